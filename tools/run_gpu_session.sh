python tools/beam_eos_probe.py 5 8 12 20 2>&1 | tail -6
