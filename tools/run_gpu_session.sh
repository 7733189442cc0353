timeout 1500 python tools/beam_select_stress.py 16 2>&1 | tail -18
