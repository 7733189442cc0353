timeout -k 5 600 python tools/beam_select_stress.py 10 2>&1 | tail -11
