timeout 1200 python -m pytest tests/test_gpu_properties.py -q -x -m gpu -k "beam_operating" -s 2>&1 | tail -8
