timeout -k 5 600 python -m pytest tests/test_gpu_properties.py -q -x -m gpu -k "operating_point" -s 2>&1 | tail -6
