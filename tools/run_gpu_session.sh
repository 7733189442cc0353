timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "split3 or test_gemm" 2>&1 | tail -3
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vatex or median" 2>&1 | tail -3
timeout -k 5 200 python bench.py --no-legs --no-cpu-baseline --config vatex_care_large --batch 4096 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('large', d['value'], d['ms_per_step'], d['kernels'].get('enc_gemm'))"
