python tools/split_debug.py 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm_ln_split" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "care or cabase" 2>&1 | tail -3
grep -h "care" gpurun_out/bf16_err.jsonl | tail -3
timeout 600 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care --batch 16384 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('care', d['value'], d['ms_per_step'], d['kernels']['enc_gemm'])"
