mkdir -p gpurun_out/r2w
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm_ln" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "care or cabase" 2>&1 | tail -8
timeout 600 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care --batch 16384 > gpurun_out/r2w/care.log 2>&1; tail -1 gpurun_out/r2w/care.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('care', d['value'], d['ms_per_step']); 
for k,v in d['kernels'].items(): print(k, v)"
