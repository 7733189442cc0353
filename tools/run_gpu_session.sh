mkdir -p gpurun_out/r2y
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "greedy or embed or ln" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_properties.py -q -x -m gpu 2>&1 | tail -2
run() { timeout 600 python bench.py --no-legs --no-cpu-baseline > gpurun_out/r2y/$1.log 2>&1; tail -1 gpurun_out/r2y/$1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['value'], d['ms_per_step'], 'cross %.1f' % k['step_cross_attn']['avg_us'])"; }
CARE_UPDATE_RPW4=1 run new
CARE_UPDATE_RPW4=0 run old
CARE_UPDATE_RPW4=1 run new
CARE_UPDATE_RPW4=0 run old
