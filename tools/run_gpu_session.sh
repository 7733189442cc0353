mkdir -p gpurun_out/r2y
timeout 1500 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -3
run() { timeout 600 python bench.py --no-legs --no-cpu-baseline > gpurun_out/r2y/$1.log 2>&1; tail -1 gpurun_out/r2y/$1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['value'], d['ms_per_step'], ' '.join('%s %.1f' % (t.replace('step_',''), k[t]['avg_us']) for t in k))"; }
run a
run b
