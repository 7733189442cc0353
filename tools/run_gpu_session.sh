mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fused_beam or vocab_argmax or beam" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -m gpu -k "beam" 2>&1 | tail -4
for rep in 1 2; do
timeout 600 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care_beam5 --beam 5 --batch 4096 > gpurun_out/r2z/beam.log 2>&1; tail -1 gpurun_out/r2z/beam.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('beam5', d['value'], d['ms_per_step'], ' '.join('%s %.1f' % (t.replace('step_',''), k[t]['avg_us']) for t in k))"
done
