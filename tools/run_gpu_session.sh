mkdir -p gpurun_out/r2z
for rep in 1 2; do for sp in 1 0; do
CARE_BEAM_SPARSE=$sp timeout 600 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care_beam5 --beam 5 --batch 4096 > gpurun_out/r2z/beam.log 2>&1; tail -1 gpurun_out/r2z/beam.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('sparse=$sp', d['value'], d['ms_per_step'], ' '.join('%s %.1f' % (t.replace('step_',''), k[t]['avg_us']) for t in list(k)[:4]))"
done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -x -m gpu -k "beam" 2>&1 | tail -2
python tools/beam_select_stress.py 8 2>&1 | tail -2
