mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fused_beam" 2>&1 | tail -2
python tools/beam_select_stress.py 8 2>&1 | tail -1
for i in 1 2; do timeout 600 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care_beam5 --beam 5 --batch 4096 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('beam5 B4096', d['value'], d['ms_per_step'], 'stats %.1f collect %.1f' % (k['beam_vocab_stats']['avg_us'], k['beam_vocab_collect']['avg_us']))"; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -x -m gpu -k "beam" 2>&1 | tail -2
