timeout -k 5 400 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "vocab_argmax or fused_beam or greedy" 2>&1 | tail -2
for v in "" prev "" prev "" prev; do
lib=""; [ -n "$v" ] && lib=care_amd/dbg/libcare_hip_$v.so
CARE_HIP_LIB=$lib timeout -k 5 200 python bench.py --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('variant [$v]', d['value'], d['ms_per_step'], 'vocab %.1f cross %.1f' % (k['step_vocab_argmax']['avg_us'], k['step_cross_attn']['avg_us']))"
done
