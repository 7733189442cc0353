cd $GRAFT_REPO_ROOT
for c in 0 213 214 413; do
  CARE_TILE_CFG64=$c timeout 300 python bench.py --config vatex_care_large --batch 4096 --no-legs --no-cpu-baseline --steps 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('cfg64=$c', d['value'], d['ms_per_step'], 'reduce', k['step_head_reduce']['avg_us'], 'expand', k['step_head_expand']['avg_us'], 'attn', k['step_cross_attn']['avg_us'])"
done
CARE_LAT2_SLOTS=3 timeout 300 python bench.py --config vatex_care_large --batch 4096 --no-legs --no-cpu-baseline --steps 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('slots3', d['value'], d['ms_per_step'], 'attn', k['step_cross_attn']['avg_us'])"
