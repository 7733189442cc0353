cd $GRAFT_REPO_ROOT
for S in 2 4 8 15 29; do
  CARE_SEGMENT_STEPS=$S python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S=$S', d['value'], d['ms_per_step'])"
done
