cd $GRAFT_REPO_ROOT
for B in 32768 16384; do
for spec in "1 1" "1 0" "2 0"; do
  set -- $spec
  CARE_EARLY_EXIT=$2 python bench.py --batch $B --lanes $1 --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$B lanes=$1 early_exit=$2', d['value'], d['ms_per_step'])"
done; done
