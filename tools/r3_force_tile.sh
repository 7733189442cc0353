cd $GRAFT_REPO_ROOT
for ft in 0 1; do
  for spec in "128 5" "512 5" "128 1" "1024 1" "2048 1" "4096 1"; do
    set -- $spec
    if [ "$2" = "5" ]; then A="--beam 5 --config msrvtt_care_beam5"; else A=""; fi
    CARE_FORCE_TILE=$ft python bench.py --batch $1 $A --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('force_tile=$ft B=$1 beam=$2', d['value'], d['ms_per_step'])"
  done
done
