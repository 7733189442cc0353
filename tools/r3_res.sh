cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py -x -q -k "resident" 2>&1 | tail -3
for rep in 1 2; do
for B in 1 128; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 3 --no-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('B=$B', d['value'], d['ms_per_step'], d['decoder_step_us'])
except Exception as e: print('B=$B failed', e)"
done; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
