cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "greedy_bf16 and resident" 2>&1 | tail -3
timeout 300 python tools/resident_prof.py 1 128 2>&1 | grep -v amdgpu.ids
for B in 1 16 128 256; do
  CARE_RESIDENT_MAX_ROWS=256 timeout 300 python bench.py --batch $B --steps 20 --warmup 3 --no-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('B=$B', d['value'], d['ms_per_step'], d['decoder_step_us'])
except Exception as e: print('B=$B failed', e)"
done
