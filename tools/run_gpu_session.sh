timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -x -m gpu -k "beam" 2>&1 | tail -3
for B in 128 4096; do
timeout 600 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care_beam5 --beam 5 --batch $B --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$B', d['value'], d['ms_per_step'])"
done
