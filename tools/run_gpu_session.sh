for ee in 1 0 1 0; do
CARE_EARLY_EXIT=$ee timeout -k 5 200 python bench.py --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('early_exit=$ee', d['value'], d['ms_per_step'], d['kernels']['step_cross_attn']['avg_us'])"
done
