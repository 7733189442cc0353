mkdir -p gpurun_out/r2h
timeout 1500 python bench.py > gpurun_out/r2h/bench_full2.log 2>&1; tail -1 gpurun_out/r2h/bench_full2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
for k,v in d.get('legs',{}).items(): print(' ', k, {a:b for a,b in v.items() if a in ('captions_per_s','ms_per_pass','decoder_step_us','speedup_vs_fixed_29')})"
