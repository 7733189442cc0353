R=$PWD; O=$R/gpurun_out/r2m; mkdir -p $O
timeout -k 5 900 python -m pytest tests -q -m gpu 2>&1 | tail -2 > $O/tests.log; cat $O/tests.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 5 600 python bench.py > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['kernels'].items(): print('  %-20s %4d x %8.1f' % (k, v['launches'], v['avg_us']))
for k,v in d.get('legs',{}).items(): print(' ', k, {a:b for a,b in v.items() if a in ('captions_per_s','ms_per_pass','speedup_vs_fixed_29')})"
cd /tmp; export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/prof_stats.log 2>&1
cd $R
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -3 $O/kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
