#!/bin/bash
# rocprofv3 evidence of the round-3 build (run on the GPU box through gpurun; summaries are made by tools/r3_profiles_post.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03prof; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_vatex -o t -- $B --steps 5 --warmup 2 --config vatex_care_large --batch 4096 > $O/trace_vatex.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- $B --steps 1 --warmup 2 --no-graph > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- $B --steps 1 --warmup 2 --no-graph > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq -o t -- python3 $R/tools/pmc_target.py 32768 > $O/sq.log 2>&1
# keep what the post-processor reads (the raw traces are large)
for d in trace trace_vatex; do f=$(find $O/$d -name "t_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
for d in fetch write sq; do f=$(find $O/$d -name "t_counter_collection.csv" | head -1); [ -n "$f" ] && python3 - "$f" > $O/${d}_counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    for c, v in sorted(cs.items()):
        print("%s\t%s\t%.1f\t%d" % (k, c, sum(v) / len(v), len(v)))
PY
done
rm -rf $O/trace $O/trace_vatex $O/fetch $O/write $O/sq
tail -1 $O/trace.log | cut -c1-200
ls -la $O
