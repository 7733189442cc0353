#!/bin/bash
# kernel trace of the small-batch decode (graph replay): per-kernel durations at B = 128 and B = 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 128 1; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sb_trace_B$B -o sb -- python3 $R/bench.py --batch $B --steps 20 --warmup 3 --no-legs --no-cpu-baseline > $R/gpurun_out/sb_trace_B$B.log 2>&1
  f=$(ls $R/gpurun_out/sb_trace_B$B/*/sb_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -40 $f > $R/gpurun_out/sb_stats_B$B.csv
  tail -1 $R/gpurun_out/sb_trace_B$B.log | cut -c1-400
done
