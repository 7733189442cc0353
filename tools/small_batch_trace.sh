#!/bin/bash
# kernel trace of the small-batch decode (graph replay): per-kernel durations at B = 128 / 1 greedy and 128 x beam 5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sb
run() {  # name, bench args
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb/$name -o sb -- python3 $R/bench.py "$@" --steps 20 --warmup 3 --no-legs --no-cpu-baseline > $R/gpurun_out/sb/$name.log 2>&1
  f=$(find $R/gpurun_out/sb/$name -name "sb_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/sb/${name}_kernel_stats.csv
  rm -rf $R/gpurun_out/sb/$name
  tail -1 $R/gpurun_out/sb/$name.log | cut -c1-300
}
run greedy_B128 --batch 128
run greedy_B1 --batch 1
run beam5_B128 --batch 128 --beam 5 --config msrvtt_care_beam5
