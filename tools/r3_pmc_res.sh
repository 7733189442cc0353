#!/bin/bash
# L2 counters of the resident decode kernel (one pass each, counters in a run of their own)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcres
for B in 1 128; do
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcres/B${B}_$tag -o p -- python3 $R/bench.py --batch $B --steps 2 --warmup 2 --no-legs --no-cpu-baseline --no-graph > $R/gpurun_out/pmcres/B${B}_$tag.log 2>&1
  f=$(find $R/gpurun_out/pmcres/B${B}_$tag -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "B=$B" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "decode_resident" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
for c, d in acc.items():
    v = d["v"]
    print(sys.argv[2], c, "launches", len(v), "avg per launch %.4g" % (sum(v) / len(v)))
PY
  rm -rf $R/gpurun_out/pmcres/B${B}_$tag
done; done
