#!/bin/bash
# rocprofv3 evidence of the current build (run on the GPU box through gpurun; tools/profiles_post.py turns the output into
# profiles/<round>_* and regenerates the "Readings" of profiles/README.md):   ROUND=r05 bash tools/profiles.sh
# Counters are collected in passes of their own (kernel-trace only beside --pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; ROUND=${ROUND:-r06}; O=$R/gpurun_out/${ROUND}prof; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-legs"
stats() {  # name, command...: kernel trace with --stats, keeps the summary csv
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name "t_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/$name
}
stats trace $B --steps 5 --warmup 2
stats trace_vatex $B --steps 5 --warmup 2 --config vatex_care_large --batch 4096
stats trace_vatex16k $B --steps 3 --warmup 2 --config vatex_care_large --batch 16384
stats greedy_B128 $B --batch 128 --steps 20 --warmup 3
stats greedy_B1 $B --batch 1 --steps 20 --warmup 3
stats beam5_B128 $B --batch 128 --beam 5 --config msrvtt_care_beam5 --steps 20 --warmup 3
stats beam5_B1 $B --batch 1 --beam 5 --config msrvtt_care_beam5 --steps 20 --warmup 3
stats train_B64 python3 $R/tools/train_prof.py 64 10
stats train_B512 python3 $R/tools/train_prof.py 512 5
export CARE_TRAIN_GEMM=f32   # (every training product in the exact-f32 form; the default "auto" above takes the split products
stats train_B512_f32 python3 $R/tools/train_prof.py 512 5   #  for these sizes.  Exported here: nothing but the program goes behind `--`)
unset CARE_TRAIN_GEMM
stats trace_fp16 $B --steps 5 --warmup 2 --dtype fp16
stats beam5_chain_B128 python3 $R/tools/chain_prof.py 128 3
stats beam5_multilaunch_B512 python3 $R/tools/beam_sweep.py --only multi-launch 512
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -o t -- "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name "t_counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/${name}_counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    for c, v in sorted(cs.items()):
        print("%s\t%s\t%.1f\t%d" % (k, c, sum(v) / len(v), len(v)))
PY
  rm -rf $O/$name
}
pmc fetch FETCH_SIZE $B --steps 1 --warmup 2 --no-graph
pmc write WRITE_SIZE $B --steps 1 --warmup 2 --no-graph
pmc sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" python3 $R/tools/pmc_target.py 32768
pmc sq_resident "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_IFETCH" python3 $R/bench.py --no-cpu-baseline --no-legs --batch 128 --beam 5 --config msrvtt_care_beam5 --steps 3 --warmup 2 --no-graph
pmc icache_resident "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" python3 $R/bench.py --no-cpu-baseline --no-legs --batch 128 --beam 5 --config msrvtt_care_beam5 --steps 3 --warmup 2 --no-graph
# the phases of the resident beam launch ONE KERNEL EACH (the chained step, csrc/decode_chain.hip): counters per phase
pmc2() {  # name, counters, command...: aggregated per kernel by tools/pmc_agg.py
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -o t -- "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name "t_counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_agg.py $f chain_ > $O/${name}_counters.txt
  rm -rf $O/$name
}
pmc2 chain_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" python3 $R/tools/chain_prof.py 128 1
pmc2 chain_sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" python3 $R/tools/chain_prof.py 128 1
cd $R
python3 tools/greedy_sweep.py 1 64 128 256 512 1024 2048 4096 > $O/greedy_sweep.txt 2>&1
python3 tools/greedy_sweep.py --config vatex_care_large 1 32 64 128 >> $O/greedy_sweep.txt 2>&1
python3 tools/resident_prof.py 1 128 > $O/resident_phase_clocks.txt 2>&1
python3 tools/resident_prof.py --beam 5 --config msrvtt_care 1 128 >> $O/resident_phase_clocks.txt 2>&1
python3 tools/beam_sweep.py 1 4 16 32 64 128 256 512 819 > $O/beam_sweep.txt 2>&1
python3 tools/beam_sweep.py --beam 8 1 16 80 >> $O/beam_sweep.txt 2>&1    # beam sizes 6 .. 8: the launch's second instance
python3 tools/beam_sweep.py --beam 6 1 16 106 >> $O/beam_sweep.txt 2>&1
tail -1 $O/trace.log | cut -c1-200
ls -la $O
