mkdir -p gpurun_out/r2p
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2p/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $R/gpurun_out/r2p/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2p/fetch -- python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-legs > $R/gpurun_out/r2p/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2p/write -- python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-legs > $R/gpurun_out/r2p/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/r2p/sq -- python3 $R/tools/pmc_target.py 32768 > $R/gpurun_out/r2p/sq.log 2>&1
cd $R
find gpurun_out/r2p/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2p/kernel_stats.csv
python tools/pmc_parse.py gpurun_out/r2p/fetch > gpurun_out/r2p/fetch_summary.txt
python tools/pmc_parse.py gpurun_out/r2p/write > gpurun_out/r2p/write_summary.txt
python tools/pmc_parse.py gpurun_out/r2p/sq > gpurun_out/r2p/sq_summary.txt
tail -1 gpurun_out/r2p/stats.log | cut -c1-300
head -30 gpurun_out/r2p/kernel_stats.csv
rm -rf gpurun_out/r2p/stats gpurun_out/r2p/fetch gpurun_out/r2p/write gpurun_out/r2p/sq
timeout 900 python bench.py > gpurun_out/r2p/bench_full.log 2>&1; tail -1 gpurun_out/r2p/bench_full.log | cut -c1-400
