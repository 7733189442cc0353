# full GPU suite + full bench + profiles of the bench default (run on the GPU box from the repo root)
R=$PWD; O=$R/gpurun_out/r2g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py > $O/bench_full.log 2>&1; tail -c 300 $O/bench_full.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-legs > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-legs > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_parse.py $O/pmc_fetch > $O/pmc_fetch.txt 2>&1; python tools/pmc_parse.py $O/pmc_write > $O/pmc_write.txt 2>&1
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -6 $O/kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
