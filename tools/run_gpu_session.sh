mkdir -p gpurun_out/r2f
timeout 1200 python -m pytest tests/test_gpu_properties.py -q --maxfail=8 -k "early_exit or active_slots" > gpurun_out/r2f/t_ee.log 2>&1; echo "ee rc=$?"; tail -30 gpurun_out/r2f/t_ee.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q --maxfail=12 > gpurun_out/r2f/t_par.log 2>&1; echo "parity rc=$?"; tail -8 gpurun_out/r2f/t_par.log
timeout 600 python bench.py > gpurun_out/r2f/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2f/bench.log | cut -c1-200
