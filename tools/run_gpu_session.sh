mkdir -p gpurun_out/r2s
timeout 1200 python -m pytest tests/test_gpu_properties.py -q --maxfail=8 -k "beam" > gpurun_out/r2s/t_b.log 2>&1; echo "beam tests rc=$?"; tail -30 gpurun_out/r2s/t_b.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r2s/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -6 gpurun_out/r2s/t_all.log
