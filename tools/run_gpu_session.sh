mkdir -p gpurun_out/r2u
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r2u/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -6 gpurun_out/r2u/t_all.log
