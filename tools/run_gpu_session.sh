mkdir -p gpurun_out/r2h
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r2h/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -25 gpurun_out/r2h/t_all.log
