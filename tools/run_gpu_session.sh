mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_kernels.py -q -k "gemm_ln" > gpurun_out/r2e/t_ln.log 2>&1; echo "ln tests rc=$?"; tail -3 gpurun_out/r2e/t_ln.log
timeout 300 python tools/gemm_bench.py ln > gpurun_out/r2e/ln_packed.log 2>&1
LN_LEAN=1 timeout 300 python tools/gemm_bench.py ln > gpurun_out/r2e/ln_packed_lean.log 2>&1
for d in 6 15; do CARE_HIP_LIB=care_amd/dbg/libcare_hip_dbg$d.so timeout 300 python tools/gemm_bench.py ln > gpurun_out/r2e/ln_dbg$d.log 2>&1; done
for f in ln_packed ln_packed_lean ln_dbg6 ln_dbg15; do echo "== $f"; grep gemm_ln gpurun_out/r2e/$f.log | head -14; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q --maxfail=12 > gpurun_out/r2e/t_par.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r2e/t_par.log
timeout 600 python bench.py > gpurun_out/r2e/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2e/bench.log | cut -c1-200
