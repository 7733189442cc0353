mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_kernels.py -q -k "gemm_ln" > gpurun_out/r2a/t_ln.log 2>&1; echo "ln tests rc=$?"
for v in 0 1 2 3; do CARE_LN_V2=$v timeout 300 python tools/gemm_bench.py ln > gpurun_out/r2a/ln_v$v.log 2>&1; done
CARE_LN_RG=2 CARE_LN_V2=1 timeout 300 python tools/gemm_bench.py ln > gpurun_out/r2a/ln_v1_forced.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q --maxfail=12 > gpurun_out/r2a/t_par.log 2>&1; echo "parity rc=$?"
timeout 600 python bench.py > gpurun_out/r2a/bench.log 2>&1; echo "bench rc=$?"
tail -3 gpurun_out/r2a/t_ln.log; tail -5 gpurun_out/r2a/t_par.log; cat gpurun_out/r2a/ln_v1.log
