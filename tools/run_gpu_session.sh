timeout -k 5 400 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm_store or gemm_bf16 or test_gemm" 2>&1 | tail -2
timeout -k 5 600 python -m pytest tests/test_gpu_properties.py -q -x -m gpu 2>&1 | tail -2
