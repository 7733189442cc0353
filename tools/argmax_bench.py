"""Time the fused vocabulary-argmax GEMM (care_gemm_argmax_bf16) alone, hipGraph-timed (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"
for M in [int(a) for a in sys.argv[1:]] or [32768, 16384, 4096]:
    N, K = 10547, 512
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    parts = _lib.argmax_parts(N, M, True)
    pm, ps = torch.empty(M, parts, device=DEV), torch.empty(M, parts, device=DEV)
    pi = torch.empty(M, parts, device=DEV, dtype=torch.int32)
    p = lambda t: t.data_ptr()
    t = time_call(lambda: _lib.call("care_gemm_argmax_bf16", p(A), K, 1, p(W), p(pm), p(pi), p(ps), None, None, M, N, K), iters=10)
    print("lib=%s argmax M=%6d parts=%d: %7.1f us (%6.1f TF = %4.1f%% of 2.5 PF)" % (
        os.path.basename(os.environ.get("CARE_HIP_LIB", "default")), M, parts, t, 2.0 * M * N * K / t / 1e6, 2.0 * M * N * K / t / 1e6 / 25), flush=True)
