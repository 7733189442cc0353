"""care_gemm (fp32 weights) on the concept head's small-batch shapes: one wave per 16 x 16 tile against the LDS-tiled kernel
(CARE_GEMM_FEW_TILES=0), us per launch (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"
p = lambda t: t.data_ptr()
for M, N, K in [(1, 500, 2048), (128, 500, 2048), (128, 500, 1536), (1, 512, 512), (128, 512, 512), (256, 500, 2048), (16, 1536, 512)]:
    A, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) * 0.02, torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    line = "M=%4d N=%5d K=%4d " % (M, N, K)
    for few in ("1", "0"):
        os.environ["CARE_GEMM_FEW_TILES"] = few
        t = time_call(lambda: _lib.call("care_gemm", p(A), K, p(W), 0, p(b), p(out), N, 0, None, 0, 0, N, M, N, K, 0))
        line += " %s %6.1f us |" % ("wave per tile" if few == "1" else "LDS-tiled", t)
    print(line, flush=True)
