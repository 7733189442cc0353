import os, sys
sys.path.insert(0, ".")
import torch
from care_amd import _lib
from tools.gemm_bench import time_call
p = lambda t: t.data_ptr()
dev = "cuda"
for M in (32768, 16384, 8192):
    for (N, K, act, split, out_bf) in [(1536, 512, 0, 512, None), (2048, 512, 1, None, True), (512, 512, 0, None, False), (1024, 512, 0, None, True)]:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        if split:  # q fp32 | kv bf16
            c0 = torch.empty(M, split, device=dev); c1 = torch.empty(M, N - split, device=dev, dtype=torch.bfloat16)
            tail = (p(bias), p(c0), split, 0, p(c1), N - split, 1, split, M, N, K, act)
        else:
            c0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if out_bf else torch.float32)
            tail = (p(bias), p(c0), N, 1 if out_bf else 0, None, 0, 0, N, M, N, K, act)
        t0 = time_call(lambda: _lib.call("care_gemm_bf16", p(A), K, 1, p(W), *tail))
        line = "M=%5d N=%4d K=%3d act=%d %s: A-stationary %6.1f us |" % (M, N, K, act, "split" if split else ("bf16" if out_bf else "fp32"), t0)
        for cfg in ("4412", "222"):
            os.environ["CARE_TILE_CFG"] = cfg
            t = time_call(lambda: _lib.call("care_gemm_tile", p(A), K, p(W), *tail))
            line += " tile %s %6.1f |" % (cfg, t)
        os.environ.pop("CARE_TILE_CFG", None)
        print(line, flush=True)
