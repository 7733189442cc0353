"""In-kernel timeline of the wave-specialised embedder kernel (tools build -DCARE_LN3_DBG=64: workgroup 0 stamps its first
256 K steps with s_memtime).   CARE_HIP_LIB=tools/lib/ln3_ts.so python tools/emb_ts.py [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib

DEV = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = 32768 * 28
p = lambda t: t.data_ptr()
A = torch.randn(M, K, device=DEV)
W = (torch.randn(512, K, device=DEV) * 0.05).to(torch.bfloat16)
bias, g, b = (torch.randn(512, device=DEV) for _ in range(3))
outb = torch.empty(M, 512, device=DEV, dtype=torch.bfloat16)
Wp = torch.empty_like(W)
_lib.call("care_pack_ln_weight", p(W), p(Wp), 512, K)
ts = torch.zeros(12 * 256 * 4, device=DEV, dtype=torch.int64)
for _ in range(3):
    _lib.call("care_gemm_ln_packed", p(A), K, 0, p(Wp), p(bias), None, 512, p(g), p(b), 1e-12, p(ts), p(outb), 512, M, 512, K, M, M, 0)
torch.cuda.synchronize()
t = ts.cpu().view(12, 256, 4).numpy().astype("int64")
t0 = t[0, 0, 0]
nk = K // 32
print("shader-clock ticks (s_memtime) relative to wave 0's first step; columns per step g:")
print("  compute wave 0: start, MFMAs issued, [epilogue done]   | W loader 8: top, issued, landed(g+1), past B_g | A loader 10: top, stored, loads issued")
for gi in list(range(0, min(256, 3 * nk))):
    c, w, a = t[0, gi], t[8, gi], t[10, gi]
    f = lambda x: "%8d" % (x - t0) if x else "       -"
    print("g=%3d  C %s %s %s | W %s %s %s %s | A %s %s %s" % (gi, f(c[0]), f(c[1]), f(c[2]), f(w[0]), f(w[1]), f(w[2]), f(w[3]), f(a[0]), f(a[1]), f(a[2])))
