"""In-kernel timeline of the LDS-tiled GEMM (tools build -DCARE_TILE_DBG=64: workgroup 0 stamps s_memtime per wave and K step).
    python tools/variant_lib.py gemm_tile.hip tools/lib/tile_ts.so -DCARE_TILE_DBG=64
    CARE_HIP_LIB=tools/lib/tile_ts.so python tools/tile_ts.py [M N K]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib

DEV = "cuda:0"
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 4096, 4096)
p = lambda t: t.data_ptr()
A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device=DEV)
out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
os.environ.setdefault("CARE_TILE_CFG", "4412")
for _ in range(3):
    _lib.call("care_gemm_tile", p(A), K, p(W), p(bias), p(out), N, 1, None, 0, 0, N, M, N, K, 0)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["CARE_HIP_LIB"])
buf = np.zeros(16 * 64 * 4, dtype=np.uint64)
assert lib.care_tile_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(16, 64, 4).astype(np.int64)
t0 = t[:, 0, 0].min()
nk = K // 64
print("workgroup 0, %d x %d x %d, %d K steps; ticks of s_memtime relative to the first stamp" % (M, N, K, nk))
print("per step: top, landed (vmcnt), past the barrier, DMAs issued (cfg 4415: MFMAs issued)  - waves 0, 7, 15")
for kt in range(min(nk, 24)):
    print("kt=%2d " % kt + " | ".join("w%-2d %7d %7d %7d %7d" % ((w,) + tuple(int(x - t0) for x in t[w, kt])) for w in (0, 7, 15)))
if nk < 64:
    print("epilogue: " + " | ".join("w%-2d %7d -> %7d" % (w, t[w, nk, 0] - t0, t[w, nk, 1] - t0) for w in (0, 7, 15)))
d = np.diff(t[0, : min(nk, 64), 0])
print("wave 0: ticks per K step: median %d, mean %d" % (np.median(d), d.mean()))
