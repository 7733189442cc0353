"""In-kernel timeline of the 256-row store GEMM (tools build -DCARE_S32_DBG=64: workgroup 0 stamps s_memtime per wave and tile).
    python tools/variant_lib.py gemm_store32.hip tools/lib/s32_ts.so -DCARE_S32_DBG=64
    CARE_HIP_LIB=tools/lib/s32_ts.so python tools/s32_ts.py [rows] [N]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
K = 512
p = lambda t: t.data_ptr()
A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device=DEV)
out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
for _ in range(3):  # FFN1's call: bf16 out, ReLU
    _lib.call("care_gemm_bf16", p(A), K, 1, p(W), p(bias), p(out), N, 1, None, 0, 0, N, M, N, K, 1)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["CARE_HIP_LIB"])
buf = np.zeros(8 * 64 * 4, dtype=np.uint64)
assert lib.care_s32_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 64, 4).astype(np.int64)
t0 = t[:, 1, 0].min()
print("workgroup 0, %d x %d x %d; per tile: top, landed (vmcnt), past the barrier, MFMAs + woven epilogue issued - waves 0, 4, 7" % (M, N, K))
for it in range(1, 20):
    print("it=%2d " % it + " | ".join("w%d %7d %7d %7d %7d" % ((w,) + tuple(int(x - t0) for x in t[w, it])) for w in (0, 4, 7)))
dd = np.diff(t[0, 1:30, 0])
print("wave 0: ticks per tile: median %d, mean %d" % (np.median(dd), dd.mean()))
