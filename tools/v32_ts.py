"""In-kernel timeline of the vocabulary arg-max kernel (tools build -DCARE_V32_DBG=64: workgroup 0 stamps s_memtime per wave, tile).
    python tools/variant_lib.py gemm_vocab.hip tools/lib/v32_ts.so -DCARE_V32_DBG=64
    CARE_HIP_LIB=tools/lib/v32_ts.so python tools/v32_ts.py [rows]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib

DEV = "cuda:0"
rows, d, V = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 512, 10547
p = lambda t: t.data_ptr()
xb = (torch.randn(rows, d, device=DEV)).to(torch.bfloat16)
W = (torch.randn(V, d, device=DEV) * 0.05).to(torch.bfloat16)
parts = _lib.argmax_parts(V, rows, True)
pmax = torch.empty(rows, parts, device=DEV); pidx = torch.empty(rows, parts, device=DEV, dtype=torch.int32); psum = torch.empty(rows, parts, device=DEV)
for _ in range(3):
    _lib.call("care_gemm_argmax_bf16", p(xb), d, 1, p(W), p(pmax), p(pidx), p(psum), None, None, rows, V, d)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["CARE_HIP_LIB"])
buf = np.zeros(8 * 64 * 4, dtype=np.uint64)
assert lib.care_v32_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 64, 4).astype(np.int64)
t0 = t[:, 1, 0].min()
print("workgroup 0, %d rows; per tile: top, landed (vmcnt), past the barrier, MFMAs issued - waves 0 (multiplies first), 4 (statistics first), 7" % rows)
for it in range(1, 24):
    print("it=%2d " % it + " | ".join("w%d %7d %7d %7d %7d" % ((w,) + tuple(int(x - t0) for x in t[w, it])) for w in (0, 4, 7)))
dd = np.diff(t[0, 1:60, 0])
print("wave 0: ticks per tile: median %d, mean %d" % (np.median(dd), dd.mean()))
