"""A few launches of the decode-step kernels at the bench shape, for `rocprofv3 --pmc ... -- python3 tools/pmc_target.py`."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None
rows, Lk, H, d, V = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 84, 8, 512, 10547
xb = torch.randn(rows, d, device=DEV).to(torch.bfloat16)
W = (torch.randn(V, d, device=DEV) * 0.05).to(torch.bfloat16)
parts = _lib.argmax_parts(V, rows, True)
pmax = torch.empty(rows, parts, device=DEV); pidx = torch.empty(rows, parts, device=DEV, dtype=torch.int32); psum = torch.empty(rows, parts, device=DEV)
mem = torch.randn(rows, Lk, d, device=DEV).to(torch.bfloat16)
qt = (torch.randn(rows, H, d, device=DEV) * 0.1).to(torch.bfloat16)
ct = torch.empty(rows, H, d, device=DEV, dtype=torch.bfloat16)
bias = torch.randn(H, Lk, device=DEV)
W1 = (torch.randn(2048, d, device=DEV) * 0.05).to(torch.bfloat16); b1 = torch.randn(2048, device=DEV)
hid = torch.empty(rows, 2048, device=DEV, dtype=torch.bfloat16)
for _ in range(3):
    _lib.call("care_gemm_argmax_bf16", p(xb), d, 1, p(W), p(pmax), p(pidx), p(psum), None, None, rows, V, d)
    _lib.call("care_attention_latent", p(qt), H * d, p(mem), Lk * d, d, 1, Lk, p(bias), Lk, p(ct), H * d, rows, H, d)
    _lib.call("care_gemm_bf16", p(xb), d, 1, p(W1), p(b1), p(hid), 2048, 1, None, 0, 0, 2048, rows, 2048, d, 1)
torch.cuda.synchronize()
