"""A few launches of the hot kernels at the bench shapes, for `rocprofv3 --pmc ... -- python3 tools/pmc_target.py [rows]`:
vocabulary arg-max, absorbed cross-attention (d = 512 and the two-wave d = 1024 form), the 256-row store GEMMs (QKV, FFN1),
fused dense / FFN2 + residual + LayerNorm, the LDS-tiled GEMM (256 x 256 tiles) and the sequence attention."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None
rows, Lk, H, d, V = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 84, 8, 512, 10547
bf = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).to(torch.bfloat16)
xb = bf(rows, d)
W = bf(V, d, sc=0.05)
parts = _lib.argmax_parts(V, rows, True)
pmax = torch.empty(rows, parts, device=DEV); pidx = torch.empty(rows, parts, device=DEV, dtype=torch.int32); psum = torch.empty(rows, parts, device=DEV)
mem = bf(rows, Lk, d)
qt = bf(rows, H, d, sc=0.1)
ct = torch.empty(rows, H, d, device=DEV, dtype=torch.bfloat16)
bias = torch.randn(H, Lk, device=DEV)
W1 = bf(2048, d, sc=0.05); b1 = torch.randn(2048, device=DEV)
hid = torch.empty(rows, 2048, device=DEV, dtype=torch.bfloat16)
Wqkv = bf(3 * d, d, sc=0.05); bqkv = torch.randn(3 * d, device=DEV)
q = torch.empty(rows, d, device=DEV); cache = torch.empty(rows, 2 * d, device=DEV, dtype=torch.bfloat16)
# fused dense / FFN2 + residual + LayerNorm (packed weights)
Wo = bf(d, d, sc=0.05); W2 = bf(d, 2048, sc=0.03)
Wop, W2p = torch.empty_like(Wo), torch.empty_like(W2)
_lib.call("care_pack_ln_weight", p(Wo), p(Wop), d, d)
_lib.call("care_pack_ln_weight", p(W2), p(W2p), d, 2048)
bo, g, be = torch.randn(d, device=DEV), torch.randn(d, device=DEV), torch.randn(d, device=DEV)
res = torch.randn(rows, d, device=DEV); out = torch.empty(rows, d, device=DEV); outb = torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)
# d_model = 1024 kernels at 4096 rows
r2, d2, H2, Lk2 = 4096, 1024, 16, 114
A2 = bf(r2, d2); Wf = bf(4096, d2, sc=0.03); bff = torch.randn(4096, device=DEV); o2 = torch.empty(r2, 4096, device=DEV, dtype=torch.bfloat16)
mem2 = bf(r2, Lk2, d2); qt2 = bf(r2, H2, d2, sc=0.08); ct2 = torch.empty(r2, H2, d2, device=DEV, dtype=torch.bfloat16); bias2 = torch.randn(H2, Lk2, device=DEV)
# sequence attention (teacher forcing), 4096 sequences x 29 queries over 84 keys
ns, t = 4096, 29
qs = bf(ns * t, d); kv = bf(ns * Lk, 2 * d); cs = torch.empty(ns * t, d, device=DEV, dtype=torch.bfloat16)
for _ in range(3):
    _lib.call("care_gemm_argmax_bf16", p(xb), d, 1, p(W), p(pmax), p(pidx), p(psum), None, None, rows, V, d)
    _lib.call("care_attention_latent", p(qt), H * d, p(mem), Lk * d, d, 1, Lk, p(bias), Lk, p(ct), H * d, rows, H, d)
    _lib.call("care_gemm_bf16", p(xb), d, 1, p(W1), p(b1), p(hid), 2048, 1, None, 0, 0, 2048, rows, 2048, d, 1)
    _lib.call("care_gemm_bf16", p(xb), d, 1, p(Wqkv), p(bqkv), p(q), d, 0, p(cache), 2 * d, 1, d, rows, 3 * d, d, 0)
    _lib.call("care_gemm_ln_packed", p(xb), d, 1, p(Wop), p(bo), p(res), d, p(g), p(be), 1e-12, p(out), p(outb), d, rows, d, d, rows, rows, 0)
    _lib.call("care_gemm_ln_packed", p(hid), 2048, 1, p(W2p), p(bo), p(res), d, p(g), p(be), 1e-12, None, p(outb), d, rows, d, 2048, rows, rows, 0)
    _lib.call("care_gemm_tile", p(A2), d2, p(Wf), p(bff), p(o2), 4096, 1, None, 0, 0, 4096, r2, 4096, d2, 1)
    _lib.call("care_attention_latent", p(qt2), H2 * d2, p(mem2), Lk2 * d2, d2, 1, Lk2, p(bias2), Lk2, p(ct2), H2 * d2, r2, H2, d2)
    _lib.call("care_attention_seq", p(qs), d, p(kv), p(kv[:, d:]), Lk * 2 * d, 2 * d, 1, Lk, 0, t, None, 0, 0, p(bias), Lk, p(cs), d, ns, H)
torch.cuda.synchronize()
