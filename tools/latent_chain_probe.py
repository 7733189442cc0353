"""expand -> absorbed attention -> reduce against the projected-K/V formulation (torch fp32), plus timings."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    Lk = int(sys.argv[2]) if len(sys.argv) > 2 else 84
    H, d, dh = 8, 512, 64
    torch.manual_seed(1)
    mem = torch.randn(rows, Lk, d, device=DEV).to(torch.bfloat16)
    q = (torch.randn(rows, d, device=DEV)).to(torch.bfloat16)
    Wk = (torch.randn(d, d, device=DEV) * 0.04).to(torch.bfloat16)
    Wv = (torch.randn(d, d, device=DEV) * 0.04).to(torch.bfloat16)
    bk = torch.randn(d, device=DEV) * 0.1
    bv = torch.randn(d, device=DEV) * 0.1
    bias = torch.randn(H, Lk, device=DEV) * 0.5
    # wkt[h][c][e] = Wk[h*64+e][c] / 8
    wkt = (Wk.float().view(H, dh, d).permute(0, 2, 1) / 8.0).contiguous().to(torch.bfloat16)
    qt = torch.empty(rows, H, d, device=DEV, dtype=torch.bfloat16)
    ct = torch.empty(rows, H, d, device=DEV, dtype=torch.bfloat16)
    ctx = torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)

    def expand():
        _lib.call("care_head_expand", p(q), d, p(wkt), p(qt), H * d, rows, H)

    def latent():
        _lib.call("care_attention_latent", p(qt), H * d, p(mem), Lk * d, d, 1, Lk, p(bias), Lk, p(ct), H * d, rows, H, d)

    def reduce():
        _lib.call("care_head_reduce", p(ct), H * d, p(Wv), p(bv), p(ctx), d, rows, H)

    expand(); latent(); reduce()
    torch.cuda.synchronize()
    n = min(rows, 300)
    sl = slice(rows - n, rows)  # the tail rows (ragged tiles)
    qf, mf = q[sl].float(), mem[sl].float()
    qt_ref = torch.einsum("rhe,hce->rhc", qf.view(n, H, dh), wkt.float())
    print("expand  max err %.4g (ref max %.3g)" % ((qt[sl].float() - qt_ref).abs().max().item(), qt_ref.abs().max().item()))
    K = (mf @ Wk.float().t() + bk).view(n, Lk, H, dh)
    V = (mf @ Wv.float().t() + bv).view(n, Lk, H, dh)
    s = torch.einsum("rhe,rjhe->rhj", qf.view(n, H, dh), K) / 8.0 + bias[None]
    pr = torch.softmax(s, -1)
    ref = torch.einsum("rhj,rjhe->rhe", pr, V).reshape(n, d)
    err = (ctx[sl].float() - ref).abs()
    print("chain   max err %.4g mean %.4g (ref max %.3g)" % (err.max().item(), err.mean().item(), ref.abs().max().item()))
    ct_ref = torch.einsum("rhj,rjc->rhc", pr, mf)
    ctx_ref2 = torch.einsum("rhc,hec->rhe", ct[sl].float(), Wv.float().view(H, dh, d)).reshape(n, d) + bv
    print("reduce  max err %.4g" % (ctx[sl].float() - ctx_ref2).abs().max().item())
    te, tl, tr = time_call(expand, 10), time_call(latent, 10), time_call(reduce, 10)
    print("rows=%d Lk=%d: expand %.1f us (%.2f TB/s), latent %.1f us, reduce %.1f us (%.2f TB/s) -> %.1f us" %
          (rows, Lk, te, rows * (d * 2 + H * d * 2) / te / 1e6, tl, tr, rows * (H * d * 2 + d * 2) / tr / 1e6, te + tl + tr))


if __name__ == "__main__":
    main()
