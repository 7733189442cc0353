"""Absorbed cross-attention kernel: numerics against torch and time against the K/V kernel (GPU box)."""
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
    H, d = 8, 512
    torch.manual_seed(0)
    mem = (torch.randn(rows, Lk, d, device=DEV)).to(torch.bfloat16)
    qt = (torch.randn(rows, H, d, device=DEV) * 0.1).to(torch.bfloat16)
    bias = torch.randn(H, Lk, device=DEV) * 0.5
    ct = torch.zeros(rows, H, d, device=DEV, dtype=torch.bfloat16)
    for use_bias in (True, False):
        b = bias if use_bias else None
        _lib.call("care_attention_latent", p(qt), H * d, p(mem), Lk * d, d, 1, Lk, p(b), Lk, p(ct), H * d, rows, H, d)
        torch.cuda.synchronize()
        n = min(rows, 256)
        s = torch.einsum("rhc,rjc->rhj", qt[:n].float(), mem[:n].float())
        if use_bias:
            s = s + bias[None]
        pr = torch.softmax(s, -1)
        ref = torch.einsum("rhj,rjc->rhc", pr, mem[:n].float())
        err = (ct[:n].float() - ref).abs().max().item()
        print("bias=%s rows=%d Lk=%d  max|ct - ref| = %.4g  (ref max %.3g)" % (use_bias, rows, Lk, err, ref.abs().max().item()))
    t_lat = time_call(lambda: _lib.call("care_attention_latent", p(qt), H * d, p(mem), Lk * d, d, 1, Lk, p(bias), Lk,
                                        p(ct), H * d, rows, H, d), iters=10)
    byt = rows * (Lk * d * 2 + 2 * H * d * 2)
    print("latent: %.1f us  %.2f TB/s of %.1f MB" % (t_lat, byt / t_lat / 1e6, byt / 1e6))
    q = torch.randn(rows, d, device=DEV)
    kv = torch.randn(rows * Lk, 2 * d, device=DEV).to(torch.bfloat16)
    ctx = torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)
    t_kv = time_call(lambda: _lib.call("care_attention", p(q), d, p(kv), p(kv[:, d:]), 1, Lk * 2 * d, 2 * d, 1, None, 0,
                                       Lk, 0, 1, 0, None, 0, 0, p(bias), Lk, p(ctx), d, 1, rows, H), iters=10)
    byt2 = rows * (2 * Lk * d * 2 + d * 4 + d * 2)
    print("K/V kernel: %.1f us  %.2f TB/s of %.1f MB" % (t_kv, byt2 / t_kv / 1e6, byt2 / 1e6))


if __name__ == "__main__":
    main()
