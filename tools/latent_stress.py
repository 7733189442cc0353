"""Randomised stress of care_attention_latent (GPU box): shapes, pairing, bias; against torch, and run-to-run bit equality."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = torch.Generator().manual_seed(1234)
    worst = 0.0
    for it in range(iters):
        H = [8, 8, 8, 4, 16][int(torch.randint(0, 5, (1,), generator=g))]
        rpk = [1, 1, 2, 3, 5][int(torch.randint(0, 5, (1,), generator=g))]
        clips = int(torch.randint(1, 1500, (1,), generator=g))
        rows = clips * rpk if torch.rand(1, generator=g) < 0.8 else clips * rpk + int(torch.randint(1, max(2, rpk), (1,), generator=g))
        nkeys = int(torch.randint(1, 129, (1,), generator=g))
        use_bias = bool(torch.rand(1, generator=g) < 0.5)
        d = 512
        nclips = (rows + rpk - 1) // rpk
        torch.manual_seed(it)
        mem = torch.randn(nclips, nkeys, d, device=DEV).to(torch.bfloat16)
        qt = (torch.randn(rows, H, d, device=DEV) * 0.12).to(torch.bfloat16)
        bias = torch.randn(H, nkeys, device=DEV) * 0.7 if use_bias else None
        outs = []
        for rep in range(2):
            ct = torch.full((rows, H, d), float("nan"), device=DEV, dtype=torch.bfloat16)
            _lib.call("care_attention_latent", p(qt), H * d, p(mem), nkeys * d, d, rpk, nkeys, p(bias), nkeys, p(ct), H * d, rows, H, d)
            outs.append(ct)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), ("run-to-run difference", it, H, rpk, rows, nkeys)
        ct = outs[0]
        n = min(rows, 400)
        sel = torch.randperm(rows, generator=g)[:n].to(DEV)
        clip_of = sel // rpk
        m = mem.float()[clip_of]
        s = torch.einsum("rhc,rjc->rhj", qt.float()[sel], m)
        if use_bias:
            s = s + bias[None]
        ref = torch.einsum("rhj,rjc->rhc", torch.softmax(s, -1), m)
        assert torch.isfinite(ct.float()).all(), ("non-finite", it, H, rpk, rows, nkeys)
        err = (ct.float()[sel] - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        worst = max(worst, err)
        assert err < 1.2e-2, (err, it, H, rpk, rows, nkeys, use_bias)
    print("latent stress: %d cases ok, worst relative error %.3g" % (iters, worst))


if __name__ == "__main__":
    main()
