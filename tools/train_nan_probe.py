"""Which product of a training step first yields a non-finite value, and where it comes from (debug probe):
python tools/train_nan_probe.py [fixture]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import GoldenCase
from care_amd import get_framework, training
from care_amd._lib import call, ptr

name = sys.argv[1] if len(sys.argv) > 1 else "msrvtt_cabase_b3"
g = GoldenCase(name)
opt, P, feats, ids = g.build()
opt.update(encoder_dropout_prob=0.0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, hidden_act="gelu")
model = get_framework(opt)
model.load_state_dict(P, strict=True)
model.to("cuda:0").train()
orig = training._mm_x3
n = [0]


def dissect(A, Bt, a_t, b_t):
    M = A.shape[1] if a_t else A.shape[0]
    K = A.shape[0] if a_t else A.shape[1]
    N = Bt.shape[1] if b_t else Bt.shape[0]
    slabs = training._x3_slabs(M, N, K)
    ks = ((K + slabs - 1) // slabs + 63) // 64 * 64
    As, Bs = training._slab_major(A, slabs, ks, a_t), training._slab_major(Bt, slabs, ks, b_t)
    print("  M N K slabs ks", M, N, K, slabs, ks, "As", tuple(As.shape), As.is_contiguous(), "Bs", tuple(Bs.shape), Bs.is_contiguous(),
          "finite", bool(torch.isfinite(As).all()), bool(torch.isfinite(Bs).all()))
    slots = torch.zeros(2, device=A.device, dtype=torch.int32)
    call("care_absmax", ptr(As), ks, slabs * M, ks, slots.data_ptr())
    call("care_absmax", ptr(Bs), ks, slabs * N, ks, slots.data_ptr() + 4)
    torch.cuda.synchronize()
    print("  amax slots", slots.view(torch.float32).tolist(), "torch", float(As.abs().max()), float(Bs.abs().max()))
    a2 = torch.empty(slabs * M, 2 * ks, device=A.device, dtype=torch.float16)
    w3 = torch.empty(slabs * N, 3 * ks, device=A.device, dtype=torch.float16)
    call("care_split2_act_scaled", ptr(As), ks, ptr(a2), slabs * M, ks, slots.data_ptr())
    call("care_split3_weight_scaled", ptr(Bs), ks, ptr(w3), slabs * N, ks, slots.data_ptr() + 4)
    torch.cuda.synchronize()
    print("  a2 finite", bool(torch.isfinite(a2).all()), "absmax", float(a2.float().abs().max()), "w3 finite", bool(torch.isfinite(w3).all()),
          "absmax", float(w3.float().abs().max()))
    out = torch.empty(slabs * M, N, device=A.device, dtype=torch.float32)
    call("care_gemm_tile_split3_scaled", ptr(a2), ptr(w3), None, ptr(out), N, M, N, ks, slots.data_ptr(), slots.data_ptr() + 4, slabs)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(out)
    print("  slab outputs non-finite", int(bad.sum()), "of", out.numel(), "slabs with bad", sorted(set((bad.nonzero()[:, 0] // M).tolist()))[:12],
          "cols", sorted(set(bad.nonzero()[:, 1].tolist()))[:12])


def probe(A, Bt, bias=None, a_t=False, b_t=False):
    out = orig(A, Bt, bias, a_t, b_t)
    n[0] += 1
    ok_in = bool(torch.isfinite(A).all()) and bool(torch.isfinite(Bt).all())
    if not bool(torch.isfinite(out).all()) or not ok_in:
        print("product %d: A %s (a_t %s) Bt %s (b_t %s) inputs finite %s, output non-finite %d of %d" % (
            n[0], tuple(A.shape), a_t, tuple(Bt.shape), b_t, ok_in, int((~torch.isfinite(out)).sum()), out.numel()), flush=True)
        dissect(A, Bt, a_t, b_t)
        again = orig(A, Bt, bias, a_t, b_t)
        print("  the same call again: non-finite", int((~torch.isfinite(again)).sum()))
        sys.exit(1)
    return out


training._mm_x3 = probe
batch = {"feats": [f.to("cuda:0") for f in feats], "input_ids": ids.to("cuda:0")}
out = model(batch)
loss = (out["logits"] * torch.randn_like(out["logits"])).sum()
loss.backward()
print("no non-finite product in %d products" % n[0])
