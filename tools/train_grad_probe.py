"""Per-parameter gradient error of the training mode against the oracle's autograd (GPU box)."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import GoldenCase
from test_gpu_training import _build, _loss, NO_DROP
from oracle import care_cpu

name = sys.argv[1] if len(sys.argv) > 1 else "msrvtt_cabase_b3"
opt, P, feats, ids, model = _build(GoldenCase(name), **NO_DROP)
model.train()
batch = {"feats": [f.to("cuda:0") for f in feats], "input_ids": ids.to("cuda:0")}
out = model(batch)
_loss(out, "cuda:0").backward()
Pc = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in P.items()}
ref = care_cpu.feedforward_step(Pc, opt, feats, ids)
_loss(ref, "cpu").backward()
for k in ("logits", "hidden_states", "preds_attr", "encoder_hidden_states"):
    if k in ref and ref[k] is not None:
        print(k, float((out[k].detach().cpu().reshape(-1) - ref[k].detach().reshape(-1)).abs().max()))
for k, p in model.named_parameters():
    g = Pc[k].grad
    if g is None or p.grad is None:
        print("%-60s ref %s mine %s" % (k, g is not None, p.grad is not None)); continue
    sc = float(g.abs().max()); df = float((p.grad.cpu() - g).abs().max())
    print("%-60s scale %.3e diff %.3e rel %.2e" % (k, sc, df, df / max(sc, 1e-12)))
