"""Beam search over small batches: the resident launch (csrc/decode_resident_beam.hip) against the multi-launch search,
ms per pass (hipGraph replay, encode included) and us per decoder step.   python tools/beam_sweep.py [B ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

dev = torch.device("cuda:0")
opt = make_opt("msrvtt_care_beam5")
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype("bf16")
model.to(dev)
eng = model.engine()
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32, 51, 64, 96, 128]:
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    res = {}
    for name, cap in (("resident", 640), ("multi-launch", 0)):
        eng.resident_beam_max_rows = cap
        run = lambda: eng.translate_beam(feats, 5, 5, use_graph=True, lean=True)
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20
    print("B = %4d (%4d rows): resident %.3f ms (%.1f us / step)   multi-launch %.3f ms (%.1f us / step)   ratio %.2f" % (
        B, 5 * B, res["resident"] * 1e3, res["resident"] * 1e6 / eng.T, res["multi-launch"] * 1e3,
        res["multi-launch"] * 1e6 / eng.T, res["multi-launch"] / res["resident"]), flush=True)
