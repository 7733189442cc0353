"""Beam search (beam 5) over small and medium batches: the resident launch (csrc/decode_resident_beam.hip), the chained step
(csrc/decode_chain.hip) and the multi-launch search, ms per pass (hipGraph replay, encode included, early exit off: all 29
steps) and us per decoder step.   python tools/beam_sweep.py [--mode bf16|fp16] [--config NAME] [--beam K] [B ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

args = sys.argv[1:]
mode = args.pop(args.index("--mode") + 1) if "--mode" in args else "bf16"
config = args.pop(args.index("--config") + 1) if "--config" in args else "msrvtt_care_beam5"
only = args.pop(args.index("--only") + 1) if "--only" in args else None
K = int(args.pop(args.index("--beam") + 1)) if "--beam" in args else 5
args = [a for a in args if not a.startswith("--")]
dev = torch.device("cuda:0")
opt = make_opt(config)
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype(mode)
model.to(dev)
eng = model.engine()
for B in [int(a) for a in args] or [1, 4, 16, 32, 51, 64, 96, 128, 256, 512, 819]:
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    res = {}
    for name, cap_r, cap_c in (("resident", 640, 0), ("chain", 0, 1 << 20), ("multi-launch", 0, 0)):
        eng.resident_beam_max_rows, eng.chain_beam_max_rows = cap_r, cap_c
        if only and name != only:
            continue
        if name == "resident" and not eng.resident_beam_ok(B, K, K):
            continue
        if name == "chain" and not eng.chain_beam_ok(B, K, K):
            continue
        run = lambda: eng.translate_beam(feats, K, K, use_graph=True, lean=True, early_exit=False)
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        n = 20 if B <= 256 else 8
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n
    print("B = %4d (%5d rows) %s: " % (B, K * B, mode) + "   ".join(
        "%s %.3f ms (%.1f us / step, %.1f K captions/s)" % (k, v * 1e3, v * 1e6 / eng.T, B / v / 1e3) for k, v in res.items()), flush=True)
