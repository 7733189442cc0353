"""Greedy decode over small and medium batches: us per decoder step of the whole pass (hipGraph replay, encode included).
python tools/greedy_sweep.py [--config NAME] [--mode bf16|fp16] [B ...]"""
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
config = args.pop(args.index("--config") + 1) if "--config" in args else "msrvtt_base_ami"
args = [a for a in args if not a.startswith("--")]
dev = torch.device("cuda:0")
opt = make_opt(config)
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype(mode)
model.to(dev)
eng = model.engine()
for B in [int(a) for a in args] or [1, 16, 64, 128, 256]:
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    run = lambda: eng.translate_greedy(feats, use_graph=True, lean=True)
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    n = 30 if B <= 512 else 8
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s %s B = %5d: %.3f ms per pass, %.2f us / step, %.1f K captions/s (%s)" % (
        config, mode, B, dt * 1e3, dt * 1e6 / eng.T, B / dt / 1e3, "resident" if eng.last_decode.get("resident") else "multi-launch"), flush=True)
