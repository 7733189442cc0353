"""Phase clocks of one step of the resident decode (csrc/decode_resident.hip; CARE_RESIDENT_PROF_STEP): per phase the
time workgroup 0 spent working (wait returned -> arrive) and the time from its arrive to the next phase's start."""
import os
import sys

os.environ.setdefault("CARE_RESIDENT_PROF_STEP", "3")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from care_amd import _lib, get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

dev = torch.device("cuda:0")
opt = make_opt("msrvtt_base_ami")
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype("bf16")
model.to(dev)
eng = model.engine()
names = ["qkv", "self_attn", "dense1", "q2", "cross_attn", "dense2", "ffn1", "ffn2", "vocab"]
for B in [int(a) for a in sys.argv[1:]] or [1, 128]:
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    for _ in range(3):
        eng.translate_greedy(feats, use_graph=False, lean=True)
    torch.cuda.synchronize()
    nb = _lib.load().care_decode_resident_scratch(B, eng.d, eng.ff, eng.V)
    sc = eng.ws("r_scratch", (nb,), torch.uint8)
    t = sc[2048:2048 + 8 * (2 * len(names) + 2)].view(torch.int64).cpu().tolist()
    vt = t[-4:]
    t = t[:-4] + [t[-4], t[-1]]
    print("  vocab: A rows %.2f us, items %.2f us, merge + store %.2f us" % ((vt[1] - vt[0]) / 100.0, (vt[2] - vt[1]) / 100.0, (vt[3] - vt[2]) / 100.0))
    print(flush=True); print("B = %d: step total %.2f us" % (B, (t[-1] - t[0]) / 100.0))
    for i, n in enumerate(names):
        work = (t[2 * i + 1] - t[2 * i]) / 100.0
        gap = (t[2 * i + 2] - t[2 * i + 1]) / 100.0 if 2 * i + 2 < len(t) else float("nan")
        print("  %-10s work %6.2f us   barrier+prefetch %6.2f us" % (n, work, gap))
