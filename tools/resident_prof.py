"""Phase clocks of one step of the resident decode (csrc/decode_resident.hip; care_decode_resident_debug): per phase the
time workgroup 0 spent working (wait returned -> arrive) and the time from its arrive to the next phase's start.

    python tools/resident_prof.py [--beam BM] [--config NAME] B [B ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from care_amd import _lib, get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

argv = sys.argv[1:]
bm, config = 1, "msrvtt_base_ami"
while argv and argv[0].startswith("--"):
    if argv[0] == "--beam":
        bm = int(argv[1])
    elif argv[0] == "--config":
        config = argv[1]
    argv = argv[2:]
dev = torch.device("cuda:0")
opt = make_opt(config)
model = get_framework(opt).eval()
model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
model.set_compute_dtype("bf16")
model.to(dev)
eng = model.engine()
_lib.load().care_decode_resident_debug(int(os.environ.get("CARE_RESIDENT_PROF_STEP", "3")), 0)
# (phase, marks): a GEMM phase stamps wait-returned / A rows in LDS / items done / stores issued, the others begin / end
# (the advance: begin / rows' candidates in LDS / end)
phases = [("qkv", 4), ("self_attn", 2), ("dense1", 4)]
for a in range(2 if eng.attr_att else 1):
    phases += [("q%d" % (a + 2), 4), ("static_attn%d" % a, 2), ("dense%d" % (a + 2), 4)]
phases += [("ffn1", 4), ("ffn2", 2), ("vocab", 4)] + ([("advance", 5)] if bm > 1 else [])
for B in [int(a) for a in argv] or [1, 128]:
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    run = (lambda: eng.translate_beam(feats, bm, bm, use_graph=False, lean=True)) if bm > 1 else \
          (lambda: eng.translate_greedy(feats, use_graph=False, lean=True))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert eng.last_decode.get("resident"), "this batch does not take the resident form"
    if bm > 1:
        nb = _lib.load().care_decode_resident_beam_scratch(B, bm, eng.d, eng.ff, eng.V)
        sc = eng.ws("rb_scratch", (nb,), torch.uint8)
    else:
        nb = _lib.load().care_decode_resident_scratch(B, eng.d, eng.ff, eng.V)
        sc = eng.ws("r_scratch", (nb,), torch.uint8)
    nmarks = sum(n for _, n in phases)
    t = [v / 100.0 for v in sc[2048:2048 + 8 * nmarks].view(torch.int64).cpu().tolist()]
    print(flush=True); print("B = %d%s: step total %.2f us (workgroup 0)" % (B, " x beam %d" % bm if bm > 1 else "", t[-1] - t[0]))
    i = 0
    for name, n in phases:
        m = t[i:i + n]
        nxt = t[i + n] if i + n < len(t) else float("nan")
        if n == 4:
            detail = "A rows %5.2f  items %5.2f  epilogue %5.2f" % (m[1] - m[0], m[2] - m[1], m[3] - m[2])
        elif n == 5:
            detail = "first row: groups %5.2f  logits %5.2f  | all rows %5.2f  advance %5.2f" % (m[1] - m[0], m[2] - m[1], m[3] - m[0], m[4] - m[3])
        else:
            detail = ""
        print("  %-13s work %6.2f us   hand-off + prefetch %6.2f us   %s" % (name, m[-1] - m[0], nxt - m[-1], detail))
        i += n
