"""Beam-5 early exit on models whose EOS row is boosted by different factors (GPU box): steps run, compactions, time."""
import sys, time
import torch
sys.path.insert(0, ".")
from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

dev = torch.device("cuda:0")
B = 4096
for boost in [float(a) for a in sys.argv[1:]] or [5.0, 8.0, 12.0, 20.0]:
    opt = make_opt("msrvtt_care_beam5")
    model = get_framework(opt).eval()
    P = synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                         row_scale={"cls_head.tgt_word_prj.weight": {3: boost}})
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype("bf16")
    model.to(dev)
    eng = model.engine()
    gen = torch.Generator(device=dev); gen.manual_seed(2000)
    feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
    out = {}
    for ee in (True, False):
        run = lambda: eng.translate_beam(feats, 5, 5, use_graph=True, lean=True, early_exit=ee)
        for _ in range(3):
            r = run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = run()
        torch.cuda.synchronize(); out[ee] = (time.perf_counter() - t0) / 5
        if ee:
            st = dict(eng.last_decode)
            hyps = r[0] if isinstance(r, tuple) else r
    print("boost %.0f: early %.2f ms fixed %.2f ms  %s" % (boost, out[True] * 1e3, out[False] * 1e3, st), flush=True)
