"""Bit-exact regression of the resident decode across library builds (GPU box):
    CARE_HIP_LIB=<old libcare_hip.so> python tools/ab_resident.py save ; python tools/ab_resident.py check
30 (model, batch) cases; every case also run three times (eager, captured, replayed) and compared with itself."""
import os, sys, torch
sys.path.insert(0, ".")
from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict
mode = sys.argv[1]
dev = torch.device("cuda:0")
out = {}
for cfg in ("msrvtt_base_ami", "msrvtt_care", "msrvtt_cabase"):
    opt = make_opt(cfg); model = get_framework(opt).eval()
    boost = {"cls_head.tgt_word_prj.weight": {3: 4.0, 0: 3.0}}
    model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()], row_scale=boost), strict=True)
    model.set_compute_dtype("bf16"); model.to(dev); eng = model.engine()
    for B in (1, 5, 16, 17, 33, 64, 100, 128, 200, 256):
        gen = torch.Generator(device=dev); gen.manual_seed(B)
        feats = [torch.randn(s, generator=gen, device=dev) for s in feat_shapes(opt, B)]
        for rep in range(3):
            _, fed, length, score = eng.translate_greedy(feats, use_graph=rep > 0, lean=True)
            torch.cuda.synchronize()
            key = "%s_%d" % (cfg, B)
            cur = (fed.cpu().clone(), length.cpu().clone(), score.cpu().clone(), int(eng.last_decode["steps"]))
            if rep == 0:
                out[key] = cur
            else:
                assert all(torch.equal(a, b) for a, b in zip(cur[:3], out[key][:3])) and cur[3] == out[key][3], (key, rep)
path = "gpurun_out/ab_resident.pt"
if mode == "save":
    torch.save(out, path); print("saved", len(out))
else:
    ref = torch.load(path)
    bad = 0
    for k, v in out.items():
        r = ref[k]
        ok = torch.equal(v[0], r[0]) and torch.equal(v[1], r[1]) and torch.equal(v[2], r[2]) and v[3] == r[3]
        if not ok:
            bad += 1
            print("MISMATCH", k, "steps", v[3], r[3], "len eq", torch.equal(v[1], r[1]), "score maxdiff", float((v[2] - r[2]).abs().max()))
    print("checked", len(out), "mismatches", bad)
