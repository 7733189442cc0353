"""Hidden-state / memory / logsumexp error of the 16-bit modes against the reference fixtures, and whether the greedy /
beam winners are the reference's: one line of JSON per (fixture, mode).  python tools/h16_err.py [mode ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GoldenCase, golden_names  # noqa: E402

from care_amd import get_framework, get_translator  # noqa: E402

modes = sys.argv[1:] or ["bf16", "fp16"]
worst = {m: dict(hidden_max=0.0, hidden_mean=0.0, mem_max=0.0, lse_max=0.0, same=0, clips=0) for m in modes}
for name in golden_names():
    g = GoldenCase(name)
    opt, P, feats, ids = g.build()
    z = g.z
    for mode in modes:
        model = get_framework(opt).eval()
        model.load_state_dict(P, strict=True)
        model.set_compute_dtype(mode)
        model.to("cuda:0")
        dev = [f.to("cuda:0") for f in feats]
        out = model.feedforward_step({"feats": dev, "input_ids": ids.to("cuda:0")})
        n = z["tf_hidden_states"].shape[0]
        diff = np.abs(out["hidden_states"][:n].float().cpu().numpy() - z["tf_hidden_states"])
        mem = float(np.max(np.abs(out["encoder_hidden_states"][0].float().cpu().numpy() - z["encoder_hidden_states_clip0"])))
        lse = float(np.max(np.abs(torch.logsumexp(out["logits"], -1).float().cpu().numpy() - z["tf_logits_lse"])))
        hyps, scores = get_translator(opt).translate_batch([model], {"feats": dev})
        ref_hyps, _ = g.hyps()
        same = sum(int(h[0] == r[0]) for h, r in zip(hyps, ref_hyps))
        rec = dict(case=name, mode=mode, hidden_max=float(diff.max()), hidden_mean=float(diff.mean()), mem_max=mem, lse_max=lse,
                   same=same, clips=len(hyps))
        print(json.dumps(rec), flush=True)
        w = worst[mode]
        for k in ("hidden_max", "hidden_mean", "mem_max"):
            w[k] = max(w[k], rec[k])
        if "peaked" not in name:
            w["lse_max"] = max(w["lse_max"], lse)
        w["same"] += same
        w["clips"] += len(hyps)
print(json.dumps(dict(worst=worst)))
