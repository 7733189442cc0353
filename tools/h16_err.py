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

def block_budget(config=os.environ.get("H16_CONFIG", "msrvtt_base_ami"), B=64, seed=3):
    """Where a 16-bit mode's hidden-state error comes from, block by block: the decoder's intermediate rows (the auxiliary
    entries of `feedforward_step`: Decoder/Transformer.py:239-252) of the 16-bit mode against fp32 mode of the same engine, on
    whole tensors (B clips x 29 positions).  `python tools/h16_err.py --blocks [mode ...]`"""
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_input_ids, synth_state_dict

    opt = make_opt(config)
    model = get_framework(opt).eval()
    model.load_state_dict(synth_state_dict(0, [(k, tuple(v.shape)) for k, v in model.state_dict().items()]), strict=True)
    model.to("cuda:0")
    feats = [f.to("cuda:0") for f in synth_feats(seed, feat_shapes(opt, B))]
    ids = synth_input_ids(seed, B, opt["max_len"] - 1, opt["vocab_size"]).to("cuda:0")
    keys = [("encoder_hidden_states", "memory rows (embedder Linear + LayerNorm)"), ("input_embs", "embedding sum + LayerNorm"),
            ("self_embs", "+ self-attention block (QKV, softmax, dense, LayerNorm)"),
            ("cross_embs", "+ cross-attention block (query, K/V, softmax, dense, LayerNorm)"),
            ("hidden_states", "+ FFN block (dense1, activation, dense2, LayerNorm) = the hidden state")]
    outs = {}
    for mode in ["fp32"] + [m for m in (sys.argv[2:] or ["fp16", "bf16"])]:
        model.set_compute_dtype(mode)
        out = model.feedforward_step({"feats": feats, "input_ids": ids})
        outs[mode] = {k: out[k].float().clone() for k, _ in keys}
    for mode in outs:
        if mode == "fp32":
            continue
        for k, what in keys:
            diff = (outs[mode][k] - outs["fp32"][k]).abs()
            print(json.dumps(dict(mode=mode, after=what, max_abs=round(float(diff.max()), 6), mean_abs=round(float(diff.mean()), 7),
                                  rms_of_values=round(float(outs["fp32"][k].pow(2).mean().sqrt()), 3))), flush=True)


if "--blocks" in sys.argv[1:2]:
    block_budget()
else:
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
