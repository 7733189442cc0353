import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import GoldenCase, golden_names
from care_amd import get_framework, get_translator
for name in golden_names():
    gc = GoldenCase(name); opt, P, feats, ids = gc.build(); z = gc.z
    m = get_framework(opt).eval(); m.load_state_dict(P); m.set_compute_dtype("bf16"); m.to("cuda:0")
    f = [x.to("cuda:0") for x in feats]
    enc = m.encoding_phase(f)
    e1 = np.abs(enc["encoder_hidden_states"][0].cpu().numpy() - z["encoder_hidden_states_clip0"]).max()
    out = m.feedforward_step({"feats": f, "input_ids": ids.to("cuda:0")})
    n = z["tf_hidden_states"].shape[0]
    e2 = np.abs(out["hidden_states"][:n].cpu().numpy() - z["tf_hidden_states"]).max()
    e2m = np.abs(out["hidden_states"][:n].cpu().numpy() - z["tf_hidden_states"]).mean()
    lse = np.abs(torch.logsumexp(out["logits"], -1).cpu().numpy() - z["tf_logits_lse"]).max()
    extra = ""
    if "preds_attr" in z:
        extra = " preds %.1e labels_eq %s" % (np.abs(enc["preds_attr"].cpu().numpy() - z["preds_attr"]).max(), np.array_equal(enc["semantic_labels"].cpu().numpy(), z["semantic_labels"]))
    hyps, scores = get_translator(opt).translate_batch([m], {"feats": f})
    rh, rs = gc.hyps()
    same = sum(h == r for h, r in zip(hyps, rh))
    print("%-32s enc %.2e tf_hidden max %.2e mean %.2e lse %.1e%s hyps_equal %d/%d" % (name, e1, e2, e2m, lse, extra, same, len(rh)))
