"""Projected-K/V vs absorbed cross-attention on the same clips (bf16): identical greedy prefixes and the log-prob
difference after 1 and 5 steps (measured: max 2e-3 / mean 5e-4 per step for Base and CARE).  Run on the GPU box."""
import sys, torch
sys.path.insert(0, ".")
from care_amd import get_framework
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict
for config in ("msrvtt_base_ami", "msrvtt_care"):
    opt = make_opt(config)
    model = get_framework(opt).eval()
    P = synth_state_dict(11, [(k, tuple(v.shape)) for k, v in model.state_dict().items()])
    model.load_state_dict(P, strict=True); model.set_compute_dtype("bf16"); model.to("cuda:0")
    eng = model.engine(); eng.LATENT_MIN_ROWS = 1
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    feats = [torch.randn(s, generator=gen, device="cuda:0") for s in feat_shapes(opt, 256)]
    res = {}
    for latent in (True, False):
        eng.latent = latent
        enc = eng.encode(feats)
        for steps in (1, 5):
            fed, length, score = eng.greedy(enc["encoder_hidden_states"], enc.get("semantic_hidden_states"), steps=steps,
                                            sem_embs=enc.get("semantic_embs"))
            res[(latent, steps)] = (fed.clone(), score.clone())
    for steps in (1, 5):
        fa, sa = res[(True, steps)]; fb, sb = res[(False, steps)]
        same = (fa[:, 1:steps + 1] == fb[:, 1:steps + 1]).all(1)
        print(config, "steps", steps, "identical prefixes %d/256" % int(same.sum()),
              "score diff on identical: max %.4f mean %.5f" % ((sa - sb)[same].abs().max().item(), (sa - sb)[same].abs().mean().item()))
