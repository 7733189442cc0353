"""GPU: the HIP path (through the C ABI) against the reference's golden outputs and the oracle.

Tolerances (north_star): fp32 mode - hidden states / concept outputs within 1e-5 of the
reference, greedy and beam token ids identical; bf16 mode - see test_bf16_*.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL_FP32 = 1e-5


def _model(opt, P, dtype="fp32"):
    from care_amd import get_framework

    model = get_framework(opt).eval()
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype(dtype)
    return model.to("cuda:0")


def _dev(feats):
    return [f.to("cuda:0") for f in feats]


def _maxdiff(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.max(np.abs(a - np.asarray(b))))


def test_encoding_phase_fp32(golden):
    opt, P, feats, _ = golden.build()
    z = golden.z
    enc = _model(opt, P).encoding_phase(_dev(feats))
    assert _maxdiff(enc["encoder_hidden_states"][0], z["encoder_hidden_states_clip0"]) < ATOL_FP32
    for i, m in enumerate(enc["mean_encoder_hidden_states"]):
        assert _maxdiff(m, z["mean_encoder_hidden_states_%d" % i]) < ATOL_FP32
    if "preds_attr" in z:
        assert _maxdiff(enc["preds_attr"], z["preds_attr"]) < ATOL_FP32
        assert _maxdiff(enc["avg_prob_attr"], z["avg_prob_attr"]) < ATOL_FP32
        assert np.array_equal(enc["semantic_labels"].cpu().numpy(), z["semantic_labels"])
        assert _maxdiff(enc["semantic_hidden_states"], z["semantic_hidden_states"]) < ATOL_FP32
        assert enc["attribute_prediction_prj"] is not None


def test_teacher_forced_fp32(golden):
    opt, P, feats, ids = golden.build()
    z = golden.z
    out = _model(opt, P).feedforward_step({"feats": _dev(feats), "input_ids": ids.to("cuda:0")})
    n = z["tf_hidden_states"].shape[0]
    assert _maxdiff(out["hidden_states"][:n], z["tf_hidden_states"]) < ATOL_FP32
    logits = out["logits"]
    assert logits.shape == (ids.shape[0], ids.shape[1], opt["vocab_size"])
    assert _maxdiff(torch.logsumexp(logits, -1), z["tf_logits_lse"]) < 2e-5
    top = logits.topk(8, dim=-1)
    assert _maxdiff(top[0], z["tf_logits_top8_val"]) < 2e-5
    assert np.array_equal(top[1].cpu().numpy(), z["tf_logits_top8_idx"])


def test_translate_batch_fp32(golden):
    from care_amd import get_translator

    opt, P, feats, _ = golden.build()
    ref_hyps, ref_scores = golden.hyps()
    hyps, scores = get_translator(opt).translate_batch([_model(opt, P)], {"feats": _dev(feats)})
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
        assert all(isinstance(s, float) for s in a)
    assert all(isinstance(t, int) for hs in hyps for h in hs for t in h)
