"""CPU: the oracle (oracle/care_cpu.py) against the reference's own outputs (tests/golden).

This is what pins the oracle.  Tolerance 1e-6 absolute on O(1) hidden states: the
oracle runs the same torch CPU kernels in the same order as the reference, so in
practice the match is exact or within one ulp.
"""
import numpy as np
import torch

from oracle import care_cpu

ATOL = 1e-6


def _close(a, b, atol=ATOL):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=0, atol=atol)


def test_encoding_phase_matches_reference(golden):
    opt, P, feats, _ = golden.build()
    z = golden.z
    with torch.no_grad():
        enc = care_cpu.encoding_phase(P, opt, feats)
    _close(enc["encoder_hidden_states"][0], z["encoder_hidden_states_clip0"])
    for i, m in enumerate(enc["mean_encoder_hidden_states"]):
        _close(m, z["mean_encoder_hidden_states_%d" % i])
    if "preds_attr" in z:
        _close(enc["preds_attr"], z["preds_attr"])
        _close(enc["avg_prob_attr"], z["avg_prob_attr"])
        assert np.array_equal(enc["semantic_labels"].numpy(), z["semantic_labels"])
        if "semantic_hidden_states" in z:
            _close(enc["semantic_hidden_states"], z["semantic_hidden_states"])
        if "semantic_embs_clip0" in z:
            _close(enc["semantic_embs"][0], z["semantic_embs_clip0"])
        assert float(z["concept_topk_min_gap"]) > 0.0, "fixture has an exact concept tie (parity unpinned there)"
    else:
        assert "preds_attr" not in enc


def test_teacher_forced_forward_matches_reference(golden):
    opt, P, feats, ids = golden.build()
    z = golden.z
    with torch.no_grad():
        out = care_cpu.feedforward_step(P, opt, feats, ids)
    assert np.array_equal(ids.numpy(), z["tf_input_ids"])
    n = z["tf_hidden_states"].shape[0]
    _close(out["hidden_states"][:n], z["tf_hidden_states"])
    logits = out["logits"]
    _close(torch.logsumexp(logits, -1), z["tf_logits_lse"], atol=1e-5)
    top = logits.topk(8, dim=-1)
    _close(top[0], z["tf_logits_top8_val"], atol=1e-5)
    assert np.array_equal(top[1].numpy(), z["tf_logits_top8_idx"])


def test_translate_batch_matches_reference(golden):
    opt, P, feats, _ = golden.build()
    ref_hyps, ref_scores = golden.hyps()
    hyps, scores = care_cpu.translate_batch(P, opt, feats)
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)
    for hs in hyps:
        for h in hs:
            assert all(isinstance(t, int) for t in h)


def test_state_dict_inventory(golden):
    """Structural known answers recorded by the reference itself:
    18,218,884 parameters for MSRVTT CARE base (notebooks/retrieval_robustness.ipynb:186-187)."""
    m = golden.meta
    if m["config"] in ("msrvtt_care", "msrvtt_care_beam5"):
        assert m["n_params"] == 18218884
