"""CPU: the oracle (oracle/care_cpu.py) against the reference's own outputs (tests/golden).

This is what pins the oracle.  Tolerance 1e-6 absolute on O(1) hidden states: the
oracle runs the same torch CPU kernels in the same order as the reference, so in
practice the match is exact or within one ulp.
"""
import numpy as np
import pytest
import torch

from oracle import care_cpu

ATOL = 1e-6
_FF = {}


def _feedforward(golden):
    """The oracle's teacher-forced forward with every auxiliary output, once per fixture."""
    if golden.name not in _FF:
        opt, P, feats, ids = golden.build()
        with torch.no_grad():
            _FF[golden.name] = care_cpu.feedforward_step(P, opt, feats, ids, auxiliary=True)
    return _FF[golden.name]


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
        if "semantic_labels" in z:
            assert np.array_equal(enc["semantic_labels"].numpy(), z["semantic_labels"])
        else:  # (G0L0: the concept head without a SemanticContainer)
            assert "semantic_labels" not in enc
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
    out = _feedforward(golden)
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


@pytest.mark.parametrize("name", __import__("conftest").ensemble_names())
def test_translate_batch_of_an_ensemble_matches_reference(name):
    """Model ensembling (models/Translator.py:39-52,112-133): every member encodes its own feature list and decodes the shared
    prefixes, the step's word log-probabilities are the members' log_softmax averaged - the reference Translator over two and
    three reference models, greedy and beam 5 (oracle/gen_golden.py ENSEMBLE_CASES)."""
    from conftest import EnsembleCase

    case = EnsembleCase(name)
    opts, Ps, feats = case.build()
    ref_hyps, ref_scores = case.hyps()
    hyps, scores, gaps = care_cpu.translate_batch_ensemble(Ps, opts, feats, return_gaps=True)
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)
    np.testing.assert_allclose([g["select"] for g in gaps], case.z["gap_select"], rtol=0, atol=1e-4)
    # an ensemble of one is the single model (the same code path, no averaging)
    one, one_scores = care_cpu.translate_batch_ensemble(Ps[:1], opts[:1], feats[:1])
    assert (one, one_scores) == care_cpu.translate_batch(Ps[0], opts[0], feats[0])


def test_state_dict_inventory(golden):
    """Structural known answers recorded by the reference itself:
    18,218,884 parameters for MSRVTT CARE base (notebooks/retrieval_robustness.ipynb:186-187)."""
    m = golden.meta
    if m["config"] in ("msrvtt_care", "msrvtt_care_beam5"):
        assert m["n_params"] == 18218884


def test_decision_gaps_recorded_with_the_fixture(golden):
    """The smallest decision margins of every clip's search (oracle/care_cpu.py `return_gaps`) are
    stored beside the reference outputs; the bf16 GPU tests read them to decide what a near-tie may
    excuse.  The "peaked" fixtures were chosen (gen_golden.py `search-peaked`) so that nothing needs
    excusing: every greedy step is decided by >= 0.1, the beam-5 winner was never within 0.1 of being
    pruned and finishes >= 0.05 ahead of the runner-up."""
    opt, P, feats, _ = golden.build()
    z = golden.z
    _, _, gaps = care_cpu.translate_batch(P, opt, feats, return_gaps=True)
    np.testing.assert_allclose([g["select"] for g in gaps], z["gap_select"], rtol=0, atol=1e-5)
    np.testing.assert_allclose([min(g["rank"], 1e30) for g in gaps], z["gap_rank"], rtol=0, atol=1e-5)
    np.testing.assert_allclose([min(g["best_slack"], 1e30) for g in gaps], z["gap_best_slack"], rtol=0, atol=1e-5)
    if "peaked" in golden.name:
        if opt["beam_size"] == 1:
            assert z["gap_select"].min() >= 0.1
        else:
            assert z["gap_best_slack"].min() >= 0.1 and z["gap_rank"].min() >= 0.05


def test_score_hypothesis_reproduces_the_beam_scores(golden):
    """care_cpu.score_hypothesis (the audit helper of the bf16 beam tests): teacher-forced rescoring of
    a reference hypothesis gives the reference's own length-normalised score."""
    opt, P, feats, _ = golden.build()
    ref_hyps, ref_scores = golden.hyps()
    with torch.no_grad():
        enc = care_cpu.encoding_phase(P, opt, feats)
    inputs = care_cpu.inputs_for_decoder(opt, enc)
    for i in range(min(2, len(ref_hyps))):
        one = {k: v[i:i + 1] for k, v in inputs.items()}
        assert abs(care_cpu.score_hypothesis(P, opt, one, ref_hyps[i][0]) - ref_scores[i][0]) < 2e-5


AUX_POS = [0, 5, 28]


def check_auxiliary(out, z, close):
    """The auxiliary entries of the decoder dict (models/Decoder/Transformer.py:239-252) against the
    fixture: attention probabilities, pre-residual contexts, intermediate embeddings (clip 0)."""
    for k in ("hidden_states", "all_hidden_states", "all_intra_attentions", "all_inter_attentions", "attention_probs",
              "context", "text_context", "self_embs", "cross_embs", "input_embs", "input_embs_exclude_bos", "sentence_embs"):
        assert k in out, k
    assert len(out["all_hidden_states"]) == int(z["aux_n_hidden_states"])
    t = z["aux_attention_probs"].shape[0]
    assert tuple(out["attention_probs"].shape[1:]) == z["aux_attention_probs"].shape
    assert tuple(out["input_embs_exclude_bos"].shape[1:]) == (t - 1, out["input_embs"].shape[2])
    close(out["attention_probs"][0], z["aux_attention_probs"])
    close(out["all_intra_attentions"][-1][0], z["aux_intra_attention"])
    close(out["all_inter_attentions"][-1][0][:, AUX_POS], z["aux_inter_attention"])
    for k in ("context", "text_context", "self_embs", "cross_embs", "input_embs", "sentence_embs"):
        close(out[k][0][AUX_POS], z["aux_" + k])
    if "aux_attr_attention" in z:
        close(out["attr_attention_probs"][-1][0][:, AUX_POS], z["aux_attr_attention"])
        assert out["gate_probs"] == () if "gate_probs" in out else True


def test_decoder_auxiliary_outputs_match_reference(golden):
    check_auxiliary(_feedforward(golden), golden.z, lambda a, b: _close(a, b, atol=2e-6))
