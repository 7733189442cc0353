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
        if "semantic_labels" in z:
            assert np.array_equal(enc["semantic_labels"].cpu().numpy(), z["semantic_labels"])
        else:  # (G0L0: the concept head without a SemanticContainer)
            assert "semantic_labels" not in enc
        if "semantic_hidden_states" in z:
            assert _maxdiff(enc["semantic_hidden_states"], z["semantic_hidden_states"]) < ATOL_FP32
        if "semantic_embs_clip0" in z:
            assert _maxdiff(enc["semantic_embs"][0], z["semantic_embs_clip0"]) < ATOL_FP32
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


def test_decoder_auxiliary_outputs_fp32(golden):
    """feedforward_step returns the reference decoder's whole dict (Decoder/Transformer.py:239-252):
    attention probabilities of every block, pre-residual contexts, intermediate embeddings."""
    from test_oracle_golden import check_auxiliary

    opt, P, feats, ids = golden.build()
    out = _model(opt, P).feedforward_step({"feats": _dev(feats), "input_ids": ids.to("cuda:0")})
    check_auxiliary(out, golden.z, lambda a, b: np.testing.assert_allclose(a.detach().float().cpu().numpy(), b, rtol=0, atol=2e-5))
    pr = out["all_inter_attentions"][-1]
    assert pr.shape == (ids.shape[0], opt["num_attention_heads"], ids.shape[1], out["attention_probs"].shape[2])
    assert float((pr.sum(-1) - 1).abs().max()) < 1e-5
    # a decode-loop style call skips them unless asked
    model = _model(opt, P)
    enc = model.encoding_phase(_dev(feats))
    step = model.decoding_phase(ids[:, :3].to("cuda:0"), model.prepare_inputs_for_decoder(enc, {}), last_time_step_logits=True)
    assert "attention_probs" not in step and step["logits"].shape == (ids.shape[0], opt["vocab_size"])


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


@pytest.mark.parametrize("name", __import__("conftest").ensemble_names())
@pytest.mark.parametrize("mode", ["fp32", "fp16", "bf16"])
def test_translate_batch_of_an_ensemble(name, mode):
    """Model ensembling through the drop-in Translator (models/Translator.py:39-52,112-133): two and three models - CARE, Base,
    G1L0; one feature list for all or one per model - greedy and beam 5, against the reference Translator's own output over
    the reference models (tests/golden/ensemble).  fp32 mode: the reference's hypotheses and scores; 16-bit modes: the same
    unless the fixture's search had a decision closer than the mode's noise (the recorded margins are 2e-4 .. 5e-4)."""
    from conftest import EnsembleCase
    from care_amd import get_translator

    case = EnsembleCase(name)
    opts, Ps, feats = case.build()
    models = [_model(o, P, mode) for o, P in zip(opts, Ps)]
    batch = {"feats": [_dev(f) for f in feats]} if case.meta["own_feats"] else {"feats": _dev(feats[0])}
    tr = get_translator(opts[0])
    eager = tr.translate_batch(models, batch, use_graph=False)
    hyps, scores = tr.translate_batch(models, batch)
    for _ in range(3):   # the first pass of a key runs eagerly, the second is captured into a hipGraph, later ones replay it
        assert tr.translate_batch(models, batch) == (hyps, scores) == eager
    del batch
    for _ in range(4):   # ... and over recycled device buffers (freed, allocated again at the same addresses: replays)
        fresh = {"feats": [_dev(f) for f in feats]} if case.meta["own_feats"] else {"feats": _dev(feats[0])}
        assert tr.translate_batch(models, fresh) == (hyps, scores)
        del fresh
    ref_hyps, ref_scores = case.hyps()
    assert [len(h) for h in hyps] == [len(h) for h in ref_hyps]
    assert all(isinstance(t, int) for hs in hyps for h in hs for t in h) and all(isinstance(x, float) for sc in scores for x in sc)
    if mode == "fp32":
        assert hyps == ref_hyps
        for a, b in zip(scores, ref_scores):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
        return
    tol = 5e-2 if mode == "bf16" else 1e-2
    z = case.z
    for i, (hs, rs) in enumerate(zip(hyps, ref_hyps)):
        if hs == rs:
            np.testing.assert_allclose(scores[i], ref_scores[i], rtol=0, atol=tol)
        else:  # a flip needs a near-tie of the reference search of that clip
            assert min(z["gap_select"][i], z["gap_rank"][i], z["gap_best_slack"][i]) < tol, (i, hs, rs)
            assert abs(scores[i][0] - ref_scores[i][0]) < 10 * tol
    # one feature list per model and the same list for all are the same search when the lists are equal
    if not case.meta["own_feats"] and mode == "fp16":
        again, again_scores = get_translator(opts[0]).translate_batch(models, {"feats": [_dev(feats[0]) for _ in models]})
        assert (again, again_scores) == (hyps, scores)


@pytest.mark.parametrize("config,beam,B,mode", [("msrvtt_base_ami", 1, 96, "bf16"), ("msrvtt_care_beam5", 5, 64, "bf16"),
                                                 ("msrvtt_base_ami", 1, 3072, "fp16"), ("msrvtt_care_beam5", 5, 640, "bf16"),
                                                 ("msrvtt_care", 1, 7, "fp32")])
def test_translate_batches_equals_translate_batch(config, beam, B, mode):
    """The pipelined entry (lists of batch k assembled while batch k + 1 decodes: in the host waits of the segmented
    passes, after the asynchronous resident launches) against one translate_batch call per batch, on host-fed batches of
    different content through FeaturePrefetcher(depth=3) - and CaptionRunner.translate_steps over the same loader.  A model
    that ends its captions at mixed lengths (EOS row x 5), so slicing, early exit and compaction are all live."""
    from care_amd import get_translator
    from care_amd.checkpoint import CaptionRunner
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.data import FeaturePrefetcher
    from care_amd.synth import synth_state_dict

    opt = make_opt(config, beam_size=beam, topk=min(beam, 2))
    runner = CaptionRunner(opt)
    model = runner.captioner.eval()
    P = synth_state_dict(3, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                         row_scale={"cls_head.tgt_word_prj.weight": {3: 5.0, 0: 2.0}})
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype(mode)
    model.to("cuda:0")
    tr = get_translator(opt)
    gen = torch.Generator().manual_seed(11)
    host = [[torch.randn(s, generator=gen).pin_memory() for s in feat_shapes(opt, B)] for _ in range(4)]
    loader = lambda: ({"feats": f} for f in FeaturePrefetcher(iter(host), "cuda:0", depth=3))
    want = [tr.translate_batch([model], b) for b in loader()]
    assert len({len(h[0]) for hyps, _ in want for h in hyps}) > 3            # mixed lengths
    assert want[0] != want[1]                                               # different batches
    for _ in range(3):                                                      # eager, capture, replay
        assert list(tr.translate_batches([model], loader())) == want
    assert list(runner.translate_steps(loader())) == want
    assert model.engine().idle_hook is None


# ----------------------------------------------------------------------------- bf16 mode
# A bf16 rounding of an O(1) activation is already up to 2^-8 = 3.9e-3, so the 1e-3 of the
# north star is not attainable by any path that stores or multiplies bf16.
# Bars asserted here = the worst value MEASURED over the fixtures on MI355X + 25% (max-abs,
# mean-abs of the teacher-forced hidden states against the fp32 reference; bench.py reports the same
# two numbers for the benchmarked model in its JSON line); concept outputs exact (they stay fp32);
# logsumexp of the logits <= 1e-3; greedy ids identical wherever the reference's own
# top-1/top-2 margin exceeds the bf16 noise (audited against the oracle's margins).
BF16_MAX, BF16_MEAN = 1.9e-2, 3.0e-3  # measured over the 20 fixtures: 1.60e-2 / 2.35e-3 (tools/h16_err.py)
# logsumexp of the logits: bf16 noise of a logit scales with the norm of its vocabulary row, so the
# `peaked` fixtures (rows scaled by 12 / 20, logits up to +-30) get a bar of their own
BF16_LSE, BF16_LSE_PEAKED = 1e-4, 3.5e-2  # measured 6.3e-5; 2.2e-2 .. 2.6e-2 on the peaked fixtures

# ----------------------------------------------------------------------------- the 16-bit modes, side by side
# `fp16` (round 5) runs the SAME kernels compiled for IEEE half (libcare_hip_f16.so): 11 significand bits instead of 8 at
# bf16's bytes and MFMA rate.  north_star asks a 16-bit mode for 1e-3 on hidden states: fp16 mode's MEAN error is
# 3.0e-4 and its MAX 1.84e-3 over the 20 fixtures (tools/h16_err.py on an MI355X; a single fp16 rounding of an O(1..4)
# LayerNorm output is already up to 2^-11 .. 2^-9 = 4.9e-4 .. 2e-3, so no fp16-storing path can hold the max at 1e-3) -
# the bars below are 2.5e-3 max (measured 1.84e-3 through the module API, 2.05e-3 through the fused teacher-forced path,
# + 20 %) and 4e-4 mean (north_star's 1e-3 with room to spare); every other bar is bf16's divided by ~6-8, the ratio
# of the two roundings.
MODES = {
    "bf16": dict(max=BF16_MAX, mean=BF16_MEAN, lse=BF16_LSE, lse_peaked=BF16_LSE_PEAKED, greedy_tie=5e-3, score=2e-2, beam_tie=2e-2,
                 ppl=3e-3),
    "fp16": dict(max=2.5e-3, mean=4.0e-4, lse=1.5e-5, lse_peaked=5e-3, greedy_tie=1e-3, score=3e-3, beam_tie=3e-3, ppl=5e-4),
}
H16 = sorted(MODES)
PRELN_SCALE = 1.6  # hidden-state bars of the pre-LN fixtures (see test_encoding_and_teacher_forced_bf16)


# concept models, bf16 mode: the embedder multiplies fp32 operands as three passes over FP16 hi/lo pieces
# (a_hi w_hi + a_hi w_lo + a_lo w_hi); what is dropped is ~2^-22 per product: the fp32 bars hold
SPLIT_MEM, SPLIT_PREDS = ATOL_FP32, ATOL_FP32


def _record(name, **values):
    """Measured errors -> gpurun_out/bf16_err.jsonl (scratch; the bars above are set from it)."""
    import json
    import os

    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "bf16_err.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=name, **values)) + "\n")


@pytest.mark.parametrize("mode", H16)
def test_encoding_and_teacher_forced_bf16(golden, mode):
    bar = MODES[mode]
    opt, P, feats, ids = golden.build()
    z = golden.z
    model = _model(opt, P, mode)
    enc = model.encoding_phase(_dev(feats))
    if "preds_attr" in z and opt["encoder"] == "Embedder":
        # the concept path keeps fp32 operands in bf16 mode (hi/lo split products, care_gemm_ln_split):
        # same labels; probabilities and memory within the split's error of the reference
        assert _maxdiff(enc["preds_attr"], z["preds_attr"]) < SPLIT_PREDS
        if "semantic_labels" in z:
            assert np.array_equal(enc["semantic_labels"].cpu().numpy(), z["semantic_labels"])
        assert _maxdiff(enc["encoder_hidden_states"][0], z["encoder_hidden_states_clip0"]) < SPLIT_MEM
    else:
        assert _maxdiff(enc["encoder_hidden_states"][0], z["encoder_hidden_states_clip0"]) < bar["max"]
    out = model.feedforward_step({"feats": _dev(feats), "input_ids": ids.to("cuda:0")})
    n = z["tf_hidden_states"].shape[0]
    diff = np.abs(out["hidden_states"][:n].float().cpu().numpy() - z["tf_hidden_states"])
    lse = _maxdiff(torch.logsumexp(out["logits"], -1), z["tf_logits_lse"])
    _record(golden.name + "#" + mode, hidden_max=float(diff.max()), hidden_mean=float(diff.mean()), lse_max=lse,
            mem_max=_maxdiff(enc["encoder_hidden_states"][0], z["encoder_hidden_states_clip0"]))
    # pre-LN decoders: the residual stream is never normalised, so the final LayerNorm sees 16-bit noise accumulated over three
    # un-normalised sums - measured 1.5 - 1.9 x the post-LN fixtures' error in both modes (bf16 2.8e-2 / 4.6e-3, fp16 3.6e-3 / 5.7e-4)
    k = PRELN_SCALE if "preln" in golden.name else 1.0
    assert diff.max() < k * bar["max"] and diff.mean() < k * bar["mean"], (diff.max(), diff.mean())
    assert lse < (bar["lse_peaked"] if "peaked" in golden.name else bar["lse"]), lse


def _audit_greedy(P, opt, one, h, r, tol):
    """A bf16 greedy caption `h` that differs from the oracle's `r`: the first differing step must be
    a near-tie of the ORACLE's distribution on the common prefix (margin < tol), and the token the
    GPU took must be within tol of the best."""
    from oracle import care_cpu

    t = next((k for k in range(min(len(h), len(r))) if h[k] != r[k]), None)
    assert t is not None, "one caption is a prefix of the other: {} vs {}".format(h, r)
    prefix = torch.tensor([[2] + r[:t]])
    logp = torch.log_softmax(care_cpu.decoding_phase(P, opt, prefix, one, True)["logits"], dim=1)[0]
    top2 = logp.topk(2)[0]
    assert float(top2[0] - top2[1]) < tol, "bf16 greedy diverged at a clear-margin step ({:.4f})".format(
        float(top2[0] - top2[1]))
    assert float(logp.max() - logp[h[t]]) < tol


GREEDY_TIE_TOL = 5e-3   # log-prob units; a step decided by less than this may flip in bf16
CLEAR_MARGIN = 0.1      # a clip whose every step is decided by more than this must be bit-exact


@pytest.mark.parametrize("mode", H16)
@pytest.mark.parametrize("form", ["projected", "absorbed", "resident"])
def test_greedy_bf16_matches_up_to_near_ties(golden, form, mode):
    """bf16 greedy ids vs the oracle.  A clip whose reference search never saw a margin below
    CLEAR_MARGIN (fixture `gap_select`) must come out bit-exact - that is every clip of the `peaked`
    fixtures; any other divergence must start at a step where the oracle's own top-1/top-2 log-prob
    margin is below GREEDY_TIE_TOL (random-init logits are nearly flat).  Both forms of the
    cross-attention of the multi-launch decode: the absorbed form (the default wherever the model allows it) and
    projected K/V (`engine.latent = False`); and the resident form (the whole decode of a small batch as one launch,
    csrc/decode_resident.hip - what the Translator runs at these batch sizes by default)."""
    from care_amd import get_translator
    from oracle import care_cpu

    opt, P, feats, _ = golden.build()
    if opt.get("beam_size", 1) != 1:
        pytest.skip("greedy audit")
    bar = MODES[mode]
    model = _model(opt, P, mode)
    eng = model.engine()
    absorbed = form == "absorbed"
    if absorbed and not eng.latent_capable:
        pytest.skip("the absorbed cross-attention needs bf16 mode and a d_model of 512 / 768 / 1024")
    if form == "resident":
        if not eng.resident_ok(feats[0].shape[0]):
            pytest.skip("the resident decode covers d_model = 512 in bf16 mode")
    else:
        eng.resident_max_rows = 0
        eng.latent = absorbed  # the default is the absorbed form wherever the model allows it, at every batch size
        assert eng.latent_for(feats[0].shape[0]) == absorbed
    hyps, scores = get_translator(opt).translate_batch([model], {"feats": _dev(feats)})
    if form == "resident":
        assert eng.last_decode.get("resident") and 1 <= int(eng.last_decode["steps"]) <= eng.T
    ref_hyps, ref_scores = golden.hyps()
    gap = golden.z["gap_select"]
    enc = care_cpu.encoding_phase(P, opt, feats)
    inputs = care_cpu.inputs_for_decoder(opt, enc)
    for i, (h, r) in enumerate(zip(hyps, ref_hyps)):
        h, r = h[0], r[0]
        if gap[i] >= CLEAR_MARGIN:
            assert h == r, "clip {}: every reference step decided by >= {} but bf16 ids differ".format(i, gap[i])
        if h == r:
            assert abs(scores[i][0] - ref_scores[i][0]) < (bar["lse_peaked"] if "peaked" in golden.name else bar["score"])
            continue
        _audit_greedy(P, opt, {k: v[i:i + 1] for k, v in inputs.items()}, h, r, bar["greedy_tie"])
    if "peaked" in golden.name:
        assert hyps == ref_hyps


BEAM_SCORE_TOL = 2e-2   # length-normalised log-prob: bf16 noise on a hypothesis' own score
BEAM_TIE_TOL = 2e-2     # how close two hypotheses / a pruning decision must be to count as a tie


@pytest.mark.parametrize("mode", H16)
@pytest.mark.parametrize("form", ["projected", "absorbed", "small", "resident", "chain"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_beam_bf16_vs_oracle(golden, form, use_graph, mode):
    """bf16 beam search (fused two-pass selection, device beam state) vs the reference, per clip:
      * the score the GPU reports for its best hypothesis is that hypothesis' exact fp32 score
        (oracle teacher-forced rescoring) within bf16 noise - whatever path the search took;
      * if the best hypothesis differs from the reference's, the two are a near-tie under exact
        scoring, or the reference's winner was within BEAM_TIE_TOL of being pruned at some step
        (fixture `gap_best_slack`), or its final lead over the runner-up was that small (`gap_rank`);
      * a clip with clear margins everywhere (the `peaked` fixture) must be bit-exact.
    use_graph: the hipGraph-captured pass (third call on the same buffers replays) vs eager.
    form: the cross-attention of the large-batch pass, projected or absorbed, with its fused embedder; `small`: the
    multi-launch search with the small-batch forms (engine.small_forms: unfused embedder, projected K/V); `resident`:
    what the engine runs by default for a batch this small - the whole search as ONE launch
    (csrc/decode_resident_beam.hip: group lists in the vocabulary phase, recomputed candidate logits, Beam.advance)."""
    from care_amd import get_translator
    from oracle import care_cpu

    opt, P, feats, _ = golden.build()
    if opt.get("beam_size", 1) == 1:
        pytest.skip("beam audit")
    bar = MODES[mode]
    model = _model(opt, P, mode)
    eng = model.engine()
    absorbed = form == "absorbed"
    if absorbed and not eng.latent_capable:
        pytest.skip("the absorbed cross-attention needs bf16 mode and a d_model of 512 / 768 / 1024")
    bm, need = int(opt["beam_size"]), max(int(opt["beam_size"]), int(opt.get("topk", 1)))
    if form == "resident":
        if not eng.resident_beam_ok(feats[0].shape[0], bm, need):
            pytest.skip("the resident beam search covers d_model = 512 in bf16 mode, beam_size <= 5")
    elif form == "chain":
        eng.resident_beam_max_rows, eng.chain_beam_max_rows = 0, 4096  # (the chain is opt-in: engine.chain_beam_max_rows)
        if not eng.chain_beam_ok(feats[0].shape[0], bm, need):
            pytest.skip("the chained beam step covers 16-bit modes, beam_size <= 5")
    elif form == "small":
        eng.resident_beam_max_rows = 0
        eng.chain_beam_max_rows = 0
        if not eng.small_forms(feats[0].shape[0]):
            pytest.skip("the small-batch forms cover d_model = 512 in bf16 mode")
    else:
        eng.resident_beam_max_rows = 0
        eng.chain_beam_max_rows = 0
        eng.resident_max_rows = 0
        eng.latent = absorbed
    # the per-row top-k: two passes of the vocabulary GEMM (what large batches use) in the graph variant,
    # logits + care_beam_select (what the engine picks for a batch this small) in the eager one
    eng.BEAM_FUSED_MIN_ROWS = 1 if use_graph else type(eng).BEAM_FUSED_MIN_ROWS
    dev = _dev(feats)
    tr = get_translator(opt)
    for _ in range(3 if use_graph else 1):  # first sight (eager), capture, replay
        hyps, scores = tr.translate_batch([model], {"feats": dev}, use_graph=use_graph)
    if form == "resident":
        assert eng.last_decode.get("resident") and 1 <= int(eng.last_decode["steps"]) <= eng.T
    if form == "chain":
        assert eng.last_decode.get("chain") and 1 <= int(eng.last_decode["steps"]) <= eng.T
    if use_graph:
        assert any(isinstance(v, tuple) for k, v in eng._graphs.items() if k[0] in ("beam", "bseg0", "bres", "bchain")), "beam pass was not captured"
    ref_hyps, ref_scores = golden.hyps()
    z = golden.z
    enc = care_cpu.encoding_phase(P, opt, feats)
    inputs = care_cpu.inputs_for_decoder(opt, enc)
    for i, (hs, rs) in enumerate(zip(hyps, ref_hyps)):
        one = {k: v[i:i + 1] for k, v in inputs.items()}
        h, r = hs[0], rs[0]
        exact_h = care_cpu.score_hypothesis(P, opt, one, h)
        assert abs(scores[i][0] - exact_h) < (bar["lse_peaked"] if "peaked" in golden.name else bar["score"]), \
            (i, scores[i][0], exact_h)
        clear = z["gap_best_slack"][i] >= CLEAR_MARGIN and z["gap_rank"][i] >= 0.05
        if clear:
            assert h == r, "clip {}: clear reference margins but the bf16 beam winner differs".format(i)
        if h != r:
            near_tie = abs(exact_h - ref_scores[i][0]) < bar["beam_tie"]
            assert near_tie or z["gap_best_slack"][i] < bar["beam_tie"] or z["gap_rank"][i] < bar["beam_tie"], \
                "clip {}: bf16 beam winner {} (exact {:.4f}) vs reference {} ({:.4f}) with clear margins".format(
                    i, h, exact_h, r, ref_scores[i][0])
    if "peaked" in golden.name:
        assert [h[0] for h in hyps] == [r[0] for r in ref_hyps]


def test_checkpoint_ingestion_and_prefetcher_gpu(tmp_path):
    """Lightning-layout checkpoint -> CaptionRunner -> captions identical to the reference fixture;
    features fed through the pinned double-buffered prefetcher (care_amd/data.py)."""
    from care_amd.checkpoint import load_model
    from care_amd.data import FeaturePrefetcher
    from conftest import GoldenCase
    from test_next_rows_cpu import _fake_lightning_checkpoint

    golden = GoldenCase("msrvtt_care_eos_b4")
    opt, P, feats, _ = golden.build()
    path = str(tmp_path / "m.ckpt")
    _fake_lightning_checkpoint(path, {**opt, "beam_size": 5}, P, {})
    runner = load_model(path, new_opt_used_to_override={"beam_size": 1}, device="cuda:0", replace_paths=False)
    ref_hyps, _ = golden.hyps()
    seen = []
    for dev_feats in FeaturePrefetcher([feats, feats, feats], "cuda:0"):
        hyps, _ = runner.translate_step({"feats": dev_feats})
        seen.append(hyps)
    assert len(seen) == 3 and all(h == ref_hyps for h in seen)
    vocab = {i: "w%d" % i for i in range(opt["vocab_size"])}
    out = runner.translate_step({"feats": _dev(feats), "video_ids": ["video%d" % i for i in range(4)]}, vocab=vocab)
    assert out[0]["image_id"] == "video0" and isinstance(out[0]["caption"], str) and isinstance(out[0]["score"], float)


def test_checkpoint_list_decodes_as_an_ensemble_gpu(tmp_path):
    """models.load_model over a list of checkpoints (-> Wrapper.ModelEnsemble) -> EnsembleRunner -> the reference Translator's
    hypotheses over the same two models (tests/golden/ensemble), fp32 mode; and through translate_step with the union of the
    members' modalities split per member (Wrapper.py:680-693)."""
    from care_amd.checkpoint import load_model
    from conftest import EnsembleCase
    from test_next_rows_cpu import _fake_lightning_checkpoint

    case = EnsembleCase("ens_care_base_greedy_b3")
    opts, Ps, feats = case.build()
    paths = []
    for i, (o, P) in enumerate(zip(opts, Ps)):
        paths.append(str(tmp_path / ("m%d.ckpt" % i)))
        _fake_lightning_checkpoint(paths[-1], o, P, {})
    runner = load_model(paths, device="cuda:0", replace_paths=False, compute_dtype="fp32")
    hyps, scores = runner.translator.translate_batch(runner.captioner, {"feats": [_dev(f) for f in feats]})
    ref_hyps, ref_scores = case.hyps()
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    # translate_step: one tensor per modality of the union, every member picks its own
    union = runner.get_opt()["modality"]
    by_mod = {}
    for o, fl in zip(opts, feats):
        for ch, t in zip(o["modality"], fl):
            by_mod.setdefault(ch, t)
    split = runner.preprocess_batch_before_translate_step({"feats": [by_mod[ch].to("cuda:0") for ch in union]})
    if runner.need_to_split_feats:
        assert [len(fl) for fl in split["feats"]] == [len(o["modality"]) for o in opts]
    got = runner.translate_step({"feats": [by_mod[ch].to("cuda:0") for ch in union]})
    assert len(got[0]) == case.meta["batch"]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_metrics_step_gpu(golden, dtype):
    """Fused teacher-forced scoring (no logits in HBM in bf16 mode) -> word accuracy / perplexity;
    concept F1@k / mAP on the device predictions; both against the reference's own criteria."""
    from care_amd.metrics import concept_metrics, language_metrics

    opt, P, feats, ids = golden.build()
    z = golden.z
    model = _model(opt, P, dtype)
    enc = model.encoding_phase(_dev(feats))
    labels = torch.from_numpy(z["tf_labels"])
    if dtype != "fp32":   # the one-pass entry point (lean encode where nothing else reads the memory) gives the same numbers
        lp1, pr1, _ = model.engine().metrics_step(_dev(feats), ids.to("cuda:0"), labels)
        lp1, pr1 = lp1.clone(), pr1.clone()
    logp, pred = model.engine().score_teacher_forced(ids.to("cuda:0"), labels, enc["encoder_hidden_states"],
                                                     enc.get("semantic_hidden_states"), enc.get("semantic_embs"))
    if dtype != "fp32":
        assert torch.equal(pr1, pred) and (lp1 - logp).abs().max().item() < 1e-5
    m = language_metrics(logp, pred, labels)
    tol = 1e-4 if dtype == "fp32" else MODES[dtype]["ppl"]
    assert abs(m["Perplexity"] / z["metrics_lang"][1] - 1) < tol
    if dtype == "fp32":
        assert abs(m["Word Acc0"] - z["metrics_lang"][0]) < 1e-6
    if "metrics_attr" in z and (dtype == "fp32" or opt["encoder"] == "Embedder"):
        c = concept_metrics(enc["preds_attr"], torch.from_numpy(z["labels_attr"]))
        got = [c["F1-%02d" % k] for k in (5, 10, 20, 30, 40, 50)] + [c["mAP"]]
        np.testing.assert_allclose(got, z["metrics_attr"], rtol=1e-4, atol=1e-6)


def test_bench_under_torchrun_with_rccl():
    """bench.py launched the way the driver launches it (torch.distributed.run, nccl = RCCL):
    one rank on the one GPU of this box, the process group and the all-gather forced on."""
    import json
    import os
    import subprocess
    import sys

    import socket

    with socket.socket() as sk:  # a free port: two test runs may share a host
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CARE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "2", "--batch", "256", "--no-cpu-baseline", "--no-legs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
    d = line["distributed"]  # the N > 1 report: who took part, what each rank did, what the exchange costs
    assert d["rccl_ranks_seen"] == [0] and d["per_rank_captions_per_s"][0] > 0 and d["all_gather_us"] > 0


@pytest.mark.parametrize("mode", H16)
def test_teacher_forced_fast_path_bf16(golden, mode):
    """The teacher-forced forward on the decode path's kernels (engine._decode_full_fast: store GEMMs, fused dense +
    LayerNorm, one-wave-per-(sequence, head) attention): hidden states against the fp32 reference at the bf16 bars,
    and against the unfused sequence of the same mode; the fused scoring (no logits in memory) against the reference's
    own word accuracy / perplexity inputs."""
    bar = MODES[mode]
    opt, P, feats, ids = golden.build()
    z = golden.z
    model = _model(opt, P, mode)
    eng = model.engine()
    if not eng.tf_fast_ok(ids.shape[1], False):
        pytest.skip("fast teacher-forced path: bf16, d_model = 512")
    batch = {"feats": _dev(feats), "input_ids": ids.to("cuda:0")}
    fast = model.feedforward_step(batch, output_auxiliary=False)
    assert "attention_probs" not in fast
    slow = model.feedforward_step(batch)          # with the auxiliary dict: the unfused sequence
    n = z["tf_hidden_states"].shape[0]
    err = (fast["hidden_states"][:n].float().cpu() - torch.from_numpy(z["tf_hidden_states"])).abs()
    assert err.max().item() < bar["max"] and err.mean().item() < bar["mean"], (err.max().item(), err.mean().item())
    d2 = (fast["hidden_states"].float() - slow["hidden_states"].float()).abs()
    assert d2.max().item() < bar["max"] and d2.mean().item() < bar["mean"]
    lse = _maxdiff(torch.logsumexp(fast["logits"], -1), z["tf_logits_lse"])
    assert lse < (bar["lse_peaked"] if "peaked" in golden.name else 2 * bar["lse"]), lse
    _record(golden.name + "#tf_fast#" + mode, hid_max=err.max().item(), hid_mean=err.mean().item())


# ----------------------------------------------------------------------------- fp16x3 mode
# fp32 storage, every GEMM as three fp16 MFMA passes over hi/lo pieces of both operands (what is dropped is ~2^-22
# of a product): the mode between bf16 and the exact-f32 MFMA.  Bars: hidden states and concept probabilities
# within FP16X3_ATOL of the reference (measured worst over the fixtures: see gpurun_out/bf16_err.jsonl, case
# `<fixture>#fp16x3`), token ids as the reference's wherever its own margins exceed 1e-4.
FP16X3_ATOL = 1.2e-5  # north_star's fp32 bar is 1e-5; the worst measured over the 20 fixtures is 1.1e-5 (a memory row), hidden states 7.9e-6


def test_fp16x3_mode_matches_the_reference(golden):
    from care_amd import get_translator

    opt, P, feats, ids = golden.build()
    z = golden.z
    model = _model(opt, P, "fp16x3")
    assert model.engine().split3 and len(model.engine()._w3) > 5
    out = model.feedforward_step({"feats": _dev(feats), "input_ids": ids.to("cuda:0")})
    n = z["tf_hidden_states"].shape[0]
    hid = _maxdiff(out["hidden_states"][:n], z["tf_hidden_states"])
    mem = _maxdiff(out["encoder_hidden_states"][0], z["encoder_hidden_states_clip0"])
    lse = _maxdiff(torch.logsumexp(out["logits"], -1), z["tf_logits_lse"])
    rec = dict(hidden_max=hid, mem_max=mem, lse_max=lse)
    if "preds_attr" in z:
        rec["preds_max"] = _maxdiff(out["preds_attr"], z["preds_attr"])
        assert rec["preds_max"] < ATOL_FP32
        if "semantic_labels" in z:
            assert np.array_equal(out["semantic_labels"].cpu().numpy(), z["semantic_labels"])
    _record(golden.name + "#fp16x3", **rec)
    assert hid < ATOL_FP32 and mem < FP16X3_ATOL, rec  # hidden states at north_star's fp32 bar itself
    assert lse < (2e-3 if "peaked" in golden.name else 1e-4), rec
    ref_hyps, ref_scores = golden.hyps()
    hyps, scores = get_translator(opt).translate_batch([model], {"feats": _dev(feats)})
    if hyps != ref_hyps:
        # only a reference near-tie may flip: audit every differing clip against the oracle's margins
        from oracle import care_cpu
        enc = care_cpu.encoding_phase(P, opt, feats)
        inputs = care_cpu.inputs_for_decoder(opt, enc)
        assert opt.get("beam_size", 5) == 1 or get_translator(opt).beam_size == 1, "beam search differs in fp16x3 mode"
        for j, (h, r) in enumerate(zip(hyps, ref_hyps)):
            if h != r:
                _audit_greedy(P, opt, {k: v[j:j + 1] for k, v in inputs.items()}, h[0], r[0], 1e-4)
    else:
        for a, b in zip(scores, ref_scores):
            np.testing.assert_allclose(a, b, rtol=0, atol=2e-4)


@pytest.mark.parametrize("max_len", [12, 45, 64])
@pytest.mark.parametrize("config,beam", [("msrvtt_care", 1), ("msrvtt_care", 5), ("msrvtt_base_ami", 5)])
def test_other_caption_lengths_against_the_oracle(config, beam, max_len):
    """opts.py --max_len (default 30): captions of up to 11, 44 and 63 tokens - the position table, the self-attention cache, the
    beam tables and the resident launches' lane-per-position layouts (T <= 32 / T <= 63) all follow it.  fp32 mode: the
    oracle's hypotheses and scores; fp16 mode (the resident launches at 6 clips): the same winners but for near-ties."""
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict
    from oracle import care_cpu

    B = 6
    opt = make_opt(config, max_len=max_len, beam_size=beam, topk=min(beam, 2))
    model = get_framework(opt).eval()
    P = synth_state_dict(91 + max_len, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                         row_scale={"cls_head.tgt_word_prj.weight": {3: 2.5 if max_len > 30 else 4.0, 0: 3.0}})
    model.load_state_dict(P, strict=True)
    model.to("cuda:0")
    feats = synth_feats(91 + max_len, feat_shapes(opt, B))
    ref_hyps, ref_scores, gaps = care_cpu.translate_batch(P, opt, feats, return_gaps=True)
    assert max(len(h[0]) for h in ref_hyps) > min(max_len - 1, 20) // 2   # captions that do run on
    tr = get_translator(opt)
    hyps, scores = tr.translate_batch([model], {"feats": _dev(feats)})
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    model.set_compute_dtype("fp16")
    for _ in range(3):
        h16, _ = tr.translate_batch([model], {"feats": _dev(feats)})
    assert model.engine().last_decode.get("resident") == (max_len - 1 <= 63)
    for i in range(B):
        if h16[i][0] != ref_hyps[i][0]:
            g = gaps[i]
            assert min(g["select"], g["best_slack"], g["rank"]) < 1e-2, (i, g)


@pytest.mark.parametrize("name,config,over", [
    ("frames8", "msrvtt_care", dict(n_frames=8)),                       # opts.py --n_frames (MSVD runs use 8 ... 60 frames)
    ("frames36", "msrvtt_base_ami", dict(n_frames=36)),                  # 108 memory rows
    ("frames60", "msrvtt_base_ami", dict(n_frames=60)),                  # 180 memory rows: beyond every attention kernel's 128 keys - REFUSED
    ("vocab2003", "msrvtt_care", dict(vocab_size=2003)),                # a corpus of its own: V is read from info_corpus.pkl (opts.py:349)
    ("vocab20011", "msrvtt_base_ami", dict(vocab_size=20011)),          # ... beyond the resident launches' 16384 columns
    ("alpha07", "msrvtt_care", dict(beam_alpha=0.7)),                   # opts.py --beam_alpha
    ("topk30k12", "msrvtt_care", dict(use_attr_topk=12, attribute_prediction_k=300)),   # tasks.yaml:40-41 are options too
    ("d256", "msrvtt_base_ami", dict(dim_hidden=256, num_attention_heads=4, intermediate_size=1024)),   # a width outside archs.yaml
    ("layers2", "msrvtt_care", dict(num_hidden_layers_decoder=2)),
    # options of the classes on the path whose oracle restatement was checked against the reference itself (round 6, CPU):
    ("sinusoid_pe", "msrvtt_care", dict(trainable_pe=False)),           # Embeddings.py:116-119: the fixed sinusoid table
    ("no_qkv_bias", "msrvtt_base_ami", dict(mha_exclude_bias=True)),    # opts.py --mha_exclude_bias
    ("no_hybrid_bias", "msrvtt_care", dict(add_hybrid_attention_bias=False)),
    ("decoder_mi", "msrvtt_base_ami", dict(modality_for_decoder="mi")), # Encoder.py:125-138: the decoder sees two of three modalities
    ("predictor_mi", "msrvtt_care", dict(modality_for_predictor="mi")),
    ("eps1e-6", "msrvtt_base_ami", dict(layer_norm_eps=1e-6)),
    ("modality_ai", "msrvtt_base_ami", dict(modality="ai")),
    ("share_prj", "msrvtt_care", dict(attribute_prediction_share_prj=True)),
    ("retrieval10", "msrvtt_care", dict(retrieval_topk=10)),            # tasks.yaml:44: rows of the retrieval modality
    ("dims_64_1024_768", "msrvtt_base_ami", dict(dim_a=64, dim_m=1024, dim_i=768)),   # other feature extractors (feats.yaml)
    ("dim_i_500", "msvd_base_i", dict(dim_i=500)),                      # a width that is no multiple of 32
    ("vocab100", "msrvtt_base_ami", dict(vocab_size=100)),              # tiny vocabularies (below the fused selections' minimum)
    ("vocab130", "msrvtt_care", dict(vocab_size=130)),
    ("d128", "msrvtt_base_ami", dict(dim_hidden=128, num_attention_heads=2, intermediate_size=512)),
    ("d192", "msrvtt_care", dict(dim_hidden=192, num_attention_heads=3, intermediate_size=768)),
    ("d320_ff1280", "msrvtt_base_ami", dict(dim_hidden=320, num_attention_heads=5, intermediate_size=1280)),
    ("d512_ff1536", "msrvtt_care", dict(intermediate_size=1536)),       # a standard width with an FFN the resident launches do not cover
])
def test_options_outside_the_shipped_configurations_against_the_oracle(name, config, over):
    """The shapes a user's own checkpoint may have - other frame counts, vocabularies, beam_alpha, concept counts, widths and
    depths than config/*.yaml ships: fp32 mode must give the oracle's beam-5 hypotheses and scores, the 16-bit mode must run
    (whatever forms the shape admits) and agree but for near-ties - or the model must refuse LOUDLY, never answer wrongly."""
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict
    from oracle import care_cpu

    B = 5
    opt = make_opt(config, beam_size=5, topk=2, **over)
    try:
        model = get_framework(opt).eval()
        P = synth_state_dict(7, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                             row_scale={"cls_head.tgt_word_prj.weight": {3: 4.0, 0: 3.0}})
        model.load_state_dict(P, strict=True)
        model.to("cuda:0")
        feats = synth_feats(7, feat_shapes(opt, B))
        tr = get_translator(opt)
        hyps, scores = tr.translate_batch([model], {"feats": _dev(feats)})
    except (NotImplementedError, ValueError) as exc:   # a loud refusal that names what is outside the path
        assert name == "frames60" and "128 keys" in str(exc), "{}: refused with `{}`".format(name, exc)
        return
    assert name != "frames60"
    ref_hyps, ref_scores, gaps = care_cpu.translate_batch(P, opt, feats, return_gaps=True)
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    # the teacher-forced forward (Framework.py:215-237) of the same model: hidden states, logits, concept probabilities
    from care_amd.synth import synth_input_ids
    ids = synth_input_ids(7, B, opt["max_len"] - 1, opt["vocab_size"])
    with torch.no_grad():
        tf_ref = care_cpu.feedforward_step(P, opt, feats, ids)
    tf = model.feedforward_step({"feats": _dev(feats), "input_ids": ids.to("cuda:0")})
    assert _maxdiff(tf["hidden_states"], tf_ref["hidden_states"].numpy()) < ATOL_FP32
    assert _maxdiff(torch.logsumexp(tf["logits"], -1), torch.logsumexp(tf_ref["logits"], -1).numpy()) < 1e-4
    if "preds_attr" in tf_ref:
        assert _maxdiff(tf["preds_attr"], tf_ref["preds_attr"].numpy()) < 1e-5
    model.set_compute_dtype("fp16")
    for _ in range(3):
        h16, s16 = tr.translate_batch([model], {"feats": _dev(feats)})
    for i in range(B):
        if h16[i][0] != ref_hyps[i][0]:
            g = gaps[i]
            assert min(g["select"], g["best_slack"], g["rank"]) < 1e-2, (i, g)
        else:
            assert abs(s16[i][0] - ref_scores[i][0]) < 2e-2
    tf16 = model.feedforward_step({"feats": _dev(feats), "input_ids": ids.to("cuda:0")})
    assert _maxdiff(tf16["hidden_states"], tf_ref["hidden_states"].numpy()) < 1e-2
    logp, pred, _ = model.engine().metrics_step(_dev(feats), ids.to("cuda:0"), torch.roll(ids, -1, 1).to("cuda:0"))
    want = torch.log_softmax(tf_ref["logits"], -1).gather(-1, torch.roll(ids, -1, 1).unsqueeze(-1)).squeeze(-1)
    assert _maxdiff(logp.view(B, -1), want.numpy()) < 5e-2


@pytest.mark.parametrize("name", ["msrvtt_care_beam5_eos_b4", "msrvtt_base_ami_eos_b4", "msrvtt_cabase_beam5_long_b3", "msrvtt_care_g1l0_beam5_b2"])
def test_the_reference_translator_loop_over_this_module(name):
    """Only `get_framework` swapped: the reference's OWN Translator (models/Translator.py:35-133: encoding_phase ->
    prepare_inputs_for_decoder -> auto_enlarge -> per step decoding_phase(last_time_step_logits=True, decoder_rnn_hidden_states=None) ->
    log_softmax -> Beam.advance -> collect_active_part) drives this module's public API - restated here with the oracle's beam
    (misc/Decoding/Beam.py's semantics) - and must arrive at the reference's hypotheses: every tensor the loop touches (enlarged
    and index_selected encoder outputs, growing input_ids, [N, V] logits) goes through the module API, not the engine's passes."""
    from conftest import GoldenCase
    from oracle.care_cpu import HostBeam

    golden = GoldenCase(name)
    opt, P, feats, _ = golden.build()
    model = _model(opt, P)
    bm, n_best, max_len = int(opt.get("beam_size", 5)), int(opt.get("topk", 1)), int(opt["max_len"])
    with torch.no_grad():
        enc = model.encoding_phase(_dev(feats))
        inputs = model.prepare_inputs_for_decoder(enc, {"feats": _dev(feats)})
        inputs = {k: v.repeat_interleave(bm, dim=0) if isinstance(v, torch.Tensor) else v for k, v in inputs.items()}
        n = inputs["encoder_hidden_states"].shape[0] // bm
        beams = [HostBeam(bm, max_len, n_best) for _ in range(n)]
        active = list(range(n))
        for t in range(1, max_len):
            ids = torch.stack([beams[i].prefixes() for i in active]).view(-1, t).to("cuda:0")
            out = model.decoding_phase(input_ids=ids, inputs_for_decoder=inputs, decoder_rnn_hidden_states=None, last_time_step_logits=True)
            logp = torch.log_softmax(out["logits"], dim=1).view(len(active), bm, -1).cpu()
            still = [pos for pos, i in enumerate(active) if not beams[i].advance(logp[pos])]
            if not still:
                break
            if len(still) != len(active):
                sel = torch.tensor(still, device="cuda:0")
                inputs = {k: v.view(len(active), -1).index_select(0, sel).view(len(still) * bm, *v.shape[1:]) if isinstance(v, torch.Tensor) else v
                          for k, v in inputs.items()}
                active = [active[pos] for pos in still]
    hyps, scores = [], []
    for b in beams:
        ranked = b.ranked(float(opt.get("beam_alpha", 1.0)))
        n_best = min(n_best, len(ranked))
        hyps.append([b.hypothesis(t_, k) for _, t_, k, _ in ranked[:n_best]])
        scores.append([s for s, _, _, _ in ranked[:n_best]])
    ref_hyps, ref_scores = golden.hyps()
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)


@pytest.mark.parametrize("mode", ["fp32", "fp16"])
def test_feature_tensors_as_a_loader_may_hand_them_over(mode):
    """The same clips as fp32 contiguous device tensors (the fixture's form), on the CPU, as non-contiguous views, as float64,
    with extra tensors behind the modalities' (Framework.py:151-159 takes the first len(modality)), one clip at a time - the
    same captions; as 16-bit tensors (a loader that rounds its features): the captions of the rounded features."""
    from conftest import GoldenCase
    from care_amd import get_translator

    golden = GoldenCase("msrvtt_care_beam5_eos_b4")
    opt, P, feats, _ = golden.build()
    model = _model(opt, P, mode)
    tr = get_translator(opt)
    base = tr.translate_batch([model], {"feats": _dev(feats)})
    if mode == "fp32":
        assert base[0] == golden.hyps()[0]
    assert tr.translate_batch([model], {"feats": [f.clone() for f in feats]}) == base                       # host tensors
    wide = [torch.cat([f, f], dim=-1).to("cuda:0") for f in feats]
    assert tr.translate_batch([model], {"feats": [w[..., : f.shape[-1]] for w, f in zip(wide, feats)]}) == base   # strided views
    assert tr.translate_batch([model], {"feats": [f.double().to("cuda:0") for f in feats]}) == base
    assert tr.translate_batch([model], {"feats": _dev(feats) + [torch.zeros(4, 3, device="cuda:0")]}) == base
    one_by_one = [tr.translate_batch([model], {"feats": [f[i: i + 1].to("cuda:0") for f in feats]}) for i in range(4)]
    if mode == "fp32":
        assert [h[0][0] for h, _ in one_by_one] == [h[0] for h in base[0]]   # (n_best shrinks across the clips of ONE batch: winners only)
    for dt in (torch.bfloat16, torch.float16):
        rounded = [f.to(dt) for f in feats]
        as16 = tr.translate_batch([model], {"feats": [r.to("cuda:0") for r in rounded]})
        assert as16 == tr.translate_batch([model], {"feats": [r.float().to("cuda:0") for r in rounded]})
    with pytest.raises(ValueError):
        tr.translate_batch([model], {"feats": _dev(feats)[:2]})                                              # a modality is missing
    with pytest.raises((ValueError, RuntimeError)):
        tr.translate_batch([model], {"feats": [f[:, :5].to("cuda:0") for f in feats]})                       # 5 frames instead of 28
    with pytest.raises(ValueError, match="without clips"):
        tr.translate_batch([model], {"feats": [f[:0].to("cuda:0") for f in feats]})                          # an empty batch


def test_copies_and_pickles_of_a_used_module_decode_like_the_original():
    """copy.deepcopy(model) and torch.save(model) / torch.load of a module whose engine already ran (captured graphs, device
    workspaces): the engine is no part of the module's state - a copy builds its own; in-place edits of the parameters are seen
    by the next call."""
    import copy
    import io
    from conftest import GoldenCase
    from care_amd import get_translator

    golden = GoldenCase("msrvtt_care_beam5_eos_b4")
    opt, P, feats, _ = golden.build()
    model = _model(opt, P)
    tr = get_translator(opt)
    dev = {"feats": _dev(feats)}
    for _ in range(3):
        base = tr.translate_batch([model], dev)
    assert base[0] == golden.hyps()[0]
    twin = copy.deepcopy(model)
    assert twin._engine is None and model._engine is not None
    assert tr.translate_batch([twin], dev) == base == tr.translate_batch([model], dev)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    assert tr.translate_batch([torch.load(buf, weights_only=False)], dev) == base
    with torch.no_grad():
        model.cls_head.tgt_word_prj.weight[3] += 5.0     # the EOS row up: shorter captions
    after = tr.translate_batch([model], dev)
    assert after != base and tr.translate_batch([twin], dev) == base
    fresh = _model(opt, {k: v.detach().cpu() for k, v in model.state_dict().items()})
    assert after == tr.translate_batch([fresh], dev)


def test_ensemble_of_models_of_different_widths_against_the_oracle():
    """Members need one vocabulary and one max_len, not one architecture: a d_model 512 CARE model, a d_model 1024 one and a Base
    model (other modalities: a feature list per member) decode together - against the oracle's ensemble search (pinned on the
    reference Translator by tests/golden/ensemble), fp32 mode; and with the members in DIFFERENT compute modes."""
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict
    from oracle import care_cpu

    B = 3
    opts = [make_opt("msrvtt_care", beam_size=5, topk=2), make_opt("vatex_care_large"), make_opt("msrvtt_base_ami")]
    models, Ps, feats = [], [], []
    for i, o in enumerate(opts):
        m = get_framework(o).eval()
        P = synth_state_dict(40 + i, [(k, tuple(v.shape)) for k, v in m.state_dict().items()],
                             row_scale={"cls_head.tgt_word_prj.weight": {3: 4.0, 0: 3.0}})
        m.load_state_dict(P, strict=True)
        models.append(m.to("cuda:0")); Ps.append(P); feats.append(synth_feats(40 + i, feat_shapes(o, B)))
    ref_hyps, ref_scores, gaps = care_cpu.translate_batch_ensemble(Ps, opts, feats, return_gaps=True)
    tr = get_translator(opts[0])
    batch = {"feats": [_dev(f) for f in feats]}
    hyps, scores = tr.translate_batch(models, batch)
    assert hyps == ref_hyps
    for a, b in zip(scores, ref_scores):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    models[1].set_compute_dtype("fp16")
    models[2].set_compute_dtype("bf16")
    for _ in range(3):
        mixed, _ = tr.translate_batch(models, batch)
    for i in range(B):
        if mixed[i][0] != ref_hyps[i][0]:
            g = gaps[i]
            assert min(g["select"], g["best_slack"], g["rank"]) < 5e-2, (i, g)


@pytest.mark.parametrize("mode", ["fp32", "fp16"])
def test_side_streams_and_worker_threads(mode):
    """A serving process: the call under another current stream, from a worker thread, first used on a worker thread (graph
    capture off the main thread), and two threads with a model each at the same time (two resident launches competing for
    the chip in fp16 mode: a launch that times out is decoded again by the multi-launch pass) - always the same captions."""
    import threading
    from conftest import GoldenCase
    from care_amd import get_translator

    golden = GoldenCase("msrvtt_care_beam5_eos_b4")
    opt, P, feats, _ = golden.build()
    model, other = _model(opt, P, mode), _model(opt, P, mode)
    tr, tr2 = get_translator(opt), get_translator(opt)
    dev = {"feats": _dev(feats)}
    for _ in range(3):
        base = tr.translate_batch([model], dev)
    if mode == "fp32":
        assert base[0] == golden.hyps()[0]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(3):
            assert tr.translate_batch([model], dev) == base
    torch.cuda.synchronize()
    assert tr.translate_batch([model], dev) == base

    def run(box, m, t, n):
        try:
            for _ in range(n):
                box["r"] = t.translate_batch([m], dev)
        except Exception as exc:   # noqa: BLE001 - reported below
            box["e"] = repr(exc)

    box = {}
    th = threading.Thread(target=run, args=(box, other, tr2, 4))   # `other` has never run: its first passes happen on this thread
    th.start(); th.join()
    assert box.get("e") is None and box["r"] == base
    boxes = [{}, {}]
    ths = [threading.Thread(target=run, args=(boxes[0], model, tr, 20)), threading.Thread(target=run, args=(boxes[1], other, tr2, 20))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert [b.get("e") for b in boxes] == [None, None] and boxes[0]["r"] == base and boxes[1]["r"] == base
    # two COLD models started on two threads at the same moment: their first passes and graph captures coincide (captures are
    # serialised process-wide, engine._CAPTURE_LOCK: two at once abort the process inside torch's generator registry)
    cold = [_model(opt, P, mode), _model(opt, P, mode)]
    trs = [get_translator(opt), get_translator(opt)]
    boxes = [{}, {}]
    ths = [threading.Thread(target=run, args=(boxes[i], cold[i], trs[i], 12)) for i in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert [b.get("e") for b in boxes] == [None, None] and boxes[0]["r"] == base and boxes[1]["r"] == base


@pytest.mark.parametrize("mode", ["fp32", "fp16"])
def test_threads_sharing_one_module_take_turns(mode):
    """A thread pool in front of ONE module (an nn.Module in eval mode looks stateless to its callers): four threads, each with
    clips of its own, 25 calls each through one shared Translator and through the module API - an engine's workspaces, result
    block and graphs are shared state, so calls take turns (engine.lock) and every thread gets ITS captions."""
    import threading
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict

    opt = make_opt("msrvtt_care", beam_size=5, topk=1)
    model = get_framework(opt).eval()
    model.load_state_dict(synth_state_dict(5, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                                           row_scale={"cls_head.tgt_word_prj.weight": {3: 5.0}}), strict=True)
    model.set_compute_dtype(mode)
    model.to("cuda:0")
    tr = get_translator(opt)
    inputs = [{"feats": _dev(synth_feats(100 + i, feat_shapes(opt, 6 + i)))} for i in range(4)]
    want = [tr.translate_batch([model], b, use_graph=False) for b in inputs]
    want_mem = [model.encoding_phase(b["feats"])["encoder_hidden_states"].clone() for b in inputs]
    assert len({str(w[0]) for w in want}) == 4
    errors = []

    def run(i):
        try:
            for k in range(25):
                if tr.translate_batch([model], inputs[i]) != want[i]:
                    errors.append("thread {} call {}: another thread's captions".format(i, k))
                if not torch.equal(model.encoding_phase(inputs[i]["feats"])["encoder_hidden_states"], want_mem[i]):
                    errors.append("thread {} call {}: another thread's memory".format(i, k))
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    ths = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors[:3]
