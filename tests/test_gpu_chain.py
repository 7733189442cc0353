"""GPU: the chained beam step (csrc/decode_chain.hip, care_decode_chain_beam: the phases of the resident beam launch as
kernels of their own - models/Translator.py:77-143, misc/Decoding/Beam.py:45-85) against the resident launch (the SAME
arithmetic per row: identical bits asked), the multi-launch search and the CPU oracle.  The golden beam fixtures run
through it in tests/test_gpu_parity.py (form `chain`)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_properties import PEAKED_ROWS, _setup  # noqa: E402
from test_gpu_resident_beam import _beam, _best  # noqa: E402


def _chain(eng, feats, bm=5, need=5, **kw):
    keep = eng.resident_beam_max_rows, eng.chain_beam_max_rows
    eng.resident_beam_max_rows, eng.chain_beam_max_rows = 0, 4096
    try:
        out = _beam(eng, feats, bm, need, **kw)
        assert eng.last_decode.get("chain"), "the pass did not take the chained step"
    finally:
        eng.resident_beam_max_rows, eng.chain_beam_max_rows = keep
    return out


# rows = clips x 5: one row tile (5), K-split forms (60), one row tile per workgroup (65, 255), several tiles per
# weight fetch (260, 640), both 16-bit modes
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("config,B", [("msrvtt_care", 1), ("msrvtt_base_ami", 12), ("msrvtt_cabase", 13), ("msrvtt_care", 51),
                                      ("msrvtt_care", 52), ("msrvtt_base_ami", 128)])
def test_chain_is_bit_identical_to_the_resident_launch(config, B, mode):
    """The chain's kernels ARE the resident launch's phases (plain instead of agent-scope accesses): finished lists,
    lengths and scores must be identical, bit for bit - on a model whose clips end at mixed steps (EOS / PAD rows
    boosted) so that finished lists, frozen clips and the early exit are all exercised."""
    boost = {"cls_head.tgt_word_prj.weight": {3: 14.0, 0: 3.0}}
    opt, P, model, feats = _setup(config, B, mode, boost=boost)
    eng = model.engine()
    eng.resident_max_rows = 256
    eng.resident_beam_max_rows = 640
    rs = _beam(eng, feats, use_graph=False)
    assert eng.last_decode.get("resident")
    ch = _chain(eng, feats, use_graph=False)
    # finished lists, lengths and tokens: identical.  Scores: identical up to the ORDER in which a row's log-sum-exp partials
    # are merged - one partial per vocabulary part, and the two forms cut the vocabulary into a different number of parts at
    # some row counts (the resident grid is the CU count, the chain's is sized by the work): 1e-6 of a log-probability
    assert torch.equal(rs[0], ch[0]) and torch.equal(rs[2], ch[2]) and torch.equal(rs[3], ch[3])
    assert (rs[1] - ch[1]).abs().max().item() < 2e-4
    # every form of the chain (K-split items / one row tile / several row tiles per weight fetch): the same bits
    import os
    for form in ("0", "1", "3"):
        os.environ["CARE_CHAIN_FORM"] = form
        try:
            alt = _chain(eng, feats, use_graph=False)
        finally:
            del os.environ["CARE_CHAIN_FORM"]
        # (the forms differ in the row tiles per vocabulary part: the log-sum-exp partials merge in another order - scores only)
        assert torch.equal(rs[0], alt[0]) and torch.equal(rs[2], alt[2]) and torch.equal(rs[3], alt[3]), form
        assert (rs[1] - alt[1]).abs().max().item() < 2e-4, form


@pytest.mark.parametrize("config,B", [("msrvtt_care", 30), ("msrvtt_base_ami", 200)])
def test_chain_is_deterministic_replayable_segmented_and_batch_independent(config, B):
    """Run after run, eager or replayed from the captured segment graphs: identical finished lists; segments of 1, 8 or
    all 29 steps; early exit == the fixed-length pass; a clip's hypotheses do not depend on the batch it rides in."""
    boost = {"cls_head.tgt_word_prj.weight": {3: 14.0, 0: 3.0}}
    opt, P, model, feats = _setup(config, B, "bf16", boost=boost)
    eng = model.engine()
    eng.resident_max_rows = 256
    full = _chain(eng, feats, use_graph=False)
    steps = int(eng.last_decode["steps"])
    assert int(full[0].min()) >= 1 and steps <= eng.T
    for it in range(3):  # first sight, capture, replay
        again = _chain(eng, feats, use_graph=True)
        for a, b in zip(full, again):
            assert torch.equal(a, b)
    assert any(isinstance(v, tuple) for k, v in eng._graphs.items() if k[0] == "bchain"), "no segment was captured"
    for seg in (1, 5):
        eng.chain_segment_steps = seg
        other = _chain(eng, feats, use_graph=False)
        assert int(eng.last_decode["steps"]) <= steps + seg
        for a, b in zip(full, other):
            assert torch.equal(a, b)
    eng.chain_segment_steps = 8
    fixed = _chain(eng, feats, use_graph=False, early_exit=False)
    assert int(eng.last_decode["steps"]) == eng.T
    for a, b in zip(full, fixed):
        assert torch.equal(a, b)
    for lo, n in ((0, 1), (3, 12), (B - 7, 7), (B // 2, 13)):
        sub = [f[lo:lo + n].contiguous() for f in feats]
        part = _chain(eng, sub, use_graph=False)
        assert torch.equal(part[0], full[0][lo:lo + n])
        for i in range(n):
            k = int(part[0][i])
            assert torch.equal(part[2][i, :k], full[2][lo + i, :k])
            assert torch.equal(part[3][i, :k], full[3][lo + i, :k])
            assert (part[1][i, :k] - full[1][lo + i, :k]).abs().max().item() < 2e-4


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("config,B", [("msrvtt_care", 256), ("msrvtt_base_ami", 819)])
def test_chain_beyond_the_resident_rows_against_multi_launch_and_oracle(config, B, mode):
    """1280 / 4095 rows - beyond what one resident launch holds: the chained search and the multi-launch search must report
    the same winner wherever the oracle's search is decided by clear margins and nearly always otherwise; a sample of
    clips is audited against the oracle (its winner, or a near-tie under exact scoring)."""
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN, MODES

    bar = MODES[mode]
    opt, P, model, feats = _setup(config, B, mode, seed=189, boost=PEAKED_ROWS)
    opt = dict(opt, beam_size=5)
    eng = model.engine()
    eng.resident_max_rows = 256
    eng.chain_beam_max_rows = 0
    eng.resident_beam_max_rows = 0
    ml = _beam(eng, feats, use_graph=False)
    assert not eng.last_decode.get("chain") and not eng.last_decode.get("resident")
    eng.chain_beam_max_rows = 4096
    assert eng.chain_beam_ok(B, 5, 5) and not eng.resident_beam_ok(B, 5, 5)
    ch = _beam(eng, feats, use_graph=False)
    assert eng.last_decode.get("chain")
    same = 0
    for i in range(B):
        (ha, sa), (hb, sb) = _best(*ch, i), _best(*ml, i)
        if ha == hb:
            same += 1
            assert abs(sa - sb) < bar["score"]
    assert same >= B - max(1, B // 16), "{} of {} winners differ between the two forms".format(B - same, B)
    idx = sorted(set(int(i) for i in torch.linspace(0, B - 1, 8).round().tolist()))
    sample = [f[idx].cpu() for f in feats]
    hyps, scores, gaps = care_cpu.translate_batch(P, opt, sample, return_gaps=True)
    inputs = care_cpu.inputs_for_decoder(opt, care_cpu.encoding_phase(P, opt, sample))
    ref_same = 0
    for j, i in enumerate(idx):
        h, s = _best(*ch, i)
        one = {k: v[j:j + 1] for k, v in inputs.items()}
        exact = care_cpu.score_hypothesis(P, opt, one, h)
        assert abs(s - exact) < bar["lse_peaked"], (i, s, exact)  # the reported score IS the hypothesis' exact score
        r = hyps[j][0]
        ref_same += int(h == r)
        if h == r or h == _best(*ml, i)[0]:
            continue  # the reference's winner, or the winner of the multi-launch search of this mode (audited on its own)
        # a winner of its own: a near-tie under exact scoring, a reference winner that was close to being pruned or
        # overtaken, or a hypothesis the reference's search pruned that scores BETTER under exact scoring (beam search is
        # not exact: which of two candidates at the pruning edge survives is decided at 16-bit noise level, and the
        # stored margins of the reference only follow ITS winner)
        assert (abs(exact - scores[j][0]) < bar["beam_tie"] or gaps[j]["best_slack"] < bar["beam_tie"] or
                gaps[j]["rank"] < bar["beam_tie"] or exact > scores[j][0]), (i, h, exact, r, scores[j][0])
    assert ref_same >= len(idx) - 2, "{} of {} sampled winners are the reference's".format(ref_same, len(idx))


def test_chain_covers_topk_above_beam_size_and_small_beams():
    opt, P, model, feats = _setup("msrvtt_care", 9, "bf16", seed=189, boost=PEAKED_ROWS)
    eng = model.engine()
    eng.resident_max_rows = 256
    for bm, need in ((5, 8), (2, 2), (3, 4), (4, 4)):
        eng.resident_beam_max_rows = 640
        rs = _beam(eng, feats, bm, need, use_graph=False)
        assert eng.last_decode.get("resident")
        ch = _chain(eng, feats, bm, need, use_graph=False)
        assert torch.equal(rs[0], ch[0]) and torch.equal(rs[2], ch[2]) and torch.equal(rs[3], ch[3]), (bm, need)
        assert (rs[1] - ch[1]).abs().max().item() < 2e-4


def test_chain_shape_rules():
    from care_amd.configs import make_opt
    from care_amd.engine import HipEngine

    e = HipEngine(make_opt("msrvtt_care"), "bf16")
    assert not e.chain_beam_ok(128, 5, 5)  # off by default (slower than the resident / multi-launch forms)
    e.chain_beam_max_rows = 4096
    assert e.chain_beam_ok(128, 5, 5) and e.chain_beam_ok(819, 5, 5) and e.chain_beam_ok(1, 2, 2)
    assert not e.chain_beam_ok(820, 5, 5) and not e.chain_beam_ok(8, 6, 6) and not e.chain_beam_ok(8, 1, 1)
    e.chain_beam_max_rows = 0
    assert not e.chain_beam_ok(1, 5, 5)
    for dtype, ok in (("fp16", True), ("fp32", False)):
        e = HipEngine(make_opt("msrvtt_care"), dtype)
        e.chain_beam_max_rows = 4096
        assert e.chain_beam_ok(128, 5, 5) == ok
