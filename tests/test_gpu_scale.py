"""GPU: 16-bit agreement at the size of the MSRVTT test split (north_star: "greedy captions on MSRVTT-test
identical to the reference").  2990 clips (the split the reference's notebooks decode: notebooks/retrieval_robustness.ipynb:188)
of the peaked CARE model - a softmax as peaked as a trained model's (oracle/gen_golden.py PEAKED) - go through the drop-in
Translator at translate.py's batch of 128 (translate.py:137: the resident launches), greedy and beam 5, in both 16-bit
modes; the captions are compared with the CPU oracle's (greedy: all 2990; beam 5: all 2990 against the engine's fp32 mode,
which a 128-clip oracle sample pins), the counts go to gpurun_out/audit.jsonl, and every differing clip must be a near-tie
of the REFERENCE's own distribution.  No real MSRVTT features or checkpoints exist offline: synthetic features, seeded weights.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_CLIPS, BATCH, SEED = 2990, 128, 373
_CACHE = {}


def _setup():
    from care_amd import get_framework
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_state_dict
    from test_gpu_properties import PEAKED_ROWS

    if "model" not in _CACHE:
        opt = make_opt("msrvtt_care")
        model = get_framework(opt).eval()
        P = synth_state_dict(SEED, [(k, tuple(v.shape)) for k, v in model.state_dict().items()], row_scale=PEAKED_ROWS)
        model.load_state_dict(P, strict=True)
        model.to("cuda:0")
        gen = torch.Generator().manual_seed(SEED)
        feats = [torch.randn(s, generator=gen) for s in feat_shapes(opt, N_CLIPS)]
        _CACHE.update(opt=opt, P=P, model=model, feats=feats)
    return _CACHE["opt"], _CACHE["P"], _CACHE["model"], _CACHE["feats"]


def _batches(feats):
    for lo in range(0, N_CLIPS, BATCH):
        yield {"feats": [f[lo: lo + BATCH].to("cuda:0") for f in feats]}


def _translate(model, opt, mode, beam):
    """All 2990 clips through the drop-in API (pipelined entry), batches of 128: (hyps, scores) per clip."""
    from care_amd import get_translator

    model.set_compute_dtype(mode)
    tr = get_translator(dict(opt, beam_size=beam, topk=1))
    hyps, scores = [], []
    for h, s in tr.translate_batches([model], _batches(_CACHE["feats"])):
        hyps += h
        scores += s
    assert len(hyps) == N_CLIPS
    if mode != "fp32":
        assert model.engine().last_decode.get("resident"), "batch 128 is the resident launches' operating point"
    return hyps, scores


# Concept selection (pred_attribute.py:262-264: the top-k of sigmoid probabilities, SORTED - the rank is the concept's position
# embedding) has ties of its own: of these 2990 clips 21 have two neighbours of the sorted top-(k + 1) probabilities within 1e-6
# of each other and 6 within 3e-7 - five fp32 ulps at 0.76 .. 0.97 (*measured* on the oracle, round 6).  The probabilities come
# from a sum of 1536 fp32 products, so ANY other summation order - torch's CPU blocking, the exact-f32 matrix cores' K loop, the
# small batches' K ranges side by side (engine_encode._concept_linear) - may swap such a pair, and with it the semantic memory
# the whole caption is decoded against.  A clip whose oracle concepts have such a pair is explained by it.
CONCEPT_TIE = 2e-6


def _concept_gap(P, opt, one) -> float:
    from oracle import care_cpu

    p = care_cpu.encoding_phase(P, opt, one)["preds_attr"]
    top, _ = p.topk(opt["use_attr_topk"] + 1, dim=1)
    return float((top[:, :-1] - top[:, 1:]).min())


def _oracle_greedy():
    """The CPU oracle's greedy captions, scores and margins of all 2990 clips (once per session: ~25 s on 16 host threads)."""
    from oracle import care_cpu

    if "oracle_greedy" not in _CACHE:
        opt, P, _, feats = _setup()
        torch.set_num_threads(16)
        hyps, scores, gaps = [], [], []
        for lo in range(0, N_CLIPS, 256):
            h, s, g = care_cpu.translate_batch(P, dict(opt, beam_size=1), [f[lo: lo + 256] for f in feats], return_gaps=True)
            hyps += h
            scores += s
            gaps += g
        _CACHE["oracle_greedy"] = (hyps, scores, gaps)
    return _CACHE["oracle_greedy"]


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_greedy_captions_of_2990_clips_against_the_oracle(mode):
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN, _audit_greedy
    from test_gpu_properties import _audit_record

    opt, P, model, feats = _setup()
    ref, ref_scores, gaps = _oracle_greedy()
    assert len({len(r[0]) for r in ref}) > 5                         # a model that ends its captions at mixed lengths
    got, _ = _translate(model, opt, mode, 1)
    differ = [i for i in range(N_CLIPS) if got[i][0] != ref[i][0]]
    clear = sum(1 for g in gaps if g["select"] >= CLEAR_MARGIN)
    tie_tol = 5e-2 if mode == "bf16" else 1e-2                       # (peaked rows scale the logit noise: test_gpu_properties)
    concept_ties = []
    for i in differ:
        one = [f[i: i + 1] for f in feats]
        if _concept_gap(P, opt, one) < CONCEPT_TIE:   # (see CONCEPT_TIE: another memory, not another decision over the same one)
            concept_ties.append(i)
            continue
        assert gaps[i]["select"] < CLEAR_MARGIN, "clip {}: every reference step decided by >= {} but the {} ids differ".format(
            i, CLEAR_MARGIN, mode)
        inputs = care_cpu.inputs_for_decoder(opt, care_cpu.encoding_phase(P, opt, one))
        _audit_greedy(P, opt, inputs, got[i][0], ref[i][0], tie_tol)
    same = N_CLIPS - len(differ)
    _audit_record(test="msrvtt_test_scale_greedy", mode=mode, clips=N_CLIPS, identical=same, clear_margin_clips=clear,
                  differing=differ[:32], concept_rank_ties=concept_ties)
    # fp16: >= 99 % of the captions are the reference's; bf16 (8 significand bits): >= 96 %
    assert same >= (0.99 if mode == "fp16" else 0.96) * N_CLIPS, "{}: {} of {} greedy captions identical".format(mode, same, N_CLIPS)


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_beam5_winners_of_2990_clips(mode):
    """Beam 5 (translate.py:144's default) at the same scale.  The CPU oracle runs beam search at ~15 captions/s, so the
    reference for all 2990 clips is the engine's fp32 mode - itself held to the oracle here on the first 128 clips (identical
    winners, scores within 1e-4), as it is on every fixture; a differing 16-bit winner must, in the ORACLE's own
    search of that clip, show a near-tie the 16-bit noise can flip (see the loop below)."""
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN
    from test_gpu_properties import _audit_record

    opt, P, model, feats = _setup()
    if "fp32_beam" not in _CACHE:
        ref, ref_scores = _translate(model, opt, "fp32", 5)
        torch.set_num_threads(16)
        o_hyps, o_scores = care_cpu.translate_batch(P, dict(opt, beam_size=5, topk=1), [f[:128] for f in feats])
        assert [h[0] for h in o_hyps] == [h[0] for h in ref[:128]]
        assert max(abs(a[0] - b[0]) for a, b in zip(o_scores, ref_scores[:128])) < 1e-4
        _CACHE["fp32_beam"] = (ref, ref_scores)
    ref, ref_scores = _CACHE["fp32_beam"]
    got, got_scores = _translate(model, opt, mode, 5)
    differ = [i for i in range(N_CLIPS) if got[i][0] != ref[i][0]]
    # fp16: the greedy test's tie tolerance.  bf16: CLEAR_MARGIN - the same rule as for greedy decoding (a clip may differ only
    # if some decision of the reference search was closer than 0.1); between 0.05 and 0.1 a SINGLE bf16 step does not flip, but
    # a beam's decisions compare sums of several steps' log-probabilities of two hypotheses, whose errors add (*measured*: the
    # one clip of 2990 that needed more than 0.05 had select 0.058, rank 0.092)
    tol = CLEAR_MARGIN if mode == "bf16" else 1e-2
    better, concept_ties = 0, []
    for i in differ[:48]:  # (a full oracle search + two exact rescorings per clip)
        # A beam search is path dependent: a flip at ANY near-tie of the reference search - the beam_size-th against the next
        # candidate of a step (`select`), the winner's ancestry against pruning (`best_slack`), the finished list's order
        # (`rank`) - can change the winner, for better or worse.  So a differing clip must show such a near-tie in the ORACLE's
        # own search of that clip, or score (exactly) within the tolerance of the reference winner.
        one = [f[i: i + 1] for f in feats]
        if _concept_gap(P, opt, one) < CONCEPT_TIE:   # (see CONCEPT_TIE; also where the fp32 mode and the oracle may part)
            concept_ties.append(i)
            continue
        o_hyps, o_scores, gaps = care_cpu.translate_batch(P, dict(opt, beam_size=5, topk=1), one, return_gaps=True)
        assert o_hyps[0][0] == ref[i][0], "clip {}: the fp32-mode winner is not the oracle's".format(i)
        inputs = care_cpu.inputs_for_decoder(opt, care_cpu.encoding_phase(P, opt, one))
        mine, theirs = (care_cpu.score_hypothesis(P, opt, inputs, h) for h in (got[i][0], ref[i][0]))
        better += mine > theirs
        g = gaps[0]
        assert abs(mine - theirs) < tol or min(g["select"], g["best_slack"], g["rank"]) < tol, \
            "clip {}: the {} winner scores {:.4f}, the reference's {:.4f}, and the reference search has no near-tie ({})".format(
                i, mode, mine, theirs, g)
    same = N_CLIPS - len(differ)
    _audit_record(test="msrvtt_test_scale_beam5", mode=mode, clips=N_CLIPS, identical=same, differing=differ[:32],
                  audited=min(len(differ), 48), audited_with_a_better_exact_score=int(better), concept_rank_ties=concept_ties)
    assert same >= (0.98 if mode == "fp16" else 0.93) * N_CLIPS, "{}: {} of {} beam winners identical".format(mode, same, N_CLIPS)


LARGE_CLIPS, LARGE_SEED = 1024, 189


@pytest.mark.parametrize("config,LARGE_BATCH,floor", [("vatex_care_large", 32, 0.97), ("msvd_base_i", 128, 0.98),
                                                      ("msrvtt_base_ami", 128, 0.98), ("care_median_gelu", 40, 0.97)])
def test_greedy_captions_of_the_other_models_against_the_oracle(config, LARGE_BATCH, floor):
    """The same question for the d_model 1024 model (`vatex_care_large`: 16 heads, ff 4096, BASELINE configs[3]) at its share of
    translate.py's batch per GPU (32 clips: the resident launch's K-split forms) and for the one-modality model without a concept
    head (`msvd_base_i`, BASELINE configs[0]; 28 memory rows), the headline model (`msrvtt_base_ami`) and the d_model 768 GELU model: 1024 clips of the peaked model, fp16 mode, greedy, through the
    drop-in Translator, against the CPU oracle's captions; every differing clip a near-tie of the reference's own distribution
    (or of its concept ranks), the counts in gpurun_out/audit.jsonl."""
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_state_dict
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN, _audit_greedy
    from test_gpu_properties import PEAKED_ROWS, _audit_record

    opt = make_opt(config)
    model = get_framework(opt).eval()
    P = synth_state_dict(LARGE_SEED, [(k, tuple(v.shape)) for k, v in model.state_dict().items()], row_scale=PEAKED_ROWS)
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype("fp16")
    model.to("cuda:0")
    gen = torch.Generator().manual_seed(LARGE_SEED)
    feats = [torch.randn(s, generator=gen) for s in feat_shapes(opt, LARGE_CLIPS)]
    tr = get_translator(dict(opt, beam_size=1, topk=1))
    got = []
    batches = ({"feats": [f[lo: lo + LARGE_BATCH].to("cuda:0") for f in feats]} for lo in range(0, LARGE_CLIPS, LARGE_BATCH))
    for h, _ in tr.translate_batches([model], batches):
        got += h
    assert len(got) == LARGE_CLIPS
    torch.set_num_threads(16)
    ref, gaps = [], []
    for lo in range(0, LARGE_CLIPS, 128):
        h, _, g = care_cpu.translate_batch(P, dict(opt, beam_size=1), [f[lo: lo + 128] for f in feats], return_gaps=True)
        ref += h
        gaps += g
    differ = [i for i in range(LARGE_CLIPS) if got[i][0] != ref[i][0]]
    concept_ties = []
    for i in differ:
        one = [f[i: i + 1] for f in feats]
        if opt.get("use_attr") and _concept_gap(P, opt, one) < CONCEPT_TIE:
            concept_ties.append(i)
            continue
        assert gaps[i]["select"] < CLEAR_MARGIN, "clip {}: every reference step decided by >= {} but the fp16 ids differ".format(i, CLEAR_MARGIN)
        inputs = care_cpu.inputs_for_decoder(opt, care_cpu.encoding_phase(P, opt, one))
        _audit_greedy(P, opt, inputs, got[i][0], ref[i][0], 1e-2)
    same = LARGE_CLIPS - len(differ)
    _audit_record(test="other_models_scale_greedy", mode="fp16", config=config, clips=LARGE_CLIPS, identical=same,
                  clear_margin_clips=sum(1 for g in gaps if g["select"] >= CLEAR_MARGIN), differing=differ[:32], concept_rank_ties=concept_ties)
    # *measured* round 6, vatex_care_large: 1003 of 1024 identical, all 21 others audited near-ties - on a model where only 365 of
    # the 1024 reference searches have every step decided by >= 0.1 (it never emits EOS: 29 decisions per clip)
    assert same >= floor * LARGE_CLIPS, "{} of {} greedy captions identical".format(same, LARGE_CLIPS)


def test_beam5_winners_of_the_large_model():
    """Beam 5 of the d_model 1024 model (translate.py's default decode at BASELINE configs[3]'s 32 clips per GPU: the resident beam
    launch's K-split forms), 512 clips through the pipelined Translator over recycled device buffers: the fp16 winners against
    the engine's fp32 mode (multi-launch, exact f32), a sample of the differing clips audited in the ORACLE's own search."""
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_state_dict
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN
    from test_gpu_properties import PEAKED_ROWS, _audit_record

    clips, batch = 512, 32
    opt = make_opt("vatex_care_large")
    model = get_framework(opt).eval()
    P = synth_state_dict(LARGE_SEED, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                         row_scale={"cls_head.tgt_word_prj.weight": {**PEAKED_ROWS["cls_head.tgt_word_prj.weight"], 3: 30.0}})
    model.load_state_dict(P, strict=True)
    model.to("cuda:0")
    gen = torch.Generator().manual_seed(LARGE_SEED)
    feats = [torch.randn(s, generator=gen) for s in feat_shapes(opt, clips)]
    tr = get_translator(dict(opt, beam_size=5, topk=1))
    res = {}
    for mode in ("fp32", "fp16"):
        model.set_compute_dtype(mode)
        got = []
        batches = ({"feats": [f[lo: lo + batch].to("cuda:0") for f in feats]} for lo in range(0, clips, batch))
        for h, _ in tr.translate_batches([model], batches):
            got += h
        res[mode] = got
        if mode == "fp16":
            assert model.engine().last_decode.get("resident")
    differ = [i for i in range(clips) if res["fp16"][i][0] != res["fp32"][i][0]]
    torch.set_num_threads(16)
    for i in differ[:6]:
        one = [f[i: i + 1] for f in feats]
        if _concept_gap(P, opt, one) < CONCEPT_TIE:
            continue
        o_hyps, _, gaps = care_cpu.translate_batch(P, dict(opt, beam_size=5, topk=1), one, return_gaps=True)
        assert o_hyps[0][0] == res["fp32"][i][0], "clip {}: the fp32-mode winner is not the oracle's".format(i)
        g = gaps[0]
        assert min(g["select"], g["best_slack"], g["rank"]) < CLEAR_MARGIN, (i, g)
    same = clips - len(differ)
    _audit_record(test="large_model_scale_beam5", mode="fp16", config="vatex_care_large", clips=clips, identical=same, differing=differ[:32],
                  mean_length=sum(len(h[0]) for h in res["fp32"]) / clips)
    assert same >= 0.93 * clips, "{} of {} beam winners identical".format(same, clips)


@pytest.mark.parametrize("config,B,beam", [("vatex_care_large", 32, 5), ("msrvtt_care", 128, 1), ("care_median_gelu", 40, 1)])
def test_runner_over_the_prefetcher_equals_eager_passes(config, B, beam):
    """translate.py's loop as this repository offers it: CaptionRunner.translate_steps over FeaturePrefetcher(depth=3) - pinned
    staging, H2D on a side stream into three rotating device slots, the pipelined Translator, hipGraph replays per slot - over
    14 batches (the last one ragged), against one eager pass per batch."""
    from care_amd.checkpoint import CaptionRunner
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.data import FeaturePrefetcher
    from care_amd.synth import synth_state_dict
    from test_gpu_properties import PEAKED_ROWS

    opt = make_opt(config, beam_size=beam, topk=1)
    runner = CaptionRunner(opt)
    P = synth_state_dict(LARGE_SEED, [(k, tuple(v.shape)) for k, v in runner.captioner.state_dict().items()], row_scale=PEAKED_ROWS)
    runner.captioner.load_state_dict(P, strict=True)
    runner.captioner.set_compute_dtype("fp16")
    runner.eval().to("cuda:0")
    n = 13 * B + B // 3
    gen = torch.Generator().manual_seed(LARGE_SEED + 1)
    host = [torch.randn(s, generator=gen) for s in feat_shapes(opt, n)]
    batches = [[f[lo: lo + B] for f in host] for lo in range(0, n, B)]
    got = []
    for hyps, _ in runner.translate_steps(({"feats": f} for f in FeaturePrefetcher(batches, "cuda:0", depth=3))):
        got += hyps
    assert len(got) == n
    eng = runner.captioner.engine()
    want = []
    for b in batches:
        dev = [f.to("cuda:0") for f in b]
        if beam == 1:
            _, fed, length, _ = eng.translate_greedy(dev, use_graph=False, lean=True)
            want += [[fed[i, 1: int(length[i]) + 1].tolist()] for i in range(len(length))]
        else:
            hyps, _ = runner.translator.translate_batch([runner.captioner], {"feats": dev}, use_graph=False)
            want += hyps
    assert got == want
