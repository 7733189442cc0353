"""GPU: the resident decode of small batches (csrc/decode_resident.hip, care_decode_resident) - the whole greedy step loop
of Translator.translate_batch (models/Translator.py:77-143) as one launch - against the multi-launch decode, the CPU
oracle, and its own invariants (rows independent of the batch they ride in, replay == eager, early exit == fixed length).
The golden fixtures run through it in tests/test_gpu_parity.py (form `resident`)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_properties import PEAKED_ROWS, _setup  # noqa: E402


def _run(eng, feats, **kw):
    _, fed, length, score = eng.translate_greedy(feats, **kw)
    return fed.clone(), length.clone(), score.clone()


def _caption_equal(fa, la, fb, lb):
    """Rows whose captions (BOS .. the token at `length`) agree."""
    T1 = fa.shape[1]
    keep = torch.arange(T1, device=fa.device).unsqueeze(0) <= la.unsqueeze(1)
    return (la == lb) & ((fa * keep) == (fb * keep)).all(dim=1)


@pytest.mark.parametrize("config,B", [("msrvtt_base_ami", 1), ("msrvtt_base_ami", 17), ("msrvtt_base_ami", 128),
                                      ("msrvtt_care", 5), ("msrvtt_care", 100), ("msrvtt_cabase", 33), ("msvd_base_i", 64),
                                      # d_model 1024 / 768 (round 4: the K-split forms, BASELINE configs[3]'s 32 clips per GPU)
                                      ("vatex_care_large", 32), ("vatex_care_large", 3), ("care_median_gelu", 64)])
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_resident_decode_against_multi_launch_and_oracle(config, B, mode):
    """Peaked (trained-like) logits: the resident form and the multi-launch form (projected cross K/V: the same rounding
    points) must give the same caption wherever the oracle's every step is decided by a clear margin, and nearly
    always otherwise: at most B // 32 captions may differ between the forms (measured: 3 of 128; B // 16 until round 4),
    and a differing caption must part from the oracle's at a step the oracle's own margin decides by less than 5e-2.  A
    margin is the difference of TWO log-probabilities, and on THIS model, whose logits reach +-30, the bf16 noise on
    one is 2.2 - 2.6e-2 (DESIGN.md section 7, BF16_LSE_PEAKED = 3.5e-2 is its bar): flips at margins of 2.1e-2, 2.4e-2
    and 3.7e-2 were seen when the bar was tried at 5e-3 / 3.5e-2; the 5e-3 of the flat random-init fixtures
    (GREEDY_TIE_TOL) is below this model's noise floor;
    scores within the bf16 bar; the resident path must actually have run."""
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN, MODES, _audit_greedy

    # fp16 mode (round 5): 8 x less noise on a log-probability, so the bars of the MULTI-LAUNCH audit
    # hold for the resident form there - a differing caption must part from the oracle at a margin below 1e-2 (twice the
    # 5e-3 noise bar of one log-probability on this model), scores agree within that bar (measured 3.4e-3), the forms within B // 64 captions
    lse_bar = MODES[mode]["lse_peaked"]
    tie_tol, forms_slack, score_bar = (5e-2, max(1, B // 32), 2e-2) if mode == "bf16" else (1e-2, max(1, B // 64), 5e-3)
    opt, P, model, feats = _setup(config, B, mode, seed=189, boost=PEAKED_ROWS)
    eng = model.engine()
    eng.latent = False
    ml = _run(eng, feats, use_graph=False)
    assert not eng.last_decode.get("resident")
    eng.resident_max_rows = 128
    assert eng.resident_ok(B)
    rs = _run(eng, feats, use_graph=False)
    assert eng.last_decode.get("resident") and 1 <= int(eng.last_decode["steps"]) <= eng.T
    same = _caption_equal(rs[0], rs[1], ml[0], ml[1])
    assert int(same.sum()) >= B - forms_slack, "{} of {} captions differ between the two forms".format(B - int(same.sum()), B)
    n = rs[1].clamp(min=1).float()
    assert ((rs[2] - ml[2]).abs() / n)[same].max().item() < score_bar
    idx = sorted(set(int(i) for i in torch.linspace(0, B - 1, min(B, 12)).round().tolist()))
    sample = [f[idx].cpu() for f in feats]
    hyps, scores, gaps = care_cpu.translate_batch(P, opt, sample, return_gaps=True)
    inputs = care_cpu.inputs_for_decoder(opt, care_cpu.encoding_phase(P, opt, sample))
    for j, i in enumerate(idx):
        k = int(rs[1][i])
        h, r = rs[0][i, 1:k + 1].tolist(), hyps[j][0]
        if gaps[j]["select"] >= CLEAR_MARGIN:
            assert h == r, "clip {}: clear margins ({:.3f}) but the resident ids differ".format(i, gaps[j]["select"])
        if h == r:
            assert abs(float(rs[2][i]) / k - scores[j][0]) < lse_bar
        else:
            _audit_greedy(P, opt, {kk: v[j:j + 1] for kk, v in inputs.items()}, h, r, tie_tol)


@pytest.mark.parametrize("config,B", [("msrvtt_base_ami", 100), ("msrvtt_care", 37)])
def test_resident_rows_do_not_depend_on_their_batch(config, B):
    """A clip decodes to the same tokens alone, in a chunk, or in the full batch (rows are independent in every phase;
    only the order in which the log-sum-exp partials of the vocabulary phase merge follows the grid: scores 1e-4), run
    after run, eager or replayed from the captured graph."""
    boost = {"cls_head.tgt_word_prj.weight": {3: 4.0, 0: 3.0}}  # early EOS at mixed steps, generated PADs
    opt, P, model, feats = _setup(config, B, "bf16", boost=boost)
    eng = model.engine()
    eng.resident_max_rows = 128
    full = _run(eng, feats, use_graph=False)
    assert eng.last_decode.get("resident")
    assert len(set(full[1].tolist())) > 3 and int(full[1].max()) <= eng.T
    for it in range(3):  # first sight, capture, replay
        again = _run(eng, feats, use_graph=True)
        for a, b in zip(full, again):
            assert torch.equal(a, b)
    assert any(isinstance(v, tuple) for k, v in eng._graphs.items() if k[0] == "gres"), "pass was not captured"
    for lo, n in ((0, 1), (3, 16), (B - 7, 7), (B // 2, 17)):
        sub = [f[lo:lo + n].contiguous() for f in feats]
        f_s, l_s, s_s = _run(eng, sub, use_graph=False)
        assert torch.equal(l_s, full[1][lo:lo + n])
        assert bool(_caption_equal(f_s, l_s, full[0][lo:lo + n], full[1][lo:lo + n]).all())
        assert (s_s - full[2][lo:lo + n]).abs().max().item() < 1e-4


def test_resident_two_decoder_layers_against_oracle():
    """num_hidden_layers_decoder = 2: the second layer's QKV phase normalises the first layer's FFN sum on load."""
    from care_amd import get_framework
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_state_dict
    from oracle import care_cpu
    from test_gpu_parity import CLEAR_MARGIN, _audit_greedy

    opt = make_opt("msrvtt_base_ami", num_hidden_layers_decoder=2)
    model = get_framework(opt).eval()
    P = synth_state_dict(189, [(k, tuple(v.shape)) for k, v in model.state_dict().items()], row_scale=PEAKED_ROWS)
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype("bf16")
    model.to("cuda:0")
    eng = model.engine()
    assert eng.n_layers == 2 and eng.resident_ok(9)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(4)
    feats = [torch.randn(s, generator=gen, device="cuda:0") for s in feat_shapes(opt, 9)]
    fed, length, score = _run(eng, feats, use_graph=False)
    assert eng.last_decode.get("resident")
    sample = [f.cpu() for f in feats]
    hyps, scores, gaps = care_cpu.translate_batch(P, opt, sample, return_gaps=True)
    inputs = care_cpu.inputs_for_decoder(opt, care_cpu.encoding_phase(P, opt, sample))
    exact = 0
    for i in range(9):
        k = int(length[i])
        h, r = fed[i, 1:k + 1].tolist(), hyps[i][0]
        if gaps[i]["select"] >= CLEAR_MARGIN:
            assert h == r
        if h == r:
            exact += 1
        else:
            _audit_greedy(P, opt, {kk: v[i:i + 1] for kk, v in inputs.items()}, h, r, 5e-2)
    assert exact >= 7


def test_resident_early_exit_equals_fixed_length():
    """The device-side `every row has ended` exit (Translator.py:77-81) stops after the step at which the last clip
    ended; tokens / lengths / scores are those of the pass that runs all 29 steps."""
    boost = {"cls_head.tgt_word_prj.weight": {3: 6.0}}
    opt, P, model, feats = _setup("msrvtt_base_ami", 48, "bf16", boost=boost)
    eng = model.engine()
    eng.resident_max_rows = 128
    fixed = _run(eng, feats, use_graph=False, early_exit=False)
    assert int(eng.last_decode["steps"]) == eng.T
    early = _run(eng, feats, use_graph=False, early_exit=True)
    steps = int(eng.last_decode["steps"])
    assert steps == int(fixed[1].max()) < eng.T
    assert torch.equal(early[1], fixed[1]) and torch.equal(early[2], fixed[2])
    assert bool(_caption_equal(early[0], early[1], fixed[0], fixed[1]).all())
    assert int(early[0][:, steps + 1:].abs().sum()) == 0  # nothing was fed after the exit


def test_resident_entry_point_rejects_what_it_does_not_cover():
    from care_amd import _lib

    lib = _lib.load()
    layers = (_lib.ResidentLayer * 1)()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    p = buf.data_ptr()
    args = lambda d=512, heads=8, ff=2048, rows=4, T=29, V=10547: (
        _lib.ctypes.addressof(layers), 1, p, p, None, 1, p, p, 1e-12, p, V, d, heads, ff, 1, rows, T, T, 2, 3, 0, p, T + 1, p, p, p, p,
        buf.numel(), 1, 0, None)
    assert lib.care_decode_resident(*args(d=768, heads=12)) == -3       # CARE_ESHAPE: d_model != 512
    assert lib.care_decode_resident(*args(ff=3072)) == -3
    assert lib.care_decode_resident(*args(T=200)) == -3
    assert lib.care_decode_resident(*args(rows=0)) == -1                # CARE_EINVAL
    assert lib.care_decode_resident(*args()) == -1                      # layer pointers are NULL


@pytest.fixture(params=[0, 1], ids=["fence-free", "fenced"])
def handoff(request):
    """Both forms of the hand-off between the phases of a resident launch (care_resident_set_fenced): the fence-free one
    (what the validated configuration takes by default) and the one with an agent-scope release / acquire pair (what any
    other device or partition mode gets)."""
    from care_amd import _lib

    libs = [_lib.load(), _lib.load(variant="f16")]
    for lib in libs:
        lib.care_resident_set_fenced(request.param)
    assert libs[0].care_resident_fenced() == request.param
    yield request.param
    for lib in libs:
        lib.care_resident_set_fenced(-1)


def test_resident_handoff_default_is_fence_free_on_the_validated_configuration():
    from care_amd import _lib

    lib = _lib.load()
    lib.care_resident_set_fenced(-1)
    props = torch.cuda.get_device_properties(0)
    validated = props.gcnArchName.startswith("gfx950") and props.multi_processor_count == 256
    assert lib.care_resident_fenced() == (0 if validated else 1)


def test_resident_barrier_watchdog_aborts_instead_of_hanging(handoff):
    """A grid that cannot meet at its hand-offs must give up, not hang: simulated by a counter state in which more
    workgroups than exist would have to arrive (care_decode_resident_debug(0, 1): the kernel expects producers that are
    not there).  Every row's length reads -1 - no caller can mistake a re-used workspace's old rows for results - and
    the Translator decodes the batch once more through the multi-launch path instead of failing; the same for beam
    search (every clip's count of finished hypotheses = -1)."""
    from care_amd import _lib, get_translator

    boost = {"cls_head.tgt_word_prj.weight": {3: 6.0}}
    opt, P, model, feats = _setup("msrvtt_base_ami", 5, "bf16", boost=boost)
    eng = model.engine()
    eng.resident_max_rows = eng.resident_beam_max_rows = 0   # what the multi-launch path gives
    want = get_translator(opt).translate_batch([model], {"feats": feats}, use_graph=False)
    opt5 = dict(opt, beam_size=5)
    want5 = get_translator(opt5).translate_batch([model], {"feats": feats}, use_graph=False)
    eng.resident_max_rows, eng.resident_beam_max_rows = 128, 640
    _lib.load().care_decode_resident_debug(0, 1)
    try:
        _, fed, length, score = eng.translate_greedy(feats, use_graph=False, lean=True)
        torch.cuda.synchronize()
        assert length.tolist() == [-1] * 5 and int(eng.last_decode["steps"]) == -1
        got = get_translator(opt).translate_batch([model], {"feats": feats}, use_graph=False)
        assert got[0] == want[0]
        _, nfin, _, _, _ = eng.translate_beam(feats, 5, 5, use_graph=False, lean=True)
        torch.cuda.synchronize()
        assert nfin.tolist() == [-1] * 5 and int(eng.last_decode["steps"]) == -1
        got5 = get_translator(opt5).translate_batch([model], {"feats": feats}, use_graph=False)
        assert got5[0] == want5[0]
    finally:
        _lib.load().care_decode_resident_debug(0, 0)
    _, fed, length, score = eng.translate_greedy(feats, use_graph=False, lean=True)
    assert int(length.min()) >= 1
    _, nfin, _, _, _ = eng.translate_beam(feats, 5, 5, use_graph=False, lean=True)
    assert int(nfin.min()) >= 1 and eng.last_decode.get("resident")


def test_resident_launches_under_contention_never_hang(handoff):
    """Two resident launches at a time (two engines, two threads, a HIP stream each) beside a stream of filler kernels
    that keep the CUs busy: a resident launch needs every one of its workgroups on the chip at once, which nothing
    guarantees here.  Whatever the scheduler does, every translate_batch call must come back - with the captions of
    an undisturbed run (a launch that gave up at its watchdog is decoded again by the multi-launch path) - and the
    whole exercise must end in bounded time: never a hang, never a stale or partial result."""
    import threading
    import time

    from care_amd import get_translator

    boost = {"cls_head.tgt_word_prj.weight": {3: 6.0}}
    jobs = []
    for config, B, beam in (("msrvtt_base_ami", 96, 1), ("msrvtt_care", 24, 5)):
        opt, P, model, feats = _setup(config, B, "bf16", boost=boost)
        opt = dict(opt, beam_size=beam)
        eng = model.engine()
        eng.resident_max_rows, eng.resident_beam_max_rows = 128, 640
        tr = get_translator(opt)
        want = tr.translate_batch([model], {"feats": feats}, use_graph=False)   # undisturbed
        assert eng.last_decode.get("resident")
        # ... and what the multi-launch pass - the one that decodes a batch again when its resident launch gave up - reports for
        # the same clips: another 16-bit rounding form of the same arithmetic (absorbed cross-attention, the large-batch embedder),
        # whose captions may part from the resident launch's at a near-tie.  A call must come back with one of the two.
        knob = "resident_max_rows" if beam == 1 else "resident_beam_max_rows"   # (the knob Translator._finish turns off for the second decode)
        keep = getattr(eng, knob)
        setattr(eng, knob, 0)
        want_ml = tr.translate_batch([model], {"feats": feats}, use_graph=False)
        assert not eng.last_decode.get("resident")
        setattr(eng, knob, keep)
        jobs.append((tr, model, feats, (want, want_ml)))
    stop = threading.Event()
    errors, done = [], [0, 0]

    def filler():
        with torch.cuda.stream(torch.cuda.Stream()):
            a = torch.randn(4096, 4096, device="cuda:0")
            while not stop.is_set():
                for _ in range(20):
                    a = (a @ a).clamp_(-1.0, 1.0)
                torch.cuda.current_stream().synchronize()

    def worker(k):
        tr, model, feats, (want, want_ml) = jobs[k]
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(12):
                    got = tr.translate_batch([model], {"feats": feats}, use_graph=False)
                    if got[0] == want_ml[0]:
                        done[k] += 1
                        continue
                    if got[0] != want[0]:
                        import json, os
                        os.makedirs("gpurun_out", exist_ok=True)
                        json.dump(dict(job=k, passes_done=done[k], differing=[i for i, (a, b) in enumerate(zip(got[0], want[0])) if a != b],
                                       got=got[0][:6], want=want[0][:6], got_scores=got[1][:6], want_scores=want[1][:6],
                                       last_decode={a: (int(b) if torch.is_tensor(b) else b) for a, b in model.engine().last_decode.items()}),
                                  open("gpurun_out/contention_failure.json", "w"))
                    assert got[0] == want[0], "job {}: captions changed under contention (last pass: {}; caption lengths {})".format(
                        k, {a: (int(b) if torch.is_tensor(b) else b) for a, b in model.engine().last_decode.items()},
                        sorted({len(h[0]) for h in got[0]}))
                    done[k] += 1
        except Exception as exc:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(exc)))

    t0 = time.time()
    threads = [threading.Thread(target=filler)] + [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads[1:]:
        th.join(timeout=240)
    stop.set()
    threads[0].join(timeout=60)
    torch.cuda.synchronize()
    assert not any(th.is_alive() for th in threads), "a translate_batch call did not return: {} / {} passes done".format(*done)
    assert not errors, errors
    assert done == [12, 12] and time.time() - t0 < 240


def test_unsupported_vocabulary_falls_back_to_the_multi_launch_decode():
    """A vocabulary beyond the resident kernels' 16384 columns (opts.py takes vocab_size from the corpus): resident_ok /
    resident_beam_ok say no, small batches decode through the multi-launch path - greedy and beam search - instead of
    raising CARE_ESHAPE on every call."""
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats, synth_state_dict

    opt = make_opt("msrvtt_base_ami", vocab_size=16400)
    model = get_framework(opt).eval()
    P = synth_state_dict(3, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                         row_scale={"cls_head.tgt_word_prj.weight": {3: 6.0}})
    model.load_state_dict(P, strict=True)
    model.set_compute_dtype("bf16")
    model.to("cuda:0")
    eng = model.engine()
    feats = [f.to("cuda:0") for f in synth_feats(3, feat_shapes(opt, 6))]
    assert not eng.resident_ok(6) and not eng.resident_beam_ok(6, 5, 5)
    hyps, scores = get_translator(opt).translate_batch([model], {"feats": feats})
    assert not eng.last_decode.get("resident") and len(hyps) == 6 and all(1 <= len(h[0]) <= eng.T for h in hyps)
    hyps5, _ = get_translator(dict(opt, beam_size=5)).translate_batch([model], {"feats": feats})
    assert not eng.last_decode.get("resident") and len(hyps5) == 6


@pytest.mark.parametrize("config,B,bm,resident", [
    ("vatex_care_large", 32, 1, True), ("care_median_gelu", 32, 1, True), ("msrvtt_care", 32, 1, True),
    ("vatex_care_large", 8, 5, True), ("msrvtt_care", 16, 5, True), ("msrvtt_care", 8, 8, True),
    # the multi-launch passes (segments of the early-exit forms, each a graph of its own; the fused beam selections)
    ("msrvtt_base_ami", 1024, 1, False), ("msrvtt_care", 300, 5, False), ("vatex_care_large", 256, 1, False),
    ("msrvtt_care", 2048, 5, False), ("msrvtt_care", 40, 1, False)])
def test_graph_replays_over_recycled_input_buffers_equal_eager_passes(config, B, bm, resident):
    """A loader that frees a batch's device tensors and allocates the next batch's - at the same addresses, so the pass replays
    its hipGraph - must get what an eager pass over the same data gives, batch after batch.  Round 6 found the d_model 768 /
    1024 greedy passes ending after ONE step from the second replay on: the launcher cleared the launch's sync area with
    hipMemsetAsync, and the memset node of a captured graph replayed as a fill with a 16-byte pattern of two pointers (ROCm 7.2) -
    hand-off counters and the count of ended rows started at garbage.  The launchers zero the area with a kernel of their own
    now (csrc/decode_resident.h, res_zero_kernel)."""
    from care_amd.configs import feat_shapes
    from test_gpu_properties import PEAKED_ROWS, _setup

    opt, P, model, _ = _setup(config, 1, "fp16", seed=189, boost=PEAKED_ROWS)
    eng = model.engine()
    eng.resident_max_rows, eng.resident_beam_max_rows = (256, 640) if resident else (0, 0)
    gen = torch.Generator().manual_seed(5)
    host = [torch.randn(s, generator=gen) for s in feat_shapes(opt, 8 * B)]
    replays = 0
    for k in range(8):
        dev = [f[k * B: (k + 1) * B].to("cuda:0") for f in host]
        before = len([v for v in eng._graphs.values() if v != "seen"])
        if bm == 1:
            got = [t.clone() for t in eng.translate_greedy(dev, use_graph=True, lean=True)[1:]]
            steps = int(eng.last_decode["steps"])
            want = [t.clone() for t in eng.translate_greedy(dev, use_graph=False, lean=True)[1:]]
        else:
            got = [t.clone() for t in eng.translate_beam(dev, bm, bm, use_graph=True, lean=True)[1:]]
            steps = int(eng.last_decode["steps"])
            want = [t.clone() for t in eng.translate_beam(dev, bm, bm, use_graph=False, lean=True)[1:]]
        assert bool(eng.last_decode.get("resident")) == resident
        replays += int(before > 0)
        assert steps == int(eng.last_decode["steps"]), (k, steps)
        for a, b in zip(got, want):
            assert torch.equal(a, b), "batch {}: the graph pass and the eager pass differ".format(k)
        del dev
    if replays < 2:   # (the caching allocator placed the new batches elsewhere: every pass ran eagerly - nothing was put to the test)
        pytest.skip("the recycled buffers did not come back at the same addresses ({} replays)".format(replays))


def test_mixed_traffic_on_one_engine_equals_eager_passes():
    """What a service does to one model: batches of changing sizes, greedy and beam, a metrics step in between, another model's
    passes interleaved, the compute mode switched and switched back, the workspace budget forcing evictions - every graph-path
    result against an eager pass of the same engine over the same (recycled) tensors."""
    from care_amd.configs import feat_shapes
    from test_gpu_properties import PEAKED_ROWS, _setup

    opt, P, model, _ = _setup("msrvtt_care", 1, "fp16", seed=189, boost=PEAKED_ROWS)
    opt2, P2, other, _ = _setup("vatex_care_large", 1, "fp16", seed=189, boost=PEAKED_ROWS)
    for m in (model, other):
        m.engine().resident_max_rows, m.engine().resident_beam_max_rows = 256, 640
    gen = torch.Generator().manual_seed(11)
    host = {id(model): [torch.randn(s, generator=gen) for s in feat_shapes(opt, 640)],
            id(other): [torch.randn(s, generator=gen) for s in feat_shapes(opt2, 640)]}
    plan = [(model, 128, 1), (model, 64, 5), (other, 32, 1), (model, 128, 1), (model, 300, 1), (other, 32, 5), (model, 64, 5),
            (model, 128, 5), (model, 128, 1), (other, 32, 1), (model, 46, 1), (model, 128, 1), (model, 300, 1), (other, 32, 1)]
    checked = 0
    for round_ in range(3):
        if round_ == 1:   # the other 16-bit mode and back: engines are rebuilt, graphs and workspaces start over
            model.set_compute_dtype("bf16")
            model.set_compute_dtype("fp16")
            model.engine().resident_max_rows, model.engine().resident_beam_max_rows = 256, 640
        if round_ == 2:   # a budget that does not hold two shapes' workspaces: every other pass evicts
            model.engine().ws_budget_bytes = 64 << 20
        for i, (m, B, bm) in enumerate(plan):
            eng = m.engine()
            lo = (37 * (i + round_)) % (640 - B)
            dev = [f[lo: lo + B].to("cuda:0") for f in host[id(m)]]
            run = (lambda g: eng.translate_greedy(dev, use_graph=g, lean=True)[1:]) if bm == 1 else \
                  (lambda g: eng.translate_beam(dev, bm, bm, use_graph=g, lean=True)[1:])
            got = [t.clone() for t in run(True)]
            want = [t.clone() for t in run(False)]
            for a, b in zip(got, want):
                assert torch.equal(a, b), "round {} pass {} ({} clips, beam {}): graph path and eager pass differ".format(round_, i, B, bm)
            checked += 1
            if i % 5 == 4 and m is model:   # a metrics step (teacher-forced pass, its own workspaces and side stream) in between
                ids = torch.randint(4, opt["vocab_size"], (B, eng.T), device="cuda:0")
                ids[:, 0] = 1
                a = eng.metrics_step(dev, ids, ids)[0].clone()
                b = eng.metrics_step(dev, ids, ids)[0].clone()
                assert torch.equal(a, b)
            del dev
    assert checked == 3 * len(plan)
