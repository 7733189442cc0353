"""CPU: host-side logic, the drop-in seam, and the C-ABI library (no compute calls)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_matches_reference_inventory(golden):
    """Keys and shapes equal the reference's (captured by gen_golden.py), so its checkpoints load strict."""
    from care_amd import get_framework
    from care_amd.configs import make_opt

    m = golden.meta
    model = get_framework(make_opt(m["config"], **m["overrides"]))
    ours = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    assert sorted(ours) == sorted((k, s) for k, s in m["state_dict"])
    assert sum(p.numel() for p in model.parameters()) == m["n_params"]
    opt, P, _, _ = golden.build()
    model.load_state_dict(P, strict=True)


def test_reference_recorded_parameter_count():
    # notebooks/retrieval_robustness.ipynb:186-187: 18,218,884 parameters, vocab 10,547
    from care_amd import get_framework
    from care_amd.configs import make_opt

    model = get_framework(make_opt("msrvtt_care"))
    assert sum(p.numel() for p in model.parameters()) == 18218884
    assert model.backbone is None and model.pointer is None
    assert model.input_keys_for_decoder == ["encoder_hidden_states", "semantic_hidden_states"]
    assert model.get_keys_to_device() == ["feats", "input_ids"]
    assert get_framework(make_opt("msrvtt_base_ami")).input_keys_for_decoder == ["encoder_hidden_states"]


def test_init_weights_follow_reference_scheme():
    # models/Framework.py:115-134: LN = (1, 0), Linear bias 0, PAD embedding row 0, hybrid_bias 0
    from care_amd import get_framework
    from care_amd.configs import make_opt

    sd = get_framework(make_opt("msrvtt_care")).state_dict()
    assert torch.all(sd["decoder.embedding.LayerNorm.weight"] == 1) and torch.all(sd["decoder.embedding.LayerNorm.bias"] == 0)
    assert torch.all(sd["decoder.embedding.word_embeddings.weight"][0] == 0)
    assert torch.all(sd["decoder.layers.0.ffn.dense1.bias"] == 0)
    assert torch.all(sd["decoder.layers.0.inter_attention.SDPA.hybrid_bias"] == 0)
    assert sd["decoder.layers.0.inter_attention.SDPA.hybrid_bias"].shape == (8, 114)


def test_factory_errors_match_reference():
    from care_amd import get_framework, get_translator
    from care_amd.configs import make_opt

    with pytest.raises(ValueError, match="can not find the class"):
        get_framework(make_opt("msrvtt_base_ami", encoder="NoSuchEncoder"))
    with pytest.raises(ValueError, match="can not find the class"):
        get_translator(make_opt("msrvtt_base_ami", decoding_type="NARFormer"))
    with pytest.raises(ValueError):
        get_framework(make_opt("msrvtt_base_ami", decoder="SingleLayerRNNDecoder"))
    model = get_framework(make_opt("msrvtt_care"))
    with pytest.raises(KeyError, match="semantic_hidden_states"):
        model.prepare_inputs_for_decoder({"encoder_hidden_states": torch.zeros(1)}, {"feats": []})


def test_no_silent_fallback_on_cpu_or_in_training():
    from care_amd import get_framework, get_translator
    from care_amd.configs import feat_shapes, make_opt
    from care_amd.synth import synth_feats

    opt = make_opt("msvd_base_i")
    model = get_framework(opt).eval()
    feats = synth_feats(0, feat_shapes(opt, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.encoding_phase(feats)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_translator(opt).translate_batch([model], {"feats": feats})
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # training mode too: autograd over the HIP kernels
        model.train().feedforward_step({"feats": feats, "input_ids": torch.zeros(2, 29, dtype=torch.long)})
    with pytest.raises(NotImplementedError, match="eval mode"):
        model.train().encoding_phase(feats)
    with pytest.raises(TypeError):
        get_translator(opt).translate_batch([torch.nn.Linear(2, 2)], {"feats": feats})


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "care_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_library_is_bound_to_its_sources(tmp_path, monkeypatch):
    """The hash of csrc/ + include/care_hip.h + flags is compiled into the library; a library from other sources is
    detected without being loaded, and _lib refuses it when it cannot rebuild."""
    from care_amd import _lib, build

    build.build_all()
    for variant in build.VARIANTS:
        assert not build.needs_build(variant)
        assert build.embedded_hash(build.lib_path(variant)) == build.source_hash(variant)
        assert _lib.load(variant=variant).care_source_hash().decode() == build.source_hash(variant)
    assert build.source_hash("") != build.source_hash("f16")                  # the flags are part of the hash
    assert build.source_hash("", ("-DRES_NOINLINE",)) != build.source_hash("")  # ... a tool build's too
    # a stale library: same file, other tree hash
    monkeypatch.setattr(build, "source_hash", lambda variant="", flags_extra=(): "0" * 32)
    assert build.needs_build("") and build.needs_build("f16")
    monkeypatch.setenv("CARE_NO_REBUILD", "1")
    with pytest.raises(_lib.CareHipError, match="built from other sources"):
        _lib._ensure_current("", build.lib_path(""))
    # a file that is no library of this build system carries no hash
    junk = tmp_path / "libjunk.so"
    junk.write_bytes(b"\x7fELF" + b"\0" * 64)
    assert build.embedded_hash(str(junk)) is None


def test_abi_library_exports_every_declared_symbol():
    """The .so loads and exports exactly the entry points include/care_hip.h declares."""
    import ctypes

    from care_amd import _lib, build

    build.build_all()
    header = open(os.path.join(ROOT, "include", "care_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|void|const char\*)\s+(care_\w+)\s*\(", header, re.M))
    assert len(declared) >= 18
    for variant in build.VARIANTS:  # the bf16 and the fp16 library: one source, one ABI
        lib = ctypes.CDLL(build.lib_path(variant))
        for name in declared:
            assert hasattr(lib, name), (variant, name)
    assert declared == set(_lib.exported_symbols())
    loaded = _lib.load()
    assert loaded.care_h16() == b"bf16" and _lib.load(variant="f16").care_h16() == b"fp16"
    assert _lib.load(variant="f16").care_build_flags() == b"-DCARE_H16_FP16" and loaded.care_build_flags() == b""
    # the two structs of care_decode_resident: ctypes mirrors against the C compiler's layout of the header
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "care_hip.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", '
                             'sizeof(care_resident_attn), sizeof(care_resident_layer), offsetof(care_resident_layer, n_att), '
                             'offsetof(care_resident_attn, bias)); return 0; }\n')
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(td, "sz")], check=True)
        sizes = [int(v) for v in subprocess.run([os.path.join(td, "sz")], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.ResidentAttn), ctypes.sizeof(_lib.ResidentLayer), _lib.ResidentLayer.n_att.offset,
                     _lib.ResidentAttn.bias.offset]
    assert loaded.care_decode_resident_scratch(1, 512, 2048, 10547) == 53248 + 16 * (512 * 18 + 2048 * 2 + 165 * 12)
    assert loaded.care_decode_resident_scratch(0, 512, 2048, 10547) < 0
    # ... and of the beam launch: + the group lists [rows16, parts, 5] x (value, group), the bf16 hidden rows and (round 6) the
    # bf16 input rows of the next step
    assert loaded.care_decode_resident_beam_scratch(3, 5, 512, 2048, 10547) == 53248 + 16 * (512 * 18 + 2048 * 2 + 165 * 12 + 165 * 40 + 1024 + 1024)
    assert loaded.care_decode_resident_beam_scratch(0, 5, 512, 2048, 10547) < 0
    assert loaded.care_version() == int(re.search(r"#define CARE_ABI_VERSION (\d+)", header).group(1))
    assert loaded.care_arch() == b"gfx950"
    assert loaded.care_argmax_parts(10547) == 166
    # the K ranges of the training backward's few-tile products (csrc/backward.hip: care_gemm_kn_splits): dx = dlogits W at
    # 64 clips is split, dW and the d x d products are not; every slab gets a range of a multiple of 16
    assert loaded.care_gemm_kn_splits(1856, 512, 10547) == 5
    assert loaded.care_gemm_kn_splits(10547, 512, 1856) == 1 and loaded.care_gemm_kn_splits(1856, 512, 512) == 1
    for M, N, K in [(64, 64, 2048), (300, 100, 2049), (1856, 512, 10547), (128, 512, 4100)]:
        ks = loaded.care_gemm_kn_splits(M, N, K)
        chunk = -(-(-(-K // ks)) // 16) * 16
        assert 1 <= ks <= 8 and -(-K // chunk) == ks
    assert loaded.care_gemm_kn_splits(0, 1, 1) < 0
    assert loaded.care_argmax_parts_bf16(1024, 10547) % 8 == 0
    # the column split of the A-stationary GEMM (csrc/gemm_as.hip: pick_ns), at its measured optima
    for rows, parts in [(1, 512), (4096, 16), (8192, 8), (16384, 4), (20480, 8), (32768, 2), (65536, 1)]:
        assert loaded.care_argmax_parts_bf16(rows, 10547) == parts, rows
    assert loaded.care_argmax_parts_bf16_min(4096, 10547, 512, 1, 8) == 16   # 128-row panels: the greedy split
    # from 8192 bf16 rows the 256-row kernel: whole launch rounds over the 256 CUs (csrc/gemm_vocab.hip)
    for rows, parts in [(8192, 8), (12288, 16), (20480, 16), (32768, 8)]:
        assert loaded.care_argmax_parts_bf16_min(rows, 10547, 512, 1, 8) == parts, rows
    assert loaded.care_argmax_parts_bf16_min(32768, 10547, 512, 0, 8) == 8    # fp32 rows stay on the 128-row kernel


def test_synth_generator_is_deterministic_and_portable():
    from care_amd.synth import synth_state_dict, tensor_sha256, uniform

    u = uniform(7, "x", (4,))
    np.testing.assert_array_equal(u, uniform(7, "x", (4,)))
    assert not np.array_equal(u, uniform(8, "x", (4,)))
    sd = synth_state_dict(3, [("a.LayerNorm.weight", (8,)), ("a.LayerNorm.bias", (8,)), ("a.dense.weight", (8, 8)),
                              ("a.dense.bias", (8,))], row_scale={"a.dense.weight": {2: 10.0}})
    assert abs(float(sd["a.LayerNorm.weight"].mean()) - 1.0) < 0.1
    assert float(sd["a.dense.weight"][2].abs().max()) > float(sd["a.dense.weight"][1].abs().max())
    # pinned digest: any change of the generator invalidates tests/golden (regenerate with oracle/gen_golden.py)
    assert tensor_sha256(sd["a.dense.weight"]) == tensor_sha256(synth_state_dict(
        3, [("a.dense.weight", (8, 8))], row_scale={"a.dense.weight": {2: 10.0}})["a.dense.weight"])


def test_translator_result_assembly_matches_reference_quirks():
    """Host-side assembly (no GPU): n_best shrinks across clips exactly like Translator.py:211-220."""
    from care_amd.translator import Translator_ARFormer

    tr = Translator_ARFormer({"beam_size": 5, "topk": 3, "beam_alpha": 1.0, "max_len": 30})

    class FakeEngine:
        T = 29

        def translate_beam(self, feats, bm, need, use_graph=True, lean=False):
            nfin = torch.tensor([5, 1, 5], dtype=torch.int32)
            fscore = torch.tensor([[-4.0, -2.0, -9.0, -8.0, -7.0, 0, 0, 0, 0, 0],
                                   [-3.0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                                   [-1.0, -2.0, -3.0, -4.0, -5.0, 0, 0, 0, 0, 0]])
            flen = torch.tensor([[2, 2, 3, 4, 7, 0, 0, 0, 0, 0], [3, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                                 [1, 1, 1, 1, 1, 0, 0, 0, 0, 0]], dtype=torch.int32)
            fhyp = torch.arange(3 * 10 * 30, dtype=torch.int32).view(3, 10, 30)
            return None, nfin, fscore, flen, fhyp

    hyps, scores = tr._beam(FakeEngine(), [])
    assert [len(h) for h in hyps] == [3, 1, 1]          # clip 1 has one hypothesis -> clip 2 is cut to one
    assert scores[0] == [-1.0, -1.0, -2.0]              # -2/2, -7/7 (stable order), -4/2
    assert hyps[0][0] == [30, 31] and hyps[0][1] == list(range(120, 127))
    assert scores[2] == [-1.0]


def _assemble_per_clip(tr, kind, arrays):
    """The reference's own loops (Translator.py:211-220 over Beam.sort_finished / get_hypothesis, Beam.py:91-132), clip by
    clip with python scalars: what the vectorised assembly of care_amd/translator.py must reproduce value for value."""
    hyps, scores, n_best = [], [], tr.topk
    if kind == "greedy":
        length, score, fed = arrays
        for i in range(len(length)):
            n = int(length[i])
            n_best = min(n_best, 1)
            hyps.append([[int(v) for v in fed[i, 1: n + 1]]][:n_best])
            scores.append([float(score[i]) / (n ** tr.beam_alpha)][:n_best])
        return hyps, scores
    nfin, fscore, flen, fhyp = arrays
    for i in range(len(nfin)):
        items = [[float(fscore[i, j]) / (int(flen[i, j]) ** tr.beam_alpha), j] for j in range(int(nfin[i]))]
        items.sort(key=lambda a: -a[0])
        n_best = min(n_best, len(items))
        hyps.append([[int(v) for v in fhyp[i, j, : int(flen[i, j])]] for _, j in items[:n_best]])
        scores.append([s for s, _ in items[:n_best]])
    return hyps, scores


@pytest.mark.parametrize("alpha", [1.0, 0.7, 0.0])
@pytest.mark.parametrize("topk", [1, 3, 8])
def test_translator_vectorised_assembly_equals_the_per_clip_loops(alpha, topk):
    """Random finished lists - ties between scores, clips with fewer finished hypotheses than topk (the n_best shrink),
    every length 1 .. 29 - through the numpy assembly and through the reference's per-clip loops: identical lists and
    identical doubles.  Chunked conversion (CHUNK_ROWS) included."""
    from care_amd.translator import Translator_ARFormer

    tr = Translator_ARFormer({"beam_size": 5, "topk": topk, "beam_alpha": alpha, "max_len": 30})
    tr.CHUNK_ROWS = 7
    rng = np.random.default_rng(5 + topk)
    B, cap, T = 61, max(5, topk) + 5, 29
    nfin = rng.integers(max(1, topk - 1), cap + 1, size=B).astype(np.int32)
    nfin[40] = min(2, topk)     # from here on at most two hypotheses per clip
    fscore = np.round(rng.normal(-20, 6, size=(B, cap)), 0).astype(np.float32)   # rounded: many exact ties
    flen = rng.integers(1, T + 1, size=(B, cap)).astype(np.int32)
    fhyp = rng.integers(0, 10547, size=(B, cap, T + 1)).astype(np.int32)
    got = tr._assemble_beam(nfin, fscore, flen, fhyp)
    want = _assemble_per_clip(tr, "beam", (nfin, fscore, flen, fhyp))
    assert got == want
    assert all(type(s) is float for row in got[1] for s in row) and all(type(t) is int for row in got[0] for h in row for t in h)
    length = rng.integers(1, T + 1, size=B).astype(np.int32)
    score = rng.normal(-30, 5, size=B).astype(np.float32)
    fed = rng.integers(0, 10547, size=(B, T + 1)).astype(np.int32)
    assert tr._assemble_greedy(length, score, fed) == _assemble_per_clip(tr, "greedy", (length, score, fed))


def test_translate_batches_is_translate_batch_one_batch_behind():
    """The pipelined entry on a stand-in engine (host tensors): same results in the same order; a second
    stand-in whose pass waits on the host calls engine.idle_hook there, and the previous batch's lists are built in those waits."""
    from care_amd.translator import Translator_ARFormer
    from care_amd.framework import TransformerSeq2Seq

    tr = Translator_ARFormer({"beam_size": 1, "topk": 1, "beam_alpha": 1.0, "max_len": 30})

    class FakeEngine:
        T = 29

        def translate_greedy(self, feats, use_graph=True, lean=False):
            B = feats[0].shape[0]
            g = torch.Generator().manual_seed(B)
            return (None, torch.randint(4, 99, (B, 30), generator=g, dtype=torch.int32),
                    torch.randint(1, 30, (B,), generator=g, dtype=torch.int32), -torch.rand(B, generator=g))

    class FakeModel(TransformerSeq2Seq):
        def __init__(self):
            torch.nn.Module.__init__(self)

        def engine(self):
            return FakeEngine()

    model = FakeModel()
    batches = [{"feats": [torch.zeros(b, 1, 1)]} for b in (3, 9, 2, 17)]
    one_by_one = [tr.translate_batch([model], b) for b in batches]
    assert list(tr.translate_batches([model], iter(batches))) == one_by_one
    assert [len(h) for h, _ in one_by_one] == [3, 9, 2, 17]

    class WaitingEngine(FakeEngine):
        idle_hook = None
        pieces = 0

        def translate_greedy(self, feats, use_graph=True, lean=False):
            while self.idle_hook is not None and self.idle_hook():   # what engine._host_count does between two segments
                WaitingEngine.pieces += 1
            return FakeEngine.translate_greedy(self, feats, use_graph, lean)

    eng = WaitingEngine()
    model.engine = lambda: eng
    tr.CHUNK_ROWS = 2
    assert list(tr.translate_batches([model], iter(batches))) == one_by_one
    assert WaitingEngine.pieces >= (3 + 9 + 2) // 2 and eng.idle_hook is None


def test_translate_batch_hands_a_list_of_models_to_the_ensemble_search():
    """Several models (models/Translator.py:39-52): the first model's engine runs the search with the others as members; each gets
    `batch['feats'][index]` when the batch carries one feature list per model (Wrapper.ModelEnsemble), else the same list;
    greedy is the beam search with beam_size 1; the results are the beam search's block."""
    from care_amd.translator import Translator_ARFormer
    from care_amd.framework import TransformerSeq2Seq

    calls = []

    class FakeEngine:
        T = 29

        def __init__(self, tag):
            self.tag = tag

        def translate_beam_ensemble(self, others, feats_list, bm, need, use_graph=True):
            calls.append((self.tag, [o.tag for o in others], [[tuple(f.shape) for f in fl] for fl in feats_list], bm, need))
            B = feats_list[0][0].shape[0]
            cap = need + bm
            nfin = torch.full((B,), need, dtype=torch.int32)
            fscore = -torch.arange(1, B * cap + 1, dtype=torch.float32).view(B, cap)
            flen = torch.full((B, cap), 2, dtype=torch.int32)
            fhyp = torch.arange(B * cap * 30, dtype=torch.int32).view(B, cap, 30) % 97
            return None, nfin, fscore, flen, fhyp

    class FakeModel(TransformerSeq2Seq):
        def __init__(self, tag):
            torch.nn.Module.__init__(self)
            self._e = FakeEngine(tag)

        def engine(self):
            return self._e

    models = [FakeModel("a"), FakeModel("b"), FakeModel("c")]
    tr = Translator_ARFormer({"beam_size": 1, "topk": 1, "beam_alpha": 1.0, "max_len": 30})
    hyps, scores = tr.translate_batch(models, {"feats": [torch.zeros(4, 28, 8), torch.zeros(4, 28, 16)]})
    assert calls[-1] == ("a", ["b", "c"], [[(4, 28, 8), (4, 28, 16)]] * 3, 1, 1)
    assert [len(h) for h in hyps] == [1] * 4 and hyps[0][0] == [0, 1] and scores[0] == [-0.5]
    tr5 = Translator_ARFormer({"beam_size": 5, "topk": 8, "beam_alpha": 1.0, "max_len": 30})
    own = [[torch.zeros(2, 28, 8)], [torch.zeros(2, 28, 16)], [torch.zeros(2, 28, 4)]]
    tr5.translate_batch(models, {"feats": own})
    assert calls[-1] == ("a", ["b", "c"], [[(2, 28, 8)], [(2, 28, 16)], [(2, 28, 4)]], 5, 8)
    with pytest.raises(ValueError):
        tr5.translate_batch(models, {"feats": own[:2]})
    with pytest.raises(TypeError):
        tr5.translate_batch([models[0], torch.nn.Linear(2, 2)], {"feats": own})


def test_sharding_bounds_and_records():
    from care_amd.sharding import pack_records, shard_bounds, unpack_records

    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3, 3), (3, 6, 3), (6, 9, 3), (9, 10, 3)]
    assert shard_bounds(2, 3, 4) == (2, 2, 1)
    fed = torch.arange(2 * 30, dtype=torch.int32).view(2, 30)
    rec = pack_records(fed, torch.tensor([5, 7]), torch.tensor([-1.5, -2.25]), per=3)
    assert rec.shape == (3, 33) and rec[2].abs().sum() == 0
    f, l, s = unpack_records(rec)
    assert torch.equal(f, fed) and l.tolist() == [5, 7] and s.tolist() == [-1.5, -2.25]


def test_graph_cache_is_lru_capped_per_kind():
    """engine._graph_put / _graph_get (no GPU needed): per-kind caps, least recently used entries go first, "seen"
    markers count like graphs, other kinds are untouched."""
    from care_amd.configs import make_opt
    from care_amd.engine import HipEngine

    eng = HipEngine(make_opt("msrvtt_base_ami"), "bf16")
    cap = eng.GRAPH_CAPS["gseg0"]
    for i in range(cap + 5):
        eng._graph_put(("gseg0", i), "seen")
    assert [k[1] for k in eng._graphs if k[0] == "gseg0"] == list(range(5, cap + 5))
    assert eng._graph_get(("gseg0", 5)) == "seen"            # touched: now the most recent
    eng._graph_put(("gseg0", 999), ("graph", "out"))
    kept = [k[1] for k in eng._graphs if k[0] == "gseg0"]
    assert 5 in kept and 6 not in kept and kept[-1] == 999 and len(kept) == cap
    for i in range(200):
        eng._graph_put(("gseg", 0, i), "seen")
    assert sum(1 for k in eng._graphs if k[0] == "gseg") == eng.GRAPH_CAPS["gseg"]
    assert sum(1 for k in eng._graphs if k[0] == "gseg0") == cap
    assert eng._graph_get(("nope",)) is None


def test_compute_modes_and_shape_routing_flags():
    from care_amd.configs import make_opt
    from care_amd.engine import HipEngine

    base = HipEngine(make_opt("msrvtt_base_ami"), "bf16")
    assert base.as_ok and base.bf_act and base.latent_capable and not base.split3
    large = HipEngine(make_opt("vatex_care_large"), "bf16")
    assert not large.as_ok and large.bf_act and large.latent_capable          # tile GEMMs, two-wave absorbed attention
    median = HipEngine(make_opt("care_median_gelu"), "bf16")
    assert not median.as_ok and median.bf_act and median.latent_capable       # three-wave absorbed attention
    x3 = HipEngine(make_opt("msrvtt_care"), "fp16x3")
    assert x3.split3 and not x3.bf and x3.wt == torch.float32 and not x3.bf_act
    with pytest.raises(ValueError):
        HipEngine(make_opt("msrvtt_care"), "fp8")


def test_small_batch_form_rules(monkeypatch):
    """Which batches take the small-batch forms (resident greedy decode, unfused embedder, projected K/V under beam
    search): bf16 mode, d_model = 512, up to `resident_max_rows` clips (256, CARE_RESIDENT_MAX_ROWS); off -> one set of
    forms at every size, and the absorbed cross-attention again follows the model alone."""
    from care_amd.configs import make_opt
    from care_amd.engine import HipEngine

    monkeypatch.delenv("CARE_RESIDENT_MAX_ROWS", raising=False)
    base = HipEngine(make_opt("msrvtt_base_ami"), "bf16")
    assert base.resident_max_rows == 256
    assert base.resident_ok(1) and base.resident_ok(256) and not base.resident_ok(257) and not base.resident_ok(0)
    assert base.small_forms(128) and not base.small_forms(320)
    assert all(base.latent_for(r) for r in (1, 640, 1 << 20))          # the multi-launch form: absorbed at every size
    base._small_pass = True                                            # ... except inside a small beam pass
    assert not base.latent_for(640)
    base._small_pass = False
    base.resident_max_rows = 0
    assert not base.resident_ok(1) and not base.small_forms(1)
    for cfg, dtype in (("msrvtt_base_ami", "fp32"), ("msrvtt_care", "fp16x3"), ("vatex_care_large", "fp32")):
        e = HipEngine(make_opt(cfg), dtype)
        assert not e.resident_ok(8) and not e.small_forms(8), (cfg, dtype)   # bf16 mode only
    for cfg in ("vatex_care_large", "care_median_gelu"):   # d_model 1024 / 768 (ff = 4 d_model): greedy, up to 128 rows
        e = HipEngine(make_opt(cfg), "bf16")
        assert e.resident_ok(1) and e.resident_ok(32) and e.resident_ok(128) and not e.resident_ok(129), cfg
        # ... and, round 5, beam search (160 rows at d_model 1024, 256 at 768: engine_resident.RESIDENT_WIDE_BEAM_MAX_ROWS)
        assert not e.small_forms(8) and e.resident_beam_ok(4, 5, 5) and not e.resident_beam_ok(52, 5, 5), cfg
    # 16-bit modes: fp16 (libcare_hip_f16.so) takes every small-batch form bf16 does
    h = HipEngine(make_opt("msrvtt_care"), "fp16")
    assert h.variant == "f16" and h.h16 == torch.float16 and h.resident_ok(128) and h.resident_beam_ok(128, 5, 5) and h.small_forms(8)
    # pre-LN decoders run the multi-launch unfused forms only (the resident phases normalise after the residual sum)
    pre = HipEngine(make_opt("msrvtt_care", transformer_pre_ln=True), "bf16")
    assert pre.pre_ln and not pre.resident_ok(8) and not pre.resident_beam_ok(8, 5, 5) and not pre.ln_fusable(1 << 20)
    assert not pre.tf_fast_ok(29, False)
    # mid-size forms: the beam selection from group maxima below BEAM_FUSED_MIN_ROWS (16-bit modes, beam <= 5)
    assert base.beam_groups_for(640, 5) and base.beam_groups_for(8191, 5) and not base.beam_groups_for(8192, 5)
    assert base.beam_groups_for(640, 8) and not base.beam_groups_for(640, 9)
    assert not HipEngine(make_opt("msrvtt_care"), "fp32").beam_groups_for(640, 5)
    assert base.MID_TILE_ROWS == (1280, 16384)
    two = HipEngine(make_opt("msrvtt_base_ami", num_hidden_layers_decoder=2), "bf16")
    assert two.resident_ok(8)
    # every shape limit of care_decode_resident is checked here, so an unsupported model takes the multi-launch decode
    # instead of failing on every small batch: a vocabulary beyond 64 x 64 x 4 columns (opts.py takes it from the corpus)
    big_v = HipEngine(make_opt("msrvtt_base_ami", vocab_size=16385), "bf16")
    assert not big_v.resident_ok(8) and not big_v.resident_beam_ok(8, 5, 5) and big_v.small_forms(8)
    assert HipEngine(make_opt("msrvtt_base_ami", vocab_size=16384), "bf16").resident_ok(8)
    # beam search: rows = clips x beam_size up to resident_beam_max_rows (640), beam_size <= 5
    assert base.resident_beam_max_rows == 640
    base.resident_max_rows = 256
    assert base.resident_beam_ok(128, 5, 5) and not base.resident_beam_ok(129, 5, 5) and base.resident_beam_ok(4, 6, 6) and not base.resident_beam_ok(4, 9, 9)
    assert base.resident_beam_ok(1, 5, 8) and not base.resident_beam_ok(1, 1, 1)
    monkeypatch.setenv("CARE_RESIDENT_MAX_ROWS", "64")
    assert HipEngine(make_opt("msrvtt_care"), "bf16").resident_max_rows == 64


def test_prefix_guidance_is_refused_not_decoded_as_the_additive_form():
    """`use_attr_flags` Gp.. (use_attr_type 'pp_emb_...') and use_attr_type 'prefix' prepend the guidance to the decoder's input
    sequence (Embeddings.py:155-157, Decoder/Transformer.py:131-160): a layout this path does not build - it must say so."""
    from care_amd import get_framework
    from care_amd.configs import make_opt

    for t in ("pp_emb_concat", "prefix"):
        with pytest.raises(ValueError, match="prefix guidance"):
            get_framework(make_opt("msrvtt_care", use_attr_type=t))
    get_framework(make_opt("msrvtt_care", use_attr_type="emb_concat"))


def test_options_that_change_the_layout_of_the_path_are_named_not_ignored():
    """A model configured with another decoding scheme, fusion, relative positions or compositional sub-layers is another
    network: get_framework says so (its checkpoint would not load either)."""
    from care_amd import get_framework
    from care_amd.configs import make_opt

    for over in (dict(decoding_type="NARFormer"), dict(fusion="channel_concat"), dict(RPE=True), dict(compositional_intra=True),
                 dict(compositional_ffn=True), dict(with_category=True), dict(attr_layer_pos="parallel", use_attr_type="_att"),
                 dict(attribute_prediction_flags="VA"), dict(decoder="SingleLayerRNNDecoder"), dict(encoder="TransformerEncoder")):
        with pytest.raises(ValueError):
            get_framework(make_opt("msrvtt_care", **over))


def test_translator_names_the_beam_sizes_it_covers():
    from care_amd import get_translator

    for bm in (0, 9, 16):
        with pytest.raises(ValueError, match="beam_size"):
            get_translator({"decoding_type": "ARFormer", "beam_size": bm})
    assert get_translator({"decoding_type": "ARFormer", "beam_size": 8}).beam_size == 8
