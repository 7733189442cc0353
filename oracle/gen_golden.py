"""Generate tests/golden/*.npz by running the GENUINE reference in the build container.

    python -m oracle.gen_golden            # needs /root/reference (read-only)

The reference ships no tests or golden vectors, so parity is pinned on
outputs of the reference itself: for each case below the reference model is built from
an explicit `opt` (care_amd/configs.py), loaded with the deterministic synthetic state
dict (care_amd/synth.py), run on deterministic synthetic inputs, and its outputs are
stored.  Only data (inputs are regenerated from seeds, outputs stored) and this script
are committed; no reference source or bytecode enters the repo.
"""
import json
import os
import sys

import numpy as np
import torch

from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import (GENERATOR_VERSION, synth_feats, synth_input_ids, synth_labels, synth_labels_attr,
                            synth_state_dict, tensor_sha256)
from oracle.ref_import import import_reference

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

EOS_ROW, PAD_ROW = 3, 0
AUX_POS = [0, 5, 28]  # positions of clip 0 whose auxiliary decoder outputs are stored
VOCAB_W = "cls_head.tgt_word_prj.weight"

# "Peaked" cases: 40 frequent-word rows of the vocabulary projection are scaled by 12 and the EOS
# row by 20, so the softmax is peaked like a trained model's, and the seed was SEARCHED (oracle,
# `python -m oracle.gen_golden search-peaked <config> <B> <first> <last>`) for a case whose every
# decision is clear: greedy top-1/top-2 log-prob margin >= 0.1 at every step of every clip; beam 5:
# the winning hypothesis never within 0.1 of being pruned and >= 0.05 ahead of the runner-up.  On
# these, "bf16 ids bit-exact" is a hard assertion (tests/test_gpu_parity.py), not a near-tie audit.
PEAKED = {VOCAB_W: {**{r: 12.0 for r in range(6, 46)}, EOS_ROW: 20.0}}

# name, config, B, seed, opt overrides, row_scale
CASES = [
    ("msvd_base_i_b10", "msvd_base_i", 10, 11, {}, {}),
    ("msrvtt_base_ami_b2", "msrvtt_base_ami", 2, 12, {}, {}),
    ("msrvtt_base_ami_eos_b4", "msrvtt_base_ami", 4, 13, {}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
    ("msrvtt_care_b2", "msrvtt_care", 2, 14, {}, {}),
    ("msrvtt_care_eos_b4", "msrvtt_care", 4, 15, {}, {VOCAB_W: {EOS_ROW: 6.0, PAD_ROW: 3.0}}),
    ("vatex_care_large_b2", "vatex_care_large", 2, 16, {}, {}),
    ("msrvtt_care_beam5_b3", "msrvtt_care_beam5", 3, 17, {}, {}),
    ("msrvtt_care_beam5_eos_b4", "msrvtt_care_beam5", 4, 15, {"topk": 3}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
    ("msrvtt_care_beam5_eos2_b4", "msrvtt_care_beam5", 4, 32, {"topk": 3}, {VOCAB_W: {EOS_ROW: 6.0, PAD_ROW: 3.0}}),
    ("msrvtt_care_beam5_eos_b1", "msrvtt_care_beam5", 1, 15, {}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
    ("msrvtt_base_ami_beam5_eos_b3", "msrvtt_base_ami", 3, 20, {"beam_size": 5}, {VOCAB_W: {EOS_ROW: 3.5}}),
    ("care_median_gelu_b2", "care_median_gelu", 2, 21, {}, {}),
    ("base_ami_mte_b2", "base_ami_mte", 2, 22, {}, {}),
    ("msrvtt_cabase_b3", "msrvtt_cabase", 3, 23, {}, {VOCAB_W: {EOS_ROW: 4.0}}),
    ("msrvtt_cabase_beam5_b2", "msrvtt_cabase", 2, 24, {"beam_size": 5}, {VOCAB_W: {EOS_ROW: 3.5}}),
    # CABase under beam search with captions that run on (EOS x 2.5: winners of 11, 8 and 26 tokens): the third attention
    # block over the concept rows across many beam re-orderings (the _b2 case above ends after 3 tokens)
    ("msrvtt_cabase_beam5_long_b3", "msrvtt_cabase", 3, 41, {"beam_size": 5, "topk": 2}, {VOCAB_W: {EOS_ROW: 2.5}}),
    ("msrvtt_care_beam5_topk8_b4", "msrvtt_care_beam5", 4, 15, {"topk": 8}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
    ("msrvtt_base_ami_peaked_b4", "msrvtt_base_ami", 4, 189, {}, PEAKED),
    ("msrvtt_care_peaked_b3", "msrvtt_care", 3, 373, {}, PEAKED),
    ("msrvtt_care_peaked_beam5_b3", "msrvtt_care_beam5", 3, 373, {}, PEAKED),
    # pre-LN decoders (opts.py:68 `--transformer_pre_ln`; off in every shipped config, an option of the classes on the path):
    # LayerNorm in front of every sub-block, no LayerNorm after the embedding, a final one in front of the head
    ("msrvtt_base_ami_preln_b3", "msrvtt_base_ami", 3, 51, {"transformer_pre_ln": True}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
    ("msrvtt_care_preln_beam5_b2", "msrvtt_care_beam5", 2, 52, {"transformer_pre_ln": True}, {VOCAB_W: {EOS_ROW: 4.0}}),
    ("msrvtt_cabase_preln_b2", "msrvtt_cabase", 2, 53, {"transformer_pre_ln": True}, {VOCAB_W: {EOS_ROW: 4.0}}),
    # the ablation rows of scripts/exp_ablation_main.sh:34,37,63,66: global guidance without the concept rows (G1L0:
    # use_attr_type "emb_", no hybrid bias), and the concept head beside an unguided decoder (G0L0: use_attr off)
    ("msrvtt_care_g1l0_b3", "msrvtt_care_g1l0", 3, 61, {}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
    ("msrvtt_care_g1l0_beam5_b2", "msrvtt_care_g1l0", 2, 62, {"beam_size": 5}, {VOCAB_W: {EOS_ROW: 4.0}}),
    ("msrvtt_care_g0l0_b3", "msrvtt_care_g0l0", 3, 63, {}, {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}),
]


def pad_hyps(hyps, width):
    """List[B][n_best][<=width] -> int32 [B, n_best_max, width] (-1 fill) + lengths."""
    nb = max(len(h) for h in hyps)
    arr = -np.ones((len(hyps), nb, width), dtype=np.int32)
    lens = np.zeros((len(hyps), nb), dtype=np.int32)
    for i, hs in enumerate(hyps):
        for j, h in enumerate(hs):
            arr[i, j, : len(h)] = h
            lens[i, j] = len(h)
    return arr, lens


def run_case(get_framework, get_translator, name, cfg, B, seed, overrides, row_scale):
    opt = make_opt(cfg, **overrides)
    torch.manual_seed(0)
    model = get_framework(opt).eval()
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = synth_state_dict(seed, shapes, row_scale=row_scale)
    model.load_state_dict(sd, strict=True)
    feats = synth_feats(seed, feat_shapes(opt, B))
    T = opt["max_len"] - 1
    input_ids = synth_input_ids(seed, B, T, opt["vocab_size"])

    rec = {}
    with torch.no_grad():
        enc = model.encoding_phase([f.clone() for f in feats])
        # memory rows of clip 0 only (keeps the fixture small); the per-modality means
        # below cover every clip, and the concept rows are the tail of the memory.
        rec["encoder_hidden_states_clip0"] = enc["encoder_hidden_states"][0].numpy()
        for i, m in enumerate(enc["mean_encoder_hidden_states"]):
            rec["mean_encoder_hidden_states_%d" % i] = m.numpy()
        for k in ("preds_attr", "avg_prob_attr", "semantic_labels", "semantic_hidden_states"):
            if enc.get(k) is not None:
                rec[k] = enc[k].numpy()
        if enc.get("semantic_embs") is not None and "concat" not in opt.get("use_attr_type", ""):
            rec["semantic_embs_clip0"] = enc["semantic_embs"][0].numpy()  # not part of the memory here
        if "preds_attr" in enc:
            top = enc["preds_attr"].topk(opt["use_attr_topk"] + 1, dim=1)[0]
            rec["concept_topk_min_gap"] = np.float64((top[:, :-1] - top[:, 1:]).min().item())

        out = model.feedforward_step({"feats": [f.clone() for f in feats], "input_ids": input_ids})
        rec["tf_input_ids"] = input_ids.numpy()
        rec["tf_hidden_states"] = out["hidden_states"][: min(B, 4)].numpy()
        logits = out["logits"]
        rec["tf_logits_lse"] = torch.logsumexp(logits, dim=-1).numpy()
        top = logits.topk(8, dim=-1)
        rec["tf_logits_top8_val"] = top[0].numpy()
        rec["tf_logits_top8_idx"] = top[1].numpy().astype(np.int32)
        # the decoder's auxiliary dict entries (Decoder/Transformer.py:239-252), clip 0, a few positions
        pos = AUX_POS
        rec["aux_attention_probs"] = out["attention_probs"][0].numpy()                    # [t, Lk] (mean over heads)
        rec["aux_intra_attention"] = out["all_intra_attentions"][-1][0].numpy()           # [H, t, t]
        rec["aux_inter_attention"] = out["all_inter_attentions"][-1][0][:, pos].numpy()   # [H, 3, Lk]
        for k in ("context", "text_context", "self_embs", "cross_embs", "input_embs", "sentence_embs"):
            rec["aux_" + k] = out[k][0][pos].numpy()                                      # [3, d]
        rec["aux_n_hidden_states"] = np.int64(len(out["all_hidden_states"]))
        if opt.get("use_attr") and len(out.get("attr_attention_probs", ())):
            rec["aux_attr_attention"] = out["attr_attention_probs"][-1][0][:, pos].numpy()

        # metrics step with the reference's OWN criteria (misc/Crit/crit_lang.py, crit_attribute.py)
        from misc.Crit.crit_attribute import NoisyOrMIL
        from misc.Crit.crit_lang import LanguageGeneration

        labels = synth_labels(input_ids)
        lang = LanguageGeneration({**opt, "label_smoothing": 0.0})
        lang.reset_recorder()
        lang({"logits": logits, "labels": labels})
        rec["tf_labels"] = labels.numpy()
        rec["metrics_lang"] = np.asarray(lang.get_info()[1], dtype=np.float64)      # [Word Acc0, Perplexity]
        if "preds_attr" in out and out["preds_attr"] is not None:
            labels_attr = synth_labels_attr(seed, B, opt["attribute_prediction_k"])
            crit = NoisyOrMIL({**opt, "calculate_mAP": True})
            crit.reset_recorder()
            crit({"preds_attr": out["preds_attr"], "avg_prob_attr": out["avg_prob_attr"], "labels_attr": labels_attr})
            rec["labels_attr"] = labels_attr.numpy()
            rec["metrics_attr"] = np.asarray(crit.get_info()[1], dtype=np.float64)  # F1@5..50, mAP

        translator = get_translator(opt)
        hyps, scores = translator.translate_batch([model], {"feats": [f.clone() for f in feats]})
        arr, lens = pad_hyps(hyps, T)
        rec["hyps"] = arr
        rec["hyp_lens"] = lens
        sc = np.full(lens.shape, np.nan, dtype=np.float64)
        for i, s in enumerate(scores):
            sc[i, : len(s)] = s
        rec["hyp_scores"] = sc

    # decision margins of the search (oracle/care_cpu.py, which the fixtures pin to the reference at
    # 1e-6): what the bf16 tests may and may not excuse as a near-tie
    from oracle import care_cpu

    o_hyps, _, gaps = care_cpu.translate_batch(sd, opt, feats, return_gaps=True)
    assert o_hyps == hyps, "oracle and reference disagree on {}".format(name)
    rec["gap_select"] = np.asarray([g["select"] for g in gaps], dtype=np.float64)
    rec["gap_rank"] = np.asarray([min(g["rank"], 1e30) for g in gaps], dtype=np.float64)
    rec["gap_best_slack"] = np.asarray([min(g["best_slack"], 1e30) for g in gaps], dtype=np.float64)

    meta = dict(name=name, config=cfg, batch=B, seed=seed, overrides=overrides,
                row_scale={k: {str(r): f for r, f in v.items()} for k, v in row_scale.items()},
                generator_version=GENERATOR_VERSION,
                state_dict=[[k, list(s)] for k, s in shapes],
                n_params=int(sum(p.numel() for p in model.parameters())),
                sha256={"feats0": tensor_sha256(feats[0]), VOCAB_W: tensor_sha256(sd[VOCAB_W]),
                        "input_ids": tensor_sha256(input_ids)},
                torch_version=torch.__version__)
    rec["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **rec)
    lens_s = lens.tolist()
    print("{:32s} params={} hyp_lens={} size={:.0f}KB".format(name, meta["n_params"], lens_s, os.path.getsize(path) / 1024))


# Model ensembling (models/Translator.py:39-52,112-133; models/Wrapper.py ModelEnsemble feeds one feature list per model):
# name, [(config, seed, overrides, row_scale) per model], B, translator overrides (the search options are the FIRST model's opt),
# whether each model gets feature lists of its own
ENS_EOS = {VOCAB_W: {EOS_ROW: 4.0, PAD_ROW: 3.0}}
ENSEMBLE_CASES = [
    ("ens_care_x2_beam5_b3", [("msrvtt_care_beam5", 17, {}, ENS_EOS), ("msrvtt_care_beam5", 71, {}, ENS_EOS)], 3, {"topk": 2}, False),
    ("ens_care_base_greedy_b3", [("msrvtt_care", 14, {}, ENS_EOS), ("msrvtt_base_ami", 12, {}, ENS_EOS)], 3, {}, True),
    ("ens_x3_beam5_b2", [("msrvtt_care_beam5", 15, {}, ENS_EOS), ("msrvtt_base_ami", 20, {}, ENS_EOS),
                         ("msrvtt_care_g1l0", 62, {}, ENS_EOS)], 2, {"beam_size": 5, "topk": 3}, True),
]
ENS_DIR = os.path.join(OUT_DIR, "ensemble")


def run_ensemble_case(get_framework, get_translator, name, members, B, t_over, own_feats):
    """The reference Translator over a list of reference models; stored: the hypotheses and scores, the recipe of every member
    (tests/conftest.py EnsembleCase regenerates weights and features from the seeds) and the oracle's decision margins."""
    from oracle import care_cpu

    models, opts, sds, feats_list, metas = [], [], [], [], []
    for cfg, seed, overrides, row_scale in members:
        opt = make_opt(cfg, **{**overrides, **(t_over if not models else {})})
        torch.manual_seed(0)
        model = get_framework(opt).eval()
        shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        sd = synth_state_dict(seed, shapes, row_scale=row_scale)
        model.load_state_dict(sd, strict=True)
        # one feature list per model (seeded by the member) or the first member's for all (the modalities must agree then)
        fseed = seed if own_feats else members[0][1]
        feats = synth_feats(fseed, feat_shapes(opt, B))
        models.append(model); opts.append(opt); sds.append(sd); feats_list.append(feats)
        metas.append(dict(config=cfg, seed=seed, feats_seed=fseed, overrides={**overrides, **(t_over if len(models) == 1 else {})},
                          row_scale={k: {str(r): f for r, f in v.items()} for k, v in row_scale.items()},
                          state_dict=[[k, list(sh)] for k, sh in shapes],
                          sha256={"feats0": tensor_sha256(feats[0]), VOCAB_W: tensor_sha256(sd[VOCAB_W])}))
    translator = get_translator(opts[0])
    batch = {"feats": [[f.clone() for f in fl] for fl in feats_list] if own_feats else [f.clone() for f in feats_list[0]]}
    hyps, scores = translator.translate_batch(models, batch)
    T = opts[0]["max_len"] - 1
    arr, lens = pad_hyps(hyps, T)
    sc = np.full(lens.shape, np.nan, dtype=np.float64)
    for i, s_ in enumerate(scores):
        sc[i, : len(s_)] = s_
    o_hyps, o_scores, gaps = care_cpu.translate_batch_ensemble(sds, opts, feats_list, return_gaps=True)
    assert o_hyps == hyps, "oracle and reference disagree on {}".format(name)
    assert max(abs(a - b) for x, y in zip(o_scores, scores) for a, b in zip(x, y)) < 1e-5
    rec = dict(hyps=arr, hyp_lens=lens, hyp_scores=sc,
               gap_select=np.asarray([g["select"] for g in gaps], dtype=np.float64),
               gap_rank=np.asarray([min(g["rank"], 1e30) for g in gaps], dtype=np.float64),
               gap_best_slack=np.asarray([min(g["best_slack"], 1e30) for g in gaps], dtype=np.float64),
               meta_json=np.array(json.dumps(dict(name=name, batch=B, own_feats=own_feats, members=metas,
                                                  generator_version=GENERATOR_VERSION, torch_version=torch.__version__))))
    os.makedirs(ENS_DIR, exist_ok=True)
    path = os.path.join(ENS_DIR, name + ".npz")
    np.savez_compressed(path, **rec)
    print("{:32s} members={} hyp_lens={} min gaps select {:.3g} rank {:.3g} size={:.0f}KB".format(
        name, len(members), lens.tolist(), float(rec["gap_select"].min()), float(rec["gap_rank"].min()), os.path.getsize(path) / 1024))


def host_helpers_golden():
    """Known answers of the host-side helpers next to the path: frame sampling
    (misc/utils.py:307-317) and detokenisation (misc/utils.py:117-137), from the reference itself."""
    from misc.utils import get_uniform_ids_from_k_snippets, resampling, to_sentence

    vocab = {i: "w%d" % i for i in range(20)}
    hyps = [[7, 8, 9, 3, 5], [7, 0, 9], [3], [], [6, 6, 3, 3], [11, 12, 13]]
    cases = {
        "uniform": {"%d,%d" % (l, k): get_uniform_ids_from_k_snippets(l, k)
                    for l, k in [(60, 28), (60, 8), (60, 60), (32, 28), (100, 7)]},
        "resampling": {"%d,%d" % (a, b): resampling(a, b) for a, b in [(10, 28), (27, 28), (60, 28)]},
        "to_sentence": [{"hyp": h, "plain": to_sentence(h, vocab), "add_eos": to_sentence(h, vocab, add_eos=True)}
                        for h in hyps],
    }
    json.dump(cases, open(os.path.join(OUT_DIR, "host_helpers.json"), "w"))
    print("host_helpers.json written")


def search_peaked(cfg, B, first, last):
    """Seed search for the peaked cases (oracle only; the chosen seed is then run through the reference)."""
    from care_amd import get_framework as build_framework
    from oracle import care_cpu

    for seed in range(first, last):
        line = "seed %d" % seed
        for bm in (1, 5):
            opt = make_opt(cfg, beam_size=bm)
            shapes = [(k, tuple(v.shape)) for k, v in build_framework(opt).state_dict().items()]
            sd = synth_state_dict(seed, shapes, row_scale=PEAKED)
            hyps, _, gaps = care_cpu.translate_batch(sd, opt, synth_feats(seed, feat_shapes(opt, B)), return_gaps=True)
            line += " | beam %d: select %.4f slack %.4f rank %.4f lens %s" % (
                bm, min(g["select"] for g in gaps), min(g["best_slack"] for g in gaps),
                min(g["rank"] for g in gaps), [len(h[0]) for h in hyps])
            if bm == 1 and min(g["select"] for g in gaps) < 0.1:
                break
        print(line, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "search-peaked":
        return search_peaked(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    get_framework, get_translator = import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    only = set(sys.argv[1:])
    if not only or "host_helpers" in only:
        host_helpers_golden()
    for case in CASES:
        if only and case[0] not in only:
            continue
        run_case(get_framework, get_translator, *case)
    for case in ENSEMBLE_CASES:
        if only and case[0] not in only:
            continue
        run_ensemble_case(get_framework, get_translator, *case)


if __name__ == "__main__":
    main()
