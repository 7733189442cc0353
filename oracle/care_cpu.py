"""CPU oracle: a functional torch-fp32 restatement of the reference's captioning forward path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the genuine
reference by `oracle/gen_golden.py` -> `tests/golden/*.npz` (checked in
tests/test_oracle_golden.py at 1e-6) -- parity status: PINNED on synthetic weights
and inputs; unpinned on real checkpoints/datasets (none exist offline).

It executes the reference algorithm *as written*: no KV cache (the whole decoder is
re-run on the full prefix every step, `models/Translator.py:71-75,118-123`), beam
inputs materialised `beam_size` times (`misc/utils.py:244-279`), per-instance host
beam bookkeeping (`misc/Decoding/Beam.py`).  That makes it both the parity checker and
the "port" CPU baseline timed by bench.py.

Everything is a pure function of `(P, opt, inputs)` where `P` is a state dict with the
reference's parameter names; there are no nn.Modules here.
All file:line citations are relative to /root/reference.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

PAD, BOS, EOS = 0, 2, 3  # config/Constants.py:1-4


# --------------------------------------------------------------------------- primitives
def _linear(P, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, P[prefix + ".weight"], P.get(prefix + ".bias"))


def _layer_norm(P, prefix: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], eps)


def _activation(name: str, x: torch.Tensor) -> torch.Tensor:
    # models/components/activations.py:3-10 (nn.GELU() = exact erf form)
    if name == "relu":
        return torch.relu(x)
    if name == "gelu":
        return F.gelu(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "linear":
        return x
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "leakyrelu":
        return F.leaky_relu(x)
    raise KeyError(name)


def _split_heads(x: torch.Tensor, n_heads: int) -> torch.Tensor:
    b, l, d = x.shape
    return x.view(b, l, n_heads, d // n_heads).permute(0, 2, 1, 3)


def attention_block(P, prefix: str, opt: dict, x: torch.Tensor, memory: Optional[torch.Tensor],
                    mask: Optional[torch.Tensor], aux: Optional[dict] = None) -> torch.Tensor:
    """Multi-head attention sub-block, post-LN (every shipped config) or pre-LN (`transformer_pre_ln`, opts.py:68).

    `MultiHeadAttention.forward` (models/components/SubLayers.py:40-81) around
    `ScaledDotProductAttention.forward` (models/components/Attention.py:69-131):
    Q/K/V Linear with bias -> scores / sqrt(head) -> masked_fill(-1e9) -> + hybrid_bias
    (after the mask, Attention.py:104-111) -> softmax -> PV -> dense -> + residual -> LN.
    Dropout is identity in eval mode.  `mask`: bool, True = masked, [B, Lq, Lk].
    """
    H = opt["num_attention_heads"]
    pre_ln = opt.get("transformer_pre_ln", False)
    x_in = x
    if pre_ln:  # SubLayers.py:55-56: the LayerNorm comes first; keys / values of a self-attention are the normalised rows too (:62-63)
        x = _layer_norm(P, prefix + ".LayerNorm", x, opt["layer_norm_eps"])
    kv = x if memory is None else memory
    q = _split_heads(_linear(P, prefix + ".SDPA.query", x), H)
    k = _split_heads(_linear(P, prefix + ".SDPA.key", kv), H)
    v = _split_heads(_linear(P, prefix + ".SDPA.value", kv), H)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1), -1e9)
    hb = P.get(prefix + ".SDPA.hybrid_bias")
    if hb is not None:
        scores = scores + hb[None, :, None, :]
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
    ctx = ctx.view(ctx.shape[0], ctx.shape[1], -1)
    context = _linear(P, prefix + ".dense", ctx)  # SubLayers.py:69-70 `context` (dropout is identity)
    out = context + x_in  # :75-76 the residual is the block's INPUT tensor (un-normalised in a pre-LN block)
    if not pre_ln:
        out = _layer_norm(P, prefix + ".LayerNorm", out, opt["layer_norm_eps"])  # :78-79
    if aux is not None:
        aux.update(probs=probs, context=context, embs=out)
    return out


def ffn_block(P, prefix: str, opt: dict, x: torch.Tensor) -> torch.Tensor:
    """`PositionwiseFeedForward.forward` (SubLayers.py:137-152), post-LN."""
    pre_ln = opt.get("transformer_pre_ln", False)
    xin = _layer_norm(P, prefix + ".LayerNorm", x, opt["layer_norm_eps"]) if pre_ln else x  # :140-141
    h = _activation(opt["hidden_act"], _linear(P, prefix + ".dense1", xin))
    out = _linear(P, prefix + ".dense2", h) + x
    return out if pre_ln else _layer_norm(P, prefix + ".LayerNorm", out, opt["layer_norm_eps"])  # :149-150


# --------------------------------------------------------------------------- encoder
def encode_modality(P, opt: dict, ch: str, feats: torch.Tensor) -> torch.Tensor:
    """One stream of `MultipleStreams` (models/Encoder.py:51-76).

    `Embedder`: Linear -> LayerNorm -> Dropout (Encoder.py:165-168).
    `MultiTransformerEncoder`: Linear -> TransformerEncoderBase (Encoder.py:190-193,
    244-298): + trainable position embedding, LayerNorm, then `EncoderLayer`s
    (models/components/Layers.py:16-52; unmasked self-attention + FFN).
    """
    prefix = "encoder.Encoder_{}".format(ch.upper())
    eps = opt["layer_norm_eps"]
    if opt["encoder"] == "Embedder":
        return _layer_norm(P, prefix + ".1", _linear(P, prefix + ".0", feats), eps)
    if opt["encoder"] == "MultiTransformerEncoder":
        h = _linear(P, prefix + ".0", feats)
        n = h.shape[1]
        if opt.get("trainable_pe", False):
            pos = P[prefix + ".1.position_embeddings.weight"][:n]
        else:
            pos = P[prefix + ".1.position_embeddings.pe"][0, :n]
        h = _layer_norm(P, prefix + ".1.LayerNorm", h + pos.unsqueeze(0), eps)
        for li in range(opt["num_hidden_layers_encoder"]):
            lp = "{}.1.layers.{}".format(prefix, li)
            h = attention_block(P, lp + ".intra_attention", opt, h, None, None)
            h = ffn_block(P, lp + ".ffn", opt, h)
        return h
    raise ValueError("encoder `{}` is outside the hot path".format(opt["encoder"]))


def concept_probabilities(scores: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """`prepare_merged_probs` without a mask (models/Predictor/pred_attribute.py:17-46).

    Noisy-OR over the sequence axis, restated literally: it is *not* sigmoid(scores)
    bit for bit even at seq_len 1 (hard part 4).
    """
    probs = torch.sigmoid(scores)
    raw = torch.log(torch.clamp(1.0 - probs, 1e-12, 1))
    preds = 1.0 - torch.exp(raw.sum(dim=1))
    return preds, probs.mean(dim=(1, 2))


def encoding_phase(P, opt: dict, feats: List[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`Seq2SeqBase.encoding_phase` (models/Framework.py:150-187) for the Transformer branch.

    Encoder streams + per-modality means (Encoder.py:85-123), modality filtering for
    the decoder / predictor (Encoder.py:125-153), `Predictor_attribute` with mean
    pooling + channel concat (pred_attribute.py:78-131), `SemanticContainer`
    (pred_attribute.py:262-289 + Embeddings.py:53-87) and the 'concat' of concept rows
    onto the memory (Framework.py:184-185).
    """
    modality = opt["modality"]
    assert len(feats) >= len(modality)
    streams = [encode_modality(P, opt, ch, f) for ch, f in zip(modality, feats)]
    means = [s.mean(1) for s in streams]
    dec_mod = opt.get("modality_for_decoder") or modality
    pred_mod = opt.get("modality_for_predictor") or modality
    out: Dict[str, torch.Tensor] = {}
    dec_streams = [s for ch, s in zip(modality, streams) if ch in dec_mod]
    out["encoder_hidden_states"] = torch.cat(dec_streams, dim=1)  # fusion = temporal_concat
    out["mean_encoder_hidden_states"] = [m for ch, m in zip(modality, means) if ch in dec_mod]

    if "attribute" in opt.get("crits", []):
        pred_means = [m for ch, m in zip(modality, means) if ch in pred_mod]
        assert opt.get("attribute_prediction_mean_pooling") and opt.get("attribute_prediction_channel_concat"), \
            "only the CARE predictor configuration (tasks.yaml:19-20) is on the hot path"
        h = torch.cat(pred_means, dim=-1).unsqueeze(1)  # [B, 1, n_mod*d]
        scores = _linear(P, "predictor.nets.0.prj", h)
        preds_attr, avg_prob = concept_probabilities(scores)
        out["preds_attr"] = preds_attr
        out["avg_prob_attr"] = avg_prob

        if "SemanticContainer" in opt.get("predictors_to_be_added", []):
            sp = "predictor.nets.1"
            _, labels = preds_attr.topk(opt["use_attr_topk"], dim=1, sorted=True, largest=True)
            if "L0" in opt.get("use_attr_flags", ""):  # pred_attribute.py:243-252,276-277: no concept embeddings
                out["semantic_embs"] = None
            else:
                emb = P[sp + ".attr_embs.word_embeddings.weight"][labels]
                emb = emb + P[sp + ".attr_embs.position_embeddings.weight"][: labels.shape[1]].unsqueeze(0)
                out["semantic_embs"] = _layer_norm(P, sp + ".attr_embs.LayerNorm", emb, opt["layer_norm_eps"])
            out["semantic_labels"] = labels
            if "emb" in opt.get("use_attr_type", ""):
                # pred_attribute.py:279: detached unless `global_semantic_guidance_not_detach` (same forward values)
                src = preds_attr if opt.get("global_semantic_guidance_not_detach") else preds_attr.detach()
                out["semantic_hidden_states"] = F.linear(
                    src, P[sp + ".semantic2hidden.weight"], P.get(sp + ".semantic2hidden.bias"))
            if "concat" in opt.get("use_attr_type", ""):
                out["encoder_hidden_states"] = torch.cat(
                    (out["encoder_hidden_states"], out["semantic_embs"]), dim=1)
    return out


# --------------------------------------------------------------------------- decoder
def decoder_embeddings(P, opt: dict, input_ids: torch.Tensor,
                       semantic_hidden_states: Optional[torch.Tensor]) -> torch.Tensor:
    """`Embeddings.forward` (models/components/Embeddings.py:134-188), trainable PE."""
    t = input_ids.shape[1]
    e = P["decoder.embedding.word_embeddings.weight"][input_ids]
    if opt.get("trainable_pe", False):
        pos = P["decoder.embedding.position_embeddings.weight"][:t]
    else:
        pos = P["decoder.embedding.position_embeddings.pe"][0, :t]
    e = e + pos.unsqueeze(0)
    if "emb" in opt.get("use_attr_type", ""):
        e = e + semantic_hidden_states.unsqueeze(1).expand_as(e)
    if opt.get("transformer_pre_ln", False):  # Embeddings.py:130-131: no LayerNorm in a pre-LN decoder's embedding
        return e
    return _layer_norm(P, "decoder.embedding.LayerNorm", e, opt["layer_norm_eps"])


def decoder_forward(P, opt: dict, input_ids: torch.Tensor, inputs: Dict[str, torch.Tensor],
                    aux: Optional[dict] = None) -> torch.Tensor:
    """`TransformerDecoder.forward` (models/Decoder/Transformer.py:161-268), ARFormer branch.

    Self-attention mask = key is PAD OR strictly-future (:169-174); cross mask all-False
    (:179-180); one or more `DecoderLayer`s (Layers.py:157-228: intra -> inter -> ffn).
    Returns `hidden_states [N, t, d]`.
    """
    n, t = input_ids.shape
    key_pad = input_ids.eq(PAD).unsqueeze(1).expand(-1, t, -1)
    causal = torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1).unsqueeze(0)
    self_mask = key_pad | causal
    h = decoder_embeddings(P, opt, input_ids, inputs.get("semantic_hidden_states"))
    memory = inputs["encoder_hidden_states"]
    if aux is not None:  # the other entries of the dict at Decoder/Transformer.py:239-252
        aux.update(all_hidden_states=[h], all_intra_attentions=(), all_inter_attentions=(), attr_attention_probs=(),
                   input_embs=h, input_embs_exclude_bos=h[:, 1:, :],
                   sentence_embs=P["decoder.embedding.word_embeddings.weight"][input_ids])
    for li in range(opt["num_hidden_layers_decoder"]):
        lp = "decoder.layers.{}".format(li)
        a1, a2, a3 = ({}, {}, {}) if aux is not None else (None, None, None)
        h = attention_block(P, lp + ".intra_attention", opt, h, None, self_mask, a1)
        h = attention_block(P, lp + ".inter_attention", opt, h, memory, None, a2)
        if (lp + ".attr_attention.dense.weight") in P:
            # CABase, attr_layer_pos = 'cross2attr' (Layers.py:139-154,218-225): a third post-LN
            # attention block whose keys/values are the concept embeddings, no mask
            assert opt.get("attr_layer_pos", "cross2attr") == "cross2attr"
            h = attention_block(P, lp + ".attr_attention", opt, h, inputs["semantic_embs"], None, a3)
        h = ffn_block(P, lp + ".ffn", opt, h)
        if aux is not None:
            aux["all_hidden_states"].append(h)
            aux["all_intra_attentions"] += (a1["probs"],)
            aux["all_inter_attentions"] += (a2["probs"],)
            aux.update(text_context=a1["context"], self_embs=a1["embs"], context=a2["context"], cross_embs=a2["embs"])
            if a3:
                aux["attr_attention_probs"] += (a3["probs"],)
    if aux is not None:
        aux["attention_probs"] = aux["all_inter_attentions"][-1].mean(1)
    if opt.get("transformer_pre_ln", False):  # Decoder/Transformer.py:80-81,233-234: the final LayerNorm of a pre-LN decoder
        assert opt["encoder"] == "Embedder", "pre-LN with a self-attention encoder is outside the hot path"
        h = _layer_norm(P, "decoder.LayerNorm", h, opt["layer_norm_eps"])
    return h


def decoding_phase(P, opt: dict, input_ids: torch.Tensor, inputs: Dict[str, torch.Tensor],
                   last_time_step_logits: bool = False, auxiliary: bool = False) -> Dict[str, torch.Tensor]:
    """`TransformerSeq2Seq.decoding_phase` (Framework.py:240-269) + `NaiveHead` (Head.py:26-32).
    auxiliary: also the decoder's other dict entries (Decoder/Transformer.py:239-252)."""
    aux = {} if auxiliary else None
    hidden = decoder_forward(P, opt, input_ids, inputs, aux)
    w = P["cls_head.tgt_word_prj.weight"]
    logits = F.linear(hidden[:, -1, :], w) if last_time_step_logits else F.linear(hidden, w)
    return {**(aux or {}), "hidden_states": hidden, "logits": logits}


def inputs_for_decoder(opt: dict, enc: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`get_framework` key list (Framework.py:20-33) + `prepare_inputs_for_decoder` (:189-204)."""
    keys = ["encoder_hidden_states"]
    if opt.get("use_attr", False) and ("prefix" in opt["use_attr_type"] or "att" in opt["use_attr_type"].lower()):
        keys.append("semantic_embs")
    if "emb" in opt.get("use_attr_type", ""):
        keys.append("semantic_hidden_states")
    return {k: enc[k] for k in keys}


def feedforward_step(P, opt: dict, feats: List[torch.Tensor], input_ids: torch.Tensor,
                     auxiliary: bool = False) -> Dict[str, torch.Tensor]:
    """Teacher-forced forward, `Seq2SeqBase.feedforward_step` (Framework.py:215-234)."""
    enc = encoding_phase(P, opt, feats)
    dec = decoding_phase(P, opt, input_ids, inputs_for_decoder(opt, enc), last_time_step_logits=False, auxiliary=auxiliary)
    return {**enc, **dec}


# --------------------------------------------------------------------------- beam search
class HostBeam:
    """Per-clip beam state; semantics of `misc/Decoding/Beam.py` (restated, not copied).

    Quirks kept on purpose (hard part 6): the first step looks only
    at row 0 (:55-56); a beam whose last token is EOS stays in the tensor with all its
    continuations at -1e20 (:52-54); the clip is done once `max(size, topk)` hypotheses
    ended with EOS (:10,38-43) or at `max_len` tokens, when all live beams are taken if
    none finished (:79-84); final ranking by `score / t**alpha` (:91-101).
    """

    def __init__(self, size: int, max_len: int, n_best: int):
        self.size = size
        self.max_len = max_len
        self.need = max(size, n_best)
        self.scores = torch.zeros(size)
        self.parents: List[torch.Tensor] = []
        self.tokens: List[torch.Tensor] = [torch.full((size,), BOS, dtype=torch.long)]
        self.finished: List[list] = []
        self.done = False
        # tie audit (test infrastructure, not part of the reference): the smallest gap this clip's
        # search ever saw at the selection boundary (size-th vs next candidate) and between
        # neighbours inside the selected set
        self.gap_select = float("inf")
        self.gap_order = float("inf")
        # per live beam: how close its ancestry ever came to being pruned (candidate value minus
        # the best candidate that was NOT selected, minimum over the steps so far)
        self.slack = [float("inf")] * size

    def prefixes(self) -> torch.Tensor:
        order = torch.sort(self.scores, 0, True)[1]
        return torch.tensor([self._walk(int(k), len(self.parents), with_bos=True) for k in order], dtype=torch.long)

    def _walk(self, k: int, length: int, with_bos: bool) -> List[int]:
        seq = []
        for j in range(length - 1, -1, -1):
            seq.append(int(self.tokens[j + 1][k]))
            k = int(self.parents[j][k])
        if with_bos:
            seq.append(int(self.tokens[0][k]))
        return seq[::-1]

    def advance(self, logp: torch.Tensor) -> bool:
        vocab = logp.shape[1]
        if self.parents:
            cand = logp + self.scores.unsqueeze(1)
            ended = self.tokens[-1].eq(EOS)
            cand[ended] = -1e20
        else:
            cand = logp[0]
        best, flat = cand.reshape(-1).topk(self.size, 0, True, True)
        wide = cand.reshape(-1).topk(self.size + 1, 0, True, True)[0]
        self.gap_select = min(self.gap_select, float(wide[self.size - 1] - wide[self.size]))
        if self.size > 1:
            self.gap_order = min(self.gap_order, float((wide[:-2] - wide[1:-1]).min()))
        self.scores = best
        parent = flat // vocab
        self.slack = [min(self.slack[int(parent[i])], float(best[i] - wide[self.size])) for i in range(self.size)]
        self.parents.append(parent)
        self.tokens.append(flat - parent * vocab)
        for i in range(self.size):
            if int(self.tokens[-1][i]) == EOS:
                self.finished.append([float(self.scores[i]), len(self.parents), i, self.slack[i]])
                if len(self.finished) >= self.need:
                    self.done = True
                    return True
        if len(self.tokens) == self.max_len:
            self.done = True
            if not self.finished:
                for i in range(self.size):
                    self.finished.append([float(self.scores[i]), len(self.parents), i, self.slack[i]])
        return self.done

    def ranked(self, alpha: float):
        return sorted(([s / (t ** alpha), t, k, sl] for s, t, k, sl in self.finished), key=lambda a: -a[0])

    def hypothesis(self, t: int, k: int) -> List[int]:
        return self._walk(k, t, with_bos=False)


def _repeat_rows(x: torch.Tensor, times: int) -> torch.Tensor:
    # misc/utils.py:244-258 `enlarge`: each clip repeated `beam_size` times, materialised
    return x.unsqueeze(1).repeat(1, times, *([1] * (x.dim() - 1))).reshape(x.shape[0] * times, *x.shape[1:])


def translate_batch(P, opt: dict, feats: List[torch.Tensor], return_trace: bool = False, return_gaps: bool = False):
    """`Translator_ARFormer.translate_batch` (models/Translator.py:35-85) for one model.

    Greedy decoding is beam search with `beam_size == 1` (models/Wrapper.py:34-35).
    Returns `(batch_hyps, batch_scores)` exactly like the reference: python ints without
    BOS and including EOS when emitted; python floats (length-normalised).
    With `return_trace`, also the per-step top-2 log-prob margins of every live row
    (the tie audit of SURVEY.md 8(c)).  With `return_gaps`, also one dict per clip with the
    smallest decision margins of its search: `select` (beam_size-th vs next candidate; for
    greedy the top-1/top-2 log-prob margin), `order` (neighbours inside the selected set),
    `rank` (neighbours among the n_best + 1 best finished hypotheses, length-normalised) and
    `best_slack` (how close the winning hypothesis' ancestry ever came to being pruned).
    """
    return translate_batch_ensemble([P], [opt], [feats], return_trace=return_trace, return_gaps=return_gaps)


def translate_batch_ensemble(Ps, opts, feats_list, return_trace: bool = False, return_gaps: bool = False):
    """`Translator_ARFormer.translate_batch` for a LIST of models (models/Translator.py:39-52,112-133: every model encodes its
    own features - `batch['feats'][index]` when the batch carries one feature list per model, Wrapper.ModelEnsemble - and
    decodes the shared prefixes; the step's word log-probabilities are the models' log_softmax averaged equally, :130-131).
    The search options (beam_size, topk, beam_alpha, max_len) are the translator's, i.e. `opts[0]`'s."""
    opt = opts[0]
    bm = int(opt.get("beam_size", 5))
    n_best = int(opt.get("topk", 1))
    alpha = float(opt.get("beam_alpha", 1.0))
    max_len = int(opt.get("max_len", 30))
    with torch.no_grad():
        all_inputs = []
        for P, o, feats in zip(Ps, opts, feats_list):
            enc = encoding_phase(P, o, feats)
            all_inputs.append({k: _repeat_rows(v, bm) for k, v in inputs_for_decoder(o, enc).items()})
        n_clips = all_inputs[0]["encoder_hidden_states"].shape[0] // bm
        beams = [HostBeam(bm, max_len, n_best) for _ in range(n_clips)]
        active = list(range(n_clips))
        margins = []
        for t in range(1, max_len):
            ids = torch.stack([beams[i].prefixes() for i in active]).view(-1, t)
            logps = [torch.log_softmax(decoding_phase(P, o, ids, inputs, last_time_step_logits=True)["logits"], dim=1)
                     for P, o, inputs in zip(Ps, opts, all_inputs)]
            logp = logps[0] if len(logps) == 1 else torch.stack(logps, dim=0).mean(0)
            if return_trace:
                top2 = logp.topk(2, dim=1)[0]
                margins.append(float((top2[:, 0] - top2[:, 1]).min()))
            logp = logp.view(len(active), bm, -1)
            still = [pos for pos, i in enumerate(active) if not beams[i].advance(logp[pos])]
            if not still:
                break
            if len(still) != len(active):
                # models/Translator.py:145-209: finished clips are removed from every cached tensor
                sel = torch.tensor(still, dtype=torch.long)
                for inputs in all_inputs:
                    for k, v in inputs.items():
                        inputs[k] = v.view(len(active), -1).index_select(0, sel).view(len(still) * bm, *v.shape[1:])
                active = [active[pos] for pos in still]
        # models/Translator.py:211-220: `n_best = min(n_best, len(scores))` is re-assigned
        # inside the loop over clips, so once one clip has fewer finished hypotheses than
        # `topk`, every LATER clip is truncated to that count as well.  Kept as is.
        hyps, scores, gaps = [], [], []
        for b in beams:
            ranked = b.ranked(alpha)
            head = [r[0] for r in ranked[: n_best + 1]]
            gaps.append(dict(select=b.gap_select, order=b.gap_order, best_slack=ranked[0][3],
                             rank=min([a - c for a, c in zip(head, head[1:])] or [float("inf")])))
            n_best = min(n_best, len(ranked))
            hyps.append([b.hypothesis(t, k) for _, t, k, _ in ranked[:n_best]])
            scores.append([s for s, _, _, _ in ranked[:n_best]])
    if return_gaps:
        return hyps, scores, gaps
    if return_trace:
        return hyps, scores, margins
    return hyps, scores


def score_hypothesis(P, opt: dict, inputs: Dict[str, torch.Tensor], hyp: List[int]) -> float:
    """Exact fp32 score of ONE hypothesis of ONE clip as the beam would report it: sum of the
    log-probs of its tokens (teacher-forced on its own prefix) / len**alpha (Beam.py:91-101).
    `inputs` = decoder inputs of that clip (batch dim 1).  Audit helper for the bf16 beam tests."""
    alpha = float(opt.get("beam_alpha", 1.0))
    with torch.no_grad():
        ids = torch.tensor([[BOS] + list(hyp[:-1])], dtype=torch.long)
        logp = torch.log_softmax(decoding_phase(P, opt, ids, inputs, False)["logits"][0], dim=-1)
        total = float(logp[torch.arange(len(hyp)), torch.tensor(hyp)].sum())
    return total / (len(hyp) ** alpha)
