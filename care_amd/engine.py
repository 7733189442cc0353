"""Host-side orchestration of the HIP kernels for the captioning forward path.

`HipEngine` owns the packed device weights and the workspaces, and issues the kernels of
libcare_hip.so (include/care_hip.h) on torch's current HIP stream.  PyTorch is used for
device memory and streams only; every FLOP of the path runs in hand-written kernels.
There is no CPU or eager fallback: without the library or a GPU, calls raise.

Compared with the reference the engine
  * projects the cross-attention K/V of the static memory ONCE per clip and shares them
    between the beams of a clip (reference: every step, every beam copy);
  * keeps an incremental self-attention K/V cache (reference: full-prefix recompute);
  * keeps greedy/beam state on the device for all 29 steps (reference: D2H sync per step);
  * writes the encoder streams and concept rows straight into the [B, Lk, d] memory
    (reference: torch.cat copies).
These are numerically equivalent re-orderings of the same math (decoder is causal and
post-LN, eval-mode dropout is identity).
"""
import collections
import contextlib
import ctypes
import functools
import os
import threading
import weakref
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT_CODES, CARE_BF16, CARE_F32, ptr
from .constants import BOS, EOS, PAD
from .engine_beam import BeamMixin
from .engine_decode import DecodeMixin
from .engine_encode import EncodeMixin
from .engine_resident import ResidentMixin
from .engine_util import _LaneOutputs, device_props  # noqa: F401


_CAPTURE_LOCK = threading.Lock()   # hipGraph captures are serialised across engines and threads (HipEngine._replay)


class HipEngine(EncodeMixin, DecodeMixin, ResidentMixin, BeamMixin):
    def __init__(self, opt: dict, dtype: str = "fp32"):
        if dtype not in ("fp32", "bf16", "fp16", "fp16x3"):
            raise ValueError("compute dtype must be 'fp32', 'bf16', 'fp16' or 'fp16x3', got {!r}".format(dtype))
        self.opt = opt
        self.dtype = dtype
        # The 16-bit modes are ONE set of kernels compiled for two storage types (care_amd/build.py): `bf16` runs
        # libcare_hip.so, `fp16` the same sources with IEEE half as the 16-bit type (libcare_hip_f16.so: 11 significand
        # bits instead of 8 at the same bytes and the same MFMA rate; range 65504 - every 16-bit tensor of this path is a
        # LayerNorm output, an attention context, a projected key / value or an FFN hidden value, all O(1 .. 100); raw
        # features are multiplied from fp32 LDS stages rounded per fragment, so |feature| < 65504 is the one input
        # condition, checked by `check_fp16_range`).  fp32 and fp16x3 modes use the default library.
        self.variant = "f16" if dtype == "fp16" else ""
        self.h16 = torch.float16 if dtype == "fp16" else torch.bfloat16
        self.call = functools.partial(_lib.call, variant=self.variant)
        self.wt = self.h16 if dtype in ("bf16", "fp16") else torch.float32
        # 'fp16x3': fp32 storage everywhere (activations, K/V caches, weights' masters) like 'fp32', but every GEMM
        # multiplies hi/lo fp16 pieces of both operands - a_hi w_hi + a_hi w_lo + a_lo w_hi, three fp16 MFMA passes,
        # what is dropped is ~2^-22 of a product - on the LDS-tiled kernel instead of the exact-f32 MFMA (1/16 of
        # the 16-bit rate): fp32-GRADE results (the reference's 1e-5 bars, token ids as in fp32 mode) at about
        # twice fp32 mode's throughput.  The mode between bf16 (1.5e-2) and exact fp32.
        self.split3 = dtype == "fp16x3"
        self._w3: Dict[int, torch.Tensor] = {}
        self.d = int(opt["dim_hidden"])
        self.H = int(opt["num_attention_heads"])
        if self.d != self.H * 64:
            raise ValueError("the attention kernel is specialised for head dim 64 (archs.yaml:15-26)")
        self.ff = int(opt["intermediate_size"])
        if self.ff % 64:
            raise ValueError("intermediate_size {} is no multiple of 64 (the K of FFN dense2 on the matrix cores)".format(self.ff))
        self.V = int(opt["vocab_size"])
        self.T = int(opt["max_len"]) - 1
        self.eps = float(opt["layer_norm_eps"])
        self.act = ACT_CODES[opt["hidden_act"]]
        self.modality = opt["modality"]
        self.dec_mod = opt.get("modality_for_decoder") or self.modality
        self.pred_mod = opt.get("modality_for_predictor") or self.modality
        self.has_concepts = "attribute" in opt.get("crits", [])
        self.has_container = "SemanticContainer" in opt.get("predictors_to_be_added", [])
        # local guidance off (`use_attr_flags` ..L0, pred_attribute.py:243-252): the container has no concept embeddings -
        # labels and the global-guidance vector only, no concept rows in the memory
        self.has_attr_embs = self.has_container and "L0" not in opt.get("use_attr_flags", "")
        self.use_attr_type = opt.get("use_attr_type", "") if self.has_container else ""
        self.concat = "concat" in self.use_attr_type
        if self.has_container and not self.has_attr_embs and (self.concat or "att" in self.use_attr_type.lower()):
            raise ValueError("use_attr_flags {!r} (no concept embeddings) with use_attr_type {!r}".format(
                opt.get("use_attr_flags"), self.use_attr_type))
        self.sem = "emb" in self.use_attr_type
        self.attr_att = bool(opt.get("use_attr", False)) and "att" in self.use_attr_type.lower()
        self.topk = int(opt.get("use_attr_topk", 30))
        self.k_attr = int(opt.get("attribute_prediction_k", 500))
        self.n_layers = int(opt["num_hidden_layers_decoder"])
        # pre-LN decoder (opts.py:68 `--transformer_pre_ln`; SubLayers.py:55,78,140,149; Embeddings.py:130; Decoder/Transformer.py:80,
        # 233-234): LayerNorm BEFORE every sub-block, the residual sums left as they are, no LayerNorm after the embedding, one
        # final LayerNorm in front of the head.  Runs through the multi-launch unfused forms in every mode (round 5).
        self.pre_ln = bool(opt.get("transformer_pre_ln", False))
        if self.pre_ln and opt["encoder"] != "Embedder":
            raise ValueError("transformer_pre_ln with a self-attention encoder is outside the hot path")
        self.rows_of = {ch: (int(opt["retrieval_topk"]) if ch == "r" else int(opt["n_frames"])) for ch in self.modality}
        # feature widths that are no multiple of 32 are padded with zero columns to the next multiple of 128 (engine_encode._prep_one)
        self.feat_pad = {ch: ((-int(opt["dim_" + ch])) % 128 if int(opt["dim_" + ch]) % 32 else 0) for ch in self.modality}
        self.mem_off = {}
        off = 0
        for ch in self.modality:
            if ch in self.dec_mod:
                self.mem_off[ch] = off
                off += self.rows_of[ch]
        self.concept_off = off
        self.Lk = off + (self.topk if self.concat else 0)
        if self.Lk > 128:
            raise ValueError("memory length {} > 128 keys is outside the attention kernel".format(self.Lk))
        self.w: Dict[str, torch.Tensor] = {}
        self.device = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._ws_used: Dict[tuple, int] = {}     # workspace key -> number of the last pass that asked for it
        self._ws_bytes = 0
        self._gen = 0                            # pass counter (_begin_pass)
        # one pass at a time per engine: its workspaces, result block and graphs are the engine's, not the caller's - threads
        # that share a module take turns (the Translator and the module API hold it for the length of a call)
        self.lock = threading.RLock()
        # rows (clips x beam) the running pass STARTED with: what the row-count switches of a decode step look at
        # (ln_fusable) - compaction shrinks the row count mid-pass, and a clip's arithmetic must not change with it
        self._form_rows: Optional[int] = None
        # byte budget of the cached workspaces (None: 60 % of the device's memory, CARE_WS_BUDGET_GB overrides)
        self.ws_budget_bytes: Optional[int] = None
        self._graphs: "collections.OrderedDict[tuple, object]" = collections.OrderedDict()
        self._lane = 0
        self.lanes = 1  # batch lanes of a graph-replayed greedy pass (lanes_for)
        self.latent = os.environ.get("CARE_LATENT", "1") != "0"
        self._ws_cap = None
        # early termination + active-set compaction of the greedy loop (greedy_early_exit); the fixed
        # 29-step pass remains available (`early_exit=False`, CARE_EARLY_EXIT=0)
        self.early_exit = os.environ.get("CARE_EARLY_EXIT", "1") != "0"
        self.segment_steps = int(os.environ.get("CARE_SEGMENT_STEPS", "4"))
        # greedy batches of up to this many clips decode as ONE resident launch (greedy_resident); 0 turns it off
        self.resident_max_rows = int(os.environ.get("CARE_RESIDENT_MAX_ROWS", "256"))
        # beam search as one resident launch (csrc/decode_resident_beam.hip) up to this many rows = clips x beam_size
        # (translate.py's default: 128 clips x beam 5); 0: the multi-launch search at every size
        self.resident_beam_max_rows = int(os.environ.get("CARE_RESIDENT_BEAM_MAX_ROWS", "640"))
        # ... or as a CHAIN of kernels per step (csrc/decode_chain.hip: the resident launch's phases as launches of their
        # own, bit-identical results, no residency condition) up to this many rows.  OFF by default (0): *measured*
        # (tools/beam_sweep.py, profiles/r05_beam_sweep.txt) the chain is slower than the resident launch wherever that
        # applies (640 rows: 207 against 173 us per step) and slower than the multi-launch search beyond (1280 rows:
        # 370 against 282) - DESIGN.md 4.2f says where its time goes
        self.chain_beam_max_rows = int(os.environ.get("CARE_CHAIN_BEAM_MAX_ROWS", "0"))
        self.chain_segment_steps = int(os.environ.get("CARE_CHAIN_SEGMENT_STEPS", "8"))

    def _code(self, t: Optional[torch.Tensor]) -> int:
        """dtype code of a tensor argument (include/care_hip.h): CARE_F32, or the library's ONE 16-bit code for a tensor
        in THIS engine's 16-bit type.  The other 16-bit type - a bf16 tensor handed to an fp16-mode engine or the reverse -
        would be reinterpreted bit for bit by the kernels: refused here."""
        if t is None or t.dtype == torch.float32:
            return CARE_F32
        if t.dtype == self.h16:
            return CARE_BF16
        raise TypeError("a {} tensor in a `{}`-mode engine (its 16-bit type is {})".format(t.dtype, self.dtype, self.h16))

    @property
    def lib(self):
        """The ctypes library of this engine's compute mode (libcare_hip.so, or libcare_hip_f16.so for 'fp16')."""
        return _lib.load(variant=self.variant)

    # ------------------------------------------------------------------ weights
    def load_weights(self, sd: Dict[str, torch.Tensor], device) -> None:
        """Pack a reference-named state dict into the device layout the kernels read."""
        self.lib  # (loads - and, when stale, rebuilds - the library of this mode: raises without it)
        if not torch.cuda.is_available():
            raise _lib.CareHipError("no HIP device: the captioning path has no CPU fallback")
        self.device = torch.device(device)
        f32 = lambda t: t.detach().to(self.device, torch.float32).contiguous()
        wt = lambda t: t.detach().to(self.device, torch.float32).to(self.wt).contiguous()
        w = {}
        opt, d = self.opt, self.d
        # Concept detection ends in a DISCRETE choice (top-30 of 500, pred_attribute.py:264) whose
        # neighbouring probabilities differ by ~1e-4, below bf16 operand noise (~1.5e-3 measured).
        # So with a concept head the feature-embedding GEMMs keep fp32 operands even in bf16 mode:
        # as three fp16 MFMA passes over hi/lo pieces (care_gemm_ln_split: fp32-grade, memory within
        # ~5e-6 of the reference) where the fused kernel applies, in exact f32 MFMA
        # otherwise (or with CARE_ENC_SPLIT=0); everything downstream of the choice is bf16.
        enc_wt = f32 if (self.has_concepts and opt["encoder"] == "Embedder") else wt
        for ch in self.modality:
            p = "encoder.Encoder_{}".format(ch.upper())
            w["enc_w_" + ch], w["enc_b_" + ch] = enc_wt(sd[p + ".0.weight"]), f32(sd[p + ".0.bias"])
            if self.feat_pad.get(ch, 0):   # (see _prep_one: zero columns against the zero columns appended to the features)
                w["enc_w_" + ch] = torch.nn.functional.pad(w["enc_w_" + ch], (0, self.feat_pad[ch])).contiguous()
            if opt["encoder"] == "Embedder":
                w["enc_g_" + ch], w["enc_be_" + ch] = f32(sd[p + ".1.weight"]), f32(sd[p + ".1.bias"])
            elif opt["encoder"] == "MultiTransformerEncoder":
                q = p + ".1"
                if opt.get("trainable_pe", False):
                    w["enc_pos_" + ch] = f32(sd[q + ".position_embeddings.weight"])
                else:
                    w["enc_pos_" + ch] = f32(sd[q + ".position_embeddings.pe"][0])
                w["enc_g_" + ch], w["enc_be_" + ch] = f32(sd[q + ".LayerNorm.weight"]), f32(sd[q + ".LayerNorm.bias"])
                for li in range(int(opt["num_hidden_layers_encoder"])):
                    self._pack_attn(w, sd, "{}.layers.{}.intra_attention".format(q, li), "enc{}{}_sa".format(ch, li), True, wt, f32)
                    self._pack_ffn(w, sd, "{}.layers.{}.ffn".format(q, li), "enc{}{}_ffn".format(ch, li), wt, f32)
            else:
                raise ValueError("encoder `{}` is outside the hot path (SURVEY.md 8(a))".format(opt["encoder"]))
        if self.has_concepts:
            if not (opt.get("attribute_prediction_mean_pooling") and opt.get("attribute_prediction_channel_concat")):
                raise ValueError("only the CARE concept head (mean pooling + channel concat) is on the hot path")
            w["attr_w"], w["attr_b"] = f32(sd["predictor.nets.0.prj.weight"]), f32(sd["predictor.nets.0.prj.bias"])
            if self.has_container:
                sp = "predictor.nets.1"
                if self.has_attr_embs:
                    w["attr_word"] = f32(sd[sp + ".attr_embs.word_embeddings.weight"])
                    w["attr_pos"] = f32(sd[sp + ".attr_embs.position_embeddings.weight"])
                    w["attr_g"], w["attr_be"] = f32(sd[sp + ".attr_embs.LayerNorm.weight"]), f32(sd[sp + ".attr_embs.LayerNorm.bias"])
                else:
                    w["attr_word"] = w["attr_pos"] = w["attr_g"] = w["attr_be"] = None
                if self.sem:
                    kp = self._kpad()
                    s2h = torch.zeros(d, kp, device=self.device, dtype=torch.float32)
                    s2h[:, : self.k_attr] = f32(sd[sp + ".semantic2hidden.weight"])
                    w["s2h_w"] = s2h
                    b = sd.get(sp + ".semantic2hidden.bias")
                    w["s2h_b"] = f32(b) if b is not None else None
        e = "decoder.embedding"
        w["word"] = f32(sd[e + ".word_embeddings.weight"])
        if opt.get("trainable_pe", False):
            w["pos"] = f32(sd[e + ".position_embeddings.weight"])
        else:
            w["pos"] = f32(sd[e + ".position_embeddings.pe"][0])
        if self.pre_ln:  # no LayerNorm after the embedding sum (Embeddings.py:130); the decoder's final one instead
            w["emb_g"] = w["emb_be"] = None
            w["dec_g"], w["dec_be"] = f32(sd["decoder.LayerNorm.weight"]), f32(sd["decoder.LayerNorm.bias"])
        else:
            w["emb_g"], w["emb_be"] = f32(sd[e + ".LayerNorm.weight"]), f32(sd[e + ".LayerNorm.bias"])
        for li in range(self.n_layers):
            lp = "decoder.layers.{}".format(li)
            self._pack_attn(w, sd, lp + ".intra_attention", "d{}_sa".format(li), True, wt, f32)
            self._pack_attn(w, sd, lp + ".inter_attention", "d{}_ca".format(li), False, wt, f32)
            hb = sd.get(lp + ".inter_attention.SDPA.hybrid_bias")
            if hb is not None and hb.shape[1] != self.Lk:
                raise ValueError("hybrid_bias length {} != memory length {}".format(hb.shape[1], self.Lk))
            w["d{}_hb".format(li)] = f32(hb) if hb is not None else None
            if self.attr_att:
                self._pack_attn(w, sd, lp + ".attr_attention", "d{}_aa".format(li), False, wt, f32)
            self._pack_ffn(w, sd, lp + ".ffn", "d{}_ffn".format(li), wt, f32)
        w["vocab"] = wt(sd["cls_head.tgt_word_prj.weight"])
        if self.as_ok and d == 512:
            # weights of the fused Linear -> LayerNorm GEMMs also in the K-step-major order the kernel
            # streams (csrc/gemm_ln.hip, care_pack_ln_weight): 1 KB of full cache lines per DMA instruction
            for name in [k for k in w if k.startswith("enc_w_") or k.endswith("_o_w") or k.endswith("_ffn_w2")]:
                W = w[name]
                if W is not None and W.dtype == self.h16 and W.shape[0] == 512 and W.shape[1] % 128 == 0:
                    Wp = torch.empty_like(W)
                    self.call("care_pack_ln_weight", ptr(W), ptr(Wp), 512, W.shape[1])
                    w[name + "#packed"] = Wp
                elif (W is not None and name.startswith("enc_w_") and W.dtype == torch.float32 and W.shape[0] == 512 and
                      W.shape[1] % 64 == 0 and opt["encoder"] == "Embedder" and os.environ.get("CARE_ENC_SPLIT", "1") != "0"):
                    Ws = torch.empty(3 * W.shape[1] * 512, device=self.device, dtype=self.h16)
                    self.call("care_pack_ln_weight_split", ptr(W), ptr(Ws), 512, W.shape[1])
                    w[name + "#split"] = Ws
        if self.bf and self.has_concepts and opt["encoder"] == "Embedder" and os.environ.get("CARE_ENC_SPLIT", "1") != "0":
            # concept models the fused kernel does not cover (d_model != 512): the embedder GEMM with split products
            # through the generic kernel (care_gemm_split3) instead of its exact-f32 MFMA
            for ch in self.modality:
                W = w["enc_w_" + ch]
                # (d_model = 512 too: small batches take this form instead of the fused kernel, see encode(small=True))
                if W.dtype == torch.float32 and W.shape[1] % 64 == 0:
                    W3 = torch.empty(W.shape[0], 3 * W.shape[1], device=self.device, dtype=torch.float16)
                    self.call("care_split3_weight", ptr(W), ptr(W3), W.shape[0], W.shape[1])
                    w["enc_w_" + ch + "#split3"] = W3
        self._w3 = {}
        if self.split3:
            for name, W in w.items():
                if (isinstance(W, torch.Tensor) and W.dim() == 2 and W.dtype == torch.float32 and W.shape[1] % 64 == 0 and
                        "#" not in name and not name.startswith(("word", "pos", "attr_word", "attr_pos", "enc_pos_"))):
                    W3 = torch.empty(W.shape[0], 3 * W.shape[1], device=self.device, dtype=torch.float16)
                    self.call("care_split3_weight", ptr(W), ptr(W3), W.shape[0], W.shape[1])
                    self._w3[W.data_ptr()] = W3
        self.w = w
        self._graphs.clear()
        self._epoch = getattr(self, "_epoch", 0) + 1   # (graphs of OTHER engines that stepped this one as an ensemble member: their keys carry it)

    def _pack_attn(self, w, sd, p, name, self_attn, wt, f32):
        wq, wk, wv = (sd[p + ".SDPA.{}.weight".format(n)] for n in ("query", "key", "value"))
        bq, bk, bv = (sd.get(p + ".SDPA.{}.bias".format(n)) for n in ("query", "key", "value"))
        zeros = lambda m: torch.zeros(m.shape[0], dtype=torch.float32)
        bq, bk, bv = (b if b is not None else zeros(m) for b, m in ((bq, wq), (bk, wk), (bv, wv)))
        if self_attn:  # one [3d, d] projection
            w[name + "_qkv_w"] = wt(torch.cat([wq, wk, wv], 0))
            w[name + "_qkv_b"] = f32(torch.cat([bq, bk, bv], 0))
        else:
            w[name + "_q_w"], w[name + "_q_b"] = wt(wq), f32(bq)
            w[name + "_kv_w"], w[name + "_kv_b"] = wt(torch.cat([wk, wv], 0)), f32(torch.cat([bk, bv], 0))
            if self.latent_capable:
                # absorbed cross-attention (csrc/attention_latent.hip): W_k moves to the query side as
                # wkt[h][c][e] = W_k[h*64+e][c] / sqrt(64) (b_k only shifts every score of a head by one
                # constant, which the softmax cancels); W_v, b_v are applied to the latent context
                H = self.H
                wkt = wk.detach().to(torch.float32).view(H, 64, self.d).permute(0, 2, 1) * 0.125
                w[name + "_wkt"] = wkt.contiguous().to(self.device, self.h16)
                w[name + "_v_w"], w[name + "_v_b"] = wt(wv), f32(bv)
        w[name + "_o_w"], w[name + "_o_b"] = wt(sd[p + ".dense.weight"]), f32(sd[p + ".dense.bias"])
        w[name + "_g"], w[name + "_be"] = f32(sd[p + ".LayerNorm.weight"]), f32(sd[p + ".LayerNorm.bias"])

    def _pack_ffn(self, w, sd, p, name, wt, f32):
        w[name + "_w1"], w[name + "_b1"] = wt(sd[p + ".dense1.weight"]), f32(sd[p + ".dense1.bias"])
        w[name + "_w2"], w[name + "_b2"] = wt(sd[p + ".dense2.weight"]), f32(sd[p + ".dense2.bias"])
        w[name + "_g"], w[name + "_be"] = f32(sd[p + ".LayerNorm.weight"]), f32(sd[p + ".LayerNorm.bias"])

    def _kpad(self) -> int:
        return (self.k_attr + 31) // 32 * 32

    # ------------------------------------------------------------------ helpers
    def ws(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        """Named, cached workspace (allocated on first use, reused afterwards); one namespace per
        batch lane (lanes_for), so concurrent lanes never share a buffer."""
        caps = self._ws_cap
        if caps is not None and len(shape) >= 1:
            # compacted decode (greedy_early_exit / beam_early_exit): `n` active slots of `cap` use the first
            # n rows of ONE full-size buffer instead of a workspace of their own for every row count
            for cur, cap in (caps if isinstance(caps, list) else [caps]):
                if shape[0] == cur and cur != cap:
                    return self._ws_get(name, (cap,) + tuple(shape[1:]), dtype)[:cur]
        return self._ws_get(name, shape, dtype)

    def _ws_get(self, name, shape, dtype):
        key = (self._lane, name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = torch.empty(shape, device=self.device, dtype=dtype)
            self._ws[key] = t
            self._ws_bytes += t.numel() * t.element_size()
        self._ws_used[key] = self._gen
        return t

    # While a segmented pass waits for the host-visible counter between two of its segments (greedy_early_exit,
    # beam_early_exit, the chained beam step), the caller may have host work of its own: `idle_hook` (None, or a callable that
    # does ONE small piece of it and returns False once it has none left) is called in that wait instead of blocking in the
    # copy.  The Translator's pipelined entry assembles the previous batch's python lists there (care_amd/translator.py).
    idle_hook = None

    def _host_count(self, cnt: torch.Tensor) -> int:
        """The value of a one-element device counter, on the host: the one round trip per segment of the segmented passes."""
        hook = self.idle_hook
        if hook is None:
            return int(cnt.item())
        if getattr(self, "_cnt_pin", None) is None:
            self._cnt_pin = torch.empty(1, dtype=torch.int32).pin_memory()
            self._cnt_ev = torch.cuda.Event()
        self._cnt_pin.copy_(cnt, non_blocking=True)
        self._cnt_ev.record()
        while not self._cnt_ev.query():
            if not hook():
                break
        self._cnt_ev.synchronize()
        return int(self._cnt_pin[0])

    def ws_block(self, name: str, parts):
        """The RESULTS of a decode as views of ONE cached workspace: `parts` = [(shape, dtype), ...] -> the tensors, each
        256-byte aligned inside the block, in the order given.  What the Translator hands back lives in a single
        contiguous range, so one device-to-host copy fetches it (care_amd/translator.py::_fetch)."""
        sizes = [int(torch.empty(0, dtype=dt).element_size()) for _, dt in parts]
        offs, total = [], 0
        for (shape, dt), es in zip(parts, sizes):
            offs.append(total)
            n = es
            for s in shape:
                n *= int(s)
            total += (n + 255) // 256 * 256
        blk = self._ws_get(name, (max(total, 256),), torch.uint8)
        out = []
        for (shape, dt), es, off in zip(parts, sizes, offs):
            n = es
            for s in shape:
                n *= int(s)
            out.append(blk[off: off + n].view(dt).view(tuple(shape)))
        return out

    # Captured graphs per kind (first element of the key).  The whole-pass kinds are keyed on the caller's feature
    # buffers, the segment kinds on (buffer set, first step, rows): a loader with ragged last batches, or a caller
    # that allocates fresh feature tensors for every batch, must not grow them without limit.
    GRAPH_CAPS = {"greedy": 8, "beam": 8, "gseg0": 8, "bseg0": 8, "gseg": 96, "bseg": 96, "tf": 8, "gres": 8}

    def _graph_get(self, key):
        entry = self._graphs.get(key)
        if entry is not None:
            self._graphs.move_to_end(key)
        return entry

    def _graph_put(self, key, entry):
        """Insert / update a graph entry; the least recently used entries of the same kind beyond its cap go
        (a captured graph releases its private memory pool with its last reference; 'seen' markers go too)."""
        self._graphs[key] = entry
        self._graphs.move_to_end(key)
        cap = self.GRAPH_CAPS.get(key[0], 8)
        same = [k for k in self._graphs if k[0] == key[0]]
        for k in same[: max(0, len(same) - cap)]:
            del self._graphs[k]

    def _begin_pass(self):
        """Top of every public pass (never inside one): count it, and when the cached workspaces exceed their byte
        budget drop the least recently used ones - all but those of the previous pass, so a loop over one batch shape
        stays warm.  Captured graphs hold raw workspace addresses, so every graph goes with them (the next passes
        run eagerly once and re-capture)."""
        self._gen += 1
        self._form_rows = None
        self._small_pass = False
        budget = self.ws_budget_bytes
        if budget is None:
            env = os.environ.get("CARE_WS_BUDGET_GB")
            budget = int(float(env) * (1 << 30)) if env else int(0.6 * device_props(self.device).total_memory)
            self.ws_budget_bytes = budget
        if self._ws_bytes <= budget:
            return
        for key in sorted(self._ws, key=lambda k: self._ws_used.get(k, 0)):
            if self._ws_bytes <= budget // 2 or self._ws_used.get(key, 0) >= self._gen - 1:
                break
            t = self._ws.pop(key)
            self._ws_used.pop(key, None)
            self._ws_bytes -= t.numel() * t.element_size()
        self._graphs.clear()
        self._epoch = getattr(self, "_epoch", 0) + 1   # (graphs of OTHER engines that stepped this one as an ensemble member: their keys carry it)

    @property
    def bf(self) -> bool:
        return self.dtype in ("bf16", "fp16")

    @property
    def as_ok(self) -> bool:
        """bf16 mode AND d_model fits the A-stationary kernel (K = d <= 512, d % 128 == 0):
        GEMM-input activations then live as bf16 mirrors.  Otherwise (fp32 mode, d = 768/1024)
        every GEMM takes fp32 activations through the generic kernel."""
        return self.bf and self.d <= 512 and self.d % 128 == 0

    @property
    def bf_act(self) -> bool:
        """bf16 mode with bf16 MIRRORS of the GEMM-input activations (any d_model % 64 == 0): K = d <= 512 goes to
        the A-stationary kernels (as_ok), larger K (d_model 768 / 1024, their FFNs) to the LDS-tiled bf16 kernel
        (csrc/gemm_tile.hip)."""
        return self.bf and self.d % 64 == 0 and self.ff % 64 == 0

    @property
    def latent_capable(self) -> bool:
        """Shapes / dtype the absorbed cross-attention kernels cover: bf16 mode, head dim 64, d_model = 512 (one wave
        per row; per-head projections in csrc/heads.hip), 1024 (two waves per row of 512 dims each) or 768 (three waves
        of 256; per-head projections of both as one batched launch of the LDS-tiled GEMM)."""
        if self.H * 64 != self.d or self.H > 16:
            return False
        if os.environ.get("CARE_LATENT_WIDE", "1") == "0" and self.d != 512:
            return False
        return (self.as_ok and self.d == 512) or (self.bf_act and self.d in (768, 1024))

    @property
    def latent_ok(self) -> bool:
        """Absorbed cross-attention (one bf16 copy of the memory instead of projected K and V) is in
        use; `self.latent = False` (or CARE_LATENT=0) keeps the projected-K/V kernels."""
        return bool(self.latent) and self.latent_capable

    @property
    def act_dtype(self):
        """dtype of activations that are ONLY GEMM inputs (attention context, FFN hidden)."""
        return self.h16 if self.bf_act else torch.float32

    def wsb(self, name: str, shape) -> Optional[torch.Tensor]:
        """bf16 mirror workspace of a GEMM-input activation (None unless as_ok)."""
        return self.ws(name + "#bf", shape, self.h16) if self.bf_act else None

    # *measured* (round 5, tools/greedy_sweep.py / beam_sweep.py with CARE_FORCE_TILE = 0 / 1, us per decoder step of the whole
    # pass): greedy 512 clips 137 / 141, 1024 166 / 167, 2048 248 / 232, 4096 365 / 339, 8192 590 / 568; beam 5 over 256
    # clips (1280 rows) 284 / 272, 512 357 / 315, 1024 528 / 496, 2048 (10240 rows) 732 / 700; at 20480 rows the pass does
    # not move and at 32768 the A-stationary kernels win in situ
    MID_TILE_ROWS = (1280, 16384)
    # small batches (the resident decodes): frame rows from which the embedder runs as the fused loader-wave kernel instead
    # of GEMM + LayerNorm launches side by side (*measured* whole pass, fused / unfused: 64 clips 1.709 / 1.652 ms, 128 clips 1.809 /
    # 1.803, 256 clips 2.637 / 2.689: from 192 clips; CARE_FUSED_SMALL_MIN_ROWS overrides)
    FUSED_SMALL_MIN_ROWS = int(os.environ.get("CARE_FUSED_SMALL_MIN_ROWS", "5376"))

    def gemm(self, A, W, bias, out, act=0, out2=None, n_split=None, tag=None, tile=False):
        """out = act(A @ W^T + bias).  bf16 weights + bf16 A -> A-stationary kernel (csrc/gemm_as.hip) for
        K <= 512, the LDS-tiled bf16 kernel (csrc/gemm_tile.hip) for larger K; anything else -> the generic
        fp32-activation kernel (csrc/gemm.hip)."""
        M, K = A.shape
        N = W.shape[0]
        assert W.shape[1] == K and A.stride(1) == 1 and out.stride(-1) == 1
        tail = (ptr(bias), ptr(out), out.stride(0), self._code(out), ptr(out2),
                out2.stride(0) if out2 is not None else 0, self._code(out2), N if n_split is None else n_split, M, N, K, act)
        if W.dtype == self.h16 and A.dtype == self.h16:
            if K % 64 or A.stride(0) % 8:
                raise ValueError("bf16 A operand needs K % 64 == 0 and a 16-byte aligned row stride (got K = {})".format(K))
            # K <= 512: the A-stationary kernel (a 128-row panel's activations in registers for the whole K) - except
            # between MID_TILE_ROWS rows, where its 128 / 256-row panels leave most of the chip idle and the LDS-tiled
            # kernel's 128 x 128 tiles do not.  The two kernels add K in the same order: BIT-IDENTICAL outputs
            # (tests/test_gpu_kernels.py::test_tile_and_a_stationary_gemm_agree_bit_for_bit), so the switch is by the
            # CURRENT row count and changes no caption.
            mid = self.MID_TILE_ROWS[0] <= M < self.MID_TILE_ROWS[1] and os.environ.get("CARE_FORCE_TILE", "") != "0"
            if K <= 512 and K % 128 == 0 and not tile and not mid and os.environ.get("CARE_FORCE_TILE", "0") != "1":
                self.call("care_gemm_bf16", ptr(A), A.stride(0), self._code(A), ptr(W), *tail, tag=tag)
            else:
                self.call("care_gemm_tile", ptr(A), A.stride(0), ptr(W), *tail, tag=tag)
        else:
            if A.dtype != torch.float32:
                raise ValueError("generic GEMM takes fp32 activations")
            W3 = self._w3.get(W.data_ptr()) if self.split3 else None
            if W3 is not None and A.stride(0) % 4 == 0:
                a2 = self.ws("split_a2", (M, 2 * K), torch.float16)
                self.call("care_split2_act", ptr(A), A.stride(0), ptr(a2), M, K)
                self.call("care_gemm_tile_split3", ptr(a2), ptr(W3), *tail, tag=tag)
            else:
                self.call("care_gemm", ptr(A), A.stride(0), ptr(W), self._code(W), *tail, tag=tag)
        return out

    # up to this many rows the vocabulary arg-max of a d_model <= 512 model runs on the LDS-tiled kernel too (measured crossover
    # between 2048 and 4096 rows on msrvtt_base_ami: 2048 rows 246 -> 241 us / step, 4096 rows 346 -> 356)
    VOCAB_TILE_MAX_ROWS = int(os.environ.get("CARE_VOCAB_TILE_MAX_ROWS", "2048"))

    def _vocab_as(self, rows: int) -> bool:
        """The A-stationary vocabulary kernel (against the LDS-tiled one)?  Decided by the pass's INITIAL row count
        (`_form_rows`), like ln_fusable: compaction must not move a clip from one kernel's summation order to the
        other's mid-pass."""
        return self.as_ok and (self._form_rows or rows) > self.VOCAB_TILE_MAX_ROWS

    def vocab_parts(self, rows: int) -> int:
        """Column groups per row of the fused vocabulary arg-max for `rows` rows (the kernel vocab_argmax picks)."""
        if self._vocab_as(rows):
            return self.lib.care_argmax_parts_bf16(rows, self.V)
        if self.bf_act or (self.split3 and self.w["vocab"].data_ptr() in self._w3):
            return self.lib.care_argmax_parts_tile(self.V)
        return self.lib.care_argmax_parts(self.V)

    def vocab_argmax(self, x, xb, rows, pmax, pidx, psum, labels=None, plab=None, tag="step_vocab_argmax"):
        """Per-row (max, arg-max, sum exp) partials of the vocabulary projection of the last hidden state
        (NaiveHead + log_softmax + top-1, Head.py:26-32 / Translator.py:127); the [rows, V] logits never exist.
        bf16, d <= 512: A-stationary kernels; bf16, larger d: the LDS-tiled kernel; fp32 mode: exact-f32 MFMA."""
        d, W = self.d, self.w["vocab"]
        if self._vocab_as(rows):
            self.call("care_gemm_argmax_bf16", ptr(xb), d, self._code(xb), ptr(W), ptr(pmax), ptr(pidx), ptr(psum), ptr(labels),
                 ptr(plab), rows, self.V, d, tag=tag)
        elif self.bf_act:
            self.call("care_gemm_tile_argmax", ptr(xb), d, ptr(W), ptr(pmax), ptr(pidx), ptr(psum), ptr(labels), ptr(plab),
                 rows, self.V, d, tag=tag)
        else:
            if labels is not None:
                raise ValueError("label logits come from the bf16 kernels only (fp32 mode scores materialised logits)")
            W3 = self._w3.get(W.data_ptr()) if self.split3 else None
            if W3 is not None:
                a2 = self.ws("split_a2v", (rows, 2 * d), torch.float16)
                self.call("care_split2_act", ptr(x), x.stride(0), ptr(a2), rows, d)
                self.call("care_gemm_tile_split3_argmax", ptr(a2), ptr(W3), ptr(pmax), ptr(pidx), ptr(psum), rows, self.V, d, tag=tag)
            else:
                self.call("care_gemm_argmax", ptr(x), d, ptr(W), self._code(W), ptr(pmax), ptr(pidx), ptr(psum), rows, self.V, d, tag=tag)

    def add_ln(self, x, res, g, be, out, outb=None, grp=None, out_grp_rows=None, out_row_off=0, pos=None, nslab=1, tag=None):
        """out = LN(sum of the nslab slabs of x + res); x is [rows, d] or [nslab, rows, d]."""
        rows, d = x.shape[-2], x.shape[-1]
        grp = rows if grp is None else grp
        out_grp_rows = grp if out_grp_rows is None else out_grp_rows
        self.call("care_add_ln", ptr(x), x.stride(-2), ptr(res), res.stride(0) if res is not None else 0, ptr(pos), ptr(g),
             ptr(be), self.eps, ptr(out), ptr(outb), out.stride(-2), rows, d, grp, out_grp_rows, out_row_off,
             nslab, x.stride(0) if nslab > 1 else 0, tag=tag)
        return out

    # Unfused FFN2 (K = ff) from this many rows on the LDS-tiled kernel (one product over the whole K, no slabs) instead of
    # the split-K slabs of the A-stationary kernel
    FFN2_TILE_MIN_ROWS = 1 << 30

    def ln_fusable(self, rows: int) -> bool:
        """Whether dense -> (+res) -> LayerNorm runs as ONE kernel (csrc/gemm_ln.hip): bf16 mode,
        d_model = 512, and enough 64-row panels to occupy the chip (below ~10 K rows the A-stationary
        GEMM + LayerNorm kernel pair is faster)."""
        # *measured* (Base `ami`, whole pass): 8192 rows 435 K captions/s unfused vs 420 K fused, 12288 rows
        # 440 K vs 446-451 K
        return self.as_ok and self.d == 512 and not self.pre_ln and rows >= int(os.environ.get("CARE_LN_MIN_ROWS", "10240"))

    # -- the two halves of a sub-block's LayerNorm placement (post-LN: after the residual sum; pre-LN: in front of the block)
    def _ln_in(self, x, xb, g, be, tag):
        """The sub-block's input: x itself (post-LN), or LayerNorm(x) (pre-LN, SubLayers.py:55,140) with its 16-bit mirror."""
        if not self.pre_ln:
            return x, xb
        h, hb = self.ws(tag + "preln", tuple(x.shape)), self.wsb(tag + "preln", tuple(x.shape))
        self.add_ln(x, None, g, be, h, hb)
        return h, hb

    def _res_ln(self, o, x, g, be, out, outb, **kw):
        """The sub-block's output: LayerNorm(o + x) (post-LN, SubLayers.py:73-79), or o + x as it is (pre-LN)."""
        if self.pre_ln:
            return self.add_ln(o, x, None, None, out, outb, **kw)
        return self.add_ln(o, x, g, be, out, outb, **kw)

    def gemm_ln(self, A, W, bias, res, g, be, out, outb, grp=None, out_grp_rows=None, out_row_off=0, pos=None, tag=None,
                Wp=None):
        """out = LN(A W^T + bias + res [+ pos]); Wp: the same weight in care_pack_ln_weight order (preferred)."""
        rows, K = A.shape
        grp = rows if grp is None else grp
        out_grp_rows = grp if out_grp_rows is None else out_grp_rows
        if Wp is not None and pos is None and os.environ.get("CARE_LN_PACKED", "1") != "0":
            self.call("care_gemm_ln_packed", ptr(A), A.stride(0), self._code(A), ptr(Wp), ptr(bias), ptr(res),
                 res.stride(0) if res is not None else 0, ptr(g), ptr(be), self.eps, ptr(out), ptr(outb),
                 (out if out is not None else outb).stride(-2), rows, self.d, K, grp, out_grp_rows, out_row_off, tag=tag)
            return out
        self.call("care_gemm_ln", ptr(A), A.stride(0), self._code(A), ptr(W), ptr(bias), ptr(res),
             res.stride(0) if res is not None else 0, ptr(pos), ptr(g), ptr(be), self.eps, ptr(out), ptr(outb),
             (out if out is not None else outb).stride(-2), rows, self.d, K, grp, out_grp_rows, out_row_off, tag=tag)
        return out

    def attention(self, Q, K, V, ctx, kv_batch_stride, kv_row_stride, rows_per_kv, nkeys, anc=None, causal=False,
                  seq=1, pad_tok=None, bias=None, tag=None):
        rows = Q.shape[0]
        self.call("care_attention", ptr(Q), Q.stride(0), ptr(K), ptr(V), self._code(K), kv_batch_stride, kv_row_stride,
             rows_per_kv, ptr(anc), anc.stride(0) if anc is not None else 0, nkeys, 1 if causal else 0, seq, 0,
             ptr(pad_tok), pad_tok.stride(0) if pad_tok is not None else 0, PAD, ptr(bias),
             bias.stride(0) if bias is not None else 0, ptr(ctx), ctx.stride(0), self._code(ctx), rows, self.H, tag=tag)
        return ctx

    def attention_probs(self, Q, K, kv_batch_stride, kv_row_stride, rows_per_kv, nkeys, causal=False, seq=1,
                        pad_tok=None, bias=None):
        """[rows, H, nkeys] fp32 probabilities of one attention (auxiliary outputs of the teacher-forced
        forward; the fused attention kernels never materialise them)."""
        rows = Q.shape[0]
        probs = torch.empty(rows, self.H, nkeys, device=self.device)
        self.call("care_attention_probs", ptr(Q), Q.stride(0), ptr(K), self._code(K), kv_batch_stride, kv_row_stride, rows_per_kv,
             nkeys, 1 if causal else 0, seq, ptr(pad_tok), pad_tok.stride(0) if pad_tok is not None else 0, PAD,
             ptr(bias), bias.stride(0) if bias is not None else 0, ptr(probs), rows, self.H)
        return probs

    def _replay(self, key, fn, use_graph=True):
        """fn() eagerly the first time `key` is seen (allocates every workspace), captured into a hipGraph the
        second time, replayed afterwards.  Returns fn's result (static tensors once captured)."""
        if not use_graph:
            return fn()
        entry = self._graph_get(key)
        if entry is None:
            self._graph_put(key, "seen")
            return fn()
        if entry == "seen":
            # ONE capture at a time in the process (torch registers the default generator's state with the graph being
            # captured: two threads capturing at once abort the process - "The graph should be registered to the state",
            # found by tools/soak.py), and in thread-local error mode, so that what OTHER threads do meanwhile - launches,
            # allocations, event queries of their own passes - does not invalidate this capture
            with _CAPTURE_LOCK:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = fn()
            entry = (graph, out)
            self._graph_put(key, entry)
        entry[0].replay()
        return entry[1]
