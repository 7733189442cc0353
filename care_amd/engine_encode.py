"""HipEngine, the encoder side: feature preparation, `encode` (Embedder / MultiTransformerEncoder, concept head, semantic
container - reference models/Framework.py:150-187, Encoder.py, Predictor/pred_attribute.py), the static keys / values of the
cross-attention (projected once per clip, or the absorbed form's bf16 memory).  Methods of care_amd.engine.HipEngine."""
import contextlib
import ctypes
import os
import weakref
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT_CODES, CARE_BF16, CARE_F32, ptr
from .constants import BOS, EOS, PAD
from .engine_util import _LaneOutputs


class EncodeMixin:
    # ------------------------------------------------------------------ encoder + concept head
    @property
    def lean_ok(self) -> bool:
        """The captioning loop of a model WITHOUT a concept head consumes nothing of the encoder but the
        bf16 memory (the A operand of the cross-K/V projection, or what the absorbed cross-attention
        reads).  `encode(..., lean=True)` then skips what nobody reads: the fp32 copy of the memory
        (5.6 GB of stores at B = 32768) and the per-modality frame means (a second pass over it)."""
        if os.environ.get("CARE_LEAN", "1") == "0":  # A/B switch
            return False
        return (self.as_ok and self.d == 512 and not self.has_concepts and self.opt["encoder"] == "Embedder" and
                all(ch in self.dec_mod and int(self.opt["dim_" + ch]) % 32 == 0 for ch in self.modality))

    @property
    def feats_bf16_ok(self) -> bool:
        """bf16 feature tensors are taken as they are (no widening copy): the fused embedder of a model without a
        concept head multiplies bf16-rounded features anyway, so features a loader rounded on the host (the same
        round-to-nearest-even) give bit-identical products at half the PCIe / HBM bytes."""
        return (self.as_ok and self.d == 512 and not self.has_concepts and self.opt["encoder"] == "Embedder" and
                all(int(self.opt["dim_" + ch]) % 128 == 0 for ch in self.modality))

    def _prep_one(self, f, ch=None):
        if f.dtype == self.h16 and self.feats_bf16_ok:
            return f.to(self.device).contiguous()
        f = f.to(self.device, torch.float32)
        pad = self.feat_pad.get(ch, 0) if ch is not None else 0
        if pad and f.shape[-1] + pad == self.w["enc_w_" + ch].shape[1]:
            # a feature width that is no multiple of 32 (no shipped extractor; feats.yaml): zero columns up to the next multiple
            # of 128, against zero columns of the weight (load_weights) - one more copy of the features, every kernel's K rule met
            f = torch.nn.functional.pad(f, (0, pad))
        return f.contiguous()

    def _prep_feats(self, feats):
        return [self._prep_one(f, ch) for f, ch in zip(feats[: len(self.modality)], self.modality)]

    def encode(self, feats: List[torch.Tensor], lean: bool = False, static: bool = False, small: bool = False) -> Dict[str, torch.Tensor]:
        """`Seq2SeqBase.encoding_phase` (models/Framework.py:150-187); outputs are fresh tensors.
        lean (translate path only, see lean_ok): returns just {"encoder_hidden_states": bf16 memory}.
        static (translate path only): the memory lives in engine-owned buffers that the next call
        overwrites - so that decode segments captured as hipGraphs keep reading valid addresses.
        small (the resident decode's batches, <= resident_max_rows clips): the embedder as GEMM + LayerNorm launches
        instead of the fused kernel, whose 128-row blocks leave most of the chip idle below ~1000 clips (*measured*
        128 clips: 118 us per modality fused)."""
        w, d, opt = self.w, self.d, self.opt
        if len(feats) < len(self.modality):
            raise ValueError("expected {} feature tensors, got {}".format(len(self.modality), len(feats)))
        B = feats[0].shape[0]
        lean = lean and self.lean_ok
        new = (lambda name, shape, dt=torch.float32: self.ws("enc_out_" + name, shape, dt)) if static else \
              (lambda name, shape, dt=torch.float32: torch.empty(shape, device=self.device, dtype=dt))
        mem = None if lean else new("mem", (B, self.Lk, d))
        memb = new("memb", (B, self.Lk, d), self.h16) if self.bf_act else None
        means = None if lean else new("means", (B, len(self.modality) * d))
        # small batches, Embedder: the modalities' launches are a few microseconds of latency-bound work each - they run
        # side by side on streams of their own (forked from / joined to the caller's stream; also inside a capture)
        cur = torch.cuda.current_stream()
        side = []
        if small and opt["encoder"] == "Embedder" and len(self.modality) > 1:
            if len(getattr(self, "_enc_streams", ())) < len(self.modality) - 1:
                self._enc_streams = [torch.cuda.Stream(device=self.device) for _ in range(len(self.modality) - 1)]
            side = self._enc_streams[: len(self.modality) - 1]
        for mi, ch in enumerate(self.modality):
            st = side[mi - 1] if side and mi > 0 else None
            sfx = "_" + ch if side else ""
            if st is not None:
                st.wait_stream(cur)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                x = self._prep_one(feats[mi], ch)
                n = x.shape[1]
                if n != self.rows_of[ch]:
                    raise ValueError("modality `{}`: {} rows, expected {}".format(ch, n, self.rows_of[ch]))
                x2 = x.view(B * n, x.shape[2])
                Ws = w.get("enc_w_" + ch + "#split")
                fused = (opt["encoder"] == "Embedder" and self.as_ok and d == 512 and
                         (w["enc_w_" + ch].dtype == self.h16 or Ws is not None) and x2.shape[1] % 32 == 0)
                if small and fused and (Ws is None or w.get("enc_w_" + ch + "#split3") is not None):
                    # (concept models: the split products through the LDS-tiled kernel, below) - unless the loader-wave
                    # kernel takes the rows (version 3 of csrc/gemm_ln.hip: whole 128-row blocks, FUSED_SMALL_MIN_ROWS of them)
                    fused = (B * n) % 128 == 0 and B * n >= self.FUSED_SMALL_MIN_ROWS and x2.shape[1] % 128 == 0
                W3 = w.get("enc_w_" + ch + "#split3")
                if fused:
                    lin = None
                elif W3 is not None:
                    lin = self.ws("enc_lin" + sfx, (B * n, d))
                    if os.environ.get("CARE_ENC_TILE", "1") != "0":  # fp16 pieces of the features once, then the LDS-tiled kernel
                        a2 = self.ws("enc_a2" + sfx, (B * n, 2 * x2.shape[1]), torch.float16)
                        self.call("care_split2_act", ptr(x2), x2.stride(0), ptr(a2), B * n, x2.shape[1], tag="enc_split")
                        self.call("care_gemm_tile_split3", ptr(a2), ptr(W3), ptr(w["enc_b_" + ch]), ptr(lin), lin.stride(0), CARE_F32,
                             None, 0, 0, d, B * n, d, x2.shape[1], 0, tag="enc_gemm")
                    else:
                        self.call("care_gemm_split3", ptr(x2), x2.stride(0), ptr(W3), ptr(w["enc_b_" + ch]), ptr(lin), lin.stride(0),
                             B * n, d, x2.shape[1], tag="enc_gemm")
                else:
                    lin = self.gemm(x2, w["enc_w_" + ch], w["enc_b_" + ch], self.ws("enc_lin" + sfx, (B * n, d)), tag="enc_gemm")
                in_mem = ch in self.dec_mod
                if in_mem:
                    dst, dstb, grp_rows, off = mem, memb, self.Lk, self.mem_off[ch]
                    if dst is None and not fused:  # lean + unfused: the LayerNorm kernel writes an fp32 row too
                        dst = self.ws("enc_mem_f32", (B, self.Lk, d))
                else:
                    dst, dstb, grp_rows, off = self.ws("enc_side_" + ch, (B, n, d)), None, n, 0
                ln_kw = dict(grp=n, out_grp_rows=grp_rows, out_row_off=off)
                if fused and Ws is not None:  # the same, fp32 operands as hi/lo fp16 pieces (concept models)
                    self.call("care_gemm_ln_split", ptr(x2), x2.stride(0), ptr(Ws), ptr(w["enc_b_" + ch]), ptr(w["enc_g_" + ch]),
                         ptr(w["enc_be_" + ch]), self.eps, ptr(dst), ptr(dstb), dst.stride(-2), B * n, d, x2.shape[1], n,
                         grp_rows, off, tag="enc_gemm")
                elif fused:  # Linear + bias + LayerNorm in one kernel, raw fp32 features streamed by LDS-DMA
                    self.gemm_ln(x2, w["enc_w_" + ch], w["enc_b_" + ch], None, w["enc_g_" + ch], w["enc_be_" + ch],
                                 dst, dstb, tag="enc_gemm", Wp=w.get("enc_w_" + ch + "#packed"), **ln_kw)
                elif opt["encoder"] == "Embedder":
                    self.add_ln(lin, None, w["enc_g_" + ch], w["enc_be_" + ch], dst, dstb, **ln_kw)
                else:  # MultiTransformerEncoder
                    h, hb = self.ws("enc_h0", (B * n, d)), self.wsb("enc_h0", (B * n, d))
                    self.add_ln(lin, None, w["enc_g_" + ch], w["enc_be_" + ch], h, hb, grp=n, pos=w["enc_pos_" + ch])
                    n_enc = int(opt["num_hidden_layers_encoder"])
                    for li in range(n_enc):
                        nm = "enc{}{}".format(ch, li)
                        h1, h1b = self._mha_self_full(nm + "_sa", h, hb, n, None, False, "enc_")
                        if li == n_enc - 1:
                            self._ffn(nm + "_ffn", h1, h1b, dst, dstb, "enc_", **ln_kw)
                        else:
                            h, hb = self.ws("enc_h%d" % (li + 1), (B * n, d)), self.wsb("enc_h%d" % (li + 1), (B * n, d))
                            self._ffn(nm + "_ffn", h1, h1b, h, hb, "enc_")
                if not lean:
                    self.call("care_group_mean", ptr(dst), d, grp_rows, off, n, ptr(means), means.stride(0), mi * d, B, d)
        for st in side:
            cur.wait_stream(st)
        if lean:
            return {"encoder_hidden_states": memb}
        out: Dict[str, torch.Tensor] = {"encoder_hidden_states": mem}
        out["mean_encoder_hidden_states"] = [means[:, mi * d:(mi + 1) * d] for mi, ch in enumerate(self.modality)
                                             if ch in self.dec_mod]
        if self.has_concepts:
            if self.pred_mod == self.modality:
                pm = means
            else:
                pm = torch.cat([means[:, mi * d:(mi + 1) * d] for mi, ch in enumerate(self.modality)
                                if ch in self.pred_mod], dim=1).contiguous()
            kp = self._kpad()
            scores = self.gemm(pm, w["attr_w"], w["attr_b"], self.ws("attr_scores", (B, kp)))
            preds = new("preds", (B, kp))
            avg = new("avg", (B,))
            self.call("care_concept_finish", ptr(scores), kp, ptr(preds), kp, ptr(avg), B, self.k_attr)
            out["preds_attr"] = preds[:, : self.k_attr]
            out["avg_prob_attr"] = avg
            if self.has_container:
                labels = new("labels", (B, self.topk), torch.int64)
                if not self.has_attr_embs:  # (..L0: the labels alone, pred_attribute.py:264,276-277)
                    dst, dstb, grp_rows, off = None, None, self.topk, 0
                elif self.concat:
                    dst, dstb, grp_rows, off = mem, memb, self.Lk, self.concept_off
                else:
                    dst, dstb, grp_rows, off = new("sem_embs", (B, self.topk, d)), None, self.topk, 0
                self.call("care_concept_topk_embed", ptr(preds), kp, self.k_attr, self.topk, ptr(w["attr_word"]),
                     ptr(w["attr_pos"]), ptr(w["attr_g"]), ptr(w["attr_be"]), self.eps, ptr(labels), ptr(dst),
                     ptr(dstb), d, grp_rows, off, B, d)
                out["semantic_labels"] = labels
                out["semantic_embs"] = dst[:, off: off + self.topk] if dst is not None else None
                if self.sem:
                    out["semantic_hidden_states"] = self.gemm(preds, w["s2h_w"], w["s2h_b"], new("sem_hidden", (B, d)))
                else:
                    out["semantic_hidden_states"] = None
        # bf16 mirror of the memory: the A operand of the cross-K/V projection (internal).  Matched by
        # tensor IDENTITY (weakref), not by address: another tensor may later live at the same address.
        self._mem_mirror = (weakref.ref(mem), memb)
        return out

    # ------------------------------------------------------------------ cross K/V (once per clip)
    # the resident decodes' cross K/V from this many memory rows up go through the LDS-tiled GEMM: the A-stationary kernels
    # want many 128- / 256-row panels, and 128 clips are 42 panels of 256 on 256 CUs (*measured* 10752 x 1024 x 512:
    # 36.4 against 19.6 us; 5376 rows 20.6 / 12.6; 84 rows 6.1 / 8.1 - below the threshold nothing changes)
    # Round 5: the small-batch decodes take the LDS-tiled kernel at EVERY row count - one kernel, one K order (K
    # steps of 64 into one accumulator per output, whatever the tile shape), so a clip's K/V bits do not depend on the batch
    # it rides in (84 rows: + 2 us per pass).  -1: the A-stationary kernel instead (tuning).
    RESIDENT_CKV_TILE_ROWS = 0

    def cross_kv(self, mem: torch.Tensor, tag="ckv", resident=False, tile=False) -> List[torch.Tensor]:
        """K/V of the static memory for every decoder layer: [B, Lk, 2d] in the weight dtype.

        The reference re-projects them at every step for every beam copy
        (Attention.py:63-67 called from Layers.py:206-213); here once per clip.
        `resident`: for the one-launch decodes of small batches (their own form of the arithmetic already, resident_ok).
        `tile`: the LDS-tiled kernel at every row count (the teacher-forced pass: *measured* round 6, 344064 x 1024 x 512 in
        situ 472 -> 410 us, alone 566 -> 420; the two kernels' outputs are bit-identical, tests/test_gpu_kernels.py).
        """
        B, Lk, d = mem.shape
        mem = mem.contiguous()
        ref, memb = getattr(self, "_mem_mirror", (None, None))
        src = memb if (self.bf_act and memb is not None and ref is not None and ref() is mem) else mem
        src2 = src.view(B * Lk, d)
        out = []
        for li in range(self.n_layers):
            nm = "d{}_ca".format(li)
            kv = self.ws("{}{}".format(tag, li), (B * Lk, 2 * d), self.wt)
            out.append(self.gemm(src2, self.w[nm + "_kv_w"], self.w[nm + "_kv_b"], kv, tag="cross_kv_gemm",
                                 tile=src2.dtype == self.h16 and (tile or (resident and self.RESIDENT_CKV_TILE_ROWS >= 0))))
        return out

    LATENT_MIN_ROWS = 1

    def latent_for(self, rows: int) -> bool:
        """Absorbed cross-attention for a decode over `rows` rows?  The FORM OF THE ARITHMETIC is a
        property of the model and its compute mode (bf16, d_model = 512: absorbed; otherwise projected
        K/V), NOT of the batch a clip happens to be in: the two forms are two bf16 roundings of the same
        algebra, and switching between them by row count (round 1: from 2048 rows) made a clip's
        caption depend on the size of its batch wherever two tokens were nearly tied.  The price: the
        absorbed form has two more launches per step, which small, launch-bound batches feel
        (*measured* round 1: -15% at 32 rows, -1% at 1024; +5% at 2048, +13% at 16384).
        `engine.latent = False` (CARE_LATENT=0) selects projected K/V for every size instead;
        LATENT_MIN_ROWS > 1 restores a row threshold (tuning only).
        This is the MULTI-LAUNCH decode.  Greedy batches of <= resident_max_rows clips (256) take the resident decode
        instead (resident_ok): one launch, projected K/V - a deliberate exception to the rule above, bought with
        2 x the small-batch step rate; `resident_max_rows = 0` restores one form at every size."""
        return self.latent_ok and rows >= self.LATENT_MIN_ROWS and not getattr(self, "_small_pass", False)

    Q_TILE_MIN_ROWS = int(os.environ.get("CARE_Q_TILE_MIN_ROWS", "8192"))

    def cross_src(self, mem: torch.Tensor, rows: int):
        """What the decoder's cross-attention reads at every step: per-layer projected K/V
        (cross_kv, a list of [B*Lk, 2d] tensors), or - absorbed form - the bf16 memory itself
        ([B, Lk, d], shared by all layers; a tuple marks it)."""
        if not self.latent_for(rows):
            return self.cross_kv(mem)
        mem = mem.contiguous()
        if mem.dtype == self.h16:  # lean encode: the bf16 memory is all there is
            return (mem,) * self.n_layers
        ref, memb = getattr(self, "_mem_mirror", (None, None))
        if not (memb is not None and ref is not None and ref() is mem):
            memb = self.ws("lat_mem", tuple(mem.shape), self.h16)
            memb.copy_(mem)
        return (memb,) * self.n_layers

    def attr_kv(self, sem_embs: torch.Tensor, tag="akv") -> Optional[List[torch.Tensor]]:
        """K/V of the concept embeddings [B, topk, d] for the attr_attention block (CABase)."""
        if not self.attr_att:
            return None
        B, n, d = sem_embs.shape
        src = sem_embs.to(self.device, torch.float32).contiguous().view(B * n, d)
        if self.bf_act:  # the bf16 kernels want a bf16 operand
            srcb = self.ws(tag + "_srcb", (B * n, d), self.h16)
            srcb.copy_(src)
            src = srcb
        out = []
        for li in range(self.n_layers):
            nm = "d{}_aa".format(li)
            kv = self.ws("{}{}".format(tag, li), (B * n, 2 * d), self.wt)
            out.append(self.gemm(src, self.w[nm + "_kv_w"], self.w[nm + "_kv_b"], kv))
        return out
