"""HipEngine, the small-batch forms: which batches decode as ONE resident launch (csrc/decode_resident*.hip) or as chained
kernels per step (csrc/decode_chain.hip), and their drivers (models/Translator.py:77-143 on the device).  Methods of
care_amd.engine.HipEngine."""
import ctypes
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT_CODES, CARE_BF16, CARE_F32, ptr
from .constants import BOS, EOS, PAD
from .engine_util import _LaneOutputs, device_props


class ResidentMixin:
    # ------------------------------------------------------------------ resident decode of small batches
    RESIDENT_MAX_V = 64 * 64 * 4  # csrc/decode_resident.hip: 64 lanes x RES_NP column-group partials of 64 columns

    def resident_ok(self, rows: int) -> bool:
        """Greedy decode of `rows` clips as one resident launch (csrc/decode_resident.hip)?  bf16 mode, d_model = 512;
        a form of its own next to the multi-launch one: projected cross K/V, the same rounding points, sums in another
        order - so which of two nearly tied tokens wins can differ between a batch of <= resident_max_rows clips and a
        larger one (the audit of tests/test_gpu_properties.py counts such rows)."""
        if not (0 < rows <= self.resident_max_rows and self._resident_model_ok()):
            return False
        if rows >= self._refused_from("greedy"):
            return False
        if self.d != 512 and rows > self.RESIDENT_WIDE_MAX_ROWS:
            return False
        return self._resident_fits(rows)

    RESIDENT_WIDE_MAX_ROWS = 128  # d_model 768 / 1024: the K-split forms only (csrc/decode_resident.hip, template D)

    def _resident_model_ok(self, beam: bool = False) -> bool:
        """Every model-side limit care_decode_resident / care_decode_resident_beam enforce (CARE_ESHAPE otherwise):
        bf16 mode; d_model 512 (ff 512 / 1024 / 2048), or - greedy only - d_model 768 / 1024 with ff = 4 d_model."""
        if self.pre_ln:  # (the resident phases normalise AFTER the residual sum: post-LN decoders only)
            return False
        if not (self.bf and self.wt == self.h16 and self.T <= 128 and self.n_layers <= 4 and
                (not self.attr_att or self.topk <= 128) and self.V <= self.RESIDENT_MAX_V and self.Lk <= 128):
            return False
        if self.d == 512:
            return bool(self.as_ok and self.ff in (512, 1024, 2048))
        return bool(self.d in (768, 1024) and self.ff == 4 * self.d and self.bf_act)  # (greedy and - round 5 - beam search)

    def _refused_from(self, kind: str) -> int:
        """Smallest row count at which a resident launch of `kind` ("greedy" / "beam") was refused on this device
        (CARE_ESHAPE: fewer co-resident workgroups than the launch needs - a partition with few CUs); launches below it are
        still tried, the chained step (no residency condition) never looks here."""
        return getattr(self, "_resident_refused", {}).get(kind, 1 << 30)

    def _note_refused(self, kind: str, rows: int) -> None:
        d = self.__dict__.setdefault("_resident_refused", {})
        d[kind] = min(d.get(kind, 1 << 30), rows)

    def _resident_fits(self, rows: int, per_tile: int = 1) -> bool:
        """one workgroup per CU at most, and at least one per group of `per_tile` 16-row tiles (a partitioned GPU has fewer CUs)"""
        if self.device is not None and torch.cuda.is_available():
            if getattr(self, "_cus", None) is None:
                self._cus = device_props(self.device).multi_processor_count
            return ((rows + 15) // 16 + per_tile - 1) // per_tile <= self._cus // 8 * 8
        return True

    RESIDENT_BEAM_MAX = 8  # csrc/decode_resident.h RES_BMK: 5 in the launch's first instance, 8 in its second (decode_resident_beam_wide.hip)
    # beam search of the d_model 768 / 1024 models as one resident launch up to this many rows (160 at d_model 1024).  *Measured*
    # (round 5, tools/beam_sweep.py, us per step of the whole pass, resident / multi-launch): d_model 1024 - 5 rows 103 / 207, 40
    # rows 131 / 214, 125 rows 171 / 225, 160 rows 196 / 225, 200 rows 232 / 228; d_model 768 - 5 rows 88 / 186, 160 rows 152 / 202,
    # 255 rows 183 / 207
    RESIDENT_WIDE_BEAM_MAX_ROWS = int(os.environ.get("CARE_RESIDENT_WIDE_BEAM_MAX_ROWS", "256"))

    def resident_beam_ok(self, clips: int, bm: int, need: int) -> bool:
        """Beam search over `clips` clips as one resident launch (csrc/decode_resident_beam.hip)?  The limits of
        care_decode_resident_beam: the greedy launch's, beam_size <= 8, a hypothesis' positions one per lane (T <= 63)."""
        rows = clips * bm
        if not (0 < rows <= self.resident_beam_max_rows and 1 < bm <= self.RESIDENT_BEAM_MAX and need >= 1 and
                self._resident_model_ok(beam=True) and self.T <= 63 and self.V >= 16 * self.RESIDENT_BEAM_MAX):
            return False
        if rows >= self._refused_from("beam"):
            return False
        if self.d != 512 and rows > (self.RESIDENT_WIDE_BEAM_MAX_ROWS if self.d <= 768 else min(160, self.RESIDENT_WIDE_BEAM_MAX_ROWS)):
            return False  # (d_model 768 / 1024: the K-split forms; see RESIDENT_WIDE_BEAM_MAX_ROWS)
        # (care_decode_resident_beam packs two row tiles per workgroup only in its forms for MORE than 512 rows; up to 512
        # rows it needs a workgroup per 16-row tile - a partitioned device with fewer CUs than tiles must not be promised
        # the resident form: ADVICE r4)
        return self._resident_fits(rows, 2 if rows > 512 else 1)

    def chain_beam_ok(self, clips: int, bm: int, need: int) -> bool:
        """Beam search over `clips` clips with every step a chain of kernels (csrc/decode_chain.hip)?  The model-side
        limits of the resident beam launch (its phases are the chain's kernels); no residency condition, so the row
        count is bounded only by where the large-batch forms take over (`chain_beam_max_rows`)."""
        rows = clips * bm
        return bool(0 < rows <= self.chain_beam_max_rows and 1 < bm <= 5 and need >= 1 and self.d == 512 and   # (RES_BMK of decode_chain.hip)
                    self._resident_model_ok(beam=True) and self.T <= 63 and self.V >= 16 * self.RESIDENT_BEAM_MAX)

    def small_forms(self, clips: int) -> bool:
        """Batches of <= resident_max_rows clips (bf16, d_model = 512) take the small-batch forms of the pass: the
        embedder as GEMM + LayerNorm launches side by side per modality (encode(small=True)) and, for greedy decoding,
        the resident decode.  `resident_max_rows = 0`: one set of forms at every batch size."""
        return 0 < clips <= self.resident_max_rows and self.as_ok and self.d == 512

    def _resident_layers(self, tag: str, rows: int, rows_per_clip: int, ckv, akv, Lk: int):
        """care_resident_layer[] of this model for a resident launch over `rows` rows (self-attention caches in the
        workspaces `tag`skv*; static K/V per clip, shared by its `rows_per_clip` rows)."""
        w, d, T = self.w, self.d, self.T
        layers = self._res_layers = (_lib.ResidentLayer * self.n_layers)()  # kept: bench.py re-issues the recorded call
        for li in range(self.n_layers):
            L, sa, ffn = layers[li], "d{}_sa".format(li), "d{}_ffn".format(li)
            L.qkv_w, L.qkv_b, L.o_w, L.o_b = ptr(w[sa + "_qkv_w"]), ptr(w[sa + "_qkv_b"]), ptr(w[sa + "_o_w"]), ptr(w[sa + "_o_b"])
            L.ln_g, L.ln_b = ptr(w[sa + "_g"]), ptr(w[sa + "_be"])
            L.self_kv = ptr(self.ws(tag + "skv%d" % li, (rows, T, 2 * d), self.h16))
            blocks = [("d{}_ca".format(li), ckv[li], Lk, w["d{}_hb".format(li)])]
            if self.attr_att:
                blocks.append(("d{}_aa".format(li), akv[li], self.topk, None))
            L.n_att = len(blocks)
            for a, (nm, kv, nkeys, hb) in enumerate(blocks):
                A = L.att[a]
                A.q_w, A.q_b, A.o_w, A.o_b = ptr(w[nm + "_q_w"]), ptr(w[nm + "_q_b"]), ptr(w[nm + "_o_w"]), ptr(w[nm + "_o_b"])
                A.ln_g, A.ln_b = ptr(w[nm + "_g"]), ptr(w[nm + "_be"])
                A.kv, A.kv_batch_stride, A.nkeys, A.rows_per_kv = ptr(kv), nkeys * 2 * d, nkeys, rows_per_clip
                A.bias, A.bias_ld = ptr(hb), (hb.stride(0) if hb is not None else 0)
            L.w1, L.b1, L.w2, L.b2 = ptr(w[ffn + "_w1"]), ptr(w[ffn + "_b1"]), ptr(w[ffn + "_w2"]), ptr(w[ffn + "_b2"])
            L.ffn_g, L.ffn_b = ptr(w[ffn + "_g"]), ptr(w[ffn + "_be"])
        return layers

    def beam_resident(self, mem: torch.Tensor, sem: Optional[torch.Tensor], bm: int, need: int,
                      sem_embs: Optional[torch.Tensor] = None, early_exit: bool = True):
        """Beam search of B clips x bm beams in ONE launch (care_decode_resident_beam): the step loop of
        Translator.translate_batch (models/Translator.py:77-143) with Beam.advance (misc/Decoding/Beam.py:45-85) on the
        device, stopping once every clip is done (Translator.py:77-81).  Returns the per-clip results of engine.beam:
        nfin [B], fscore / flen [B, need + bm], fhyp [B, need + bm, T + 1]; no host synchronisation here."""
        B, Lk, d = mem.shape
        T, w, N, cap = self.T, self.w, mem.shape[0] * bm, need + bm
        sem = sem.to(self.device, torch.float32).contiguous() if sem is not None else None
        ckv = self.cross_kv(mem, tag="rb_ckv", resident=True)
        akv = self.attr_kv(sem_embs, tag="rb_akv") if self.attr_att else None
        tok = self.ws("rb_tok", (N, T + 1), torch.int32)
        anc = [self.ws("rb_anc%d" % i, (N, T + 1), torch.int32) for i in range(2)]
        scores, done = self.ws("rb_scores", (N,)), self.ws("rb_done", (B,), torch.int32)
        nfin, fscore, flen, fhyp = self.ws_block("rb_out", self._beam_out_parts(B, cap))
        layers = self._resident_layers("rb_", N, bm, ckv, akv, Lk)
        nbytes = self.lib.care_decode_resident_beam_scratch(B, bm, d, self.ff, self.V)
        scratch = self.ws("rb_scratch", (nbytes,), torch.uint8)
        self.call("care_decode_resident_beam", ctypes.addressof(layers), self.n_layers, ptr(w["word"]), ptr(w["pos"]), ptr(sem),
             ptr(w["emb_g"]), ptr(w["emb_be"]), self.eps, ptr(w["vocab"]), self.V, d, self.H, self.ff, self.act, B, bm, need, T, T,
             BOS, EOS, PAD, ptr(tok), T + 1, ptr(anc[0]), ptr(anc[1]), ptr(scores), ptr(done), ptr(nfin), ptr(fscore), ptr(flen),
             ptr(fhyp), cap, ptr(scratch), nbytes, int(bool(early_exit)), int(os.environ.get("CARE_RESIDENT_BLOCKS", "0")),
             tag="decode_resident_beam")
        self.last_decode = dict(clips=B, steps=scratch[8:12].view(torch.int32)[0], compactions=0, resident=True,
                                row_steps=None)
        return nfin, fscore, flen, fhyp

    def _chain_state(self, B: int, bm: int, need: int):
        T, N, cap = self.T, B * bm, need + bm
        return dict(tok=self.ws("cb_tok", (N, T + 1), torch.int32),
                    anc=[self.ws("cb_anc%d" % i, (N, T + 1), torch.int32) for i in range(2)],
                    scores=self.ws("cb_scores", (N,)), done=self.ws("cb_done", (B,), torch.int32),
                    **dict(zip(("nfin", "fscore", "flen", "fhyp"), self.ws_block("cb_out", self._beam_out_parts(B, cap)))),
                    idx=self.ws("cb_idx", (B,), torch.int32), cnt=self.ws("cb_cnt", (1,), torch.int32))

    def beam_chain_steps(self, mem: torch.Tensor, sem: Optional[torch.Tensor], bm: int, need: int, t0: int, t1: int,
                         sem_embs: Optional[torch.Tensor] = None, count_live: bool = True):
        """Steps t0 .. t1 of the beam search of B clips x bm beams as chains of kernels (care_decode_chain_beam: 10
        launches per step for a one-layer decoder; models/Translator.py:77-143, misc/Decoding/Beam.py:45-85), the beam
        state of csrc/beam.hip in the `cb_` workspaces; t0 == 1 also projects the clips' static K/V and initialises the
        state.  Ends with the partition of the clips by `done` (care_active_slots: cb_cnt = clips still live).  No host
        synchronisation here."""
        B, Lk, d = mem.shape
        T, w, N, cap = self.T, self.w, mem.shape[0] * bm, need + bm
        sem = sem.to(self.device, torch.float32).contiguous() if sem is not None else None
        kvs = self.__dict__.setdefault("_chain_kv", {})
        if t0 == 1:  # (static workspaces: the handles of a (clips, beam) stay valid for the later segments' graphs)
            kvs[(B, bm)] = (self.cross_kv(mem, tag="cb_ckv", resident=True),
                            self.attr_kv(sem_embs, tag="cb_akv") if self.attr_att else None)
        ckv, akv = kvs[(B, bm)]
        v = self._chain_state(B, bm, need)
        layers = self._resident_layers("cb_", N, bm, ckv, akv, Lk)
        nbytes = self.lib.care_decode_chain_beam_scratch(B, bm, d, self.ff, self.V)
        scratch = self.ws("cb_scratch", (nbytes,), torch.uint8)
        self.call("care_decode_chain_beam", ctypes.addressof(layers), self.n_layers, ptr(w["word"]), ptr(w["pos"]), ptr(sem),
                  ptr(w["emb_g"]), ptr(w["emb_be"]), self.eps, ptr(w["vocab"]), self.V, d, self.H, self.ff, self.act, B, bm, need, T,
                  t0, t1, BOS, EOS, PAD, ptr(v["tok"]), T + 1, ptr(v["anc"][0]), ptr(v["anc"][1]), ptr(v["scores"]), ptr(v["done"]),
                  ptr(v["nfin"]), ptr(v["fscore"]), ptr(v["flen"]), ptr(v["fhyp"]), cap, ptr(scratch), nbytes,
                  int(os.environ.get("CARE_CHAIN_FORM", "-1")), tag="decode_chain_beam")
        if count_live:
            self.call("care_active_slots", ptr(v["done"]), B, ptr(v["idx"]), ptr(v["cnt"]))
        return v

    def translate_beam_chain(self, feats: List[torch.Tensor], bm: int, need: int, use_graph: bool = True, lean: bool = False,
                             early_exit: bool = True):
        """encode + beam search with chained steps.  The pass runs in segments of `chain_segment_steps` steps, each a
        hipGraph of its own (the first with the encoder and the static K/V projection); between segments the host reads
        ONE counter - the clips still live - and stops when none is (`if not active_inst_idx_list: break`,
        models/Translator.py:77-81).  early_exit=False: all T steps in one graph.  No compaction: the chain serves the
        row counts below those at which moving the survivors pays (engine.beam_early_exit)."""
        B, T = feats[0].shape[0], self.T
        S = max(1, self.chain_segment_steps) if early_exit else T
        fkey = (tuple(f.data_ptr() for f in feats), tuple(tuple(f.shape) for f in feats))
        box = {}

        def first():
            self._form_rows = B * bm
            enc = self.encode(feats, lean, static=True, small=self.small_forms(B))
            box["enc"] = enc
            v = self.beam_chain_steps(enc["encoder_hidden_states"], enc.get("semantic_hidden_states"), bm, need, 1, min(S, T),
                                      sem_embs=enc.get("semantic_embs"), count_live=early_exit)
            return enc, v

        enc, v = self._replay(("bchain", 0, S, bm, need, bool(lean), bool(early_exit)) + fkey, first, use_graph)
        t = min(S, T) + 1
        stats = dict(clips=B, steps=t - 1, row_steps=B * bm * (t - 1), compactions=0, chain=True)
        self.last_decode = stats
        while t <= T:
            if early_exit and self._host_count(v["cnt"]) == 0:
                break
            t1 = min(t + S - 1, T)
            tt = t
            self._replay(("bchain", tt, t1, bm, need, B, bool(lean)) + fkey,
                         lambda: self.beam_chain_steps(enc["encoder_hidden_states"], enc.get("semantic_hidden_states"), bm, need,
                                                       tt, t1, sem_embs=enc.get("semantic_embs")), use_graph)
            stats["steps"] = t1
            stats["row_steps"] += B * bm * (t1 - tt + 1)
            t = t1 + 1
        return enc, v["nfin"], v["fscore"], v["flen"], v["fhyp"]

    def greedy_resident(self, mem: torch.Tensor, sem: Optional[torch.Tensor], sem_embs: Optional[torch.Tensor] = None,
                        steps: Optional[int] = None, early_exit: bool = True):
        """Greedy decoding of B clips in ONE launch: the step loop of Translator.translate_batch with beam_size 1
        (models/Translator.py:77-143) runs on the device, phases of a step separated by grid barriers, and stops once
        every clip has ended (Translator.py:77-81).  Returns device tensors fed int32 [B, T + 1], length int32 [B],
        score fp32 [B]; `self.last_decode["steps"]` is a 0-dim DEVICE tensor (no host synchronisation here)."""
        B, Lk, d = mem.shape
        T, w = self.T, self.w
        steps = T if steps is None else steps
        sem = sem.to(self.device, torch.float32).contiguous() if sem is not None else None
        ckv = self.cross_kv(mem, tag="r_ckv", resident=True)
        akv = self.attr_kv(sem_embs, tag="r_akv") if self.attr_att else None
        length, score, fed = self.ws_block("r_out", [((B,), torch.int32), ((B,), torch.float32), ((B, T + 1), torch.int32)])
        fin = self.ws("r_fin", (B,), torch.int32)
        layers = self._resident_layers("r_", B, 1, ckv, akv, Lk)
        nbytes = self.lib.care_decode_resident_scratch(B, d, self.ff, self.V)
        scratch = self.ws("r_scratch", (nbytes,), torch.uint8)
        self.call("care_decode_resident", ctypes.addressof(layers), self.n_layers, ptr(w["word"]), ptr(w["pos"]), ptr(sem), 1,
             ptr(w["emb_g"]), ptr(w["emb_be"]), self.eps, ptr(w["vocab"]), self.V, d, self.H, self.ff, self.act, B, T, steps,
             BOS, EOS, PAD, ptr(fed), T + 1, ptr(score), ptr(length), ptr(fin), ptr(scratch), nbytes,
             int(bool(early_exit)), int(os.environ.get("CARE_RESIDENT_BLOCKS", "0")), tag="decode_resident")
        self.last_decode = dict(clips=B, steps=scratch[8:12].view(torch.int32)[0], compactions=0, resident=True)
        return fed, length, score
