"""HipEngine, beam search over many launches: the forms of the per-row top-k (logits + select, group maxima of the tiled
vocabulary product, two fused passes), the segmented search with early exit and compaction, and `translate_beam`, which
picks between all forms of the search (models/Translator.py:35-133, misc/Decoding/Beam.py).  Methods of care_amd.engine.HipEngine."""
import ctypes
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT_CODES, CARE_BF16, CARE_F32, ptr
from .constants import BOS, EOS, PAD
from .engine_util import _LaneOutputs


class BeamMixin:
    # Rows (clips x beam) from which the per-row top-k of beam search runs as two passes of the vocabulary GEMM on the
    # 256-row panels (statistics -> threshold -> sparse collect -> pick: no [rows, V] logits in memory).  Below it the
    # 16-bit modes take ONE pass of the LDS-tiled kernel that keeps group maxima (beam_groups_for; round 5), fp32 mode and
    # beam sizes above 5 the materialised logits + care_beam_select.  All forms pick the same columns in the same order
    # (tests/test_gpu_kernels.py::test_fused_beam_selection_..., test_beam_selection_from_group_maxima_...).
    # *Measured* round 5 (beam 5, us per step of the whole pass, groups / two-pass): 5120 rows 422 / 497, 10240 rows 726 /
    # 700, 20480 rows 1266 / 1162 - the group maxima are 12 KB per row and step.  Fixed for a pass by its INITIAL row count.
    BEAM_FUSED_MIN_ROWS = int(os.environ.get("CARE_BEAM_FUSED_MIN_ROWS", "8192"))

    def _beam_out_parts(self, B: int, cap: int):
        """Per-clip results of a beam search as parts of one block (engine.ws_block): nfin [B], fscore / flen [B, cap],
        fhyp [B, cap, T + 1]."""
        return [((B,), torch.int32), ((B, cap), torch.float32), ((B, cap), torch.int32), ((B, cap, self.T + 1), torch.int32)]

    def _beam_sparse_ws(self, tag: str, rows: int):
        """Workspaces of the sparse second pass (csrc/beam_sparse.hip) - (tile maxima [tiles, rows] fp32, per-tile
        row counts + work-unit prefix sums [2 tiles + 1], per-tile row lists [tiles, rows]) - or None where the 256-row statistics kernel does not apply."""
        if os.environ.get("CARE_BEAM_SPARSE", "1") == "0" or not self.lib.care_beam_sparse_applies(rows, self.V, self.d, 1):
            return None
        tiles = (self.V + 31) // 32
        return (self.ws(tag + "stmax", (tiles, rows)), self.ws(tag + "stcount", (2 * tiles + 1,), torch.int32),
                self.ws(tag + "stlist", (tiles, rows), torch.int32))

    def beam_fused_for(self, rows: int) -> bool:
        if os.environ.get("CARE_BEAM_FUSED", "1") == "0":
            return False
        return self.as_ok and rows >= self.BEAM_FUSED_MIN_ROWS

    # Beam selection below BEAM_FUSED_MIN_ROWS in the 16-bit modes (beam_size <= 8): the vocabulary product on the LDS-tiled
    # kernel keeping per (row, 64-column part) the maximum, sum exp and the maxima of its sixteen 4-column groups
    # (care_gemm_tile_beam), then one wave per row picks the bm best groups and recomputes their 4 bm logits
    # (care_beam_pick_groups) - two launches and 12 KB per row instead of the [rows, V] fp32 logits written and read back
    # (*measured* round 5, beam 5, us per step of the whole multi-launch pass, logits + care_beam_select / groups: 160 rows
    # 183 / 175, 640 rows 229 / 213, 1280 rows 274 / 245, 2560 rows 320 / 284).
    # The form is fixed for a pass by its INITIAL row count, like the fused two-pass selection's.
    BEAM_GROUPS_MIN_ROWS = int(os.environ.get("CARE_BEAM_GROUPS_MIN_ROWS", "1"))

    def beam_groups_for(self, rows: int, bm: int) -> bool:
        rows = self._form_rows or rows
        return bool(self.bf_act and not self.beam_fused_for(rows) and bm <= 8 and rows >= self.BEAM_GROUPS_MIN_ROWS and
                    128 <= self.V <= 16384 and self.d % 64 == 0)

    def _beam_groups_select(self, tag, xb, N, bm, cval, cidx):
        parts = (self.V + 63) // 64
        pmax, psum = self.ws(tag + "gpmax", (N, parts)), self.ws(tag + "gpsum", (N, parts))
        gmax = self.ws(tag + "ggmax", (N, parts, 16))
        self.call("care_gemm_tile_beam", ptr(xb), xb.stride(0), ptr(self.w["vocab"]), ptr(pmax), ptr(psum), ptr(gmax), N, self.V,
                  self.d, tag="beam_vocab_groups")
        self.call("care_beam_pick_groups", ptr(pmax), ptr(psum), ptr(gmax), parts, bm, ptr(xb), xb.stride(0), ptr(self.w["vocab"]),
                  self.V, self.d, ptr(cval), ptr(cidx), N, tag="beam_pick_groups")

    # ------------------------------------------------------------------ beam search with early exit + compaction
    def _beam_steps(self, v, t0, t1, bm, need):
        """Steps t0 .. t1 of the beam search on the n clips (n * bm rows) of state `v`; ends with the
        partition of the clip slots (care_active_slots on `done`)."""
        n, T, d = v["n"], self.T, self.d
        N, cap = n * bm, need + bm
        B = v["B"]
        self._ws_cap = [(n, B), (N, B * bm)]
        tag = v["tag"]
        cval, cidx = self.ws(tag + "cval", (N, bm)), self.ws(tag + "cidx", (N, bm), torch.int32)
        fused_sel = self.beam_fused_for(B * bm)  # one form for the whole pass, whatever the compaction leaves
        groups_sel = self.beam_groups_for(B * bm, bm)
        if groups_sel:
            pass
        elif fused_sel:
            s_parts = self.lib.care_argmax_parts_bf16_min(N, self.V, d, 1, 8)  # bf16 rows (code 1)
            s_cap = 64
            s_pmax, s_psum = self.ws(tag + "spmax", (N, s_parts)), self.ws(tag + "spsum", (N, s_parts))
            s_pidx = self.ws(tag + "spidx", (N, s_parts), torch.int32)
            s_thr, s_cnt = self.ws(tag + "sthr", (N,)), self.ws(tag + "scnt", (N,), torch.int32)
            s_cval, s_cidx = self.ws(tag + "scval", (N, s_cap)), self.ws(tag + "scidx", (N, s_cap), torch.int32)
            sparse = self._beam_sparse_ws(tag, N)
        else:
            vpad = (self.V + 63) // 64 * 64
            logits = self.ws(tag + "logits", (N, vpad))[:, : self.V]
        for t in range(t0, t1 + 1):
            a_old, a_new = v["anc"][(t - 1) & 1], v["anc"][t & 1]
            x, xb = self._decode_step(t, N, bm, v["tok"], a_old, v["sem"], v["ckv"], v["skv"], self.Lk, tag, akv=v["akv"])
            if groups_sel:
                self._beam_groups_select(tag, xb, N, bm, cval, cidx)
            elif fused_sel:
                if sparse is not None:
                    # second pass only over the (tile, row) products whose tile maximum reaches the row's threshold
                    self.call("care_gemm_argmax_bf16_tiles", ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), ptr(s_pmax),
                         ptr(s_pidx), ptr(s_psum), ptr(sparse[0]), N, self.V, d, 8, tag="beam_vocab_stats")
                    self.call("care_beam_threshold", ptr(s_pmax), s_parts, bm, ptr(s_thr), ptr(s_cnt), N)
                    self.call("care_beam_sparse_collect", ptr(xb), d, ptr(self.w["vocab"]), ptr(sparse[0]), ptr(s_thr),
                         ptr(s_cnt), ptr(s_cval), ptr(s_cidx), s_cap, ptr(sparse[1]), ptr(sparse[2]), N, self.V, d,
                         tag="beam_vocab_collect")
                else:
                    self.call("care_gemm_argmax_bf16_min", ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), ptr(s_pmax),
                         ptr(s_pidx), ptr(s_psum), N, self.V, d, 8, tag="beam_vocab_stats")
                    self.call("care_beam_threshold", ptr(s_pmax), s_parts, bm, ptr(s_thr), ptr(s_cnt), N)
                    self.call("care_gemm_collect_bf16", ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), ptr(s_thr), ptr(s_cnt),
                         ptr(s_cval), ptr(s_cidx), s_cap, N, self.V, d, tag="beam_vocab_collect")
                self.call("care_beam_pick", ptr(s_pmax), ptr(s_psum), s_parts, ptr(s_cnt), ptr(s_cval), ptr(s_cidx), s_cap,
                     bm, ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), self.V, d, ptr(cval), ptr(cidx), N)
            else:
                src = xb if xb is not None else x
                chunk = int(os.environ.get("CARE_BEAM_CHUNK", "0")) or max(128, (176 << 20) // (logits.stride(0) * 4) // 128 * 128)
                for lo in range(0, N, chunk):
                    hi = min(N, lo + chunk)
                    self.gemm(src[lo:hi], self.w["vocab"], None, logits[lo:hi], tag="step_vocab_logits")
                    self.call("care_beam_select", ptr(logits[lo:hi]), logits.stride(0), self.V, bm, ptr(cval[lo:hi]),
                         ptr(cidx[lo:hi]), hi - lo, 4 if self._small_pass else 1, tag="step_beam_select")
            self.call("care_beam_advance", ptr(cval), ptr(cidx), ptr(v["scores"]), bm, ptr(v["tok"]), ptr(a_old), ptr(a_new),
                 ptr(v["done"]), ptr(v["nfin"]), cap, ptr(v["fscore"]), ptr(v["flen"]), ptr(v["fhyp"]), t, T, need, EOS,
                 self.V, T + 1, n)
        self.call("care_active_slots", ptr(v["done"]), n, ptr(v["idx"]), ptr(v["cnt"]))

    def beam_early_exit(self, feats: List[torch.Tensor], bm: int, need: int, lean: bool = False, use_graph: bool = True):
        """encode + beam search that stops when every clip is done and drops finished clips between
        segments (models/Translator.py:77-81,194-209), like greedy_early_exit: the clip-level state
        (memory, finished lists ...) and the bm rows of every surviving clip (tokens, scores, K/V cache,
        ancestor tables - whose entries are physical row numbers and are renumbered) move to the front of
        a second buffer set.  Results per CLIP: nfin [B], fscore / flen [B, need + bm], fhyp [B, need + bm, T + 1]."""
        feats = self._prep_feats(feats)
        B, T, d = feats[0].shape[0], self.T, self.d
        cap = need + bm
        S = max(1, self.segment_steps) * (1 if B * bm >= 2048 else 2)
        out = dict(zip(("nfin", "fscore", "flen", "fhyp"), self.ws_block("be_out", self._beam_out_parts(B, cap))))
        idx, cnt = self.ws("be_idx", (B,), torch.int32), self.ws("be_cnt", (1,), torch.int32)
        fkey = (tuple(f.data_ptr() for f in feats), tuple(tuple(f.shape) for f in feats))

        def state(par, n):
            N = n * bm
            self._ws_cap = [(n, B), (N, B * bm)]
            tag = "b%d_" % par
            return dict(tag=tag, n=n, B=B, idx=idx, cnt=cnt,
                        tok=self.ws(tag + "tok", (N, T + 1), torch.int32),
                        anc=[self.ws(tag + "anc%d" % i, (N, T + 1), torch.int32) for i in range(2)],
                        scores=self.ws(tag + "scores", (N,)),
                        skv=[self.ws(tag + "skv%d" % li, (N, T, 2 * d), self.wt) for li in range(self.n_layers)],
                        done=self.ws(tag + "done", (n,), torch.int32), nfin=self.ws(tag + "nfin", (n,), torch.int32),
                        fscore=self.ws(tag + "fscore", (n, cap)), flen=self.ws(tag + "flen", (n, cap), torch.int32),
                        fhyp=self.ws(tag + "fhyp", (n, cap, T + 1), torch.int32), clip=self.ws(tag + "clip", (n,), torch.int32))

        def first_segment():
            self._ws_cap = None
            enc = self.encode(feats, lean, static=True, small=self.small_forms(B))
            mem, sem = enc["encoder_hidden_states"], enc.get("semantic_hidden_states")
            v = state(0, B)
            N = B * bm
            v["tok"].fill_(EOS); v["tok"][:, 0] = BOS
            rows = self._arange(N)
            for a in v["anc"]:
                a.copy_(rows.unsqueeze(1).expand(N, T + 1))
            for k in ("scores", "done", "nfin", "fscore", "flen", "fhyp"):
                v[k].zero_()
            torch.add(self._arange(B), 0, out=v["clip"])   # (an elementwise kernel, not a memcpy node in the captured graph: see csrc/decode_resident.h, res_zero_kernel)
            v["sem"] = sem.to(self.device, torch.float32).contiguous() if sem is not None else None
            self._ws_cap = None
            v["ckv"] = self.cross_src(mem, N)
            v["akv"] = self.attr_kv(enc.get("semantic_embs")) if self.attr_att else None
            self._beam_steps(v, 1, min(S, T), bm, need)
            return enc, v

        replayable = lambda key, fn: self._replay(key, fn, use_graph)

        def flush(v):
            """finished lists of every slot of v -> the per-clip outputs"""
            n = v["n"]
            self._call_rows("care_scatter_rows", v["nfin"].view(n, 1), out["nfin"].view(B, 1), v["clip"], n)
            for k in ("fscore", "flen"):
                self._call_rows("care_scatter_rows", v[k], out[k], v["clip"], n)
            self._call_rows("care_scatter_rows", v["fhyp"].view(n, -1), out["fhyp"].view(B, -1), v["clip"], n)

        try:
            self._form_rows = B * bm
            enc, v = replayable(("bseg0", bm, need, self.latent_ok and not self._small_pass, bool(lean), S) + fkey, first_segment)
            par, t = 0, min(S, T) + 1
            stats = dict(clips=B, steps=t - 1, row_steps=B * bm * (t - 1), compactions=0)
            self.last_decode = stats
            while True:
                active = self._host_count(cnt)
                if active == 0 or t > T:
                    break
                m = self._slot_bucket(active, B)
                if m * 4 <= v["n"] * 3 and v["n"] * bm >= 2048:
                    flush(v)
                    v = self._compact_beam(v, state(par ^ 1, m), idx, active, bm)
                    par ^= 1
                    stats["compactions"] += 1
                t1 = min(t + S - 1, T)
                vv = v
                replayable(("bseg", par, t, t1, v["n"], B, bm, need, self.latent_ok and not self._small_pass), lambda: self._beam_steps(vv, t, t1, bm, need))
                stats["steps"] = t1
                stats["row_steps"] += v["n"] * bm * (t1 - t + 1)
                t = t1 + 1
            flush(v)
        finally:
            self._ws_cap = None
        return enc, out["nfin"], out["fscore"], out["flen"], out["fhyp"]

    def _compact_beam(self, v, w, idx, active, bm):
        """The first w['n'] clips of the partition `idx` (unfinished first, finished ones as padding) and their rows
        -> buffer set `w`; ancestor entries are renumbered to the rows' new places."""
        n, m, B = v["n"], w["n"], v["B"]
        N, M = n * bm, m * bm
        self._ws_cap = [(m, B), (M, B * bm), (n, B), (N, B * bm)]
        tag = w["tag"]
        idx_r = self.ws(tag + "idx_r", (M,), torch.int32)
        self.call("care_expand_index", ptr(idx), m, bm, ptr(idx_r))
        cmap = self.ws(tag + "cmap", (n,), torch.int32)
        cmap.zero_()  # clips that are dropped map to clip 0: nothing references their rows any more
        self._call_rows("care_scatter_rows", self._arange(m).view(m, 1), cmap.view(n, 1), idx, m)
        for k in ("done", "nfin", "clip"):
            self._call_rows("care_gather_rows", v[k].view(n, 1), w[k].view(m, 1), idx, m)
        for k in ("fscore", "flen"):
            self._call_rows("care_gather_rows", v[k], w[k], idx, m)
        self._call_rows("care_gather_rows", v["fhyp"].view(n, -1), w["fhyp"].view(m, -1), idx, m)
        self._call_rows("care_gather_rows", v["tok"], w["tok"], idx_r, M)
        self._call_rows("care_gather_rows", v["scores"].view(N, 1), w["scores"].view(M, 1), idx_r, M)
        for a, b in zip(v["anc"], w["anc"]):
            self._call_rows("care_gather_rows", a, b, idx_r, M)
            self.call("care_remap_rows", ptr(b), b.numel(), ptr(cmap), bm)
        for a, b in zip(v["skv"], w["skv"]):
            self._call_rows("care_gather_rows", a, b, idx_r, M)

        def moved(name, src, per=1):
            if src is None:
                return None
            s2 = src.view(n, -1)
            dst = self.ws(tag + name, (m, s2.shape[1]), src.dtype)
            self._call_rows("care_gather_rows", s2, dst, idx, m)
            return dst.view((m * per,) + tuple(src.shape[1:])) if per > 1 else dst.view((m,) + tuple(src.shape[1:]))

        w["sem"] = moved("sem", v["sem"])
        if isinstance(v["ckv"], tuple):
            w["ckv"] = (moved("mem", v["ckv"][0]),) * len(v["ckv"])
        else:
            w["ckv"] = [moved("ckv%d" % i, kv, self.Lk) for i, kv in enumerate(v["ckv"])]
        w["akv"] = [moved("akv%d" % i, kv, self.topk) for i, kv in enumerate(v["akv"])] if v["akv"] is not None else None
        w["clip"][active:].fill_(-1)
        return w

    def translate_beam(self, feats: List[torch.Tensor], bm: int, need: int, use_graph: bool = True, lean: bool = False,
                       early_exit: Optional[bool] = None):
        """encode + beam search of one batch, replayed from a hipGraph when the input buffers repeat
        (same policy as translate_greedy).  Returns (enc_outputs, nfin, fscore, flen, fhyp)."""
        feats = self._prep_feats(feats)
        self._begin_pass()
        # beam search over a small batch: projected cross K/V (two launches less per step than the absorbed form, the
        # beams of a clip share its K/V rows in cache; *measured* 128 clips x 5: 5.97 -> 5.47 ms per pass)
        self._small_pass = self.small_forms(feats[0].shape[0])
        ee = self.early_exit if early_exit is None else early_exit
        if self.resident_beam_ok(feats[0].shape[0], bm, need):  # encode + ONE resident launch for the whole search
            def run_resident():
                self._form_rows = feats[0].shape[0] * bm
                enc = self.encode(feats, lean, static=True, small=True)
                return (enc,) + tuple(self.beam_resident(enc["encoder_hidden_states"], enc.get("semantic_hidden_states"), bm, need,
                                                         sem_embs=enc.get("semantic_embs"), early_exit=ee))
            key = ("bres", bm, need, bool(lean), bool(ee), tuple(f.data_ptr() for f in feats), tuple(tuple(f.shape) for f in feats))
            try:
                out = self._replay(key, run_resident, use_graph)
            except _lib.CareHipError as exc:
                # refused before anything was enqueued (CARE_ESHAPE: the device admits fewer resident workgroups than the
                # launch needs - a partition with few CUs, another occupancy): this engine keeps the multi-launch search
                if "CARE_ESHAPE" not in str(exc):
                    raise
                self._note_refused("beam", feats[0].shape[0] * bm)
                out = None
            if out is not None:
                nb = self.lib.care_decode_resident_beam_scratch(feats[0].shape[0], bm, self.d, self.ff, self.V)
                self.last_decode = dict(clips=feats[0].shape[0], steps=self.ws("rb_scratch", (nb,), torch.uint8)[8:12].view(torch.int32)[0],
                                        compactions=0, resident=True, row_steps=None)
                return out
        if self.chain_beam_ok(feats[0].shape[0], bm, need):  # every step a chain of ~10 kernels (csrc/decode_chain.hip)
            return self.translate_beam_chain(feats, bm, need, use_graph, lean, ee)
        if ee:
            return self.beam_early_exit(feats, bm, need, lean, use_graph)

        def run():
            self._form_rows = feats[0].shape[0] * bm
            enc = self.encode(feats, lean, small=self.small_forms(feats[0].shape[0]))
            return (enc,) + tuple(self.beam(enc["encoder_hidden_states"], enc.get("semantic_hidden_states"), bm, need,
                                            sem_embs=enc.get("semantic_embs")))

        key = ("beam", bm, need, self.latent_ok and not self._small_pass, bool(lean), tuple(f.data_ptr() for f in feats), tuple(tuple(f.shape) for f in feats))
        return self._replay(key, run, use_graph)

    def translate_beam_ensemble(self, others: list, feats_list: List[List[torch.Tensor]], bm: int, need: int, use_graph: bool = True):
        """encode + beam search of one batch by a LIST of models (model ensembling, models/Translator.py:39-52,112-133): this
        engine and `others` each encode their own feature list and decode the SHARED prefixes step by step; the step's word
        log-probabilities are the members' log_softmax averaged (care_ensemble_select) and ONE beam state machine
        (care_beam_advance, this engine's) advances on them.  Greedy decoding is bm = 1 (models/Wrapper.py:34-35).  Every member
        runs its multi-launch step with the vocabulary logits in memory - off the hot path ("out of scope beyond
        accepting the list"; built in round 6 so that a list of checkpoints decodes at all): no resident launch, no fused selection.
        Returns (enc_outputs of this engine, nfin, fscore, flen, fhyp) like translate_beam."""
        engines = [self] + list(others)
        B = feats_list[0][0].shape[0]
        for e in engines:
            if (e.T, e.V) != (self.T, self.V) or e.device != self.device:
                raise ValueError("ensemble members must share max_len, the vocabulary and the device")
        if len(engines) > 8:
            raise ValueError("at most 8 ensemble members (care_ensemble_select)")
        prepped = []
        for e, feats in zip(engines, feats_list):
            feats = e._prep_feats(feats)
            if feats[0].shape[0] != B:
                raise ValueError("ensemble members must see the same clips")
            e._begin_pass()
            prepped.append(feats)

        def run():
            members, enc0 = [], None
            for e, feats in zip(engines, prepped):
                e._form_rows = B * bm
                enc = e.encode(feats, False)
                enc0 = enc if enc0 is None else enc0
                members.append((e, enc["encoder_hidden_states"], enc.get("semantic_hidden_states"), enc.get("semantic_embs")))
            # (the engines ride along in the result: a captured graph keeps its members alive, so their ids in the key stay theirs)
            return (enc0,) + tuple(self.beam(members[0][1], members[0][2], bm, need, sem_embs=members[0][3], others=members[1:])) + (tuple(engines),)

        # replayed from a hipGraph like the single-model passes; the key carries every member's identity, feature buffers and
        # epoch (an engine that dropped workspaces or re-packed weights since the capture: engine._epoch)
        key = ("ens", bm, need, tuple((id(e), getattr(e, "_epoch", 0)) for e in engines),
               tuple(tuple((f.data_ptr(), tuple(f.shape)) for f in feats) for feats in prepped))
        return self._replay(key, run, use_graph)[:-1]

    def beam(self, mem: torch.Tensor, sem: Optional[torch.Tensor], bm: int, need: int,
             sem_embs: Optional[torch.Tensor] = None, others=()):
        """Beam search of B clips x bm beams, state on the device (csrc/beam.hip).  others: further ensemble members as
        (engine, mem, sem, sem_embs) - see translate_beam_ensemble."""
        B, Lk, d = mem.shape
        T, N = self.T, mem.shape[0] * bm
        mem = mem.to(self.device, mem.dtype if mem.dtype == self.h16 else torch.float32)  # bf16: lean encode
        sem = sem.to(self.device, torch.float32).contiguous() if sem is not None else None
        cap = need + bm
        tok = self.ws("b_tok", (N, T + 1), torch.int32)
        anc = [self.ws("b_anc%d" % i, (N, T + 1), torch.int32) for i in range(2)]
        tok.fill_(EOS); tok[:, 0] = BOS
        rows = torch.arange(N, device=self.device, dtype=torch.int32)
        for a in anc:
            a.copy_(rows.unsqueeze(1).expand(N, T + 1))
        scores = self.ws("b_scores", (N,)); scores.zero_()
        done = self.ws("b_done", (B,), torch.int32); done.zero_()
        nfin, fscore, flen, fhyp = self.ws_block("b_out", self._beam_out_parts(B, cap))
        nfin.zero_(); fscore.zero_(); flen.zero_(); fhyp.zero_()
        cval = self.ws("b_cval", (N, bm))
        cidx = self.ws("b_cidx", (N, bm), torch.int32)
        vpad = (self.V + 63) // 64 * 64  # 16-byte aligned row stride -> the GEMM's vector store path
        if others:
            # model ensembling: every member steps on the shared prefixes (tok / ancestors) with state of its own, its
            # vocabulary logits in memory; the averaged log-probabilities' top bm -> the one state machine
            per = []
            for e, m, s_, se in [(self, mem, sem, sem_embs)] + list(others):
                m = m.to(e.device, m.dtype if m.dtype == e.h16 else torch.float32)
                s_ = s_.to(e.device, torch.float32).contiguous() if s_ is not None else None
                per.append(dict(e=e, sem=s_, ckv=e.cross_src(m, N), akv=e.attr_kv(se) if e.attr_att else None, Lk=m.shape[1],
                                skv=[e.ws("b_skv%d" % li, (N, T, 2 * e.d), e.wt) for li in range(e.n_layers)],
                                logits=e.ws("b_logits", (N, vpad))))
            import ctypes
            rows_ptr = (ctypes.c_void_p * len(per))(*[p["logits"].data_ptr() for p in per])
            for t in range(1, T + 1):
                a_old, a_new = anc[(t - 1) & 1], anc[t & 1]
                for p in per:
                    e = p["e"]
                    x, xb = e._decode_step(t, N, bm, tok, a_old, p["sem"], p["ckv"], p["skv"], p["Lk"], "b_", akv=p["akv"])
                    e.gemm(xb if xb is not None else x, e.w["vocab"], None, p["logits"][:, : self.V])
                self.call("care_ensemble_select", rows_ptr, len(per), vpad, self.V, bm, ptr(cval), ptr(cidx), N)
                self.call("care_beam_advance", ptr(cval), ptr(cidx), ptr(scores), bm, ptr(tok), ptr(a_old), ptr(a_new),
                     ptr(done), ptr(nfin), cap, ptr(fscore), ptr(flen), ptr(fhyp), t, T, need, EOS, self.V, T + 1, B)
            return nfin, fscore, flen, fhyp
        fused_sel = self.beam_fused_for(B * bm)
        groups_sel = self.beam_groups_for(B * bm, bm)
        if groups_sel:
            logits = None
        elif fused_sel:
            s_parts = self.lib.care_argmax_parts_bf16_min(N, self.V, d, 1, 8)  # bf16 rows (code 1)
            s_cap = 64
            s_pmax, s_psum = self.ws("b_spmax", (N, s_parts)), self.ws("b_spsum", (N, s_parts))
            s_pidx = self.ws("b_spidx", (N, s_parts), torch.int32)
            s_thr, s_cnt = self.ws("b_sthr", (N,)), self.ws("b_scnt", (N,), torch.int32)
            s_cval, s_cidx = self.ws("b_scval", (N, s_cap)), self.ws("b_scidx", (N, s_cap), torch.int32)
            sparse = self._beam_sparse_ws("b_", N)
            logits = None
        else:
            logits = self.ws("b_logits", (N, vpad))[:, : self.V]
        ckv = self.cross_src(mem, N)
        akv = self.attr_kv(sem_embs) if self.attr_att else None
        skv = [self.ws("b_skv%d" % li, (N, T, 2 * d), self.wt) for li in range(self.n_layers)]
        for t in range(1, T + 1):
            a_old, a_new = anc[(t - 1) & 1], anc[t & 1]
            x, xb = self._decode_step(t, N, bm, tok, a_old, sem, ckv, skv, Lk, "b_", akv=akv)
            if groups_sel:
                self._beam_groups_select("b_", xb, N, bm, cval, cidx)
                self.call("care_beam_advance", ptr(cval), ptr(cidx), ptr(scores), bm, ptr(tok), ptr(a_old), ptr(a_new),
                     ptr(done), ptr(nfin), cap, ptr(fscore), ptr(flen), ptr(fhyp), t, T, need, EOS, self.V, T + 1, B)
                continue
            if fused_sel:
                # fused selection (csrc/beam.hip): statistics GEMM -> threshold -> candidate pass -> pick;
                # the [N, V] logits never exist
                if sparse is not None:
                    self.call("care_gemm_argmax_bf16_tiles", ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), ptr(s_pmax),
                         ptr(s_pidx), ptr(s_psum), ptr(sparse[0]), N, self.V, d, 8, tag="beam_vocab_stats")
                    self.call("care_beam_threshold", ptr(s_pmax), s_parts, bm, ptr(s_thr), ptr(s_cnt), N)
                    self.call("care_beam_sparse_collect", ptr(xb), d, ptr(self.w["vocab"]), ptr(sparse[0]), ptr(s_thr),
                         ptr(s_cnt), ptr(s_cval), ptr(s_cidx), s_cap, ptr(sparse[1]), ptr(sparse[2]), N, self.V, d,
                         tag="beam_vocab_collect")
                else:
                    self.call("care_gemm_argmax_bf16_min", ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), ptr(s_pmax),
                         ptr(s_pidx), ptr(s_psum), N, self.V, d, 8, tag="beam_vocab_stats")
                    self.call("care_beam_threshold", ptr(s_pmax), s_parts, bm, ptr(s_thr), ptr(s_cnt), N)
                    self.call("care_gemm_collect_bf16", ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), ptr(s_thr), ptr(s_cnt),
                         ptr(s_cval), ptr(s_cidx), s_cap, N, self.V, d, tag="beam_vocab_collect")
                self.call("care_beam_pick", ptr(s_pmax), ptr(s_psum), s_parts, ptr(s_cnt), ptr(s_cval), ptr(s_cidx), s_cap,
                     bm, ptr(xb), d, self._code(xb), ptr(self.w["vocab"]), self.V, d, ptr(cval), ptr(cidx), N)
                self.call("care_beam_advance", ptr(cval), ptr(cidx), ptr(scores), bm, ptr(tok), ptr(a_old), ptr(a_new),
                     ptr(done), ptr(nfin), cap, ptr(fscore), ptr(flen), ptr(fhyp), t, T, need, EOS, self.V, T + 1, B)
                continue
            # vocabulary logits -> per-row top-bm, in row chunks whose logits (chunk x vpad x 4 B) stay
            # inside the 256 MB Infinity Cache between the GEMM's stores and beam_select's loads
            src = xb if xb is not None else x
            # (*measured*, 20480 rows x 10560: chunks of 4096 rows = 173 MB +4% on the whole beam pass;
            # 5120 rows = 216 MB no gain, 2048 rows +1%)
            chunk = int(os.environ.get("CARE_BEAM_CHUNK", "0")) or max(128, (176 << 20) // (vpad * 4) // 128 * 128)
            for lo in range(0, N, chunk):
                hi = min(N, lo + chunk)
                self.gemm(src[lo:hi], self.w["vocab"], None, logits[lo:hi])
                self.call("care_beam_select", ptr(logits[lo:hi]), logits.stride(0), self.V, bm, ptr(cval[lo:hi]),
                     ptr(cidx[lo:hi]), hi - lo, 4 if self._small_pass else 1)
            self.call("care_beam_advance", ptr(cval), ptr(cidx), ptr(scores), bm, ptr(tok), ptr(a_old), ptr(a_new),
                 ptr(done), ptr(nfin), cap, ptr(fscore), ptr(flen), ptr(fhyp), t, T, need, EOS, self.V, T + 1, B)
        return nfin, fscore, flen, fhyp
