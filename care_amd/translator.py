"""Drop-in for the reference's `models.Translator.get_translator` (models/Translator.py:14-19).

`Translator_ARFormer.translate_batch(models, batch, *args, **kwargs)` keeps the reference's
contract (models/Translator.py:35-85): it returns `(batch_hyps, batch_scores)`
with `batch_hyps[i][j]` a python list of ints (no BOS, EOS included when emitted) and
`batch_scores[i][j]` a python float (sum of log-probs / length**beam_alpha).

What changed is where the work happens: the 29 decode steps run on the device without a
host round trip (greedy: fused vocab-GEMM/argmax + state update kernels; beam: device
beam state machine, csrc/beam.hip); the host only assembles the python lists at the end.

The hand-back (round 6): a decode's results live in ONE contiguous device block (engine.ws_block); `_launch` enqueues the pass
and ONE asynchronous copy of that block into a pinned host buffer, `_finish` waits for the copy's event and builds the python
lists from numpy views with one `.tolist()` per array and list slicing - no per-clip tensor indexing, no `.item()`.
`translate_batches(models, batches)` is the pipelined form of the same call: batch k's lists are assembled while batch
k + 1 runs on the device (what `translate.py`'s loop over the loader does serially: translate.py:34-51) - on the caller's
own thread, piece by piece: in the host waits of a segmented pass (engine.idle_hook) and after an asynchronous launch.
"""
import contextlib
import gc
from typing import Iterable, List

import numpy as np
import torch

from . import _lib
from .framework import TransformerSeq2Seq


def get_translator(opt: dict) -> object:
    name = "Translator_{}".format(opt["decoding_type"])
    if name != "Translator_ARFormer":
        raise ValueError("We can not find the class `{}` in {}".format(name, __file__))
    return Translator_ARFormer(opt)


class _Pending:
    """One launched batch: the event of its device-to-host copy, the numpy views of the pinned block it lands in, what a
    second try through the multi-launch forms needs (`retry`), and the assembly in progress (`gen`: a generator that builds
    the python lists a piece per `next`; `result` once it has finished)."""
    __slots__ = ("kind", "event", "arrays", "retry", "gen", "result", "error", "aborted")

    def __init__(self, kind, event, arrays, retry):
        self.kind, self.event, self.arrays, self.retry = kind, event, arrays, retry
        self.gen = self.result = self.error = None
        self.aborted = False


class Translator_ARFormer(object):
    # rows of a result array converted to python lists per piece of the assembly: small enough (~0.2 ms) that a piece
    # run inside a pass's host wait (engine.idle_hook) does not delay that pass's next segment
    CHUNK_ROWS = 512

    def __init__(self, opt: dict = {}):
        self.beam_size = opt.get("beam_size", 5)
        self.beam_alpha = opt.get("beam_alpha", 1.0)
        self.topk = opt.get("topk", 1)
        self.max_len = opt.get("max_len", 30)
        self.ar_token_id = opt.get("ar_token_id", None)
        if self.ar_token_id is not None:
            raise ValueError("`ar_token_id` (NACF joint training) is outside the hot path")
        if not 1 <= int(self.beam_size) <= 8:
            # (care_beam_select keeps a row's beam_size best columns in registers, care_beam_advance a clip's beam_size^2
            # candidates one per lane of a wave: csrc/beam.hip MAXBM)
            raise ValueError("beam_size {} is outside the hot path: 1 .. 8 (translate.py's default is 5)".format(self.beam_size))
        # length ** beam_alpha (Beam.py:93) for every possible length, computed by the interpreter's own `**` so that the
        # vectorised division below yields the doubles `score / t ** alpha` yields
        tab = []
        for n in range(self.max_len + 2):
            try:
                tab.append(float(n ** self.beam_alpha))
            except ZeroDivisionError:
                tab.append(0.0)
        self._len_pow = np.asarray(tab, dtype=np.float64)
        self._pinned = {}   # slot -> pinned uint8 host buffer (grown on demand, reused across calls)
        self._events = {}   # slot -> torch.cuda.Event
        self._slot = 0

    # ------------------------------------------------------------------ the reference's entry point
    @staticmethod
    def _device_of(models):
        """The device the first model lives on, as the CURRENT device of everything below (kernels are launched on the current
        device with the model's pointers: a process that drives several GPUs need not have set it); a no-op otherwise."""
        p = next(iter(models[0].parameters()), None) if len(models) else None
        return torch.cuda.device(p.device) if p is not None and p.is_cuda else contextlib.nullcontext()

    @staticmethod
    @contextlib.contextmanager
    def _turn(models):
        """The engines of `models`, held for the length of a call (threads sharing a module take turns: an engine's workspaces
        and result block are its own) - in a fixed order, so that two ensembles over the same models cannot wait for each other."""
        with contextlib.ExitStack() as stack:
            for m in sorted((m for m in models if isinstance(m, TransformerSeq2Seq)), key=id):
                stack.enter_context(getattr(m.engine(), "lock", contextlib.nullcontext()))
            yield

    def translate_batch(self, models: List[torch.nn.Module], batch: dict, *args, **kwargs):
        with self._device_of(models), self._turn(models):
            return self._finish(self._launch(models, batch, kwargs))

    def translate_batches(self, models: List[torch.nn.Module], batches: Iterable[dict], **kwargs):
        """`translate_batch` over an iterable of batches, one batch behind: yields `(batch_hyps, batch_scores)` of batch k
        after batch k + 1 has been launched, so the host-side list assembly of one batch overlaps the device pass of the
        next - inside that pass's host waits where it has any (the segmented large-batch passes: engine.idle_hook), after
        its asynchronous launch where it has none (the resident decodes).  Results and their order are those of calling
        `translate_batch` per batch.  The feature tensors of a batch must stay untouched until its results have been
        yielded (FeaturePrefetcher: depth >= 3) - only so that a resident decode that timed out can be run again; nothing
        else reads them after the launch."""
        prev = None
        for batch in batches:
            with self._device_of(models), self._turn(models):   # (per step: a generator holds neither the caller's current device nor the engines between yields)
                cur = self._launch(models, batch, kwargs, overlap=prev)
                done = self._finish(prev) if prev is not None else None
            if done is not None:
                yield done
            prev = cur
        if prev is not None:
            with self._device_of(models), self._turn(models):
                done = self._finish(prev)
            yield done

    # ------------------------------------------------------------------ launch: the device pass + one D2H copy
    _ABORTED = ("the resident decode timed out at a hand-off: its workgroups never became resident together (another "
                "long-running kernel holds CUs?); set CARE_RESIDENT_MAX_ROWS=0 / CARE_RESIDENT_BEAM_MAX_ROWS=0")

    def _engines_and_feats(self, models, batch):
        """The engines of `models` and each one's feature list: `batch['feats']` is one list of tensors for all of them, or - from
        Wrapper.ModelEnsemble - one such list per model (models/Translator.py:45-48)."""
        if len(models) < 1:
            raise ValueError("translate_batch needs at least one model")
        feats = batch["feats"]
        own = isinstance(feats[0], list)
        if own and len(feats) < len(models):
            raise ValueError("{} feature lists for {} models".format(len(feats), len(models)))
        engines, feats_list = [], []
        for i, model in enumerate(models):
            if not isinstance(model, TransformerSeq2Seq):
                raise TypeError("translate_batch needs care_amd framework modules, got {}".format(type(model)))
            engine = model.engine()
            if engine.T != self.max_len - 1:
                raise ValueError("translator max_len {} != model max_len {}".format(self.max_len, engine.T + 1))
            engines.append(engine)
            feats_list.append(list(feats[i] if own else feats))
            if len(feats_list[-1]) == 0 or feats_list[-1][0].shape[0] == 0:
                # (the reference fails on such a batch too - torch.stack of no prefixes, Translator.py:106 - a few calls further in)
                raise ValueError("translate_batch got a batch without clips")
        return engines, feats_list

    def _engine_and_feats(self, models, batch):
        engines, feats_list = self._engines_and_feats(models[:1], batch)
        return engines[0], feats_list[0]

    def _launch(self, models, batch, kwargs, overlap: "_Pending" = None) -> _Pending:
        with torch.no_grad():
            if len(models) > 1:   # model ensembling (Translator.py:112-133): the members step side by side
                return self._launch_ensemble(*self._engines_and_feats(models, batch), use_graph=kwargs.get("use_graph", True))
            engine, feats = self._engine_and_feats(models, batch)
            use_graph = kwargs.get("use_graph", True)
            hook = self._hook_for(overlap) if overlap is not None else None
            if hook is not None:
                engine.idle_hook = hook   # the pass's host waits assemble the previous batch (engine._host_count)
            try:
                if self.beam_size == 1:
                    return self._launch_greedy(engine, feats, use_graph)
                return self._launch_beam(engine, feats, use_graph)
            finally:
                if hook is not None:
                    engine.idle_hook = None

    def _launch_ensemble(self, engines, feats_list, use_graph=True) -> _Pending:
        """Several models: their log-probabilities averaged step by step, one beam state machine (greedy = beam_size 1, as in
        the reference: models/Wrapper.py:34-35); the results come back in the beam search's block."""
        need = max(self.beam_size, self.topk)
        _, nfin, fscore, flen, fhyp = engines[0].translate_beam_ensemble(engines[1:], feats_list, self.beam_size, need, use_graph=use_graph)
        event, arrays = self._fetch([nfin, fscore, flen, fhyp])
        return _Pending("beam", event, arrays, None)

    def _launch_greedy(self, engine, feats, use_graph, retry=True) -> _Pending:
        _, fed, length, score = engine.translate_greedy(list(feats), use_graph=use_graph, lean=True)
        event, arrays = self._fetch([length, score, fed])
        return _Pending("greedy", event, arrays, (engine, feats) if retry else None)

    def _launch_beam(self, engine, feats, use_graph=True, retry=True) -> _Pending:
        # Beam.specific_nums_of_sents = max(size, topk) (Beam.py:10): with topk > beam_size a clip keeps
        # decoding until topk hypotheses have ended (or max_len)
        need = max(self.beam_size, self.topk)
        _, nfin, fscore, flen, fhyp = engine.translate_beam(list(feats), self.beam_size, need, use_graph=use_graph, lean=True)
        event, arrays = self._fetch([nfin, fscore, flen, fhyp])
        return _Pending("beam", event, arrays, (engine, feats) if retry else None)

    def _fetch(self, tensors):
        """Enqueue the copy of the result tensors to pinned host memory (ONE copy when they are parts of one block -
        engine.ws_block - else one per tensor into consecutive ranges of the same buffer) and record an event behind it.
        Returns (event, numpy views of the host buffer); nothing is waited for here."""
        if not tensors[0].is_cuda:   # (host tensors: the unit tests' stand-in engine)
            return None, [t.numpy() for t in tensors]
        tensors = [t if t.is_contiguous() else t.contiguous() for t in tensors]
        sizes = [t.numel() * t.element_size() for t in tensors]
        ptrs = [t.data_ptr() for t in tensors]
        lo = min(ptrs)
        hi = max(p + n for p, n in zip(ptrs, sizes))
        st = tensors[0].untyped_storage()
        one = (hi - lo <= sum(sizes) + 256 * len(tensors) and st.data_ptr() <= lo and hi <= st.data_ptr() + st.nbytes() and
               all(t.untyped_storage().data_ptr() == st.data_ptr() for t in tensors))
        total = (hi - lo) if one else sum((n + 15) // 16 * 16 for n in sizes)
        slot = self._slot
        self._slot = (slot + 1) % 3   # three buffers in rotation: one being assembled, one in flight, one being launched
        host = self._pinned.get(slot)
        if host is None or host.numel() < total:
            host = self._pinned[slot] = torch.empty(max(total, 4096) * 5 // 4, dtype=torch.uint8).pin_memory()
        if one:
            src = torch.empty(0, dtype=torch.uint8, device=tensors[0].device).set_(st, lo - st.data_ptr(), (hi - lo,))
            host[: hi - lo].copy_(src, non_blocking=True)
            offs = [p - lo for p in ptrs]
        else:
            offs, off = [], 0
            for t, n in zip(tensors, sizes):
                host[off: off + n].copy_(t.view(-1).view(torch.uint8), non_blocking=True)
                offs.append(off)
                off += (n + 15) // 16 * 16
        event = self._events.get(slot)
        if event is None:
            event = self._events[slot] = torch.cuda.Event()
        event.record()
        base = host.numpy()
        kinds = {torch.int32: np.int32, torch.float32: np.float32}
        return event, [base[off: off + n].view(kinds[t.dtype]).reshape(tuple(t.shape)) for t, n, off in zip(tensors, sizes, offs)]

    # ------------------------------------------------------------------ finish: wait for the copy, build the lists
    def _greedy(self, engine, feats, use_graph=True):
        return self._finish(self._launch_greedy(engine, feats, use_graph))

    def _beam(self, engine, feats, use_graph=True):
        return self._finish(self._launch_beam(engine, feats, use_graph))

    @staticmethod
    def _gave_up(p: _Pending) -> bool:
        # length (greedy) / nfin (beam): negative = the resident launch gave up (include/care_hip.h).  ZERO is no result either - a
        # finished clip has at least one token / one hypothesis: a clip that comes back without is treated like a launch that gave
        # up (decoded again by the multi-launch pass; an error if that pass reports the same), never handed on as an empty caption
        # (seen once, round 6: a greedy job of test_resident_launches_under_contention_never_hang[fenced] came back with lengths 0
        # in one full-suite run of about ten; not reproduced in isolation)
        first = p.arrays[0]
        return bool(first.size) and int(first.min()) <= 0

    def _hook_for(self, p: _Pending):
        """engine.idle_hook that advances batch p's assembly by one piece per call (False: nothing left to do here).
        It never launches anything - it runs in the middle of ANOTHER batch's pass: a batch whose resident launch gave up
        is left to `_finish`."""
        if p.result is not None or p.error is not None or p.aborted:
            return None

        def hook():
            if p.result is not None or p.error is not None or p.aborted:
                return False
            try:
                if p.gen is None:
                    if p.event is not None and not p.event.query():
                        return True   # (its copy is still in flight: keep polling)
                    if self._gave_up(p):
                        p.aborted = True
                        return False
                    p.gen = self._assemble(p)
                next(p.gen)
            except StopIteration as stop:
                p.result = stop.value
                return False
            except BaseException as exc:  # noqa: BLE001 - re-raised by _finish
                p.error = exc
                return False
            return True
        return hook

    def _finish(self, p: _Pending):
        if p.error is not None:
            raise p.error
        if p.result is not None:
            return p.result
        if p.gen is None:
            if p.event is not None:
                p.event.synchronize()
            if self._gave_up(p):
                # care_decode_resident / care_decode_resident_beam gave up waiting for its workgroups: once more through
                # the multi-launch decode, which needs no co-residency; raise only if that fails too
                if p.retry is None:
                    raise _lib.CareHipError(self._ABORTED)
                engine, feats = p.retry
                knob = "resident_max_rows" if p.kind == "greedy" else "resident_beam_max_rows"
                keep = getattr(engine, knob)
                setattr(engine, knob, 0)
                try:
                    with torch.no_grad():
                        q = (self._launch_greedy if p.kind == "greedy" else self._launch_beam)(engine, feats, False, retry=False)
                    q.event.synchronize()
                finally:
                    setattr(engine, knob, keep)
                if self._gave_up(q):
                    raise _lib.CareHipError(self._ABORTED)
                p.arrays, p.aborted = q.arrays, False
            p.gen = self._assemble(p)
        try:
            while True:
                next(p.gen)
        except StopIteration as stop:
            p.result = stop.value
        return p.result

    def _assemble(self, p: _Pending):
        return self._assemble_greedy_gen(*p.arrays) if p.kind == "greedy" else self._assemble_beam_gen(*p.arrays)

    @staticmethod
    def _run(gen):
        try:
            while True:
                next(gen)
        except StopIteration as stop:
            return stop.value

    def _assemble_greedy(self, length, score, fed):
        return self._run(self._assemble_greedy_gen(length, score, fed))

    def _assemble_beam(self, nfin, fscore, flen, fhyp):
        return self._run(self._assemble_beam_gen(nfin, fscore, flen, fhyp))

    def _norm(self, score, length):
        """score / length ** alpha as doubles (Beam.py:91-95), for whole arrays."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return score.astype(np.float64) / self._len_pow[np.clip(length, 0, len(self._len_pow) - 1)]

    def _token_lists(self, out, tokens, lengths, wrap=False):
        """Generator: [n, w] int32 + [n] lengths -> n python lists appended to `out`, row i cut to lengths[i] (wrap: each in
        a list of its own); ONE conversion per CHUNK_ROWS rows, then a yield; no copy of a row that is already as long as
        its chunk's longest."""
        n = tokens.shape[0]
        for lo in range(0, n, self.CHUNK_ROWS):
            ln = lengths[lo: lo + self.CHUNK_ROWS]
            width = int(ln.max()) if ln.size else 0
            rows = tokens[lo: lo + self.CHUNK_ROWS, :width].tolist()
            if int(ln.min()) == width:
                out.extend([[r] for r in rows] if wrap else rows)
            elif wrap:
                out.extend([[r[:k]] for r, k in zip(rows, ln.tolist())])
            else:
                out.extend([r[:k] for r, k in zip(rows, ln.tolist())])
            yield

    def _assemble_greedy_gen(self, length, score, fed):
        n_best = min(self.topk, 1)   # (Translator.py:211-220 with one finished hypothesis per clip)
        B = int(length.shape[0])
        if n_best < 1:
            return [[] for _ in range(B)], [[] for _ in range(B)]
        # (a million small objects with no cycles among them: the cyclic collector's generation scans are a third of the
        # assembly's time at 32768 clips, so it sits out - and comes back as it was found - while the lists are built)
        was = gc.isenabled()
        gc.disable()
        try:
            hyps = []
            yield from self._token_lists(hyps, fed[:, 1:], length, wrap=True)
            scores = [[s] for s in self._norm(score, length).tolist()]
        finally:
            if was:
                gc.enable()
        # (an explicit gc.collect(1) here - the collector's look at the 3 B new lists as a piece of its own - was tried: one
        # piece of 20 - 30 ms stalls the pass whose host wait it runs in: 0.76 -> 0.50 x the engine pass at 32768 clips)
        return hyps, scores

    def _assemble_beam_gen(self, nfin, fscore, flen, fhyp):
        B, cap = int(fscore.shape[0]), int(fscore.shape[1])
        hyps, scores = [[] for _ in range(B)], [[] for _ in range(B)]
        if B == 0:
            return hyps, scores
        nfin = np.minimum(nfin.astype(np.int64), cap)
        norm = self._norm(fscore, flen)
        # best first, finished order among equals: a STABLE sort on the negated score, like Beam.sort_finished (Beam.py:91-101)
        key = np.where(np.arange(cap)[None, :] < nfin[:, None], -norm, np.inf)
        order = np.argsort(key, axis=1, kind="stable")
        # Translator.py:211-220 re-assigns n_best inside the loop over clips: it shrinks for every later clip once a
        # clip has fewer finished hypotheses.  Reproduced as is: the running minimum.
        n_best = np.minimum.accumulate(np.minimum(nfin, self.topk))
        yield
        was = gc.isenabled()
        gc.disable()
        try:
            for r in range(int(n_best.max())):
                clips = np.nonzero(n_best > r)[0]
                sel = order[clips, r]
                toks = []
                yield from self._token_lists(toks, fhyp[clips, sel], flen[clips, sel])
                for i, h, s in zip(clips.tolist(), toks, norm[clips, sel].tolist()):
                    hyps[i].append(h)
                    scores[i].append(s)
        finally:
            if was:
                gc.enable()
        return hyps, scores
