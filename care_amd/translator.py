"""Drop-in for the reference's `models.Translator.get_translator` (models/Translator.py:14-19).

`Translator_ARFormer.translate_batch(models, batch, *args, **kwargs)` keeps the reference's
contract (models/Translator.py:35-85): it returns `(batch_hyps, batch_scores)`
with `batch_hyps[i][j]` a python list of ints (no BOS, EOS included when emitted) and
`batch_scores[i][j]` a python float (sum of log-probs / length**beam_alpha).

What changed is where the work happens: the 29 decode steps run on the device without a
host round trip (greedy: fused vocab-GEMM/argmax + state update kernels; beam: device
beam state machine, csrc/beam.hip); the host only assembles the python lists at the end.
"""
from typing import List

import torch

from . import _lib
from .framework import TransformerSeq2Seq


def get_translator(opt: dict) -> object:
    name = "Translator_{}".format(opt["decoding_type"])
    if name != "Translator_ARFormer":
        raise ValueError("We can not find the class `{}` in {}".format(name, __file__))
    return Translator_ARFormer(opt)


class Translator_ARFormer(object):
    def __init__(self, opt: dict = {}):
        self.beam_size = opt.get("beam_size", 5)
        self.beam_alpha = opt.get("beam_alpha", 1.0)
        self.topk = opt.get("topk", 1)
        self.max_len = opt.get("max_len", 30)
        self.ar_token_id = opt.get("ar_token_id", None)
        if self.ar_token_id is not None:
            raise ValueError("`ar_token_id` (NACF joint training) is outside the hot path")

    def translate_batch(self, models: List[torch.nn.Module], batch: dict, *args, **kwargs):
        if len(models) != 1:
            raise NotImplementedError("model ensembling (Translator.py:131) is outside the hot path")
        model = models[0]
        if not isinstance(model, TransformerSeq2Seq):
            raise TypeError("translate_batch needs a care_amd framework module, got {}".format(type(model)))
        feats = batch["feats"]
        if isinstance(feats[0], list):
            feats = feats[0]
        with torch.no_grad():
            engine = model.engine()
            if engine.T != self.max_len - 1:
                raise ValueError("translator max_len {} != model max_len {}".format(self.max_len, engine.T + 1))
            if self.beam_size == 1:
                return self._greedy(engine, feats, kwargs.get("use_graph", True))
            return self._beam(engine, feats, kwargs.get("use_graph", True))

    _ABORTED = ("the resident decode timed out at a hand-off: its workgroups never became resident together (another "
                "long-running kernel holds CUs?); set CARE_RESIDENT_MAX_ROWS=0 / CARE_RESIDENT_BEAM_MAX_ROWS=0")

    def _greedy(self, engine, feats, use_graph):
        _, fed, length, score = engine.translate_greedy(list(feats), use_graph=use_graph, lean=True)
        fed, length, score = fed.cpu(), length.cpu().tolist(), score.cpu()
        if length and min(length) < 0:  # care_decode_resident gave up waiting for its workgroups (include/care_hip.h)
            # once more through the multi-launch decode, which needs no co-residency; raise only if that fails too
            keep, engine.resident_max_rows = engine.resident_max_rows, 0
            try:
                _, fed, length, score = engine.translate_greedy(list(feats), use_graph=False, lean=True)
                fed, length, score = fed.cpu(), length.cpu().tolist(), score.cpu()
            finally:
                engine.resident_max_rows = keep
            if length and min(length) < 0:
                raise _lib.CareHipError(self._ABORTED)
        hyps, scores = [], []
        n_best = self.topk
        for i, n in enumerate(length):
            n_best = min(n_best, 1)
            hyps.append([fed[i, 1: n + 1].tolist()][:n_best])
            scores.append([score[i].item() / (n ** self.beam_alpha)][:n_best])
        return hyps, scores

    def _beam(self, engine, feats, use_graph=True):
        # Beam.specific_nums_of_sents = max(size, topk) (Beam.py:10): with topk > beam_size a clip keeps
        # decoding until topk hypotheses have ended (or max_len)
        need = max(self.beam_size, self.topk)
        _, nfin, fscore, flen, fhyp = engine.translate_beam(list(feats), self.beam_size, need, use_graph=use_graph,
                                                             lean=True)
        nfin, fscore, flen, fhyp = nfin.cpu().tolist(), fscore.cpu(), flen.cpu(), fhyp.cpu()
        if nfin and min(nfin) < 0:  # care_decode_resident_beam aborted: the multi-launch search instead (see _greedy)
            keep, engine.resident_beam_max_rows = engine.resident_beam_max_rows, 0
            try:
                _, nfin, fscore, flen, fhyp = engine.translate_beam(list(feats), self.beam_size, need, use_graph=False, lean=True)
                nfin, fscore, flen, fhyp = nfin.cpu().tolist(), fscore.cpu(), flen.cpu(), fhyp.cpu()
            finally:
                engine.resident_beam_max_rows = keep
            if nfin and min(nfin) < 0:
                raise _lib.CareHipError(self._ABORTED)
        hyps, scores = [], []
        n_best = self.topk
        for i, nf in enumerate(nfin):
            items = [[fscore[i, j].item() / (int(flen[i, j]) ** self.beam_alpha), j] for j in range(nf)]
            items.sort(key=lambda a: -a[0])  # stable, like Beam.sort_finished (Beam.py:91-101)
            # Translator.py:211-220 re-assigns n_best inside the loop over clips: it shrinks for
            # every later clip once a clip has fewer finished hypotheses.  Reproduced as is.
            n_best = min(n_best, len(items))
            hyps.append([fhyp[i, j, : int(flen[i, j])].tolist() for _, j in items[:n_best]])
            scores.append([s for s, _ in items[:n_best]])
        return hyps, scores
