"""Checkpoint + `opt` ingestion without PyTorch-Lightning.

The reference saves Lightning checkpoints (`train.py:76-96`, `models/Wrapper.py:27`):
a pickled dict with `state_dict` (model keys prefixed `captioner.`; criterion buffers may
sit beside them) and `hyper_parameters = {'opt': {...}, 'new_opt_used_to_override': {...}}`.
`models/__init__.py:93-152` loads them through `LightningModule.load_from_checkpoint`,
merges `{**opt, **new_opt_used_to_override}` (`Wrapper.py:29,402-403`) and rewrites the
dataset paths stored in `opt`.  Lightning is not needed for any of that; this module reads
the same file with an ALLOW-LIST unpickler - checkpoints are the "pre-trained models released by
others" case, so nothing outside tensors, plain containers, numpy scalars / arrays and `_codecs.encode`
(a bytes literal under pickle protocol 2) is ever imported or called (Lightning's `AttributeDict` becomes a
dict; any other global - `torch.storage._load_from_bytes`, a nested unrestricted load, included - raises) -
and returns a `CaptionRunner` that owns the care_amd captioner and translator.
"""
import os
import pickle
from typing import Any, Dict, Optional

import torch

from .framework import get_framework
from .translator import get_translator

PATH_KEYS = ["feats_a", "feats_m", "feats_i", "feats_o", "feats_t", "feats_r", "reference", "info_corpus"]
DEFAULT_BASE_DATA_PATH = "/data/video_datasets"  # config/Constants.py:21


class _AttrDict(dict):
    """Stand-in for pytorch_lightning.utilities.parsing.AttributeDict (a dict subclass)."""

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)


# Globals a reference checkpoint can legitimately name.  Everything else is refused: a pickle GLOBAL
# opcode is an import + attribute lookup and REDUCE calls the result, i.e. arbitrary code.
_ALLOWED_GLOBALS = {
    "collections": {"OrderedDict", "defaultdict", "Counter"},   # Counter: MultiStepLR milestones in lr_schedulers
    "_codecs": {"encode"},   # how pickle protocol 2 (torch.save's default) spells the bytes of a numpy scalar / array
    "builtins": {"dict", "list", "tuple", "set", "frozenset", "int", "float", "str", "bool", "bytes", "complex",
                 "slice", "range", "bytearray"},
    "argparse": {"Namespace"},
    "torch": {"Size", "device", "dtype", "float32", "float64", "float16", "bfloat16", "int64", "int32", "int16",
              "int8", "uint8", "bool", "FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage",
              "LongStorage", "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "Tensor"},
    "torch._utils": {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state"},
    # NOT `torch.storage._load_from_bytes`: it is torch.load(BytesIO(b), weights_only=False) - an unrestricted
    # nested unpickle of attacker bytes; zip-format checkpoints never name it
    "torch.storage": {"UntypedStorage", "TypedStorage"},
    "torch.nn.parameter": {"Parameter"},
    "numpy": {"dtype", "ndarray", "float32", "float64", "int64", "int32", "bool_"},
    "numpy.core.multiarray": {"scalar", "_reconstruct"},
    "numpy._core.multiarray": {"scalar", "_reconstruct"},
}
_LIGHTNING = ("pytorch_lightning", "lightning", "lightning_fabric", "lightning_utilities")


class _Unpickler(pickle.Unpickler):
    """Allow-list unpickler: Lightning's container classes map to dict (Lightning is never
    imported), the globals in _ALLOWED_GLOBALS resolve normally, anything else raises."""

    def find_class(self, module, name):
        if module.split(".")[0] in _LIGHTNING:
            return _AttrDict
        if name in _ALLOWED_GLOBALS.get(module, ()):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(
            "checkpoint names the global `{}.{}`, which a CARE checkpoint has no use for; refusing to "
            "import it (care_amd/checkpoint.py allow-list)".format(module, name))


class _PickleModule:
    """Duck-typed `pickle_module` for torch.load."""
    __name__ = "care_amd_checkpoint_pickle"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps = staticmethod(pickle.dump), staticmethod(pickle.dumps)


def read_checkpoint(path: str) -> Dict[str, Any]:
    """Return {'state_dict': captioner weights (prefix stripped), 'opt', 'new_opt', 'extra_keys'}."""
    ckpt = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    if "state_dict" not in ckpt or "hyper_parameters" not in ckpt:
        raise ValueError("{} is not a Lightning checkpoint of the reference (state_dict / hyper_parameters)".format(path))
    hp = ckpt["hyper_parameters"]
    if "opt" not in hp:
        raise ValueError("hyper_parameters has no `opt` (Wrapper.py:27 saves opt and new_opt_used_to_override)")
    sd, extra = {}, []
    for k, v in ckpt["state_dict"].items():
        if k.startswith("captioner."):
            sd[k[len("captioner."):]] = v
        else:
            extra.append(k)
    return {"state_dict": sd, "opt": dict(hp["opt"]), "new_opt": dict(hp.get("new_opt_used_to_override", {}) or {}),
            "extra_keys": extra}


def replace_data_paths(opt: Dict[str, Any], base_data_path: Optional[str]) -> Dict[str, Any]:
    """models/__init__.py:127-148: re-root the stored dataset paths under `base_data_path`."""
    if "info_corpus" not in opt:
        return opt
    ori = os.path.dirname(opt["info_corpus"])
    if os.path.basename(ori) != opt.get("dataset"):
        raise AssertionError("info_corpus `{}` does not live in a `{}` directory".format(opt["info_corpus"], opt.get("dataset")))
    ori = os.path.dirname(ori)
    now = base_data_path if base_data_path is not None else DEFAULT_BASE_DATA_PATH

    def rep(item):
        if isinstance(item, (list, tuple)):
            return [rep(x) for x in item]
        if not isinstance(item, str):
            raise AssertionError("path entries must be str")
        return item.replace(ori, now)

    opt = dict(opt)
    for key in PATH_KEYS:
        if key in opt:
            opt[key] = rep(opt[key])
    return opt


class CaptionRunner:
    """Minimal non-Lightning stand-in for `models.Wrapper.ModelBase` around the hot path.

    Holds `captioner` (get_framework) and `translator` (get_translator) exactly like
    `ModelBase.__init__` (Wrapper.py:29-36) and offers `get_opt`, `get_keys_to_device`,
    `translate_step` (Wrapper.py:158-212 without the metric bookkeeping).
    """

    def __init__(self, opt: Dict[str, Any], new_opt_used_to_override: Optional[Dict[str, Any]] = None):
        self.opt = dict(opt)
        self.new_opt_used_to_override = dict(new_opt_used_to_override or {})
        newest = self.get_opt()
        self.captioner = get_framework(newest)
        self.translator = get_translator(newest)

    def get_opt(self) -> Dict[str, Any]:
        return {**self.opt, **self.new_opt_used_to_override}

    def get_keys_to_device(self, *args, **kwargs):
        return self.captioner.get_keys_to_device(*args, **kwargs)

    def eval(self):
        self.captioner.eval()
        return self

    def to(self, device):
        self.captioner.to(device)
        return self

    @staticmethod
    def _records(batch, hyps, scores, vocab):
        from .text import to_sentence

        if vocab is None:
            return hyps, scores
        out = []
        for i, (hs, ss) in enumerate(zip(hyps, scores)):
            vid = batch["video_ids"][i] if "video_ids" in batch else i
            out.append({"image_id": vid, "caption": to_sentence(hs[0], vocab), "score": ss[0]})
        return out

    def translate_step(self, batch: Dict[str, Any], vocab: Optional[Dict[int, str]] = None):
        hyps, scores = self.translator.translate_batch(models=[self.captioner], batch=batch, vocab=vocab)
        return self._records(batch, hyps, scores, vocab)

    def translate_steps(self, batches, vocab: Optional[Dict[int, str]] = None):
        """`translate_step` over a loader (translate.py:34-51's loop), pipelined: batch k's captions are assembled on the
        host while batch k + 1 decodes (Translator_ARFormer.translate_batches).  Yields what translate_step returns, in
        order; a batch's feature tensors must stay in place until its results are out (FeaturePrefetcher(depth=3))."""
        seen = []

        def tap():
            for b in batches:
                seen.append(b)
                yield b

        for hyps, scores in self.translator.translate_batches([self.captioner], tap(), vocab=vocab):
            yield self._records(seen.pop(0), hyps, scores, vocab)


class EnsembleRunner(CaptionRunner):
    """`models.Wrapper.ModelEnsemble` (Wrapper.py:617-714) around the hot path: several checkpoints decode together, their word
    log-probabilities averaged at every step (Translator_ARFormer.translate_batch over the list).  `captioner` is the LIST of
    frameworks; `opt` is the first checkpoint's with the members' feature paths merged and - when the members' modalities
    differ - `modality` = the union of theirs, the loader then yields one tensor per modality of the union and
    `preprocess_batch_before_translate_step` hands every member the tensors of its own modalities (Wrapper.py:680-693).
    The union's order: first appearance over the members (the reference walks `list(set(...))`, an order that depends on the
    interpreter's string hashing; any loader built from `get_opt()['modality']` sees the same order as this runner)."""

    def __init__(self, opts, new_opt_used_to_override: Optional[Dict[str, Any]] = None):
        if not opts:
            raise ValueError("an ensemble needs at least one checkpoint")
        opt, modalities, full = None, [], ""
        for o in opts:
            modalities.append(o["modality"])
            full += o["modality"]
            if opt is None:
                opt = dict(o)
                continue
            for ch in o["modality"]:   # Wrapper.py:653-663: the same modality must mean the same feature files
                key = "feats_" + ch
                if ch in opt["modality"]:
                    if list(o.get(key, [])) != list(opt.get(key, [])):
                        raise AssertionError("checkpoints disagree on {}: {} / {}".format(key, o.get(key), opt.get(key)))
                elif key in o:
                    opt[key] = o[key]
        self.need_to_split_feats = len(set(modalities)) != 1
        self.modality_of_all_checkpoints = modalities
        if self.need_to_split_feats:
            opt["modality"] = "".join(dict.fromkeys(full))
        self.opt = opt
        self.new_opt_used_to_override = dict(new_opt_used_to_override or {})
        # (the members are built from their own stored options, Wrapper.py:641; the overrides reach the translator only, :676)
        self.captioner = [get_framework(dict(o)) for o in opts]
        self.translator = get_translator(self.get_opt())

    def get_keys_to_device(self, *args, **kwargs):
        return self.captioner[0].get_keys_to_device(*args, **kwargs)

    def eval(self):
        for m in self.captioner:
            m.eval()
        return self

    def to(self, device):
        for m in self.captioner:
            m.to(device)
        return self

    def preprocess_batch_before_translate_step(self, batch):
        """Wrapper.py:680-693: `batch['feats']` (one tensor per modality of the union) -> one feature list per member."""
        if self.need_to_split_feats:
            union = self.get_opt()["modality"]
            batch = dict(batch)
            batch["feats"] = [[batch["feats"][union.index(ch)] for ch in modality] for modality in self.modality_of_all_checkpoints]
        return batch

    def translate_step(self, batch: Dict[str, Any], vocab: Optional[Dict[int, str]] = None):
        batch = self.preprocess_batch_before_translate_step(batch)
        hyps, scores = self.translator.translate_batch(models=self.captioner, batch=batch, vocab=vocab)
        return self._records(batch, hyps, scores, vocab)

    def translate_steps(self, batches, vocab: Optional[Dict[int, str]] = None):
        for batch in batches:   # (the ensemble search is eager: nothing to pipeline behind it)
            yield self.translate_step(batch, vocab)


def load_model(checkpoint_path: str, new_opt_used_to_override: Optional[Dict[str, Any]] = None, device="cuda:0",
               strict: bool = True, replace_paths: bool = True, base_data_path: Optional[str] = None,
               compute_dtype: Optional[str] = None) -> CaptionRunner:
    """`models.load_model` (models/__init__.py:93-152): one checkpoint -> CaptionRunner, a list of them -> EnsembleRunner.

    Like the reference (`new_opt_used_to_override={}` by default, handed to `load_from_checkpoint`,
    where it REPLACES the saved hyper-parameter), the overrides stored in the checkpoint are dropped
    unless the caller passes them again; `read_checkpoint(path)["new_opt"]` returns the stored ones."""
    if isinstance(checkpoint_path, (list, tuple)):   # models/__init__.py:104-113 -> ModelEnsemble
        cks = [read_checkpoint(p) for p in checkpoint_path]
        opts = [replace_data_paths(ck["opt"], base_data_path) if replace_paths else ck["opt"] for ck in cks]
        runner = EnsembleRunner(opts, dict(new_opt_used_to_override or {}))
        for model, ck in zip(runner.captioner, cks):
            missing, unexpected = model.load_state_dict(ck["state_dict"], strict=strict)
            if strict and (missing or unexpected):
                raise RuntimeError("checkpoint / model key mismatch: missing {} unexpected {}".format(missing, unexpected))
            if compute_dtype is not None:
                model.set_compute_dtype(compute_dtype)
        runner.eval()
        return runner.to(device) if device is not None else runner
    ck = read_checkpoint(checkpoint_path)
    override = dict(new_opt_used_to_override or {})
    opt = replace_data_paths(ck["opt"], base_data_path) if replace_paths else ck["opt"]
    runner = CaptionRunner(opt, override)
    missing, unexpected = runner.captioner.load_state_dict(ck["state_dict"], strict=strict)
    if strict and (missing or unexpected):
        raise RuntimeError("checkpoint / model key mismatch: missing {} unexpected {}".format(missing, unexpected))
    if compute_dtype is not None:
        runner.captioner.set_compute_dtype(compute_dtype)
    runner.eval()
    return runner.to(device) if device is not None else runner
