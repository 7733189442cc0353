"""Import the genuine reference (read-only, /root/reference) in the build container.

Only used by `gen_golden.py` and by the optional `tests/test_oracle_vs_reference.py`
(skipped when /root/reference is absent, e.g. on the GPU box).  Two third-party
packages the reference imports eagerly are absent from this image and irrelevant to
the forward path; they are replaced by empty stand-in *modules* (not reference code):
`pytorch_lightning` (models/__init__.py:1 -> Wrapper.py) and `pycocoevalcap`
(misc/cocoeval.py:4-9).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CARE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "Framework.py"))


def _stub_modules():
    import torch.nn as nn

    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            def save_hyperparameters(self, *a, **k):
                pass

        pl.LightningModule = LightningModule
        pl.seed_everything = lambda *a, **k: None
        sys.modules["pytorch_lightning"] = pl
    for name in ("pycocoevalcap", "pycocoevalcap.tokenizer", "pycocoevalcap.tokenizer.ptbtokenizer",
                 "pycocoevalcap.bleu", "pycocoevalcap.bleu.bleu", "pycocoevalcap.meteor",
                 "pycocoevalcap.meteor.meteor", "pycocoevalcap.rouge", "pycocoevalcap.rouge.rouge",
                 "pycocoevalcap.cider", "pycocoevalcap.cider.cider", "pycocoevalcap.spice",
                 "pycocoevalcap.spice.spice"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            for cls in ("PTBTokenizer", "Bleu", "Meteor", "Rouge", "Cider", "Spice"):
                setattr(mod, cls, type(cls, (), {}))
            sys.modules[name] = mod


def import_reference():
    """Return (get_framework, get_translator) of the genuine reference."""
    if not reference_available():
        raise RuntimeError("reference tree not found at {}".format(REFERENCE_ROOT))
    _stub_modules()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from models.Framework import get_framework  # noqa: E402
    from models.Translator import get_translator  # noqa: E402

    return get_framework, get_translator
