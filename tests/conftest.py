import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


class GoldenCase:
    """One fixture: reference outputs + everything needed to regenerate its inputs."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(str(self.z["meta_json"]))

    _built = {}

    def build(self):
        """(opt, state dict, feats, input_ids) regenerated from the seeds - memoised per fixture: the
        portable generator costs seconds per model and a dozen tests ask for the same one."""
        if self.name not in GoldenCase._built:
            GoldenCase._built[self.name] = self._build()
        opt, P, feats, ids = GoldenCase._built[self.name]
        return dict(opt), dict(P), list(feats), ids

    def _build(self):
        import torch  # noqa: F401
        from care_amd.configs import feat_shapes, make_opt
        from care_amd.synth import synth_feats, synth_input_ids, synth_state_dict, tensor_sha256

        m = self.meta
        opt = make_opt(m["config"], **m["overrides"])
        row_scale = {k: {int(r): f for r, f in v.items()} for k, v in m["row_scale"].items()}
        P = synth_state_dict(m["seed"], [(k, tuple(s)) for k, s in m["state_dict"]], row_scale=row_scale)
        feats = synth_feats(m["seed"], feat_shapes(opt, m["batch"]))
        ids = synth_input_ids(m["seed"], m["batch"], opt["max_len"] - 1, opt["vocab_size"])
        # the regenerated tensors must be the ones the reference saw
        assert tensor_sha256(feats[0]) == m["sha256"]["feats0"]
        assert tensor_sha256(P["cls_head.tgt_word_prj.weight"]) == m["sha256"]["cls_head.tgt_word_prj.weight"]
        assert tensor_sha256(ids) == m["sha256"]["input_ids"]
        return opt, P, feats, ids

    def hyps(self):
        """Reference hypotheses as List[B][n_best][len] of python ints, and scores."""
        arr, lens, sc = self.z["hyps"], self.z["hyp_lens"], self.z["hyp_scores"]
        hyps, scores = [], []
        for i in range(arr.shape[0]):
            hs, ss = [], []
            for j in range(arr.shape[1]):
                if lens[i, j] > 0:
                    hs.append([int(x) for x in arr[i, j, : lens[i, j]]])
                    ss.append(float(sc[i, j]))
            hyps.append(hs)
            scores.append(ss)
        return hyps, scores


ENSEMBLE_DIR = os.path.join(GOLDEN_DIR, "ensemble")


def ensemble_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ENSEMBLE_DIR, "*.npz")))


class EnsembleCase(GoldenCase):
    """One model-ensembling fixture (oracle/gen_golden.py ENSEMBLE_CASES): the reference Translator's hypotheses over a LIST of
    reference models, and the recipe of every member."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(ENSEMBLE_DIR, name + ".npz"))
        self.meta = json.loads(str(self.z["meta_json"]))

    def build(self):
        """(opts, state dicts, feature lists) of the members, regenerated from the seeds."""
        from care_amd.configs import feat_shapes, make_opt
        from care_amd.synth import synth_feats, synth_state_dict, tensor_sha256

        if self.name not in GoldenCase._built:
            opts, Ps, feats = [], [], []
            for m in self.meta["members"]:
                opt = make_opt(m["config"], **m["overrides"])
                row_scale = {k: {int(r): f for r, f in v.items()} for k, v in m["row_scale"].items()}
                P = synth_state_dict(m["seed"], [(k, tuple(sh)) for k, sh in m["state_dict"]], row_scale=row_scale)
                f = synth_feats(m["feats_seed"], feat_shapes(opt, self.meta["batch"]))
                assert tensor_sha256(f[0]) == m["sha256"]["feats0"]
                assert tensor_sha256(P["cls_head.tgt_word_prj.weight"]) == m["sha256"]["cls_head.tgt_word_prj.weight"]
                opts.append(opt); Ps.append(P); feats.append(f)
            GoldenCase._built[self.name] = (opts, Ps, feats)
        opts, Ps, feats = GoldenCase._built[self.name]
        return [dict(o) for o in opts], [dict(P) for P in Ps], [list(f) for f in feats]


@pytest.fixture(params=golden_names())
def golden(request):
    return GoldenCase(request.param)
