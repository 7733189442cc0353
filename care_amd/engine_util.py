"""Small helpers shared by the parts of care_amd.engine.HipEngine."""
from typing import Optional

import torch



class _LaneOutputs(dict):
    """Encoder outputs of a pass that ran as several batch lanes.  Every value is per clip (first
    dim = clips of the lane), so the full-batch tensor is the concatenation of the lanes'; it is
    built on access only - the captioning loop never reads these (translator.py), and
    `encoder_hidden_states` alone is 2.9 GB at B = 16384."""

    def __init__(self, parts):
        super().__init__((k, None) for k in parts[0])
        self._parts = parts

    @staticmethod
    def _join(vals):
        if vals[0] is None:
            return None
        if isinstance(vals[0], (list, tuple)):
            return [torch.cat([v[i] for v in vals], 0) for i in range(len(vals[0]))]
        return torch.cat(vals, 0)

    def __getitem__(self, k):
        super().__getitem__(k)  # KeyError for unknown names
        # joined on EVERY access: the lanes' tensors are static graph outputs that the next replay
        # overwrites, so a cached concatenation would go stale
        return self._join([pt[k] for pt in self._parts])

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]


import threading as _threading

_PROPS, _PROPS_LOCK = {}, _threading.Lock()


def device_props(device):
    """torch.cuda.get_device_properties, once per device and under a lock: threads that touch the GPU for the first time at the
    same moment race inside torch's lazy initialisation ("Invalid device id" from a worker thread; tools/soak.py)."""
    key = torch.device(device).index or 0
    with _PROPS_LOCK:
        if key not in _PROPS:
            torch.cuda.init()
            _PROPS[key] = torch.cuda.get_device_properties(key)
        return _PROPS[key]
