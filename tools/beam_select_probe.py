"""Time care_beam_select on the beam-5 decode shape (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"
rows, V, ld, bm = 20480, 10547, 10560, 5
buf = torch.randn(rows, ld, device=DEV) * 2.0
cv = torch.zeros(rows, bm, device=DEV)
ci = torch.zeros(rows, bm, device=DEV, dtype=torch.int32)
t = time_call(lambda: _lib.call("care_beam_select", buf.data_ptr(), ld, V, bm, cv.data_ptr(), ci.data_ptr(), rows), iters=5)
print("beam_select %d x %d: %.1f us  (%.2f TB/s)" % (rows, V, t, rows * V * 4 / t / 1e6))
