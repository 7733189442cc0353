"""care_attention_latent: time per launch over the memory length (GPU box).  python tools/latent_sweep.py [rows] [Lk ...]"""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    lks = [int(a) for a in sys.argv[2:]] or [64, 80, 84, 96]
    H, d = 8, 512
    torch.manual_seed(0)
    qt = (torch.randn(rows, H, d, device=DEV) * 0.1).to(torch.bfloat16)
    ct = torch.zeros(rows, H, d, device=DEV, dtype=torch.bfloat16)
    for Lk in lks:
        mem = torch.randn(rows, Lk, d, device=DEV).to(torch.bfloat16)
        t = time_call(lambda: _lib.call("care_attention_latent", p(qt), H * d, p(mem), Lk * d, d, 1, Lk, None, 0,
                                        p(ct), H * d, rows, H, d), iters=10)
        byt = rows * (Lk * d * 2 + 2 * H * d * 2)
        print("rows %d Lk %3d: %7.1f us  %.2f TB/s algorithmic  (%.2f us per key-row K)" % (rows, Lk, t, byt / t / 1e6, t / Lk), flush=True)
        del mem


if __name__ == "__main__":
    main()
