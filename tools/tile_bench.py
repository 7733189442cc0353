"""Micro-benchmark of the LDS-tiled bf16 GEMM (csrc/gemm_tile.hip) on d_model = 768 / 1024 decode shapes, every
tile shape / ring depth (CARE_TILE_CFG) interleaved in one process (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"


def main():
    shapes = [(4096, 3072, 1024), (4096, 1024, 1024), (4096, 4096, 1024), (4096, 1024, 4096), (4096, 10547, 1024),
              (16384, 3072, 1024), (16384, 1024, 1024), (16384, 4096, 1024), (16384, 1024, 4096), (16384, 10547, 1024),
              (4096 * 114, 2048, 1024), (4096, 2304, 768), (4096, 10547, 768), (16384, 2304, 768), (16384, 768, 3072),
              (4096 * 29, 1536, 512), (4096 * 29, 512, 2048), (4096 * 29, 2048, 512), (4096 * 29, 10547, 512),
              (4096, 4096, 4096), (8192, 8192, 8192), (640, 4096, 1024)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    cfgs = os.environ.get("TILE_BENCH_CFGS", "222,4412,90").split(",")
    p = lambda t: t.data_ptr()
    for M, N, K in shapes:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N + (-N) % 8, device=DEV, dtype=torch.bfloat16)
        parts = _lib.load().care_argmax_parts_tile(N)
        pm, ps = torch.empty(M, parts, device=DEV), torch.empty(M, parts, device=DEV)
        pi = torch.empty(M, parts, device=DEV, dtype=torch.int32)
        fl = 2.0 * M * N * K
        line = "M=%6d N=%5d K=%4d " % (M, N, K)
        first = None
        for cfg in cfgs:
            os.environ["CARE_TILE_CFG"] = cfg
            out.zero_()
            t = time_call(lambda: _lib.call("care_gemm_tile", p(A), K, p(W), p(bias), p(out), out.stride(0), 1, None, 0, 0, N,
                                            M, N, K, 0))
            same = ""
            if first is None:
                first = out.clone()
            else:
                same = " =" if torch.equal(first.view(torch.int16), out.view(torch.int16)) else " DIFFERS"
            line += " %s: %7.1f us %6.1f TF%s |" % (cfg, t, fl / t / 1e6, same)
        if N > 8192:
            for cfg in ("4412", "90"):
                os.environ["CARE_TILE_CFG"] = cfg
                t = time_call(lambda: _lib.call("care_gemm_tile_argmax", p(A), K, p(W), p(pm), p(pi), p(ps), None, None, M, N, K))
                line += " argmax %s: %7.1f us %6.1f TF |" % (cfg, t, fl / t / 1e6)
        del os.environ["CARE_TILE_CFG"]
        print(line, flush=True)


if __name__ == "__main__":
    main()
