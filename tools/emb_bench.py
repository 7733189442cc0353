"""The feature embedder's launches alone (fused Linear + LayerNorm on raw fp32 features, bf16 output only - the lean
encode of the headline pass): M = B * 28 frame rows, K = 2048 / 512 / 128.   python tools/emb_bench.py [B] [K ...]
CARE_HIP_LIB=<tool build> selects an ablation / experiment library (tools/variant_lib.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import _lib
from tools.gemm_bench import time_call

DEV = "cuda:0"


def split_main(B, Ks):
    """the concept models' embedder: fp32 weight as fp16 hi / lo pieces (care_gemm_ln_split), version 3 against version 2"""
    M = B * 28
    p = lambda t: t.data_ptr()
    for K in Ks:
        A = torch.randn(M, K, device=DEV)
        W = torch.randn(512, K, device=DEV) * 0.05
        bias, g, b = (torch.randn(512, device=DEV) for _ in range(3))
        Ws = torch.empty(512, 3 * K, device=DEV, dtype=torch.float16)
        _lib.call("care_pack_ln_weight_split", p(W), p(Ws), 512, K)
        res = {}
        for v3 in ("1", "0"):
            os.environ["CARE_LN_V3"] = v3
            outb = torch.zeros(M, 512, device=DEV, dtype=torch.bfloat16)
            fn = lambda: _lib.call("care_gemm_ln_split", p(A), K, p(Ws), p(bias), p(g), p(b), 1e-12, None, p(outb), 512, M, 512, K, M, M, 0)
            t = time_call(fn, iters=4)
            res[v3] = outb
            print("split embedder version %s  M=%6d K=%4d: %8.1f us  %6.1f TF (3 passes)" % ("3" if v3 == "1" else "2", M, K, t, 6.0 * M * 512 * K / t / 1e6), flush=True)
        del os.environ["CARE_LN_V3"]
        x = A[:4096].double() @ W.double().t() + bias.double()
        y = torch.nn.functional.layer_norm(x, (512,), g.double(), b.double(), 1e-12)
        print("    the two versions bit-identical: %s;  max |err| vs fp64 of the first 4096 rows: %.3e" %
              (torch.equal(res["1"].view(torch.int16), res["0"].view(torch.int16)), (res["1"][:4096].double() - y).abs().max().item()), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--split":
        return split_main(int(sys.argv[2]) if len(sys.argv) > 2 else 16384, [int(a) for a in sys.argv[3:]] or [2048, 512, 128])
    if len(sys.argv) > 1 and sys.argv[1] == "--sweep":  # version 3 (forced from one block) against version 2 by row count
        os.environ["EMB_CHECK"] = "0"
        for B in (128, 256, 512, 768, 1024, 1536, 2048, 4096):
            for v3 in ("0", "1"):
                os.environ["CARE_LN_V3"], os.environ["CARE_LN_V3_MIN"] = v3, "1"
                sys.argv = [sys.argv[0], str(B), "2048", "512", "128"]
                print("version", 3 if v3 == "1" else 2, end="  ")
                main()
        return
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    Ks = [int(a) for a in sys.argv[2:]] or [2048, 512, 128]
    M = B * 28
    p = lambda t: t.data_ptr()
    tag = os.environ.get("CARE_HIP_LIB", "default")
    for K in Ks:
        A = torch.randn(M, K, device=DEV)
        W = (torch.randn(512, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias, g, b = (torch.randn(512, device=DEV) for _ in range(3))
        outb = torch.empty(M, 512, device=DEV, dtype=torch.bfloat16)
        Wp = torch.empty_like(W)
        _lib.call("care_pack_ln_weight", p(W), p(Wp), 512, K)
        t = time_call(lambda: _lib.call("care_gemm_ln_packed", p(A), K, 0, p(Wp), p(bias), None, 512, p(g), p(b), 1e-12, None, p(outb),
                                        512, M, 512, K, M, M, 0), iters=4)
        if os.environ.get("EMB_CHECK", "1") != "0":  # the packed entry (version 3 where it applies) against the unpacked one (version 2)
            ref = torch.empty_like(outb)
            _lib.call("care_gemm_ln", p(A), K, 0, p(W), p(bias), None, 512, None, p(g), p(b), 1e-12, None, p(ref), 512, M, 512, K, M, M, 0)
            torch.cuda.synchronize()
            same = torch.equal(ref.view(torch.int16), outb.view(torch.int16))
            x = (A[:4096].to(torch.bfloat16).double() @ W.double().t()) + bias.double()
            y = torch.nn.functional.layer_norm(x, (512,), g.double(), b.double(), 1e-12)
            print("    bit-identical to the unpacked entry: %s;  max |err| vs fp64 of the first 4096 rows: %.3e" %
                  (same, (outb[:4096].double() - y).abs().max().item()), flush=True)
            del ref
        nbytes = M * K * 4 + M * 512 * 2
        print("%-40s M=%6d K=%4d: %8.1f us  %6.1f TF  %5.2f TB/s" % (os.path.basename(tag), M, K, t, 2.0 * M * 512 * K / t / 1e6, nbytes / t / 1e6),
              flush=True)
        del A, outb


if __name__ == "__main__":
    main()
