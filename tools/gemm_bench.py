"""Micro-benchmark of the GEMM entry points on decode-step shapes (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib

DEV = "cuda:0"


def time_call(fn, iters=20):
    """GPU time per launch: `iters` launches captured in a hipGraph (no host launch overhead)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (5 * iters)


def main():
    import os
    shapes = [(1024, 512, 512), (4096, 512, 512), (4096, 1536, 512), (4096, 2048, 512), (4096, 10547, 512),
              (8192, 512, 512), (4096 * 84, 1024, 512), (4096 * 28, 512, 512)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    print("CARE_AS_NS=%s CARE_AS_MAP=%s" % (os.environ.get("CARE_AS_NS"), os.environ.get("CARE_AS_MAP")))
    for M, N, K in shapes:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        Af = A.float()
        W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        p = lambda t: t.data_ptr()
        if K <= 512:
            t_as = time_call(lambda: _lib.call("care_gemm_bf16", p(A), K, 1, p(W), p(bias), p(out), N, 0, None, 0, 0, N, M, N, K, 0))
        else:
            t_as = float("nan")
        t_gen = time_call(lambda: _lib.call("care_gemm", p(Af), K, p(W), 1, p(bias), p(out), N, 0, None, 0, 0, N, M, N, K, 0))
        fl = 2.0 * M * N * K
        print("M=%6d N=%5d K=%4d  A-stationary %7.1f us (%6.1f TF)   generic(fp32 A) %7.1f us (%6.1f TF)" %
              (M, N, K, t_as, fl / t_as / 1e6, t_gen, fl / t_gen / 1e6), flush=True)


def ln_main():
    """care_gemm_ln (fused LayerNorm epilogue) on embedder / decode shapes."""
    import os
    print("CARE_LN_RG=%s CARE_LN_V2=%s" % (os.environ.get("CARE_LN_RG"), os.environ.get("CARE_LN_V2")))
    for M, K, f32 in [(917504, 2048, True), (917504, 512, True), (917504, 128, True), (458752, 2048, True), (458752, 512, True),
                      (458752, 128, True), (65536, 512, False), (65536, 2048, False), (32768, 512, False), (32768, 2048, False),
                      (16384, 512, False), (16384, 2048, False), (4096, 512, False), (4096, 2048, False)]:
        A = torch.randn(M, K, device=DEV)
        Ain = A if f32 else A.to(torch.bfloat16)
        W = (torch.randn(512, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias, g, b = (torch.randn(512, device=DEV) for _ in range(3))
        res = torch.randn(M, 512, device=DEV)
        out = torch.empty(M, 512, device=DEV)
        outb = torch.empty(M, 512, device=DEV, dtype=torch.bfloat16)
        p = lambda t: t.data_ptr()
        resp = None if f32 else p(res)  # the embedder (raw fp32 features) has no residual
        outp = None if os.environ.get("LN_LEAN") else p(out)  # LN_LEAN=1: bf16 output only (the lean encode)
        if os.environ.get("LN_PACKED", "1") != "0":
            Wp = torch.empty_like(W)
            _lib.call("care_pack_ln_weight", p(W), p(Wp), 512, K)
            t = time_call(lambda: _lib.call("care_gemm_ln_packed", p(Ain), K, 0 if f32 else 1, p(Wp), p(bias), resp, 512,
                                            p(g), p(b), 1e-12, outp, p(outb), 512, M, 512, K, M, M, 0), iters=5)
        else:
            t = time_call(lambda: _lib.call("care_gemm_ln", p(Ain), K, 0 if f32 else 1, p(W), p(bias), resp, 512, None,
                                            p(g), p(b), 1e-12, outp, p(outb), 512, M, 512, K, M, M, 0), iters=5)
        nbytes = M * K * (4 if f32 else 2) + M * 512 * (6 if f32 else 10)  # A in; fp32 + bf16 out; (+ fp32 residual in)
        print("gemm_ln M=%6d K=%4d A=%s: %8.1f us (%6.1f TF, %5.2f TB/s)" % (M, K, "f32" if f32 else "bf16", t, 2.0 * M * 512 * K / t / 1e6,
                                                                          nbytes / t / 1e6), flush=True)


def f32_main():
    """care_gemm with f32 weights (exact f32 MFMA): the embedder of the concept models and the fp32 parity mode."""
    for M, N, K in [(458752, 512, 2048), (458752, 512, 512), (458752, 512, 128), (16384, 512, 1536), (16384, 2048, 512)]:
        A = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) * 0.05
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        p = lambda t: t.data_ptr()
        t = time_call(lambda: _lib.call("care_gemm", p(A), K, p(W), 0, p(bias), p(out), N, 0, None, 0, 0, N, M, N, K, 0), iters=5)
        fl = 2.0 * M * N * K
        print("f32 M=%6d N=%5d K=%4d  %8.1f us (%6.1f TF of 157.3)" % (M, N, K, t, fl / t / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ln":
        ln_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "f32":
        f32_main()
    else:
        main()
