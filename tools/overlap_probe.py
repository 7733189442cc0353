"""Do an HBM-bound attention launch and an MFMA-bound GEMM launch overlap when issued on two HIP
streams?  Times each alone and both together (hipGraph-captured, run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None


def graph_time(fns, iters=10):
    """fns: list of (stream, callable); each stream runs its callable `iters` times, all forked from and
    joined to the capture stream."""
    for st, fn in fns:
        with torch.cuda.stream(st):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for st, fn in fns:
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(iters):
                    fn()
        for st, _ in fns:
            cur.wait_stream(st)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (3 * iters)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    Lk, d, H, V = 84, 512, 8, 10547
    q = torch.randn(rows, d, device=DEV)
    kv = torch.randn(rows * Lk, 2 * d, device=DEV).to(torch.bfloat16)
    ctx = torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)
    xb = torch.randn(rows, d, device=DEV).to(torch.bfloat16)
    W = (torch.randn(V, d, device=DEV) * 0.05).to(torch.bfloat16)
    parts = _lib.argmax_parts(V, rows, True)
    pmax = torch.empty(rows, parts, device=DEV)
    pidx = torch.empty(rows, parts, device=DEV, dtype=torch.int32)
    psum = torch.empty(rows, parts, device=DEV)
    W1 = (torch.randn(2048, d, device=DEV) * 0.05).to(torch.bfloat16)
    b1 = torch.randn(2048, device=DEV)
    hid = torch.empty(rows, 2048, device=DEV, dtype=torch.bfloat16)

    def attn():
        _lib.call("care_attention", p(q), d, p(kv), p(kv[:, d:]), 1, Lk * 2 * d, 2 * d, 1, None, 0, Lk, 0, 1, 0,
                  None, 0, 0, None, 0, p(ctx), d, 1, rows, H)

    def vocab():
        _lib.call("care_gemm_argmax_bf16", p(xb), d, 1, p(W), p(pmax), p(pidx), p(psum), None, None, rows, V, d)

    def ffn1():
        _lib.call("care_gemm_bf16", p(xb), d, 1, p(W1), p(b1), p(hid), 2048, 1, None, 0, 0, 2048, rows, 2048, d, 1)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ta = graph_time([(s1, attn)])
    for name, fn in (("vocab_argmax", vocab), ("ffn1", ffn1)):
        tg = graph_time([(s2, fn)])
        both = graph_time([(s1, attn), (s2, fn)])
        print("rows=%d  attention alone %.1f us, %s alone %.1f us, together %.1f us  (sum %.1f, max %.1f) -> overlap "
              "efficiency %.2f" % (rows, ta, name, tg, both, ta + tg, max(ta, tg), (ta + tg - both) / min(ta, tg)))


if __name__ == "__main__":
    main()
