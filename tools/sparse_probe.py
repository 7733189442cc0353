"""How many (tile, row) products the sparse second pass of the beam selection recomputes (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import _lib
from tools.gemm_bench import time_call
DEV = "cuda:0"
p = lambda t: t.data_ptr() if t is not None else None
call = _lib.call
M, V, K, bm = 20480, 10547, 512, 5
torch.manual_seed(0)
A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
W = (torch.randn(V, K, device=DEV) * 0.05).to(torch.bfloat16)
parts = _lib.load().care_argmax_parts_bf16_min(M, V, K, 1, 8)
tiles = (V + 31) // 32
pm, ps = torch.empty(M, parts, device=DEV), torch.empty(M, parts, device=DEV)
pi = torch.empty(M, parts, device=DEV, dtype=torch.int32)
tmx = torch.empty(tiles, M, device=DEV)
thr = torch.empty(M, device=DEV); cnt = torch.zeros(M, device=DEV, dtype=torch.int32)
cap = 64
cval = torch.empty(M, cap, device=DEV); cidx = torch.empty(M, cap, device=DEV, dtype=torch.int32)
tcount = torch.zeros(2 * tiles + 1, device=DEV, dtype=torch.int32); tlist = torch.empty(tiles, M, device=DEV, dtype=torch.int32)
call("care_gemm_argmax_bf16_tiles", p(A), K, 1, p(W), p(pm), p(pi), p(ps), p(tmx), M, V, K, 8)
call("care_beam_threshold", p(pm), parts, bm, p(thr), p(cnt), M)
call("care_beam_sparse_collect", p(A), K, p(W), p(tmx), p(thr), p(cnt), p(cval), p(cidx), cap, p(tcount), p(tlist), M, V, K)
torch.cuda.synchronize()
print("parts %d; (tile,row) pairs %d = %.2f per row; per tile mean %.0f max %d; candidates per row mean %.2f max %d" % (
    parts, int(tcount[:tiles].sum()), float(tcount[:tiles].sum()) / M, float(tcount[:tiles].float().mean()), int(tcount[:tiles].max()), float(cnt.float().mean()), int(cnt.max())))
def sparse():
    call("care_beam_threshold", p(pm), parts, bm, p(thr), p(cnt), M)
    call("care_beam_sparse_collect", p(A), K, p(W), p(tmx), p(thr), p(cnt), p(cval), p(cidx), cap, p(tcount), p(tlist), M, V, K)
def dense():
    call("care_beam_threshold", p(pm), parts, bm, p(thr), p(cnt), M)
    call("care_gemm_collect_bf16", p(A), K, 1, p(W), p(thr), p(cnt), p(cval), p(cidx), cap, M, V, K)
print("threshold + sparse collect %.1f us; threshold + dense collect %.1f us" % (time_call(sparse, 10), time_call(dense, 10)))
print("stats with tile maxima %.1f us; without %.1f us" % (
    time_call(lambda: call("care_gemm_argmax_bf16_tiles", p(A), K, 1, p(W), p(pm), p(pi), p(ps), p(tmx), M, V, K, 8), 10),
    time_call(lambda: call("care_gemm_argmax_bf16_min", p(A), K, 1, p(W), p(pm), p(pi), p(ps), M, V, K, 8), 10)))
