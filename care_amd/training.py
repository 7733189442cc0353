"""Training-mode forward of the captioning path under torch.autograd.

The reference trains through `LightningModule.training_step` -> `self.captioner(batch)` ->
`Seq2SeqBase.feedforward_step` (models/Wrapper.py:423-435, models/Framework.py:215-237) with dropout active and
PyTorch's autograd recording every op.  Here the same forward is a chain of `torch.autograd.Function`s whose
forward AND backward are HIP kernels behind the C ABI (include/care_hip.h):

  * every nn.Linear: care_gemm (exact f32 MFMA) forward; backward = care_gemm_kn on the operands as they lie in memory
    (dx = dy W, dW = dy^T x: no transposed copies) + a two-level care_strided_sum for the bias (_colsum);
    from a few GFLOP per product on (TRAIN_GEMM "auto", round 6): the three products as SPLIT PRODUCTS at the 16-bit matrix
    rate - three fp16 MFMA passes over hi / lo pieces of operands pre-scaled by an exact power of two each (care_absmax ->
    care_split_pieces (the operands read as they lie, transposed or not) -> care_gemm_tile_split3_scaled, K in slabs for few-tile products;
    ~2^-22 per product, gradients within the same 1e-4 of the oracle's autograd);
  * LayerNorm (+ residual): care_add_ln / care_ln_bwd; activations: care_act; dropout: care_dropout (a counter-based
    generator keyed by (seed, element): the backward re-creates the forward's mask; RNG parity with torch is not a
    goal);
  * attention: care_attention_probs (softmax(QK^T / 8 + mask + bias), Attention.py:83-118) -> care_attn_pv
    (dropout(P) V) forward, care_attn_bwd backward (dQ, dK, dV and the hybrid-bias gradient) - each as small exact-f32 MFMA
    products per (sequence, head) since round 6 (csrc/backward.hip, csrc/attention.hip);
  * embeddings: care_gather_rows / care_scatter_add_rows, care_add_pos_sem; the concept head: care_concept_finish /
    care_concept_bwd; mean pooling: care_group_mean / care_bcast_rows.

torch itself only moves data (transposes, zero padding to the kernels' K % 32, cat / slicing) and runs the
autograd engine (which also sums the gradients of a tensor with several consumers).  fp32 arithmetic; the module's
own nn.Parameters are the operands, so `loss.backward()` fills their `.grad` and any torch optimiser steps them.

Scope: the `Embedder` and `MultiTransformerEncoder` encoders and every decoder variant of the hot path (Base, CARE,
CABase); the training-only
sparse-sampling branch of the concept head (pred_attribute.py:100-119, off by default) and scheduled sampling (RNN
decoders only, Framework.py:221-232) are outside it, like in eval mode.
"""
from typing import Any, Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT_CODES, CARE_F32, call, ptr
from .constants import PAD


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


def _pad_cols(t: torch.Tensor, mult: int = 32) -> torch.Tensor:
    """[M, K] -> contiguous [M, ceil(K / mult) * mult], zero filled (data movement only)."""
    M, K = t.shape
    Kp = (K + mult - 1) // mult * mult
    if Kp == K and t.is_contiguous():
        return t
    out = torch.zeros(M, Kp, device=t.device, dtype=torch.float32)
    out[:, :K] = t
    return out


def _mm(A: torch.Tensor, Bt: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A [M, K] . Bt [N, K]^T (+ bias) on the exact-f32 MFMA kernel; K is zero-padded to its multiple of 32."""
    A, Bt = _pad_cols(A), _pad_cols(Bt)
    M, K = A.shape
    N = Bt.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    call("care_gemm", ptr(A), A.stride(0), ptr(Bt), CARE_F32, ptr(bias), ptr(out), N, CARE_F32, None, 0, 0, N, M, N, K, 0)
    return out


# GEMM arithmetic of training mode (CARE_TRAIN_GEMM / set_train_gemm()):
#   "f32"    the exact-f32 MFMA on the operands as they lie in memory (care_gemm / care_gemm_kn);
#   "fp16x3" every product as three fp16 MFMA passes over hi / lo pieces of operands pre-scaled by an exact power of two each
#            (care_gemm_tile_split3_scaled: ~2^-22 relative per product - fp32-grade - at the 16-bit matrix rate / 3);
#   "auto"   (default) per product: the split form from X3_MIN_FLOPS on, the exact form below.
# *Measured* round 6 (one MI355X, msrvtt_care, forward + backward, ms per step, f32 / fp16x3 / auto): 512 clips 15.7 / 11.3 / 10.8,
# 64 clips 4.0 / 4.6 / 3.9 (19.5 / 15.8 and 4.5 / 5.1 before the attention kernels moved to the matrix cores) - the split form pays its fixed costs (two |max| sweeps, two piece writes, a slab sum: ~6 launches) back from a few
# GFLOP per product on; below, the exact kernel on the operands as they lie is faster.  Gradients of both forms meet the same
# 1e-4 of the oracle's autograd (tests/test_gpu_training.py), so a step may mix them.
import os as _os
TRAIN_GEMM = _os.environ.get("CARE_TRAIN_GEMM", "auto")
X3_MIN_FLOPS = 6.0e9   # 2 M N K of a product from which "auto" takes the split form (512 clips: every decoder product; 64 clips: the vocabulary's)


def _use_x3(M: int, N: int, K: int) -> bool:
    return TRAIN_GEMM == "fp16x3" or (TRAIN_GEMM == "auto" and 2.0 * M * N * K >= X3_MIN_FLOPS)


def _fixed_pe(P, Bf, key):
    """The sinusoid table of a model without trainable position embeddings: a frozen nn.Parameter `pe` [1, n, d] like the
    reference's (Embeddings.py:24) - among the module's parameters, not its buffers."""
    t = P[key] if key in P else Bf[key]
    return t.detach()[0]


def set_train_gemm(mode: str) -> None:
    global TRAIN_GEMM
    if mode not in ("auto", "fp16x3", "f32"):
        raise ValueError("training GEMM mode must be 'auto', 'fp16x3' or 'f32', got {!r}".format(mode))
    TRAIN_GEMM = mode


def _x3_slabs(M: int, N: int, K: int) -> int:
    """K ranges a split product is cut into: enough 128 x 128 output tiles x slabs to occupy the chip twice over, slabs of at
    least 256 columns (dW = dy^T x of a d x d weight at 512 clips: 16 tiles over K = 14848 -> 32 slabs of 512)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 256 or K < 1024:
        return 1
    return max(1, min((512 + tiles - 1) // tiles, K // 256, 64))


def _absmax_slot(t: torch.Tensor) -> torch.Tensor:
    """|max| of a tensor as the bit pattern care_split_pieces / care_gemm_tile_split3_scaled derive its power-of-two scale from:
    a 1-element int32 device tensor (care_absmax; the layout does not matter, so the tensor is swept as it lies)."""
    slot = torch.empty(1, device=t.device, dtype=torch.int32)
    flat = t.numel()
    if flat % 4 == 0 and t.is_contiguous():
        call("care_absmax", ptr(t), flat, 1, flat, slot.data_ptr())
    else:
        p4 = _pad_cols(t if t.dim() == 2 else t.reshape(1, -1), 4)
        call("care_absmax", ptr(p4), p4.shape[1], p4.shape[0], p4.shape[1], slot.data_ptr())
    return slot


def _mm_x3(A: torch.Tensor, Bt: torch.Tensor, bias: Optional[torch.Tensor] = None, a_t: bool = False, b_t: bool = False,
           a_slot: Optional[torch.Tensor] = None, b_slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """op(A) [M, K] . op(Bt) [N, K]^T (+ bias) as split products of pre-scaled operands (see TRAIN_GEMM): absolute maxima ->
    power-of-two scales -> fp16 hi / lo pieces -> the LDS-tiled product over the 3 K virtual columns, unscaled in its epilogue;
    K in slabs (added in order) when the output has few tiles.  a_t / b_t: the operand is given TRANSPOSED ([K, M] / [K, N], as
    dy and x lie in memory for dW = dy^T x) - care_split_pieces reads it as it lies.  a_slot / b_slot: the operand's |max| from
    an earlier product of the same tensor (_absmax_slot: x and W in the forward, dy once for dx and dW).  Every step a kernel
    on the current stream; the scales stay on the device."""
    A, Bt = _f32c(A), _f32c(Bt)
    M = A.shape[1] if a_t else A.shape[0]
    K = A.shape[0] if a_t else A.shape[1]
    N = Bt.shape[1] if b_t else Bt.shape[0]
    slabs = _x3_slabs(M, N, K) if bias is None else 1
    ks = ((K + slabs - 1) // slabs + 63) // 64 * 64
    dev = A.device
    a_slot = _absmax_slot(A) if a_slot is None else a_slot
    b_slot = _absmax_slot(Bt) if b_slot is None else b_slot
    a2 = torch.empty(slabs * M, 2 * ks, device=dev, dtype=torch.float16)
    w3 = torch.empty(slabs * N, 3 * ks, device=dev, dtype=torch.float16)
    call("care_split_pieces", ptr(A), A.stride(0), M, K, int(a_t), slabs, ks, ptr(a2), 2, a_slot.data_ptr())
    call("care_split_pieces", ptr(Bt), Bt.stride(0), N, K, int(b_t), slabs, ks, ptr(w3), 3, b_slot.data_ptr())
    out = torch.empty(slabs * M, N, device=dev, dtype=torch.float32)
    call("care_gemm_tile_split3_scaled", ptr(a2), ptr(w3), ptr(bias), ptr(out), N, M, N, ks, a_slot.data_ptr(), b_slot.data_ptr(), slabs)
    return out if slabs == 1 else _strided_sum(out, M, slabs, 1, M)


def _mm_kn(A: torch.Tensor, B: torch.Tensor, a_is_km: bool) -> torch.Tensor:
    """op(A) . B with B [K, N]; a_is_km: A is stored [K, M] (care_gemm_kn: the operands as they lie, no transposed copies)."""
    K, N = B.shape
    M = A.shape[1] if a_is_km else A.shape[0]
    ks = _lib.load().care_gemm_kn_splits(M, N, K)
    if ks > 1:  # few output tiles, long K (dx = dlogits W): K ranges into slabs, added in order
        slabs = torch.empty(ks * M, N, device=A.device, dtype=torch.float32)
        call("care_gemm_kn_splitk", ptr(A), A.stride(0), int(a_is_km), ptr(B), B.stride(0), ptr(slabs), N, M * N, M, N, K, ks)
        return _strided_sum(slabs, M, ks, 1, M)  # row r of slab k = row k M + r
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    call("care_gemm_kn", ptr(A), A.stride(0), int(a_is_km), ptr(B), B.stride(0), ptr(out), N, M, N, K)
    return out


def _strided_sum(x2: torch.Tensor, rows: int, terms: int, row_stride: int, term_stride: int, scale: float = 1.0):
    d = x2.shape[1]
    out = torch.empty(rows, d, device=x2.device, dtype=torch.float32)
    call("care_strided_sum", ptr(x2), x2.stride(0), ptr(out), d, rows, d, terms, row_stride, term_stride, scale)
    return out


def _colsum(dy: torch.Tensor) -> torch.Tensor:
    """Column sums of [M, d] (a bias gradient) in two levels: S row slabs summed side by side (S x d / 64 workgroups), then the S
    partial rows - care_strided_sum twice.  One level is d / 64 workgroups walking all M rows: 250 us per call at M = 14848,
    a quarter of a 512-clip training step (*measured* round 6, rocprofv3)."""
    M, d = dy.shape
    S = next((s for s in range(min(256, M // 8), 1, -1) if M % s == 0), 1)
    if S < 8:
        return _strided_sum(dy, 1, M, 0, 1).view(-1)
    part = _strided_sum(dy, S, M // S, M // S, 1)   # slab s = rows s L .. s L + L - 1
    return _strided_sum(part, 1, S, 0, 1).view(-1)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        x, W = _f32c(x), _f32c(W)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        ctx.x3 = _use_x3(x.shape[0], W.shape[0], W.shape[1])   # (the three products of a layer have the same 2 M N K)
        if ctx.x3:   # the operands' |max| once: the backward's products of x and W take them from here
            ctx.slots = (_absmax_slot(x), _absmax_slot(W))
            return _mm_x3(x, W, _f32c(b) if b is not None else None, a_slot=ctx.slots[0], b_slot=ctx.slots[1])
        return _mm(x, W, _f32c(b) if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f32c(dy)
        if ctx.x3:
            # the same split products; the operands transposed into the [rows, K] layout the tiled kernel streams (torch: data
            # movement only): dx = dy W = dy (W^T)^T, dW = dy^T x = dy^T (x^T)^T
            dy_slot = _absmax_slot(dy)
            dx = _mm_x3(dy, W, b_t=True, a_slot=dy_slot, b_slot=ctx.slots[1]) if ctx.needs_input_grad[0] else None
            dW = _mm_x3(dy, x, a_t=True, b_t=True, a_slot=dy_slot, b_slot=ctx.slots[0]) if ctx.needs_input_grad[1] else None
            db = _colsum(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            return dx, dW, db
        dx = _mm_kn(dy, W, False) if ctx.needs_input_grad[0] else None   # dy [M, out] W [out, in]
        dW = _mm_kn(dy, x, True) if ctx.needs_input_grad[1] else None    # dy^T [out, M] x [M, in]
        db = _colsum(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dW, db


class _AddLN(torch.autograd.Function):
    """y = LayerNorm(x + res) * gamma + beta (res optional)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        x = _f32c(x)
        res = _f32c(res) if res is not None else None
        rows, d = x.shape
        out = torch.empty_like(x)
        call("care_add_ln", ptr(x), d, ptr(res), d if res is not None else 0, None, ptr(gamma), ptr(beta), eps, ptr(out), None,
             d, rows, d, rows, rows, 0, 1, 0)
        ctx.save_for_backward(x, res if res is not None else x.new_empty(0), gamma)
        ctx.has_res, ctx.eps = res is not None, eps
        return out

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        rows, d = x.shape
        ds = torch.empty_like(x)
        dg, db = torch.zeros(d, device=x.device), torch.zeros(d, device=x.device)
        call("care_ln_bwd", ptr(x), d, ptr(res) if ctx.has_res else None, d if ctx.has_res else 0, ptr(gamma), ptr(dy), d,
             ctx.eps, ptr(ds), d, ptr(dg), ptr(db), rows, d)
        return ds, (ds if ctx.has_res else None), dg, db, None


class _Add(torch.autograd.Function):
    """y = x + res, the plain residual sum of a pre-LN sub-block (SubLayers.py:55,78,140,149: no LayerNorm behind it) -
    care_add_ln with gamma == beta == NULL forward, the identity twice backward."""

    @staticmethod
    def forward(ctx, x, res):
        x, res = _f32c(x), _f32c(res)
        rows, d = x.shape
        out = torch.empty_like(x)
        call("care_add_ln", ptr(x), d, ptr(res), d, None, None, None, 0.0, ptr(out), None, d, rows, d, rows, rows, 0, 1, 0)
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, act):
        z = _f32c(z)
        ctx.save_for_backward(z)
        ctx.act = act
        out = torch.empty_like(z)
        call("care_act", ptr(z), None, ptr(out), z.numel(), act)
        return out

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        dy = _f32c(dy)
        out = torch.empty_like(z)
        call("care_act", ptr(z), ptr(dy), ptr(out), z.numel(), ctx.act)
        return out, None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _f32c(x)
        ctx.p, ctx.seed = p, seed
        out = torch.empty_like(x)
        call("care_dropout", ptr(x), ptr(out), x.numel(), p, seed)
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        out = torch.empty_like(dy)
        call("care_dropout", ptr(dy), ptr(out), dy.numel(), ctx.p, ctx.seed)
        return out, None, None


class _Gather(torch.autograd.Function):
    """rows = table[idx] (nn.Embedding); the gradient of row `skip` (padding_idx) stays zero."""

    @staticmethod
    def forward(ctx, table, idx32, skip):
        table = _f32c(table)
        n, d = idx32.numel(), table.shape[1]
        out = torch.empty(n, d, device=table.device)
        call("care_gather_rows", ptr(table), table.stride(0) * 4, ptr(out), d * 4, ptr(idx32), n, d * 4)
        ctx.save_for_backward(idx32)
        ctx.shape, ctx.skip = tuple(table.shape), skip
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx32,) = ctx.saved_tensors
        dy = _f32c(dy)
        dt = torch.zeros(ctx.shape, device=dy.device)
        call("care_scatter_add_rows", ptr(dy), dy.stride(0), ptr(idx32), ptr(dt), dt.stride(0), dy.shape[0], dy.shape[1], ctx.skip)
        return dt, None, None


class _AddPosSem(torch.autograd.Function):
    """out[r] = x[r] + pos[r % seq] + sem[r // sem_div] (Embeddings.py:170-176; pos / sem optional)."""

    @staticmethod
    def forward(ctx, x, pos, sem, seq, sem_div):
        x = _f32c(x)
        pos = _f32c(pos) if pos is not None else None
        sem = _f32c(sem) if sem is not None else None
        rows, d = x.shape
        out = torch.empty_like(x)
        call("care_add_pos_sem", ptr(x), ptr(pos), ptr(sem), ptr(out), rows, d, seq, sem_div)
        ctx.seq, ctx.sem_div, ctx.rows = seq, sem_div, rows
        ctx.n_sem = sem.shape[0] if sem is not None else 0
        ctx.has_pos = pos is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        dpos = _strided_sum(dy, ctx.seq, ctx.rows // ctx.seq, 1, ctx.seq) if ctx.has_pos else None
        dsem = _strided_sum(dy, ctx.n_sem, ctx.sem_div, ctx.sem_div, 1) if ctx.n_sem else None
        return dy, dpos, dsem, None, None


class _GroupMean(torch.autograd.Function):
    """[G * n, d] -> [G, d]: mean over the n rows of a group (`item.mean(1)`, Encoder.py:106)."""

    @staticmethod
    def forward(ctx, x, n):
        x = _f32c(x)
        rows, d = x.shape
        out = torch.empty(rows // n, d, device=x.device)
        call("care_group_mean", ptr(x), d, n, 0, n, ptr(out), d, 0, rows // n, d)
        ctx.n, ctx.rows = n, rows
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        out = torch.empty(ctx.rows, dy.shape[1], device=dy.device)
        call("care_bcast_rows", ptr(dy), dy.stride(0), ptr(out), dy.shape[1], ctx.rows, dy.shape[1], ctx.n, 1.0 / ctx.n)
        return out, None


class _ConceptFinish(torch.autograd.Function):
    """scores [B, k] -> (preds [B, k], avg [B]) (prepare_merged_probs at seq_len 1, pred_attribute.py:17-46)."""

    @staticmethod
    def forward(ctx, scores):
        scores = _f32c(scores)
        B, k = scores.shape
        preds, avg = torch.empty_like(scores), torch.empty(B, device=scores.device)
        call("care_concept_finish", ptr(scores), k, ptr(preds), k, ptr(avg), B, k)
        ctx.save_for_backward(scores)
        return preds, avg

    @staticmethod
    def backward(ctx, dpreds, davg):
        (scores,) = ctx.saved_tensors
        B, k = scores.shape
        ds = torch.empty_like(scores)
        dp = _f32c(dpreds) if dpreds is not None else None
        da = _f32c(davg) if davg is not None else None
        call("care_concept_bwd", ptr(scores), k, ptr(dp), k, ptr(da), ptr(ds), k, B, k)
        return ds


class _Attention(torch.autograd.Function):
    """softmax(Q K^T / 8 + mask + bias) -> dropout -> . V for nseq sequences of `seq` queries and `nkeys` keys each
    (ScaledDotProductAttention.forward after the projections, Attention.py:83-131)."""

    @staticmethod
    def forward(ctx, q, k, v, bias, pad_tok, nseq, seq, nkeys, heads, causal, p_drop, seed):
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        d = q.shape[1]
        bias = _f32c(bias) if bias is not None else None
        probs = torch.empty(nseq * seq, heads, nkeys, device=q.device)
        call("care_attention_probs", ptr(q), d, ptr(k), CARE_F32, nkeys * d, d, seq, nkeys, 1 if causal else 0, seq, ptr(pad_tok),
             pad_tok.stride(0) if pad_tok is not None else 0, PAD, ptr(bias), bias.stride(0) if bias is not None else 0,
             ptr(probs), nseq * seq, heads)
        out = torch.empty(nseq * seq, d, device=q.device)
        call("care_attn_pv", ptr(probs), ptr(v), nkeys * d, d, ptr(out), d, nseq, seq, nkeys, heads, p_drop, seed)
        ctx.save_for_backward(q, k, v, probs)
        ctx.geo = (nseq, seq, nkeys, heads, p_drop, seed)
        ctx.bias_shape = tuple(bias.shape) if bias is not None else None
        return out

    @staticmethod
    def backward(ctx, dctx):
        q, k, v, probs = ctx.saved_tensors
        nseq, seq, nkeys, heads, p_drop, seed = ctx.geo
        dctx = _f32c(dctx)
        d = q.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dbias = torch.zeros(ctx.bias_shape, device=q.device) if ctx.bias_shape is not None else None
        call("care_attn_bwd", ptr(q), d, ptr(k), ptr(v), nkeys * d, d, ptr(probs), ptr(dctx), d, ptr(dq), d, ptr(dk), ptr(dv),
             nkeys * d, d, ptr(dbias), dbias.stride(0) if dbias is not None else 0, nseq, seq, nkeys, heads, p_drop, seed)
        return dq, dk, dv, dbias, None, None, None, None, None, None, None, None


class _Seeds:
    """One 64-bit seed per dropout site and call, derived from torch's generator (torch.manual_seed reproduces a run)."""

    def __init__(self):
        self.base = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.n = 0

    def next(self) -> int:
        self.n += 1
        return (self.base + 0x9E3779B97F4A7C15 * self.n) % (2 ** 63)


def training_forward(model, batch: Dict[str, Any], **kwargs) -> Dict[str, Any]:
    """`Seq2SeqBase.feedforward_step` in training mode (Framework.py:215-237): returns the dict the reference's
    criteria read (`logits`, `preds_attr`, `avg_prob_attr`, ... - Crit/crit_lang.py:20, crit_attribute.py:17)."""
    opt = model.opt
    P = dict(model.named_parameters())
    Bf = dict(model.named_buffers())
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("the model is on `{}`: move it to the MI355X (there is no CPU fallback)".format(dev))
    if opt["encoder"] not in ("Embedder", "MultiTransformerEncoder"):
        raise NotImplementedError("training mode covers the `Embedder` and `MultiTransformerEncoder` encoders")
    pre_ln = bool(opt.get("transformer_pre_ln", False))  # opts.py:68: LayerNorm in FRONT of every sub-block (round 6: trains too)
    if pre_ln and opt["encoder"] != "Embedder":
        raise NotImplementedError("transformer_pre_ln with a self-attention encoder is outside the hot path (as in eval mode)")
    d, H = int(opt["dim_hidden"]), int(opt["num_attention_heads"])
    eps = float(opt["layer_norm_eps"])
    act = ACT_CODES[opt["hidden_act"]]
    p_enc = float(opt.get("encoder_dropout_prob", 0.5))
    p_hid = float(opt.get("hidden_dropout_prob", 0.5))
    p_att = float(opt.get("attention_probs_dropout_prob", 0.1))
    seeds = _Seeds()
    drop = lambda x, p: _Dropout.apply(x, p, seeds.next()) if p > 0.0 else x

    def mha(pre, xq, kv2, nseq, seq, n_keys, causal, pad_tok, bias):
        """Multi-head attention sub-block (SubLayers.py:40-81) on [nseq * seq, d] queries: post-LN, or - pre_ln - LayerNorm first
        (the normalised rows are the keys / values of a self-attention too, :55-63) and the plain residual sum after."""
        sd = pre + ".SDPA."
        h_in = _AddLN.apply(xq, None, P[pre + ".LayerNorm.weight"], P[pre + ".LayerNorm.bias"], eps) if pre_ln else xq
        kv_in = h_in if kv2 is xq else kv2
        q = _Linear.apply(h_in, P[sd + "query.weight"], P.get(sd + "query.bias"))
        k = _Linear.apply(kv_in, P[sd + "key.weight"], P.get(sd + "key.bias"))
        v = _Linear.apply(kv_in, P[sd + "value.weight"], P.get(sd + "value.bias"))
        ctx_ = _Attention.apply(q, k, v, bias, pad_tok, nseq, seq, n_keys, H, causal, p_att, seeds.next())
        o = drop(_Linear.apply(ctx_, P[pre + ".dense.weight"], P[pre + ".dense.bias"]), p_hid)
        if pre_ln:
            return _Add.apply(o, xq)
        return _AddLN.apply(o, xq, P[pre + ".LayerNorm.weight"], P[pre + ".LayerNorm.bias"], eps)

    def ffn(fp, x2):
        """PositionwiseFeedForward (SubLayers.py:137-152), post-LN or pre-LN."""
        h_in = _AddLN.apply(x2, None, P[fp + ".LayerNorm.weight"], P[fp + ".LayerNorm.bias"], eps) if pre_ln else x2
        h = _Act.apply(_Linear.apply(h_in, P[fp + ".dense1.weight"], P[fp + ".dense1.bias"]), act)
        f = drop(_Linear.apply(h, P[fp + ".dense2.weight"], P[fp + ".dense2.bias"]), p_hid)
        if pre_ln:
            return _Add.apply(f, x2)
        return _AddLN.apply(f, x2, P[fp + ".LayerNorm.weight"], P[fp + ".LayerNorm.bias"], eps)

    modality = opt["modality"]
    dec_mod = opt.get("modality_for_decoder") or modality
    pred_mod = opt.get("modality_for_predictor") or modality
    has_concepts = "attribute" in opt.get("crits", [])
    if has_concepts and not (opt.get("attribute_prediction_mean_pooling") and opt.get("attribute_prediction_channel_concat")):
        # (the eval engine makes the same restriction, engine.load_weights; computing the mean-pooled channel-concat
        # head for a model configured otherwise would train the wrong function)
        raise NotImplementedError("training mode covers the concept head with attribute_prediction_mean_pooling and "
                                  "attribute_prediction_channel_concat (pred_attribute.py:78-131) only")
    has_container = "SemanticContainer" in opt.get("predictors_to_be_added", [])
    for key in ("global_semantic_guidance_not_detach", "attr_embs_no_dropout", "attribute_prediction_sparse_sampling"):
        # (pred_attribute.py:279: gradients through the guidance vector; :251: no dropout on the concept rows; :100-119: the
        # sparse-sampling branch of the concept head - training-only switches this backward does not implement)
        if has_concepts and opt.get(key, False):
            raise NotImplementedError("training mode does not cover `{}`".format(key))
    use_attr_type = opt.get("use_attr_type", "") if has_container else ""
    topk = int(opt.get("use_attr_topk", 30))

    feats = batch["feats"]
    if isinstance(feats[0], list):
        feats = feats[0]
    B = feats[0].shape[0]
    out: Dict[str, Any] = {}

    # ---- encoder streams (Encoder.py:165-168: Linear -> LayerNorm -> Dropout), means (Encoder.py:106)
    streams, means = {}, {}
    for mi, ch in enumerate(modality):
        x = feats[mi].to(dev, torch.float32).contiguous()
        n = x.shape[1]
        pre = "encoder.Encoder_{}".format(ch.upper())
        h = _Linear.apply(x.view(B * n, x.shape[2]), P[pre + ".0.weight"], P[pre + ".0.bias"])
        if opt["encoder"] == "Embedder":
            h = drop(_AddLN.apply(h, None, P[pre + ".1.weight"], P[pre + ".1.bias"], eps), p_enc)
        else:  # TransformerEncoderBase (Encoder.py:244-298): + position, LayerNorm, dropout, unmasked self-attention + FFN layers
            q1 = pre + ".1"
            pos_e = P[q1 + ".position_embeddings.weight"] if opt.get("trainable_pe", False) else _fixed_pe(P, Bf, q1 + ".position_embeddings.pe")
            h = _AddPosSem.apply(h, pos_e[:n], None, n, n)
            h = drop(_AddLN.apply(h, None, P[q1 + ".LayerNorm.weight"], P[q1 + ".LayerNorm.bias"], eps), p_hid)
            for li in range(int(opt["num_hidden_layers_encoder"])):
                lp = "{}.layers.{}".format(q1, li)
                h = mha(lp + ".intra_attention", h, h, B, n, n, False, None, None)
                h = ffn(lp + ".ffn", h)
        streams[ch] = h.view(B, n, d)
        means[ch] = _GroupMean.apply(h, n)
    mem = torch.cat([streams[ch] for ch in modality if ch in dec_mod], dim=1)
    out["mean_encoder_hidden_states"] = [means[ch] for ch in modality if ch in dec_mod]

    # ---- concept head + semantic container (pred_attribute.py:78-131,262-289)
    sem_hidden = sem_embs = None
    if has_concepts:
        pm = torch.cat([means[ch] for ch in modality if ch in pred_mod], dim=1)
        scores = _Linear.apply(pm, P["predictor.nets.0.prj.weight"], P["predictor.nets.0.prj.bias"])
        preds, avg = _ConceptFinish.apply(scores)
        out["preds_attr"], out["avg_prob_attr"] = preds, avg
        out["attribute_prediction_prj"] = model.predictor.nets[0].prj
        if has_container:
            sp = "predictor.nets.1"
            k_attr = preds.shape[1]
            labels = torch.empty(B, topk, device=dev, dtype=torch.int64)
            scratch = torch.empty(B * topk, d, device=dev)
            pd = preds.detach().contiguous()
            s2h_in = preds if opt.get("global_semantic_guidance_not_detach") else pd  # pred_attribute.py:279
            local = "L0" not in opt.get("use_attr_flags", "")  # pred_attribute.py:243-252: no concept embeddings without local guidance
            if local:
                call("care_concept_topk_embed", ptr(pd), k_attr, k_attr, topk, ptr(P[sp + ".attr_embs.word_embeddings.weight"]),
                     ptr(P[sp + ".attr_embs.position_embeddings.weight"]), ptr(P[sp + ".attr_embs.LayerNorm.weight"]),
                     ptr(P[sp + ".attr_embs.LayerNorm.bias"]), eps, ptr(labels), ptr(scratch), None, d, topk, 0, B, d)
            else:
                call("care_concept_topk_embed", ptr(pd), k_attr, k_attr, topk, None, None, None, None, eps, ptr(labels), None, None,
                     d, topk, 0, B, d)
            out["semantic_labels"] = labels
            sem_embs = None
            if local:
                e = _Gather.apply(P[sp + ".attr_embs.word_embeddings.weight"], labels.view(-1).to(torch.int32), -1)
                e = _AddPosSem.apply(e, P[sp + ".attr_embs.position_embeddings.weight"][:topk], None, topk, topk)
                e = _AddLN.apply(e, None, P[sp + ".attr_embs.LayerNorm.weight"], P[sp + ".attr_embs.LayerNorm.bias"], eps)
                if not opt.get("attr_embs_no_dropout", False):
                    e = drop(e, p_hid)
                sem_embs = e.view(B, topk, d)
            out["semantic_embs"] = sem_embs
            if "concat" in use_attr_type:
                mem = torch.cat([mem, sem_embs], dim=1)
            if "emb" in use_attr_type:
                sem_hidden = _Linear.apply(s2h_in, P[sp + ".semantic2hidden.weight"], P.get(sp + ".semantic2hidden.bias"))
            out["semantic_hidden_states"] = sem_hidden
    out["encoder_hidden_states"] = mem
    Lk = mem.shape[1]
    mem2 = mem.reshape(B * Lk, d)

    # ---- decoder (Decoder/Transformer.py:161-268), teacher forced
    ids = batch["input_ids"].to(dev)
    N, t = ids.shape
    if N != B:
        raise ValueError("training mode takes one caption per clip ({} captions for {} clips)".format(N, B))
    ids32 = ids.to(torch.int32).contiguous()
    e = "decoder.embedding"
    pos_table = P[e + ".position_embeddings.weight"] if opt.get("trainable_pe", False) else _fixed_pe(P, Bf, e + ".position_embeddings.pe")
    x = _Gather.apply(P[e + ".word_embeddings.weight"], ids32.view(-1), PAD)
    x = _AddPosSem.apply(x, pos_table[:t], sem_hidden, t, t)
    if pre_ln:  # Embeddings.py:130-131: no LayerNorm behind the embedding sum of a pre-LN decoder
        x = drop(x, p_hid)
    else:
        x = drop(_AddLN.apply(x, None, P[e + ".LayerNorm.weight"], P[e + ".LayerNorm.bias"], eps), p_hid)

    attr_att = bool(opt.get("use_attr", False)) and "att" in use_attr_type.lower()
    for li in range(int(opt["num_hidden_layers_decoder"])):
        lp = "decoder.layers.{}".format(li)
        x1 = mha(lp + ".intra_attention", x, x, N, t, t, True, ids32, None)
        x2 = mha(lp + ".inter_attention", x1, mem2, N, t, Lk, False, None, P.get(lp + ".inter_attention.SDPA.hybrid_bias"))
        if attr_att:
            x2 = mha(lp + ".attr_attention", x2, sem_embs.reshape(B * topk, d), N, t, topk, False, None,
                     P.get(lp + ".attr_attention.SDPA.hybrid_bias"))
        x = ffn(lp + ".ffn", x2)
    if pre_ln:  # Decoder/Transformer.py:80-81,233-234: the decoder's final LayerNorm, in front of the head
        x = _AddLN.apply(x, None, P["decoder.LayerNorm.weight"], P["decoder.LayerNorm.bias"], eps)
    hidden = drop(x, p_hid)  # Decoder/Transformer.py:236-237
    logits = _Linear.apply(hidden, P["cls_head.tgt_word_prj.weight"], None)
    out["hidden_states"] = hidden.view(N, t, d)
    out["logits"] = logits.view(N, t, -1)
    out["schedule_sampling_prob"] = 0
    return out
