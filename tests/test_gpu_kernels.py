"""GPU: each entry point of the C ABI against a plain PyTorch fp32 reference of the same op.

bf16 kernels are compared with the same math on bf16-ROUNDED operands accumulated in
fp32/fp64 (that is what the MFMA computes), so the tolerance only has to cover summation
order: 2e-3 absolute on O(1..30) outputs.  fp32 kernels: 1e-4 at K <= 2048.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _call(name, *args):
    from care_amd import _lib

    _lib.call(name, *args)


def _p(t):
    return None if t is None else t.data_ptr()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


GEMM_SHAPES = [
    (1, 512, 512), (5, 10547, 512), (64, 512, 512), (100, 1536, 512), (257, 2048, 512), (300, 512, 2048),
    (1024, 512, 128), (28 * 7, 512, 2048), (130, 640, 640), (96, 500, 2048), (200, 768, 768),
    (1000, 48, 512), (129, 1040, 256), (4096, 512, 512), (513, 10547, 384),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("mode", ["f32", "bf16_generic", "bf16_as_f32A", "bf16_as_bf16A"])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm(M, N, K, mode, act):
    if mode != "f32" and act == 2 and M > 64:
        pytest.skip("activation covered on small shapes")
    A = _rand(M, K, seed=1)
    W = _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    bias = _rand(N, seed=3)
    out = torch.full((M, N), float("nan"), device=DEV)
    if mode == "f32":
        _call("care_gemm", _p(A), K, _p(W), 0, _p(bias), _p(out), N, 0, None, 0, 0, N, M, N, K, act)
        ref = A.double() @ W.double().t() + bias.double()
        tol = 2e-4
    else:
        Wb = W.to(torch.bfloat16).contiguous()
        ref = _bf(A).double() @ _bf(W).double().t() + bias.double()
        tol = 3e-3
        if mode == "bf16_generic":
            if K % 64:
                pytest.skip("generic bf16 kernel needs K % 64 == 0")
            _call("care_gemm", _p(A), K, _p(Wb), 1, _p(bias), _p(out), N, 0, None, 0, 0, N, M, N, K, act)
        else:
            if K % 128 or K > 512:
                pytest.skip("A-stationary kernel needs K % 128 == 0 and K <= 512 (K > 512: split-K entry point)")
            Ain = A if mode == "bf16_as_f32A" else A.to(torch.bfloat16).contiguous()
            _call("care_gemm_bf16", _p(Ain), K, 0 if mode == "bf16_as_f32A" else 1, _p(Wb), _p(bias), _p(out), N, 0,
                  None, 0, 0, N, M, N, K, act)
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize("kernel", ["care_gemm", "care_gemm_bf16"])
def test_gemm_split_destinations_and_bf16_out(kernel):
    M, K, d = 77, 512, 512
    A, W, bias = _rand(M, K, seed=4), _rand(3 * d, K, seed=5, scale=0.05), _rand(3 * d, seed=6)
    Wb = W.to(torch.bfloat16).contiguous()
    q = torch.zeros(M, d, device=DEV)
    cache = torch.zeros(M, 29, 2 * d, device=DEV, dtype=torch.bfloat16)
    dst = cache[:, 7, :]
    if kernel == "care_gemm":
        _call(kernel, _p(A), K, _p(Wb), 1, _p(bias), _p(q), d, 0, _p(dst), dst.stride(0), 1, d, M, 3 * d, K, 0)
    else:
        _call(kernel, _p(A), K, 0, _p(Wb), _p(bias), _p(q), d, 0, _p(dst), dst.stride(0), 1, d, M, 3 * d, K, 0)
    ref = _bf(A) @ _bf(W).t() + bias
    torch.cuda.synchronize()
    assert (q - ref[:, :d]).abs().max().item() < 3e-3
    assert (cache[:, 7, :].float() - ref[:, d:]).abs().max().item() < 2e-2  # bf16 output rounding
    assert cache[:, 6, :].abs().max().item() == 0 and cache[:, 8, :].abs().max().item() == 0


@pytest.mark.parametrize("M,N,act,out_bf16", [(8192, 2048, 1, True), (8192 + 100, 512, 0, False), (12000, 2048, 2, True),
                                              (33000, 512, 0, True), (8192, 4096, 1, True)])
def test_gemm_store_256_row_panels(M, N, act, out_bf16):
    """csrc/gemm_store32.hip (bf16 activations, K = 512, M >= 8192): bias + activation + 16-byte stores of
    16 consecutive columns per lane, against torch on the bf16-rounded operands; a ragged last panel."""
    K = 512
    A = _rand(M, K, seed=61).to(torch.bfloat16).contiguous()
    W = _rand(N, K, seed=62, scale=0.05)
    bias = _rand(N, seed=63)
    Wb = W.to(torch.bfloat16).contiguous()
    out = torch.full((M + 3, N + 8), 7.0, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dst = out[:M, :N]
    _call("care_gemm_bf16", _p(A), K, 1, _p(Wb), _p(bias), _p(dst), dst.stride(0), 1 if out_bf16 else 0, None, 0, 0, N, M, N, K, act)
    ref = A.float() @ Wb.float().t() + bias
    ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref) if act == 2 else ref)
    torch.cuda.synchronize()
    assert (dst.float() - ref).abs().max().item() < (3e-2 if out_bf16 else 3e-3)
    assert float(out[M:].float().min()) == 7.0 and float(out[:, N:].float().min()) == 7.0   # nothing written outside


def test_gemm_store_256_row_panels_split_destinations():
    """The QKV projection of a decode step at 16384 rows: q fp32 [M, 512] and k | v bf16 straight into
    position 7 of the [M, 29, 1024] cache (columns >= n_split, own leading dimension)."""
    M, K, d = 16384 + 5, 512, 512
    A = _rand(M, K, seed=64).to(torch.bfloat16).contiguous()
    W, bias = _rand(3 * d, K, seed=65, scale=0.05), _rand(3 * d, seed=66)
    Wb = W.to(torch.bfloat16).contiguous()
    q = torch.zeros(M, d, device=DEV)
    cache = torch.zeros(M, 29, 2 * d, device=DEV, dtype=torch.bfloat16)
    dst = cache[:, 7, :]
    _call("care_gemm_bf16", _p(A), K, 1, _p(Wb), _p(bias), _p(q), d, 0, _p(dst), dst.stride(0), 1, d, M, 3 * d, K, 0)
    ref = A.float() @ Wb.float().t() + bias
    torch.cuda.synchronize()
    assert (q - ref[:, :d]).abs().max().item() < 3e-3
    assert (cache[:, 7, :].float() - ref[:, d:]).abs().max().item() < 3e-2
    assert cache[:, 6, :].abs().max().item() == 0 and cache[:, 8, :].abs().max().item() == 0


@pytest.mark.parametrize("M", [1, 3, 64, 129, 1000])
@pytest.mark.parametrize("mode", ["f32", "bf16_generic", "bf16_as"])
def test_gemm_argmax(M, mode):
    from care_amd import _lib

    N, K = 10547, 512
    A = _rand(M, K, seed=7)
    W = _rand(N, K, seed=8, scale=0.05)
    if mode == "f32":
        parts = _lib.argmax_parts(N)
        ref = A.double() @ W.double().t()
        Wd, code = W, 0
    else:
        Wd, code = W.to(torch.bfloat16).contiguous(), 1
        parts = _lib.argmax_parts(N, M, mode == "bf16_as")
        ref = _bf(A).double() @ _bf(W).double().t()
    pmax = torch.empty(M, parts, device=DEV)
    pidx = torch.empty(M, parts, device=DEV, dtype=torch.int32)
    psum = torch.empty(M, parts, device=DEV)
    if mode == "bf16_as":
        _call("care_gemm_argmax_bf16", _p(A), K, 0, _p(Wd), _p(pmax), _p(pidx), _p(psum), None, None, M, N, K)
    else:
        _call("care_gemm_argmax", _p(A), K, _p(Wd), code, _p(pmax), _p(pidx), _p(psum), M, N, K)
    fed = torch.zeros(M, 30, device=DEV, dtype=torch.int32)
    score = torch.zeros(M, device=DEV)
    length = torch.zeros(M, device=DEV, dtype=torch.int32)
    fin = torch.zeros(M, device=DEV, dtype=torch.int32)
    _call("care_greedy_update", _p(pmax), _p(pidx), _p(psum), parts, _p(fed), 30, _p(score), _p(length), _p(fin), 1,
          29, 3, M)
    torch.cuda.synchronize()
    top2 = ref.topk(2, dim=1)
    safe = (top2[0][:, 0] - top2[0][:, 1]) > (1e-5 if mode == "f32" else 1e-3)
    assert torch.equal(fed[:, 1][safe].long(), top2[1][:, 0][safe])
    logp = torch.log_softmax(ref, dim=1).gather(1, fed[:, 1:2].long()).squeeze(1)
    assert (score.double() - logp).abs().max().item() < (1e-4 if mode == "f32" else 2e-3)
    assert torch.all(length == 1)


@pytest.mark.parametrize("M,N,min_parts", [(8192, 10547, 1), (8192 + 77, 10547, 1), (12288, 5000, 8), (33000, 10547, 1),
                                           (8192, 130, 1), (20480, 10547, 8), (9000, 10547, 24)])
def test_vocab_argmax_256_row_panels(M, N, min_parts, monkeypatch):
    """csrc/gemm_vocab.hip (bf16 activations, K = 512, M >= 8192): per column range (max, lowest argmax,
    sum-exp) of the logits - against the bf16-rounded product in fp64, against the 128-row kernel of
    gemm_as.hip (CARE_V32_MIN_ROWS beyond M switches it off), with exact ties (duplicated W rows: the
    lower column must win) and a ragged last tile."""
    from care_amd import _lib

    K = 512
    A = _rand(M, K, seed=17).to(torch.bfloat16).contiguous()
    W = _rand(N, K, seed=18, scale=0.05)
    W[N // 2 + 5] = W[11]            # exact ties between far-apart columns and ...
    W[N - 1] = W[N - 2]              # ... inside the ragged last tile
    A[5] = (W[11] * 40).to(torch.bfloat16)   # rows whose maximum IS the tied column
    A[M - 1] = (W[N - 2] * 40).to(torch.bfloat16)
    Wb = W.to(torch.bfloat16).contiguous()
    parts = (_lib.load().care_argmax_parts_bf16_min(M, N, K, 1, min_parts) if min_parts > 1 else
             _lib.load().care_argmax_parts_bf16(M, N))

    def run():
        pm, ps = torch.full((M, parts), float("nan"), device=DEV), torch.full((M, parts), float("nan"), device=DEV)
        pi = torch.full((M, parts), -7, device=DEV, dtype=torch.int32)
        if min_parts > 1:
            _call("care_gemm_argmax_bf16_min", _p(A), K, 1, _p(Wb), _p(pm), _p(pi), _p(ps), M, N, K, min_parts)
        else:
            _call("care_gemm_argmax_bf16", _p(A), K, 1, _p(Wb), _p(pm), _p(pi), _p(ps), None, None, M, N, K)
        torch.cuda.synchronize()
        return pm, pi, ps

    pm, pi, ps = run()
    # merge the ranges like care_greedy_update does: max, lowest index among equal maxima, log-sum-exp
    best = pm.max(1).values
    lse = torch.log((ps.double() * torch.exp(pm.double() - best.double().unsqueeze(1))).sum(1)) + best.double()
    cand = torch.where(pm == best.unsqueeze(1), pi, torch.full_like(pi, 0x7fffffff)).min(1).values
    ref = A.float().double() @ Wb.float().double().t()
    assert (lse - torch.logsumexp(ref, 1)).abs().max().item() < 2e-3
    top2 = ref.topk(2, dim=1)
    safe = (top2[0][:, 0] - top2[0][:, 1]) > 1e-3
    assert torch.equal(cand[safe].long(), top2[1][:, 0][safe])
    assert int(cand[5]) == 11 and int(cand[M - 1]) == N - 2        # exact ties: the lower column
    monkeypatch.setenv("CARE_V32_MIN_ROWS", str(1 << 30))
    # (read once per process: the switch is compiled as a static; compare through the ablation env instead)
    assert (best - ref.max(1).values.float()).abs().max().item() < 2e-3


@pytest.mark.parametrize("d", [512, 768, 1024, 2048])
@pytest.mark.parametrize("with_res", [False, True])
def test_add_ln_fixed_length_kernel(d, with_res):
    """care_add_ln takes a kernel specialised by row length for d in {512, 768, 1024, 2048} without a position table or
    slabs (every load issued before the first use); a table of zeros sends the same rows through the general kernel:
    same arithmetic in the same order up to the compiler's fma contraction - the fp32 rows agree to the last bit or
    two; ragged row count, nothing written past the rows."""
    rows, grp = 37 * 4 + 3, 7
    x, res = _rand(rows, d, seed=11, scale=3.0), _rand(rows, d, seed=12)
    g, b = _rand(d, seed=13) + 1.0, _rand(d, seed=14)
    zeros = torch.zeros(grp, d, device=DEV)
    outs = []
    for pos in (None, zeros):
        out = torch.full((rows + grp, d), float("nan"), device=DEV)
        outb = torch.zeros(rows + grp, d, device=DEV, dtype=torch.bfloat16)
        _call("care_add_ln", _p(x), d, _p(res) if with_res else None, d, _p(pos) if pos is not None else None, _p(g), _p(b), 1e-12,
              _p(out), _p(outb), d, rows, d, grp, grp, 0, 1, 0)
        outs.append((out, outb))
    torch.cuda.synchronize()
    assert (outs[0][0][:rows] - outs[1][0][:rows]).abs().max().item() <= 3e-6
    assert (outs[0][1][:rows].float() - outs[1][1][:rows].float()).abs().max().item() <= 4e-2  # a bf16 rounding boundary at most
    ref = torch.nn.functional.layer_norm((x + res) if with_res else x, (d,), g, b, 1e-12)
    assert (outs[0][0][:rows] - ref).abs().max().item() < 2e-5
    assert bool(torch.isnan(outs[0][0][rows:]).all())  # nothing written past the rows


def test_add_ln_group_mean_embed():
    rows, d, grp = 56, 512, 28
    x, res = _rand(rows, d, seed=9), _rand(rows, d, seed=10)
    g, b = _rand(d, seed=11), _rand(d, seed=12)
    out = torch.zeros(2, 100, d, device=DEV)
    outb = torch.zeros(2, 100, d, device=DEV, dtype=torch.bfloat16)
    _call("care_add_ln", _p(x), d, _p(res), d, None, _p(g), _p(b), 1e-12, _p(out), _p(outb), d, rows, d, grp, 100, 30, 1, 0)
    ref = torch.nn.functional.layer_norm(x + res, (d,), g, b, 1e-12).view(2, grp, d)
    torch.cuda.synchronize()
    assert (out[:, 30:58] - ref).abs().max().item() < 1e-5
    assert out[:, :30].abs().max().item() == 0 and out[:, 58:].abs().max().item() == 0
    assert torch.equal(outb, out.to(torch.bfloat16))  # the mirror is the fp32 result rounded once
    means = torch.zeros(2, 3 * d, device=DEV)
    _call("care_group_mean", _p(out), d, 100, 30, grp, _p(means), 3 * d, d, 2, d)
    torch.cuda.synchronize()
    assert (means[:, d:2 * d] - ref.mean(1)).abs().max().item() < 1e-6
    # embedding: tokens [3, 29], teacher-forced layout, with a global semantic vector
    V, T = 1000, 29
    word, pos, sem = _rand(V, d, seed=13), _rand(30, d, seed=14), _rand(3, d, seed=15)
    tok = torch.randint(0, V, (3, T), dtype=torch.int32).to(DEV)
    o2 = torch.zeros(3 * T, d, device=DEV)
    _call("care_embed_ln", _p(tok), T, 0, None, 0, _p(word), _p(pos), 0, _p(sem), T, _p(g), _p(b), 1e-12, _p(o2), None,
          d, 3 * T, T, d)
    ref2 = torch.nn.functional.layer_norm(word[tok.long()] + pos[:T].unsqueeze(0) + sem.unsqueeze(1), (d,), g, b, 1e-12)
    torch.cuda.synchronize()
    assert (o2.view(3, T, d) - ref2).abs().max().item() < 1e-5


@pytest.mark.parametrize("kv_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ctx_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nkeys,causal", [(114, False), (84, False), (29, True), (7, False)])
def test_attention(kv_dtype, ctx_dtype, nkeys, causal):
    B, H, d, per = 3, 8, 512, (29 if causal else 5)
    rows = B * per
    q = _rand(rows, d, seed=16)
    kv = (_rand(B * nkeys, 2 * d, seed=17)).to(kv_dtype).contiguous()
    bias = _rand(H, nkeys, seed=18)
    tok = torch.randint(0, 4, (B, nkeys), dtype=torch.int32).to(DEV)
    tok[:, 0] = 2
    ctx = torch.zeros(rows, d, device=DEV, dtype=ctx_dtype)
    _call("care_attention", _p(q), d, _p(kv), _p(kv[:, d:]), 1 if kv_dtype == torch.bfloat16 else 0, nkeys * 2 * d,
          2 * d, per, None, 0, nkeys, 1 if causal else 0, per, 0, _p(tok), nkeys, 0, _p(bias), nkeys, _p(ctx), d,
          1 if ctx_dtype == torch.bfloat16 else 0, rows, H)
    kvf = kv.float().view(B, nkeys, 2, H, 64)
    k, v = kvf[:, :, 0].permute(0, 2, 1, 3), kvf[:, :, 1].permute(0, 2, 1, 3)        # [B,H,Lk,64]
    qh = q.view(B, per, H, 64).permute(0, 2, 1, 3)
    s = qh @ k.transpose(-1, -2) / 8.0
    mask = tok.eq(0).view(B, 1, 1, nkeys).expand(B, H, per, nkeys).clone()
    if causal:
        mask |= torch.triu(torch.ones(per, nkeys, dtype=torch.bool, device=DEV), 1).view(1, 1, per, nkeys)
    s = s.masked_fill(mask, -1e9) + bias.view(1, H, 1, nkeys)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(rows, d)
    torch.cuda.synchronize()
    assert (ctx.float() - ref).abs().max().item() < (2e-5 if ctx_dtype == torch.float32 else 2e-2)


def test_concept_kernels():
    B, k, kp, d, topk = 5, 500, 512, 512, 30
    scores = _rand(B, kp, seed=19, scale=1.5)
    preds = torch.full((B, kp), float("nan"), device=DEV)
    avg = torch.zeros(B, device=DEV)
    _call("care_concept_finish", _p(scores), kp, _p(preds), kp, _p(avg), B, k)
    p = torch.sigmoid(scores[:, :k])
    ref = 1.0 - torch.exp(torch.log(torch.clamp(1.0 - p, 1e-12, 1)))
    torch.cuda.synchronize()
    assert (preds[:, :k] - ref).abs().max().item() < 1e-6
    assert preds[:, k:].abs().max().item() == 0
    assert (avg - p.mean(1)).abs().max().item() < 1e-6
    word, pos = _rand(k, d, seed=20), _rand(topk, d, seed=21)
    g, b = _rand(d, seed=22), _rand(d, seed=23)
    preds[1, 17] = preds[1, 3] = 0.999  # an exact tie: order must be index-ascending
    labels = torch.zeros(B, topk, device=DEV, dtype=torch.int64)
    mem = torch.zeros(B, 114, d, device=DEV)
    _call("care_concept_topk_embed", _p(preds), kp, k, topk, _p(word), _p(pos), _p(g), _p(b), 1e-12, _p(labels),
          _p(mem), None, d, 114, 84, B, d)
    torch.cuda.synchronize()
    pr = preds[:, :k].cpu().numpy()
    for i in range(B):
        order = sorted(range(k), key=lambda j: (-pr[i, j], j))[:topk]
        assert labels[i].tolist() == order
    ref_e = torch.nn.functional.layer_norm(word[labels] + pos.unsqueeze(0), (d,), g, b, 1e-12)
    assert (mem[:, 84:] - ref_e).abs().max().item() < 1e-5
    assert mem[:, :84].abs().max().item() == 0


@pytest.mark.parametrize("waves", [1, 4])
@pytest.mark.parametrize("rows,V,ld,bm", [(7, 10547, 10547, 5), (7, 10547, 10560, 5), (130, 10547, 10560, 8),
                                          (5, 300, 320, 1), (9, 17, 20, 5), (3, 5, 8, 5), (4, 1029, 1032, 3),
                                          (6, 20004, 20008, 5)])
def test_beam_select(rows, V, ld, bm, waves):
    """The kernels behind care_beam_select: a wave per row, four waves per row (few rows: 16-byte aligned rows both)
    and the block-per-row fallback (ld % 4 != 0): top-bm (value desc, index asc) as log-probabilities."""
    buf = torch.full((rows, ld), float("nan"), device=DEV)
    logits = buf[:, :V]
    logits.copy_(_rand(rows, V, seed=24 + V, scale=2.0))
    if V > 5000:
        logits[2, 100] = logits[2, 5000] = logits[2].max() + 1.0  # tie for the top spot, far apart
        logits[3, 64] = logits[3, 65] = logits[3, 4 * 64 + 1] = logits[3].max() + 0.5  # ties within / across lanes
    if V >= 17:
        logits[1, 3] = float("-inf")
    cv = torch.zeros(rows, bm, device=DEV)
    ci = torch.zeros(rows, bm, device=DEV, dtype=torch.int32)
    _call("care_beam_select", _p(buf), ld, V, bm, _p(cv), _p(ci), rows, waves)
    lp = torch.log_softmax(logits.double(), dim=1)
    torch.cuda.synchronize()
    x = logits.cpu().numpy()
    for r in range(rows):
        order = sorted(range(V), key=lambda j: (-x[r, j], j))[:bm]
        assert ci[r].tolist() == order
    assert (cv.double() - lp.gather(1, ci.long())).abs().max().item() < 1e-5


def test_rejected_arguments_raise():
    from care_amd import _lib

    A = _rand(4, 48)
    W = _rand(8, 48)
    out = torch.zeros(4, 8, device=DEV)
    with pytest.raises(_lib.CareHipError, match="ESHAPE"):
        _call("care_gemm", _p(A), 48, _p(W), 0, None, _p(out), 8, 0, None, 0, 0, 8, 4, 8, 48, 0)
    with pytest.raises(_lib.CareHipError, match="EINVAL"):
        _call("care_gemm", None, 48, _p(W), 0, None, _p(out), 8, 0, None, 0, 0, 8, 4, 8, 64, 0)


@pytest.mark.parametrize("M", [5, 300, 1024])
def test_gemm_splitk_slabs_summed_by_add_ln(M):
    N, K = 512, 2048
    A = _rand(M, K, seed=30).to(torch.bfloat16)
    W = _rand(N, K, seed=31, scale=0.03)
    Wb = W.to(torch.bfloat16).contiguous()
    bias, res = _rand(N, seed=32), _rand(M, N, seed=33)
    g, b = _rand(N, seed=34), _rand(N, seed=35)
    slabs = torch.full((K // 512, M, N), float("nan"), device=DEV)
    _call("care_gemm_bf16_splitk", _p(A), K, 1, _p(Wb), _p(bias), _p(slabs), N, slabs.stride(0), M, N, K)
    out = torch.zeros(M, N, device=DEV)
    _call("care_add_ln", _p(slabs), N, _p(res), N, None, _p(g), _p(b), 1e-12, _p(out), None, N, M, N, M, M, 0,
          K // 512, slabs.stride(0))
    y = A.float().double() @ _bf(W).double().t() + bias.double()
    ref = torch.nn.functional.layer_norm(y.float() + res, (N,), g, b, 1e-12)
    torch.cuda.synchronize()
    assert (slabs.sum(0).double() - y).abs().max().item() < 3e-3
    assert (out - ref).abs().max().item() < 3e-3


@pytest.mark.parametrize("M", [3, 200, 1030])
def test_fused_label_scoring(M):
    from care_amd import _lib

    N, K = 10547, 512
    A = _rand(M, K, seed=40).to(torch.bfloat16)
    W = (_rand(N, K, seed=41, scale=0.05)).to(torch.bfloat16).contiguous()
    labels = torch.randint(0, N, (M,), dtype=torch.int32).to(DEV)
    labels[0] = N - 1
    parts = _lib.argmax_parts(N, M, True)
    pm, ps, pl = (torch.empty(M, parts, device=DEV) for _ in range(3))
    pi = torch.empty(M, parts, device=DEV, dtype=torch.int32)
    _call("care_gemm_argmax_bf16", _p(A), K, 1, _p(W), _p(pm), _p(pi), _p(ps), _p(labels), _p(pl), M, N, K)
    logp, pred = torch.empty(M, device=DEV), torch.empty(M, device=DEV, dtype=torch.int32)
    _call("care_score_partials", _p(pm), _p(pi), _p(ps), _p(pl), parts, _p(logp), _p(pred), M)
    logits = (A.float() @ W.float().t()).contiguous()
    ref = torch.log_softmax(logits.double(), dim=1).gather(1, labels.long().unsqueeze(1)).squeeze(1)
    logp2, pred2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV, dtype=torch.int32)
    _call("care_score_logits", _p(logits), N, N, _p(labels), _p(logp2), _p(pred2), M)
    torch.cuda.synchronize()
    assert (logp.double() - ref).abs().max().item() < 2e-3
    assert (logp2.double() - ref).abs().max().item() < 1e-4
    top2 = logits.topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(pred[safe].long(), logits.argmax(1)[safe]) and torch.equal(pred2.long(), logits.argmax(1))


@pytest.mark.parametrize("form", ["auto", "128-row blocks", "packed W", "packed W, 128-row blocks"])
@pytest.mark.parametrize("a_f32", [True, False])
@pytest.mark.parametrize("M,K", [(7, 128), (64, 512), (200, 2048), (28 * 300, 512), (128 * 300 + 5, 128), (40000, 2048),
                                 (28 * 1171, 32), (1000, 96), (129, 64)])
def test_gemm_ln_fused(a_f32, M, K, form, monkeypatch):
    """form: the library picks 64- or 128-row blocks by launch rounds; CARE_LN_RG=2 forces the 128-row
    kernel (wave-specialised loaders, LayerNorm in the accumulator registers) onto every shape, ragged
    last blocks and K shorter than its ring depth included."""
    if "128-row" in form:
        monkeypatch.setenv("CARE_LN_RG", "2")
    packed = "packed" in form
    if packed and K % (64 if a_f32 else 128):
        pytest.skip("the packed-weight kernels move A in 256-byte row pieces")
    d, grp = 512, 28 if M % 28 == 0 else M
    A = _rand(M, K, seed=50)
    W = _rand(d, K, seed=51, scale=1 / math.sqrt(K))
    bias, g, b = _rand(d, seed=52), _rand(d, seed=53), _rand(d, seed=54)
    res = _rand(M, d, seed=55)
    pos = _rand(grp, d, seed=56) if (grp == 28 and not packed) else None
    Ain = A if a_f32 else A.to(torch.bfloat16).contiguous()
    Wb = W.to(torch.bfloat16).contiguous()
    ngrp = M // grp
    out = torch.zeros(ngrp, grp + 9, d, device=DEV)
    outb = torch.zeros(ngrp, grp + 9, d, device=DEV, dtype=torch.bfloat16)
    if packed:  # same arithmetic on the re-laid-out weight: bit-identical to the plain-weight call
        Wp = torch.empty_like(Wb)
        _call("care_pack_ln_weight", _p(Wb), _p(Wp), d, K)
        _call("care_gemm_ln_packed", _p(Ain), K, 0 if a_f32 else 1, _p(Wp), _p(bias), _p(res), d, _p(g), _p(b), 1e-12,
              _p(out), _p(outb), d, M, d, K, grp, grp + 9, 4)
        out2 = torch.zeros_like(out)
        _call("care_gemm_ln", _p(Ain), K, 0 if a_f32 else 1, _p(Wb), _p(bias), _p(res), d, None, _p(g), _p(b), 1e-12,
              _p(out2), None, d, M, d, K, grp, grp + 9, 4)
        assert torch.equal(out, out2)
    else:
        _call("care_gemm_ln", _p(Ain), K, 0 if a_f32 else 1, _p(Wb), _p(bias), _p(res), d, _p(pos), _p(g), _p(b), 1e-12,
              _p(out), _p(outb), d, M, d, K, grp, grp + 9, 4)
    y = (_bf(A).double() @ _bf(W).double().t()).float() + bias + res
    if pos is not None:
        y = (y.view(ngrp, grp, d) + pos.unsqueeze(0)).view(M, d)
    ref = torch.nn.functional.layer_norm(y, (d,), g, b, 1e-12).view(ngrp, grp, d)
    torch.cuda.synchronize()
    assert (out[:, 4:4 + grp] - ref).abs().max().item() < 4e-3
    assert out[:, :4].abs().max().item() == 0 and out[:, 4 + grp:].abs().max().item() == 0
    assert torch.equal(outb, out.to(torch.bfloat16))


@pytest.mark.parametrize("variant", ["", "f16"])
@pytest.mark.parametrize("out32", [False, True])
@pytest.mark.parametrize("M,K", [(128 * 8, 2048), (128 * 21, 512), (128 * 192, 128), (28 * 1024, 512), (28 * 1024, 2048), (128 * 300, 128),
                                 (128 * 300, 512), (128 * 513, 128), (128 * 777, 256),
                                 # ragged row counts: whole multiples of lcm(128, frames) rows on version 3, the rest on version 2
                                 (28 * 2990, 512), (28 * 333, 128), (128 * 40 + 77, 256)])
def test_gemm_ln_loader_waves(M, K, out32, variant):
    """The embedder's form at batch sizes (raw fp32 features, no residual, packed weight, whole 128-row blocks, >= 192 of
    them): version 3 of the fused kernel - loader waves, persistent workgroups, csrc/gemm_ln.hip - against the plain-weight
    entry, which stays on version 2: the same bits (a row's arithmetic does not depend on the launch rule), in both
    libraries, with the grouped output rows of the encoder (28 frames into a memory of 84 rows) and workgroups that own one,
    two and three blocks."""
    from care_amd import _lib

    h16 = torch.float16 if variant else torch.bfloat16
    call = lambda name, *a: _lib.call(name, *a, variant=variant)
    d, grp = 512, 28 if M % 28 == 0 else M
    A = _rand(M, K, seed=150)
    W = _rand(d, K, seed=151, scale=1 / math.sqrt(K)).to(h16).contiguous()
    bias, g, b = (None if K == 256 else _rand(d, seed=152)), _rand(d, seed=153), _rand(d, seed=154)   # (one shape without a bias)
    ngrp = M // grp
    Wp = torch.empty_like(W)
    call("care_pack_ln_weight", _p(W), _p(Wp), d, K)
    outs = []
    for packed in (True, False):
        out = torch.zeros(ngrp, grp + 56, d, device=DEV) if out32 else None
        outb = torch.zeros(ngrp, grp + 56, d, device=DEV, dtype=h16)
        if packed:
            call("care_gemm_ln_packed", _p(A), K, 0, _p(Wp), _p(bias), None, d, _p(g), _p(b), 1e-12, _p(out), _p(outb), d, M, d, K, grp, grp + 56, 28)
        else:
            call("care_gemm_ln", _p(A), K, 0, _p(W), _p(bias), None, d, None, _p(g), _p(b), 1e-12, _p(out), _p(outb), d, M, d, K, grp, grp + 56, 28)
        outs.append((out, outb))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))
    if out32:
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[0][0].to(h16))
    rows = slice(0, 4096)
    y = A[rows].to(h16).double() @ W.double().t() + (bias.double() if bias is not None else 0.0)
    ref = torch.nn.functional.layer_norm(y, (d,), g.double(), b.double(), 1e-12)
    got = outs[0][1][:, 28:28 + grp].reshape(M, d)[rows].double()
    assert (got - ref).abs().max().item() < (4e-3 if variant else 3.2e-2)
    assert outs[0][1][:, :28].abs().max().item() == 0 and outs[0][1][:, 28 + grp:].abs().max().item() == 0


@pytest.mark.parametrize("variant", ["", "f16"])
@pytest.mark.parametrize("out32", [False, True])
@pytest.mark.parametrize("M,K", [(128 * 8, 2048), (128 * 21, 512), (128 * 192, 128), (28 * 1024, 512), (28 * 1024, 2048), (128 * 300, 128),
                                 (128 * 513, 256), (28 * 2990, 512), (128 * 40 + 77, 128)])
def test_gemm_ln_split_loader_waves(M, K, out32, variant, monkeypatch):
    """The concept models' embedder (care_gemm_ln_split on whole 128-row blocks, >= 8 of them): version 3 -
    the hi / lo pieces of the features made once by the loader waves, two of the three weight images fetched - against
    version 2 (CARE_LN_V3=0, read per call): the same bits, and the split product's accuracy against float64."""
    from care_amd import _lib

    h16 = torch.float16 if variant else torch.bfloat16
    call = lambda name, *a: _lib.call(name, *a, variant=variant)
    d, grp = 512, 28 if M % 28 == 0 else M
    A = _rand(M, K, seed=160)
    W = _rand(d, K, seed=161, scale=1 / math.sqrt(K))
    bias, g, b = _rand(d, seed=162), _rand(d, seed=163), _rand(d, seed=164)
    ngrp = M // grp
    Ws = torch.empty(d, 3 * K, device=DEV, dtype=torch.float16)
    call("care_pack_ln_weight_split", _p(W), _p(Ws), d, K)
    outs = []
    for v3 in ("1", "0"):
        monkeypatch.setenv("CARE_LN_V3", v3)
        out = torch.zeros(ngrp, grp + 56, d, device=DEV) if out32 else None
        outb = torch.zeros(ngrp, grp + 56, d, device=DEV, dtype=h16)
        call("care_gemm_ln_split", _p(A), K, _p(Ws), _p(bias), _p(g), _p(b), 1e-12, _p(out), _p(outb), d, M, d, K, grp, grp + 56, 28)
        outs.append((out, outb))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))
    assert outs[0][1][:, :28].abs().max().item() == 0 and outs[0][1][:, 28 + grp:].abs().max().item() == 0
    if out32:
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[0][0].to(h16))
        rows = slice(0, 4096)
        ref = torch.nn.functional.layer_norm(A[rows].double() @ W.double().t() + bias.double(), (d,), g.double(), b.double(), 1e-12)
        assert (outs[0][0][:, 28:28 + grp].reshape(M, d)[rows].double() - ref).abs().max().item() < 2e-5 * math.sqrt(K / 64)


@pytest.mark.parametrize("split", [False, True])
def test_gemm_ln_loader_waves_random_shapes(split, monkeypatch):
    """Seeded sweep over block counts (one, two, three blocks per workgroup and uneven shares), K, frame groups, output
    offsets and the optional fp32 output: version 3 (both forms) against version 2, bit for bit, untouched rows left at zero."""
    from care_amd import _lib

    rng = np.random.RandomState(1234 + int(split))
    d = 512
    for trial in range(14):
        blocks = int(rng.choice([8, 9, 31, 64, 255, 256, 257, 300, 511, 513, 600, 777]))
        K = int(rng.choice([128, 256, 384, 512, 1024, 2048] if blocks < 400 else [128, 256, 384]))
        M = 128 * blocks
        grp = int(rng.choice([g for g in (M, 16, 32, 64, 28) if M % g == 0]))
        pad, off = int(rng.randint(0, 40)), 0
        off = int(rng.randint(0, pad + 1))
        out32 = bool(rng.randint(0, 2))
        A = _rand(M, K, seed=200 + trial)
        W = _rand(d, K, seed=300 + trial, scale=1 / math.sqrt(K))
        bias, g, b = _rand(d, seed=400 + trial), _rand(d, seed=500 + trial), _rand(d, seed=600 + trial)
        ngrp = M // grp
        if split:
            Wx = torch.empty(d, 3 * K, device=DEV, dtype=torch.float16)
            _lib.call("care_pack_ln_weight_split", _p(W), _p(Wx), d, K)
        else:
            Wb = W.to(torch.bfloat16).contiguous()
            Wx = torch.empty_like(Wb)
            _lib.call("care_pack_ln_weight", _p(Wb), _p(Wx), d, K)
        outs = []
        for v3 in ("1", "0"):
            monkeypatch.setenv("CARE_LN_V3", v3)
            out = torch.zeros(ngrp, grp + pad, d, device=DEV) if out32 else None
            outb = torch.zeros(ngrp, grp + pad, d, device=DEV, dtype=torch.bfloat16)
            if split:
                _lib.call("care_gemm_ln_split", _p(A), K, _p(Wx), _p(bias), _p(g), _p(b), 1e-12, _p(out), _p(outb), d, M, d, K, grp, grp + pad, off)
            else:
                _lib.call("care_gemm_ln_packed", _p(A), K, 0, _p(Wx), _p(bias), None, d, _p(g), _p(b), 1e-12, _p(out), _p(outb), d, M, d, K, grp,
                          grp + pad, off)
            outs.append((out, outb))
        torch.cuda.synchronize()
        what = (trial, blocks, K, grp, pad, off, out32)
        assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16)), what
        assert outs[0][1].abs().max().item() > 0.1, what
        if out32:
            assert torch.equal(outs[0][0], outs[1][0]), what
        if pad:
            assert outs[0][1][:, :off].abs().max().item() == 0 if off else True, what
            assert outs[0][1][:, off + grp:].abs().max().item() == 0 if off < pad else True, what
        del A, outs


@pytest.mark.parametrize("M,N,K", [(70, 1024, 64), (28 * 41, 1024, 2048), (4500, 768, 512), (28 * 1200, 1024, 128)])
def test_gemm_split3_products(M, N, K):
    """care_gemm_split3: the generic GEMM with fp32 operands as fp16 hi/lo pieces (one product over 3K) - against the
    same three products in float64 and against the exact product (what it stands in for)."""
    A = _rand(M, K, seed=70)
    W = _rand(N, K, seed=71, scale=1 / math.sqrt(K))
    bias = _rand(N, seed=72)
    W3 = torch.empty(N, 3 * K, device=DEV, dtype=torch.float16)
    C = torch.full((M, N + 8), float("nan"), device=DEV)
    _call("care_split3_weight", _p(W), _p(W3), N, K)
    _call("care_gemm_split3", _p(A), K, _p(W3), _p(bias), _p(C), N + 8, M, N, K)
    torch.cuda.synchronize()
    hi = lambda x: x.to(torch.float16).float()
    lo = lambda x: (x - hi(x)).to(torch.float16).float()
    assert torch.equal(W3[:, :K].float(), hi(W)) and torch.equal(W3[:, K:2 * K].float(), lo(W)) and torch.equal(W3[:, 2 * K:], W3[:, :K])
    y3 = hi(A).double() @ hi(W).double().t() + hi(A).double() @ lo(W).double().t() + lo(A).double() @ hi(W).double().t() + bias
    got = C[:, :N]
    assert (got - y3).abs().max().item() < 3e-6 * math.sqrt(K / 64)
    assert (got - (A.double() @ W.double().t() + bias)).abs().max().item() < 4e-6 * math.sqrt(K / 64)
    assert torch.isnan(C[:, N:]).all()


@pytest.mark.parametrize("M,K", [(56, 64), (28 * 37, 2048), (28 * 9 + 5, 4096), (300, 1024), (28 * 400, 2048)])
def test_gemm_ln_split_products(M, K):
    """care_gemm_ln_split: fp32 operands as hi/lo FP16 pieces, three MFMA passes.  Against the SAME three
    products in float64 (what the kernel computes, up to fp32 accumulation order) and against the plain
    fp32 Linear -> LayerNorm (what it stands in for: the dropped lo x lo term is ~2^-22 of a product)."""
    d, grp = 512, 28 if M % 28 == 0 else M
    A = _rand(M, K, seed=60)
    W = _rand(d, K, seed=61, scale=1 / math.sqrt(K))
    bias, g, b = _rand(d, seed=62), _rand(d, seed=63), _rand(d, seed=64)
    ngrp = M // grp
    out = torch.zeros(ngrp, grp + 3, d, device=DEV)
    outb = torch.zeros(ngrp, grp + 3, d, device=DEV, dtype=torch.bfloat16)
    Ws = torch.empty(3 * K * d, device=DEV, dtype=torch.bfloat16)
    _call("care_pack_ln_weight_split", _p(W), _p(Ws), d, K)
    _call("care_gemm_ln_split", _p(A), K, _p(Ws), _p(bias), _p(g), _p(b), 1e-12, _p(out), _p(outb), d, M, d, K, grp, grp + 3, 2)
    hi = lambda x: x.to(torch.float16).float()
    lo = lambda x: (x - hi(x)).to(torch.float16).float()
    y3 = (hi(A).double() @ hi(W).double().t() + hi(A).double() @ lo(W).double().t() + lo(A).double() @ hi(W).double().t())
    ln = lambda y: torch.nn.functional.layer_norm(y.float() + bias, (d,), g, b, 1e-12).view(ngrp, grp, d)
    torch.cuda.synchronize()
    got = out[:, 2:2 + grp]
    assert (got - ln(y3)).abs().max().item() < 3e-6 * math.sqrt(K / 64)  # fp32 accumulation only
    err = (got - ln(A.double() @ W.double().t())).abs()  # fp32 accumulation order + the 2^-22 term, gamma up to ~4
    assert err.mean().item() < 2e-6 and err.max().item() < 3e-5
    assert out[:, :2].abs().max().item() == 0 and out[:, 2 + grp:].abs().max().item() == 0
    assert torch.equal(outb, out.to(torch.bfloat16))
    # a bf16 weight image, a residual or an unpacked K are refused, not silently mishandled
    from care_amd._lib import CareHipError
    with pytest.raises(CareHipError):
        _call("care_gemm_ln_split", _p(A), K + 32, _p(Ws), _p(bias), _p(g), _p(b), 1e-12, _p(out), None, d, M, d, K + 32, grp, grp + 3, 2)


# --------------------------------------------------------------------------------------------
# absorbed cross-attention (attention_latent.hip, heads.hip)
@pytest.mark.parametrize("rows,heads", [(1, 8), (7, 8), (16, 8), (250, 8), (1029, 8), (33, 4)])
def test_head_expand_and_reduce(rows, heads):
    d = heads * 64 if heads == 8 else 512
    q = _rand(rows, 512, seed=1).to(torch.bfloat16)
    wkt = _rand(heads, 512, 64, seed=2, scale=0.05).to(torch.bfloat16)
    qt = torch.full((rows, heads, 512), float("nan"), device=DEV, dtype=torch.bfloat16)
    _call("care_head_expand", _p(q), 512, _p(wkt), _p(qt), heads * 512, rows, heads)
    ref = torch.einsum("rhe,hce->rhc", q.float()[:, : heads * 64].reshape(rows, heads, 64), wkt.float())
    assert torch.isfinite(qt.float()).all()
    assert (qt.float() - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())

    ct = _rand(rows, heads, 512, seed=3).to(torch.bfloat16)
    wv = _rand(512, 512, seed=4, scale=0.05).to(torch.bfloat16)
    bv = _rand(512, seed=5)
    for bias in (bv, None):
        ctx = torch.full((rows, 512), float("nan"), device=DEV, dtype=torch.bfloat16)
        _call("care_head_reduce", _p(ct), heads * 512, _p(wv), _p(bias), _p(ctx), 512, rows, heads)
        ref = torch.einsum("rhc,hec->rhe", ct.float(), wv.float()[: heads * 64].reshape(heads, 64, 512)).reshape(rows, -1)
        if bias is not None:
            ref = ref + bias[: heads * 64]
        got = ctx.float()[:, : heads * 64]
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < 8e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("rows,nkeys,rows_per_kv,use_bias", [(1, 84, 1, True), (5, 7, 1, False), (37, 16, 1, True),
                                                             (130, 114, 1, True), (100, 84, 5, True),
                                                             (1030, 128, 1, False), (64, 33, 2, True),
                                                             (35, 57, 5, True), (9, 20, 3, False), (7, 20, 2, True),
                                                             (5 * 1100, 84, 5, False), (12, 40, 3, "H4")])
def test_attention_latent(rows, nkeys, rows_per_kv, use_bias):
    """ct[r][h] = softmax_j(qt[r][h] . mem[clip][j] + bias[h][j]) . mem[clip] against torch on the same
    bf16 operands (the kernel rounds the probabilities and the output to bf16).  Rows that share a clip
    (rows_per_kv > 1, whole clips) are taken two per wave - pairs, the odd row out, clip edges."""
    H, d = (4 if use_bias == "H4" else 8), 512
    clips = (rows + rows_per_kv - 1) // rows_per_kv
    mem = _rand(clips, nkeys, d, seed=11).to(torch.bfloat16)
    qt = _rand(rows, H, d, seed=12, scale=0.12).to(torch.bfloat16)
    bias = _rand(H, nkeys, seed=13, scale=0.7) if use_bias else None
    ct = torch.full((rows, H, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    _call("care_attention_latent", _p(qt), H * d, _p(mem), nkeys * d, d, rows_per_kv, nkeys, _p(bias), nkeys, _p(ct),
          H * d, rows, H, d)
    clip_of = torch.arange(rows, device=DEV) // rows_per_kv
    m = mem.float()[clip_of]
    s = torch.einsum("rhc,rjc->rhj", qt.float(), m)
    if use_bias:
        s = s + bias[None]
    ref = torch.einsum("rhj,rjc->rhc", torch.softmax(s, -1), m)
    assert torch.isfinite(ct.float()).all()
    err = (ct.float() - ref).abs()
    assert err.max().item() < 1.2e-2 * max(1.0, ref.abs().max().item()) and err.mean().item() < 2e-3


def test_attention_latent_rejects_bad_arguments():
    from care_amd import _lib

    H, d = 8, 512
    mem = _rand(2, 20, d).to(torch.bfloat16)
    qt = _rand(2, H, d).to(torch.bfloat16)
    ct = torch.empty(2, H, d, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.CareHipError):  # d_model other than 512 / 768 / 1024
        _call("care_attention_latent", _p(qt), H * d, _p(mem), 20 * d, d, 1, 20, None, 0, _p(ct), H * d, 2, H, 640)
    with pytest.raises(_lib.CareHipError):  # more than 128 keys
        _call("care_attention_latent", _p(qt), H * d, _p(mem), 20 * d, d, 1, 129, None, 0, _p(ct), H * d, 2, H, d)
    with pytest.raises(_lib.CareHipError):  # misaligned query
        _call("care_attention_latent", _p(qt) + 2, H * d, _p(mem), 20 * d, d, 1, 20, None, 0, _p(ct), H * d, 2, H, d)


@pytest.mark.parametrize("M,K,a_f32", [(200, 2048, True), (28 * 300, 512, True), (1030, 512, False)])
def test_gemm_ln_bf16_only_output(M, K, a_f32):
    """care_gemm_ln with out = NULL: only the bf16 mirror is written, with the same values."""
    A = _rand(M, K, seed=31)
    Ain = A if a_f32 else A.to(torch.bfloat16)
    W = _rand(512, K, seed=32, scale=0.05).to(torch.bfloat16)
    bias, g, b = _rand(512, seed=33), _rand(512, seed=34), _rand(512, seed=35)
    out = torch.empty(M, 512, device=DEV)
    ob1 = torch.empty(M, 512, device=DEV, dtype=torch.bfloat16)
    ob2 = torch.full((M, 512), float("nan"), device=DEV, dtype=torch.bfloat16)
    args = (_p(Ain), K, 0 if a_f32 else 1, _p(W), _p(bias), None, 0, None, _p(g), _p(b), 1e-12)
    _call("care_gemm_ln", *args, _p(out), _p(ob1), 512, M, 512, K, M, M, 0)
    _call("care_gemm_ln", *args, None, _p(ob2), 512, M, 512, K, M, M, 0)
    assert torch.equal(ob1, ob2)
    from care_amd import _lib
    with pytest.raises(_lib.CareHipError):
        _call("care_gemm_ln", *args, None, None, 512, M, 512, K, M, M, 0)


def test_beam_select_massive_ties_take_the_overflow_path():
    """All-equal rows (every logit reaches the threshold: the per-wave candidate list overflows and
    the chunk is re-scanned from memory), rows with a plateau of ties around the k-th best, and a row
    that is -inf except for three entries: the order is still (value desc, index asc)."""
    rows, V, ld, bm = 6, 10547, 10560, 5
    buf = torch.zeros(rows, ld, device=DEV)
    logits = buf[:, :V]
    logits[1] = -3.25
    logits[2] = _rand(V, seed=77)
    logits[2, 2000:2600] = logits[2].max() + 1.0       # a 600-wide plateau at the top
    logits[3] = _rand(V, seed=78)
    logits[3, 5:9000:3] = 9.0                          # ~3000 ties spread over all chunks
    logits[4] = float("-inf")
    logits[4, [7, 4100, 10546]] = torch.tensor([1.0, 2.0, 1.0], device=DEV)
    logits[5] = _rand(V, seed=79)
    cv = torch.zeros(rows, bm, device=DEV)
    ci = torch.zeros(rows, bm, device=DEV, dtype=torch.int32)
    _call("care_beam_select", _p(buf), ld, V, bm, _p(cv), _p(ci), rows, 1)
    torch.cuda.synchronize()
    x = logits.cpu().numpy()
    lp = torch.log_softmax(logits.double(), dim=1)
    for r in range(rows):
        order = sorted(range(V), key=lambda j: (-x[r, j], j))[:bm]
        got, ref = cv[r].double().cpu(), lp[r, order].cpu()
        fin = torch.isfinite(ref)
        nf = int(fin.sum())
        # -inf is never a candidate: with fewer than bm finite logits the remaining slots are (-inf, 0)
        assert ci[r].tolist() == order[:nf] + [0] * (bm - nf), (r, ci[r].tolist(), order)
        assert torch.equal(torch.isfinite(got), fin)
        assert (got[fin] - ref[fin]).abs().max().item() < 1e-5


@pytest.mark.parametrize("rows,parts,d,with_sem", [(5, 7, 512, True), (130, 166, 512, False), (33, 512, 1024, True)])
def test_greedy_update_embed_equals_the_two_kernels(rows, parts, d, with_sem):
    """care_greedy_update_embed == care_greedy_update followed by the next step's care_embed_ln."""
    V, T, t = 300, 29, 4
    pmax = _rand(rows, parts, seed=41)
    psum = _rand(rows, parts, seed=42).abs() + 0.1
    pidx = torch.randint(0, V, (rows, parts), device=DEV, dtype=torch.int32)
    word, pos = _rand(V, d, seed=43), _rand(T + 1, d, seed=44)
    sem = _rand(rows, d, seed=45) if with_sem else None
    g, b = _rand(d, seed=46), _rand(d, seed=47)

    def state():
        fed = torch.zeros(rows, T + 1, device=DEV, dtype=torch.int32)
        score = _rand(rows, seed=48).clone()
        length = torch.zeros(rows, device=DEV, dtype=torch.int32)
        fin = (torch.arange(rows, device=DEV) % 3 == 0).to(torch.int32)
        return fed, score, length, fin

    f1, s1, l1, n1 = state()
    x1 = torch.empty(rows, d, device=DEV); x1b = torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)
    _call("care_greedy_update", _p(pmax), _p(pidx), _p(psum), parts, _p(f1), T + 1, _p(s1), _p(l1), _p(n1), t, T, 3, rows)
    _call("care_embed_ln", _p(f1), T + 1, t, None, 0, _p(word), _p(pos), t, _p(sem), 1, _p(g), _p(b), 1e-12, _p(x1),
          _p(x1b), d, rows, 1, d)
    f2, s2, l2, n2 = state()
    x2 = torch.empty(rows, d, device=DEV); x2b = torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)
    _call("care_greedy_update_embed", _p(pmax), _p(pidx), _p(psum), parts, _p(f2), T + 1, _p(s2), _p(l2), _p(n2), t, T,
          3, rows, _p(word), _p(pos), _p(sem), 1, _p(g), _p(b), 1e-12, _p(x2), _p(x2b), d, d)
    assert torch.equal(f1, f2) and torch.equal(s1, s2) and torch.equal(l1, l2) and torch.equal(n1, n2)
    assert torch.equal(x1, x2) and torch.equal(x1b, x2b)


@pytest.mark.parametrize("M,V,bm", [(5, 10547, 5), (300, 10547, 5), (130, 700, 8), (64, 10547, 1), (8192 + 40, 10547, 5),
                                    (12288, 3000, 8)])
def test_fused_beam_selection_equals_logits_plus_beam_select(M, V, bm):
    """statistics GEMM -> threshold -> candidate GEMM -> pick  ==  store-mode GEMM + care_beam_select,
    including rows with tied logits (duplicated vocabulary rows) and a plateau of > 64 ties at the
    top, which overflows the candidate list and takes the exact in-kernel recomputation.  From 8192 rows
    both GEMM passes run on the 256-row panels of csrc/gemm_vocab.hip (statistics / collect modes)."""
    from care_amd import _lib

    K = 512
    A = _rand(M, K, seed=51).to(torch.bfloat16)
    W = _rand(V, K, seed=52, scale=0.05).to(torch.bfloat16)
    W[17] = W[400]                       # an exact tie between two columns for every row
    if V > 5000:
        W[3000:3100] = W[2999]           # a 101-column plateau: ties everywhere it reaches the top
        A[1] = (W[2999].float() * 40).to(torch.bfloat16)   # row 1: the plateau IS the top -> overflow path
    ld = (V + 63) // 64 * 64
    logits = torch.empty(M, ld, device=DEV)
    _call("care_gemm_bf16", _p(A), K, 1, _p(W), None, _p(logits), ld, 0, None, 0, 0, V, M, V, K, 0)
    ref_v = torch.zeros(M, bm, device=DEV); ref_i = torch.zeros(M, bm, device=DEV, dtype=torch.int32)
    _call("care_beam_select", _p(logits), ld, V, bm, _p(ref_v), _p(ref_i), M, 1)

    parts = _lib.load().care_argmax_parts_bf16_min(M, V, K, 1, 8)
    assert parts >= 8
    pmax = torch.empty(M, parts, device=DEV); psum = torch.empty(M, parts, device=DEV)
    pidx = torch.empty(M, parts, device=DEV, dtype=torch.int32)
    thr = torch.empty(M, device=DEV); cnt = torch.full((M,), -1, device=DEV, dtype=torch.int32)
    cap = 64
    cval = torch.empty(M, cap, device=DEV); cidx = torch.empty(M, cap, device=DEV, dtype=torch.int32)
    got_v = torch.zeros(M, bm, device=DEV); got_i = torch.zeros(M, bm, device=DEV, dtype=torch.int32)
    _call("care_gemm_argmax_bf16_min", _p(A), K, 1, _p(W), _p(pmax), _p(pidx), _p(psum), M, V, K, 8)
    _call("care_beam_threshold", _p(pmax), parts, bm, _p(thr), _p(cnt), M)
    _call("care_gemm_collect_bf16", _p(A), K, 1, _p(W), _p(thr), _p(cnt), _p(cval), _p(cidx), cap, M, V, K)
    _call("care_beam_pick", _p(pmax), _p(psum), parts, _p(cnt), _p(cval), _p(cidx), cap, bm, _p(A), K, 1, _p(W), V, K,
          _p(got_v), _p(got_i), M)
    torch.cuda.synchronize()
    if _lib.load().care_beam_sparse_applies(M, V, K, 1):
        # the sparse second pass (csrc/beam_sparse.hip): the same statistics + the maxima of every (tile, row), then
        # only the (tile, row) products that can hold a candidate - the same candidate SETS, hence the same picks
        tiles = (V + 31) // 32
        tmx = torch.full((tiles, M), float("nan"), device=DEV)
        pm2, ps2, pi2 = torch.empty_like(pmax), torch.empty_like(psum), torch.empty_like(pidx)
        thr2 = torch.empty_like(thr); cnt2 = torch.full_like(cnt, -1)
        cval2, cidx2 = torch.empty_like(cval), torch.empty_like(cidx)
        tcount = torch.full((2 * tiles + 1,), -3, device=DEV, dtype=torch.int32)  # counts, then work-unit prefix sums
        tlist = torch.empty(tiles, M, device=DEV, dtype=torch.int32)
        sp_v, sp_i = torch.zeros_like(got_v), torch.zeros_like(got_i)
        _call("care_gemm_argmax_bf16_tiles", _p(A), K, 1, _p(W), _p(pm2), _p(pi2), _p(ps2), _p(tmx), M, V, K, 8)
        _call("care_beam_threshold", _p(pm2), parts, bm, _p(thr2), _p(cnt2), M)
        _call("care_beam_sparse_collect", _p(A), K, _p(W), _p(tmx), _p(thr2), _p(cnt2), _p(cval2), _p(cidx2), cap,
              _p(tcount), _p(tlist), M, V, K)
        _call("care_beam_pick", _p(pm2), _p(ps2), parts, _p(cnt2), _p(cval2), _p(cidx2), cap, bm, _p(A), K, 1, _p(W), V, K,
              _p(sp_v), _p(sp_i), M)
        torch.cuda.synchronize()
        assert torch.equal(pm2, pmax) and torch.equal(pi2, pidx) and torch.equal(ps2, psum)
        assert torch.equal(tmx.max(0).values, pmax.max(1).values)      # the map's row maxima are the rows' maxima
        assert torch.equal(cnt2, cnt)                                   # the same number of candidates per row ...
        assert torch.equal(sp_i, got_i) and torch.equal(sp_v, got_v)    # ... and the same picks, bit for bit
        assert int(tcount[:tiles].sum()) >= M and int(tcount[:tiles].max()) <= M
        assert int(tcount[2 * tiles]) == int(((tcount[:tiles] + 127) // 128).sum())
    assert int(cnt.min()) >= bm                      # at least bm candidates reach every threshold
    if V > 5000:
        assert int(cnt[1]) > cap                     # the plateau row did overflow
    over = cnt > cap
    assert torch.equal(got_i[~over], ref_i[~over])   # bit-identical logits -> identical order, ties included
    assert (got_v[~over] - ref_v[~over]).abs().max().item() < 2e-5
    # overflow rows: recomputed with an fp32 fma chain instead of the MFMA - same columns unless two
    # non-identical columns differ by rounding; the duplicated columns stay tied (column asc)
    if bool(over.any()):
        assert torch.equal(got_i[over], ref_i[over])
        assert (got_v[over] - ref_v[over]).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,V,K,bm", [(5, 10547, 512, 5), (300, 10547, 512, 5), (640, 10547, 512, 4), (130, 700, 512, 2), (64, 10547, 1024, 5),
                                      (2560, 10547, 512, 5), (777, 9000, 768, 3), (33, 128, 512, 5), (300, 10547, 512, 8), (64, 3000, 512, 6)])
def test_beam_selection_from_group_maxima_equals_logits_plus_beam_select(M, V, K, bm):
    """care_gemm_tile_beam -> care_beam_pick_groups  ==  the logits of the SAME tile kernel + care_beam_select: identical
    columns in identical order (the recomputed candidates are the tile kernel's bits), log-probabilities within 2e-5
    (the log-sum-exp is merged from 64-column parts); with exact ties between columns, a plateau of equal logits across
    groups and parts at the top of one row, and a ragged last part."""
    from care_amd import _lib

    A = _rand(M, K, seed=61).to(torch.bfloat16)
    W = _rand(V, K, seed=62, scale=0.05).to(torch.bfloat16)
    W[17] = W[40]                        # an exact tie between two columns of every row (same part, different groups)
    if V > 5000:
        W[3000:3100] = W[2999]           # a 101-column plateau over two parts ...
        A[1] = (W[2999].float() * 40).to(torch.bfloat16)   # ... which IS the top of row 1: the bm lowest columns win
    ld = (V + 63) // 64 * 64
    logits = torch.empty(M, ld, device=DEV)
    _call("care_gemm_tile", _p(A), K, _p(W), None, _p(logits), ld, 0, None, 0, 0, V, M, V, K, 0)
    ref_v = torch.zeros(M, bm, device=DEV); ref_i = torch.zeros(M, bm, device=DEV, dtype=torch.int32)
    _call("care_beam_select", _p(logits), ld, V, bm, _p(ref_v), _p(ref_i), M, 1)
    parts = _lib.load().care_argmax_parts_tile(V)
    pmax = torch.full((M, parts), float("nan"), device=DEV); psum = torch.full((M, parts), float("nan"), device=DEV)
    gmax = torch.full((M, parts, 16), float("nan"), device=DEV)
    got_v = torch.zeros(M, bm, device=DEV); got_i = torch.zeros(M, bm, device=DEV, dtype=torch.int32)
    _call("care_gemm_tile_beam", _p(A), K, _p(W), _p(pmax), _p(psum), _p(gmax), M, V, K)
    _call("care_beam_pick_groups", _p(pmax), _p(psum), _p(gmax), parts, bm, _p(A), K, _p(W), V, K, _p(got_v), _p(got_i), M)
    torch.cuda.synchronize()
    # the epilogue's statistics against the logits themselves
    lg = logits[:, :V]
    pad = torch.full((M, parts * 64 - V), float("-inf"), device=DEV)
    blocks = torch.cat([lg, pad], 1).view(M, parts, 16, 4)
    assert torch.equal(gmax, blocks.max(-1).values)
    assert torch.equal(pmax, blocks.view(M, parts, 64).max(-1).values)
    assert torch.equal(got_i, ref_i)     # bit-identical logits -> identical order, ties included
    assert (got_v - ref_v).abs().max().item() < 2e-5
    if V > 5000:
        assert got_i[1].tolist() == [2999, 3000, 3001, 3002, 3003, 3004, 3005, 3006][:bm]


TILE_SHAPES = [(1, 1024, 1024), (700, 520, 64), (300, 256, 128), (5, 10547, 1024), (64, 768, 768), (100, 3072, 1024), (257, 4096, 1024), (300, 1024, 4096),
               (130, 640, 640), (200, 768, 3072), (1000, 48, 512), (4096, 1024, 1024), (513, 10547, 768), (777, 2304, 768),
               (4096 + 19, 3072, 1024), (2048, 512, 2048)]


@pytest.mark.parametrize("M,N,K", TILE_SHAPES)
@pytest.mark.parametrize("act,out_bf16", [(0, False), (1, True), (2, False)])
@pytest.mark.parametrize("cfg", ["", "223", "42", "2222", "2422", "4412", "4414", "2224"])
def test_gemm_tile(M, N, K, act, out_bf16, cfg, monkeypatch):
    """csrc/gemm_tile.hip (bf16 A, bf16 W, any K % 64 == 0: the d_model 768 / 1024 layers) in every tile shape / ring
    depth, against torch on the same bf16 operands; ragged edges in M and N; nothing written outside the destination."""
    if cfg and (act == 2 or M < 64) and cfg != "42":
        pytest.skip("tile shapes covered on the other combinations")
    if cfg:
        monkeypatch.setenv("CARE_TILE_CFG", cfg)
    A = _rand(M, K, seed=71).to(torch.bfloat16).contiguous()
    W = _rand(N, K, seed=72, scale=1 / math.sqrt(K))
    bias = _rand(N, seed=73)
    Wb = W.to(torch.bfloat16).contiguous()
    ld = (N + 15) // 8 * 8
    out = torch.full((M + 2, ld), 7.0, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dst = out[:M, :N]
    _call("care_gemm_tile", _p(A), K, _p(Wb), _p(bias), _p(dst), dst.stride(0), 1 if out_bf16 else 0, None, 0, 0, N, M, N, K, act)
    ref = A.double() @ Wb.double().t() + bias.double()
    ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref) if act == 2 else ref)
    torch.cuda.synchronize()
    assert (dst.double() - ref).abs().max().item() < (4e-2 if out_bf16 else 3e-3)
    assert float(out[M:].float().min()) == 7.0 and float(out[:, N:].float().min()) == 7.0


@pytest.mark.parametrize("M,N,act,split", [(640, 1536, 0, True), (1280, 512, 0, False), (2560, 2048, 1, False), (4099, 512, 0, False),
                                           (12000, 1536, 0, True), (16000, 2048, 2, False)])
def test_tile_and_a_stationary_gemm_agree_bit_for_bit(M, N, act, split):
    """engine.gemm moves the K = 512 products of a decode step from the A-stationary kernel (csrc/gemm_as.hip) to the
    LDS-tiled one (csrc/gemm_tile.hip) between engine.MID_TILE_ROWS rows: both add K in ascending order into one
    accumulator per output, so every output - fp32 or rounded to bf16, one destination or q | K,V split - is the same
    bit pattern and the switch changes no caption at any batch size."""
    K = 512
    A = _rand(M, K, seed=81).to(torch.bfloat16).contiguous()
    Wb = _rand(N, K, seed=82, scale=1 / math.sqrt(K)).to(torch.bfloat16).contiguous()
    bias = _rand(N, seed=83)
    outs = []
    for fn in ("care_gemm_bf16", "care_gemm_tile"):
        if split:   # q fp32 | K, V bf16 (the QKV product writes the cache directly)
            o0 = torch.zeros(M, 512, device=DEV)
            o1 = torch.zeros(M, N - 512, device=DEV, dtype=torch.bfloat16)
            tail = (_p(bias), _p(o0), 512, 0, _p(o1), N - 512, 1, 512, M, N, K, act)
        else:
            o0 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16 if act else torch.float32)
            o1 = None
            tail = (_p(bias), _p(o0), N, 1 if act else 0, None, 0, 0, N, M, N, K, act)
        if fn == "care_gemm_bf16":
            _call(fn, _p(A), K, 1, _p(Wb), *tail)
        else:
            _call(fn, _p(A), K, _p(Wb), *tail)
        outs.append((o0, o1))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0])
    if split:
        assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].float().abs().max()) > 0.1


def test_gemm_tile_split_destinations_and_odd_leading_dimension():
    """QKV of a d_model = 1024 decode step: q fp32 and k | v bf16 straight into position 7 of the cache; and the
    vocabulary logits with an odd leading dimension (scalar store path)."""
    M, K, d = 300, 1024, 1024
    A = _rand(M, K, seed=74).to(torch.bfloat16).contiguous()
    W, bias = _rand(3 * d, K, seed=75, scale=0.03), _rand(3 * d, seed=76)
    Wb = W.to(torch.bfloat16).contiguous()
    q = torch.zeros(M, d, device=DEV)
    cache = torch.zeros(M, 29, 2 * d, device=DEV, dtype=torch.bfloat16)
    dst = cache[:, 7, :]
    _call("care_gemm_tile", _p(A), K, _p(Wb), _p(bias), _p(q), d, 0, _p(dst), dst.stride(0), 1, d, M, 3 * d, K, 0)
    ref = A.float() @ Wb.float().t() + bias
    torch.cuda.synchronize()
    assert (q - ref[:, :d]).abs().max().item() < 3e-3
    assert (cache[:, 7, :].float() - ref[:, d:]).abs().max().item() < 4e-2
    assert cache[:, 6, :].abs().max().item() == 0 and cache[:, 8, :].abs().max().item() == 0
    N = 10547
    Wv = _rand(N, K, seed=77, scale=0.03).to(torch.bfloat16).contiguous()
    logits = torch.full((M, N), float("nan"), device=DEV)
    _call("care_gemm_tile", _p(A), K, _p(Wv), None, _p(logits), N, 0, None, 0, 0, N, M, N, K, 0)
    torch.cuda.synchronize()
    assert (logits - A.float() @ Wv.float().t()).abs().max().item() < 3e-3


@pytest.mark.parametrize("M,K", [(1, 1024), (3, 768), (129, 1024), (1000, 1024), (4096 + 7, 1024), (300, 512)])
@pytest.mark.parametrize("cfg", ["", "42", "2422", "2223", "4414"])
def test_gemm_tile_argmax(M, K, cfg, monkeypatch):
    """The fused vocabulary arg-max of the LDS-tiled kernel: per 64-column group (max, lowest arg-max, sum exp) and the
    label logit, reduced by care_greedy_update / care_score_partials, against the bf16 product in fp64; exact ties
    between far-apart columns and inside the ragged last tile go to the lower column."""
    from care_amd import _lib

    if cfg:
        monkeypatch.setenv("CARE_TILE_CFG", cfg)
    N = 10547
    A = _rand(M, K, seed=81).to(torch.bfloat16).contiguous()
    W = _rand(N, K, seed=82, scale=0.05)
    W[N // 2 + 5] = W[11]
    W[N - 1] = W[N - 2]
    A[0] = (W[11] * 40).to(torch.bfloat16)
    A[M - 1] = (W[N - 2] * 40).to(torch.bfloat16)
    Wb = W.to(torch.bfloat16).contiguous()
    parts = _lib.load().care_argmax_parts_tile(N)
    ref = A.double() @ Wb.double().t()
    labels = torch.randint(0, N, (M,), device=DEV, dtype=torch.int32)
    pm, ps = torch.full((M, parts), float("nan"), device=DEV), torch.full((M, parts), float("nan"), device=DEV)
    pl = torch.full((M, parts), float("nan"), device=DEV)
    pi = torch.full((M, parts), -7, device=DEV, dtype=torch.int32)
    _call("care_gemm_tile_argmax", _p(A), K, _p(Wb), _p(pm), _p(pi), _p(ps), _p(labels), _p(pl), M, N, K)
    logp, pred = torch.empty(M, device=DEV), torch.empty(M, device=DEV, dtype=torch.int32)
    _call("care_score_partials", _p(pm), _p(pi), _p(ps), _p(pl), parts, _p(logp), _p(pred), M)
    pm2, ps2 = torch.full((M, parts), float("nan"), device=DEV), torch.full((M, parts), float("nan"), device=DEV)
    pi2 = torch.full((M, parts), -7, device=DEV, dtype=torch.int32)
    _call("care_gemm_tile_argmax", _p(A), K, _p(Wb), _p(pm2), _p(pi2), _p(ps2), None, None, M, N, K)
    torch.cuda.synchronize()
    assert torch.equal(pm, pm2) and torch.equal(pi, pi2) and torch.equal(ps, ps2)
    top2 = ref.topk(2, dim=1)
    safe = (top2[0][:, 0] - top2[0][:, 1]) > 1e-3
    assert torch.equal(pred[safe].long(), top2[1][:, 0][safe])
    if M > 1:
        assert int(pred[0]) == 11 and int(pred[M - 1]) == N - 2      # ties: the lower column
    want = torch.log_softmax(ref, dim=1).gather(1, labels.long().unsqueeze(1)).squeeze(1)
    assert (logp.double() - want).abs().max().item() < 2e-3


@pytest.mark.parametrize("M,N,K", [(300, 1024, 2048), (4096 + 33, 1024, 512), (28 * 64, 768, 128), (70000, 1024, 128)])
def test_gemm_tile_split3(M, N, K):
    """fp32-grade products on the LDS-tiled kernel (fp16 pieces, three MFMA passes as one product over 3K with the A
    columns wrapping): against the exact product in fp64, and against the generic kernel's care_gemm_split3."""
    A = _rand(M, K, seed=91)
    W = _rand(N, K, seed=92, scale=1 / math.sqrt(K))
    bias = _rand(N, seed=93)
    W3 = torch.empty(N, 3 * K, device=DEV, dtype=torch.float16)
    _call("care_split3_weight", _p(W), _p(W3), N, K)
    A2 = torch.empty(M, 2 * K, device=DEV, dtype=torch.float16)
    _call("care_split2_act", _p(A), K, _p(A2), M, K)
    out = torch.full((M, N), float("nan"), device=DEV)
    _call("care_gemm_tile_split3", _p(A2), _p(W3), _p(bias), _p(out), N, 0, None, 0, 0, N, M, N, K, 0)
    old = torch.full((M, N), float("nan"), device=DEV)
    _call("care_gemm_split3", _p(A), K, _p(W3), _p(bias), _p(old), N, M, N, K)
    ref = A.double() @ W.double().t() + bias.double()
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() < 2e-5
    assert (out - old).abs().max().item() < 1e-5


@pytest.mark.parametrize("nseq,seq,nkeys,per_kv,kind", [(7, 29, 29, 1, "self"), (5, 30, 30, 1, "self"), (3, 12, 12, 1, "self"),
                                                        (9, 29, 84, 3, "cross"), (4, 29, 114, 1, "cross_bias"),
                                                        (6, 17, 28, 2, "cross"), (2, 29, 128, 1, "cross_bias"),
                                                        (3000, 29, 114, 5, "cross_bias"), (2048, 29, 29, 1, "self")])
def test_attention_seq(nseq, seq, nkeys, per_kv, kind):
    """csrc/attention_seq.hip (one wave per (sequence, head), K / V read once for all query positions) against torch
    on the same bf16 operands: causal + key-padding mask (-1e9 before the bias) for self-attention; per-(head, key)
    bias and key/value blocks shared by `per_kv` consecutive sequences for cross-attention."""
    H, d = 8, 512
    rows = nseq * seq
    g = torch.Generator().manual_seed(nseq * 131 + nkeys)
    if kind == "self":
        qkv = (torch.randn(rows, 3 * d, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
        Q, K, V = qkv, qkv[:, d:], qkv[:, 2 * d:]
        ldq, kbs, krs = 3 * d, seq * 3 * d, 3 * d
        tok = torch.randint(1, 50, (nseq, seq), generator=g).to(DEV).to(torch.int32)
        tok[:, 0] = 1
        tok[0, 3] = 0
        tok[nseq - 1, seq - 2:] = 0       # PAD keys (id 0), also at a row's own position
        bias = None
        Kf = qkv[:, d:2 * d].float().view(nseq, seq, H, 64)
        Vf = qkv[:, 2 * d:].float().view(nseq, seq, H, 64)
        Qf = qkv[:, :d].float().view(nseq, seq, H, 64)
    else:
        nkv = (nseq + per_kv - 1) // per_kv
        assert nseq % per_kv == 0
        q = (torch.randn(rows, d, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
        kv = (torch.randn(nkv * nkeys, 2 * d, generator=g)).to(DEV).to(torch.bfloat16)
        Q, K, V = q, kv, kv[:, d:]
        ldq, kbs, krs = d, nkeys * 2 * d, 2 * d
        tok = None
        bias = (torch.randn(H, nkeys, generator=g) * 0.5).to(DEV) if kind == "cross_bias" else None
        idx = torch.arange(nseq, device=DEV) // per_kv
        Kf = kv[:, :d].float().view(nkv, nkeys, H, 64)[idx]
        Vf = kv[:, d:].float().view(nkv, nkeys, H, 64)[idx]
        Qf = q.float().view(nseq, seq, H, 64)
    ctx = torch.full((rows, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    _call("care_attention_seq", _p(Q), ldq, _p(K), _p(V), kbs, krs, per_kv, nkeys, 1 if kind == "self" else 0, seq,
          _p(tok), seq if tok is not None else 0, 0, _p(bias), nkeys if bias is not None else 0, _p(ctx), d, nseq, H)
    sc = torch.einsum("sqhe,skhe->shqk", Qf, Kf) / 8.0
    if kind == "self":
        mask = (tok == 0)[:, None, None, :] | torch.triu(torch.ones(seq, seq, device=DEV, dtype=torch.bool), 1)[None, None]
        sc = sc.masked_fill(mask, -1e9)
    if bias is not None:
        sc = sc + bias[None, :, None, :]
    pr = torch.softmax(sc, dim=-1)
    ref = torch.einsum("shqk,skhe->sqhe", pr, Vf).reshape(rows, d)
    torch.cuda.synchronize()
    got = ctx.float()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 4e-2      # bf16 probabilities and bf16 output
    assert (got - ref).abs().mean().item() < 3e-3


@pytest.mark.parametrize("d", [1024, 768])
@pytest.mark.parametrize("rows,nkeys,rows_per_kv,H,use_bias", [(1, 114, 1, 16, True), (300, 114, 1, 16, True), (77, 84, 1, 16, False),
                                                                (4096 + 5, 114, 1, 16, True), (35, 57, 5, 16, True),
                                                                (12, 40, 3, 12, False), (640, 114, 5, 16, True)])
def test_attention_latent_d1024(rows, nkeys, rows_per_kv, H, use_bias, d):
    """The several-waves-per-row form of the absorbed cross-attention (d_model = 1024: two waves of 512 dims; 768:
    three waves of 256 dims, 12 heads; the partial scores cross through LDS) against torch on the same bf16 operands."""
    H = H if d == 1024 else 12
    clips = (rows + rows_per_kv - 1) // rows_per_kv
    mem = _rand(clips, nkeys, d, seed=11).to(torch.bfloat16)
    qt = _rand(rows, H, d, seed=12, scale=0.09).to(torch.bfloat16)
    bias = _rand(H, nkeys, seed=13, scale=0.7) if use_bias else None
    ct = torch.full((rows, H, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    _call("care_attention_latent", _p(qt), H * d, _p(mem), nkeys * d, d, rows_per_kv, nkeys, _p(bias), nkeys, _p(ct),
          H * d, rows, H, d)
    clip_of = torch.arange(rows, device=DEV) // rows_per_kv
    m = mem.float()[clip_of]
    s = torch.einsum("rhc,rjc->rhj", qt.float(), m)
    if use_bias:
        s = s + bias[None]
    ref = torch.einsum("rhj,rjc->rhc", torch.softmax(s, -1), m)
    torch.cuda.synchronize()
    assert torch.isfinite(ct.float()).all()
    err = (ct.float() - ref).abs()
    assert err.max().item() < 1.2e-2 * max(1.0, ref.abs().max().item()) and err.mean().item() < 2e-3


@pytest.mark.parametrize("rows", [1, 130, 4096 + 9])
def test_gemm_tile_batched_head_projections(rows):
    """care_gemm_tile_batched in its two uses (d_model = 1024, 16 heads): qt[r][h] = wkt[h] q[r, 64 h : 64 h + 64] and
    ctx[r, 64 h : 64 h + 64] = W_v[64 h : 64 h + 64, :] ct[r][h] + b_v - against torch on the same bf16 operands."""
    H, d = 16, 1024
    q = _rand(rows, d, seed=21).to(torch.bfloat16)
    wkt = _rand(H, d, 64, seed=22, scale=0.05).to(torch.bfloat16)
    qt = torch.full((rows, H, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    _call("care_gemm_tile_batched", _p(q), d, 64, _p(wkt), 64, d * 64, None, 0, _p(qt), H * d, d, 1, H, rows, d, 64)
    ref = torch.einsum("rhe,hce->rhc", q.float().view(rows, H, 64), wkt.float())
    torch.cuda.synchronize()
    assert (qt.float() - ref).abs().max().item() < 2e-2
    ct = _rand(rows, H, d, seed=23, scale=0.5).to(torch.bfloat16)
    wv = _rand(d, d, seed=24, scale=0.03).to(torch.bfloat16)
    bv = _rand(d, seed=25)
    ctx = torch.full((rows, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    _call("care_gemm_tile_batched", _p(ct), H * d, d, _p(wv), d, 64 * d, _p(bv), 64, _p(ctx), d, 64, 1, H, rows, 64, d)
    ref2 = torch.einsum("rhc,hec->rhe", ct.float(), wv.float().view(H, 64, d)).reshape(rows, d) + bv
    torch.cuda.synchronize()
    assert (ctx.float() - ref2).abs().max().item() < 3e-2


@pytest.mark.parametrize("rows,nkeys,H,use_bias", [(1, 84, 8, True), (128, 114, 8, True), (200, 28, 8, False), (256, 128, 8, True),
                                                   (3, 16, 8, False), (77, 17, 4, True)])
def test_attention_latent_few_rows_is_bit_identical_to_the_streaming_kernel(rows, nkeys, H, use_bias):
    """Up to 256 rows the absorbed cross-attention runs one wave per row with the head of the row's stream in flight
    (attention_latent_few_kernel): the same arithmetic in the same order as the streaming kernel - so a launch of
    these rows alone equals, bit for bit, their slice of a 300-row launch (which takes the streaming kernel)."""
    d = 512
    big = 300
    mem = _rand(big, nkeys, d, seed=11).to(torch.bfloat16)
    qt = _rand(big, H, d, seed=12, scale=0.12).to(torch.bfloat16)
    bias = _rand(H, nkeys, seed=13, scale=0.7) if use_bias else None
    out_big = torch.full((big, H, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    out_few = torch.full((rows, H, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    args = lambda o, n: (_p(qt), H * d, _p(mem), nkeys * d, d, 1, nkeys, _p(bias), nkeys, _p(o), H * d, n, H, d)
    _call("care_attention_latent", *args(out_big, big))
    _call("care_attention_latent", *args(out_few, rows))
    torch.cuda.synchronize()
    assert torch.isfinite(out_few.float()).all()
    assert torch.equal(out_few, out_big[:rows])
    s = torch.einsum("rhc,rjc->rhj", qt.float(), mem.float())
    if use_bias:
        s = s + bias[None]
    ref = torch.einsum("rhj,rjc->rhc", torch.softmax(s, -1), mem.float())
    assert (out_big.float() - ref).abs().max().item() < 1.2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K,ks", [(1856, 512, 10547, 0), (300, 100, 2049, 3), (64, 64, 4096, 8), (1856, 512, 10547, 7)])
@pytest.mark.parametrize("a_is_km", [False, True])
def test_gemm_kn_split_k(M, N, K, ks, a_is_km):
    """care_gemm_kn_splitk: K ranges into slabs + care_strided_sum in order = the product (ks = 0: the count
    care_gemm_kn_splits picks); two runs are bit-identical (no atomics); a ksplit that leaves a slab without a range is refused."""
    from care_amd import _lib

    lib = _lib.load()
    if ks == 0:
        ks = lib.care_gemm_kn_splits(M, N, K)
        assert ks > 1
    A = _rand(K, M, seed=4) if a_is_km else _rand(M, K, seed=4)
    B = _rand(K, N, seed=5)
    outs = []
    for _ in range(2):
        slabs = torch.full((ks * M, N), float("nan"), device="cuda:0")
        _call("care_gemm_kn_splitk", _p(A), A.stride(0), int(a_is_km), _p(B), B.stride(0), _p(slabs), N, M * N, M, N, K, ks)
        C = torch.empty(M, N, device="cuda:0")
        _call("care_strided_sum", _p(slabs), N, _p(C), N, M, N, ks, 1, M, 1.0)
        outs.append(C)
    ref = ((A.t() if a_is_km else A).double() @ B.double())
    assert (outs[0].double() - ref).abs().max().item() < 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item())
    assert torch.equal(outs[0], outs[1])
    assert lib.care_gemm_kn_splits(10547, 512, 1856) == 1 and lib.care_gemm_kn_splits(1856, 512, 512) == 1
    with pytest.raises(_lib.CareHipError):  # 17 ranges of 16 cover K = 260 in 17 slabs, not 20
        _call("care_gemm_kn_splitk", _p(A), A.stride(0), int(a_is_km), _p(B), B.stride(0), _p(slabs), N, M * N, M, N, 260, 20)


@pytest.mark.parametrize("M,N,K", [(1856, 512, 10547), (10547, 512, 1856), (64, 64, 16), (1, 1, 1), (70, 33, 50), (512, 2048, 1856)])
@pytest.mark.parametrize("a_is_km", [False, True])
def test_gemm_kn_transposed_operands(M, N, K, a_is_km):
    """care_gemm_kn (the backward products of an nn.Linear on the operands as they lie: dx = dy W, dW = dy^T x) against
    torch's fp64 product of the same fp32 operands: exact-f32 MFMA sums K terms in fp32, K <= 10547 here."""
    A = _rand(K, M, seed=1) if a_is_km else _rand(M, K, seed=1)
    B = _rand(K, N, seed=2)
    C = torch.full((M, N), float("nan"), device="cuda:0")
    _call("care_gemm_kn", _p(A), A.stride(0), int(a_is_km), _p(B), B.stride(0), _p(C), N, M, N, K)
    ref = ((A.t() if a_is_km else A).double() @ B.double())
    err = (C.double() - ref).abs().max().item()
    assert err < 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item()), err
    # strided views (leading dimensions beyond the row length)
    if M >= 64 and K >= 16:
        Abig = _rand(A.shape[0], A.shape[1] + 5, seed=3)
        Av = Abig[:, : A.shape[1]]
        C2 = torch.empty(M, N + 3, device="cuda:0")
        _call("care_gemm_kn", _p(Av), Av.stride(0), int(a_is_km), _p(B), B.stride(0), _p(C2), C2.stride(0), M, N, K)
        ref2 = ((Av.t() if a_is_km else Av).double() @ B.double())
        assert (C2[:, :N].double() - ref2).abs().max().item() < 2e-6 * math.sqrt(K) * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(1, 500, 2048), (128, 500, 1536), (128, 512, 512), (5, 33, 96), (17, 1536, 512), (255, 500, 2048),
                                   (16, 8192, 64)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_f32_gemm_of_few_tiles_is_bit_identical_to_the_lds_tiled_kernel(M, N, K, act, monkeypatch):
    """care_gemm with fp32 weights and <= 512 tiles of 16 x 16 (the concept head of small batches, the fp32 mode at a few rows): a
    wave per tile, operands from registers in the LDS-tiled kernel's K order - the SAME BITS as that kernel (CARE_GEMM_FEW_TILES=0),
    for fp32 and 16-bit outputs, split destinations and leading dimensions beyond the row."""
    h16 = torch.bfloat16   # (2-byte outputs of the loaded library's 16-bit type, compared as bits)
    Abig = _rand(M, K + 8, seed=21)
    A = Abig[:, :K]
    W = _rand(N, K, seed=22, scale=1 / math.sqrt(K))
    bias = _rand(N, seed=23)
    ns = N if N % 32 else N // 2   # split destinations where the kernel takes them (n_split % 16 == 0)
    outs = {}
    for few in ("1", "0"):
        monkeypatch.setenv("CARE_GEMM_FEW_TILES", few)
        c0 = torch.full((M, ns + 3), float("nan"), device=DEV)
        c1 = torch.full((M, N - ns + 5), float("nan"), device=DEV, dtype=h16) if ns < N else None
        _call("care_gemm", _p(A), A.stride(0), _p(W), 0, _p(bias), _p(c0), c0.stride(0), 0, _p(c1) if c1 is not None else None,
              c1.stride(0) if c1 is not None else 0, 1, ns, M, N, K, act)
        nb = torch.full((M, N), float("nan"), device=DEV)
        _call("care_gemm", _p(A), A.stride(0), _p(W), 0, None, _p(nb), N, 0, None, 0, 0, N, M, N, K, 0)
        torch.cuda.synchronize()
        outs[few] = (c0, c1, nb)
    for a, b in zip(outs["1"], outs["0"]):
        if a is not None:
            assert torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16),
                               b.view(torch.int32 if b.dtype == torch.float32 else torch.int16))   # (NaN padding included)
    ref = A.double() @ W.double().t()
    assert (outs["1"][2].double() - ref).abs().max().item() < 2e-4
