// beam_pick.hip - the per-row top beam_size of log_softmax(x W^T) from the GROUP MAXIMA of the LDS-tiled vocabulary product
// (care_gemm_tile_beam, csrc/gemm_tile.hip): beam search over a few hundred to a few thousand rows.
//
// Replaces torch.log_softmax(logits, dim=1) (models/Translator.py:127) + the per-row part of Beam.advance's flattened top-k
// (misc/Decoding/Beam.py:60), like care_beam_select / care_beam_pick.  At these row counts the materialised logits cost a
// [rows, V] fp32 round trip per step (2560 rows: 108 MB written by the GEMM and read back by care_beam_select, 47 + 43 us) and
// the two-pass fused selection of the large batches needs 256-row panels it cannot fill.  Here the vocabulary product runs
// ONCE on the LDS-tiled kernel, whose epilogue keeps, per (row, 64-column part), the maximum, the sum of exponentials and the
// maxima of the part's sixteen 4-column groups (72 B per part instead of 256 B of logits).  A row's bm best logits lie in
// its bm best groups - the bm-th largest group maximum is a lower bound of the bm-th best logit, and a logit at least that
// large lifts its group's maximum that high - and those lie in its bm best PARTS.  One wave per row:
//   1. the parts' (max, sum exp) -> the row's log-sum-exp; the bm parts with the largest maxima;
//   2. their 16 bm group maxima -> the bm best groups;
//   3. the 4 bm logits of those groups RECOMPUTED: two MFMA tiles whose rows are the gathered weight rows and whose other
//      operand is the row's bf16 hidden state - the same v_mfma_f32_16x16x32 with the same operand roles and K in the same
//      ascending order into one accumulator as the tile kernel, so the same bits as the logits the maxima came from;
//   4. the bm best of them (value desc, column asc) as log-probabilities -> cand_val / cand_idx [rows, bm].
#include "decode_resident.h"

namespace {

constexpr int PK_NP = 4;  // parts per lane: V <= 64 * 64 * PK_NP
constexpr int PK_NQ = 8;  // K fragments (of 32) in flight per operand
constexpr int PK_BM = 8;  // beam sizes up to 8 (care_beam_select's and care_beam_advance's limit): two MFMA tiles of four groups

__global__ __launch_bounds__(256, 4) void beam_pick_groups_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                               const float* __restrict__ gmax, int parts, int bm,
                                                               const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W,
                                                               int V, int K, float* __restrict__ cand_val,
                                                               int32_t* __restrict__ cand_idx, int rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kg = lane >> 4;
  const int r = blockIdx.x * 4 + wave;
  if (r >= rows) return;  // (wave-uniform; no workgroup barrier below)
  // ---- 1. log-sum-exp of the row, its bm best parts
  float pm[PK_NP], ps[PK_NP];
#pragma unroll
  for (int k = 0; k < PK_NP; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < parts;
    pm[k] = pmax[(int64_t)r * parts + (ok ? c : 0)];
    ps[k] = psum[(int64_t)r * parts + (ok ? c : 0)];
    if (!ok) { pm[k] = -INFINITY; ps[k] = 0.f; }
  }
  float mloc = pm[0];
#pragma unroll
  for (int k = 1; k < PK_NP; ++k) mloc = fmaxf(mloc, pm[k]);
  const float M = care_wave_max_dpp(mloc);
  float sloc = 0.f;
#pragma unroll
  for (int k = 0; k < PK_NP; ++k) sloc += pm[k] == -INFINITY ? 0.f : ps[k] * expf(pm[k] - M);
  const float logS = logf(wave_sum_dpp(sloc));
  unsigned long long hk[PK_NP];
#pragma unroll
  for (int k = 0; k < PK_NP; ++k) hk[k] = pm[k] == -INFINITY ? 0ull : key_of(pm[k], (unsigned)(lane + 64 * k));
  // entry e = lane + 64 j (j < 2) of the bm x 16 group maxima: group e % 16 of the (e / 16)-th best part
  int mypart[2] = {-1, -1};
#pragma unroll
  for (int k = 0; k < PK_BM; ++k) {
    if (k >= bm) break;
    unsigned long long loc = hk[0];
#pragma unroll
    for (int q = 1; q < PK_NP; ++q) loc = max_u64(loc, hk[q]);
    const unsigned long long best = wave_max_u64(loc);
#pragma unroll
    for (int q = 0; q < PK_NP; ++q)
      if (hk[q] == best) hk[q] = 0ull;
    const int part = best ? (int)key_idx(best) : -1;
    if ((lane >> 4) == k) mypart[0] = part;
    if (((lane + 64) >> 4) == k) mypart[1] = part;
  }
  // ---- 2. the bm best groups among the 16 bm group maxima
  unsigned long long gk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bool ok = mypart[j] >= 0;
    const float v = gmax[((int64_t)r * parts + (ok ? mypart[j] : 0)) * 16 + l16];
    gk[j] = (ok && v != -INFINITY) ? key_of(v, (unsigned)(mypart[j] * 16 + l16)) : 0ull;
  }
  if (gk[1] > gk[0]) { const unsigned long long x = gk[0]; gk[0] = gk[1]; gk[1] = x; }
  int gsel[PK_BM];
#pragma unroll
  for (int k = 0; k < PK_BM; ++k) {
    gsel[k] = 0;
    if (k >= bm) continue;
    const unsigned long long best = wave_max_u64(gk[0]);
    gsel[k] = best ? (int)key_idx(best) : 0;
    if (gk[0] == best) { gk[0] = gk[1]; gk[1] = 0ull; }
  }
  // ---- 3. the 4 bm logits of those groups again: MFMA rows = the columns of groups (tile 0: groups 0 .. 3, tile 1: group 4);
  // lane (l16, kg) fetches the weight row of column 4 gsel[4 tile + l16 / 4] + l16 % 4 and receives the logits of columns
  // 4 gsel[4 tile + kg] + 0 .. 3 (every column of the MFMA = this row's hidden state)
  int gl[2], gout[2];
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    gl[tile] = gout[tile] = gsel[tile * 4 < PK_BM ? tile * 4 : 0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (tile * 4 + q < PK_BM) {
        if ((l16 >> 2) == q) gl[tile] = gsel[tile * 4 + q];
        if (kg == q) gout[tile] = gsel[tile * 4 + q];
      }
  }
  const bf16_t* w0 = W + (int64_t)min(gl[0] * 4 + (l16 & 3), V - 1) * K + kg * 8;
  const bf16_t* w1 = W + (int64_t)min(gl[1] * 4 + (l16 & 3), V - 1) * K + kg * 8;
  const bf16_t* ar = A + (int64_t)r * lda + kg * 8;
  f32x4 vt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // (K in chunks of PK_NQ fragments - 96 registers of fragments, four waves per SIMD: a row is a chain of four dependent
  // memory round trips, so rows in flight are what a launch of a few thousand rows needs; one accumulator chain per tile,
  // ascending K)
  for (int k0 = 0; k0 < K; k0 += 32 * PK_NQ) {
    const int nq = min(PK_NQ, (K - k0) >> 5);
    bf16x8 wf0[PK_NQ], wf1[PK_NQ], af[PK_NQ];
#pragma unroll
    for (int q = 0; q < PK_NQ; ++q) {
      const int o = k0 + min(q, nq - 1) * 32;
      wf0[q] = *reinterpret_cast<const bf16x8*>(w0 + o);
      if (bm > 4) wf1[q] = *reinterpret_cast<const bf16x8*>(w1 + o);
      af[q] = *reinterpret_cast<const bf16x8*>(ar + o);
    }
#pragma unroll
    for (int q = 0; q < PK_NQ; ++q)
      if (q < nq) {
        vt[0] = care_mfma_16x16x32_h16(wf0[q], af[q], vt[0], 0, 0, 0);
        if (bm > 4) vt[1] = care_mfma_16x16x32_h16(wf1[q], af[q], vt[1], 0, 0, 0);
      }
  }
  if (bm <= 4) vt[1] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  // ---- 4. one candidate per lane (l16 < 8: tile l16 / 4, element l16 % 4 of the lane's group), bm rounds of arg-best
  const int ctile = (l16 >> 2) & 1, ce = l16 & 3, cgrp = ctile * 4 + kg;
  const f32x4 vsel = ctile ? vt[1] : vt[0];
  const float cval = ce == 0 ? vsel[0] : ce == 1 ? vsel[1] : ce == 2 ? vsel[2] : vsel[3];
  const int ccol = (ctile ? gout[1] : gout[0]) * 4 + ce;
  unsigned long long ck = (l16 < 8 && cgrp < bm && ccol < V) ? key_of(cval, (unsigned)ccol) : 0ull;
#pragma unroll
  for (int k = 0; k < PK_BM; ++k) {
    if (k >= bm) break;
    const unsigned long long best = wave_max_u64(ck);
    if (ck == best) ck = 0ull;
    if (lane == 0) {  // log_softmax = (x - max) - log(sum), as csrc/beam.hip; no candidate left: (-inf, 0)
      cand_val[(int64_t)r * bm + k] = best ? (key_val(best) - M) - logS : -INFINITY;
      cand_idx[(int64_t)r * bm + k] = best ? (int)key_idx(best) : 0;
    }
  }
}

}  // namespace

extern "C" int care_beam_pick_groups(const float* pmax, const float* psum, const float* gmax, int parts, int bm, const void* A,
                                     int64_t lda, const void* W, int V, int K, float* cand_val, int32_t* cand_idx, int rows,
                                     void* stream) {
  if (!pmax || !psum || !gmax || !A || !W || !cand_val || !cand_idx || rows < 1 || parts < 1 || V < 1) return CARE_EINVAL;
  if (bm < 1 || bm > PK_BM || parts > 64 * PK_NP || parts != (V + 63) / 64 || K < 32 || K % 32 || V < 4 * PK_BM * 4) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda & 7) || (K & 7)) return CARE_EALIGN;
  hipLaunchKernelGGL(beam_pick_groups_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, pmax, psum, gmax, parts, bm,
                     (const bf16_t*)A, lda, (const bf16_t*)W, V, K, cand_val, cand_idx, rows);
  return care_launch_status();
}
