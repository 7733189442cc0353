// heads.hip - the two per-head projections around the absorbed cross-attention
// (attention_latent.hip): both are batched-over-heads GEMMs with one tiny dimension (64), bound
// by the [rows, heads, 512] bf16 tensor they write / read, so they are built W-STATIONARY: a wave
// keeps the MFMA fragments of its slice of the head's weight in 64 VGPRs for its whole life and
// streams 16-row tiles of activations past them.
//
//   care_head_expand:  qt[r][h][c]  = sum_e q[r][h*64+e] * wkt[h][c][e]          (K = 64,  N = 512 per head)
//   care_head_reduce:  ctx[r][h*64+e] = sum_c ct[r][h][c] * wv[h*64+e][c] + bv   (K = 512, N = 64 per head)
#include "care_common.h"

// cache policy of the two big streams (q~ out of the expansion, c~ into the reduction): see attention_latent.hip
#ifndef CARE_HEADS_ST_NT
#define CARE_HEADS_ST_NT 0
#endif
#ifndef CARE_HEADS_LD_AUX
#define CARE_HEADS_LD_AUX 0
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// expand: block = (head, row-tile stride); wave w owns output columns [128 w, 128 w + 128).
// D = W_frag (row operand: column c) x q_frag (column operand: batch row), so a lane ends up with
// consecutive output columns of one batch row (16-byte bf16 stores, see the column permutation).
__global__ __launch_bounds__(256) void head_expand_kernel(const bf16_t* q, int64_t ldq, const bf16_t* wkt, bf16_t* qt,
                                                          int64_t ldo, int rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int h = blockIdx.x;
  // MFMA tile nt, D row i computes output column  (nt>>1)*32 + (i>>2)*8 + (nt&1)*4 + (i&3)  of the
  // wave's 128, so that after tiles 2k and 2k+1 a lane (D rows 4 fg .. 4 fg + 3 of each) holds the
  // 8 CONSECUTIVE columns k*32 + fg*8 .. +7: one 16-byte store per tile pair.
  const bf16_t* wh = wkt + (int64_t)h * 512 * 64 + (int64_t)(wave * 128 + (fr >> 2) * 8 + (fr & 3)) * 64 + fg * 8;
  bf16x8 bw[8][2];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      bw[nt][ks] = *reinterpret_cast<const bf16x8*>(wh + ((nt >> 1) * 32 + (nt & 1) * 4) * 64 + ks * 32);

  const int tiles = (rows + 15) / 16;
  const bf16_t* qh = q + h * 64 + fg * 8;
  auto load_a = [&](int T, bf16x8 (&af)[2]) {
    const bf16_t* src = qh + (int64_t)min(T * 16 + fr, rows - 1) * ldq;
    af[0] = *reinterpret_cast<const bf16x8*>(src);
    af[1] = *reinterpret_cast<const bf16x8*>(src + 32);
  };
  bf16x8 af[2], an[2];
  int T = blockIdx.y;
  if (T < tiles) load_a(T, af);
  for (; T < tiles; T += gridDim.y) {
    const int Tn = T + gridDim.y;
    load_a(min(Tn, tiles - 1), an);  // prefetch (the last iteration re-reads its own tile)
    const int r = T * 16 + fr;
    // rows past the end are clamped to the last row: they recompute and rewrite ITS values
    bf16_t* out = qt + (int64_t)min(r, rows - 1) * ldo + h * 512 + wave * 128 + fg * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16x8 o;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = care_mfma_16x16x32_h16(bw[2 * k + half][0], af[0], acc, 0, 0, 0);
        acc = care_mfma_16x16x32_h16(bw[2 * k + half][1], af[1], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[half * 4 + j] = (bf16_t)acc[j];
      }
      if (CARE_HEADS_ST_NT) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(out + k * 32));
      else *reinterpret_cast<bf16x8*>(out + k * 32) = o;
    }
    af[0] = an[0]; af[1] = an[1];
  }
}

// ---------------------------------------------------------------------------------------------
// reduce: block = (head, row-tile stride); wave w owns output columns h*64 + [16 w, 16 w + 16) and
// keeps W_v rows of those columns (16 x 512) as 16 fragments.  Tiles of 16 latent-context rows
// (1 KiB each, contiguous) stream through a 3-slot LDS ring by LDS-DMA, 4 rows per wave, two
// tiles ahead; one raw barrier per tile, counted waits.
constexpr int RED_SLOTS = 3;
constexpr int RED_TILE = 16 * 1024;

__global__ __launch_bounds__(256) void head_reduce_kernel(const bf16_t* ct, int64_t ldc, const bf16_t* wv,
                                                          const float* bv, bf16_t* ctx, int64_t ldo, int rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int h = blockIdx.x;
  const bf16_t* wh = wv + (int64_t)(h * 64 + wave * 16 + fr) * 512 + fg * 8;
  bf16x8 bw[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bw[ks] = *reinterpret_cast<const bf16x8*>(wh + ks * 32);
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (bv) bias = *reinterpret_cast<const f32x4*>(bv + h * 64 + wave * 16 + fg * 4);

  const int tiles = (rows + 15) / 16;
  const int nmine = blockIdx.y < tiles ? (tiles - blockIdx.y + gridDim.y - 1) / gridDim.y : 0;
  // LDS row i of a tile = latent context of batch row T*16+i (clamped), 16-byte chunk k at k ^ i
  auto stage = [&](int n, int slot) {
    const int T = blockIdx.y + n * gridDim.y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = wave * 4 + j;
      const unsigned char* g = reinterpret_cast<const unsigned char*>(ct + (int64_t)min(T * 16 + i, rows - 1) * ldc +
                                                                      h * 512) + ((lane ^ i) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(smem + slot * RED_TILE + i * 1024),
                                       16, 0, CARE_HEADS_LD_AUX);
    }
  };
  int roff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) roff[r] = fr * 1024 + ((((r * 4 + fg) ^ fr) & 15) << 4);

  if (nmine > 0) stage(0, 0);
  if (nmine > 1) stage(1, 1);
  for (int n = 0; n < nmine; ++n) {
    // VM ops younger than tile n's DMAs: [store of tile n-1 (1)] + [DMAs of tile n+1 (4)] when they exist
    if (n + 1 < nmine) {
      if (n == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* sb = smem + (n % RED_SLOTS) * RED_TILE;
    f32x4 acc = bias;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + roff[ks & 3] + (ks >> 2) * 256);
      acc = care_mfma_16x16x32_h16(bw[ks], a, acc, 0, 0, 0);
    }
    // lane (batch row fr, group fg) holds output columns h*64 + wave*16 + fg*4 + 0..3
    const int T = blockIdx.y + n * gridDim.y;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)acc[j];
    *reinterpret_cast<bf16x4*>(ctx + (int64_t)min(T * 16 + fr, rows - 1) * ldo + h * 64 + wave * 16 + fg * 4) = o;
    __builtin_amdgcn_sched_barrier(0);
    // slot of tile n-1: every wave passed this iteration's barrier after reading it
    if (n + 2 < nmine) stage(n + 2, (n + 2) % RED_SLOTS);
  }
}

}  // namespace

extern "C" int care_head_expand(const void* q, int64_t ldq, const void* wkt, void* qt, int64_t ldo, int rows, int heads,
                                void* stream) {
  if (!q || !wkt || !qt || rows <= 0 || heads <= 0) return CARE_EINVAL;
  if ((ldq % 8) || (ldo % 8) || !care_aligned16(q) || !care_aligned16(wkt) || !care_aligned16(qt)) return CARE_EALIGN;
  const int tiles = (rows + 15) / 16;
  const int nb = min(tiles, max(1, 1024 / heads));
  hipLaunchKernelGGL(head_expand_kernel, dim3(heads, nb), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(q), ldq, reinterpret_cast<const bf16_t*>(wkt),
                     reinterpret_cast<bf16_t*>(qt), ldo, rows);
  return care_launch_status();
}

extern "C" int care_head_reduce(const void* ct, int64_t ldc, const void* wv, const float* bv, void* ctx, int64_t ldo,
                                int rows, int heads, void* stream) {
  if (!ct || !wv || !ctx || rows <= 0 || heads <= 0) return CARE_EINVAL;
  if ((ldc % 8) || (ldo % 4) || !care_aligned16(ct) || !care_aligned16(wv) || !care_aligned16(ctx) ||
      (bv && !care_aligned16(bv)))
    return CARE_EALIGN;
  const int tiles = (rows + 15) / 16;
  const int nb = min(tiles, max(1, 768 / heads));
  hipLaunchKernelGGL(head_reduce_kernel, dim3(heads, nb), dim3(256), RED_SLOTS * RED_TILE, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(ct), ldc, reinterpret_cast<const bf16_t*>(wv), bv,
                     reinterpret_cast<bf16_t*>(ctx), ldo, rows);
  return care_launch_status();
}
