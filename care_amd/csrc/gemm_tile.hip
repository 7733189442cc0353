// gemm_tile.hip - C = act(A W^T + bias) for bf16 A [M, K] and bf16 W [N, K] with ANY K % 64 == 0:
// the nn.Linear layers of the d_model = 768 / 1024 architectures (config/archs.yaml:15-26: K = 768, 1024,
// 3072, 4096), where the A-stationary kernels (csrc/gemm_as.hip, gemm_store32.hip, gemm_vocab.hip: the
// activations of a row panel resident in registers for the WHOLE K) run out of registers at K > 512.
//
// Classic output-tile decomposition, built for gfx950:
//   * a workgroup of WGM x WGN waves owns a (64 WGM) x (64 WGN) output tile, a wave 64 x 64 of it
//     (16 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 64 VGPRs);
//   * BOTH operands stream through an LDS ring in K steps of 64 (rows of 128 bytes) by LDS-DMA
//     (global_load_lds, 16 B per lane: one wave instruction = 8 rows x 128 B, full cache lines), STAGES - 1
//     steps in flight; the LDS image is lane-linear, so the bank swizzle (16-byte chunk ^= row & 7, the
//     conflict-free pattern of ds_read_b128's 16-lane groups) is applied to the per-lane SOURCE address
//     and again on the fragment read;
//   * one raw s_barrier per K step: [counted s_waitcnt vmcnt -> barrier -> issue step t + STAGES - 1 ->
//     multiply step t]; the wait precedes the barrier that precedes the read (LDS-DMA data is ordered for a
//     ds_read only by the issuing wave's vmcnt + a barrier), the re-staged slot was read one barrier ago;
//   * MFMA operands are swapped (D = W_tile A_tile^T) and the W rows of a wave's 64 columns are permuted
//     over the MFMA rows WHEN STAGED (LDS row 16 nt + 4 g + r holds column 16 g + 4 nt + r), so a lane ends
//     up with 16 CONSECUTIVE output columns of one row: 16-byte stores, and the row statistics of the
//     arg-max epilogue are a 16-value chain in one lane + two cross-lane steps;
//   * epilogues: bias + activation + (split) store in bf16 / fp32 - or per-row (max, first arg-max,
//     sum exp) of the 64 columns of every wave, the vocabulary projection of greedy decoding
//     (models/Head.py:26-32, Translator.py:127), optionally with the label logit (teacher-forced scoring);
//   * blocks are numbered XCD-aware: the blocks of one XCD (blockIdx % 8) walk a contiguous run of tiles in bands of 8
//     row tiles (tile_of_block), so the 32 tiles an XCD works on at a time share 8 A panels and 4 W panels in its L2.
// What bounds it (*measured*, round 4): 4096^3 1296 TFLOP/s, 8192^3 1372 on random operands
// (0.52-0.55 of the dense bf16 peak; zero-filled operands read 18-20 % higher: clocks).  With the LDS-DMA removed the
// loop runs at 1.8-2.0 PFLOP/s, with 15 of 16 MFMAs removed no faster than whole: the tile is bound by FETCHING its
// operands (tools/micro/dma_rate.hip: this tile's fetch alone runs at 24.5-25 TB/s chip-wide in pieces of 8 rows x 128 B,
// 15.7-17.4 TB/s in pieces of 16 rows x 64 B).  Two-wave-group variants of the 256 x 256 tile (8 waves of 128 x 64, the
// groups one barrier apart so one multiplies while the other reads and issues DMA; K steps of 32 in rings of 4 and 5, K
// steps of 64 in row groups with 64 KB in flight; one persistent with the next tile's first stages in flight across the
// epilogue) were built, verified, and landed within 2 % of this kernel on every shape: not kept.  Round 5: the wave-specialised
// persistent form that took the embedder's fused kernel from 2.86 to 2.07 ms (csrc/gemm_ln.hip version 3: 8 compute waves of
// 128 x 64 + 4 loader waves, A ring of three stages, W ring of two) - same bits, 5 - 13 % SLOWER here on every shape
// (8192^3 1226 against 1310 TFLOP/s, 118784 x 2048 x 512 732 against 845): this tile is not bound by its streams (37 GB/s
// per CU against the 57 the embedder's streams reach) but by two 128-accumulator waves per SIMD with 168 registers, which
// cannot double-buffer their fragments - the sixteen 64-accumulator waves hide the LDS latency by occupancy.  Not kept.
// In-kernel stamps of this kernel (tools/tile_ts.py, 16384 x 4096 x 4096): 3700 ticks per K step against 2048 of MFMA for the four
// waves of a SIMD - every wave waits ~450 ticks for its own stage to land (one stage of prefetch: the stage issued behind a
// barrier is awaited before the next), the first wave ~1700 at the barrier for the last, whose four DMA pieces took 1200 to
// issue behind the other fifteen's.  Weaving the pieces into the step's MFMAs (one behind every eight) took 6 % fewer ticks
// per step and 3 - 10 % MORE time on every shape (8192^3 1232 against 1365 TFLOP/s): not kept either.  Nor was the tile with
// the operand streams on opposite halves of the workgroup and in opposite phase (waves 8 - 15 issue W stage kt + 1 behind the
// barrier and then multiply; waves 0 - 7 multiply first and issue A stage kt + 2 afterwards, into a ring of three A stages +
// two W stages = 160 KB): bit-identical, within +- 4 % on every shape (8192^3 1368 against 1380) - its stamps show the A
// waves now idle 1750 ticks at the barrier for the W waves, whose blocked issue (~1000 ticks for 32 KB through the CU's
// in-order memory path) + MFMAs + the 450-tick landing wait of a one-stage-ahead W ring are the period again.  The 64 KB a
// 256 x 256 x 64 step fetches need 1300 - 2000 ticks of that path against 2048 of MFMA, and 160 KB of LDS hold no third stage.
#include <cstdlib>

#include "care_common.h"

namespace {

struct TArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  void* C0; int64_t ldc0; int c0_bf16;
  void* C1; int64_t ldc1; int c1_bf16;
  int n_split, M, N, K, act;
  float* pmax; int32_t* pidx; float* psum; int parts;
  const int32_t* labels; float* plab;
  float* gmax;  // EPI_BEAM: per (row, 64-column part) the maxima of its sixteen 4-column groups
  int tiles_m, tiles_n, group;  // group: row tiles per band of the block -> tile map
  int a_wrap;  // K steps (of 64) after which the A columns start over: see care_gemm_tile_split3 (INT_MAX otherwise)
  int64_t a_bs, w_bs, c_bs; int bias_bs;  // batched launches (blockIdx.y): element offsets per batch of A, W, C0, bias
  // split products of PRE-SCALED operands (care_gemm_tile_split3_scaled): the |max| bit patterns the two operands' power-of-two
  // scales were derived from (care_absmax); the epilogue multiplies the accumulators by 1 / (scale_a scale_b).  NULL: no scaling.
  const unsigned* amax_a; const unsigned* amax_b;
};

// The power of two that brings a tensor's largest magnitude into [2^14, 2^15) - the top of fp16's range, so that the LOW piece
// of a hi / lo split (<= 2^-11 of its element) stays a normal fp16 number for every element within 2^-14 of the largest:
// gradients of ~1e-6 split as exactly as activations of ~1.  bits = the fp32 pattern of the maximum (care_absmax); zero,
// denormal or non-finite maxima: no scaling.  pow2_unscale: the reciprocal of the product of two such scales.
__device__ __forceinline__ int pow2_scale_exp(unsigned bits) {
  const int E = (int)(bits >> 23) & 255;
  return (E == 0 || E == 255) ? 0 : min(max(141 - E, -100), 100);
}
__device__ __forceinline__ float pow2_of(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }

enum { EPI_STORE = 0, EPI_ARGMAX = 1, EPI_ARGMAX_LAB = 2, EPI_BEAM = 3 };

// Block -> output tile.  Blocks are dealt to the 8 XCDs round robin and in order, so the 32 workgroups an XCD runs at a
// time are 32 CONSECUTIVE numbers of its run (bijective for any grid size); inside a run the tiles go in bands of
// `group` row tiles, row tile fastest inside a band: those 32 tiles are `group` row tiles x 32 / group column tiles and
// the XCD's L2 fetches group + 32 / group operand panels for them, not 32 + 1 (all row tiles of one column tile:
// *measured* M = 65536, N = 4096, K = 1024: every A panel missed the L2, the kernel ran at the 8 TB/s of the
// Infinity Cache with 1 MFMA in 16 removed and at 1.3 x the rate with the loads removed).
__device__ __forceinline__ void tile_of_block(const int tiles_m, const int tiles_n, const int group, int& tm, int& tn) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  const int band = group * tiles_n, b = t / band, o = t - b * band;
  const int rows = min(group, tiles_m - b * group);
  tm = b * group + o % rows;
  tn = o / rows;
}

__device__ __forceinline__ float t_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int N_>
__device__ __forceinline__ void t_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory");
}

// The epilogue of a wave's (16 MT) x 64 outputs:
// acc[m][n][j] = out[row row0 + 16 m + fr][column col0 + 4 n + j], col0 = (the wave's first column) + 16 fg
// GELU is a template parameter: erff inlined for the 16 MT values of a lane behind a run-time test was ~ 9000 instructions
// of epilogue, more than the instruction cache holds - *measured* 13 100 cycles per tile and wave group with the
// activation OFF (s_memtime stamps around the epilogue of a persistent variant of this tile), a sixth of a 256 x 256 x 1024 tile.
template <int EPI, int MT, bool GELU>
__device__ __forceinline__ void tile_epilogue(const TArgs& p, f32x4 (&acc)[MT][4], int row0, int col0, int fr, int fg) {
  if constexpr (EPI == EPI_STORE) {
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = 0.0f;
    if (p.bias) {
      if (col0 + 16 <= p.N && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col0 + 4 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[4 * g + j] = b4[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (col0 + j < p.N) bv[j] = p.bias[col0 + j];
      }
    }
    float alpha = 1.0f;
    if (p.amax_a) {  // (kernel-uniform) pre-scaled split operands: back to the true magnitude, exactly (a power of two)
      const int ea = pow2_scale_exp(*p.amax_a), eb = pow2_scale_exp(*p.amax_b);
      alpha = pow2_of(-ea) * pow2_of(-eb);
    }
    const bool second = col0 >= p.n_split;  // n_split % 16 == 0: a lane's 16 columns never straddle it
    unsigned char* C = reinterpret_cast<unsigned char*>(second ? p.C1 : p.C0);
    const int64_t ld = second ? p.ldc1 : p.ldc0;
    const bool isb = (second ? p.c1_bf16 : p.c0_bf16) != 0;
    const int cc = col0 - (second ? p.n_split : 0);
    const int nvalid = min(16, (second ? p.N : min(p.N, p.n_split)) - col0);  // columns of this lane inside the destination
    const bool vec = nvalid == 16 && ((ld & 7) == 0) && ((cc & 7) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    // the wave's 64 columns whole and in one destination (wave-uniform): the four lanes of a row trade 16-byte chunks so
    // that ONE store instruction writes 64 contiguous bytes of every row (bf16) - *measured* on the 256 x 256 tile: a
    // lane storing its own 32 bytes as two 16-byte pieces (64 separate 16-byte writes per instruction) took 13 200
    // cycles per wave group and tile, a sixth of the tile's time at K = 1024
    const int colw = col0 - fg * 16;
    const bool whole = vec && colw + 64 <= (second ? p.N : min(p.N, p.n_split)) && (colw >= p.n_split || colw + 64 <= p.n_split);
    const bool wide = isb && whole, wide32 = !isb && whole;
    const bool wide_u = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_ballot_w64(!wide) == 0 && (int)(__builtin_amdgcn_ballot_w64(!wide) >> 32) == 0);
    // fp32 outputs: a lane's 16 columns are FOUR 16-byte chunks of the row's sixteen; a 4 x 4 transpose over the row's four
    // lanes (v_permlane32_swap, then v_permlane16_swap: lane fg ends up with chunks fg, fg + 4, fg + 8, fg + 12) makes every
    // store instruction 64 contiguous bytes per row here too
    const bool wide32_u = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_ballot_w64(!wide32) == 0 && (int)(__builtin_amdgcn_ballot_w64(!wide32) >> 32) == 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = row0 + m * 16 + fr;
      float v[16];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = (p.amax_a ? acc[m][n][j] * alpha : acc[m][n][j]) + bv[4 * n + j];
          if constexpr (GELU) x = t_gelu(x);
          else x = p.act == CARE_ACT_RELU ? fmaxf(x, 0.0f) : x;
          v[4 * n + j] = x;
        }
      if (wide_u) {
        bf16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o0[j] = (bf16_t)v[j]; o1[j] = (bf16_t)v[8 + j]; }
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 lo = __builtin_bit_cast(i32x4, o0), hi = __builtin_bit_cast(i32x4, o1), x, y;
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // lanes fg < 2: (own lo, partner's lo); fg >= 2: (partner's hi, own hi) - chunks (c, c + 4)
          const auto r = __builtin_amdgcn_permlane32_swap(lo[d], hi[d], false, false);
          x[d] = (int)r[0]; y[d] = (int)r[1];
        }
        const int c = fg < 2 ? 2 * fg : 2 * fg - 3;
        if (row < p.M) {
          bf16_t* dst = reinterpret_cast<bf16_t*>(C) + (int64_t)row * ld + (cc - fg * 16) + 8 * c;
          *reinterpret_cast<i32x4*>(dst) = x;
          *reinterpret_cast<i32x4*>(dst + 32) = y;
        }
        continue;
      }
      if (wide32_u) {
        int x[4][4], y[4][4];  // [chunk slot][dword]
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // step 1 (lane ^ 32): chunks (0, 2) and (1, 3) trade halves
          const auto r02 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v[d]), __builtin_bit_cast(int, v[8 + d]), false, false);
          const auto r13 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v[4 + d]), __builtin_bit_cast(int, v[12 + d]), false, false);
          x[0][d] = (int)r02[0]; x[2][d] = (int)r02[1];
          x[1][d] = (int)r13[0]; x[3][d] = (int)r13[1];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // step 2 (lane ^ 16): slots (0, 1) and (2, 3)
          const auto r01 = __builtin_amdgcn_permlane16_swap(x[0][d], x[1][d], false, false);
          const auto r23 = __builtin_amdgcn_permlane16_swap(x[2][d], x[3][d], false, false);
          y[0][d] = (int)r01[0]; y[1][d] = (int)r01[1];
          y[2][d] = (int)r23[0]; y[3][d] = (int)r23[1];
        }
        if (row < p.M) {
          typedef int i32x4 __attribute__((ext_vector_type(4)));
          float* dst = reinterpret_cast<float*>(C) + (int64_t)row * ld + (cc - fg * 16) + 4 * fg;
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<i32x4*>(dst + 16 * k) = i32x4{y[k][0], y[k][1], y[k][2], y[k][3]};
        }
        continue;
      }
      if (row >= p.M || nvalid <= 0) continue;
      if (isb) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(C) + (int64_t)row * ld + cc;
        if (vec) {
          bf16x8 o0, o1;
#pragma unroll
          for (int j = 0; j < 8; ++j) { o0[j] = (bf16_t)v[j]; o1[j] = (bf16_t)v[8 + j]; }
          *reinterpret_cast<bf16x8*>(dst) = o0;
          *reinterpret_cast<bf16x8*>(dst + 8) = o1;
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (j < nvalid) dst[j] = (bf16_t)v[j];
        }
      } else {
        float* dst = reinterpret_cast<float*>(C) + (int64_t)row * ld + cc;
        if (vec) {
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dst + 4 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (j < nvalid) dst[j] = v[j];
        }
      }
    }
  } else {
    // per row of this wave's 64 columns: max, FIRST arg-max, sum exp(x - max) [, the logit of the label column]
    const int part = (col0 - fg * 16) >> 6;  // one partial per 64 columns, whatever the tile shape
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = row0 + m * 16 + fr;
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = col0 + 4 * n + j;
          const float x = col < p.N ? acc[m][n][j] : -INFINITY;
          if (x > best) { best = x; bi = col; }  // columns ascend: the first maximum is kept
        }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      float s = 0.0f;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col0 + 4 * n + j < p.N) s += __expf(acc[m][n][j] - best);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      float lv = -INFINITY;
      if constexpr (EPI == EPI_ARGMAX_LAB) {
        const int lab = p.labels[min(row, p.M - 1)];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col0 + 4 * n + j == lab) lv = acc[m][n][j];
        lv = fmaxf(lv, __shfl_xor(lv, 16, 64));
        lv = fmaxf(lv, __shfl_xor(lv, 32, 64));
      }
      if (fg == 0 && row < p.M && part < p.parts) {
        const int64_t o = (int64_t)row * p.parts + part;
        p.pmax[o] = best; p.psum[o] = s;
        if constexpr (EPI != EPI_BEAM) p.pidx[o] = bi;
        if constexpr (EPI == EPI_ARGMAX_LAB) p.plab[o] = lv;
      }
      if constexpr (EPI == EPI_BEAM) {
        // beam search: the maxima of this lane's four 4-column groups (groups 4 fg .. 4 fg + 3 of the part) - a row's bm best
        // logits lie in its bm best groups (the bm-th largest group maximum is a lower bound of the bm-th best logit),
        // which care_beam_pick_groups recomputes; one 16-byte store per lane and row
        if (row < p.M && part < p.parts) {
          f32x4 gm;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            float g4 = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col0 + 4 * n + j < p.N) g4 = fmaxf(g4, acc[m][n][j]);
            gm[n] = g4;
          }
          *reinterpret_cast<f32x4*>(p.gmax + ((int64_t)row * p.parts + part) * 16 + fg * 4) = gm;
        }
      }
    }
  }
}

// BK = K elements per ring stage: 64 (128-byte LDS rows, two MFMA k-steps per stage) or 32 (64-byte rows, one k-step:
// twice the stages in the same LDS, i.e. more K steps of prefetch for the same bytes in use).  Swizzle of a 64-byte
// row's four 16-byte chunks: chunk ^= G[(row >> 2) & 3], G = {0, 3, 2, 1} - the 16 lanes of every ds_read_b128 lane
// group ({0-3, 12-15, 20-27}, ...) then fall on 16 different 16-byte slots of the 256-byte bank row.
#ifndef CARE_TILE_DBG
#define CARE_TILE_DBG 0  // tools (tools/variant_lib.py, tools/tile_ts.py): 64 - workgroup 0 stamps s_memtime per wave and K step
#endif
#if CARE_TILE_DBG & 64
__device__ unsigned long long tile_stamps[16 * 64 * 4];
#define TILE_STAMP(kt, k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (kt) < 64 && lane == 0 && wave < 16) tile_stamps[(wave * 64 + (kt)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TILE_STAMP(kt, k) do { } while (0)
#endif

template <int WGM, int WGN, int WTM, int STAGES, int EPI, bool F16 = false, int BK = 64, bool GELU = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tile_kernel(TArgs p) {
  constexpr int NW = WGM * WGN, BM = 64 * WTM * WGM, BN = 64 * WGN;
  constexpr int MT = 4 * WTM;             // 16-row accumulator tiles of a wave (its (64 WTM) x 64 outputs)
  constexpr int ROWB = BK * 2;            // bytes of an LDS row
  constexpr int RPP = 1024 / ROWB;        // rows per 1-KB DMA piece (one wave instruction)
  constexpr int CPR = ROWB / 16;          // 16-byte chunks per row
  constexpr int PIECES = (BM + BN) / RPP; // DMA pieces of one K step
  constexpr int P = PIECES / NW;          // pieces per wave and K step
  static_assert(BK == 64 || BK == 32, "K step");
  static_assert(PIECES % NW == 0, "pieces must divide over the waves");
  static_assert(STAGES >= 2 && STAGES <= 6, "ring depth");
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  int tm, tn;
  tile_of_block(p.tiles_m, p.tiles_n, p.group, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  if (gridDim.y > 1) {  // batched: one independent product per blockIdx.y (the per-head projections of the absorbed attention)
    const int b = blockIdx.y;
    p.A += (int64_t)b * p.a_bs;
    p.W += (int64_t)b * p.w_bs;
    p.C0 = reinterpret_cast<unsigned char*>(p.C0) + (int64_t)b * p.c_bs * (p.c0_bf16 ? 2 : 4);
    if (p.bias) p.bias += (int64_t)b * p.bias_bs;
  }

  // ---- staging: piece q = wave * P + i covers LDS rows 8 q .. 8 q + 7 of the stage image (A rows first)
  const int prow = lane / CPR;
  // source chunk of this lane's LDS slot (the piece starts at a multiple of RPP rows: row & 7 == prow, (row >> 2) & 3 == (prow >> 2) & 3)
  const int sc = (BK == 64 ? ((lane & 7) ^ prow) : ((lane & 3) ^ ((0x1230 >> (4 * ((prow >> 2) & 3))) & 3))) << 4;
  const unsigned char* src[P];
  bool is_a[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    is_a[i] = (wave * P + i) * RPP < BM;
    const int R = (wave * P + i) * RPP + prow;
    if ((wave * P + i) * RPP < BM) {
      const int row = min(m0 + R, p.M - 1);  // clamped, never branched around: rows past M are not stored
      src[i] = reinterpret_cast<const unsigned char*>(p.A) + (int64_t)row * p.lda * 2 + sc;
    } else {
      const int Rb = R - BM;  // LDS row 16 nt + 4 g + r of a wave's 64 columns holds column 16 g + 4 nt + r
      const int col = (Rb & ~63) + 16 * ((Rb >> 2) & 3) + 4 * ((Rb >> 4) & 3) + (Rb & 3);
      const int row = min(n0 + col, p.N - 1);
      src[i] = reinterpret_cast<const unsigned char*>(p.W) + (int64_t)row * p.ldw * 2 + sc;
    }
  }
  auto issue = [&](int kt, int slot) {
    const int kta = F16 ? kt - (kt >= p.a_wrap ? p.a_wrap : 0) : kt;  // split products: the A columns start over
#pragma unroll
    for (int i = 0; i < P; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)((F16 && is_a[i]) ? kta : kt) * ROWB),
                                       (__attribute__((address_space(3))) void*)(smem + slot * STAGE_BYTES +
                                                                                 (wave * P + i) * 1024),
                                       16, 0, 0);
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);

  const int fr = lane & 15, fg = lane >> 4;
  const int a_row = (wm * 64 * WTM + fr) * ROWB, b_row = (BM + wn * 64 + fr) * ROWB;
  const int sw0 = BK == 64 ? ((fg ^ (fr & 7)) << 4) : ((fg ^ ((0x1230 >> (4 * ((fr >> 2) & 3))) & 3)) << 4);
  const int sw1 = (((4 + fg) ^ (fr & 7)) << 4);

  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_sched_barrier(0);  // the MFMAs of step kt - 1 (and the waits on their fragments) stay above the barrier
    // step kt has landed when at most the STAGES - 2 younger steps of this wave are outstanding
    TILE_STAMP(kt, 0);
    if (kt + STAGES - 2 < nk) t_wait_vm<(STAGES - 2) * P>();
    else t_wait_vm<0>();
    TILE_STAMP(kt, 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    TILE_STAMP(kt, 2);
    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    __builtin_amdgcn_sched_barrier(0);
    TILE_STAMP(kt, 3);
    const unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int sw = kk ? sw1 : sw0;
      bf16x8 fa[MT], fb[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(st + b_row + n * 16 * ROWB + sw);
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(st + a_row + m * 16 * ROWB + sw);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          if constexpr (F16) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[n]), __builtin_bit_cast(f16x8, fa[m]),
                                                               acc[m][n], 0, 0, 0);
          } else {
            acc[m][n] = care_mfma_16x16x32_h16(fb[n], fa[m], acc[m][n], 0, 0, 0);
          }
    }
  }

  TILE_STAMP(nk, 0);
  tile_epilogue<EPI, MT, GELU>(p, acc, m0 + wm * 64 * WTM, n0 + wn * 64 + fg * 16, fr, fg);
  TILE_STAMP(nk, 1);
}

int tile_group(int tiles_m) {
  static const int g = [] { const char* e = getenv("CARE_TILE_GROUP"); return e ? atoi(e) : 8; }();
  return g > 0 ? g : tiles_m;  // 0: the whole column of row tiles (the map before the bands)
}

template <int WGM, int WGN, int WTM, int STAGES, int EPI, bool F16 = false, int BK = 64, bool GELU = false>
int launch_tile(TArgs& p, hipStream_t st, int batch = 1) {
  if constexpr (EPI == EPI_STORE && !GELU)
    if (p.act == CARE_ACT_GELU) return launch_tile<WGM, WGN, WTM, STAGES, EPI, F16, BK, true>(p, st, batch);
  constexpr int BM = 64 * WTM * WGM, BN = 64 * WGN;
  constexpr int lds = STAGES * (BM + BN) * BK * 2;
  if (BK == 32 && F16) p.a_wrap *= 2;  // a_wrap is handed over in K steps of 64
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.group = tile_group(p.tiles_m);
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};
    const int rc = care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_tile_kernel<WGM, WGN, WTM, STAGES, EPI, F16, BK, GELU>), lds, done);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((gemm_tile_kernel<WGM, WGN, WTM, STAGES, EPI, F16, BK, GELU>), dim3(p.tiles_m * p.tiles_n, batch), dim3(64 * WGM * WGN), lds, st, p);
  return care_launch_status();
}

// Tile shape (CARE_TILE_CFG overrides, tuning): 256 x 256 tiles of 16 waves (half the L2 -> LDS bytes per flop of
// 128 x 128, four waves per SIMD) as soon as they occupy half the 256 CUs, 128 x 128 tiles of 4 waves (two workgroups
// per CU) below.  *Measured* (tools/tile_bench.py, one MI355X, bf16 out): M = 4096, K = 1024: N = 4096 952 vs 701
// TFLOP/s, N = 3072 (192 big tiles) 843 vs 580, N = 2304 / K = 768 (144) 576 vs 484, N = 1024 (64) 314 vs 483;
// M = 466944, N = 2048 929 vs 682; 8 waves of 128 x 64 (2422) and 256 x 128 tiles (422, 2222) lie in between.
// (K <= 512 - the d_model 512 products routed here - with between half a chip and three quarters of big tiles: 16384 x
// 512 x 512 22.8 us as 128 big tiles, 16.2 as 512 small ones; at K = 768 and 144 big tiles the big tile still wins.)
int pick_cfg(int M, int N, int K) {
  if (const char* e = getenv("CARE_TILE_CFG")) return atoi(e);
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  return t256 >= 192 || (t256 >= 128 && K > 512) ? 4412 : 222;
}

template <int EPI>
int dispatch(TArgs& p, hipStream_t st) {
  switch (pick_cfg(p.M, p.N, p.K)) {
    case 42: return launch_tile<4, 2, 1, 3, EPI>(p, st);
    case 423: return launch_tile<4, 2, 1, 3, EPI>(p, st);
    case 422: return launch_tile<4, 2, 1, 2, EPI>(p, st);
    case 223: return launch_tile<2, 2, 1, 3, EPI>(p, st);
    case 2222: return launch_tile<2, 2, 2, 2, EPI>(p, st);   // 256 x 128, 4 waves of 128 x 64
    case 2422: return launch_tile<2, 4, 2, 2, EPI>(p, st);   // 256 x 256, 8 waves of 128 x 64
    case 4412: return launch_tile<4, 4, 1, 2, EPI>(p, st);   // 256 x 256, 16 waves of 64 x 64
    case 4414: return launch_tile<4, 4, 1, 4, EPI, false, 32>(p, st);   // ... K steps of 32, four stages
    case 4413: return launch_tile<4, 4, 1, 3, EPI, false, 32>(p, st);
    case 2224: return launch_tile<2, 2, 1, 4, EPI, false, 32>(p, st);   // 128 x 128, K steps of 32, four stages (64 KB)
    case 2226: return launch_tile<2, 2, 1, 6, EPI, false, 32>(p, st);
    case 122: return launch_tile<1, 2, 1, 2, EPI>(p, st);   // 64 x 128, 2 waves
    case 123: return launch_tile<1, 2, 1, 3, EPI>(p, st);
    case 212: return launch_tile<2, 1, 1, 2, EPI>(p, st);   // 128 x 64, 2 waves
    case 242: return launch_tile<2, 4, 1, 2, EPI>(p, st);   // 128 x 256, 8 waves
    case 2223: return launch_tile<2, 2, 2, 3, EPI>(p, st);
    case 112: return launch_tile<1, 1, 1, 2, EPI>(p, st);   // 64 x 64, ONE wave: few-row products that a 128 x 128 tiling
    case 114: return launch_tile<1, 1, 1, 4, EPI>(p, st);   // leaves on a fraction of the CUs (640 x 512: 20 tiles)
    case 124: return launch_tile<1, 2, 1, 4, EPI>(p, st);
    case 214: return launch_tile<2, 1, 1, 4, EPI>(p, st);
    case 224: return launch_tile<2, 2, 1, 4, EPI>(p, st);   // 128 x 128, four stages of 32 KB
    default: return launch_tile<2, 2, 1, 2, EPI>(p, st);
  }
}

int tile_check(const void* A, int64_t lda, const void* W, int M, int N, int K) {
  if (!A || !W || M <= 0 || N <= 0 || K <= 0) return CARE_EINVAL;
  if (K % 64 != 0) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % 8) != 0) return CARE_EALIGN;
  return 0;
}

}  // namespace

extern "C" int care_gemm_tile(const void* A, int64_t lda, const void* W, const float* bias, void* C0, int64_t ldc0,
                              int c0_dtype, void* C1, int64_t ldc1, int c1_dtype, int n_split, int M, int N, int K,
                              int act, void* stream) {
  int rc = tile_check(A, lda, W, M, N, K);
  if (rc) return rc;
  if (!C0 || n_split <= 0 || n_split > N || (n_split < N && !C1)) return CARE_EINVAL;
  if (n_split % 16 != 0 && n_split != N) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if ((c0_dtype != CARE_F32 && c0_dtype != CARE_BF16) || (C1 && c1_dtype != CARE_F32 && c1_dtype != CARE_BF16))
    return CARE_EDTYPE;
  TArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.ldw = K; p.bias = bias;
  p.C0 = C0; p.ldc0 = ldc0; p.c0_bf16 = c0_dtype == CARE_BF16;
  p.C1 = C1; p.ldc1 = ldc1; p.c1_bf16 = c1_dtype == CARE_BF16;
  p.n_split = n_split; p.M = M; p.N = N; p.K = K; p.act = act;
  return dispatch<EPI_STORE>(p, (hipStream_t)stream);
}

// `batch` independent products in one launch: product b multiplies A + b a_bs ([M, lda]) by W + b w_bs ([N, K] rows of
// length K... leading dimension ldw) into C + b c_bs, bias + b bias_bs.  The per-head projections on either side of the
// absorbed cross-attention for d_model = 1024 (the roles of care_head_expand / care_head_reduce, csrc/heads.hip):
//   expand: A = q + 64 h (K = 64), W = wkt[h] [d, 64], C = qt + d h   -> [rows, heads, d]
//   reduce: A = ct + d h (K = d),  W = W_v rows 64 h .. (ldw = d), C = ctx + 64 h (N = 64), bias = b_v + 64 h
extern "C" int care_gemm_tile_batched(const void* A, int64_t lda, int64_t a_bs, const void* W, int64_t ldw, int64_t w_bs,
                                      const float* bias, int bias_bs, void* C, int64_t ldc, int64_t c_bs, int c_dtype,
                                      int batch, int M, int N, int K, void* stream) {
  int rc = tile_check(A, lda, W, M, N, K);
  if (rc) return rc;
  if (!C || batch <= 0 || batch > 65535) return CARE_EINVAL;
  if (c_dtype != CARE_F32 && c_dtype != CARE_BF16) return CARE_EDTYPE;
  if ((a_bs % 8) || (w_bs % 8) || (ldw % 8) || (c_bs % 8)) return CARE_EALIGN;
  TArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.ldw = ldw; p.bias = bias;
  p.C0 = C; p.ldc0 = ldc; p.c0_bf16 = c_dtype == CARE_BF16; p.n_split = N; p.M = M; p.N = N; p.K = K; p.act = CARE_ACT_NONE;
  p.a_bs = a_bs; p.w_bs = w_bs; p.c_bs = c_bs; p.bias_bs = bias_bs;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 64) {  // 64-column tiles: the A stream is read once
    // *measured* (head reduce, 4096 rows x 16 heads, K = 1024): 128-row tiles / 3 stages 38.9 us, 256 / 3 39.8,
    // 256 / 2 41.5, 128 / 4 56.0
    static const int cfg64 = [] { const char* e = getenv("CARE_TILE_CFG64"); return e ? atoi(e) : 213; }();
    if (cfg64 == 412) return launch_tile<4, 1, 1, 2, EPI_STORE>(p, st, batch);
    if (cfg64 == 413) return launch_tile<4, 1, 1, 3, EPI_STORE>(p, st, batch);
    return launch_tile<2, 1, 1, 3, EPI_STORE>(p, st, batch);
  }
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256) * batch;
  return t256 >= 128 ? launch_tile<4, 4, 1, 2, EPI_STORE>(p, st, batch) : launch_tile<2, 2, 1, 2, EPI_STORE>(p, st, batch);
}

// fp32 [M, K] -> fp16 pieces [M, 2K]: x_hi = fp16(x) | x_lo = fp16(x - x_hi); amax != NULL: of x * the power-of-two scale of
// pow2_scale_exp (exact in fp32)
__global__ void split2_act_kernel(const float* A, int64_t lda, _Float16* out, int M, int K, const unsigned* amax) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 each
  const int kq = K >> 2;
  if (i >= (int64_t)M * kq) return;
  const int r = (int)(i / kq), c = (int)(i % kq) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(A + (int64_t)r * lda + c);
  if (amax) {
    const float sc = pow2_of(pow2_scale_exp(*amax));
    v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
  }
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) { hi[j] = (_Float16)v[j]; lo[j] = (_Float16)(v[j] - (float)hi[j]); }
  _Float16* row = out + (int64_t)r * 2 * K;
  *reinterpret_cast<f16x4*>(row + c) = hi;
  *reinterpret_cast<f16x4*>(row + K + c) = lo;
}

extern "C" int care_split2_act(const float* A, int64_t lda, void* A2, int M, int K, void* stream) {
  if (!A || !A2 || M <= 0 || K <= 0) return CARE_EINVAL;
  if (K % 4 != 0 || lda % 4 != 0 || !care_aligned16(A) || !care_aligned16(A2)) return CARE_EALIGN;
  const int64_t total = (int64_t)M * (K >> 2);
  hipLaunchKernelGGL(split2_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, lda,
                     reinterpret_cast<_Float16*>(A2), M, K, (const unsigned*)nullptr);
  return care_launch_status();
}

// The fp16 pieces of one operand of a scaled split product, straight from the tensor as it lies in memory:
//   src: the operand [rows, K] (ld >= K), or - transposed - its transpose [K, rows] (ld >= rows: dy and x as they lie for
//        dW = dy^T x, W for dx = dy W);
//   out: [slabs][rows][pieces * ks] fp16, slab s = columns s ks .. s ks + ks - 1 of the operand (zeros past K), each row
//        hi | lo (pieces = 2: the A operand) or hi | lo | hi (pieces = 3: the W operand) of x * the power-of-two scale of *amax.
// One 64 x 64 tile per workgroup through LDS, so that both the reads (along the source's contiguous dimension) and the writes
// (along kk) are whole cache lines - round 6's first form made a transposed slab-major fp32 copy first (torch) and split that:
// 3.4 ms of copies per 512-clip training step.
__global__ __launch_bounds__(256) void split_pieces_kernel(const float* src, int64_t ld, int rows, int K, int transposed, int slabs,
                                                           int ks, _Float16* out, int pieces, const unsigned* amax) {
  __shared__ float tile[64][65];
  const int tiles_k = (slabs * ks) >> 6;                       // ks % 64 == 0
  const int tk = blockIdx.x % tiles_k, tr = blockIdx.x / tiles_k;
  const int k0 = tk * 64, r0 = tr * 64;
  const float sc = pow2_of(pow2_scale_exp(*amax));
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 4 rows of 64 per pass
  if (transposed) {  // contiguous along r: tile[k][r]
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int k = k0 + p * 4 + ty, r = r0 + tx;
      tile[p * 4 + ty][tx] = (k < K && r < rows) ? src[(int64_t)k * ld + r] : 0.f;
    }
  } else {           // contiguous along k: tile[r][k]
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int r = r0 + p * 4 + ty, k = k0 + tx;
      tile[p * 4 + ty][tx] = (k < K && r < rows) ? src[(int64_t)r * ld + k] : 0.f;
    }
  }
  __syncthreads();
  const int slab = k0 / ks, kk = k0 - slab * ks + tx;          // a 64-column tile lies inside one slab (ks % 64 == 0)
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int rr = p * 4 + ty, r = r0 + rr;
    if (r >= rows) continue;
    const float x = (transposed ? tile[tx][rr] : tile[rr][tx]) * sc;
    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
    _Float16* o = out + ((int64_t)slab * rows + r) * pieces * ks + kk;
    o[0] = hi;
    o[ks] = lo;
    if (pieces == 3) o[2 * ks] = hi;
  }
}

extern "C" int care_split_pieces(const float* src, int64_t ld, int rows, int K, int transposed, int slabs, int ks, void* out,
                                 int pieces, const void* amax, void* stream) {
  if (!src || !out || !amax || rows <= 0 || K <= 0 || slabs <= 0 || ks <= 0 || (pieces != 2 && pieces != 3)) return CARE_EINVAL;
  if (ks % 64 != 0 || (int64_t)slabs * ks < K || ld < (transposed ? rows : K)) return CARE_ESHAPE;
  const int64_t blocks = (int64_t)((slabs * (int64_t)ks) >> 6) * ((rows + 63) / 64);
  if (blocks > 0x7fffffff) return CARE_ESHAPE;
  hipLaunchKernelGGL(split_pieces_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld, rows, K, transposed, slabs,
                     ks, reinterpret_cast<_Float16*>(out), pieces, reinterpret_cast<const unsigned*>(amax));
  return care_launch_status();
}

// max |x| of an fp32 matrix as a bit pattern (non-negative floats order like their patterns): *slot = max(*slot, ...) - the
// function zeroes the slot first.  NaNs are skipped (their patterns would win every comparison).
__global__ void absmax_kernel(const float* A, int64_t lda, int M, int K, unsigned* slot) {
  const int kq = K >> 2;
  const int64_t total = (int64_t)M * kq;
  const bool dense = lda == K;  // (rows back to back: one linear sweep, no division per element - 65 us per call before)
  unsigned best = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t off = dense ? i * 4 : (i / kq) * lda + (i % kq) * 4;
    const uint4 u = *reinterpret_cast<const uint4*>(A + off);
    const unsigned b0 = u.x & 0x7fffffffu, b1 = u.y & 0x7fffffffu, b2 = u.z & 0x7fffffffu, b3 = u.w & 0x7fffffffu;
    if (b0 <= 0x7f800000u && b0 > best) best = b0;
    if (b1 <= 0x7f800000u && b1 > best) best = b1;
    if (b2 <= 0x7f800000u && b2 > best) best = b2;
    if (b3 <= 0x7f800000u && b3 > best) best = b3;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)best, o, 64);
    if (other > best) best = other;
  }
  // ONE atomic per workgroup (16 K wave-level atomics on one address were most of this kernel: 58 us per call on average)
  __shared__ unsigned wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned b = wmax[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = wmax[w] > b ? wmax[w] : b;
    if (b) atomicMax(slot, b);
  }
}

// (a kernel, not hipMemsetAsync: a memset node of a captured graph replays wrong on ROCm 7.2 - csrc/decode_resident.h, res_zero_kernel)
__global__ void absmax_clear_kernel(unsigned* slot) { *slot = 0u; }

extern "C" int care_absmax(const float* A, int64_t lda, int M, int K, void* slot, void* stream) {
  if (!A || !slot || M <= 0 || K <= 0) return CARE_EINVAL;
  if (K % 4 != 0 || lda % 4 != 0 || !care_aligned16(A)) return CARE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(absmax_clear_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned*>(slot));
  const int64_t total = (int64_t)M * (K >> 2);
  const int64_t want = (total + 2047) / 2048;  // ~8 float4 per thread, at most 2048 workgroups (8 per CU)
  const unsigned blocks = (unsigned)(want < 2048 ? (want > 0 ? want : 1) : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, A, lda, M, K, reinterpret_cast<unsigned*>(slot));
  return care_launch_status();
}

// C = act(A W^T + bias) with fp32-GRADE products on the LDS-tiled kernel: A2 = care_split2_act(A) [M, 2K] (hi | lo),
// W3 = care_split3_weight(W) [N, 3K] (hi | lo | hi); one product over the virtual 3K columns
// a_hi w_hi + a_hi w_lo + a_lo w_hi (the A columns start over after the first K: a_hi, a_hi again, then a_lo),
// v_mfma_f32_16x16x32_f16.  The contract of care_gemm_split3 (csrc/gemm.hip) at about twice its rate, with the
// destinations / activation of care_gemm.
static void split3_args(TArgs& p, const void* A2, const void* W3, int M, int N, int K) {
  p.A = reinterpret_cast<const bf16_t*>(A2); p.lda = 2 * (int64_t)K; p.W = reinterpret_cast<const bf16_t*>(W3); p.ldw = 3 * (int64_t)K;
  p.M = M; p.N = N; p.K = 3 * K; p.a_wrap = K >> 6; p.n_split = N;
}

extern "C" int care_gemm_tile_split3(const void* A2, const void* W3, const float* bias, void* C0, int64_t ldc0, int c0_dtype,
                                     void* C1, int64_t ldc1, int c1_dtype, int n_split, int M, int N, int K, int act,
                                     void* stream) {
  int rc = tile_check(A2, 2 * (int64_t)K, W3, M, N, K);
  if (rc) return rc;
  if (!C0 || n_split <= 0 || n_split > N || (n_split < N && !C1)) return CARE_EINVAL;
  if (n_split % 16 != 0 && n_split != N) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if ((c0_dtype != CARE_F32 && c0_dtype != CARE_BF16) || (C1 && c1_dtype != CARE_F32 && c1_dtype != CARE_BF16))
    return CARE_EDTYPE;
  TArgs p{};
  split3_args(p, A2, W3, M, N, K);
  p.bias = bias; p.C0 = C0; p.ldc0 = ldc0; p.c0_bf16 = c0_dtype == CARE_BF16;
  p.C1 = C1; p.ldc1 = ldc1; p.c1_bf16 = c1_dtype == CARE_BF16; p.n_split = n_split; p.act = act;
  hipStream_t st = (hipStream_t)stream;
  return pick_cfg(M, N, 3 * K) == 4412 ? launch_tile<4, 4, 1, 2, EPI_STORE, true>(p, st) : launch_tile<2, 2, 1, 2, EPI_STORE, true>(p, st);
}

// ... of operands split after a power-of-two pre-scale each (care_absmax -> care_split_pieces):
// C [M, ldc] fp32 = (A2 W3^T) / (scale_a scale_b) + bias.  The training-mode products (care_amd/training.py): gradients of
// 1e-6 .. 1e-3, whose unscaled low pieces would be fp16 denormals, keep the ~2^-22 product error of the inference mode.
// slabs > 1 (products with few output tiles and a long reduction - dW = dy^T x, dx = dy W): the operands come as `slabs`
// consecutive matrices A2 [slabs][M, 2K], W3 [slabs][N, 3K] - K ranges of the real product, split range by range - and slab s
// is multiplied into C + s M ldc; the caller adds the slabs in order (care_strided_sum).  bias must be NULL then.
extern "C" int care_gemm_tile_split3_scaled(const void* A2, const void* W3, const float* bias, float* C, int64_t ldc, int M, int N,
                                            int K, const void* amax_a, const void* amax_b, int slabs, void* stream) {
  int rc = tile_check(A2, 2 * (int64_t)K, W3, M, N, K);
  if (rc) return rc;
  if (!C || !amax_a || !amax_b || slabs < 1 || slabs > 65535 || (slabs > 1 && bias)) return CARE_EINVAL;
  TArgs p{};
  split3_args(p, A2, W3, M, N, K);
  p.bias = bias; p.C0 = C; p.ldc0 = ldc; p.c0_bf16 = 0; p.n_split = N; p.act = CARE_ACT_NONE;
  p.amax_a = reinterpret_cast<const unsigned*>(amax_a); p.amax_b = reinterpret_cast<const unsigned*>(amax_b);
  p.a_bs = (int64_t)M * 2 * K; p.w_bs = (int64_t)N * 3 * K; p.c_bs = (int64_t)M * ldc; p.bias_bs = 0;
  hipStream_t st = (hipStream_t)stream;
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256) * slabs;
  return t256 >= 192 ? launch_tile<4, 4, 1, 2, EPI_STORE, true>(p, st, slabs) : launch_tile<2, 2, 1, 2, EPI_STORE, true>(p, st, slabs);
}

// the fused vocabulary arg-max (care_gemm_tile_argmax) on split products: fp32-grade logits, never written
extern "C" int care_gemm_tile_split3_argmax(const void* A2, const void* W3, float* pmax, int32_t* pidx, float* psum, int M,
                                            int N, int K, void* stream) {
  int rc = tile_check(A2, 2 * (int64_t)K, W3, M, N, K);
  if (rc) return rc;
  if (!pmax || !pidx || !psum) return CARE_EINVAL;
  TArgs p{};
  split3_args(p, A2, W3, M, N, K);
  p.pmax = pmax; p.pidx = pidx; p.psum = psum; p.parts = care_argmax_parts_tile(N);
  hipStream_t st = (hipStream_t)stream;
  return pick_cfg(M, N, 3 * K) == 4412 ? launch_tile<4, 4, 1, 2, EPI_ARGMAX, true>(p, st) : launch_tile<2, 2, 1, 2, EPI_ARGMAX, true>(p, st);
}
extern "C" int care_argmax_parts_tile(int N) { return N > 0 ? (N + 63) / 64 : CARE_EINVAL; }

extern "C" int care_gemm_tile_beam(const void* A, int64_t lda, const void* W, float* pmax, float* psum, float* gmax, int M,
                                   int N, int K, void* stream) {
  int rc = tile_check(A, lda, W, M, N, K);
  if (rc) return rc;
  if (!pmax || !psum || !gmax || !care_aligned16(gmax)) return CARE_EINVAL;
  TArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.ldw = K;
  p.n_split = N; p.M = M; p.N = N; p.K = K;
  p.pmax = pmax; p.psum = psum; p.gmax = gmax; p.parts = care_argmax_parts_tile(N);
  return dispatch<EPI_BEAM>(p, (hipStream_t)stream);
}

extern "C" int care_gemm_tile_argmax(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                                     const int32_t* labels, float* plab, int M, int N, int K, void* stream) {
  int rc = tile_check(A, lda, W, M, N, K);
  if (rc) return rc;
  if (!pmax || !pidx || !psum || ((labels != nullptr) != (plab != nullptr))) return CARE_EINVAL;
  TArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.ldw = K;
  p.n_split = N; p.M = M; p.N = N; p.K = K;
  p.pmax = pmax; p.pidx = pidx; p.psum = psum; p.parts = care_argmax_parts_tile(N);
  p.labels = labels; p.plab = plab;
  return labels ? dispatch<EPI_ARGMAX_LAB>(p, (hipStream_t)stream) : dispatch<EPI_ARGMAX>(p, (hipStream_t)stream);
}

#if CARE_TILE_DBG & 64
extern "C" int care_tile_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tile_stamps), sizeof(tile_stamps)); }
#endif
