// gemm.hip - C = act(A * W^T + bias) on the gfx950 matrix cores.
//
// One templated kernel serves every nn.Linear of the path (see include/care_hip.h).
//   * A is fp32 in HBM.  WT = float : exact f32 MFMA (v_mfma_f32_16x16x4_f32), the parity mode.
//                        WT = bf16  : A is rounded to bf16 while it is staged into LDS,
//                                     v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
//   * Both operand tiles are kept in LDS as rows of 128 bytes (32 f32 / 64 bf16 along K),
//     XOR-swizzled in 16-byte chunks (chunk ^= row & 7) so that the ds_read_b128 of an MFMA
//     fragment (16 rows x one chunk per 16-lane group) is bank-conflict free.
//   * 256 threads = 4 waves in a 2x2 grid; each wave owns a (BM/2)x(BN/2) block of 16x16
//     MFMA tiles.  Global->register->LDS staging, double-buffered: the loads of K-step
//     t+1 are in flight while step t is multiplied.
//   * f32 K order: lane group g = lane>>4 supplies k = 16*kk + 4*g + j to MFMA j of read kk
//     for BOTH operands, so each lane reads 16 contiguous bytes; only the summation order
//     differs from ascending k.
//   * Epilogues: bias + activation + (split) store, or the fused per-row
//     (max, argmax, sum-exp) of the greedy vocabulary projection.
#include <cstdlib>

#include "care_common.h"

namespace {

struct GemmArgs {
  const float* A; int64_t lda;
  const void* W;
  const float* bias;
  void* C0; int64_t ldc0; int c0_bf16;
  void* C1; int64_t ldc1; int c1_bf16;
  int n_split;
  int M, N, K;
  int act;
  float* pmax; int32_t* pidx; float* psum; int parts;
};

template <typename WT> struct KTraits;
template <> struct KTraits<float>  { static constexpr int BK = 32; };
template <> struct KTraits<bf16_t> { static constexpr int BK = 64; };

// GELU is a template parameter of the kernel (erff inlined behind a run-time test for every one of a lane's 64 outputs was
// most of the 128 x 128 kernel's 7800 instructions: see csrc/gemm_tile.hip, tile_epilogue)
template <bool GELU>
__device__ __forceinline__ float apply_act(float v, int act) {
  if constexpr (GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return act == CARE_ACT_RELU ? fmaxf(v, 0.0f) : v;
}

__device__ __forceinline__ void store_out(void* C, int64_t ld, int is_bf16, int row, int col, float v) {
  if (is_bf16) reinterpret_cast<bf16_t*>(C)[(int64_t)row * ld + col] = (bf16_t)v;
  else reinterpret_cast<float*>(C)[(int64_t)row * ld + col] = v;
}

// SPLIT (16-bit W only): the split product of fp32 operands, A = A_hi + A_lo, W = W_hi + W_lo in FP16 pieces,
//   C = A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T  as ONE product over 3K: W is the [N, 3K] concatenation W_hi | W_lo | W_hi
//   (care_split3_weight), the A tile of virtual column block k0 comes from real columns k0 % K, converted to its
//   high piece in the first two thirds and to its low piece in the last; v_mfma_f32_16x16x32_f16.  What is dropped
//   (A_lo W_lo) is ~2^-22 of a product: fp32-grade, at a third of the bf16 rate instead of the exact-f32 MFMA's 1/16.
template <typename WT, int BM, int BN, bool ARGMAX, bool SPLIT = false, bool GELU = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int BK = KTraits<WT>::BK;
  constexpr bool BF = sizeof(WT) == 2;
  static_assert(!SPLIT || BF, "split products take 16-bit weight pieces");
  const int KW = SPLIT ? 3 * p.K : p.K;  // columns of W (and of the virtual product)
  constexpr int MT = BM / 32, NT = BN / 32;
  // A staging: f32 mode 8 16-byte chunks per row, bf16 mode 16 float4 per row (-> 8 bytes each)
  constexpr int A_CH = BF ? 16 : 8;
  constexpr int A_IT = BM * A_CH / 256;
  constexpr int W_IT = BN * 8 / 256;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                    // 2 x BM x 128
  unsigned char* sB = smem + 2 * BM * 128;     // 2 x BN x 128

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM;
  const int n0 = (blockIdx.x % tiles_n) * BN;

  // native vector types: arrays of HIP's struct-based float4/uint4 are not promoted to registers
  // here (the staging then lives in scratch memory and every tile waits for its global loads)
  f32x4 ra[A_IT];
  u32x4 rw[W_IT];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int c = i * 256 + tid;
      int row = c / A_CH, ch = c % A_CH;
      // rows past M are clamped, not branched around: their products are never stored, and a
      // branch per unrolled load would serialise the loads behind vmcnt(0) waits
      const int gr = min(m0 + row, p.M - 1);
      ra[i] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)gr * p.lda + (SPLIT ? k0 % p.K : k0) + ch * 4);
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      int c = i * 256 + tid;
      int row = c >> 3, ch = c & 7;
      const int gn = min(n0 + row, p.N - 1);
      rw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.W) +
                                              ((int64_t)gn * KW + k0) * sizeof(WT) + ch * 16);
    }
  };
  auto store_tile = [&](int buf, int k0) {
    const bool lo_piece = SPLIT && k0 >= 2 * p.K;
    unsigned char* a = sA + buf * BM * 128;
    unsigned char* b = sB + buf * BN * 128;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int c = i * 256 + tid;
      int row = c / A_CH, ch = c % A_CH;
      if constexpr (SPLIT) {  // fp16 pieces, moved as whole 32-bit pairs
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 v;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          f16x2 pr = __builtin_convertvector(f32x2{ra[i][e], ra[i][e + 1]}, f16x2);
          if (lo_piece) pr = __builtin_convertvector(f32x2{ra[i][e] - (float)pr[0], ra[i][e + 1] - (float)pr[1]}, f16x2);
          v[e >> 1] = __builtin_bit_cast(unsigned, pr);
        }
        *reinterpret_cast<u32x2*>(a + row * 128 + ((((ch >> 1) ^ (row & 7))) << 4) + (ch & 1) * 8) = v;
      } else if constexpr (BF) {
        bf16x4 v;
        v[0] = (bf16_t)ra[i][0]; v[1] = (bf16_t)ra[i][1]; v[2] = (bf16_t)ra[i][2]; v[3] = (bf16_t)ra[i][3];
        *reinterpret_cast<bf16x4*>(a + row * 128 + ((((ch >> 1) ^ (row & 7))) << 4) + (ch & 1) * 8) = v;
      } else {
        *reinterpret_cast<f32x4*>(a + row * 128 + ((ch ^ (row & 7)) << 4)) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      int c = i * 256 + tid;
      int row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(b + row * 128 + ((ch ^ (row & 7)) << 4)) = rw[i];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = KW / BK;
  load_tile(0);
  store_tile(0, 0);
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // unconditional prefetch (the last iteration re-loads its own tile): a conditional one keeps
    // the staging registers in scratch memory and exposes the whole global-load latency
    load_tile(min(kt + 1, nk - 1) * BK);
    const unsigned char* a = sA + buf * BM * 128 + (wm * (BM / 2) + fr) * 128;
    const unsigned char* b = sB + buf * BN * 128 + (wn * (BN / 2) + fr) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 4 + fg) ^ (fr & 7)) << 4;
      if constexpr (BF) {
        bf16x8 fa[MT], fb[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(a + m * 16 * 128 + off);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(b + n * 16 * 128 + off);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if constexpr (SPLIT) {
              typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[m]), __builtin_bit_cast(f16x8, fb[n]),
                                                                 acc[m][n], 0, 0, 0);
            } else {
              acc[m][n] = care_mfma_16x16x32_h16(fa[m], fb[n], acc[m][n], 0, 0, 0);
            }
      } else {
        f32x4 fa[MT], fb[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const f32x4*>(a + m * 16 * 128 + off);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const f32x4*>(b + n * 16 * 128 + off);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][j], fb[n][j], acc[m][n], 0, 0, 0);
      }
    }
    store_tile(buf ^ 1, min(kt + 1, nk - 1) * BK);
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  const int row_base = m0 + wm * (BM / 2) + fg * 4;
  const int col_base = n0 + wn * (BN / 2) + fr;
  if constexpr (!ARGMAX) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = col_base + n * 16;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.0f;
      const bool second = col >= p.n_split;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row_base + m * 16 + j;
          if (row >= p.M) continue;
          float v = apply_act<GELU>(acc[m][n][j] + bv, p.act);
          if (!second) store_out(p.C0, p.ldc0, p.c0_bf16, row, col, v);
          else store_out(p.C1, p.ldc1, p.c1_bf16, row, col - p.n_split, v);
        }
    }
  } else {
    // per row of this wave's (BM/2) x (BN/2) block: max, first argmax, sum exp(x - max)
    const int part = (n0 / BN) * 2 + wn;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int col = col_base + n * 16;
          const float v = col < p.N ? acc[m][n][j] : -INFINITY;
          if (v > best) { best = v; bi = col; }   // columns ascend with n: first max kept
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const float ov = __shfl_xor(best, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        float s = 0.0f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int col = col_base + n * 16;
          if (col < p.N) s += __expf(acc[m][n][j] - best);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
        const int row = row_base + m * 16 + j;
        if (fr == 0 && row < p.M) {
          const int64_t o = (int64_t)row * p.parts + part;
          p.pmax[o] = best; p.pidx[o] = bi; p.psum[o] = s;
        }
      }
  }
}

template <typename WT, int BM, int BN, bool ARGMAX, bool SPLIT = false, bool GELU = false>
int launch(const GemmArgs& p, hipStream_t st) {
  if constexpr (!ARGMAX && !GELU)
    if (p.act == CARE_ACT_GELU) return launch<WT, BM, BN, ARGMAX, SPLIT, true>(p, st);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t lds = 2 * (BM + BN) * 128;
  hipLaunchKernelGGL((gemm_kernel<WT, BM, BN, ARGMAX, SPLIT, GELU>), dim3(tiles), dim3(256), lds, st, p);
  return care_launch_status();
}

// Few output tiles and exact f32 (small batches: the concept head's [clips <= 128, 2048] x [2048, 500] and [clips, 512] x
// [512, 512]; every Linear of the fp32 mode at a few rows): gemm_kernel gives a wave 2 x 2 tiles of 16 x 16 - four chains of
// K / 4 dependent MFMAs through one SIMD's matrix pipe, a barrier per 32 columns - on 8 - 16 workgroups: 50 us at 1 clip,
// 53 us at 128 for the concept scores (*measured* rocprofv3 round 6, twice per pass).  Here a wave - a workgroup of its own, so
// the tiles spread over the chip - owns ONE 16 x 16 tile and hands the matrix core the SAME operands in the SAME order straight
// from registers - a lane's 16 contiguous bytes of its row per 16 K columns, element j to MFMA j: the f32 K order of the header -
// so the outputs are BIT-IDENTICAL to gemm_kernel<float>'s (tests/test_gpu_kernels.py) and the one chain of K / 4 MFMAs is the
// whole critical path; no LDS, no barrier, 8 blocks of 16 columns in flight per operand (4 / 2 when K / 16 is no multiple of 8).
// *Measured* (tools/few_tiles_probe.py, us, this kernel / the LDS-tiled one): 1 x 500 x 2048 11.8 / 50.3, 128 x 500 x 2048 12.9 /
// 52.8, 128 x 512 x 512 5.2 / 17.0, 256 x 500 x 2048 17.9 / 53.0; 4 blocks in flight the same, 2: 18 - 22; four waves per
// workgroup (64 CUs instead of 256): 20 - 32.
constexpr long FEW_MAX_TILES = 512;   // 16 x 16 tiles (each streams 2 x 16 x K floats from L2: beyond this the LDS tiles' reuse wins)

template <bool GELU, int U>   // U blocks of 16 columns in flight per operand (K / 16 % U == 0)
__global__ __launch_bounds__(64) void gemm_few_tiles_f32_kernel(GemmArgs p) {
  const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
  const int tiles_n = (p.N + 15) / 16;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  // (rows / columns past M / N are clamped, like gemm_kernel: their products are never stored)
  const float* a = p.A + (int64_t)min(tm * 16 + fr, p.M - 1) * p.lda + fg * 4;
  const float* b = reinterpret_cast<const float*>(p.W) + (int64_t)min(tn * 16 + fr, p.N - 1) * p.K + fg * 4;
  const int nb = p.K / 16;
  f32x4 ra[U], rb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    ra[u] = *reinterpret_cast<const f32x4*>(a + u * 16);
    rb[u] = *reinterpret_cast<const f32x4*>(b + u * 16);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < nb; kb += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][j], rb[u][j], acc, 0, 0, 0);
      const int nxt = min(kb + u + U, nb - 1);   // unconditional refill of the slot just used (the tail re-reads the last block)
      ra[u] = *reinterpret_cast<const f32x4*>(a + nxt * 16);
      rb[u] = *reinterpret_cast<const f32x4*>(b + nxt * 16);
    }
  }
  const int col = tn * 16 + fr;
  if (col >= p.N) return;
  const float bv = p.bias ? p.bias[col] : 0.0f;
  const bool second = col >= p.n_split;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = tm * 16 + fg * 4 + j;
    if (row >= p.M) continue;
    const float v = apply_act<GELU>(acc[j] + bv, p.act);
    if (!second) store_out(p.C0, p.ldc0, p.c0_bf16, row, col, v);
    else store_out(p.C1, p.ldc1, p.c1_bf16, row, col - p.n_split, v);
  }
}

// W [N, K] fp32 -> [N, 3K] fp16 pieces  W_hi | W_lo | W_hi  (the operand of care_gemm_split3)
__global__ void split3_weight_kernel(const float* W, unsigned short* out, int N, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * K) return;
  const int n = (int)(i / K), k = (int)(i % K);
  const _Float16 hi = (_Float16)W[i];
  const _Float16 lo = (_Float16)(W[i] - (float)hi);
  unsigned short* row = out + (int64_t)n * 3 * K;
  row[k] = __builtin_bit_cast(unsigned short, hi);
  row[K + k] = __builtin_bit_cast(unsigned short, lo);
  row[2 * K + k] = __builtin_bit_cast(unsigned short, hi);
}

int check_common(const float* A, int64_t lda, const void* W, int wdtype, int M, int N, int K) {
  if (!A || !W || M <= 0 || N <= 0 || K <= 0) return CARE_EINVAL;
  if (wdtype != CARE_F32 && wdtype != CARE_BF16) return CARE_EDTYPE;
  if (K % (wdtype == CARE_BF16 ? 64 : 32) != 0) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % 4) != 0) return CARE_EALIGN;
  return 0;
}

}  // namespace

extern "C" int care_gemm(const float* A, int64_t lda, const void* W, int wdtype, const float* bias,
                         void* C0, int64_t ldc0, int c0_dtype, void* C1, int64_t ldc1, int c1_dtype,
                         int n_split, int M, int N, int K, int act, void* stream) {
  int rc = check_common(A, lda, W, wdtype, M, N, K);
  if (rc) return rc;
  if (!C0 || n_split <= 0 || n_split > N || (n_split < N && !C1)) return CARE_EINVAL;
  if (n_split % 16 != 0 && n_split != N) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if ((c0_dtype != CARE_F32 && c0_dtype != CARE_BF16) || (C1 && c1_dtype != CARE_F32 && c1_dtype != CARE_BF16))
    return CARE_EDTYPE;
  GemmArgs p{};
  p.A = A; p.lda = lda; p.W = W; p.bias = bias;
  p.C0 = C0; p.ldc0 = ldc0; p.c0_bf16 = c0_dtype == CARE_BF16;
  p.C1 = C1; p.ldc1 = ldc1; p.c1_bf16 = c1_dtype == CARE_BF16;
  p.n_split = n_split; p.M = M; p.N = N; p.K = K; p.act = act;
  hipStream_t st = (hipStream_t)stream;
  // 128 x 128 tiles only when they fill the chip at least twice over (or exactly once, bf16):
  // *measured* (M = 4096, K = 1024 bf16 / 512 f32) the 64 x 64 tile wins below that - 256 big
  // tiles leave half the 512 (bf16, 2 per CU) / 768 (f32, 3 per CU) workgroup slots empty
  // (43.5 vs 29.5 us), 768 run as one and a half waves (77.9 vs 67.7 us).
  const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long slots = wdtype == CARE_BF16 ? 512 : 768;
  bool big = big_tiles >= 2 * slots || (wdtype == CARE_BF16 && big_tiles >= slots && big_tiles % slots == 0);
  if (const char* e = getenv("CARE_GEMM_TILE")) big = atoi(e) >= 128;  // tuning override
  if (wdtype == CARE_BF16)
    return big ? launch<bf16_t, 128, 128, false>(p, st) : launch<bf16_t, 64, 64, false>(p, st);
  // few 16 x 16 tiles: one per wave, the same bits (CARE_GEMM_FEW_TILES=0: the LDS-tiled kernel at every size - the tests' reference)
  const long t16 = (long)((M + 15) / 16) * ((N + 15) / 16);
  const char* few = getenv("CARE_GEMM_FEW_TILES");
  if (t16 <= FEW_MAX_TILES && !(few && atoi(few) == 0)) {
    const unsigned tiles = (unsigned)t16;
    const bool gelu = act == CARE_ACT_GELU;
#define FEW_LAUNCH(UU) do { if (gelu) hipLaunchKernelGGL((gemm_few_tiles_f32_kernel<true, UU>), dim3(tiles), dim3(64), 0, st, p); \
                            else hipLaunchKernelGGL((gemm_few_tiles_f32_kernel<false, UU>), dim3(tiles), dim3(64), 0, st, p); } while (0)
    if (K % 128 == 0) FEW_LAUNCH(8); else if (K % 64 == 0) FEW_LAUNCH(4); else FEW_LAUNCH(2);   // (K % 32 == 0: check_common)
#undef FEW_LAUNCH
    return care_launch_status();
  }
  return big ? launch<float, 128, 128, false>(p, st) : launch<float, 64, 64, false>(p, st);
}

extern "C" int care_split3_weight(const float* W, void* W3, int N, int K, void* stream) {
  if (!W || !W3 || N <= 0 || K <= 0) return CARE_EINVAL;
  const int64_t total = (int64_t)N * K;
  hipLaunchKernelGGL(split3_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                     reinterpret_cast<unsigned short*>(W3), N, K);
  return care_launch_status();
}

// C = A W^T + bias with fp32-grade products at a third of the bf16 rate (see gemm_kernel, SPLIT): A [M, K] fp32,
// W3 = care_split3_weight(W) [N, 3K] fp16 pieces, C fp32 [M, ldc].  K % 64 == 0.
extern "C" int care_gemm_split3(const float* A, int64_t lda, const void* W3, const float* bias, float* C, int64_t ldc,
                                int M, int N, int K, void* stream) {
  int rc = check_common(A, lda, W3, CARE_BF16, M, N, K);
  if (rc) return rc;
  if (!C) return CARE_EINVAL;
  GemmArgs p{};
  p.A = A; p.lda = lda; p.W = W3; p.bias = bias;
  p.C0 = C; p.ldc0 = ldc; p.c0_bf16 = 0; p.C1 = nullptr; p.n_split = N; p.M = M; p.N = N; p.K = K; p.act = CARE_ACT_NONE;
  hipStream_t st = (hipStream_t)stream;
  const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  return big_tiles >= 512 ? launch<bf16_t, 128, 128, false, true>(p, st) : launch<bf16_t, 64, 64, false, true>(p, st);
}

extern "C" int care_argmax_parts(int N) { return N > 0 ? 2 * ((N + 127) / 128) : CARE_EINVAL; }

extern "C" int care_gemm_argmax(const float* A, int64_t lda, const void* W, int wdtype, float* pmax,
                                int32_t* pidx, float* psum, int M, int N, int K, void* stream) {
  int rc = check_common(A, lda, W, wdtype, M, N, K);
  if (rc) return rc;
  if (!pmax || !pidx || !psum) return CARE_EINVAL;
  GemmArgs p{};
  p.A = A; p.lda = lda; p.W = W; p.M = M; p.N = N; p.K = K; p.n_split = N;
  p.pmax = pmax; p.pidx = pidx; p.psum = psum; p.parts = care_argmax_parts(N);
  hipStream_t st = (hipStream_t)stream;
  const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  const bool big = big_tiles >= 192;
  if (wdtype == CARE_BF16)
    return big ? launch<bf16_t, 128, 128, true>(p, st) : launch<bf16_t, 64, 128, true>(p, st);
  return big ? launch<float, 128, 128, true>(p, st) : launch<float, 64, 128, true>(p, st);
}
