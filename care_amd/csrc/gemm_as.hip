// gemm_as.hip - "A-stationary" bf16 GEMM for the K = d_model shapes of the decoder (gfx950).
//
// Almost every GEMM of the path has K = d_model = 512 (QKV, Wq, Wo, FFN1, cross-K/V, vocab)
// and a tall-skinny or huge-N shape.  A classic LDS-tiled kernel re-stages A for every
// N-tile and synchronises every k-step; at K = 512 that is all prologue.  Here instead:
//
//   * a workgroup owns a panel of 128 rows (4 waves x 32 rows).  Each wave loads the MFMA
//     A-fragments of its 32 rows for a whole K-chunk of 512 ONCE, straight from global memory
//     into 128 VGPRs (all 32 loads in flight at once), and keeps them there;
//   * W is streamed in tiles of 32 output columns x 512 k (32 KiB) through a double-buffered
//     LDS ring with LDS-DMA (global_load_lds, 16 B per lane, one 1-KiB W row per
//     wave-instruction).  The LDS image is lane-linear, so the bank-conflict swizzle
//     (16-byte chunk ^= row & 15) is applied to the per-lane SOURCE address and again on the
//     ds_read_b128 of the B fragment (conflict-free: see the group analysis in DESIGN.md);
//   * one barrier per W tile (64 MFMA 16x16x32 per wave between barriers); the next tile's
//     DMA is issued right after the barrier and lands during the MFMAs;
//   * block -> (panel, column range) mapping is XCD-aware: the 8 XCDs each take 1/8 of the
//     column ranges and walk the panels, so every XCD L2 streams its slice of W once.
//
// MODES  STREAM_STORE : K <= 512, any N: bias + activation + (split) store per tile.
//        STREAM_ARGMAX: K <= 512: running per-row (max, argmax, sum-exp) over the block's
//                       column range (greedy vocabulary projection; logits never stored).
//        MULTI_STORE  : K > 512 (FFN2, wide feature embedders): 4 tiles (128 columns) per
//                       block with accumulators persistent over the K-chunks.
// A may be bf16 (no conversion) or fp32 (rounded to bf16 on load, exactly what the generic
// kernel does while staging).  W is bf16 [N, K].  Requires K % 128 == 0.
#include "care_common.h"

namespace {

enum { STREAM_STORE = 0, STREAM_ARGMAX = 1, MULTI_STORE = 2 };

struct AsArgs {
  const void* A; int64_t lda;
  const bf16_t* W;
  const float* bias;
  void* C0; int64_t ldc0; int c0_bf16;
  void* C1; int64_t ldc1; int c1_bf16;
  int n_split, M, N, K, act;
  int panels, ns;        // grid decomposition
  int kslices;           // split-K: slice s multiplies K range [512 s, 512 s + 512) into slab s of C0
  int64_t ldw;           // row stride of W in elements (== full K)
  int64_t slab_stride;   // elements between consecutive fp32 slabs of C0
  float* pmax; int32_t* pidx; float* psum;
};

__device__ __forceinline__ float as_act(float v, int act) {
  if (act == CARE_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == CARE_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

template <typename AT>
__device__ __forceinline__ bf16x8 load_a_frag(const AT* p);
template <>
__device__ __forceinline__ bf16x8 load_a_frag<bf16_t>(const bf16_t* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
template <>
__device__ __forceinline__ bf16x8 load_a_frag<float>(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  bf16x8 v;
  v[0] = (bf16_t)a.x; v[1] = (bf16_t)a.y; v[2] = (bf16_t)a.z; v[3] = (bf16_t)a.w;
  v[4] = (bf16_t)b.x; v[5] = (bf16_t)b.y; v[6] = (bf16_t)b.z; v[7] = (bf16_t)b.w;
  return v;
}

// FULL: K % 512 == 0, every K-chunk has all 16 k-steps.  The unrolled loads/MFMAs then carry
// no run-time predicate: a predicate (even wave-uniform) makes hipcc branch around every
// load and wait vmcnt(0) per element - 32 dependent round trips instead of 32 loads in flight.
template <typename AT, int MODE, bool FULL>
__global__ __launch_bounds__(256, 2) void gemm_as_kernel(AsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x 32 KiB W tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- XCD-aware block decomposition (blocks b and b+8 share an XCD)
  const int per_slice = (MODE == MULTI_STORE) ? (int)gridDim.x : p.panels * p.ns;
  const int slice = blockIdx.x / per_slice, bslot = blockIdx.x % per_slice;
  const int xcd = bslot & 7, idx = bslot >> 3;
  int panel, ns;
  if (MODE == MULTI_STORE) {  // blocks that share an A panel are neighbours on one XCD
    ns = idx % p.ns;
    panel = (idx / p.ns) * 8 + xcd;
  } else {                    // blocks that share a W column range are neighbours on one XCD
    panel = idx % p.panels;
    ns = (idx / p.panels) * 8 + xcd;
  }
  if (panel >= p.panels || ns >= p.ns) return;
  const int tiles_total = (p.N + 31) >> 5;
  int t0, t1;
  if (MODE == MULTI_STORE) { t0 = ns * 4; t1 = min(t0 + 4, tiles_total); }
  else {
    const int tpb = (tiles_total + p.ns - 1) / p.ns;
    t0 = ns * tpb; t1 = min(t0 + tpb, tiles_total);
  }
  const int m0 = panel * 128 + wave * 32;
  if (t0 >= t1) {  // empty column range (ns is rounded up to a multiple of 8)
    if (MODE == STREAM_ARGMAX && lane < 32 && m0 + lane < p.M) {
      const int64_t o = (int64_t)(m0 + lane) * p.ns + ns;
      p.pmax[o] = -INFINITY; p.pidx[o] = 0x7fffffff; p.psum[o] = 0.f;
    }
    return;
  }

  const AT* Ap = reinterpret_cast<const AT*>(p.A) + slice * 512;
  const bf16_t* Wp = p.W + slice * 512;
  const float* biasp = slice == 0 ? p.bias : nullptr;
  int arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) arow[mt] = min(m0 + mt * 16 + fr, p.M - 1);

  bf16x8 a[2][16];
  auto load_a = [&](int kc) {
    const int ksn = FULL ? 16 : min(16, (p.K - kc * 512) >> 5);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (FULL || ks < ksn) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          a[mt][ks] = load_a_frag<AT>(Ap + (int64_t)arow[mt] * p.lda + kc * 512 + ks * 32 + fg * 8);
      }
  };

  // stage W tile `tile`, K-chunk kc into LDS buffer `buf`: wave w copies rows 8w..8w+7,
  // one 1-KiB row per LDS-DMA instruction; lane = chunk slot, source chunk = slot ^ (row & 15)
  auto stage = [&](int tile, int kc, int buf) {
    const int kbytes = FULL ? 1024 : min(512, p.K - kc * 512) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wave * 8 + i;
      const int n = min(tile * 32 + row, p.N - 1);
      const int src_chunk = lane ^ (row & 15);
      if (FULL || src_chunk * 16 < kbytes) {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(Wp + (int64_t)n * p.ldw + kc * 512) + src_chunk * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + buf * 32768 + row * 1024),
                                         16, 0, 0);
      }
    }
  };

  f32x4 acc[MODE == MULTI_STORE ? 4 : 1][2][2];
  auto zero_acc = [&](int q) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[q][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // B fragments are read one k-step ahead of the MFMAs that use them, so the LDS latency
  // is covered by the previous step's four MFMAs instead of being exposed before each group.
  auto compute = [&](int buf, int kc, int q) {
    const int ksn = FULL ? 16 : min(16, (p.K - kc * 512) >> 5);
    const unsigned char* b0 = smem + buf * 32768 + fr * 1024;
    bf16x8 fb[2][2];
    fb[0][0] = *reinterpret_cast<const bf16x8*>(b0 + ((fg ^ fr) << 4));
    fb[0][1] = *reinterpret_cast<const bf16x8*>(b0 + 16 * 1024 + ((fg ^ fr) << 4));
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (FULL || ks < ksn) {
        if (ks + 1 < 16 && (FULL || ks + 1 < ksn)) {
          const int off = (((ks + 1) * 4 + fg) ^ fr) << 4;
          fb[(ks + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(b0 + off);
          fb[(ks + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(b0 + 16 * 1024 + off);
        }
        acc[q][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ks], fb[ks & 1][0], acc[q][0][0], 0, 0, 0);
        acc[q][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][ks], fb[ks & 1][0], acc[q][1][0], 0, 0, 0);
        acc[q][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ks], fb[ks & 1][1], acc[q][0][1], 0, 0, 0);
        acc[q][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][ks], fb[ks & 1][1], acc[q][1][1], 0, 0, 0);
      }
  };
  // Epilogue.  Everything that selects a destination is wave-uniform (a 16-column MFMA tile
  // never straddles n_split), the bias comes from LDS (a global load here would make hipcc
  // drain the in-flight LDS-DMA of the next tile with vmcnt(0)), and interior tiles take a
  // guard-free path.
  const float* sbias = reinterpret_cast<const float*>(smem + 65536);
  const int bias0 = t0 * 32;
  auto store_tile = [&](int tile, int q) {
    const bool interior = (m0 + 32 <= p.M) && (tile * 32 + 32 <= p.N);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int cbase = tile * 32 + nt * 16;
      if (cbase >= p.N) continue;
      const bool second = cbase >= p.n_split;
      unsigned char* C = reinterpret_cast<unsigned char*>(second ? p.C1 : p.C0) + (int64_t)slice * p.slab_stride * 4;
      const int64_t ld = second ? p.ldc1 : p.ldc0;
      const bool isb = (second ? p.c1_bf16 : p.c0_bf16) != 0;
      const int col = cbase + fr;
      const int cc = col - (second ? p.n_split : 0);
      const float bv = biasp ? sbias[col - bias0] : 0.0f;
      float v[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[mt][j] = acc[q][mt][nt][j] + bv;
      if (p.act == CARE_ACT_RELU) {  // one uniform branch per tile, not one per element
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[mt][j] = fmaxf(v[mt][j], 0.0f);
      } else if (p.act == CARE_ACT_GELU) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[mt][j] = as_act(v[mt][j], CARE_ACT_GELU);
      }
      const int64_t rb = (int64_t)(m0 + fg * 4) * ld + cc;
      if (interior) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t o = rb + (int64_t)(mt * 16 + j) * ld;
            if (isb) reinterpret_cast<bf16_t*>(C)[o] = (bf16_t)v[mt][j];
            else reinterpret_cast<float*>(C)[o] = v[mt][j];
          }
      } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = m0 + mt * 16 + fg * 4 + j;
            if (row < p.M && col < p.N) {
              const int64_t o = rb + (int64_t)(mt * 16 + j) * ld;
              if (isb) reinterpret_cast<bf16_t*>(C)[o] = (bf16_t)v[mt][j];
              else reinterpret_cast<float*>(C)[o] = v[mt][j];
            }
          }
      }
    }
  };
  if (MODE != STREAM_ARGMAX && biasp) {  // this block's bias slice -> LDS (<= 2048 columns)
    float* sb = reinterpret_cast<float*>(smem + 65536);
    const int nb = min((t1 - t0) * 32, p.N - bias0);
    for (int i = tid; i < nb; i += 256) sb[i] = biasp[bias0 + i];
  }

  if constexpr (MODE == MULTI_STORE) {
    const int nkc = (p.K + 511) / 512, ntl = t1 - t0, total = nkc * ntl;
#pragma unroll
    for (int q = 0; q < 4; ++q) zero_acc(q);
    stage(t0, 0, 0);
    int cur = 0;
    for (int kc = 0; kc < nkc; ++kc) {
      load_a(kc);
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        if (tl < ntl) {
          __syncthreads();
          const int nxt = kc * ntl + tl + 1;
          if (nxt < total) stage(t0 + nxt % ntl, nxt / ntl, cur ^ 1);
          compute(cur, kc, tl);
          cur ^= 1;
        }
      }
    }
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
      if (tl < ntl) store_tile(t0 + tl, tl);
  } else {
    load_a(0);
    // running (max, argmax, sum-exp) of the 8 rows this lane sees, over its column residue
    float rm[8], rs[8];
    int ri[8];
    if constexpr (MODE == STREAM_ARGMAX) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { rm[i] = -INFINITY; rs[i] = 0.f; ri[i] = 0x7fffffff; }
    }
    stage(t0, 0, 0);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
      __syncthreads();
      if (t + 1 < t1) stage(t + 1, 0, cur ^ 1);
      zero_acc(0);
      compute(cur, 0, 0);
      if constexpr (MODE == STREAM_STORE) {
        store_tile(t, 0);
      } else {
        const int c0 = t * 32 + fr, c1 = c0 + 16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = mt * 4 + j;
            const float v0 = c0 < p.N ? acc[0][mt][0][j] : -INFINITY;
            const float v1 = c1 < p.N ? acc[0][mt][1][j] : -INFINITY;
            const float tm = fmaxf(v0, v1);
            const int ti = v1 > v0 ? c1 : c0;
            const float mn = fmaxf(rm[i], tm);
            rs[i] = rs[i] * __expf(rm[i] - mn) + __expf(v0 - mn) + __expf(v1 - mn);
            ri[i] = tm > rm[i] ? ti : ri[i];
            rm[i] = mn;
          }
      }
      cur ^= 1;
    }
    if constexpr (MODE == STREAM_ARGMAX) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float m = rm[i], s = rs[i];
        int id = ri[i];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
          const int oi = __shfl_xor(id, o, 64);
          const float mn = fmaxf(m, om);
          s = s * __expf(m - mn) + os * __expf(om - mn);
          if (om > m || (om == m && oi < id)) id = oi;
          m = mn;
        }
        const int row = m0 + (i >> 2) * 16 + fg * 4 + (i & 3);
        if (fr == 0 && row < p.M) {
          const int64_t o = (int64_t)row * p.ns + ns;
          p.pmax[o] = m; p.pidx[o] = id; p.psum[o] = s;
        }
      }
    }
  }
}

template <typename AT, int MODE>
int launch_as(const AsArgs& p, int blocks, hipStream_t st) {
  if (p.K % 512 == 0) hipLaunchKernelGGL((gemm_as_kernel<AT, MODE, true>), dim3(blocks), dim3(256), 65536 + 8192, st, p);
  else hipLaunchKernelGGL((gemm_as_kernel<AT, MODE, false>), dim3(blocks), dim3(256), 65536 + 8192, st, p);
  return care_launch_status();
}

int as_check(const void* A, int64_t lda, int a_dtype, const void* W, int M, int N, int K) {
  if (!A || !W || M <= 0 || N <= 0 || K <= 0) return CARE_EINVAL;
  if (a_dtype != CARE_F32 && a_dtype != CARE_BF16) return CARE_EDTYPE;
  if (K % 128 != 0) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % 8) != 0) return CARE_EALIGN;
  return 0;
}

// number of column ranges: a multiple of 8 (one share per XCD) giving >= ~2 blocks per CU
int pick_ns(int panels, int tiles_total) {
  int ns = 8;
  while (ns < tiles_total && (long)panels * ns < 512) ns += 8;
  return ns;
}

}  // namespace

extern "C" int care_gemm_bf16(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias, void* C0,
                              int64_t ldc0, int c0_dtype, void* C1, int64_t ldc1, int c1_dtype, int n_split, int M,
                              int N, int K, int act, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!C0 || n_split <= 0 || n_split > N || (n_split < N && !C1)) return CARE_EINVAL;
  if (n_split % 16 != 0 && n_split != N) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias;
  p.C0 = C0; p.ldc0 = ldc0; p.c0_bf16 = c0_dtype == CARE_BF16;
  p.C1 = C1; p.ldc1 = ldc1; p.c1_bf16 = c1_dtype == CARE_BF16;
  p.n_split = n_split; p.M = M; p.N = N; p.K = K; p.act = act;
  p.panels = (M + 127) / 128;
  p.kslices = 1; p.ldw = K; p.slab_stride = 0;
  const int tiles_total = (N + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  if (K > 512) {
    p.ns = (tiles_total + 3) / 4;
    const int blocks = ((p.panels + 7) / 8) * 8 * p.ns;
    return a_dtype == CARE_BF16 ? launch_as<bf16_t, MULTI_STORE>(p, blocks, st)
                                : launch_as<float, MULTI_STORE>(p, blocks, st);
  }
  p.ns = pick_ns(p.panels, tiles_total);
  while (bias && (tiles_total + p.ns - 1) / p.ns > 64) p.ns += 8;  // bias slice must fit its LDS region
  const int blocks = p.panels * p.ns;
  return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_STORE>(p, blocks, st)
                              : launch_as<float, STREAM_STORE>(p, blocks, st);
}

extern "C" int care_argmax_parts_bf16(int M, int N) {
  if (M <= 0 || N <= 0) return CARE_EINVAL;
  return pick_ns((M + 127) / 128, (N + 31) / 32);
}

extern "C" int care_gemm_argmax_bf16(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                                     int32_t* pidx, float* psum, int M, int N, int K, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!pmax || !pidx || !psum) return CARE_EINVAL;
  if (K > 512) return CARE_ESHAPE;
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.M = M; p.N = N; p.K = K; p.n_split = N;
  p.kslices = 1; p.ldw = K;
  p.pmax = pmax; p.pidx = pidx; p.psum = psum;
  p.panels = (M + 127) / 128;
  p.ns = pick_ns(p.panels, (N + 31) / 32);
  const int blocks = p.panels * p.ns;
  hipStream_t st = (hipStream_t)stream;
  return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_ARGMAX>(p, blocks, st)
                              : launch_as<float, STREAM_ARGMAX>(p, blocks, st);
}

// Split-K variant for K > 512 at small M (FFN dense2 of a decode step): K/512 slices, slice s
// writes its partial product (bias in slice 0) to fp32 slab s = C + s * slab_stride.  The
// consumer (care_add_ln with nslab = K/512) sums the slabs, so no separate reduction pass.
extern "C" int care_gemm_bf16_splitk(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias,
                                     float* C, int64_t ldc, int64_t slab_stride, int M, int N, int K, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!C || K % 512 != 0 || K < 1024) return C ? CARE_ESHAPE : CARE_EINVAL;
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias;
  p.C0 = C; p.ldc0 = ldc; p.c0_bf16 = 0; p.n_split = N; p.M = M; p.N = N; p.K = 512; p.act = CARE_ACT_NONE;
  p.panels = (M + 127) / 128;
  p.kslices = K / 512; p.ldw = K; p.slab_stride = slab_stride;
  const int tiles_total = (N + 31) / 32;
  p.ns = pick_ns(p.panels * p.kslices, tiles_total);
  while (bias && (tiles_total + p.ns - 1) / p.ns > 64) p.ns += 8;
  const int blocks = p.panels * p.ns * p.kslices;
  hipStream_t st = (hipStream_t)stream;
  return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_STORE>(p, blocks, st)
                              : launch_as<float, STREAM_STORE>(p, blocks, st);
}
