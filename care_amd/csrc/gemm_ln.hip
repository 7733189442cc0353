// gemm_ln.hip - out = LayerNorm(A W^T + bias + res [+ pos]) * gamma + beta for N = d_model = 512.
//
// One workgroup owns FULL output rows (64 or 128 rows x all 512 columns), so the post-LN
// epilogues of the reference (Linear -> LayerNorm in the Embedder, dense -> +residual ->
// LayerNorm in MultiHeadAttention and PositionwiseFeedForward) are fused into the GEMM:
// no intermediate [M, 512] fp32 round trip, no separate LayerNorm launch, no split-K slabs.
//
//   * both operands stream through a 3-slot LDS ring in K steps of 32, filled by LDS-DMA
//     (global_load_lds, 16 B per lane) two steps ahead; A may be fp32 (raw features: staged as
//     fp32, rounded to bf16 when the fragment is read - no register staging) or bf16;
//   * LDS images are lane-linear, so the bank-conflict swizzles are applied to the DMA SOURCE
//     address and again on the ds_read_b128: 64-byte rows (bf16 A, W): chunk ^= (row & 8) >> 2;
//     128-byte rows (fp32 A): chunk ^= table[(row & 15) >> 1] - both brute-forced against the
//     real 16-lane grouping of ds_read_b128 (tools/lds_swizzle_search.py);
//   * waves: RG row groups x 4 column groups; a wave owns 64 rows x 128 columns = 4 x 8 MFMA
//     tiles (128 accumulator VGPRs), so every B fragment feeds 4 MFMAs and every A fragment 8;
//   * waits are counted (vmcnt(N) = the DMA instructions of the one younger stage) with a raw
//     s_barrier per K step;
//   * epilogue: the accumulators (swapped operand order: a lane holds 4 consecutive columns of a
//     row) are parked in LDS and finished row-wise, one wave per row, so every residual / bias /
//     gamma / beta load and both stores are coalesced and the statistics are wave shuffles.
//
// Tried and dropped (round 1): an A-stationary variant with 16 rows per wave and the accumulators
// of all 512 columns in registers (W streamed once per 64-row block through 16-KiB DMA tiles,
// LayerNorm finished in registers).  One MFMA per 1-KiB B-fragment read makes it LDS-read bound:
// 43.7 us (K = 512) / 103.6 us (K = 2048) at M = 16384 against 37 / 75 us here, where a wave's
// 64 x 128 tile feeds 32 MFMAs from 12 fragment reads.  With 32 rows per wave (256 accumulator
// registers in AGPRs, one wave per SIMD) for the embedder shape (M = 458752, fp32 features) it is
// not LDS-bound any more and still loses: 2.41 ms (K = 2048) / 0.90 ms (K = 512) against 2.09 / 0.80.
#include <cstdlib>
#include <type_traits>

#include "care_common.h"

namespace {

constexpr int LN_N = 512;

struct LnArgs {
  const void* A; int64_t lda;
  const bf16_t* W;      // [512, K] bf16
  const float* bias; const float* res; int64_t ldres; const float* pos;
  const float* gamma; const float* beta; float eps;
  float* out; bf16_t* outb; int64_t ldo;
  int M, K, grp, out_grp_rows, out_row_off;
  int w_packed;  // W is in the K-step-major DMA order of care_pack_ln_weight (version-2 kernels only)
};

template <int N>
__device__ __forceinline__ void ln_wait_vm() {
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// AF32: A is fp32.  RG: row groups of 64 rows (BM = 64 * RG, 4 * RG waves).
template <bool AF32, int RG>
__global__ __launch_bounds__(256 * RG, RG == 1 ? 1 : 2) void gemm_ln_kernel(LnArgs p) {
  constexpr int BM = 64 * RG, NW = 4 * RG;
  constexpr int A_ROWB = AF32 ? 128 : 64;                 // bytes per A row per K step
  constexpr int A_BYTES = BM * A_ROWB, W_BYTES = LN_N * 64, SLOT = A_BYTES + W_BYTES;
  constexpr int NA = A_BYTES / 1024, NWI = W_BYTES / 1024;  // DMA instructions per stage
  constexpr int NPW = (NA + NWI) / NW;                     // per wave: 5, 6, 9 or 10
  static_assert((NA + NWI) % NW == 0, "DMA instructions must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 3 slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 2, cg = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  constexpr unsigned HTAB = (2u) | (3u << 3) | (4u << 6) | (2u << 9) | (5u << 12) | (7u << 15) | (4u << 18) | (1u << 21);

  // ---- per-lane DMA source pointers at K step 0 (advance by 64 B (bf16) / 128 B (fp32) per step)
  const unsigned char* src[NPW];
  int dst_off[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave * NPW + i;  // wave-uniform instruction index: [0, NA) -> A, [NA, NA+NWI) -> W
    if (q < NA) {
      if constexpr (AF32) {        // 8 rows x 8 chunks per instruction
        const int row = q * 8 + (lane >> 3), pch = lane & 7;
        const int ch = pch ^ ((HTAB >> (3 * ((row & 15) >> 1))) & 7);
        const int grow = min(m0 + row, p.M - 1);
        src[i] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const float*>(p.A) + (int64_t)grow * p.lda) + ch * 16;
      } else {                     // 16 rows x 4 chunks per instruction
        const int row = q * 16 + (lane >> 2), pch = lane & 3;
        const int ch = pch ^ ((row & 8) >> 2);
        const int grow = min(m0 + row, p.M - 1);
        src[i] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const bf16_t*>(p.A) + (int64_t)grow * p.lda) + ch * 16;
      }
      dst_off[i] = q * 1024;
    } else {
      const int n = (q - NA) * 16 + (lane >> 2), pch = lane & 3;
      const int ch = pch ^ ((n & 8) >> 2);
      src[i] = reinterpret_cast<const unsigned char*>(p.W + (int64_t)n * p.K) + ch * 16;
      dst_off[i] = A_BYTES + (q - NA) * 1024;
    }
  }
  auto stage = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = wave * NPW + i;
      const int step = (q < NA) ? A_ROWB : 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)kt * step),
                                       (__attribute__((address_space(3))) void*)(smem + slot * SLOT + dst_off[i]),
                                       16, 0, 0);
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment offsets inside a slot
  int a_off[4][AF32 ? 2 : 1], w_off[8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = rg * 64 + mt * 16 + fr;
    if constexpr (AF32) {
      const int x = (HTAB >> (3 * (fr >> 1))) & 7;
      a_off[mt][0] = row * 128 + (((2 * fg) ^ x) << 4);
      a_off[mt][1] = row * 128 + (((2 * fg + 1) ^ x) << 4);
    } else {
      a_off[mt][0] = row * 64 + ((fg ^ ((fr & 8) >> 2)) << 4);
    }
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) w_off[nt] = A_BYTES + (cg * 128 + nt * 16 + fr) * 64 + ((fg ^ ((fr & 8) >> 2)) << 4);

  const int nk = p.K >> 5;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; stage kt+1 (NPW younger DMA instructions) may stay in flight
    if (kt + 1 < nk) ln_wait_vm<NPW>(); else ln_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) stage(kt + 2, (kt + 2) % 3);  // the slot every wave finished reading last iteration
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* sl = smem + (kt % 3) * SLOT;
    bf16x8 fa[4], fb[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(sl + w_off[nt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (AF32) {
        const float4 lo = *reinterpret_cast<const float4*>(sl + a_off[mt][0]);
        const float4 hi = *reinterpret_cast<const float4*>(sl + a_off[mt][1]);
        fa[mt][0] = (bf16_t)lo.x; fa[mt][1] = (bf16_t)lo.y; fa[mt][2] = (bf16_t)lo.z; fa[mt][3] = (bf16_t)lo.w;
        fa[mt][4] = (bf16_t)hi.x; fa[mt][5] = (bf16_t)hi.y; fa[mt][6] = (bf16_t)hi.z; fa[mt][7] = (bf16_t)hi.w;
      } else {
        fa[mt] = *reinterpret_cast<const bf16x8*>(sl + a_off[mt][0]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)  // swapped: lane holds 4 consecutive columns of one row
        acc[mt][nt] = care_mfma_16x16x32_h16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
  }

  // ------------------------------------------------------------------ epilogue
  // acc[mt][nt][j] = C[row 64 rg + 16 mt + fr][col 128 cg + 16 nt + 4 fg + j] (block-local row).
  // The accumulators are parked in LDS (64 rows x 512 fp32 = 128 KiB, one row group at a time,
  // 16-byte chunk index ^= row & 15 against bank conflicts) and the rest is done ROW-WISE: one wave
  // per full row, lane l owning columns [4l, 4l+4) and [256+4l, 256+4l+4).  Bias / residual /
  // position / gamma / beta loads and both stores are then fully coalesced 16-byte accesses and
  // the LayerNorm statistics are plain wave shuffles (no spills, no partial-line stores).
  float* tile = reinterpret_cast<float*>(smem);
  float4 gm[2], bt[2], bs[2];  // row-invariant vectors of this lane's 2 x 4 columns
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int col = (h * 64 + lane) * 4;
    gm[h] = *reinterpret_cast<const float4*>(p.gamma + col);
    bt[h] = *reinterpret_cast<const float4*>(p.beta + col);
    bs[h] = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll 1
  for (int g = 0; g < RG; ++g) {
    __syncthreads();  // ring (or the previous row group's tile) no longer read by any wave
    if (rg == g) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const int row = mt * 16 + fr;
          const int chunk = (cg * 32 + nt * 4 + fg) ^ fr;
          *reinterpret_cast<f32x4*>(tile + row * LN_N + chunk * 4) = acc[mt][nt];
        }
    }
    __syncthreads();
    // 4 rows per iteration with every load issued before the first use: with one workgroup per
    // CU nothing else hides the residual-load latency (one row at a time cost ~2 us per row).
    constexpr int UNR = 4, RPW = 64 / NW;  // rows per wave: 16 (RG = 1) or 8 (RG = 2)
#pragma unroll 1
    for (int rb = 0; rb < RPW; rb += UNR) {
      float4 v[UNR][2], rs[UNR][2], ps[UNR][2];
      int grow[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = wave * RPW + rb + u;
        grow[u] = m0 + g * 64 + r;
        const int gc = min(grow[u], p.M - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int chunk = h * 64 + lane;
          v[u][h] = *reinterpret_cast<const float4*>(tile + r * LN_N + ((chunk ^ (r & 15)) * 4));
          if (p.res) rs[u][h] = *reinterpret_cast<const float4*>(p.res + (int64_t)gc * p.ldres + chunk * 4);
          if (p.pos) ps[u][h] = *reinterpret_cast<const float4*>(p.pos + (int64_t)(gc % p.grp) * LN_N + chunk * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (grow[u] >= p.M) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v[u][h].x += bs[h].x; v[u][h].y += bs[h].y; v[u][h].z += bs[h].z; v[u][h].w += bs[h].w;
          if (p.res) { v[u][h].x += rs[u][h].x; v[u][h].y += rs[u][h].y; v[u][h].z += rs[u][h].z; v[u][h].w += rs[u][h].w; }
          if (p.pos) { v[u][h].x += ps[u][h].x; v[u][h].y += ps[u][h].y; v[u][h].z += ps[u][h].z; v[u][h].w += ps[u][h].w; }
        }
        const float s = ((v[u][0].x + v[u][0].y) + (v[u][0].z + v[u][0].w)) + ((v[u][1].x + v[u][1].y) + (v[u][1].z + v[u][1].w));
        const float mean = care_wave_sum(s) * (1.0f / LN_N);
        float q = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float a = v[u][h].x - mean, b = v[u][h].y - mean, c = v[u][h].z - mean, e = v[u][h].w - mean;
          q += (a * a + b * b) + (c * c + e * e);
        }
        const float rstd = 1.0f / sqrtf(care_wave_sum(q) * (1.0f / LN_N) + p.eps);
        const int64_t orow = (int64_t)(grow[u] / p.grp) * p.out_grp_rows + p.out_row_off + (grow[u] % p.grp);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = (h * 64 + lane) * 4;
          float4 o;
          o.x = (v[u][h].x - mean) * rstd * gm[h].x + bt[h].x;
          o.y = (v[u][h].y - mean) * rstd * gm[h].y + bt[h].y;
          o.z = (v[u][h].z - mean) * rstd * gm[h].z + bt[h].z;
          o.w = (v[u][h].w - mean) * rstd * gm[h].w + bt[h].w;
          if (p.out) *reinterpret_cast<float4*>(p.out + orow * p.ldo + col) = o;
          if (p.outb) {
            bf16x4 ob;
            ob[0] = (bf16_t)o.x; ob[1] = (bf16_t)o.y; ob[2] = (bf16_t)o.z; ob[3] = (bf16_t)o.w;
            *reinterpret_cast<bf16x4*>(p.outb + orow * p.ldo + col) = ob;
          }
        }
      }
    }
  }
}

template <bool AF32, int RG>
int launch_ln(const LnArgs& p, hipStream_t st) {
  constexpr int BM = 64 * RG;
  constexpr size_t ring = 3 * (BM * (AF32 ? 128 : 64) + LN_N * 64);
  constexpr size_t lds = ring > 64 * LN_N * 4 ? ring : 64 * LN_N * 4;  // the epilogue parks 64 x 512 fp32
  const int blocks = (p.M + BM - 1) / BM;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_ln_kernel<AF32, RG>), (int)lds, lds_ok)) return e;
  hipLaunchKernelGGL((gemm_ln_kernel<AF32, RG>), dim3(blocks), dim3(256 * RG), lds, st, p);
  return care_launch_status();
}


// ------------------------------------------------------------------------------------------------
// Version 2 (round 2): 64- or 128-row blocks (RG = 1, 2), loads specialised by wave, LayerNorm
// finished in the accumulator registers, fragment reads software-pipelined across the K-step barrier.
//
// What was wrong with the kernel above: (1) its fp32-A form read LDS through HIP's `float4`, and for
// such a read hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front while an LDS-DMA is in flight
// (ext_vector types do not trigger it): every prefetched stage was drained before the K step could
// start - the embedder ran at 2.0 TB/s; (2) at two waves per SIMD (RG = 2) it wanted ~270 registers
// of 256 and spilled; (3) every wave issued loads of BOTH operands into one ring, so - vmcnt being
// in issue order - the short-latency W stream (L2 hits) and the long-latency A stream (HBM) could be
// prefetched no deeper than each other; (4) the epilogue parked the accumulators in LDS (128 KB per
// row group, four barriers).  Here:
//   * the first half of the waves streams W, the second half streams A; each operand has its OWN
//     ring (W: NSW slots of 32 KB; A: NSA slots of 64 RG x 128 B (fp32) or 64 B (bf16)) and each
//     loader waits on its own counted vmcnt, so raw fp32 features run NSA - 1 K steps ahead;
//   * ONE barrier per K step, in the MIDDLE of the step's MFMAs: a wave reads the second half of the
//     step's B fragments at the top, issues half of the MFMAs, then waits for its own stream's next
//     stage, meets the barrier, reads the A fragments and first B half of the NEXT step and issues the
//     other half of the MFMAs - so LDS reads and barrier skew are covered by matrix work of the same
//     wave; the DMA instructions of the stage that takes the slot freed at that barrier are issued
//     one by one between those MFMAs (a burst costs ~150 cycles per instruction up front);
//   * all waves compute (RG row groups x 4 column groups, 64 x 128 per wave, 128 accumulators);
//   * the LayerNorm is finished IN the accumulator registers: the W rows are permuted over the MFMA
//     rows when the fragment is read (tile pair p, lane group fg -> columns 32 p + 8 fg + [0, 8)) so
//     a lane holds 8 consecutive columns of its 4 rows; row statistics = per-lane sums, two
//     xor-shuffles over the lane groups, and a small exchange between the four column-group waves
//     through LDS (two-pass: mean, then centred squares); bias / residual / gamma / beta are read
//     with 16-byte loads and every store is 16 bytes per lane (64 or 128 contiguous bytes per row).
//     The per-row arithmetic does not depend on RG: a row's result is bit-identical whatever block
//     size the launch rule picks (batch-composition invariance, tests/test_gpu_properties.py).
#ifndef CARE_LN_ST_NT
#define CARE_LN_ST_NT 0  // bf16 output stores non-temporal (ablation)
#endif
#ifndef CARE_LN_A_AUX
#define CARE_LN_A_AUX 0  // cache policy of the A stream's DMA (2 = nt: every A byte is read once)
#endif
#ifndef CARE_LN_DBG
#define CARE_LN_DBG 0  // ablation builds (tools/variant_lib.py): 1 no MFMA, 2 no W DMA, 4 no A DMA, 8 no fragment reads, 16 no epilogue, 32 no stores
#endif
// the step's MFMA: bf16 operands, or - split products - the same registers holding fp16 pieces
template <bool F16>
__device__ __forceinline__ f32x4 ln2_mfma(const bf16x8& b, const bf16x8& a, const f32x4& c) {
  if constexpr (F16) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
  } else {
    return care_mfma_16x16x32_h16(b, a, c, 0, 0, 0);
  }
}

template <int N>
__device__ __forceinline__ void ln2_wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until all but `younger` stages (PER DMA instructions each) of this wave's stream have landed
template <int PER, int MAXY>
__device__ __forceinline__ void ln2_wait_stages(int younger) {
  static_assert(MAXY * PER <= 63, "vmcnt range");
  if constexpr (MAXY >= 6) { if (younger >= 6) { ln2_wait_vm<6 * PER>(); return; } }
  if constexpr (MAXY >= 5) { if (younger == 5) { ln2_wait_vm<5 * PER>(); return; } }
  if constexpr (MAXY >= 4) { if (younger == 4) { ln2_wait_vm<4 * PER>(); return; } }
  if constexpr (MAXY >= 3) { if (younger == 3) { ln2_wait_vm<3 * PER>(); return; } }
  if constexpr (MAXY >= 2) { if (younger == 2) { ln2_wait_vm<2 * PER>(); return; } }
  if constexpr (MAXY >= 1) { if (younger == 1) { ln2_wait_vm<PER>(); return; } }
  ln2_wait_vm<0>();
}

// EPI: which optional epilogue operand exists - 0: none (the Embedder), 1: the residual (decoder
// sub-blocks); a position table goes to the round-1 kernel (nothing on the path uses one here).
// A template parameter because a run-time condition per load makes hipcc branch around every load and
// wait for it on the spot, and duplicating the epilogue behind ONE run-time branch made it spill the
// 128 accumulators (700 B of scratch per lane).
// REP = 3 (fp32 A, packed split weight only): the split product a_hi w_hi + a_hi w_lo + a_lo w_hi with
// a = a_hi + a_lo, w = w_hi + w_lo in FP16 pieces (11 significant bits each: what is dropped, a_lo w_lo, is
// ~2^-22 of a product - fp32-grade; bf16 pieces would leave 2^-17) - three fp16 MFMA passes, the same rate as
// bf16, instead of the 16x slower exact-f32 MFMA (concept models: the embedder feeds a discrete top-30 choice,
// see engine.load_weights).  Feature magnitudes must stay below fp16's 65504 (ViT / ResNet / VGGish features are O(1..100)).  Every REAL K step is three consecutive virtual steps that
// share the A stage (converted as hi, hi, lo) and take the W stages w_hi, w_lo, w_hi of
// care_pack_ln_weight_split.
template <bool AF32, int RG, int NSW, int NSA, int EPI, int REP = 1>
__global__ __launch_bounds__(256 * RG, RG) void gemm_ln2_kernel(LnArgs p) {
  static_assert(REP == 1 || (REP == 3 && AF32), "split products need the fp32 operand");
  constexpr int BM = 64 * RG, NW = 4 * RG, NLD = NW / 2;  // NLD loader waves per operand
  // A moves in MACRO stages of 256 contiguous bytes per row (2 K steps of fp32, 4 of bf16): with 128
  // (64) bytes per row and step every DRAM page was visited for one cache line at a time and the raw
  // feature stream ran at 3.8 TB/s even with nothing else in the kernel (ablation, round 2)
  constexpr int A_ROWB = 256, KSUB = AF32 ? 2 : 4;   // bytes per A row per macro stage; K steps per macro stage
  constexpr int A_BYTES = BM * A_ROWB, W_BYTES = LN_N * 64;
  constexpr int A_BASE = NSW * W_BYTES;
  constexpr int NWI = W_BYTES / 1024 / NLD;          // DMA instructions per stage per W wave: 16 / 8
  constexpr int NAI = A_BYTES / 1024 / NLD;          // per macro stage per A wave: 8
  static_assert(NSW >= 2 && NSA >= 2, "ring depths");
  // NSA == 2 (bf16 A: a macro stage is 4 K steps): the whole of macro stage mac + 1 goes out in the FIRST step of
  // macro stage mac (three steps of lookahead) - the LDS this saves buys a third W stage: with two, the weight
  // tile of step kt + 1 was requested from L2 only one step (~1000 MFMA cycles) ahead and every step waited for it.
  constexpr bool AE = NSA == 2;
  static_assert(!AE || (!AF32 && REP == 1), "early A issue: bf16 A only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 2, cg = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const bool w_loader = wave < NLD;
  constexpr unsigned HTAB = (2u) | (3u << 3) | (4u << 6) | (2u << 9) | (5u << 12) | (7u << 15) | (4u << 18) | (1u << 21);

  // ---- DMA sources.  W: instruction q of a stage covers W rows [16 q, 16 q + 16) x 4 chunks; the lane's
  // part of the address - row (lane >> 2) of the 16, swizzled chunk - is the same for every q, so one
  // 32-bit lane offset + a wave-uniform base per instruction (SGPR base + VGPR offset addressing); a
  // PACKED W (care_pack_ln_weight) is the LDS image of every stage back to back: 1 KB per instruction,
  // lane-linear, full cache lines.  A: 4 rows x 16 chunks per instruction (256 contiguous bytes per
  // row), LDS position (row, c) takes source chunk c ^ (row & 15); rows clamped to the last one.
  const unsigned w_lane = p.w_packed ? (unsigned)lane * 16u
                                     : (unsigned)(lane >> 2) * (unsigned)p.K * 2u + (unsigned)(((lane & 3) ^ (((lane >> 2) & 8) >> 2)) << 4);
  unsigned a_src[NAI];  // byte offset from the block's first row (< 128 rows x lda: 32 bits; wave-uniform 64-bit base)
#pragma unroll
  for (int i = 0; i < NAI; ++i) {
    const int row = ((w_loader ? 0 : wave - NLD) * NAI + i) * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ (row & 15);
    a_src[i] = (unsigned)(min(m0 + row, p.M - 1) - m0) * (unsigned)p.lda * (AF32 ? 4u : 2u) + (unsigned)ch * 16u;
  }
  const unsigned char* a_blk = reinterpret_cast<const unsigned char*>(p.A) + (int64_t)m0 * p.lda * (AF32 ? 4 : 2);
  // one DMA instruction i of stage kt (W) / macro stage m (A) of this wave's stream
  // (stage to fetch, ring position it takes - they differ only past the end of K, see the schedule)
  auto dma_w = [&](int kt, int slot, int i) {
    if (CARE_LN_DBG & 2) return;
    const int q = wave * NWI + i;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.W) +
                                (p.w_packed ? (int64_t)kt * W_BYTES + q * 1024 : (int64_t)(q * 16) * p.K * 2 + (int64_t)kt * 64);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + w_lane),
                                     (__attribute__((address_space(3))) void*)(smem + (slot % NSW) * W_BYTES + q * 1024),
                                     16, 0, 0);
  };
  auto dma_a = [&](int m, int slot, int i) {
    if (CARE_LN_DBG & 4) return;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_blk + (int64_t)m * A_ROWB + a_src[i]),
                                     (__attribute__((address_space(3))) void*)(smem + A_BASE + (slot % NSA) * A_BYTES + ((wave - NLD) * NAI + i) * 1024),
                                     16, 0, CARE_LN_A_AUX);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment offsets.  W: MFMA row i = 4 fg' + j of tile nt reads W row
  // 128 cg + 32 (nt >> 1) + 8 (i >> 2) + 4 (nt & 1) + (i & 3)  (i = this lane's fr as the READER of row i),
  // so that the lane that ends up with rows 4 fg + [0, 4) of tiles 2p and 2p + 1 owns 8 consecutive columns.
  const int w_off0 = (cg * 128 + 8 * (fr >> 2) + (fr & 3)) * 64 + ((fg ^ ((fr & 4) >> 1)) << 4);
  // A: tile mt is rows 64 rg + 16 mt + fr and (row & 15) = fr for every tile, so tile mt = tile 0 +
  // mt * 16 rows (an immediate).  K step j of a macro stage: fp32 chunks 8 j + 2 fg, + 1; bf16 chunk
  // 4 j + fg; swizzled by ^ fr - all XORs, so step j is the lane offset ^ (j * 128) (^ (j * 64)) and
  // the second fp32 chunk is ^ 16.  Conflict-free for the 16-lane groups of ds_read_b128: lanes
  // {0-3, 12-15} of lane group 0 and {4-11} of lane group 1 land on 16 distinct 16-byte slots.
  const int a_lane = A_BASE + (rg * 64 + fr) * A_ROWB + (((AF32 ? 2 * fg : fg) ^ fr) << 4);
  // raw fragment reads (no conversion, no waits): B half h of step kt; A tiles [2 part, 2 part + 2) of step kt
  auto read_b = [&](int kt, int h, bf16x8 (&f)[4]) {
    if (CARE_LN_DBG & 8) { for (int q = 0; q < 4; ++q) { f[q] = bf16x8{}; asm volatile("" : "+v"(f[q])); } return; }
    const unsigned char* sw = smem + (kt % NSW) * W_BYTES + w_off0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nt = h * 4 + q;
      f[q] = *reinterpret_cast<const bf16x8*>(sw + ((nt >> 1) * 32 + (nt & 1) * 4) * 64);
    }
  };
  using araw_t = typename std::conditional<AF32, f32x4, bf16x8>::type;  // [tile][AF32 ? lo/hi : 1]
  auto read_a = [&](int kt, int part, araw_t (&r)[2][AF32 ? 2 : 1]) {
    if (CARE_LN_DBG & 8) { for (int t = 0; t < 2; ++t) for (int c = 0; c < (AF32 ? 2 : 1); ++c) { r[t][c] = araw_t{}; asm volatile("" : "+v"(r[t][c])); } return; }
    const int kr = kt / REP;  // real K step of virtual step kt
    const int jx = (kr % KSUB) * (AF32 ? 128 : 64);
    const unsigned char* sa = smem + ((kr / KSUB) % NSA) * A_BYTES + (a_lane ^ jx);
    const unsigned char* sx = smem + ((kr / KSUB) % NSA) * A_BYTES + (a_lane ^ jx ^ 16);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      r[t][0] = *reinterpret_cast<const araw_t*>(sa + (2 * part + t) * 16 * A_ROWB);
      if constexpr (AF32) r[t][1] = *reinterpret_cast<const araw_t*>(sx + (2 * part + t) * 16 * A_ROWB);
    }
  };
  auto cvt_a = [&](const araw_t (&r)[2][AF32 ? 2 : 1], bf16x8 (&f)[4], int part, int kt) {
    const bool lo = REP == 3 && kt % REP == 2;  // third pass of a real step: the low piece a - bf16(a)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (AF32) {  // pairs -> one v_cvt_pk_bf16_f32 each
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef h16_t bf16x2 __attribute__((ext_vector_type(2)));
        if constexpr (REP == 3) {  // split products: fp16 pieces (11 bits each) in the bf16-typed fragment registers
          // (whole 32-bit pairs are moved: an element-wise bit_cast of the halves made hipcc drop the odd ones)
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
          u32x4v packed;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              f16x2 pr = __builtin_convertvector(f32x2{r[t][c][e], r[t][c][e + 1]}, f16x2);
              if (lo) pr = __builtin_convertvector(f32x2{r[t][c][e] - (float)pr[0], r[t][c][e + 1] - (float)pr[1]}, f16x2);
              packed[2 * c + (e >> 1)] = __builtin_bit_cast(unsigned, pr);
            }
          f[2 * part + t] = __builtin_bit_cast(bf16x8, packed);
        } else {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              bf16x2 pr = __builtin_convertvector(f32x2{r[t][c][e], r[t][c][e + 1]}, bf16x2);
              f[2 * part + t][4 * c + e] = pr[0]; f[2 * part + t][4 * c + e + 1] = pr[1];
            }
        }
      } else {
        f[2 * part + t] = r[t][0];
      }
    }
  };

  const int nk = (p.K >> 5) * REP, nmac = (p.K >> 5) / KSUB;  // the launcher guarantees K % (32 KSUB) == 0; nk: virtual steps
  // Issue schedule (the same instruction counts in every step, so every wait is a constant):
  //   W wave: the NWI instructions of stage kt + NSW go out in the second half of step kt (its slot was
  //           read for the last time before this step's barrier);
  //   A wave: macro stage mac - 1 + NSA (the slot of macro stage mac - 1, free for the whole of macro
  //           stage mac) goes out NAS = NAI / KSUB instructions per step over the steps of macro stage mac.
  // Past the end of K the stage index is clamped: the last stage is simply fetched again into a slot
  // nobody will read - no tail cases, and the vmcnt arithmetic stays exact.
  constexpr int NAS = NAI / KSUB;
  static_assert(NAI % KSUB == 0 && 16 % NAI == 0, "A issue schedule");
  if (w_loader) {
#pragma unroll 1
    for (int s = 0; s < NSW; ++s)
#pragma unroll
      for (int i = 0; i < NWI; ++i) dma_w(min(s, nk - 1), s, i);
    ln2_wait_vm<(NSW - 1) * NWI>();
  } else {
#pragma unroll 1
    for (int s = 0; s < NSA - 1; ++s)   // macro stages 0 .. NSA - 2; NSA - 1 follows during macro stage 0
#pragma unroll
      for (int i = 0; i < NAI; ++i) dma_a(min(s, nmac - 1), s, i);
    ln2_wait_vm<(NSA - 2) * NAI>();
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // Two fragment sets in ping-pong (the K loop is unrolled by two: no register copies at the loop edge).
  bf16x8 faA[4], fbA[4], faB[4], fbB[4];   // step kt: all A tiles, B half 0
  {
    araw_t r0[2][AF32 ? 2 : 1], r1[2][AF32 ? 2 : 1];
    read_a(0, 0, r0); read_a(0, 1, r1); read_b(0, 0, fbA);
    cvt_a(r0, faA, 0, 0); cvt_a(r1, faA, 1, 0);
  }
  // One K step: fa / fb0 hold its fragments, fan / fbn receive the next step's.
  auto kstep = [&](int kt, bf16x8 (&fa)[4], bf16x8 (&fb0)[4], bf16x8 (&fan)[4], bf16x8 (&fbn)[4]) {
    const bool more = kt + 1 < nk;                       // a next step exists
    const int kr = kt / REP, mode = kt % REP;            // real step; 0 .. REP - 1: which product of the split
    const int mac = kr / KSUB, sub = kr % KSUB;
    const bool a_issue = mode == 0;                      // the A stream moves once per real step
    const bool a_edge = sub == KSUB - 1 && mode == REP - 1;  // step kt + 1 opens macro stage mac + 1
    bf16x8 fb1[4];
    araw_t ra[2][AF32 ? 2 : 1], rb[2][AF32 ? 2 : 1];
    __builtin_amdgcn_sched_barrier(0);
    read_b(kt, 1, fb1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (CARE_LN_DBG & 1) asm volatile("" :: "v"(fb0[q]), "v"(fa[mt]));
        else acc[mt][q] = ln2_mfma<REP == 3>(fb0[q], fa[mt], acc[mt][q]);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      // my stream's stage kt + 1 has landed (its younger instructions stay in flight); my reads of stage kt are done
      if (w_loader) ln2_wait_vm<(NSW - 2) * NWI>();
      else if (a_edge) ln2_wait_vm<AE ? 0 : REP == 1 ? (NSA - 3) * NAI + (KSUB - 1) * NAS : (NSA - 2) * NAI>();  // all of macro stage mac - 1 + NSA is out by now when REP > 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_a(kt + 1, 0, ra);
      read_b(kt + 1, 0, fbn);
    }
    __builtin_amdgcn_sched_barrier(0);
    // second half of the MFMAs with this step's DMA instructions woven in one at a time; the A tiles of
    // the next step arrive in two parts
    const int wst = min(kt + NSW, nk - 1), wslot = kt + NSW;           // W stage and the slot it takes
    const int ast = min(mac - 1 + NSA, nmac - 1), aslot = mac - 1 + NSA;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (CARE_LN_DBG & 1) asm volatile("" :: "v"(fb1[q]), "v"(fa[mt]));
        else acc[mt][4 + q] = ln2_mfma<REP == 3>(fb1[q], fa[mt], acc[mt][4 + q]);
        const int m = mt * 4 + q;  // 0..15
        if ((m + 1) % (16 / NWI) == 0 || (m + 1) % (16 / (AE ? NAI : NAS)) == 0) {
          if (w_loader) { if ((m + 1) % (16 / NWI) == 0) dma_w(wst, wslot, m / (16 / NWI)); }
          else if (AE) { if (sub == 0 && (m + 1) % (16 / NAI) == 0) dma_a(ast, aslot, m / (16 / NAI)); }
          else { if (a_issue && (m + 1) % (16 / NAS) == 0) dma_a(ast, aslot, sub * NAS + m / (16 / NAS)); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mt == 1 && more) {  // first A part has had 8 MFMAs to arrive: convert it, fetch the second part
        cvt_a(ra, fan, 0, kt + 1);
        read_a(kt + 1, 1, rb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) cvt_a(rb, fan, 1, kt + 1);
  };

#pragma unroll 1
  for (int kt = 0; kt < nk; kt += 2) {  // nk is even (K % 64 == 0)
    kstep(kt, faA, fbA, faB, fbB);
    kstep(kt + 1, faB, fbB, faA, fbA);
  }

  // ------------------------------------------------------------------ epilogue (in registers)
  // acc[mt][2p + e][j] = C[row 64 rg + 16 mt + fr][col 128 cg + 32 p + 8 fg + 4 e + j]
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the rings are dead: their first bytes become the statistics exchange
  if (CARE_LN_DBG & 16) {  // (ablation: no epilogue)
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) s += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
    if (s == 12345.678f && p.outb) p.outb[0] = (bf16_t)s;
    return;
  }
  float* stat = reinterpret_cast<float*>(smem);            // [2 passes][BM rows][4 column groups]
  const int col0 = cg * 128 + 8 * fg;
  int m0e = m0;
  asm volatile("" : "+s"(m0e));  // everything below depends on it: no epilogue address arithmetic is hoisted into the K loop's register budget
  int grow[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) grow[mt] = m0e + rg * 64 + mt * 16 + fr;

  // pass 1: v = acc + bias (+ residual) (+ position), row sums
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};
  {
    int dep = 0;
    int64_t roff[4], poff[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int gc = min(grow[mt], p.M - 1);
      roff[mt] = (int64_t)gc * p.ldres;
      poff[mt] = (int64_t)(gc % p.grp) * LN_N;
    }
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      const int c = col0 + 32 * pq;
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
      if (p.bias) { b0 = *reinterpret_cast<const float4*>(p.bias + c); b1 = *reinterpret_cast<const float4*>(p.bias + c + 4); }
      float4 r[4][2], q[4][2];
#pragma unroll
      for (int mth = 0; mth < 2; ++mth) {
#pragma unroll
      for (int mt = 2 * mth; mt < 2 * mth + 2; ++mt) {
        if constexpr (EPI == 1) {
          r[mt][0] = *reinterpret_cast<const float4*>(p.res + roff[mt] + c + dep);
          r[mt][1] = *reinterpret_cast<const float4*>(p.res + roff[mt] + c + 4 + dep);
        }
        if constexpr (EPI == 2) {
          q[mt][0] = *reinterpret_cast<const float4*>(p.pos + poff[mt] + c);
          q[mt][1] = *reinterpret_cast<const float4*>(p.pos + poff[mt] + c + 4);
        }
      }
#pragma unroll
      for (int mt = 2 * mth; mt < 2 * mth + 2; ++mt) {
        f32x4& v0 = acc[mt][2 * pq];
        f32x4& v1 = acc[mt][2 * pq + 1];
        v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
        v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
        if constexpr (EPI == 1) {
          v0[0] += r[mt][0].x; v0[1] += r[mt][0].y; v0[2] += r[mt][0].z; v0[3] += r[mt][0].w;
          v1[0] += r[mt][1].x; v1[1] += r[mt][1].y; v1[2] += r[mt][1].z; v1[3] += r[mt][1].w;
        }
        if constexpr (EPI == 2) {
          v0[0] += q[mt][0].x; v0[1] += q[mt][0].y; v0[2] += q[mt][0].z; v0[3] += q[mt][0].w;
          v1[0] += q[mt][1].x; v1[1] += q[mt][1].y; v1[2] += q[mt][1].z; v1[3] += q[mt][1].w;
        }
        rsum[mt] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
      }
      // two rows x one column group of loads in flight at a time (registers).  Neither sched_barrier nor a "memory"
      // clobber holds the loads of the later groups back (plain loads the kernel never clobbers: the compiler issued all
      // 32 of them ahead of the first add and, with the 128 accumulators live, spilled half - 88 B of scratch per lane
      // and a vmcnt(0) per spilled piece).  A data dependency does: `dep` (always 0) is an output of an asm statement
      // that reads a sum of this group, and part of the next group's addresses.
      asm volatile("" : "+v"(dep) : "v"(rsum[2 * mth]), "v"(rsum[2 * mth + 1]));
      __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float mean[4], rstd[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    float s = rsum[mt];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (fg == 0) stat[(rg * 64 + mt * 16 + fr) * 4 + cg] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const float4 t = *reinterpret_cast<const float4*>(stat + (rg * 64 + mt * 16 + fr) * 4);
    mean[mt] = ((t.x + t.y) + (t.z + t.w)) * (1.0f / LN_N);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float a = acc[mt][nt][0] - mean[mt], b = acc[mt][nt][1] - mean[mt];
      const float c = acc[mt][nt][2] - mean[mt], e = acc[mt][nt][3] - mean[mt];
      q += (a * a + b * b) + (c * c + e * e);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (fg == 0) stat[BM * 4 + (rg * 64 + mt * 16 + fr) * 4 + cg] = q;
  }
  __syncthreads();
  int64_t orow[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const float4 t = *reinterpret_cast<const float4*>(stat + BM * 4 + (rg * 64 + mt * 16 + fr) * 4);
    rstd[mt] = 1.0f / sqrtf(((t.x + t.y) + (t.z + t.w)) * (1.0f / LN_N) + p.eps);
    orow[mt] = (int64_t)(grow[mt] / p.grp) * p.out_grp_rows + p.out_row_off + (grow[mt] % p.grp);
  }
  // (the weight / bias pointers pass through an empty asm here: loads through them cannot be hoisted above this point,
  // i.e. into the first pass, where the 128 accumulators leave no room for them)
  const float* gam = p.gamma;
  const float* bet = p.beta;
  asm volatile("" : "+s"(gam), "+s"(bet));
#pragma unroll
  for (int pq = 0; pq < 4; ++pq) {
    const int c = col0 + 32 * pq;
    const float4 g0 = *reinterpret_cast<const float4*>(gam + c), g1 = *reinterpret_cast<const float4*>(gam + c + 4);
    const float4 e0 = *reinterpret_cast<const float4*>(bet + c), e1 = *reinterpret_cast<const float4*>(bet + c + 4);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (grow[mt] >= p.M) continue;
      const f32x4 v0 = acc[mt][2 * pq], v1 = acc[mt][2 * pq + 1];
      float4 o0, o1;
      o0.x = (v0[0] - mean[mt]) * rstd[mt] * g0.x + e0.x; o0.y = (v0[1] - mean[mt]) * rstd[mt] * g0.y + e0.y;
      o0.z = (v0[2] - mean[mt]) * rstd[mt] * g0.z + e0.z; o0.w = (v0[3] - mean[mt]) * rstd[mt] * g0.w + e0.w;
      o1.x = (v1[0] - mean[mt]) * rstd[mt] * g1.x + e1.x; o1.y = (v1[1] - mean[mt]) * rstd[mt] * g1.y + e1.y;
      o1.z = (v1[2] - mean[mt]) * rstd[mt] * g1.z + e1.z; o1.w = (v1[3] - mean[mt]) * rstd[mt] * g1.w + e1.w;
      if (CARE_LN_DBG & 32) { if (o0.x + o0.y + o0.z + o0.w + o1.x + o1.y + o1.z + o1.w != 12345.678f) continue; }  // (ablation: no stores)
      if (p.out) {
        float* op = p.out + orow[mt] * p.ldo + c;
        *reinterpret_cast<float4*>(op) = o0;
        *reinterpret_cast<float4*>(op + 4) = o1;
      }
      if (p.outb) {
        bf16x8 ob;
        ob[0] = (bf16_t)o0.x; ob[1] = (bf16_t)o0.y; ob[2] = (bf16_t)o0.z; ob[3] = (bf16_t)o0.w;
        ob[4] = (bf16_t)o1.x; ob[5] = (bf16_t)o1.y; ob[6] = (bf16_t)o1.z; ob[7] = (bf16_t)o1.w;
        if (CARE_LN_ST_NT) __builtin_nontemporal_store(ob, reinterpret_cast<bf16x8*>(p.outb + orow[mt] * p.ldo + c));
        else *reinterpret_cast<bf16x8*>(p.outb + orow[mt] * p.ldo + c) = ob;
      }
    }
  }
}

template <bool AF32, int RG, int NSW, int NSA, int EPI, int REP = 1>
int launch_ln2e(const LnArgs& p, hipStream_t st) {
  constexpr int BM = 64 * RG;
  constexpr int LDS = NSW * LN_N * 64 + NSA * BM * 256;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_ln2_kernel<AF32, RG, NSW, NSA, EPI, REP>), LDS, lds_ok)) return e;
  hipLaunchKernelGGL((gemm_ln2_kernel<AF32, RG, NSW, NSA, EPI, REP>), dim3((p.M + BM - 1) / BM), dim3(256 * RG), LDS, st, p);
  return care_launch_status();
}

template <bool AF32, int RG, int NSW, int NSA>
int launch_ln2(const LnArgs& p, hipStream_t st) {
  if (p.res) return launch_ln2e<AF32, RG, NSW, NSA, 1>(p, st);
  return launch_ln2e<AF32, RG, NSW, NSA, 0>(p, st);
}

// ------------------------------------------------------------------------------------------------
// Version 3 (round 5): the feature embedder's form - raw fp32 features, no residual - with the two operand streams on
// waves OF THEIR OWN and persistent workgroups.
//
// What was wrong with version 2 on that shape (M = 917504, K = 2048: 2.86 ms; tools/variant_lib.py ablations, round 5):
// the two streams alone - in a kernel that does nothing else - take 1.50 ms (tools/micro/emb_stream.hip: the 7.5-GB
// feature stream from HBM and the 14.7 GB every block re-reads of the packed weight from L2 share the CU's vector memory
// path, whose returns are in order: 57 GB/s per CU in all, and a wave spends the whole K step ISSUING its 4 - 8 pieces
// against that back pressure), the K loop without them 1.33 ms, and together 2.45 ms - every wave was a loader, so every
// wave's MFMAs queued behind its own blocked loads.  Then 0.40 ms of epilogue (LayerNorm + 0.94 GB of stores) during
// which the one workgroup of a CU had nothing in flight.  Here:
//   * 12 waves: 8 compute (2 row groups x 4 column groups of 64 x 128, as before - the K step and the LayerNorm in the
//     accumulator registers are version 2's, bit for bit) and 4 loaders that never touch the matrix pipe: two stream W
//     by LDS-DMA (16 pieces of 1 KB per K step each, 3 stages of 32 KB), two stream the features through REGISTERS
//     (global_load_dwordx4, two macro stages of 32 KB in flight per CU beyond the LDS ring), round them to the 16-bit
//     type ONCE (version 2 converted in every one of the four column-group waves) and write the image the fragment reads
//     want: 128-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) - conflict-free for the 16-lane
//     groups of ds_read_b128.  Three waves per SIMD: 168 registers each; a compute wave holds 128 accumulators, 4 A
//     fragments and two pairs of B fragments;
//   * persistent workgroups (one per CU, blocks blockIdx.x + i gridDim.x): the streams run on INTO the next block while
//     the compute waves finish the LayerNorm - all three weight stages and the first feature stages of the next block
//     are in LDS or in flight when its K loop starts;
//   * one barrier per K step for all 12 waves (B_g: the reads of stage g are done, stage g + 1 has landed) and the two
//     of the statistics exchange per block, which the loaders join.
constexpr int WS_NSW = 3, WS_NSA = 2;
constexpr int LN3_MIN_BLOCKS = 8;  // version 3 from this many whole 128-row blocks (tools/emb_bench.py --sweep: ahead of version 2 from 28 blocks down, K = 2048: 55 against 92 us)
constexpr int WS_W_BYTES = LN_N * 64;                         // a K step of the packed weight
constexpr int WS_A_BYTES = 128 * 128;                         // a macro stage (2 K steps): 128 rows x 64 K x 2 B
constexpr int WS_A_BASE = WS_NSW * WS_W_BYTES;
constexpr int WS_STAT = WS_A_BASE + WS_NSA * WS_A_BYTES;      // [2 passes][128 rows][4 column groups] floats
constexpr int WS_PAR = WS_STAT + 2 * 128 * 4 * 4;             // bias | gamma | beta, 512 floats each
constexpr int WS_LDS = WS_PAR + 3 * LN_N * 4;
#ifndef CARE_LN3_DBG
#define CARE_LN3_DBG 0  // ablation builds (tools/variant_lib.py): 1 no MFMA, 2 no W DMA, 4 no feature loads, 8 no fragment reads, 16 no epilogue, 32 no stores
#endif
#ifndef CARE_LN3_A_NT
#define CARE_LN3_A_NT 1  // the feature loads are non-temporal (every byte is read once: 5.4 -> 6.0 TB/s alone, tools/micro/emb_stream.hip)
#endif

template <bool OUT32>
__global__ __launch_bounds__(768) void gemm_ln3_kernel(LnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = p.K >> 5, nmac = nk >> 1;                       // the launcher guarantees K % 128 == 0: nmac is even
  const int nblk = (p.M + 127) >> 7;
  const int mine = (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // >= 1: gridDim.x <= nblk
  const int total = mine * nk, total_mac = mine * nmac;          // steps / macro stages of this workgroup's streams
  // (tools: CARE_LN3_DBG & 64 - workgroup 0 stamps its first 256 steps into p.out: [wave][step][4] shader-clock values)
  auto stamp = [&](int g, int k) {
    if (!(CARE_LN3_DBG & 64) || blockIdx.x != 0 || g >= 256 || lane != 0) return;
    reinterpret_cast<unsigned long long*>(p.out)[(wave * 256 + g) * 4 + k] = __builtin_amdgcn_s_memtime();
  };

  if (wave >= 10) {
    // ---------------------------------------------------------------- feature loader (two waves: rows 64 al + [0, 64))
    // Piece i of a macro stage (i < 16): rows 64 al + 4 i + (lane >> 4), the 16-byte chunk lane & 15 of their 256 bytes.
    // The loads are inline asm with counted waits of their own: hipcc's scoreboard loses the in-order count across the
    // loop edge and drains the stream (vmcnt(15) .. vmcnt(0): every load of the YOUNGER stages too) before the first use.
    // A register set is written by a load here and next touched by the wait below, whose "+v" operands order its uses.
    const int al = wave - 10, q4 = lane >> 4, c16 = lane & 15;
    const int64_t ldab = p.lda * 4;
    const unsigned voff = (unsigned)q4 * (unsigned)ldab + (unsigned)c16 * 16u;   // (the launcher: 128 rows of lda floats < 4 GB)
    const int lw = (64 * al + q4) * 128 + ((((c16 >> 1) ^ (q4 >> 1)) << 4) | ((c16 & 1) << 3));
    f32x4 R0[16], R1[16];
    int lbi = 0, lmac = 0;  // block / macro stage in it of the next load
    const unsigned char* lrow = reinterpret_cast<const unsigned char*>(p.A) + ((int64_t)blockIdx.x * 128 + 64 * al) * ldab;
    auto load_half = [&](f32x4 (&R)[16], int h) {  // pieces [8 h, 8 h + 8) of the next macro stage of the stream
#pragma unroll
      for (int i = 8 * h; i < 8 * h + 8; ++i) {
        const unsigned char* base = lrow + (int64_t)(4 * i) * ldab + (int64_t)lmac * 256;
        if (CARE_LN3_DBG & 4) { asm volatile("" : "+v"(R[i])); continue; }
        // ("+v": the set keeps its registers - an "=v" result may land elsewhere and be COPIED home at the loop edge, in flight)
        if (CARE_LN3_A_NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : "+v"(R[i]) : "v"(voff), "s"(base) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_nop 0" : "+v"(R[i]) : "v"(voff), "s"(base) : "memory");
      }
      if (h == 1 && ++lmac == nmac) {  // past the end of the stream the last macro stage is fetched again (nobody reads it):
        if (lbi + 1 < mine) { lmac = 0; ++lbi; lrow += (int64_t)gridDim.x * 128 * ldab; }   // no tail cases, constant waits
        else lmac = nmac - 1;
      }
    };
    // ... of the OLDEST stage in flight have landed (the 24 younger pieces may fly)
    auto landed = [&](f32x4 (&R)[16], int h, bool steady) {
      if (steady) asm volatile("s_waitcnt vmcnt(24)" : "+v"(R[8 * h]), "+v"(R[8 * h + 1]), "+v"(R[8 * h + 2]), "+v"(R[8 * h + 3]), "+v"(R[8 * h + 4]), "+v"(R[8 * h + 5]), "+v"(R[8 * h + 6]), "+v"(R[8 * h + 7])::"memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[8 * h]), "+v"(R[8 * h + 1]), "+v"(R[8 * h + 2]), "+v"(R[8 * h + 3]), "+v"(R[8 * h + 4]), "+v"(R[8 * h + 5]), "+v"(R[8 * h + 6]), "+v"(R[8 * h + 7])::"memory");
    };
    auto store_half = [&](int m, f32x4 (&R)[16], int h) {  // ... rounded, into the LDS slot of macro stage m
      unsigned char* dst = smem + WS_A_BASE + (m & 1) * WS_A_BYTES;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef h16_t hx2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 8 * h; i < 8 * h + 8; ++i) {
        const hx2 lo = __builtin_convertvector(f32x2{R[i][0], R[i][1]}, hx2), hi = __builtin_convertvector(f32x2{R[i][2], R[i][3]}, hx2);
        bf16x4 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
        *reinterpret_cast<bf16x4*>(dst + i * 512 + (lw ^ (((2 * i) & 7) << 4))) = o;
      }
    };
    // half h of macro stage m + 1 goes from its registers into the slot macro stage m - 1 left at B_2m-1, and the registers
    // take the same half of macro stage m + 3 (straight-line: a branch around either would put copies of the set at the join)
    auto produce = [&](int m, f32x4 (&R)[16], int h) {
      landed(R, h, true);
      store_half(m + 1, R, h);
      __builtin_amdgcn_sched_barrier(0);
      load_half(R, h);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) { R0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; R1[i] = R0[i]; }
    load_half(R0, 0); load_half(R0, 1);
    load_half(R1, 0); load_half(R1, 1);
    landed(R0, 0, false); landed(R0, 1, false);   // (the prologue waits for everything: once per launch)
    store_half(0, R0, 0); store_half(0, R0, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_half(R0, 0); load_half(R0, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // B_start
    // period m: the compute waves read macro stage m (steps 2m, 2m + 1)
    auto period = [&](int m, f32x4 (&R)[16]) {
      stamp(2 * m, 0);
      produce(m, R, 0);
      stamp(2 * m, 1);
      if (m > 0 && m % nmac == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }  // the previous block's LayerNorm: E1, E2
      __builtin_amdgcn_s_barrier();   // B_2m
      stamp(2 * m + 1, 0);
      produce(m, R, 1);
      stamp(2 * m + 1, 1);
      __builtin_amdgcn_s_barrier();   // B_2m+1
    };
#pragma unroll 1
    for (int m = 0; m < total_mac; m += 2) {
      period(m, R1);
      period(m + 1, R0);
    }
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();  // the last block's E1, E2
    return;
  }

  if (wave >= 8) {
    // ---------------------------------------------------------------- weight loader (two waves: half a stage each)
    const int wl = wave - 8;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.W) + wl * 16384 + lane * 16;
    int ikt = 0, islot = 0, issued = 0;  // K step / ring slot / number of the next stage to issue
    auto issue = [&]() {
      const unsigned char* s = wsrc + (int64_t)ikt * WS_W_BYTES;
      unsigned char* d = smem + islot * WS_W_BYTES + wl * 16384;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (!(CARE_LN3_DBG & 2)) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024),
                                         (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
      ikt = ikt + 1 == nk ? 0 : ikt + 1;
      islot = islot + 1 == WS_NSW ? 0 : islot + 1;
      ++issued;
    };
    while (issued < min(WS_NSW - 1, total)) issue();
    if (issued > 1) ln2_wait_vm<16>(); else ln2_wait_vm<0>();
    __builtin_amdgcn_s_barrier();  // B_start: stage 0 has landed
    int kt = 0;
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
      // stages < g are read: stage g + NSW - 1 takes the slot of stage g - 1
      stamp(g, 0);
      if (issued < min(g + WS_NSW, total)) issue();
      stamp(g, 1);
      if (g > 0 && kt == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }  // the previous block's E1, E2
      if (issued > g + 2) ln2_wait_vm<16>(); else ln2_wait_vm<0>();  // stage g + 1 has landed
      stamp(g, 2);
      __builtin_amdgcn_s_barrier();  // B_g
      stamp(g, 3);
      kt = kt + 1 == nk ? 0 : kt + 1;
    }
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();  // the last block's E1, E2
    return;
  }

  // ------------------------------------------------------------------ compute waves
  const int rg = wave >> 2, cg = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int w_off0 = (cg * 128 + 8 * (fr >> 2) + (fr & 3)) * 64 + ((fg ^ ((fr & 4) >> 1)) << 4);   // as version 2
  const int a_lane = WS_A_BASE + (rg * 64 + fr) * 128 + ((fg ^ ((fr >> 1) & 7)) << 4);
  float* stat = reinterpret_cast<float*>(smem + WS_STAT);
  const float* par = reinterpret_cast<const float*>(smem + WS_PAR);
  for (int i = tid; i < 3 * LN_N; i += 512)   // the epilogue's vectors, once per workgroup (tid < 512 here)
    reinterpret_cast<float*>(smem + WS_PAR)[i] = i < LN_N ? (p.bias ? p.bias[i] : 0.f) : i < 2 * LN_N ? p.gamma[i - LN_N] : p.beta[i - 2 * LN_N];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // B_start
  int wslot = 0, g = 0;
#pragma unroll 1
  for (int bi = 0; bi < mine; ++bi) {
    f32x4 acc[4][8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const unsigned char* sw = smem + wslot * WS_W_BYTES + w_off0;
      const unsigned char* sa = smem + ((g >> 1) & 1) * WS_A_BYTES + (a_lane ^ ((g & 1) << 6));
      bf16x8 fa[4], fb[2][2];
      stamp(g, 0);
      auto rd = [&](const unsigned char* q) {
        if (CARE_LN3_DBG & 8) { bf16x8 z = {}; asm volatile("" : "+v"(z)); return z; }
        return *reinterpret_cast<const bf16x8*>(q);
      };
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa[mt] = rd(sa + mt * 16 * 128);
      fb[0][0] = rd(sw);
      fb[0][1] = rd(sw + 4 * 64);
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        if (pr < 3) {  // the next pair of B fragments is requested BEFORE this pair's MFMAs (hipcc would reuse the registers)
          fb[(pr + 1) & 1][0] = rd(sw + (pr + 1) * 32 * 64);
          fb[(pr + 1) & 1][1] = rd(sw + ((pr + 1) * 32 + 4) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (CARE_LN3_DBG & 1) { asm volatile("" :: "v"(fb[pr & 1][0]), "v"(fb[pr & 1][1]), "v"(fa[mt])); continue; }
          acc[mt][2 * pr] = care_mfma_16x16x32_h16(fb[pr & 1][0], fa[mt], acc[mt][2 * pr], 0, 0, 0);
          acc[mt][2 * pr + 1] = care_mfma_16x16x32_h16(fb[pr & 1][1], fa[mt], acc[mt][2 * pr + 1], 0, 0, 0);
        }
      }
      wslot = wslot + 1 == WS_NSW ? 0 : wslot + 1;
      stamp(g, 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of stage g is in registers
      __builtin_amdgcn_s_barrier();  // B_g
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------------------------------------------------------- epilogue (version 2's arithmetic, no residual; in registers)
    // acc[mt][2p + e][j] = C[row 64 rg + 16 mt + fr][col 128 cg + 32 p + 8 fg + 4 e + j]
    const int m0 = ((int)blockIdx.x + bi * (int)gridDim.x) * 128;
    if (CARE_LN3_DBG & 16) {  // (ablation: no epilogue; the loaders still make its two barriers)
      float s = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) s += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
      if (s == 12345.678f && p.outb) p.outb[0] = (bf16_t)s;
      __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
      continue;
    }
    // (lane coordinates through an empty asm: every address below is recomputed per block - hoisted out of the block loop they
    // would sit in scratch across the K loop, whose 164 registers are all taken)
    int lid, grp = p.grp;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
    asm volatile("" : "+s"(grp));
    const int fre = lid & 15, fge = lid >> 4;
    auto xlane = [&](float v, int mask) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lid ^ mask) << 2, __builtin_bit_cast(int, v))); };
    const int col0e = cg * 128 + 8 * fge;
    const int srow = (rg * 64 + fre) * 4;   // this lane's first row in the statistics exchange (+ 64 floats per tile)
    // pass 1: v = acc + bias, row sums
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(par + col0e + 32 * pq), b1 = *reinterpret_cast<const f32x4*>(par + col0e + 32 * pq + 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4& v0 = acc[mt][2 * pq];
        f32x4& v1 = acc[mt][2 * pq + 1];
        v0 += b0; v1 += b1;
        rsum[mt] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float s = rsum[mt];
      s += xlane(s, 16);
      s += xlane(s, 32);
      if (fge == 0) stat[srow + mt * 64 + cg] = s;
    }
    __syncthreads();  // E1
    // pass 2: centred squares
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 t = *reinterpret_cast<const float4*>(stat + srow + mt * 64);
      const float mean = ((t.x + t.y) + (t.z + t.w)) * (1.0f / LN_N);
      float q = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        acc[mt][nt] -= mean;   // (v - mean) is what the third pass wants too
        const float a = acc[mt][nt][0], b = acc[mt][nt][1], c = acc[mt][nt][2], e = acc[mt][nt][3];
        q += (a * a + b * b) + (c * c + e * e);
      }
      q += xlane(q, 16);
      q += xlane(q, 32);
      if (fge == 0) stat[128 * 4 + srow + mt * 64 + cg] = q;
    }
    __syncthreads();  // E2
    // pass 3: ((v - mean) * rstd) * gamma + beta, rounded, 16 bytes per lane and row (64 contiguous bytes per row and store)
    int orow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 t = *reinterpret_cast<const float4*>(stat + 128 * 4 + srow + mt * 64);
      const float rstd = 1.0f / sqrtf(((t.x + t.y) + (t.z + t.w)) * (1.0f / LN_N) + p.eps);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[mt][nt] *= rstd;
      const int grow = m0 + rg * 64 + mt * 16 + fre;   // < M: the launcher sends whole blocks only
      orow[mt] = (grow / grp) * p.out_grp_rows + p.out_row_off + (grow % grp);
    }
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      const int c = col0e + 32 * pq;
      typedef h16_t hx4 __attribute__((ext_vector_type(4)));
      hx4 h0[4];
      {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(par + LN_N + c), e0 = *reinterpret_cast<const f32x4*>(par + 2 * LN_N + c);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 o = acc[mt][2 * pq] * g0 + e0;
          if (OUT32 && !(CARE_LN3_DBG & 96)) *reinterpret_cast<f32x4*>(p.out + (int64_t)orow[mt] * p.ldo + c) = o;
          h0[mt] = hx4{(h16_t)o[0], (h16_t)o[1], (h16_t)o[2], (h16_t)o[3]};
        }
      }
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(par + LN_N + c + 4), e1 = *reinterpret_cast<const f32x4*>(par + 2 * LN_N + c + 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 o = acc[mt][2 * pq + 1] * g1 + e1;
        if (OUT32 && !(CARE_LN3_DBG & 96)) *reinterpret_cast<f32x4*>(p.out + (int64_t)orow[mt] * p.ldo + c + 4) = o;
        if (CARE_LN3_DBG & 32) { if (o[0] + o[1] + o[2] + o[3] != 12345.678f) continue; }
        if (p.outb) {
          bf16x8 ob;
          ob[0] = h0[mt][0]; ob[1] = h0[mt][1]; ob[2] = h0[mt][2]; ob[3] = h0[mt][3];
          ob[4] = (h16_t)o[0]; ob[5] = (h16_t)o[1]; ob[6] = (h16_t)o[2]; ob[7] = (h16_t)o[3];
          *reinterpret_cast<bf16x8*>(p.outb + (int64_t)orow[mt] * p.ldo + c) = ob;
        }
      }
    }
    stamp(g - 1, 2);
  }
}

int launch_ln3(const LnArgs& p, hipStream_t st) {
  static std::atomic<unsigned long long> lds_ok{0};
  static std::atomic<unsigned long long> lds_ok32{0};
  const bool out32 = p.out != nullptr && !(CARE_LN3_DBG & 64);
  if (const int e = out32 ? care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_ln3_kernel<true>), WS_LDS, lds_ok32)
                          : care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_ln3_kernel<false>), WS_LDS, lds_ok)) return e;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    return n;
  }();
  const int nblk = (p.M + 127) / 128;
  if (out32) hipLaunchKernelGGL(gemm_ln3_kernel<true>, dim3(nblk < cus ? nblk : cus), dim3(768), WS_LDS, st, p);
  else hipLaunchKernelGGL(gemm_ln3_kernel<false>, dim3(nblk < cus ? nblk : cus), dim3(768), WS_LDS, st, p);
  return care_launch_status();
}

// ---- the split products (REP = 3 of version 2: a_hi w_hi + a_hi w_lo + a_lo w_hi in fp16 pieces) in version 3's form.
// A stage is one REAL K step of 32: the weight's w_hi | w_lo images (64 KB; care_pack_ln_weight_split's third image is its
// first again and is NOT fetched - version 2 streamed all three) and the features' a_hi | a_lo images (128 rows x 64 B
// each, written by the loader waves: the four column-group waves of version 2 each split every fragment themselves); two
// stages of each fill the 160 KB, the statistics exchange of the LayerNorm borrows the feature slot that was read last
// (a third barrier hands it back).  Per accumulator the three products of a step run in version 2's order: the same bits.
constexpr int WSS_W_STAGE = 2 * WS_W_BYTES;
constexpr int WSS_A_IMG = 128 * 64, WSS_A_STAGE = 2 * WSS_A_IMG;
constexpr int WSS_A_BASE = 2 * WSS_W_STAGE;
constexpr int WSS_LDS = WSS_A_BASE + 2 * WSS_A_STAGE;
static_assert(WSS_LDS <= 160 * 1024, "LDS budget");

template <bool OUT32>
__global__ __launch_bounds__(768) void gemm_ln3s_kernel(LnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = p.K >> 5;                                        // the launcher guarantees K % 128 == 0: nk % 4 == 0
  const int nblk = (p.M + 127) >> 7;
  const int mine = (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = mine * nk;

  if (wave >= 10) {
    // ---------------------------------------------------------------- feature loader (two waves: rows 64 al + [0, 64))
    // piece i of a stage (i < 8): rows 64 al + 8 i + (lane >> 3), the 16-byte fp32 chunk lane & 7 of their 128 bytes;
    // stage s lives in register set s % 4 (the halves of R0, R1), three younger stages in flight behind the one awaited
    const int al = wave - 10, q8 = lane >> 3, c8 = lane & 7;
    const int64_t ldab = p.lda * 4;
    const unsigned voff = (unsigned)q8 * (unsigned)ldab + (unsigned)c8 * 16u;
    const int lw = (64 * al + q8) * 64 + (((c8 >> 1) << 4) | ((c8 & 1) << 3));
    f32x4 R0[16], R1[16];
    int lbi = 0, lkt = 0;
    const unsigned char* lrow = reinterpret_cast<const unsigned char*>(p.A) + ((int64_t)blockIdx.x * 128 + 64 * al) * ldab;
    auto load_half = [&](f32x4 (&R)[16], int h) {  // the next stage of the stream into pieces [8 h, 8 h + 8) of a set
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned char* base = lrow + (int64_t)(8 * i) * ldab + (int64_t)lkt * 128;
        asm volatile("global_load_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : "+v"(R[8 * h + i]) : "v"(voff), "s"(base) : "memory");
      }
      if (++lkt == nk) {  // past the end of the stream the last stage is fetched again (nobody reads it)
        if (lbi + 1 < mine) { lkt = 0; ++lbi; lrow += (int64_t)gridDim.x * 128 * ldab; }
        else lkt = nk - 1;
      }
    };
    auto landed = [&](f32x4 (&R)[16], int h, bool steady) {
      if (steady) asm volatile("s_waitcnt vmcnt(24)" : "+v"(R[8 * h]), "+v"(R[8 * h + 1]), "+v"(R[8 * h + 2]), "+v"(R[8 * h + 3]), "+v"(R[8 * h + 4]), "+v"(R[8 * h + 5]), "+v"(R[8 * h + 6]), "+v"(R[8 * h + 7])::"memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[8 * h]), "+v"(R[8 * h + 1]), "+v"(R[8 * h + 2]), "+v"(R[8 * h + 3]), "+v"(R[8 * h + 4]), "+v"(R[8 * h + 5]), "+v"(R[8 * h + 6]), "+v"(R[8 * h + 7])::"memory");
    };
    auto store_half = [&](int s, f32x4 (&R)[16], int h) {  // stage s: a = a_hi + a_lo in fp16 pieces, into its slot
      unsigned char* dst = smem + WSS_A_BASE + (s & 1) * WSS_A_STAGE;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 x = R[8 * h + i];
        const f16x2 h0 = __builtin_convertvector(f32x2{x[0], x[1]}, f16x2), h1 = __builtin_convertvector(f32x2{x[2], x[3]}, f16x2);
        const f16x2 l0 = __builtin_convertvector(f32x2{x[0] - (float)h0[0], x[1] - (float)h0[1]}, f16x2);
        const f16x2 l1 = __builtin_convertvector(f32x2{x[2] - (float)h1[0], x[3] - (float)h1[1]}, f16x2);
        const int o = i * 512 + (lw ^ ((i & 1) << 5));  // row & 8 = 8 (i & 1): the chunk swizzle of a 64-byte row, chunk ^= (row & 8) >> 2
        *reinterpret_cast<f16x4*>(dst + o) = f16x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<f16x4*>(dst + WSS_A_IMG + o) = f16x4{l0[0], l0[1], l1[0], l1[1]};
      }
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) { R0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; R1[i] = R0[i]; }
    load_half(R0, 0); load_half(R0, 1); load_half(R1, 0); load_half(R1, 1);
    landed(R0, 0, true);
    store_half(0, R0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_half(R0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // B_start
    // step g: stage g + 1 goes from its registers (set (g + 1) % 4) into the slot stage g - 1 left at B_g-1; the registers
    // take stage g + 5.  At a block's first step the LayerNorm of its predecessor runs first, in the slot to be written.
    int kt = 0;
    auto step = [&](int g, f32x4 (&R)[16], int h) {
      if (g > 0 && kt == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }  // E1, E2, E3
      landed(R, h, true);
      store_half(g + 1, R, h);
      __builtin_amdgcn_sched_barrier(0);
      load_half(R, h);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // B_g
      kt = kt + 1 == nk ? 0 : kt + 1;
    };
#pragma unroll 1
    for (int g = 0; g < total; g += 4) {
      step(g, R0, 1); step(g + 1, R1, 0); step(g + 2, R1, 1); step(g + 3, R0, 0);
    }
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();  // the last block's E1, E2, E3
    return;
  }

  if (wave >= 8) {
    // ---------------------------------------------------------------- weight loader (two waves: 32 of a stage's 64 pieces each)
    const int wl = wave - 8;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.W) + wl * 32768 + lane * 16;
    int ikt = 0;
    auto issue = [&](int s) {  // stage s: the hi (wl = 0) or lo (wl = 1) image of its K step
      const unsigned char* src = wsrc + (int64_t)ikt * (3 * WS_W_BYTES);
      unsigned char* d = smem + (s & 1) * WSS_W_STAGE + wl * 32768;
#pragma unroll
      for (int i = 0; i < 32; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
      ikt = ikt + 1 == nk ? 0 : ikt + 1;
    };
    issue(0);
    ln2_wait_vm<0>();
    __builtin_amdgcn_s_barrier();  // B_start
    int kt = 0;
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
      if (g + 1 < total) issue(g + 1);  // into the slot stage g - 1 left at B_g-1 (also while the LayerNorm runs)
      if (g > 0 && kt == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
      ln2_wait_vm<0>();
      __builtin_amdgcn_s_barrier();  // B_g
      kt = kt + 1 == nk ? 0 : kt + 1;
    }
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
    return;
  }

  // ------------------------------------------------------------------ compute waves
  const int rg = wave >> 2, cg = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int w_off0 = (cg * 128 + 8 * (fr >> 2) + (fr & 3)) * 64 + ((fg ^ ((fr & 4) >> 1)) << 4);   // as version 2
  const int a_lane = WSS_A_BASE + (rg * 64 + fr) * 64 + ((fg ^ ((fr & 8) >> 2)) << 4);
  float* stat = reinterpret_cast<float*>(smem + WSS_A_BASE + WSS_A_STAGE);   // feature slot 1: read last in every block (nk is even)
  __builtin_amdgcn_s_barrier();  // B_start
  int g = 0;
#pragma unroll 1
  for (int bi = 0; bi < mine; ++bi) {
    f32x4 acc[4][8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const unsigned char* sw = smem + (g & 1) * WSS_W_STAGE + w_off0;
      const unsigned char* sa = smem + (g & 1) * WSS_A_STAGE + a_lane;
      bf16x8 fa[4], fb[2][2];
      auto rd = [&](const unsigned char* q) { return *reinterpret_cast<const bf16x8*>(q); };
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {  // a_hi w_hi, a_hi w_lo, a_lo w_hi
        const unsigned char* swp = sw + (pass == 1 ? WS_W_BYTES : 0);
        if (pass != 1) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) fa[mt] = rd(sa + (pass == 2 ? WSS_A_IMG : 0) + mt * 16 * 64);
        }
        fb[0][0] = rd(swp);
        fb[0][1] = rd(swp + 4 * 64);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          if (pr < 3) {
            fb[(pr + 1) & 1][0] = rd(swp + (pr + 1) * 32 * 64);
            fb[(pr + 1) & 1][1] = rd(swp + ((pr + 1) * 32 + 4) * 64);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            acc[mt][2 * pr] = ln2_mfma<true>(fb[pr & 1][0], fa[mt], acc[mt][2 * pr]);
            acc[mt][2 * pr + 1] = ln2_mfma<true>(fb[pr & 1][1], fa[mt], acc[mt][2 * pr + 1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of stage g is in registers
      __builtin_amdgcn_s_barrier();  // B_g
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------------------------------------------------------- epilogue (version 2's arithmetic; the vectors from global memory)
    const int m0 = ((int)blockIdx.x + bi * (int)gridDim.x) * 128;
    int lid, grp = p.grp;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
    asm volatile("" : "+s"(grp));
    const int fre = lid & 15, fge = lid >> 4;
    auto xlane = [&](float v, int mask) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lid ^ mask) << 2, __builtin_bit_cast(int, v))); };
    const int col0e = cg * 128 + 8 * fge;
    const int srow = (rg * 64 + fre) * 4;
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + col0e + 32 * pq); b1 = *reinterpret_cast<const f32x4*>(p.bias + col0e + 32 * pq + 4); }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4& v0 = acc[mt][2 * pq];
        f32x4& v1 = acc[mt][2 * pq + 1];
        v0 += b0; v1 += b1;
        rsum[mt] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float s = rsum[mt];
      s += xlane(s, 16);
      s += xlane(s, 32);
      if (fge == 0) stat[srow + mt * 64 + cg] = s;
    }
    __syncthreads();  // E1
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 t = *reinterpret_cast<const float4*>(stat + srow + mt * 64);
      const float mean = ((t.x + t.y) + (t.z + t.w)) * (1.0f / LN_N);
      float q = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        acc[mt][nt] -= mean;
        const float a = acc[mt][nt][0], b = acc[mt][nt][1], c = acc[mt][nt][2], e = acc[mt][nt][3];
        q += (a * a + b * b) + (c * c + e * e);
      }
      q += xlane(q, 16);
      q += xlane(q, 32);
      if (fge == 0) stat[128 * 4 + srow + mt * 64 + cg] = q;
    }
    __syncthreads();  // E2
    int orow[4];
    float rstd[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 t = *reinterpret_cast<const float4*>(stat + 128 * 4 + srow + mt * 64);
      rstd[mt] = 1.0f / sqrtf(((t.x + t.y) + (t.z + t.w)) * (1.0f / LN_N) + p.eps);
    }
    __syncthreads();  // E3: the exchange is read - the feature loader may write its slot
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[mt][nt] *= rstd[mt];
      const int grow = m0 + rg * 64 + mt * 16 + fre;
      orow[mt] = (grow / grp) * p.out_grp_rows + p.out_row_off + (grow % grp);
    }
    const float* gam = p.gamma;
    const float* bet = p.beta;
    asm volatile("" : "+s"(gam), "+s"(bet));
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      const int c = col0e + 32 * pq;
      typedef h16_t hx4 __attribute__((ext_vector_type(4)));
      hx4 h0[4];
      {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + c), e0 = *reinterpret_cast<const f32x4*>(bet + c);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 o = acc[mt][2 * pq] * g0 + e0;
          if (OUT32) *reinterpret_cast<f32x4*>(p.out + (int64_t)orow[mt] * p.ldo + c) = o;
          h0[mt] = hx4{(h16_t)o[0], (h16_t)o[1], (h16_t)o[2], (h16_t)o[3]};
        }
      }
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gam + c + 4), e1 = *reinterpret_cast<const f32x4*>(bet + c + 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 o = acc[mt][2 * pq + 1] * g1 + e1;
        if (OUT32) *reinterpret_cast<f32x4*>(p.out + (int64_t)orow[mt] * p.ldo + c + 4) = o;
        if (p.outb) {
          bf16x8 ob;
          ob[0] = h0[mt][0]; ob[1] = h0[mt][1]; ob[2] = h0[mt][2]; ob[3] = h0[mt][3];
          ob[4] = (h16_t)o[0]; ob[5] = (h16_t)o[1]; ob[6] = (h16_t)o[2]; ob[7] = (h16_t)o[3];
          *reinterpret_cast<bf16x8*>(p.outb + (int64_t)orow[mt] * p.ldo + c) = ob;
        }
      }
    }
  }
}

int launch_ln3s(const LnArgs& p, hipStream_t st) {
  static std::atomic<unsigned long long> lds_ok{0}, lds_ok32{0};
  const bool out32 = p.out != nullptr;
  if (const int e = out32 ? care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_ln3s_kernel<true>), WSS_LDS, lds_ok32)
                          : care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_ln3s_kernel<false>), WSS_LDS, lds_ok)) return e;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    return n;
  }();
  const int nblk = (p.M + 127) / 128;
  if (out32) hipLaunchKernelGGL(gemm_ln3s_kernel<true>, dim3(nblk < cus ? nblk : cus), dim3(768), WSS_LDS, st, p);
  else hipLaunchKernelGGL(gemm_ln3s_kernel<false>, dim3(nblk < cus ? nblk : cus), dim3(768), WSS_LDS, st, p);
  return care_launch_status();
}

}  // namespace

static int gemm_ln_impl(const void* A, int64_t lda, int a_dtype, const void* W, int w_packed, const float* bias,
                        const float* res, int64_t ldres, const float* pos, const float* gamma, const float* beta,
                        float eps, float* out, void* out_bf16, int64_t ldo, int M, int N, int K, int grp,
                        int out_grp_rows, int out_row_off, void* stream) {
  if (!A || !W || !gamma || !beta || (!out && !out_bf16) || M <= 0 || K <= 0 || grp <= 0) return CARE_EINVAL;
  if (a_dtype != CARE_F32 && a_dtype != CARE_BF16) return CARE_EDTYPE;
  if (N != LN_N || K % 32 != 0) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % (a_dtype == CARE_BF16 ? 8 : 4)) || (ldo % 4) ||
      (res && (ldres % 4)) || (out && !care_aligned16(out)) || (out_bf16 && !care_aligned16(out_bf16)) ||
      (bias && !care_aligned16(bias)))
    return CARE_EALIGN;
  // A ragged row count in the embedder's form (2990 test clips x 28 frames): the whole multiples of lcm(128, grp) rows go to
  // version 3 (whole 128-row blocks, whole output groups), the remainder to version 2 - row for row the same bits.
  if (w_packed && a_dtype == CARE_F32 && !res && !pos && K % 128 == 0 && M % 128 != 0 && grp <= 4096) {
    const bool ungrouped = grp >= M;  // one group: output row = out_row_off + row
    int64_t unit = ungrouped ? 128 : grp;
    while (unit % 128) unit += grp;   // lcm(128, grp): 896 rows for 28 frames
    const int64_t head = (M / unit) * unit;
    if (head >= 128 * LN3_MIN_BLOCKS && head < M) {
      const int64_t adv = (ungrouped ? head : (head / grp) * out_grp_rows) * ldo;  // output elements the head covers
      int rc = gemm_ln_impl(A, lda, a_dtype, W, w_packed, bias, res, ldres, pos, gamma, beta, eps, out, out_bf16, ldo, (int)head, N, K,
                            ungrouped ? (int)head : grp, out_grp_rows, out_row_off, stream);
      if (rc) return rc;
      return gemm_ln_impl(reinterpret_cast<const float*>(A) + head * lda, lda, a_dtype, W, w_packed, bias, res, ldres, pos, gamma, beta, eps,
                          out ? out + adv : nullptr, out_bf16 ? reinterpret_cast<bf16_t*>(out_bf16) + adv : nullptr, ldo, (int)(M - head), N, K,
                          ungrouped ? (int)(M - head) : grp, out_grp_rows, out_row_off, stream);
    }
  }
  LnArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias; p.res = res; p.ldres = ldres;
  p.pos = pos; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out = out; p.outb = reinterpret_cast<bf16_t*>(out_bf16);
  p.ldo = ldo; p.M = M; p.K = K; p.grp = grp; p.out_grp_rows = out_grp_rows; p.out_row_off = out_row_off;
  p.w_packed = w_packed;
  hipStream_t st = (hipStream_t)stream;
  // 64-row blocks (RG = 1) or 128-row blocks (RG = 2, one launch round of them takes ~1.3x as long):
  // whichever needs less time in whole rounds over the 256 CUs.  *Measured* K = 512: M = 16384
  // 38 vs 44 us (RG 1 wins, one round each), M = 20480 68 vs 47.5 us (RG 1 needs two rounds),
  // M = 32768 70 vs 53 us.
  const long rounds1 = ((M + 63) / 64 + 255) / 256, rounds2 = ((M + 127) / 128 + 255) / 256;
  bool big = 10 * rounds1 > 13 * rounds2;
  if (const char* e = getenv("CARE_LN_RG")) big = atoi(e) >= 2;  // tuning override
  static const int v2 = [] { const char* e = getenv("CARE_LN_V2"); return e ? atoi(e) : 1; }();  // A/B switch: 0 = round-1 kernel
  // version 2 moves A in 256-byte pieces of a row: K must be a whole number of them
  const bool v2_ok = !pos && K % (a_dtype == CARE_F32 ? 64 : 128) == 0;
  if (w_packed && !(v2 && v2_ok)) return CARE_ESHAPE;  // only the version-2 kernels read the packed order
  // version 3 (loader waves, persistent workgroups): raw fp32 features, no residual, at least a workgroup per CU's worth of
  // 128-row blocks.  A row's arithmetic is version 2's: the launch rule does not change a result.
  const char* v3e = getenv("CARE_LN_V3");  // A/B switch (read per call: the tests compare the two forms in one process)
  const char* v3m = getenv("CARE_LN_V3_MIN");  // tuning: the fewest 128-row blocks version 3 takes
  const bool v3_ok = (!v3e || atoi(v3e)) && v2 && a_dtype == CARE_F32 && !res && K % 128 == 0 && M % 128 == 0 &&
                     M >= 128 * (v3m ? atoi(v3m) : LN3_MIN_BLOCKS) && lda * 4 * 128 < (1ll << 32);
  if (v3_ok && w_packed == 1) return launch_ln3(p, st);
  if (w_packed == 2) {  // split products (care_pack_ln_weight_split): fp32 A, no residual
    if (a_dtype != CARE_F32 || res) return CARE_ESHAPE;
    if (v3_ok) return launch_ln3s(p, st);
    return big ? launch_ln2e<true, 2, 2, 3, 0, 3>(p, st) : launch_ln2e<true, 1, 3, 4, 0, 3>(p, st);
  }
  if (v2 && v2_ok) {
    if (a_dtype == CARE_F32) return big ? launch_ln2<true, 2, 2, 3>(p, st) : launch_ln2<true, 1, 3, 4>(p, st);
    static const int ae = [] { const char* e = getenv("CARE_LN_AE"); return e ? atoi(e) : 1; }();  // A/B switch
    if (big && ae) return launch_ln2<false, 2, 3, 2>(p, st);
    return big ? launch_ln2<false, 2, 2, 3>(p, st) : launch_ln2<false, 1, 3, 4>(p, st);
  }
  if (a_dtype == CARE_F32) return big ? launch_ln<true, 2>(p, st) : launch_ln<true, 1>(p, st);
  return big ? launch_ln<false, 2>(p, st) : launch_ln<false, 1>(p, st);
}

extern "C" int care_gemm_ln(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias,
                            const float* res, int64_t ldres, const float* pos, const float* gamma, const float* beta,
                            float eps, float* out, void* out_bf16, int64_t ldo, int M, int N, int K, int grp,
                            int out_grp_rows, int out_row_off, void* stream) {
  return gemm_ln_impl(A, lda, a_dtype, W, 0, bias, res, ldres, pos, gamma, beta, eps, out, out_bf16, ldo, M, N, K, grp,
                      out_grp_rows, out_row_off, stream);
}

extern "C" int care_gemm_ln_packed(const void* A, int64_t lda, int a_dtype, const void* W_packed, const float* bias,
                                   const float* res, int64_t ldres, const float* gamma, const float* beta, float eps,
                                   float* out, void* out_bf16, int64_t ldo, int M, int N, int K, int grp,
                                   int out_grp_rows, int out_row_off, void* stream) {
  return gemm_ln_impl(A, lda, a_dtype, W_packed, 1, bias, res, ldres, nullptr, gamma, beta, eps, out, out_bf16, ldo, M, N,
                      K, grp, out_grp_rows, out_row_off, stream);
}

extern "C" int care_gemm_ln_split(const void* A, int64_t lda, const void* W_split, const float* bias, const float* gamma,
                                  const float* beta, float eps, float* out, void* out_bf16, int64_t ldo, int M, int N,
                                  int K, int grp, int out_grp_rows, int out_row_off, void* stream) {
  return gemm_ln_impl(A, lda, CARE_F32, W_split, 2, bias, nullptr, 0, nullptr, gamma, beta, eps, out, out_bf16, ldo, M, N, K,
                      grp, out_grp_rows, out_row_off, stream);
}

namespace {
// W [512, K] fp32 -> the stream of the split-product kernels: per real K step the three 32-KB LDS images
// w_hi, w_lo, w_hi (w_hi = fp16(w), w_lo = fp16(w - w_hi)), rows and swizzled chunks as in pack_ln_weight_kernel
__global__ void pack_ln_weight_split_kernel(const float* W, bf16_t* Wp, int K) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk of one of the three images
  const int64_t total = (int64_t)LN_N * K / 8 * 3;
  if (slot >= total) return;
  const int c = (int)(slot & 3), n = (int)((slot >> 2) % LN_N);
  const int64_t img = slot / (4 * LN_N);     // 3 kt + j
  const int kt = (int)(img / 3), j = (int)(img % 3);
  const int src_chunk = c ^ ((n & 8) >> 2);
  const float* src = W + (int64_t)n * K + kt * 32 + src_chunk * 8;
  bf16x8 o;
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f16x8 oh;
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // fp16 pieces in the 16-bit slots
    const _Float16 hi = (_Float16)src[e];
    oh[e] = j == 1 ? (_Float16)(src[e] - (float)hi) : hi;
  }
  o = __builtin_bit_cast(bf16x8, oh);
  reinterpret_cast<bf16x8*>(Wp)[slot] = o;
}
}  // namespace

extern "C" int care_pack_ln_weight_split(const float* W, void* W_split, int N, int K, void* stream) {
  if (!W || !W_split) return CARE_EINVAL;
  if (N != LN_N || K <= 0 || K % 64 != 0) return CARE_ESHAPE;
  if (!care_aligned16(W) || !care_aligned16(W_split)) return CARE_EALIGN;
  const int64_t total = (int64_t)LN_N * K / 8 * 3;
  hipLaunchKernelGGL(pack_ln_weight_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                     reinterpret_cast<bf16_t*>(W_split), K);
  return care_launch_status();
}

namespace {
// W [512, K] bf16 (nn.Linear layout) -> the K-step-major order the version-2 kernels stream: for every K
// step of 32 the 32-KB LDS image of that step, rows in order, the four 16-byte chunks of a row at their
// swizzled positions (position c of row n holds chunk c ^ ((n & 8) >> 2)).
__global__ void pack_ln_weight_kernel(const bf16_t* W, bf16_t* Wp, int K) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk per thread
  const int64_t total = (int64_t)LN_N * K / 8;
  if (slot >= total) return;
  const int c = (int)(slot & 3), n = (int)((slot >> 2) % LN_N);
  const int kt = (int)(slot / (4 * LN_N));
  const int src_chunk = c ^ ((n & 8) >> 2);
  reinterpret_cast<bf16x8*>(Wp)[slot] = *reinterpret_cast<const bf16x8*>(W + (int64_t)n * K + kt * 32 + src_chunk * 8);
}
}  // namespace

extern "C" int care_pack_ln_weight(const void* W, void* W_packed, int N, int K, void* stream) {
  if (!W || !W_packed || W == W_packed) return CARE_EINVAL;
  if (N != LN_N || K <= 0 || K % 32 != 0) return CARE_ESHAPE;
  if (!care_aligned16(W) || !care_aligned16(W_packed)) return CARE_EALIGN;
  const int64_t total = (int64_t)LN_N * K / 8;
  hipLaunchKernelGGL(pack_ln_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(W), reinterpret_cast<bf16_t*>(W_packed), K);
  return care_launch_status();
}
