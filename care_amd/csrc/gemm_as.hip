// gemm_as.hip - "A-stationary" bf16 GEMM for the K = d_model shapes of the decoder (gfx950).
//
// Almost every GEMM of the path has K = d_model = 512 (QKV, Wq, Wo, FFN1, cross-K/V, vocab)
// and a tall-skinny or huge-N shape.  A classic LDS-tiled kernel re-stages A for every
// N-tile and synchronises every k-step; at K = 512 that is all prologue.  Here instead:
//
//   * a workgroup owns a panel of 128 rows (4 waves x 32 rows).  Each wave loads the MFMA
//     A-fragments of its 32 rows for the whole K (<= 512) ONCE, straight from global memory
//     into 128 VGPRs (all 32 loads in flight at once), and keeps them there;
//   * W is streamed in tiles of 16 output columns x 512 k (16 KiB) through a 4-deep LDS ring
//     filled by LDS-DMA (global_load_lds, 16 B per lane, one 1-KiB W row per
//     wave-instruction) with THREE tiles in flight: the DMA latency (~2 us under load) is
//     several times the 32 MFMAs a wave spends on a tile, so one tile of prefetch leaves the
//     kernel latency-bound (measured: 18 % MFMA occupancy).  Waits are counted
//     (s_waitcnt vmcnt(N) with N = the exact number of younger VM operations, output stores
//     included) and the barrier is a raw s_barrier, so neither the in-flight DMA nor the
//     previous tiles' stores are drained;
//   * the LDS image is lane-linear, so the bank-conflict swizzle (16-byte chunk ^= row & 15)
//     is applied to the per-lane SOURCE address and again on the ds_read_b128 of the B
//     fragment (conflict-free for every 16-lane read group);
//   * in the store mode the MFMA operands are swapped (D = tile of (A W^T)^T), so a lane holds
//     4 consecutive output columns of one row: one 16-byte (fp32) / 8-byte (bf16) store per
//     16x16 tile per lane;
//   * block -> (panel, column range) mapping is XCD-aware: the sharers of the bigger operand
//     are XCD neighbours (tall A: the blocks of a panel; wide W: the panels walking one range).
//
// MODES  STREAM_STORE : bias + activation + (split) store per tile; with kslices > 1 it is
//                       the split-K form (slice s -> fp32 slab s, summed by care_add_ln).
//        STREAM_ARGMAX: running per-row (max, argmax, sum-exp) over the block's column range
//                       (greedy vocabulary projection; the logits are never stored).
// A may be bf16 (no conversion) or fp32 (rounded to bf16 on load).  W is bf16 [N, ldw].
// Requires K % 128 == 0 and K <= 512 per slice.
#include <cstdlib>

#include "care_common.h"

// Ablation builds only (tools/gemm_bench.py): -DCARE_AS_DBG=<bits> 1 no stores, 2 no MFMA, 4 no W DMA, 16 no argmax statistics, 32 no B-fragment LDS reads,
// 8 no A loads.  Compile-time so that the shipped kernel carries no debug branches.
#ifndef CARE_AS_DBG
#define CARE_AS_DBG 0
#endif

namespace {

// _LAB: also the logit of a label column.  _COLLECT: no statistics; every logit >= its row's threshold is
// appended to a small per-row candidate list (second pass of the fused beam-search selection).
enum { STREAM_STORE = 0, STREAM_ARGMAX = 1, STREAM_ARGMAX_LAB = 2, STREAM_COLLECT = 3 };

constexpr int TILE_N = 16;             // output columns per W tile
constexpr int TILE_BYTES = TILE_N * 1024;
constexpr int RING = 4;                // LDS ring slots
constexpr int AHEAD = 3;               // tiles in flight
constexpr int BIAS_OFF = RING * TILE_BYTES;
constexpr int BIAS_MAX = 2048;         // bias columns a block can stage
constexpr int LDS_BYTES = BIAS_OFF + BIAS_MAX * 4;

struct AsArgs {
  const void* A; int64_t lda;
  const bf16_t* W;
  const float* bias;
  void* C0; int64_t ldc0; int c0_bf16;
  void* C1; int64_t ldc1; int c1_bf16;
  int n_split, M, N, K, act;
  int panels, ns;        // grid decomposition
  int panel_major;       // 1: blocks sharing an A panel are XCD neighbours; 0: blocks sharing a W range are
  int kslices;           // split-K: slice s multiplies K range [512 s, 512 s + 512) into slab s of C0
  int64_t ldw;           // row stride of W in elements (== full K)
  int64_t slab_stride;   // elements between consecutive fp32 slabs of C0
  float* pmax; int32_t* pidx; float* psum;
  const int32_t* labels; float* plab;   // optional (ARGMAX): logit of column labels[row] per partial
  // COLLECT mode re-uses idle fields (a bigger kernel-argument block costs the ARGMAX kernel a spill):
  //   bias = threshold per row, pidx = count per row, pmax = candidate values [M, cap],
  //   C1 = candidate columns int32 [M, cap], act = cap
  int total_items;       // kslices * per-slice items (grid may be smaller: persistent blocks)
};

__device__ __forceinline__ float as_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

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

// s_waitcnt vmcnt(n) needs an immediate: dispatch over the even counts that can occur.
// lgkmcnt(0) rides along so this wave's LDS writes/reads are complete before the barrier.
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 20: asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
}

// FULL: K == 512, every unrolled load / MFMA is unpredicated.  A run-time predicate (even a
// wave-uniform one) makes hipcc branch around every load and wait vmcnt(0) per element.
template <typename AT, int MODE, bool FULL>
__global__ __launch_bounds__(256, 2) void gemm_as_kernel(AsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool IS_ARGMAX = MODE == STREAM_ARGMAX || MODE == STREAM_ARGMAX_LAB;
  constexpr bool IS_COLLECT = MODE == STREAM_COLLECT;
  constexpr bool HAS_LAB = MODE == STREAM_ARGMAX_LAB;  // 16 more live VGPRs: its own instantiation
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- persistent workgroups: the grid is at most ~2 blocks per CU and each block walks work
  // items (K slice, A panel, column range) with stride gridDim.x (a multiple of 8, so a block
  // keeps its XCD class).  Launching one tiny workgroup per item is dispatch-bound: 21 504
  // workgroups of the cross-K/V projection cost 406 us with every instruction ablated.
  const int per_slice = p.total_items / p.kslices;
  for (int item = blockIdx.x; item < p.total_items; item += gridDim.x) {
  const int slice = item / per_slice, bslot = item % per_slice;
  const int xcd = bslot & 7, idx = bslot >> 3;
  int panel, ns;
  if (p.panel_major) { ns = idx % p.ns; panel = (idx / p.ns) * 8 + xcd; }
  else { panel = idx % p.panels; ns = (idx / p.panels) * 8 + xcd; }
  if (panel >= p.panels || ns >= p.ns) continue;
  const int tiles_total = (p.N + TILE_N - 1) / TILE_N;
  const int tpb = (tiles_total + p.ns - 1) / p.ns;
  const int t0 = ns * tpb, t1 = min(t0 + tpb, tiles_total);
  const int m0 = panel * 128 + wave * 32;
  if (t0 >= t1) {  // empty column range (ns is rounded up to a multiple of 8)
    if (IS_ARGMAX && lane < 32 && m0 + lane < p.M) {
      const int64_t o = (int64_t)(m0 + lane) * p.ns + ns;
      p.pmax[o] = -INFINITY; p.pidx[o] = 0x7fffffff; p.psum[o] = 0.f;
      if (HAS_LAB && p.plab) p.plab[o] = -INFINITY;
    }
    continue;
  }
  // every wave must be done reading the ring (and the bias) of the previous item
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (MODE == STREAM_COLLECT) {  // candidate staging count (see collect); ordered by the first tile's barrier
    if (tid == 0) {
      const unsigned cb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + BIAS_OFF;
      const int zero = 0;
      asm volatile("ds_write_b32 %0, %1" :: "v"(cb), "v"(zero) : "memory");
    }
  }

  const AT* Ap = reinterpret_cast<const AT*>(p.A) + slice * 512;
  const bf16_t* Wp = p.W + slice * 512;
  const float* biasp = slice == 0 ? p.bias : nullptr;
  const int ksn = FULL ? 16 : (p.K >> 5);
  const int kbytes = FULL ? 1024 : p.K * 2;

  // ---- A fragments of this wave's 32 rows, whole K, resident for the life of the block
  int arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) arow[mt] = min(m0 + mt * 16 + fr, p.M - 1);
  bf16x8 a[2][16];

  const int col0 = t0 * TILE_N;
  // per-lane source pointers of this wave's 4 rows at tile 0; tile t adds t * 16 rows
  const unsigned char* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    wrow[i] = reinterpret_cast<const unsigned char*>(Wp + (int64_t)(wave * 4 + i) * p.ldw) + ((lane ^ (wave * 4 + i)) << 4);
  const int64_t wtile = (int64_t)TILE_N * p.ldw * 2;  // bytes between consecutive tiles
  auto stage = [&](int tile, int slot) {
    if (CARE_AS_DBG & 4) return;
    const bool whole = tile * TILE_N + TILE_N <= p.N;  // uniform; only the last tile can be ragged
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 4 + i;
      if (FULL || (lane ^ row) * 16 < kbytes) {
        const unsigned char* g = wrow[i] + (int64_t)tile * wtile;
        if (!whole) g -= (int64_t)max(tile * TILE_N + row - (p.N - 1), 0) * p.ldw * 2;  // clamp to the last W row
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + slot * TILE_BYTES + row * 1024),
                                         16, 0, 0);
      }
    }
  };

  f32x4 acc[2];
  // B fragments are read BDEPTH k-steps ahead of the MFMAs that use them: one ds_read_b128 has
  // ~130+ cycles of latency but feeds only 32 cycles of MFMA, so a shallow prefetch leaves the
  // loop LDS-latency-bound (measured: ~2600 cycles per tile instead of ~600).
  constexpr int BDEPTH = IS_ARGMAX ? 4 : 8;  // the argmax modes need the registers (running max, index, sum, reference: they would spill)
  // chunk (ks*4 + fg) ^ fr  ==  (ks & ~3)*4 + ((ks & 3) ^ (fr >> 2))*4 + (fg ^ (fr & 3)): per lane
  // only FOUR distinct byte offsets (r = ks & 3) plus the compile-time 256 * (ks >> 2).
  int boff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) boff[r] = fr * 1024 + (((r ^ (fr >> 2)) * 4 + (fg ^ (fr & 3))) << 4);
  auto compute = [&](int slot) {
    const unsigned char* sb = smem + slot * TILE_BYTES;
    bf16x8 fb[BDEPTH];
#pragma unroll
    for (int ks = 0; ks < BDEPTH; ++ks)
      if (FULL || ks < ksn) fb[ks] = *reinterpret_cast<const bf16x8*>(sb + boff[ks & 3] + (ks >> 2) * 256);
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (FULL || ks < ksn) {
        const bf16x8 b = fb[ks % BDEPTH];
        if (ks + BDEPTH < 16 && (FULL || ks + BDEPTH < ksn))
          fb[ks % BDEPTH] = *reinterpret_cast<const bf16x8*>(sb + boff[(ks + BDEPTH) & 3] + ((ks + BDEPTH) >> 2) * 256);
        if (CARE_AS_DBG & 2) { asm volatile("" :: "v"(b)); continue; }
        if constexpr (MODE != STREAM_STORE) {  // D[row of A][col = W row]: a lane sees 1 column, 4 rows
          acc[0] = care_mfma_16x16x32_h16(a[0][ks], b, acc[0], 0, 0, 0);
          acc[1] = care_mfma_16x16x32_h16(a[1][ks], b, acc[1], 0, 0, 0);
        } else {  // swapped: a lane holds 4 CONSECUTIVE output columns of one row
          acc[0] = care_mfma_16x16x32_h16(b, a[0][ks], acc[0], 0, 0, 0);
          acc[1] = care_mfma_16x16x32_h16(b, a[1][ks], acc[1], 0, 0, 0);
        }
#ifndef CARE_AS_NOPIN
        __builtin_amdgcn_sched_barrier(0);  // pinned k-step order (see compute_woven)
#endif
      }
  };

  // Store mode: acc[mt][j] = C[row m0 + 16 mt + fr][col 16 tile + 4 fg + j].  Everything that
  // selects a destination is wave-uniform (a 16-column tile never straddles n_split).
  // Returns the number of VM store instructions issued when that number is exact (2), else -1.
  auto store_tile = [&](int tile, float4 bv) -> int {
    const int cbase = tile * TILE_N;
    const bool second = cbase >= p.n_split;
    unsigned char* C = reinterpret_cast<unsigned char*>(second ? p.C1 : p.C0) + (int64_t)slice * p.slab_stride * 4;
    const int64_t ld = second ? p.ldc1 : p.ldc0;
    const bool isb = (second ? p.c1_bf16 : p.c0_bf16) != 0;
    const int col = cbase + fg * 4;
    const int cc = col - (second ? p.n_split : 0);
    const bool vec = (m0 + 32 <= p.M) && (cbase + TILE_N <= p.N) && (ld % 4 == 0);
    if (CARE_AS_DBG & 1) { asm volatile("" :: "v"(acc[0]), "v"(acc[1])); return 0; }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float v[4] = {acc[mt][0] + bv.x, acc[mt][1] + bv.y, acc[mt][2] + bv.z, acc[mt][3] + bv.w};
      if (p.act == CARE_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
      } else if (p.act == CARE_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = as_gelu(v[j]);
      }
      const int row = m0 + mt * 16 + fr;
      const int64_t o = (int64_t)row * ld + cc;
      if (vec) {
        if (isb) {
          bf16x4 ob;
#pragma unroll
          for (int j = 0; j < 4; ++j) ob[j] = (bf16_t)v[j];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(C) + o) = ob;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else if (row < p.M) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < p.N) {
            if (isb) reinterpret_cast<bf16_t*>(C)[o + j] = (bf16_t)v[j];
            else reinterpret_cast<float*>(C)[o + j] = v[j];
          }
      }
    }
    return vec ? 2 : -1;
  };

  // running (max, argmax, sum-exp) of the 8 rows this lane sees, over its column residue
  float rm[8], rs[8], rref[IS_ARGMAX ? 8 : 1], rl[HAS_LAB ? 8 : 1];
  int ri[8], lab[HAS_LAB ? 8 : 1];
  if constexpr (IS_ARGMAX) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // finite sentinel, not -inf: an update with a masked (-inf) logit then gives exp(-inf) = 0, not NaN
      rm[i] = -1e30f; rs[i] = 0.f; ri[i] = 0x7fffffff; rref[i] = -1e30f;
      const int row = min(m0 + (i >> 2) * 16 + fg * 4 + (i & 3), p.M - 1);
      if constexpr (HAS_LAB) { rl[i] = -INFINITY; lab[i] = p.labels ? p.labels[row] : -1; }
    }
  }

  // COLLECT: the thresholds of the 8 rows this lane sees (+inf for rows past M: nothing is appended)
  float cth[IS_COLLECT ? 8 : 1];
  if constexpr (IS_COLLECT) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + (i >> 2) * 16 + fg * 4 + (i & 3);
      cth[i] = row < p.M ? p.bias[row] : INFINITY;
    }
  }
  // Candidates are staged in the block's LDS bias area (idle in this mode): [0] = count, then
  // (value, row-in-panel << 24 | column) pairs, flushed to the global lists once per work item.  A
  // global atomic with return inside the tile loop is a full memory round trip that also drains the
  // W tiles in flight, and about every second tile of a wave holds a candidate (417 us per launch
  // against 257 us for the statistics pass); LDS atomics do not touch vmcnt (asm: through C++ hipcc
  // would put a vmcnt(0) in front of them, possible alias with the LDS-DMA).
  constexpr int LCAP = (BIAS_MAX * 4 - 8) / 8;
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + BIAS_OFF;
  auto append_global = [&](int row, float v0, int c0) {
    const int pos = atomicAdd(&p.pidx[row], 1);
    if (pos < p.act) {
      p.pmax[(int64_t)row * p.act + pos] = v0;
      reinterpret_cast<int32_t*>(p.C1)[(int64_t)row * p.act + pos] = c0;
    }
  };
  auto collect = [&](const f32x4 (&av)[2], int tile) {
    const int c0 = tile * TILE_N + fr;
    // ONE wave-uniform test per tile: a divergent branch per logit costs more than the MFMAs
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) hit |= av[i >> 2][i & 3] >= cth[IS_COLLECT ? i : 0];
    if (__builtin_expect(__any(hit && c0 < p.N), 0)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v0 = av[i >> 2][i & 3];
        if (c0 < p.N && v0 >= cth[IS_COLLECT ? i : 0]) {
          const int rloc = wave * 32 + (i >> 2) * 16 + fg * 4 + (i & 3);
          int pos;
          const int one = 1;
          asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(lbase), "v"(one) : "memory");
          if (pos < LCAP) {
            const unsigned a = lbase + 8 + (unsigned)pos * 8;
            const int packed = (rloc << 24) | c0;
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(a), "v"(v0), "v"(packed) : "memory");
          } else {
            append_global(m0 - wave * 32 + rloc, v0, c0);  // staging area full
          }
        }
      }
    }
  };
  // flush of the staged candidates: after the tile loop of the item, all 256 threads
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  auto collect_flush = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int n;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(n) : "v"(lbase) : "memory");
    n = min(n, LCAP);
    for (int e = tid; e < n; e += 256) {
      i32x2 pr;
      const unsigned a = lbase + 8 + (unsigned)e * 8;
      asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pr) : "v"(a) : "memory");
      append_global(m0 - wave * 32 + (int)((unsigned)pr[1] >> 24), __builtin_bit_cast(float, pr[0]), pr[1] & 0xffffff);
    }
  };

  // online (max, sum-exp) of logit i (= m-tile i>>2, register i&3) of a finished tile, with ONE exp
  // per logit: e = exp(-|m - v|) is the rescale factor when v is the new max and the new term otherwise
  // Per logit: the exact running maximum and its column (compare, select, max), and the sum of
  // exponentials against a LAZY reference rref (subtract, exp, add): 5 vector instructions and one exp,
  // no branch, instead of the 13 + 1 of the textbook online softmax (select-heavy: its new maximum
  // rescales the sum every time).  The statistics cost more issue cycles than the 32 MFMAs of a tile,
  // so this is what sets the kernel's matrix-pipe occupancy.  rref is looked after ONCE PER TILE
  // (argmax_ref): set from the first tile's logits, moved - with the sum rescaled - when the running
  // maximum has got more than 20 ahead of it (e^20 per term is far from fp32 overflow; a row whose
  // logits jump by more than 88 inside one tile would overflow the sum: the final merge then falls
  // back to max-only, which is what such a sum equals in fp32 anyway).
  auto argmax_one = [&](const f32x4 (&av)[2], int c0, int i) {
    const float v0 = c0 < p.N ? av[i >> 2][i & 3] : -INFINITY;
    ri[i] = v0 > rm[i] ? c0 : ri[i];
    rm[i] = fmaxf(rm[i], v0);
    rs[i] += __expf(v0 - rref[i]);  // a masked (-inf) logit adds exp(-inf) = 0
    if constexpr (HAS_LAB) rl[HAS_LAB ? i : 0] = c0 == lab[HAS_LAB ? i : 0] ? v0 : rl[HAS_LAB ? i : 0];
  };
  // before the statistics of a tile: first tile -> the reference is the tile's own logit; later ->
  // wave-uniform test whether any running maximum has left its reference behind (rare)
  auto argmax_ref = [&](const f32x4 (&av)[2], bool first) {
    if (first) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rref[i] = fmaxf(av[i >> 2][i & 3], -1e30f);
      return;
    }
    float far = rm[0] - rref[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) far = fmaxf(far, rm[i] - rref[i]);
    if (__builtin_amdgcn_ballot_w64(far > 20.0f) != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        rs[i] *= __expf(rref[i] - rm[i]);
        rref[i] = rm[i];
      }
    }
  };
  auto argmax_update = [&](const f32x4 (&av)[2], int tile, bool first) {
    argmax_ref(av, first);
    const int c0 = tile * TILE_N + fr;
#pragma unroll
    for (int i = 0; i < 8; ++i) argmax_one(av, c0, i);
  };
  // Software pipeline of the argmax mode: the MFMAs of tile t with the statistics of tile t-1 (held
  // in accp) woven in by hand, one logit after every second k-step, and the order PINNED by a
  // sched_barrier per k-step.  Left to the scheduler (statistics in the same region as the MFMAs)
  // hipcc issues the 16 dependent MFMAs of one accumulator back to back - each then waits out the
  // full MFMA latency and the tile takes 1.8x as long; done strictly after the MFMAs the statistics
  // cost about as many issue cycles again (13 VALU ops + 1 exp per logit vs 32 MFMAs).
  f32x4 accp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  auto compute_woven = [&](int slot, int prev_tile) {
    const unsigned char* sb = smem + slot * TILE_BYTES;
    const int c0 = prev_tile * TILE_N + fr;
    bf16x8 fb[BDEPTH];
#pragma unroll
    for (int ks = 0; ks < BDEPTH; ++ks)
      if (FULL || ks < ksn) {
        if (CARE_AS_DBG & 32) { fb[ks] = bf16x8{}; asm volatile("" : "+v"(fb[ks])); }
        else fb[ks] = *reinterpret_cast<const bf16x8*>(sb + boff[ks & 3] + (ks >> 2) * 256);
      }
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (FULL || ks < ksn) {
        const bf16x8 b = fb[ks % BDEPTH];
        if (ks + BDEPTH < 16 && (FULL || ks + BDEPTH < ksn) && !(CARE_AS_DBG & 32))
          fb[ks % BDEPTH] = *reinterpret_cast<const bf16x8*>(sb + boff[(ks + BDEPTH) & 3] + ((ks + BDEPTH) >> 2) * 256);
        if (CARE_AS_DBG & 2) asm volatile("" :: "v"(b));
        else {
          acc[0] = care_mfma_16x16x32_h16(a[0][ks], b, acc[0], 0, 0, 0);
          acc[1] = care_mfma_16x16x32_h16(a[1][ks], b, acc[1], 0, 0, 0);
        }
      }
      if ((ks & 1) && !(CARE_AS_DBG & 16)) argmax_one(accp, c0, ks >> 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: up to AHEAD tiles in flight (4 VM ops each), then the A loads: one combined latency
  const int ntl = t1 - t0;
  const int npro = min(AHEAD, ntl);
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < ntl) stage(t0 + i, i);
  __builtin_amdgcn_sched_barrier(0);

  // A fragment loads: issued right behind the prologue DMAs so both latencies overlap
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
    if (FULL || ks < ksn) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (CARE_AS_DBG & 8) { a[mt][ks] = bf16x8{}; asm volatile("" : "+v"(a[mt][ks])); }
        else a[mt][ks] = load_a_frag<AT>(Ap + (int64_t)arow[mt] * p.lda + ks * 32 + fg * 8);
      }
    }
  __builtin_amdgcn_sched_barrier(0);

  // ---- bias slice of this block's column range -> LDS, issued BEHIND the prologue DMAs and the A
  // loads so that all three latencies overlap (iteration 0 drains everything anyway); read back
  // with an asm ds_read, see below
  if (MODE == STREAM_STORE && biasp) {
    float* sb = reinterpret_cast<float*>(smem + BIAS_OFF);
    const int nb = min((t1 - t0) * TILE_N, p.N - col0);
    for (int i = tid; i < ((nb + 3) & ~3); i += 256) sb[i] = i < nb ? biasp[col0 + i] : 0.f;
  }

  // stage W tile `tile` into ring slot `slot`: wave w copies rows 4w..4w+3, one 1-KiB row per
  // LDS-DMA instruction; lane = chunk slot, source chunk = slot ^ (row & 15).  4 VM ops / wave.

  // VM operations this wave issued in each of the last AHEAD iterations, oldest first:
  // DMA count and store count (-1 = unknown: guarded scalar stores -> fall back to vmcnt(0)).
  int hist_dma[AHEAD], hist_st[AHEAD];
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) { hist_dma[i] = 0; hist_st[i] = 0; }

  for (int t = t0; t < t1; ++t) {
    const int it = t - t0;
    const int slot = it % RING;
    // ---- wait until tile t's DMA has landed, leaving every YOUNGER VM op in flight.
    // Tile t was staged in iteration it-AHEAD before that iteration's stores (or in the
    // prologue).  Younger ops = stores(it-AHEAD) + sum over the AHEAD-1 iterations since of
    // (DMA + stores) + the prologue tiles staged after tile t (first AHEAD iterations only).
    int younger = it < AHEAD ? 4 * (npro - 1 - it) : 0;
    bool exact = true;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
      if (hist_st[i] < 0) exact = false;
      younger += hist_st[i] + (i > 0 ? hist_dma[i] : 0);
    }
    // iteration 0 drains everything (A fragments + prologue tiles were issued together)
    wait_vm(exact && it > 0 ? younger : 0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // bias of this lane's 4 columns: an asm ds_read (hipcc would put a vmcnt(0) drain in front
    // of a C++ LDS read because the in-flight LDS-DMA might alias it)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == STREAM_STORE && biasp) {
      const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
      const unsigned addr = base + (unsigned)(BIAS_OFF + ((t * TILE_N - col0) + fg * 4) * 4);
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(addr) : "memory");
    }

    // ---- refill the ring slot consumed in the previous iteration
    int n_dma = 0;
    if (t + AHEAD < t1) { stage(t + AHEAD, (it + AHEAD) % RING); n_dma = 4; }
    __builtin_amdgcn_sched_barrier(0);

    int n_st = 0;
    if constexpr (MODE == STREAM_STORE) {
      compute(slot);
      __builtin_amdgcn_sched_barrier(0);
      n_st = store_tile(t, bv);
    } else if constexpr (HAS_LAB) {  // 16 more live registers: no room for a second accumulator pair
      compute(slot);
      argmax_update(acc, t, t == t0);
    } else if constexpr (IS_COLLECT) {
      // appends are rare extra VM operations: the counted waits then over-wait (never under-wait)
      compute(slot);
      collect(acc, t);
    } else {
      // iteration 0 has no previous tile: a column index past N masks the dummy statistics (no branch)
      if (it > 0) argmax_ref(accp, it == 1);
      compute_woven(slot, it > 0 ? t - 1 : (1 << 26));
      accp[0] = acc[0]; accp[1] = acc[1];
    }
#pragma unroll
    for (int i = 0; i + 1 < AHEAD; ++i) { hist_dma[i] = hist_dma[i + 1]; hist_st[i] = hist_st[i + 1]; }
    hist_dma[AHEAD - 1] = n_dma;
    hist_st[AHEAD - 1] = n_st;
  }

  if constexpr (IS_COLLECT) collect_flush();
  if constexpr (IS_ARGMAX) {
    if constexpr (!HAS_LAB) argmax_update(accp, t1 - 1, t1 - t0 == 1);  // the last tile's statistics
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // the sum relative to the true maximum (rows that saw no logit: rm = rref = -1e30, s = 0)
      float m = rm[i], lv = HAS_LAB ? rl[HAS_LAB ? i : 0] : 0.f;
      float s = rs[i] > 0.f ? rs[i] * __expf(rref[IS_ARGMAX ? i : 0] - rm[i]) : 0.f;
      if (!(s < 3.0e38f)) s = 1.0f;  // the sum overflowed (a > 88 jump inside one tile): it is its largest term
      int id = ri[i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        if constexpr (HAS_LAB) lv = fmaxf(lv, __shfl_xor(lv, o, 64));
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
        if (HAS_LAB && p.plab) p.plab[o] = lv;
      }
    }
  }
  }  // work items
}

template <typename AT, int MODE>
int launch_as(const AsArgs& p, int blocks, hipStream_t st) {
  if (p.K == 512) hipLaunchKernelGGL((gemm_as_kernel<AT, MODE, true>), dim3(blocks), dim3(256), LDS_BYTES, st, p);
  else hipLaunchKernelGGL((gemm_as_kernel<AT, MODE, false>), dim3(blocks), dim3(256), LDS_BYTES, st, p);
  return care_launch_status();
}

int as_check(const void* A, int64_t lda, int a_dtype, const void* W, int M, int N, int K) {
  if (!A || !W || M <= 0 || N <= 0 || K <= 0) return CARE_EINVAL;
  if (a_dtype != CARE_F32 && a_dtype != CARE_BF16) return CARE_EDTYPE;
  if (K % 128 != 0) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % 8) != 0) return CARE_EALIGN;
  return 0;
}

// Number of column ranges per panel.  Every extra range re-loads the panel's A fragments (the
// expensive, fragment-shaped loads, worth ~24 W tiles of time), every launch "round" of 512
// persistent workgroups (2 per CU) costs one item time, and an item sweeps tiles / ns tiles:
//     cost(ns) = ceil(panels * ns / 512) * (24 + tiles / ns)
// minimised over ns in {1, 2, 4} (panel-major mapping only) and the multiples of 8 (one share per
// XCD); ties go to the smaller split.  It reproduces the measured optima - 1 panel: as many ranges
// as there are tiles (<= 512), 32 panels: ns = 16, 128 panels: ns = 4 (N = 512 .. 10547), 256 panels:
// ns = 2 - and picks ns = 8 for the vocabulary GEMM at 160 panels, where 4 gives 1.25 rounds of big
// items (measured: +5% on the beam pass).
int pick_ns(int panels, int tiles_total) {
  if (const char* e = getenv("CARE_AS_NS")) return atoi(e);  // tuning override (tools/gemm_bench.py)
  auto cost = [&](int ns) {
    const long rounds = ((long)panels * ns + 511) / 512;
    return (double)rounds * (24.0 + (double)tiles_total / ns);
  };
  auto next = [](int ns) { return ns < 4 ? ns * 2 : (ns == 4 ? 8 : ns + 8); };
  double best = 1e30;
  for (int ns = 1; ns <= tiles_total && ns <= 512; ns = next(ns)) best = cost(ns) < best ? cost(ns) : best;
  for (int ns = 1; ns <= tiles_total && ns <= 512; ns = next(ns))
    if (cost(ns) <= best) return ns;  // the smallest split among the best
  return 1;
}

// Grid size and XCD mapping of a launch.  The sharers of the BIGGER operand are made XCD
// neighbours; panel-major needs the panel count rounded up to a multiple of 8.
int plan_stream(AsArgs& p, bool has_bias) {
  const int tiles_total = (p.N + TILE_N - 1) / TILE_N;
  while (has_bias && ((tiles_total + p.ns - 1) / p.ns) * TILE_N > BIAS_MAX) p.ns += 8;
  p.panel_major = ((long)p.M > (long)p.N || p.ns % 8 != 0) ? 1 : 0;
  if (const char* e = getenv("CARE_AS_MAP")) p.panel_major = atoi(e);
  const int per_slice = p.panel_major ? ((p.panels + 7) / 8) * 8 * p.ns : p.panels * p.ns;
  p.total_items = per_slice * p.kslices;
  int max_blocks = 512;  // 2 workgroups per CU (LDS 72 KiB, <= 256 VGPRs each)
  if (const char* e = getenv("CARE_AS_BLOCKS")) max_blocks = atoi(e);
  return p.total_items < max_blocks ? p.total_items : max_blocks;
}

}  // namespace

// the 256-row-panel store kernel for large row counts (csrc/gemm_store32.hip)
extern "C" int care_store32_applies(int M, int N, int K, int a_dtype, int n_split, int has_bias_cols);
extern "C" int care_store32_launch(const void* A, int64_t lda, const void* W, const float* bias, void* C0, int64_t ldc0,
                                   int c0_bf16, void* C1, int64_t ldc1, int c1_bf16, int n_split, int M, int N, int act,
                                   void* stream);

extern "C" int care_gemm_bf16(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias, void* C0,
                              int64_t ldc0, int c0_dtype, void* C1, int64_t ldc1, int c1_dtype, int n_split, int M,
                              int N, int K, int act, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (K > 512) return CARE_ESHAPE;  // use care_gemm_bf16_splitk
  if (!C0 || n_split <= 0 || n_split > N || (n_split < N && !C1)) return CARE_EINVAL;
  if (n_split % 16 != 0 && n_split != N) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if (care_store32_applies(M, N, K, a_dtype, n_split, 0)) {
    const int rc32 = care_store32_launch(A, lda, W, bias, C0, ldc0, c0_dtype == CARE_BF16, C1, ldc1, c1_dtype == CARE_BF16,
                                         n_split, M, N, act, stream);
    if (rc32 != CARE_ESHAPE && rc32 != CARE_EALIGN) return rc32;  // shapes / alignments it does not take: the 128-row kernel
  }
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias;
  p.C0 = C0; p.ldc0 = ldc0; p.c0_bf16 = c0_dtype == CARE_BF16;
  p.C1 = C1; p.ldc1 = ldc1; p.c1_bf16 = c1_dtype == CARE_BF16;
  p.n_split = n_split; p.M = M; p.N = N; p.K = K; p.act = act;
  p.panels = (M + 127) / 128;
  p.kslices = 1; p.ldw = K; p.slab_stride = 0;
  p.ns = pick_ns(p.panels, (N + TILE_N - 1) / TILE_N);
  const int blocks = plan_stream(p, bias != nullptr);
  hipStream_t st = (hipStream_t)stream;
  return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_STORE>(p, blocks, st)
                              : launch_as<float, STREAM_STORE>(p, blocks, st);
}

// at least `min_parts` column ranges (the fused beam selection needs >= beam_size sub-maxima per row)
static int pick_ns_min(int panels, int tiles_total, int min_parts) {
  const int ns = pick_ns(panels, tiles_total);
  return ns >= min_parts ? ns : (min_parts + 7) / 8 * 8;
}

extern "C" int care_argmax_parts_bf16(int M, int N) {
  if (M <= 0 || N <= 0) return CARE_EINVAL;
  return pick_ns((M + 127) / 128, (N + TILE_N - 1) / TILE_N);
}

// the 256-row-panel kernels for large row counts (csrc/gemm_vocab.hip)
extern "C" int care_vocab32_applies(int M, int N, int K, int a_dtype, int has_labels);
extern "C" int care_vocab32_launch(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                                   int M, int N, int ns, void* stream);
extern "C" int care_collect32_launch(const void* A, int64_t lda, const void* W, const float* thr, int32_t* cnt, float* cval,
                                     int32_t* cidx, int cap, int M, int N, int ns, void* stream);
extern "C" int care_vocab32_ranges(int M, int N, int min_parts);
extern "C" int care_vocab32_launch_tiles(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                                         float* tile_max, int M, int N, int ns, void* stream);

// column ranges of the `_min` statistics pass (beam search): on the 256-row kernel its own rule (whole launch rounds)
static bool min_uses_v32(int M, int N, int K, int a_dtype, int min_parts) {
  return min_parts <= 32 && care_vocab32_applies(M, N, K, a_dtype, 0);
}
extern "C" int care_argmax_parts_bf16_min(int M, int N, int K, int a_dtype, int min_parts) {
  if (M <= 0 || N <= 0 || K <= 0 || min_parts <= 0) return CARE_EINVAL;
  if (min_uses_v32(M, N, K, a_dtype, min_parts)) return care_vocab32_ranges(M, N, min_parts);
  return pick_ns_min((M + 127) / 128, (N + TILE_N - 1) / TILE_N, min_parts);
}

static int gemm_argmax_bf16(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax, int32_t* pidx,
                           float* psum, const int32_t* labels, float* plab, int M, int N, int K, int min_parts,
                           bool min_entry, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!pmax || !pidx || !psum || ((labels != nullptr) != (plab != nullptr))) return CARE_EINVAL;
  if (K > 512) return CARE_ESHAPE;
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.M = M; p.N = N; p.K = K; p.n_split = N;
  p.kslices = 1; p.ldw = K;
  p.pmax = pmax; p.pidx = pidx; p.psum = psum; p.labels = labels; p.plab = plab;
  p.panels = (M + 127) / 128;
  p.ns = pick_ns_min(p.panels, (N + TILE_N - 1) / TILE_N, min_parts);
  // 256-row panels: half the W streaming.  The greedy entry keeps the range count of care_argmax_parts_bf16, the
  // `_min` entry the one of care_argmax_parts_bf16_min.
  if (care_vocab32_applies(M, N, K, a_dtype, labels != nullptr)) {
    if (min_entry && min_uses_v32(M, N, K, a_dtype, min_parts)) p.ns = care_vocab32_ranges(M, N, min_parts);
    return care_vocab32_launch(A, lda, W, pmax, pidx, psum, M, N, p.ns, stream);
  }
  const int blocks = plan_stream(p, false);
  hipStream_t st = (hipStream_t)stream;
  if (labels && plab)
    return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_ARGMAX_LAB>(p, blocks, st)
                                : launch_as<float, STREAM_ARGMAX_LAB>(p, blocks, st);
  return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_ARGMAX>(p, blocks, st)
                              : launch_as<float, STREAM_ARGMAX>(p, blocks, st);
}

extern "C" int care_gemm_argmax_bf16(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                                     int32_t* pidx, float* psum, const int32_t* labels, float* plab, int M, int N,
                                     int K, void* stream) {
  return gemm_argmax_bf16(A, lda, a_dtype, W, pmax, pidx, psum, labels, plab, M, N, K, 1, false, stream);
}

extern "C" int care_gemm_argmax_bf16_min(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                                         int32_t* pidx, float* psum, int M, int N, int K, int min_parts,
                                         void* stream) {
  if (min_parts <= 0) return CARE_EINVAL;
  return gemm_argmax_bf16(A, lda, a_dtype, W, pmax, pidx, psum, nullptr, nullptr, M, N, K, min_parts, true, stream);
}

// The `_min` statistics pass that also writes the maximum of every (32-column tile, row): tile_max [ceil(N / 32), M]
// fp32 - the map the sparse second pass (care_beam_sparse_collect) works from.  Shapes of the 256-row kernel only
// (care_beam_sparse_applies); the partials are exactly those of care_gemm_argmax_bf16_min.
extern "C" int care_gemm_argmax_bf16_tiles(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                                           int32_t* pidx, float* psum, float* tile_max, int M, int N, int K,
                                           int min_parts, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!pmax || !pidx || !psum || !tile_max || min_parts <= 0) return CARE_EINVAL;
  if (!min_uses_v32(M, N, K, a_dtype, min_parts)) return CARE_ESHAPE;
  return care_vocab32_launch_tiles(A, lda, W, pmax, pidx, psum, tile_max, M, N, care_vocab32_ranges(M, N, min_parts), stream);
}

extern "C" int care_gemm_collect_bf16(const void* A, int64_t lda, int a_dtype, const void* W, const float* thr,
                                      int32_t* cnt, float* cval, int32_t* cidx, int cap, int M, int N, int K,
                                      void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!thr || !cnt || !cval || !cidx || cap <= 0) return CARE_EINVAL;
  if (K > 512 || a_dtype != CARE_BF16 || N >= (1 << 24)) return CARE_ESHAPE;  // columns are staged in 24 bits
  if (care_vocab32_applies(M, N, K, a_dtype, 0))  // the range count only balances the launch here (lists are per row)
    return care_collect32_launch(A, lda, W, thr, cnt, cval, cidx, cap, M, N, care_vocab32_ranges(M, N, 1), stream);
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.M = M; p.N = N; p.K = K; p.n_split = N;
  p.kslices = 1; p.ldw = K;
  p.bias = thr; p.pidx = cnt; p.pmax = cval; p.C1 = cidx; p.act = cap;  // see AsArgs
  p.panels = (M + 127) / 128;
  p.ns = pick_ns(p.panels, (N + TILE_N - 1) / TILE_N);
  const int blocks = plan_stream(p, false);
  return launch_as<bf16_t, STREAM_COLLECT>(p, blocks, (hipStream_t)stream);
}

// Split-K form for K > 512 (FFN dense2, wide feature embedders): K/512 slices, slice s writes its
// partial product (bias in slice 0) to fp32 slab s = C + s * slab_stride.  The consumer
// (care_add_ln with nslab = K/512) sums the slabs, so no separate reduction pass exists.
extern "C" int care_gemm_bf16_splitk(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias,
                                     float* C, int64_t ldc, int64_t slab_stride, int M, int N, int K, void* stream) {
  int rc = as_check(A, lda, a_dtype, W, M, N, K);
  if (rc) return rc;
  if (!C) return CARE_EINVAL;
  if (K % 512 != 0 || K < 1024) return CARE_ESHAPE;
  AsArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias;
  p.C0 = C; p.ldc0 = ldc; p.c0_bf16 = 0; p.n_split = N; p.M = M; p.N = N; p.K = 512; p.act = CARE_ACT_NONE;
  p.panels = (M + 127) / 128;
  p.kslices = K / 512; p.ldw = K; p.slab_stride = slab_stride;
  p.ns = pick_ns(p.panels * p.kslices, (N + TILE_N - 1) / TILE_N);
  const int blocks = plan_stream(p, bias != nullptr);
  hipStream_t st = (hipStream_t)stream;
  return a_dtype == CARE_BF16 ? launch_as<bf16_t, STREAM_STORE>(p, blocks, st)
                              : launch_as<float, STREAM_STORE>(p, blocks, st);
}
