// gemm_vocab.hip - the greedy vocabulary projection at large row counts:
//     per row  (max, argmax, sum-exp)  of  hidden [M, 512] bf16  x  W^T [512, V] bf16   (Head.py:26-32 +
//     log_softmax + top-1: Translator.py:127, Beam.py:58-70), the [M, V] logits never stored.
//
// Why a second kernel beside gemm_as.hip's STREAM_ARGMAX mode.  Ablation of that kernel at M = 32768
// (tools/variant_lib.py, round 2): 381 us in all; with the MFMAs, the statistics AND the B-fragment reads
// removed it still takes 177 us - the time to stream W from L2 into LDS: every 128-row panel sweeps
// the whole 10.8 MB, 2.76 GB per launch at the ~16 TB/s the L2 -> LDS path gives (65 GB/s per CU),
// i.e. 128 flop per streamed byte caps the kernel at 2 PFLOP/s before any arithmetic happens.  MFMA
// issue (32 x 16x16x32 per 16-column tile), 13-instruction-per-logit statistics over 8 rows per lane
// and the fragment reads then ADD to that instead of hiding in it (114 + 62 + 76 us).  Here:
//   * a workgroup is 8 waves x 32 rows = a 256-row panel: W is streamed half as often (256 flop/B);
//   * W tiles are 32 columns x 512 k (32 KiB) through a 4-slot LDS ring, three in flight, one barrier
//     per tile = per 1024 MFMA cycles of every wave;
//   * v_mfma_f32_32x32x16_bf16 with the operands swapped (D = (X W^T)^T): half the MFMA instructions
//     per flop, and a lane ends up with 16 columns of ONE row, so the running statistics of a row are
//     four registers in one lane (maximum, its column, sum of exponentials, the sum's lazy reference)
//     instead of 8 rows x 3 registers per lane: maximum by v_max3 over the tile, ONE compare per tile
//     for "new maximum?" (the column search sits in a rarely taken branch), sub / exp / add per logit;
//   * the statistics of tile t are woven between the MFMAs of tile t + 1 (order pinned per k-step);
//   * the last tile of a range is simply fetched again past the end (constant vmcnt arithmetic).
// In-kernel stamps (round 5, tools/v32_ts.py, 32768 rows): a tile takes ~3700 cycles of a SIMD for the 2048 of its two waves'
// MFMAs.  A wave's 32 MFMAs take ~1740, not 1024: its partner's ~520 cycles of statistics issue ADD to them even though
// they come from the other wave (one vector issue port per SIMD), then the roles swap - 2 x (1024 + ~520 + LDS reads) +
// ~650 of barrier and landing waits per tile = the 0.46 of peak this kernel has run at since round 2; the matrix pipe is
// never short of work, the SIMD's issue port is.  (The sum of exponentials as two partial sums with v_pk_fma_f32 / v_pk_add_f32 -
// 16 vector instructions fewer per tile and wave: 350 against 323 us at 32768 rows on the same box.  Not kept.)
// Used by care_gemm_argmax_bf16 for bf16 A, K = 512, M >= 8192 without label logits; everything else
// (and the beam-search / scoring variants) stays on gemm_as.hip.
#include <cstdlib>
#include <type_traits>

#include "care_common.h"

namespace {

constexpr int VT_N = 32;                 // columns per W tile
constexpr int VT_BYTES = VT_N * 1024;    // K = 512 bf16
constexpr int V_RING = 4, V_AHEAD = 3;
constexpr int V_LDS = V_RING * VT_BYTES;
constexpr int V_ROWS = 256;              // rows per workgroup (8 waves x 32)
constexpr float V_LOG2E = 1.4426950408889634f;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct VArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W;
  float* pmax; int32_t* pidx; float* psum;
  int M, N, ns, panels, total_items;
  // collect mode (beam search, second pass): logits >= thr[row] are appended to the row's candidate list
  const float* thr; int32_t* cnt; float* cval; int32_t* cidx; int cap;
  // statistics mode, optional: the maximum of every (tile, row) -> tile_max[tile * M + row]: the map from which the
  // SPARSE second pass of the beam selection (csrc/beam_sparse.hip) recomputes only the tiles that hold a candidate
  float* tile_max;
};

enum { V_ARGMAX = 0, V_COLLECT = 1 };
constexpr int V_STAGE = 8192;                  // collect: LDS staging of the candidates of one work item
constexpr int V_LCAP = (V_STAGE - 8) / 8;

#ifndef V32_STAGGER
#define V32_STAGGER 1  // the two waves of a SIMD take MFMAs / statistics in opposite order (0: both multiply first)
#endif
#ifndef CARE_V32_DBG
#define CARE_V32_DBG 0  // ablation: 2 no MFMA, 16 no statistics, 32 no fragment reads
#endif

#if CARE_V32_DBG & 64  // tools (tools/variant_lib.py, tools/v32_ts.py): workgroup 0 stamps s_memtime per wave and tile
__device__ unsigned long long v32_stamps[8 * 64 * 4];
#define V32_STAMP(it, k) do { if (blockIdx.x == 0 && (it) < 64 && lane == 0) v32_stamps[(wave * 64 + (it)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define V32_STAMP(it, k) do { } while (0)
#endif

template <int BDEPTH, int MODE>
__global__ __launch_bounds__(512, 2) void vocab_argmax32_kernel(VArgs p) {
  constexpr bool COLLECT = MODE == V_COLLECT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tiles_total = (p.N + VT_N - 1) / VT_N;
  const int tpb = (tiles_total + p.ns - 1) / p.ns;

  // Work distribution: the panels x ns (panel, range) items in panel-major order, cut into gridDim.x contiguous
  // spans - a block's consecutive items are consecutive column ranges of the SAME panel, so a RUN of them keeps the
  // activations in registers and the W ring streaming straight through the range boundaries (only the statistics are
  // closed and written there).  Any item count balances: 4096 clips x beam 5 = 80 panels x 16 ranges = 5 per block.
  const int i_lo = (int)((long)blockIdx.x * p.total_items / gridDim.x), i_hi = (int)((long)(blockIdx.x + 1) * p.total_items / gridDim.x);
  for (int item = i_lo; item < i_hi;) {
    const int panel = item / p.ns, ra = item % p.ns;
    const int rb = min(p.ns, ra + (i_hi - item));  // ranges ra .. rb - 1 of this panel
    item += rb - ra;
    const int t0 = ra * tpb, t1 = min(rb * tpb, tiles_total);
    const int m0 = panel * V_ROWS + wave * 32;
    const int row = m0 + r;
    // ranges past the last tile (ns does not divide the tile count evenly) hold nothing
    if (!COLLECT && h == 0 && row < p.M)
      for (int range = max(ra, (tiles_total + tpb - 1) / tpb); range < rb; ++range) {
        const int64_t o = (int64_t)row * p.ns + range;
        p.pmax[o] = -INFINITY; p.pidx[o] = 0x7fffffff; p.psum[o] = 0.f;
      }
    if (t0 >= t1) continue;
    // every wave is done reading the ring of the previous run
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // collect: candidates are staged in LDS behind the ring - [0] = count, then (value, row-in-panel << 24 | column)
    // pairs - and flushed to the global lists once per run (a global atomic with return inside the tile loop
    // is a memory round trip that also drains the W tiles in flight); LDS atomics via asm (no vmcnt involvement)
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + V_LDS;
    if (COLLECT && tid == 0) {  // ordered before the first append by the first tile's barrier
      const int zero = 0;
      asm volatile("ds_write_b32 %0, %1" ::"v"(lbase), "v"(zero) : "memory");
    }
    const float thr = COLLECT ? (row < p.M ? p.thr[row] : INFINITY) : 0.f;

    // ---- W tile -> ring slot: wave w copies rows 4w .. 4w + 3 of the 32 (one 1-KiB row per DMA
    // instruction); lane = chunk slot, source chunk = slot ^ (row & 15): conflict-free ds_read_b128
    const unsigned char* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wrow[i] = reinterpret_cast<const unsigned char*>(p.W) + ((lane ^ ((wave * 4 + i) & 15)) << 4);
    auto stage = [&](int tile, int slot) {
      tile = min(tile, t1 - 1);  // past the end: the last tile again, into a slot nobody will read
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = min(tile * VT_N + wave * 4 + i, p.N - 1);  // ragged last tile: clamp to the last W row
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[i] + (int64_t)n * 1024),
                                         (__attribute__((address_space(3))) void*)(smem + slot * VT_BYTES + (wave * 4 + i) * 1024),
                                         16, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < V_AHEAD; ++i) stage(t0 + i, i);
    __builtin_amdgcn_sched_barrier(0);

    // ---- activations of this wave's 32 rows, whole K, resident: fragment ks = X[row][16 ks + 8 h + (0..7)]
    bf16x8 a[32];
    const bf16_t* arow = p.A + (int64_t)min(row, p.M - 1) * p.lda + h * 8;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(arow + ks * 16);
    __builtin_amdgcn_sched_barrier(0);

    // per-lane W fragment offset: row r of the tile, chunk (2 ks + h) ^ (r & 15); the ks part is an
    // XOR of the low four chunk bits with (2 ks & 15) and an add of 256 per 8 k-steps
    const int bswz = r * 1024;
    auto boff = [&](int ks) { return bswz + ((((2 * ks + h) ^ (r & 15)) & 15) << 4) + ((2 * ks) >> 4) * 256; };

    float rm = -1e30f, rs = 0.f, rref = -1e30f;
    float rref2 = -1e30f * V_LOG2E;  // rref * log2(e): one fma + one exp2 per logit (MFMA and VALU time ADD on a SIMD, §4.1c)
    int ri = 0x7fffffff;
    f32x16 acc, accp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accp[i] = 0.f; }

    // column of accumulator register i of this lane inside a tile
    auto col_in_tile = [&](int i) { return (i & 3) + 8 * (i >> 2) + 4 * h; };
    // statistics of one finished tile held in accp: pieces woven between the MFMAs of the next tile
    float tmax = -INFINITY;
    auto stat_piece = [&](int i) {  // logit i of 16
      tmax = fmaxf(tmax, accp[i]);
      if (!COLLECT) rs += __builtin_amdgcn_exp2f(fmaf(accp[i], V_LOG2E, -rref2));
    };
    auto append_global = [&](int grow, float v, int c) {
      const int pos = atomicAdd(&p.cnt[grow], 1);
      if (pos < p.cap) {
        p.cval[(int64_t)grow * p.cap + pos] = v;
        p.cidx[(int64_t)grow * p.cap + pos] = c;
      }
    };
    auto stat_open = [&](int tile, bool first) {
      if (tile * VT_N + VT_N > p.N) {  // ragged last tile of the vocabulary: columns past N never win nor count
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (tile * VT_N + col_in_tile(i) >= p.N) accp[i] = -INFINITY;
      }
      if (first && !COLLECT) { rref = fmaxf(accp[0], -1e30f); rref2 = rref * V_LOG2E; }
      tmax = -INFINITY;
    };
    auto stat_close = [&](int tile) {
      if constexpr (COLLECT) {
        // ONE wave-uniform test per tile (about every second tile of a wave holds a candidate of one of its rows)
        if (__builtin_expect(__any(tmax >= thr), 0)) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float v = accp[i];
            const int c = tile * VT_N + col_in_tile(i);
            if (v >= thr && c < p.N) {
              int pos;
              const int one = 1;
              asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(lbase), "v"(one) : "memory");
              if (pos < V_LCAP) {
                const unsigned a = lbase + 8 + (unsigned)pos * 8;
                const int packed = ((wave * 32 + r) << 24) | c;
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(a), "v"(v), "v"(packed) : "memory");
              } else {
                append_global(row, v, c);  // staging area full
              }
            }
          }
        }
        return;
      }
      if (p.tile_max) {  // one 128-byte store per wave and tile (rows past M repeat row M - 1: the same value)
        const float tm = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        if (h == 0) p.tile_max[(int64_t)tile * p.M + min(row, p.M - 1)] = tm;
      }
      if (tmax > rm) {  // a new maximum of this row: rare after the first tiles -> divergent branch
        int c = 0;
#pragma unroll
        for (int i = 15; i >= 0; --i) c = accp[i] == tmax ? col_in_tile(i) : c;  // lowest column among equals
        ri = tile * VT_N + c;
        rm = tmax;
      }
      if (rm - rref > 20.0f) {  // keep the sum's reference within e^20 of the maximum
        rs *= __expf(rref - rm);
        rref = rm; rref2 = rref * V_LOG2E;
      }
    };

    // one tile: wait + barrier, refill the freed slot, 32 MFMAs with (STATS) the previous tile's statistics
    // woven in; the first tile of a range has no predecessor - a compile-time variant, not a flag tested
    // at every weaving point
    auto tile_body = [&](int t, auto with_stats) {
      constexpr bool STATS = decltype(with_stats)::value && !(CARE_V32_DBG & 16);
      const int it = t - t0;
      // tile t has landed; the two younger tiles (4 DMA instructions each) stay in flight.  Iteration 0
      // drains everything: the A fragments were issued behind the prologue DMAs.
      V32_STAMP(it, 0);
      if constexpr (!decltype(with_stats)::value) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (COLLECT || !p.tile_max) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else {  // + the tile_max stores issued since this tile's DMA went out (iterations it - 3 .. it - 1 that had statistics)
        switch (min(it - 1, 3)) {
          case 0: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory"); break;
        }
      }
      V32_STAMP(it, 1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      V32_STAMP(it, 2);
      stage(t + V_AHEAD, (it + V_AHEAD) % V_RING);  // the slot every wave finished reading last iteration
      __builtin_amdgcn_sched_barrier(0);

      const unsigned char* sb = smem + (it % V_RING) * VT_BYTES;
      // A wave cannot issue VALU work in the shadow of its OWN MFMAs (tools/micro/mfma_valu_overlap.hip: they add), so
      // the statistics of tile t - 1 (~180 VALU instructions) only overlap the matrix pipe if the OTHER wave of the SIMD
      // is multiplying meanwhile.  The per-tile barrier starts all eight waves together: left alone both waves of a SIMD
      // queue for the matrix pipe and then for the VALU.  So the two waves of a SIMD (w and w + 4) take the two halves
      // of a tile step in opposite order: waves 0-3 multiply first, waves 4-7 do their statistics first.
      const bool stats_first = (V32_STAGGER == 1 && wave >= 4) || (V32_STAGGER == 2 && (wave & 1)) || (V32_STAGGER == 3 && (wave & 2));
      auto stats_prev = [&]() {
        stat_open(t - 1, (t - 1) % tpb == 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) stat_piece(i);
        stat_close(t - 1);
      };
      if constexpr (STATS) {
        if (stats_first) {
          stats_prev();
          asm volatile("" : "+v"(rs), "+v"(rm), "+v"(rref), "+v"(rref2));  // done before the first MFMA, not sunk behind the chain
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 fb[BDEPTH];
#pragma unroll
      for (int ks = 0; ks < BDEPTH; ++ks) {
        if (CARE_V32_DBG & 32) { fb[ks] = bf16x8{}; asm volatile("" : "+v"(fb[ks])); }
        else fb[ks] = *reinterpret_cast<const bf16x8*>(sb + boff(ks));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        const bf16x8 b = fb[ks % BDEPTH];
        if (ks + BDEPTH < 32 && !(CARE_V32_DBG & 32)) fb[ks % BDEPTH] = *reinterpret_cast<const bf16x8*>(sb + boff(ks + BDEPTH));
        if (CARE_V32_DBG & 2) asm volatile("" :: "v"(b));
        else acc = care_mfma_32x32x16_h16(b, a[ks], acc, 0, 0, 0);  // D[n][m]: lane = row m, 16 columns n
        __builtin_amdgcn_sched_barrier(0);
      }
      V32_STAMP(it, 3);
      if constexpr (STATS) {
        if (!stats_first) stats_prev();
      }
      accp = acc;
    };
    // the statistics of a finished range: merge the two lanes of a row (columns 4 h + ...), write (max, argmax,
    // sum relative to the max), start over
    auto close_range = [&](int range) {
      float m = rm, sx = rs > 0.f ? rs * __expf(rref - rm) : 0.f;
      if (!(sx < 3.0e38f)) sx = 1.0f;  // the sum overflowed (a > 88 jump inside one tile): it is its largest term
      int id = ri;
      const float om = __shfl_xor(m, 32, 64), os = __shfl_xor(sx, 32, 64);
      const int oi = __shfl_xor(id, 32, 64);
      const float mn = fmaxf(m, om);
      sx = sx * __expf(m - mn) + os * __expf(om - mn);
      if (om > m || (om == m && oi < id)) id = oi;
      if (h == 0 && row < p.M) {
        const int64_t o = (int64_t)row * p.ns + range;
        p.pmax[o] = mn; p.pidx[o] = id; p.psum[o] = sx;
      }
      rm = -1e30f; rs = 0.f; rref = -1e30f; rref2 = -1e30f * V_LOG2E; ri = 0x7fffffff;
    };
    tile_body(t0, std::false_type{});
#pragma unroll 1
    for (int t = t0 + 1; t < t1; ++t) {
      tile_body(t, std::true_type{});                       // ... with the statistics of tile t - 1
      if (!COLLECT && t % tpb == 0) close_range(t / tpb - 1);  // which was the last one of its range
    }
    // the last tile's statistics
    if (!(CARE_V32_DBG & 16)) {
      stat_open(t1 - 1, (t1 - 1) % tpb == 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) stat_piece(i);
      stat_close(t1 - 1);
    }
    if constexpr (COLLECT) {  // flush the staged candidates: all 512 threads
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      typedef int i32x2 __attribute__((ext_vector_type(2)));
      int n;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(n) : "v"(lbase) : "memory");
      n = min(n, V_LCAP);
      for (int e = tid; e < n; e += 512) {
        i32x2 pr;
        const unsigned a = lbase + 8 + (unsigned)e * 8;
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pr) : "v"(a) : "memory");
        append_global(panel * V_ROWS + (int)((unsigned)pr[1] >> 24), __builtin_bit_cast(float, pr[0]), pr[1] & 0xffffff);
      }
    } else {
      close_range((t1 - 1) / tpb);
    }
  }
}

}  // namespace

// Row count from which the 256-row kernel is used (below it the 128-row panels of gemm_as.hip fill the chip better).
extern "C" int care_vocab32_applies(int M, int N, int K, int a_dtype, int has_labels) {
  static const int min_rows = [] { const char* e = getenv("CARE_V32_MIN_ROWS"); return e ? atoi(e) : 8192; }();
  return K == 512 && a_dtype == CARE_BF16 && !has_labels && M >= min_rows && N >= 4 * VT_N;
}

extern "C" int care_vocab32_launch_tiles(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                                         float* tile_max, int M, int N, int ns, void* stream);
extern "C" int care_vocab32_launch(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                                   int M, int N, int ns, void* stream) {
  return care_vocab32_launch_tiles(A, lda, W, pmax, pidx, psum, nullptr, M, N, ns, stream);
}
extern "C" int care_vocab32_launch_tiles(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                                         float* tile_max, int M, int N, int ns, void* stream) {
  VArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W);
  p.pmax = pmax; p.pidx = pidx; p.psum = psum; p.M = M; p.N = N; p.ns = ns; p.tile_max = tile_max;
  p.panels = (M + V_ROWS - 1) / V_ROWS;
  p.total_items = p.panels * ns;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&vocab_argmax32_kernel<6, V_ARGMAX>), V_LDS, lds_ok)) return e;
  const int blocks = p.total_items < 256 ? p.total_items : 256;  // one workgroup per CU (128 KiB of LDS), persistent
  hipLaunchKernelGGL((vocab_argmax32_kernel<6, V_ARGMAX>), dim3(blocks), dim3(512), V_LDS, (hipStream_t)stream, p);
  return care_launch_status();
}

// Second pass of the fused beam selection (care_gemm_collect_bf16) on the same 256-row panels.
extern "C" int care_collect32_launch(const void* A, int64_t lda, const void* W, const float* thr, int32_t* cnt, float* cval,
                                     int32_t* cidx, int cap, int M, int N, int ns, void* stream) {
  VArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W);
  p.thr = thr; p.cnt = cnt; p.cval = cval; p.cidx = cidx; p.cap = cap; p.M = M; p.N = N; p.ns = ns;
  p.panels = (M + V_ROWS - 1) / V_ROWS;
  p.total_items = p.panels * ns;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&vocab_argmax32_kernel<6, V_COLLECT>), V_LDS + V_STAGE, lds_ok))
    return e;
  const int blocks = p.total_items < 256 ? p.total_items : 256;
  hipLaunchKernelGGL((vocab_argmax32_kernel<6, V_COLLECT>), dim3(blocks), dim3(512), V_LDS + V_STAGE, (hipStream_t)stream, p);
  return care_launch_status();
}

// Column ranges for the 256-row panels: whole launch rounds over the 256 CUs, each work item paying ~3 tile times of
// prologue (A fragments, ring fill).  32768 rows -> 8 (4 rounds of 42 tiles), 20480 rows (4096 clips x beam 5) -> 16
// (5 rounds of 21 tiles instead of 3 of 42: 640 items over 256 CUs leave half the chip idle in the third round).
extern "C" int care_vocab32_ranges(int M, int N, int min_parts) {
  const int panels = (M + V_ROWS - 1) / V_ROWS, tiles = (N + VT_N - 1) / VT_N;
  int best = 0;
  long best_cost = 0;
  for (int ns = 8; ns <= 32; ns += 8) {
    if (ns < min_parts && ns < 32) continue;
    const long rounds = ((long)panels * ns + 255) / 256;
    const long cost = rounds * ((tiles + ns - 1) / ns + 3);
    if (!best || cost < best_cost) { best = ns; best_cost = cost; }
  }
  return best;
}

#if CARE_V32_DBG & 64
extern "C" int care_v32_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(v32_stamps), sizeof(v32_stamps)); }
#endif
