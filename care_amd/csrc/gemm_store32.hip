// gemm_store32.hip - C = act(A W^T + bias) for bf16 A [M, 512], bf16 W [N, 512] at large row counts:
// the QKV, Wq and FFN1 projections of a decode step (Attention.py:53-67, SubLayers.py:129-141).
//
// Same structure as csrc/gemm_vocab.hip (see there for the ablation that motivates it: with 128-row
// panels the L2 -> LDS stream of W bounds the A-stationary kernel before any arithmetic):
//   * a workgroup is 8 waves x 32 rows = a 256-row panel with its A fragments resident in registers;
//   * W tiles of 32 output columns x 512 k through a 4-slot LDS ring (LDS-DMA, three in flight), one
//     barrier per tile; the bias slice of the block's column range is staged in LDS once;
//   * v_mfma_f32_32x32x16_bf16 with swapped operands; the W rows of a tile are permuted over the MFMA
//     rows when the tile is STAGED (LDS row n takes column 16 ((n >> 2) & 1) + 4 (n >> 3) + (n & 3)), so
//     a lane ends up with 16 CONSECUTIVE output columns of one row: two 16-byte bf16 stores (or four
//     fp32 ones) per tile, 64 (128) contiguous bytes per row from the two lanes of a row - the
//     16x16 form stores 8 bytes per lane, 32-byte row pieces;
//   * the epilogue of tile t (bias, activation, convert, store) is woven between the MFMAs of tile
//     t + 1; its stores are counted in the vmcnt arithmetic of the ring.
// Two destinations like care_gemm_bf16 (columns < n_split -> C0, the rest -> C1; each fp32 or bf16,
// own leading dimension): the QKV projection writes q and, straight into the cache, k | v.
// In-kernel stamps (round 5, tools/s32_ts.py, 32768 x 2048 x 512): ~4200 cycles per tile for the 2048 of a SIMD's two
// waves' MFMAs - both waves multiply at once, each with ~100 vector instructions of epilogue woven in, and a phase takes
// 2650 - 3350 (the second-dispatched half of the workgroup is the slower one); ~330 of landing wait and up to ~900 at the
// barrier follow.  The activation as a template parameter (a run-time test was a v_max + v_cndmask pair per output) is
// worth ~1 %: the vector work is not what the phase waits for.
#include <cstdlib>
#include <type_traits>

#include "care_common.h"

namespace {

constexpr int ST_N = 32;
constexpr int ST_BYTES = ST_N * 1024;
constexpr int S_RING = 4, S_AHEAD = 3;
constexpr int S_BIAS_OFF = S_RING * ST_BYTES;
constexpr int S_BIAS_MAX = 4096;         // bias columns a block can stage
constexpr int S_LDS = S_BIAS_OFF + S_BIAS_MAX * 4;
constexpr int S_ROWS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CARE_S32_ST_NT
#define CARE_S32_ST_NT 0  // non-temporal output stores (ablation)
#endif
#ifndef CARE_S32_QUAD
#define CARE_S32_QUAD 1  // fp32 tiles: transpose the pieces inside lane quads before storing (0: one row per lane)
#endif
#ifndef CARE_S32_DBG
#define CARE_S32_DBG 0  // ablation builds: 1 no stores, 2 no bias reads, 4 no activation / conversion either, 8 lane-linear stores, 64 stamps
#endif

struct SArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W;
  const float* bias;
  void* C0; int64_t ldc0; int c0_bf16;
  void* C1; int64_t ldc1; int c1_bf16;
  int n_split, M, N, act, ns, panels, total_items;
};

__device__ __forceinline__ float s_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// GELU is a template parameter: erff inlined behind a run-time test in every epilogue piece made the kernel 6872
// instructions (55 KB) against 2658 without it - the relu / plain launches of the headline pass carried it along.
#if CARE_S32_DBG & 64  // tools (tools/variant_lib.py, tools/s32_ts.py): workgroup 0 stamps s_memtime per wave and tile
__device__ unsigned long long s32_stamps[8 * 64 * 4];
#define S32_STAMP(it, k) do { if (blockIdx.x == 0 && (it) < 64 && lane == 0) s32_stamps[(wave * 64 + (it)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S32_STAMP(it, k) do { } while (0)
#endif

template <int BDEPTH, int ACT>  // ACT: CARE_ACT_NONE / RELU / GELU (a run-time test per value was a v_max + v_cndmask pair on every output)
__global__ __launch_bounds__(512, 2) void gemm_store32_kernel(SArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tiles_total = p.N / ST_N;  // the launcher guarantees N % 32 == 0 and n_split % 32 == 0
  const int tpb = (tiles_total + p.ns - 1) / p.ns;

  for (int item = blockIdx.x; item < p.total_items; item += gridDim.x) {
    const int range = item % p.ns, panel = item / p.ns;
    const int t0 = range * tpb, t1 = min(t0 + tpb, tiles_total);
    if (t0 >= t1) continue;
    const int m0 = panel * S_ROWS + wave * 32;
    const int row = m0 + r;
    // every wave is done with the ring and the bias slice of the previous item
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- W tile -> ring slot: wave w copies LDS rows 4w .. 4w + 3; LDS row n holds output column
    // perm(n) of the tile, lane = chunk slot, source chunk = slot ^ (n & 15)
    unsigned wlane[4];
    int wcol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = wave * 4 + i;
      wlane[i] = (unsigned)((lane ^ (n & 15)) << 4);
      wcol[i] = 16 * ((n >> 2) & 1) + 4 * (n >> 3) + (n & 3);
    }
    auto stage = [&](int tile, int slot) {
      tile = min(tile, t1 - 1);  // past the end: the last tile again, into a slot nobody will read
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(p.W) +
                                                            (int64_t)(tile * ST_N + wcol[i]) * 1024 + wlane[i]),
            (__attribute__((address_space(3))) void*)(smem + slot * ST_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < S_AHEAD; ++i) stage(t0 + i, i);
    __builtin_amdgcn_sched_barrier(0);

    // ---- activations of this wave's 32 rows, whole K, resident
    bf16x8 a[32];
    const bf16_t* arow = p.A + (int64_t)min(row, p.M - 1) * p.lda + h * 8;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(arow + ks * 16);
    __builtin_amdgcn_sched_barrier(0);

    // ---- bias slice of the range -> LDS (behind the DMAs and the A loads: one combined latency)
    const int col0 = t0 * ST_N;
    if (p.bias) {
      float* sb = reinterpret_cast<float*>(smem + S_BIAS_OFF);
      const int nb = (t1 - t0) * ST_N;
      for (int i = tid; i < nb; i += 512) sb[i] = p.bias[col0 + i];
    }

    // The destination is a property of the TILE (wave-uniform): a range may straddle n_split, a 32-column
    // tile cannot (n_split % 32 == 0).  VM store instructions per tile: 2 (bf16) or 4 (fp32), none for a
    // wave whose rows are all past M - the last three iterations' counts enter the vmcnt arithmetic.
    const bool wave_rows = m0 < p.M;
    int st_hist0 = 0, st_hist1 = 0, st_hist2 = 0;  // stores issued in the iterations it - 1, it - 2, it - 3

    const int bswz = r * 1024;
    auto boff = [&](int ks) { return bswz + ((((2 * ks + h) ^ (r & 15)) & 15) << 4) + ((2 * ks) >> 4) * 256; };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    f32x16 accA, accB;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA[i] = 0.f; accB[i] = 0.f; }

    // epilogue piece g (0..3) of a finished tile: columns 16 h + 4 g .. + 3 of this lane's row
    // (accumulator registers 4 g .. 4 g + 3 hold output columns 16 h + 4 g + (0..3) after the staging permutation)
    auto out_piece = [&](const f32x16& ap, int tile, int g, float (&v)[16]) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && !(CARE_S32_DBG & 2)) {  // asm read: hipcc would drain the in-flight DMAs in front of a C++ LDS read here
        const unsigned addr = lds0 + (unsigned)(S_BIAS_OFF + ((tile * ST_N - col0) + 16 * h + 4 * g) * 4);
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(addr) : "memory");
      }
      v[4 * g + 0] = ap[4 * g + 0] + bv.x; v[4 * g + 1] = ap[4 * g + 1] + bv.y;
      v[4 * g + 2] = ap[4 * g + 2] + bv.z; v[4 * g + 3] = ap[4 * g + 3] + bv.w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ACT == CARE_ACT_GELU) v[4 * g + j] = s_gelu(v[4 * g + j]);
        else if constexpr (ACT == CARE_ACT_RELU) v[4 * g + j] = fmaxf(v[4 * g + j], 0.0f);
      }
    };
    auto out_store = [&](int tile, const float (&v)[16]) -> int {
      const bool second = tile * ST_N >= p.n_split;
      unsigned char* C = reinterpret_cast<unsigned char*>(second ? p.C1 : p.C0);
      const int64_t ld = second ? p.ldc1 : p.ldc0;
      const bool isb = (second ? p.c1_bf16 : p.c0_bf16) != 0;
      const int cshift = second ? p.n_split : 0;
      if (!isb && CARE_S32_QUAD && !(CARE_S32_DBG & 9)) {  // wave-uniform: every lane takes part in the exchange
        // fp32 tile: a lane holds 64 contiguous bytes of ITS row, so a store instruction is 64 separate 16-byte
        // pieces.  A 4 x 4 transpose of the 16-byte pieces inside every quad of lanes (rows 4q .. 4q + 3, DPP
        // quad_perm, two butterfly steps per component) makes lane j hold piece j of each of the quad's four
        // rows: store i then writes 64 contiguous bytes per quad.  *Measured* QKV (a third of its columns: fp32 q)
        // 87.5 -> 69 us at 32768 rows; the same idea for the bf16 tiles (lane pairs swapping one piece) changed nothing.
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = v[i];
        const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int g = 0; g < 4; g += 2) {  // step 1: pieces (g, g + 1) against lane ^ 1
            const float send = b0 ? t[4 * g + c] : t[4 * (g + 1) + c];
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
            if (b0) t[4 * g + c] = recv; else t[4 * (g + 1) + c] = recv;
          }
#pragma unroll
          for (int g = 0; g < 2; ++g) {     // step 2: pieces (g, g + 2) against lane ^ 2
            const float send = b1 ? t[4 * g + c] : t[4 * (g + 2) + c];
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
            if (b1) t[4 * g + c] = recv; else t[4 * (g + 2) + c] = recv;
          }
        }
        // now t[4 i .. 4 i + 3] = piece (lane & 3) of row 4 (r >> 2) + i
        const int rq = m0 + (r & ~3);
        float* base = reinterpret_cast<float*>(C) + (tile * ST_N - cshift) + 16 * h + 4 * (lane & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rq + i < p.M)
            *reinterpret_cast<f32x4*>(base + (int64_t)(rq + i) * ld) = f32x4{t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
      } else if (row < p.M && !(CARE_S32_DBG & 1)) {
        int64_t o = (int64_t)row * ld + (tile * ST_N - cshift) + 16 * h;
        if (CARE_S32_DBG & 8) {  // ablation: the same bytes to lane-linear (wrong) addresses - fully coalesced instructions
          const int64_t ob = (int64_t)min(m0, p.M - 40) * ld + (tile * ST_N - cshift);
          if (isb) {
            bf16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (bf16_t)v[j]; o1[j] = (bf16_t)v[8 + j]; }
            bf16_t* dst = reinterpret_cast<bf16_t*>(C) + ob + lane * 8;
            *reinterpret_cast<bf16x8*>(dst) = o0;
            *reinterpret_cast<bf16x8*>(dst + 512) = o1;
          } else {
            float* dst = reinterpret_cast<float*>(C) + ob + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dst + 256 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          }
        } else if (isb) {
          bf16x8 o0, o1;
#pragma unroll
          for (int j = 0; j < 8; ++j) { o0[j] = (bf16_t)v[j]; o1[j] = (bf16_t)v[8 + j]; }
          bf16_t* dst = reinterpret_cast<bf16_t*>(C) + o;
          if (CARE_S32_ST_NT) {
            __builtin_nontemporal_store(o0, reinterpret_cast<bf16x8*>(dst));
            __builtin_nontemporal_store(o1, reinterpret_cast<bf16x8*>(dst + 8));
          } else {
            *reinterpret_cast<bf16x8*>(dst) = o0;
            *reinterpret_cast<bf16x8*>(dst + 8) = o1;
          }
        } else {
          float* dst = reinterpret_cast<float*>(C) + o;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 o = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            if (CARE_S32_ST_NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst + 4 * g));
            else *reinterpret_cast<f32x4*>(dst + 4 * g) = o;
          }
        }
      }
      if (CARE_S32_DBG & 1) { asm volatile("" :: "v"(v[0]), "v"(v[5]), "v"(v[10]), "v"(v[15])); return 0; }
      // store instructions surely issued (a LOWER bound keeps the vmcnt arithmetic safe): the quad form's store i has an
      // active lane only if row m0 + i exists
      if (!isb && CARE_S32_QUAD && !(CARE_S32_DBG & 9)) return wave_rows ? min(4, p.M - m0) : 0;
      return wave_rows ? (isb ? 2 : 4) : 0;
    };

    auto wait_tile = [&]() {
      // the tile has landed when at most the DMAs of the two younger tiles (8) and the stores issued since
      // its own DMA are outstanding: its DMA went out three iterations ago (or in the prologue) AHEAD of
      // that iteration's stores
      switch (8 + st_hist0 + st_hist1 + st_hist2) {
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;  // odd counts (ragged last panel): wait for more
      }
    };

    auto tile_body = [&](int t, auto with_prev, f32x16& ac, f32x16& ap) {
      constexpr bool PREV = decltype(with_prev)::value;
      const int it = t - t0;
      S32_STAMP(it, 0);
      if constexpr (!PREV) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // also the A fragments and the bias slice
      else wait_tile();
      S32_STAMP(it, 1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      S32_STAMP(it, 2);
      stage(t + S_AHEAD, (it + S_AHEAD) % S_RING);
      __builtin_amdgcn_sched_barrier(0);

      const unsigned char* sb = smem + (it % S_RING) * ST_BYTES;
      bf16x8 fb[BDEPTH];
#pragma unroll
      for (int ks = 0; ks < BDEPTH; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(sb + boff(ks));
      float v[16];
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        const bf16x8 b = fb[ks % BDEPTH];
        if (ks + BDEPTH < 32) fb[ks % BDEPTH] = *reinterpret_cast<const bf16x8*>(sb + boff(ks + BDEPTH));
        if (ks == 0) ac = care_mfma_32x32x16_h16(b, a[0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else ac = care_mfma_32x32x16_h16(b, a[ks], ac, 0, 0, 0);
        if constexpr (PREV) {
          if (ks % 6 == 5 && ks / 6 < 4) out_piece(ap, t - 1, ks / 6, v);   // pieces after k-steps 5, 11, 17, 23
          if (ks == 27) { st_hist2 = st_hist1; st_hist1 = st_hist0; st_hist0 = out_store(t - 1, v); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      S32_STAMP(it, 3);
      if constexpr (!PREV) { st_hist2 = st_hist1; st_hist1 = st_hist0; st_hist0 = 0; }
    };
    tile_body(t0, std::false_type{}, accA, accB);
    int t = t0 + 1;
#pragma unroll 1
    for (; t + 1 < t1; t += 2) {
      tile_body(t, std::true_type{}, accB, accA);
      tile_body(t + 1, std::true_type{}, accA, accB);
    }
    auto last_out = [&](const f32x16& al) {
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) out_piece(al, t1 - 1, g, v);
      out_store(t1 - 1, v);
    };
    if (t < t1) {
      tile_body(t, std::true_type{}, accB, accA);
      last_out(accB);
    } else {
      last_out(accA);
    }
  }
}

}  // namespace

extern "C" int care_store32_applies(int M, int N, int K, int a_dtype, int n_split, int has_bias_cols) {
  static const int min_rows = [] { const char* e = getenv("CARE_S32_MIN_ROWS"); return e ? atoi(e) : 8192; }();
  return K == 512 && a_dtype == CARE_BF16 && M >= min_rows && N % ST_N == 0 && n_split % ST_N == 0 && N >= 4 * ST_N;
}

extern "C" int care_store32_launch(const void* A, int64_t lda, const void* W, const float* bias, void* C0, int64_t ldc0,
                                   int c0_bf16, void* C1, int64_t ldc1, int c1_bf16, int n_split, int M, int N, int act,
                                   void* stream) {
  SArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias;
  p.C0 = C0; p.ldc0 = ldc0; p.c0_bf16 = c0_bf16; p.C1 = C1; p.ldc1 = ldc1; p.c1_bf16 = c1_bf16;
  p.n_split = n_split; p.M = M; p.N = N; p.act = act;
  p.panels = (M + S_ROWS - 1) / S_ROWS;
  // column ranges per panel: enough work items for one workgroup per CU (256), whole tiles per range; the
  // bias slice of a range must fit its LDS staging area
  const int tiles = N / ST_N;
  int ns = 1;
  while ((p.panels * ns < 256 && ns * 2 <= tiles) || ((tiles + ns - 1) / ns) * ST_N > S_BIAS_MAX) ns *= 2;
  if (((tiles + ns - 1) / ns) * ST_N > S_BIAS_MAX) return CARE_ESHAPE;
  p.ns = ns;
  p.total_items = p.panels * ns;
  // 16-byte stores: destinations and leading dimensions must allow them
  const int e0 = c0_bf16 ? 8 : 4, e1 = c1_bf16 ? 8 : 4;
  if ((ldc0 % e0) || !care_aligned16(C0) || (n_split < N && ((ldc1 % e1) || !care_aligned16(C1)))) return CARE_EALIGN;
  const int blocks = p.total_items < 256 ? p.total_items : 256;
  auto go = [&](auto actc) -> int {
    constexpr int A = decltype(actc)::value;
    static std::atomic<unsigned long long> lds_ok{0};
    if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_store32_kernel<4, A>), S_LDS, lds_ok)) return e;
    hipLaunchKernelGGL((gemm_store32_kernel<4, A>), dim3(blocks), dim3(512), S_LDS, (hipStream_t)stream, p);
    return 0;
  };
  if (const int e = act == CARE_ACT_GELU ? go(std::integral_constant<int, CARE_ACT_GELU>{})
                  : act == CARE_ACT_RELU ? go(std::integral_constant<int, CARE_ACT_RELU>{})
                                         : go(std::integral_constant<int, CARE_ACT_NONE>{}))
    return e;
  return care_launch_status();
}

#if CARE_S32_DBG & 64
extern "C" int care_s32_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s32_stamps), sizeof(s32_stamps)); }
#endif
