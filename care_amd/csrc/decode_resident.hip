// decode_resident.hip - the WHOLE greedy decode of a small batch as one resident launch.
//
// A decoder step of a small batch (1 .. a few hundred caption rows) is launch-bound: ~15 dependent kernels of
// 4-5 us each (rocprofv3, profiles/r03_small_batch_multilaunch_*), none of which has more than a few microseconds of
// work.  Here one grid of <= one workgroup per CU stays resident for all T steps and walks the phases of a step
// separated by producer-counted hand-offs (no fences, no full barriers - see GridSync): per decoder layer
//
//   QKV projection (+ token choice of the previous step + embedding) | self-attention | dense + residual |
//   { query projection | attention over the static keys | dense + residual } per cross / attribute block |
//   FFN dense1 + activation | FFN dense2 + residual | ... | vocabulary partial arg-max
//
// (reference: models/Decoder.py + models/components/Layers.py:157-228 driven step by step from
// models/Translator.py:77-143; Head.py:26-32; the early exit of Translator.py:77-81 is the device-side
// `all rows ended` counter, read by every workgroup once the phase that advances it is complete).
//
// Layout of a phase (DESIGN.md 4.2d has the measurements):
//   * GEMM phases: a workgroup owns one 16-row tile (its A rows go through LDS as bf16 once, then live in registers
//     as MFMA fragments) and walks column items - 16 columns with the 4 waves splitting K, or 64 columns with a wave
//     per 16 x 16 tile (gemm_phase).  The W fragments (v_mfma_f32_16x16x32_bf16 A operand, 16 B per lane straight from
//     the [N, K] row-major weight) of the first item are requested BEFORE the workgroup waits for the previous phase - weights
//     do not depend on the previous phase - and double-buffered across items.
//   * LayerNorm is applied ON LOAD: a phase stores the pre-LayerNorm sum (dense + bias + residual, fp32) and
//     every consumer normalises the 16 rows it needs (it reads all K columns anyway); the consumer of column
//     item 0 also stores the normalised fp32 rows, the residual of the phase after next.
//   * attention phases: one wave per (row, head), 8 key slots x 8 dim chunks per wave-wide 16-byte load (the
//     layout of csrc/attention.hip), scores / softmax in registers.
//   * the token choice (max / arg-max / sum-exp over the vocabulary partials) is folded into the first phase of
//     the next step; the rows' score / length / end flags are advanced there by an otherwise idle workgroup
//     (PhaseMap::helper) or the workgroup of column item 0; the last step's choice runs after the loop.
//   * column items follow blockIdx % 8 = the XCD a workgroup runs on, so every weight byte lives in ONE L2 (PhaseMap).
// Rounding points are those of the multi-launch bf16 path with projected cross K/V (bf16 A operands, K/V
// caches and contexts, fp32 accumulators, LayerNorm and softmax statistics); sums run in another order, the same
// order at every row count.
#include "decode_resident.h"

namespace {

// KCF = ff / 512; RB: row tiles per workgroup in the vocabulary phase; SM (few row tiles): QKV and FFN dense1 as
// 16-column K-split items too (the same bits either way: gemm_phase adds K in the same order in both forms)
// D = d_model.  D = 768 / 1024 (ff = 4 D; config/archs.yaml:15-26): the K-split forms in EVERY GEMM phase (a wave's K range
// is D / 4: 6 / 8 fragments), the vocabulary phase included, and FFN dense2 over two workgroups per column tile - up to 128
// rows (BASELINE configs[3]: 32 clips per GPU).
template <int KCF, int RB, bool SM, bool HF, int D = 512>  // HF (ff = 2048, <= 64 rows): FFN dense2 over two workgroups per column tile
__global__ __launch_bounds__(256, 1) void decode_resident_kernel(RArgs p_by_value) {  // (read through the kernarg segment: res_args)
  const ResKArgs kargs = RES_KARGS();
#define p (res_args(kargs))
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
  GridSync gs{p.sync, (unsigned)p.ghost, -1, false, 0, 0, 0u};
  gs.fenced = p.fenced != 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = p.d;
  constexpr bool WIDE = D != 512;
  const float* y2 = ((KCF == 4 && HF) || WIDE) ? p.y2 : nullptr;  // the second K half of FFN dense2, added by its consumers
  bool ended = false;
  // the phase a phase consumes: its position within the step, its producers, how many times it has run
  int sl = 0, sl_prev = 0;
  unsigned np_prev = 0, ex_prev = 0;
#define RES_PHASE(CALL)                     \
  do {                                      \
    gs.prev = sl_prev;                      \
    gs.want = np_prev * ex_prev;            \
    gs.cur = sl;                            \
    const unsigned np_this = (CALL);        \
    sl_prev = sl;                           \
    np_prev = np_this;                      \
    ex_prev = (unsigned)t;                  \
    ++sl;                                   \
  } while (0)
  for (int t = 1; t <= p.steps && !ended && !gs.dead; ++t) {
    gs.slot = (p.prof_step == t && blockIdx.x == 0) ? 0 : -1;
    sl = 0;
    for (int l = 0; l < p.n_layers; ++l) {
      const RLayer& L = p.L[l];
      if (l == 0) RES_PHASE((gemm_phase<D, A_EMBED, E_QKV, SM, 1, D>(p, gs, true, sA, L.qkv_w, L.qkv_b, 3 * d, nullptr, p.emb_g, p.emb_be, true, t, L.skv)));
      else RES_PHASE((gemm_phase<D, A_LN, E_QKV, SM, 1, D>(p, gs, true, sA, L.qkv_w, L.qkv_b, 3 * d, p.y, p.L[l - 1].fg, p.L[l - 1].fbe, true, t, L.skv, y2)));
      if (l == 0 && t > 1) {
        // Every row ended with the token chosen in the phase above?  EVERY workgroup must come to the same answer before
        // it goes on.  The workgroups with self-attention items wait for the phase anyway and read the count themselves;
        // workgroup 0 (always one of them) publishes the verdict of step t as 2 t + ended in a word of its own line,
        // which is all the others poll - 250 of them polling the 8 counter shards held up the arrivals at one row.
        const int bpx0 = (gridDim.x & 7) == 0 ? (int)(gridDim.x >> 3) : (int)gridDim.x;
        const int x0 = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) : 0, xs0 = (gridDim.x & 7) == 0 ? 8 : 1;
        const bool attn_part = ((gridDim.x & 7) == 0 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) < ((p.R - x0 + xs0 - 1) / xs0) * p.H &&
                               bpx0 > 0;
        bool all_ended;
        if (attn_part) {
          gs.prev = sl_prev;
          gs.want = np_prev * ex_prev;
          gs.wait();
          all_ended = p.early && __hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)p.R;
          if (blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(p.sync + 64, 2u * (unsigned)t + (all_ended ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __shared__ unsigned s_verdict;
          if (threadIdx.x == 0) {
            unsigned v, spins = 0;
            while ((v = __hip_atomic_load(p.sync + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 2u * (unsigned)t) {
              __builtin_amdgcn_s_sleep(2);
              if ((++spins & 63u) == 0 && __hip_atomic_load(p.sync + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v = ~0u; break; }
            }
            s_verdict = v;
          }
          __syncthreads();
          const unsigned v = s_verdict;
          if (v == ~0u) gs.dead = true;
          all_ended = (v & 1u) != 0 && !gs.dead;
          __syncthreads();
        }
        if (gs.dead) break;
        if (all_ended) {
          ended = true;
          if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = (unsigned)(t - 1);
          break;
        }
        RES_PHASE((p.T <= 32 ? attn_phase<true, 4, false, D>(p, gs, false, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0)
                              : attn_phase<true, RES_MAXKB, false, D>(p, gs, false, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0)));
      } else {
        RES_PHASE((p.T <= 32 ? attn_phase<true, 4, false, D>(p, gs, true, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0)
                              : attn_phase<true, RES_MAXKB, false, D>(p, gs, true, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0)));
      }
      RES_PHASE((gemm_phase<D, A_BF16, E_RES, true, 1, D>(p, gs, true, sA, L.o_w, L.o_b, d, p.ctx, nullptr, nullptr, false, t, nullptr)));
      const float* g = L.g;
      const float* be = L.be;
      for (int a = 0; a < L.n_att; ++a) {
        const RAttn& A = L.att[a];
        RES_PHASE((gemm_phase<D, A_LN, E_Q, true, 1, D>(p, gs, true, sA, A.q_w, A.q_b, d, p.y, g, be, true, t, nullptr)));
        RES_PHASE((A.nkeys <= 64 ? attn_phase<false, 8, false, D>(p, gs, true, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, nullptr, A.bias, A.bias_ld)
                                  : attn_phase<false, RES_MAXKB, false, D>(p, gs, true, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, nullptr, A.bias, A.bias_ld)));
        RES_PHASE((gemm_phase<D, A_BF16, E_RES, true, 1, D>(p, gs, true, sA, A.o_w, A.o_b, d, p.ctx, nullptr, nullptr, false, t, nullptr)));
        g = A.g; be = A.be;
      }
      RES_PHASE((gemm_phase<D, A_LN, E_ACT, SM, 1, D>(p, gs, true, sA, L.w1, L.b1, p.ff, p.y, g, be, true, t, nullptr)));
      if constexpr (WIDE) RES_PHASE((ffn2_phase<true, 512 * KCF, D>(p, gs, sA, L.w2, L.b2)));
      else if constexpr (KCF == 4) RES_PHASE((ffn2_phase<HF>(p, gs, sA, L.w2, L.b2)));
      else RES_PHASE((gemm_phase<512 * KCF, A_BF16, E_RES, true>(p, gs, true, sA, L.w2, L.b2, d, p.h, nullptr, nullptr, false, t, nullptr)));
    }
    if (ended) break;
    const RLayer& LL = p.L[p.n_layers - 1];
    RES_PHASE((gemm_phase<D, A_LN, E_VOCAB, WIDE, RB, D>(p, gs, true, sA, p.vocab, nullptr, p.V, p.y, LL.fg, LL.fbe, false, t, nullptr, y2)));
  }
#undef RES_PHASE
  if (!ended && !gs.dead) {  // every workgroup: the vocabulary partials of the last step are complete
    gs.prev = sl_prev;
    gs.want = np_prev * ex_prev;
    gs.wait();
  }
  if (gs.dead) {  // aborted (GridSync::wait): say so where the host looks anyway - EVERY row's length, so that no
    // caller mistakes the previous batch's rows in a re-used workspace for results
    // (EVERY workgroup that gave up, each for all rows - a few hundred words: striped over the grid, up to 256 rows were workgroup
    // 0's alone to mark)
    for (int r = threadIdx.x; r < p.R; r += 256) cst_i(p.length + r, -1);
    if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = 0xffffffffu;
    return;
  }
  if (!ended) {  // the token of the last step
    for (int rb = (blockIdx.x * 4 + wave) * 4; rb < p.R; rb += gridDim.x * 16) {
      int tok[4];
      select4<true>(p, rb, p.steps, lane, tok);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = (unsigned)p.steps;
  }
}

#undef p

std::atomic<unsigned long long> g_res_lds_done[8];
std::atomic<int> g_res_ok[8];  // residency checked (res_check_residency) for this instantiation

}  // namespace

std::atomic<int> care_res_dbg_prof{0}, care_res_dbg_ghost{0};  // care_decode_resident_debug (shared with decode_resident_beam.hip)
std::atomic<int> care_res_fenced_mode{-1};                      // care_resident_set_fenced

extern "C" {

void care_resident_set_fenced(int mode) { care_res_fenced_mode.store(mode < 0 ? -1 : (mode ? 1 : 0)); }
int care_resident_fenced(void) { return res_fenced_for_device(); }

void care_decode_resident_debug(int prof_step, int ghost) {
  care_res_dbg_prof.store(prof_step);
  care_res_dbg_ghost.store(ghost);
}

int64_t care_decode_resident_scratch(int rows, int d, int ff, int V) {
  if (rows < 1 || d < 1 || ff < 1 || V < 1) return CARE_EINVAL;
  const int64_t R16 = (rows + 15) / 16 * 16;
  int64_t parts = d == 512 ? (V + 63) / 64 : (V + 15) / 16;  // column items of the vocabulary phase (16 columns each when d_model > 512)
  if (parts > 64 * RES_NP) parts = 64 * RES_NP;
  // sync | xres, y, y2, q fp32 [R16, d] | ctx bf16 [R16, d] | h bf16 [R16, ff] | pmax, pidx, psum [R16, parts]
  return RES_SYNC_BYTES + R16 * d * 4 * 4 + R16 * d * 2 + R16 * ff * 2 + R16 * parts * 12;
}

int care_decode_resident(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                         const float* sem, int sem_div, const float* emb_g, const float* emb_b, float eps,
                         const void* vocab_w, int V, int d, int heads, int ff, int act, int rows, int T, int steps,
                         int bos, int eos, int pad, int32_t* fed, int fed_stride, float* score, int32_t* length,
                         int32_t* finished, void* scratch, int64_t scratch_bytes, int early_exit, int blocks, void* stream) {
  if (!layers || !word || !pos || !emb_g || !emb_b || !vocab_w || !fed || !score || !length || !finished || !scratch)
    return CARE_EINVAL;
  if (n_layers < 1 || n_layers > RES_MAX_LAYERS || rows < 1 || T < 1 || steps < 1 || steps > T || V < 1 || fed_stride < T + 1)
    return CARE_EINVAL;
  const bool wide = d != 512;  // d_model 768 / 1024 with ff = 4 d_model, up to 128 rows (decode_resident_kernel's D)
  if (heads * 64 != d || T > 8 * RES_MAXKB || V > 64 * 64 * RES_NP) return CARE_ESHAPE;
  if (!wide && ff != 512 && ff != 1024 && ff != 2048) return CARE_ESHAPE;
  if (wide && ((d != 768 && d != 1024) || ff != 4 * d || rows > 128)) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if (scratch_bytes < care_decode_resident_scratch(rows, d, ff, V) || !care_aligned16(scratch)) return CARE_EINVAL;
  RArgs p{};
  if (const int rc = res_fill_layers(p, layers, n_layers)) return rc;
  p.word = word; p.pos = pos; p.sem = sem; p.sem_div = sem_div > 0 ? sem_div : 1; p.emb_g = emb_g; p.emb_be = emb_b; p.eps = eps;
  p.vocab = (const bf16_t*)vocab_w; p.V = V;
  p.d = d; p.H = heads; p.ff = ff; p.act = act; p.R = rows; p.T = T; p.steps = steps; p.bos = bos; p.eos = eos; p.pad = pad; p.early = early_exit;
  p.prof_step = care_res_dbg_prof.load();        // tools: phase clocks of that step -> scratch + 2048
  p.ghost = care_res_dbg_ghost.load() ? 8 : 0;
  p.fenced = res_fenced_for_device();   // tests: phases whose producers never all arrive (the watchdog)
  p.fed = fed; p.fed_stride = fed_stride; p.score = score; p.length = length; p.fin = finished;
  const int64_t R16 = (rows + 15) / 16 * 16;
  p.parts = wide ? ((V + 15) / 16 < 64 * RES_NP ? (V + 15) / 16 : 64 * RES_NP) : (V + 63) / 64;  // (the layout's stride; the launch's count below)
  unsigned char* b = (unsigned char*)scratch;
  p.sync = (unsigned*)b; b += RES_SYNC_BYTES;
  p.xres = (float*)b; b += R16 * d * 4;
  p.y = (float*)b; b += R16 * d * 4;
  p.y2 = (float*)b; b += R16 * d * 4;
  p.q = (float*)b; b += R16 * d * 4;
  p.ctx = (bf16_t*)b; b += R16 * d * 2;
  p.h = (bf16_t*)b; b += R16 * ff * 2;
  p.pmax = (float*)b; b += R16 * p.parts * 4;
  p.pidx = (int32_t*)b; b += R16 * p.parts * 4;
  p.psum = (float*)b;

  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return (int)e;
  const ResKnobs& kn = res_knobs();
  // every workgroup must be resident (they wait for one another): at most one per CU; no more than the widest phase
  // has items (the vocabulary groups x row tiles, or a wave per (row, head))
  const int RT = (int)(R16 / 16), CIV = wide ? (V + 15) / 16 : (V + 63) / 64;  // column items of the vocabulary phase
  // row tiles a workgroup multiplies per fetch of its vocabulary fragments: 2 from 128 rows (ff = 2048 builds) - the 8
  // workgroups of an XCD that share a column item ask its L2 for the same 64 KB at the same time, and two row tiles per
  // fetch halve those requests while doubling the rows a workgroup normalises (*measured*, same box, us per step:
  // 128 rows 77.6 -> 76.5, 96 rows 73.1 -> 73.0; ms per pass with 1 / 2 / 4 row tiles: 192 rows 3.20 / 3.04 / 3.24,
  // 256 rows 3.38 / 3.25 / 3.44)
  int rb = (ff == 2048 && rows >= 128 && !wide) ? 2 : 1;
  if (kn.rb >= 0 && ff == 2048) rb = kn.rb >= 2 ? 2 : 1;  // tuning (CARE_RESIDENT_RB)
  const int RG = (RT + rb - 1) / rb;
  int want = RT * CIV;
  if ((rows * heads + 3) / 4 > want) want = (rows * heads + 3) / 4;
  int grid = blocks > 0 ? blocks : want;
  grid = (grid + 7) / 8 * 8;  // whole rounds over the 8 XCDs (PhaseMap)
  if (grid > cus) grid = cus;
  if (grid < RT) return CARE_ESHAPE;  // a workgroup per 16-row tile at least
  {  // vocabulary partials per row = workgroups per row tile that have a column item (PhaseMap)
    const int nper = ((grid & 7) == 0 && (grid >> 3) >= RG) ? 8 * ((grid >> 3) / RG) : grid / RG;
    p.parts = nper < CIV ? nper : CIV;
    if (p.parts > 64 * RES_NP) return CARE_ESHAPE;
  }
  const int kmax = wide ? (ff / 2 > d ? ff / 2 : d) : (ff > d ? ff : d);  // (wide: FFN dense2's tile holds a K half)
  int lds = 16 * (kmax + 8) * 2;
  if (rb * 16 * (512 + 8) * 2 > lds) lds = rb * 16 * (512 + 8) * 2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(256);
  int rc;
  // (the residency check comes before anything is enqueued: a refused launch leaves the stream untouched)
#define RES_LAUNCH(KCF, RB, SM, HF, SLOT) RES_LAUNCH_D(KCF, RB, SM, HF, 512, SLOT)
#define RES_LAUNCH_D(KCF, RB, SM, HF, DM, SLOT)                                                                         \
  do {                                                                                                                  \
    const void* kfn = (const void*)decode_resident_kernel<KCF, RB, SM, HF, DM>;                                         \
    if ((rc = care_allow_dynamic_lds(kfn, lds, g_res_lds_done[SLOT]))) return rc;                                        \
    if (!g_res_ok[SLOT].load(std::memory_order_acquire)) {                                                              \
      if ((rc = res_check_residency(kfn, lds, grid, cus))) return rc;                                                   \
      g_res_ok[SLOT].store(1, std::memory_order_release);                                                               \
    }                                                                                                                   \
    if ((e = res_zero_words(p.sync, RES_SYNC_BYTES, st)) != hipSuccess) return (int)e;                               \
    hipLaunchKernelGGL((decode_resident_kernel<KCF, RB, SM, HF, DM>), g, blk, lds, st, p);                              \
  } while (0)
  // QKV / FFN dense1 in 16-column K-split items up to 64 rows (*measured* us / step with / without: 1 row 43.6 / 47.1,
  // 32 rows 60.7 / 65.0, 64 rows 66.0 / 67.7, 128 rows 76.9 / 77.5 with 8.1 against 4.6 us in FFN dense1); FFN dense2
  // over two workgroups per column tile up to 64 rows (ffn2_phase; *measured* ms per pass with / without: 1 row 1.28 / 1.33,
  // 16 rows 1.59 / 1.68, 32 rows 1.69 / 1.86, 64 rows 1.94 / 2.00)
  const bool small = kn.small >= 0 ? kn.small != 0 : rows <= 64;       // tuning (CARE_RESIDENT_SMALL)
  const bool half = rows <= (kn.half_rows >= 0 ? kn.half_rows : 64);   // tuning (CARE_RESIDENT_HALF_ROWS)
  if (d == 768) RES_LAUNCH_D(6, 1, true, true, 768, 6);
  else if (d == 1024) RES_LAUNCH_D(8, 1, true, true, 1024, 7);
  else if (ff == 512) RES_LAUNCH(1, 1, true, false, 0);
  else if (ff == 1024) RES_LAUNCH(2, 1, true, false, 1);
  else if (rb == 2) RES_LAUNCH(4, 2, false, false, 3);
  else if (small && half) RES_LAUNCH(4, 1, true, true, 5);
  else if (small) RES_LAUNCH(4, 1, true, false, 4);
  else RES_LAUNCH(4, 1, false, false, 2);
#undef RES_LAUNCH
#undef RES_LAUNCH_D
  return care_launch_status();
}

}  // extern "C"
