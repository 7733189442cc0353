// decode_resident_beam.hip - BEAM SEARCH over a small batch as one resident launch.
//
// The reference's default decode is beam search (translate.py:144: beam_size 5, batch 128, translate.py:208-209: batch 1):
// per step Translator_ARFormer.predict_word / beam_decode_step (models/Translator.py:91-133) runs the decoder on
// clips x beam rows, takes log_softmax, and Beam.advance (misc/Decoding/Beam.py:45-85) picks the beam_size best of the
// beam_size x V continuations of every clip, re-orders the beams and collects ended hypotheses.  Multi-launch that is
// ~25 dependent launches per step (190 us per step at 128 clips x 5: profiles/r03_small_batch_beam5_B128_*); here the
// whole search is ONE launch on the phase machinery of decode_resident.h:
//
//   QKV (+ embedding of the tokens the previous step's advance wrote) | self-attention THROUGH THE ANCESTOR TABLE |
//   dense + residual | { query | attention over the clip's static keys (shared by its beams) | dense + residual } |
//   FFN dense1 | FFN dense2 + residual | vocabulary: per row (max, sum exp) AND its RES_BMK best 4-column groups |
//   beam advance: one workgroup per clip
//
// * Nothing of the [rows, V] logits is stored.  The vocabulary phase keeps, per lane and row, a sorted list of the best
//   4-column groups (one branch-free insertion per 16 x 16 tile), merged per (row, workgroup).  A row's bm best logits lie in
//   its bm best groups (gemm_phase, E_VOCABK), so the advance phase RECOMPUTES just those 4 bm logits per row - the same
//   v_mfma_f32_16x16x32_bf16 chain over the same operands, so the same bits - and selects among them.
// * The beam state is that of csrc/beam.hip (same tables, same quirks of Beam.py): tokphys / ancestor tables (re-ordering
//   beams never moves K / V), finished lists per clip; a clip that is done keeps its tables valid and its rows run on,
//   ignored.  `every clip is done` (Translator.py:77-81) ends the launch.
// * Rows per step go up to 640 (128 clips x 5), so the GEMM phases take several 16-row tiles per workgroup and fetch of
//   the weight fragments (gemm_phase's RTB) - at 40 row tiles a workgroup per (row tile, column item) would stream every
//   weight matrix 40 times through the L2s per step.
#include "decode_beam_phase.h"

namespace {

// KCF = ff / 512.  Row tiles per workgroup and weight fetch: RQ (QKV), RD (the N = 512 products), RF (FFN dense1), RV
// (vocabulary); SM (<= 64 rows): QKV and FFN dense1 in 16-column K-split items, FFN dense2 over two workgroups per
// column tile (the forms of decode_resident.hip).
// D = d_model: 512, or 768 / 1024 (config/archs.yaml:15-26, ff = 4 D, up to 128 rows): the K-split forms in every GEMM phase,
// the vocabulary included, FFN dense2 over two workgroups per column tile - the forms of decode_resident.hip's wide instances.
template <int KCF, int RQ, int RD, int RF, int RV, bool SM, bool KD, int D = 512>  // KD: the N = d_model products in K-split items
__global__ __launch_bounds__(256, 1) void decode_resident_beam_kernel(RArgs p_by_value) {  // (read through the kernarg segment: res_args)
  const ResKArgs kargs = RES_KARGS();
#define p (res_args(kargs))
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
  GridSync gs{p.sync, (unsigned)p.ghost, -1, false, 0, 0, 0u};
  gs.fenced = p.fenced != 0;
  const int d = p.d;
  constexpr bool WIDE = D != 512;
  constexpr bool HF = (KCF == 4 && SM) || WIDE;
  const float* y2 = HF ? p.y2 : nullptr;  // the second K half of FFN dense2, added by its consumers
  bool ended = false;
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
  int t_run = 0;
  {  // the clips' state and the input rows of step 1 (beam_init_phase): a hand-off slot of its own, run once
    gs.prev = 0; gs.want = 0u; gs.cur = RES_MAX_SLOTS - 1;
    np_prev = beam_init_phase<D>(p, gs);
    sl_prev = RES_MAX_SLOTS - 1;
    ex_prev = 1u;
  }
  for (int t = 1; t <= p.steps && !gs.dead; ++t) {
    gs.slot = (p.prof_step == t && blockIdx.x == 0) ? 0 : -1;
    sl = 0;
    bool first_waited = false;
    if (t > 1) {
      // Every clip done with the advance of step t - 1?  EVERY workgroup must come to the same answer before it goes
      // on: the workgroups of the first phase wait for the advance anyway and read the count; workgroup 0 (always one
      // of them) publishes the verdict as 2 t + ended in a word of its own, which is all the others poll.
      const int RTq = (p.R + 15) >> 4, RGq = (RTq + RQ - 1) / RQ, CIq = SM ? (3 * d + 15) >> 4 : (3 * d + 63) >> 6;
      const PhaseMap pq(RGq, CIq);
      bool all_ended;
      if (pq.has) {
        gs.prev = sl_prev;
        gs.want = np_prev * ex_prev;
        gs.wait();
        first_waited = true;
        all_ended = p.early && __hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)p.nclips;
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
      if (all_ended) { ended = true; break; }
    }
    t_run = t;
    for (int l = 0; l < p.n_layers; ++l) {
      const RLayer& L = p.L[l];
      // (layer 0: the rows come embedded and normalised from the phase that chose their tokens - p.xa, the residual in p.xres)
      if (l == 0) RES_PHASE((gemm_phase<D, A_BF16, E_QKV, SM, RQ, D>(p, gs, !first_waited, sA, L.qkv_w, L.qkv_b, 3 * d, p.xa, nullptr, nullptr, false, t, L.skv)));
      else RES_PHASE((gemm_phase<D, A_LN, E_QKV, SM, RQ, D>(p, gs, true, sA, L.qkv_w, L.qkv_b, 3 * d, p.y, p.L[l - 1].fg, p.L[l - 1].fbe, true, t, L.skv, y2)));
      RES_PHASE((p.T <= 32 ? attn_phase<true, 4, true, D>(p, gs, true, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0, p.anc[(t - 1) & 1])
                            : attn_phase<true, RES_MAXKB, true, D>(p, gs, true, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0, p.anc[(t - 1) & 1])));
      RES_PHASE((gemm_phase<D, A_BF16, E_RES, KD, RD, D>(p, gs, true, sA, L.o_w, L.o_b, d, p.ctx, nullptr, nullptr, false, t, nullptr)));
      const float* g = L.g;
      const float* be = L.be;
      for (int a = 0; a < L.n_att; ++a) {
        const RAttn& A = L.att[a];
        RES_PHASE((gemm_phase<D, A_LN, E_Q, KD, RD, D>(p, gs, true, sA, A.q_w, A.q_b, d, p.y, g, be, true, t, nullptr)));
        // more (row, head) pairs than waves: a wave per (clip, head) with the clip's keys fetched once for its beams
        if (p.R * p.H > 4 * (int)gridDim.x)
          RES_PHASE((A.nkeys <= 64 ? attn_shared_phase<8, D>(p, gs, true, A.kv, A.kv_bs, p.bm, A.nkeys, A.bias, A.bias_ld)
                                    : attn_shared_phase<RES_MAXKB, D>(p, gs, true, A.kv, A.kv_bs, p.bm, A.nkeys, A.bias, A.bias_ld)));
        else
          RES_PHASE((A.nkeys <= 64 ? attn_phase<false, 8, false, D>(p, gs, true, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, nullptr, A.bias, A.bias_ld)
                                    : attn_phase<false, RES_MAXKB, false, D>(p, gs, true, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, nullptr, A.bias, A.bias_ld)));
        RES_PHASE((gemm_phase<D, A_BF16, E_RES, KD, RD, D>(p, gs, true, sA, A.o_w, A.o_b, d, p.ctx, nullptr, nullptr, false, t, nullptr)));
        g = A.g; be = A.be;
      }
      RES_PHASE((gemm_phase<D, A_LN, E_ACT, SM, RF, D>(p, gs, true, sA, L.w1, L.b1, p.ff, p.y, g, be, true, t, nullptr)));
      if constexpr (WIDE) RES_PHASE((ffn2_phase<true, 512 * KCF, D>(p, gs, sA, L.w2, L.b2)));
      else if constexpr (KCF == 4) RES_PHASE((ffn2_phase<HF>(p, gs, sA, L.w2, L.b2)));
      else RES_PHASE((gemm_phase<512 * KCF, A_BF16, E_RES, true>(p, gs, true, sA, L.w2, L.b2, d, p.h, nullptr, nullptr, false, t, nullptr)));
    }
    const RLayer& LL = p.L[p.n_layers - 1];
    RES_PHASE((gemm_phase<D, A_LN, E_VOCABK, WIDE, RV, D>(p, gs, true, sA, p.vocab, nullptr, p.V, p.y, LL.fg, LL.fbe, false, t, nullptr, y2, p.vcap)));
    RES_PHASE((beam_advance_phase<D>(p, gs, t, smem)));  // (+ the input rows of step t + 1)
  }
#undef RES_PHASE
  if (gs.dead) {  // aborted (GridSync::wait): every clip's count of finished hypotheses = -1 - written by EVERY workgroup that gave
    // up, each for all clips (a few hundred words): the host must see it whichever workgroups gave up (striped over the grid,
    // 128 clips were workgroup 0's alone to mark)
    for (int b = threadIdx.x; b < p.nclips; b += 256) cst_i(p.nfin + b, -1);
    if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = 0xffffffffu;
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = (unsigned)(ended ? t_run : p.steps);
}

#undef p

constexpr int RES_BEAM_KERNELS = 10;
std::atomic<unsigned long long> g_resb_lds_done[RES_BEAM_KERNELS];
std::atomic<int> g_resb_ok[RES_BEAM_KERNELS];

// parts (vocabulary partials per row) of a grid: workgroups per row group that have a column item (PhaseMap)
inline int beam_parts(int grid, int RG, int CIV) {
  const int nper = ((grid & 7) == 0 && (grid >> 3) >= RG) ? 8 * ((grid >> 3) / RG) : grid / RG;
  return nper < CIV ? nper : CIV;
}

}  // namespace

extern "C" {

#if CARE_RES_BMK == 5
// beam sizes 6 .. 8: this file compiled once more with 8 groups per (row, vocabulary part) - decode_resident_beam_wide.hip
int64_t care_decode_resident_beam8_scratch(int clips, int beam, int d, int ff, int V);
int care_decode_resident_beam8(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                               const float* sem, const float* emb_g, const float* emb_b, float eps, const void* vocab_w,
                               int V, int d, int heads, int ff, int act, int clips, int beam, int need, int T, int steps,
                               int bos, int eos, int pad, int32_t* tok, int stride, int32_t* anc0, int32_t* anc1,
                               float* scores, int32_t* done, int32_t* nfin, float* fscore, int32_t* flen, int32_t* fhyp,
                               int fin_cap, void* scratch, int64_t scratch_bytes, int early_exit, int blocks, void* stream);
#else
#define care_decode_resident_beam_scratch care_decode_resident_beam8_scratch
#define care_decode_resident_beam care_decode_resident_beam8
#endif

int64_t care_decode_resident_beam_scratch(int clips, int beam, int d, int ff, int V) {
  if (clips < 1 || beam < 1 || d < 1 || ff < 1 || V < 1) return CARE_EINVAL;
#if CARE_RES_BMK == 5
  if (beam > RES_BMK) return care_decode_resident_beam8_scratch(clips, beam, d, ff, V);
#endif
  const int64_t rows = (int64_t)clips * beam, R16 = (rows + 15) / 16 * 16;
  int64_t parts = d == 512 ? (V + 63) / 64 : (V + 15) / 16;  // column items of the vocabulary phase (16 columns each when d_model > 512)
  if (parts > 64 * RES_NP) parts = 64 * RES_NP;
  // sync | xres, y, y2, q fp32 [R16, d] | ctx bf16 [R16, d] | h bf16 [R16, ff] | pmax, pidx, psum [R16, parts] |
  // gval, ggid [R16, parts, RES_BMK] | hn bf16 [R16, d]
  return RES_SYNC_BYTES + R16 * d * 4 * 4 + R16 * d * 2 + R16 * ff * 2 + R16 * parts * 12 + R16 * parts * RES_BMK * 8 + R16 * d * 2 * 2;  // (... | xa bf16 [R16, d])
}

int care_decode_resident_beam(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                              const float* sem, const float* emb_g, const float* emb_b, float eps, const void* vocab_w,
                              int V, int d, int heads, int ff, int act, int clips, int beam, int need, int T, int steps,
                              int bos, int eos, int pad, int32_t* tok, int stride, int32_t* anc0, int32_t* anc1,
                              float* scores, int32_t* done, int32_t* nfin, float* fscore, int32_t* flen, int32_t* fhyp,
                              int fin_cap, void* scratch, int64_t scratch_bytes, int early_exit, int blocks, void* stream) {
  if (!layers || !word || !pos || !emb_g || !emb_b || !vocab_w || !tok || !anc0 || !anc1 || !scores || !done || !nfin ||
      !fscore || !flen || !fhyp || !scratch)
    return CARE_EINVAL;
  if (n_layers < 1 || n_layers > RES_MAX_LAYERS || clips < 1 || beam < 1 || need < 1 || fin_cap < 1 || T < 1 || steps < 1 ||
      steps > T || V < 1 || stride < T + 1)
    return CARE_EINVAL;
#if CARE_RES_BMK == 5
  if (beam > RES_BMK && beam <= 8)
    return care_decode_resident_beam8(layers, n_layers, word, pos, sem, emb_g, emb_b, eps, vocab_w, V, d, heads, ff, act, clips, beam,
                                      need, T, steps, bos, eos, pad, tok, stride, anc0, anc1, scores, done, nfin, fscore, flen, fhyp,
                                      fin_cap, scratch, scratch_bytes, early_exit, blocks, stream);
#endif
  // the advance phase keeps a hypothesis' positions one per lane (T + 1 <= 64), a clip's candidates one per lane
  // (beam^2 <= 64) and RES_BMK groups per row
  const bool wide = d != 512;  // d_model 768 / 1024 with ff = 4 d_model, up to 128 rows (the kernel's D)
  if (heads * 64 != d || T > 63 || V > 64 * 64 * RES_NP || beam > RES_BMK || V < 4 * RES_BMK * 4) return CARE_ESHAPE;
  if (!wide && ff != 512 && ff != 1024 && ff != 2048) return CARE_ESHAPE;
  if (wide && ((d != 768 && d != 1024) || ff != 4 * d || (int64_t)clips * beam > 256)) return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if (scratch_bytes < care_decode_resident_beam_scratch(clips, beam, d, ff, V) || !care_aligned16(scratch)) return CARE_EINVAL;
  const int rows = clips * beam;
  RArgs p{};
  if (const int rc = res_fill_layers(p, layers, n_layers)) return rc;
  for (int l = 0; l < n_layers; ++l)
    for (int a = 0; a < p.L[l].n_att; ++a)
      if (p.L[l].att[a].rows_per_kv != beam) return CARE_EINVAL;  // the beams of a clip share its static keys (attn_shared_phase)
  p.word = word; p.pos = pos; p.sem = sem; p.sem_div = beam; p.emb_g = emb_g; p.emb_be = emb_b; p.eps = eps;
  p.vocab = (const bf16_t*)vocab_w; p.V = V;
  p.d = d; p.H = heads; p.ff = ff; p.act = act; p.R = rows; p.T = T; p.steps = steps; p.bos = bos; p.eos = eos; p.pad = pad; p.early = early_exit;
  p.prof_step = care_res_dbg_prof.load();
  p.ghost = care_res_dbg_ghost.load() ? 8 : 0;
  p.fenced = res_fenced_for_device();
  p.fed = tok; p.fed_stride = stride; p.score = scores; p.length = nullptr; p.fin = nullptr;
  p.bm = beam; p.nclips = clips; p.need = need; p.fin_cap = fin_cap;
  p.anc[0] = anc0; p.anc[1] = anc1; p.done = done; p.nfin = nfin; p.fscore = fscore; p.flen = flen; p.fhyp = fhyp;
  const int64_t R16 = (rows + 15) / 16 * 16;
  int64_t maxparts = wide ? (V + 15) / 16 : (V + 63) / 64;
  if (maxparts > 64 * RES_NP) maxparts = 64 * RES_NP;
  unsigned char* b = (unsigned char*)scratch;
  p.sync = (unsigned*)b; b += RES_SYNC_BYTES;
  p.xres = (float*)b; b += R16 * d * 4;
  p.y = (float*)b; b += R16 * d * 4;
  p.y2 = (float*)b; b += R16 * d * 4;
  p.q = (float*)b; b += R16 * d * 4;
  p.ctx = (bf16_t*)b; b += R16 * d * 2;
  p.h = (bf16_t*)b; b += R16 * ff * 2;
  p.pmax = (float*)b; b += R16 * maxparts * 4;
  p.pidx = (int32_t*)b; b += R16 * maxparts * 4;
  p.psum = (float*)b; b += R16 * maxparts * 4;
  p.gval = (float*)b; b += R16 * maxparts * RES_BMK * 4;
  p.ggid = (int32_t*)b; b += R16 * maxparts * RES_BMK * 4;
  p.hn = (bf16_t*)b; b += R16 * d * 2;
  p.xa = (bf16_t*)b;

  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return (int)e;
  // The forms by row count (CARE_RESIDENT_BEAM_CFG = 0 .. 3 forces one): up to 64 rows the K-split forms of the
  // greedy launch; up to 512 rows one row tile per workgroup (two in the vocabulary phase from 128 rows); beyond, several
  // row tiles per workgroup and weight fetch in every GEMM phase (2 / 2 / 2 / 4: form 3; 4 / 2 / 4 / 4: form 2).
  const ResKnobs& kn = res_knobs();
  // (*measured*, us per step of the whole pass, forms 1 / 2 / 3: 260 rows 149 / 162 / 154, 480 rows 161 / 174 / 168,
  // 640 rows 196 / 193 / 188)
  int cfg = rows <= 64 ? 0 : rows <= 512 ? 1 : 3;
  if (kn.beam_cfg >= 0 && kn.beam_cfg <= 3) cfg = kn.beam_cfg;
  if (ff != 2048 || wide) cfg = 0;  // (one form for the narrow FFNs and for d_model 768 / 1024)
  const int RT = (int)(R16 / 16), CIV = wide ? (V + 15) / 16 : (V + 63) / 64;
  const int rv = cfg >= 2 ? 4 : (cfg == 1 && rows >= 128 && ff == 2048) ? 2 : 1;
  const int rq = cfg == 2 ? 4 : cfg == 3 ? 2 : 1, rdd = cfg >= 2 ? 2 : 1, rf = cfg == 2 ? 4 : cfg == 3 ? 2 : 1;
  const int RG = (RT + rv - 1) / rv;
  int want = RG * CIV;
  if ((rows * heads + 3) / 4 > want) want = (rows * heads + 3) / 4;
  if (clips > want) want = clips;
  int grid = blocks > 0 ? blocks : want;
  grid = (grid + 7) / 8 * 8;  // whole rounds over the 8 XCDs (PhaseMap)
  if (grid > cus) grid = cus;
  // a workgroup per row group of every phase at least (the widest count of row groups: one row tile per workgroup)
  const int RGmax = cfg >= 2 ? (RT + rdd - 1) / rdd : RT;
  if (grid < RGmax) return CARE_ESHAPE;
  // the advance phase merges parts x RES_BMK list entries per row, RES_MAXE per lane: the vocabulary phase runs on the
  // first vcap workgroups only - 6 per (row group, XCD) where the grid allows: 48 partial lists per row (*measured* 1 clip
  // x beam 5: 165 lists per row made the advance phase 25 us)
  // ... a single row group (<= 16 rows at one tile per workgroup) takes every workgroup and the two-step fetch instead
  constexpr int LISTS = 64 * RES_MAXE / RES_BMK >= 48 ? 48 : 64 * RES_MAXE / RES_BMK / 8 * 8;   // (48 at 5 groups per list, 32 at 8)
  int vcap = RG == 1 ? grid : (LISTS * RG < grid ? LISTS * RG : grid);
  while (vcap > 8 && beam_parts(vcap, RG, CIV) > maxparts) vcap -= 8;
  p.vcap = vcap;
  p.parts = beam_parts(vcap, RG, CIV);
  if (p.parts > maxparts || p.parts > 64 * RES_NP || p.parts < 1) return CARE_ESHAPE;
  const int kmax = wide ? (ff / 2 > d ? ff / 2 : d) : (ff > d ? ff : d);  // (wide: FFN dense2's tile holds a K half)
  int lds = 16 * (kmax + 8) * 2;
  const int rmax = rv > rq ? (rv > rf ? rv : rf) : (rq > rf ? rq : rf);
  if (rmax * 16 * (512 + 8) * 2 > lds) lds = rmax * 16 * (512 + 8) * 2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(256);
  int rc;
#define RESB_LAUNCH(KCF, RQ, RD, RF, RV, SM, KD, SLOT) RESB_LAUNCH_D(KCF, RQ, RD, RF, RV, SM, KD, 512, SLOT)
#define RESB_LAUNCH_D(KCF, RQ, RD, RF, RV, SM, KD, DM, SLOT)                                                              \
  do {                                                                                                                  \
    const void* kfn = (const void*)decode_resident_beam_kernel<KCF, RQ, RD, RF, RV, SM, KD, DM>;                            \
    if ((rc = care_allow_dynamic_lds(kfn, lds, g_resb_lds_done[SLOT]))) return rc;                                       \
    if (!g_resb_ok[SLOT].load(std::memory_order_acquire)) {                                                             \
      if ((rc = res_check_residency(kfn, lds, grid, cus))) return rc;                                                   \
      g_resb_ok[SLOT].store(1, std::memory_order_release);                                                              \
    }                                                                                                                   \
    if ((e = res_zero_words(p.sync, RES_SYNC_BYTES, st)) != hipSuccess) return (int)e;                               \
    hipLaunchKernelGGL((decode_resident_beam_kernel<KCF, RQ, RD, RF, RV, SM, KD, DM>), g, blk, lds, st, p);                 \
  } while (0)
  if (d == 768) RESB_LAUNCH_D(6, 1, 1, 1, 1, true, true, 768, 8);
  else if (d == 1024) RESB_LAUNCH_D(8, 1, 1, 1, 1, true, true, 1024, 9);
  else if (ff == 512) RESB_LAUNCH(1, 1, 1, 1, 1, true, true, 0);
  else if (ff == 1024) RESB_LAUNCH(2, 1, 1, 1, 1, true, true, 1);
  else if (cfg == 0) RESB_LAUNCH(4, 1, 1, 1, 1, true, true, 2);
  else if (cfg == 1 && rv == 2) RESB_LAUNCH(4, 1, 1, 1, 2, false, true, 3);
  else if (cfg == 1) RESB_LAUNCH(4, 1, 1, 1, 1, false, true, 4);
  else if (cfg == 3) RESB_LAUNCH(4, 2, 2, 2, 4, false, false, 6);
  else RESB_LAUNCH(4, 4, 2, 4, 4, false, false, 5);
#undef RESB_LAUNCH
#undef RESB_LAUNCH_D
  return care_launch_status();
}

}  // extern "C"
