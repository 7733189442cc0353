// decode_chain.hip - a BEAM-SEARCH STEP as a chain of kernels: the phases of the resident beam launch
// (decode_resident_beam.hip), each as a launch of its own with a grid and a register budget of its own.
//
// Why (round 5): the one-launch search stops paying at a few hundred rows.  Its grid is one workgroup
// per CU for ALL phases (1 wave per SIMD, 512 registers, a 78 K-instruction body whose phase code is cold once per step),
// a hand-off between phases costs 3.5 us at 640 rows - more than a kernel boundary inside a hipGraph (1.5 - 1.9 us) -
// and every byte that crosses a hand-off travels the coherent (sc1) path past the L2s.  Here every phase is a kernel:
//
//   QKV (embedding + LayerNorm on load) | self-attention through the ancestor table | dense + residual |
//   { query (LayerNorm on load) | attention over the clip's static keys | dense + residual } per cross / attribute block |
//   FFN dense1 (LayerNorm on load) | FFN dense2 + residual | vocabulary: (max, sum exp) + best groups per (row, part) |
//   advance (log-softmax of the recomputed candidates + Beam.advance, one workgroup per clip)
//
// = 10 launches per step for the one-layer models, captured with all steps of a segment into one hipGraph by the engine.
// The DEVICE code is decode_resident.h's, compiled with RES_PLAIN_IO: plain loads / stores instead of agent-scope ones
// (nothing is handed over inside a launch; the kernel boundary orders the phases) and an empty GridSync.  So a row's
// arithmetic - operand roundings, K order, LayerNorm on load, group lists, recomputed candidate logits - is the resident
// launch's, bit for bit: tests/test_gpu_chain.py asserts IDENTICAL hypotheses and scores between the two forms.
// What differs is the execution: grids sized by the phase's items (hundreds to thousands of workgroups), two workgroups
// per CU where the registers allow, no residency condition - so no row limit from the CU count: 64 .. 4096+ rows.
//
// Reference: models/Translator.py:77-143 (step loop, predict_word's log_softmax), misc/Decoding/Beam.py:45-85
// (Beam.advance and its quirks), models/components/Layers.py:157-228 (DecoderLayer), Head.py:26-32.
#define RES_PLAIN_IO 1
#include "decode_beam_phase.h"

namespace {

constexpr unsigned CH_NO_GHOST = 0u;
__device__ __forceinline__ GridSync chain_sync(const RArgs& p) { return GridSync{p.sync, CH_NO_GHOST, -1, false, 0, 0, 0u}; }

// One GEMM phase as a kernel (gemm_phase's template parameters; the operands of THIS phase are kernel arguments, RArgs
// carries the batch-wide state).  MINB: workgroups per CU the register allocation must admit.
template <int K, int AMODE, int EPI, bool KSPLIT, int RTB, int D, int MINB>
__global__ __launch_bounds__(256, MINB) void chain_gemm_kernel(RArgs p, const bf16_t* W, const float* bias, int N, const void* asrc,
                                                               const float* g, const float* be, int write_x, int t, bf16_t* skv,
                                                               const float* asrc2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  GridSync gs = chain_sync(p);
  gemm_phase<K, AMODE, EPI, KSPLIT, RTB, D>(p, gs, false, reinterpret_cast<bf16_t*>(smem), W, bias, N, asrc, g, be, write_x != 0, t, skv, asrc2, 0);
}

template <bool HALF, int KF, int D>
__global__ __launch_bounds__(256, HALF ? 2 : 1) void chain_ffn2_kernel(RArgs p, const bf16_t* W, const float* bias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  GridSync gs = chain_sync(p);
  ffn2_phase<HALF, KF, D>(p, gs, reinterpret_cast<bf16_t*>(smem), W, bias);
}

template <bool SELF, int NKB, bool ANC, int D>
__global__ __launch_bounds__(256, NKB > 8 ? 1 : 2) void chain_attn_kernel(RArgs p, const bf16_t* KV, int64_t kv_bs, int rows_per_kv, int nk,
                                                            const int32_t* pad_tok, const float* bias, int bias_ld, const int32_t* anc) {
  GridSync gs = chain_sync(p);
  attn_phase<SELF, NKB, ANC, D>(p, gs, false, KV, kv_bs, rows_per_kv, nk, pad_tok, bias, bias_ld, anc);
}

template <int NKB>
__global__ __launch_bounds__(256, NKB > 8 ? 1 : 2) void chain_attn_shared_kernel(RArgs p, const bf16_t* KV, int64_t kv_bs, int bm, int nk,
                                                                   const float* bias, int bias_ld) {
  GridSync gs = chain_sync(p);
  attn_shared_phase<NKB>(p, gs, false, KV, kv_bs, bm, nk, bias, bias_ld);
}

__global__ __launch_bounds__(256, 1) void chain_advance_kernel(RArgs p, int t) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(RES_BEAM_LDS + 15) / 16 * 16];
  GridSync gs = chain_sync(p);
  beam_advance_phase(p, gs, t, smem);
}

__global__ __launch_bounds__(256, 1) void chain_init_kernel(RArgs p) {
  GridSync gs = chain_sync(p);
  beam_init_phase(p, gs);
}

// grid of a GEMM phase: 8 XCD slots x row groups x column-item slots per XCD (PhaseMap: workgroup b -> XCD b & 7, slot
// b >> 3 = cs * RG + row group; its items are c0 = x + 8 cs, + 8 nsl, ...): one item per workgroup when nsl covers the items
inline int gemm_grid(int RG, int CI, int nsl_cap) {
  int nsl = (CI + 7) / 8;
  if (nsl_cap > 0 && nsl > nsl_cap) nsl = nsl_cap;
  return 8 * RG * nsl;
}

struct ChainKnobs {
  int cfg, shared_min;  // -1: not set
  ChainKnobs() {
    auto geti = [](const char* n) { const char* e = getenv(n); return e ? atoi(e) : -1; };
    cfg = geti("CARE_CHAIN_CFG"); shared_min = geti("CARE_CHAIN_SHARED_MIN_ROWS");
  }
};
inline const ChainKnobs& chain_knobs() {
  static const ChainKnobs k;
  return k;
}

}  // namespace

extern "C" {

int64_t care_decode_chain_beam_scratch(int clips, int beam, int d, int ff, int V) {
  return care_decode_resident_beam_scratch(clips, beam, d, ff, V);  // the same buffers (no hand-off counters are used)
}

int care_decode_chain_beam(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                           const float* sem, const float* emb_g, const float* emb_b, float eps, const void* vocab_w,
                           int V, int d, int heads, int ff, int act, int clips, int beam, int need, int T, int t0, int t1,
                           int bos, int eos, int pad, int32_t* tok, int stride, int32_t* anc0, int32_t* anc1,
                           float* scores, int32_t* done, int32_t* nfin, float* fscore, int32_t* flen, int32_t* fhyp,
                           int fin_cap, void* scratch, int64_t scratch_bytes, int form, void* stream) {
  if (!layers || !word || !pos || !emb_g || !emb_b || !vocab_w || !tok || !anc0 || !anc1 || !scores || !done || !nfin ||
      !fscore || !flen || !fhyp || !scratch)
    return CARE_EINVAL;
  if (n_layers < 1 || n_layers > RES_MAX_LAYERS || clips < 1 || beam < 1 || need < 1 || fin_cap < 1 || T < 1 || t0 < 1 ||
      t1 < t0 || t1 > T || V < 1 || stride < T + 1)
    return CARE_EINVAL;
  // (the limits of the resident beam launch's phases: a hypothesis' positions one per lane, a clip's candidates one per
  // lane, RES_BMK groups per row; no limit on the rows)
  if (d != 512 || heads * 64 != d || (ff != 512 && ff != 1024 && ff != 2048) || T > 63 || V > 64 * 64 * RES_NP || beam > RES_BMK ||
      V < 4 * RES_BMK * 4)
    return CARE_ESHAPE;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  if (scratch_bytes < care_decode_chain_beam_scratch(clips, beam, d, ff, V) || !care_aligned16(scratch)) return CARE_EINVAL;
  const int64_t rows64 = (int64_t)clips * beam;
  if (rows64 > (1 << 20)) return CARE_ESHAPE;
  const int rows = (int)rows64;
  RArgs p{};
  if (const int rc = res_fill_layers(p, layers, n_layers)) return rc;
  for (int l = 0; l < n_layers; ++l)
    for (int a = 0; a < p.L[l].n_att; ++a)
      if (p.L[l].att[a].rows_per_kv != beam) return CARE_EINVAL;  // the beams of a clip share its static keys
  p.word = word; p.pos = pos; p.sem = sem; p.sem_div = beam; p.emb_g = emb_g; p.emb_be = emb_b; p.eps = eps;
  p.vocab = (const bf16_t*)vocab_w; p.V = V;
  p.d = d; p.H = heads; p.ff = ff; p.act = act; p.R = rows; p.T = T; p.steps = t1; p.bos = bos; p.eos = eos; p.pad = pad; p.early = 0;
  p.prof_step = 0; p.ghost = 0; p.fenced = 0;
  p.fed = tok; p.fed_stride = stride; p.score = scores; p.length = nullptr; p.fin = nullptr;
  p.bm = beam; p.nclips = clips; p.need = need; p.fin_cap = fin_cap;
  p.anc[0] = anc0; p.anc[1] = anc1; p.done = done; p.nfin = nfin; p.fscore = fscore; p.flen = flen; p.fhyp = fhyp;
  const int64_t R16 = ((int64_t)rows + 15) / 16 * 16;
  int64_t maxparts = (V + 63) / 64;
  if (maxparts > 64 * RES_NP) maxparts = 64 * RES_NP;
  unsigned char* b = (unsigned char*)scratch;
  p.sync = (unsigned*)b; b += RES_SYNC_BYTES;  // (word 1: clips done - the advance kernel's counter)
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

  // Forms by row count (CARE_CHAIN_CFG / `form` >= 0 force one), the resident beam launch's (same bits in every form of a
  // phase, see gemm_phase): 0 (<= 64 rows) K-split items everywhere, FFN dense2 over two workgroups per column tile;
  // 1 one row tile per workgroup, 64-column items for QKV / FFN dense1 / vocabulary (two row tiles per vocabulary fetch
  // from 128 rows); 3 two row tiles per weight fetch in QKV / the N = 512 products / FFN dense1, four in the vocabulary.
  const ChainKnobs& kn = chain_knobs();
  int cfg = rows <= 64 ? 0 : rows <= 256 ? 1 : 3;
  if (kn.cfg >= 0 && kn.cfg <= 3) cfg = kn.cfg;
  if (form >= 0 && form <= 3) cfg = form;
  if (cfg == 2) cfg = 3;
  if (ff != 2048) cfg = 0;  // (one form for the narrow FFNs)
  const int RT = (int)(R16 / 16), CIV = (V + 63) / 64;
  const int rv = cfg >= 2 ? 4 : (cfg == 1 && rows >= 128) ? 2 : 1;
  const int rq = cfg == 3 ? 2 : 1, rdd = cfg == 3 ? 2 : 1, rf = cfg == 3 ? 2 : 1;
  const bool sm = cfg == 0, kd = cfg < 2;
  const int RGv = (RT + rv - 1) / rv;
  // vocabulary parts per row: 48 (6 column-item slots per XCD; the advance merges parts x RES_BMK list entries per row,
  // RES_MAXE per lane); a single row group takes a part per item and the two-step list fetch (beam_advance_phase)
  const int nsl_v = RGv == 1 ? (CIV + 7) / 8 : 6;
  p.vcap = 0;
  p.parts = 8 * nsl_v < CIV ? 8 * nsl_v : CIV;
  if (p.parts > maxparts || p.parts < 1) return CARE_ESHAPE;
  const int shared_min = kn.shared_min >= 0 ? kn.shared_min : 128;  // rows from which the beams of a clip share a K/V fetch
  const bool shared = beam > 1 && rows >= shared_min;

  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (t0 == 1 && (e = res_zero_words(p.sync, 4096, st)) != hipSuccess) return (int)e;
  int rc = 0;
#define CH_LAUNCH(KERNEL, GRID, LDS, ...)                                                               \
  do {                                                                                                  \
    static std::atomic<unsigned long long> lds_done{0};                                                 \
    const int lds_ = (LDS);                                                                             \
    if (lds_ > 32768 && (rc = care_allow_dynamic_lds((const void*)KERNEL, lds_, lds_done))) return rc;  \
    hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(256), lds_, st, __VA_ARGS__);                           \
  } while (0)
  auto lds_a = [](int rtb, int K) { return rtb * 16 * (K + 8) * 2; };
  const bf16_t* nob = nullptr;
  const float* nof = nullptr;

  if (t0 == 1) CH_LAUNCH(chain_init_kernel, clips, 0, p);  // the clips' state + the input rows of step 1
  for (int t = t0; t <= t1; ++t) {
    const int32_t* anc_old = p.anc[(t - 1) & 1];
    for (int l = 0; l < p.n_layers; ++l) {
      const RLayer& L = p.L[l];
      const float* y2 = (ff == 2048 && sm) ? p.y2 : nof;  // the second K half of FFN dense2 (two workgroups per column tile)
      // ---- QKV (+ embedding / LayerNorm on load)
      {
        const int N = 3 * d, CI = sm ? (N + 15) / 16 : (N + 63) / 64, RG = (RT + rq - 1) / rq, G = gemm_grid(RG, CI, 0);
        const void* asrc = l == 0 ? (const void*)p.xa : (const void*)p.y;  // (layer 0: embedded + normalised by the advance / init kernel)
        const float* g = l == 0 ? p.emb_g : p.L[l - 1].fg;
        const float* be = l == 0 ? p.emb_be : p.L[l - 1].fbe;
        const float* a2 = l == 0 ? nof : y2;
        if (l == 0) {
          if (sm) CH_LAUNCH((chain_gemm_kernel<512, A_BF16, E_QKV, true, 1, 512, 2>), G, lds_a(1, 512), p, L.qkv_w, L.qkv_b, N, asrc, nof, nof, 0, t, L.skv, nof);
          else if (rq == 1) CH_LAUNCH((chain_gemm_kernel<512, A_BF16, E_QKV, false, 1, 512, 2>), G, lds_a(1, 512), p, L.qkv_w, L.qkv_b, N, asrc, nof, nof, 0, t, L.skv, nof);
          else CH_LAUNCH((chain_gemm_kernel<512, A_BF16, E_QKV, false, 2, 512, 2>), G, lds_a(2, 512), p, L.qkv_w, L.qkv_b, N, asrc, nof, nof, 0, t, L.skv, nof);
        } else {
          if (sm) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_QKV, true, 1, 512, 2>), G, lds_a(1, 512), p, L.qkv_w, L.qkv_b, N, asrc, g, be, 1, t, L.skv, a2);
          else if (rq == 1) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_QKV, false, 1, 512, 2>), G, lds_a(1, 512), p, L.qkv_w, L.qkv_b, N, asrc, g, be, 1, t, L.skv, a2);
          else CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_QKV, false, 2, 512, 2>), G, lds_a(2, 512), p, L.qkv_w, L.qkv_b, N, asrc, g, be, 1, t, L.skv, a2);
        }
      }
      // ---- self-attention through the ancestor table: a wave per (row, head)
      {
        const int items = ((rows + 7) / 8) * heads, G = 8 * ((items + 3) / 4);
        if (T <= 32) CH_LAUNCH((chain_attn_kernel<true, 4, true, 512>), G, 0, p, (const bf16_t*)L.skv, (int64_t)T * 2 * d, 1, t, (const int32_t*)p.fed, nof, 0, anc_old);
        else CH_LAUNCH((chain_attn_kernel<true, RES_MAXKB, true, 512>), G, 0, p, (const bf16_t*)L.skv, (int64_t)T * 2 * d, 1, t, (const int32_t*)p.fed, nof, 0, anc_old);
      }
      // ---- dense + residual, then per static-key block: query | attention | dense + residual
      auto dense_res = [&](const bf16_t* W, const float* bias) -> int {
        const int CI = kd ? d / 16 : d / 64, RG = (RT + rdd - 1) / rdd, G = gemm_grid(RG, CI, 0);
        if (kd) CH_LAUNCH((chain_gemm_kernel<512, A_BF16, E_RES, true, 1, 512, 2>), G, lds_a(1, 512), p, W, bias, d, (const void*)p.ctx, nof, nof, 0, t, (bf16_t*)nullptr, nof);
        else CH_LAUNCH((chain_gemm_kernel<512, A_BF16, E_RES, false, 2, 512, 2>), G, lds_a(2, 512), p, W, bias, d, (const void*)p.ctx, nof, nof, 0, t, (bf16_t*)nullptr, nof);
        return 0;
      };
      if ((rc = dense_res(L.o_w, L.o_b))) return rc;
      const float* g = L.g;
      const float* be = L.be;
      for (int a = 0; a < L.n_att; ++a) {
        const RAttn& A = L.att[a];
        {
          const int CI = kd ? d / 16 : d / 64, RG = (RT + rdd - 1) / rdd, G = gemm_grid(RG, CI, 0);
          if (kd) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_Q, true, 1, 512, 2>), G, lds_a(1, 512), p, A.q_w, A.q_b, d, (const void*)p.y, g, be, 1, t, (bf16_t*)nullptr, nof);
          else CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_Q, false, 2, 512, 2>), G, lds_a(2, 512), p, A.q_w, A.q_b, d, (const void*)p.y, g, be, 1, t, (bf16_t*)nullptr, nof);
        }
        if (shared) {
          const int items = ((clips + 7) / 8) * heads, G = 8 * ((items + 3) / 4);
          if (A.nkeys <= 64) CH_LAUNCH((chain_attn_shared_kernel<8>), G, 0, p, A.kv, A.kv_bs, beam, A.nkeys, A.bias, A.bias_ld);
          else CH_LAUNCH((chain_attn_shared_kernel<RES_MAXKB>), G, 0, p, A.kv, A.kv_bs, beam, A.nkeys, A.bias, A.bias_ld);
        } else {
          const int items = ((rows + 7) / 8) * heads, G = 8 * ((items + 3) / 4);
          const int32_t* noi = nullptr;
          if (A.nkeys <= 64) CH_LAUNCH((chain_attn_kernel<false, 8, false, 512>), G, 0, p, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, noi, A.bias, A.bias_ld, noi);
          else CH_LAUNCH((chain_attn_kernel<false, RES_MAXKB, false, 512>), G, 0, p, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, noi, A.bias, A.bias_ld, noi);
        }
        if ((rc = dense_res(A.o_w, A.o_b))) return rc;
        g = A.g; be = A.be;
      }
      // ---- FFN dense1 (+ activation), FFN dense2 + residual
      {
        const int N = ff, CI = sm ? (N + 15) / 16 : (N + 63) / 64, RG = (RT + rf - 1) / rf, G = gemm_grid(RG, CI, 0);
        if (sm) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_ACT, true, 1, 512, 2>), G, lds_a(1, 512), p, L.w1, L.b1, N, (const void*)p.y, g, be, 1, t, (bf16_t*)nullptr, nof);
        else if (rf == 1) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_ACT, false, 1, 512, 2>), G, lds_a(1, 512), p, L.w1, L.b1, N, (const void*)p.y, g, be, 1, t, (bf16_t*)nullptr, nof);
        else CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_ACT, false, 2, 512, 2>), G, lds_a(2, 512), p, L.w1, L.b1, N, (const void*)p.y, g, be, 1, t, (bf16_t*)nullptr, nof);
      }
      if (ff == 2048) {
        if (sm) CH_LAUNCH((chain_ffn2_kernel<true, 2048, 512>), gemm_grid(RT, d / 8, 0), 16 * (1024 + 8) * 2, p, L.w2, L.b2);
        else CH_LAUNCH((chain_ffn2_kernel<false, 2048, 512>), gemm_grid(RT, d / 16, 0), 16 * (2048 + 8) * 2, p, L.w2, L.b2);
      } else if (ff == 1024) {
        CH_LAUNCH((chain_gemm_kernel<1024, A_BF16, E_RES, true, 1, 512, 2>), gemm_grid(RT, d / 16, 0), lds_a(1, 1024), p, L.w2, L.b2, d, (const void*)p.h, nof, nof, 0, t, (bf16_t*)nullptr, nof);
      } else {
        CH_LAUNCH((chain_gemm_kernel<512, A_BF16, E_RES, true, 1, 512, 2>), gemm_grid(RT, d / 16, 0), lds_a(1, 512), p, L.w2, L.b2, d, (const void*)p.h, nof, nof, 0, t, (bf16_t*)nullptr, nof);
      }
    }
    // ---- vocabulary: per (row, part) the maximum, the sum of exponentials and the RES_BMK best 4-column groups
    {
      const RLayer& LL = p.L[p.n_layers - 1];
      const float* y2 = (ff == 2048 && sm) ? p.y2 : nof;
      const int G = gemm_grid(RGv, CIV, nsl_v);
      if (rv == 1) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_VOCABK, false, 1, 512, 2>), G, lds_a(1, 512), p, p.vocab, nof, V, (const void*)p.y, LL.fg, LL.fbe, 0, t, (bf16_t*)nullptr, y2);
      else if (rv == 2) CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_VOCABK, false, 2, 512, 2>), G, lds_a(2, 512), p, p.vocab, nof, V, (const void*)p.y, LL.fg, LL.fbe, 0, t, (bf16_t*)nullptr, y2);
      else CH_LAUNCH((chain_gemm_kernel<512, A_LN, E_VOCABK, false, 4, 512, 1>), G, lds_a(4, 512), p, p.vocab, nof, V, (const void*)p.y, LL.fg, LL.fbe, 0, t, (bf16_t*)nullptr, y2);
    }
    // ---- log-softmax of the candidates + Beam.advance: a workgroup per clip
    CH_LAUNCH(chain_advance_kernel, clips, 0, p, t);
  }
#undef CH_LAUNCH
  (void)nob;
  return care_launch_status();
}

}  // extern "C"
