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
#include "decode_resident.h"

namespace {

constexpr int RES_MAXE = 4;   // group-list entries per lane in the advance phase: parts x RES_BMK <= 64 x RES_MAXE (parts <= 48: RArgs::vcap)

// One phase: log_softmax + Beam.advance for every clip (models/Translator.py:127, misc/Decoding/Beam.py:45-85).
// One workgroup per clip.  Waves 0 .. 3 take the clip's rows (row i -> wave i % 4): log-sum-exp of the row from the
// vocabulary partials, the row's bm best groups from the parts' lists (per-lane sorted lists, then bm rounds of
// `largest head in the wave`), the 4 bm logits of those groups recomputed (see above), the bm best of them as
// log-probabilities (value desc, column asc) -> LDS.  The LAST wave (one row at most) then runs the state machine of
// csrc/beam.hip's beam_advance_wave_kernel on the bm x bm candidates (lane = candidate; ancestor rows one position per
// lane).  What bounds the phase is the number of DEPENDENT memory round trips (~1.5 us each through the coherent
// path), so: every row's partials are requested before the first is used (a wave with two rows has both in flight),
// and ALL the state the advance reads - flags, scores, the clip's rows of the ancestor and token tables, whichever
// parents win - is requested by its wave at the top of the phase: after the barrier it computes and stores.
struct BeamRowIn {
  float pm[RES_NP], ps[RES_NP];
  float ev[RES_MAXE];
  int eg[RES_MAXE];
};

RES_PHASE_FN unsigned beam_advance_phase(const RArgs& p, GridSync& gs, int t, unsigned char* scratch_lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kg = lane >> 4;
  const int G = gridDim.x, bm = p.bm, stride = p.fed_stride;
  const unsigned nprod = (unsigned)(p.nclips < G ? p.nclips : G);
  const bool participant = (int)blockIdx.x < p.nclips;
  float (*s_cv)[RES_BMK] = reinterpret_cast<float (*)[RES_BMK]>(scratch_lds);
  int (*s_ci)[RES_BMK] = reinterpret_cast<int (*)[RES_BMK]>(scratch_lds + 8 * RES_BMK * 4);
  if (gs.dead) return nprod;
  if (participant) gs.wait();
  if (gs.dead) return nprod;
  gs.mark();
  const int32_t* anc_old = p.anc[(t - 1) & 1];
  int32_t* anc_new = p.anc[t & 1];
  // Few row tiles (one clip x beam 5 = ONE tile): the vocabulary phase runs on every workgroup the matrix has column
  // items for (165 partial lists per row; capped at 48 it took 13 instead of 5 us), and the lists are fetched in TWO
  // steps: the row's best bm PARTS first - a part's list is sorted, so every group at least as large as the bm-th
  // largest list head sits in one of the bm parts with the largest heads (the heads = the partial maxima the log-sum-exp
  // reads anyway) - then those bm x RES_BMK entries.  One more dependent fetch, 25 entries instead of 825.
  const bool two_step = p.parts * RES_BMK > 64 * RES_MAXE;
  const int NE = two_step ? RES_BMK * RES_BMK : p.parts * RES_BMK, NEL = (NE + 63) >> 6;
  for (int c = blockIdx.x; c < p.nclips; c += G) {
    const int row0 = c * bm;
    // ---- the advance wave's state, requested before anything else (lane = position j of the tables)
    int st_done = 0, st_nf = 0, st_anc[RES_BMK], st_tok[RES_BMK];
    float st_sc[RES_BMK];
    if (wave == 3) {
      st_done = cld_i(p.done + c);
      st_nf = cld_i(p.nfin + c);
#pragma unroll
      for (int i = 0; i < RES_BMK; ++i) {
        const int64_t o = (int64_t)(row0 + (i < bm ? i : 0)) * stride + (lane < stride ? lane : 0);
        st_anc[i] = cld_i(anc_old + o);
        st_tok[i] = cld_i(p.fed + o);
        st_sc[i] = cld_f(p.score + row0 + (i < bm ? i : 0));
      }
    }
    // ---- the partials of this wave's rows (i = wave, wave + 4), all requested before the first is used
    BeamRowIn in[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int i = wave + 4 * s;
      if (i >= bm) break;  // (wave-uniform)
      const int r = row0 + i;
#pragma unroll
      for (int k = 0; k < RES_NP; ++k) {
        const int cpart = lane + 64 * k;
        const bool ok = cpart < p.parts;
        in[s].pm[k] = cld_f(p.pmax + (int64_t)r * p.parts + (ok ? cpart : 0));
        in[s].ps[k] = cld_f(p.psum + (int64_t)r * p.parts + (ok ? cpart : 0));
        if (!ok) { in[s].pm[k] = -INFINITY; in[s].ps[k] = 0.f; }
      }
      if (!two_step) {
#pragma unroll
        for (int k = 0; k < RES_MAXE; ++k)
          if (k < NEL) {
            const int e = lane + 64 * k;
            const bool ok = e < NE;
            in[s].ev[k] = cld_f(p.gval + (int64_t)r * NE + (ok ? e : 0));
            in[s].eg[k] = cld_i(p.ggid + (int64_t)r * NE + (ok ? e : 0));
            if (!ok) { in[s].ev[k] = -INFINITY; in[s].eg[k] = 0x7fffffff; }
          }
      }
    }
    if (two_step) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int i = wave + 4 * s;
        if (i >= bm) break;  // (wave-uniform)
        const int r = row0 + i;
        // the bm parts with the largest heads (value desc, part asc: a lower part holds lower columns)
        unsigned long long hk[RES_NP];
#pragma unroll
        for (int k = 0; k < RES_NP; ++k) hk[k] = in[s].pm[k] == -INFINITY ? 0ull : key_of(in[s].pm[k], (unsigned)(lane + 64 * k));
        int myp = 0;  // lane j < bm * RES_BMK fetches entry j % RES_BMK of the (j / RES_BMK)-th best part
#pragma unroll
        for (int k = 0; k < RES_BMK; ++k) {
          unsigned long long loc = hk[0];
#pragma unroll
          for (int q = 1; q < RES_NP; ++q) loc = max_u64(loc, hk[q]);
          const unsigned long long best = wave_max_u64(loc);
#pragma unroll
          for (int q = 0; q < RES_NP; ++q)
            if (hk[q] == best) hk[q] = 0ull;
          if (lane / RES_BMK == k) myp = best ? (int)key_idx(best) : -1;
        }
        const bool ok = lane < RES_BMK * RES_BMK && myp >= 0;
        const int64_t o = ((int64_t)r * p.parts + (ok ? myp : 0)) * RES_BMK + lane % RES_BMK;
        in[s].ev[0] = cld_f(p.gval + o);
        in[s].eg[0] = cld_i(p.ggid + o);
        if (!ok) { in[s].ev[0] = -INFINITY; in[s].eg[0] = 0x7fffffff; }
      }
    }
    // The three stages of a row, as lambdas so that a wave with two rows can run the second row's group selection while
    // the first row's weight fragments travel:
    //   groups: the row's bm best groups from the parts' lists;
    //   issue : their 4 bm logits again - MFMA rows = the columns of groups (tile 0: groups 0 .. 3, tile 1: 4 .. 7), lane
    //           (l16, kg) fetches the weight row of column 4 gsel[4 tile + l16 / 4] + l16 % 4 and receives the logits of
    //           columns 4 gsel[4 tile + kg] + 0 .. 3; both tiles' weight rows and the row's hidden state (B fragments:
    //           every column of the MFMA = this row) are requested together;
    //   finish: log-sum-exp of the row, the products in gemm_phase's accumulation order (two chains over even / odd k
    //           fragments), one candidate per lane (l16 < 8: tile l16 / 4, element l16 % 4 of the lane's group), bm
    //           rounds of arg-best (value desc, column asc) -> LDS.
    auto groups = [&](int sl, int (&gsel)[RES_BMK]) {
      unsigned long long hk[RES_BMK];
#pragma unroll
      for (int k = 0; k < RES_BMK; ++k) hk[k] = 0ull;
#pragma unroll
      for (int k = 0; k < RES_MAXE; ++k)
        if (k < NEL) {
          unsigned long long x = in[sl].ev[k] == -INFINITY ? 0ull : key_of(in[sl].ev[k], (unsigned)in[sl].eg[k]);
#pragma unroll
          for (int j = 0; j < RES_BMK; ++j) {
            const bool gt = x > hk[j];
            const unsigned long long hi = gt ? x : hk[j];
            x = gt ? hk[j] : x;
            hk[j] = hi;
          }
        }
#pragma unroll
      for (int k = 0; k < RES_BMK; ++k) {
        const unsigned long long best = wave_max_u64(hk[0]);
        gsel[k] = best ? (int)key_idx(best) : 0;
        if (hk[0] == best) {
#pragma unroll
          for (int j = 0; j + 1 < RES_BMK; ++j) hk[j] = hk[j + 1];
          hk[RES_BMK - 1] = 0ull;
        }
      }
    };
    auto issue = [&](int r, const int (&gsel)[RES_BMK], int (&gout)[2], bf16x8 (&wf0)[16], bf16x8 (&wf1)[16], bf16x8 (&af)[16]) {
      int gl[2];
#pragma unroll
      for (int tile = 0; tile < 2; ++tile) {
        gl[tile] = gout[tile] = gsel[tile * 4 < RES_BMK ? tile * 4 : 0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (tile * 4 + q < RES_BMK) {
            if ((l16 >> 2) == q) gl[tile] = gsel[tile * 4 + q];
            if (kg == q) gout[tile] = gsel[tile * 4 + q];
          }
      }
      load_w<16>(wf0, p.vocab + (int64_t)min(gl[0] * 4 + (l16 & 3), p.V - 1) * 512 + kg * 8);
      if (bm > 4) load_w<16>(wf1, p.vocab + (int64_t)min(gl[1] * 4 + (l16 & 3), p.V - 1) * 512 + kg * 8);
#pragma unroll
      for (int q = 0; q < 16; ++q) af[q] = cld_b8(p.hn + (int64_t)r * 512 + kg * 8 + q * 32);
    };
    auto finish = [&](int sl, int i, const int (&gout)[2], const bf16x8 (&wf0)[16], const bf16x8 (&wf1)[16], const bf16x8 (&af)[16]) {
      float mloc = in[sl].pm[0];
#pragma unroll
      for (int k = 1; k < RES_NP; ++k) mloc = fmaxf(mloc, in[sl].pm[k]);
      const float M = care_wave_max_dpp(mloc);
      float sloc = 0.f;
#pragma unroll
      for (int k = 0; k < RES_NP; ++k) sloc += in[sl].pm[k] == -INFINITY ? 0.f : in[sl].ps[k] * expf(in[sl].pm[k] - M);
      const float logS = logf(wave_sum_dpp(sloc));
      f32x4 vt[2];
      {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          acc0 = care_mfma_16x16x32_h16(wf0[q], af[q], acc0, 0, 0, 0);
          acc1 = care_mfma_16x16x32_h16(wf0[q + 1], af[q + 1], acc1, 0, 0, 0);
        }
        vt[0] = acc0 + acc1;
      }
      vt[1] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (bm > 4) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          acc0 = care_mfma_16x16x32_h16(wf1[q], af[q], acc0, 0, 0, 0);
          acc1 = care_mfma_16x16x32_h16(wf1[q + 1], af[q + 1], acc1, 0, 0, 0);
        }
        vt[1] = acc0 + acc1;
      }
      const int ctile = (l16 >> 2) & 1, ce = l16 & 3, cgrp = ctile * 4 + kg;
      const f32x4 vsel = ctile ? vt[1] : vt[0];
      const float cval = ce == 0 ? vsel[0] : ce == 1 ? vsel[1] : ce == 2 ? vsel[2] : vsel[3];
      const int ccol = (ctile ? gout[1] : gout[0]) * 4 + ce;
      unsigned long long ck = (l16 < 8 && cgrp < bm && ccol < p.V) ? key_of(cval, (unsigned)ccol) : 0ull;
#pragma unroll
      for (int k = 0; k < RES_BMK; ++k) {
        if (k >= bm) break;
        const unsigned long long best = wave_max_u64(ck);
        if (ck == best) ck = 0ull;
        if (lane == 0) {  // log_softmax = (x - max) - log(sum), as csrc/beam.hip; no candidate left: (-inf, 0)
          s_cv[i][k] = best ? (key_val(best) - M) - logS : -INFINITY;
          s_ci[i][k] = best ? (int)key_idx(best) : 0;
        }
      }
    };
    if (wave < bm) {
      const bool two = wave + 4 < bm;  // (wave-uniform)
      int gsel[RES_BMK], gsel1[RES_BMK], gout[2];
      bf16x8 wf0[16], wf1[16], af[16];
      groups(0, gsel);
      gs.mark();
      issue(row0 + wave, gsel, gout, wf0, wf1, af);
      if (two) groups(1, gsel1);  // (while the first row's fragments travel)
      finish(0, wave, gout, wf0, wf1, af);
      gs.mark();
      if (two) {
        issue(row0 + wave + 4, gsel1, gout, wf0, wf1, af);
        finish(1, wave + 4, gout, wf0, wf1, af);
      }
    }
    __syncthreads();
    gs.mark();
    if (wave == 3) {
      // ---------------- Beam.advance (csrc/beam.hip beam_advance_wave_kernel) on the prefetched state ----------------
      const int b = c;
      if (st_done) {
        // frozen clip: keep the tables valid so the (ignored) rows keep reading defined memory
#pragma unroll
        for (int i = 0; i < RES_BMK; ++i)
          if (i < bm) {
            const int64_t o = (int64_t)(row0 + i) * stride;
            if (lane < t) cst_i(anc_new + o + lane, st_anc[i]);
            if (lane == 0) { cst_i(anc_new + o + t, row0 + i); cst_i(p.fed + o + t, p.eos); }
          }
      } else {
        // candidate pool, lane = i * bm + j: (value, flat index i * V + col); ended beams offer nothing
        // (Beam.py:52-54: the token at position t - 1 of the hypothesis in slot i, which lives at anc_old[i][t - 1],
        // a row of this clip); first step: row 0 only (Beam.py:55-56)
        const int n_src = (t == 1) ? 1 : bm;
        const int ci = lane / bm, cj = lane % bm;
        bool live = lane < n_src * bm;
        // tokens at position t - 1 of the clip's physical rows (lane t - 1 holds them), then per slot through its ancestor
        int ended_mask = 0;
        if (t > 1) {
#pragma unroll
          for (int i = 0; i < RES_BMK; ++i)
            if (i < bm) {
              const int prow = __shfl(st_anc[i], t - 1, 64) - row0;  // physical row (within the clip) of slot i's last token
              int tk = p.eos + 1;
#pragma unroll
              for (int j = 0; j < RES_BMK; ++j)
                if (j < bm && prow == j) tk = __shfl(st_tok[j], t - 1, 64);
              if (tk == p.eos) ended_mask |= 1 << i;
            }
        }
        float v = -INFINITY;
        int col = 0;
        if (live) {
          if ((ended_mask >> ci) & 1) live = false;
          v = s_cv[ci][cj];
          col = s_ci[ci][cj];
          if (t > 1) {
            float so = st_sc[0];
#pragma unroll
            for (int i = 1; i < RES_BMK; ++i)
              if (ci == i) so = st_sc[i];
            v = v + so;
          }
        }
        unsigned long long key = live ? key_of(v, (unsigned)(ci * p.V + col)) : 0ull;
        float sc[RES_BMK];
        int parent[RES_BMK], tok[RES_BMK];
#pragma unroll
        for (int k = 0; k < RES_BMK; ++k) {
          sc[k] = -1e20f; parent[k] = 0; tok[k] = p.eos;
          if (k < bm) {
            const unsigned long long best = wave_max_u64(key);
            if (best) {  // no candidate left: every beam has ended (possible once topk > beam_size)
              const unsigned flat = key_idx(best);
              sc[k] = key_val(best); parent[k] = (int)(flat / (unsigned)p.V); tok[k] = (int)(flat - (unsigned)parent[k] * (unsigned)p.V);
              if (key == best) key = 0ull;
            }
          }
        }
        // rewire ancestors (lane = position): slot i inherits its parent's row of the old table
        int anew[RES_BMK];
#pragma unroll
        for (int i = 0; i < RES_BMK; ++i) {
          anew[i] = 0;
          if (i < bm) {
            int a = st_anc[0];
#pragma unroll
            for (int j = 1; j < RES_BMK; ++j)
              if (parent[i] == j) a = st_anc[j];
            anew[i] = lane < t ? a : 0;
          }
        }
#pragma unroll
        for (int i = 0; i < RES_BMK; ++i)
          if (i < bm) {
            const int64_t dst = (int64_t)(row0 + i) * stride;
            if (lane < t) cst_i(anc_new + dst + lane, anew[i]);
            if (lane == 0) { cst_i(anc_new + dst + t, row0 + i); cst_i(p.fed + dst + t, tok[i]); cst_f(p.score + row0 + i, sc[i]); }
          }
        // finished hypotheses, in beam order, stop as soon as `need` are collected (Beam.py:72-77)
        int nf = st_nf;
        bool is_done = false;
        auto record = [&](int i_anew, int i_tok, float i_sc) {  // hypothesis of one beam: positions 1..t, one per lane
          if (nf < p.fin_cap) {
            const int64_t slot = (int64_t)b * p.fin_cap + nf;
            if (lane == 0) { cst_f(p.fscore + slot, i_sc); cst_i(p.flen + slot, t); }
            if (lane >= 1 && lane <= t) {
              // position `lane` of the hypothesis lives at physical row i_anew (a row of this clip: its token is in
              // the prefetched table); position t is the token just chosen
              int token = i_tok;
              if (lane < t) {
                token = st_tok[0];
#pragma unroll
                for (int j = 1; j < RES_BMK; ++j)
                  if (i_anew - row0 == j) token = st_tok[j];
              }
              cst_i(p.fhyp + slot * stride + (lane - 1), token);
            }
          }
          ++nf;
        };
        // topk > beam_size: a clip can run out of live beams before `need` hypotheses have ended; it then ends with
        // the hypotheses it has (csrc/beam.hip: the reference's -1e20 continuations are never reported)
        if (sc[0] <= -1e19f) is_done = true;
#pragma unroll
        for (int i = 0; i < RES_BMK; ++i)
          if (i < bm && !is_done && tok[i] == p.eos && sc[i] > -1e19f) {
            record(anew[i], tok[i], sc[i]);
            if (nf >= p.need) is_done = true;
          }
        if (!is_done && t >= p.T) {  // Beam.py:79-84
          is_done = true;
          if (nf == 0) {
#pragma unroll
            for (int i = 0; i < RES_BMK; ++i)
              if (i < bm) record(anew[i], tok[i], sc[i]);
          }
        }
        if (lane == 0) {
          cst_i(p.nfin + b, nf);
          if (is_done) {
            cst_i(p.done + b, 1);
            __hip_atomic_fetch_add(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    if (c + G < p.nclips) __syncthreads();  // (s_cv is written again)
  }
  gs.mark();
  gs.arrive(participant);
  return nprod;
}

// KCF = ff / 512.  Row tiles per workgroup and weight fetch: RQ (QKV), RD (the N = 512 products), RF (FFN dense1), RV
// (vocabulary); SM (<= 64 rows): QKV and FFN dense1 in 16-column K-split items, FFN dense2 over two workgroups per
// column tile (the forms of decode_resident.hip).
template <int KCF, int RQ, int RD, int RF, int RV, bool SM, bool KD>  // KD: the N = 512 products in K-split items
__global__ __launch_bounds__(256, 1) void decode_resident_beam_kernel(RArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
  GridSync gs{p.sync, (unsigned)p.ghost, -1, false, 0, 0, 0u};
  gs.fenced = p.fenced != 0;
  const int d = p.d;
  constexpr bool HF = KCF == 4 && SM;
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
      if (l == 0) RES_PHASE((gemm_phase<512, A_EMBEDB, E_QKV, SM, RQ>(p, gs, !first_waited, sA, L.qkv_w, L.qkv_b, 3 * d, nullptr, p.emb_g, p.emb_be, true, t, L.skv)));
      else RES_PHASE((gemm_phase<512, A_LN, E_QKV, SM, RQ>(p, gs, true, sA, L.qkv_w, L.qkv_b, 3 * d, p.y, p.L[l - 1].fg, p.L[l - 1].fbe, true, t, L.skv, y2)));
      RES_PHASE((p.T <= 32 ? attn_phase<true, 4, true>(p, gs, true, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0, p.anc[(t - 1) & 1])
                            : attn_phase<true, RES_MAXKB, true>(p, gs, true, L.skv, (int64_t)p.T * 2 * d, 1, t, p.fed, nullptr, 0, p.anc[(t - 1) & 1])));
      RES_PHASE((gemm_phase<512, A_BF16, E_RES, KD, RD>(p, gs, true, sA, L.o_w, L.o_b, d, p.ctx, nullptr, nullptr, false, t, nullptr)));
      const float* g = L.g;
      const float* be = L.be;
      for (int a = 0; a < L.n_att; ++a) {
        const RAttn& A = L.att[a];
        RES_PHASE((gemm_phase<512, A_LN, E_Q, KD, RD>(p, gs, true, sA, A.q_w, A.q_b, d, p.y, g, be, true, t, nullptr)));
        // more (row, head) pairs than waves: a wave per (clip, head) with the clip's keys fetched once for its beams
        if (p.R * p.H > 4 * (int)gridDim.x)
          RES_PHASE((A.nkeys <= 64 ? attn_shared_phase<8>(p, gs, true, A.kv, A.kv_bs, p.bm, A.nkeys, A.bias, A.bias_ld)
                                    : attn_shared_phase<RES_MAXKB>(p, gs, true, A.kv, A.kv_bs, p.bm, A.nkeys, A.bias, A.bias_ld)));
        else
          RES_PHASE((A.nkeys <= 64 ? attn_phase<false, 8>(p, gs, true, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, nullptr, A.bias, A.bias_ld)
                                    : attn_phase<false, RES_MAXKB>(p, gs, true, A.kv, A.kv_bs, A.rows_per_kv, A.nkeys, nullptr, A.bias, A.bias_ld)));
        RES_PHASE((gemm_phase<512, A_BF16, E_RES, KD, RD>(p, gs, true, sA, A.o_w, A.o_b, d, p.ctx, nullptr, nullptr, false, t, nullptr)));
        g = A.g; be = A.be;
      }
      RES_PHASE((gemm_phase<512, A_LN, E_ACT, SM, RF>(p, gs, true, sA, L.w1, L.b1, p.ff, p.y, g, be, true, t, nullptr)));
      if constexpr (KCF == 4) RES_PHASE((ffn2_phase<HF>(p, gs, sA, L.w2, L.b2)));
      else RES_PHASE((gemm_phase<512 * KCF, A_BF16, E_RES, true>(p, gs, true, sA, L.w2, L.b2, d, p.h, nullptr, nullptr, false, t, nullptr)));
    }
    const RLayer& LL = p.L[p.n_layers - 1];
    RES_PHASE((gemm_phase<512, A_LN, E_VOCABK, false, RV>(p, gs, true, sA, p.vocab, nullptr, p.V, p.y, LL.fg, LL.fbe, false, t, nullptr, y2, p.vcap)));
    RES_PHASE((beam_advance_phase(p, gs, t, smem)));
  }
#undef RES_PHASE
  if (gs.dead) {  // aborted (GridSync::wait): every clip's count of finished hypotheses = -1
    for (int b = blockIdx.x * 256 + threadIdx.x; b < p.nclips; b += gridDim.x * 256) cst_i(p.nfin + b, -1);
    if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = 0xffffffffu;
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) p.sync[2] = (unsigned)(ended ? t_run : p.steps);
}

constexpr int RES_BEAM_KERNELS = 8;
std::atomic<unsigned long long> g_resb_lds_done[RES_BEAM_KERNELS];
std::atomic<int> g_resb_ok[RES_BEAM_KERNELS];

// parts (vocabulary partials per row) of a grid: workgroups per row group that have a column item (PhaseMap)
inline int beam_parts(int grid, int RG, int CIV) {
  const int nper = ((grid & 7) == 0 && (grid >> 3) >= RG) ? 8 * ((grid >> 3) / RG) : grid / RG;
  return nper < CIV ? nper : CIV;
}

}  // namespace

extern "C" {

int64_t care_decode_resident_beam_scratch(int clips, int beam, int d, int ff, int V) {
  if (clips < 1 || beam < 1 || d < 1 || ff < 1 || V < 1) return CARE_EINVAL;
  const int64_t rows = (int64_t)clips * beam, R16 = (rows + 15) / 16 * 16;
  int64_t parts = (V + 63) / 64;
  if (parts > 64 * RES_NP) parts = 64 * RES_NP;
  // sync | xres, y, y2, q fp32 [R16, d] | ctx bf16 [R16, d] | h bf16 [R16, ff] | pmax, pidx, psum [R16, parts] |
  // gval, ggid [R16, parts, RES_BMK] | hn bf16 [R16, d]
  return RES_SYNC_BYTES + R16 * d * 4 * 4 + R16 * d * 2 + R16 * ff * 2 + R16 * parts * 12 + R16 * parts * RES_BMK * 8 + R16 * d * 2;
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
  // the advance phase keeps a hypothesis' positions one per lane (T + 1 <= 64), a clip's candidates one per lane
  // (beam^2 <= 64) and RES_BMK groups per row
  if (d != 512 || heads * 64 != d || (ff != 512 && ff != 1024 && ff != 2048) || T > 63 || V > 64 * 64 * RES_NP || beam > RES_BMK ||
      V < 4 * RES_BMK * 4)
    return CARE_ESHAPE;
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
  int64_t maxparts = (V + 63) / 64;
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
  p.hn = (bf16_t*)b;

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
  if (ff != 2048) cfg = 0;  // (one form for the narrow FFNs)
  const int RT = (int)(R16 / 16), CIV = (V + 63) / 64;
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
  int vcap = RG == 1 ? grid : (48 * RG < grid ? 48 * RG : grid);
  while (vcap > 8 && beam_parts(vcap, RG, CIV) > maxparts) vcap -= 8;
  p.vcap = vcap;
  p.parts = beam_parts(vcap, RG, CIV);
  if (p.parts > maxparts || p.parts > 64 * RES_NP || p.parts < 1) return CARE_ESHAPE;
  const int kmax = ff > d ? ff : d;
  int lds = 16 * (kmax + 8) * 2;
  const int rmax = rv > rq ? (rv > rf ? rv : rf) : (rq > rf ? rq : rf);
  if (rmax * 16 * (512 + 8) * 2 > lds) lds = rmax * 16 * (512 + 8) * 2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(256);
  int rc;
#define RESB_LAUNCH(KCF, RQ, RD, RF, RV, SM, KD, SLOT)                                                                      \
  do {                                                                                                                  \
    const void* kfn = (const void*)decode_resident_beam_kernel<KCF, RQ, RD, RF, RV, SM, KD>;                                \
    if ((rc = care_allow_dynamic_lds(kfn, lds, g_resb_lds_done[SLOT]))) return rc;                                       \
    if (!g_resb_ok[SLOT].load(std::memory_order_acquire)) {                                                             \
      if ((rc = res_check_residency(kfn, lds, grid, cus))) return rc;                                                   \
      g_resb_ok[SLOT].store(1, std::memory_order_release);                                                              \
    }                                                                                                                   \
    if ((e = hipMemsetAsync(p.sync, 0, RES_SYNC_BYTES, st)) != hipSuccess) return (int)e;                               \
    hipLaunchKernelGGL((decode_resident_beam_kernel<KCF, RQ, RD, RF, RV, SM, KD>), g, blk, lds, st, p);                     \
  } while (0)
  if (ff == 512) RESB_LAUNCH(1, 1, 1, 1, 1, true, true, 0);
  else if (ff == 1024) RESB_LAUNCH(2, 1, 1, 1, 1, true, true, 1);
  else if (cfg == 0) RESB_LAUNCH(4, 1, 1, 1, 1, true, true, 2);
  else if (cfg == 1 && rv == 2) RESB_LAUNCH(4, 1, 1, 1, 2, false, true, 3);
  else if (cfg == 1) RESB_LAUNCH(4, 1, 1, 1, 1, false, true, 4);
  else if (cfg == 3) RESB_LAUNCH(4, 2, 2, 2, 4, false, false, 6);
  else RESB_LAUNCH(4, 4, 2, 4, 4, false, false, 5);
#undef RESB_LAUNCH
  return care_launch_status();
}

}  // extern "C"
