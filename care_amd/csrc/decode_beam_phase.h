// decode_beam_phase.h - the beam-advance phase shared by the resident beam launch (decode_resident_beam.hip: one phase of
// the launch) and the chained beam step (decode_chain.hip: a kernel of its own).  See decode_resident_beam.hip.
#pragma once
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

// D = d_model.  D > 512 (the `median` / `large` architectures, whose vocabulary phase runs in K-split items): the candidates'
// logits are recomputed in the K-split form's order - per K quarter two accumulator chains, the quarters added as
// q0 + ((q1 + q2) + q3) - a quarter's fragments at a time.
template <int D = 512>
RES_PHASE_FN unsigned beam_advance_phase(const RArgs& p, GridSync& gs, int t, unsigned char* scratch_lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kg = lane >> 4;
  const int G = gridDim.x, bm = p.bm, stride = p.fed_stride;
  const unsigned nprod = (unsigned)(p.nclips < G ? p.nclips : G);
  const bool participant = (int)blockIdx.x < p.nclips;
  float (*s_cv)[RES_BMK] = reinterpret_cast<float (*)[RES_BMK]>(scratch_lds);
  int (*s_ci)[RES_BMK] = reinterpret_cast<int (*)[RES_BMK]>(scratch_lds + 8 * RES_BMK * 4);
  int* s_tok = reinterpret_cast<int*>(scratch_lds + 16 * RES_BMK * 4);  // the tokens just chosen, slot by slot (RES_BEAM_LDS bytes in all)
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
    const bf16_t* wide_w[2] = {nullptr, nullptr};  // D > 512: this lane's weight rows / the row's hidden state (issue -> finish)
    const bf16_t* wide_a = nullptr;
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
      if constexpr (D == 512) {
        load_w<16>(wf0, p.vocab + (int64_t)min(gl[0] * 4 + (l16 & 3), p.V - 1) * 512 + kg * 8);
        if (bm > 4) load_w<16>(wf1, p.vocab + (int64_t)min(gl[1] * 4 + (l16 & 3), p.V - 1) * 512 + kg * 8);
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] = cld_b8(p.hn + (int64_t)r * 512 + kg * 8 + q * 32);
      } else {  // (the fragments are fetched a K quarter at a time by `finish`)
        wide_w[0] = p.vocab + (int64_t)min(gl[0] * 4 + (l16 & 3), p.V - 1) * D + kg * 8;
        wide_w[1] = p.vocab + (int64_t)min(gl[1] * 4 + (l16 & 3), p.V - 1) * D + kg * 8;
        wide_a = p.hn + (int64_t)r * D + kg * 8;
      }
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
      if constexpr (D == 512) {
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
      } else {
        constexpr int QF = D / 128;  // fragments of a K quarter (what a wave of the K-split vocabulary phase multiplies)
        f32x4 part[2][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          bf16x8 a[QF], w0[QF], w1[QF];
#pragma unroll
          for (int i = 0; i < QF; ++i) {
            a[i] = cld_b8(wide_a + (qq * QF + i) * 32);
            w0[i] = *reinterpret_cast<const bf16x8*>(wide_w[0] + (qq * QF + i) * 32);
            if (bm > 4) w1[i] = *reinterpret_cast<const bf16x8*>(wide_w[1] + (qq * QF + i) * 32);
          }
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < QF; i += 2) {
            a0 = care_mfma_16x16x32_h16(w0[i], a[i], a0, 0, 0, 0);
            a1 = care_mfma_16x16x32_h16(w0[i + 1], a[i + 1], a1, 0, 0, 0);
            if (bm > 4) {
              b0 = care_mfma_16x16x32_h16(w1[i], a[i], b0, 0, 0, 0);
              b1 = care_mfma_16x16x32_h16(w1[i + 1], a[i + 1], b1, 0, 0, 0);
            }
          }
          part[0][qq] = a0 + a1;
          part[1][qq] = b0 + b1;
        }
        vt[0] = part[0][0] + ((part[0][1] + part[0][2]) + part[0][3]);
        vt[1] = bm > 4 ? part[1][0] + ((part[1][1] + part[1][2]) + part[1][3]) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
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
            if (lane == 0) { cst_i(anc_new + o + t, row0 + i); cst_i(p.fed + o + t, p.eos); s_tok[i] = p.eos; }
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
            if (lane == 0) { cst_i(anc_new + dst + t, row0 + i); cst_i(p.fed + dst + t, tok[i]); cst_f(p.score + row0 + i, sc[i]); s_tok[i] = tok[i]; }
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
            // (the returned value is waited for: the count is at L2 BEFORE this workgroup's arrival is - every workgroup that reads it
            // behind the hand-off must come to the same `all rows ended` verdict; a fire-and-forget add may land after the arrival)
            const unsigned ended_before = __hip_atomic_fetch_add(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(ended_before));
          }
        }
      }
    }
    if (t < p.T) {
      // the input rows of step t + 1, once per row: embedding + LayerNorm of the tokens just chosen (embed_ln_store_row), four
      // rows per wave side by side; the rows of a clip that is done embed the EOS its table carries - defined bytes, ignored
      __syncthreads();
      if (wave * 4 < bm) {
        const int i = wave * 4 + (lane >> 4);
        embed_ln_store_row<D>(p, row0 + i, i < bm, s_tok[i < bm ? i : 0], t);
      }
    }
    if (c + G < p.nclips) __syncthreads();  // (s_cv / s_tok are written again)
  }
  gs.mark();
  gs.arrive(participant);
  return nprod;
}

constexpr int RES_BEAM_LDS = 16 * RES_BMK * 4 + 8 * 4;  // s_cv + s_ci + s_tok of beam_advance_phase

// Before step 1: the state of every clip's rows as the host-side initialisation of engine.beam leaves it - token table
// [BOS, EOS ...], both ancestor tables = the row itself, scores 0, the clip's flags and (zeroed) finished lists - and the
// input rows of step 1 (BOS at position 0 + the clip's guidance vector, normalised).  One workgroup per clip, like the advance.
template <int D = 512>
RES_PHASE_FN unsigned beam_init_phase(const RArgs& p, GridSync& gs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = gridDim.x, bm = p.bm, stride = p.fed_stride;
  const unsigned nprod = (unsigned)(p.nclips < G ? p.nclips : G);
  const bool participant = (int)blockIdx.x < p.nclips;
  if (gs.dead) return nprod;
  for (int c = blockIdx.x; c < p.nclips; c += G) {
    const int row0 = c * bm;
    for (int i = wave; i < bm; i += 4) {
      const int r = row0 + i;
      for (int col = lane; col <= p.T; col += 64) {
        cst_i(p.fed + (int64_t)r * stride + col, col == 0 ? p.bos : p.eos);
        cst_i(p.anc[0] + (int64_t)r * stride + col, r);
        cst_i(p.anc[1] + (int64_t)r * stride + col, r);
      }
      if (lane == 0) cst_f(p.score + r, 0.f);
    }
    if (threadIdx.x == 0) { cst_i(p.done + c, 0); cst_i(p.nfin + c, 0); }
    for (int k = threadIdx.x; k < p.fin_cap; k += 256) { cst_f(p.fscore + (int64_t)c * p.fin_cap + k, 0.f); cst_i(p.flen + (int64_t)c * p.fin_cap + k, 0); }
    for (int k = threadIdx.x; k < p.fin_cap * stride; k += 256) cst_i(p.fhyp + (int64_t)c * p.fin_cap * stride + k, 0);
    if (wave * 4 < bm) {
      const int i = wave * 4 + (lane >> 4);
      embed_ln_store_row<D>(p, row0 + i, i < bm, p.bos, 0);
    }
  }
  gs.arrive(participant);
  return nprod;
}

}  // namespace
