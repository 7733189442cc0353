// beam.hip - beam-search step on the device (beam_size > 1).
//
// Replaces, per decode step, Translator_ARFormer.predict_word's log_softmax
// (models/Translator.py:127) and Beam.advance (misc/Decoding/Beam.py:45-85) for every clip,
// without the per-step D2H sync / .item() loop of the reference.  Two kernels:
//
//   care_beam_select : per row of logits [rows, V]: log-sum-exp and the beam_size best
//                      columns as log-probabilities (value desc, index asc).  The global top
//                      beam_size of a clip's beam_size x V candidates always lies inside the
//                      union of the per-row top beam_size, so nothing else leaves this kernel.
//   care_beam_advance: one thread per clip runs the beam state machine on those
//                      beam_size^2 candidates, including every quirk the reference has
//                      (first step looks at row 0 only; a beam that ended with EOS offers no
//                      continuation; hypotheses are collected in beam order until `need` are
//                      finished; forced finish at max_steps), and rewires the ancestor table
//                      that the incremental self-attention K/V cache is addressed through.
//
// Ancestor table: anc[row][j] = PHYSICAL row that holds position j (token, self-attn K/V) of the
// hypothesis currently living in beam slot `row`.  Re-ordering beams is a copy of <= 30 ints per
// row instead of a gather over the whole K/V cache.
#include "care_common.h"

namespace {

constexpr int MAXBM = 8;

__global__ __launch_bounds__(256) void beam_select_kernel(const float* logits, int64_t ldl, int V, int bm,
                                                          float* cand_val, int32_t* cand_idx, int rows) {
  __shared__ float sval[256 * MAXBM];
  __shared__ int sidx[256 * MAXBM];
  __shared__ float sred[8];
  __shared__ int sredi[8];
  __shared__ float s_bcast[2];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)r * ldl;

  // pass 1: row max + thread-local top-bm (sorted: value desc, index asc)
  float tv[MAXBM];
  int ti[MAXBM];
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) {
    const float v = x[c];
    mx = fmaxf(mx, v);
    if (v > tv[bm - 1]) {  // strictly greater: equal values keep the earlier (lower) index
      float cv = v; int ci = c;
#pragma unroll
      for (int j = 0; j < MAXBM; ++j) {
        if (j < bm && (cv > tv[j])) {
          const float ov = tv[j]; const int oi = ti[j];
          tv[j] = cv; ti[j] = ci; cv = ov; ci = oi;
        }
      }
    }
  }
  mx = care_wave_max(mx);
  if (lane == 0) sred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  // pass 2: sum exp(x - max)
  float se = 0.f;
  for (int c = tid; c < V; c += 256) se += expf(x[c] - mx);
  se = care_wave_sum(se);
  if (lane == 0) sred[wave] = se;
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) { sval[tid * MAXBM + j] = tv[j]; sidx[tid * MAXBM + j] = ti[j]; }
  __syncthreads();
  const float logsum = logf((sred[0] + sred[1]) + (sred[2] + sred[3]));
  __syncthreads();

  // bm rounds of block-wide arg-best over the 256 thread-local list heads
  int head = 0;  // next unconsumed entry of this thread's sorted list
  for (int k = 0; k < bm; ++k) {
    float v = head < bm ? sval[tid * MAXBM + head] : -INFINITY;
    int id = head < bm ? sidx[tid * MAXBM + head] : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; }
    }
    if (lane == 0) { sred[wave] = v; sredi[wave] = id; }
    __syncthreads();
    if (tid == 0) {
      float bv = sred[0]; int bi = sredi[0];
      for (int w = 1; w < 4; ++w)
        if (sred[w] > bv || (sred[w] == bv && sredi[w] < bi)) { bv = sred[w]; bi = sredi[w]; }
      s_bcast[0] = bv; sredi[4] = bi;
      cand_val[(int64_t)r * bm + k] = (bv - mx) - logsum;  // log_softmax = (x - max) - log(sum)
      cand_idx[(int64_t)r * bm + k] = bi == 0x7fffffff ? 0 : bi;  // fewer than bm finite logits: (-inf, 0)
    }
    __syncthreads();
    const int win = sredi[4];
    if (head < bm && sidx[tid * MAXBM + head] == win) ++head;  // column indices are unique
    __syncthreads();
  }
}

// Model ensembling (models/Translator.py:112-133): the step's word log-probabilities are the members' log_softmax rows averaged
// equally, `torch.stack(word_probs).mean(0)`, and Beam.advance takes its top-k from that average.  One workgroup per row: every
// member's (max, log sum exp) first, then one walk over the columns forming avg[c] = (sum_m ((x_m[c] - max_m) - logsum_m)) / n -
// torch's log_softmax and mean, member by member in list order - into thread-local top-bm lists, then beam_select_kernel's
// selection rounds.  The averaged [rows, V] array never exists; an average of log-probabilities is NOT renormalised (the
// reference does not), which is why care_beam_select (it subtracts the row's own log-sum-exp) cannot be reused on it.
constexpr int MAX_MODELS = 8;
struct EnsArgs { const float* x[MAX_MODELS]; int n; };

__global__ __launch_bounds__(256) void ensemble_select_kernel(EnsArgs a, int64_t ldl, int V, int bm, float* cand_val,
                                                              int32_t* cand_idx, int rows) {
  __shared__ float sval[256 * MAXBM];
  __shared__ int sidx[256 * MAXBM];
  __shared__ float sred[8];
  __shared__ int sredi[8];
  __shared__ float s_mx[MAX_MODELS], s_ls[MAX_MODELS];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int m = 0; m < a.n; ++m) {
    const float* x = a.x[m] + (int64_t)r * ldl;
    float mx = -INFINITY;
    for (int c = tid; c < V; c += 256) mx = fmaxf(mx, x[c]);
    mx = care_wave_max(mx);
    if (lane == 0) sred[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float se = 0.f;
    for (int c = tid; c < V; c += 256) se += expf(x[c] - mx);
    se = care_wave_sum(se);
    if (lane == 0) sred[wave] = se;
    __syncthreads();
    if (tid == 0) { s_mx[m] = mx; s_ls[m] = logf((sred[0] + sred[1]) + (sred[2] + sred[3])); }
    __syncthreads();
  }
  float tv[MAXBM];
  int ti[MAXBM];
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  const float fn = (float)a.n;
  for (int c = tid; c < V; c += 256) {
    float v = 0.f;
    for (int m = 0; m < a.n; ++m) v += (a.x[m][(int64_t)r * ldl + c] - s_mx[m]) - s_ls[m];
    v = v / fn;   // (torch's mean: the sum divided by the count)
    if (v > tv[bm - 1]) {  // strictly greater: equal values keep the earlier (lower) index
      float cv = v; int ci = c;
#pragma unroll
      for (int j = 0; j < MAXBM; ++j) {
        if (j < bm && (cv > tv[j])) {
          const float ov = tv[j]; const int oi = ti[j];
          tv[j] = cv; ti[j] = ci; cv = ov; ci = oi;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) { sval[tid * MAXBM + j] = tv[j]; sidx[tid * MAXBM + j] = ti[j]; }
  __syncthreads();
  int head = 0;  // next unconsumed entry of this thread's sorted list
  for (int k = 0; k < bm; ++k) {
    float v = head < bm ? sval[tid * MAXBM + head] : -INFINITY;
    int id = head < bm ? sidx[tid * MAXBM + head] : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; }
    }
    if (lane == 0) { sred[wave] = v; sredi[wave] = id; }
    __syncthreads();
    if (tid == 0) {
      float bv = sred[0]; int bi = sredi[0];
      for (int w = 1; w < 4; ++w)
        if (sred[w] > bv || (sred[w] == bv && sredi[w] < bi)) { bv = sred[w]; bi = sredi[w]; }
      sredi[4] = bi;
      cand_val[(int64_t)r * bm + k] = bv;
      cand_idx[(int64_t)r * bm + k] = bi == 0x7fffffff ? 0 : bi;
    }
    __syncthreads();
    const int win = sredi[4];
    if (head < bm && sidx[tid * MAXBM + head] == win) ++head;  // column indices are unique
    __syncthreads();
  }
}

// One WAVE per row, one pass over HBM.  The row is taken in chunks of 64 logits per lane held in
// registers (16 x 16-byte loads issued together: 16 KiB of the row in flight per wave):
//   A. branch-free per logit: online (max, sum-exp) with a single exp, and the lane's chunk maximum;
//   B. a threshold tau = max(k-th entry of the wave's top-bm list so far, k-th largest of the 64 lane
//      maxima of this chunk) - both are lower bounds of the final k-th best, so nothing >= the final
//      k-th best is dropped - and every logit >= tau is appended to a per-wave LDS list (a divergent
//      branch that is almost never taken: ~10 candidates per chunk);
//   C. ONE rolled copy of the exact selection: the listed candidates, one per lane, enter the wave's
//      sorted top-bm list (replicated in every lane) by best-first rounds of wave arg-best
//      (value desc, index asc).
// If a chunk lists more candidates than the LDS list holds (thousands of equal logits), the chunk is
// re-scanned from memory 64 logits at a time through the same rounds.
// Measured for 20480 x 10547 (a beam-5 step of 4096 clips): 251 us = 3.4 TB/s.  History: 256-thread
// two-pass kernel (above, still the fallback for rows that are not 16-byte aligned) 735 us = 28% of
// a beam-5 pass; per-lane top-bm lists 406 us (the divergent insertion path runs in nearly every
// iteration); the rounds inlined per logit: 32 K instructions, slower still (instruction fetch);
// this structure with xor-shuffle reductions for the threshold and arg-best rounds for the listed
// candidates 350-420 us - each such reduction is six DEPENDENT ds_bpermute round trips, so the
// threshold now uses a DPP maximum (care_wave_max_dpp), only in the first chunk, and the listed
// candidates are inserted from LDS broadcast reads.  Ablation: streaming skeleton ~200 us,
// statistics ~45 us with the fast exp (accurate expf: +57 us; the fp32 beam fixtures do not move).
constexpr int BS_CAP = 192;  // candidates per wave and chunk
#ifndef CARE_BS_DBG
#define CARE_BS_DBG 0  // ablation builds (tools/beam_select_probe.py): 1 no statistics, 2 no candidate list / selection, 4 accurate expf
#endif
#if CARE_BS_DBG & 4
#define BS_EXP expf
#else
#define BS_EXP __expf
#endif

// SPLIT = 4 (few rows: a launch of rows / 4 workgroups leaves most CUs idle and every wave walks its 42-KB row in
// three dependent chunks): the four waves of a workgroup share ONE row - a quarter of the 16-byte groups each - and
// wave 0 merges the four sorted lists and (max, sum exp) pairs through LDS.  The same bm columns in the same order;
// the log-sum-exp is added in another order (log-probabilities differ in the last bits).
template <int SPLIT>
__global__ __launch_bounds__(256) void beam_select_wave_kernel(const float* logits, int64_t ldl, int V, int bm,
                                                               float* cand_val, int32_t* cand_idx, int rows) {
  __shared__ float lval[4][BS_CAP];
  __shared__ int lidx[4][BS_CAP];
  __shared__ int lcnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = SPLIT == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (r >= rows) return;  // wave-uniform (SPLIT = 1: no block-wide barrier below; SPLIT = 4: workgroup-uniform)
  const float* x = logits + (int64_t)r * ldl;
  float tv[MAXBM];  // the wave's top-bm so far, identical in all lanes, sorted (value desc, index asc)
  int ti[MAXBM];
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  float m = -1e30f, s = 0.f;  // finite sentinel: a lane that saw nothing merges as exp(-1e30 - max) = 0

  auto insert = [&](float cv, int ci) {  // wave-uniform (cv, ci) into the sorted list
#pragma unroll
    for (int j = 0; j < MAXBM; ++j) {
      if (j < bm && (cv > tv[j] || (cv == tv[j] && ci < ti[j]))) {
        const float ov = tv[j]; const int oi = ti[j];
        tv[j] = cv; ti[j] = ci; cv = ov; ci = oi;
      }
    }
  };
  // best-first rounds: every lane may hold one candidate (v, c); c = INT_MAX means none
  auto rounds = [&](float v, int c) {
    bool cand = c != 0x7fffffff && v > -INFINITY && (v > tv[bm - 1] || (v == tv[bm - 1] && c < ti[bm - 1]));
    while (__any(cand)) {
      float bv = cand ? v : -INFINITY;
      int bc = cand ? c : 0x7fffffff;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oc = __shfl_xor(bc, o, 64);
        if (oc != 0x7fffffff && (bc == 0x7fffffff || ov > bv || (ov == bv && oc < bc))) { bv = ov; bc = oc; }
      }
      float cv = bv;
      int ci = bc;
#pragma unroll
      for (int j = 0; j < MAXBM; ++j) {
        if (j < bm && (cv > tv[j] || (cv == tv[j] && ci < ti[j]))) {
          const float ov = tv[j]; const int oi = ti[j];
          tv[j] = cv; ti[j] = ci; cv = ov; ci = oi;
        }
      }
      cand = cand && c != bc && (v > tv[bm - 1] || (v == tv[bm - 1] && c < ti[bm - 1]));
    }
  };

  const int nv4 = V >> 2;  // whole 16-byte groups; the V % 4 tail is handled at the end
  const int per = SPLIT == 1 ? nv4 : (nv4 + SPLIT - 1) / SPLIT;  // this wave's groups: [g0, g1)
  const int g0 = SPLIT == 1 ? 0 : min(nv4, wave * per), g1 = SPLIT == 1 ? nv4 : min(nv4, g0 + per);
  for (int base = g0; base < g1; base += 1024) {  // 16 groups per lane: group (base + u*64 + lane)
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (int64_t)min(base + u * 64 + lane, nv4 - 1) * 4);
    // ---- A: statistics
    float qmax[4];  // maximum of each quarter of the chunk (16 logits per lane)
#pragma unroll
    for (int q = 0; q < 4; ++q) qmax[q] = -INFINITY;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (base + u * 64 + lane >= g1) v[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e0 = v[u][j];
        qmax[u >> 2] = fmaxf(qmax[u >> 2], e0);
        if (CARE_BS_DBG & 1) continue;
        if constexpr (SPLIT == 1) {  // online: one exp per logit, a serial chain through (m, s) - hidden by the other waves
          const bool up = e0 > m;
          const float mn = up ? e0 : m;
          const float e = BS_EXP((up ? m : e0) - mn);  // rescale factor if e0 is the new max, else the new term
          s = up ? fmaf(s, e, 1.0f) : s + e;
          m = mn;
        }
      }
    }
    const float lmax = fmaxf(fmaxf(qmax[0], qmax[1]), fmaxf(qmax[2], qmax[3]));
    if constexpr (SPLIT > 1) {  // few rows, one wave per SIMD: the lane's maximum first, then 64 INDEPENDENT exps
      if (!(CARE_BS_DBG & 1) && lmax > -INFINITY) {
        float sc = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) sc += BS_EXP(v[u][j] - lmax);
        const float mn = fmaxf(m, lmax);
        s = s * BS_EXP(m - mn) + sc * BS_EXP(lmax - mn);
        m = mn;
      }
    }
    // ---- B: threshold and candidate list
    float tau = tv[bm - 1];
    if (base == g0) {  // first chunk: the list is empty; k-th largest lane maximum (ties only make it more conservative)
      float y = lmax, kth = -INFINITY;
      for (int k = 0; k < bm; ++k) {
        kth = care_wave_max_dpp(y);
        if (y == kth) y = -INFINITY;
      }
      tau = kth;
    }
    if (lane == 0) lcnt[wave] = 0;
    if (CARE_BS_DBG & 2) { m = fmaxf(m, tau); continue; }
    // one wave-uniform test per quarter (a skipped per-logit branch is a TAKEN branch: 192 of them per
    // row cost more than the statistics); only a quarter that holds a candidate in some lane is scanned
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (__builtin_expect(__any(qmax[q] >= tau && qmax[q] > -INFINITY), 0)) {
#pragma unroll
        for (int u = q * 4; u < q * 4 + 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (__builtin_expect(v[u][j] >= tau && v[u][j] > -INFINITY, 0)) {
              const int pos = atomicAdd(&lcnt[wave], 1);
              if (pos < BS_CAP) { lval[wave][pos] = v[u][j]; lidx[wave][pos] = (base + u * 64 + lane) * 4 + j; }
            }
      }
    const int n = lcnt[wave];  // same wave: LDS operations complete in order
    // ---- C: selection.  The listed candidates enter the list one after the other, read by every
    // lane from the same LDS address (broadcast) - no cross-lane reduction (an arg-best round is six
    // dependent ds_bpermute round trips; ~20 of them per row cost more than the statistics).
    if (n <= BS_CAP) {
      for (int i = 0; i < n; ++i) insert(lval[wave][i], lidx[wave][i]);
    } else {  // list overflow: the chunk again, from memory
      const int c_end = min((base + 1024) * 4, g1 * 4);
      for (int c0 = base * 4; c0 < c_end; c0 += 64) {
        const int c = c0 + lane;
        rounds(c < c_end ? x[c < c_end ? c : 0] : 0.f, c < c_end ? c : 0x7fffffff);
      }
    }
  }
  if (SPLIT == 1 || wave == SPLIT - 1) {  // the V % 4 tail logits
    const int c = nv4 * 4 + lane;
    const bool in = c < V;
    const float e0 = in ? x[in ? c : 0] : -INFINITY;
    const bool up = e0 > m;
    const float mn = up ? e0 : m;
    const float e = BS_EXP((up ? m : e0) - mn);
    s = up ? fmaf(s, e, 1.0f) : s + e;
    m = mn;
    if (lane == 0) lcnt[wave] = 0;
    if (in && e0 > -INFINITY && (e0 > tv[bm - 1] || (e0 == tv[bm - 1] && c < ti[bm - 1]))) {  // -inf is never a candidate
      const int pos = atomicAdd(&lcnt[wave], 1);  // at most 3 entries
      lval[wave][pos] = e0; lidx[wave][pos] = c;
    }
    const int n = lcnt[wave];
    for (int i = 0; i < n; ++i) insert(lval[wave][i], lidx[wave][i]);
  }

  float mx = care_wave_max(m);
  float ssum = care_wave_sum(s * BS_EXP(m - mx));
  if constexpr (SPLIT > 1) {
    __shared__ float pv[SPLIT][MAXBM], pmx[SPLIT], psm[SPLIT];
    __shared__ int pi[SPLIT][MAXBM];
    if (lane == 0) {
      pmx[wave] = mx; psm[wave] = ssum;
#pragma unroll
      for (int k = 0; k < MAXBM; ++k) { pv[wave][k] = tv[k]; pi[wave][k] = ti[k]; }
    }
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < SPLIT; ++w) {
      for (int k = 0; k < bm; ++k)
        if (pi[w][k] != 0x7fffffff) insert(pv[w][k], pi[w][k]);
      const float om = pmx[w], nm = fmaxf(mx, om);
      ssum = ssum * BS_EXP(mx - nm) + psm[w] * BS_EXP(om - nm);
      mx = nm;
    }
  }
  const float logsum = logf(ssum);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < MAXBM; ++k)
      if (k < bm) {
        cand_val[(int64_t)r * bm + k] = (tv[k] - mx) - logsum;  // log_softmax = (x - max) - log(sum)
        cand_idx[(int64_t)r * bm + k] = ti[k] == 0x7fffffff ? 0 : ti[k];  // fewer than bm finite logits: (-inf, 0)
      }
  }
}

// One WAVE per clip: the bm x bm candidates one per lane (bm rounds of wave-wide arg-best: value
// desc, flat index asc - the flattened topk order of Beam.py:60), the ancestor rows copied one
// position per lane.  (A one-thread-per-clip version ran ~200 dependent memory operations in series:
// 195 us per step for 4096 clips, 9% of a beam-5 pass.)  The ancestor table must fit a wave:
// stride = max_len <= 64.
__global__ __launch_bounds__(256) void beam_advance_wave_kernel(
    const float* cand_val, const int32_t* cand_idx, float* scores, int bm, int32_t* tokphys, const int32_t* anc_old,
    int32_t* anc_new, int32_t* done, int32_t* n_fin, float* fin_score, int32_t* fin_len, int32_t* fin_hyp, int fin_cap,
    int t, int max_steps, int need, int eos_id, int V, int stride, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // wave-uniform
  const int row0 = b * bm;

  if (done[b]) {
    // frozen clip: keep the tables valid so the (ignored) rows keep reading defined memory
    for (int i = 0; i < bm; ++i) {
      const int64_t o = (int64_t)(row0 + i) * stride;
      if (lane < t) anc_new[o + lane] = anc_old[o + lane];
      if (lane == 0) { anc_new[o + t] = row0 + i; tokphys[o + t] = eos_id; }
    }
    return;
  }

  // --- candidate pool, lane c = i * bm + j: (value, flat index i*V + col); ended beams offer nothing
  // (Beam.py:52-54); first step: row 0 only (Beam.py:55-56)
  const int n_src = (t == 1) ? 1 : bm;
  const int ci = lane / bm, cj = lane % bm;
  bool live = lane < n_src * bm;
  if (live && t > 1) {
    const int prow = anc_old[(int64_t)(row0 + ci) * stride + (t - 1)];
    if (tokphys[(int64_t)prow * stride + (t - 1)] == eos_id) live = false;
  }
  float v = -INFINITY;
  int col = 0;
  long flat = 0x7fffffffffffffffL;
  if (live) {
    v = cand_val[(int64_t)(row0 + ci) * bm + cj];
    col = cand_idx[(int64_t)(row0 + ci) * bm + cj];
    if (t > 1) v = v + scores[row0 + ci];
    flat = (long)ci * V + col;
  }
  float sc[MAXBM];
  int parent[MAXBM], tok[MAXBM];
#pragma unroll
  for (int k = 0; k < MAXBM; ++k) {
    sc[k] = -1e20f; parent[k] = 0; tok[k] = eos_id;
    if (k < bm) {
      float bv = live ? v : -INFINITY;
      long bf = live ? flat : 0x7fffffffffffffffL;
      int bl = live ? lane : -1;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int olo = __shfl_xor((int)(bf & 0xffffffffL), o, 64), ohi = __shfl_xor((int)(bf >> 32), o, 64);
        const int ol = __shfl_xor(bl, o, 64);
        const long of = ((long)ohi << 32) | (unsigned int)olo;
        // a lane without a candidate (-1) never wins; among candidates: value desc, flat index asc
        if (ol >= 0 && (bl < 0 || ov > bv || (ov == bv && of < bf))) { bv = ov; bf = of; bl = ol; }
      }
      if (bl >= 0) {  // no candidate left: every beam has ended (possible once topk > beam_size), see below
        sc[k] = bv; parent[k] = bl / bm; tok[k] = __shfl(col, bl, 64);
        if (lane == bl) live = false;
      }
    }
  }

  // --- rewire ancestors (lane = position), record tokens and scores
  int anew[MAXBM];  // anc_new[row0 + i][lane] for lane < t
#pragma unroll
  for (int i = 0; i < MAXBM; ++i) {
    anew[i] = 0;
    if (i < bm) {
      const int64_t dst = (int64_t)(row0 + i) * stride, src = (int64_t)(row0 + parent[i]) * stride;
      if (lane < t) { anew[i] = anc_old[src + lane]; anc_new[dst + lane] = anew[i]; }
      if (lane == 0) { anc_new[dst + t] = row0 + i; tokphys[dst + t] = tok[i]; scores[row0 + i] = sc[i]; }
    }
  }

  // --- finished hypotheses, in beam order, stop as soon as `need` are collected (Beam.py:72-77)
  int nf = n_fin[b];
  bool is_done = false;
  auto record = [&](int i_anew, int i_tok, float i_sc) {  // hypothesis of one beam: positions 1..t, one per lane
    if (nf < fin_cap) {
      const int64_t slot = (int64_t)b * fin_cap + nf;
      if (lane == 0) { fin_score[slot] = i_sc; fin_len[slot] = t; }
      if (lane >= 1 && lane <= t) {
        // position t is the token just chosen (not read back from memory this wave has just written)
        const int token = lane < t ? tokphys[(int64_t)i_anew * stride + lane] : i_tok;
        fin_hyp[slot * stride + (lane - 1)] = token;
      }
    }
    ++nf;
  };
  // topk > beam_size (need > bm): a clip can run out of live beams before `need` hypotheses have ended.
  // The reference then keeps extending the ended beams from -1e20 rows, i.e. from a topk over exact
  // ties whose order torch leaves unspecified (parity unpinned); here the clip simply ends with the
  // hypotheses it has - such -1e20 continuations are never reported.
  if (sc[0] <= -1e19f) is_done = true;
#pragma unroll
  for (int i = 0; i < MAXBM; ++i)
    if (i < bm && !is_done && tok[i] == eos_id && sc[i] > -1e19f) {
      record(anew[i], tok[i], sc[i]);
      if (nf >= need) is_done = true;
    }
  if (!is_done && t >= max_steps) {  // Beam.py:79-84
    is_done = true;
    if (nf == 0) {
#pragma unroll
      for (int i = 0; i < MAXBM; ++i)
        if (i < bm) record(anew[i], tok[i], sc[i]);
    }
  }
  if (lane == 0) {
    n_fin[b] = nf;
    if (is_done) done[b] = 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused beam selection (bf16 mode): no [rows, V] logits in HBM.
//   pass 1  care_gemm_argmax_bf16_min : the fused-statistics vocabulary GEMM with >= 8 column ranges
//           per row -> per range (max, argmax, sum-exp);
//   care_beam_threshold               : tau[row] = the bm-th largest of the row's range maxima - bm
//           distinct logits are >= tau, so tau is a lower bound of the row's bm-th best - and the
//           candidate counter is reset;
//   pass 2  care_gemm_collect_bf16    : the same GEMM, epilogue = append every logit >= tau[row] to the
//           row's candidate list (about 10 per row);
//   care_beam_pick                    : the top bm of the list (value desc, column asc) as log-probs.
// A row whose list overflows (a plateau of equal logits at the threshold) is recomputed exactly: the
// pick kernel evaluates its V dot products itself.
__global__ __launch_bounds__(256) void beam_threshold_kernel(const float* pmax, int parts, int bm, float* thr,
                                                             int32_t* cnt, int rows) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float tv[MAXBM];
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) tv[j] = -INFINITY;
  for (int c = 0; c < parts; ++c) {
    float cv = pmax[(int64_t)r * parts + c];
#pragma unroll
    for (int j = 0; j < MAXBM; ++j)
      if (j < bm && cv > tv[j]) { const float o = tv[j]; tv[j] = cv; cv = o; }
  }
  float t = tv[0];
#pragma unroll
  for (int j = 1; j < MAXBM; ++j)
    if (j < bm) t = tv[j];
  thr[r] = t;  // -inf when fewer than bm ranges hold a finite maximum: everything is a candidate -> exact path
  cnt[r] = 0;
}

template <typename AT>
__global__ __launch_bounds__(256) void beam_pick_kernel(const float* pmax, const float* psum, int parts, const int32_t* cnt,
                                                        const float* cval, const int32_t* cidx, int cap, int bm,
                                                        const AT* A, int64_t lda, const bf16_t* W, int V, int K,
                                                        float* cand_val, int32_t* cand_idx, int rows) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  // log-sum-exp of the row from the pass-1 partials
  float m = -INFINITY;
  for (int c = lane; c < parts; c += 64) m = fmaxf(m, pmax[(int64_t)r * parts + c]);
  const float mx = care_wave_max(m);
  float se = 0.f;
  for (int c = lane; c < parts; c += 64) se += psum[(int64_t)r * parts + c] * expf(pmax[(int64_t)r * parts + c] - mx);
  const float logsum = logf(care_wave_sum(se));

  float tv[MAXBM];  // per-LANE sorted top-bm of the candidates this lane holds
  int ti[MAXBM];
#pragma unroll
  for (int j = 0; j < MAXBM; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  auto take = [&](float v, int c) {
    float cv = v; int ci = c;
#pragma unroll
    for (int j = 0; j < MAXBM; ++j)
      if (j < bm && (cv > tv[j] || (cv == tv[j] && ci < ti[j]))) {
        const float ov = tv[j]; const int oi = ti[j];
        tv[j] = cv; ti[j] = ci; cv = ov; ci = oi;
      }
  };
  const int n = cnt[r];
  if (n <= cap) {
    for (int i = lane; i < n; i += 64) take(cval[(int64_t)r * cap + i], cidx[(int64_t)r * cap + i]);
  } else {
    // overflow: the exact answer from the row's logits, recomputed here (fp32 accumulation of the
    // same bf16 products; only rows with thousands of tied logits come this way)
    const AT* a = A + (int64_t)r * lda;
    for (int c = lane; c < V; c += 64) {
      const bf16_t* w = W + (int64_t)c * K;
      float d = 0.f;
      for (int k = 0; k < K; k += 8) {
        float av[8], wv[8];
        care_load8(a + k, av);
        care_load8(w + k, wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) d = fmaf((float)(bf16_t)av[i], wv[i], d);
      }
      take(d, c);
    }
  }
  // merge the lanes' lists: bm rounds of wave-wide arg-best over the heads
  for (int k = 0; k < bm; ++k) {
    float v = tv[0];
    int id = ti[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; }
    }
    if (lane == 0) {
      cand_val[(int64_t)r * bm + k] = (v - mx) - logsum;
      cand_idx[(int64_t)r * bm + k] = id == 0x7fffffff ? 0 : id;
    }
    if (ti[0] == id && id != 0x7fffffff) {  // column indices are unique: exactly one lane pops its head
#pragma unroll
      for (int j = 0; j + 1 < MAXBM; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
      tv[MAXBM - 1] = -INFINITY; ti[MAXBM - 1] = 0x7fffffff;
    }
  }
}

}  // namespace

extern "C" int care_beam_select(const float* logits, int64_t ldl, int V, int bm, float* cand_val, int32_t* cand_idx,
                                int rows, int waves_per_row, void* stream) {
  if (!logits || !cand_val || !cand_idx || rows <= 0 || V <= 0) return CARE_EINVAL;
  if (bm <= 0 || bm > MAXBM || bm > V || (waves_per_row != 1 && waves_per_row != 4)) return CARE_ESHAPE;
  if ((ldl % 4) == 0 && care_aligned16(logits) && waves_per_row == 4)
    hipLaunchKernelGGL(beam_select_wave_kernel<4>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, bm,
                       cand_val, cand_idx, rows);
  else if ((ldl % 4) == 0 && care_aligned16(logits))
    hipLaunchKernelGGL(beam_select_wave_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                       V, bm, cand_val, cand_idx, rows);
  else
    hipLaunchKernelGGL(beam_select_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, bm, cand_val,
                       cand_idx, rows);
  return care_launch_status();
}

extern "C" int care_ensemble_select(const float* const* logits, int n_models, int64_t ldl, int V, int bm, float* cand_val,
                                    int32_t* cand_idx, int rows, void* stream) {
  if (!logits || !cand_val || !cand_idx || rows <= 0 || V <= 0 || n_models <= 0 || ldl < V) return CARE_EINVAL;
  if (bm <= 0 || bm > MAXBM || bm > V || n_models > MAX_MODELS) return CARE_ESHAPE;
  EnsArgs a{};
  a.n = n_models;
  for (int m = 0; m < n_models; ++m) {
    if (!logits[m]) return CARE_EINVAL;
    a.x[m] = logits[m];
  }
  hipLaunchKernelGGL(ensemble_select_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, ldl, V, bm, cand_val, cand_idx, rows);
  return care_launch_status();
}

extern "C" int care_beam_threshold(const float* pmax, int parts, int bm, float* thr, int32_t* cnt, int rows,
                                   void* stream) {
  if (!pmax || !thr || !cnt || rows <= 0 || parts <= 0) return CARE_EINVAL;
  if (bm <= 0 || bm > MAXBM) return CARE_ESHAPE;
  hipLaunchKernelGGL(beam_threshold_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, pmax, parts,
                     bm, thr, cnt, rows);
  return care_launch_status();
}

extern "C" int care_beam_pick(const float* pmax, const float* psum, int parts, const int32_t* cnt, const float* cval,
                              const int32_t* cidx, int cap, int bm, const void* A, int64_t lda, int a_dtype,
                              const void* W, int V, int K, float* cand_val, int32_t* cand_idx, int rows,
                              void* stream) {
  if (!pmax || !psum || !cnt || !cval || !cidx || !A || !W || !cand_val || !cand_idx || rows <= 0 || parts <= 0 ||
      cap <= 0 || V <= 0 || K <= 0)
    return CARE_EINVAL;
  if (bm <= 0 || bm > MAXBM || bm > V || (K % 8)) return CARE_ESHAPE;
  if (a_dtype != CARE_F32 && a_dtype != CARE_BF16) return CARE_EDTYPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % 8)) return CARE_EALIGN;
  const dim3 grid((rows + 3) / 4), block(256);
  if (a_dtype == CARE_BF16)
    hipLaunchKernelGGL(beam_pick_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, pmax, psum, parts, cnt, cval, cidx,
                       cap, bm, reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), V, K,
                       cand_val, cand_idx, rows);
  else
    hipLaunchKernelGGL(beam_pick_kernel<float>, grid, block, 0, (hipStream_t)stream, pmax, psum, parts, cnt, cval, cidx,
                       cap, bm, reinterpret_cast<const float*>(A), lda, reinterpret_cast<const bf16_t*>(W), V, K,
                       cand_val, cand_idx, rows);
  return care_launch_status();
}

extern "C" int care_beam_advance(const float* cand_val, const int32_t* cand_idx, float* scores, int bm,
                                 int32_t* tokphys, const int32_t* anc_old, int32_t* anc_new, int32_t* done,
                                 int32_t* n_fin, int fin_cap, float* fin_score, int32_t* fin_len, int32_t* fin_hyp,
                                 int t, int max_steps, int need, int eos_id, int V, int stride, int B, void* stream) {
  if (!cand_val || !cand_idx || !scores || !tokphys || !anc_old || !anc_new || !done || !n_fin || !fin_score ||
      !fin_len || !fin_hyp || B <= 0)
    return CARE_EINVAL;
  if (bm <= 0 || bm > MAXBM || t <= 0 || t >= stride || stride > 64 || need > fin_cap) return CARE_ESHAPE;
  hipLaunchKernelGGL(beam_advance_wave_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, cand_val,
                     cand_idx, scores, bm, tokphys, anc_old, anc_new, done, n_fin, fin_score, fin_len, fin_hyp,
                     fin_cap, t, max_steps, need, eos_id, V, stride, B);
  return care_launch_status();
}
