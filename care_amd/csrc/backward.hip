// backward.hip - the kernels the TRAINING-mode forward / backward needs beside the forward path's own
// (care_amd/training.py: torch.autograd.Functions over this ABI; models/Wrapper.py:423-435 -> Framework.py:215-237
// under autograd in the reference).  fp32 throughout; the matrix products of a backward pass re-use care_gemm on
// transposed operands, what is here is everything that is not a GEMM: LayerNorm / activation / softmax-attention /
// concept-head backward, dropout, the scatter-add of embedding gradients and the reductions of broadcasts.
// These are correctness-first row kernels (one wave per row or per (sequence, head)); training throughput is not the
// benchmarked path.
#include "care_common.h"

namespace {

#define BST ((hipStream_t)stream)

// ------------------------------------------------------------------ LayerNorm backward
// s = x (+ res); y = (s - mean) * rstd * gamma + beta.  ds = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
// dgamma += dy * xhat, dbeta += dy into zero-initialised [d] buffers.  One wave per row at a time, LN_BWD_ROWS rows per
// workgroup: a lane keeps the dgamma / dbeta contributions of its columns in registers over its wave's rows, the four
// waves meet in LDS, and the workgroup adds ONE value per column to the global sums (the first version issued two
// atomics per element - 1.9 M onto 512 addresses for [1856, 512]: 90 us per launch, rocprofv3 round 4).  d <= 2048.
// MAXC = columns per lane (d <= 64 MAXC) is a template parameter: sized for d = 2048 the unrolled loops carried four
// times the instructions and registers a d_model = 512 row needs (35 us per launch at [1856, 512], rocprofv3 round 4).
constexpr int LN_BWD_ROWS = 16;
template <int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* x, int64_t ldx, const float* res, int64_t ldres,
                                                     const float* gamma, const float* dy, int64_t lddy, float eps, float* ds,
                                                     int64_t ldds, float* dgamma, float* dbeta, int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float s_dg[3][64 * MAXC], s_db[3][64 * MAXC];
  float ag[MAXC], ab[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  for (int rr = wave; rr < LN_BWD_ROWS; rr += 4) {
    const int r = blockIdx.x * LN_BWD_ROWS + rr;
    if (r >= rows) break;  // (wave-uniform)
    float s[MAXC], g[MAXC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      s[i] = 0.f;
      if (c < d) {
        s[i] = x[(int64_t)r * ldx + c] + (res ? res[(int64_t)r * ldres + c] : 0.f);
        sum += s[i];
      }
    }
    const float mean = care_wave_sum(sum) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < d) { const float a = s[i] - mean; q += a * a; }
    const float rstd = 1.0f / sqrtf(care_wave_sum(q) / (float)d + eps);
    float mg = 0.f, mgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      g[i] = 0.f;
      if (c < d) {
        const float xh = (s[i] - mean) * rstd;
        const float dyv = dy[(int64_t)r * lddy + c];
        g[i] = dyv * gamma[c];
        mg += g[i];
        mgx += g[i] * xh;
        ag[i] += dyv * xh;
        ab[i] += dyv;
        s[i] = xh;
      }
    }
    mg = care_wave_sum(mg) / (float)d;
    mgx = care_wave_sum(mgx) / (float)d;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < d) ds[(int64_t)r * ldds + c] = rstd * (g[i] - mg - s[i] * mgx);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < d) { s_dg[wave - 1][lane + 64 * i] = ag[i]; s_db[wave - 1][lane + 64 * i] = ab[i]; }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < d) {
        atomicAdd(dgamma + c, ag[i] + ((s_dg[0][c] + s_dg[1][c]) + s_dg[2][c]));
        atomicAdd(dbeta + c, ab[i] + ((s_db[0][c] + s_db[1][c]) + s_db[2][c]));
      }
    }
  }
}

// ------------------------------------------------------------------ activation, dropout
__device__ __forceinline__ float b_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float b_gelu_grad(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// dy == nullptr: out = act(z); else out = dy * act'(z)
__global__ void act_kernel(const float* z, const float* dy, float* out, int64_t n, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = z[i];
  if (!dy) out[i] = act == CARE_ACT_RELU ? fmaxf(v, 0.f) : (act == CARE_ACT_GELU ? b_gelu(v) : v);
  else out[i] = dy[i] * (act == CARE_ACT_RELU ? (v > 0.f ? 1.f : 0.f) : (act == CARE_ACT_GELU ? b_gelu_grad(v) : 1.f));
}

// counter-based uniform in [0, 1): a 64-bit mix (splitmix64 finaliser) of (seed, element index) - stateless, so the
// backward pass re-creates the forward's mask from the same (seed, index)
__device__ __forceinline__ float b_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// out = x * keep / (1 - p), keep = uniform(seed, i) >= p  (nn.Dropout in training mode; the same call is its backward)
__global__ void dropout_kernel(const float* x, float* out, int64_t n, float p, unsigned long long seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = b_uniform(seed, (unsigned long long)i) >= p ? x[i] * (1.0f / (1.0f - p)) : 0.f;
}

// ------------------------------------------------------------------ reductions of broadcasts, scatter-add
// out[r][c] = scale * sum_{k < terms} x[(r * row_stride + k * term_stride)][c]
__global__ void strided_sum_kernel(const float* x, int64_t ldx, float* out, int64_t ldo, int rows, int d, int terms,
                                   int64_t row_stride, int64_t term_stride, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * d) return;
  const int r = (int)(i / d), c = (int)(i % d);
  float s = 0.f;
  for (int k = 0; k < terms; ++k) s += x[((int64_t)r * row_stride + (int64_t)k * term_stride) * ldx + c];
  out[(int64_t)r * ldo + c] = s * scale;
}

// The same sum for MANY terms per output element (a bias gradient: one output row, terms = every row of dy): the
// one-thread-per-element kernel above walks its terms in series - 515 us for [1856, 512] -> [512], 8.2 of the 14.2 ms of a
// training step (rocprofv3, round 4).  Here a workgroup owns (output row, 64 columns): 16 waves, wave w sums terms
// w, w + 16, ... (64 consecutive floats per wave and term), the 16 partial rows meet in LDS and are added in wave
// order - the same order every run.
__global__ __launch_bounds__(1024) void strided_sum_tile_kernel(const float* x, int64_t ldx, float* out, int64_t ldo, int rows, int d,
                                                                  int terms, int64_t row_stride, int64_t term_stride, float scale) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tiles = (d + 63) >> 6, r = blockIdx.x / tiles, c = (blockIdx.x % tiles) * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (c < d) {
    const float* xp = x + (int64_t)r * row_stride * ldx + c;
    int k = w;
    for (; k + 16 < terms; k += 32) {  // two independent chains
      s0 += xp[(int64_t)k * term_stride * ldx];
      s1 += xp[(int64_t)(k + 16) * term_stride * ldx];
    }
    if (k < terms) s0 += xp[(int64_t)k * term_stride * ldx];
  }
  part[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && c < d) {
    float s = part[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) s += part[q][lane];
    out[(int64_t)r * ldo + c] = s * scale;
  }
}

// dst[i][c] = scale * src[i / grp][c]  (backward of a mean / sum over groups of `grp` consecutive rows; also a broadcast add's forward operand)
__global__ void bcast_rows_kernel(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int d, int grp, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * d) return;
  const int r = (int)(i / d), c = (int)(i % d);
  dst[(int64_t)r * ldd + c] = scale * src[(int64_t)(r / grp) * lds_ + c];
}

// out[r] = x[r] + pos[r % seq] + sem[r / sem_div]  (Embeddings.forward before its LayerNorm; pos / sem optional)
__global__ void add_pos_sem_kernel(const float* x, const float* pos, const float* sem, float* out, int rows, int d, int seq,
                                   int sem_div) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * d) return;
  const int r = (int)(i / d), c = (int)(i % d);
  float v = x[i];
  if (pos) v += pos[(int64_t)(r % seq) * d + c];
  if (sem) v += sem[(int64_t)(r / sem_div) * d + c];
  out[i] = v;
}

// table[idx[i]][c] += src[i][c]  (embedding backward); rows with idx < 0 or == skip_idx are skipped (padding_idx)
__global__ void scatter_add_rows_kernel(const float* src, int64_t lds_, const int32_t* idx, float* table, int64_t ldt, int rows,
                                        int d, int skip_idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * d) return;
  const int r = (int)(i / d), c = (int)(i % d);
  const int t = idx[r];
  if (t < 0 || t == skip_idx) return;
  atomicAdd(table + (int64_t)t * ldt + c, src[(int64_t)r * lds_ + c]);
}

// ------------------------------------------------------------------ concept head backward
// p = sigmoid(s); preds = 1 - exp(log(clamp(1 - p, 1e-12, 1))) (= p while 1 - p >= 1e-12), avg = mean_k p
// (pred_attribute.py:17-46 at seq_len 1).  ds = (dpreds [1 - p >= 1e-12] + davg / k) * p (1 - p)
__global__ void concept_bwd_kernel(const float* scores, int64_t lds_, const float* dpreds, int64_t ldp, const float* davg,
                                   float* ds, int64_t ldo, int B, int k) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * k) return;
  const int b = (int)(i / k), c = (int)(i % k);
  const float p = 1.0f / (1.0f + expf(-scores[(int64_t)b * lds_ + c]));
  float g = davg ? davg[b] / (float)k : 0.f;
  if (dpreds && (1.0f - p) >= 1e-12f) g += dpreds[(int64_t)b * ldp + c];
  ds[(int64_t)b * ldo + c] = g * p * (1.0f - p);
}

// ------------------------------------------------------------------ attention: P V and the backward of softmax(QK^T / 8 + ...) V
struct AttnBArgs {
  const float* Q; int64_t ldq;           // [nseq * seq, ldq], head h at columns 64 h
  const float* K; const float* V; int64_t kv_bs, kv_rs;  // key j of sequence s at base + s * kv_bs + j * kv_rs + 64 h
  const float* P;                        // probabilities AFTER softmax, BEFORE dropout: [nseq * seq, heads, nkeys]
  const float* dctx; int64_t ldd;        // [nseq * seq, ldd]
  float* ctx; int64_t ldc;               // forward output
  float* dQ; int64_t lddq; float* dK; float* dV; int64_t dkv_bs, dkv_rs;
  float* dbias; int bias_ld;             // [heads, bias_ld] (atomics) or null
  int nseq, seq, nkeys, heads;
  float p_drop; unsigned long long seed;
};

__device__ __forceinline__ float attn_keep(const AttnBArgs& a, int64_t pidx) {
  if (a.p_drop <= 0.f) return 1.0f;
  return b_uniform(a.seed, (unsigned long long)pidx) >= a.p_drop ? 1.0f / (1.0f - a.p_drop) : 0.f;
}

// ctx[s, i, 64 h + e] = sum_j keep(i, j) P[s, i, h, j] V[s, j, 64 h + e]: one wave per (sequence, head, query i), lane = e.
// The row of probabilities is fetched (and its dropout mask evaluated) ONCE, key j by lane j % 64, and handed round by
// readlane; the first version - a wave per (sequence, head) walking all queries, every lane loading and hashing
// every (i, j) - took 355 us per launch at 64 clips (rocprofv3, round 4).
__global__ __launch_bounds__(256) void attn_pv_kernel(AttnBArgs a) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6), e = threadIdx.x & 63;
  if (item >= a.nseq * a.heads * a.seq) return;  // (wave-uniform)
  const int i = item % a.seq, sh = item / a.seq, s = sh / a.heads, h = sh % a.heads;
  const float* Vb = a.V + (int64_t)s * a.kv_bs + h * 64 + e;
  const int64_t prow = ((int64_t)(s * a.seq + i) * a.heads + h) * a.nkeys;
  float acc = 0.f;
  for (int j0 = 0; j0 < a.nkeys; j0 += 64) {
    const int j = j0 + e;
    const float pd = j < a.nkeys ? a.P[prow + j] * attn_keep(a, prow + j) : 0.f;
    const int n = min(64, a.nkeys - j0);
    for (int jj = 0; jj < n; ++jj) {
      const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pd), jj));
      acc = fmaf(pj, Vb[(int64_t)(j0 + jj) * a.kv_rs], acc);
    }
  }
  a.ctx[(int64_t)(s * a.seq + i) * a.ldc + h * 64 + e] = acc;
}

// One wave per (sequence, head).  LDS: dS row [nkeys], dK / dV accumulators [nkeys][64].
//   dPd[j] = dctx_i . V_j ; dP = dPd * keep ; dS = P (dP - sum_j dP P) ; dQ_i = dS K / 8 ; dK_j += dS_j Q_i / 8 ;
//   dV_j += (P keep)_ij dctx_i ; dbias[h][j] += dS_j
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnBArgs a) {
  extern __shared__ float sh[];
  float* sdS = sh;                    // [nkeys]
  float* sPd = sh + 128;              // [nkeys]
  float* sdc = sh + 256;              // [64] dctx row
  float* sq = sh + 320;               // [64] Q row
  float* sdK = sh + 384;              // [nkeys][64]
  float* sdV = sdK + a.nkeys * 64;    // [nkeys][64]
  const int s = blockIdx.x / a.heads, h = blockIdx.x % a.heads, lane = threadIdx.x;
  const float* Kb = a.K + (int64_t)s * a.kv_bs + h * 64;
  const float* Vb = a.V + (int64_t)s * a.kv_bs + h * 64;
  for (int j = 0; j < a.nkeys; ++j) { sdK[j * 64 + lane] = 0.f; sdV[j * 64 + lane] = 0.f; }
  __syncthreads();
  for (int i = 0; i < a.seq; ++i) {
    const int64_t row = (int64_t)s * a.seq + i;
    const int64_t prow = (row * a.heads + h) * a.nkeys;
    sdc[lane] = a.dctx[row * a.ldd + h * 64 + lane];
    sq[lane] = a.Q[row * a.ldq + h * 64 + lane];
    __syncthreads();
    // lanes over keys
    float dot = 0.f;
    float dP[2], Pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = lane + 64 * t;
      dP[t] = 0.f; Pv[t] = 0.f;
      if (j < a.nkeys) {
        float acc = 0.f;
        const float* vj = Vb + (int64_t)j * a.kv_rs;
        for (int e = 0; e < 64; ++e) acc = fmaf(sdc[e], vj[e], acc);
        const float keep = attn_keep(a, prow + j);
        Pv[t] = a.P[prow + j];
        dP[t] = acc * keep;
        sPd[j] = Pv[t] * keep;
        dot += dP[t] * Pv[t];
      }
    }
    dot = care_wave_sum(dot);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = lane + 64 * t;
      if (j < a.nkeys) {
        const float dS = Pv[t] * (dP[t] - dot);
        sdS[j] = dS;
        if (a.dbias) atomicAdd(a.dbias + (int64_t)h * a.bias_ld + j, dS);
      }
    }
    __syncthreads();
    // lanes over the 64 dims
    float dq = 0.f;
    const float qe = sq[lane], de = sdc[lane];
    for (int j = 0; j < a.nkeys; ++j) {
      const float dS = sdS[j];
      dq = fmaf(dS, Kb[(int64_t)j * a.kv_rs + lane], dq);
      sdK[j * 64 + lane] = fmaf(dS, qe, sdK[j * 64 + lane]);
      sdV[j * 64 + lane] = fmaf(sPd[j], de, sdV[j * 64 + lane]);
    }
    a.dQ[row * a.lddq + h * 64 + lane] = dq * 0.125f;
    __syncthreads();
  }
  for (int j = 0; j < a.nkeys; ++j) {
    a.dK[(int64_t)s * a.dkv_bs + (int64_t)j * a.dkv_rs + h * 64 + lane] = sdK[j * 64 + lane] * 0.125f;
    a.dV[(int64_t)s * a.dkv_bs + (int64_t)j * a.dkv_rs + h * 64 + lane] = sdV[j * 64 + lane];
  }
}

// The same backward as FOUR small exact-f32 matrix products per (sequence, head) on the matrix cores (round 6; seq <= 32 queries,
// nkeys <= 128): one workgroup of 4 waves,
//   dPd^T [keys x queries] = V_h [keys x 64] . dctx_h^T          (v_mfma_f32_16x16x4_f32: bit-wise an fp32 fma chain per output)
//   dS = P o (dPd o keep - rowsum(dPd o keep o P))               (a wave per query row, lanes over keys)
//   dQ [queries x 64] = dS . K_h / 8 ;  dK [keys x 64] = dS^T . Q_h / 8 ;  dV [keys x 64] = (P o keep)^T . dctx_h
// dS and P o keep live in LDS ([32][132] each), Q_h and dctx_h too ([32][68]); K_h / V_h fragments come straight from memory.
// The one-wave form above walks the queries in series with its dK / dV accumulators in LDS (two read-modify-writes per key and
// query) and reads V through 64 different cache lines per instruction: *measured* round 6, 512 clips: 2.85 ms per launch for the
// cross-attention (114 keys), 3.1 ms of a 15.3 ms training step (profiles/r06_training_step_B512_kernel_stats.csv).
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(AttnBArgs a) {
  __shared__ float sQ[32][68], sD[32][68], sS[32][132], sP[32][132];
  const int s = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  const float* Kb = a.K + (int64_t)s * a.kv_bs + h * 64;
  const float* Vb = a.V + (int64_t)s * a.kv_bs + h * 64;
  const int KT = (a.nkeys + 15) >> 4, QT = (a.seq + 15) >> 4;
  for (int i = tid; i < 32 * 64; i += 256) {
    const int q = i >> 6, e = i & 63;
    const int64_t row = (int64_t)s * a.seq + q;
    sQ[q][e] = q < a.seq ? a.Q[row * a.ldq + h * 64 + e] : 0.f;
    sD[q][e] = q < a.seq ? a.dctx[row * a.ldd + h * 64 + e] : 0.f;
  }
  __syncthreads();
  // ---- dPd[key][query]: tile (kt, qt) -> sS[query][key]
  for (int ti = wave; ti < KT * QT; ti += 4) {
    const int kt = ti / QT, qt = ti - kt * QT;
    const int key = kt * 16 + l16;
    const float* vrow = Vb + (int64_t)min(key, a.nkeys - 1) * a.kv_rs + kg;
    float fa[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) fa[ks] = key < a.nkeys ? vrow[4 * ks] : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], sD[qt * 16 + l16][4 * ks + kg], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) sS[qt * 16 + l16][kt * 16 + 4 * kg + e] = acc[e];
  }
  __syncthreads();
  // ---- softmax backward per query row (rows past the sequence and keys past nkeys: zeros)
  for (int i = wave; i < 32; i += 4) {
    const int64_t prow = (((int64_t)s * a.seq + i) * a.heads + h) * a.nkeys;
    float dP[2], Pv[2], Pk[2];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = lane + 64 * t;
      dP[t] = Pv[t] = Pk[t] = 0.f;
      if (i < a.seq && j < a.nkeys) {
        const float keep = attn_keep(a, prow + j);
        Pv[t] = a.P[prow + j];
        dP[t] = sS[i][j] * keep;
        Pk[t] = Pv[t] * keep;
        dot += dP[t] * Pv[t];
      }
    }
    dot = care_wave_sum(dot);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = lane + 64 * t;
      sS[i][j] = Pv[t] * (dP[t] - dot);
      sP[i][j] = Pk[t];
    }
  }
  __syncthreads();
  if (a.dbias && tid < a.nkeys) {  // column sums over this sequence's queries, then one atomic per (head, key)
    float c = 0.f;
    for (int i = 0; i < a.seq; ++i) c += sS[i][tid];
    atomicAdd(a.dbias + (int64_t)h * a.bias_ld + tid, c);
  }
  // ---- dQ[query][dim] = dS . K / 8: tile (qt, dt)
  for (int ti = wave; ti < QT * 4; ti += 4) {
    const int qt = ti >> 2, dt = ti & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KT * 4; ++ks) {
      const int key = 4 * ks + kg;
      const float fb = key < a.nkeys ? Kb[(int64_t)key * a.kv_rs + dt * 16 + l16] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sS[qt * 16 + l16][key], fb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = qt * 16 + 4 * kg + e;
      if (q < a.seq) a.dQ[((int64_t)s * a.seq + q) * a.lddq + h * 64 + dt * 16 + l16] = acc[e] * 0.125f;
    }
  }
  // ---- dK[key][dim] = dS^T . Q / 8 and dV[key][dim] = (P keep)^T . dctx: tile (kt, dt)
  for (int ti = wave; ti < KT * 4; ti += 4) {
    const int kt = ti >> 2, dt = ti & 3;
    f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks >= QT * 4) break;
      const int q = 4 * ks + kg;
      ak = __builtin_amdgcn_mfma_f32_16x16x4f32(sS[q][kt * 16 + l16], sQ[q][dt * 16 + l16], ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_16x16x4f32(sP[q][kt * 16 + l16], sD[q][dt * 16 + l16], av, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = kt * 16 + 4 * kg + e;
      if (key < a.nkeys) {
        const int64_t o = (int64_t)s * a.dkv_bs + (int64_t)key * a.dkv_rs + h * 64 + dt * 16 + l16;
        a.dK[o] = ak[e] * 0.125f;
        a.dV[o] = av[e];
      }
    }
  }
}

// ... and the forward's ctx = (P o keep) V in the same form (seq <= 32, nkeys <= 128): a workgroup per (sequence, head), the
// masked probabilities of its <= 32 queries in LDS, ctx[query][dim] tiles on the exact-f32 matrix cores with V_h fragments
// straight from memory (attn_pv_kernel above: a wave per query walking the keys with readlane broadcasts, 0.78 ms per step at 512 clips).
__global__ __launch_bounds__(256) void attn_pv_mfma_kernel(AttnBArgs a) {
  __shared__ float sP[32][132];
  const int s = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  const float* Vb = a.V + (int64_t)s * a.kv_bs + h * 64;
  const int KT = (a.nkeys + 15) >> 4, QT = (a.seq + 15) >> 4;
  for (int i = tid; i < 32 * 128; i += 256) {
    const int q = i >> 7, j = i & 127;
    float v = 0.f;
    if (q < a.seq && j < a.nkeys) {
      const int64_t pidx = (((int64_t)s * a.seq + q) * a.heads + h) * a.nkeys + j;
      v = a.P[pidx] * attn_keep(a, pidx);
    }
    sP[q][j] = v;
  }
  __syncthreads();
  for (int ti = wave; ti < QT * 4; ti += 4) {
    const int qt = ti >> 2, dt = ti & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KT * 4; ++ks) {
      const int key = 4 * ks + kg;
      const float fb = key < a.nkeys ? Vb[(int64_t)key * a.kv_rs + dt * 16 + l16] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sP[qt * 16 + l16][key], fb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = qt * 16 + 4 * kg + e;
      if (q < a.seq) a.ctx[((int64_t)s * a.seq + q) * a.ldc + h * 64 + dt * 16 + l16] = acc[e];
    }
  }
}

inline unsigned grid1d(int64_t n) { return (unsigned)((n + 255) / 256); }


// ------------------------------------------------------------------ C = op(A) B, exact f32, K-major operands
// The two products of an nn.Linear's backward with the operands as they lie in memory (the first version made
// .t().contiguous() copies for care_gemm, which multiplies A [M, K] by W [N, K]^T only: 49 copy launches, 0.58 ms of a
// 5.2 ms step, rocprofv3 round 4):
//   dx [M, N] = dy [M, K] W [K, N]      (TA = false: A row-major [M, K]);
//   dW [M, N] = dy^T x, dy [K, M], x [K, N]   (TA = true: A is stored [K, M], the reduction index is its ROW).
// B is [K, N] row-major in both.  64 x 64 or 128 x 128 tiles, K steps of 16 through LDS as [k][m] / [k][n] (+ 4 floats of padding),
// v_mfma_f32_16x16x4_f32 (bit-wise an fp32 fma chain per output), a wave owns a quarter of the tile; the next step's tile
// travels in registers while the current one is multiplied.  Any M, N, K, any leading dimensions (scalar, coalesced loads).
template <bool TA, int BT>  // BT x BT output tiles (64 or 128): a wave owns (BT / 2) x (BT / 2) outputs
__global__ __launch_bounds__(256) void gemm_kn_kernel(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                                       int M, int N, int K, int kchunk, int64_t c_slab) {
  constexpr int BM = BT, BN = BT, BK = 16, LDT = BT + 4, NT = BT / 32, NL = BT * BK / 256;  // NT 16 x 16 tiles per wave and side
  __shared__ float sA[2][BK][LDT], sB[2][BK][LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
  f32x4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ra[NL], rb[NL];
  // element q = tid + 256 i of a tile: TA / B: (k = q / BT, c = q % BT) - consecutive floats of a row per wave;
  // !TA: (m = q / 16, k = q % 16) - 16 consecutive floats of 4 rows per wave-quarter
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + 256 * i;
      if constexpr (TA) {
        const int k = q / BT, m = q % BT;
        ra[i] = (k0 + k < K && m0 + m < M) ? A[(int64_t)(k0 + k) * lda + m0 + m] : 0.f;
      } else {
        const int m = q >> 4, k = q & 15;
        ra[i] = (k0 + k < K && m0 + m < M) ? A[(int64_t)(m0 + m) * lda + k0 + k] : 0.f;
      }
      const int kb = q / BT, n = q % BT;
      rb[i] = (k0 + kb < K && n0 + n < N) ? B[(int64_t)(k0 + kb) * ldb + n0 + n] : 0.f;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + 256 * i;
      if constexpr (TA) sA[buf][q / BT][q % BT] = ra[i];
      else sA[buf][q & 15][q >> 4] = ra[i];
      sB[buf][q / BT][q % BT] = rb[i];
    }
  };
  // split K (gridDim.y > 1): this block multiplies K elements [kbeg, kend) into slab blockIdx.y of C
  const int kbeg = blockIdx.y * kchunk;
  K = min(K, kbeg + kchunk);
  C += (int64_t)blockIdx.y * c_slab;
  fetch(kbeg);
  stage(0);
  __syncthreads();
  const int nk = (K - kbeg + BK - 1) / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch(kbeg + (kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float fa[NT], fb[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) fa[a] = sA[buf][kk + kg][wm + 16 * a + l16];
#pragma unroll
      for (int b = 0; b < NT; ++b) fb[b] = sB[buf][kk + kg][wn + 16 * b + l16];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) stage(buf ^ 1);
    __syncthreads();
  }
  // D[4 kg + e][l16] of each 16 x 16 tile
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = m0 + wm + 16 * a + 4 * kg + e, c = n0 + wn + 16 * b + l16;
        if (r < M && c < N) C[(int64_t)r * ldc + c] = acc[a][b][e];
      }
}

}  // namespace

extern "C" int care_ln_bwd(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* gamma, const float* dy,
                           int64_t lddy, float eps, float* ds, int64_t ldds, float* dgamma, float* dbeta, int rows, int d,
                           void* stream) {
  if (!x || !gamma || !dy || !ds || !dgamma || !dbeta || rows <= 0 || d <= 0) return CARE_EINVAL;
  if (d > 2048) return CARE_ESHAPE;
  const dim3 grid((rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS);
#define LN_BWD_LAUNCH(C) hipLaunchKernelGGL(ln_bwd_kernel<C>, grid, dim3(256), 0, BST, x, ldx, res, ldres, gamma, dy, lddy, eps, ds, ldds, dgamma, dbeta, rows, d)
  if (d <= 512) LN_BWD_LAUNCH(8);
  else if (d <= 1024) LN_BWD_LAUNCH(16);
  else LN_BWD_LAUNCH(32);
#undef LN_BWD_LAUNCH
  return care_launch_status();
}

extern "C" int care_act(const float* z, const float* dy, float* out, int64_t n, int act, void* stream) {
  if (!z || !out || n <= 0) return CARE_EINVAL;
  if (act < CARE_ACT_NONE || act > CARE_ACT_GELU) return CARE_EDTYPE;
  hipLaunchKernelGGL(act_kernel, dim3(grid1d(n)), dim3(256), 0, BST, z, dy, out, n, act);
  return care_launch_status();
}

extern "C" int care_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream) {
  if (!x || !out || n <= 0 || !(p >= 0.f && p < 1.f)) return CARE_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid1d(n)), dim3(256), 0, BST, x, out, n, p, (unsigned long long)seed);
  return care_launch_status();
}

extern "C" int care_strided_sum(const float* x, int64_t ldx, float* out, int64_t ldo, int rows, int d, int terms,
                                int64_t row_stride, int64_t term_stride, float scale, void* stream) {
  if (!x || !out || rows <= 0 || d <= 0 || terms <= 0) return CARE_EINVAL;
  if (terms >= 64 && (int64_t)rows * ((d + 63) / 64) < (1 << 20)) {  // few outputs, many terms each: a workgroup per (row, 64 columns)
    hipLaunchKernelGGL(strided_sum_tile_kernel, dim3(rows * ((d + 63) / 64)), dim3(1024), 0, BST, x, ldx, out, ldo, rows, d, terms,
                       row_stride, term_stride, scale);
    return care_launch_status();
  }
  hipLaunchKernelGGL(strided_sum_kernel, dim3(grid1d((int64_t)rows * d)), dim3(256), 0, BST, x, ldx, out, ldo, rows, d, terms,
                     row_stride, term_stride, scale);
  return care_launch_status();
}

extern "C" int care_bcast_rows(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int d, int grp, float scale,
                               void* stream) {
  if (!src || !dst || rows <= 0 || d <= 0 || grp <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(grid1d((int64_t)rows * d)), dim3(256), 0, BST, src, lds_, dst, ldd, rows, d, grp, scale);
  return care_launch_status();
}

extern "C" int care_add_pos_sem(const float* x, const float* pos, const float* sem, float* out, int rows, int d, int seq,
                                int sem_div, void* stream) {
  if (!x || !out || rows <= 0 || d <= 0 || seq <= 0 || sem_div <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(add_pos_sem_kernel, dim3(grid1d((int64_t)rows * d)), dim3(256), 0, BST, x, pos, sem, out, rows, d, seq, sem_div);
  return care_launch_status();
}

extern "C" int care_scatter_add_rows(const float* src, int64_t lds_, const int32_t* idx, float* table, int64_t ldt, int rows,
                                     int d, int skip_idx, void* stream) {
  if (!src || !idx || !table || rows <= 0 || d <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid1d((int64_t)rows * d)), dim3(256), 0, BST, src, lds_, idx, table, ldt, rows,
                     d, skip_idx);
  return care_launch_status();
}

extern "C" int care_concept_bwd(const float* scores, int64_t lds_, const float* dpreds, int64_t ldp, const float* davg, float* ds,
                                int64_t ldo, int B, int k, void* stream) {
  if (!scores || !ds || B <= 0 || k <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(concept_bwd_kernel, dim3(grid1d((int64_t)B * k)), dim3(256), 0, BST, scores, lds_, dpreds, ldp, davg, ds, ldo, B, k);
  return care_launch_status();
}

extern "C" int care_attn_pv(const float* P, const float* V, int64_t kv_bs, int64_t kv_rs, float* ctx, int64_t ldc, int nseq,
                            int seq, int nkeys, int heads, float p_drop, uint64_t seed, void* stream) {
  if (!P || !V || !ctx || nseq <= 0 || seq <= 0 || nkeys <= 0 || heads <= 0) return CARE_EINVAL;
  AttnBArgs a{};
  a.P = P; a.V = V; a.kv_bs = kv_bs; a.kv_rs = kv_rs; a.ctx = ctx; a.ldc = ldc;
  a.nseq = nseq; a.seq = seq; a.nkeys = nkeys; a.heads = heads; a.p_drop = p_drop; a.seed = seed;
  static const bool one_wave = [] { const char* e = getenv("CARE_ATTN_BWD_MFMA"); return e && atoi(e) == 0; }();
  if (seq <= 32 && nkeys <= 128 && !one_wave) {
    hipLaunchKernelGGL(attn_pv_mfma_kernel, dim3(nseq * heads), dim3(256), 0, BST, a);
    return care_launch_status();
  }
  hipLaunchKernelGGL(attn_pv_kernel, dim3((nseq * heads * seq + 3) / 4), dim3(256), 0, BST, a);
  return care_launch_status();
}

extern "C" int care_attn_bwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t kv_bs, int64_t kv_rs,
                             const float* P, const float* dctx, int64_t ldd, float* dQ, int64_t lddq, float* dK, float* dV,
                             int64_t dkv_bs, int64_t dkv_rs, float* dbias, int bias_ld, int nseq, int seq, int nkeys, int heads,
                             float p_drop, uint64_t seed, void* stream) {
  if (!Q || !K || !V || !P || !dctx || !dQ || !dK || !dV || nseq <= 0 || seq <= 0 || nkeys <= 0 || heads <= 0) return CARE_EINVAL;
  if (nkeys > 128) return CARE_ESHAPE;
  AttnBArgs a{};
  a.Q = Q; a.ldq = ldq; a.K = K; a.V = V; a.kv_bs = kv_bs; a.kv_rs = kv_rs; a.P = P; a.dctx = dctx; a.ldd = ldd;
  a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.dV = dV; a.dkv_bs = dkv_bs; a.dkv_rs = dkv_rs; a.dbias = dbias; a.bias_ld = bias_ld;
  a.nseq = nseq; a.seq = seq; a.nkeys = nkeys; a.heads = heads; a.p_drop = p_drop; a.seed = seed;
  static const bool one_wave = [] { const char* e = getenv("CARE_ATTN_BWD_MFMA"); return e && atoi(e) == 0; }();
  if (seq <= 32 && !one_wave) {  // the matrix-core form: a workgroup per (sequence, head)
    hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(nseq * heads), dim3(256), 0, BST, a);
    return care_launch_status();
  }
  const int lds = (384 + 2 * nkeys * 64) * 4;
  static std::atomic<unsigned long long> ok{0};
  if (lds > 64 * 1024)
    if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&attn_bwd_kernel), lds, ok)) return e;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(nseq * heads), dim3(64), lds, BST, a);
  return care_launch_status();
}

// ksplit > 1: K is cut into ksplit ranges (multiples of 16), range s multiplied into slab s of C (C + s c_slab, each
// [M, ldc]); the caller adds the slabs (care_strided_sum: terms = ksplit, in order - deterministic).  For products with
// few output tiles and a long K: dx = dlogits W at 64 clips is 232 tiles of 64 x 64 with K = 10547 - 523 us as one
// workgroup of one wave per SIMD per tile (rocprofv3, round 4).
extern "C" int care_gemm_kn_splitk(const float* A, int64_t lda, int a_is_km, const float* B, int64_t ldb, float* C, int64_t ldc,
                                   int64_t c_slab, int M, int N, int K, int ksplit, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ksplit <= 0 || ksplit > 64) return CARE_EINVAL;
  if (lda < (a_is_km ? M : K) || ldb < N || ldc < N || (ksplit > 1 && c_slab < (int64_t)M * ldc)) return CARE_EINVAL;
  // 128 x 128 tiles where they fill the chip (16 flop per loaded byte at 64 x 64: the vocabulary-sized dW ran at 28 TFLOP/s)
  const int64_t big = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  const int bt = big >= 200 ? 128 : 64;
  const int64_t tiles = (int64_t)((M + bt - 1) / bt) * ((N + bt - 1) / bt);
  if (tiles > 0x7fffffff) return CARE_ESHAPE;
  const int kchunk = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
  const int splits = (K + kchunk - 1) / kchunk;  // <= ksplit; the slabs past it are not written
  if (splits != ksplit) return CARE_ESHAPE;      // (pick ksplit so that every slab gets a range: care_gemm_kn_splits)
#define KN_LAUNCH(TA, BT) hipLaunchKernelGGL((gemm_kn_kernel<TA, BT>), dim3((unsigned)tiles, ksplit), dim3(256), 0, BST, A, lda, B, ldb, C, ldc, M, N, K, kchunk, c_slab)
  if (a_is_km) { if (bt == 128) KN_LAUNCH(true, 128); else KN_LAUNCH(true, 64); }
  else { if (bt == 128) KN_LAUNCH(false, 128); else KN_LAUNCH(false, 64); }
#undef KN_LAUNCH
  return care_launch_status();
}

// how many K ranges care_gemm_kn_splitk should be given for this product (1: do not split)
extern "C" int care_gemm_kn_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return CARE_EINVAL;
  const int64_t big = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  const int bt = big >= 200 ? 128 : 64;
  const int64_t tiles = (int64_t)((M + bt - 1) / bt) * ((N + bt - 1) / bt);
  if (tiles >= 512 || K < 2048) return 1;
  int ks = (int)((1024 + tiles - 1) / tiles);
  ks = ks < 2 ? 2 : (ks > 8 ? 8 : ks);
  for (; ks > 1; --ks) {  // every slab must get a range
    const int kchunk = ((K + ks - 1) / ks + 15) / 16 * 16;
    if ((K + kchunk - 1) / kchunk == ks) break;
  }
  return ks;
}

extern "C" int care_gemm_kn(const float* A, int64_t lda, int a_is_km, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                            int N, int K, void* stream) {
  return care_gemm_kn_splitk(A, lda, a_is_km, B, ldb, C, ldc, 0, M, N, K, 1, stream);
}
