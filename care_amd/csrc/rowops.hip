// rowops.hip - HBM-bound row kernels: LayerNorm epilogues, embeddings, concept head, greedy state.
//
// All of these move a few KB per row and do O(d) work, so the design rule is simply:
// one 64-lane wave per row, every access a coalesced float4, statistics in fp32 via
// wave shuffles (no LDS), 4 rows per 256-thread workgroup.  A row of d <= 2048 fp32
// stays in registers between the statistics and the normalisation (read once, write once).
#include "care_common.h"

namespace {

constexpr int MAXV = 8;  // float4 per lane: d <= 64 * 4 * 8 = 2048

// LayerNorm of one row held as v[0..nv) float4 per lane (chunk index = lane + 64*i).
// `outb` (optional): bf16 mirror of the normalised row, the A operand of the next bf16 GEMM.
__device__ __forceinline__ void row_layernorm(float4 (&v)[MAXV], int nv4, int lane, int d,
                                              const float* gamma, const float* beta, float eps,
                                              float* out, bf16_t* outb = nullptr) {
  if (gamma == nullptr) {  // (kernel-uniform) no LayerNorm: the plain sum - the sub-blocks of a pre-LN decoder
    // (models/components/SubLayers.py:55,78: `context + input_tensor` without the LayerNorm) and its un-normalised embedding
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv4) {
        *reinterpret_cast<float4*>(out + c4 * 4) = v[i];
        if (outb) {
          bf16x4 ob;
          ob[0] = (bf16_t)v[i].x; ob[1] = (bf16_t)v[i].y; ob[2] = (bf16_t)v[i].z; ob[3] = (bf16_t)v[i].w;
          *reinterpret_cast<bf16x4*>(outb + c4 * 4) = ob;
        }
      }
    }
    return;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + 64 * i < nv4) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = care_wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + 64 * i < nv4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
      q += (a * a + b * b) + (c * c + e * e);
    }
  const float var = care_wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv4) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
      const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(out + c4 * 4) = o;
      if (outb) {
        bf16x4 ob;
        ob[0] = (bf16_t)o.x; ob[1] = (bf16_t)o.y; ob[2] = (bf16_t)o.z; ob[3] = (bf16_t)o.w;
        *reinterpret_cast<bf16x4*>(outb + c4 * 4) = ob;
      }
    }
  }
}

__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

__global__ __launch_bounds__(256) void add_ln_kernel(const float* x, int64_t ldx, const float* res, int64_t ldres,
                                                     const float* pos, const float* gamma, const float* beta,
                                                     float eps, float* out, bf16_t* outb, int64_t ldo, int rows, int d,
                                                     int grp, int out_grp_rows, int out_row_off, int nslab,
                                                     int64_t slab_stride) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int nv4 = d >> 2;
  float4 v[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv4) {
      v[i] = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c4 * 4);
      for (int sl = 1; sl < nslab; ++sl)  // split-K partial products of the preceding GEMM
        add4(v[i], *reinterpret_cast<const float4*>(x + sl * slab_stride + (int64_t)r * ldx + c4 * 4));
      if (res) add4(v[i], *reinterpret_cast<const float4*>(res + (int64_t)r * ldres + c4 * 4));
      if (pos) add4(v[i], *reinterpret_cast<const float4*>(pos + (int64_t)(r % grp) * d + c4 * 4));
    }
  }
  const int64_t orow = (int64_t)(r / grp) * out_grp_rows + out_row_off + (r % grp);
  row_layernorm(v, nv4, lane, d, gamma, beta, eps, out + orow * ldo, outb ? outb + orow * ldo : nullptr);
}

// The same kernel for d = 256 NV exactly, no position table, no split-K slabs (every LayerNorm launch of the d_model
// 768 / 1024 decode steps): with the row length a template parameter every load of the row, the residual, gamma and
// beta is issued before the first use - the general kernel above tests `c4 < nv4`, `res`, `pos` and the slab count at run
// time, hipcc branches around each load and waits for it at the join, and a row is three or four HBM round trips in
// series (*measured* 16384 rows x 1024: 70 -> 41 us per launch, 3.4 -> 5.8 TB/s of its 235 MB).  The arithmetic is
// row_layernorm's in the same order; hipcc contracts the two bodies' multiply-adds differently, so a few rows in a
// hundred differ in the last bit or two of their fp32 values (1.4e-6 at most, tests/test_gpu_kernels.py).  Which kernel
// a LayerNorm takes depends on the model's d and the call site (position table, slabs), never on the batch.
template <int NV, bool RES>
__global__ __launch_bounds__(256) void add_ln_fixed_kernel(const float* x, int64_t ldx, const float* res, int64_t ldres,
                                                           const float* gamma, const float* beta, float eps, float* out,
                                                           bf16_t* outb, int64_t ldo, int rows, int d, int grp, int out_grp_rows,
                                                           int out_row_off) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float4 v[NV], rs[NV], g[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + (lane + 64 * i) * 4);
  if constexpr (RES) {
#pragma unroll
    for (int i = 0; i < NV; ++i) rs[i] = *reinterpret_cast<const float4*>(res + (int64_t)r * ldres + (lane + 64 * i) * 4);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = *reinterpret_cast<const float4*>(gamma + (lane + 64 * i) * 4);
    b[i] = *reinterpret_cast<const float4*>(beta + (lane + 64 * i) * 4);
  }
  if constexpr (RES) {
#pragma unroll
    for (int i = 0; i < NV; ++i) add4(v[i], rs[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = care_wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    q += (a * a + bb * bb) + (c * c + e * e);
  }
  const float var = care_wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + eps);
  const int64_t orow = (int64_t)(r / grp) * out_grp_rows + out_row_off + (r % grp);
  float* o_row = out + orow * ldo;
  bf16_t* ob_row = outb ? outb + orow * ldo : nullptr;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    float4 o;
    o.x = (v[i].x - mean) * rstd * g[i].x + b[i].x;
    o.y = (v[i].y - mean) * rstd * g[i].y + b[i].y;
    o.z = (v[i].z - mean) * rstd * g[i].z + b[i].z;
    o.w = (v[i].w - mean) * rstd * g[i].w + b[i].w;
    *reinterpret_cast<float4*>(o_row + c4 * 4) = o;
    if (ob_row) {
      bf16x4 ob;
      ob[0] = (bf16_t)o.x; ob[1] = (bf16_t)o.y; ob[2] = (bf16_t)o.z; ob[3] = (bf16_t)o.w;
      *reinterpret_cast<bf16x4*>(ob_row + c4 * 4) = ob;
    }
  }
}

__global__ __launch_bounds__(256) void group_mean_kernel(const float* x, int64_t ldx, int in_grp_rows, int in_row_off,
                                                         int grp, float* out, int64_t ldo, int col_off, int groups,
                                                         int d) {
  // one thread per (group, float4 column): consecutive threads read consecutive float4
  const int nv4 = d >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)groups * nv4) return;
  const int g = (int)(idx / nv4), c4 = (int)(idx % nv4);
  const float* p = x + ((int64_t)g * in_grp_rows + in_row_off) * ldx + c4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < grp; ++i) add4(s, *reinterpret_cast<const float4*>(p + (int64_t)i * ldx));
  const float inv = 1.0f / (float)grp;
  s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
  *reinterpret_cast<float4*>(out + (int64_t)g * ldo + col_off + c4 * 4) = s;
}

__global__ __launch_bounds__(256) void concept_finish_kernel(const float* scores, int64_t lds, float* preds,
                                                             int64_t ldp, float* avg, int B, int k) {
  // one wave per clip
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float psum = 0.f;
  for (int c = lane; c < ldp; c += 64) {
    float out = 0.f;
    if (c < k) {
      const float s = scores[(int64_t)b * lds + c];
      const float p = 1.0f / (1.0f + expf(-s));
      psum += p;
      const float raw = logf(fminf(fmaxf(1.0f - p, 1e-12f), 1.0f));
      out = 1.0f - expf(raw);
    }
    preds[(int64_t)b * ldp + c] = out;
  }
  psum = care_wave_sum(psum);
  if (lane == 0) avg[b] = psum / (float)k;
}

// (1024 threads: the 30 embedded rows of a clip are dependent load -> LayerNorm -> store chains, a row per wave; four waves took
// them eight deep - *measured* round 5: 44.6 us per launch at 128 clips, on the critical path of every small concept batch)
__global__ __launch_bounds__(1024) void concept_topk_embed_kernel(const float* preds, int64_t ldp, int k, int topk,
                                                                 const float* word, const float* pos,
                                                                 const float* gamma, const float* beta, float eps,
                                                                 int64_t* labels, float* out, bf16_t* outb,
                                                                 int64_t ldo, int out_grp_rows, int out_row_off,
                                                                 int d) {
  // one workgroup per clip.  Rank by counting: rank(i) = #{j : v[j] > v[i] or (v[j] == v[i] and j < i)};
  // the element of rank r < topk is label r (value desc, index asc) - no iterative selection.
  __shared__ __attribute__((aligned(16))) float sv[1024];
  __shared__ int slab[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int k4 = (k + 3) & ~3;
  for (int i = tid; i < k4; i += 1024) sv[i] = i < k ? preds[(int64_t)b * ldp + i] : -INFINITY;  // pads never outrank
  __syncthreads();
  for (int i = tid; i < k; i += 1024) {
    const float vi = sv[i];
    int rank = 0;
    for (int j = 0; j < k4; j += 4) {  // four comparands per LDS read
      const float4 vj = *reinterpret_cast<const float4*>(sv + j);
      rank += (vj.x > vi) || (vj.x == vi && j < i);
      rank += (vj.y > vi) || (vj.y == vi && j + 1 < i);
      rank += (vj.z > vi) || (vj.z == vi && j + 2 < i);
      rank += (vj.w > vi) || (vj.w == vi && j + 3 < i);
    }
    if (rank < topk) slab[rank] = i;
  }
  __syncthreads();
  if (tid < topk) labels[(int64_t)b * topk + tid] = slab[tid];
  if (!word) return;  // labels only: a model without local guidance (use_attr_flags ..L0) has no concept embeddings
  const int lane = tid & 63, wave = tid >> 6, nv4 = d >> 2;
  for (int j = wave; j < topk; j += 16) {
    const float* w = word + (int64_t)slab[j] * d;
    const float* pp = pos + (int64_t)j * d;
    float4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv4) {
        v[i] = *reinterpret_cast<const float4*>(w + c4 * 4);
        add4(v[i], *reinterpret_cast<const float4*>(pp + c4 * 4));
      }
    }
    const int64_t orow = (int64_t)b * out_grp_rows + out_row_off + j;
    row_layernorm(v, nv4, lane, d, gamma, beta, eps, out + orow * ldo, outb ? outb + orow * ldo : nullptr);
  }
}

__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* tokens, int tok_stride, int tok_off,
                                                       const int32_t* anc, int anc_stride, const float* word,
                                                       const float* pos, int pos0, const float* sem, int sem_div,
                                                       const float* gamma, const float* beta, float eps, float* out,
                                                       bf16_t* outb, int64_t ldo, int rows, int seq, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int s = r / seq, i = r % seq, col = tok_off + i;
  const int trow = anc ? anc[(int64_t)s * anc_stride + col] : s;
  const int tok = tokens[(int64_t)trow * tok_stride + col];
  const float* w = word + (int64_t)tok * d;
  const float* pp = pos + (int64_t)(pos0 + i) * d;
  const float* sm = sem ? sem + (int64_t)(r / sem_div) * d : nullptr;
  const int nv4 = d >> 2;
  float4 v[MAXV];
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {
    const int c4 = lane + 64 * c;
    if (c4 < nv4) {
      v[c] = *reinterpret_cast<const float4*>(w + c4 * 4);
      add4(v[c], *reinterpret_cast<const float4*>(pp + c4 * 4));
      if (sm) add4(v[c], *reinterpret_cast<const float4*>(sm + c4 * 4));
    }
  }
  row_layernorm(v, nv4, lane, d, gamma, beta, eps, out + (int64_t)r * ldo, outb ? outb + (int64_t)r * ldo : nullptr);
}

__global__ __launch_bounds__(256) void greedy_update_kernel(const float* pmax, const int32_t* pidx, const float* psum,
                                                            int parts, int32_t* fed, int fed_stride, float* score,
                                                            int32_t* length, int32_t* finished, int t, int max_steps,
                                                            int eos_id, int rows) {
  // one wave per row: reduce the column-group partials (lowest index wins ties)
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < parts; c += 64) {
    const float v = pmax[(int64_t)r * parts + c];
    const int id = pidx[(int64_t)r * parts + c];
    if (v > best || (v == best && id < bi)) { best = v; bi = id; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < parts; c += 64)
    s += psum[(int64_t)r * parts + c] * expf(pmax[(int64_t)r * parts + c] - best);
  s = care_wave_sum(s);
  if (lane == 0) {
    fed[(int64_t)r * fed_stride + t] = bi;  // always feed the token: frozen rows still run
    if (!finished[r]) {
      score[r] += -logf(s);
      length[r] = t;
      if (bi == eos_id || t >= max_steps) finished[r] = 1;
    }
  }
}

// greedy_update + the NEXT step's embedding: after the token of step t is chosen the same wave embeds
// it (word + position t [+ semantic row]) and LayerNorms it into the activations of step t+1 - one
// launch per step less (a decode step is launch-latency bound at the reference's batch sizes).
__global__ __launch_bounds__(256) void greedy_update_embed_kernel(
    const float* pmax, const int32_t* pidx, const float* psum, int parts, int32_t* fed, int fed_stride, float* score,
    int32_t* length, int32_t* finished, int t, int max_steps, int eos_id, int rows, const float* word, const float* pos,
    const float* sem, int sem_div, const float* gamma, const float* beta, float eps, float* out, bf16_t* outb,
    int64_t ldo, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < parts; c += 64) {
    const float v = pmax[(int64_t)r * parts + c];
    const int id = pidx[(int64_t)r * parts + c];
    if (v > best || (v == best && id < bi)) { best = v; bi = id; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < parts; c += 64)
    s += psum[(int64_t)r * parts + c] * expf(pmax[(int64_t)r * parts + c] - best);
  s = care_wave_sum(s);
  if (lane == 0) {
    fed[(int64_t)r * fed_stride + t] = bi;  // always feed the token: frozen rows still run
    if (!finished[r]) {
      score[r] += -logf(s);
      length[r] = t;
      if (bi == eos_id || t >= max_steps) finished[r] = 1;
    }
  }
  // ---- embedding of that token at position t: the input of decode step t + 1
  const float* w = word + (int64_t)bi * d;
  const float* pp = pos + (int64_t)t * d;
  const float* sm = sem ? sem + (int64_t)(r / sem_div) * d : nullptr;
  const int nv4 = d >> 2;
  float4 v[MAXV];
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {
    const int c4 = lane + 64 * c;
    if (c4 < nv4) {
      v[c] = *reinterpret_cast<const float4*>(w + c4 * 4);
      add4(v[c], *reinterpret_cast<const float4*>(pp + c4 * 4));
      if (sm) add4(v[c], *reinterpret_cast<const float4*>(sm + c4 * 4));
    }
  }
  row_layernorm(v, nv4, lane, d, gamma, beta, eps, out + (int64_t)r * ldo, outb ? outb + (int64_t)r * ldo : nullptr);
}

// teacher-forced scoring from the fused vocabulary partials: per row the arg-max column and the
// log-probability of the label column, log_softmax(x)[label] = x[label] - max - log(sum exp(x - max))
// lab_parts: column groups of plab per row (== parts: the fused label logits of care_gemm_argmax_bf16; 1: the logit of
// the label column computed on its own by care_label_logits)
__global__ __launch_bounds__(256) void score_partials_kernel(const float* pmax, const int32_t* pidx, const float* psum,
                                                             const float* plab, int parts, float* logp, int32_t* pred,
                                                             int rows, int lab_parts) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float best = -INFINITY, lv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < parts; c += 64) {
    const float v = pmax[(int64_t)r * parts + c];
    const int id = pidx[(int64_t)r * parts + c];
    if (v > best || (v == best && id < bi)) { best = v; bi = id; }
    if (c < lab_parts) lv = fmaxf(lv, plab[(int64_t)r * lab_parts + c]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    lv = fmaxf(lv, __shfl_xor(lv, o, 64));
  }
  float s = 0.f;
  for (int c = lane; c < parts; c += 64)
    s += psum[(int64_t)r * parts + c] * expf(pmax[(int64_t)r * parts + c] - best);
  s = care_wave_sum(s);
  if (lane == 0) { logp[r] = (lv - best) - logf(s); pred[r] = bi; }
}

// the same from materialised logits [rows, ld] (fp32 mode, and the checker of the fused form)
__global__ __launch_bounds__(256) void score_logits_kernel(const float* logits, int64_t ld, int V, const int32_t* labels,
                                                           float* logp, int32_t* pred, int rows) {
  __shared__ float sm[4];
  __shared__ int si[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid; c < V; c += 256) {
    const float v = x[c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { sm[wave] = best; si[wave] = bi; }
  __syncthreads();
  best = sm[0]; bi = si[0];
  for (int w = 1; w < 4; ++w)
    if (sm[w] > best || (sm[w] == best && si[w] < bi)) { best = sm[w]; bi = si[w]; }
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf(x[c] - best);
  s = care_wave_sum(s);
  if (lane == 0) sm[wave] = s;
  __syncthreads();
  if (tid == 0) {
    const float tot = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    logp[r] = (x[labels[r]] - best) - logf(tot);
    pred[r] = bi;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int care_add_ln(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* pos,
                           const float* gamma, const float* beta, float eps, float* out, void* out_bf16,
                           int64_t ldo, int rows, int d, int grp, int out_grp_rows, int out_row_off, int nslab,
                           int64_t slab_stride, void* stream) {
  if (!x || ((gamma == nullptr) != (beta == nullptr)) || !out || rows <= 0 || d <= 0 || grp <= 0) return CARE_EINVAL;
  if (d % 4 != 0 || d > 2048 || nslab < 1 || nslab > 16) return CARE_ESHAPE;
  if ((ldx % 4) || (ldo % 4) || (res && (ldres % 4)) || (slab_stride % 4) || !care_aligned16(x) || !care_aligned16(out))
    return CARE_EALIGN;
  if (gamma && !pos && nslab == 1 && (d == 512 || d == 768 || d == 1024 || d == 2048)) {
    bf16_t* ob = reinterpret_cast<bf16_t*>(out_bf16);
    const dim3 grid((rows + 3) / 4);
#define ADD_LN_FIXED(NV)                                                                                                     \
  do {                                                                                                                       \
    if (res) hipLaunchKernelGGL((add_ln_fixed_kernel<NV, true>), grid, dim3(256), 0, ST, x, ldx, res, ldres, gamma, beta, eps, out, ob, ldo, rows, d, grp, out_grp_rows, out_row_off); \
    else hipLaunchKernelGGL((add_ln_fixed_kernel<NV, false>), grid, dim3(256), 0, ST, x, ldx, res, ldres, gamma, beta, eps, out, ob, ldo, rows, d, grp, out_grp_rows, out_row_off);   \
  } while (0)
    if (d == 512) ADD_LN_FIXED(2);
    else if (d == 768) ADD_LN_FIXED(3);
    else if (d == 1024) ADD_LN_FIXED(4);
    else ADD_LN_FIXED(8);
#undef ADD_LN_FIXED
    return care_launch_status();
  }
  hipLaunchKernelGGL(add_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, x, ldx, res, ldres, pos, gamma, beta, eps,
                     out, reinterpret_cast<bf16_t*>(out_bf16), ldo, rows, d, grp, out_grp_rows, out_row_off, nslab,
                     slab_stride);
  return care_launch_status();
}

extern "C" int care_group_mean(const float* x, int64_t ldx, int in_grp_rows, int in_row_off, int grp, float* out,
                               int64_t ldo, int col_off, int groups, int d, void* stream) {
  if (!x || !out || groups <= 0 || d <= 0 || grp <= 0) return CARE_EINVAL;
  if (d % 4 != 0) return CARE_ESHAPE;
  if ((ldx % 4) || (ldo % 4) || (col_off % 4)) return CARE_EALIGN;
  const int64_t n = (int64_t)groups * (d / 4);
  hipLaunchKernelGGL(group_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST, x, ldx, in_grp_rows,
                     in_row_off, grp, out, ldo, col_off, groups, d);
  return care_launch_status();
}

extern "C" int care_concept_finish(const float* scores, int64_t lds, float* preds, int64_t ldp, float* avg, int B,
                                   int k, void* stream) {
  if (!scores || !preds || !avg || B <= 0 || k <= 0 || ldp < k) return CARE_EINVAL;
  hipLaunchKernelGGL(concept_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, scores, lds, preds, ldp, avg, B, k);
  return care_launch_status();
}

extern "C" int care_concept_topk_embed(const float* preds, int64_t ldp, int k, int topk, const float* word,
                                       const float* pos, const float* gamma, const float* beta, float eps,
                                       int64_t* labels, float* out, void* out_bf16, int64_t ldo, int out_grp_rows,
                                       int out_row_off, int B, int d, void* stream) {
  if (!preds || !labels || B <= 0) return CARE_EINVAL;
  if (word && (!pos || !gamma || !beta || !out)) return CARE_EINVAL;  // (word == NULL: the labels alone)
  if (k <= 0 || k > 1024 || topk <= 0 || topk > 64 || topk > k || d % 4 != 0 || d > 2048) return CARE_ESHAPE;
  hipLaunchKernelGGL(concept_topk_embed_kernel, dim3(B), dim3(1024), 0, ST, preds, ldp, k, topk, word, pos, gamma, beta,
                     eps, labels, out, reinterpret_cast<bf16_t*>(out_bf16), ldo, out_grp_rows, out_row_off, d);
  return care_launch_status();
}

extern "C" int care_embed_ln(const int32_t* tokens, int tok_stride, int tok_off, const int32_t* anc, int anc_stride,
                             const float* word, const float* pos, int pos0, const float* sem, int sem_div,
                             const float* gamma, const float* beta, float eps, float* out, void* out_bf16,
                             int64_t ldo, int rows, int seq, int d, void* stream) {
  if (!tokens || !word || !pos || ((gamma == nullptr) != (beta == nullptr)) || !out || rows <= 0 || seq <= 0 || sem_div <= 0) return CARE_EINVAL;
  if (d % 4 != 0 || d > 2048) return CARE_ESHAPE;
  if (ldo % 4) return CARE_EALIGN;
  hipLaunchKernelGGL(embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, tokens, tok_stride, tok_off, anc,
                     anc_stride, word, pos, pos0, sem, sem_div, gamma, beta, eps, out, reinterpret_cast<bf16_t*>(out_bf16),
                     ldo, rows, seq, d);
  return care_launch_status();
}

extern "C" int care_greedy_update(const float* pmax, const int32_t* pidx, const float* psum, int parts, int32_t* fed,
                                  int fed_stride, float* score, int32_t* length, int32_t* finished, int t,
                                  int max_steps, int eos_id, int rows, void* stream) {
  if (!pmax || !pidx || !psum || !fed || !score || !length || !finished || rows <= 0 || parts <= 0) return CARE_EINVAL;
  if (t <= 0 || t >= fed_stride) return CARE_ESHAPE;
  hipLaunchKernelGGL(greedy_update_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, pmax, pidx, psum, parts, fed,
                     fed_stride, score, length, finished, t, max_steps, eos_id, rows);
  return care_launch_status();
}

extern "C" int care_greedy_update_embed(const float* pmax, const int32_t* pidx, const float* psum, int parts, int32_t* fed,
                                        int fed_stride, float* score, int32_t* length, int32_t* finished, int t,
                                        int max_steps, int eos_id, int rows, const float* word, const float* pos,
                                        const float* sem, int sem_div, const float* gamma, const float* beta, float eps,
                                        float* out, void* out_bf16, int64_t ldo, int d, void* stream) {
  if (!pmax || !pidx || !psum || !fed || !score || !length || !finished || rows <= 0 || parts <= 0 || !word || !pos ||
      ((gamma == nullptr) != (beta == nullptr)) || !out)  // (gamma == beta == NULL: no LayerNorm, as care_embed_ln)
    return CARE_EINVAL;
  if (t <= 0 || t >= fed_stride || d <= 0 || d > 64 * 4 * MAXV || (d % 4) || (sem && sem_div <= 0)) return CARE_ESHAPE;
  if ((ldo % 4) || !care_aligned16(word) || !care_aligned16(pos) || !care_aligned16(out) || (sem && !care_aligned16(sem)))
    return CARE_EALIGN;
  hipLaunchKernelGGL(greedy_update_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, pmax, pidx, psum, parts, fed,
                     fed_stride, score, length, finished, t, max_steps, eos_id, rows, word, pos, sem, sem_div, gamma, beta,
                     eps, out, reinterpret_cast<bf16_t*>(out_bf16), ldo, d);
  return care_launch_status();
}

extern "C" int care_score_partials(const float* pmax, const int32_t* pidx, const float* psum, const float* plab,
                                   int parts, float* logp, int32_t* pred, int rows, void* stream) {
  if (!pmax || !pidx || !psum || !plab || !logp || !pred || rows <= 0 || parts <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(score_partials_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, pmax, pidx, psum, plab, parts,
                     logp, pred, rows, parts);
  return care_launch_status();
}

// logit of ONE column per row: out[r] = A[r, :] . W[col[r], :] (bf16 operands, fp32 accumulation) - one wave per row
__global__ __launch_bounds__(256) void label_logits_kernel(const bf16_t* A, int64_t lda, const bf16_t* W, const int32_t* col,
                                                           float* out, int rows, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int c = min(max(col[r], 0), N - 1);
  const bf16_t* a = A + (int64_t)r * lda;
  const bf16_t* w = W + (int64_t)c * K;
  float s = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + k);
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf((float)av[j], (float)wv[j], s);
  }
  s = care_wave_sum(s);
  if (lane == 0) out[r] = s;
}

extern "C" int care_label_logits(const void* A, int64_t lda, const void* W, const int32_t* col, float* out, int rows,
                                 int N, int K, void* stream) {
  if (!A || !W || !col || !out || rows <= 0 || N <= 0 || K <= 0) return CARE_EINVAL;
  if (K % 8 != 0 || lda % 8 != 0 || !care_aligned16(A) || !care_aligned16(W)) return CARE_EALIGN;
  hipLaunchKernelGGL(label_logits_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, reinterpret_cast<const bf16_t*>(A), lda,
                     reinterpret_cast<const bf16_t*>(W), col, out, rows, N, K);
  return care_launch_status();
}

extern "C" int care_score_partials_lab(const float* pmax, const int32_t* pidx, const float* psum, int parts,
                                       const float* lab_logit, float* logp, int32_t* pred, int rows, void* stream) {
  if (!pmax || !pidx || !psum || !lab_logit || !logp || !pred || rows <= 0 || parts <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(score_partials_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, pmax, pidx, psum, lab_logit, parts,
                     logp, pred, rows, 1);
  return care_launch_status();
}

extern "C" int care_score_logits(const float* logits, int64_t ld, int V, const int32_t* labels, float* logp,
                                 int32_t* pred, int rows, void* stream) {
  if (!logits || !labels || !logp || !pred || rows <= 0 || V <= 0) return CARE_EINVAL;
  hipLaunchKernelGGL(score_logits_kernel, dim3(rows), dim3(256), 0, ST, logits, ld, V, labels, logp, pred, rows);
  return care_launch_status();
}

