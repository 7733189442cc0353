// attention.hip - single-query-row scaled-dot-product attention, head dim 64 (HBM-bound).
//
// The decode step reads, per (row, head), Lk keys and Lk values of 64 elements once and
// does 2*2*Lk*64 flops on them: ~1 flop/byte, so this is a pure streaming kernel and the
// matrix cores are not involved (north_star: "MFMA only for the dense GEMMs").
//
// Mapping: one 64-lane wave per (query row, head); 4 waves per workgroup.
//   * A wave-wide 16-byte load covers 8 keys x 64 elements (bf16) - lane = (key & 7) * 8 +
//     dim_chunk - i.e. 1 KiB of contiguous-per-key, full-line traffic per instruction.
//   * q.k: 8 FMAs per lane, then a 3-step xor-shuffle over the 8 dim chunks.
//   * softmax over the <= 128 scores a lane group holds in registers: row max / row sum via
//     xor shuffles across the 8 key slots (8,16,32) - no LDS at all.
//   * P.V: each lane accumulates its 8 dims over its key slot, 3 xor-shuffles merge the 8
//     slots, lanes 0-7 store 256 contiguous bytes of context.
//   * NKB (key blocks of 8) is a template parameter: the key loop is fully unrolled, K and V loads
//     are issued ahead of the arithmetic, and no load inside the loop forces a vmcnt(0) drain
//     (the mask / bias values are fetched first, see EXTRA below).
// Masking follows the reference exactly: masked keys get -1e9 (not -inf), the hybrid bias
// is added AFTER the mask (models/components/Attention.py:104-111).
#include <cstdlib>
#include <type_traits>

#include "care_common.h"

// K / V rows are read once per launch (the decoder's cache: 1.9 GB at 32768 rows, re-read every step): loaded
// non-temporal they do not push the weights of the GEMMs around them out of L2.  -DCARE_ATT_NT=0: plain loads.
#ifndef CARE_ATT_NT
#define CARE_ATT_NT 1
#endif

namespace {

// STREAM = the row is read by this wave only (greedy decoding).  With an ancestor table (beam search) the beams of
// a clip read the same cache rows and the plain policy keeps them for one another (*measured* beam 5, 20480 rows:
// self-attention 76 us plain, 100 us non-temporal).
template <bool STREAM, typename T>
__device__ __forceinline__ T att_stream_load(const T* ptr) {
  if constexpr (STREAM && CARE_ATT_NT && std::is_same<T, bf16x8>::value) return __builtin_nontemporal_load(ptr);
  else return *ptr;
}

struct AttnArgs {
  const float* Q; int64_t ldq;
  const void* K; const void* V;
  int64_t kv_batch_stride, kv_row_stride;
  int rows_per_kv;
  const int32_t* anc; int anc_stride;
  int nkeys, causal, seq, causal_off;
  const int32_t* pad_tok; int pad_stride, pad_id;
  const float* bias; int bias_ld;
  void* ctx; int64_t ldctx; int ctx_bf16;
  int rows, heads;
};

// EXTRA = the launch has a key-padding mask and/or an additive bias.  Their per-key values are
// fetched BEFORE the K/V loads (vmcnt retires in order, so a load issued inside the score loop
// would wait for every K/V byte and then cost one memory round trip per key block): lane
// (slot, chunk) fetches the values of key blocks `chunk` and `chunk + 8` of its slot, and the score
// loop pulls them across the 8 chunk lanes with one shuffle each.
template <typename KT, int NKB, bool ANC, bool EXTRA>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= p.rows * p.heads) return;
  const int r = item / p.heads, h = item % p.heads;
  const int slot = lane >> 3, chunk = lane & 7;

  int nk = p.nkeys;
  if (p.causal) nk = min(nk, (r % p.seq) + 1 + p.causal_off);
  const int kvb_default = r / p.rows_per_kv;

  constexpr int NX = NKB > 8 ? 2 : 1;
  float xbias[NX];
  int xpad[NX];
  if constexpr (EXTRA) {
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      const int j = (x * 8 + chunk) * 8 + slot;
      const int jc = j < nk ? j : 0;
      xbias[x] = p.bias ? p.bias[h * p.bias_ld + jc] : 0.f;
      int tok = p.pad_id + 1;
      if (p.pad_tok) {
        const int kvb = ANC ? p.anc[(int64_t)r * p.anc_stride + jc] : kvb_default;
        tok = p.pad_tok[(int64_t)kvb * p.pad_stride + jc];
      }
      xpad[x] = tok == p.pad_id;
    }
  }

  float q[8];
  care_load8(p.Q + (int64_t)r * p.ldq + h * 64 + chunk * 8, q);

  const KT* Kb = reinterpret_cast<const KT*>(p.K) + h * 64 + chunk * 8;
  const KT* Vb = reinterpret_cast<const KT*>(p.V) + h * 64 + chunk * 8;
  static_assert(sizeof(KT) == 2 || sizeof(KT) == 4, "bf16 or fp32 K/V");

  // ---- issue every K and V load of this (row, head) before the first use: 2*NKB 16-byte
  // loads per lane in flight, so the HBM latency is paid once, not once for K and once for V
  using Frag = typename std::conditional<sizeof(KT) == 2, bf16x8, float4>::type;
  constexpr int FPK = sizeof(KT) == 2 ? 1 : 2;  // 16-byte pieces per 8 elements
  Frag kf[NKB][FPK], vf[NKB][FPK];
  // ANC (beam search): key j of row r lives in the cache row anc[r][j]; otherwise every key of
  // the row comes from one K/V block and no per-key index is kept in registers
  int kvbs[ANC ? NKB : 1];
  if constexpr (ANC) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int j = kb * 8 + slot;
      kvbs[kb] = p.anc[(int64_t)r * p.anc_stride + (j < nk ? j : 0)];
    }
  } else {
    kvbs[0] = kvb_default;
  }
  auto kv_off = [&](int kb) {
    const int j = kb * 8 + slot;
    return (int64_t)kvbs[ANC ? kb : 0] * p.kv_batch_stride + (int64_t)(j < nk ? j : 0) * p.kv_row_stride;
  };
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int64_t off = kv_off(kb);
#pragma unroll
    for (int f = 0; f < FPK; ++f) kf[kb][f] = att_stream_load<!ANC>(reinterpret_cast<const Frag*>(Kb + off + f * 4));
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int64_t off = kv_off(kb);
#pragma unroll
    for (int f = 0; f < FPK; ++f) vf[kb][f] = att_stream_load<!ANC>(reinterpret_cast<const Frag*>(Vb + off + f * 4));
  }
#ifdef CARE_ATT_PIN
  // Ablation: pin every load above this line (all K/V bytes of the wave in flight at once, ~40 more
  // VGPRs, one wave per SIMD fewer).  Measured SLOWER than letting the scheduler sink some loads
  // towards their use: Lk = 114 bf16 677 us vs 649 us per launch at 16384 rows (same box).
  __builtin_amdgcn_sched_barrier(0);
#endif
  auto unpack = [](const Frag (&fr)[FPK], float (&o)[8]) {
    if constexpr (sizeof(KT) == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (float)fr[0][i];
    } else {
      o[0] = fr[0].x; o[1] = fr[0].y; o[2] = fr[0].z; o[3] = fr[0].w;
      o[4] = fr[1].x; o[5] = fr[1].y; o[6] = fr[1].z; o[7] = fr[1].w;
    }
  };

  // ---- scores
  float s[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int j = kb * 8 + slot;
    float kv[8];
    unpack(kf[kb], kv);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = fmaf(q[i], kv[i], d);
    // 8-lane sum through DPP (quad_perm, quad_perm, row_half_mirror): the xor-shuffle version is
    // three dependent ds_bpermute round trips per key block
    d += care_dpp_x1(d);
    d += care_dpp_x2(d);
    d += care_dpp_m8(d);
    d *= 0.125f;  // 1/sqrt(64), exact
    if constexpr (EXTRA) {
      const int src = (lane & 56) | (kb & 7);
      const int padded = __shfl(xpad[kb >> 3], src, 64);
      const float bj = __shfl(xbias[kb >> 3], src, 64);
      if (padded) d = -1e9f;  // masked_fill(-1e9) first, the hybrid bias is added after it
      d += bj;
    }
    s[kb] = j < nk ? d : -INFINITY;
  }

  // ---- softmax over all keys (each score is replicated on the 8 chunk lanes of its slot)
  float m = s[0];
#pragma unroll
  for (int kb = 1; kb < NKB; ++kb) m = fmaxf(m, s[kb]);
  m = fmaxf(m, care_dpp_x8(m));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    s[kb] = expf(s[kb] - m);  // -inf -> 0 for the padding slots
    sum += s[kb];
  }
  sum += care_dpp_x8(sum);
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;

  // ---- context
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    float vv[8];
    unpack(vf[kb], vv);
    const float pw = s[kb] * inv;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(pw, vv[i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += care_dpp_x8(acc[i]);
    acc[i] += __shfl_xor(acc[i], 16, 64);
    acc[i] += __shfl_xor(acc[i], 32, 64);
  }
  if (slot == 0) {
    const int64_t o = (int64_t)r * p.ldctx + h * 64 + chunk * 8;
    if (p.ctx_bf16) {  // context feeds a bf16 GEMM only: store it rounded, half the bytes
      bf16x8 ob;
#pragma unroll
      for (int i = 0; i < 8; ++i) ob[i] = (bf16_t)acc[i];
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.ctx) + o) = ob;
    } else {
      float* of = reinterpret_cast<float*>(p.ctx) + o;
      *reinterpret_cast<float4*>(of) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(of + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-per-wave variant for short key ranges (the decoder's self-attention: <= 32 keys, bf16 cache,
// 8 heads): lane = (head, 8-dim chunk), so one wave-wide 16-byte load is a key's WHOLE K (or V) row
// - 1 KiB, contiguous - and the keys are serial inside a lane: the q.k reduction stays within 8
// lanes (three DPP steps), the softmax and the P.V sum need no cross-lane traffic at all, the
// context row leaves as one contiguous 1-KiB (bf16) store, and exactly ceil(nk / 4) * 4 keys are
// read.  The (row, head)-per-wave kernel above reads blocks of 8 keys per 8 lanes and merges the 8
// key slots by xor-shuffles: at short prefixes it runs at ~2.5 TB/s (118 us per launch for steps
// 1-8 at 32768 rows against a 50 us HBM time).  *Measured* over the 29 steps of a pass at 32768
// rows, same box: 241-250 us per launch with the kernel above, 227-244 us with this one - a few
// per cent; short launches are bound by wave start-up and latency rather than by the access shape.
template <int NK4, bool ANC>
__global__ __launch_bounds__(256) void attention_row_kernel(AttnArgs p) {
  constexpr int NK = NK4 * 4;
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r >= p.rows) return;
  const int h = lane >> 3;
  int nk = p.nkeys;
  if (p.causal) nk = min(nk, (r % p.seq) + 1 + p.causal_off);
  const int kvb_default = r / p.rows_per_kv;

  // per-key cache block (beam search: the ancestor that holds position j), mask and bias terms
  int kvb[NK];
  float add[NK];  // -1e9 replaces the score of a padded key (flag in `padded`), the bias is added after
  bool padded[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const int jj = j < nk ? j : 0;
    kvb[j] = ANC ? p.anc[(int64_t)r * p.anc_stride + jj] : kvb_default;
    padded[j] = p.pad_tok ? p.pad_tok[(int64_t)kvb[j] * p.pad_stride + jj] == p.pad_id : false;
    add[j] = p.bias ? p.bias[h * p.bias_ld + jj] : 0.f;
  }
  float q[8];
  care_load8(p.Q + (int64_t)r * p.ldq + lane * 8, q);
  const bf16_t* Kb = reinterpret_cast<const bf16_t*>(p.K) + lane * 8;
  const bf16_t* Vb = reinterpret_cast<const bf16_t*>(p.V) + lane * 8;
  bf16x8 kf[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j)
    kf[j] = att_stream_load<!ANC>(reinterpret_cast<const bf16x8*>(Kb + (int64_t)kvb[j] * p.kv_batch_stride + (int64_t)(j < nk ? j : 0) * p.kv_row_stride));
  __builtin_amdgcn_sched_barrier(0);  // all K rows of the wave in flight before the first dot product

  float s[NK];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = fmaf(q[i], (float)kf[j][i], d);
    d += care_dpp_x1(d);
    d += care_dpp_x2(d);
    d += care_dpp_m8(d);
    d *= 0.125f;  // 1/sqrt(64), exact
    if (padded[j]) d = -1e9f;  // masked_fill(-1e9) first, the hybrid bias is added after it
    d += add[j];
    s[j] = j < nk ? d : -INFINITY;
    m = fmaxf(m, s[j]);
  }
  // the V rows travel while the softmax is computed
  bf16x8 vf[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j)
    vf[j] = att_stream_load<!ANC>(reinterpret_cast<const bf16x8*>(Vb + (int64_t)kvb[j] * p.kv_batch_stride + (int64_t)(j < nk ? j : 0) * p.kv_row_stride));
  __builtin_amdgcn_sched_barrier(0);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    s[j] = expf(s[j] - m);  // -inf -> 0 for the keys past nk
    sum += s[j];
  }
  const float inv = 1.0f / sum;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const float pw = s[j] * inv;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(pw, (float)vf[j][i], acc[i]);
  }
  const int64_t o = (int64_t)r * p.ldctx + lane * 8;
  if (p.ctx_bf16) {
    bf16x8 ob;
#pragma unroll
    for (int i = 0; i < 8; ++i) ob[i] = (bf16_t)acc[i];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.ctx) + o) = ob;
  } else {
    float* of = reinterpret_cast<float*>(p.ctx) + o;
    *reinterpret_cast<float4*>(of) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(of + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

template <bool ANC>
int launch_attention_row(const AttnArgs& p, hipStream_t st) {
  const dim3 grid((p.rows + 3) / 4), block(256);
  switch ((p.nkeys + 3) / 4) {
    case 1: hipLaunchKernelGGL((attention_row_kernel<1, ANC>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((attention_row_kernel<2, ANC>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((attention_row_kernel<3, ANC>), grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL((attention_row_kernel<4, ANC>), grid, block, 0, st, p); break;
    case 5: hipLaunchKernelGGL((attention_row_kernel<5, ANC>), grid, block, 0, st, p); break;
    case 6: hipLaunchKernelGGL((attention_row_kernel<6, ANC>), grid, block, 0, st, p); break;
    case 7: hipLaunchKernelGGL((attention_row_kernel<7, ANC>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((attention_row_kernel<8, ANC>), grid, block, 0, st, p); break;
  }
  return care_launch_status();
}

template <typename KT>
int launch_attention(const AttnArgs& p, hipStream_t st) {
  const int items = p.rows * p.heads;
  const dim3 grid((items + 3) / 4), block(256);
  // key-block count specialised to the prefix length: a step-1 self-attention loads 8 key slots, not 32
  const bool extra = p.pad_tok || p.bias;
#define CARE_ATT_LAUNCH2(NKB, ANC)                                                               \
  do {                                                                                            \
    if (extra) hipLaunchKernelGGL((attention_kernel<KT, NKB, ANC, true>), grid, block, 0, st, p); \
    else hipLaunchKernelGGL((attention_kernel<KT, NKB, ANC, false>), grid, block, 0, st, p);      \
  } while (0)
#define CARE_ATT_LAUNCH(NKB)            \
  do {                                  \
    if (p.anc) CARE_ATT_LAUNCH2(NKB, true); \
    else CARE_ATT_LAUNCH2(NKB, false);  \
  } while (0)
  if (p.nkeys <= 8) CARE_ATT_LAUNCH(1);
  else if (p.nkeys <= 16) CARE_ATT_LAUNCH(2);
  else if (p.nkeys <= 24) CARE_ATT_LAUNCH(3);
  else if (p.nkeys <= 32) CARE_ATT_LAUNCH(4);
  else if (p.nkeys <= 88) CARE_ATT_LAUNCH(11);
  else if (p.nkeys <= 104) CARE_ATT_LAUNCH(13);
  else if (p.nkeys <= 120) CARE_ATT_LAUNCH(15);
  else CARE_ATT_LAUNCH(16);
#undef CARE_ATT_LAUNCH
#undef CARE_ATT_LAUNCH2
  return care_launch_status();
}

}  // namespace

extern "C" int care_attention(const float* Q, int64_t ldq, const void* K, const void* V, int kv_dtype,
                              int64_t kv_batch_stride, int64_t kv_row_stride, int rows_per_kv, const int32_t* anc,
                              int anc_stride, int nkeys, int causal, int seq, int causal_off, const int32_t* pad_tok,
                              int pad_stride, int pad_id, const float* bias, int bias_ld, void* ctx, int64_t ldctx,
                              int ctx_dtype, int rows, int heads, void* stream) {
  if (!Q || !K || !V || !ctx || rows <= 0 || heads <= 0 || nkeys <= 0 || rows_per_kv <= 0 || seq <= 0)
    return CARE_EINVAL;
  if (nkeys > 128) return CARE_ESHAPE;
  if (kv_dtype != CARE_F32 && kv_dtype != CARE_BF16) return CARE_EDTYPE;
  if (ctx_dtype != CARE_F32 && ctx_dtype != CARE_BF16) return CARE_EDTYPE;
  if ((ldq % 4) || (ldctx % 8) || (kv_batch_stride % 8) || (kv_row_stride % 8) || !care_aligned16(Q) ||
      !care_aligned16(K) || !care_aligned16(V) || !care_aligned16(ctx))
    return CARE_EALIGN;
  AttnArgs p{};
  p.Q = Q; p.ldq = ldq; p.K = K; p.V = V;
  p.kv_batch_stride = kv_batch_stride; p.kv_row_stride = kv_row_stride; p.rows_per_kv = rows_per_kv;
  p.anc = anc; p.anc_stride = anc_stride;
  p.nkeys = nkeys; p.causal = causal; p.seq = seq; p.causal_off = causal_off;
  p.pad_tok = pad_tok; p.pad_stride = pad_stride; p.pad_id = pad_id;
  p.bias = bias; p.bias_ld = bias_ld; p.ctx = ctx; p.ldctx = ldctx; p.ctx_bf16 = ctx_dtype == CARE_BF16;
  p.rows = rows; p.heads = heads;
  hipStream_t st = (hipStream_t)stream;
  static int rowk = -1;  // CARE_ATT_ROW=0 keeps the (row, head)-per-wave kernel for every shape (A/B tuning)
  if (rowk < 0) { const char* e = getenv("CARE_ATT_ROW"); rowk = e ? atoi(e) : 1; }
  if (rowk && kv_dtype == CARE_BF16 && heads == 8 && nkeys <= 32 && (ldq % 8) == 0)
    return anc ? launch_attention_row<true>(p, st) : launch_attention_row<false>(p, st);
  return kv_dtype == CARE_BF16 ? launch_attention<bf16_t>(p, st) : launch_attention<float>(p, st);
}

// ---------------------------------------------------------------------------------------------
// care_attention_probs: the softmax probabilities themselves, [rows, heads, nkeys] fp32 - the
// `attention_probs` the reference's decoder returns next to its hidden states
// (models/Decoder/Transformer.py:239-252: all_intra_attentions / all_inter_attentions /
// attention_probs; read by regularisers and notebooks, never by decoding).  The fused kernels above
// keep them in registers; this one is the off-path companion for the teacher-forced forward: one wave
// per (row, head), lane j owns keys j and j + 64, same score arithmetic (scale, -1e9 mask, bias after
// the mask).
namespace {
template <typename KT>
__global__ __launch_bounds__(256) void attention_probs_kernel(AttnArgs p, float* probs) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= p.rows * p.heads) return;
  const int r = item / p.heads, h = item % p.heads;
  int nk = p.nkeys;
  if (p.causal) nk = min(nk, (r % p.seq) + 1 + p.causal_off);
  const int kvb = r / p.rows_per_kv;
  const float* q = p.Q + (int64_t)r * p.ldq + h * 64;
  const KT* Kb = reinterpret_cast<const KT*>(p.K) + (int64_t)kvb * p.kv_batch_stride + h * 64;
  float s[2];
  float m = -INFINITY;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = lane + 64 * u;
    s[u] = -INFINITY;
    if (j < nk) {
      const KT* kr = Kb + (int64_t)j * p.kv_row_stride;
      float d = 0.f;
      for (int i = 0; i < 64; ++i) d = fmaf(q[i], (float)kr[i], d);
      d *= 0.125f;
      if (p.pad_tok && p.pad_tok[(int64_t)kvb * p.pad_stride + j] == p.pad_id) d = -1e9f;
      if (p.bias) d += p.bias[h * p.bias_ld + j];
      s[u] = d;
    }
    m = fmaxf(m, s[u]);
  }
  m = care_wave_max(m);
  float e0 = s[0] == -INFINITY ? 0.f : expf(s[0] - m), e1 = s[1] == -INFINITY ? 0.f : expf(s[1] - m);
  const float inv = 1.0f / care_wave_sum(e0 + e1);
  float* o = probs + ((int64_t)r * p.heads + h) * p.nkeys;
  if (lane < p.nkeys) o[lane] = e0 * inv;        // keys past the causal bound get exactly 0, like the reference's
  if (lane + 64 < p.nkeys) o[lane + 64] = e1 * inv;  // softmax of -1e9-masked scores would in fp32
}
// The same probabilities for whole query sequences of <= 32 positions over fp32 keys (the training forward, the auxiliary
// entries of the teacher-forced decoder): a workgroup per (sequence, head), S^T[key][query] = K_h Q_h^T on the exact-f32 matrix
// cores (v_mfma_f32_16x16x4_f32: an fp32 fma chain per output, like the loop above), the score rows in LDS, one wave per row
// for scale / mask / bias / softmax - the same arithmetic per element.  (The per-(row, head) wave above reads its keys through 64
// different cache lines per instruction: 0.34 ms per launch at 512 clips x 29 positions x 114 keys, rocprofv3 round 6.)
__global__ __launch_bounds__(256) void attention_probs_seq_kernel(AttnArgs p, float* probs) {
  __shared__ float sQ[32][68], sS[32][132];
  const int sq = blockIdx.x / p.heads, h = blockIdx.x % p.heads;          // sequence = rows sq * seq .. + seq - 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  const int r0 = sq * p.seq, kvb = r0 / p.rows_per_kv;                    // (rows_per_kv % seq == 0: one key block per sequence)
  const float* Kb = reinterpret_cast<const float*>(p.K) + (int64_t)kvb * p.kv_batch_stride + h * 64;
  const int KT = (p.nkeys + 15) >> 4, QT = (p.seq + 15) >> 4;
  for (int i = tid; i < 32 * 64; i += 256) {
    const int q = i >> 6, e = i & 63;
    sQ[q][e] = q < p.seq ? p.Q[(int64_t)(r0 + q) * p.ldq + h * 64 + e] : 0.f;
  }
  __syncthreads();
  for (int ti = wave; ti < KT * QT; ti += 4) {
    const int kt = ti / QT, qt = ti - kt * QT;
    const int key = kt * 16 + l16;
    const float* krow = Kb + (int64_t)min(key, p.nkeys - 1) * p.kv_row_stride + kg;
    float fa[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) fa[ks] = krow[4 * ks];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], sQ[qt * 16 + l16][4 * ks + kg], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) sS[qt * 16 + l16][kt * 16 + 4 * kg + e] = acc[e];
  }
  __syncthreads();
  for (int i = wave; i < p.seq; i += 4) {
    int nk = p.nkeys;
    if (p.causal) nk = min(nk, i + 1 + p.causal_off);
    float sc[2];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u;
      sc[u] = -INFINITY;
      if (j < nk) {
        float d = sS[i][j] * 0.125f;
        if (p.pad_tok && p.pad_tok[(int64_t)kvb * p.pad_stride + j] == p.pad_id) d = -1e9f;
        if (p.bias) d += p.bias[h * p.bias_ld + j];
        sc[u] = d;
      }
      m = fmaxf(m, sc[u]);
    }
    m = care_wave_max(m);
    const float e0 = sc[0] == -INFINITY ? 0.f : expf(sc[0] - m), e1 = sc[1] == -INFINITY ? 0.f : expf(sc[1] - m);
    const float inv = 1.0f / care_wave_sum(e0 + e1);
    float* o = probs + ((int64_t)(r0 + i) * p.heads + h) * p.nkeys;
    if (lane < p.nkeys) o[lane] = e0 * inv;
    if (lane + 64 < p.nkeys) o[lane + 64] = e1 * inv;
  }
}
}  // namespace

extern "C" int care_attention_probs(const float* Q, int64_t ldq, const void* K, int kv_dtype, int64_t kv_batch_stride,
                                    int64_t kv_row_stride, int rows_per_kv, int nkeys, int causal, int seq,
                                    const int32_t* pad_tok, int pad_stride, int pad_id, const float* bias, int bias_ld,
                                    float* probs, int rows, int heads, void* stream) {
  if (!Q || !K || !probs || rows <= 0 || heads <= 0 || nkeys <= 0 || nkeys > 128 || rows_per_kv <= 0) return CARE_EINVAL;
  if (kv_dtype != CARE_F32 && kv_dtype != CARE_BF16) return CARE_EDTYPE;
  if (causal && seq <= 0) return CARE_EINVAL;
  AttnArgs p{};
  p.Q = Q; p.ldq = ldq; p.K = K; p.kv_batch_stride = kv_batch_stride; p.kv_row_stride = kv_row_stride;
  p.rows_per_kv = rows_per_kv; p.nkeys = nkeys; p.causal = causal; p.seq = seq; p.causal_off = 0;
  p.pad_tok = pad_tok; p.pad_stride = pad_stride; p.pad_id = pad_id; p.bias = bias; p.bias_ld = bias_ld;
  p.rows = rows; p.heads = heads;
  // whole sequences of <= 32 positions over fp32 keys, one key block per sequence: the matrix-core form
  static const bool row_form = [] { const char* e = getenv("CARE_PROBS_SEQ"); return e && atoi(e) == 0; }();
  if (kv_dtype == CARE_F32 && seq >= 8 && seq <= 32 && rows % seq == 0 && rows_per_kv % seq == 0 && !row_form) {
    hipLaunchKernelGGL(attention_probs_seq_kernel, dim3((rows / seq) * heads), dim3(256), 0, (hipStream_t)stream, p, probs);
    return care_launch_status();
  }
  const dim3 grid((rows * heads + 3) / 4), block(256);
  if (kv_dtype == CARE_BF16) hipLaunchKernelGGL(attention_probs_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p, probs);
  else hipLaunchKernelGGL(attention_probs_kernel<float>, grid, block, 0, (hipStream_t)stream, p, probs);
  return care_launch_status();
}
