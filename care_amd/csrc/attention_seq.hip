// attention_seq.hip - scaled-dot-product attention for WHOLE query sequences (Lq <= 32 positions), head dim 64,
// bf16 operands: the teacher-forced forward (models/Framework.py:215-237 -> Decoder/Transformer.py:161-268 ->
// Attention.py:83-131) with Lq = 29, where the decode-step kernel (csrc/attention.hip: one wave per (row, head))
// would re-read a sequence's keys and values once per query position.
//
// One WAVE owns a (sequence, head): its K and V rows are read from memory ONCE for all query positions.
//   * S^T[key][q] = K[key][:] . Q[q][:] / 8 on v_mfma_f32_16x16x32_bf16: both operands are dim-contiguous in
//     memory (a lane's fragment = one 16-byte global load of a K or Q row), keys on the MFMA rows so that a lane
//     holds 4 consecutive keys of ONE query column;
//   * mask / bias exactly like Attention.py:104-111: key-padding mask -> -1e9, THEN the additive per-(head, key)
//     bias; causal keys (j > position) and the padding of the last key tile are excluded;
//   * softmax over the keys of a query = 4 x tiles values in one lane + two xor-shuffles (16, 32);
//   * O^T[dim][q] += V^T[dim][key] . P^T[key][q] on v_mfma_f32_16x16x16_bf16: the B operand is the S^T accumulator
//     layout as it stands (4 keys of a query column per lane), the A operand the head's V rows staged in LDS by
//     LDS-DMA ([key][64 dims], 128-byte rows) and read with the transposing ds_read_b64_tr_b16; the 16-byte
//     chunks of a row are swizzled (chunk ^= ((row >> 1) & 3) << 1, on the DMA SOURCE address and on the read)
//     so that the 8 key rows of a half-wave's read fall on 8 different bank groups;
//   * the context is normalised by the row sum at the end and stored as bf16, 8 bytes per lane.
// Arithmetic per (sequence, head) is ~1 MFLOP - the kernel is bound by the K / V bytes, which is the point.
#include <cstdlib>

#include "care_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct SeqArgs {
  const bf16_t* Q; int64_t ldq;
  const bf16_t* K; const bf16_t* V;
  int64_t kv_batch_stride, kv_row_stride;  // elements
  int seqs_per_kv, nkeys, causal, seq;
  const int32_t* pad_tok; int pad_stride, pad_id;
  const float* bias; int bias_ld;
  bf16_t* ctx; int64_t ldctx;
  int nseq, heads;
};

constexpr int SEQ_WAVES = 4;

// NK16: 16-key tiles (nkeys <= 16 NK16); QT: 16-query tiles (seq <= 16 QT)
template <int NK16, int QT>
__global__ __launch_bounds__(SEQ_WAVES * 64) void attention_seq_kernel(SeqArgs p) {
  constexpr int VROWS = NK16 * 16;
  constexpr int VBYTES = VROWS * 128;
  constexpr int WBYTES = VBYTES + VROWS * 4;   // + the additive per-key term of this (sequence, head)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* vimg = smem + wave * WBYTES;
  float* kterm = reinterpret_cast<float*>(vimg + VBYTES);
  const int fr = lane & 15, fg = lane >> 4;
  const int items = p.nseq * p.heads;
  const int stride = gridDim.x * SEQ_WAVES;

  // transposed-read offsets of the PV phase: 16-lane group fg covers keys 4 fg .. 4 fg + 3 of a key tile;
  // lane 4 q + pp of the group supplies key row q, dims 4 pp .. 4 pp + 3 of the 16-dim block
  const int tq = fr >> 2, tp = fr & 3;
  int toff[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int row = fg * 4 + tq;  // + 16 kb: (row >> 1) & 3 does not change with multiples of 8
    toff[dt] = row * 128 + ((((dt * 2 + (tp >> 1)) ^ (((row >> 1) & 3) << 1)) & 7) << 4) + 8 * (tp & 1);
  }

  for (int item = blockIdx.x * SEQ_WAVES + wave; item < items; item += stride) {
    const int s = item / p.heads, h = item % p.heads;
    const int kvb = s / p.seqs_per_kv;
    const bf16_t* Kb = p.K + (int64_t)kvb * p.kv_batch_stride + h * 64;
    const bf16_t* Vb = p.V + (int64_t)kvb * p.kv_batch_stride + h * 64;
    // every read of the previous item's LDS image is complete
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- V rows -> LDS: instruction i covers key rows 8 i .. 8 i + 7 (clamped), lane = (row, chunk slot)
#pragma unroll
    for (int i = 0; i < NK16 * 2; ++i) {
      const int row = i * 8 + (lane >> 3);
      const int key = min(row, p.nkeys - 1);
      const int chunk = (lane & 7) ^ (((row >> 1) & 3) << 1);
      const unsigned char* g = reinterpret_cast<const unsigned char*>(Vb + (int64_t)key * p.kv_row_stride) + (chunk << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(vimg + i * 1024), 16, 0, 0);
    }
    // ---- per-key additive term: hybrid bias (0 without one), -1e9 on padded keys BEFORE the bias, -inf past nkeys
    for (int j = lane; j < VROWS; j += 64) {
      float v = -INFINITY;
      if (j < p.nkeys) {
        v = p.bias ? p.bias[h * p.bias_ld + j] : 0.0f;
        if (p.pad_tok && p.pad_tok[(int64_t)kvb * p.pad_stride + j] == p.pad_id) v += -1e9f;  // masked_fill, then + bias
      }
      kterm[j] = v;
    }

    // ---- Q fragments (B operand): query qt * 16 + fr, dims ks * 32 + 8 fg ..
    bf16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = min(qt * 16 + fr, p.seq - 1);
      const bf16_t* qrow = p.Q + ((int64_t)s * p.seq + q) * p.ldq + h * 64 + fg * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 32);
    }

    // ---- S^T tiles: lane (query column fr of tile qt, group fg) holds keys 16 kb + 4 fg + r
    f32x4 st[NK16][QT];
#pragma unroll
    for (int kb = 0; kb < NK16; ++kb) {
      const int key = min(kb * 16 + fr, p.nkeys - 1);
      const bf16_t* krow = Kb + (int64_t)key * p.kv_row_stride + fg * 8;
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(krow);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(krow + 32);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        a = care_mfma_16x16x32_h16(k0, qf[qt][0], a, 0, 0, 0);
        a = care_mfma_16x16x32_h16(k1, qf[qt][1], a, 0, 0, 0);
        st[kb][qt] = a;
      }
    }
    // kterm was written by this wave's lanes: make it visible to all of them (a wave is its own workgroup here)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- scale, mask, bias, softmax statistics per query column
    float inv[QT];
    s16x4 pb[NK16][QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int qpos = qt * 16 + fr;
      float m = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < NK16; ++kb) {
        const f32x4 kt4 = *reinterpret_cast<const f32x4*>(kterm + kb * 16 + fg * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb * 16 + fg * 4 + r;
          // a padded key (term <= -1e8): -1e9 REPLACES the score (masked_fill), the bias is already inside the term;
          // past nkeys the term is -inf
          float x = kt4[r] <= -1e8f ? kt4[r] : st[kb][qt][r] * 0.125f + kt4[r];
          if (p.causal && key > qpos) x = -INFINITY;
          st[kb][qt][r] = x;
          m = fmaxf(m, x);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.0f;
#pragma unroll
      for (int kb = 0; kb < NK16; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __expf(st[kb][qt][r] - m);
          l += pr;
          const bf16_t hb = (bf16_t)pr;
          pb[kb][qt][r] = __builtin_bit_cast(short, hb);
        }
      }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      inv[qt] = 1.0f / l;
    }

    // ---- O^T[dim][q] += V^T . P^T
    f32x4 acc[4][QT];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned vbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)vimg;
#pragma unroll
    for (int kb = 0; kb < NK16; ++kb) {
      s16x4 a[4];
      asm volatile(
          "ds_read_b64_tr_b16 %0, %4 offset:%8\n\t"
          "ds_read_b64_tr_b16 %1, %5 offset:%8\n\t"
          "ds_read_b64_tr_b16 %2, %6 offset:%8\n\t"
          "ds_read_b64_tr_b16 %3, %7 offset:%8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
          : "v"(vbase + toff[0]), "v"(vbase + toff[1]), "v"(vbase + toff[2]), "v"(vbase + toff[3]), "n"(kb * 2048)
          : "memory");
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          acc[dt][qt] = care_mfma_16x16x16_h16(a[dt], pb[kb][qt], acc[dt][qt], 0, 0, 0);
    }

    // ---- store: lane (query column fr, group fg) holds dims 16 dt + 4 fg + r
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qt * 16 + fr;
      if (q >= p.seq) continue;
      bf16_t* out = p.ctx + ((int64_t)s * p.seq + q) * p.ldctx + h * 64 + fg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[dt][qt][r] * inv[qt]);
        *reinterpret_cast<bf16x4*>(out + dt * 16) = o;
      }
    }
  }
}

template <int NK16, int QT>
int launch_seq(const SeqArgs& p, hipStream_t st) {
  constexpr int LDS = SEQ_WAVES * (NK16 * 16 * 128 + NK16 * 16 * 4);
  static std::atomic<unsigned long long> ok{0};
  if (LDS > 64 * 1024)
    if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&attention_seq_kernel<NK16, QT>), LDS, ok)) return e;
  const int items = p.nseq * p.heads;
  const int blocks = min((items + SEQ_WAVES - 1) / SEQ_WAVES, 2048);
  hipLaunchKernelGGL((attention_seq_kernel<NK16, QT>), dim3(blocks), dim3(SEQ_WAVES * 64), LDS, st, p);
  return care_launch_status();
}

}  // namespace

extern "C" int care_attention_seq(const void* Q, int64_t ldq, const void* K, const void* V, int64_t kv_batch_stride,
                                  int64_t kv_row_stride, int seqs_per_kv, int nkeys, int causal, int seq,
                                  const int32_t* pad_tok, int pad_stride, int pad_id, const float* bias, int bias_ld,
                                  void* ctx, int64_t ldctx, int nseq, int heads, void* stream) {
  if (!Q || !K || !V || !ctx || nseq <= 0 || heads <= 0 || nkeys <= 0 || seq <= 0 || seqs_per_kv <= 0) return CARE_EINVAL;
  if (nkeys > 128 || seq > 32) return CARE_ESHAPE;
  if ((ldq % 8) || (ldctx % 4) || (kv_batch_stride % 8) || (kv_row_stride % 8) || !care_aligned16(Q) || !care_aligned16(K) ||
      !care_aligned16(V) || (reinterpret_cast<uintptr_t>(ctx) & 7))
    return CARE_EALIGN;
  SeqArgs p{};
  p.Q = reinterpret_cast<const bf16_t*>(Q); p.ldq = ldq;
  p.K = reinterpret_cast<const bf16_t*>(K); p.V = reinterpret_cast<const bf16_t*>(V);
  p.kv_batch_stride = kv_batch_stride; p.kv_row_stride = kv_row_stride;
  p.seqs_per_kv = seqs_per_kv; p.nkeys = nkeys; p.causal = causal; p.seq = seq;
  p.pad_tok = pad_tok; p.pad_stride = pad_stride; p.pad_id = pad_id; p.bias = bias; p.bias_ld = bias_ld;
  p.ctx = reinterpret_cast<bf16_t*>(ctx); p.ldctx = ldctx; p.nseq = nseq; p.heads = heads;
  hipStream_t st = (hipStream_t)stream;
  const int nk16 = (nkeys + 15) / 16;
  if (seq <= 16) {
    if (nk16 <= 2) return launch_seq<2, 1>(p, st);
    if (nk16 <= 6) return launch_seq<6, 1>(p, st);
    return launch_seq<8, 1>(p, st);
  }
  if (nk16 <= 2) return launch_seq<2, 2>(p, st);
  if (nk16 <= 6) return launch_seq<6, 2>(p, st);
  return launch_seq<8, 2>(p, st);
}
