// attention_latent.hip - cross-attention with W_k / W_v absorbed into the query / context side.
//
// The reference projects the memory to per-head keys and values (Attention.py:63-67) and every
// decoder step reads both: 2 * Lk * d elements per row and step - the dominant HBM traffic of the
// whole path.  Algebraically
//     scores[h][j] = q_h . (W_k,h mem_j + b_k,h) = (W_k,h^T q_h) . mem_j + const_h
//     ctx_h        = sum_j p[h][j] (W_v,h mem_j + b_v,h) = W_v,h (sum_j p[h][j] mem_j) + b_v,h
// (const_h is the same for every key of a head, so the softmax does not see it), so K and V can
// both be read from ONE bf16 copy of the memory row: Lk * d elements per row and step plus the
// expanded query  qt[h] = W_k,h^T q_h / sqrt(d_h)  (H * d) in and the latent context
// ct[h] = sum_j p[h][j] mem_j  (H * d) out.  For Lk = 84, d = 512, H = 8: 102 KB instead of 175 KB.
//
// This kernel is the middle part: ct = softmax(qt . mem^T + bias) . mem per row, heads batched on
// the MFMA N axis.  One WAVE owns a row at a time (no workgroup barrier anywhere):
//   * the row's memory streams through a wave-private LDS ring in chunks of 16 keys (16 KiB), one
//     LDS-DMA instruction per key row (1 KiB, full lines), one chunk ahead of the arithmetic;
//   * S^T[key][head]  = mem_chunk [16 x 512] . qt^T [512 x 16]: 16 MFMA 16x16x32, A fragments read
//     row-wise from LDS (ds_read_b128), B = qt fragments resident in 64 VGPRs for the whole row;
//   * online softmax per head: a lane holds 4 keys of one head, max / sum by two xor-shuffles;
//   * ct^T[dim][head] += mem_chunk^T [512 x 16] . P^T [16 x 16]: 32 MFMA 16x16x16 whose A operand is
//     the SAME LDS image read with the transposing ds_read_b64_tr_b16 and whose B operand is the
//     S^T accumulator layout as it stands (no shuffle, no second copy of the tile);
//   * the 16-byte-chunk swizzle chunk ^= ((row & 7) << 1) | (row >> 3) (applied to the per-lane DMA
//     SOURCE address, the LDS image itself is lane-linear) makes the transposed reads conflict-free
//     and leaves the row reads 2-way.
// bf16 mode only (the fp32 parity mode keeps projected K/V and csrc/attention.hip).
//
// What bounds it (measured, 16384 rows, Lk = 84, 1.68 GB per launch): the DMA ring alone (no
// arithmetic, no stores) streams the memory at 6.2 TB/s; with q~ loads and c~ stores 296 us; the
// full kernel 316-334 us = 5.0-5.3 TB/s.  Removing the per-chunk accumulator copies (fixed softmax
// reference, slow-path redo) and coalescing the stores through LDS changed NOTHING on the same box:
// the arithmetic sits in slack.  The limit is bytes in flight - the 128 KiB of LDS ring per CU is
// ~27 MB chip-wide, which at the ~5 us loaded latency is ~5 TB/s (the K/V kernel buffers 264 KiB
// per CU in registers and reaches 6.1).  3 waves x 3 slots is slower (fewer waves), a fifth wave
// does not fit the 160 KiB.
#include "care_common.h"

#ifndef CARE_LAT_DBG
#define CARE_LAT_DBG 0  // ablation builds (tools/latent_probe.py): 1 no ct stores, 2 no qt loads, 4 no arithmetic
#endif

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int LAT_D = 512;
constexpr int CH_KEYS = 16;
constexpr int CH_BYTES = CH_KEYS * LAT_D * 2;  // 16 KiB

struct LatArgs {
  const bf16_t* qt; int64_t ldq;      // [rows][heads][512], row stride in elements
  const bf16_t* mem; int64_t mem_bs, mem_rs;  // memory block of clip b at mem + b*mem_bs, key stride mem_rs
  int rows_per_kv, nkeys;
  const float* bias; int bias_ld;     // [heads][nkeys] or null
  bf16_t* ct; int64_t ldc;            // [rows][heads][512]
  int rows, heads;
};

__device__ __forceinline__ int lat_swz(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }

template <int WAVES, int NSLOT>
__global__ __launch_bounds__(WAVES * 64, 1) void attention_latent_kernel(LatArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* ring = smem + wave * (NSLOT * CH_BYTES);
  const int fr = lane & 15, fg = lane >> 4;
  const int total_waves = gridDim.x * WAVES;
  const int gw = blockIdx.x * WAVES + wave;
  const int nch = (p.nkeys + CH_KEYS - 1) / CH_KEYS;

  // one chunk = 16 LDS-DMA instructions; LDS row i of the chunk holds key c*16+i (clamped), its
  // 16-byte chunk k stored at position k ^ lat_swz(i)
  auto stage = [&](int row, int c, int slot) {
    const bf16_t* base = p.mem + (int64_t)(row / p.rows_per_kv) * p.mem_bs;
#pragma unroll
    for (int i = 0; i < CH_KEYS; ++i) {
      const int key = min(c * CH_KEYS + i, p.nkeys - 1);
      const unsigned char* g = reinterpret_cast<const unsigned char*>(base + (int64_t)key * p.mem_rs) +
                               ((lane ^ lat_swz(i)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ring + slot * CH_BYTES + i * 1024),
                                       16, 0, 0);
    }
  };

  // per-lane LDS offsets.  Row read (S phase): key row fr, global 16-byte chunk ks*4 + fg.
  int roff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) roff[r] = fr * 1024 + ((((r * 4 + fg) ^ lat_swz(fr)) & 15) << 4);
  // Transposed read (PV phase): 16-lane group fg covers keys fg*4..fg*4+3; lane 4q+pp of the group
  // supplies row q, columns 4pp..4pp+3 of the 16-dim block mt -> chunk mt*2 + (pp >> 1), half pp & 1.
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int trow = fg * 4 + tq;
  int toff[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
    toff[m] = trow * 1024 + ((((m * 2 + (tp >> 1)) ^ lat_swz(trow)) & 15) << 4) + 8 * (tp & 1);

  const int headc = min(fr, p.heads - 1);

  // additive per-(head, key) term, the same for every row: hybrid bias (0 without one) for valid
  // keys, -inf for the padding keys of the last chunk.  Staged in LDS once per block - a global
  // load inside the chunk loop would be YOUNGER than the next chunk's DMAs and drain them.
  float* sbias = reinterpret_cast<float*>(smem + WAVES * NSLOT * CH_BYTES);  // [16][128]
  for (int i = threadIdx.x; i < 16 * 128; i += WAVES * 64) {
    const int h = min(i >> 7, p.heads - 1), key = i & 127;
    sbias[i] = key < p.nkeys ? (p.bias ? p.bias[h * p.bias_ld + key] : 0.f) : -INFINITY;
  }
  __syncthreads();
  int t = 0;  // chunks consumed so far by this wave -> ring slot
  if (gw < p.rows) stage(gw, 0, 0);

  for (int row = gw; row < p.rows; row += total_waves) {
    // expanded query of this row: B operand of the S MFMAs, resident for all chunks
    bf16x8 qf[16];
    const bf16_t* qrow = p.qt + (int64_t)row * p.ldq + headc * LAT_D + fg * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (CARE_LAT_DBG & 2) { qf[ks] = bf16x8{}; asm volatile("" : "+v"(qf[ks])); }
      else qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 32);
    }

    f32x4 acc[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_ref = -INFINITY, l_part = 0.f;

    for (int c = 0; c < nch; ++c, ++t) {
      const int slot = t % NSLOT;
      // next chunk (of this row or of the wave's next row) into the slot consumed one iteration ago
      const bool more_here = c + 1 < nch;
      const int nrow = more_here ? row : row + total_waves;
      const bool have_next = nrow < p.rows;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS reads of that slot are done
      if (have_next) {
        stage(nrow, more_here ? c + 1 : 0, (t + 1) % NSLOT);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // everything but the 16 newest DMAs
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* sb = ring + slot * CH_BYTES;
      if (CARE_LAT_DBG & 4) continue;

      // ---- S^T[key][head] for the 16 keys of the chunk
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + roff[ks & 3] + (ks >> 2) * 256);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], s, 0, 0, 0);
      }
      // lane (head fr, group fg) holds keys c*16 + fg*4 + r
      s += *reinterpret_cast<const f32x4*>(sbias + fr * 128 + c * CH_KEYS + fg * 4);

      // ---- online softmax over the keys of each head, with a LAZY reference maximum: the
      // exponentials are taken against m_ref, which only moves (and only then are the 128
      // accumulator registers rescaled) when a chunk's maximum exceeds it by more than 16 -
      // softmax is shift-invariant, exp(16) ~ 9e6 is harmless in fp32/bf16, and in the common case
      // the accumulators are touched by nothing but the MFMAs (they stay in AGPRs).
      float cm = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      if (__any(cm > m_ref + 16.0f)) {  // wave-uniform; always taken on a row's first chunk (m_ref = -inf)
        const float m_new = fmaxf(m_ref, cm);
        const float alpha = __expf(m_ref - m_new);  // exp(-inf) = 0 on the first chunk
        m_ref = m_new;
        l_part *= alpha;
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m] *= alpha;
      }
      s16x4 pb;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __expf(s[r] - m_ref);
        psum += pr;
        const bf16_t h = (bf16_t)pr;
        pb[r] = __builtin_bit_cast(short, h);
      }
      l_part += psum;

      // ---- ct^T[dim][head] += mem_chunk^T . P^T
      // The transposed reads are asm: through the intrinsic hipcc puts an s_waitcnt vmcnt(0) in
      // front of the first one (possible alias with the LDS-DMA in flight), which would wait for
      // the NEXT chunk as well.  8 reads + their lgkmcnt wait per statement.
      const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sb;
      unsigned tad[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) tad[m] = sbase + toff[m];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s16x4 a[8];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8 offset:%16\n\t"
            "ds_read_b64_tr_b16 %1, %9 offset:%16\n\t"
            "ds_read_b64_tr_b16 %2, %10 offset:%16\n\t"
            "ds_read_b64_tr_b16 %3, %11 offset:%16\n\t"
            "ds_read_b64_tr_b16 %4, %12 offset:%16\n\t"
            "ds_read_b64_tr_b16 %5, %13 offset:%16\n\t"
            "ds_read_b64_tr_b16 %6, %14 offset:%16\n\t"
            "ds_read_b64_tr_b16 %7, %15 offset:%16\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
            : "v"(tad[0]), "v"(tad[1]), "v"(tad[2]), "v"(tad[3]), "v"(tad[4]), "v"(tad[5]), "v"(tad[6]), "v"(tad[7]),
              "n"(g * 256)
            : "memory");
#pragma unroll
        for (int m = 0; m < 8; ++m)
          acc[g * 8 + m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[m], pb, acc[g * 8 + m], 0, 0, 0);
      }
    }

    // ---- normalise and store: lane (head fr, group fg) holds dims m*16 + fg*4 + 0..3
    float l = l_part;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (fr < p.heads && !(CARE_LAT_DBG & 1)) {
      bf16_t* out = p.ct + (int64_t)row * p.ldc + fr * LAT_D + fg * 4;
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[m][r] * inv);
        *reinterpret_cast<bf16x4*>(out + m * 16) = o;
      }
    }
  }
}

template <int WAVES, int NSLOT>
int launch_latent(const LatArgs& p, hipStream_t st) {
  constexpr int LDS = WAVES * NSLOT * CH_BYTES + 16 * 128 * 4;
  static std::atomic<unsigned long long> lds_ok{0};  // per device (care_common.h)
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&attention_latent_kernel<WAVES, NSLOT>), LDS, lds_ok))
    return e;
  const int blocks = min((p.rows + WAVES - 1) / WAVES, 256);
  hipLaunchKernelGGL((attention_latent_kernel<WAVES, NSLOT>), dim3(blocks), dim3(WAVES * 64), LDS, st, p);
  return care_launch_status();
}

}  // namespace

extern "C" int care_attention_latent(const void* qt, int64_t ldq, const void* mem, int64_t mem_batch_stride,
                                     int64_t mem_row_stride, int rows_per_kv, int nkeys, const float* bias,
                                     int bias_ld, void* ct, int64_t ldc, int rows, int heads, int d, void* stream) {
  if (!qt || !mem || !ct || rows <= 0 || heads <= 0 || nkeys <= 0 || rows_per_kv <= 0) return CARE_EINVAL;
  if (d != LAT_D || heads > 16 || nkeys > 128) return CARE_ESHAPE;
  if ((ldq % 8) || (ldc % 4) || (mem_batch_stride % 8) || (mem_row_stride % 8) || !care_aligned16(qt) ||
      !care_aligned16(mem) || !care_aligned16(ct) || mem_row_stride < LAT_D)
    return CARE_EALIGN;
  LatArgs p{};
  p.qt = reinterpret_cast<const bf16_t*>(qt); p.ldq = ldq;
  p.mem = reinterpret_cast<const bf16_t*>(mem); p.mem_bs = mem_batch_stride; p.mem_rs = mem_row_stride;
  p.rows_per_kv = rows_per_kv; p.nkeys = nkeys; p.bias = bias; p.bias_ld = bias_ld;
  p.ct = reinterpret_cast<bf16_t*>(ct); p.ldc = ldc; p.rows = rows; p.heads = heads;
  hipStream_t st = (hipStream_t)stream;
  // tuning: 0 = 4 waves x 2 slots, 1 = 3 waves x 3 slots (read once; initialisation is thread-safe)
  static const int cfg = [] { const char* e = getenv("CARE_LAT_CFG"); return e ? atoi(e) : 0; }();
  if (cfg == 1) return launch_latent<3, 3>(p, st);
  return launch_latent<4, 2>(p, st);
}
