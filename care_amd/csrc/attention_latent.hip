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
//   * the row's expanded query is one more stage of the same ring in front of its chunks (round 2; see stage_q);
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
// What bounds it.  Round 1 read the kernel as limited by bytes in flight (DMA ring alone 6.2 TB/s, full kernel
// 5.0-5.3).  Round 2's ablation builds (tools/variant_lib.py, 32768 rows x 84 keys) found the actual costs: the
// query load at the row start was waited for with vmcnt(0) - the wave's DMA queue drained once per row (658 us;
// without q~ loads and c~ stores 465) - fixed by sending the query through the ring; the stream's cache policy
// (673 -> 574 us with non-temporal loads and stores); and the c~ stores, whose cost is their bytes (no stores:
// 513 us).  Arithmetic is hidden completely, and the ring depth no longer matters (3 waves x 2 slots within 1 %
// of 4 x 2 and 3 x 3): 545-615 us per launch = 5.4-6.2 TB/s, the memory system's rate for this read/write mix.
#include "care_common.h"

#ifndef CARE_LAT_DBG
#define CARE_LAT_DBG 0  // ablation builds (tools/variant_lib.py): 1 no ct stores, 2 no qt loads, 4 no arithmetic, 8 direct 8-byte stores
#endif

// Cache policy of the streams (ablation: -DCARE_LAT_LD_AUX=0 -DCARE_LAT_ST_NT=0).  The clips' memory, q~ and c~ are
// each touched once per launch: marked non-temporal they stop evicting one another (and the next kernels' weights)
// from L2 / the memory-side cache.  *Measured* 32768 rows x 84 keys, same box: 673 us plain, 658 nt stores only,
// 629 nt loads only, 574 both; a later box: 622-670 plain, 571-608 all three nt, 535-568 with plain q~ loads.
#ifndef CARE_LAT_LD_AUX
#define CARE_LAT_LD_AUX 2  // cache-policy immediate of the LDS-DMA loads: 1 sc0, 2 nt, 16 sc1
#endif
#ifndef CARE_LAT_Q_AUX
#define CARE_LAT_Q_AUX CARE_LAT_LD_AUX  // q~ too (whole pass, B = 32768: 57.4 ms plain, 55.0 with plain q~ loads, 54.7 all nt)
#endif
#ifndef CARE_LAT_ST_NT
#define CARE_LAT_ST_NT 1
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
  int paired;                         // two rows of a clip per wave (see the kernel)
};

__device__ __forceinline__ int lat_swz(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }

// s_waitcnt vmcnt(n) for a wave-uniform n in 0..32 (or 48); anything else waits for MORE than asked (always safe)
__device__ __forceinline__ void lat_wait_vm(int n) {
  switch (n) {
#define LAT_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    LAT_VM_CASE(0) LAT_VM_CASE(1) LAT_VM_CASE(2) LAT_VM_CASE(3) LAT_VM_CASE(4) LAT_VM_CASE(5) LAT_VM_CASE(6) LAT_VM_CASE(7)
    LAT_VM_CASE(8) LAT_VM_CASE(9) LAT_VM_CASE(10) LAT_VM_CASE(11) LAT_VM_CASE(12) LAT_VM_CASE(13) LAT_VM_CASE(14)
    LAT_VM_CASE(15) LAT_VM_CASE(16)
    LAT_VM_CASE(17) LAT_VM_CASE(18) LAT_VM_CASE(19) LAT_VM_CASE(20) LAT_VM_CASE(21) LAT_VM_CASE(22) LAT_VM_CASE(23)
    LAT_VM_CASE(24) LAT_VM_CASE(25) LAT_VM_CASE(26) LAT_VM_CASE(27) LAT_VM_CASE(28) LAT_VM_CASE(29) LAT_VM_CASE(30)
    LAT_VM_CASE(31) LAT_VM_CASE(32) LAT_VM_CASE(48)
#undef LAT_VM_CASE
    default:
      if (n > 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      break;
  }
}

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
                                       16, 0, CARE_LAT_LD_AUX);
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

  // MFMA column fr = (sub-row, head).  With <= 8 heads and rows that share a clip's memory (beam search:
  // rows_per_kv consecutive rows per clip) a wave takes TWO rows of the clip at once - columns 0..7 the
  // first, 8..15 the second - so the memory streams once per pair instead of once per row, at the same
  // MFMA cost (the N axis is 16 wide either way).  Work items: `ipc` per clip, the last one of a clip
  // with an odd row count is half empty.
  const bool paired = p.paired != 0;
  const int ipc = paired ? (p.rows_per_kv + 1) / 2 : 1;
  const int items = paired ? (p.rows / p.rows_per_kv) * ipc : p.rows;
  const int ncols = paired ? 2 * p.heads : p.heads;  // columns in use
  const int nq = ncols <= 8 ? 8 : 16;                // DMA instructions of a query stage
  auto col_sub = [&](int i) { return paired && i >= p.heads ? 1 : 0; };
  auto col_head = [&](int i) { return paired ? (i >= p.heads ? i - p.heads : i) : min(i, p.heads - 1); };
  const int csub = col_sub(fr), headc = fr < ncols ? col_head(fr) : 0;
  auto first_row = [&](int item) { return paired ? (item / ipc) * p.rows_per_kv + 2 * (item % ipc) : item; };
  // output staging (see the store phase): my 8-byte piece of global chunk 2 m + (fg >> 1) of column fr
  int woff[8];
#pragma unroll
  for (int mm = 0; mm < 8; ++mm) woff[mm] = fr * 1024 + ((((2 * mm + (fg >> 1)) ^ lat_swz(fr)) & 15) << 4) + 8 * (fg & 1);
  // a query stage of 8 rows: the idle columns 8..15 read rows 0..7 again (defined values, never stored)
  const int frq = fr < nq ? fr : fr - 8;
  int qoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qoff[r] = frq * 1024 + ((((r * 4 + fg) ^ lat_swz(frq)) & 15) << 4);

  // The expanded query of an item travels through the SAME ring as one more stage in front of the
  // item's memory chunks: LDS row i = the 1 KB of column i (same swizzle, so the B fragments are read
  // exactly like a chunk's A fragments).  A plain global load at the row start would be waited for
  // with vmcnt(0) - the wave's whole DMA queue drained once per row (measured: 597 -> 465 us of the
  // 658 us launch at 32768 rows x 84 keys were the q~ loads and c~ stores).
  auto stage_q = [&](int item, int slot) {
    const int row = first_row(item);
    const bool second = paired && (row % p.rows_per_kv) + 1 < p.rows_per_kv;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i >= nq) break;
      const int ci = i < ncols ? i : 0;
      const int r = row + (col_sub(ci) && second ? 1 : 0);
      const unsigned char* g = reinterpret_cast<const unsigned char*>(p.qt + (int64_t)r * p.ldq + col_head(ci) * LAT_D) +
                               ((lane ^ lat_swz(i)) << 4);
      if (CARE_LAT_DBG & 2) continue;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ring + slot * CH_BYTES + i * 1024),
                                       16, 0, CARE_LAT_Q_AUX);
    }
  };

  // additive per-(head, key) term, the same for every row: hybrid bias (0 without one) for valid
  // keys, -inf for the padding keys of the last chunk.  Staged in LDS once per block - a global
  // load inside the chunk loop would be YOUNGER than the next chunk's DMAs and drain them.
  float* sbias = reinterpret_cast<float*>(smem + WAVES * NSLOT * CH_BYTES);  // [16][128]
  for (int i = threadIdx.x; i < 16 * 128; i += WAVES * 64) {
    const int h = min(i >> 7, p.heads - 1), key = i & 127;
    sbias[i] = key < p.nkeys ? (p.bias ? p.bias[h * p.bias_ld + key] : 0.f) : -INFINITY;
  }
  __syncthreads();
  int t = 0;  // stages consumed so far by this wave -> ring slot
  if (gw < items) stage_q(gw, 0);
  int n_stored = 0;  // store instructions of the previous item: younger than the stage being waited for

  for (int item = gw; item < items; item += total_waves) {
    const int row = first_row(item);
    // is my column a real (row, head)?  The second row of a pair must belong to the same clip.
    const bool col_ok = fr < ncols && (csub == 0 || (row % p.rows_per_kv) + 1 < p.rows_per_kv);
    const int myrow = row + (col_ok ? csub : 0);
    // ---- query stage: B operand of the S MFMAs, resident for all chunks
    bf16x8 qf[16];
    {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage(row, 0, (t + 1) % NSLOT);
      // the query has landed; younger: chunk 0 (16) and, in program order before it, the previous item's stores
      // (loads and stores retire in issue order on gfx9-family vmcnt, which is what hipcc itself assumes)
      lat_wait_vm(16 + n_stored);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* sq = ring + (t % NSLOT) * CH_BYTES;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (CARE_LAT_DBG & 2) { qf[ks] = bf16x8{}; asm volatile("" : "+v"(qf[ks])); }
        else qf[ks] = *reinterpret_cast<const bf16x8*>(sq + qoff[ks & 3] + (ks >> 2) * 256);
      }
      ++t;
    }

    f32x4 acc[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_ref = -INFINITY, l_part = 0.f;

    for (int c = 0; c < nch; ++c, ++t) {
      const int slot = t % NSLOT;
      // next chunk (of this row or of the wave's next row) into the slot consumed one iteration ago
      const bool more_here = c + 1 < nch;
      const bool have_next = more_here || item + total_waves < items;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS reads of that slot are done
      if (more_here) {
        stage(row, c + 1, (t + 1) % NSLOT);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // everything but the 16 newest DMAs
      } else if (have_next) {
        stage_q(item + total_waves, (t + 1) % NSLOT);
        if ((CARE_LAT_DBG & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nq == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
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
        s = care_mfma_16x16x32_h16(a, qf[ks], s, 0, 0, 0);
      }
      // lane (head fr, group fg) holds keys c*16 + fg*4 + r
      s += *reinterpret_cast<const f32x4*>(sbias + headc * 128 + c * CH_KEYS + fg * 4);

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
          acc[g * 8 + m] = care_mfma_16x16x16_h16(a[m], pb, acc[g * 8 + m], 0, 0, 0);
      }
    }

    // ---- normalise and store: lane (head fr, group fg) holds dims m*16 + fg*4 + 0..3
    float l = l_part;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // Through the ring slot of the item's last chunk (consumed, not restaged before the next query turn): every
    // lane writes its 8-byte pieces at the swizzled chunk positions, then each column's 1 KB goes out as ONE store
    // instruction of full lines (the direct form was 32 instructions of 32-byte pieces per column: 86 us of the
    // 599 us launch at 32768 rows x 84 keys).
    n_stored = 0;
    if (CARE_LAT_DBG & 8) {  // ablation: the direct form
      if (col_ok) {
        bf16_t* out = p.ct + (int64_t)myrow * p.ldc + headc * LAT_D + fg * 4;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[m][r] * inv);
          *reinterpret_cast<bf16x4*>(out + m * 16) = o;
        }
      }
      n_stored = 32;
    } else if (!(CARE_LAT_DBG & 1)) {
      const unsigned ob = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(ring + ((t - 1) % NSLOT) * CH_BYTES);
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[m][r] * inv);
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ob + woff[m & 7]), "v"(o), "n"((m >> 3) * 256) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool second = paired && (row % p.rows_per_kv) + 1 < p.rows_per_kv;
      const unsigned rb = ob + lane * 16;
#pragma unroll
      for (int i0 = 0; i0 < 16; i0 += 8) {
        if (i0 >= ncols) break;
        f32x4 v[8];
        asm volatile(
            "ds_read_b128 %0, %8 offset:%9\n\t"
            "ds_read_b128 %1, %8 offset:%9+1024\n\t"
            "ds_read_b128 %2, %8 offset:%9+2048\n\t"
            "ds_read_b128 %3, %8 offset:%9+3072\n\t"
            "ds_read_b128 %4, %8 offset:%9+4096\n\t"
            "ds_read_b128 %5, %8 offset:%9+5120\n\t"
            "ds_read_b128 %6, %8 offset:%9+6144\n\t"
            "ds_read_b128 %7, %8 offset:%9+7168\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(rb), "n"(i0 * 1024)
            : "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = i0 + k;
          if (i < ncols && (col_sub(i) == 0 || second)) {
            unsigned char* dst = reinterpret_cast<unsigned char*>(p.ct + (int64_t)(row + col_sub(i)) * p.ldc + col_head(i) * LAT_D) +
                                 ((lane ^ lat_swz(i)) << 4);
            if (CARE_LAT_ST_NT) __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(dst));
            else *reinterpret_cast<f32x4*>(dst) = v[k];
            ++n_stored;
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------ d_model = 1024 / 768: several waves per row
// The same algorithm with the row's dims cut into NW slices of DW: a workgroup of NW waves owns a row, wave w streams and
// multiplies dims [DW w, DW w + DW) - its own ring, its own query fragments and accumulators (NW = 4, DW = 256 for
// d_model 1024 - round 4; NW = 2, DW = 512, the single-wave kernel's footprint per wave, before; NW = 3, DW = 256 for
// d_model 768: 512-byte LDS rows, two key rows per LDS-DMA instruction).  What the slices share is the score: each wave has the partial dot products over
// its dims, they swap them through NW x 1 KB of LDS (double-buffered by chunk parity, one raw s_barrier per chunk - the
// DMAs in flight are not drained) and every wave adds the NW partials in the SAME order, so all take identical softmax
// decisions; 16 / 12 heads fill the MFMA N axis.  Per row and step Lk x d x 2 B of memory + 2 x H x d x 2 B of q~ / c~
// instead of 2 x Lk x d x 2 B of projected K and V.
template <int NW, int DW, int NSLOT>
__global__ __launch_bounds__(64 * NW, 1) void attention_latentN_kernel(LatArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int DN = NW * DW;              // d_model
  constexpr int ROWB = DW * 2;             // bytes of an LDS row (a key's slice)
  constexpr int RPI = 1024 / ROWB;         // key rows per LDS-DMA instruction
  constexpr int NDMA = CH_KEYS / RPI;      // DMA instructions per chunk (and per query stage of 16 columns)
  constexpr int CHB = CH_KEYS * ROWB;      // bytes of a chunk image
  constexpr int KS = DW / 32;              // MFMA k-steps of the score product
  constexpr int NACC = DW / 16;            // 16-dim accumulator tiles of the context
  constexpr int CMASK = ROWB / 16 - 1;     // chunk-in-row mask of a DMA lane
  static_assert(DW == 512 || DW == 256, "slice width");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = which slice of the dims
  unsigned char* ring = smem + wave * (NSLOT * CHB);
  float* sbias = reinterpret_cast<float*>(smem + NW * NSLOT * CHB);              // [16][128]
  float* xch = reinterpret_cast<float*>(smem + NW * NSLOT * CHB + 16 * 128 * 4);  // [2 parities][NW waves][64 lanes][4]
  const int fr = lane & 15, fg = lane >> 4;
  const int nch = (p.nkeys + CH_KEYS - 1) / CH_KEYS;
  const int items = p.rows;
  const int ncols = p.heads;                // <= 16
  const int nq = (RPI == 1 && ncols <= 8) ? 8 : 16;   // LDS rows of a query stage
  const int lrow = RPI == 1 ? 0 : (lane >> 5);        // row of a DMA instruction this lane copies
  const int lchunk = lane & CMASK;

  auto stage = [&](int row, int c, int slot) {
    const bf16_t* base = p.mem + (int64_t)(row / p.rows_per_kv) * p.mem_bs + wave * DW;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int r = i * RPI + lrow;
      const int key = min(c * CH_KEYS + r, p.nkeys - 1);
      const unsigned char* g = reinterpret_cast<const unsigned char*>(base + (int64_t)key * p.mem_rs) + ((lchunk ^ lat_swz(r)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ring + slot * CHB + i * 1024), 16, 0,
                                       CARE_LAT_LD_AUX);
    }
  };
  auto stage_q = [&](int row, int slot) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      if (i * RPI >= nq) break;
      const int r = i * RPI + lrow;
      const int hcol = min(r, ncols - 1);
      const unsigned char* g = reinterpret_cast<const unsigned char*>(p.qt + (int64_t)row * p.ldq + hcol * DN + wave * DW) +
                               ((lchunk ^ lat_swz(r)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ring + slot * CHB + i * 1024), 16, 0,
                                       CARE_LAT_Q_AUX);
    }
  };
  const int nq_dma = (nq + RPI - 1) / RPI;   // DMA instructions of a query stage

  int roff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) roff[r] = fr * ROWB + ((((r * 4 + fg) ^ lat_swz(fr)) & 15) << 4);
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int trow = fg * 4 + tq;
  int toff[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) toff[m] = trow * ROWB + ((((m * 2 + (tp >> 1)) ^ lat_swz(trow)) & 15) << 4) + 8 * (tp & 1);
  int woff[8];
#pragma unroll
  for (int mm = 0; mm < 8; ++mm) woff[mm] = fr * ROWB + ((((2 * mm + (fg >> 1)) ^ lat_swz(fr)) & 15) << 4) + 8 * (fg & 1);
  const int frq = fr < nq ? fr : fr - 8;
  int qoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qoff[r] = frq * ROWB + ((((r * 4 + fg) ^ lat_swz(frq)) & 15) << 4);
  const int headc = fr < ncols ? fr : 0;

  for (int i = threadIdx.x; i < 16 * 128; i += 64 * NW) {
    const int h = min(i >> 7, p.heads - 1), key = i & 127;
    sbias[i] = key < p.nkeys ? (p.bias ? p.bias[h * p.bias_ld + key] : 0.f) : -INFINITY;
  }
  __syncthreads();
  int t = 0, par = 0;
  if ((int)blockIdx.x < items) stage_q(blockIdx.x, 0);
  int n_stored = 0;

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int row = item;
    bf16x8 qf[KS];
    {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage(row, 0, (t + 1) % NSLOT);
      lat_wait_vm(NDMA + n_stored);   // the query has landed; younger: chunk 0 and, before it, the previous item's stores
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* sq = ring + (t % NSLOT) * CHB;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(sq + qoff[ks & 3] + (ks >> 2) * 256);
      ++t;
    }
    f32x4 acc[NACC];
#pragma unroll
    for (int m = 0; m < NACC; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_ref = -INFINITY, l_part = 0.f;

    for (int c = 0; c < nch; ++c, ++t, par ^= 1) {
      const int slot = t % NSLOT;
      const bool more_here = c + 1 < nch;
      const bool have_next = more_here || item + (int)gridDim.x < items;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more_here) {
        stage(row, c + 1, (t + 1) % NSLOT);
        lat_wait_vm(NDMA);
      } else if (have_next) {
        stage_q(item + gridDim.x, (t + 1) % NSLOT);
        lat_wait_vm(nq_dma);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* sb = ring + slot * CHB;

      // ---- partial S^T over this wave's dims, summed with the other slices' in a fixed order
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + roff[ks & 3] + (ks >> 2) * 256);
        s = care_mfma_16x16x32_h16(a, qf[ks], s, 0, 0, 0);
      }
      {
        const unsigned xb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)xch;
        const unsigned mine = xb + (unsigned)(((par * NW + wave) * 64 + lane) * 16);
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(mine), "v"(s) : "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 o[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const unsigned src = xb + (unsigned)(((par * NW + w) * 64 + lane) * 16);
          asm volatile("ds_read_b128 %0, %1" : "=v"(o[w]) : "v"(src) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        s = o[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += o[w];
      }
      s += *reinterpret_cast<const f32x4*>(sbias + headc * 128 + c * CH_KEYS + fg * 4);

      float cm = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      if (__any(cm > m_ref + 16.0f)) {
        const float m_new = fmaxf(m_ref, cm);
        const float alpha = __expf(m_ref - m_new);
        m_ref = m_new;
        l_part *= alpha;
#pragma unroll
        for (int m = 0; m < NACC; ++m) acc[m] *= alpha;
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

      const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sb;
      unsigned tad[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) tad[m] = sbase + toff[m];
#pragma unroll
      for (int g = 0; g < NACC / 8; ++g) {
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
          acc[g * 8 + m] = care_mfma_16x16x16_h16(a[m], pb, acc[g * 8 + m], 0, 0, 0);
      }
    }

    // ---- normalise and store this wave's DW dims of every head
    float l = l_part;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    n_stored = 0;
    const unsigned ob = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(ring + ((t - 1) % NSLOT) * CHB);
#pragma unroll
    for (int m = 0; m < NACC; ++m) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[m][r] * inv);
      asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ob + woff[m & 7]), "v"(o), "n"((m >> 3) * 256) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned rb = ob + lane * 16;
    // one instruction moves 1 KB of the staged image: one column (DW = 512) or two (DW = 256), each as whole lines
#pragma unroll
    for (int i0 = 0; i0 < NDMA; i0 += 8) {
      if (i0 * RPI >= ncols) break;
      f32x4 v[8];
      asm volatile(
          "ds_read_b128 %0, %8 offset:%9\n\t"
          "ds_read_b128 %1, %8 offset:%9+1024\n\t"
          "ds_read_b128 %2, %8 offset:%9+2048\n\t"
          "ds_read_b128 %3, %8 offset:%9+3072\n\t"
          "ds_read_b128 %4, %8 offset:%9+4096\n\t"
          "ds_read_b128 %5, %8 offset:%9+5120\n\t"
          "ds_read_b128 %6, %8 offset:%9+6144\n\t"
          "ds_read_b128 %7, %8 offset:%9+7168\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
          : "v"(rb), "n"(i0 * 1024)
          : "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = (i0 + k) * RPI + lrow;   // the column this lane's 16 bytes belong to
        if ((i0 + k) * RPI < ncols) {          // wave-uniform: the instruction holds at least one real column
          if (i < ncols) {
            unsigned char* dst = reinterpret_cast<unsigned char*>(p.ct + (int64_t)row * p.ldc + i * DN + wave * DW) +
                                 ((lchunk ^ lat_swz(i)) << 4);
            if (CARE_LAT_ST_NT) __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(dst));
            else *reinterpret_cast<f32x4*>(dst) = v[k];
          }
          ++n_stored;
        }
      }
    }
  }
}

template <int NW, int DW, int NSLOT>
int launch_latentN(const LatArgs& p, hipStream_t st) {
  constexpr int LDS = NW * NSLOT * CH_KEYS * DW * 2 + 16 * 128 * 4 + 2 * NW * 64 * 16;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&attention_latentN_kernel<NW, DW, NSLOT>), LDS, lds_ok)) return e;
  const int blocks = min(p.rows, 512);  // two workgroups per CU
  hipLaunchKernelGGL((attention_latentN_kernel<NW, DW, NSLOT>), dim3(blocks), dim3(64 * NW), LDS, st, p);
  return care_launch_status();
}

// ------------------------------------------------------------------ few rows (a launch-latency-bound decode step)
// With one row per wave and at most one wave per row-slot of the chip, the single-wave kernel above is a latency chain:
// six or eight chunks, each requested one chunk ahead (*measured* by rocprofv3 inside the replayed graph: 17 us at 1
// row, 20 us at 128 rows - the longest kernel of a small-batch decode step).  This variant is the SAME arithmetic in
// the SAME order (chunk after chunk, the same lazy-reference softmax: bit-identical outputs) with the row's query
// stage and its first PF = 3 chunks in flight at once (64 LDS-DMA instructions, the 6-bit vmcnt's reach) and every
// later chunk requested three ahead: one wave per workgroup, a ring of PF + 1 chunk slots + the query slot.  Single
// items per wave only (rows <= workgroups): no cross-row prefetch, no store accounting.
template <int PF>
__global__ __launch_bounds__(64, 1) void attention_latent_few_kernel(LatArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NCS = PF + 1;                 // chunk slots
  const int lane = threadIdx.x & 63;
  unsigned char* ring = smem;                 // [NCS chunk slots][query slot]
  float* sbias = reinterpret_cast<float*>(smem + (NCS + 1) * CH_BYTES);  // [16][128]
  const int fr = lane & 15, fg = lane >> 4;
  const int nch = (p.nkeys + CH_KEYS - 1) / CH_KEYS;
  const int row = blockIdx.x;
  const int ncols = p.heads;
  const int nq = ncols <= 8 ? 8 : 16;

  auto stage = [&](int c, int slot) {
    const bf16_t* base = p.mem + (int64_t)(row / p.rows_per_kv) * p.mem_bs;
#pragma unroll
    for (int i = 0; i < CH_KEYS; ++i) {
      const int key = min(c * CH_KEYS + i, p.nkeys - 1);
      const unsigned char* g = reinterpret_cast<const unsigned char*>(base + (int64_t)key * p.mem_rs) + ((lane ^ lat_swz(i)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ring + slot * CH_BYTES + i * 1024), 16, 0, 0);
    }
  };
  // query stage first (it is needed first), then the first PF chunks
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i >= nq) break;
    const int hcol = min(i, ncols - 1);
    const unsigned char* g = reinterpret_cast<const unsigned char*>(p.qt + (int64_t)row * p.ldq + hcol * LAT_D) + ((lane ^ lat_swz(i)) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(ring + NCS * CH_BYTES + i * 1024), 16, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < PF; ++c)
    if (c < nch) stage(c, c);

  int roff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) roff[r] = fr * 1024 + ((((r * 4 + fg) ^ lat_swz(fr)) & 15) << 4);
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int trow = fg * 4 + tq;
  int toff[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) toff[m] = trow * 1024 + ((((m * 2 + (tp >> 1)) ^ lat_swz(trow)) & 15) << 4) + 8 * (tp & 1);
  const int frq = fr < nq ? fr : fr - 8;
  int qoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qoff[r] = frq * 1024 + ((((r * 4 + fg) ^ lat_swz(frq)) & 15) << 4);
  const int headc = fr < ncols ? fr : 0;

  for (int i = lane; i < 16 * 128; i += 64) {   // ordinary loads: they retire BEHIND the DMAs issued above (in-order vmcnt)
    const int h = min(i >> 7, p.heads - 1), key = i & 127;
    sbias[i] = key < p.nkeys ? (p.bias ? p.bias[h * p.bias_ld + key] : 0.f) : -INFINITY;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // one wave per workgroup: no barrier needed
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 qf[16];
  {
    const unsigned char* sq = ring + NCS * CH_BYTES;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(sq + qoff[ks & 3] + (ks >> 2) * 256);
  }
  f32x4 acc[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_ref = -INFINITY, l_part = 0.f;

  for (int c = 0; c < nch; ++c) {
    const int slot = c % NCS;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of chunk c - 1 (whose slot is re-staged now) are done
    if (c >= 1 && c + PF - 1 < nch && c + PF - 1 >= PF) stage(c + PF - 1, (c + PF - 1) % NCS);
    // chunk c has landed when only the younger chunks' DMAs are outstanding
    {
      const int younger = min(c + PF - 1, nch - 1) - c;   // chunks requested after chunk c (0 .. PF - 1)
      if (c < PF) {
        // chunks 0 .. PF - 1 were all waited for together with the bias loads above
      } else if (younger >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* sb = ring + slot * CH_BYTES;

    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + roff[ks & 3] + (ks >> 2) * 256);
      s = care_mfma_16x16x32_h16(a, qf[ks], s, 0, 0, 0);
    }
    s += *reinterpret_cast<const f32x4*>(sbias + headc * 128 + c * CH_KEYS + fg * 4);
    float cm = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
    cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
    if (__any(cm > m_ref + 16.0f)) {
      const float m_new = fmaxf(m_ref, cm);
      const float alpha = __expf(m_ref - m_new);
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
        acc[g * 8 + m] = care_mfma_16x16x16_h16(a[m], pb, acc[g * 8 + m], 0, 0, 0);
    }
  }

  float l = l_part;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (fr < ncols) {   // lane (head fr, group fg) holds dims m * 16 + fg * 4 + 0..3: 8-byte stores (one row: nothing to coalesce with)
    bf16_t* out = p.ct + (int64_t)row * p.ldc + headc * LAT_D + fg * 4;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[m][r] * inv);
      *reinterpret_cast<bf16x4*>(out + m * 16) = o;
    }
  }
}

int launch_latent_few(const LatArgs& p, hipStream_t st) {
  constexpr int PF = 3;
  constexpr int LDS = (PF + 2) * CH_BYTES + 16 * 128 * 4;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&attention_latent_few_kernel<PF>), LDS, lds_ok)) return e;
  hipLaunchKernelGGL((attention_latent_few_kernel<PF>), dim3(p.rows), dim3(64), LDS, st, p);
  return care_launch_status();
}

template <int WAVES, int NSLOT>
int launch_latent(const LatArgs& p, hipStream_t st) {
  constexpr int LDS = WAVES * NSLOT * CH_BYTES + 16 * 128 * 4;
  static std::atomic<unsigned long long> lds_ok{0};  // per device (care_common.h)
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&attention_latent_kernel<WAVES, NSLOT>), LDS, lds_ok))
    return e;
  const bool paired = p.paired != 0;
  const int items = paired ? (p.rows / p.rows_per_kv) * ((p.rows_per_kv + 1) / 2) : p.rows;
  const int blocks = min((items + WAVES - 1) / WAVES, 256);
  hipLaunchKernelGGL((attention_latent_kernel<WAVES, NSLOT>), dim3(blocks), dim3(WAVES * 64), LDS, st, p);
  return care_launch_status();
}

}  // namespace

extern "C" int care_attention_latent(const void* qt, int64_t ldq, const void* mem, int64_t mem_batch_stride,
                                     int64_t mem_row_stride, int rows_per_kv, int nkeys, const float* bias,
                                     int bias_ld, void* ct, int64_t ldc, int rows, int heads, int d, void* stream) {
  if (!qt || !mem || !ct || rows <= 0 || heads <= 0 || nkeys <= 0 || rows_per_kv <= 0) return CARE_EINVAL;
  if ((d != LAT_D && d != 2 * LAT_D && d != 768) || heads > 16 || nkeys > 128) return CARE_ESHAPE;
  if ((ldq % 8) || (ldc % 4) || (mem_batch_stride % 8) || (mem_row_stride % 8) || !care_aligned16(qt) ||
      !care_aligned16(mem) || !care_aligned16(ct) || mem_row_stride < LAT_D)
    return CARE_EALIGN;
  LatArgs p{};
  p.qt = reinterpret_cast<const bf16_t*>(qt); p.ldq = ldq;
  p.mem = reinterpret_cast<const bf16_t*>(mem); p.mem_bs = mem_batch_stride; p.mem_rs = mem_row_stride;
  p.rows_per_kv = rows_per_kv; p.nkeys = nkeys; p.bias = bias; p.bias_ld = bias_ld;
  p.ct = reinterpret_cast<bf16_t*>(ct); p.ldc = ldc; p.rows = rows; p.heads = heads;
  static const int pair_ok = [] { const char* e = getenv("CARE_LAT_PAIR"); return e ? atoi(e) : 1; }();  // A/B switch
  p.paired = pair_ok && heads <= 8 && rows_per_kv > 1 && rows % rows_per_kv == 0;
  hipStream_t st = (hipStream_t)stream;
  if (d == 2 * LAT_D) {  // d_model = 1024: four waves per row, 256 dims each (attention_latentN_kernel)
    p.paired = 0;
    // *measured* (vatex_care_large, 16384 rows x 84 keys, 16 heads, same box): two waves x 512 dims 952 us per launch,
    // four x 256 with a ring of 2 chunks 809 (6.05 TB/s of the algorithmic bytes), with 3 chunks 831: twice the waves per
    // CU (8 instead of 4 at two workgroups per CU) cover the per-chunk exchange of partial scores and its barrier
    static const int ncfg = [] { const char* e = getenv("CARE_LATN_CFG"); return e ? atoi(e) : 1; }();  // tuning / A-B
    if (ncfg == 0) return launch_latentN<2, 512, 2>(p, st);
    if (ncfg == 2) return launch_latentN<4, 256, 3>(p, st);
    return launch_latentN<4, 256, 2>(p, st);
  }
  if (d == 768) {        // d_model = 768: three waves per row, 256 dims each
    p.paired = 0;
    return launch_latentN<3, 256, 2>(p, st);
  }
  // few rows: one wave per row with the whole head of the row's stream in flight (attention_latent_few_kernel)
  static const int few_rows = [] { const char* e = getenv("CARE_LAT_FEW_ROWS"); return e ? atoi(e) : 256; }();
  if (rows <= few_rows && !p.paired) return launch_latent_few(p, st);
  // tuning: 0 = 4 waves x 2 slots, 1 = 3 waves x 3 slots (read once; initialisation is thread-safe).  (Round 4: the
  // several-waves-per-row kernel as two waves of 256 dims at this d_model: 736 / 582 / 572 us with rings of 2 / 3 / 4
  // chunks against 560 here - the exchange of partial scores buys nothing when a row fits one wave.)
  static const int cfg = [] { const char* e = getenv("CARE_LAT_CFG"); return e ? atoi(e) : 0; }();
  if (cfg == 1) return launch_latent<3, 3>(p, st);
  if (cfg == 2) return launch_latent<3, 2>(p, st);
  if (cfg == 3) return launch_latent<2, 4>(p, st);
  return launch_latent<4, 2>(p, st);
}
