// beam_sparse.hip - the second pass of the fused beam selection, only where the first pass says it matters.
//
// The fused per-row top-k of beam search (log_softmax + the per-row part of Beam.advance's top-k: Translator.py:127,
// Beam.py:60) runs the vocabulary GEMM twice: pass 1 for the row statistics and a lower bound thr[r] of the row's
// bm-th best logit, pass 2 to collect the logits >= thr[r] - about ten per row, out of 10 547.  With the maximum of
// every (32-column tile, row) written by pass 1 (csrc/gemm_vocab.hip, `tile_max`), the logits that can reach thr[r]
// are known to lie in the few tiles whose maximum does: pass 2 becomes
//   1. bin:      for every tile the list of rows with tile_max[tile][row] >= thr[row]      (one read of the 27-MB map);
//   2. recompute: per tile, 32 listed rows at a time: the same v_mfma_f32_32x32x16_bf16 chain as pass 1 (same operand
//                 roles, same k order: bit-identical logits), operands straight from global memory / L2 - gathered
//                 activation rows, the tile's 32 W rows - and the logits >= thr[row] appended to the row's list.
// ~2 % of the dense pass's arithmetic.  The candidate lists then go to care_beam_pick as before.
#include <cstdlib>

#include "care_common.h"

#ifndef SP_BLOCKS_DEF
#define SP_BLOCKS_DEF 1024
#endif

namespace {

constexpr int SP_N = 32;       // columns per tile (gemm_vocab.hip's VT_N)
constexpr int SP_BLOCKS = SP_BLOCKS_DEF;  // persistent workgroups of the recompute launch

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void sparse_zero_kernel(int32_t* tcount, int tiles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < tiles) tcount[i] = 0;
}

// grid (ceil(M / 2048), tiles), 256 threads x 8 rows: tlist[tile][..tcount[tile]) = rows whose maximum in the tile reaches
// their threshold.  ONE atomic per workgroup (a same-address atomic per wave cost 70 us: 320 of them per tile queue up).
constexpr int BIN_RPT = 8;
__global__ __launch_bounds__(256) void sparse_bin_kernel(const float* tile_max, const float* thr, int32_t* tcount,
                                                         int32_t* tlist, int M) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int tile = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 256 * BIN_RPT + tid;
  unsigned hot = 0;
#pragma unroll
  for (int i = 0; i < BIN_RPT; ++i) {
    const int row = row0 + i * 256;
    if (row < M && tile_max[(int64_t)tile * M + row] >= thr[row]) hot |= 1u << i;
  }
  const int mine = __popc(hot);
  // exclusive prefix over the workgroup: wave scan by DPP-free shuffles, then the four wave totals
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { if (w < wave) before += wsum[w]; total += wsum[w]; }
  if (total == 0) return;
  if (tid == 0) base_s = atomicAdd(&tcount[tile], total);
  __syncthreads();
  int pos = base_s + before + incl - mine;
#pragma unroll
  for (int i = 0; i < BIN_RPT; ++i)
    if (hot & (1u << i)) tlist[(int64_t)tile * M + pos++] = row0 + i * 256;
}

// one workgroup: unit_start[t] = number of 128-entry work units of the tiles before t (unit_start[tiles] = all of them);
// a tile with a frequent token is hot for every row - its list must spread over the chip, not over 16 workgroups
__global__ __launch_bounds__(512) void sparse_scan_kernel(const int32_t* tcount, int32_t* unit_start, int tiles) {
  __shared__ int part[512];
  const int tid = threadIdx.x;
  const int per = (tiles + 511) / 512;
  int mine = 0;
  for (int i = 0; i < per; ++i) {
    const int t = tid * per + i;
    if (t < tiles) mine += (tcount[t] + 127) >> 7;
  }
  part[tid] = mine;
  __syncthreads();
  for (int o = 1; o < 512; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - mine;
  for (int i = 0; i < per; ++i) {
    const int t = tid * per + i;
    if (t < tiles) { unit_start[t] = run; run += (tcount[t] + 127) >> 7; }
  }
  if (tid == 511) unit_start[tiles] = part[511];
}

struct SpArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W;
  const float* thr;
  const int32_t* tcount; const int32_t* tlist; const int32_t* unit_start; int tiles;
  int32_t* cnt; float* cval; int32_t* cidx; int cap;
  int M, N;
};

constexpr int SP_LDS = 32 * 1024;  // the W tile

// Persistent workgroups over the work units (tile, 128 consecutive entries of the tile's list): wave w of the workgroup takes
// 32 of them.  The W tile goes through LDS once per unit (whole 1-KB rows by LDS-DMA, swizzled like gemm_vocab.hip's
// ring; kept when the next unit is of the same tile); the listed activation rows are gathered straight into the fragments
// (lane = row: 16-byte pieces of 32 different rows per load instruction - slow per wave, but with 32 KB of LDS two
// workgroups share a CU; staging them in LDS too, one workgroup per CU, took 109 us, both operands gathered 168 us).
__global__ __launch_bounds__(256) void sparse_collect_kernel(SpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int total = p.unit_start[p.tiles];
  // fragment offset inside the 32-row image: row r, chunk (2 ks + h) ^ (r & 15) (gemm_vocab.hip `boff`)
  const int bswz = r * 1024;
  auto boff = [&](int ks) { return bswz + ((((2 * ks + h) ^ (r & 15)) & 15) << 4) + ((2 * ks) >> 4) * 256; };
  int staged = -1;
  for (int u = blockIdx.x; u < total; u += gridDim.x) {
    int lo = 0, hi = p.tiles - 1;  // the tile of unit u: the last one whose first unit is <= u
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (p.unit_start[mid] <= u) lo = mid; else hi = mid - 1;
    }
    const int tile = lo;
    const int count = p.tcount[tile];
    const int e0 = (u - p.unit_start[tile]) * 128 + wave * 32;
    if (tile != staged) {  // wave-uniform, the same for the whole workgroup
      __syncthreads();     // everybody is done with the previous tile's image
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // wave w copies rows 8w .. 8w + 7; rows past N clamped, their columns are masked below
        const int n = wave * 8 + i;
        const unsigned char* g = reinterpret_cast<const unsigned char*>(p.W) + (int64_t)min(tile * SP_N + n, p.N - 1) * 1024 +
                                 ((lane ^ (n & 15)) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + n * 1024), 16, 0, 0);
      }
    }
    const bool live = e0 < count;
    const bool valid = live && e0 + r < count;
    const int row = live ? p.tlist[(int64_t)tile * p.M + (valid ? e0 + r : e0)] : 0;  // travels with the W rows
    if (tile != staged) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      staged = tile;
    }
    if (!live) continue;
    const bf16_t* arow = p.A + (int64_t)row * p.lda + h * 8;
    const float th = p.thr[row];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // The activation fragments come in two batches of 16, every load of a batch in flight before its first MFMA
    // (pinned: left alone the scheduler interleaves ONE load per MFMA - 32 dependent memory round trips per 32 rows).
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x8 a[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = *reinterpret_cast<const bf16x8*>(arow + (half * 16 + k) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 16; ++k) {  // operand roles and k order of the first pass: bit-identical logits
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + boff(half * 16 + k));
        acc = care_mfma_32x32x16_h16(b, a[k], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ONE atomic per lane for all its candidates (an atomic with return per candidate is a memory round trip each)
    unsigned hits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = tile * SP_N + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (valid && c < p.N && acc[i] >= th) hits |= 1u << i;
    }
    if (hits) {
      int pos = atomicAdd(&p.cnt[row], __popc(hits));
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (hits & (1u << i)) {
          if (pos < p.cap) {
            p.cval[(int64_t)row * p.cap + pos] = acc[i];
            p.cidx[(int64_t)row * p.cap + pos] = tile * SP_N + (i & 3) + 8 * (i >> 2) + 4 * h;
          }
          ++pos;
        }
    }
  }
}

}  // namespace

extern "C" int care_vocab32_applies(int M, int N, int K, int a_dtype, int has_labels);

// Whether the sparse second pass covers the shape (the 256-row statistics kernel writes the tile maxima).
extern "C" int care_beam_sparse_applies(int M, int N, int K, int a_dtype) {
  return care_vocab32_applies(M, N, K, a_dtype, 0) && K == 512;
}

extern "C" int care_beam_sparse_collect(const void* A, int64_t lda, const void* W, const float* tile_max, const float* thr,
                                        int32_t* cnt, float* cval, int32_t* cidx, int cap, int32_t* tcount, int32_t* tlist,
                                        int M, int N, int K, void* stream) {
  if (!A || !W || !tile_max || !thr || !cnt || !cval || !cidx || !tcount || !tlist || cap <= 0 || M <= 0 || N <= 0) return CARE_EINVAL;
  if (K != 512 || (lda % 8) || !care_aligned16(A) || !care_aligned16(W)) return CARE_ESHAPE;
  const int tiles = (N + SP_N - 1) / SP_N;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sparse_zero_kernel, dim3((tiles + 255) / 256), dim3(256), 0, st, tcount, tiles);
  hipLaunchKernelGGL(sparse_bin_kernel, dim3((M + 256 * BIN_RPT - 1) / (256 * BIN_RPT), tiles), dim3(256), 0, st, tile_max, thr,
                     tcount, tlist, M);
  SpArgs p{};
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.thr = thr;
  int32_t* unit_start = tcount + tiles;  // tcount is [2 tiles + 1]: the counts, then the units' exclusive prefix sums
  hipLaunchKernelGGL(sparse_scan_kernel, dim3(1), dim3(512), 0, st, tcount, unit_start, tiles);
  p.tcount = tcount; p.tlist = tlist; p.unit_start = unit_start; p.tiles = tiles; p.cnt = cnt; p.cval = cval; p.cidx = cidx; p.cap = cap; p.M = M; p.N = N;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int e = care_allow_dynamic_lds(reinterpret_cast<const void*>(&sparse_collect_kernel), SP_LDS, lds_ok)) return e;
  hipLaunchKernelGGL(sparse_collect_kernel, dim3(SP_BLOCKS), dim3(256), SP_LDS, st, p);
  return care_launch_status();
}
