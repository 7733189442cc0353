// gemm_ln.hip - out = LayerNorm(A W^T + bias + res [+ pos]) * gamma + beta for N = d_model = 512.
//
// One workgroup owns FULL output rows (64 or 128 rows x all 512 columns), so the post-LN
// epilogues of the reference (Linear -> LayerNorm in the Embedder, dense -> +residual ->
// LayerNorm in MultiHeadAttention and PositionwiseFeedForward) are fused into the GEMM:
// no intermediate [M, 512] fp32 round trip, no separate LayerNorm launch, no split-K slabs.
//
//   * both operands stream through a 3-slot LDS ring in K steps of 32, filled by LDS-DMA
//     (global_load_lds, 16 B per lane) two steps ahead; A may be fp32 (raw features: staged as
//     fp32, rounded to bf16 when the fragment is read - no register staging) or bf16;
//   * LDS images are lane-linear, so the bank-conflict swizzles are applied to the DMA SOURCE
//     address and again on the ds_read_b128: 64-byte rows (bf16 A, W): chunk ^= (row & 8) >> 2;
//     128-byte rows (fp32 A): chunk ^= table[(row & 15) >> 1] - both brute-forced against the
//     real 16-lane grouping of ds_read_b128 (tools/lds_swizzle_search.py);
//   * waves: RG row groups x 4 column groups; a wave owns 64 rows x 128 columns = 4 x 8 MFMA
//     tiles (128 accumulator VGPRs), so every B fragment feeds 4 MFMAs and every A fragment 8;
//   * waits are counted (vmcnt(N) = the DMA instructions of the one younger stage) with a raw
//     s_barrier per K step;
//   * accumulators use the swapped operand order (a lane holds 4 consecutive columns of a row):
//     row statistics need 2 shuffles + one LDS exchange between the 4 column groups, and every
//     store is 16 bytes (fp32) / 8 bytes (bf16 mirror).
#include <cstdlib>

#include "care_common.h"

namespace {

constexpr int LN_N = 512;

struct LnArgs {
  const void* A; int64_t lda;
  const bf16_t* W;      // [512, K] bf16
  const float* bias; const float* res; int64_t ldres; const float* pos;
  const float* gamma; const float* beta; float eps;
  float* out; bf16_t* outb; int64_t ldo;
  int M, K, grp, out_grp_rows, out_row_off;
};

template <int N>
__device__ __forceinline__ void ln_wait_vm() {
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// AF32: A is fp32.  RG: row groups of 64 rows (BM = 64 * RG, 4 * RG waves).
template <bool AF32, int RG>
__global__ __launch_bounds__(256 * RG, RG == 1 ? 1 : 2) void gemm_ln_kernel(LnArgs p) {
  constexpr int BM = 64 * RG, NW = 4 * RG;
  constexpr int A_ROWB = AF32 ? 128 : 64;                 // bytes per A row per K step
  constexpr int A_BYTES = BM * A_ROWB, W_BYTES = LN_N * 64, SLOT = A_BYTES + W_BYTES;
  constexpr int NA = A_BYTES / 1024, NWI = W_BYTES / 1024;  // DMA instructions per stage
  constexpr int NPW = (NA + NWI) / NW;                     // per wave: 5, 6, 9 or 10
  static_assert((NA + NWI) % NW == 0, "DMA instructions must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 3 slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 2, cg = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  constexpr unsigned HTAB = (2u) | (3u << 3) | (4u << 6) | (2u << 9) | (5u << 12) | (7u << 15) | (4u << 18) | (1u << 21);

  // ---- per-lane DMA source pointers at K step 0 (advance by 64 B (bf16) / 128 B (fp32) per step)
  const unsigned char* src[NPW];
  int dst_off[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave * NPW + i;  // wave-uniform instruction index: [0, NA) -> A, [NA, NA+NWI) -> W
    if (q < NA) {
      if constexpr (AF32) {        // 8 rows x 8 chunks per instruction
        const int row = q * 8 + (lane >> 3), pch = lane & 7;
        const int ch = pch ^ ((HTAB >> (3 * ((row & 15) >> 1))) & 7);
        const int grow = min(m0 + row, p.M - 1);
        src[i] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const float*>(p.A) + (int64_t)grow * p.lda) + ch * 16;
      } else {                     // 16 rows x 4 chunks per instruction
        const int row = q * 16 + (lane >> 2), pch = lane & 3;
        const int ch = pch ^ ((row & 8) >> 2);
        const int grow = min(m0 + row, p.M - 1);
        src[i] = reinterpret_cast<const unsigned char*>(reinterpret_cast<const bf16_t*>(p.A) + (int64_t)grow * p.lda) + ch * 16;
      }
      dst_off[i] = q * 1024;
    } else {
      const int n = (q - NA) * 16 + (lane >> 2), pch = lane & 3;
      const int ch = pch ^ ((n & 8) >> 2);
      src[i] = reinterpret_cast<const unsigned char*>(p.W + (int64_t)n * p.K) + ch * 16;
      dst_off[i] = A_BYTES + (q - NA) * 1024;
    }
  }
  auto stage = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = wave * NPW + i;
      const int step = (q < NA) ? A_ROWB : 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)kt * step),
                                       (__attribute__((address_space(3))) void*)(smem + slot * SLOT + dst_off[i]),
                                       16, 0, 0);
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment offsets inside a slot
  int a_off[4][AF32 ? 2 : 1], w_off[8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = rg * 64 + mt * 16 + fr;
    if constexpr (AF32) {
      const int x = (HTAB >> (3 * (fr >> 1))) & 7;
      a_off[mt][0] = row * 128 + (((2 * fg) ^ x) << 4);
      a_off[mt][1] = row * 128 + (((2 * fg + 1) ^ x) << 4);
    } else {
      a_off[mt][0] = row * 64 + ((fg ^ ((fr & 8) >> 2)) << 4);
    }
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) w_off[nt] = A_BYTES + (cg * 128 + nt * 16 + fr) * 64 + ((fg ^ ((fr & 8) >> 2)) << 4);

  const int nk = p.K >> 5;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; stage kt+1 (NPW younger DMA instructions) may stay in flight
    if (kt + 1 < nk) ln_wait_vm<NPW>(); else ln_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) stage(kt + 2, (kt + 2) % 3);  // the slot every wave finished reading last iteration
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* sl = smem + (kt % 3) * SLOT;
    bf16x8 fa[4], fb[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(sl + w_off[nt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (AF32) {
        const float4 lo = *reinterpret_cast<const float4*>(sl + a_off[mt][0]);
        const float4 hi = *reinterpret_cast<const float4*>(sl + a_off[mt][1]);
        fa[mt][0] = (bf16_t)lo.x; fa[mt][1] = (bf16_t)lo.y; fa[mt][2] = (bf16_t)lo.z; fa[mt][3] = (bf16_t)lo.w;
        fa[mt][4] = (bf16_t)hi.x; fa[mt][5] = (bf16_t)hi.y; fa[mt][6] = (bf16_t)hi.z; fa[mt][7] = (bf16_t)hi.w;
      } else {
        fa[mt] = *reinterpret_cast<const bf16x8*>(sl + a_off[mt][0]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)  // swapped: lane holds 4 consecutive columns of one row
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
  }

  // ------------------------------------------------------------------ epilogue
  // acc[mt][nt][j] = C[row m0 + 64 rg + 16 mt + fr][col 128 cg + 16 nt + 4 fg + j]
  __syncthreads();  // every wave is done with the ring: reuse it for the row statistics
  float* red = reinterpret_cast<float*>(smem);  // [BM][4] partial sums of the 4 column groups
  int rows[4];
  int64_t orow[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    rows[mt] = m0 + rg * 64 + mt * 16 + fr;
    const int rc = min(rows[mt], p.M - 1);
    orow[mt] = (int64_t)(rc / p.grp) * p.out_grp_rows + p.out_row_off + (rc % p.grp);
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int col = cg * 128 + nt * 16 + fg * 4;
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int rc = min(rows[mt], p.M - 1);
      float4 add = bv;
      if (p.res) {
        const float4 r = *reinterpret_cast<const float4*>(p.res + (int64_t)rc * p.ldres + col);
        add.x += r.x; add.y += r.y; add.z += r.z; add.w += r.w;
      }
      if (p.pos) {
        const float4 r = *reinterpret_cast<const float4*>(p.pos + (int64_t)(rc % p.grp) * LN_N + col);
        add.x += r.x; add.y += r.y; add.z += r.z; add.w += r.w;
      }
      acc[mt][nt][0] += add.x; acc[mt][nt][1] += add.y; acc[mt][nt][2] += add.z; acc[mt][nt][3] += add.w;
    }
  }
  float mean[4], rstd[4];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float s = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = pass == 0 ? acc[mt][nt][j] : acc[mt][nt][j] - mean[mt];
          s += pass == 0 ? d : d * d;
        }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (fg == 0) red[(rg * 64 + mt * 16 + fr) * 4 + cg] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 t = *reinterpret_cast<const float4*>(red + (rg * 64 + mt * 16 + fr) * 4);
      const float tot = (t.x + t.y) + (t.z + t.w);
      if (pass == 0) mean[mt] = tot * (1.0f / LN_N);
      else rstd[mt] = 1.0f / sqrtf(tot * (1.0f / LN_N) + p.eps);
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int col = cg * 128 + nt * 16 + fg * 4;
    const float4 g = *reinterpret_cast<const float4*>(p.gamma + col);
    const float4 b = *reinterpret_cast<const float4*>(p.beta + col);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (rows[mt] >= p.M) continue;
      float4 o;
      o.x = (acc[mt][nt][0] - mean[mt]) * rstd[mt] * g.x + b.x;
      o.y = (acc[mt][nt][1] - mean[mt]) * rstd[mt] * g.y + b.y;
      o.z = (acc[mt][nt][2] - mean[mt]) * rstd[mt] * g.z + b.z;
      o.w = (acc[mt][nt][3] - mean[mt]) * rstd[mt] * g.w + b.w;
      const int64_t off = orow[mt] * p.ldo + col;
      *reinterpret_cast<float4*>(p.out + off) = o;
      if (p.outb) {
        bf16x4 ob;
        ob[0] = (bf16_t)o.x; ob[1] = (bf16_t)o.y; ob[2] = (bf16_t)o.z; ob[3] = (bf16_t)o.w;
        *reinterpret_cast<bf16x4*>(p.outb + off) = ob;
      }
    }
  }
}

template <bool AF32, int RG>
int launch_ln(const LnArgs& p, hipStream_t st) {
  constexpr int BM = 64 * RG;
  constexpr size_t lds = 3 * (BM * (AF32 ? 128 : 64) + LN_N * 64);
  const int blocks = (p.M + BM - 1) / BM;
  hipLaunchKernelGGL((gemm_ln_kernel<AF32, RG>), dim3(blocks), dim3(256 * RG), lds, st, p);
  return care_launch_status();
}

}  // namespace

extern "C" int care_gemm_ln(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias,
                            const float* res, int64_t ldres, const float* pos, const float* gamma, const float* beta,
                            float eps, float* out, void* out_bf16, int64_t ldo, int M, int N, int K, int grp,
                            int out_grp_rows, int out_row_off, void* stream) {
  if (!A || !W || !gamma || !beta || !out || M <= 0 || K <= 0 || grp <= 0) return CARE_EINVAL;
  if (a_dtype != CARE_F32 && a_dtype != CARE_BF16) return CARE_EDTYPE;
  if (N != LN_N || K % 32 != 0) return CARE_ESHAPE;
  if (!care_aligned16(A) || !care_aligned16(W) || (lda % (a_dtype == CARE_BF16 ? 8 : 4)) || (ldo % 4) ||
      (res && (ldres % 4)) || !care_aligned16(out) || (bias && !care_aligned16(bias)))
    return CARE_EALIGN;
  LnArgs p{};
  p.A = A; p.lda = lda; p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias; p.res = res; p.ldres = ldres;
  p.pos = pos; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out = out; p.outb = reinterpret_cast<bf16_t*>(out_bf16);
  p.ldo = ldo; p.M = M; p.K = K; p.grp = grp; p.out_grp_rows = out_grp_rows; p.out_row_off = out_row_off;
  hipStream_t st = (hipStream_t)stream;
  bool big = (M + 127) / 128 >= 256;  // enough 128-row panels to fill the chip
  if (const char* e = getenv("CARE_LN_RG")) big = atoi(e) >= 2;  // tuning override
  if (a_dtype == CARE_F32) return big ? launch_ln<true, 2>(p, st) : launch_ln<true, 1>(p, st);
  return big ? launch_ln<false, 2>(p, st) : launch_ln<false, 1>(p, st);
}
