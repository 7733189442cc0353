// decode_resident.h - device code shared by the resident decodes (decode_resident.hip: greedy; decode_resident_beam.hip:
// beam search): hand-offs between phases, the GEMM / attention phases, LayerNorm on load.  See decode_resident.hip.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "care_common.h"

namespace {

#ifdef RES_NOINLINE
#define RES_PHASE_FN __device__ __attribute__((noinline))
#else
#define RES_PHASE_FN __device__ __forceinline__
#endif
constexpr int RES_MAX_LAYERS = 4;
constexpr int RES_MAXKB = 16;  // key blocks of 8: <= 128 keys per attention

struct RAttn {
  const bf16_t* q_w; const float* q_b; const bf16_t* o_w; const float* o_b; const float* g; const float* be;
  const bf16_t* kv; int64_t kv_bs; int nkeys, rows_per_kv; const float* bias; int bias_ld;
};
struct RLayer {
  const bf16_t* qkv_w; const float* qkv_b; const bf16_t* o_w; const float* o_b; const float* g; const float* be;
  bf16_t* skv;
  RAttn att[2]; int n_att;
  const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2; const float* fg; const float* fbe;
};
struct RArgs {
  RLayer L[RES_MAX_LAYERS]; int n_layers;
  const float *word, *pos, *sem; int sem_div; const float *emb_g, *emb_be; float eps;
  const bf16_t* vocab; int V;
  int d, H, ff, act, R, T, steps, bos, eos, pad, early, prof_step, ghost;
  int32_t* fed; int fed_stride; float* score; int32_t* length; int32_t* fin;
  unsigned* sync; float* xres; float* y; float* y2; float* q; bf16_t* ctx; bf16_t* h;
  float* pmax; int32_t* pidx; float* psum; int parts;
  // beam search (decode_resident_beam.hip): R = nclips * bm rows, `fed` = the token table addressed through the ancestor
  // tables (csrc/beam.hip), `score` = the running scores of the beam slots
  int bm, nclips, need, fin_cap;
  int32_t* anc[2]; int32_t* done; int32_t* nfin; float* fscore; int32_t* flen; int32_t* fhyp;
  float* gval; int32_t* ggid;  // per (row, vocabulary part): the RES_BMK best 4-column groups (maximum, group number)
  int vcap;                    // workgroups of the vocabulary phase (PhaseMap gcap): parts <= 48
  int fenced;                  // hand-offs with an agent-scope release / acquire pair (GridSync; care_resident_set_fenced)
  bf16_t* hn;                  // the normalised last hidden rows (bf16 [R16, 512]): the B operand of the recomputed logits
  bf16_t* xa;                  // beam search: the embedded + normalised input rows of the NEXT step (bf16 [R16, d]), written by the
                               // phase that chooses the tokens (beam_advance_phase / beam_init_phase): the QKV phase's A operand
};

// The argument block of a resident launch is ~150 pointers and sizes.  Taken by value into a kernel whose step loop calls
// ten inlined phases, its loop-invariant fields are hoisted out of the loop into SGPRs the kernel does not have (round 4:
// 560 - 600 SGPR spills, ~110 v_readlane per tile block of a phase's epilogue).  The resident kernels therefore read it
// where it lies - the kernarg segment (constant address space, s_load) - through a pointer that an empty asm statement
// makes opaque at every use: a field is loaded in the phase that needs it and lives no longer.
typedef const __attribute__((address_space(4))) RArgs* ResKArgs;
__device__ __forceinline__ const RArgs& res_args(ResKArgs k) {
  asm volatile("" : "+s"(k));
  return *(const RArgs*)k;
}
#define RES_KARGS() ((ResKArgs)__builtin_amdgcn_kernarg_segment_ptr())

// ---------------------------------------------------------------------------------------------------------------
// Data that one workgroup writes and another reads INSIDE the launch moves with agent-scope (sc1) accesses: stores
// write through to the device's coherence point, loads miss the caches that are not coherent across XCDs (a CU's L1,
// another XCD's L2) - relaxed 4- / 8-byte atomics, which is how the compiler spells them.  No cache-wide writeback or
// invalidate is needed then, and that is the point: *measured* (tools/micro/barrier_bench.hip, 256 workgroups) a
// barrier with __threadfence() on either side costs 17 us (one thread fences) to 39 us (every wave does), the
// counters alone 3.7 us, two-level counters 1.9 us.  Weights, embeddings and the cross K/V (written before the
// launch) use plain loads and stay in the L2s for all T steps.
// RES_PLAIN_IO (decode_chain.hip: the same phases as KERNELS of their own, one launch per phase): nothing is handed over
// inside a launch there, so every accessor below is a plain load / store (kernel boundaries order them) and GridSync is empty.
#ifdef RES_PLAIN_IO
#define RES_SC1 ""
__device__ __forceinline__ unsigned long long cld8(const void* p) { return *reinterpret_cast<const unsigned long long*>(p); }
__device__ __forceinline__ void cst8(void* p, unsigned long long v) { *reinterpret_cast<unsigned long long*>(p) = v; }
__device__ __forceinline__ float cld_f(const float* p) { return *p; }
__device__ __forceinline__ int cld_i(const int32_t* p) { return *p; }
__device__ __forceinline__ void cst_f(float* p, float v) { *p = v; }
__device__ __forceinline__ void cst_i(int32_t* p, int v) { *p = v; }
#else
#define RES_SC1 " sc1"
__device__ __forceinline__ unsigned long long cld8(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cst8(void* p, unsigned long long v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float cld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int cld_i(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cst_f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cst_i(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
struct U2 { unsigned long long a, b; };
__device__ __forceinline__ float4 cld_f4(const float* p) {
#ifdef RES_PLAIN_IO
  return *reinterpret_cast<const float4*>(p);
#else
  U2 u{cld8(p), cld8(p + 2)};
  return __builtin_bit_cast(float4, u);
#endif
}
// 16-byte write-through stores: ONE global_store_dwordx4 sc1 (an 8-byte agent-scope atomic store is the widest the
// compiler spells; two of them per 16 bytes are twice the instructions and, per MI355X_MICROARCH.md, 2.7 x the time per
// byte).  Inline asm: the compiler does not count it in its vmcnt bookkeeping - its own waits then wait for more, never
// for less (vmcnt retires in order) - and GridSync::arrive drains vmcnt(0) before it signals.
__device__ __forceinline__ void cst16(void* p, f32x4 v) {
#if defined(RES_PLAIN_IO)
  *reinterpret_cast<f32x4*>(p) = v;
#elif defined(RES_NO_ST16)
  const U2 u = __builtin_bit_cast(U2, v);
  cst8(p, u.a);
  cst8(reinterpret_cast<unsigned long long*>(p) + 1, u.b);
#else
  // (the s_nop: a VMEM store of more than 64 bits must not be followed directly by a write of its data registers - a
  // hazard the compiler resolves for its own stores and cannot see inside an asm statement; without it rows came out wrong)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void cst_f4(float* p, float4 v) { cst16(p, f32x4{v.x, v.y, v.z, v.w}); }
__device__ __forceinline__ bf16x8 cld_b8(const bf16_t* p) {
#ifdef RES_PLAIN_IO
  return *reinterpret_cast<const bf16x8*>(p);
#else
  U2 u{cld8(p), cld8(p + 4)};
  return __builtin_bit_cast(bf16x8, u);
#endif
}
__device__ __forceinline__ void cst_b8(bf16_t* p, bf16x8 v) { cst16(p, __builtin_bit_cast(f32x4, v)); }
__device__ __forceinline__ void cst_b4(bf16_t* p, bf16x4 v) { cst8(p, __builtin_bit_cast(unsigned long long, v)); }

// Synchronisation between phases: producer-counted hand-offs instead of full grid barriers.  Only the workgroups that
// HAVE WORK in a phase (`participants`) count themselves in when their stores are acknowledged - one atomic add on the
// phase's counter, sharded 8 ways by blockIdx % 8 (the XCD; 128-byte lines of their own) - and only the participants
// of the NEXT phase wait, polling the 8 shards until their sum reaches producers x executions.  Against the two-level
// barrier with a published generation (every workgroup: group counter -> top counter -> flag -> poll) that is one
// memory round trip less per phase and no waiting for workgroups that have nothing to do.  Why it is enough:
//   * a phase reads what the previous phase wrote, or older data whose writers the previous phase's producers waited
//     for in turn (residual rows, caches, decode state): the order is transitive along the chain of phases;
//   * a buffer is overwritten in phase k + 2 at the earliest by workgroups that waited for ALL producers of phase
//     k + 1, and the readers of the old contents in phase k + 1 are among those producers;
//   * every workgroup waits for the first phase of a step (the token choice): all of them read the same `rows ended`.
// sync (unsigned): [1] rows ended, [2] steps run, [33] abort flag, [64] verdict of the step (2 t + all rows ended),
// [512 ..] phase clocks (tools), [1024 + (32 slot + 4
// shard) * 8 ...]: counter of (slot, shard), slot = the phase's position within a step.
constexpr int RES_MAX_SLOTS = 48;
constexpr int RES_SYNC_BYTES = 4096 + RES_MAX_SLOTS * 8 * 128;
struct GridSync {
  unsigned* sync;
  unsigned ghost;  // tests: producers that never arrive are expected too (the watchdog)
  int slot;        // >= 0: workgroup 0 records the device clock at the phase boundaries of one step (sync + 512, 8-byte slots)
  bool dead;
  int cur, prev;   // position of this phase / of the phase it consumes within the step
  unsigned want;   // what the counters of `prev` must add up to: its producers x the times it has run (0: nothing to wait for)
  __device__ __forceinline__ void mark() {
    if (slot >= 0 && slot < 96 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(sync + 512)[slot++] = wall_clock64();
  }
  __device__ __forceinline__ unsigned* counter(int sl, int shard) const { return sync + 1024 + (sl * 8 + shard) * 32; }
  // The hand-off is the form MI355X_MICROARCH.md lists as measured-valid on gfx950 (visibility table, first row): EVERY
  // store of handed-off bytes is an sc1 (agent-scope, write-through) store, every storing wave drains them
  // (s_waitcnt vmcnt(0), as inline asm: also a compiler barrier, and invisible to the pass that drops a builtin wait),
  // the workgroup meets at its barrier (a workgroup-scope release / acquire pair for the compiler: no memory operation
  // moves across it), ONE lane adds to the phase's sharded counter; the consumer polls every shard with sc1 loads,
  // joins its workgroup's barrier, and EVERY load of handed-off bytes is an sc1 load to registers.  That is not what the
  // HIP memory model promises for relaxed atomics (a formal race; no cache-wide release / acquire is executed) - hence
  // the arch check below, the stress tests of tests/test_gpu_resident.py and `fenced`: the same protocol with an
  // agent-scope release before the add and an acquire after the poll (+ ~1 .. 4 us per hand-off, MI355X_MICROARCH.md fence
  // table; *measured* here: DESIGN.md 4.2d) - a RUN-TIME choice of every launch (RArgs::fenced): the host side takes the
  // fence-free form only on the configuration the stress tests validated (a gfx950 device with all 256 CUs in one
  // partition) unless care_resident_set_fenced says otherwise (res_fenced_for_device).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "decode_resident: the fence-free hand-off is validated for gfx950 only"
#endif
  bool fenced = false;
  __device__ __forceinline__ void arrive(bool participant) {
#ifdef RES_PLAIN_IO
    return;  // (the phase is a kernel of its own: the kernel boundary hands its output over)
#endif
    if (!participant) return;       // workgroup-uniform
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      if (fenced) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(counter(cur, blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // A workgroup that has spun for ~2 s (workgroups that never became resident: another kernel holds CUs for good, e.g.
  // a second resident launch on another stream) raises the abort flag; every workgroup then leaves at its next wait,
  // phases turn into no-ops and length[0] = -1 tells the host (care_decode_resident).
  __device__ __forceinline__ void wait() {
#ifdef RES_PLAIN_IO
    return;
#endif
    __shared__ int s_dead;
    if (want == 0) return;  // the first phase of the launch
    if (threadIdx.x < 64) {
      const unsigned need = want + ghost;
      const unsigned* c = counter(prev, threadIdx.x & 7);
      int d = 0;
      unsigned spins = 0;
      unsigned long long t0 = 0;
      for (;;) {
        unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // lanes 0 .. 7: the 8 shards
        // (the 8 shards through the DPP data path: lane ^ 1, lane ^ 2, the mirrored lane of the 8 - three dependent
        // ds_bpermute round trips per poll before)
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
        if (v >= need) break;  // (the same sum in every group of 8 lanes)
#ifndef RES_POLL_SLEEP
#define RES_POLL_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(RES_POLL_SLEEP);
        if ((++spins & 63u) == 0) {
          if (__hip_atomic_load(sync + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { d = 1; break; }
          if ((spins & 4095u) == 0) {
            const unsigned long long now = wall_clock64();  // 100 MHz
            if (!t0) t0 = now;
            else if (now - t0 > 200000000ull) {
              __hip_atomic_store(sync + 33, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              d = 1;
              break;
            }
          }
        }
      }
      if (threadIdx.x == 0) {
        s_dead = d;
        if (fenced) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    }
    __syncthreads();
    dead = dead || s_dead != 0;
  }
};

__device__ __forceinline__ float res_act(float v, int act) {
  if (act == CARE_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == CARE_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

// exp on the hardware's exp2 (v_exp_f32, ~1 ulp): the sum-exp of the vocabulary phase and the softmax weights of the
// attention phases, whose accurate expf was a third of their vector instructions; arguments are <= 0 there, exp(-inf) = 0.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// sum over the 16 lanes of a DPP row, in every lane: lane ^ 1, lane ^ 2, the mirrored lane of the 8-lane half, of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += care_dpp<0xB1>(v);
  v += care_dpp<0x4E>(v);
  v += care_dpp<0x141>(v);
  v += care_dpp<0x140>(v);
  return v;
}

__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// lane ^ 8 / ^ 16 / ^ 32 exchanges of the attention reductions without the LDS crossbar: row_ror:8 (DPP), and the
// gfx950 row / half swaps - v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of
// its second, v_permlane32_swap the upper half of the first with the lower half of the second; with both operands = v the
// two results are (lower partner, upper partner) in every lane.  a + b == b + a: the sums are those of __shfl_xor, bit for bit.
__device__ __forceinline__ float x16_sum(float v) {
  const int iv = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float x32_sum(float v) {
  const int iv = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float x16_max(float v) {
  const int iv = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
__device__ __forceinline__ float x32_max(float v) {
  const int iv = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}

// the partner lane's value of the lane ^ 16 / lane ^ 32 exchange (the swaps return (lower partner, upper partner))
__device__ __forceinline__ int x16_other(int v) {
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return (threadIdx.x & 16) ? (int)r[0] : (int)r[1];
}
__device__ __forceinline__ int x32_other(int v) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
}
__device__ __forceinline__ float x16_other(float v) { return __builtin_bit_cast(float, x16_other(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float x32_other(float v) { return __builtin_bit_cast(float, x32_other(__builtin_bit_cast(int, v))); }

// (max, arg-max, sum exp) merge; ties go to the lower column
__device__ __forceinline__ void amax_merge(float& m, int& i, float& s, float om, int oi, float os) {
  const float nm = fmaxf(m, om);
  const float a = m == -INFINITY ? 0.f : s * expf(m - nm);
  const float b = om == -INFINITY ? 0.f : os * expf(om - nm);
  if (om > m || (om == m && oi < i)) i = oi;
  m = nm;
  s = a + b;
}

// Wave-wide sum through the DPP data path (the scan of care_wave_max_dpp with + : row_shr 1, 2, 4, 8, then
// row_bcast15 / row_bcast31; lanes without a source add 0); every lane gets the total.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define RES_DPP_ADD(CTRL, ROWMASK) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false))
  RES_DPP_ADD(0x111, 0xF);
  RES_DPP_ADD(0x112, 0xF);
  RES_DPP_ADD(0x114, 0xF);
  RES_DPP_ADD(0x118, 0xF);
  RES_DPP_ADD(0x142, 0xA);
  RES_DPP_ADD(0x143, 0xC);
#undef RES_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sortable 64-bit keys (beam search): a float mapped to an unsigned that orders like the float (-inf lowest) in the
// high word, ~index in the low word - the maximum key = the largest value, the LOWEST index among equals (the order
// of torch.topk on distinct values / csrc/beam.hip: value desc, index asc).
__device__ __forceinline__ unsigned f_ord(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float f_unord(unsigned k) { return __builtin_bit_cast(float, k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }
__device__ __forceinline__ unsigned long long key_of(float v, unsigned idx) { return ((unsigned long long)f_ord(v) << 32) | (0xffffffffu - idx); }
__device__ __forceinline__ float key_val(unsigned long long k) { return f_unord((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned key_idx(unsigned long long k) { return 0xffffffffu - (unsigned)k; }
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long k) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xF, 0xF, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, true);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long max_u64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
// every lane of a 16-lane row gets the row's maximum: lane ^ 1, lane ^ 2 (quad_perm), then the mirrored lane of the
// 8-lane half and of the row - after each step the groups that meet hold their own maxima (no LDS round trip)
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long k) {
  k = max_u64(k, dpp_u64<0xB1>(k));
  k = max_u64(k, dpp_u64<0x4E>(k));
  k = max_u64(k, dpp_u64<0x141>(k));
  k = max_u64(k, dpp_u64<0x140>(k));
  return k;
}
// ... and every lane of the wave the wave's: the four row maxima through scalar registers
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  k = row16_max_u64(k);
  unsigned long long m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, q * 16);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), q * 16);
    m = max_u64(m, ((unsigned long long)hi << 32) | lo);
  }
  return m;
}

constexpr int RES_NP = 4;  // column-group partials per lane: V <= 64 * 64 * RES_NP

// Token choice of step `ts` for the 4 rows r0 .. r0 + 3 of a wave (all lanes get the tokens): reduce the
// column-group partials of the vocabulary phase (ties: the lowest column); WRITER: also advance the rows' state
// (Translator.py:91-109 / the top-1 of Beam.advance).  Every load of the 4 rows is issued before the first use.
template <bool WRITER>
__device__ __forceinline__ void select4(const RArgs& p, int r0, int ts, int lane, int (&tok)[4]) {
  float pm[4][RES_NP], ps[4][RES_NP];
  int pi[4][RES_NP], fin[4];
  float sc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + i;
    const bool rok = r < p.R;
#pragma unroll
    for (int k = 0; k < RES_NP; ++k) {
      const int c = lane + 64 * k;
      const bool ok = rok && c < p.parts;
      const int64_t o = (int64_t)(rok ? r : 0) * p.parts + (c < p.parts ? c : 0);
      pm[i][k] = cld_f(p.pmax + o);
      pi[i][k] = cld_i(p.pidx + o);
      if (WRITER) ps[i][k] = cld_f(p.psum + o);
      if (!ok) { pm[i][k] = -INFINITY; pi[i][k] = 0x7fffffff; if (WRITER) ps[i][k] = 0.f; }
    }
    if (WRITER) { fin[i] = cld_i(p.fin + (rok ? r : 0)); sc[i] = cld_f(p.score + (rok ? r : 0)); }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (r0 + i >= p.R) break;  // rows past the batch (wave-uniform)
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < RES_NP; ++k)
      if (pm[i][k] > best || (pm[i][k] == best && pi[i][k] < bi)) { best = pm[i][k]; bi = pi[i][k]; }
    {  // the wave's best (value desc, column asc) as the maximum of a sortable 64-bit key: DPP + readlane, no LDS round trips
      // (six dependent ds_bpermute pairs per row before)
      const unsigned long long kbest = wave_max_u64(key_of(best, (unsigned)bi));
      best = key_val(kbest);
      bi = (int)key_idx(kbest);
    }
    tok[i] = bi;
    if (WRITER) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < RES_NP; ++k) s += pm[i][k] == -INFINITY ? 0.f : ps[i][k] * expf(pm[i][k] - best);
      s = wave_sum_dpp(s);
      const int r = r0 + i;
      if (lane == 0 && r < p.R) {
        cst_i(p.fed + (int64_t)r * p.fed_stride + ts, bi);  // ended rows keep running, frozen
        if (!fin[i]) {
          cst_f(p.score + r, sc[i] - logf(s));
          cst_i(p.length + r, ts);
          if (bi == p.eos || ts >= p.T) {
            cst_i(p.fin + r, 1);
            // (the returned value is waited for: the count is at L2 BEFORE this workgroup's arrival is - every workgroup that reads it
            // behind the hand-off must come to the same `all rows ended` verdict; a fire-and-forget add may land after the arrival)
            const unsigned ended_before = __hip_atomic_fetch_add(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(ended_before));
          }
        }
      }
    }
  }
}

// The decode state of rows rb .. rb + 3 at the top of step t: fresh rows at t = 1 (fed = BOS, zeros), else the token
// choice of step t - 1 with its score / length / end-flag update.
__device__ __forceinline__ void advance_state4(const RArgs& p, int rb, int t, int lane, int (&tok)[4]) {
  if (t > 1) {
    select4<true>(p, rb, t - 1, lane, tok);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (rb + i < p.R) {
      const int r = rb + i;
      for (int c = lane; c <= p.T; c += 64) cst_i(p.fed + (int64_t)r * p.fed_stride + c, c == 0 ? p.bos : 0);
      if (lane == 0) { cst_f(p.score + r, 0.f); cst_i(p.length + r, 0); cst_i(p.fin + r, 0); }
    }
}
__device__ __forceinline__ void advance_state4(const RArgs& p, int rb, int t, int lane) {
  int tok[4];
  advance_state4(p, rb, t, lane, tok);
}

enum { A_EMBED = 0, A_LN = 1, A_BF16 = 2, A_EMBEDB = 3 };  // A_EMBEDB: beam search - the token is in the table already
enum { E_QKV = 0, E_Q = 1, E_RES = 2, E_ACT = 3, E_VOCAB = 4, E_VOCABK = 5 };  // E_VOCABK: beam search - best groups per row
// beam search: groups kept per (row, vocabulary part) = the largest beam size of the resident form.  5 everywhere (the reference's
// default beam) except in decode_resident_beam_wide.hip, which compiles decode_resident_beam.hip once more with 8 for beam sizes 6 .. 8
#ifndef CARE_RES_BMK
#define CARE_RES_BMK 5
#endif
constexpr int RES_BMK = CARE_RES_BMK;

// Beam search, step 1: the state of rows rb .. rb + 3 as the host-side initialisation of engine.beam leaves it - token
// table [BOS, EOS ...], both ancestor tables = the row itself, scores 0 - and, by the first row of a clip, the clip's
// flags and (zeroed) finished lists.
__device__ __forceinline__ void beam_init_rows4(const RArgs& p, int rb, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (rb + i < p.R) {
      const int r = rb + i;
      for (int c = lane; c <= p.T; c += 64) {
        cst_i(p.fed + (int64_t)r * p.fed_stride + c, c == 0 ? p.bos : p.eos);
        cst_i(p.anc[0] + (int64_t)r * p.fed_stride + c, r);
        cst_i(p.anc[1] + (int64_t)r * p.fed_stride + c, r);
      }
      if (lane == 0) cst_f(p.score + r, 0.f);
      if (r % p.bm == 0) {
        const int b = r / p.bm;
        if (lane == 0) { cst_i(p.done + b, 0); cst_i(p.nfin + b, 0); }
        for (int c = lane; c < p.fin_cap; c += 64) { cst_f(p.fscore + (int64_t)b * p.fin_cap + c, 0.f); cst_i(p.flen + (int64_t)b * p.fin_cap + c, 0); }
        for (int c = lane; c < p.fin_cap * p.fed_stride; c += 64) cst_i(p.fhyp + (int64_t)b * p.fin_cap * p.fed_stride + c, 0);
      }
    }
}

// 16 rows of the A operand -> LDS (bf16 [16][lda]), d = 512.  A_EMBED / A_LN: a wave owns 4 rows SIDE BY SIDE - the 16
// lanes of DPP row q take row 4 wave + q, lane `sub` its columns 4 sub + 64 k (k < 8: 256 contiguous bytes per row
// and load instruction) - so the two LayerNorm statistics of the four rows are ONE pair of 4-step DPP reductions
// (row16_sum) instead of eight 7-step wave reductions in series: the A stage was most of the time of every GEMM phase
// (*measured*, tools/resident_prof.py: 8.0 of 9.2 us in the QKV phase at 128 rows, 16.9 of 21.2 at 640 rows).
// All loads of the rows are issued together: fetch_a_rows requests them ...
// The token this lane's row (r0 + 4 wave + lane / 16) feeds at step t.  Requested for EVERY row tile of the workgroup
// before the first word row is fetched (gemm_phase): the word row depends on the token, vmcnt retires in issue order, so
// token -> word row per tile in turn is two memory round trips PER TILE in series (*measured* 640 rows, 4 tiles per
// workgroup: 19.8 us of the QKV phase).
template <int AMODE>
__device__ __forceinline__ int fetch_token(const RArgs& p, int r0, int t, bool writer) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4;
  const int rb = r0 + wave * 4, row = rb + q;
  int tok[4] = {p.bos, p.bos, p.bos, p.bos};
  if constexpr (AMODE == A_EMBED) {
    if (writer) advance_state4(p, rb, t, lane, tok);
    else if (t > 1) select4<false>(p, rb, t - 1, lane, tok);
    return row < p.R ? (q == 0 ? tok[0] : q == 1 ? tok[1] : q == 2 ? tok[2] : tok[3]) : 0;
  } else {  // beam search: beam_advance_phase wrote the token of slot r at position t - 1 (tokphys[r][t - 1], beam.hip)
    if (t == 1) {
      if (writer) beam_init_rows4(p, rb, lane);
      return row < p.R ? p.bos : 0;
    }
    const int tk = cld_i(p.fed + (int64_t)(row < p.R ? row : 0) * p.fed_stride + (t - 1));
    return row < p.R ? tk : 0;
  }
}

// NV 16-byte pieces of a handed-over fp32 row (base + 256 k bytes) as NV global_load_dwordx4 sc1 - the widest agent-scope
// atomic load the compiler spells is 8 bytes, two instructions per piece, and a row group with the second FFN half in
// flight was 64 load instructions per lane (vmcnt counts 63).  ONE asm statement issues the loads AND waits for them: the
// compiler cannot know when an asm load's destination becomes valid, so nothing may touch it in between (plain loads
// requested before the statement stay in flight under it; the wait covers them too).
template <int NV>
__device__ __forceinline__ void cld16_row(const float* base, float4 (&v)[NV]) {
  static_assert(NV == 4 || NV == 8 || NV == 12 || NV == 16, "row pieces");
  f32x4 r[NV];
  if constexpr (NV == 4) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %4, off offset:256" RES_SC1 "\n\t"
        "global_load_dwordx4 %2, %4, off offset:512" RES_SC1 "\n\tglobal_load_dwordx4 %3, %4, off offset:768" RES_SC1 "\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(base)
        : "memory");
  } else if constexpr (NV == 8) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %8, off offset:256" RES_SC1 "\n\t"
        "global_load_dwordx4 %2, %8, off offset:512" RES_SC1 "\n\tglobal_load_dwordx4 %3, %8, off offset:768" RES_SC1 "\n\t"
        "global_load_dwordx4 %4, %8, off offset:1024" RES_SC1 "\n\tglobal_load_dwordx4 %5, %8, off offset:1280" RES_SC1 "\n\t"
        "global_load_dwordx4 %6, %8, off offset:1536" RES_SC1 "\n\tglobal_load_dwordx4 %7, %8, off offset:1792" RES_SC1 "\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(base)
        : "memory");
  } else if constexpr (NV == 12) {
    asm volatile(
        "global_load_dwordx4 %0, %12, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %12, off offset:256" RES_SC1 "\n\t"
        "global_load_dwordx4 %2, %12, off offset:512" RES_SC1 "\n\tglobal_load_dwordx4 %3, %12, off offset:768" RES_SC1 "\n\t"
        "global_load_dwordx4 %4, %12, off offset:1024" RES_SC1 "\n\tglobal_load_dwordx4 %5, %12, off offset:1280" RES_SC1 "\n\t"
        "global_load_dwordx4 %6, %12, off offset:1536" RES_SC1 "\n\tglobal_load_dwordx4 %7, %12, off offset:1792" RES_SC1 "\n\t"
        "global_load_dwordx4 %8, %12, off offset:2048" RES_SC1 "\n\tglobal_load_dwordx4 %9, %12, off offset:2304" RES_SC1 "\n\t"
        "global_load_dwordx4 %10, %12, off offset:2560" RES_SC1 "\n\tglobal_load_dwordx4 %11, %12, off offset:2816" RES_SC1 "\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
          "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11])
        : "v"(base)
        : "memory");
  } else {
    asm volatile(
        "global_load_dwordx4 %0, %16, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %16, off offset:256" RES_SC1 "\n\t"
        "global_load_dwordx4 %2, %16, off offset:512" RES_SC1 "\n\tglobal_load_dwordx4 %3, %16, off offset:768" RES_SC1 "\n\t"
        "global_load_dwordx4 %4, %16, off offset:1024" RES_SC1 "\n\tglobal_load_dwordx4 %5, %16, off offset:1280" RES_SC1 "\n\t"
        "global_load_dwordx4 %6, %16, off offset:1536" RES_SC1 "\n\tglobal_load_dwordx4 %7, %16, off offset:1792" RES_SC1 "\n\t"
        "global_load_dwordx4 %8, %16, off offset:2048" RES_SC1 "\n\tglobal_load_dwordx4 %9, %16, off offset:2304" RES_SC1 "\n\t"
        "global_load_dwordx4 %10, %16, off offset:2560" RES_SC1 "\n\tglobal_load_dwordx4 %11, %16, off offset:2816" RES_SC1 "\n\t"
        "global_load_dwordx4 %12, %16, off offset:3072" RES_SC1 "\n\tglobal_load_dwordx4 %13, %16, off offset:3328" RES_SC1 "\n\t"
        "global_load_dwordx4 %14, %16, off offset:3584" RES_SC1 "\n\tglobal_load_dwordx4 %15, %16, off offset:3840" RES_SC1 "\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
          "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
        : "v"(base)
        : "memory");
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = make_float4(r[k][0], r[k][1], r[k][2], r[k][3]);
}

// ... two rows of 8 pieces (y and the second FFN half y2) in ONE statement: 16 loads in flight, one wait
__device__ __forceinline__ void cld16_row2(const float* a, const float* b, float4 (&v)[8], float4 (&w)[8]) {
  f32x4 r[16];
  asm volatile(
      "global_load_dwordx4 %0, %16, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %16, off offset:256" RES_SC1 "\n\t"
      "global_load_dwordx4 %2, %16, off offset:512" RES_SC1 "\n\tglobal_load_dwordx4 %3, %16, off offset:768" RES_SC1 "\n\t"
      "global_load_dwordx4 %4, %16, off offset:1024" RES_SC1 "\n\tglobal_load_dwordx4 %5, %16, off offset:1280" RES_SC1 "\n\t"
      "global_load_dwordx4 %6, %16, off offset:1536" RES_SC1 "\n\tglobal_load_dwordx4 %7, %16, off offset:1792" RES_SC1 "\n\t"
      "global_load_dwordx4 %8, %17, off" RES_SC1 "\n\tglobal_load_dwordx4 %9, %17, off offset:256" RES_SC1 "\n\t"
      "global_load_dwordx4 %10, %17, off offset:512" RES_SC1 "\n\tglobal_load_dwordx4 %11, %17, off offset:768" RES_SC1 "\n\t"
      "global_load_dwordx4 %12, %17, off offset:1024" RES_SC1 "\n\tglobal_load_dwordx4 %13, %17, off offset:1280" RES_SC1 "\n\t"
      "global_load_dwordx4 %14, %17, off offset:1536" RES_SC1 "\n\tglobal_load_dwordx4 %15, %17, off offset:1792" RES_SC1 "\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
        "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
      : "v"(a), "v"(b)
      : "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = make_float4(r[k][0], r[k][1], r[k][2], r[k][3]);
    w[k] = make_float4(r[8 + k][0], r[8 + k][1], r[8 + k][2], r[8 + k][3]);
  }
}

// ... N (4 or 8) pieces at addresses of their own / 16 pieces at 8 bases -+ 2048 bytes (two tile rows 4096 bytes apart)
__device__ __forceinline__ void cld16_x4(const void* a0, const void* a1, const void* a2, const void* a3, bf16x8 (&v)[4]) {
  f32x4 r[4];
  asm volatile(
      "global_load_dwordx4 %0, %4, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %5, off" RES_SC1 "\n\t"
      "global_load_dwordx4 %2, %6, off" RES_SC1 "\n\tglobal_load_dwordx4 %3, %7, off" RES_SC1 "\n\ts_waitcnt vmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
      : "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(bf16x8, r[k]);
}
__device__ __forceinline__ void cld16_x8(const void* const (&a)[8], bf16x8 (&v)[8]) {
  f32x4 r[8];
  asm volatile(
      "global_load_dwordx4 %0, %8, off" RES_SC1 "\n\tglobal_load_dwordx4 %1, %9, off" RES_SC1 "\n\t"
      "global_load_dwordx4 %2, %10, off" RES_SC1 "\n\tglobal_load_dwordx4 %3, %11, off" RES_SC1 "\n\t"
      "global_load_dwordx4 %4, %12, off" RES_SC1 "\n\tglobal_load_dwordx4 %5, %13, off" RES_SC1 "\n\t"
      "global_load_dwordx4 %6, %14, off" RES_SC1 "\n\tglobal_load_dwordx4 %7, %15, off" RES_SC1 "\n\ts_waitcnt vmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
      : "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = __builtin_bit_cast(bf16x8, r[k]);
}
// piece i at `first + 4096 i` bytes, i < 16 (the 16 rows of a K = 2048 bf16 tile): 8 bases, offsets -2048 / +2048
__device__ __forceinline__ void cld16_x16_rows4k(const unsigned char* first, bf16x8 (&v)[16]) {
  f32x4 r[16];
  const unsigned char* b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = first + 8192 * j + 2048;
  asm volatile(
      "global_load_dwordx4 %0, %16, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %1, %16, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %2, %17, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %3, %17, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %4, %18, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %5, %18, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %6, %19, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %7, %19, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %8, %20, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %9, %20, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %10, %21, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %11, %21, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %12, %22, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %13, %22, off offset:2048" RES_SC1 "\n\t"
      "global_load_dwordx4 %14, %23, off offset:-2048" RES_SC1 "\n\tglobal_load_dwordx4 %15, %23, off offset:2048" RES_SC1 "\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
        "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
      : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
      : "memory");
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = __builtin_bit_cast(bf16x8, r[k]);
}

template <int AMODE, int D = 512>
__device__ __forceinline__ void fetch_a_rows(const RArgs& p, int r0, int t, int mytok, const float* ysrc, const float* ysrc2,
                                             float4 (&v)[D / 64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, sub = lane & 15;
  constexpr int d = D, NV = D / 64;
  const int rb = r0 + wave * 4, row = rb + q, rc = row < p.R ? row : 0;
  if constexpr (AMODE == A_EMBED || AMODE == A_EMBEDB) {
    const float* pp = p.pos + (int64_t)(t - 1) * d + sub * 4;
    const float* w = p.word + (int64_t)mytok * d + sub * 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k] = *reinterpret_cast<const float4*>(w + 64 * k);
      add4(v[k], *reinterpret_cast<const float4*>(pp + 64 * k));
    }
    if (p.sem) {
      const float* sm = p.sem + (int64_t)(rc / p.sem_div) * d + sub * 4;
#pragma unroll
      for (int k = 0; k < NV; ++k) add4(v[k], *reinterpret_cast<const float4*>(sm + 64 * k));
    }
  } else {
#ifdef RES_LD8
    const float* y = ysrc + (int64_t)rc * d + sub * 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = cld_f4(y + 64 * k);
    if (ysrc2) {  // the second K half of a two-workgroup FFN dense2 (ffn2_phase<true>): y = y + y2
      const float* y2 = ysrc2 + (int64_t)rc * d + sub * 4;
#pragma unroll
      for (int k = 0; k < NV; ++k) add4(v[k], cld_f4(y2 + 64 * k));
    }
#else
    if (ysrc2) {  // the second K half of a two-workgroup FFN dense2 (ffn2_phase<true>): y = y + y2
      if constexpr (NV == 8) {
        float4 w2[NV];
        cld16_row2(ysrc + (int64_t)rc * d + sub * 4, ysrc2 + (int64_t)rc * d + sub * 4, v, w2);
#pragma unroll
        for (int k = 0; k < NV; ++k) add4(v[k], w2[k]);
      } else {
        // d_model 768 / 1024: the second row arrives in pieces of 8 (+ 4) float4 added as they land - two whole rows of 12 /
        // 16 pieces side by side (+ the asm's own outputs) were 128 - 192 registers of the A stage (round 4's scratch)
        cld16_row<NV>(ysrc + (int64_t)rc * d + sub * 4, v);
        float4 w8[8];
        cld16_row<8>(ysrc2 + (int64_t)rc * d + sub * 4, w8);
#pragma unroll
        for (int k = 0; k < 8; ++k) add4(v[k], w8[k]);
        if constexpr (NV == 16) {
          cld16_row<8>(ysrc2 + (int64_t)rc * d + sub * 4 + 512, w8);
#pragma unroll
          for (int k = 0; k < 8; ++k) add4(v[8 + k], w8[k]);
        } else {
          float4 w4[4];
          cld16_row<4>(ysrc2 + (int64_t)rc * d + sub * 4 + 512, w4);
#pragma unroll
          for (int k = 0; k < 4; ++k) add4(v[8 + k], w4[k]);
        }
      }
    } else {
      cld16_row<NV>(ysrc + (int64_t)rc * d + sub * 4, v);
    }
#endif
  }
}

// ... and their LayerNorm into the LDS tile (+ the fp32 rows for the residual when write_x, + the bf16 rows when write_hn)
// d_model 512: in registers (16 float4).  d_model 768 / 1024: through LDS - 24 / 32 float4 per lane next to the row pieces
// were what spilled in the wide instances (round 4: 216 / 412 B of scratch per lane); an LDS read does not queue behind the
// loop's write-through stores either (lgkmcnt, not vmcnt).
template <int D = 512>
struct LnGB {
  float4 g[D > 512 ? 1 : D / 64], b[D > 512 ? 1 : D / 64];
  const float* sg; const float* sb;
  __device__ __forceinline__ float4 G(int k) const {
    if constexpr (D > 512) return *reinterpret_cast<const float4*>(sg + (threadIdx.x & 15) * 4 + 64 * k);
    else return g[k];
  }
  __device__ __forceinline__ float4 B(int k) const {
    if constexpr (D > 512) return *reinterpret_cast<const float4*>(sb + (threadIdx.x & 15) * 4 + 64 * k);
    else return b[k];
  }
};
template <int D = 512>
__device__ __forceinline__ void load_gb(LnGB<D>& w, const float* g, const float* be) {
  if constexpr (D > 512) {
    __shared__ __attribute__((aligned(16))) float s_g[D], s_b[D];
    __syncthreads();  // (the previous phase's readers of the two rows are past them)
    for (int i = threadIdx.x; i < D / 4; i += 256) {
      reinterpret_cast<float4*>(s_g)[i] = reinterpret_cast<const float4*>(g)[i];
      reinterpret_cast<float4*>(s_b)[i] = reinterpret_cast<const float4*>(be)[i];
    }
    __syncthreads();
    w.sg = s_g; w.sb = s_b;
  } else {
    const int sub = threadIdx.x & 15;
#pragma unroll
    for (int k = 0; k < D / 64; ++k) {
      w.g[k] = *reinterpret_cast<const float4*>(g + sub * 4 + 64 * k);
      w.b[k] = *reinterpret_cast<const float4*>(be + sub * 4 + 64 * k);
    }
  }
}
// (the weights come in registers, requested before the rows: loaded inside the loop below they would each wait behind
// the loop's write-through stores - the compiler keeps a plain load behind an atomic store it cannot tell apart from it)
// write_x / write_hn: bit k set = this workgroup stores column slice k (columns 4 sub + 64 k) of the normalised rows -
// the fp32 rows (the residual of the phase after next) / the bf16 rows (beam search: the advance phase's B operand).  The
// first 8 workgroups of a row group take a slice each (gemm_phase): one workgroup writing all 2 KB of every row in
// write-through stores was what the phase's consumers waited for (*measured* 640 rows: 17.6 us in the QKV phase).
template <int D = 512>
__device__ __forceinline__ void finish_a_rows(const RArgs& p, int r0, const float4 (&v)[D / 64], const LnGB<D>& w,
                                              unsigned write_x, bf16_t* sA, int lda, unsigned write_hn = 0u) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, sub = lane & 15;
  constexpr int d = D, NV = D / 64;
  const int rr = wave * 4 + q, r = r0 + rr;
  const bool live = r < p.R;  // rows past the batch: zeros into the tile
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  const float mean = row16_sum(s) * (1.0f / d);
  float qq = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, e = v[k].w - mean;
    qq += (a * a + b * b) + (cc * cc + e * e);
  }
  const float var = row16_sum(qq) * (1.0f / d);
  const float rstd = 1.0f / sqrtf(var + p.eps);
  bf16_t* dst = sA + rr * lda + sub * 4;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float4 gg = w.G(k), bb = w.B(k);
    float4 o;
    o.x = (v[k].x - mean) * rstd * gg.x + bb.x;
    o.y = (v[k].y - mean) * rstd * gg.y + bb.y;
    o.z = (v[k].z - mean) * rstd * gg.z + bb.z;
    o.w = (v[k].w - mean) * rstd * gg.w + bb.w;
    bf16x4 ob;
    ob[0] = (bf16_t)o.x; ob[1] = (bf16_t)o.y; ob[2] = (bf16_t)o.z; ob[3] = (bf16_t)o.w;
    if (!live) ob = bf16x4{};
    *reinterpret_cast<bf16x4*>(dst + 64 * k) = ob;
    if (live && ((write_x >> (k & 7)) & 1u)) cst_f4(p.xres + (int64_t)r * d + sub * 4 + 64 * k, o);
    if (live && ((write_hn >> (k & 7)) & 1u)) cst_b4(p.hn + (int64_t)r * d + sub * 4 + 64 * k, ob);
  }
}

// LayerNorm ONCE (round 6): Embeddings.forward (models/components/Embeddings.py:134-188) of ONE row per 16-lane DPP row - the
// word row of `tok`, position `tpos`, the clip's guidance vector, LayerNorm - stored as the fp32 residual row (xres) and the
// bf16 row the QKV phase multiplies (xa).  Called by the phase that CHOOSES the token (beam search: one workgroup per clip),
// so the ~250 workgroups of the next QKV phase load 1 KB of finished bf16 per row instead of each repeating token -> word row
// -> statistics for its 16 - 64 rows (*measured* round 5, 640 rows: 10.3 us of that phase).  Layout and arithmetic are
// fetch_a_rows<A_EMBEDB> + finish_a_rows: lane `sub` of the row holds columns 4 sub + 64 k.
template <int D = 512>
__device__ __forceinline__ void embed_ln_store_row(const RArgs& p, int row, bool valid, int tok, int tpos) {
  const int sub = threadIdx.x & 15;
  constexpr int NV = D / 64;
  const int rc = valid ? row : 0;
  // every field of the argument block ONCE (each use of `p` is a scalar load of its own, in program order: res_args) and every
  // vector load - word, position and guidance rows, LayerNorm weights - requested before the first is used: one memory round
  // trip for the whole row.  (The first version read gamma / beta inside the store loop: eight dependent round trips, 10 us.)
  const float* word = p.word; const float* pos = p.pos; const float* sem = p.sem; const int sem_div = p.sem_div;
  const float* eg = p.emb_g; const float* eb = p.emb_be; const float eps = p.eps;
  float* xres = p.xres; bf16_t* xa = p.xa;
  float4 v[NV], pv[NV], sv[NV], gg[NV], bb[NV];
  const float* pp = pos + (int64_t)tpos * D + sub * 4;
  const float* w = word + (int64_t)(valid ? tok : 0) * D + sub * 4;
  const float* sm = sem ? sem + (int64_t)(rc / sem_div) * D + sub * 4 : pp;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k] = *reinterpret_cast<const float4*>(w + 64 * k);
    pv[k] = *reinterpret_cast<const float4*>(pp + 64 * k);
    sv[k] = *reinterpret_cast<const float4*>(sm + 64 * k);
    if constexpr (D <= 512) {
      gg[k] = *reinterpret_cast<const float4*>(eg + sub * 4 + 64 * k);
      bb[k] = *reinterpret_cast<const float4*>(eb + sub * 4 + 64 * k);
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    add4(v[k], pv[k]);
    if (sem) add4(v[k], sv[k]);
  }
  if constexpr (D > 512) {  // (d_model 768 / 1024: five row images at once are 240 - 320 registers; the weights travel under the statistics)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      gg[k] = *reinterpret_cast<const float4*>(eg + sub * 4 + 64 * k);
      bb[k] = *reinterpret_cast<const float4*>(eb + sub * 4 + 64 * k);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  const float mean = row16_sum(s) * (1.0f / D);
  float qq = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, e = v[k].w - mean;
    qq += (a * a + b * b) + (cc * cc + e * e);
  }
  const float var = row16_sum(qq) * (1.0f / D);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float4 o;
    o.x = (v[k].x - mean) * rstd * gg[k].x + bb[k].x;
    o.y = (v[k].y - mean) * rstd * gg[k].y + bb[k].y;
    o.z = (v[k].z - mean) * rstd * gg[k].z + bb[k].z;
    o.w = (v[k].w - mean) * rstd * gg[k].w + bb[k].w;
    bf16x4 ob;
    ob[0] = (bf16_t)o.x; ob[1] = (bf16_t)o.y; ob[2] = (bf16_t)o.z; ob[3] = (bf16_t)o.w;
    if (valid) {
      cst_f4(xres + (int64_t)row * D + sub * 4 + 64 * k, o);
      cst_b4(xa + (int64_t)row * D + sub * 4 + 64 * k, ob);
    }
  }
}

template <int K>
__device__ __forceinline__ void load_a_bf16(const RArgs& p, int r0, const bf16_t* src, bf16_t* sA, int lda) {
  constexpr int per_row = K / 8, NC = 16 * per_row / 256;  // 16-byte chunks per thread, all in flight together
  bf16x8 v[NC];
  if constexpr (NC == 4 || NC == 8) {
    const void* a[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = threadIdx.x + 256 * i, rr = c / per_row, c8 = c - rr * per_row;
      a[i] = src + (int64_t)(r0 + rr < p.R ? r0 + rr : 0) * K + c8 * 8;
    }
    if constexpr (NC == 4) cld16_x4(a[0], a[1], a[2], a[3], v);
    else cld16_x8(a, v);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = threadIdx.x + 256 * i, rr = c / per_row;
      if (r0 + rr >= p.R) v[i] = bf16x8{};
    }
  } else {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = threadIdx.x + 256 * i, rr = c / per_row, c8 = c - rr * per_row;
      v[i] = cld_b8(src + (int64_t)(r0 + rr < p.R ? r0 + rr : 0) * K + c8 * 8);
      if (r0 + rr >= p.R) v[i] = bf16x8{};
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = threadIdx.x + 256 * i, rr = c / per_row, c8 = c - rr * per_row;
    *reinterpret_cast<bf16x8*>(sA + rr * lda + c8 * 8) = v[i];
  }
}

template <int NF>
__device__ __forceinline__ void load_w(bf16x8 (&wf)[NF], const bf16_t* wp) {
#pragma unroll
  for (int i = 0; i < NF; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wp + i * 32);
}

// Which (row tile, column items) a workgroup takes in a GEMM phase.  Workgroups are dealt to the 8 XCDs round-robin
// (blockIdx % 8) and each XCD has an L2 of its own, so the column items - the WEIGHT slices - follow the XCD: the
// workgroups of XCD x take items x, x + 8, ... for every row tile, and a weight slice is read into one L2 only,
// where it stays for all T steps (by row tile first, every XCD would stream the whole vocabulary matrix per step).
// helper: when the phase leaves workgroups without an item, one of them per row tile (XCD 7's first idle slot) is the
// row tile's `helper` - it advances the rows' decode state (score, length, end flags: select4<true>) off the path of
// the workgroups that multiply; otherwise (`helped` false) the workgroup of column item 0 does that too.
struct PhaseMap {
  int rt, c0, nper; bool has, helped, helper;
  // gcap > 0: only the first gcap workgroups take part (the others have no item) - beam search caps the vocabulary
  // phase so that a row has at most 48 partial lists for the advance phase to merge
  __device__ __forceinline__ PhaseMap(int RT, int CI, int gcap = 0) {
    const int G = gcap > 0 && gcap < (int)gridDim.x ? gcap : (int)gridDim.x, b = blockIdx.x;
    helped = helper = false;
    if ((G & 7) == 0 && (G >> 3) >= RT) {
      const int x = b & 7, j = b >> 3, nsl = (G >> 3) / RT, cs = j / RT;
      rt = j - cs * RT; c0 = x + 8 * cs; nper = 8 * nsl; has = cs < nsl && c0 < CI;
      const int csh = CI > 7 ? (CI - 7 + 7) >> 3 : 0;  // first slot of XCD 7 without an item: 7 + 8 csh >= CI
      helped = csh < nsl;
      helper = helped && x == 7 && cs == csh;
    } else {
      nper = G / RT; rt = b % RT; c0 = b / RT; has = c0 < nper && c0 < CI;
    }
    if (b >= G) has = helper = false;
    np_ = (unsigned)(RT * (nper < CI ? nper : CI));
    nh_ = helped ? (unsigned)RT : 0u;
  }
  unsigned np_, nh_;  // workgroups with an item / helpers in this phase (the same numbers in every workgroup)
};

// One GEMM phase: out[R, N] = A[R, K] W[N, K]^T (+ bias, epilogue EPI), K = 512 * KC.  A workgroup loads (and
// normalises) the 16 A rows of its row tile ONCE and walks its column items with two sets of W fragments: the next
// item's travel while the current one is multiplied.
//   KSPLIT = false: item = 64 columns, a wave owns a 16x16 output tile over the whole K (the vocabulary phase; QKV and
//                   FFN dense1 above 64 rows);
//   KSPLIT = true : item = 16 columns, the 4 waves split K and wave 0 adds the partial tiles through LDS - 4x the
//                   items, 1/4 of the W bytes per wave (what a GEMM over a few rows waits for is its CU's read rate
//                   from L2): the N = 512 phases always, QKV and FFN dense1 up to 64 rows.
// E_VOCAB keeps a running (max, arg-max, sum exp) per lane over the workgroup's items and merges lanes and waves once,
// after the last item: one partial per (row, workgroup of the row tile), p.parts of them per row.
template <int K, int AMODE, int EPI, bool KSPLIT, int RTB = 1, int D = 512>
RES_PHASE_FN unsigned gemm_phase(const RArgs& p, GridSync& gs, bool do_wait, bf16_t* sA, const bf16_t* W,
                                           const float* bias, int N, const void* asrc, const float* g, const float* be,
                                           bool write_x, int t, bf16_t* skv, const float* asrc2 = nullptr, int gcap = 0) {
  // RTB: 16-row tiles a workgroup multiplies with ONE fetch of its W fragments (their A rows side by side in LDS):
  // the weight traffic of a phase is (row tiles / RTB) x the matrix - what bounds the vocabulary phase at 128 rows.
  constexpr int NF = KSPLIT ? K / 128 : K / 32;  // (K: the reduction length; D: the model width = the row length of q / y / xres)
  constexpr int lda = K + 8;
  static_assert(NF <= 16, "two sets of W fragments: <= 128 VGPRs");
  __shared__ float s_pm[4][16];
  __shared__ int s_pi[4][16];
  __shared__ float s_ps[4][16];
  __shared__ f32x4 s_red[2][3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kg = lane >> 4;
  const int RT = (p.R + 15) >> 4, RG = (RT + RTB - 1) / RTB, CI = KSPLIT ? (N + 15) >> 4 : (N + 63) >> 6;
  const PhaseMap pm(RG, CI, gcap);
  const int r0 = pm.rt * 16 * RTB;
  // this wave's tile of column item c: columns n0(c) .. + 16; its K range starts at koff
  const int koff = KSPLIT ? wave * (K / 4) : 0;
  auto tile_n0 = [&](int c) { return KSPLIT ? c * 16 : (c * 4 + wave) * 16; };
  // (unconditional: an item past the last one re-reads the last one's rows.  A branch around the loads would make the
  // compiler wait for ALL outstanding loads - vmcnt(0) - at its join, i.e. for the fragments just requested, before the
  // MFMAs of the current item)
  auto fetch = [&](bf16x8 (&wf)[NF], int c) {
    load_w<NF>(wf, W + (int64_t)min(tile_n0(min(c, CI - 1)) + l16, N - 1) * K + koff + kg * 8);
  };
  float vm[RTB], vs[RTB];  // E_VOCAB: the lane's running partial per row tile
  int vi[RTB];
#pragma unroll
  for (int u = 0; u < RTB; ++u) { vm[u] = -INFINITY; vs[u] = 0.f; vi[u] = 0x7fffffff; }
  // E_VOCABK: the lane's RES_BMK best 4-column groups so far per row tile (maximum of the group desc; the lane's items
  // run through ascending columns and a later group enters on `>` only, so equal maxima keep the lower group).  The
  // row's bm best logits lie in its bm best groups: the bm-th largest group maximum is a lower bound of the bm-th best
  // logit, and a logit at least that large makes its group's maximum at least that large.
  constexpr int NL = EPI == E_VOCABK ? RES_BMK : 1;
  float lv[RTB][NL];
  int lg[RTB][NL];
#pragma unroll
  for (int u = 0; u < RTB; ++u)
#pragma unroll
    for (int k = 0; k < NL; ++k) { lv[u][k] = -INFINITY; lg[u][k] = 0x7fffffff; }
  int par = 0;
  // the wave's A fragments (its K range of the 16 rows): read from LDS once per phase and kept in registers - every
  // item of the workgroup multiplies the same rows (per row tile when RTB > 1)
  bf16x8 af[NF];
  auto load_af = [&](int u) {
    const bf16_t* ar = sA + (u * 16 + l16) * lda + koff + kg * 8;
#pragma unroll
    for (int i = 0; i < NF; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ar + i * 32);
  };
  // what an item's epilogue reads from memory - its bias, the residual rows - is requested by `preload` BEFORE the next
  // item's weight fragments: vmcnt retires in issue order, so a load issued behind the prefetch would make the epilogue
  // wait for the whole prefetch (the double buffer was no buffer in every phase with several items per workgroup)
  struct Pre { float4 bv; float4 xr[RTB]; };
  auto preload = [&](Pre& P, int c) {
    const int n0 = tile_n0(c), nb = n0 + kg * 4;
    P.bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI != E_VOCAB && EPI != E_VOCABK)
      if (n0 < N) P.bv = *reinterpret_cast<const float4*>(bias + nb);
#pragma unroll
    for (int u = 0; u < RTB; ++u) {
      P.xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (EPI == E_RES) {
        const int r = r0 + u * 16 + l16;
        if ((!KSPLIT || wave == 0) && n0 < N && r < p.R) P.xr[u] = cld_f4(p.xres + (int64_t)r * D + nb);
      }
    }
  };
  auto item = [&](const bf16x8 (&wf)[NF], int c, const Pre& P) {
    const int n0 = tile_n0(c), nb = n0 + kg * 4;
    const bool active = n0 < N;
#pragma unroll
    for (int u = 0; u < RTB; ++u) {
      const int r = r0 + u * 16 + l16;  // lane: row r, columns nb .. nb + 3 of its wave's tile
      if (RTB > 1 && r0 + u * 16 >= p.R) break;
      const float4 xr = P.xr[u];
      // K in quarters, each with two accumulator chains (even / odd fragments), added as q0 + ((q1 + q2) + q3): the
      // order in which the K-split form adds its four waves' tiles, so both forms give the same bits
      constexpr int NQ = (KSPLIT || EPI == E_VOCAB || EPI == E_VOCABK) ? 1 : 4, QF = NF / NQ;  // (the vocabulary phase has one form)
      f32x4 part[NQ];
      {
        if constexpr (RTB > 1) load_af(u);
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          if (active) {
#pragma unroll
            for (int i = qq * QF; i < (qq + 1) * QF; i += 2) {
              acc0 = care_mfma_16x16x32_h16(wf[i], af[i], acc0, 0, 0, 0);
              acc1 = care_mfma_16x16x32_h16(wf[i + 1], af[i + 1], acc1, 0, 0, 0);
            }
          }
          part[qq] = acc0 + acc1;
        }
      }
      f32x4 v = part[0];
      if constexpr (NQ == 4) v = part[0] + ((part[1] + part[2]) + part[3]);
      if constexpr (KSPLIT) {
        if (wave > 0) s_red[par][wave - 1][lane] = v;
        __syncthreads();  // one per tile: the partial tiles alternate between two buffers
        par ^= 1;
        if (wave > 0) continue;
        v = v + ((s_red[par ^ 1][0][lane] + s_red[par ^ 1][1][lane]) + s_red[par ^ 1][2][lane]);
      }
      if constexpr (EPI == E_VOCABK) {
        // beam search: the 16 columns of the wave's tile all exist except in the last item of the matrix (wave-uniform):
        // no per-element bounds, no column of the maximum (the group number is the lane's), one max3 chain; the sum of
        // exponentials runs against the lane's running maximum (rescaled when it moves - a later maximum is >= : no overflow)
        if (active) {
          const bool full = n0 + 16 <= N;
          float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
          if (!full) {
            if (nb + 0 >= N) x0 = -INFINITY;
            if (nb + 1 >= N) x1 = -INFINITY;
            if (nb + 2 >= N) x2 = -INFINITY;
            if (nb + 3 >= N) x3 = -INFINITY;
          }
          const float m4 = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
          {  // branch-free insertion of (m4, group) into the sorted list
            float x = m4;
            int gx = nb >> 2;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
              const bool gt = x > lv[u][k];
              const float nl = gt ? x : lv[u][k], nx = gt ? lv[u][k] : x;
              const int ng = gt ? gx : lg[u][k], ngx = gt ? lg[u][k] : gx;
              lv[u][k] = nl; lg[u][k] = ng; x = nx; gx = ngx;
            }
          }
          if (m4 > vm[u]) {
            vs[u] = vm[u] == -INFINITY ? 0.f : vs[u] * fexp(vm[u] - m4);
            vm[u] = m4;
          }
          if (vm[u] != -INFINITY) vs[u] += (fexp(x0 - vm[u]) + fexp(x1 - vm[u])) + (fexp(x2 - vm[u]) + fexp(x3 - vm[u]));
        }
      } else if constexpr (EPI == E_VOCAB) {
        if (active) {
          float m4 = -INFINITY;
          int i4 = 0x7fffffff;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nb + e < N && v[e] > m4) { m4 = v[e]; i4 = nb + e; }
          if constexpr (EPI == E_VOCABK) {  // (unreachable: the beam form is above)
            float x = m4;
            int gx = nb >> 2;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
              const bool gt = x > lv[u][k];
              const float nl = gt ? x : lv[u][k], nx = gt ? lv[u][k] : x;
              const int ng = gt ? gx : lg[u][k], ngx = gt ? lg[u][k] : gx;
              lv[u][k] = nl; lg[u][k] = ng; x = nx; gx = ngx;
            }
          }
          if (m4 > vm[u]) {  // later items hold higher columns: a tie keeps the earlier one
            vs[u] = vm[u] == -INFINITY ? 0.f : vs[u] * fexp(vm[u] - m4);
            vm[u] = m4; vi[u] = i4;
          }
          if (vm[u] != -INFINITY) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < N) vs[u] += fexp(v[e] - vm[u]);
          }
        }
      } else if (active && r < p.R) {
        const float4 bv = P.bv;
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        if constexpr (EPI == E_QKV) {
          if (nb < D) {
            cst_f4(p.q + (int64_t)r * D + nb, make_float4(v[0], v[1], v[2], v[3]));
          } else {
            bf16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = (bf16_t)v[e];
            cst_b4(skv + ((int64_t)r * p.T + (t - 1)) * (2 * D) + (nb - D), ob);
          }
        } else if constexpr (EPI == E_Q) {
          cst_f4(p.q + (int64_t)r * D + nb, make_float4(v[0], v[1], v[2], v[3]));
        } else if constexpr (EPI == E_RES) {
          cst_f4(p.y + (int64_t)r * D + nb, make_float4(v[0] + xr.x, v[1] + xr.y, v[2] + xr.z, v[3] + xr.w));
        } else {  // E_ACT
          bf16x4 ob;
#pragma unroll
          for (int e = 0; e < 4; ++e) ob[e] = (bf16_t)res_act(v[e], p.act);
          cst_b4(p.h + (int64_t)r * N + nb, ob);
        }
      }
    }
  };

  bf16x8 wa[NF], wb[NF];
  // who hands this phase's output on: the workgroups with an item (+ the helpers that advance the decode state)
  const bool participant = pm.has || (AMODE == A_EMBED && pm.helper);
  const unsigned nprod = pm.np_ + (AMODE == A_EMBED ? pm.nh_ : 0u);
  if (gs.dead) return nprod;
  // RES_W3 (-DRES_W3=1; OFF): THREE sets of W fragments when a workgroup has several items, the second requested before
  // the hand-off wait like the first - twice the weight bytes in flight per CU.  Built on the reading that an item waits
  // for its 64 KB of weights from L2 (*measured* 640 beam rows: 16.9 us for the 7 items of a vocabulary workgroup against
  // ~1.1 us of MFMA + statistics each) and *measured* SLOWER (round 5, same box, us per step of the whole pass, two sets /
  // three: beam 5 at 128 clips 169.4 / 176.2, 64 clips 141.5 / 144.3; greedy 128 clips 62.0 / 63.6, 256 clips 93.4 / 97.2;
  // d_model 1024 at 32 clips 87.6 / 91.8): the third set lives in AGPRs the compiler copies through, and the items were not
  // waiting for bytes in flight.  Kept as a switch for the next profile.
#ifndef RES_W3
#define RES_W3 0
#endif
  constexpr bool W3 = RES_W3 != 0;
  const bool multi = pm.has && pm.c0 + pm.nper < CI;
  bf16x8 wc[W3 ? NF : 1];
  if (pm.has) fetch(wa, pm.c0);
  if (W3 && multi) fetch(wb, pm.c0 + pm.nper);
  if (do_wait && participant) gs.wait();
  if (gs.dead) return nprod;
  gs.mark();
  if constexpr (AMODE == A_EMBED)
    if (pm.helper) advance_state4(p, r0 + wave * 4, t, lane);
  if (pm.has) {
    if constexpr (AMODE == A_BF16) {
#pragma unroll
      for (int u = 0; u < RTB; ++u) {
        if (RTB > 1 && u > 0 && r0 + u * 16 >= p.R) break;
        load_a_bf16<K>(p, r0 + u * 16, reinterpret_cast<const bf16_t*>(asrc), sA + u * 16 * lda, lda);
      }
    } else {  // the rows of every tile requested before the first one is normalised
      float4 av[RTB][D / 64];
      // who stores the normalised rows: column slice k by the workgroup of item k when the row group has 8 of them
      const unsigned wmask = (pm.nper >= 8 && CI >= 8) ? (pm.c0 < 8 ? 1u << pm.c0 : 0u) : (pm.c0 == 0 ? 0xffu : 0u);
      int mytok[RTB];
#pragma unroll
      for (int u = 0; u < RTB; ++u) {
        mytok[u] = 0;
        if constexpr (AMODE == A_EMBED || AMODE == A_EMBEDB)
          mytok[u] = fetch_token<AMODE>(p, r0 + u * 16, t, pm.c0 == 0 && (AMODE == A_EMBEDB || !pm.helped));
      }
#pragma unroll
      for (int u = 0; u < RTB; ++u)
        fetch_a_rows<AMODE, D>(p, r0 + u * 16, t, mytok[u], reinterpret_cast<const float*>(asrc), asrc2, av[u]);
      LnGB<D> gb;  // (after the rows' requests: the token -> word row chain is what the stage waits for)
      load_gb<D>(gb, g, be);
#pragma unroll
      for (int u = 0; u < RTB; ++u) {
        if (RTB > 1 && u > 0 && r0 + u * 16 >= p.R) break;
        finish_a_rows<D>(p, r0 + u * 16, av[u], gb, write_x ? wmask : 0u, sA + u * 16 * lda, lda, EPI == E_VOCABK ? wmask : 0u);
      }
    }
    __syncthreads();
    if constexpr (RTB == 1) load_af(0);
    gs.mark();
    Pre Pa, Pb;
    preload(Pa, pm.c0);
    if (!multi) {
      item(wa, pm.c0, Pa);  // one item: nothing to prefetch (most phases at most row counts)
    } else if constexpr (W3) {
      Pre Pc;
      const int n1 = pm.nper;
      preload(Pb, min(pm.c0 + n1, CI - 1));
      for (int c = pm.c0; c < CI; c += 3 * n1) {
        preload(Pc, min(c + 2 * n1, CI - 1));
        if constexpr (NF > 1) fetch(wc, c + 2 * n1);
        item(wa, c, Pa);
        if (c + n1 >= CI) break;
        preload(Pa, min(c + 3 * n1, CI - 1));
        fetch(wa, c + 3 * n1);
        item(wb, c + n1, Pb);
        if (c + 2 * n1 >= CI) break;
        preload(Pb, min(c + 4 * n1, CI - 1));
        fetch(wb, c + 4 * n1);
        if constexpr (NF > 1) item(wc, c + 2 * n1, Pc);
      }
    } else {
      for (int c = pm.c0; c < CI; c += 2 * pm.nper) {
        preload(Pb, min(c + pm.nper, CI - 1));
        fetch(wb, c + pm.nper);
        item(wa, c, Pa);
        if (c + pm.nper >= CI) break;
        preload(Pa, min(c + 2 * pm.nper, CI - 1));
        fetch(wa, c + 2 * pm.nper);
        item(wb, c + pm.nper, Pb);
      }
    }
    gs.mark();
    if constexpr (EPI == E_VOCAB || EPI == E_VOCABK) {
#pragma unroll
      for (int u = 0; u < RTB; ++u) {
        const int r = r0 + u * 16 + l16;
        if (RTB > 1 && r0 + u * 16 >= p.R) break;
        {  // the four column groups of a row (lane ^ 16, lane ^ 32): row / half swaps, no LDS round trips
          amax_merge(vm[u], vi[u], vs[u], x16_other(vm[u]), x16_other(vi[u]), x16_other(vs[u]));
          amax_merge(vm[u], vi[u], vs[u], x32_other(vm[u]), x32_other(vi[u]), x32_other(vs[u]));
        }
        if (u > 0) __syncthreads();  // wave 0 has read the previous tile's entries
        if (kg == 0) { s_pm[wave][l16] = vm[u]; s_pi[wave][l16] = vi[u]; s_ps[wave][l16] = vs[u]; }
        __syncthreads();
        if (wave == 0 && kg == 0 && r < p.R) {
#pragma unroll
          for (int w = 1; w < 4; ++w) amax_merge(vm[u], vi[u], vs[u], s_pm[w][l16], s_pi[w][l16], s_ps[w][l16]);
          cst_f(p.pmax + (int64_t)r * p.parts + pm.c0, vm[u]);
          cst_i(p.pidx + (int64_t)r * p.parts + pm.c0, vi[u]);
          cst_f(p.psum + (int64_t)r * p.parts + pm.c0, vs[u]);
        }
        if constexpr (EPI == E_VOCABK) {
          // the 16 lists of a row (4 column groups x 4 waves) -> its RES_BMK best groups of this part: the lists go
          // through LDS, wave w merges rows 4 w .. 4 w + 3 - one row per 16-lane DPP row, one source list per lane,
          // RES_BMK rounds of `largest head of the 16 lists` (row16_max_u64), the lane that held it moves its list up
          __shared__ float s_lv[16][16][RES_BMK];
          __shared__ int s_lg[16][16][RES_BMK];
#pragma unroll
          for (int k = 0; k < RES_BMK; ++k) { s_lv[l16][wave * 4 + kg][k] = lv[u][k]; s_lg[l16][wave * 4 + kg][k] = lg[u][k]; }
          __syncthreads();
          const int mrow = wave * 4 + (lane >> 4), msrc = lane & 15, mr = r0 + u * 16 + mrow;
          float hv[RES_BMK], ov[RES_BMK];
          int hg[RES_BMK], og[RES_BMK];
#pragma unroll
          for (int k = 0; k < RES_BMK; ++k) { hv[k] = s_lv[mrow][msrc][k]; hg[k] = s_lg[mrow][msrc][k]; }
#pragma unroll
          for (int k = 0; k < RES_BMK; ++k) {
            const unsigned long long key = key_of(hv[0], (unsigned)hg[0]), best = row16_max_u64(key);
            ov[k] = key_val(best); og[k] = (int)key_idx(best);
            if (key == best) {  // (an empty entry may be `popped` by several lanes at once: their lists are empty anyway)
#pragma unroll
              for (int j = 0; j + 1 < RES_BMK; ++j) { hv[j] = hv[j + 1]; hg[j] = hg[j + 1]; }
              hv[RES_BMK - 1] = -INFINITY; hg[RES_BMK - 1] = 0x7fffffff;
            }
          }
          if (msrc == 0 && mr < p.R) {
            const int64_t o = ((int64_t)mr * p.parts + pm.c0) * RES_BMK;
#pragma unroll
            for (int k = 0; k < RES_BMK; ++k) { cst_f(p.gval + o + k, ov[k]); cst_i(p.ggid + o + k, og[k]); }
          }
        }
      }
    }
  }
  gs.mark();
  gs.arrive(participant);
  return nprod;
}

// FFN dense2 + residual for ff = 2048: out[R, 512] = h[R, 2048] W2^T + b2 + x.  K is added in EIGHTHS of 256 (two
// accumulator chains each), as half0 = e0 + ((e1 + e2) + e3), half1 = e4 + ((e5 + e6) + e7), y = ((half0 + b) + x) + half1,
// in both forms - a row's bits do not depend on the form its batch takes:
//   HALF = false: item = 16 columns, wave w multiplies eighths 2 w and 2 w + 1, wave 0 adds the eight tiles from LDS;
//   HALF = true (<= 64 rows: 32 items per row tile leave most CUs idle and each fetches 64 KB of W2 - what the phase waits for):
//                item = (16 columns, K half), wave w multiplies eighth 4 half + w; the half-0 workgroup stores
//                (half0 + b) + x to y, the half-1 workgroup stores half1 to y2, and the consumers add y + y2 on load.
// KF = ff (2048: d_model 512; 3072 / 4096: d_model 768 / 1024, HALF form only - a wave's K range must fit 16 fragments),
// D = d_model = the width of the output rows.
template <bool HALF, int KF = 2048, int D = 512>
RES_PHASE_FN unsigned ffn2_phase(const RArgs& p, GridSync& gs, bf16_t* sA, const bf16_t* W, const float* bias) {
  constexpr int K = KF, KW = HALF ? KF / 8 : KF / 4, NF = KW / 32;  // KW: a wave's K range
  constexpr int lda = (HALF ? KF / 2 : KF) + 8;                     // the LDS tile holds the K range of the item
  constexpr int CH = HALF ? NF : 8;                                 // fragments per accumulator pair (see the sum order above)
  static_assert(NF <= 16 && NF % 2 == 0 && (HALF || NF == 16), "ffn2_phase: K range of a wave");
  __shared__ f32x4 s_e[2][7][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kg = lane >> 4;
  const int RT = (p.R + 15) >> 4, CI = HALF ? D / 8 : D / 16;
  const PhaseMap pm(RT, CI);
  const int r0 = pm.rt * 16, r = r0 + l16;
  auto k0_of = [&](int c) { return HALF ? ((c & 1) * 4 + wave) * KW : wave * KW; };
  auto n0_of = [&](int c) { return (HALF ? c >> 1 : c) * 16; };
  bf16x8 wa[NF], wb[NF];
  const unsigned nprod = pm.np_;
  if (gs.dead) return nprod;
  // (unconditional, clamped: see gemm_phase's fetch)
  auto fetchw = [&](bf16x8 (&wf)[NF], int c) {
    const int cc = min(c, CI - 1);
    load_w<NF>(wf, W + (int64_t)(n0_of(cc) + l16) * K + k0_of(cc) + kg * 8);
  };
  // what an item's epilogue reads from memory, requested BEFORE the next item's weight fragments (vmcnt retires in order)
  struct Pre { float4 xr, bv; };
  auto preload = [&](Pre& P, int c) {
    const int cc = min(c, CI - 1), nb = n0_of(cc) + kg * 4, kh = HALF ? cc & 1 : 0;
    P.xr = make_float4(0.f, 0.f, 0.f, 0.f);
    P.bv = *reinterpret_cast<const float4*>(bias + nb);
    if (wave == 0 && kh == 0 && r < p.R) P.xr = cld_f4(p.xres + (int64_t)r * D + nb);
  };
  constexpr bool W3 = RES_W3 != 0;  // three sets of W fragments, two requested before the hand-off wait: see gemm_phase
  const bool multi = pm.has && pm.c0 + pm.nper < CI;
  bf16x8 wc[W3 ? NF : 1];
  if (pm.has) fetchw(wa, pm.c0);
  if (W3 && multi) fetchw(wb, pm.c0 + pm.nper);
  if (pm.has) gs.wait();
  if (gs.dead) return nprod;
  gs.mark();
  if (pm.has) {
    int par = 0;
    bool tile_loaded = false;
    auto item = [&](const bf16x8 (&wf)[NF], int c, const Pre& P) {
      const int n0 = n0_of(c), nb = n0 + kg * 4, k0 = k0_of(c), kh = HALF ? c & 1 : 0;
      const float4 xr = P.xr;
      if (!tile_loaded || HALF) {  // the 16 rows of h (HALF: the K half of this item) -> LDS
        if (tile_loaded) __syncthreads();
        constexpr int KP = HALF ? KF / 2 : KF, per_row = KP / 8, NC = 16 * per_row / 256;
        const int kb = HALF ? kh * (KF / 2) : 0;
        bf16x8 v[NC];
        if constexpr (NC == 16 && KF == 2048 && !HALF) {
          // row i of the tile, this thread's 16 bytes: 4096 bytes apart (rows past the batch lie inside the scratch too)
          cld16_x16_rows4k(reinterpret_cast<const unsigned char*>(p.h + (int64_t)r0 * K + threadIdx.x * 8), v);
#pragma unroll
          for (int i = 0; i < NC; ++i)
            if (r0 + i >= p.R) v[i] = bf16x8{};
        } else if constexpr (NC == 8) {
          const void* a[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int q = threadIdx.x + 256 * i, rr = q / per_row, c8 = q - rr * per_row;
            a[i] = p.h + (int64_t)(r0 + rr < p.R ? r0 + rr : 0) * K + kb + c8 * 8;
          }
          cld16_x8(a, v);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int q = threadIdx.x + 256 * i, rr = q / per_row;
            if (r0 + rr >= p.R) v[i] = bf16x8{};
          }
        } else {
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            const int q = threadIdx.x + 256 * i, rr = q / per_row, c8 = q - rr * per_row;
            v[i] = cld_b8(p.h + (int64_t)(r0 + rr < p.R ? r0 + rr : 0) * K + kb + c8 * 8);
            if (r0 + rr >= p.R) v[i] = bf16x8{};
          }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int q = threadIdx.x + 256 * i, rr = q / per_row, c8 = q - rr * per_row;
          *reinterpret_cast<bf16x8*>(sA + rr * lda + c8 * 8) = v[i];
        }
        __syncthreads();
        tile_loaded = true;
      }
      const bf16_t* ar = sA + l16 * lda + (k0 - (HALF ? kh * (KF / 2) : 0)) + kg * 8;
      f32x4 e[NF / CH];
#pragma unroll
      for (int h8 = 0; h8 < NF / CH; ++h8) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = h8 * CH; i < h8 * CH + CH; i += 2) {
          acc0 = care_mfma_16x16x32_h16(wf[i], *reinterpret_cast<const bf16x8*>(ar + i * 32), acc0, 0, 0, 0);
          acc1 = care_mfma_16x16x32_h16(wf[i + 1], *reinterpret_cast<const bf16x8*>(ar + (i + 1) * 32), acc1, 0, 0, 0);
        }
        e[h8] = acc0 + acc1;
      }
      // eighth tiles of the other waves -> LDS (slot = eighth - 1 within the item's range; wave 0 keeps its first)
      if constexpr (HALF) {
        if (wave > 0) s_e[par][wave - 1][lane] = e[0];
      } else {
        if (wave == 0) s_e[par][0][lane] = e[1];
        else { s_e[par][2 * wave - 1][lane] = e[0]; s_e[par][2 * wave][lane] = e[1]; }
      }
      __syncthreads();
      if (wave == 0 && r < p.R) {
        const float4 bv = P.bv;
        if constexpr (HALF) {
          const f32x4 half = e[0] + ((s_e[par][0][lane] + s_e[par][1][lane]) + s_e[par][2][lane]);
          if (kh == 0) {
            cst_f4(p.y + (int64_t)r * D + nb, make_float4((half[0] + bv.x) + xr.x, (half[1] + bv.y) + xr.y,
                                                             (half[2] + bv.z) + xr.z, (half[3] + bv.w) + xr.w));
          } else {
            cst_f4(p.y2 + (int64_t)r * D + nb, make_float4(half[0], half[1], half[2], half[3]));
          }
        } else {
          const f32x4 half0 = e[0] + ((s_e[par][0][lane] + s_e[par][1][lane]) + s_e[par][2][lane]);
          const f32x4 half1 = s_e[par][3][lane] + ((s_e[par][4][lane] + s_e[par][5][lane]) + s_e[par][6][lane]);
          cst_f4(p.y + (int64_t)r * D + nb, make_float4(((half0[0] + bv.x) + xr.x) + half1[0], ((half0[1] + bv.y) + xr.y) + half1[1],
                                                           ((half0[2] + bv.z) + xr.z) + half1[2], ((half0[3] + bv.w) + xr.w) + half1[3]));
        }
      }
      par ^= 1;
    };
    // two sets of weight fragments: the next item's travel while the current one is multiplied (*measured* 640 rows, 5
    // items per workgroup, each waiting for its own 64 KB from L2: 18 us in this phase)
    Pre Pa, Pb;
    preload(Pa, pm.c0);
    if (!multi) {
      item(wa, pm.c0, Pa);
    } else if constexpr (W3) {
      Pre Pc;
      const int n1 = pm.nper;
      preload(Pb, pm.c0 + n1);
      for (int c = pm.c0; c < CI; c += 3 * n1) {
        preload(Pc, c + 2 * n1);
        if constexpr (NF > 1) fetchw(wc, c + 2 * n1);
        item(wa, c, Pa);
        if (c + n1 >= CI) break;
        preload(Pa, c + 3 * n1);
        fetchw(wa, c + 3 * n1);
        item(wb, c + n1, Pb);
        if (c + 2 * n1 >= CI) break;
        preload(Pb, c + 4 * n1);
        fetchw(wb, c + 4 * n1);
        if constexpr (NF > 1) item(wc, c + 2 * n1, Pc);
      }
    } else {
      for (int c = pm.c0; c < CI; c += 2 * pm.nper) {
        preload(Pb, c + pm.nper);
        fetchw(wb, c + pm.nper);
        item(wa, c, Pa);
        if (c + pm.nper >= CI) break;
        preload(Pa, c + 2 * pm.nper);
        fetchw(wa, c + 2 * pm.nper);
        item(wb, c + pm.nper, Pb);
      }
    }
  }
  gs.mark();
  gs.arrive(pm.has);
  return nprod;
}

// One (row, head) of attention on fragments in registers: lane = (key slot = lane >> 3, 8-dim chunk = lane & 7); scores /
// softmax in registers, the 8 key slots merged by three exchange steps.  Masking as the reference: masked keys get
// -1e9, the hybrid bias is added after the mask (models/components/Attention.py:104-111).
template <int NKB>
__device__ __forceinline__ void attn_one(const float (&q)[8], const bf16x8 (&kf)[NKB], const bf16x8 (&vf)[NKB], const float (&add)[NKB],
                                         const bool (&padded)[NKB], int nk, int nkb, int slot, float (&acc)[8]) {
  float s[NKB];
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
    if (kb < nkb) {
      float dd = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) dd = fmaf(q[i], (float)kf[kb][i], dd);
      dd += care_dpp_x1(dd);
      dd += care_dpp_x2(dd);
      dd += care_dpp_m8(dd);
      dd *= 0.125f;
      if (padded[kb]) dd = -1e9f;
      dd += add[kb];
      s[kb] = kb * 8 + slot < nk ? dd : -INFINITY;
      m = fmaxf(m, s[kb]);
    }
  m = fmaxf(m, care_dpp_x8(m));
  m = x16_max(m);
  m = x32_max(m);
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
    if (kb < nkb) { s[kb] = fexp(s[kb] - m); sum += s[kb]; }
  sum += care_dpp_x8(sum);
  sum = x16_sum(sum);
  sum = x32_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
    if (kb < nkb) {
      const float pw = s[kb] * inv;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(pw, (float)vf[kb][i], acc[i]);
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += care_dpp_x8(acc[i]);
    acc[i] = x16_sum(acc[i]);
    acc[i] = x32_sum(acc[i]);
  }
}

// One attention phase: ctx[r, h*64 ..] = softmax(q_h K_h^T / 8 (masked, + bias)) V_h for every (row, head); one
// wave per item, lane = (key slot, 8-dim chunk).  Masking as the reference: masked keys get -1e9, the hybrid bias
// is added after the mask (models/components/Attention.py:104-111).  The heads of a row go to the waves of ONE XCD
// (row % 8), so a row's static K/V is read into one L2.
// One (row, head) of an attention phase in flight: its query chunk, key / value fragments, mask and bias terms.
template <int NKB>
struct AttnItem {
  int r, hh;
  float q[8];
  float add[NKB];
  bool padded[NKB];
  bf16x8 kf[NKB], vf[NKB];
};

// ANC (beam search, SELF): key j of row r is position j of the hypothesis in beam slot r - cached, like its token, at the
// PHYSICAL row anc[r][j] (the ancestor table of csrc/beam.hip: re-ordering beams never moves K / V)
template <bool SELF, int NKB, bool ANC = false, int D = 512>  // SELF: the keys / values are the cache this launch writes (coherent loads), pad mask from `fed`
RES_PHASE_FN unsigned attn_phase(const RArgs& p, GridSync& gs, bool do_wait, const bf16_t* KV, int64_t kv_bs,
                                               int rows_per_kv, int nk, const int32_t* pad_tok, const float* bias,
                                               int bias_ld, const int32_t* anc = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 3, chunk = lane & 7;
  constexpr int d = D;
  const int nkb = (nk + 7) >> 3, H = p.H;
  const bool by_xcd = (gridDim.x & 7) == 0;
  const int x = by_xcd ? (int)(blockIdx.x & 7) : 0, xs = by_xcd ? 8 : 1;
  const int bpx = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;  // workgroups per XCD (or all of them)
  const int slot0 = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int myslot = slot0 + bpx * wave;
  const int nslots = bpx * 4;
  const int nrows_x = (p.R - x + xs - 1) / xs;  // rows x, x + xs, ...
  const int nitems = nrows_x * H;
  // a workgroup has an item iff its wave 0 has one (the smallest slot of the four)
  const bool participant = slot0 < nitems;
  unsigned nprod = 0;
  for (int xx = 0; xx < xs; ++xx) {
    const int it = ((p.R - xx + xs - 1) / xs) * H;
    nprod += (unsigned)(it < bpx ? it : bpx);
  }
  if (gs.dead) return nprod;
  if (do_wait && participant) gs.wait();
  if (gs.dead) return nprod;
  gs.mark();

  // every load of an item, unconditionally (li is clamped by the caller): nothing here waits for anything
  auto load = [&](AttnItem<NKB>& it, int li) {
    it.r = x + xs * (li / H);
    it.hh = li % H;
    const float* qp = p.q + (int64_t)it.r * d + it.hh * 64 + chunk * 8;
    const float4 qa = cld_f4(qp), qb = cld_f4(qp + 4);
    it.q[0] = qa.x; it.q[1] = qa.y; it.q[2] = qa.z; it.q[3] = qa.w; it.q[4] = qb.x; it.q[5] = qb.y; it.q[6] = qb.z; it.q[7] = qb.w;
    const bf16_t* kb0 = KV + (int64_t)(it.r / rows_per_kv) * kv_bs + it.hh * 64 + chunk * 8;
    int prow[NKB];
    if constexpr (ANC) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        if (kb < nkb) {
          const int j = kb * 8 + slot;
          prow[kb] = cld_i(anc + (int64_t)it.r * p.fed_stride + (j < nk ? j : 0));
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
      if (kb < nkb) {
        const int j = kb * 8 + slot, jc = j < nk ? j : 0;
        if constexpr (ANC) {
          const bf16_t* kr = KV + (int64_t)prow[kb] * kv_bs + it.hh * 64 + chunk * 8 + (int64_t)jc * 2 * d;
          it.padded[kb] = cld_i(pad_tok + (int64_t)prow[kb] * p.fed_stride + jc) == p.pad;
          it.add[kb] = 0.f;
          it.kf[kb] = cld_b8(kr);
          it.vf[kb] = cld_b8(kr + d);
          continue;
        }
        it.padded[kb] = pad_tok ? cld_i(pad_tok + (int64_t)it.r * p.fed_stride + jc) == p.pad : false;
        it.add[kb] = bias ? bias[it.hh * bias_ld + jc] : 0.f;
        if constexpr (SELF) { it.kf[kb] = cld_b8(kb0 + (int64_t)jc * 2 * d); it.vf[kb] = cld_b8(kb0 + (int64_t)jc * 2 * d + d); }
        else {
          it.kf[kb] = *reinterpret_cast<const bf16x8*>(kb0 + (int64_t)jc * 2 * d);
          it.vf[kb] = *reinterpret_cast<const bf16x8*>(kb0 + (int64_t)jc * 2 * d + d);
        }
      }
  };
  auto compute = [&](const AttnItem<NKB>& it) {
    float acc[8];
    attn_one<NKB>(it.q, it.kf, it.vf, it.add, it.padded, nk, nkb, slot, acc);
    if (slot == 0) {
      bf16x8 ob;
#pragma unroll
      for (int i = 0; i < 8; ++i) ob[i] = (bf16_t)acc[i];
      cst_b8(p.ctx + (int64_t)it.r * d + it.hh * 64 + chunk * 8, ob);
    }
  };

  if (myslot < nitems) {
    AttnItem<NKB> A;
    load(A, myslot);
    if (myslot + nslots >= nitems) {
      compute(A);  // one item per wave (up to 128 rows): nothing to overlap
    } else if constexpr (NKB <= 8) {
      // several items per wave: the next item's loads travel while the current one is computed (two register sets;
      // an item past the last re-reads the last - unconditional loads, see gemm_phase's fetch)
      AttnItem<NKB> Bq;
      for (int li = myslot; li < nitems; li += 2 * nslots) {
        load(Bq, min(li + nslots, nitems - 1));
        compute(A);
        if (li + nslots >= nitems) break;
        load(A, min(li + 2 * nslots, nitems - 1));
        compute(Bq);
      }
    } else {
      compute(A);
      for (int li = myslot + nslots; li < nitems; li += nslots) {
        load(A, li);
        compute(A);
      }
    }
  }
  gs.mark();
  gs.arrive(participant);
  return nprod;
}


// Beam search: the bm rows of a clip attend to the SAME static keys (its memory / concept rows, whatever its beams
// hold).  One wave per (clip, head) fetches the head's K / V fragments ONCE and runs its bm queries past them - a fifth
// of the fragment traffic of a wave per (row, head), and one item per wave at 128 clips instead of five in series.  The
// arithmetic per row is attn_one's: the same bits as attn_phase.  The heads of a clip go to the waves of ONE XCD.
template <int NKB, int D = 512>
RES_PHASE_FN unsigned attn_shared_phase(const RArgs& p, GridSync& gs, bool do_wait, const bf16_t* KV, int64_t kv_bs,
                                                      int bm, int nk, const float* bias, int bias_ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 3, chunk = lane & 7;
  constexpr int d = D;
  const int nkb = (nk + 7) >> 3, H = p.H, nclips = p.R / bm;
  const bool by_xcd = (gridDim.x & 7) == 0;
  const int x = by_xcd ? (int)(blockIdx.x & 7) : 0, xs = by_xcd ? 8 : 1;
  const int bpx = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int slot0 = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int myslot = slot0 + bpx * wave, nslots = bpx * 4;
  const int nitems = ((nclips - x + xs - 1) / xs) * H;
  const bool participant = slot0 < nitems;
  unsigned nprod = 0;
  for (int xx = 0; xx < xs; ++xx) {
    const int it = ((nclips - xx + xs - 1) / xs) * H;
    nprod += (unsigned)(it < bpx ? it : bpx);
  }
  if (gs.dead) return nprod;
  // the keys / values were written before the launch: the first item's fragments travel while the workgroup waits
  bf16x8 kf[NKB], vf[NKB];
  float add[NKB];
  bool padded[NKB];
  auto load_kv = [&](int li) {
    const int clip = x + xs * (li / H), hh = li % H;
    const bf16_t* kb0 = KV + (int64_t)clip * kv_bs + hh * 64 + chunk * 8;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
      if (kb < nkb) {
        const int j = kb * 8 + slot, jc = j < nk ? j : 0;
        padded[kb] = false;
        add[kb] = bias ? bias[hh * bias_ld + jc] : 0.f;
        kf[kb] = *reinterpret_cast<const bf16x8*>(kb0 + (int64_t)jc * 2 * d);
        vf[kb] = *reinterpret_cast<const bf16x8*>(kb0 + (int64_t)jc * 2 * d + d);
      }
  };
  if (myslot < nitems) load_kv(myslot);
  if (do_wait && participant) gs.wait();
  if (gs.dead) return nprod;
  gs.mark();
  for (int li = myslot; li < nitems; li += nslots) {
    const int clip = x + xs * (li / H), hh = li % H;
    if (li != myslot) load_kv(li);
    float q[RES_BMK][8];
#pragma unroll
    for (int i = 0; i < RES_BMK; ++i)
      if (i < bm) {
        const float* qp = p.q + (int64_t)(clip * bm + i) * d + hh * 64 + chunk * 8;
        const float4 qa = cld_f4(qp), qb = cld_f4(qp + 4);
        q[i][0] = qa.x; q[i][1] = qa.y; q[i][2] = qa.z; q[i][3] = qa.w; q[i][4] = qb.x; q[i][5] = qb.y; q[i][6] = qb.z; q[i][7] = qb.w;
      }
#pragma unroll
    for (int i = 0; i < RES_BMK; ++i)
      if (i < bm) {
        // (the fragments are opaque to the compiler per row: it would otherwise hoist their 2 x 8 NKB bf16 -> fp32
        // conversions out of the row loop - 256 registers at 128 keys, all of them spilled)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(kf[kb]), "+v"(vf[kb]));
        float acc[8];
        attn_one<NKB>(q[i], kf, vf, add, padded, nk, nkb, slot, acc);
        if (slot == 0) {
          bf16x8 ob;
#pragma unroll
          for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)acc[e];
          cst_b8(p.ctx + (int64_t)(clip * bm + i) * d + hh * 64 + chunk * 8, ob);
        }
      }
  }
  gs.mark();
  gs.arrive(participant);
  return nprod;
}

// ------------------------------------------------------------------------------------------------------------------
// Host side, shared by care_decode_resident and care_decode_resident_beam.

// The launch's sync area is zeroed by a KERNEL of our own in front of it, not by hipMemsetAsync: captured into a hipGraph, the
// memset node of ROCm 7.2 replays - from the second replay of some graphs on (the d_model 768 / 1024 greedy passes whose input
// tensors had been freed and allocated again between replays: *measured* round 6) - as a fill with a 16-BYTE PATTERN OF TWO
// POINTERS (the destination's own address and another): the hand-off counters start at garbage, the count of ended rows at 29355
// >= rows, and the launch "ends" after its first step with whatever tokens the unsynchronised phases produced.  A kernel node
// carries its arguments by value.
__global__ void res_zero_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}
inline hipError_t res_zero_words(unsigned* p, int bytes, hipStream_t st) {
  hipLaunchKernelGGL(res_zero_kernel, dim3((bytes / 4 + 255) / 256), dim3(256), 0, st, p, bytes / 4);
  return hipGetLastError();
}

// care_resident_layer[] -> RArgs::L; 0 or a CARE_E* code
inline int res_fill_layers(RArgs& p, const care_resident_layer* layers, int n_layers) {
  for (int l = 0; l < n_layers; ++l) {
    const care_resident_layer& s = layers[l];
    RLayer& L = p.L[l];
    if (!s.qkv_w || !s.qkv_b || !s.o_w || !s.o_b || !s.ln_g || !s.ln_b || !s.self_kv || !s.w1 || !s.b1 || !s.w2 || !s.b2 ||
        !s.ffn_g || !s.ffn_b || s.n_att < 0 || s.n_att > 2)
      return CARE_EINVAL;
    L.qkv_w = (const bf16_t*)s.qkv_w; L.qkv_b = s.qkv_b; L.o_w = (const bf16_t*)s.o_w; L.o_b = s.o_b; L.g = s.ln_g; L.be = s.ln_b;
    L.skv = (bf16_t*)s.self_kv;
    L.n_att = s.n_att;
    for (int a = 0; a < s.n_att; ++a) {
      const care_resident_attn& sa = s.att[a];
      if (!sa.q_w || !sa.q_b || !sa.o_w || !sa.o_b || !sa.ln_g || !sa.ln_b || !sa.kv || sa.rows_per_kv < 1) return CARE_EINVAL;
      if (sa.nkeys < 1 || sa.nkeys > 8 * RES_MAXKB) return CARE_ESHAPE;
      RAttn& A = L.att[a];
      A.q_w = (const bf16_t*)sa.q_w; A.q_b = sa.q_b; A.o_w = (const bf16_t*)sa.o_w; A.o_b = sa.o_b; A.g = sa.ln_g; A.be = sa.ln_b;
      A.kv = (const bf16_t*)sa.kv; A.kv_bs = sa.kv_batch_stride; A.nkeys = sa.nkeys; A.rows_per_kv = sa.rows_per_kv;
      A.bias = sa.bias; A.bias_ld = sa.bias_ld;
    }
    L.w1 = (const bf16_t*)s.w1; L.b1 = s.b1; L.w2 = (const bf16_t*)s.w2; L.b2 = s.b2; L.fg = s.ffn_g; L.fbe = s.ffn_b;
  }
  p.n_layers = n_layers;
  return 0;
}

}  // namespace
extern std::atomic<int> care_res_dbg_prof, care_res_dbg_ghost;  // decode_resident.hip (care_decode_resident_debug)
extern std::atomic<int> care_res_fenced_mode;                   // decode_resident.hip (care_resident_set_fenced): -1 auto, 0, 1
namespace {

// Which hand-off a resident launch on the current device takes: the fence-free one only where it was validated
// (tests/test_gpu_resident.py's stress / contention tests run on gfx950 with all 256 CUs in one partition); any other
// device, partition mode or an explicit care_resident_set_fenced(1) / CARE_RESIDENT_FENCED=1 gets the release / acquire
// pair.  care_resident_set_fenced(0) forces the fence-free form (the validated arch is still a compile-time condition).
inline int res_fenced_for_device() {
  int mode = care_res_fenced_mode.load();
  if (mode < 0) {
    static const int env = [] { const char* e = getenv("CARE_RESIDENT_FENCED"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    mode = env;
  }
  if (mode >= 0) return mode;
  // decided once per device ordinal (hipGetDeviceProperties is a driver round trip in front of a latency-bound launch)
  static std::atomic<int> per_device[64];
  static const bool init = [] { for (auto& a : per_device) a.store(-1); return true; }();
  (void)init;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  if (dev >= 0 && dev < 64) {
    const int known = per_device[dev].load(std::memory_order_relaxed);
    if (known >= 0) return known;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
  const bool gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
  const int fenced = (gfx950 && prop.multiProcessorCount == 256) ? 0 : 1;
  if (dev >= 0 && dev < 64) per_device[dev].store(fenced, std::memory_order_relaxed);
  return fenced;
}

// Tuning / tool / test knobs of the resident launches: the environment is read ONCE per process (first launch), the
// debug hooks are set through care_decode_resident_debug (tests, tools/resident_prof.py), never through the environment.
struct ResKnobs {
  int rb, small, half_rows, beam_cfg;  // -1: not set
  ResKnobs() {
    auto geti = [](const char* n) { const char* e = getenv(n); return e ? atoi(e) : -1; };
    rb = geti("CARE_RESIDENT_RB"); small = geti("CARE_RESIDENT_SMALL"); half_rows = geti("CARE_RESIDENT_HALF_ROWS");
    beam_cfg = geti("CARE_RESIDENT_BEAM_CFG");
  }
};
inline const ResKnobs& res_knobs() {
  static const ResKnobs k;
  return k;
}

// Every workgroup of a resident launch must be resident at the same time: the grid is at most one workgroup per CU, and
// the occupancy query must admit one workgroup of this kernel per CU (registers, LDS).  0 or a CARE_E* / hipError_t code.
inline int res_check_residency(const void* kernel, int lds, int grid, int cus) {
  int nb = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, (size_t)lds);
  if (e != hipSuccess) return (int)e;
  return (nb >= 1 && grid <= cus) ? 0 : CARE_ESHAPE;
}

}  // namespace
