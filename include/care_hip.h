/*
 * care_hip.h - C ABI of libcare_hip.so: hand-written gfx950 (MI355X) kernels for the
 * captioning forward path of yangbang18/CARE.
 *
 * The reference has NO native/FFI boundary on this path (it is pure PyTorch, SURVEY.md
 * 8(b)); its seam is the two Python factories `get_framework` / `get_translator`
 * (models/Framework.py:14-51, models/Translator.py:14-19).  The host mirror of that seam
 * lives in care_amd/framework.py and care_amd/translator.py and calls the entry points
 * below through ctypes.  Each entry point names the reference code it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers; sizes/strides are in ELEMENTS unless noted;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - returns 0 on success, a negative CARE_E* code for a rejected argument, or a
 *     positive hipError_t from the launch;
 *   - never allocates, frees or synchronises; no host-visible global state; safe to
 *     capture into a hipGraph;
 *   - activations are fp32; `wdtype` selects the storage type of weights and of the
 *     K/V caches: CARE_F32 (exact f32 MFMA, parity mode) or CARE_BF16 (bf16 MFMA with
 *     fp32 accumulation, throughput mode).  LayerNorm / softmax statistics, the concept
 *     head and all accumulators are always fp32.
 */
#ifndef CARE_HIP_H
#define CARE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARE_ABI_VERSION 22

enum { CARE_F32 = 0, CARE_BF16 = 1 };
enum { CARE_ACT_NONE = 0, CARE_ACT_RELU = 1, CARE_ACT_GELU = 2 };
enum {
  CARE_EINVAL = -1,      /* null pointer / non-positive size */
  CARE_EALIGN = -2,      /* pointer or leading dimension not 16-byte aligned */
  CARE_ESHAPE = -3,      /* shape outside what the kernels support (stated per function) */
  CARE_EDTYPE = -4       /* unknown dtype / activation code */
};

/* ABI version of the loaded library (== CARE_ABI_VERSION). */
int care_version(void);

/* Name of the code-object architecture the library was built for ("gfx950"). */
const char* care_arch(void);

/*
 * What the library was built from (care_amd/build.py; csrc/version.hip):
 *   care_source_hash: 32 hex digits of the SHA-256 over csrc/ (.hip and .h), this header and the compile flags - the host
 *     side (care_amd/_lib.py) compares it with the hash of the tree it runs from and rebuilds or refuses a stale library;
 *   care_build_flags: the extra compile flags of the build ("" for the default library, "-DCARE_H16_FP16" for the fp16 one);
 *   care_h16: the 16-bit storage / MFMA operand type the kernels were compiled for, "bf16" (libcare_hip.so) or "fp16"
 *     (libcare_hip_f16.so).  Wherever this header says CARE_BF16 / "bf16" for a storage type it means THAT type: the two
 *     libraries are one source, and the fp16 one is the compute mode `fp16` (11 significand bits at bf16's bytes and MFMA
 *     rate).  Nothing else differs between them.
 */
const char* care_source_hash(void);
const char* care_build_flags(void);
const char* care_h16(void);

/*
 * care_gemm:  C = act(A * W^T + bias), optionally split over two destinations.
 *   Replaces every nn.Linear on the path: Embedder Linear (models/Encoder.py:167),
 *   SDPA query/key/value (models/components/Attention.py:53-56,63-67), MHA dense
 *   (SubLayers.py:35,69), FFN dense1/dense2 (SubLayers.py:129-130,143-145), concept
 *   prj / semantic2hidden (models/Predictor/pred_attribute.py:62-65,260) and the vocab
 *   projection when all logits are needed (models/Head.py:26-32, Framework.py:258-259).
 *   A [M,K] fp32 row-major, leading dim lda.  W [N,K] row-major (nn.Linear layout,
 *   leading dim K) of type wdtype.  bias [N] fp32 or NULL.
 *   Columns [0,n_split) go to C0 (leading dim ldc0, type c0_dtype), columns
 *   [n_split,N) go to C1 at column (col - n_split) (ldc1, c1_dtype); pass n_split = N
 *   and C1 = NULL for a single destination.  n_split must be a multiple of 16.
 *   Requires K % 32 == 0 (f32) or K % 64 == 0 (bf16), lda % 4 == 0, 16-byte aligned A, W.
 */
int care_gemm(const float* A, int64_t lda, const void* W, int wdtype, const float* bias,
              void* C0, int64_t ldc0, int c0_dtype, void* C1, int64_t ldc1, int c1_dtype,
              int n_split, int M, int N, int K, int act, void* stream);

/*
 * care_gemm_argmax: per-row (max, argmax, sum exp) of A * W^T without writing the logits.
 *   Replaces NaiveHead + log_softmax + the top-1 of Beam.advance for greedy decoding
 *   (models/Head.py:26-32, models/Translator.py:127, misc/Decoding/Beam.py:58-70).
 *   Writes partial results for `care_argmax_parts(N)` column groups per row:
 *   pmax/psum fp32 [M, parts], pidx int32 [M, parts]; psum = sum exp(x - pmax) over the
 *   group's columns.  Ties resolve to the lowest column index.
 */
int care_argmax_parts(int N);
int care_gemm_argmax(const float* A, int64_t lda, const void* W, int wdtype,
                     float* pmax, int32_t* pidx, float* psum, int M, int N, int K, void* stream);

/*
 * care_split3_weight / care_gemm_split3: C = A W^T + bias for fp32 A and an fp32 weight with fp32-GRADE products at a
 *   third of the 16-bit MFMA rate (the exact-f32 MFMA of care_gemm runs at 1/16): operands split into fp16 pieces,
 *   A_hi W_hi + A_hi W_lo + A_lo W_hi as one product over 3K (the dropped term is ~2^-22 of a product).  The
 *   feature embedder (models/Encoder.py:167) of concept models whose d_model the fused care_gemm_ln_split does not
 *   cover.  W3: [N, 3K] 16-bit (W_hi | W_lo | W_hi), made once by care_split3_weight; C fp32; K % 64 == 0; |A| < 65504.
 */
int care_split3_weight(const float* W, void* W3, int N, int K, void* stream);
int care_gemm_split3(const float* A, int64_t lda, const void* W3, const float* bias, float* C,
                     int64_t ldc, int M, int N, int K, void* stream);

/*
 * care_gemm_bf16 / care_gemm_argmax_bf16: the same contracts as care_gemm /
 *   care_gemm_argmax for bf16 weights, implemented by the A-stationary kernel
 *   (csrc/gemm_as.hip: A panel resident in registers, W streamed through LDS by LDS-DMA).
 *   A may be fp32 (rounded to bf16 on load) or bf16 (a_dtype); lda in elements of A.
 *   Requires K % 128 == 0 and lda % 8 == 0; care_gemm_argmax_bf16 additionally K <= 512.
 *   The number of partial column groups of the argmax variant depends on M as well:
 *   care_argmax_parts_bf16(M, N).
 */
int care_gemm_bf16(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias,
                   void* C0, int64_t ldc0, int c0_dtype, void* C1, int64_t ldc1, int c1_dtype,
                   int n_split, int M, int N, int K, int act, void* stream);
int care_argmax_parts_bf16(int M, int N);
/*
 * care_gemm_bf16_splitk: K > 512 at small M (the decode-step FFN dense2, SubLayers.py:143-145):
 *   K/512 slices; slice s writes A[:, 512s:512s+512] * W[:, 512s:512s+512]^T (+ bias when
 *   s == 0) to the fp32 slab C + s * slab_stride.  The slabs are summed by the consumer,
 *   care_add_ln(nslab = K / 512, slab_stride), so no reduction pass exists.  K % 512 == 0.
 */
int care_gemm_bf16_splitk(const void* A, int64_t lda, int a_dtype, const void* W,
                          const float* bias, float* C, int64_t ldc, int64_t slab_stride,
                          int M, int N, int K, void* stream);
int care_gemm_argmax_bf16(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                          int32_t* pidx, float* psum, const int32_t* labels, float* plab,
                          int M, int N, int K, void* stream);

/*
 * care_gemm_tile / care_gemm_tile_argmax: the contracts of care_gemm_bf16 / care_gemm_argmax_bf16 for bf16 A
 *   and ANY K % 64 == 0 (csrc/gemm_tile.hip: both operands streamed through an LDS ring by LDS-DMA in K steps of
 *   64, 64 x 64 output tiles per wave) - the nn.Linear layers of the d_model = 768 / 1024 architectures
 *   (config/archs.yaml:15-26: K = 768, 1024, 3072, 4096; the same reference lines as care_gemm), and the
 *   vocabulary projection of greedy decoding / teacher-forced scoring there (models/Head.py:26-32,
 *   Translator.py:127, misc/Crit/crit_lang.py:75-103).  A bf16 [M, lda], W bf16 [N, K], lda % 8 == 0.
 *   care_gemm_tile_argmax writes care_argmax_parts_tile(N) partials per row (one per 64 columns); labels / plab
 *   (optional, both or neither) as in care_gemm_argmax_bf16.
 */
/* care_split2_act / care_gemm_tile_split3: the contract of care_gemm_split3 (fp32 A, fp32-GRADE products
 *   a_hi w_hi + a_hi w_lo + a_lo w_hi in fp16 pieces: the feature embedder of concept models, models/Encoder.py:167)
 *   on the LDS-tiled kernel: A2 = care_split2_act(A) fp16 [M, 2K] (hi | lo), W3 = care_split3_weight(W) [N, 3K];
 *   destinations, split and activation as care_gemm.  K % 64 == 0, |A| < 65504. */
int care_split2_act(const float* A, int64_t lda, void* A2, int M, int K, void* stream);
int care_gemm_tile_split3(const void* A2, const void* W3, const float* bias, void* C0, int64_t ldc0, int c0_dtype,
                          void* C1, int64_t ldc1, int c1_dtype, int n_split, int M, int N, int K, int act,
                          void* stream);
/* Split products of PRE-SCALED operands - the GEMMs of training mode (models/Wrapper.py:423-435 -> Framework.py:215-237 under
 *   autograd: every nn.Linear's forward, dx = dy W and dW = dy^T x; care_amd/training.py).  Gradients are 1e-6 .. 1e-3: the low
 *   piece of an unscaled hi / lo split would be an fp16 denormal.  care_absmax: *slot (4 bytes, device) = bit pattern of max |A|
 *   (the function zeroes it first; NaNs skipped).  care_split_pieces: the fp16 pieces of src * 2^e, e chosen from *amax so that
 *   the largest magnitude lies in [2^14, 2^15) (exact in fp32; zero / non-finite maxima: e = 0), read from the operand as it
 *   lies - [rows, K], or transposed [K, rows] - and written slab-major: [slabs][rows][pieces ks], pieces = 2 (hi | lo: the A
 *   operand, care_split2_act's layout per slab) or 3 (hi | lo | hi: the W operand, care_split3_weight's); slab s = columns
 *   s ks .. of the operand, zeros past K; ks % 64 == 0.  care_gemm_tile_split3_scaled: C [M, ldc] fp32 =
 *   (A2 W3^T) / (2^ea 2^eb) + bias, the exponents re-derived from the same two slots.  No host synchronisation anywhere:
 *   the scales never leave the device.  K % 64 == 0, lda % 4 == 0.  slabs > 1: split-K for products with few output tiles
 *   (dW = dy^T x): A2 / W3 hold `slabs` matrices of K columns each (consecutive K ranges of the product), slab s goes to
 *   C + s M ldc, the caller adds them in order (care_strided_sum); bias == NULL then. */
int care_absmax(const float* A, int64_t lda, int M, int K, void* slot, void* stream);
int care_split_pieces(const float* src, int64_t ld, int rows, int K, int transposed, int slabs, int ks, void* out,
                      int pieces, const void* amax, void* stream);
int care_gemm_tile_split3_scaled(const void* A2, const void* W3, const float* bias, float* C, int64_t ldc, int M, int N,
                                 int K, const void* amax_a, const void* amax_b, int slabs, void* stream);
/* ... and the fused vocabulary arg-max (care_gemm_tile_argmax's partials) on the same split products: the `fp16x3`
 *   compute mode (fp32 storage, every GEMM as three fp16 MFMA passes) of care_amd/engine.py. */
int care_gemm_tile_split3_argmax(const void* A2, const void* W3, float* pmax, int32_t* pidx, float* psum, int M,
                                 int N, int K, void* stream);
int care_gemm_tile(const void* A, int64_t lda, const void* W, const float* bias, void* C0, int64_t ldc0,
                   int c0_dtype, void* C1, int64_t ldc1, int c1_dtype, int n_split, int M, int N, int K,
                   int act, void* stream);
int care_argmax_parts_tile(int N);
/* care_gemm_tile_batched: `batch` independent products C_b = A_b W_b^T + bias_b in one launch (element offsets a_bs,
 *   w_bs, c_bs, bias_bs per batch; W rows of leading dimension ldw): the per-head projections on either side of the
 *   absorbed cross-attention for d_model = 1024 - the roles of care_head_expand / care_head_reduce
 *   (models/components/Attention.py:63-67 moved to the query / context side). */
int care_gemm_tile_batched(const void* A, int64_t lda, int64_t a_bs, const void* W, int64_t ldw, int64_t w_bs,
                           const float* bias, int bias_bs, void* C, int64_t ldc, int64_t c_bs, int c_dtype, int batch,
                           int M, int N, int K, void* stream);
int care_gemm_tile_argmax(const void* A, int64_t lda, const void* W, float* pmax, int32_t* pidx, float* psum,
                          const int32_t* labels, float* plab, int M, int N, int K, void* stream);

/*
 * Teacher-forced scoring (the metrics step): log-probability of the label token and the arg-max
 *   token of every row.  Replaces log_softmax + gather / max of LanguageGeneration
 *   (misc/Crit/crit_lang.py:75-103: word accuracy, perplexity) for the eval metrics step
 *   (models/Wrapper.py:182-184).
 *   care_gemm_argmax_bf16 with labels/plab non-NULL also records, per column group, the logit
 *   of column labels[row] (-inf where the group does not contain it);
 *   care_score_partials reduces the groups: logp[r] = x[label] - max - log(sum exp(x - max)),
 *   pred[r] = arg-max column.  The [rows, V] logits are never written.
 *   care_score_logits does the same from materialised logits [rows, ld] (fp32 mode / checker).
 */
int care_score_partials(const float* pmax, const int32_t* pidx, const float* psum,
                        const float* plab, int parts, float* logp, int32_t* pred, int rows,
                        void* stream);
/* The label logit on its own (so that the arg-max statistics can come from the fastest kernel, which keeps none):
 *   care_label_logits: out[r] = A[r, :] . W[col[r], :], bf16 A [rows, lda] and W [N, K], fp32 accumulation;
 *   care_score_partials_lab: care_score_partials with that one logit per row instead of per-group label partials. */
int care_label_logits(const void* A, int64_t lda, const void* W, const int32_t* col, float* out, int rows, int N,
                      int K, void* stream);
int care_score_partials_lab(const float* pmax, const int32_t* pidx, const float* psum, int parts,
                            const float* lab_logit, float* logp, int32_t* pred, int rows, void* stream);
int care_score_logits(const float* logits, int64_t ld, int V, const int32_t* labels, float* logp,
                      int32_t* pred, int rows, void* stream);

/*
 * care_greedy_update: finish the argmax over the partials and advance the greedy state.
 *   Replaces Beam.advance / Beam.done for beam_size == 1 (misc/Decoding/Beam.py:45-85)
 *   and the bookkeeping of Translator_ARFormer.beam_decode_step
 *   (models/Translator.py:91-109,135-143) without the per-step host sync.
 *   For every row r: tok = argmax, logp = -log(sum exp(x - max)).  If the row is not
 *   finished: fed[r, t] = tok, score[r] += logp, length[r] = t, and the row finishes
 *   when tok == eos_id or t == max_steps.  `fed` is int32 [rows, fed_stride]; column 0
 *   holds BOS.  Finished rows keep running (rows are independent) but are frozen.
 */
int care_greedy_update(const float* pmax, const int32_t* pidx, const float* psum, int parts,
                       int32_t* fed, int fed_stride, float* score, int32_t* length,
                       int32_t* finished, int t, int max_steps, int eos_id, int rows,
                       void* stream);

/*
 * care_greedy_update_embed: care_greedy_update for step t, and in the same launch the embedding of
 *   the chosen token at position t - word[token] + pos[t] (+ sem[r / sem_div]) -> LayerNorm ->
 *   out / out_bf16 - i.e. the care_embed_ln call of decode step t + 1 (Embeddings.forward,
 *   models/components/Embeddings.py) fused behind Beam.advance's token choice.  One launch per
 *   decoder step less.  word fp32 [V, d], pos fp32 [>= t + 1, d], sem (optional) fp32.
 */
int care_greedy_update_embed(const float* pmax, const int32_t* pidx, const float* psum, int parts,
                             int32_t* fed, int fed_stride, float* score, int32_t* length,
                             int32_t* finished, int t, int max_steps, int eos_id, int rows,
                             const float* word, const float* pos, const float* sem, int sem_div,
                             const float* gamma, const float* beta, float eps, float* out,
                             void* out_bf16, int64_t ldo, int d, void* stream);

/*
 * care_add_ln: out = LayerNorm(x + res) * gamma + beta, row-wise over d columns.
 *   Replaces nn.LayerNorm after the Embedder Linear (models/Encoder.py:167) and the
 *   post-LN residual epilogues of MultiHeadAttention / PositionwiseFeedForward
 *   (SubLayers.py:73-79,147-150).  res may be NULL.  If pos is non-NULL, pos[(r % grp), :]
 *   is added as well (TransformerEncoderBase position embedding, Encoder.py:265-279).
 *   Output row of input row r: (r / grp) * out_grp_rows + out_row_off + (r % grp), so an
 *   encoder stream lands directly inside the [B, Lk, d] cross-attention memory
 *   (the torch.cat of Encoder.py:148-149 and Framework.py:184-185 becomes an offset).
 *   out_bf16 (optional, same layout and ldo as out): a bf16 mirror of the result, the A
 *   operand of the next bf16 GEMM (the same rounding that GEMM would apply on load).
 *   nslab > 1: x is the sum of nslab fp32 slabs spaced slab_stride elements apart (the
 *   split-K partial products of care_gemm_bf16_splitk); nslab = 1 otherwise.
 *   gamma == beta == NULL: NO LayerNorm - out = x + res (+ pos): the sub-blocks of a pre-LN decoder
 *   (`transformer_pre_ln`, opts.py:68; SubLayers.py:55,78,140,149: LayerNorm BEFORE the sub-block, the residual sum left
 *   as it is); care_embed_ln likewise (Embeddings.py:130: no LayerNorm after the embedding sum).
 *   d % 4 == 0, d <= 2048.
 */
int care_add_ln(const float* x, int64_t ldx, const float* res, int64_t ldres,
                const float* pos, const float* gamma, const float* beta, float eps,
                float* out, void* out_bf16, int64_t ldo, int rows, int d, int grp,
                int out_grp_rows, int out_row_off, int nslab, int64_t slab_stride, void* stream);

/*
 * care_gemm_ln: out = LayerNorm(A W^T + bias + res + pos) * gamma + beta with N = d_model = 512,
 *   one workgroup owning full output rows (csrc/gemm_ln.hip).  Fuses the Linear -> LayerNorm of
 *   the Embedder (models/Encoder.py:167) and the dense -> dropout -> +residual -> LayerNorm
 *   epilogues of MultiHeadAttention / PositionwiseFeedForward (SubLayers.py:69-79,143-150).
 *   A [M, K] fp32 (raw features; rounded to bf16 when multiplied) or bf16; W bf16 [512, K];
 *   bias/gamma/beta fp32 [512]; res (optional) fp32 [M, ldres]; pos (optional) fp32 [grp, 512]
 *   added per row r as pos[r % grp].  Output rows are remapped exactly like care_add_ln
 *   (grp / out_grp_rows / out_row_off); out fp32 and the bf16 mirror share ldo; either may be
 *   NULL (not both): a caller that only feeds bf16 GEMMs / the absorbed cross-attention needs no
 *   fp32 copy.  Requires N == 512, K % 32 == 0.
 */
int care_gemm_ln(const void* A, int64_t lda, int a_dtype, const void* W, const float* bias,
                 const float* res, int64_t ldres, const float* pos, const float* gamma,
                 const float* beta, float eps, float* out, void* out_bf16, int64_t ldo, int M,
                 int N, int K, int grp, int out_grp_rows, int out_row_off, void* stream);

/*
 * care_pack_ln_weight / care_gemm_ln_packed: the same fused Linear -> (+residual) -> LayerNorm with
 *   the [512, K] bf16 weight re-laid-out ONCE (at weight-load time) into the order the kernel streams
 *   it: for every K step of 32 the 32-KB LDS image of that step (rows in order, 16-byte chunks at
 *   their bank-swizzled positions), so every DMA instruction reads 1 KB of full cache lines instead
 *   of sixteen 64-byte row pieces.  Same arithmetic, bit for bit, as care_gemm_ln on the plain
 *   weight.  K % 64 (fp32 A) / K % 128 (bf16 A) == 0; no position table.  W_packed: K * 1024 bytes.
 */
int care_pack_ln_weight(const void* W, void* W_packed, int N, int K, void* stream);
int care_gemm_ln_packed(const void* A, int64_t lda, int a_dtype, const void* W_packed,
                        const float* bias, const float* res, int64_t ldres, const float* gamma,
                        const float* beta, float eps, float* out, void* out_bf16, int64_t ldo,
                        int M, int N, int K, int grp, int out_grp_rows, int out_row_off,
                        void* stream);

/*
 * care_pack_ln_weight_split / care_gemm_ln_split: the Embedder's Linear -> LayerNorm
 *   (models/Encoder.py:167) for models whose memory feeds the DISCRETE concept choice
 *   (pred_attribute.py:88-125,262-264): fp32 A and an fp32 [512, K] weight, multiplied as
 *   a_hi w_hi + a_hi w_lo + a_lo w_hi with x_hi = fp16(x), x_lo = fp16(x - x_hi) - three fp16 MFMA
 *   passes accumulated in fp32 (11 bits per piece: the dropped a_lo w_lo term is ~2^-22 of a product,
 *   fp32-grade, against bf16's 2^-9 per operand).  |x| < 65504.  W_split: 3 * K * 1024 bytes (per K
 *   step of 32 the LDS images of w_hi, w_lo, w_hi).  K % 64 == 0; no residual, no position table.
 */
int care_pack_ln_weight_split(const float* W, void* W_split, int N, int K, void* stream);
int care_gemm_ln_split(const void* A, int64_t lda, const void* W_split, const float* bias,
                       const float* gamma, const float* beta, float eps, float* out,
                       void* out_bf16, int64_t ldo, int M, int N, int K, int grp,
                       int out_grp_rows, int out_row_off, void* stream);

/*
 * care_group_mean: out[g, col_off + c] = mean over the grp rows of group g of x[., c].
 *   Replaces `item.mean(1)` per modality (models/Encoder.py:106); with ldo = n_mod * d and
 *   col_off = m * d it also performs the channel concat of pred_attribute.py:88-89.
 *   Input row of (g, i): g * in_grp_rows + in_row_off + i.
 */
int care_group_mean(const float* x, int64_t ldx, int in_grp_rows, int in_row_off, int grp,
                    float* out, int64_t ldo, int col_off, int groups, int d, void* stream);

/*
 * care_concept_finish: concept probabilities from concept scores.
 *   Replaces prepare_merged_probs for seq_len == 1, restated literally
 *   (models/Predictor/pred_attribute.py:17-46): p = sigmoid(s);
 *   preds = 1 - exp(log(clamp(1 - p, 1e-12, 1))); avg = mean_k p.
 *   scores [B, lds] fp32 (k valid columns) -> preds [B, ldp] (columns >= k zeroed up to
 *   ldp so the buffer can feed care_gemm with K = ldp), avg [B].
 */
int care_concept_finish(const float* scores, int64_t lds, float* preds, int64_t ldp,
                        float* avg, int B, int k, void* stream);

/*
 * care_concept_topk_embed: top-`topk` concepts -> embedded concept rows in the memory.
 *   Replaces SemanticContainer.forward's topk + NaiveEmbeddings
 *   (models/Predictor/pred_attribute.py:262-270, models/components/Embeddings.py:53-87):
 *   labels[b, j] = index of the j-th largest preds[b, :] (order: value desc, index asc;
 *   torch leaves the order of exact ties unspecified);
 *   out[b * out_grp_rows + out_row_off + j, :] = LN(word[labels[b,j]] + pos[j]).
 *   word == NULL: the labels alone - a model without local guidance (`use_attr_flags` G1L0:
 *   pred_attribute.py:252 builds no `attr_embs`, :276-277 leaves `semantic_embs` None).
 *   k <= 1024, topk <= 64, d % 4 == 0, d <= 2048.
 */
int care_concept_topk_embed(const float* preds, int64_t ldp, int k, int topk,
                            const float* word, const float* pos, const float* gamma,
                            const float* beta, float eps, int64_t* labels, float* out,
                            void* out_bf16, int64_t ldo, int out_grp_rows, int out_row_off,
                            int B, int d, void* stream);

/*
 * care_embed_ln: decoder input embedding.
 *   Replaces Embeddings.forward (models/components/Embeddings.py:134-188):
 *   out[r] = LN(word[tok(r)] + pos[pos0 + r % seq] + sem[r / sem_div]) with
 *   tok(r) = tokens[(r / seq) * tok_stride + tok_off + r % seq] (int32).  sem may be NULL
 *   (no global semantic guidance).  seq = 1 for an incremental decode step.
 *   If anc is non-NULL (beam search) the token row is anc[(r / seq) * anc_stride + col]
 *   instead of (r / seq), col = tok_off + r % seq.
 */
int care_embed_ln(const int32_t* tokens, int tok_stride, int tok_off, const int32_t* anc,
                  int anc_stride, const float* word, const float* pos, int pos0,
                  const float* sem, int sem_div, const float* gamma, const float* beta,
                  float eps, float* out, void* out_bf16, int64_t ldo, int rows, int seq, int d,
                  void* stream);

/*
 * care_attention: scaled-dot-product attention for single query rows, head dim 64.
 *   Replaces ScaledDotProductAttention.forward after the projections
 *   (models/components/Attention.py:83-131): scores = q.k / 8 -> masked_fill(-1e9) ->
 *   + hybrid_bias[h, j] -> softmax -> P.V, for every (row, head).
 *   Q [rows, ldq] fp32 (head h at columns h*64..).  K and V of type kv_dtype; key j of
 *   query row r lives at  base + kvb * kv_batch_stride + j * kv_row_stride + h * 64  with
 *   kvb = anc ? anc[r * anc_stride + j] : r / rows_per_kv.
 *   Number of keys of row r: causal ? min(nkeys, r % seq + 1 + causal_off) : nkeys.
 *   pad_tok (int32, optional): key j of row r is masked when
 *   pad_tok[ptb * pad_stride + j] == pad_id, ptb = anc ? anc[r*anc_stride + j] : r / rows_per_kv
 *   (the key-pad mask of models/Decoder/Transformer.py:15-28,169-174).
 *   bias (optional) fp32 [H, bias_ld].  ctx [rows, ldctx] of type ctx_dtype (bf16 when the
 *   context only feeds a bf16 GEMM).  nkeys <= 128.
 */
int care_attention(const float* Q, int64_t ldq, const void* K, const void* V, int kv_dtype,
                   int64_t kv_batch_stride, int64_t kv_row_stride, int rows_per_kv,
                   const int32_t* anc, int anc_stride, int nkeys, int causal, int seq,
                   int causal_off, const int32_t* pad_tok, int pad_stride, int pad_id,
                   const float* bias, int bias_ld, void* ctx, int64_t ldctx, int ctx_dtype,
                   int rows, int heads, void* stream);

/*
 * care_attention_seq: the same attention (models/components/Attention.py:83-131) for WHOLE query sequences of
 *   seq <= 32 positions with bf16 operands - the teacher-forced forward (models/Framework.py:215-237,
 *   Decoder/Transformer.py:161-268) - one wave per (sequence, head), K and V read once for all positions,
 *   QK^T and PV on the matrix cores (csrc/attention_seq.hip).
 *   Q bf16 [nseq * seq, ldq] (head h at columns h*64..); key j of sequence s at
 *   base + (s / seqs_per_kv) * kv_batch_stride + j * kv_row_stride + h * 64 (elements, bf16) for K and V.
 *   causal: query position i sees keys j <= i.  pad_tok (int32, optional): key j is masked (-1e9, before the
 *   bias) when pad_tok[(s / seqs_per_kv) * pad_stride + j] == pad_id.  bias fp32 [heads, bias_ld] or NULL.
 *   ctx bf16 [nseq * seq, ldctx].  nkeys <= 128.
 */
int care_attention_seq(const void* Q, int64_t ldq, const void* K, const void* V, int64_t kv_batch_stride,
                       int64_t kv_row_stride, int seqs_per_kv, int nkeys, int causal, int seq,
                       const int32_t* pad_tok, int pad_stride, int pad_id, const float* bias, int bias_ld,
                       void* ctx, int64_t ldctx, int nseq, int heads, void* stream);

/*
 * care_attention_latent: the cross-attention of a decoder step with W_k / W_v absorbed into
 *   the query / context side (bf16 mode, d_model = 512).  Same reference lines as
 *   care_attention (models/components/Attention.py:63-67,83-131), re-associated:
 *     scores[h][j] = (W_k,h^T q_h / 8) . mem_j   (+ a per-head constant the softmax cancels)
 *     ct[h]        = sum_j softmax_j(scores[h][j] + bias[h][j]) mem_j
 *   so every step reads ONE bf16 copy of the memory row instead of projected K and V.
 *   qt  bf16 [rows, heads, d] (row stride ldq): the expanded, pre-scaled queries.
 *   mem bf16: key j of row r at  mem + (r / rows_per_kv) * mem_batch_stride + j * mem_row_stride.
 *   bias fp32 [heads, bias_ld] or NULL.  ct bf16 [rows, heads, d] (row stride ldc); the caller
 *   finishes ctx_h = W_v,h ct[h] + b_v,h.  nkeys <= 128, heads <= 16, d == 512 (one wave per row) or d == 1024
 *   (two waves per row, each owning 512 dims; the partial scores cross through LDS).
 */
int care_attention_latent(const void* qt, int64_t ldq, const void* mem, int64_t mem_batch_stride,
                          int64_t mem_row_stride, int rows_per_kv, int nkeys, const float* bias,
                          int bias_ld, void* ct, int64_t ldc, int rows, int heads, int d,
                          void* stream);

/*
 * care_head_expand / care_head_reduce: the per-head projections on either side of
 *   care_attention_latent (the key / value halves of models/components/Attention.py:63-67 moved
 *   from the memory to the query / context):
 *     qt[r][h][c]     = sum_e q[r][h*64+e] * wkt[h][c][e]        wkt = W_k,h^T / 8, bf16 [H][512][64]
 *     ctx[r][h*64+e]  = sum_c ct[r][h][c] * wv[h*64+e][c] + bv   wv = W_v bf16 [512][512], bv fp32 or NULL
 *   q bf16 [rows, ldq], qt / ct bf16 [rows, heads, 512] (row strides ldo / ldc), ctx bf16 [rows, ldo].
 *   d_model = 512, head dim 64.
 */
int care_head_expand(const void* q, int64_t ldq, const void* wkt, void* qt, int64_t ldo, int rows,
                     int heads, void* stream);
int care_head_reduce(const void* ct, int64_t ldc, const void* wv, const float* bv, void* ctx,
                     int64_t ldo, int rows, int heads, void* stream);

/*
 * care_beam_select: per row of logits [rows, ldl] (V valid columns): the beam_size best
 *   columns as log-probabilities.  Replaces torch.log_softmax(logits, dim=1)
 *   (models/Translator.py:127) plus the per-row part of the flattened topk of
 *   Beam.advance (misc/Decoding/Beam.py:60): cand_val[r, k] = (x - max) - log(sum exp(x - max))
 *   of the k-th best column (value desc, index asc), cand_idx[r, k] its column.  bm <= 8.
 *   waves_per_row: 1 (a 64-lane wave walks a row), or 4 for few rows (a workgroup per row, four partial lists merged:
 *   the same columns in the same order, the log-sum-exp added in another order).
 */
int care_beam_select(const float* logits, int64_t ldl, int V, int bm, float* cand_val,
                     int32_t* cand_idx, int rows, int waves_per_row, void* stream);

/*
 * care_ensemble_select: care_beam_select for a LIST of models (model ensembling, models/Translator.py:112-133): per row the
 *   beam_size best columns of the members' log_softmax rows AVERAGED equally - torch.stack(word_probs).mean(0), :130-131 - as they
 *   are (an average of log-probabilities is not renormalised; the reference's Beam.advance adds it to the beam's score as it is).
 *   logits: HOST array of n_models (<= 8) device pointers, each fp32 [rows, ldl] with V valid columns; cand_val / cand_idx
 *   [rows, bm] in the format care_beam_advance reads (value desc, column asc); bm <= 8.  The averaged array never exists.
 */
int care_ensemble_select(const float* const* logits, int n_models, int64_t ldl, int V, int bm, float* cand_val,
                         int32_t* cand_idx, int rows, void* stream);

/*
 * Fused beam selection (bf16 mode): the per-row top beam_size of log_softmax(x W^T) without the
 *   [rows, V] logits in memory.  Replaces torch.log_softmax (models/Translator.py:127) + the per-row
 *   part of Beam.advance's top-k (misc/Decoding/Beam.py:60) exactly like care_beam_select does:
 *     1. care_gemm_argmax_bf16_min : care_gemm_argmax_bf16 with at least `min_parts` column ranges
 *        (care_argmax_parts_bf16_min, called with the same M, N, K, a_dtype and min_parts, gives the
 *        count: at large row counts the kernel balances its column ranges over whole launch rounds)
 *        -> pmax / pidx / psum [M, parts];
 *     2. care_beam_threshold      : thr[r] = bm-th largest range maximum of row r (a lower bound of
 *        the row's bm-th best logit); cnt[r] = 0;
 *     3. care_gemm_collect_bf16   : the same product; every logit >= thr[r] is appended to row r's
 *        candidate list cval / cidx [M, cap] (cnt[r] counts them, also past cap);
 *     4. care_beam_pick           : cand_val / cand_idx [rows, bm] = the bm best candidates (value
 *        desc, column asc) as log-probabilities; a row with cnt > cap is recomputed exactly from
 *        A and W inside the kernel.
 */
int care_argmax_parts_bf16_min(int M, int N, int K, int a_dtype, int min_parts);
int care_gemm_argmax_bf16_min(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                              int32_t* pidx, float* psum, int M, int N, int K, int min_parts,
                              void* stream);
int care_beam_threshold(const float* pmax, int parts, int bm, float* thr, int32_t* cnt, int rows,
                        void* stream);
int care_gemm_collect_bf16(const void* A, int64_t lda, int a_dtype, const void* W, const float* thr,
                           int32_t* cnt, float* cval, int32_t* cidx, int cap, int M, int N, int K,
                           void* stream);
/*
 * The SPARSE second pass (large row counts, K = 512, bf16 rows: care_beam_sparse_applies): step 1 as
 *   care_gemm_argmax_bf16_tiles, which also writes tile_max [ceil(N / 32), M] fp32 - the maximum of every
 *   (32-column tile, row); step 3 as care_beam_sparse_collect, which lists per tile the rows whose tile
 *   maximum reaches thr[row] (tcount [2 tiles + 1], tlist [tiles, M] int32 scratch: counts, then the
 *   prefix sums of the 128-entry work units) and recomputes ONLY those
 *   (tile, row) products - the same MFMA chain as the first pass, bit-identical logits - appending the
 *   logits >= thr[row] to cval / cidx exactly like care_gemm_collect_bf16 (~2 % of its arithmetic).
 */
int care_beam_sparse_applies(int M, int N, int K, int a_dtype);
int care_gemm_argmax_bf16_tiles(const void* A, int64_t lda, int a_dtype, const void* W, float* pmax,
                                int32_t* pidx, float* psum, float* tile_max, int M, int N, int K,
                                int min_parts, void* stream);
int care_beam_sparse_collect(const void* A, int64_t lda, const void* W, const float* tile_max,
                             const float* thr, int32_t* cnt, float* cval, int32_t* cidx, int cap,
                             int32_t* tcount, int32_t* tlist, int M, int N, int K, void* stream);
int care_beam_pick(const float* pmax, const float* psum, int parts, const int32_t* cnt,
                   const float* cval, const int32_t* cidx, int cap, int bm, const void* A, int64_t lda,
                   int a_dtype, const void* W, int V, int K, float* cand_val, int32_t* cand_idx,
                   int rows, void* stream);

/*
 * Beam selection from GROUP MAXIMA of the LDS-tiled vocabulary product (16-bit modes, a few hundred to a few thousand
 *   rows; replaces the same reference lines as care_beam_select: models/Translator.py:127, misc/Decoding/Beam.py:60):
 *     1. care_gemm_tile_beam   : x W^T on csrc/gemm_tile.hip without writing logits - per (row, 64-column part)
 *        pmax / psum [M, parts] (maximum, sum exp(x - max)) and gmax [M, parts, 16]: the maxima of the part's sixteen
 *        4-column groups (-inf for groups past N); parts = care_argmax_parts_tile(N);
 *     2. care_beam_pick_groups : one wave per row - log-sum-exp from the parts; the bm parts with the largest maxima; the
 *        bm best of their 16 bm groups (a row's bm best logits lie in its bm best groups); those 4 bm logits recomputed
 *        from A and W (the tile kernel's MFMA, operand roles and K order: the same bits); cand_val / cand_idx [rows, bm]
 *        = the bm best as log-probabilities (value desc, column asc), the format care_beam_advance reads.
 *   A bf16 [M, lda] (lda % 8 == 0), W bf16 [N, K], K % 64 == 0, bm <= 8, 128 <= N <= 16384; gmax 16-byte aligned.
 */
int care_gemm_tile_beam(const void* A, int64_t lda, const void* W, float* pmax, float* psum, float* gmax, int M,
                        int N, int K, void* stream);
int care_beam_pick_groups(const float* pmax, const float* psum, const float* gmax, int parts, int bm, const void* A,
                          int64_t lda, const void* W, int V, int K, float* cand_val, int32_t* cand_idx, int rows,
                          void* stream);

/*
 * care_beam_advance: one beam-search step for every clip (B clips x bm beams), on device.
 *   Replaces Beam.advance / Beam.done (misc/Decoding/Beam.py:38-85), the re-ordering of the
 *   beams' prefixes (Beam.get_tentative_hypothesis, :112-117) and the removal of finished
 *   clips (models/Translator.py:145-209; clips are frozen instead, rows are independent).
 *   State, all device arrays: scores fp32 [B*bm]; tokphys int32 [B*bm, stride] physical
 *   token store (column 0 = BOS); anc_old -> anc_new int32 [B*bm, stride] ancestor tables
 *   (anc[row, j] = physical row holding position j of the hypothesis in beam slot `row`;
 *   initialise anc[row, 0] = row); done / n_fin int32 [B]; finished hypotheses
 *   fin_score fp32, fin_len int32 [B, fin_cap], fin_hyp int32 [B, fin_cap, stride]
 *   (tokens without BOS), recorded in the reference's order.  t = 1-based step,
 *   max_steps = max_len - 1, need = max(beam_size, topk) <= fin_cap, V = vocabulary size
 *   (ties between equal candidates resolve to the lower flattened index beam*V + column).
 *   One wave per clip: stride (= max_len) <= 64, bm <= 8.
 */
int care_beam_advance(const float* cand_val, const int32_t* cand_idx, float* scores, int bm,
                      int32_t* tokphys, const int32_t* anc_old, int32_t* anc_new,
                      int32_t* done, int32_t* n_fin, int fin_cap, float* fin_score,
                      int32_t* fin_len, int32_t* fin_hyp, int t, int max_steps, int need,
                      int eos_id, int V, int stride, int B, void* stream);

/*
 * care_attention_probs: the attention probabilities [rows, heads, nkeys] fp32 of one attention
 *   (scale 1/8, key-padding mask -> -1e9, then the optional per-head/per-key bias, softmax) - the
 *   `attention_probs` entries of the dict TransformerDecoder.forward returns
 *   (models/Decoder/Transformer.py:239-252, Attention.py:104-118).  Off the decode path: the fused
 *   kernels never materialise them; this is for the teacher-forced forward's auxiliary outputs.
 *   Q fp32 [rows, heads * 64]; K as in care_attention (no ancestor table); causal as there.
 */
int care_attention_probs(const float* Q, int64_t ldq, const void* K, int kv_dtype,
                         int64_t kv_batch_stride, int64_t kv_row_stride, int rows_per_kv, int nkeys,
                         int causal, int seq, const int32_t* pad_tok, int pad_stride, int pad_id,
                         const float* bias, int bias_ld, float* probs, int rows, int heads,
                         void* stream);

/*
 * Active-set compaction of the decode loop (csrc/compact.hip).  The reference ends a batch when every
 * instance is done and removes finished instances from every cached tensor at each step
 * (models/Translator.py:77-81,194-209); here the loop runs in segments and compacts between them.
 *   care_active_slots: idx[0 .. cnt) = the slots with finished == 0 in ascending order, idx[cnt .. n)
 *     = the finished ones in ascending order (a stable partition), count[0] = cnt.  n <= 2^30.
 *   care_gather_rows:  dst row i = src row idx[i], i < n.
 *   care_scatter_rows: dst row idx[i] = src row i, i < n; rows with idx[i] < 0 are skipped.
 *   row_bytes and the strides are multiples of 4 (16-byte vectors are used when everything allows).
 */
int care_active_slots(const int32_t* finished, int n, int32_t* idx, int32_t* count, void* stream);
int care_gather_rows(const void* src, int64_t src_stride_bytes, void* dst, int64_t dst_stride_bytes,
                     const int32_t* idx, int n, int64_t row_bytes, void* stream);
int care_scatter_rows(const void* src, int64_t src_stride_bytes, void* dst, int64_t dst_stride_bytes,
                      const int32_t* idx, int n, int64_t row_bytes, void* stream);
/* Beam search keeps bm consecutive rows per clip and ancestor tables that name physical rows:
 *   care_expand_index: idx_r[k * bm + j] = idx_c[k] * bm + j  (clip slots -> row slots), k < m;
 *   care_remap_rows:   anc[i] = cmap[anc[i] / bm] * bm + anc[i] % bm for the n entries of a table, after
 *     the rows of clip c have moved to clip cmap[c]. */
int care_expand_index(const int32_t* idx_c, int m, int bm, int32_t* idx_r, void* stream);
int care_remap_rows(int32_t* anc, int64_t n, const int32_t* cmap, int bm, void* stream);

/*
 * Training mode (models/Wrapper.py:423-435 -> models/Framework.py:215-237 under autograd; care_amd/training.py):
 *   everything of a backward pass that is not a GEMM (those re-use care_gemm on transposed operands).  fp32.
 *   care_ln_bwd: backward of care_add_ln (y = LN(x + res) gamma + beta): ds = dx = dres; dgamma / dbeta are
 *     ACCUMULATED (atomics) into zero-initialised [d] buffers.  d <= 2048.  (nn.LayerNorm of Encoder.py:167,
 *     SubLayers.py:73-79,147-150, Embeddings.py:84,185.)
 *   care_act: dy == NULL: out = act(z); else out = dy * act'(z)  (activations.py:3-16; exact-erf GELU).
 *   care_dropout: out = x * keep / (1 - p), keep = [u(seed, i) >= p] from a counter-based generator - the same call
 *     with the same seed is nn.Dropout's backward (RNG parity with torch is not a goal).
 *   care_strided_sum: out[r] = scale * sum_{k < terms} x[r * row_stride + k * term_stride]  (bias gradients, the
 *     backward of broadcast adds);  care_bcast_rows: dst[i] = scale * src[i / grp]  (backward of mean(1), Encoder.py:106).
 *   care_add_pos_sem: out[r] = x[r] + pos[r % seq] + sem[r / sem_div]  (Embeddings.py:170-176 before its LayerNorm).
 *   care_scatter_add_rows: table[idx[i]] += src[i]  (nn.Embedding backward; rows idx == skip_idx (padding_idx) skipped).
 *   care_concept_bwd: backward of care_concept_finish (pred_attribute.py:17-46) to the concept scores.
 *   care_attn_pv: ctx = dropout(P) V per (sequence, head); P [nseq * seq, heads, nkeys] from care_attention_probs.
 *   care_attn_bwd: backward of softmax(Q K^T / 8 + mask + bias) -> dropout -> . V  (Attention.py:83-131): dQ, dK, dV
 *     (written) and dbias [heads, bias_ld] (accumulated, optional).  nkeys <= 128, head dim 64; key / value block s of
 *     sequence s (no sharing between sequences: rows_per_kv = seq).
 */
int care_ln_bwd(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* gamma, const float* dy,
                int64_t lddy, float eps, float* ds, int64_t ldds, float* dgamma, float* dbeta, int rows, int d,
                void* stream);
int care_act(const float* z, const float* dy, float* out, int64_t n, int act, void* stream);
int care_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream);
int care_strided_sum(const float* x, int64_t ldx, float* out, int64_t ldo, int rows, int d, int terms,
                     int64_t row_stride, int64_t term_stride, float scale, void* stream);
int care_bcast_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int d, int grp, float scale,
                    void* stream);
int care_add_pos_sem(const float* x, const float* pos, const float* sem, float* out, int rows, int d, int seq,
                     int sem_div, void* stream);
int care_scatter_add_rows(const float* src, int64_t lds, const int32_t* idx, float* table, int64_t ldt, int rows,
                          int d, int skip_idx, void* stream);
int care_concept_bwd(const float* scores, int64_t lds, const float* dpreds, int64_t ldp, const float* davg, float* ds,
                     int64_t ldo, int B, int k, void* stream);
int care_attn_pv(const float* P, const float* V, int64_t kv_bs, int64_t kv_rs, float* ctx, int64_t ldc, int nseq,
                 int seq, int nkeys, int heads, float p_drop, uint64_t seed, void* stream);
int care_attn_bwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t kv_bs, int64_t kv_rs,
                  const float* P, const float* dctx, int64_t ldd, float* dQ, int64_t lddq, float* dK, float* dV,
                  int64_t dkv_bs, int64_t dkv_rs, float* dbias, int bias_ld, int nseq, int seq, int nkeys, int heads,
                  float p_drop, uint64_t seed, void* stream);

/*
 * care_gemm_kn: C [M, N] = op(A) B in exact f32 (v_mfma_f32_16x16x4_f32), B [K, N] row-major (ldb); a_is_km == 0: A is
 *   [M, K] row-major (lda), a_is_km != 0: A is stored [K, M] (the reduction index is its row).  The two products of an
 *   nn.Linear's backward as the operands lie in memory (training mode, models/Wrapper.py:423-435 under autograd):
 *   dx = dy W (a_is_km = 0, B = the [out, in] weight) and dW = dy^T x (a_is_km = 1, A = dy, B = x).  Any sizes and
 *   leading dimensions.
 */
int care_gemm_kn(const float* A, int64_t lda, int a_is_km, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N,
                 int K, void* stream);
/* The same product with K cut into `ksplit` ranges, range s multiplied into slab s of C (C + s * c_slab floats, each
 * [M, ldc]); the caller adds the slabs in order (care_strided_sum: terms = ksplit, row_stride = 1, term_stride = M rows:
 * deterministic).  For few output tiles
 * and a long K (dx = dlogits W: M = 1856, N = 512, K = 10547).  care_gemm_kn_splits(M, N, K) -> the ksplit to use (1: do
 * not split); any other ksplit must give every slab a range (ceil(K / ksplit) rounded up to 16) or CARE_ESHAPE.
 * Reference: the backward of nn.Linear under autograd (models/Wrapper.py:423-435 trains through it). */
int care_gemm_kn_splitk(const float* A, int64_t lda, int a_is_km, const float* B, int64_t ldb, float* C, int64_t ldc,
                        int64_t c_slab, int M, int N, int K, int ksplit, void* stream);
int care_gemm_kn_splits(int M, int N, int K);

/*
 * care_decode_resident: the whole greedy decode of a SMALL batch (1 .. a few hundred caption rows) as ONE launch.
 *   Replaces the step loop of Translator.translate_batch with beam_size 1 (models/Translator.py:77-143: embedding,
 *   models/Decoder.py + models/components/Layers.py:157-228 per layer, Head.py:26-32, the top-1 of Beam.advance, and
 *   the `no active instance` exit of Translator.py:77-81) for batches whose step is bound by launch latency: a grid of
 *   at most one workgroup per CU stays resident and walks the phases of every step, handed on through producer counters
 *   (csrc/decode_resident.hip).  bf16 weights / caches, fp32 accumulation and statistics - the rounding points of the
 *   multi-launch path with projected cross K/V.
 *   care_resident_attn: one post-LN attention block over STATIC keys (inter_attention / attr_attention): q_w [d, d],
 *   o_w [d, d] bf16, biases / LayerNorm fp32, kv bf16 [batches, nkeys, 2 d] (K | V, as care_gemm writes cross K/V) with
 *   kv_batch_stride elements between batches, row r reads batch r / rows_per_kv; bias (optional) fp32 [heads, bias_ld].
 *   care_resident_layer: qkv_w [3 d, d], o_w [d, d], w1 [ff, d], w2 [d, ff] bf16; self_kv bf16 [rows, T, 2 d] (written).
 *   word fp32 [V, d], pos fp32 [>= T, d], sem (optional) fp32 [rows / sem_div, d]; vocab_w bf16 [V, d].
 *   Outputs: fed int32 [rows, fed_stride >= T + 1] (column 0 = bos; columns after a row's end: the tokens it kept
 *   choosing while frozen, or 0 once every row had ended), score fp32 [rows] (sum of chosen log-probs), length int32
 *   [rows], finished int32 [rows].  All four are initialised by the kernel.
 *   early_exit != 0: stop after the step at which the last row ended (else all `steps` steps run).  The number of
 *   steps run is left in ((int32_t*)scratch)[2].
 *   scratch: care_decode_resident_scratch(rows, d, ff, V) bytes, 16-byte aligned.  blocks: workgroups (0 = as many as
 *   the widest phase has items, at most one per CU; every workgroup must be resident).
 *   Requires heads == d / 64, T <= 128, nkeys <= 128, n_layers <= 4, n_att <= 2, V <= 16384 and either d == 512 with ff in
 *   {512, 1024, 2048} (any row count one workgroup per 16-row tile fits) or d in {768, 1024} with ff == 4 d and
 *   rows <= 128 (config/archs.yaml:15-26: the `median` / `large` architectures, K-split forms in every phase).
 *   Every workgroup must be resident at the same time (they wait for one another): the grid is at most one workgroup
 *   per CU and the entry point refuses (CARE_ESHAPE, nothing enqueued) unless hipOccupancyMaxActiveBlocksPerMultiprocessor
 *   admits a workgroup of the kernel per CU; do not run two of these launches concurrently on different streams.  A
 *   workgroup that waits ~2 s for a phase's producers aborts the launch: EVERY row's length = -1.
 *   Tuning knobs, read from the environment ONCE per process: CARE_RESIDENT_RB, CARE_RESIDENT_SMALL,
 *   CARE_RESIDENT_HALF_ROWS (forms of three phases), CARE_RESIDENT_BEAM_CFG (the beam launch's form).
 *   care_decode_resident_debug(prof_step, ghost): tools / tests only - phase clocks of step `prof_step` into the scratch
 *   (tools/resident_prof.py); ghost != 0: phases that can never complete (the watchdog test).  Process-wide, default 0 / 0.
 *   The entry points that issue two operations: a 52-KB memset node (hand-off counters) and the kernel.
 */
typedef struct care_resident_attn {
  const void* q_w; const float* q_b; const void* o_w; const float* o_b; const float* ln_g; const float* ln_b;
  const void* kv; int64_t kv_batch_stride; int32_t nkeys; int32_t rows_per_kv; const float* bias; int32_t bias_ld;
  int32_t reserved;
} care_resident_attn;
typedef struct care_resident_layer {
  const void* qkv_w; const float* qkv_b; const void* o_w; const float* o_b; const float* ln_g; const float* ln_b;
  void* self_kv;
  care_resident_attn att[2]; int32_t n_att; int32_t reserved;
  const void* w1; const float* b1; const void* w2; const float* b2; const float* ffn_g; const float* ffn_b;
} care_resident_layer;
int64_t care_decode_resident_scratch(int rows, int d, int ff, int V);
int care_decode_resident(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                         const float* sem, int sem_div, const float* emb_g, const float* emb_b, float eps,
                         const void* vocab_w, int V, int d, int heads, int ff, int act, int rows, int T, int steps,
                         int bos, int eos, int pad, int32_t* fed, int fed_stride, float* score, int32_t* length,
                         int32_t* finished, void* scratch, int64_t scratch_bytes, int early_exit, int blocks, void* stream);

void care_decode_resident_debug(int prof_step, int ghost);

/*
 * The hand-off between the phases of a resident launch (care_decode_resident, care_decode_resident_beam): by default
 * (mode -1) fence-free - sc1 write-through stores, a drained vmcnt, one relaxed atomic add; sc1 polls and loads - on the
 * configuration that form was validated on (a gfx950 device with all 256 CUs in one partition), and with an agent-scope
 * release before the add / acquire after the poll on any other device or partition mode (+ 1 .. 4 us per hand-off).
 * care_resident_set_fenced(1 / 0) forces one form for the process (also CARE_RESIDENT_FENCED in the environment, read
 * once); care_resident_fenced() = the form a launch on the current device would take.
 */
void care_resident_set_fenced(int mode);
int care_resident_fenced(void);

/*
 * care_decode_resident_beam: BEAM SEARCH over a small batch (clips x beam <= a few hundred rows) as ONE launch.
 *   Replaces the step loop of Translator.translate_batch for beam_size > 1 (models/Translator.py:77-143 with
 *   predict_word's log_softmax :127, Beam.advance misc/Decoding/Beam.py:45-85 incl. its quirks - first step row 0 only,
 *   ended beams offer nothing, hypotheses collected in beam order until `need` = max(beam_size, topk) have ended, forced
 *   finish at max_len - and the `no active instance` exit of Translator.py:77-81): translate.py's default decode (beam 5,
 *   batch 128; --latency: batch 1).  Same machinery and rounding points as care_decode_resident; the vocabulary phase
 *   keeps every row's best 4-column groups and the advance phase recomputes their logits bit for bit
 *   (csrc/decode_resident_beam.hip), so the [rows, V] logits never exist.
 *   layers as for care_decode_resident with self_kv bf16 [clips * beam, T, 2 d] and rows_per_kv = beam in every
 *   attention block; sem (optional) fp32 [clips, d].
 *   State and outputs (all initialised by the kernel), as care_beam_advance: tok / anc0 / anc1 int32 [clips * beam, stride
 *   >= T + 1] (token table and the two ancestor tables), scores fp32 [clips * beam], done / nfin int32 [clips], fscore fp32 /
 *   flen int32 [clips, fin_cap], fhyp int32 [clips, fin_cap, stride]; fin_cap >= need + beam.  Steps run:
 *   ((int32_t*)scratch)[2].  Aborted launch (see care_decode_resident): EVERY nfin = -1.
 *   Requires heads == d / 64, T <= 63, beam <= 8 (beam 6 .. 8: a second instance of the launch with 8 groups kept per list), V <= 16384
 *   and either d == 512 with ff in {512, 1024, 2048} or d in {768, 1024}
 *   with ff == 4 d and clips * beam <= 128 (config/archs.yaml:15-26: the `median` / `large` architectures - VATEX runs
 *   translate.py with its default beam 5 - in the K-split forms of every phase).
 *   scratch: care_decode_resident_beam_scratch(clips, beam, d, ff, V) bytes, 16-byte aligned.
 */
int64_t care_decode_resident_beam_scratch(int clips, int beam, int d, int ff, int V);
int care_decode_resident_beam(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                              const float* sem, const float* emb_g, const float* emb_b, float eps, const void* vocab_w,
                              int V, int d, int heads, int ff, int act, int clips, int beam, int need, int T, int steps,
                              int bos, int eos, int pad, int32_t* tok, int stride, int32_t* anc0, int32_t* anc1,
                              float* scores, int32_t* done, int32_t* nfin, float* fscore, int32_t* flen, int32_t* fhyp,
                              int fin_cap, void* scratch, int64_t scratch_bytes, int early_exit, int blocks, void* stream);

/* care_timestamp: out[0] (uint64) = the device wall clock (constant 100 MHz) when the one-thread kernel runs.
 *   Measurement only (bench.py: the duration of a kernel inside a replayed hipGraph); no reference counterpart. */
int care_timestamp(void* out, void* stream);

/*
 * care_decode_chain_beam: steps t0 .. t1 of a BEAM SEARCH as a chain of kernels - the phases of care_decode_resident_beam,
 *   each a launch of its own (10 per step for a one-layer decoder) with a grid sized by the phase's items; for batches
 *   beyond what one resident launch serves well (a few hundred rows) up to thousands of rows.  Replaces the same reference
 *   code as care_decode_resident_beam (models/Translator.py:77-143, misc/Decoding/Beam.py:45-85) with the SAME arithmetic
 *   per row (the device code is shared; plain loads / stores instead of agent-scope ones): hypotheses and scores are
 *   bit-identical to the resident launch's.  Arguments as care_decode_resident_beam; [t0, t1] the steps to run (1-based;
 *   t0 == 1 initialises the beam state and zeroes the `clips done` counter ((uint32_t*)scratch)[1]; later calls continue
 *   on the same state and scratch); the caller decides between calls whether any clip is still live (`done`).  There is
 *   no residency condition and no hand-off protocol: any grid, any number of concurrent streams.  form: -1 = by row
 *   count (<= 64 rows K-split items; <= 256 one row tile per workgroup; beyond, 2 / 2 / 2 / 4 row tiles per weight fetch
 *   in QKV / the d x d products / FFN dense1 / the vocabulary), 0 / 1 / 3 force one - a row's bits are the same in all.
 *   scratch: care_decode_chain_beam_scratch(clips, beam, d, ff, V) bytes, 16-byte aligned.
 *   Requires d == 512, heads == 8, ff in {512, 1024, 2048}, beam <= 5, T <= 63, V <= 16384, nkeys <= 128, n_att <= 2.
 */
int64_t care_decode_chain_beam_scratch(int clips, int beam, int d, int ff, int V);
int care_decode_chain_beam(const care_resident_layer* layers, int n_layers, const float* word, const float* pos,
                           const float* sem, const float* emb_g, const float* emb_b, float eps, const void* vocab_w,
                           int V, int d, int heads, int ff, int act, int clips, int beam, int need, int T, int t0, int t1,
                           int bos, int eos, int pad, int32_t* tok, int stride, int32_t* anc0, int32_t* anc1,
                           float* scores, int32_t* done, int32_t* nfin, float* fscore, int32_t* flen, int32_t* fhyp,
                           int fin_cap, void* scratch, int64_t scratch_bytes, int form, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARE_HIP_H */
