/*
 * sparsify_hip.h -- C ABI of libsparsify_hip.so: the MI355X (gfx950) hot path of the
 * sparse-autoencoder / transcoder train step.
 *
 * The reference (EleutherAI/sparsify v1.3.0) has no C ABI: its seams are Python
 * (torch.autograd.Function / module-level function pointers). Each
 * entry point below names the reference interface it replaces (file:line under the
 * reference tree). The Python host side (sparsify_amd/) binds these with ctypes and
 * keeps the reference's operator surface; INTEGRATION.md shows the binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host";
 *  - tensors are dense row-major; N tokens, D = d_in, M latents, k active latents;
 *  - indices cross the boundary as int64 (the reference's torch.topk dtype);
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call
 *    only enqueues work on that stream and never synchronises or allocates;
 *  - dtype codes: SPHIP_F32 / SPHIP_BF16;
 *  - return value 0 = success; otherwise an SPHIP_E* code, and sphip_last_error()
 *    returns a thread-local, human readable message. Shapes are validated on the host
 *    before any kernel is launched.
 */
#ifndef SPARSIFY_HIP_H
#define SPARSIFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHIP_ABI_VERSION 3

enum { SPHIP_F32 = 0, SPHIP_BF16 = 1 };
enum { SPHIP_OK = 0, SPHIP_EINVAL = 1, SPHIP_EUNSUPPORTED = 2, SPHIP_ELAUNCH = 3, SPHIP_EWORKSPACE = 4 };

int sphip_abi_version(void);
const char* sphip_last_error(void);

/* ---------------------------------------------------------------- encoder (forward) */

/* SparseCoder.encode prologue, sparse_coder.py:191-192: xc = x - b_dec (fp32 promote).
 * Writes any of: xc_f32 (N,D) fp32, xc_bf16 (N,D) bf16 (the operand autocast would feed to
 * the linear, sparse_coder.py:205-209). b_dec may be NULL (transcoder: plain cast). */
int sphip_center_cast(const void* x, int x_dtype, const float* b_dec, float* xc_f32,
                      void* xc_bf16, int64_t N, int64_t D, void* stream);

/* fp32 -> bf16 (round to nearest even): the cast autocast applies to encoder.weight /
 * encoder.bias on every call (sparse_coder.py:205-209); here done once per optimizer step. */
int sphip_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);

/* FusedEncoder.forward, fused_encoder.py:29: pre = relu(x W^T + b).
 * x (N,D), W (M,D), bias (M) or NULL, pre (N,M); all of `dtype` (bf16: MFMA bf16 with fp32
 * accumulate, one rounding to bf16 in the epilogue; f32: exact-fp32 MFMA). relu = 0 keeps
 * the raw affine output. Requires D % 8 == 0 (bf16) / D % 4 == 0 (f32).
 * tile_max, or NULL: the maximum of pre over each block of SPHIP_TOPK_TILE consecutive
 * latents (taken in the GEMM epilogue for the large bf16 shapes, by a small extra kernel
 * otherwise); sphip_topk uses it to skip the blocks of a row that cannot hold a selected
 * latent. ceil(N/64) * T * 64 elements of `dtype`, T = ceil(M / SPHIP_TOPK_TILE); the
 * maximum of block t of token n is element ((n / 64) * T + t) * 64 + n % 64 (64 tokens of
 * one block are contiguous: the GEMM writes whole 128-byte lines). */
#define SPHIP_TOPK_TILE 128
int sphip_encoder_preacts(const void* x, const void* W, const void* bias, void* pre,
                          void* tile_max, int dtype, int64_t N, int64_t D, int64_t M, int relu,
                          void* stream);

/* The same product (fused_encoder.py:29) for one block of M_block consecutive latents of a
 * wider encoder: W, bias, pre and tile_max point at the block's first latent (row of W,
 * element of bias, column of pre, block of tile_max); ld_pre is the row stride of pre and
 * tiles_total the row length T of tile_max, i.e. those of the whole (N, M) matrix. Calls over
 * disjoint blocks fill the same pre / tile_max as one sphip_encoder_preacts call, so a
 * data-parallel trainer can start the forward on the latents whose averaged update
 * (trainer.py:457-468) has arrived while the rest is still on the wire. bf16 only,
 * M_block % SPHIP_TOPK_TILE == 0, ld_pre % 8 == 0, D % 64 == 0. */
/* 1 if sphip_encoder_preacts computes this shape with the 256x256 kernel (the one
 * sphip_encoder_preacts_block always uses), 0 for the 128x128 kernel. The two round a few
 * outputs differently (different fp32 summation order), so a caller that recomputes a subset of
 * the latents and needs the very same values (the compact AuxK selection) asks which one ran. */
int sphip_encoder_preacts_uses_tile256(int dtype, int64_t N, int64_t D, int64_t M);
int sphip_encoder_preacts_block(const void* x, const void* W, const void* bias, void* pre,
                                void* tile_max, int dtype, int64_t N, int64_t D, int64_t M_block,
                                int64_t ld_pre, int64_t tiles_total, int relu, void* stream);

/* torch.topk(pre, k, dim=-1, sorted=False), fused_encoder.py:33 -- and, with keep_mask,
 * torch.where(dead_mask[None], pre, -inf).topk(k_aux), sparse_coder.py:240-243.
 * pre (N,M) of `dtype`; keep_mask (M) uint8 or NULL; vals (N,k) of `dtype`; idx (N,k) int64.
 * The selected SET is deterministic (ties go to the lowest index); the order inside a row is
 * unspecified like sorted=False (value-descending on the usual path, index order on the
 * exact-bisection path taken for k > 1024 or rows dominated by ties).
 * Masked-out latents rank as -inf and are selected only if fewer than k are kept.
 * tile_max: the block maxima written by sphip_encoder_preacts for this `pre` (same layout),
 * or NULL. When given (and keep_mask is NULL) only the blocks whose maximum
 * reaches the k-th largest block maximum are read; the result is identical either way. */
int sphip_topk(const void* pre, int dtype, const uint8_t* keep_mask, const void* tile_max,
               int64_t N, int64_t M, int64_t k, void* vals, int64_t* idx, void* stream);

/* sparse_coder.py:256-262 (multi-top-k): `pre_acts.topk(4k)` holds the row's top-k set, whose rows
 * the main decode (:201) has summed already. idx_out (N,kw) = idx_wide with every member of the
 * narrower selection (vals_sel, idx_sel: N x k of the same row, by sphip_topk's set rule: all values
 * above its smallest plus the lowest-index ties) replaced by -1, the inactive-slot mark of the decode
 * and gradient entry points: decode(vals_wide, idx_out) + the main decode's output is the 4k
 * reconstruction with 3k instead of 4k rows gathered per token. Order inside the rows is free. */
int sphip_exclude_selected(const void* vals_wide, const int64_t* idx_wide, int64_t kw,
                           const void* vals_sel, const int64_t* idx_sel, int64_t k, int dtype,
                           int64_t N, int64_t* idx_out, void* stream);

/* ---- top-k without the (N, M) matrix (fused_encoder.py:29-33 when nobody reads pre_acts;
 * the reference always materialises it, fused_encoder.py:50, and only AuxK / multi-top-k read it,
 * sparse_coder.py:230-262). bf16, relu, k <= 256, D % 64 == 0. Sequence for one batch, all on
 * one stream (csrc/lazy.hip has the argument for exactness):
 *   1. sphip_encoder_preacts + sphip_topk over the first M_a latents (any split with
 *      M_a % 256 == 0, M_a >= k; about M/4)            -> vals_a, idx_a
 *   2. sphip_row_min(vals_a)                            -> tau (N) bf16 bits
 *   3. sphip_encoder_preacts_emit over latents [M_a, M): nothing of the product is stored but
 *      the 16-byte chunks (8 latents) holding a value above the token's tau, appended to the
 *      token's list in the workspace (capacity `cap` chunks per token, cap <= 4096); reset != 0
 *      clears the counters and the flag first (0: a further block of the same batch)
 *   4. sphip_topk_candidates -> vals, idx: the same set as sphip_topk over the whole row,
 *      value-descending, ties to the lowest index
 *   5. the first int32 of the workspace is a flag, raised when a token's list or the sort
 *      buffer (1024 entries at or above the k-th value) overflowed; rows may then be wrong.
 *      sphip_encoder_preacts_if + sphip_topk_if enqueued behind with run_if = that word redo
 *      the ordinary path over all rows when it is up and return at once when it is not
 *      (bf16, D % 64 == 0, M % 8 == 0; same arguments as the unpredicated calls otherwise).
 * Workspace: sphip_lazy_workspace_bytes(N, cap), 256-byte aligned. */
size_t sphip_lazy_workspace_bytes(int64_t N, int64_t cap);   /* 0: cap out of [1, 4096] */
int sphip_row_min(const void* vals, int64_t N, int64_t k, void* out, void* stream);
int sphip_encoder_preacts_emit(const void* x, const void* W, const void* bias, int64_t N, int64_t D,
                               int64_t M_block, int64_t first_latent, const void* tau, int64_t cap,
                               int reset, void* ws, int64_t ws_bytes, void* stream);
int sphip_topk_candidates(const void* vals_a, const int64_t* idx_a, const void* tau, int64_t N,
                          int64_t k, int64_t cap, void* ws, int64_t ws_bytes, void* vals,
                          int64_t* idx, void* stream);
int sphip_encoder_preacts_if(const int* run_if, const void* x, const void* W, const void* bias,
                             void* pre, void* tile_max, int64_t N, int64_t D, int64_t M, int relu,
                             void* stream);
int sphip_topk_if(const int* run_if, const void* pre, int dtype, const uint8_t* keep_mask,
                  const void* tile_max, int64_t N, int64_t M, int64_t k, void* vals, int64_t* idx,
                  void* stream);

/* groupmax, fused_encoder.py:34-43: max over k contiguous groups of M/k latents. M % k == 0. */
int sphip_groupmax(const void* pre, int dtype, int64_t N, int64_t M, int64_t k, void* vals,
                   int64_t* idx, void* stream);

/* ---------------------------------------------------------------- sparse decode / gathers */

/* decoder_impl / eager_decode / embedding_bag_k, utils.py:81-90, xformers.py:9-53, and the
 * grad-input gather of FusedEncoder.backward, fused_encoder.py:58-64:
 *   out[n,:] = sum_s coeff[n,s] * table[idx[n,s],:]  (+ bias[:] if bias != NULL)
 * idx (N,k) int64 in [0,M); coeff (N,k) of coeff_dtype; table (M,D) fp32; out (N,D) fp32.
 * fp32 accumulation in slot order s = 0..k-1 (deterministic). An index outside [0,M) --
 * negative, or >= M -- marks an inactive slot (latent-sharded decode: the latent lives on
 * another GPU) and is skipped, here and in the segment / gradient entry points below;
 * torch.embedding_bag (utils.py:88) raises on such an index instead: a binding that wants that
 * checks the range before the call. */
int sphip_gather_rows_weighted(const int64_t* idx, const void* coeff, int coeff_dtype,
                               const float* table, const float* bias, float* out, int64_t N,
                               int64_t k, int64_t M, int64_t D, void* stream);

/* The same decode (utils.py:81-90, xformers.py:9-53) for tables far beyond the caches, as an
 * L2-resident gather: the latents are cut into blocks of 16384 rows, one launch per block
 * gathers 128-byte row pieces from 2 MiB of the table per XCD and adds its partial sums to
 * out (the first launch writes bias + partial). Same arguments as sphip_gather_rows_weighted
 * plus a workspace of sphip_gather_l2_workspace_bytes(N, k, M, D) bytes (256-byte aligned);
 * that query returns 0 when the shape is not eligible (small tables, D % 32 != 0, fewer than
 * 8 slots per token and block, ...): call sphip_gather_rows_weighted then. The fp32 sum runs
 * block by block, slots in order inside a block (deterministic; differs from the slot-order
 * sum of sphip_gather_rows_weighted by rounding only). Negative indices are skipped. */
size_t sphip_gather_l2_workspace_bytes(int64_t N, int64_t k, int64_t M, int64_t D);
/* table_slice_major: NULL, or the copy of table that sphip_slice_major_f32 writes. The gather then
 * reads that copy: in the row-major table the 128-byte pieces of a column slice lie D*4 bytes
 * apart, a power-of-two stride that maps a latent block onto a few L2 sets (hit rate 27 %
 * measured); in the copy they are contiguous. */
int sphip_gather_rows_weighted_l2(const int64_t* idx, const void* coeff, int coeff_dtype,
                                  const float* table, const float* table_slice_major,
                                  const float* bias, float* out, int64_t N, int64_t k, int64_t M,
                                  int64_t D, void* workspace, size_t workspace_bytes, void* stream);
/* src (R, D) fp32 row-major -> dst [D/32][R][32]: every 128-byte column slice of the table as one
 * contiguous R x 128 B block (no reference counterpart: a layout for the gather above). */
int sphip_slice_major_f32(const float* src, float* dst, int64_t R, int64_t D, void* stream);

/* count_per_embedding_k + cumsum + map_embeddings_and_outputs_k, xformers.py:56-85,139-163:
 * counting sort of the N*k (token,slot) entries by latent id.
 *   seg_begin (M+1) int32: entries of latent j are perm[seg_begin[j] : seg_begin[j+1]]
 *   perm (N*k) uint32   : entry ids e = n*k + s, ascending inside a segment (the reference's
 *                         order inside a segment is arrival order of atomics; ours is fixed).
 * Negative (inactive) ids are placed after all real segments: seg_begin[M] = number of
 * active entries, and the reduction kernels never visit them (their dcoeff is 0).
 * Requires N*k < 2^31 (as xformers.py:156). workspace: sphip_segments_workspace_bytes. */
size_t sphip_segments_workspace_bytes(int64_t N, int64_t k, int64_t M);
int sphip_build_latent_segments(const int64_t* idx, int64_t N, int64_t k, int64_t M,
                                uint32_t* perm, int32_t* seg_begin, void* workspace,
                                size_t workspace_bytes, void* stream);

/* aggregate_gradient_for_embedding_k, xformers.py:88-136 (decoder) and the index_add_
 * weight/bias gradients of FusedEncoder.backward, fused_encoder.py:67-84 (encoder):
 *   gW[j,:]      = sum_{e in seg j} coeff[e] * rows[e / k, :]          (all M rows written)
 *   gbias[j]     = sum_{e in seg j} coeff[e]                           (if gbias != NULL)
 *   dcoeff[e]    = < rows[e / k, :], W[j,:] >                          (if dcoeff != NULL)
 * row_shift (D) fp32 or NULL: when given, the rows are taken as rows[n,:] - row_shift[:]
 *   for gW only (the encoder input x - b_dec, sparse_coder.py:192, without materialising
 *   it: gW[j,:] = sum c*rows - (sum c)*row_shift).
 * coeff (N,k) of coeff_dtype, rows (N,D) of rows_dtype, W (M,D) fp32, outputs fp32.
 * Sums run in a fixed order, so results are bitwise reproducible: a segment of up to 128 entries is
 *   one chain in segment order; at training sizes (rows of 8 MiB and more) a longer one is cut into
 *   128-entry chunks, each a chain, whose partial rows are added in chunk order (the reference's
 *   atomics leave the order to arrival). gbias is the same sum from every entry point of this
 *   family (sphip_segment_coeff_sums' order).
 * workspace (sphip_weight_grad_workspace_bytes): the segment-ordered (entry, coeff) list plus
 *   either a slice-major copy of rows (no dcoeff) or the per-column-slice partial dots of
 *   dcoeff (D/32 floats per entry), whichever is larger, plus the long-segment tables (list of
 *   latents above 128 entries, chunk table, one partial row of D floats per chunk: at most
 *   2*N*k/128 + 2 chunks). Take the size from the call. */
size_t sphip_weight_grad_workspace_bytes(int64_t N, int64_t k, int64_t D);
int sphip_segmented_weight_grad(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                                float* gW, float* gbias, float* dcoeff, const float* row_shift,
                                int64_t N, int64_t k, int64_t M, int64_t D, void* workspace,
                                size_t workspace_bytes, void* stream);
/* The same for a batch taken in token blocks (the reference's index_add_ / atomics accumulate over
 * any number of rows, fused_encoder.py:67-84, xformers.py:110-136; here a block's N x 128 B row
 * slices have to fit an XCD's L2, i.e. N <= 24576): call once per block of tokens with that block's
 * segments, coeff, rows and dcoeff; accumulate = 0 writes gW / gbias, != 0 adds the block's sums to
 * what an earlier call left there (fixed order: block after block). Built into the L2-sliced kernels
 * only: SPHIP_EUNSUPPORTED with accumulate != 0 for a shape that takes another form (nothing has
 * been launched then; the caller sums the blocks' outputs itself). sphip_weight_grad_accumulates
 * answers for a shape beforehand: ask for every block before the first call adds anything. */
int sphip_weight_grad_accumulates(int64_t N, int64_t k, int64_t M, int64_t D, int rows_dtype,
                                  int want_dcoeff);   /* 1: accumulate != 0 is available for this shape */
int sphip_segmented_weight_grad_acc(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                    int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                                    float* gW, float* gbias, float* dcoeff, const float* row_shift,
                                    int64_t N, int64_t k, int64_t M, int64_t D, void* workspace,
                                    size_t workspace_bytes, int accumulate, void* stream);

/* ---------------------------------------------------------------- loss (FVU / AuxK) */

/* sparse_coder.py:228: tv = sum((y - mean_0(y))^2). y (N,D) of y_dtype; out tv[0] fp32.
 * workspace: sphip_reduce_workspace_bytes(N, D). Two-pass, fixed summation order. For bf16 y
 * the mean and y - mean are rounded to bf16 as the reference's autocast does; squares and
 * sums are fp32. */
size_t sphip_reduce_workspace_bytes(int64_t N, int64_t D);
int sphip_total_variance(const void* y, int y_dtype, int64_t N, int64_t D, float* tv,
                         void* workspace, size_t workspace_bytes, void* stream);

/* sphip_total_variance for callers that cannot allocate or pass a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc). No counterpart in the reference.
 * Same kernels and the same result, bit for bit; the partial sums live in a file-scope device
 * array of the library (8.06 MiB per GPU the library runs on, held until the process ends)
 * instead of a workspace, so the call allocates nothing and takes no workspace argument.
 * Limit: D <= 8192 (SPHIP_EUNSUPPORTED beyond; use sphip_total_variance). Calls that overlap on
 * one GPU (different streams or threads) share that GPU's array and race: serialise them. */
int sphip_total_variance_static(const void* y, int y_dtype, int64_t N, int64_t D, float* tv,
                                void* stream);

/* sparse_coder.py:225,253 (and :248, :260 with other operands): e = a - b; sq[0] = sum(e^2).
 * a (N,D) of a_dtype, b (N,D) fp32, e (N,D) fp32 or NULL. */
int sphip_residual_sumsq(const void* a, int a_dtype, const float* b, float* e, float* sq,
                         int64_t N, int64_t D, void* workspace, size_t workspace_bytes,
                         void* stream);

/* out = (alpha * num[0] / den[0]) * in  -- e.g. dL/d sae_out = (-2 * go / tv) * e, the
 * backward of sparse_coder.py:253-254. num/den are device scalars (no host sync). */
int sphip_scale_by_ratio(const float* in, float* out, int64_t n, float alpha, const float* num,
                         const float* den, void* stream);

/* column sums: out[d] = sum_n a[n,d] (b_dec.grad from `+ self.b_dec`, sparse_coder.py:202;
 * mean_0 for the step-0 bias init, trainer.py:399-404, with scale = 1/N). */
int sphip_colsum(const void* a, int a_dtype, int64_t N, int64_t D, float scale, float* out,
                 void* workspace, size_t workspace_bytes, void* stream);

/* out[d] = sum_j v[j] * W[j,d]: the column sum of FusedEncoder's grad-input,
 * sum_n grad_x[n,:] = gb^T W_enc, which is all b_dec needs from it (sparse_coder.py:192 +
 * fused_encoder.py:58-64). W (M,D) fp32, v (M) fp32, out (D) fp32. */
int sphip_vec_mat(const float* v, const float* W, int64_t M, int64_t D, float scale, float* out,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- dense parameter updates */

/* set_decoder_norm_to_unit_norm, sparse_coder.py:273-279: W /= ||W||_row + eps, in place. */
int sphip_unit_norm_rows(float* W, int64_t M, int64_t D, void* stream);

/* remove_gradient_parallel_to_decoder_directions, sparse_coder.py:281-295, G in place. */
int sphip_remove_parallel_grad(float* G, const float* W, int64_t M, int64_t D, void* stream);

/* SignSGD.step, sign_sgd.py:23: p -= lr * sign(g). Optionally refreshes a bf16 shadow of p
 * in the same pass (bf16_shadow may be NULL). */
int sphip_signsgd_step(float* p, const float* g, int64_t n, float lr, void* bf16_shadow,
                       void* stream);
/* The same step (sign_sgd.py:23) for an (M, D) matrix whose slice-major copy [D/32][M][32] (see
 * sphip_slice_major_f32) is rewritten in the same pass while the rows are in registers: a decoder
 * that is neither projected nor normalised (a transcoder's, trainer.py:406-408 / 513-518 do not
 * apply) read by the L2-resident decode. slice_major may be NULL. Bit for bit sphip_signsgd_step. */
int sphip_signsgd_step_rows(float* W, const float* G, int64_t M, int64_t D, float lr,
                            float* slice_major, void* stream);

/* torch.optim.Adam's step (the "adam" optimizer of trainer.py:92-122 when bitsandbytes is absent:
 * betas, eps, no weight decay, no amsgrad) for one parameter: exp_avg / exp_avg_sq / p updated in
 * one pass, operations in the order of torch's single-tensor implementation, bias corrections
 * 1 - beta^step and the scalars 1 - beta, lr / (1 - beta1^step) computed in double on the host
 * (hence the double arguments) and rounded to fp32 once; `step` counts from 1. Optionally refreshes a bf16
 * shadow of p in the same pass (bf16_shadow may be NULL). */
int sphip_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, int64_t step, void* bf16_shadow,
                    void* stream);

/* trainer.py:513-518 fused for one decoder matrix: projected gradient (not written back)
 * followed by the SignSGD step on W, one pass over (W, G). renorm != 0 additionally applies
 * set_decoder_norm_to_unit_norm (sparse_coder.py:273-279) to the updated rows in the same
 * pass -- the normalisation that opens the next train step (trainer.py:408); the result is
 * bitwise that of the step followed by sphip_unit_norm_rows.
 * slice_major (or NULL): the [D/32][M][32] copy of W that sphip_gather_rows_weighted_l2 reads
 * (sphip_slice_major_f32) is rewritten from the finished rows while they are in registers, so
 * the next decode needs no separate copy pass; D % 32 == 0. */
int sphip_decoder_project_signsgd(float* W, const float* G, int64_t M, int64_t D, float lr,
                                  int renorm, float* slice_major, void* stream);

/* ---------------------------------------------------------------- dead-latent bookkeeping */

/* ScheduleFreeWrapper(SignSGD, momentum = beta).step() -- the reference's default optimizer
 * "signum", trainer.py:146-162 -- in one pass over (p, z, g): p holds y (where gradients are
 * taken), z the SignSGD iterate; x = lerp(y, z, 1 - 1/beta); z -= lr sign(g);
 * x = lerp(x, z, c); y = lerp(x, z, 1 - beta), lerp as torch.lerp. The schedule-free package is
 * not part of the reference tree (parity unpinned; the host class schedule_free.py states the
 * algorithm it restates). The decoder form first projects g on the orthogonal complement of
 * the row (sparse_coder.py:281-295) and, with renorm != 0, normalises the new row
 * (sparse_coder.py:273-279); D <= 8192. slice_major: NULL, or the [D/32][M][32] copy of W
 * (sphip_slice_major_f32) to rewrite with the finished rows in the same pass (D % 32 == 0). */
int sphip_schedulefree_signsgd_step(float* p, float* z, const float* g, int64_t n, float lr,
                                    float beta, float c, void* bf16_shadow, void* stream);
int sphip_decoder_schedulefree_signsgd_step(float* W, float* Z, const float* G, int64_t M,
                                            int64_t D, float lr, float beta, float c, int renorm,
                                            float* slice_major, void* stream);
/* sphip_schedulefree_signsgd_step for an (M, D) matrix with its slice-major copy rewritten in the same
 * pass and no projection (a transcoder's decoder; D % 4 == 0, D <= 8192). Bit for bit the flat step. */
int sphip_schedulefree_signsgd_step_rows(float* W, float* Z, const float* G, int64_t M, int64_t D,
                                         float lr, float beta, float c, float* slice_major,
                                         void* stream);

/* The encoder weight gradient fused with its SignSGD step (fused_encoder.py:67-77 +
 * sign_sgd.py:23): the segmented reduction of sphip_segmented_weight_grad (without dcoeff), but
 * each finished row piece g updates the parameter in place, param -= lr * sign(g) (and the
 * bf16 shadow, may be NULL), instead of being stored: the (M, D) gradient is never written nor
 * read back by an optimizer pass. Same arithmetic as the two separate calls, bit for bit. Only
 * valid where the gradient has no other consumer (one rank, no accumulation). Workspace as for
 * sphip_segmented_weight_grad. sphip_segment_coeff_sums gives the bias gradient (sum of each
 * latent's coefficients, fused_encoder.py:80-84) that the unfused kernel returns as gbias.
 * coeff_sums (M) + bdec_partials (ceil(M / 32), D), both or neither: with them the kernel also
 * leaves, per block of 32 latents, sum_j coeff_sums[j] * param_old[j, :] -- taken from the parameter
 * pieces the update reads anyway; the column sums of bdec_partials are gb^T W_enc, the
 * encoder-side term of b_dec's gradient (sparse_coder.py:192: sae_in = x - b_dec, through
 * fused_encoder.py:56-65), which otherwise costs a pass of its own over the matrix (sphip_vec_mat)
 * that must run before this call. */
int sphip_segmented_weight_grad_signsgd(const uint32_t* perm, const int32_t* seg_begin,
                                        const void* coeff, int coeff_dtype, const void* rows,
                                        int rows_dtype, const float* row_shift, int64_t N, int64_t k,
                                        int64_t M, int64_t D, float* param, float lr,
                                        void* bf16_shadow, const float* coeff_sums,
                                        float* bdec_partials, void* workspace,
                                        size_t workspace_bytes, void* stream);
int sphip_segment_coeff_sums(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                             int coeff_dtype, int64_t M, float* out, void* stream);
/* The same fusion for the reference's default optimizer "signum" (schedule-free SignSGD,
 * trainer.py:146-162; see sphip_schedulefree_signsgd_step for the update and its parity status):
 * each finished gradient row piece drives the schedule-free step of (param = y, z) in place.
 * Bit for bit sphip_segmented_weight_grad followed by sphip_schedulefree_signsgd_step. */
int sphip_segmented_weight_grad_schedulefree(const uint32_t* perm, const int32_t* seg_begin,
                                             const void* coeff, int coeff_dtype, const void* rows,
                                             int rows_dtype, const float* row_shift, int64_t N,
                                             int64_t k, int64_t M, int64_t D, float* param, float* z,
                                             float lr, float beta, float c, void* bf16_shadow,
                                             const float* coeff_sums, float* bdec_partials,
                                             void* workspace, size_t workspace_bytes, void* stream);

/* Sign exchange for data-parallel SignSGD (DDP gradient exchange, trainer.py:457-468, done as
 * reduce-scatter of the fp32 gradients + all-gather of 1-byte signs instead of an all-reduce:
 * SignSGD.step, sign_sgd.py:23, only needs sign(mean gradient); 5/8 of the bytes on the wire).
 * Sign codes are int8 -1 / 0 / +1.
 *   sphip_sign_i8:                 out = sign(g)                                (n elements)
 *   sphip_decoder_project_sign_i8: out[j,:] = sign(G[j,:] - <G[j,:],W[j,:]> W[j,:]) for the M
 *                                  rows given (a rank's row block of W_dec and of its reduced
 *                                  gradient; sparse_coder.py:281-295 + sign)
 *   sphip_apply_signs:             p -= lr * signs (optionally refreshing a bf16 shadow of p)
 *   sphip_decoder_apply_signs:     the same per decoder row, followed by the row
 *                                  normalisation when renorm != 0 (sparse_coder.py:273-279);
 *                                  slice_major: NULL, or the [D/32][sm_rows][32] copy of the
 *                                  whole matrix (sphip_slice_major_f32), of which W holds
 *                                  rows [sm_row0, sm_row0 + M): the finished rows are written
 *                                  there too (fused form only: renorm, D % 32 == 0) */
int sphip_sign_i8(const float* g, int64_t n, int8_t* out, void* stream);
int sphip_decoder_project_sign_i8(const float* W, const float* G, int64_t M, int64_t D,
                                  int8_t* out, void* stream);
int sphip_apply_signs(float* p, const int8_t* signs, int64_t n, float lr, void* bf16_shadow,
                      void* stream);
int sphip_decoder_apply_signs(float* W, const int8_t* signs, int64_t M, int64_t D, float lr,
                              int renorm, float* slice_major, int64_t sm_rows, int64_t sm_row0,
                              void* stream);
/* The schedule-free step ("signum", see sphip_schedulefree_signsgd_step) with the exchanged
 * signs in place of sign(g): sign(mean gradient), for decoder rows of the projected mean
 * gradient (sphip_decoder_project_sign_i8 on the parameter y). Same arithmetic as the one-GPU
 * kernels, so replicas and the single-GPU step agree bit for bit. */
int sphip_schedulefree_apply_signs(float* p, float* z, const int8_t* signs, int64_t n, float lr,
                                   float beta, float c, void* bf16_shadow, void* stream);
int sphip_decoder_schedulefree_apply_signs(float* W, float* Z, const int8_t* signs, int64_t M,
                                           int64_t D, float lr, float beta, float c, int renorm,
                                           float* slice_major, int64_t sm_rows, int64_t sm_row0,
                                           void* stream);

/* trainer.py:422: did_fire[idx.flatten()] = True.  did_fire (M) uint8. */
int sphip_mark_fired(const int64_t* idx, int64_t n_entries, int64_t M, uint8_t* did_fire,
                     void* stream);

/* trainer.py:531-538: counts += n_tokens; counts[did_fire] = 0; did_fire = 0. counts int64. */
int sphip_dead_counter_step(int64_t* counts, uint8_t* did_fire, int64_t M, int64_t n_tokens,
                            void* stream);

/* trainer.py:415: dead_mask = counts > threshold; also returns the number of dead latents
 * in n_dead[0] (int32, device) so the host syncs once (sparse_coder.py:231). */
int sphip_dead_mask(const int64_t* counts, int64_t M, int64_t threshold, uint8_t* mask,
                    int32_t* n_dead, void* stream);

/* AuxK over few dead latents as dense products (sparse_coder.py:230-251: the decode of the
 * k_aux = d_in/2 best dead latents per token, self.decode(auxk_acts, auxk_indices), and its
 * W_dec gradient). When the dead latents are few, the (tokens x dead latents) matrix of
 * selected activations is dense enough for the matrix cores:
 *     e_hat = A W_dead + b_dec,   W_dec.grad[dead] = A^T dL/de_hat.
 * A is exact in bf16 (the activations are bf16 under autocast); the fp32 factor is split into
 * three bf16 terms whose sum is the fp32 value, and the partial products are added in fp32,
 * so the result differs from the fp32 embedding_bag only in the order of the additions.
 *   sphip_dead_compact:          mask (M) uint8 -> slot_of (M) int32 (slot of a dead latent, else
 *                                -1) and dead_list (slot -> latent, up to list_capacity)
 *   sphip_dead_scatter_acts:     idx (N,k) int64 / acts (N,k) bf16 -> A (N, ldA) and A^T (Kd, ldT)
 *                                bf16, zero-filled first; entries on latents that are not dead
 *                                (or < 0) are skipped
 *   sphip_split_bf16_transposed: fp32 rows src[row_list ? row_list[r] : r] (r < R, C columns) ->
 *                                three (C, ldr) bf16 matrices, element [c][r]; columns R..ldr-1 zero
 *   sphip_gemm_bf16_f32acc:      out (N, ld_out) fp32 (+)= x (N,K) W (M,K)^T, bf16 operands, fp32
 *                                accumulate; accumulate != 0 adds to what out holds. K % 64 == 0
 *   sphip_scatter_rows:          dst (M, D) = 0; dst[row_list[r], :] = src[r, :] for r < R
 *   sphip_scatter_add_rows:      dst[row_list[r], :] += src[r, :] for r < R (distinct rows): the dead
 *                                rows' gradient added into the dense W_dec gradient the main decode's
 *                                backward has produced, instead of a dense (M, D) temporary and an
 *                                (M, D) addition by the autograd engine
 *   sphip_broadcast_rows:        out[n, :] = bias (the `+ b_dec` the products are added to)
 *   sphip_gather_rows_bf16:      dst (rows_out, D) bf16 = src[row_list[r], :] for r < R, zero rows
 *                                after: the dead latents' encoder rows / bias, from which
 *                                sphip_encoder_preacts recomputes their pre-activations as a compact
 *                                (N, rows_out) matrix (same kernel, same values), so that the AuxK
 *                                top-k (sparse_coder.py:240-243) reads n_dead instead of M columns;
 *                                its indices are then slots: pass slot_of = NULL to
 *                                sphip_dead_scatter_acts */
int sphip_dead_compact(const uint8_t* mask, int64_t M, int32_t* slot_of, int64_t* dead_list,
                       int64_t list_capacity, void* stream);
int sphip_dead_scatter_acts(const int64_t* idx, const void* acts_bf16, const int32_t* slot_of,
                            int64_t N, int64_t k, int64_t M, int64_t Kd, int64_t ldA, int64_t ldT,
                            void* A, void* At, void* stream);
int sphip_split_bf16_transposed(const float* src, const int64_t* row_list, int64_t R, int64_t C,
                                int64_t ldr, void* hi, void* mid, void* lo, void* stream);
int sphip_gemm_bf16_f32acc(const void* x, const void* W, float* out, int64_t N, int64_t K,
                           int64_t M, int64_t ld_out, int accumulate, void* stream);
int sphip_scatter_rows(const float* src, const int64_t* row_list, int64_t R, int64_t D, int64_t M,
                       float* dst, void* stream);
int sphip_scatter_add_rows(const float* src, const int64_t* row_list, int64_t R, int64_t D, int64_t M,
                       float* dst, void* stream);
int sphip_broadcast_rows(const float* bias, int64_t N, int64_t D, float* out, void* stream);
int sphip_gather_rows_bf16(const void* src, const int64_t* row_list, int64_t R, int64_t rows_out,
                           int64_t D, void* dst, void* stream);

/* ---------------------------------------------------------------- Muon (muon.py:11-40, 115-181) */

/* The Newton-Schulz products of quintic_newtonschulz (muon.py:34-38: A = X @ X.mT,
 * B = b * A + c * A @ A, X = a * X + B @ X, all bf16 tensors) on the bf16 MFMA GEMM kernels:
 *   out (N, ld_out) bf16 = rnd( rnd(alpha * rnd(x W^T)) + rnd(beta * C) ),  rnd = round to bf16,
 * i.e. exactly the roundings torch applies to `beta * C + alpha * (x @ W.mT)` on bf16 tensors
 * (fp32 accumulation inside the product). x (N, K), W (M, K) bf16 row-major; C (N, ld_c) bf16 or
 * NULL (no second term). K % 8 == 0; large products (K % 64 == 0, M, ld_c, ld_out % 4 == 0) take
 * the 256x256 LDS-DMA kernel of sphip_encoder_preacts. out must not alias x or W; it may alias C
 * only when every tile's C is read before any of its out is written elsewhere, i.e. never rely
 * on it. */
int sphip_gemm_bf16_axpy(const void* x, const void* W, const void* C, void* out, int64_t N, int64_t K,
                         int64_t M, int64_t ld_c, int64_t ld_out, float alpha, float beta, void* stream);

/* Muon.step up to the orthogonalisation, one pass over the gradient (muon.py:157-160, :24, :31):
 *   buf = lerp(buf, g, 1 - beta); u = nesterov ? lerp(g, buf, beta) : buf   (torch.lerp; beta = 0:
 *   u = g, buf untouched and may be NULL); u_bf16 = bf16(u); norm[0] = sum(u_bf16^2) in fp32,
 *   norm[1] = bf16(bf16(sqrt(norm[0])) + 1e-7), the denominator of `X / (X.norm() + 1e-7)` as torch
 *   computes it on a bf16 tensor. workspace: sphip_muon_workspace_bytes() bytes.
 * sphip_muon_denominator recomputes norm[1] from norm[0] (after the ranks of a row-sharded
 * Newton-Schulz have summed their norm[0]); sphip_muon_normalise divides x (n bf16) by norm[1]
 * in place, rounding to bf16. */
size_t sphip_muon_workspace_bytes(void);
int sphip_muon_momentum(const float* g, float* buf, int64_t n, float beta, int nesterov, void* u_bf16,
                        float* norm, void* workspace, size_t workspace_bytes, void* stream);
int sphip_muon_denominator(float* norm, void* stream);
int sphip_muon_normalise(void* x, int64_t n, const float* norm, void* stream);

/* dst (C, ld_dst) = src (R, C)^T for bf16 or fp32 elements; R <= ld_dst < R + 64, the padding
 * columns R..ld_dst-1 are written as zeros (so that a product contracting over R can round R up
 * to the multiple its kernel needs). Used where the reference transposes views for free:
 * muon.py:27-28, 39-40 (the Newton-Schulz iterate; here kept tall, X X^T contracts over its
 * transposed copy) and the backward of the skip term `x @ W_skip.mT` (sparse_coder.py:221-222:
 * W_skip.grad = g^T x contracts over the tokens). */
int sphip_transpose(const void* src, void* dst, int64_t R, int64_t C, int64_t ld_dst, int dtype,
                    void* stream);

/* out = a + b, n elements, a and out fp32, b bf16 or fp32: `sae_out += x @ W_skip.mT`
 * (sparse_coder.py:221-222; under autocast the product is a bf16 tensor and the sum fp32).
 * out may be a. */
int sphip_add(const float* a, const void* b, int b_dtype, float* out, int64_t n, void* stream);

/* muon.py:121-122 + :178-180: p *= decay (= 1 - lr * weight_decay); p -= step * o with
 * step = lr * 0.2 * sqrt(max(p.shape)) and o the bf16 orthogonalised update in p's layout.
 * bf16_shadow (or NULL) is refreshed with bf16(p) in the same pass. */
int sphip_muon_apply(float* p, const void* o_bf16, int64_t n, float decay, float step, void* bf16_shadow,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSIFY_HIP_H */
