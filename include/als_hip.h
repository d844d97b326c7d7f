/*
 * als_hip.h - C ABI of the MI355X (gfx950) implicit-ALS engine, libals_hip.so.
 *
 * The reference (whong92/recommender, package reclibwh) has no native interface: its
 * boundary for this path is a set of Python functions in reclibwh/core/ALS.py.  Every
 * entry point below names the reference function it stands in for.  A host binds these
 * with ctypes; signatures use plain pointers and sizes only.
 *
 * Conventions
 *   - every pointer marked [device] is device memory owned by the caller; the library
 *     never allocates or frees device memory and keeps no pointer after a call returns;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the null
 *     stream); no call synchronises with the host;
 *   - factors are float32 row-major with an explicit leading dimension (elements);
 *     CSR is int64 row pointers, int32 column indices, float32 values;
 *   - return value: ALS_OK or an error code; als_last_error_string() describes the
 *     last error on the calling thread;
 *   - results are bit-for-bit reproducible: no floating-point atomics, every reduction
 *     has a fixed order that depends only on the inputs and on `chunk_nnz`, never on
 *     the launch geometry, on timing, or on how rows are split across devices.
 */
#ifndef ALS_HIP_H
#define ALS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALS_OK 0
#define ALS_ERR_INVALID_ARG 1 /* null pointer, negative size, bad alignment, ...          */
#define ALS_ERR_UNSUPPORTED 2 /* factor count outside the supported range                 */
#define ALS_ERR_WORKSPACE 3   /* workspace smaller than the matching *_workspace_bytes()  */
#define ALS_ERR_HIP 4         /* a HIP runtime call failed (message holds hipGetErrorString) */

#define ALS_MAX_FACTORS 256         /* largest f (Gram, conjugate-gradient half-sweep, scoring) */
#define ALS_MAX_FACTORS_EXACT 128   /* largest f of the exact (Cholesky) half-sweep            */
#define ALS_DEFAULT_CHUNK_NNZ 2048  /* rows longer than this are split (multiple of 64)        */
#define ALS_GRAM_CHUNK_ROWS 256     /* rows per first-level Gram partial                       */
#define ALS_GRAM_SUPER_ROWS 65536   /* rows per second-level Gram partial (256 chunks)         */
#define ALS_GRAM_WIDE_CHUNK_ROWS 4096 /* rows per first-level partial when f > ALS_MAX_FACTORS_EXACT */

/* Description of the last error on this thread ("" if none).  Replaces the Python
 * exceptions / asserts of the reference (reclibwh/core/ALS.py:124,145). */
const char *als_last_error_string(void);

/* Library ABI version (bumped on any signature change). */
int32_t als_abi_version(void);

/* Per-call options of the half-sweeps (no reference counterpart).  NULL, or a zeroed struct, selects
 * every default; nothing here is kept after the call returns and the library holds no settings
 * of its own, so concurrent callers do not affect each other.
 *   accumulate  how  sum_j alpha r_uj y_j y_j^T  is formed for 16 < f <= 128:
 *       ALS_ACCUMULATE_BF16X3            every value is cut exactly into three bf16 pieces and the six
 *           significant piece products run on the bf16 matrix cores, accumulated in float32; error
 *           at the level of float32 rounding (the dropped products are < 2^-23 of each term);
 *       ALS_ACCUMULATE_FP16X2            every value, scaled by a power of two chosen per call, is cut into two fp16
 *           pieces (22 mantissa bits) and the three significant piece products run on the fp16 matrix
 *           cores: half the matrix work of BF16X3, error 2^-21 per term.  The scale comes from bounds:
 *           max_rating (an upper bound of the CSR's ratings, REQUIRED for this mode: a larger rating overflows
 *           fp16 and the row's factors become NaN) and the largest diagonal entry of Y^T Y, which als_gram_f32
 *           leaves behind the tile-order Gram.
 *       ALS_ACCUMULATE_FP32              float32 matrix-core fma chain in CSR order (v_mfma_f32_16x16x4_f32).
 *       ALS_ACCUMULATE_DEFAULT           FP16X2 if max_rating > 0 was given and the fixed factor has at least 16 rows
 *           per factor (n_opp >= 16 f: with fewer the Gram is close to singular and the larger product error of
 *           FP16X2 shows in the solution), else BF16X3.
 *       All are deterministic and independent of the launch geometry.  f <= 16 always uses FP32.
 *       (The Gram of als_gram_f32 takes no options: bf16 x 3 for 32 < f <= 128, float32 otherwise.)
 *   row_kernel  which per-row kernel handles 64 < f <= 128 (both return the same bits):
 *       ALS_ROW_KERNEL_WAVE       one wavefront owns a row's whole matrix (144 accumulator registers);
 *       ALS_ROW_KERNEL_STREAM     one resident wavefront per SIMD walks its share of the rows as one stream of
 *                                 32-entry blocks: factor rows come by LDS-DMA several blocks ahead, across row
 *                                 boundaries, and a block's matrix products overlap the next block's operand
 *                                 split (bf16 x 3 accumulate, 16-byte aligned factor rows; long rows' chunk
 *                                 items go the same way once every resident wavefront gets several, else
 *                                 they run on the WAVE kernel - the partial sums are the same bits);
 *       ALS_ROW_KERNEL_AUTO       the library's choice.
 *   tap_begin / tap_end  two hipEvent_t handles, or both NULL: recorded on `stream` immediately before
 *       the first and after the last per-row kernel of the call (benchmarks time the kernels with
 *       them; the final |delta| reduction lies outside).
 *   whitened / ld_whitened / short_bounds  all NULL / 0: every row is solved as an f x f system.  Given (exact
 *       half-sweep, 32 < f <= 128, f % 4 == 0): rows with at most min(f, 128) (f <= 64: 64) stored entries are
 *       solved in their entry-space form, see "Short rows" below.
 *   reserve_cus  0: the STREAM row kernel keeps one wavefront resident on every SIMD of the device (a workgroup of four
 *       per compute unit), each holding nearly all of its SIMD's vector registers - until it ends no kernel of another
 *       stream can start.  n > 0: it occupies n compute units fewer and leaves those wholly free, so that such a kernel
 *       - the RCCL all-gather of the previous row chunk in a multi-GPU run - can run beside it.  Measured on MI355X: a
 *       kernel with workgroups of several wavefronts is dispatched only when every shader engine has a free unit, so
 *       n below the number of shader engines (32 = compute units / 8) changes nothing.  The rows are handed out while
 *       the kernel runs and solved independently: the results are the same bits for every n. */
#define ALS_ACCUMULATE_DEFAULT 0
#define ALS_ACCUMULATE_BF16X3 1
#define ALS_ACCUMULATE_FP32 2
#define ALS_ACCUMULATE_FP16X2 3
#define ALS_ROW_KERNEL_AUTO 0
#define ALS_ROW_KERNEL_WAVE 1
#define ALS_ROW_KERNEL_STREAM 2
#define ALS_SHORT_MAX_GROUPS 8      /* short rows: at most 16 * 8 = 128 entries */
typedef struct {
    int32_t accumulate;
    int32_t row_kernel;
    void *tap_begin;
    void *tap_end;
    float max_rating;            /* an upper bound of the CSR's ratings, 0 = unknown (see accumulate)             */
    const float *whitened;       /* [device] table of als_whiten_f32 for Yopp, or NULL (see "Short rows" below)  */
    int64_t ld_whitened;         /* its leading dimension (>= 2 f, a multiple of 4)                              */
    const int64_t *short_bounds; /* [host] the ALS_SHORT_MAX_GROUPS + 2 entries of als_plan_short_bounds         */
    int32_t reserve_cus;         /* compute units the STREAM row kernel leaves free, 0 = none                     */
} als_sweep_opts_t;

/* ------------------------------------------------------------------------------------
 * Row schedule ("plan") - host-side, CPU only, built once per CSR matrix.
 *
 * Replaces the batching done by SparseMatRowIterator / get_pos_ratings_padded
 * (reclibwh/data/iterators.py:160-193, reclibwh/utils/utils.py:142-173): instead of
 * padded [B, Lmax] batches the device walks raw CSR rows.  A row with more than
 * chunk_nnz stored entries is split into ceil(nnz/chunk_nnz) chunk items whose partial
 * sums are combined in chunk order; items are sorted longest first.  An item is 32 bytes and
 * carries its own entry range, so a plan belongs to the row pointer it was built from: the
 * sweeps trust it (an item whose range lies outside col/val is an out-of-bounds read).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    int32_t row;    /* row of the CSR handed to als_half_sweep_csr_f32                      */
    int32_t chunk;  /* -1: whole row, solved in place; >= 0: chunk index within a long row  */
    int32_t slot;   /* partial-sum slot for a chunk item, -1 otherwise                      */
    int32_t count;  /* stored entries of this item                                          */
    int64_t begin;  /* index of its first entry in col/val (rowptr[row] + chunk*chunk_nnz): */
                    /* the kernels start from the item alone, one dependent load less       */
    int64_t reserved;
} als_item_t;

typedef struct {
    int32_t row;
    int32_t first_slot; /* its chunks own slots first_slot .. first_slot + n_chunks - 1 */
    int32_t n_chunks;
    int32_t reserved;
} als_long_row_t;

/* Sizes of the plan for rows 0..n_rows-1 of rowptr_host (host memory, n_rows+1 entries). */
int32_t als_plan_count(const int64_t *rowptr_host, int64_t n_rows, int32_t chunk_nnz, int64_t *n_items,
                       int64_t *n_long, int64_t *n_slots);
/* Fills items_host[n_items] and long_host[n_long] (host memory); upload them afterwards. */
int32_t als_plan_fill(const int64_t *rowptr_host, int64_t n_rows, int32_t chunk_nnz, als_item_t *items_host,
                      als_long_row_t *long_host);

/* ------------------------------------------------------------------------------------
 * Gram cache - replaces als_update_cache (reclibwh/core/ALS.py:235-246):
 *     Y2 <- Y^T Y over all n rows of the layer;  LambdaI <- lamb * I.
 *
 * gram_out   [device] f*f floats, row major: Y^T Y (no lambda), exactly symmetric.
 * tiles_out  [device] als_gram_tiles_bytes(f) bytes: Y^T Y + lamb*I in the register-tile
 *            order the half-sweep kernel consumes, followed (f <= 128) by sqrt(max_k (Y^T Y)_kk), the bound of
 *            |y| the fp16 x 2 accumulate takes its operand scale from (opaque to the caller).
 * The sum over rows is a fixed three-level tree (256-row chunks, 8 runs of 32 chunks per 65 536 rows, then
 * the groups in order), so a Gram assembled from als_gram_partials_f32 calls on several
 * devices equals the single-device one bit for bit.
 * f > ALS_MAX_FACTORS_EXACT: the same tree over 4096-row chunks of 64 x 64 blocks; tiles_out
 * then holds the summed partial (no lamb; only the conjugate-gradient half-sweep exists for
 * such f and it takes gram_out and lamb).
 * ------------------------------------------------------------------------------------ */
size_t als_gram_tiles_bytes(int32_t f);
size_t als_gram_partial_bytes(int32_t f);      /* bytes of one second-level partial (als_gram_partials_f32) */
size_t als_gram_workspace_bytes(int64_t n, int32_t f);
int32_t als_gram_f32(const float *Y, int64_t n, int32_t f, int64_t ldy, float lamb, float *gram_out,
                     float *tiles_out, void *workspace, size_t ws_bytes, void *stream);

/* Second-level partials for rows [row_begin, row_end) of Y; row_begin must be a multiple
 * of ALS_GRAM_SUPER_ROWS.  Writes ceil((row_end-row_begin)/ALS_GRAM_SUPER_ROWS) partials
 * of als_gram_partial_bytes(f) bytes each to super_out [device].  Workspace as for
 * als_gram_workspace_bytes(row_end - row_begin, f).  (Multi-device Gram, no reference
 * counterpart: the reference is single device.) */
int32_t als_gram_partials_f32(const float *Y, int64_t row_begin, int64_t row_end, int32_t f, int64_t ldy,
                              float *super_out, void *workspace, size_t ws_bytes, void *stream);
/* Sums n_super second-level partials in order and emits gram_out / tiles_out as above. */
int32_t als_gram_finalize_f32(const float *super_in, int64_t n_super, int32_t f, float lamb, float *gram_out,
                              float *tiles_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Half-sweep - replaces run_als_epoch / run_als_step / als_run_regression
 * (reclibwh/core/ALS.py:160-233) and ALSRef._run_single_step (ALS.py:67-88):
 * for every row u of the CSR
 *     A_u = Y^T Y + sum_j alpha r_uj y_j y_j^T + lamb I,   b_u = sum_j (1 + alpha r_uj) y_j
 *     x_u = A_u^{-1} b_u            (Cholesky; the reference forms inv(A_u), ALS.py:82,172)
 * X is updated in place (ALS.py:196-197) and *absdiff_out receives
 * sum_u sum_k |x_u,k(old) - x_u,k(new)| (ALS.py:195), accumulated in double.
 * Stored zeros contribute nothing (ALS.py:79); ratings must be >= 0 (the default accumulate mode
 * takes sqrt(alpha r); a negative rating gives NaN factors for that row).
 * A row without entries gets x_u = 0.
 *
 * rowptr/col/val [device] CSR with n_rows rows; column indices must lie in [0, n_opp).
 * row_ids        [device] optional: row u of the CSR updates X row row_ids[u] (fold-in,
 *                ALS.py:387-395); NULL = identity.
 * items/long_rows[device] the uploaded plan for this CSR (same chunk_nnz).
 * Yopp           [device] the fixed factor, n_opp rows, leading dimension ldy.
 * gram_tiles     [device] tiles_out of als_gram_f32 for Yopp (carries lamb).
 * Xown           [device] the factor being solved for, leading dimension ldx.
 * absdiff_out    [device] one double, or NULL.
 * info_out       [device] one int32, or NULL: incremented once per row whose system
 *                was not positive definite in float32 (that row's x is then NaN).
 * opts           [host] per-call options, or NULL for the defaults.
 * ------------------------------------------------------------------------------------ */
size_t als_half_sweep_workspace_bytes(int64_t n_rows, int64_t n_slots, int32_t f);
int32_t als_half_sweep_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                               const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                               const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots,
                               int32_t chunk_nnz, const float *Yopp, int64_t n_opp, int64_t ldy,
                               const float *gram_tiles, int32_t f, float alpha, float *Xown, int64_t ldx,
                               double *absdiff_out, int32_t *info_out, const als_sweep_opts_t *opts,
                               void *workspace, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Short rows - the same exact solve of als_run_regression (reclibwh/core/ALS.py:160-176; ALSRef, ALS.py:82-86)
 * for rows with fewer stored entries than factors, in the algebraically equal "entry space" form.  With
 * G = Y^T Y + lamb I = R^T R (Cholesky) and L entries (y_e, r_e), w_e = alpha r_e:
 *     x = G^-1 sum_e sqrt(w_e) t_e y_e,   (I + V V^T) t = (1 + w) / sqrt(w),   V_e = sqrt(w_e) y_e R^-1
 * - an L x L symmetric positive definite system (every eigenvalue >= 1) instead of the f x f one.  No reference
 * counterpart as code: the reference inverts the f x f matrix for every row whatever its length; results agree with
 * it to float32 rounding (the form is the better conditioned of the two).
 *
 * als_whiten_f32 prepares, once per half-sweep, the table WZ [n, ldw] of the fixed factor:
 * columns [0, f) = Y R^-1, columns [f, 2 f) = Y G^-1 (R and the inverses in float64 from `gram`, the float32
 * gram_out of als_gram_f32, plus lamb; then one GEMM on the matrix cores, float32 accuracy).  32 < f <= 128,
 * f % 4 == 0, ldy and ldw multiples of 4, 16-byte aligned pointers, ldw >= 2 f.  info_out (optional) is incremented
 * if gram + lamb I is not positive definite.  Workspace: als_whiten_workspace_bytes(f).
 *
 * als_plan_short_bounds (host) returns, for the plan als_plan_fill builds from the same arguments,
 * bounds_out[j], j = 0..ALS_SHORT_MAX_GROUPS: the index in the item list of the first whole-row item with at most
 * 16 j stored entries (whole rows are listed by entry count, descending), and bounds_out[ALS_SHORT_MAX_GROUPS + 1] =
 * n_items.  als_short_rows_max_groups(f): the largest j the kernels use for this f (0: no short-row form).
 * Pass table, leading dimension and bounds in als_sweep_opts_t to als_half_sweep_csr_f32 (or its _static_ variant):
 * rows of at most 16 * max_groups entries then take this form, all others the f x f form; a row's result depends on
 * the row and the inputs only, never on the launch shape, so sweeps stay reproducible and independent of sharding.
 * ------------------------------------------------------------------------------------ */
int32_t als_short_rows_max_groups(int32_t f);
int32_t als_plan_short_bounds(const int64_t *rowptr_host, int64_t n_rows, int32_t chunk_nnz, int64_t *bounds_out);
size_t als_whiten_workspace_bytes(int32_t f);
int32_t als_whiten_f32(const float *gram, float lamb, int32_t f, const float *Y, int64_t n, int64_t ldy, float *WZ,
                       int64_t ldw, int32_t *info_out, void *workspace, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Half-sweep for callers that cannot allocate device memory.  No counterpart in the reference: it is
 * for code under HIP graph capture, tight serving loops (the fold-in of ALSUpdateAlgo.update_user is one)
 * and harnesses that forbid hipMalloc.  Same computation, arguments and results (bit for bit) as
 * als_half_sweep_csr_f32, but without the workspace: the scratch memory lives in a buffer of
 * ALS_STATIC_SCRATCH_BYTES built into the library's code object, one copy per GPU, held until the process
 * ends.  The call allocates and frees nothing.  It returns ALS_ERR_WORKSPACE, with the sizes in the message,
 * when als_half_sweep_workspace_bytes(n_rows, n_slots, f) exceeds ALS_STATIC_SCRATCH_BYTES (16 MiB: e.g.
 * 65 536 rows with up to 400 chunk slots at f = 128, or 1 400 at f = 64).
 * `stream` must belong to the device that is current on the calling thread.  All calls on one GPU share that
 * GPU's buffer: calls that can overlap on one GPU (different streams, or different threads) race; calls on
 * one stream are ordered and safe, and no call depends on what an earlier one left in the buffer.
 * als_static_scratch_init() resolves the buffer's address on the current device ahead of time (otherwise
 * the first call on a device does it); als_static_scratch_bytes() returns the limit.
 * ------------------------------------------------------------------------------------ */
#define ALS_STATIC_SCRATCH_BYTES (16u << 20)
size_t als_static_scratch_bytes(void);
int32_t als_static_scratch_init(void);
int32_t als_half_sweep_csr_static_f32(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                                      const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                                      const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots,
                                      int32_t chunk_nnz, const float *Yopp, int64_t n_opp, int64_t ldy,
                                      const float *gram_tiles, int32_t f, float alpha, float *Xown, int64_t ldx,
                                      double *absdiff_out, int32_t *info_out, const als_sweep_opts_t *opts,
                                      void *stream);

/* ------------------------------------------------------------------------------------
 * Conjugate-gradient half-sweep, 1 <= f <= ALS_MAX_FACTORS.  Same normal equations and same
 * in-place update of Xown as als_half_sweep_csr_f32, but each row's system is advanced by
 * exactly `cg_steps` conjugate-gradient steps from the row's current value instead of being
 * solved (A_u is never formed: A_u v = gram v + sum_j alpha r_uj (y_j . v) y_j + lamb v).
 * No counterpart in the reference, which inverts A_u (reclibwh/core/ALS.py:82,172); it is the
 * solver for factor counts whose f x f system no longer fits one wavefront's registers.
 * With cg_steps >= f it reproduces the exact solve to rounding.  A row without entries, or with
 * nothing but stored zeros (which are dropped, ALS.py:79), gets its exact solution 0.
 *
 * Implementations of the same iteration, chosen by `form`:
 *   ALS_CG_FORM_MATRIX     ONE gather pass forms the row's matrix in accumulator registers exactly as
 *                          the exact half-sweep does, and the CG steps run on it in place
 *                          (matrix-vector products on the register tiles).  f <= 128: one wave per
 *                          row.  128 < f <= 256: the tiles are spread over the 4 waves of a
 *                          workgroup and only f-vectors cross waves (through LDS); this costs
 *                          O(nnz f^2) per row and pays only for rows much longer than f.
 *                          Needs gram_tiles.
 *   ALS_CG_FORM_LOCK_STEP  all rows advance in lock step, cg_steps + 1 gather passes, the Gram part
 *                          of every step as one GEMM over all rows: O(cg_steps (nnz f + f^2)) per row.
 *   ALS_CG_FORM_AUTO       MATRIX for f <= 128 when gram_tiles is given, else LOCK_STEP.
 *
 * gram       [device] f*f floats, row major: gram_out of als_gram_f32 for Yopp (no lamb).
 * gram_tiles [device] tiles_out of the same call (carries lamb), or NULL.
 * info_out   [device] incremented once per row (MATRIX) or per (row, step) (LOCK_STEP) that met
 *            p^T A p <= 0 (not positive definite).
 * Other arguments as for als_half_sweep_csr_f32.
 * ------------------------------------------------------------------------------------ */
#define ALS_CG_FORM_AUTO 0
#define ALS_CG_FORM_LOCK_STEP 1
#define ALS_CG_FORM_MATRIX 2
size_t als_half_sweep_cg_workspace_bytes(int64_t n_rows, int64_t n_slots, int32_t f);
int32_t als_half_sweep_cg_f32(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                              const int64_t *row_ids, const als_item_t *items, int64_t n_items,
                              const als_long_row_t *long_rows, int64_t n_long, int64_t n_slots,
                              int32_t chunk_nnz, const float *Yopp, int64_t n_opp, int64_t ldy, const float *gram,
                              const float *gram_tiles, float lamb, int32_t f, float alpha, int32_t cg_steps,
                              int32_t form, float *Xown, int64_t ldx, double *absdiff_out, int32_t *info_out,
                              const als_sweep_opts_t *opts, void *workspace, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Scoring - replaces ALSTrainer.predict + SimpleMFRecAlgo.recommend
 * (reclibwh/core/ALS.py:359-362, reclibwh/core/RecAlgos.py:20-32) and
 * cosine_similarity_tf / SimpleMFSimAlgo.similar (RecAlgos.py:38-73).
 *
 * als_scores_f32        scores[u][i] = <X[user_ids[u]], Y[i]> for u < nu, i < M (user_ids NULL:
 *                       rows 0..nu-1 of X).  scores is [nu, lds] row major, lds >= M.
 *                       f <= 128 with 16-byte aligned rows (f, ldx, ldy multiples of 4): every
 *                       operand is cut exactly into three bf16 pieces and the 6 largest piece
 *                       products are accumulated in float32 on the matrix cores (error per
 *                       product at float32 rounding); otherwise float32 products.  Either way a
 *                       score depends only on its two rows, never on the launch shape.
 * als_topk_rows_f32     per row of `scores`, the k largest entries in descending order, ties by
 *                       ascending column (the reference does a full numpy argsort of every row,
 *                       RecAlgos.py:32; its consumers keep the first 200).  If ex_rowptr/ex_col
 *                       (a CSR; row ex_row_ids[r] for score row r, or row r if ex_row_ids is NULL) are
 *                       given, those columns of each row are set to
 *                       -inf first, in place (the "drop rated items" filter of
 *                       reclibwh/serving/RecommenderService.py:16-29).  1 <= k <= min(1024, M).
 * als_predict_pairs_f32 out[p] = <X[u[p]], Y[i[p]]> (the ALS Keras model's output, model
 *                       template ALS.json.template:38-43).
 * als_normalize_rows_f32 out[r] = E[r] / ||E[r]||_2, rows of zero norm become 0
 *                       (RecAlgos.py:45-50); out may alias E.
 * All pointers are [device]; all work is enqueued on `stream`.
 * ------------------------------------------------------------------------------------ */
int32_t als_scores_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y, int64_t ldy,
                       int64_t M, int32_t f, float *scores, int64_t lds, void *stream);
int32_t als_topk_rows_f32(float *scores, int64_t lds, int64_t n_rows, int64_t M, const int64_t *ex_rowptr,
                          const int32_t *ex_col, const int64_t *ex_row_ids, int32_t k, int32_t *idx_out, float *score_out,
                          void *stream);
/* Fused top-k - the whole of SimpleMFRecAlgo.recommend (RecAlgos.py:20-32) followed by the
 * "drop rated items, keep the best" filter of RecommenderService.py:16-29, without materialising
 * the [nu, M] score matrix: the first S items (S = max(16 384, 16 k)) are scored and selected
 * exactly as by als_scores_f32 + als_topk_rows_f32; their k-th best score is the user's
 * threshold; the remaining items are scored tile by tile on the matrix cores, in ranges
 * [S, 4S), [4S, 16S), ..., and only scores above the threshold leave the registers (to a candidate
 * list of 8192 slots per user); after each range the candidates are ordered (score descending,
 * ties by ascending item id) and cut back to k, which tightens the threshold for the next range.
 * Same result as the two-call path.
 * ex_rowptr/ex_col: optional CSR (sorted columns) of items to exclude per user.  ex_by_user_id == 0: it has nu
 * rows, row u for output row u.  ex_by_user_id != 0: it is indexed by user id - output row u reads row
 * user_ids[u] - so the training matrix's own rowptr/col serve as they lie on the device, for any block of users
 * (the "drop rated items" filter of RecommenderService.py:16-29 with nothing built per request).
 * overflow_out [device] one int32: number of candidates that did not fit a user's list.  If it is not 0, the
 * users it happened to (and only they) have item -1 in every slot of their idx_out row and the caller must use the
 * two-call path for them; every other row is valid.
 * 1 <= k <= min(1024, M); nu <= 65535*64. */
size_t als_recommend_topk_workspace_bytes(int64_t nu, int64_t M, int32_t k);
int32_t als_recommend_topk_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const float *Y,
                               int64_t ldy, int64_t M, int32_t f, int32_t k, const int64_t *ex_rowptr,
                               const int32_t *ex_col, int32_t ex_by_user_id, int32_t *idx_out, float *score_out,
                               int32_t *overflow_out, void *workspace, size_t ws_bytes, void *stream);
/* Prepared item tables - serving ranks many blocks of users against ONE fixed item factor (the trained Y that
 * SimpleMFRecAlgo.recommend multiplies every request against, RecAlgos.py:20-32).  als_items_prepare_f32 converts the
 * first M rows of Y once into the operand form of the fp16 matrix cores: every value scaled by one power of two
 * (taken from the largest finite |y|) and cut into two fp16 pieces (22 mantissa bits for every value within 2^18 of the
 * largest), stored in the order the score kernel's lanes read them.  The table has als_items_prepared_bytes(M, f)
 * bytes (the size of the float32 factor, padded to 32 items, + 256), is opaque, belongs to the library version that
 * wrote it, and must be rebuilt when Y changes.  f <= 128, f % 4 == 0, 16-byte aligned rows (als_items_prepared_bytes
 * returns 0 for an unsupported f); the item factors must be finite.
 * als_scores_prepared_f32 / als_recommend_topk_prepared_f32 are als_scores_f32 / als_recommend_topk_f32 against such
 * a table (M_table: the M it was built with; M <= M_table: the items scored): user rows are scaled per row and cut the
 * same way, three piece products per block are accumulated in float32, the scaling is undone exactly.  A score
 * depends on its user row, its item row and the table's scale only; error <= ~2^-21 sum |x_k y_k| (als_scores_f32:
 * 2^-23).  The two calls agree with each other bit for bit, as als_scores_f32 and als_recommend_topk_f32 do; the
 * two-call path for users marked by overflow_out is als_scores_prepared_f32 + als_topk_rows_f32.
 * Workspace of the top-k call: als_recommend_topk_workspace_bytes(nu, M, k). */
size_t als_items_prepared_bytes(int64_t M, int32_t f);
int32_t als_items_prepare_f32(const float *Y, int64_t ldy, int64_t M, int32_t f, void *prepared, size_t bytes,
                              void *stream);
int32_t als_scores_prepared_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu, const void *prepared,
                                int64_t M_table, int64_t M, int32_t f, float *scores, int64_t lds, void *stream);
int32_t als_recommend_topk_prepared_f32(const float *X, int64_t ldx, const int64_t *user_ids, int64_t nu,
                                        const void *prepared, int64_t M, int32_t f, int32_t k, const int64_t *ex_rowptr,
                                        const int32_t *ex_col, int32_t ex_by_user_id, int32_t *idx_out, float *score_out,
                                        int32_t *overflow_out, void *workspace, size_t ws_bytes, void *stream);
int32_t als_predict_pairs_f32(const float *X, int64_t ldx, const float *Y, int64_t ldy, const int64_t *u,
                              const int64_t *i, int64_t n, int32_t f, float *out, void *stream);
int32_t als_normalize_rows_f32(const float *E, int64_t lde, int64_t n, int32_t f, float *out, int64_t ldo,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ALS_HIP_H */
