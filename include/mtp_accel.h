/*
 * mtp_accel.h -- C-ABI of the MI355X (gfx950) implementation of PCCSNet's
 * modality-clustering / best-of-K hot path.
 *
 * The reference (ApeironY/Modular-Trajectory-Prediction-Toolkit) is pure Python and has no
 * FFI of its own: its boundary for this path is a set of module-level Python functions
 * (PCCSNet/utils/utility.py, PCCSNet/utils/metrics.py, PCCSNet/train_steps.py:202-239).
 * The entry points below are what a ctypes binding placed behind those functions binds;
 * each comment cites the reference code the entry point replaces.  INTEGRATION.md shows the
 * binding; modular-trajectory-prediction-toolkit_amd/pccs_accel is the host-side mirror.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every array pointer is a DEVICE pointer owned by
 *    the caller (e.g. torch tensor .data_ptr()), row-major, 16-byte aligned.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *    enqueue work unless stated otherwise; they never allocate or free device memory.
 *  - Return value: 0 = OK, negative = error (MTP_E_*); mtp_last_error() then returns a
 *    thread-local message.  No exceptions cross the boundary.
 *  - float arrays are IEEE binary32, labels are int32, counters/sums are binary64.
 */
#ifndef MTP_ACCEL_H
#define MTP_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTP_OK 0
#define MTP_E_ARG (-1)      /* bad argument (shape, alignment, null pointer) */
#define MTP_E_HIP (-2)      /* a HIP runtime call failed */
#define MTP_E_WORKSPACE (-3) /* workspace too small */
#define MTP_E_COMM (-4)      /* an RCCL call failed, or librccl.so is not available */

/* Lloyd loop states (flags[MTP_FLAG_STATE]) */
#define MTP_STATE_RUNNING 0
#define MTP_STATE_STRICT 1   /* labels == labels_old: sklearn "strict convergence" */
#define MTP_STATE_TOL 2      /* sum_j shift_j^2 <= tol: needs one more E-step */
#define MTP_STATE_RELOCATE 3 /* an empty cluster needs relocation before the M-step */
#define MTP_STATE_MAXITER 4  /* max_iter reached: needs one more E-step */

/* int32 flag words shared by the Lloyd step entry points */
#define MTP_FLAG_STATE 0
#define MTP_FLAG_N_ITER 1
#define MTP_FLAG_N_EMPTY 2
#define MTP_FLAG_PEER_FAILED 7 /* mtp_lloyd_sharded_f32: a rank reported a failed far-point search */
#define MTP_FLAG_WORDS 8

const char *mtp_last_error(void);
int mtp_version(void);

/* Number of floats of the staged centroid image the E-step kernel streams through LDS
 * (MFMA operand order, 32-centroid tiles, ||c||^2 appended per tile).  0 if (k, d) uses the
 * generic E-step kernel, which reads the plain [k, d] matrix. */
int64_t mtp_staged_centers_floats(int k, int d);

/* Bytes of scratch mtp_assign_f32 / mtp_lloyd_f32 need for (k, d). */
int64_t mtp_assign_workspace_bytes(int k, int d);
int64_t mtp_lloyd_workspace_bytes(int k, int d);

/* ||A_i||^2 per row, in NumPy einsum("ij,ij->i") float order.
 * Replaces sklearn.utils.extmath.row_norms(centers, squared=True), sklearn
 * cluster/_k_means_lloyd.pyx:99. */
int mtp_row_sqnorm_f32(const float *A, int64_t n, int d, float *out, void *stream);

/* Column sums (double, += into col_sum[d]) and sums of squared deviations from `mean`
 * (double, += into col_sqdev[d]); X -= mean in place.
 * Replace X.mean(axis=0) / np.var(X, axis=0) / X -= X_mean of sklearn
 * cluster/_kmeans.py:279-287 and :1478-1481 (KMeans.fit preprocessing). */
int mtp_col_sums_f32(const float *X, int64_t n, int d, double *col_sum, void *stream);
int mtp_col_sqdev_f32(const float *X, int64_t n, int d, const double *mean, double *col_sqdev, void *stream);
int mtp_sub_rowvec_f32(float *X, int64_t n, int d, const float *vec, void *stream);

/* The same column statistics in NumPy's own float order: out[c] = the float32 sum over the rows, added one
 * row after the other (what np.add.reduce(X, axis=0) does on a C-ordered float32 array), of X[i][c] when
 * center is NULL, else of (X[i][c] - center[c])^2 with the difference and the square rounded to float
 * (numpy/_core/_methods.py _var: x = arr - arrmean; x = x * x).  With these the host forms
 * X.mean(axis=0) and np.mean(np.var(X, axis=0)) * tol exactly as sklearn cluster/_kmeans.py:279-287 and
 * :1478-1481 do, bit for bit, so that the centred points and the tolerance are the reference's.
 * out is overwritten ([d] floats). */
int mtp_col_seqsum_f32(const float *X, int64_t n, int d, const float *center, float *out, void *stream);
/* The same with the chains started from init[d] instead of zero (init may equal out): rows [lo, lo + n) of a
 * row-sharded array continue the sums of the rows before them, rank after rank, so that the sharded fit
 * (pccs_accel.dist.kmeans_fit_sharded) centres the points with NumPy's mean too. */
int mtp_col_seqsum_carry_f32(const float *X, int64_t n, int d, const float *center, const float *init, float *out,
                             void *stream);

/* E-step: labels[i] = argmin_j (||c_j||^2 - 2 x_i.c_j), strict '<', lowest index wins.
 * Replaces sklearn cluster/_k_means_lloyd.pyx:168-213 (_update_chunk_dense, E part) -- the
 * "nearest-modality assignment" of PCCSNet/train_steps.py:357.
 *   c2        [k] ||c_j||^2, or NULL to compute it (einsum order) from C
 *   labels    [n] in/out: previous labels in, new labels out
 *   n_changed NULL, or one double that is incremented by #(labels[i] changed)
 *   ws        mtp_assign_workspace_bytes(k, d) bytes of device scratch */
int mtp_assign_f32(const float *X, int64_t n, int d, const float *C, int k, const float *c2, int32_t *labels,
                   double *n_changed, void *ws, size_t ws_bytes, void *stream);

/* mtp_assign_f32 for callers that cannot allocate device memory or bring a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same result as mtp_assign_f32(..., c2 = NULL, ...): the staged centroid image and
 * ||c||^2 are written into a buffer of MTP_NOALLOC_SCRATCH_BYTES built into the library (one copy per
 * GPU, held until the process ends); (k, d) with mtp_assign_workspace_bytes(k, d) above that limit
 * are refused with MTP_E_WORKSPACE (k = 1024, d = 128 needs 549 KB).  Calls nothing that allocates.
 * All calls on one GPU share that GPU's buffer: calls that overlap on one GPU (two streams or
 * threads) race; serialise them.  Nothing has to be cleared between calls: each call rewrites
 * every byte it reads. */
#define MTP_NOALLOC_SCRATCH_BYTES (1 << 20)
int mtp_assign_noalloc_f32(const float *X, int64_t n, int d, const float *C, int k, int32_t *labels,
                           double *n_changed, void *stream);

/* M-step sums: sums[k, d] += sum of member rows, counts[k] += member count (doubles).
 * Replaces sklearn cluster/_k_means_lloyd.pyx:215-218 + :143-152 and the per-cluster
 * gather of PCCSNet/utils/utility.py:98-104 (get_modality). */
int mtp_centroid_sums_f32(const float *X, int64_t n, int d, const int32_t *labels, int k, double *sums,
                          double *counts, void *stream);

/* means[j, :] = sums[j, :] / counts[j] rounded once to float (rows with count 0 are left
 * untouched; *n_empty_out, a device int32, receives their number).
 * Replaces torch.mean(memory_data[cluster_dict[i]], dim=0), PCCSNet/utils/utility.py:103-105. */
int mtp_cluster_means_f32(const double *sums, const double *counts, int k, int d, float *means,
                          int32_t *n_empty_out, void *stream);

/* Classifier targets from the cluster labels, the hand-off of PCCSNet/train_steps.py:357 and :364-366:
 * labels_long[i] = (int64) labels[i]  (torch.from_numpy(cluster_result).long(), the PCCS_Dataset.cluster_results of
 * PCCSNet/models/auxiliary.py:101-107) and one_hot[i, c] = (c == labels[i]) as float [n, k], every element written
 * (gt_prob = torch.zeros(n, k); gt_prob[range(n), cluster_result] = 1).  Either output may be NULL.
 * Labels must lie in [0, k) (the reference raises IndexError otherwise; the host side checks). */
int mtp_label_targets_f32(const int32_t *labels, int64_t n, int k, int64_t *labels_long, float *one_hot, void *stream);

/* Inertia: *inertia += sum_i ||x_i - c_{labels[i]}||^2 (per-point term in float with
 * sklearn's grouping, sum in double).  Replaces sklearn cluster/_k_means_common.pyx:94-124.
 * per_point may be NULL; otherwise receives the n per-point squared distances. */
int mtp_inertia_f32(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                    float *per_point, void *stream);

/* ---- Lloyd iteration, step level (multi-GPU drivers interleave an all-reduce of `red`) ----
 * red      double [k*d + k + 2]: sums, counts, n_changed, spare  (the all-reduced buffer)
 * flags    int32  [MTP_FLAG_WORDS]
 * staged   float  [mtp_staged_centers_floats(k, d)] (may be NULL when that is 0)
 * c2       float  [k]
 * All three entry points are no-ops while flags[MTP_FLAG_STATE] != MTP_STATE_RUNNING. */

/* Stage the initial centres: staged/c2 from C, zero red, flags = {RUNNING, 0, 0}. */
int mtp_lloyd_begin_f32(const float *C, int k, int d, float *staged, float *c2, double *red, int32_t *flags,
                        void *stream);

/* E-step + local M-step sums of one iteration (sklearn _k_means_lloyd.pyx:118-152):
 * labels in/out, red += local sums / counts / n_changed. */
int mtp_lloyd_estep_f32(const float *X, int64_t n, int d, const float *C, const float *staged, const float *c2,
                        int k, int32_t *labels, double *red, const int32_t *flags, void *stream);

/* The two halves of mtp_lloyd_estep_f32, separately launchable (bench.py brackets each with
 * events): labels + n_changed only / local sums + counts only. */
int mtp_lloyd_assign_f32(const float *X, int64_t n, int d, const float *C, const float *staged, const float *c2,
                         int k, int32_t *labels, double *red, const int32_t *flags, void *stream);
int mtp_lloyd_accumulate_f32(const float *X, int64_t n, int d, const int32_t *labels, int k, double *red,
                             const int32_t *flags, void *stream);

/* Finish the iteration from the (all-reduced) red buffer: average (alpha = 1/w, c *= alpha),
 * centre shift, strict / tol convergence (sklearn _k_means_common.pyx:274-311 and
 * _kmeans.py:715-734).  C_new, staged, c2, shift[k] are written, red is zeroed, flags updated.
 * If a cluster is empty and relocation_done == 0 nothing is written except
 * flags = {RELOCATE, n_iter, n_empty}; the caller then runs mtp_far_points_f32 +
 * mtp_apply_relocation_f32 and calls this again with relocation_done = 1.
 * red_backup (NULL or mtp_red_len doubles): on that pause red is MOVED there (red is zeroed),
 * so that all-reduces a multi-GPU driver has already enqueued behind the pause add zeros
 * instead of re-summing the reduced buffer; the driver copies it back before relocating.
 * tol < 0 (never sklearn's): no convergence test of either kind, the loop runs max_iter iterations. */
int mtp_lloyd_mstep_f32(double *red, const float *C_old, float *C_new, int k, int d, double tol, int max_iter,
                        int relocation_done, float *staged, float *c2, float *shift, int32_t *flags,
                        double *red_backup, void *stream);

/* (v ** 2).sum() of a contiguous float32 array in NumPy's pairwise summation order (sklearn _kmeans.py:722,
 * center_shift_tot = (center_shift**2).sum(); numpy/_core/src/umath/loops_utils.h.src) -- the total shift of
 * MTP_LLOYD_SUMS_SKLEARN_1T, exposed for its parity test.  out: 1 float (device).  One thread. */
int mtp_pairwise_sumsq_f32(const float *v, int n, float *out, void *stream);

/* Empty-cluster relocation, sklearn cluster/_k_means_common.pyx:167-211.
 * mtp_far_points_f32: the m points farthest from their assigned (old) centre, by
 * descending distances = ((X - centers_old[labels]) ** 2).sum(axis=1) (:183; float32, NumPy's pairwise
 * row sums bit for bit, so the SET of far points is sklearn's), ties to the lowest index:
 *   far_val[m] float, far_idx[m] int64 (row index + idx_offset), far_lab[m] int32,
 *   far_rows[m, d] float; entries beyond the number of points are (-1, -1, -1, 0...).
 *   dist_scratch: mtp_far_points_scratch_bytes(n) bytes (the n distances + the search's partial winners). */
int64_t mtp_far_points_scratch_bytes(int64_t n);
int mtp_far_points_f32(const float *X, int64_t n, int d, const float *C_old, const int32_t *labels, int m,
                       int64_t idx_offset, float *far_val, int64_t *far_idx, int32_t *far_lab,
                       float *far_rows, float *dist_scratch, void *stream);
/* Apply: the i-th empty cluster (ascending id) takes far_rows[i]; its old cluster loses it.
 * Only the clusters that are empty when the call starts are served (sklearn fixes empty_clusters before its
 * loop, :176): a cluster that loses its only member to an earlier relocation of the same call is not.
 * Skipped entirely when far_val[0] == 0 (more clusters than distinct points). */
int mtp_apply_relocation_f32(double *red, int k, int d, int m, const float *far_val, const int32_t *far_lab,
                             const float *far_rows, void *stream);
/* Sharded relocation: the global top-m of the ranks' local far-point lists (the all-gathered outputs of
 * mtp_far_points_f32, `total` = world * m entries in rank order): descending distance, ties to the lowest
 * global row index, entries without a point (gidx < 0) ignored; positions beyond the number of points are
 * (-1, -1, -1, 0...) -- what one GPU's search over all points returns (the selection of
 * np.argpartition(distances, -n_empty), sklearn _k_means_common.pyx:186-187, in a defined order).
 * *peer_failed (device int32) = 1 if any list carries the mark -2 of a rank whose search failed.  One block. */
int mtp_far_merge_f32(const float *gval, const int64_t *gidx, const int32_t *glab, const float *grows, int total, int m,
                      int d, float *val, int64_t *idx, int32_t *lab, float *rows, int32_t *peer_failed, void *stream);


/* Whole single-GPU Lloyd run == sklearn _kmeans_single_lloyd (cluster/_kmeans.py:624-752) on
 * already-centred data.  Synchronises the stream every `check_every` iterations to read the
 * flags; kernels of iterations enqueued past convergence are no-ops.
 *   C_inout [k, d] initial centres in, final centres out;  labels_out [n]
 *   inertia_out: 1 double (device);  n_iter_out: 1 int (HOST pointer)
 *   options: 0 or a sum of MTP_LLOYD_* values:
 *     MTP_LLOYD_NO_FINAL_ESTEP (see mtp_lloyd_sharded_f32);
 *     MTP_LLOYD_SUMS_SKLEARN_1T: the bit-exact comparison mode.  By default the centroid sums are doubles, so the
 *       run does not depend on the workgroup schedule or the GPU count; sklearn's are float32 sums whose order
 *       depends on its thread scheduling (_k_means_lloyd.pyx:143-152), so no two multi-threaded sklearn runs
 *       agree bit for bit.  With ONE thread sklearn is deterministic: every (cluster, feature) sum is a float32
 *       chain over the member rows in row order (:215-218), the counts are floats, the total shift is NumPy's
 *       pairwise float32 sum of center_shift**2 (_kmeans.py:722), the inertia a float32 chain over the points
 *       (_k_means_common.pyx:116-124).  This option forms exactly those, so that a whole
 *       np.random.seed(s); KMeans(K, n_init=20).fit(X) under threadpool_limits(1) is reproduced bit for bit
 *       (fixture G15).  Slower (one wave per cluster walks the labels); one GPU only.
 *   ws: mtp_lloyd_workspace_bytes(k, d) + mtp_far_points_scratch_bytes(n) bytes of device scratch */
#define MTP_LLOYD_NO_FINAL_ESTEP 1
#define MTP_LLOYD_SUMS_SKLEARN_1T 2
int mtp_lloyd_f32(const float *X, int64_t n, int d, float *C_inout, int k, int max_iter, double tol_abs,
                  int check_every, int32_t *labels_out, double *inertia_out, int *n_iter_out, int options, void *ws,
                  size_t ws_bytes, void *stream);

/* ---- the same run with the points sharded over the GPUs of a node: one process per GPU, RCCL over xGMI ----
 *   The reference has no distributed code; this is the multi-GPU form of the entry above.
 * A communicator is RCCL's: rank 0 calls mtp_comm_unique_id (128 bytes), the bytes reach every rank by any
 * channel (torch.distributed, MPI, a file), and every rank calls mtp_comm_create -- a collective -- with its own
 * GPU current.  librccl.so is bound at run time from the process (the one PyTorch-ROCm loaded) or the system.
 * mtp_lloyd_sharded_f32: rank r holds rows [row_offset, row_offset + n_local) of the n_total centred points and the
 * same initial centres; per iteration the library issues ONE ncclAllReduce(sum) of the packed double buffer
 * [k*d sums | k counts | n_changed | spare] on `stream` between the local sums and the M-step finish, so the
 * loop runs without the host between checks; empty clusters: ncclAllGather of the ranks' local far points.
 *   labels_out [n_local] this rank's labels; C_inout: final centres, identical on every rank;
 *   inertia_out: 1 double (device), the global inertia; n_iter_out: HOST int;
 *   phase_ms_out: NULL, or 4 HOST floats = average milliseconds of (E-step, local sums, all-reduce, M-step finish)
 *     per iteration, from HIP events the library records on `stream` (created before the first iteration; the
 *     first 1024 iterations are timed);
 *   options: 0, or MTP_LLOYD_NO_FINAL_ESTEP: when the loop ends on max_iter / the tolerance the closing E-step
 *     (sklearn _kmeans.py:736-748) is left out and labels_out are those of the last iteration -- for callers
 *     that time a fixed number of iterations (bench.py) or run the closing step themselves;
 *     MTP_LLOYD_SUMS_SKLEARN_1T is accepted only when one rank holds all points;
 *   ws: mtp_lloyd_sharded_workspace_bytes(k, d, n_local, world) bytes.
 * Every rank must call with the same k, d, max_iter, tol_abs, check_every.  comm == NULL: one rank holding all
 * points (mtp_lloyd_f32 with the timing / options arguments). */
typedef struct mtp_comm mtp_comm;
int mtp_comm_unique_id(void *id_out);
int mtp_comm_create(int rank, int world, const void *id, mtp_comm **out);
int mtp_comm_destroy(mtp_comm *comm);
int64_t mtp_lloyd_sharded_workspace_bytes(int k, int d, int64_t n_local, int world);
int mtp_lloyd_sharded_f32(const mtp_comm *comm, const float *X_local, int64_t n_local, int64_t row_offset, int64_t n_total,
                          int d, float *C_inout, int k, int max_iter, double tol_abs, int check_every, int32_t *labels_out,
                          double *inertia_out, int *n_iter_out, float *phase_ms_out, int options, void *ws, size_t ws_bytes,
                          void *stream);

/* n_init restarts at once (the loop over inits of sklearn KMeans.fit, cluster/_kmeans.py:1495-1532, which
 * PCCSNet/utils/utility.py:120-121 runs with n_init=20): `batch` independent mtp_lloyd_f32 runs on the same
 * points sharing every launch, each with its own state / convergence.  d must be 64, 96 or 128.
 *   C_inout [batch, k, d]; labels_out [batch, n]; inertia_out [batch] doubles (device);
 *   n_iter_out [batch] ints (HOST);  options: 0 or MTP_LLOYD_SUMS_SKLEARN_1T (mtp_lloyd_f32);
 *   ws: mtp_lloyd_batched_workspace_bytes(k, d, batch, n) bytes. */
int64_t mtp_lloyd_batched_workspace_bytes(int k, int d, int batch, int64_t n);
int mtp_lloyd_batched_f32(const float *X, int64_t n, int d, float *C_inout, int k, int batch, int max_iter,
                          double tol_abs, int check_every, int32_t *labels_out, double *inertia_out, int *n_iter_out,
                          int options, void *ws, size_t ws_bytes, void *stream);

/* k-means++ seeding, sklearn cluster/_kmeans.py:174-272 with unit sample weights, for one
 * restart.  The caller draws the random numbers from the same NumPy RandomState in sklearn's
 * order -- center_id = rs.choice(n, p=w/w.sum()), then u = rs.uniform(size=(k-1, l)) with
 * l = 2 + int(ln k) -- and passes u as a DEVICE array.  Only enqueues (2 launches per centre);
 * indices_out[k] (int64) and centers_out[k, d] = X[indices_out] are device arrays.
 * ws: mtp_kpp_workspace_bytes(n, d, l) bytes. */
/* pot_order: how the float32 potentials (sklearn's candidates_pot / current_pot, _kmeans.py:216, :252) are
 * summed.  MTP_POT_OPENBLAS: in the float addition order of the reference stack -- NumPy handing
 * [L, n] @ [n, 1] to OpenBLAS sgemv_t and [1, n] @ [n] to sdot (0.3.29 x86-64: the sgemv_t order is shared by the SkylakeX and Haswell
 * kernels, the sdot order is the SkylakeX / AVX-512 micro-kernel's only; parity on an AVX2-only host is unpinned) --
 * so that the draws rand = u * pot, and with them the seeds, are sklearn's bit for bit;
 * MTP_POT_F64: double sums rounded to float (a draw within ~1e-7 relative of a cumulative-sum boundary can
 * then pick the neighbouring point). */
#define MTP_POT_F64 0
#define MTP_POT_OPENBLAS 1
int64_t mtp_kpp_workspace_bytes(int64_t n, int d, int l);
int mtp_kmeans_plusplus_f32(const float *X, int64_t n, int d, int k, int l, int64_t center_id, const double *u,
                            int pot_order, int64_t *indices_out, float *centers_out, void *ws, size_t ws_bytes,
                            void *stream);
/* out[j] = sum_i rows[j, i] (float32 rows [l, n], l <= 16) in the float addition order MTP_POT_OPENBLAS stands
 * for: what ``rows @ numpy.ones((n, 1), float32)`` returns through OpenBLAS 0.3.29's x86-64 sgemv_t kernels
 * (l >= 2) and ``rows @ numpy.ones(n, float32)`` through sdot (l == 1) -- sklearn's candidates_pot and
 * current_pot (cluster/_kmeans.py:252, :216).  The seeding entries use the same kernels internally; this
 * entry exposes them so the order can be checked bit for bit on its own.  ws: *_workspace_bytes(n, l). */
int64_t mtp_blas_order_rowsums_workspace_bytes(int64_t n, int l);
int mtp_blas_order_rowsums_f32(const float *rows, int l, int64_t n, float *out, void *ws, size_t ws_bytes,
                               void *stream);

/* The same two orders for rows that are cut into consecutive pieces at multiples of 4096 points -- row-sharded
 * points, one piece per rank (pccs_accel.seeding.kmeans_plusplus_sharded).  sgemv_t: a rank forms the 4096-point
 * block sums of its piece and the sum of its n & 3 trailing points (only the last piece has any),
 *   part [batch][l][nb + 1], nb = mtp_blas_order_blocks(n);
 * pot64 (optional, [batch][l] double sums of the WHOLE rows) and eps: only the rows whose double sum is within
 * (1 + eps) / (1 - eps) of the restart's smallest are walked (the only ones that can hold the smallest float32
 * sum when eps covers the chains' rounding, (1100 + blocks) * 2^-23); the block sums of the others are left as
 * the caller initialised them;
 * the pieces' parts are gathered, laid side by side in rank order and added one after the other in float
 * (mtp_blas_order_combine_f32: out[r] = sum over part[r][0 .. m), total_rows rows) -- the additions of
 * `rows @ sample_weight` (sklearn cluster/_kmeans.py:252) in OpenBLAS' order, whatever the number of ranks.
 * sdot (cluster/_kmeans.py:216): acc [batch][64] travels from rank to rank over the first n_total & -64 points
 * (main_local of them on this rank, a multiple of 64); the rank holding the last point passes finish = 1 and the
 * left_over = n_total - (n_total & -32 & -64) points that follow, and receives out [batch]. */
int64_t mtp_blas_order_blocks(int64_t n);
int mtp_blas_order_block_sums_f32(const float *rows, int l, int64_t n, int batch, const double *pot64, double eps,
                                  float *part, void *stream);
int mtp_blas_order_combine_f32(const float *part, int64_t m, int64_t total_rows, float *out, void *stream);
int mtp_blas_order_sdot_carry_f32(const float *x, int64_t row_stride, int batch, int64_t main_local, int64_t left_over,
                                  float *acc, int finish, float *out, void *stream);
/* The same for `batch` restarts sharing every launch: center_ids [batch] int64 (DEVICE), u [batch, k-1, l],
 * indices_out [batch, k], centers_out [batch, k, d]; ws: mtp_kpp_batched_workspace_bytes(n, d, l, batch). */
int64_t mtp_kpp_batched_workspace_bytes(int64_t n, int d, int l, int batch);
int mtp_kmeans_plusplus_batched_f32(const float *X, int64_t n, int d, int k, int l, int batch, const int64_t *center_ids,
                                    const double *u, int pot_order, int64_t *indices_out, float *centers_out, void *ws,
                                    size_t ws_bytes, void *stream);

/* Step-level pieces of the same seeding for ROW-SHARDED points (one process per GPU; the host side
 * -- pccs_accel.seeding.kmeans_plusplus_sharded -- locates the draws through an exchange of the ranks'
 * cumulative sums and passes every rank the same candidate rows).  sklearn cluster/_kmeans.py:222-262.
 *   mtp_kpp_row_sqnorm_f64: yy[i] = |x_i|^2 in double (the YY term of sklearn's upcast
 *     _euclidean_distances, metrics/pairwise.py:391-442), computed once per fit.
 *   mtp_kpp_candidate_dists_f32: rows_out[b, l, i] = min(closest[b, i], fp32(max(0, |c_bl|^2 - 2 c_bl.x_i + |x_i|^2)))
 *     for the n_cand (<= 8) candidate rows cand[b, l, :] of each of `batch` restarts and this rank's n
 *     points (closest == NULL: no minimum, the first centre), and pot_out[b, l] = sum_i rows_out[b, l, i]
 *     (double; the rank-local part of sklearn's candidates_pot).  d must be a multiple of 16.
 *     ws: mtp_kpp_candidate_dists_workspace_bytes(n, d, batch) bytes.
 *   mtp_kpp_count_below_f32: counts[b, j] = #{ i : closest[b, 0] + ... + closest[b, i] < thresholds[b, j] }
 *     (double sums) = np.searchsorted(stable_cumsum(closest[b]), thresholds[b, j]) on this rank's rows
 *     (_kmeans.py:234-238); with the thresholds shifted by the lower ranks' totals, the ranks' counts add
 *     up to the candidate id.  closest [batch, n] float, thresholds [batch, l] double, counts [batch, l]
 *     int64, l <= 16.  ws: mtp_kpp_count_below_workspace_bytes(n, batch) bytes. */
int64_t mtp_kpp_candidate_dists_workspace_bytes(int64_t n, int d, int batch);
int64_t mtp_kpp_count_below_workspace_bytes(int64_t n, int batch);
int mtp_kpp_count_below_f32(const float *closest, int64_t n, int batch, const double *thresholds, int l,
                            int64_t *counts, void *ws, size_t ws_bytes, void *stream);
int mtp_kpp_row_sqnorm_f64(const float *X, int64_t n, int d, double *yy, void *stream);
int mtp_kpp_candidate_dists_f32(const float *X, int64_t n, int d, const double *yy, const float *cand, int batch,
                                int n_cand, const float *closest, float *rows_out, double *pot_out, void *ws,
                                size_t ws_bytes, void *stream);

/* Best-of-K ADE/FDE.  pred [n, kstride, t, 2], gt [n, t, 2]; the first k (<= kstride)
 * candidates of every agent are scored:
 *   ADE_c = (sum_t ||gt_t - pred_c,t||) / ade_div,  FDE_c = ||gt_T - pred_c,T||
 *   mode 0: i = argmin ADE -> (ADE_i, FDE_i); mode 1: i = argmin FDE -> (ADE_i, FDE_i);
 *   mode 2: (min ADE, min FDE) independently (idx = argmin ADE, idx2 = argmin FDE).
 * argmin = first minimum.  sums[0..1] (doubles) += sum over agents of the reported pair.
 * Replaces PCCSNet/utils/metrics.py:4-23 + PCCSNet/train_steps.py:219-232, and (f64,
 * ade_div = 1) StimulusVerification/verify_predictions.py:238-247.
 * min_ade/min_fde/idx/idx2 may each be NULL. */
int mtp_best_of_k_f32(const float *pred, const float *gt, int64_t n, int k, int kstride, int t, float ade_div,
                      int mode, float *min_ade, float *min_fde, int32_t *idx, int32_t *idx2, double *sums,
                      void *stream);
int mtp_best_of_k_f64(const double *pred, const double *gt, int64_t n, int k, int kstride, int t,
                      double ade_div, int mode, double *min_ade, double *min_fde, int32_t *idx, int32_t *idx2,
                      double *sums, void *stream);

/* Per-candidate ADE / FDE ("raw" mode of metrics.py:4-23): ade[n*k], fde[n*k]. */
int mtp_ade_fde_f32(const float *pred, const float *gt, int64_t n, int k, int kstride, int t, float ade_div,
                    float *ade, float *fde, void *stream);

/* ---- decode glue around the evaluation ---- */

/* out[a, c, j, :] = M_a @ (sum_{i<=j} disp[a, c, i, :]) + last_obs[a]   (float32, [n, k, t, 2]).
 * Replaces transform_trajectory (PCCSNet/utils/data_process.py:111-122: in-place prefix sum over
 * time, float adds in time order -- with rot = last_obs = NULL the result is bit-identical) and
 * the rotate-back + last-position add of gen_prediction (PCCSNet/train_steps.py:253-255:
 * torch.bmm(mat.repeat(k,1,1), out.transpose(1,2)).transpose(1,2) + last_obs).
 * rot [n, 2, 2] or NULL (identity), last_obs [n, 2] or NULL (zero); out may alias disp. */
int mtp_decode_trajectories_f32(const float *disp, int64_t n, int k, int t, const float *rot, const float *last_obs,
                                float *out, void *stream);

/* mtp_best_of_k_f32 on the network's raw per-step displacements: the prefix sum of
 * transform_trajectory runs inside the scoring pass (PCCSNet/train_steps.py:216-220 in one
 * launch); the decoded trajectories are never written. */
int mtp_best_of_k_disp_f32(const float *disp, const float *gt, int64_t n, int k, int kstride, int t, float ade_div,
                           int mode, float *min_ade, float *min_fde, int32_t *idx, int32_t *idx2, double *sums,
                           void *stream);

/* ---- score-sorted NMS re-ranking of stored prediction sets (section 8, row f4) ---- */

/* Per agent: sort the first max_samples candidates by binned verification score (descending;
 * score // bin_size as NumPy floor_divide, key = float32(score) - float32(j)/(2*max_samples)),
 * then greedy NMS on the final predicted point (keep while fewer than best_of are kept and no
 * kept candidate is closer than nms_thresh); new list = kept ++ dropped.
 * Replaces StimulusVerification/verify_predictions.py:251-277.
 *   pred [n, mtot, t, 2]; scores [n, mtot] float32 (scores_f64 = 0) or float64 (= 1)
 *   order [n, max_samples] int32 out: original index of the candidate at each new position
 *   n_selected [n] int32 out (may be NULL): number of kept candidates (<= best_of)
 * max_samples <= min(mtot, 1024). */
int mtp_nms_rerank_f32(const float *pred, const void *scores, int scores_f64, int64_t n, int mtot, int t,
                       int max_samples, int best_of, double bin_size, double nms_thresh, int32_t *order,
                       int32_t *n_selected, void *stream);
int mtp_nms_rerank_f64(const double *pred, const void *scores, int scores_f64, int64_t n, int mtot, int t,
                       int max_samples, int best_of, double bin_size, double nms_thresh, int32_t *order,
                       int32_t *n_selected, void *stream);

/* out[a, p] = pred[a, order[a, p]] (p < m), out[a, p] = pred[a, p] (m <= p < mtot): the
 * re-ordered copy verify_predictions.py:277 builds.  order [n, m]; out must not alias pred. */
int mtp_gather_candidates_f32(const float *pred, const int32_t *order, int64_t n, int mtot, int m, int t, float *out,
                              void *stream);
int mtp_gather_candidates_f64(const double *pred, const int32_t *order, int64_t n, int mtot, int m, int t, double *out,
                              void *stream);

/* mtp_best_of_k_* over the re-ranked list without materialising it: candidate c of agent a is
 * pred[a, order[a*ostride + c]] (verify_predictions.py:280-290); idx / idx2 are positions c. */
int mtp_best_of_k_reranked_f32(const float *pred, const float *gt, const int32_t *order, int ostride, int64_t n, int k,
                               int kstride, int t, float ade_div, int mode, float *min_ade, float *min_fde,
                               int32_t *idx, int32_t *idx2, double *sums, void *stream);
int mtp_best_of_k_reranked_f64(const double *pred, const double *gt, const int32_t *order, int ostride, int64_t n,
                               int k, int kstride, int t, double ade_div, int mode, double *min_ade, double *min_fde,
                               int32_t *idx, int32_t *idx2, double *sums, void *stream);

/* ---- modality-loss soft labels (section 8, row f2) ---- */

/* last[i] = trj[i, t, :] (float32) and speed_angle[i] = cmath.polar(trj[i, t] - trj[i, 0]) in double
 * (the offset is a float32 subtraction).  trj [n, tlen, 2].
 * Replaces get_speed_and_angle, PCCSNet/utils/data_process.py:125-131 (+ last_coords, :140). */
int mtp_speed_angle_f32(const float *trj, int64_t n, int tlen, int t, float *last, double *speed_angle, void *stream);

/* Scene neighbour search, PCCSNet/utils/data_process.py:143-160: j is a neighbour of i when both are
 * in the same scene, j != i, ||last_i - last_j|| <= rad (float32) and (both speeds are 0, or
 * 0.9 s_i < s_j < 1.1 s_i and the heading difference, wrapped to [0, pi], is < 0.1 pi).
 *   scene_of [n] int32 scene index of each sample; split_marks [n_scenes + 1] int64 (device)
 * Two passes build the CSR lists in ascending j (the reference's order):
 *   count pass: counts != NULL, offsets = ids = NULL  -> counts[i]
 *   fill pass:  counts == NULL, offsets [n] (exclusive prefix of counts), ids [sum counts] int64 */
int mtp_scene_neighbors_f32(const float *last, const double *speed_angle, const int32_t *scene_of,
                            const int64_t *split_marks, int64_t n, double rad, int64_t *counts, const int64_t *offsets,
                            int64_t *ids, void *stream);

/* dense[i, c] = (#neighbours of i with label c + [c == labels[i]]) / (len(neighbours) + 1), float64
 * [n, k], every element written.  offsets [n + 1], ids: the CSR lists above; labels [n] in [0, k).
 * Replaces gen_prob + the dense fill of get_scene_data, data_process.py:172-187, :212-218. */
int mtp_soft_labels_f64(const int64_t *offsets, const int64_t *ids, const int32_t *labels, int64_t n, int k,
                        double *dense, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MTP_ACCEL_H */
