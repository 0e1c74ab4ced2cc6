// M-step of Lloyd's algorithm on gfx950 and the small kernels around it.
//
//   k_accumulate      segmented sum of member rows by label -> double sums / counts
//                     (sklearn cluster/_k_means_lloyd.pyx:215-218,143-152; get_modality's gather,
//                      PCCSNet/utils/utility.py:98-104).  HBM-bound: one pass over X.
//   k_accumulate_chain  the same sums as float32 chains in row order (MTP_LLOYD_SUMS_SKLEARN_1T: sklearn on one thread)
//   k_mstep_centers   average, centre shift, staged image + ||c||^2 of the new centres
//   k_mstep_flags     convergence decision (sklearn cluster/_kmeans.py:715-734)
//   k_inertia         sklearn cluster/_k_means_common.pyx:94-124
//   k_col_*           KMeans.fit preprocessing (mean / variance / centring)
//   k_argmax_partial, k_argmax_final, k_apply_relocation   empty-cluster relocation
//
// Sums are accumulated in double (LDS ds_add_f64, then global_atomic_add_f64): the order of the
// additions then changes the result only at the 1e-16 level, so centroids do not depend on the
// workgroup schedule or on how many GPUs share the points.
#include "accumulate_kernel.h"

// scalar fallback for d % 4 != 0 (known-answer tests): plain global double atomics
__global__ __launch_bounds__(256) void k_accumulate_scalar(const float *__restrict__ X, int64_t n, int d,
                                                           const int32_t *__restrict__ labels, int k, double *sums,
                                                           double *counts, const int32_t *state) {
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int lab = labels[row];
    if (lab < 0 || lab >= k) return;
    atomicAdd(&counts[lab], 1.0);
    for (int c = 0; c < d; ++c) atomicAdd(&sums[(size_t)lab * d + c], (double)X[row * d + c]);
}

// MTP_LLOYD_SUMS_SKLEARN_1T: the sums sklearn forms with ONE thread (cluster/_k_means_lloyd.pyx:215-218; the
// thread-local buffer every row is added into is itself added once to zeros, :143-152): per (cluster, feature) a
// float32 chain over the member rows in row order, and float32 counts.  One wave per (cluster, restart, block of
// 128 features): it scans all labels in order (L2-resident) and adds its members' rows as it meets them, lane l
// owning features l and l + 64 of the block.  The chain is what bounds it (a dependent v_add_f32 per member row,
// its row fetched a few rows ahead), not memory: this is the bit-exact comparison mode, not the fast path.
// The float results are stored into the double reduce buffer (exactly), overwriting it.
__global__ __launch_bounds__(64) void k_accumulate_chain(const float *__restrict__ X, int64_t n, int d,
                                                         const int32_t *__restrict__ labels, int k, double *sums,
                                                         double *counts, const int32_t *state, MtpBatchStride bs) {
    {
        const size_t b = blockIdx.y;  // restart
        labels += b * bs.labels;
        sums += b * bs.red;
        counts += b * bs.red;
        if (state != nullptr) state += b * bs.flags;
    }
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    const int j = blockIdx.x, lane = threadIdx.x;
    const int f0 = blockIdx.z * 128 + lane, f1 = f0 + 64;
    const bool has0 = f0 < d, has1 = f1 < d;
    float acc0 = 0.f, acc1 = 0.f, cnt = 0.f;
    int lab_next = (lane < n) ? labels[lane] : -1;
    for (int64_t base = 0; base < n; base += 64) {
        const int lab = lab_next;
        lab_next = (base + 64 + lane < n) ? labels[base + 64 + lane] : -1;  // travels while this batch is added
        unsigned long long m = __ballot(lab == j);
        while (m != 0) {  // wave-uniform
            constexpr int G = 4;  // member rows whose loads are in flight together; the adds stay in row order
            float v0[G], v1[G];
            int nr = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v0[g] = 0.f;
                v1[g] = 0.f;
                if (m != 0) {
                    const int64_t row = base + __builtin_ctzll(m);
                    m &= m - 1;
                    if (has0) v0[g] = X[row * d + f0];
                    if (has1) v1[g] = X[row * d + f1];
                    nr = g + 1;
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < nr) {
                    acc0 += v0[g];
                    acc1 += v1[g];
                    cnt += 1.0f;
                }
            }
        }
    }
    if (has0) sums[(size_t)j * d + f0] = (double)acc0;
    if (has1) sums[(size_t)j * d + f1] = (double)acc1;
    if (blockIdx.z == 0 && lane == 0) counts[j] = (double)cnt;
}

static constexpr size_t ACC_LDS_BYTES = 152 * 1024;

int mtp_launch_accumulate(const float *X, int64_t n, int d, const int32_t *labels, int k, double *sums,
                          double *counts, const int32_t *state, hipStream_t s, MtpBatchStride bs, int sums_mode) {
    if (n == 0) return MTP_OK;
    if (sums_mode == MTP_LLOYD_SUMS_SKLEARN_1T) {
        MTP_REQUIRE(k <= 65535 * 32 && bs.batch <= 65535, "mtp_centroid_sums: k / batch too large for the chain kernel");
        hipLaunchKernelGGL(k_accumulate_chain, dim3((unsigned)k, (unsigned)bs.batch, (unsigned)((d + 127) / 128)), dim3(64), 0,
                           s, X, n, d, labels, k, sums, counts, state, bs);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    if (d % 4 != 0 || (size_t)d * 8 + 4 > ACC_LDS_BYTES) {
        MTP_REQUIRE(bs.batch == 1, "mtp_centroid_sums: batched restarts need d %% 4 == 0");
        const int64_t blocks = (n + 255) / 256;
        MTP_REQUIRE(blocks <= 0x7fffffff, "mtp_centroid_sums: n too large");
        hipLaunchKernelGGL(k_accumulate_scalar, dim3((unsigned)blocks), dim3(256), 0, s, X, n, d, labels, k, sums,
                           counts, state);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    const int dq = d / 4;
    int kr = (int)(ACC_LDS_BYTES / ((size_t)d * 8 + 4));
    if (kr > k) kr = k;
    const int n_ranges = (k + kr - 1) / kr;
    kr = (k + n_ranges - 1) / n_ranges;  // balance the ranges
    const size_t lds = (size_t)kr * d * 8 + (size_t)kr * 4;
    // One round of workgroups: a CU holds as many as fit its 160 KB of LDS (one at K = 200 / D = 96 and at the headline
    // size), and the grid (chunks x ranges x restarts) must not exceed that -- 13 chunks x 20 restarts = 260
    // workgroups on 256 CUs ran as two rounds.  A chunk must also be worth its flush (kr*d doubles of atomics).
    int cus = 0;
    if (const int rc = mtp_device_cu_count(&cus)) return rc;
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    const int64_t slots = (int64_t)cus * per_cu;
    int64_t n_chunks = slots / ((int64_t)n_ranges * bs.batch);
    const int64_t min_rows = (int64_t)8 * kr + 1024;
    if (n_chunks > (n + min_rows - 1) / min_rows) n_chunks = (n + min_rows - 1) / min_rows;
    if (n_chunks < 1) n_chunks = 1;
    int64_t rows_per_chunk = (n + n_chunks - 1) / n_chunks;
    rows_per_chunk = (rows_per_chunk + 63) / 64 * 64;
    n_chunks = (n + rows_per_chunk - 1) / rows_per_chunk;
    const dim3 grid((unsigned)n_chunks, (unsigned)n_ranges, (unsigned)bs.batch);
    // a single run reads every row once, in one workgroup: non-temporal row loads (0.924 -> 0.912 ms at N = 10M,
    // 0.143 -> 0.127 ms at the 8-GPU shard size); batched restarts read the rows once per restart and keep them cached
#define MTP_ACC_LAUNCH_D(LPR, DT)                                                                                  \
    do {                                                                                                           \
        if (bs.batch == 1) {                                                                                       \
            MTP_SET_MAX_LDS_ONCE((&k_accumulate<LPR, 4, double, true, DT, true>), ACC_LDS_BYTES + 1024);             \
            hipLaunchKernelGGL((k_accumulate<LPR, 4, double, true, DT, true>), grid, dim3(1024), lds, s, X, n, d,    \
                               labels, k, kr, rows_per_chunk, sums, counts, state, bs);                            \
        } else {                                                                                                   \
            MTP_SET_MAX_LDS_ONCE((&k_accumulate<LPR, 4, double, true, DT>), ACC_LDS_BYTES + 1024);                   \
            hipLaunchKernelGGL((k_accumulate<LPR, 4, double, true, DT>), grid, dim3(1024), lds, s, X, n, d, labels,  \
                               k, kr, rows_per_chunk, sums, counts, state, bs);                                    \
        }                                                                                                          \
    } while (0)
#define MTP_ACC_LAUNCH(LPR) MTP_ACC_LAUNCH_D(LPR, 0)
#define MTP_ACC_LAUNCH_ALL_D(LPR, DT)                                                                              \
    do {                                                                                                           \
        MTP_SET_MAX_LDS_ONCE((&k_accumulate_all<LPR, 4, DT>), ACC_LDS_BYTES + 1024);                                 \
        hipLaunchKernelGGL((k_accumulate_all<LPR, 4, DT>), dim3((unsigned)(n_chunks * bs.batch)), dim3(1024), lds, s, X, n,  \
                           d, labels, k, rows_per_chunk, sums, counts, state, bs);                                 \
    } while (0)
#define MTP_ACC_LAUNCH_ALL(LPR) MTP_ACC_LAUNCH_ALL_D(LPR, 0)
    // the reference's feature sizes with d as a compile-time constant (same arithmetic, fewer instructions per row)
    if (n_ranges == 1 && (d == 96 || d == 128 || d == 64)) {
        if (d == 96)
            MTP_ACC_LAUNCH_ALL_D(32, 96);
        else if (d == 128)
            MTP_ACC_LAUNCH_ALL_D(32, 128);
        else
            MTP_ACC_LAUNCH_ALL_D(16, 64);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    if (n_ranges > 1 && (d == 96 || d == 128 || d == 64)) {
        if (d == 96)
            MTP_ACC_LAUNCH_D(32, 96);
        else if (d == 128)
            MTP_ACC_LAUNCH_D(32, 128);
        else
            MTP_ACC_LAUNCH_D(16, 64);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    if (n_ranges == 1) {  // every row is added: no selection of a range's rows
        if (dq <= 8)
            MTP_ACC_LAUNCH_ALL(8);
        else if (dq <= 16)
            MTP_ACC_LAUNCH_ALL(16);
        else if (dq <= 32)
            MTP_ACC_LAUNCH_ALL(32);
        else
            MTP_ACC_LAUNCH_ALL(64);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    if (dq <= 8)
        MTP_ACC_LAUNCH(8);
    else if (dq <= 16)
        MTP_ACC_LAUNCH(16);
    else if (dq <= 32)
        MTP_ACC_LAUNCH(32);
    else
        MTP_ACC_LAUNCH(64);
#undef MTP_ACC_LAUNCH
#undef MTP_ACC_LAUNCH_ALL
#undef MTP_ACC_LAUNCH_D
#undef MTP_ACC_LAUNCH_ALL_D
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ------------------------------------------------------------------------------------------
// M-step finish.  One block per cluster j (64 threads).  Every block first counts the empty
// clusters itself (k doubles from L2) so that no extra launch is needed for the decision.
// ------------------------------------------------------------------------------------------
__device__ inline int block_count_empty(const double *counts, int k, int *scratch) {
    int local = 0;
    for (int j = threadIdx.x; j < k; j += blockDim.x) local += (counts[j] == 0.0);
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = local;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < nw; ++w) tot += scratch[w];
    __syncthreads();
    return tot;
}

// sklearn _k_means_common.pyx:16-43 (float): groups of four, ((t0+t1)+t2)+t3, then remainder.
__device__ inline float euclid_sq_sklearn(const float *a, const float *b, int d) {
    const int n4 = d / 4, rem = d % 4;
    float result = 0.f;
    for (int i = 0; i < n4; ++i) {
        const float t0 = (a[0] - b[0]) * (a[0] - b[0]), t1 = (a[1] - b[1]) * (a[1] - b[1]);
        const float t2 = (a[2] - b[2]) * (a[2] - b[2]), t3 = (a[3] - b[3]) * (a[3] - b[3]);
        result += ((t0 + t1) + t2) + t3;
        a += 4;
        b += 4;
    }
    for (int i = 0; i < rem; ++i) result += (a[i] - b[i]) * (a[i] - b[i]);
    return result;
}

__global__ __launch_bounds__(64) void k_mstep_centers(double *red, const float *C_old, float *C_new,
                                                      int k, int d, int relocation_done, float *staged, float *c2,
                                                      float *shift, const int32_t *flags, MtpBatchStride bs,
                                                      int flag_parity) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // 2*d floats: new row, old row
    __shared__ int scratch[2];
    int img_row = blockIdx.x;
    {
        const size_t b = blockIdx.y;
        red += b * bs.red;
        flags += b * bs.flags;
        if (staged != nullptr) staged = mtp_image_of(staged, bs, b, (int)blockIdx.x, &img_row);
        c2 += b * bs.kvec;
        shift += b * bs.kvec;
        const float *c0 = C_old + b * bs.c0;
        float *c1 = C_new + b * bs.c1;
        if (flag_parity && (flags[MTP_FLAG_N_ITER] & 1)) {  // this restart's centres live in buffer 1
            C_old = c1;
            C_new = const_cast<float *>(c0);
        } else {
            C_old = c0;
            C_new = c1;
        }
    }
    if (flags[MTP_FLAG_STATE] != MTP_STATE_RUNNING) return;
    const double *counts = red + (size_t)k * d;
    const int n_empty = block_count_empty(counts, k, scratch);
    if (n_empty > 0 && !relocation_done) return;  // k_mstep_flags pauses the loop
    const int j = blockIdx.x;
    const bool valid = j < k;
    float *rn = rowbuf, *ro = rowbuf + d;
    __shared__ float c2s;
    if (valid) {
        // empty cluster after a declined relocation: sklearn copies the heaviest cluster's
        // centre (_k_means_common.pyx:291-295); its averaged value is sums/count of that row.
        int src = j;
        if (counts[j] == 0.0) {
            src = 0;
            for (int q = 1; q < k; ++q)
                if (counts[q] > counts[src]) src = q;
        }
        // (sklearn averages in place in index order, so a copy taken from a heavier cluster
        // with a HIGHER index is that cluster's not-yet-averaged sum; reproduced as is.)
        const float alpha = (src > j) ? 1.0f : 1.0f / (float)counts[src];
        for (int c = threadIdx.x; c < d; c += 64) {
            rn[c] = (float)red[(size_t)src * d + c] * alpha;
            ro[c] = C_old[(size_t)j * d + c];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            c2s = mtp_sqnorm_einsum_order(rn, d);
            c2[j] = c2s;
            shift[j] = sqrtf(euclid_sq_sklearn(rn, ro, d));
        }
        for (int c = threadIdx.x; c < d; c += 64) C_new[(size_t)j * d + c] = rn[c];
    }
    __syncthreads();
    if (staged != nullptr) mtp_write_staged_row(staged, img_row, d, rn, valid, valid ? c2s : 0.f, threadIdx.x, 64);
}

// NumPy's float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum; what
// `(center_shift ** 2).sum()` of sklearn _kmeans.py:722 runs on a contiguous float32 array): fewer than 8
// elements one after the other; up to 128 with eight accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then
// the n % 8 rest; beyond that the halves (the first a multiple of 8) summed recursively and added.
// a[i] = shift[i] * shift[i], rounded to float like center_shift ** 2.  Restated in oracle.c (orc_pairwise_sum_f32).
__device__ inline float pw_leaf_sq(const float *sh, int off, int n) {
    if (n < 8) {
        float res = -0.0f;
        for (int i = 0; i < n; ++i) res += sh[off + i] * sh[off + i];
        return res;
    }
    float r[8];
    for (int q = 0; q < 8; ++q) r[q] = sh[off + q] * sh[off + q];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int q = 0; q < 8; ++q) r[q] += sh[off + i + q] * sh[off + i + q];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += sh[off + i] * sh[off + i];
    return res;
}
__device__ inline float pairwise_sum_sq_numpy(const float *sh, int n) {
    struct Frame {
        int off, n, stage;
        float left;
    } st[40];
    int sp = 0;
    st[sp++] = Frame{0, n, 0, 0.f};
    float ret = 0.f;
    while (sp > 0) {
        Frame &f = st[sp - 1];
        if (f.n <= 128) {
            ret = pw_leaf_sq(sh, f.off, f.n);
            --sp;
            continue;
        }
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) {
            f.stage = 1;
            st[sp++] = Frame{f.off, n2, 0, 0.f};
        } else if (f.stage == 1) {
            f.left = ret;
            f.stage = 2;
            st[sp++] = Frame{f.off + n2, f.n - n2, 0, 0.f};
        } else {
            ret = f.left + ret;
            --sp;
        }
    }
    return 0.0f + ret;  // np.add.reduce starts from the identity
}

// The same order over (a[i] - b[i])^2: one row of sklearn's `distances = ((X - centers_old[labels]) ** 2).sum(axis=1)`
// (_k_means_common.pyx:183) -- float32 difference, float32 square, NumPy's pairwise sum along the row.
__device__ inline float pw_leaf_sqdiff(const float *__restrict__ a, const float *__restrict__ b, int off, int n) {
    if (n < 8) {
        float res = -0.0f;
        for (int i = 0; i < n; ++i) {
            const float v = a[off + i] - b[off + i];
            res += v * v;
        }
        return res;
    }
    float r[8];
    for (int q = 0; q < 8; ++q) {
        const float v = a[off + q] - b[off + q];
        r[q] = v * v;
    }
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int q = 0; q < 8; ++q) {
            const float v = a[off + i + q] - b[off + i + q];
            r[q] += v * v;
        }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) {
        const float v = a[off + i] - b[off + i];
        res += v * v;
    }
    return res;
}
__device__ inline float pairwise_sum_sqdiff_numpy(const float *__restrict__ a, const float *__restrict__ b, int n) {
    struct Frame {
        int off, n, stage;
        float left;
    } st[40];
    int sp = 0;
    st[sp++] = Frame{0, n, 0, 0.f};
    float ret = 0.f;
    while (sp > 0) {
        Frame &f = st[sp - 1];
        if (f.n <= 128) {
            ret = pw_leaf_sqdiff(a, b, f.off, f.n);
            --sp;
            continue;
        }
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) {
            f.stage = 1;
            st[sp++] = Frame{f.off, n2, 0, 0.f};
        } else if (f.stage == 1) {
            f.left = ret;
            f.stage = 2;
            st[sp++] = Frame{f.off + n2, f.n - n2, 0, 0.f};
        } else {
            ret = f.left + ret;
            --sp;
        }
    }
    return 0.0f + ret;
}

// distances of the empty-cluster relocation: what sklearn ranks its far points on, bit for bit.  One thread per row
// (relocations are rare: the kernel's speed does not show in a fit).
__global__ __launch_bounds__(256) void k_far_distances(const float *__restrict__ X, int64_t n, int d,
                                                       const float *__restrict__ C, const int32_t *__restrict__ labels,
                                                       float *dist) {
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x)
        dist[row] = pairwise_sum_sqdiff_numpy(X + row * d, C + (size_t)labels[row] * d, d);
}

__global__ void k_pairwise_sumsq(const float *v, int n, float *out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = pairwise_sum_sq_numpy(v, n);
}

extern "C" int mtp_pairwise_sumsq_f32(const float *v, int n, float *out, void *stream) {
    MTP_REQUIRE(v && out && n >= 0, "mtp_pairwise_sumsq: bad arguments");
    hipLaunchKernelGGL(k_pairwise_sumsq, dim3(1), dim3(64), 0, mtp_stream(stream), v, n, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// One 1024-thread block per restart: the decision needs every count and shift, and the block then
// zeroes the reduce buffer for the next iteration (1 MB at K=1024, D=128: 16 waves of 512-byte stores).
__global__ __launch_bounds__(1024) void k_mstep_flags(double *red, int k, int d, double tol, int max_iter,
                                                      int relocation_done, const float *shift, int32_t *flags,
                                                      double *red_backup, MtpBatchStride bs, int sums_mode) {
    __shared__ int scratch[16];
    __shared__ double dsum[16];
    {
        const size_t b = blockIdx.x;
        red += b * bs.red;
        shift += b * bs.kvec;
        flags += b * bs.flags;
    }
    if (flags[MTP_FLAG_STATE] != MTP_STATE_RUNNING) return;
    double *counts = red + (size_t)k * d;
    const int n_empty = block_count_empty(counts, k, scratch);
    if (n_empty > 0 && !relocation_done) {
        if (red_backup != nullptr) {  // park the reduced buffer: later all-reduces then add zeros
            const int64_t len = mtp_red_len(k, d);
            for (int64_t e = threadIdx.x; e < len; e += blockDim.x) {
                red_backup[e] = red[e];
                red[e] = 0.0;
            }
        }
        if (threadIdx.x == 0) {
            flags[MTP_FLAG_STATE] = MTP_STATE_RELOCATE;
            flags[MTP_FLAG_N_EMPTY] = n_empty;
        }
        return;
    }
    // sum_j shift_j^2: squares in float like (center_shift**2), summed in double
    double local = 0.0;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        const float sh = shift[j];
        local += (double)(sh * sh);
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) dsum[threadIdx.x >> 6] = local;
    __syncthreads();
    const double n_changed = red[(size_t)k * d + k];
    __syncthreads();
    // zero the reduce buffer for the next iteration
    const int64_t len = mtp_red_len(k, d);
    for (int64_t e = threadIdx.x; e < len; e += blockDim.x) red[e] = 0.0;
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += dsum[w];
        // sklearn's own float32 total in NumPy's order for the bit-exact comparison mode (one thread walks it)
        if (sums_mode == MTP_LLOYD_SUMS_SKLEARN_1T) tot = (double)pairwise_sum_sq_numpy(shift, k);
        const int it = flags[MTP_FLAG_N_ITER] + 1;
        flags[MTP_FLAG_N_ITER] = it;
        flags[MTP_FLAG_N_EMPTY] = 0;
        int st = MTP_STATE_RUNNING;
        // tol < 0 (sklearn's is never negative): a fixed number of iterations, no convergence test of either
        // kind -- for callers that time exactly max_iter iterations (bench.py)
        if (n_changed == 0.0 && tol >= 0.0)
            st = MTP_STATE_STRICT;
        else if (tot <= tol)
            st = MTP_STATE_TOL;
        else if (it >= max_iter)
            st = MTP_STATE_MAXITER;
        flags[MTP_FLAG_STATE] = st;
    }
}

int mtp_launch_mstep(double *red, const float *C_old, float *C_new, int k, int d, double tol, int max_iter,
                     int relocation_done, float *staged, float *c2, float *shift, int32_t *flags, double *red_backup,
                     hipStream_t s, MtpBatchStride bs, int flag_parity, int sums_mode) {
    const bool mfma = mtp_mfma_dim(d);
    const int rows = bs.group_r > 0 ? k : (mfma ? mtp_n_tiles(k) * 32 : k);  // grouped images: valid rows only
    MTP_REQUIRE(bs.batch == 1 || red_backup == nullptr, "mtp_lloyd_mstep: red_backup is not supported for batched restarts");
    hipLaunchKernelGGL(k_mstep_centers, dim3(rows, bs.batch), dim3(64), 2 * (size_t)d * sizeof(float), s, red, C_old,
                       C_new, k, d, relocation_done, mfma ? staged : nullptr, c2, shift, flags, bs, flag_parity);
    MTP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mstep_flags, dim3(bs.batch), dim3(1024), 0, s, red, k, d, tol, max_iter, relocation_done, shift,
                       flags, red_backup, bs, sums_mode);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ------------------------------------------------------------------------------------------
// get_modality finish: means = sums / counts, one rounding.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cluster_means(const double *sums, const double *counts, int k, int d,
                                                       float *means, int32_t *n_empty) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)k * d) {
        const int j = (int)(e / d);
        const double c = counts[j];
        if (c > 0.0) means[e] = (float)(sums[e] / c);
    }
    if (n_empty != nullptr && e < k && counts[e] == 0.0) atomicAdd(n_empty, 1);
}

extern "C" int mtp_cluster_means_f32(const double *sums, const double *counts, int k, int d, float *means,
                                     int32_t *n_empty_out, void *stream) {
    MTP_REQUIRE(sums && counts && means && k > 0 && d > 0, "mtp_cluster_means: bad arguments");
    hipStream_t s = mtp_stream(stream);
    if (n_empty_out) MTP_HIP(hipMemsetAsync(n_empty_out, 0, sizeof(int32_t), s));
    const int64_t tot = (int64_t)k * d;
    hipLaunchKernelGGL(k_cluster_means, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, sums, counts, k, d, means,
                       n_empty_out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ------------------------------------------------------------------------------------------
// Inertia and per-point distance to the assigned centre.  One point per LPR-lane group, 16 B
// per lane; each lane's group-of-four term is sklearn's ((t0+t1)+t2)+t3 in float, the groups
// and the points are then summed in double.
// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_inertia(const float *__restrict__ X, int64_t n, int d,
                                                 const float *__restrict__ C, const int32_t *__restrict__ labels,
                                                 double *inertia, float *per_point, size_t c_stride, size_t lab_stride) {
    // restarts along blockIdx.y: own centres, labels and sum (per_point only with one restart)
    C += blockIdx.y * c_stride;
    labels += blockIdx.y * lab_stride;
    if (inertia != nullptr) inertia += blockIdx.y;
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane / LPR, c0 = lane % LPR;
    const int dq = d / 4;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double local = 0.0;
    for (int64_t base = wave_id * RPW; base < n; base += n_waves * RPW) {
        const int64_t row = base + sub;
        double v = 0.0;
        if (row < n) {
            const float4 *xp = reinterpret_cast<const float4 *>(X + row * d);
            const float4 *cp = reinterpret_cast<const float4 *>(C + (size_t)labels[row] * d);
            for (int c = c0; c < dq; c += LPR) {
                const float4 a = xp[c], b = cp[c];
                const float t0 = (a.x - b.x) * (a.x - b.x), t1 = (a.y - b.y) * (a.y - b.y);
                const float t2 = (a.z - b.z) * (a.z - b.z), t3 = (a.w - b.w) * (a.w - b.w);
                v += (double)(((t0 + t1) + t2) + t3);
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (row < n && c0 == 0) {
            if (per_point) per_point[row] = (float)v;
            local += v;
        }
    }
    if (inertia == nullptr) return;
    __shared__ double part[4];
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if (lane == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(inertia, part[0] + part[1] + part[2] + part[3]);
}

// d % 4 != 0: one thread per row, sklearn's exact float order
__global__ __launch_bounds__(256) void k_inertia_scalar(const float *__restrict__ X, int64_t n, int d,
                                                        const float *__restrict__ C,
                                                        const int32_t *__restrict__ labels, double *inertia,
                                                        float *per_point, size_t c_stride, size_t lab_stride) {
    C += blockIdx.y * c_stride;
    labels += blockIdx.y * lab_stride;
    if (inertia != nullptr) inertia += blockIdx.y;
    double local = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
        const float v = euclid_sq_sklearn(X + row * d, C + (size_t)labels[row] * d, d);
        if (per_point) per_point[row] = v;
        local += (double)v;
    }
    if (inertia == nullptr) return;
    __shared__ double part[4];
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(inertia, part[0] + part[1] + part[2] + part[3]);
}

// batch > 1: restart b uses C + b * c_stride, labels + b * lab_stride, inertia[b] (per_point must be NULL)
int mtp_launch_inertia_batched(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                               float *per_point, int batch, size_t c_stride, size_t lab_stride, hipStream_t s) {
    if (n == 0 || batch == 0) return MTP_OK;
    MTP_REQUIRE(batch == 1 || per_point == nullptr, "mtp_inertia: per-point distances only for one restart");
    MTP_REQUIRE(batch >= 1 && batch <= 65535, "mtp_inertia: batch must be in [1, 65535]");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 grid((unsigned)blocks, (unsigned)batch), block(256);
    if (d % 4 != 0) {
        hipLaunchKernelGGL(k_inertia_scalar, grid, block, 0, s, X, n, d, C, labels, inertia, per_point, c_stride, lab_stride);
    } else if (d / 4 <= 8) {
        hipLaunchKernelGGL(k_inertia<8>, grid, block, 0, s, X, n, d, C, labels, inertia, per_point, c_stride, lab_stride);
    } else if (d / 4 <= 16) {
        hipLaunchKernelGGL(k_inertia<16>, grid, block, 0, s, X, n, d, C, labels, inertia, per_point, c_stride, lab_stride);
    } else {
        hipLaunchKernelGGL(k_inertia<32>, grid, block, 0, s, X, n, d, C, labels, inertia, per_point, c_stride, lab_stride);
    }
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

int mtp_launch_inertia(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                       float *per_point, hipStream_t s) {
    return mtp_launch_inertia_batched(X, n, d, C, labels, inertia, per_point, 1, 0, 0, s);
}

// sklearn's own inertia with one thread (_k_means_common.pyx:94-124, float32 input): per point the float
// _euclidean_dense_dense (groups of four, :16-43), the points added one after the other into a float.  Two launches:
// the per-point distances (one thread per row and restart), then one wave per restart walking them in row order.
__global__ __launch_bounds__(256) void k_sqdist_rows(const float *__restrict__ X, int64_t n, int d,
                                                     const float *__restrict__ C, const int32_t *__restrict__ labels,
                                                     float *per_point, size_t c_stride, size_t lab_stride) {
    C += blockIdx.y * c_stride;
    labels += blockIdx.y * lab_stride;
    per_point += (size_t)blockIdx.y * (size_t)n;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x)
        per_point[row] = euclid_sq_sklearn(X + row * d, C + (size_t)labels[row] * d, d);
}

__global__ __launch_bounds__(64) void k_chain_sum_f32(const float *__restrict__ v, int64_t n, double *out) {
    v += (size_t)blockIdx.x * (size_t)n;
    const int lane = threadIdx.x;
    float acc = 0.f;
    float nxt = lane < n ? v[lane] : 0.f;
    for (int64_t base = 0; base < n; base += 64) {
        const float cur = nxt;
        nxt = (base + 64 + lane < n) ? v[base + 64 + lane] : 0.f;
        const int cnt = (int)min((int64_t)64, n - base);
        if (cnt == 64) {
#pragma unroll
            for (int l = 0; l < 64; ++l) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur), l));
        } else {
            for (int l = 0; l < cnt; ++l) acc += __shfl(cur, l);
        }
    }
    if (lane == 0) out[blockIdx.x] = (double)acc;
}

int mtp_launch_inertia_chain_f32(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                                 float *per_point_scratch, int batch, size_t c_stride, size_t lab_stride, hipStream_t s) {
    if (batch == 0) return MTP_OK;
    MTP_REQUIRE(batch >= 1 && batch <= 65535, "mtp_inertia: batch must be in [1, 65535]");
    if (n > 0) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(k_sqdist_rows, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, s, X, n, d, C, labels,
                           per_point_scratch, c_stride, lab_stride);
        MTP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_chain_sum_f32, dim3((unsigned)batch), dim3(64), 0, s, per_point_scratch, n, inertia);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ------------------------------------------------------------------------------------------
// Column statistics / centring.  blockDim.x = d * rpb (rpb rows per pass), thread owns a column.
// ------------------------------------------------------------------------------------------
template <bool SQDEV>
__global__ __launch_bounds__(1024) void k_col_stats(const float *__restrict__ X, int64_t n, int d, int rpb,
                                                    int64_t rows_per_block, const double *__restrict__ mean,
                                                    double *out) {
    extern __shared__ __attribute__((aligned(16))) double part[];  // blockDim.x doubles
    const int c = threadIdx.x % d, rslot = threadIdx.x / d;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = min(n, r_begin + rows_per_block);
    const double mu = SQDEV ? mean[c] : 0.0;
    double acc = 0.0;
    for (int64_t r = r_begin + rslot; r < r_end; r += rpb) {
        const double v = (double)X[r * d + c];
        acc += SQDEV ? (v - mu) * (v - mu) : v;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rslot == 0) {
        for (int q = 1; q < rpb; ++q) acc += part[q * d + c];
        atomicAdd(&out[c], acc);
    }
}

static int launch_col_stats(bool sqdev, const float *X, int64_t n, int d, const double *mean, double *out,
                            hipStream_t s) {
    MTP_REQUIRE(d >= 1 && d <= 1024, "mtp_col_*: d must be in [1, 1024] (got %d)", d);
    if (n == 0) return MTP_OK;
    const int rpb = (256 / d) > 0 ? (256 / d) : 1;
    const int threads = d * rpb;
    int64_t blocks = (n + 511) / 512;
    if (blocks > 2048) blocks = 2048;
    const int64_t rows_per_block = (n + blocks - 1) / blocks;
    blocks = (n + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)threads * sizeof(double);
    if (sqdev)
        hipLaunchKernelGGL(k_col_stats<true>, dim3((unsigned)blocks), dim3(threads), lds, s, X, n, d, rpb, rows_per_block,
                           mean, out);
    else
        hipLaunchKernelGGL(k_col_stats<false>, dim3((unsigned)blocks), dim3(threads), lds, s, X, n, d, rpb,
                           rows_per_block, mean, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_col_sums_f32(const float *X, int64_t n, int d, double *col_sum, void *stream) {
    MTP_REQUIRE(X && col_sum, "mtp_col_sums: null pointer");
    return launch_col_stats(false, X, n, d, nullptr, col_sum, mtp_stream(stream));
}

extern "C" int mtp_col_sqdev_f32(const float *X, int64_t n, int d, const double *mean, double *col_sqdev,
                                 void *stream) {
    MTP_REQUIRE(X && mean && col_sqdev, "mtp_col_sqdev: null pointer");
    return launch_col_stats(true, X, n, d, mean, col_sqdev, mtp_stream(stream));
}

// ------------------------------------------------------------------------------------------
// Column sums in NumPy's float order.  np.add.reduce(X, axis=0) on a C-ordered float32 [n, d] array adds the
// rows one after the other into d float accumulators (checked against NumPy by the CPU suite through
// oracle.c's restatement), and X.mean(axis=0) / np.var(X, axis=0) are built on it (numpy/_core/_methods.py
// _mean, _var): a sum in another order moves the column means by an ulp, the centred points with them, and
// a few points per 7e4 change cluster.  One chain of n dependent float additions per column: the kernel is
// bound by the addition latency (~n x 6 clocks), not by memory.  A workgroup owns 16 columns; its first 256
// threads bring tiles of 512 rows x 16 columns to LDS (column-major, triple buffered, the next tile's loads are
// in flight while this one is walked), 16 lanes of a FIFTH wave walk them with 16-byte LDS reads -- a wave that
// does nothing else, so that the chain never waits behind the wave's share of the loads and deposits.
// SQ: the summand is (x - center[c])^2 formed in float like NumPy's x = arr - arrmean; x * x.
// ------------------------------------------------------------------------------------------
constexpr int SEQ_COLS = 16, SEQ_ROWS = 512, SEQ_PITCH = SEQ_ROWS + 4;  // LDS tile: [column][row], rows padded

// VEC: d % 4 == 0, a thread loads 16 bytes (4 columns of one row) at a time; else one float.
template <bool SQ, bool VEC>
__global__ __launch_bounds__(320) void k_col_seqsum(const float *__restrict__ X, int64_t n, int d,
                                                    const float *__restrict__ center, const float *__restrict__ init,
                                                    float *out) {
    // three tile buffers: one being walked, one complete, one being filled; the global loads of a tile are
    // issued two walks before the tile is needed (two register sets), so their latency stays off the chain
    __shared__ __attribute__((aligned(16))) float tile[3][SEQ_COLS * SEQ_PITCH];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * SEQ_COLS;
    // VEC: columns c0 + 4 q .. + 3 of rows r, r + 64, ...;  scalar: column c0 + lc of rows lr, lr + 16, ...
    const int q = tid & 3, vr = tid >> 2;
    const int lc = tid & (SEQ_COLS - 1), lr = tid >> 4;
    constexpr int PER = SEQ_ROWS * SEQ_COLS / 256;  // floats per thread and tile (32)
    float mu[4] = {0.f, 0.f, 0.f, 0.f};
    if (SQ) {
        if (VEC) {
            for (int e = 0; e < 4; ++e)
                if (c0 + 4 * q + e < d) mu[e] = center[c0 + 4 * q + e];
        } else if (c0 + lc < d) {
            mu[0] = center[c0 + lc];
        }
    }
    const bool loader = tid < 256;  // threads 256 .. 271 walk, 272 .. 319 only keep the barriers
    const int wl = tid - 256;
    auto load = [&](int64_t r0, float(&v)[PER]) {
        if (!loader) return;
        if (VEC) {
#pragma unroll
            for (int e = 0; e < PER / 4; ++e) {
                const int64_t r = r0 + vr + 64 * e;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < n && c0 + 4 * q < d) w = *reinterpret_cast<const float4 *>(X + r * d + c0 + 4 * q);
                v[4 * e] = w.x, v[4 * e + 1] = w.y, v[4 * e + 2] = w.z, v[4 * e + 3] = w.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int64_t r = r0 + lr + 16 * e;
                v[e] = (c0 + lc < d && r < n) ? X[r * d + c0 + lc] : 0.f;
            }
        }
    };
    // the summands are formed by the loading threads (all 256), so that the walk is additions only;
    // rows past the end are not walked
    auto store = [&](int buf, const float(&v)[PER]) {
        if (!loader) return;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            float w = v[e];
            if (SQ) {
                const float x = w - mu[VEC ? (e & 3) : 0];
                w = x * x;
            }
            if (VEC)
                tile[buf][(4 * q + (e & 3)) * SEQ_PITCH + vr + 64 * (e >> 2)] = w;
            else
                tile[buf][lc * SEQ_PITCH + lr + 16 * e] = w;
        }
    };
    // init: the chain continues from another block of rows (row-sharded points: the rank before this one)
    float acc = (init != nullptr && wl >= 0 && wl < SEQ_COLS && c0 + wl < d) ? init[c0 + wl] : 0.f;
    auto walk = [&](int buf, int64_t r0) {
        if (wl < 0 || wl >= SEQ_COLS) return;
        const int rows = (int)((n - r0) < SEQ_ROWS ? (n - r0) : SEQ_ROWS);
        const float *t = tile[buf] + wl * SEQ_PITCH;
        int r = 0;
        for (; r + 32 <= rows; r += 32) {
            float4 w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const float4 *>(t + r + 4 * e);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc = acc + w[e].x;
                acc = acc + w[e].y;
                acc = acc + w[e].z;
                acc = acc + w[e].w;
            }
        }
        for (; r < rows; ++r) acc = acc + t[r];
    };
    const int64_t n_tiles = (n + SEQ_ROWS - 1) / SEQ_ROWS;
    float va[PER], vb[PER];
    // tiles 0 and 1 into LDS, tile 2 into va
    load(0, va);
    store(0, va);
    load(SEQ_ROWS, va);
    store(1, va);
    load(2 * (int64_t)SEQ_ROWS, va);
    __syncthreads();
    // iteration t: walk tile t; tile t + 2 (loaded during iteration t - 1) goes to LDS; tile t + 3 is requested
    for (int64_t t = 0; t < n_tiles; t += 2) {
        store((int)((t + 2) % 3), va);
        load((t + 3) * SEQ_ROWS, vb);
        walk((int)(t % 3), t * SEQ_ROWS);
        __syncthreads();
        if (t + 1 >= n_tiles) break;
        store((int)((t + 3) % 3), vb);
        load((t + 4) * SEQ_ROWS, va);
        walk((int)((t + 1) % 3), (t + 1) * SEQ_ROWS);
        __syncthreads();
    }
    if (wl >= 0 && wl < SEQ_COLS && c0 + wl < d) out[c0 + wl] = acc;
}

extern "C" int mtp_col_seqsum_carry_f32(const float *X, int64_t n, int d, const float *center, const float *init,
                                        float *out, void *stream) {
    MTP_REQUIRE((X || n == 0) && out && n >= 0 && d >= 1, "mtp_col_seqsum: bad arguments");
    const dim3 grid((unsigned)((d + SEQ_COLS - 1) / SEQ_COLS)), block(320);
    hipStream_t s = mtp_stream(stream);
    const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    if (center != nullptr) {
        if (vec) hipLaunchKernelGGL((k_col_seqsum<true, true>), grid, block, 0, s, X, n, d, center, init, out);
        else hipLaunchKernelGGL((k_col_seqsum<true, false>), grid, block, 0, s, X, n, d, center, init, out);
    } else {
        if (vec) hipLaunchKernelGGL((k_col_seqsum<false, true>), grid, block, 0, s, X, n, d, center, init, out);
        else hipLaunchKernelGGL((k_col_seqsum<false, false>), grid, block, 0, s, X, n, d, center, init, out);
    }
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_col_seqsum_f32(const float *X, int64_t n, int d, const float *center, float *out, void *stream) {
    return mtp_col_seqsum_carry_f32(X, n, d, center, nullptr, out, stream);
}

__global__ __launch_bounds__(256) void k_sub_rowvec(float *X, int64_t total, int d, const float *__restrict__ vec) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        X[e] = X[e] - vec[e % d];
}

extern "C" int mtp_sub_rowvec_f32(float *X, int64_t n, int d, const float *vec, void *stream) {
    MTP_REQUIRE(X && vec && d > 0, "mtp_sub_rowvec: bad arguments");
    if (n == 0) return MTP_OK;
    const int64_t total = n * d;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_sub_rowvec, dim3((unsigned)blocks), dim3(256), 0, mtp_stream(stream), X, total, d, vec);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

__global__ __launch_bounds__(256) void k_row_sqnorm(const float *__restrict__ A, int64_t n, int d, float *out) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row < n) out[row] = mtp_sqnorm_einsum_order(A + row * d, d);
}

extern "C" int mtp_row_sqnorm_f32(const float *A, int64_t n, int d, float *out, void *stream) {
    MTP_REQUIRE(A && out && d > 0, "mtp_row_sqnorm: bad arguments");
    if (n == 0) return MTP_OK;
    hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mtp_stream(stream), A, n, d, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ------------------------------------------------------------------------------------------
// Empty-cluster relocation (sklearn _k_means_common.pyx:167-211).
// distances = ((X - C_old[labels])**2).sum(axis=1) is NumPy float arithmetic there (pairwise
// row sums); here the sklearn-grouped float form of k_inertia is used -- the same quantity
// within a few ulp, only its ranking matters.  The m farthest points are selected by m argmax
// passes (descending distance, ties to the lowest index); a selected point's distance is then
// set to -1 so that the next pass skips it.
// ------------------------------------------------------------------------------------------
// (value, index) maximum, ties to the lowest index; index -1 = "nothing yet"
__device__ __forceinline__ void argmax_merge(float &bv, long long &bi, float ov, long long oi) {
    if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
        bv = ov;
        bi = oi;
    }
}

__device__ __forceinline__ void argmax_block(float &bv, long long &bi, float *sv, long long *si) {
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const long long oi = __shfl_xor(bi, o);
        argmax_merge(bv, bi, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) argmax_merge(bv, bi, sv[w], si[w]);
}

// stage 1: every block scans a contiguous slice of dist (entries < 0 are already taken)
__global__ __launch_bounds__(1024) void k_argmax_partial(const float *__restrict__ dist, int64_t n, int64_t per_block,
                                                         float *pval, long long *pidx) {
    __shared__ float sv[16];
    __shared__ long long si[16];
    const int64_t b = (int64_t)blockIdx.x * per_block, e = min(n, b + per_block);
    float bv = -2.f;
    long long bi = -1;
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const float v = dist[i];
        if (v >= 0.f && (bi < 0 || v > bv)) {  // ascending i per thread: the first maximum is kept
            bv = v;
            bi = i;
        }
    }
    argmax_block(bv, bi, sv, si);
    if (threadIdx.x == 0) {
        pval[blockIdx.x] = bv;
        pidx[blockIdx.x] = bi;
    }
}

// stage 2 (one block): winner of the partials -> far_*[pass]; its distance is set to -1 so that the
// next pass skips it
__global__ __launch_bounds__(1024) void k_argmax_final(float *dist, const float *__restrict__ pval,
                                                       const long long *__restrict__ pidx, int n_part, int pass,
                                                       const float *__restrict__ X, int d,
                                                       const int32_t *__restrict__ labels, int64_t idx_offset,
                                                       float *far_val, int64_t *far_idx, int32_t *far_lab,
                                                       float *far_rows) {
    __shared__ float sv[16];
    __shared__ long long si[16];
    __shared__ long long win;
    float bv = -2.f;
    long long bi = -1;
    for (int q = threadIdx.x; q < n_part; q += blockDim.x) argmax_merge(bv, bi, pval[q], pidx[q]);
    argmax_block(bv, bi, sv, si);
    if (threadIdx.x == 0) {
        win = bi;
        far_val[pass] = bi >= 0 ? bv : -1.f;
        far_idx[pass] = bi >= 0 ? bi + idx_offset : -1;
        far_lab[pass] = bi >= 0 ? labels[bi] : -1;
        if (bi >= 0) dist[bi] = -1.f;
    }
    __syncthreads();
    const long long w = win;
    for (int c = threadIdx.x; c < d; c += blockDim.x) far_rows[(size_t)pass * d + c] = w >= 0 ? X[w * d + c] : 0.f;
}

extern "C" int mtp_far_points_f32(const float *X, int64_t n, int d, const float *C_old, const int32_t *labels, int m,
                                  int64_t idx_offset, float *far_val, int64_t *far_idx, int32_t *far_lab,
                                  float *far_rows, float *dist_scratch, void *stream) {
    MTP_REQUIRE(X && C_old && labels && far_val && far_idx && far_lab && far_rows && dist_scratch && m > 0 && d > 0,
                "mtp_far_points: bad arguments");
    hipStream_t s = mtp_stream(stream);
    {
        // sklearn's own `distances` (float32 row sums in NumPy's pairwise order): two points whose distances differ
        // in the last bit rank as they do there.  (Until round 3 the ranking used the inertia kernel's per-point
        // terms, groups of four added in double: a near-tie at the m-th place could pick the other point.)
        int64_t blocks = (n + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_far_distances, dim3((unsigned)blocks), dim3(256), 0, s, X, n, d, C_old, labels, dist_scratch);
        MTP_LAUNCH_CHECK();
    }
    // m argmax passes, each a grid-wide scan (<= 1024 slices) + a one-block merge; the slice winners
    // live behind the n distances in the caller's scratch (mtp_far_points_scratch_bytes), so searches
    // on different streams do not share anything
    char *after = reinterpret_cast<char *>(dist_scratch) + ((size_t)n * 4 + 255) / 256 * 256;
    float *pval = reinterpret_cast<float *>(after);
    long long *pidx = reinterpret_cast<long long *>(after + 1024 * 4);
    int n_part = (int)((n + 8191) / 8192);
    if (n_part > 1024) n_part = 1024;
    if (n_part < 1) n_part = 1;
    const int64_t per_block = (n + n_part - 1) / n_part;
    for (int p = 0; p < m; ++p) {
        hipLaunchKernelGGL(k_argmax_partial, dim3(n_part), dim3(1024), 0, s, dist_scratch, n, per_block, pval, pidx);
        MTP_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(1024), 0, s, dist_scratch, pval, pidx, n_part, p, X, d,
                           labels, idx_offset, far_val, far_idx, far_lab, far_rows);
        MTP_LAUNCH_CHECK();
    }
    return MTP_OK;
}

__global__ __launch_bounds__(256) void k_apply_relocation(double *red, int k, int d, int m,
                                                          const float *__restrict__ far_val,
                                                          const int32_t *__restrict__ far_lab,
                                                          const float *__restrict__ far_rows, int sums_mode) {
    // single block; sequential over the empty clusters like the reference loop (:197-211).
    // sklearn fixes `empty_clusters = where(weight_in_clusters == 0)` BEFORE the loop (:176): a cluster that
    // loses its only member to a relocation during the loop is not served.  The snapshot is kept in the
    // counts themselves: the originally empty clusters are marked -0.0 (a count that drops 1 -> 0 is +0.0).
    if (!(far_val[0] > 0.f)) return;  // np.max(distances) == 0: relocating is pointless
    double *counts = red + (size_t)k * d;
    for (int j = threadIdx.x; j < k; j += blockDim.x)
        if (counts[j] == 0.0) counts[j] = -0.0;
    __shared__ int s_new;   // the empty cluster served in this round (ascending id), k = none left
    if (threadIdx.x == 0) s_new = -1;
    __syncthreads();
    for (int i = 0; i < m; ++i) {
        if (threadIdx.x == 0) {
            int j = s_new + 1;
            while (j < k && !(counts[j] == 0.0 && signbit(counts[j]))) ++j;
            s_new = j;
        }
        __syncthreads();
        const int new_id = s_new, old_id = far_lab[i];
        if (new_id >= k || old_id < 0) break;  // uniform: no empty cluster / no far point left
        for (int c = threadIdx.x; c < d; c += blockDim.x) {
            const double x = (double)far_rows[(size_t)i * d + c];
            if (sums_mode == MTP_LLOYD_SUMS_SKLEARN_1T)  // the sums are float32 values there: subtract in float
                red[(size_t)old_id * d + c] = (double)((float)red[(size_t)old_id * d + c] - (float)x);
            else
                red[(size_t)old_id * d + c] -= x;
            red[(size_t)new_id * d + c] = x;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            counts[new_id] = 1.0;
            counts[old_id] -= 1.0;
        }
        __syncthreads();
    }
    for (int j = threadIdx.x; j < k; j += blockDim.x)  // unserved marks (fewer far points than empty clusters)
        if (counts[j] == 0.0) counts[j] = 0.0;
}

int mtp_launch_apply_relocation(double *red, int k, int d, int m, const float *far_val, const int32_t *far_lab,
                                const float *far_rows, int sums_mode, hipStream_t s) {
    MTP_REQUIRE(red && far_val && far_lab && far_rows && m > 0 && m <= k, "mtp_apply_relocation: bad arguments");
    hipLaunchKernelGGL(k_apply_relocation, dim3(1), dim3(256), 0, s, red, k, d, m, far_val, far_lab, far_rows, sums_mode);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_apply_relocation_f32(double *red, int k, int d, int m, const float *far_val, const int32_t *far_lab,
                                        const float *far_rows, void *stream) {
    return mtp_launch_apply_relocation(red, k, d, m, far_val, far_lab, far_rows, 0, mtp_stream(stream));
}


// ------------------------------------------------------------------------------------------
// Classifier targets from the cluster labels (PCCSNet/train_steps.py:357, :364-366):
//   labels_long[i] = (int64) labels[i]                       torch.from_numpy(cluster_result).long()
//   one_hot[i, c]  = (c == labels[i]) ? 1 : 0   float [n, k] gt_prob = zeros; gt_prob[range(n), cluster_result] = 1
// HBM-bound: n*k*4 bytes written once (every element, so the buffer needs no zeroing), 16 B per lane.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_label_targets(const int32_t *__restrict__ labels, int64_t n, int k,
                                                       int64_t *labels_long, float *one_hot) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (labels_long != nullptr)
        for (int64_t i = t0; i < n; i += stride) labels_long[i] = (int64_t)labels[i];
    if (one_hot == nullptr) return;
    if ((k & 3) == 0) {
        const int kq = k >> 2;
        const int64_t total = n * kq;
        float4 *out = reinterpret_cast<float4 *>(one_hot);
        for (int64_t e = t0; e < total; e += stride) {
            const int64_t i = e / kq;
            const int c = (int)(e - i * kq) * 4;
            const int lab = labels[i];
            out[e] = make_float4(lab == c ? 1.f : 0.f, lab == c + 1 ? 1.f : 0.f, lab == c + 2 ? 1.f : 0.f,
                                 lab == c + 3 ? 1.f : 0.f);
        }
    } else {
        const int64_t total = n * k;
        for (int64_t e = t0; e < total; e += stride) {
            const int64_t i = e / k;
            one_hot[e] = (labels[i] == (int)(e - i * k)) ? 1.f : 0.f;
        }
    }
}

extern "C" int mtp_label_targets_f32(const int32_t *labels, int64_t n, int k, int64_t *labels_long, float *one_hot,
                                     void *stream) {
    MTP_REQUIRE(n >= 0 && k >= 1, "mtp_label_targets: bad shape n=%lld k=%d", (long long)n, k);
    if (n == 0 || (labels_long == nullptr && one_hot == nullptr)) return MTP_OK;
    MTP_REQUIRE(labels != nullptr, "mtp_label_targets: labels is NULL");
    MTP_REQUIRE(one_hot == nullptr || (k & 3) != 0 || reinterpret_cast<uintptr_t>(one_hot) % 16 == 0,
                "mtp_label_targets: one_hot must be 16-byte aligned");
    const int64_t work = one_hot != nullptr ? ((k & 3) == 0 ? n * (k >> 2) : n * k) : n;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_label_targets, dim3((unsigned)blocks), dim3(256), 0, mtp_stream(stream), labels, n, k, labels_long,
                       one_hot);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
