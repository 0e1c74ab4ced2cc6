// E-step of Lloyd's algorithm on gfx950: labels[i] = argmin_j (||c_j||^2 - 2 x_i.c_j).
//
// Replaces sklearn cluster/_k_means_lloyd.pyx:168-213 (called through KMeans.fit from
// PCCSNet/utils/utility.py:121).  The [N, K] distance matrix is never written: the cross term
// runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 fmaf chain in natural k
// order == what an FMA sgemm micro-kernel computes), the ||c||^2 bias, the row argmin and the
// "label changed" count are fused into the epilogue, in registers.
//
// Mapping (MFMA 32x32x2: D[i][j] += A[i][k] B[k][j], lane l feeds A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31], result reg r of lane l is D[i=(r&3)+8(r>>2)+4(l>>5)][j=l&31]):
//   A = centroid tile (i = centroid), B = points (j = point).  A lane therefore owns ONE point
//   and sees 16 centroids per accumulator: the running (min, argmin) over all K centroids lives
//   in two registers per point tile and only the two lane halves are merged at the very end.
//   Points (B) stay in registers for the whole kernel (each wave owns 32*PT points); centroids
//   (A) stream through LDS as the pre-swizzled "staged image" (mtp_internal.h), double
//   buffered, filled with global_load_lds (no VGPR round trip).
#include "assign_kernel.h"

// ------------------------------------------------------------------------------------------
// Generic E-step for any d (used by the known-answer tests and odd feature sizes): one thread
// per point, 32-centroid tiles in LDS (plain layout), the same natural-k fmaf chain per (i, j).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assign_generic(const float *__restrict__ X, int64_t n, int d,
                                                        const float *__restrict__ C, const float *__restrict__ c2,
                                                        int k, int32_t *__restrict__ labels, double *n_changed,
                                                        const int32_t *state) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 32 * d floats
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float *x = X + (row < n ? row : n - 1) * d;
    float best = __builtin_inff();
    int bidx = 0;
    for (int j0 = 0; j0 < k; j0 += 32) {
        const int jn = min(32, k - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < jn * d; e += blockDim.x) smem[e] = C[(size_t)j0 * d + e];
        __syncthreads();
        float acc[32];
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) acc[jj] = 0.f;
        for (int kk = 0; kk < d; ++kk) {
            const float xv = x[kk];
#pragma unroll
            for (int jj = 0; jj < 32; ++jj)
                if (jj < jn) acc[jj] = fmaf(xv, smem[jj * d + kk], acc[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            if (jj < jn) {
                const float dist = fmaf(-2.0f, acc[jj], c2[j0 + jj]);
                if (dist < best) {
                    best = dist;
                    bidx = j0 + jj;
                }
            }
        }
    }
    int changed = 0;
    if (row < n) {
        const int old = labels[row];
        labels[row] = bidx;
        changed = (old != bidx);
    }
    if (n_changed != nullptr) {
        for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
        if ((threadIdx.x & 63) == 0 && changed != 0) atomicAdd(n_changed, (double)changed);
    }
}

// Stage C [k, d] -> staged image + c2 (einsum order).  One block per (padded) centroid row.
__global__ __launch_bounds__(64) void k_stage_centers(const float *__restrict__ C, int k, int d, float *staged,
                                                      float *c2, const int32_t *state, MtpBatchStride bs) {
    const int j = blockIdx.x;
    int img_row = j;
    {
        const size_t b = blockIdx.y;
        C += b * bs.c0;
        if (staged != nullptr) staged = mtp_image_of(staged, bs, b, j, &img_row);
        if (c2 != nullptr) c2 += b * bs.kvec;
        if (state != nullptr) state += b * bs.flags;
    }
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    const bool valid = j < k;
    __shared__ float c2s;
    if (threadIdx.x == 0) {
        c2s = valid ? mtp_sqnorm_einsum_order(C + (size_t)j * d, d) : __builtin_inff();
        if (valid && c2 != nullptr) c2[j] = c2s;
    }
    __syncthreads();
    if (staged != nullptr) mtp_write_staged_row(staged, img_row, d, C + (size_t)(valid ? j : 0) * d, valid, c2s, threadIdx.x, 64);
}

// Grouped images: every row invalid (zeros, ||c||^2 = +inf) before the restarts' rows are staged, so that the rows
// between k and kp of a restart and the rows behind the last restart of a group never win.  One block per image row.
__global__ __launch_bounds__(64) void k_clear_images(float *images, int d, size_t image_floats) {
    float *img = images + (size_t)blockIdx.y * image_floats;
    mtp_write_staged_row(img, blockIdx.x, d, nullptr, false, 0.f, threadIdx.x, 64);
}

int mtp_launch_clear_images(float *images, int n_images, int tiles, int d, size_t image_floats, hipStream_t s) {
    hipLaunchKernelGGL(k_clear_images, dim3(tiles * 32, n_images), dim3(64), 0, s, images, d, image_floats);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

int mtp_launch_stage_centers(const float *C, int k, int d, float *staged, float *c2, const int32_t *state,
                             hipStream_t s, MtpBatchStride bs) {
    const bool mfma = mtp_mfma_dim(d);
    // grouped images: only the k valid rows (the rest was cleared once by mtp_launch_clear_images)
    const int rows = bs.group_r > 0 ? k : (mfma ? mtp_n_tiles(k) * 32 : k);
    hipLaunchKernelGGL(k_stage_centers, dim3(rows, bs.batch), dim3(64), 0, s, C, k, d, mfma ? staged : nullptr, c2,
                       state, bs);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// One launch of the MFMA E-step with WPB waves (32 points each) per workgroup.
template <int D, int WPB>
static int launch_assign_mfma_wpb(const float *X, int64_t n, const float *staged, int k, int32_t *labels,
                                  double *n_changed, const int32_t *state, hipStream_t s, MtpBatchStride bs) {
    // 8 waves x 32 points per workgroup, 2 workgroups (4 waves/SIMD) per CU: measured 4 % faster
    // than 4 waves x 64 points at 2 waves/SIMD (tools/assign_tune.hip, N=10M K=1024 D=128)
    constexpr int PT = 1, CTW = 2, WPS = 4;
    const size_t lds = 2 * (size_t)CTW * mtp_tile_floats(D) * sizeof(float);
    MTP_SET_MAX_LDS_ONCE((&k_assign_mfma<D, PT, CTW, WPB, WPS, true>), lds);
    const int64_t pts_per_block = WPB * 32 * PT;
    const int64_t blocks = (n + pts_per_block - 1) / pts_per_block;
    MTP_REQUIRE(blocks <= 0x7fffffff, "mtp_assign: n too large");
    hipLaunchKernelGGL((k_assign_mfma<D, PT, CTW, WPB, WPS, true>), dim3((unsigned)blocks, bs.batch), dim3(64 * WPB), lds, s, X, n,
                       staged, (k + 31) / 32, labels, n_changed, state, bs);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// The chip holds 2 workgroups of 8 waves per CU (68 KB of LDS, 128 VGPRs each): one ROUND = 2 x CUs workgroups.
// Workgroups of equal cost run in lockstep, and the hardware spreads a last, partial round of up to CUs
// workgroups one per CU but puts anything beyond that two per CU on some CUs: the launch time moves in steps
// of half a round (tools/assign_sweep.sh), and N = 1.25 M points (9.54 rounds on 256 CUs) pays for 10.0.
// So a single run whose tail that shortens (see the rule in launch_assign_mfma) is issued as two launches on the
// stream: the whole rounds with the 8-wave workgroups, then the remaining points with workgroups of w = ceil(remaining waves / CUs) waves, rounded up to 4, 8, 12 or 16
// (16 = a whole CU's wave slots at this register budget) -- at most one workgroup per CU, so the tail costs
// w / 16 of a round instead of 1/2 or 1.  At N = 1.25 M: 36 + 3 tile-times per SIMD instead of 36 + 4
// (2.41 against 2.47 ms; the optimum for 32-point wave tiles, 38.15 rounded up).  The labels do not depend on the split
// (every point is scored against every centroid by the same instruction sequence).
// Batched restarts (blockIdx.y) keep the single launch: their tail is restart-major and small against 20 x N.
template <int D>
static int launch_assign_mfma(const float *X, int64_t n, const float *staged, int k, int32_t *labels,
                              double *n_changed, const int32_t *state, hipStream_t s, MtpBatchStride bs) {
    if (bs.batch != 1) return launch_assign_mfma_wpb<D, 8>(X, n, staged, k, labels, n_changed, state, s, bs);
    int cus = 0;
    const int rc = mtp_device_cu_count(&cus);
    if (rc != MTP_OK) return rc;
    const int64_t per_round = (int64_t)2 * cus * 256;  // points
    const int64_t n_main = n / per_round * per_round;
    {
        // The split is only worth a second launch where it shortens the tail: left in the single launch the last,
        // partial round costs half a round (up to one workgroup per CU) or a whole one; on its own it costs w / 16.
        // N = 10 M (76.29 rounds) pays half a round either way and stays one launch (18.61 against 18.66 ms).
        const int64_t n_tail = n - n_main;
        const int64_t w = ((n_tail + 31) / 32 + cus - 1) / cus;
        const int wq = w <= 4 ? 4 : w <= 8 ? 8 : w <= 12 ? 12 : 16;
        const double plain = n_tail * 2 <= per_round ? 0.5 : 1.0;
        if (n_main > 0 && n_tail > 0 && plain - wq / 16.0 < 0.1)
            return launch_assign_mfma_wpb<D, 8>(X, n, staged, k, labels, n_changed, state, s, bs);
    }
    if (n_main > 0) {
        const int r = launch_assign_mfma_wpb<D, 8>(X, n_main, staged, k, labels, n_changed, state, s, bs);
        if (r != MTP_OK) return r;
    }
    const int64_t n_tail = n - n_main;
    if (n_tail == 0) return MTP_OK;
    const int64_t w = ((n_tail + 31) / 32 + cus - 1) / cus;  // waves per CU, 1 ... 16
    const float *Xt = X + n_main * D;
    int32_t *lt = labels + n_main;
    // whole waves per SIMD: a workgroup's waves go round the CU's 4 SIMDs, and 10 waves take as long as 12
    if (w <= 4) return launch_assign_mfma_wpb<D, 4>(Xt, n_tail, staged, k, lt, n_changed, state, s, bs);
    if (w <= 8) return launch_assign_mfma_wpb<D, 8>(Xt, n_tail, staged, k, lt, n_changed, state, s, bs);
    if (w <= 12) return launch_assign_mfma_wpb<D, 12>(Xt, n_tail, staged, k, lt, n_changed, state, s, bs);
    return launch_assign_mfma_wpb<D, 16>(Xt, n_tail, staged, k, lt, n_changed, state, s, bs);
}

template <int D>
static int launch_assign_grouped(const float *X, int64_t n, const float *images, int32_t *labels, double *n_changed,
                                 const int32_t *state, hipStream_t s, MtpBatchStride bs) {
    constexpr int WPB = 8;
    const int group_tiles = (bs.group_r * bs.kp + 31) / 32;
    MTP_REQUIRE(group_tiles <= 64 && bs.group_r <= 8, "mtp_assign: restart group too large (%d restarts, %d tiles)", bs.group_r,
                group_tiles);
    const size_t lds = 2 * (size_t)2 * mtp_tile_floats(D) * sizeof(float);
    MTP_SET_MAX_LDS_ONCE((&k_assign_mfma_grouped<D, WPB>), lds);
    const int64_t blocks = (n + WPB * 32 - 1) / (WPB * 32);
    const int n_groups = (bs.batch + bs.group_r - 1) / bs.group_r;
    MTP_REQUIRE(blocks <= 0x7fffffff && n_groups <= 65535, "mtp_assign: n too large");
    hipLaunchKernelGGL((k_assign_mfma_grouped<D, WPB>), dim3((unsigned)blocks, (unsigned)n_groups), dim3(64 * WPB), lds, s, X, n,
                       images, bs.kp, bs.group_r, group_tiles, bs.image_floats, labels, n_changed, state, bs);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// How many restarts share a centroid image (mtp_lloyd_batched_f32).  One round = 2 workgroups on every CU; a launch
// costs rounds x (tiles per group + the fixed costs of a workgroup, ~2 tiles' worth).  0: one image per restart.
int mtp_pick_restart_group(int64_t n, int k, int batch, int cus) {
    if (batch < 2) return 0;
    const int kp = (k + 7) / 8 * 8;
    const double blocks = (double)((n + 255) / 256);
    int best_r = 0;
    double best_cost = 0.0;
    for (int r = 1; r <= batch && r <= 8; ++r) {
        const int tiles = r == 1 ? (k + 31) / 32 : (r * kp + 31) / 32;
        if (tiles > 64) break;
        const int groups = (batch + r - 1) / r;
        const double rounds = blocks * groups / (2.0 * cus);
        const double cost = (rounds < 1.0 ? 1.0 : (double)(int64_t)(rounds + 0.999)) * (tiles + 2.0);
        if (best_r == 0 || cost < best_cost * 0.97) {  // a larger group must pay for itself
            best_r = r;
            best_cost = cost;
        }
    }
    return best_r <= 1 ? 0 : best_r;
}

int mtp_launch_assign(const float *X, int64_t n, int d, const float *C, const float *staged, const float *c2, int k,
                      int32_t *labels, double *n_changed, const int32_t *state, hipStream_t s, MtpBatchStride bs) {
    if (n == 0) return MTP_OK;
    if (bs.group_r > 0) {
        MTP_REQUIRE(mtp_mfma_dim(d) && staged != nullptr, "mtp_assign: grouped restarts need d in {64, 96, 128} and the images");
        switch (d) {
            case 64: return launch_assign_grouped<64>(X, n, staged, labels, n_changed, state, s, bs);
            case 96: return launch_assign_grouped<96>(X, n, staged, labels, n_changed, state, s, bs);
            default: return launch_assign_grouped<128>(X, n, staged, labels, n_changed, state, s, bs);
        }
    }
    MTP_REQUIRE(bs.batch == 1 || mtp_mfma_dim(d), "mtp_assign: batched restarts need d in {64, 96, 128} (got %d)", d);
    if (mtp_mfma_dim(d)) {
        MTP_REQUIRE(staged != nullptr, "mtp_assign: staged centroid image missing for d=%d", d);
        switch (d) {
            case 64: return launch_assign_mfma<64>(X, n, staged, k, labels, n_changed, state, s, bs);
            case 96: return launch_assign_mfma<96>(X, n, staged, k, labels, n_changed, state, s, bs);
            case 128: return launch_assign_mfma<128>(X, n, staged, k, labels, n_changed, state, s, bs);
        }
    }
    MTP_REQUIRE(d >= 1 && d <= 1024, "mtp_assign: generic kernel supports 1 <= d <= 1024 (got %d)", d);
    MTP_REQUIRE(C != nullptr && c2 != nullptr, "mtp_assign: generic kernel needs C and c2");
    const size_t lds = 32 * (size_t)d * sizeof(float);
    MTP_SET_MAX_LDS_ONCE(&k_assign_generic, 32 * 1024 * 4);
    const int64_t blocks = (n + 255) / 256;
    MTP_REQUIRE(blocks <= 0x7fffffff, "mtp_assign: n too large");
    hipLaunchKernelGGL(k_assign_generic, dim3((unsigned)blocks), dim3(256), lds, s, X, n, d, C, c2, k, labels,
                       n_changed, state);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
