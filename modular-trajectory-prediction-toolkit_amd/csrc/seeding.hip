// k-means++ seeding on gfx950, replaying scikit-learn's algorithm and random stream.
//
// Replaces sklearn cluster/_kmeans.py:174-272 (_kmeans_plusplus, unit weights) as called once per
// restart by KMeans.fit (PCCSNet/utils/utility.py:120-121 hard-codes n_init=20): at reference
// scale (N=73 568, K=200) seeding is >50 % of the CPU wall time.
//
// The host draws the random numbers (NumPy RandomState, same order as sklearn) and uploads them
// once; everything else is enqueued without a host sync, for all restarts of a batch together: per new centre
//   k_kpp_pot_blocks (MTP_POT_OPENBLAS, the default) the float32 potentials of the candidate rows that can still
//                                win, in the addition order of NumPy + OpenBLAS (sgemv_t blocks of 4096 points;
//                                k_kpp_pot_sdot for the first centre), so that rand = u * pot carries sklearn's bits
//   k_kpp_select  (1 workgroup per restart)  pick the best of the previous candidates (argmin of potentials),
//                                then locate the L new candidates: rand_l = u_l * pot,
//                                searchsorted(cumsum(closest), rand_l) via block sums + one
//                                in-block scan per candidate, clip to N-1; gather their rows
//   k_kpp_dist_mfma / k_kpp_dist_tile / k_kpp_dist   d = fp32(max(0, (-2 x.y + |x|^2) + |y|^2)) in double for the L
//                                candidates (sklearn metrics/pairwise.py:391-442,582-653: float32
//                                input is upcast, stored back as float32, clamped), min with the
//                                closest-so-far, per-block double sums
// np.cumsum (float64 in sklearn) is replaced by double block sums + an in-block scan: differences ~1e-16 relative.
// With MTP_POT_F64 the potentials are double sums too (the seeds then differ from sklearn's in about one restart in
// four at N ~ 7e4).
#include <cstdlib>
#include <cstring>

#include "mtp_internal.h"

static constexpr int KPP_PB = 256;     // points per block sum of k_kpp_dist (one per thread); the tiled kernel uses 128
static constexpr int KPP_MAX_L = 16;
__host__ __device__ inline int kpp_ltile(int l) { return l <= 8 ? 8 : 16; }   // local trials: 2 + int(ln K) <= 16 up to K ~ 1.2e6

// Restarts are batched along blockIdx.y: restart b uses the pointers below moved by b * stride
// bytes (yy is shared: it only depends on X).
template <typename T>
__host__ __device__ __forceinline__ T *kpp_at(T *p, size_t stride, int b) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(p) + stride * (size_t)b);
}
template <typename T>
__host__ __device__ __forceinline__ const T *kpp_at(const T *p, size_t stride, int b) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(p) + stride * (size_t)b);
}

static constexpr int KPP_GB = 4096;  // OpenBLAS sgemv_t block length (NBMAX, kernel/x86_64/sgemv_t_4.c)
__host__ __device__ inline int64_t kpp_pot_blocks(int64_t n) { return ((n & ~(int64_t)3) + KPP_GB - 1) / KPP_GB; }

// Which candidate rows can still hold the smallest float32 potential.  The float32 potential in OpenBLAS order is a
// chain of at most ~1100 + nb additions of non-negative terms, so it is within (1100 + nb) * 2^-24 relative of the
// exact sum; a row whose double sum exceeds the smallest double sum by more than twice that (eps, passed in) can
// neither win nor tie, and its float32 chain need not be walked.  pot64 comes from atomics (its last bits depend
// on their order), which can only move a row across the eps margin -- where both answers give the same result.
__device__ __forceinline__ bool kpp_contender(const double *pot64, int n_rows, int row, double eps) {
    double mn = pot64[0];
    for (int l = 1; l < n_rows; ++l) mn = pot64[l] < mn ? pot64[l] : mn;
    return pot64[row] * (1.0 - eps) <= mn * (1.0 + eps);
}

struct KppWs {
    double *yy;        // [n]  |y|^2 of the upcast rows (shared by all restarts)
    float *rows;       // [2][L][n] closest-distance rows (ping-pong over steps)
    double *bsum;      // [2][L][nblk] per-block sums of those rows
    double *prefix;    // [nblk] scratch: inclusive prefix of the selected row's block sums
    double *cand_x;    // [d][LTILE] candidate rows in double, feature-major: element k of candidate l at k*LTILE + l
                       // (LTILE = 8 or 16), so one scalar load brings the k-th element of a whole candidate tile
    double *cand_xx;   // [L]
    int64_t *cand;     // [L] candidate ids
    int32_t *state;    // [0] = selected candidate of the current rows
    float *pot;        // [1] current potential (float32 like sklearn's)
    double *pot64;     // [KPP_MAX_L] double sums of the current candidate rows (atomics of the distance kernel): only
                       // to tell which rows can still be the float32 argmin (kpp_contender)
    float *pot_part;   // [L][nb4k] float32 sums of the rows' 4096-point blocks in OpenBLAS order (MTP_POT_OPENBLAS)
    size_t stride;     // bytes from one restart's block to the next
    size_t bytes;      // total for `batch` restarts
};

static inline size_t kpp_align(size_t v) { return (v + 255) / 256 * 256; }

static KppWs kpp_carve(void *ws, int64_t n, int d, int l, int batch) {
    KppWs w;
    const int64_t nblk = (n + 127) / 128;  // sized for the smaller of the two block-sum granularities
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = ws ? static_cast<char *>(ws) + off : nullptr;
        off += kpp_align(bytes);
        return p;
    };
    w.yy = static_cast<double *>(take((size_t)n * 8));
    const size_t first = off;
    w.rows = static_cast<float *>(take((size_t)2 * l * n * 4));
    w.bsum = static_cast<double *>(take((size_t)2 * l * nblk * 8));
    w.prefix = static_cast<double *>(take((size_t)nblk * 8));
    // padded to the kernels' candidate tile (8 or 16 rows): k_kpp_dist_reg multiplies all rows of the tile
    w.cand_x = static_cast<double *>(take((size_t)kpp_ltile(l) * d * 8));
    w.cand_xx = static_cast<double *>(take((size_t)l * 8));
    w.cand = static_cast<int64_t *>(take((size_t)l * 8));
    w.state = static_cast<int32_t *>(take(64));
    w.pot = static_cast<float *>(take(64));
    w.pot64 = static_cast<double *>(take(KPP_MAX_L * 8));
    w.pot_part = static_cast<float *>(take((size_t)l * kpp_pot_blocks(n) * 4 + 4));
    w.stride = off - first;
    w.bytes = first + w.stride * (size_t)batch;
    return w;
}

extern "C" int64_t mtp_kpp_workspace_bytes(int64_t n, int d, int l) {
    if (n <= 0 || d <= 0 || l <= 0 || l > KPP_MAX_L) return 0;
    return (int64_t)kpp_carve(nullptr, n, d, l, 1).bytes;
}
extern "C" int64_t mtp_kpp_batched_workspace_bytes(int64_t n, int d, int l, int batch) {
    if (n <= 0 || d <= 0 || l <= 0 || l > KPP_MAX_L || batch <= 0) return 0;
    return (int64_t)kpp_carve(nullptr, n, d, l, batch).bytes;
}

// ||row||^2 of the upcast row in the double summation order of NumPy's einsum("ij,ij->i") -- what sklearn's
// _euclidean_distances_upcast runs for XX and YY (row_norms(X.astype(float64), squared=True),
// metrics/pairwise.py:391-442): x86-64 baseline SIMD, two lane accumulators (element e -> lane e % 2), product
// rounded then added, full blocks of 8 elements consumed vector 3, 2, 1, 0, the remaining vectors in order,
// l0 + l1 (established against np.einsum itself, tests/test_oracle_golden.py; restated in oracle.c).
// The order only shows where the distance cancels: a candidate that coincides with a point (duplicate rows, fixture
// G16) must get (-2 x.y + |x|^2) + |y|^2 with |x|^2 and |y|^2 the SAME bits, as sklearn does.
// `lane2` = 0 or 1: the accumulator this call walks; the caller adds the two.
__device__ inline double kpp_sqnorm_einsum_f64_lane(const float *a, int d, int lane2) {
    double l = 0.0;
    int e = 0;
    for (; e + 8 <= d; e += 8)
        for (int v = 3; v >= 0; --v) {
            const double x = (double)a[e + 2 * v + lane2];
            l = l + x * x;
        }
    for (; e < d; e += 2)
        if (e + lane2 < d) {
            const double x = (double)a[e + lane2];
            l = l + x * x;
        }
    return l;
}
// the same chain over squares that are already formed (sq[e] = double(a[e]) * double(a[e]), rounded once, as above)
__device__ inline double kpp_sqnorm_einsum_f64_lane_sq(const double *sq, int d, int lane2) {
    double l = 0.0;
    int e = 0;
    for (; e + 8 <= d; e += 8)
        for (int v = 3; v >= 0; --v) l = l + sq[e + 2 * v + lane2];
    for (; e < d; e += 2)
        if (e + lane2 < d) l = l + sq[e + lane2];
    return l;
}
__device__ inline double kpp_sqnorm_einsum_f64(const float *a, int d) {
    return kpp_sqnorm_einsum_f64_lane(a, d, 0) + kpp_sqnorm_einsum_f64_lane(a, d, 1);
}

static constexpr int KPP_PART_LDS = 2048;    // float32 block sums of the candidate rows held in LDS by k_kpp_select
static constexpr int KPP_PREFIX_LDS = 1024;  // block-sum prefixes held in LDS (one chunk of the scan) instead of global
static constexpr int KPP_NORM_LDS = 256;  // rows up to this many features are staged in LDS for the norm chains

// |y|^2 per row in double
__global__ __launch_bounds__(256) void k_kpp_rownorm(const float *__restrict__ X, int64_t n, int d, double *yy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    yy[i] = kpp_sqnorm_einsum_f64(X + i * d, d);
}

// Distances from n_cand (<= LT) candidate rows (double, in cand_x) to every point; min with the
// previous closest row (or none when has_prev == 0); block sums.  KPP_PB/256 points per thread; every
// point row is converted to double once and used for all candidates.
template <int LT>
__global__ __launch_bounds__(256) void k_kpp_dist(const float *__restrict__ X, int64_t n, int d, int n_cand,
                                                  const double *__restrict__ cand_x,
                                                  const double *__restrict__ cand_xx, const double *__restrict__ yy,
                                                  const float *rows_prev /* [L][n] */, const int32_t *sel_prev,
                                                  int has_prev, float *rows_out /* [L][n] */, double *bsum_out,
                                                  int64_t nblk, size_t bstride, int ltile, double *pot64) {
    extern __shared__ __attribute__((aligned(16))) double sx[];  // LT * d candidate rows (zero padded)
    __shared__ double part[LT][4];
    {
        const int b = blockIdx.y;
        cand_x = kpp_at(cand_x, bstride, b);
        cand_xx = kpp_at(cand_xx, bstride, b);
        rows_prev = kpp_at(rows_prev, bstride, b);
        sel_prev = kpp_at(sel_prev, bstride, b);
        rows_out = kpp_at(rows_out, bstride, b);
        bsum_out = kpp_at(bsum_out, bstride, b);
        if (pot64 != nullptr) pot64 = kpp_at(pot64, bstride, b);
    }
    for (int e = threadIdx.x; e < LT * d; e += 256) sx[e] = e < n_cand * d ? cand_x[(size_t)(e % d) * ltile + e / d] : 0.0;
    __syncthreads();
    const float *prev = has_prev ? rows_prev + (size_t)(*sel_prev) * n : nullptr;
    double acc[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) acc[l] = 0.0;
    const int64_t base = (int64_t)blockIdx.x * KPP_PB;
    for (int r = 0; r < KPP_PB / 256; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i >= n) continue;
        const float *y = X + i * d;
        double dot[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) dot[l] = 0.0;
        for (int k = 0; k < d; ++k) {
            const double yv = (double)y[k];
#pragma unroll
            for (int l = 0; l < LT; ++l) dot[l] = fma(sx[l * d + k], yv, dot[l]);
        }
        const double yyi = yy[i];
        const float old = prev ? prev[i] : __builtin_inff();
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            if (l < n_cand) {
                double dd = -2.0 * dot[l];
                dd += cand_xx[l];
                dd += yyi;
                float f = (float)dd;
                f = f > 0.f ? f : 0.f;  // np.maximum(d, 0)
                f = f < old ? f : old;  // np.minimum(closest_dist_sq, distance_to_candidates)
                rows_out[(size_t)l * n + i] = f;
                acc[l] += (double)f;
            }
        }
    }
    // block sums (fixed tree: wave shuffles, then the 4 wave partials in order)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        double v = acc[l];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) part[l][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < n_cand) {
        const int l = threadIdx.x;
        const double t = (part[l][0] + part[l][1]) + (part[l][2] + part[l][3]);
        bsum_out[(size_t)l * nblk + blockIdx.x] = t;
        if (pot64 != nullptr) atomicAdd(pot64 + l, t);
    }
}


// The same computation as a register-tiled double-precision GEMM (feature sizes that are multiples of
// 16, at most 8 local trials).  A workgroup computes 128 points x (16 * CQ) candidates -- the
// candidate tiles of 2 * CQ restarts side by side -- from LDS chunks of 16 features; a thread owns
// an 8 x CQ block of accumulators (points tx + 16 p, candidates ty * CQ + q), so every multiply-add
// costs (8 + CQ) / (8 CQ) LDS reads instead of one, X is read once for all those restarts, and the
// candidate rows once per workgroup.  Per (candidate, point) the features are still accumulated
// one fma at a time in ascending k: the same bits as k_kpp_dist.  Block sums: a thread's 8 points
// in order, then a 16-lane shuffle tree.
static constexpr int KPP_TP = 128;  // points per workgroup == points per block sum on this path
static constexpr int KPP_KC = 16;   // features per LDS chunk

template <int CQ>
__global__ __launch_bounds__(256, 2) void k_kpp_dist_tile(const float *__restrict__ X, int64_t n, int d, int n_cand,
                                                       const double *__restrict__ cand_x0,
                                                       const double *__restrict__ cand_xx0,
                                                       const double *__restrict__ yy, const float *rows_prev0,
                                                       const int32_t *sel_prev0, int has_prev, float *rows_out0,
                                                       double *bsum_out0, int64_t nblk, size_t bstride, int batch,
                                                       double *pot64_0) {
    constexpr int NC = 16 * CQ;      // candidate columns of the tile (8 per restart)
    constexpr int RB = 2 * CQ;       // restarts per workgroup
    // LDS chunks are double buffered and the next chunk's global loads are issued before this chunk's
    // FMAs (values wait in registers): one barrier per chunk, global latency behind the arithmetic
    __shared__ __attribute__((aligned(16))) double ys_[2][KPP_KC][KPP_TP + 2];
    __shared__ __attribute__((aligned(16))) double cs_[2][KPP_KC][NC];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * KPP_TP;
    const int b0 = blockIdx.y * RB;
    double acc[8][CQ];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[p][q] = 0.0;

    constexpr int XV = KPP_TP * (KPP_KC / 4) / 256;           // float4 pieces of X per thread and chunk (2)
    constexpr int CV = (RB * KPP_KC * 4 + 255) / 256;         // double2 pieces of the candidate tiles
    float4 xreg[XV];
    double2 creg[CV];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int e = tid + 256 * i;
            const int r = e / (KPP_KC / 4), q4 = e % (KPP_KC / 4);
            int64_t row = row0 + r;
            if (row >= n) row = n - 1;
            xreg[i] = *reinterpret_cast<const float4 *>(X + row * d + kc + 4 * q4);
        }
#pragma unroll
        for (int i = 0; i < CV; ++i) {
            const int e = tid + 256 * i;  // double2 index: restart rb, pair w2 of its KPP_KC x 8 chunk
            const int rb = e / (KPP_KC * 4), w2 = e % (KPP_KC * 4);
            const int b = b0 + rb;
            creg[i] = make_double2(0.0, 0.0);
            if (e < RB * KPP_KC * 4 && b < batch)
                creg[i] = *reinterpret_cast<const double2 *>(kpp_at(cand_x0, bstride, b) + (size_t)kc * 8 + 2 * w2);
        }
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int e = tid + 256 * i;
            const int r = e / (KPP_KC / 4), q4 = e % (KPP_KC / 4);
            ys_[buf][4 * q4 + 0][r] = (double)xreg[i].x;
            ys_[buf][4 * q4 + 1][r] = (double)xreg[i].y;
            ys_[buf][4 * q4 + 2][r] = (double)xreg[i].z;
            ys_[buf][4 * q4 + 3][r] = (double)xreg[i].w;
        }
#pragma unroll
        for (int i = 0; i < CV; ++i) {
            const int e = tid + 256 * i;
            const int rb = e / (KPP_KC * 4), w2 = e % (KPP_KC * 4);
            if (e < RB * KPP_KC * 4)
                *reinterpret_cast<double2 *>(&cs_[buf][w2 >> 2][rb * 8 + 2 * (w2 & 3)]) = creg[i];
        }
    };

    fetch(0);
    for (int kc = 0, buf = 0; kc < d; kc += KPP_KC, buf ^= 1) {
        deposit(buf);
        __syncthreads();  // chunk visible; every thread is past the FMAs that read the other buffer
        if (kc + KPP_KC < d) fetch(kc + KPP_KC);
        const double (*ys)[KPP_TP + 2] = ys_[buf];
        const double (*cs)[NC] = cs_[buf];
        // operands of step kk+1 are read from LDS while the 8*CQ multiply-adds of step kk issue
        double yv[8], cv[CQ];
#pragma unroll
        for (int p = 0; p < 8; ++p) yv[p] = ys[0][tx + 16 * p];
#pragma unroll
        for (int q = 0; q < CQ; ++q) cv[q] = cs[0][ty * CQ + q];
#pragma unroll
        for (int kk = 0; kk < KPP_KC; ++kk) {
            double yn[8], cn[CQ];
            if (kk + 1 < KPP_KC) {
#pragma unroll
                for (int p = 0; p < 8; ++p) yn[p] = ys[kk + 1][tx + 16 * p];
#pragma unroll
                for (int q = 0; q < CQ; ++q) cn[q] = cs[kk + 1][ty * CQ + q];
            }
            __builtin_amdgcn_sched_barrier(0);  // the reads above are in flight during the FMAs below
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int q = 0; q < CQ; ++q) acc[p][q] = fma(cv[q], yv[p], acc[p][q]);
            if (kk + 1 < KPP_KC) {
#pragma unroll
                for (int p = 0; p < 8; ++p) yv[p] = yn[p];
#pragma unroll
                for (int q = 0; q < CQ; ++q) cv[q] = cn[q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: distance, clamp, min with the closest-so-far row, store, block sums.  All loads
    // (||y||^2, the previous closest rows) are issued before the first store: rows_out may alias
    // nothing here, but the compiler cannot know and would otherwise serialise load -> store per
    // candidate.
    double yyv[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int64_t i = row0 + tx + 16 * p;
        yyv[p] = yy[i < n ? i : n - 1];
    }
    float oldv[CQ][8];
    double cxx[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        const int col = ty * CQ + q;
        const int b = b0 + (col >> 3), l = col & 7;
        const bool use = b < batch && l < n_cand;
        cxx[q] = use ? kpp_at(cand_xx0, bstride, b)[l] : 0.0;
        const float *prev = (use && has_prev)
                                ? kpp_at(rows_prev0, bstride, b) + (size_t)(*kpp_at(sel_prev0, bstride, b)) * n
                                : nullptr;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int64_t i = row0 + tx + 16 * p;
            oldv[q][p] = (prev != nullptr && i < n) ? prev[i] : __builtin_inff();
        }
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        const int col = ty * CQ + q;
        const int b = b0 + (col >> 3), l = col & 7;
        const bool use = b < batch && l < n_cand;  // uniform per 16-lane group
        double sum = 0.0;
        if (use) {
            float *rows_out = kpp_at(rows_out0, bstride, b) + (size_t)l * n;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int64_t i = row0 + tx + 16 * p;
                if (i < n) {
                    double dd = -2.0 * acc[p][q];
                    dd += cxx[q];
                    dd += yyv[p];
                    float f = (float)dd;
                    f = f > 0.f ? f : 0.f;                        // np.maximum(d, 0)
                    f = f < oldv[q][p] ? f : oldv[q][p];          // np.minimum(closest_dist_sq, distance_to_candidates)
                    rows_out[i] = f;
                    sum += (double)f;
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
        if (use && tx == 0) {
            kpp_at(bsum_out0, bstride, b)[(size_t)l * nblk + blockIdx.x] = sum;
            if (pot64_0 != nullptr) atomicAdd(kpp_at(pot64_0, bstride, b) + l, sum);
        }
    }
}

template <int CQ>
static void launch_kpp_dist_tile(const float *X, int64_t n, int d, int n_cand, const KppWs &w, const float *rows_prev,
                                 int has_prev, float *rows_out, double *bsum_out, int64_t nblk, int batch,
                                 hipStream_t s) {
    hipLaunchKernelGGL((k_kpp_dist_tile<CQ>), dim3((unsigned)nblk, (unsigned)((batch + 2 * CQ - 1) / (2 * CQ))), dim3(256),
                       0, s, X, n, d, n_cand, w.cand_x, w.cand_xx, w.yy, rows_prev, w.state, has_prev, rows_out, bsum_out,
                       nblk, w.stride, batch, w.pot64);
}


// ------------------------------------------------------------------------------------------
// The same distances on the fp64 matrix pipe.  v_mfma_f64_16x16x4_f64 takes ONE double per lane and
// operand for 2048 flop, where the register-tiled kernel above reads (8 + CQ) doubles from LDS per
// 8*CQ multiply-adds.  (Measured with tools/mfma_f64_rate on MI355X: the instruction sustains
// 47.8 TFLOP/s, plain v_fma_f64 69.5 TFLOP/s; this kernel reaches ~38 TFLOP/s, the tiled
// multiply-add kernel 24.)  A = candidates (16 rows = the 8-candidate tiles of two
// restarts), B = points (16 columns), k = features in ascending groups of four; the accumulator
// of (candidate, point) sees the same products in the same order as the fma chain (checked
// bit for bit against k_kpp_dist_tile by tests/test_gpu_batched.py).
//   workgroup: 128 points x NC candidates, 4 waves as WP (points) x WC (candidates);
//   wave: PB 16-point blocks x QB 16-candidate blocks, 4 doubles of accumulator per block and lane:
//         lane (g = lane / 16, j = lane % 16), register r  <->  candidate 4 r + g, point j.
// LDS chunks of 8 features, double buffered, rows padded so that the four k rows one operand read
// touches fall into different 128-byte bank groups.
// ------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
static constexpr int KPP_MK = 8;   // features per LDS chunk of the MFMA kernel

// Byte strides from one restart's array to the next.  The fused seeding keeps everything in one
// workspace block per restart (all strides equal); mtp_kpp_candidate_dists_f32 runs the kernel on
// caller-owned arrays with their own pitches.
struct KppStrides {
    size_t cand_x, cand_xx, rows_prev, sel, rows_out, bsum, pot64;
};
__host__ inline KppStrides kpp_uniform_strides(size_t s) { return KppStrides{s, s, s, s, s, s, s}; }

// 16-lane sums of doubles on the vector pipe (DPP row operations on the two halves of the double) in the order of the
// xor butterfly `s += shfl_xor(s, 1); s += shfl_xor(s, 2); s += shfl_xor(s, 4); s += shfl_xor(s, 8)`: after the
// steps 1 and 2 the four lanes of a quad hold the same bits (a + b == b + a), so "the lane 4 (8) away" of steps 3
// and 4 can be any lane of the other quad (half row) -- row_half_mirror and row_mirror.  ds_bpermute took a trip
// through the LDS per step and half of the double; eight sums per wave made that the longest chain of the epilogue.
template <int CTRL>
__device__ __forceinline__ double kpp_dpp_f64(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int NS>
__device__ __forceinline__ void kpp_row16_sums(double (&s)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] += kpp_dpp_f64<0xB1>(s[i]);   // quad_perm [1, 0, 3, 2]
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] += kpp_dpp_f64<0x4E>(s[i]);   // quad_perm [2, 3, 0, 1]
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] += kpp_dpp_f64<0x141>(s[i]);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] += kpp_dpp_f64<0x140>(s[i]);  // row_mirror
}

// A workgroup lives for ~30 us at any load, and what it lives through is a sequence of DEPENDENT phases, so the
// kernel is written to keep that sequence short (round 3; the arithmetic is untouched):
//  * the first two chunks of points and candidates are requested before anything else -- the indices of the
//    closest-so-far rows (scalar loads) and |y|^2 follow, all issued without waiting for one another (they used to
//    be four conditional scalar loads, each waited for, in front of the first point load);
//  * the candidates' |x|^2 go to LDS with the first chunk (they were global loads inside the epilogue, one trip per
//    restart of the column);
//  * the closest-so-far values are requested in the LAST chunk pair, when the chunk registers are free, and have
//    landed when the MFMAs are through (they were loaded restart by restart in the epilogue, each load behind the
//    stores of the restart before, since the rows may alias);
//  * the epilogue computes all values, then reduces the column's eight sums side by side with DPP.
template <int WP, int QB, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void k_kpp_dist_mfma(const float *__restrict__ X, int64_t n, int d, int n_cand,
                                                          const double *__restrict__ cand_x0,
                                                          const double *__restrict__ cand_xx0,
                                                          const double *__restrict__ yy, const float *rows_prev0,
                                                          const int32_t *sel_prev0, int has_prev, float *rows_out0,
                                                          double *bsum_out0, int64_t nblk, KppStrides st,
                                                          int batch, double *pot64_0) {
    constexpr int WC = 4 / WP;            // waves along the candidate axis
    constexpr int PB = 8 / WP;            // 16-point blocks per wave
    constexpr int NC = WC * QB * 16;      // candidate columns per workgroup
    constexpr int RB = NC / 8;            // restarts per workgroup
    constexpr int YS = KPP_TP + 16;
    constexpr int CS = NC + 16 + ((NC % 32) ? 16 : 0);
    __shared__ __attribute__((aligned(16))) double ys_[2][KPP_MK][YS];
    __shared__ __attribute__((aligned(16))) double cs_[2][KPP_MK][CS];
    __shared__ double part[WP][NC];
    // |x|^2 of the column's candidates (0 where there is none), in the order the epilogue's lanes read them: lane
    // group g holds the candidates 4 e + g of its wave's QB blocks x 2 restarts -- QB * 4 consecutive doubles
    __shared__ __attribute__((aligned(16))) double cxx_s[4][NC / 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: restart indices / pointers stay scalar
    const int j = lane & 15, g = lane >> 4;
    const int wp = wave % WP, wc = wave / WP;
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the ncol
    // workgroups that share a block of points (one per column of restarts) get ids 8 apart: same XCD, back
    // to back -- the block is fetched from HBM once and the other columns hit in that XCD's L2.  (With the
    // columns as the grid's second dimension they ran ~nblk dispatches apart and every column streamed X
    // from memory again.)
    const int ncol = (batch + RB - 1) / RB;
    const unsigned slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int64_t bx = (int64_t)(slot / ncol) * 8 + xcd;
    const int by = slot % ncol;
    if (bx >= nblk) return;
    const int64_t row0 = bx * KPP_TP;
    const int b0 = by * RB;

    constexpr int XPIECES = KPP_TP * (KPP_MK / 4);               // float4 pieces of X per chunk (256)
    static_assert(XPIECES == 256, "one float4 of X per thread and chunk");
    constexpr int CPIECES = RB * KPP_MK * 4;                     // double2 pieces of the candidate tiles
    constexpr int CV = (CPIECES + 255) / 256;
    static_assert(NC <= 256, "one thread per candidate column");

    // Global loads run TWO chunks ahead of the MFMAs (two register sets, loop unrolled by two): one
    // chunk's MFMAs are shorter than a trip to L2 / HBM.
    struct Regs {
        float4 x;
        double2 c[CV];
    };
    Regs ra, rb;
    auto fetch = [&](int kc, Regs &rg) {
        {
            const int r = tid / (KPP_MK / 4), q4 = tid % (KPP_MK / 4);
            int64_t row = row0 + r;
            if (row >= n) row = n - 1;
            rg.x = *reinterpret_cast<const float4 *>(X + row * d + kc + 4 * q4);
        }
#pragma unroll
        for (int i = 0; i < CV; ++i) {
            const int e = tid + 256 * i;  // double2 index: restart rb, pair w2 of its KPP_MK x 8 chunk
            const int rbi = e / (KPP_MK * 4), w2 = e % (KPP_MK * 4);
            const int b = b0 + rbi;
            rg.c[i] = make_double2(0.0, 0.0);
            if (e < CPIECES && b < batch)
                rg.c[i] = *reinterpret_cast<const double2 *>(kpp_at(cand_x0, st.cand_x, b) + (size_t)kc * 8 + 2 * w2);
        }
    };
    // d is a multiple of 16 (kpp_tile_path): an even number of chunks
    fetch(0, ra);
    fetch(KPP_MK, rb);

    // behind them, none waiting for another: |x|^2 of this thread's candidate column and which of the previous rows
    // is the closest-so-far one (wave-uniform: scalar loads)
    double cxv = 0.0;
    if (tid < NC) {
        const int b = b0 + (tid >> 3), l = tid & 7;
        if (b < batch && l < n_cand) cxv = kpp_at(cand_xx0, st.cand_xx, b)[l];
    }
    int selv[QB][2];
#pragma unroll
    for (int q = 0; q < QB; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) selv[q][h] = 0;
    if (has_prev) {  // uniform; the restart index is clamped so that the four loads are unconditional and go out together
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int b = b0 + (wc * QB + q) * 2 + h;
                selv[q][h] = *kpp_at(sel_prev0, st.sel, b < batch ? b : batch - 1);
            }
    }

    f64x4 acc[PB][QB];
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[p][q] = (f64x4){0.0, 0.0, 0.0, 0.0};

    auto deposit = [&](int buf, const Regs &rg) {
        {
            const int r = tid / (KPP_MK / 4), q4 = tid % (KPP_MK / 4);
            ys_[buf][4 * q4 + 0][r] = (double)rg.x.x;
            ys_[buf][4 * q4 + 1][r] = (double)rg.x.y;
            ys_[buf][4 * q4 + 2][r] = (double)rg.x.z;
            ys_[buf][4 * q4 + 3][r] = (double)rg.x.w;
        }
#pragma unroll
        for (int i = 0; i < CV; ++i) {
            const int e = tid + 256 * i;
            const int rbi = e / (KPP_MK * 4), w2 = e % (KPP_MK * 4);
            if (e < CPIECES) *reinterpret_cast<double2 *>(&cs_[buf][w2 >> 2][rbi * 8 + 2 * (w2 & 3)]) = rg.c[i];
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int s4 = 0; s4 < KPP_MK / 4; ++s4) {
            double bv[PB], av[QB];
#pragma unroll
            for (int p = 0; p < PB; ++p) bv[p] = ys_[buf][4 * s4 + g][(wp * PB + p) * 16 + j];
#pragma unroll
            for (int q = 0; q < QB; ++q) av[q] = cs_[buf][4 * s4 + g][(wc * QB + q) * 16 + j];
#pragma unroll
            for (int q = 0; q < QB; ++q)
#pragma unroll
                for (int p = 0; p < PB; ++p)
                    acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[p], acc[p][q], 0, 0, 0);
        }
    };

    if (tid < NC) cxx_s[tid & 3][(tid >> 4) * 4 + ((tid >> 3) & 1) * 2 + ((tid >> 2) & 1)] = cxv;  // published by the first barrier below
    // all chunk pairs but the last: the loads of the pair after the next are issued as the registers come free
    int kc = 0;
    for (; kc + 2 * KPP_MK < d; kc += 2 * KPP_MK) {
        deposit(0, ra);
        __syncthreads();  // chunk visible; every wave is past the MFMAs that read this buffer two chunks ago
        fetch(kc + 2 * KPP_MK, ra);
        compute(0);
        deposit(1, rb);
        __syncthreads();
        fetch(kc + 3 * KPP_MK, rb);
        compute(1);
    }
    // the last pair: instead of chunks, the closest-so-far values and |y|^2 of this lane's points are requested
    deposit(0, ra);
    __syncthreads();
    float oldv[QB][2][PB];
#pragma unroll
    for (int q = 0; q < QB; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = b0 + (wc * QB + q) * 2 + h;
            const float *prev = kpp_at(rows_prev0, st.rows_prev, b < batch ? b : batch - 1) + (size_t)selv[q][h] * n;
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int64_t i = row0 + (wp * PB + p) * 16 + j;
                oldv[q][h][p] = __builtin_inff();
                if (has_prev) oldv[q][h][p] = prev[i < n ? i : n - 1];
            }
        }
    double yyv[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int64_t i = row0 + (wp * PB + p) * 16 + j;
        yyv[p] = yy[i < n ? i : n - 1];
    }
    compute(0);
    deposit(1, rb);
    __syncthreads();
    compute(1);

    // epilogue: distance, clamp, min with the closest-so-far row, store, block sums
    double sums[QB * 4], cxxv[QB * 4];
#pragma unroll
    for (int i = 0; i < QB * 4; ++i) cxxv[i] = cxx_s[g][wc * QB * 4 + i];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        // accumulator register r of lane group g is candidate row 4 r + g of the block: registers 0, 1
        // belong to the block's first restart (candidates g, 4 + g), registers 2, 3 to its second
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = b0 + (wc * QB + q) * 2 + h;
            const bool live = b < batch;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * h + e, l = 4 * e + g;
                const bool use = live && l < n_cand;  // uniform per 16-lane group
                const double cxx = cxxv[q * 4 + h * 2 + e];
                float *rows_out = kpp_at(rows_out0, st.rows_out, live ? b : 0) + (size_t)l * n;
                double sum = 0.0;
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int64_t i = row0 + (wp * PB + p) * 16 + j;
                    const bool ok = use && i < n;
                    // -2 * dot is exact, so one fma gives the bits of sklearn's `d *= -2; d += XX` (two roundings, the
                    // first of which changes nothing); the multiply-add chain kernel keeps the two-step form
                    double dd = __builtin_fma(-2.0, acc[p][q][r], cxx);
                    dd += yyv[p];
                    float f = (float)dd;
                    f = f > 0.f ? f : 0.f;                          // np.maximum(d, 0)
                    f = f < oldv[q][h][p] ? f : oldv[q][h][p];      // np.minimum(closest_dist_sq, distance_to_candidates)
                    if (ok) rows_out[i] = f;
                    sum += ok ? (double)f : 0.0;
                }
                sums[q * 4 + h * 2 + e] = sum;
            }
        }
    }
    kpp_row16_sums<QB * 4>(sums);
    if (j == 0) {
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) part[wp][(wc * QB + q) * 16 + 8 * h + 4 * e + g] = sums[q * 4 + h * 2 + e];
    }
    __syncthreads();
    // block sums: the WP point slices of the workgroup in ascending order
    for (int col = tid; col < NC; col += 256) {
        const int b = b0 + (col >> 3), l = col & 7;
        if (b < batch && l < n_cand) {
            double t = part[0][col];
#pragma unroll
            for (int w = 1; w < WP; ++w) t += part[w][col];
            kpp_at(bsum_out0, st.bsum, b)[(size_t)l * nblk + bx] = t;
            if (pot64_0 != nullptr) atomicAdd(kpp_at(pot64_0, st.pot64, b) + l, t);
        }
    }
}

template <int WP, int QB, int MINB = 2>
static void launch_kpp_dist_mfma(const float *X, int64_t n, int d, int n_cand, const double *cand_x,
                                 const double *cand_xx, const double *yy, const float *rows_prev, const int32_t *sel_prev,
                                 int has_prev, float *rows_out, double *bsum_out, int64_t nblk, KppStrides st, int batch,
                                 double *pot64, hipStream_t s) {
    constexpr int RB = (4 / WP) * QB * 2;
    const int64_t ncol = (batch + RB - 1) / RB;
    const dim3 grid((unsigned)(((nblk + 7) / 8) * 8 * ncol));  // 1-D, XCD-aware (see the kernel)
    hipLaunchKernelGGL((k_kpp_dist_mfma<WP, QB, MINB>), grid, dim3(256), 0, s, X, n, d, n_cand, cand_x, cand_xx, yy, rows_prev,
                       sel_prev, has_prev, rows_out, bsum_out, nblk, st, batch, pot64);
}

// 4 restarts (32 candidate rows) per workgroup column: 80 VGPRs, 6 workgroups per CU, so the grid
// quantises finely (eth: 2875 workgroups on 1536 slots) and every SIMD holds waves of 6 different
// workgroups; wider columns (8, 10, 20 restarts) measured 71-86 us against 63 us per step
static void launch_kpp_dist_mfma_any(const float *X, int64_t n, int d, int n_cand, const double *cand_x,
                                     const double *cand_xx, const double *yy, const float *rows_prev,
                                     const int32_t *sel_prev, int has_prev, float *rows_out, double *bsum_out,
                                     int64_t nblk, KppStrides st, int batch, double *pot64, hipStream_t s) {
    if (batch > 2)
        launch_kpp_dist_mfma<4, 2, 6>(X, n, d, n_cand, cand_x, cand_xx, yy, rows_prev, sel_prev, has_prev, rows_out, bsum_out,
                                      nblk, st, batch, pot64, s);
    else
        launch_kpp_dist_mfma<4, 1, 6>(X, n, d, n_cand, cand_x, cand_xx, yy, rows_prev, sel_prev, has_prev, rows_out, bsum_out,
                                      nblk, st, batch, pot64, s);
}

__host__ inline bool kpp_tile_path(int d, int l) { return d % KPP_KC == 0 && l <= 8; }

static int launch_kpp_dist(const float *X, int64_t n, int d, int n_cand, const KppWs &w, const float *rows_prev,
                           int has_prev, float *rows_out, double *bsum_out, int64_t nblk, int batch, int ltile,
                           hipStream_t s) {
    const dim3 grid((unsigned)nblk, (unsigned)batch), block(256);
    if (kpp_tile_path(d, ltile == 8 ? 8 : 16)) {
        // MTP_KPP_KERNEL=fma selects the multiply-add chain kernel the MFMA kernel is verified against
        const char *sel = getenv("MTP_KPP_KERNEL");
        if (sel == nullptr || strcmp(sel, "fma") != 0) {
            launch_kpp_dist_mfma_any(X, n, d, n_cand, w.cand_x, w.cand_xx, w.yy, rows_prev, w.state, has_prev, rows_out,
                                     bsum_out, nblk, kpp_uniform_strides(w.stride), batch, w.pot64, s);
            MTP_LAUNCH_CHECK();
            return MTP_OK;
        }
        // candidates per thread: 10 covers the reference's 20 restarts in one workgroup column
        if (batch > 8)
            launch_kpp_dist_tile<5>(X, n, d, n_cand, w, rows_prev, has_prev, rows_out, bsum_out, nblk, batch, s);
        else if (batch > 2)
            launch_kpp_dist_tile<4>(X, n, d, n_cand, w, rows_prev, has_prev, rows_out, bsum_out, nblk, batch, s);
        else
            launch_kpp_dist_tile<1>(X, n, d, n_cand, w, rows_prev, has_prev, rows_out, bsum_out, nblk, batch, s);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    if (n_cand == 1)
        hipLaunchKernelGGL(k_kpp_dist<1>, grid, block, (size_t)1 * d * 8, s, X, n, d, n_cand, w.cand_x, w.cand_xx, w.yy,
                           rows_prev, w.state, has_prev, rows_out, bsum_out, nblk, w.stride, ltile, w.pot64);
    else if (n_cand <= 8)
        hipLaunchKernelGGL(k_kpp_dist<8>, grid, block, (size_t)8 * d * 8, s, X, n, d, n_cand, w.cand_x, w.cand_xx, w.yy,
                           rows_prev, w.state, has_prev, rows_out, bsum_out, nblk, w.stride, ltile, w.pot64);
    else
        hipLaunchKernelGGL(k_kpp_dist<16>, grid, block, (size_t)16 * d * 8, s, X, n, d, n_cand, w.cand_x, w.cand_xx,
                           w.yy, rows_prev, w.state, has_prev, rows_out, bsum_out, nblk, w.stride, ltile, w.pot64);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ------------------------------------------------------------------------------------------
// Float32 potentials in the float addition order of the reference stack.  sklearn forms them as
//   candidates_pot = distance_to_candidates @ sample_weight.reshape(-1, 1)   (cluster/_kmeans.py:252)
//   current_pot    = closest_dist_sq @ sample_weight                        (:216, first centre only)
// i.e. NumPy hands [L, n] @ [n, 1] to OpenBLAS sgemv_t and [1, n] @ [n] to sdot; with unit weights the
// products are exact and only the order of the float additions shows.  The draws are rand = u * pot,
// so a potential that differs in its last bit moves a draw across a cumulative-sum boundary about once
// in a few thousand draws at N ~ 7e4 -- enough to change the seeds of one restart in four (measured).
// The orders below are those of OpenBLAS 0.3.29's x86-64 kernels (sgemv_t: shared by the SkylakeX and Haswell
// kernels; sdot: the SkylakeX / AVX-512 micro-kernel only -- Haswell's sdot runs 4 x 8 lanes over n & -32, another
// order, so on an AVX2-only host the first-centre potential is unpinned; independent of the thread count), established against NumPy itself and restated in oracle/oracle.c:
//   sgemv_t: blocks of 4096 points (the last (n & 4095) - (n & 3)), y += block sum, block by block, then the
//            n & 3 trailing points as y += ((a0 + a1) + a2).  Rows in groups: fours and a last single row
//            by an 8-lane kernel (lane l adds points l, l+8, ...; a block whose length has bit 2 set gives
//            its first four points to lanes 0..3 first; then (l + (l+4)), (h0 + h1) + (h2 + h3)), a pair
//            (L & 2) by a 4-lane kernel (lane l adds l, l+4, ...; (a0 + a1) + (a2 + a3)).
//   sdot:    n1 = n & -32 points: 4 accumulators x 16 lanes over the first n1 & -64 (point 64 s + 16 a + l),
//            folded 16 -> 8 (l + (l+8)), a last chunk of 32 as 4 x 8, ((a0 + a1) + a2) + a3, the 8-lane fold;
//            the remaining points one by one in double, rounded to float at the end.
// k_kpp_pot_blocks: one wave per (4096-point block, row, restart): the block is staged through LDS with
// coalesced loads, then one lane per accumulator walks its chain.
// ------------------------------------------------------------------------------------------
// One wave stages one row's block (16 KB of LDS) and walks its chains: every (block, row, restart) is resident
// at once and the launch lasts one chain -- the kernel is latency-, not bandwidth-bound.
static constexpr int KPP_POT_TS = KPP_GB;

__global__ __launch_bounds__(64) void k_kpp_pot_blocks(const float *__restrict__ rows0, int64_t n, int n_rows,
                                                       size_t bstride, float *part0, int64_t nb,
                                                       const double *pot64_0, double eps, size_t part_stride,
                                                       int64_t part_pitch, size_t pot_stride) {
    __shared__ __attribute__((aligned(16))) float tile[KPP_POT_TS];
    const float *rows = kpp_at(rows0, bstride, blockIdx.z);
    float *part = kpp_at(part0, part_stride, blockIdx.z);  // block sums of row j at part[j * part_pitch + block]
    const int tid = threadIdx.x;
    const int64_t m_main = n & ~(int64_t)3;
    const int64_t off = (int64_t)blockIdx.x * KPP_GB;
    const int nbk = (int)((m_main - off) < KPP_GB ? (m_main - off) : KPP_GB);  // a multiple of 4
    const int row = blockIdx.y;
    // rows that cannot hold the smallest potential are skipped (k_kpp_select applies the same test)
    if (pot64_0 != nullptr && !kpp_contender(kpp_at(pot64_0, pot_stride, blockIdx.z), n_rows, row, eps)) return;
    const float *src = rows + (size_t)row * n + off;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        float4 *dst4 = reinterpret_cast<float4 *>(tile);
#pragma unroll 16
        for (int i = tid; i < nbk / 4; i += 64) dst4[i] = src4[i];
    } else {
#pragma unroll 16
        for (int i = tid; i < nbk; i += 64) tile[i] = src[i];
    }
    __syncthreads();
    if (tid >= 8) return;
    const int l8 = tid;
    const int quads = n_rows >> 2;
    const bool pair = (n_rows & 2) && row >= 4 * quads && row < 4 * quads + 2;
    const float *x = tile;
    float acc = 0.f;
    if (!pair) {
        int i = 0;
        if (nbk & 4) {
            if (l8 < 4) acc = acc + x[l8];
            i = 4;
        }
#pragma unroll 32
        for (; i + 8 <= nbk; i += 8) acc = acc + x[i + l8];
    } else if (l8 < 4) {
#pragma unroll 32
        for (int i = 0; i + 4 <= nbk; i += 4) acc = acc + x[i + l8];
    }
    const float o4 = __shfl_down(acc, 4, 8);
    const float h = pair ? acc : acc + o4;          // lanes 0..3
    const float t = h + __shfl_down(h, 1, 8);       // lanes 0 and 2: h0 + h1, h2 + h3
    const float y = t + __shfl_down(t, 2, 8);       // lane 0
    if (l8 == 0) part[(size_t)row * part_pitch + blockIdx.x] = y;
}

// the first centre's potential (one row per restart) in sdot's order; result -> part[0]
__global__ __launch_bounds__(64) void k_kpp_pot_sdot(const float *__restrict__ rows0, int64_t n, size_t bstride,
                                                     float *part0) {
    const float *x = kpp_at(rows0, bstride, blockIdx.x);
    float *part = kpp_at(part0, bstride, blockIdx.x);
    const int lane = threadIdx.x, a = lane >> 4, l = lane & 15;
    const int64_t n1 = n & ~(int64_t)31, n64 = n1 & ~(int64_t)63;
    float acc = 0.f;
    int64_t i = 0;
    for (; i + 512 <= n64; i += 512) {  // eight independent loads in flight per lane
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[i + 64 * e + lane];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = acc + v[e];
    }
    for (; i < n64; i += 64) acc = acc + x[i + lane];
    float w = acc + __shfl_down(acc, 8, 16);  // lanes l < 8 of each accumulator: 16 -> 8
    if (n64 < n1 && l < 8) w = w + x[n64 + 8 * a + l];
    const int l7 = lane & 7;
    const float c = ((__shfl(w, l7) + __shfl(w, 16 + l7)) + __shfl(w, 32 + l7)) + __shfl(w, 48 + l7);
    const float h = c + __shfl_down(c, 4, 8);
    const float t = h + __shfl_down(h, 1, 8);
    const float y = t + __shfl_down(t, 2, 8);
    if (lane == 0) {
        double dot = n1 ? (double)y : 0.0;
        for (int64_t j = n1; j < n; ++j) dot += (double)x[j];
        part[0] = (float)dot;
    }
}

// y[j] = the block sums of row j in order, then the n & 3 trailing points (what k_kpp_select does inline)
__global__ __launch_bounds__(64) void k_kpp_pot_combine(const float *__restrict__ rows, int64_t n, int n_rows,
                                                        const float *__restrict__ part, int64_t nb, float *out) {
    const int j = threadIdx.x;
    if (j >= n_rows) return;
    float y = 0.f;
    for (int64_t b = 0; b < nb; ++b) y = y + part[(size_t)j * nb + b];
    const int tail_n = (int)(n & 3);
    const float *tail = rows + (size_t)j * n + (n - tail_n);
    if (tail_n == 3) y = y + ((tail[0] + tail[1]) + tail[2]);
    else if (tail_n == 2) y = y + (tail[0] + tail[1]);
    else if (tail_n == 1) y = y + tail[0];
    out[j] = y;
}

extern "C" int64_t mtp_blas_order_rowsums_workspace_bytes(int64_t n, int l) {
    if (n <= 0 || l <= 0) return 0;
    return (int64_t)kpp_align((size_t)l * (kpp_pot_blocks(n) + 1) * 4);
}

extern "C" int mtp_blas_order_rowsums_f32(const float *rows, int l, int64_t n, float *out, void *ws, size_t ws_bytes,
                                          void *stream) {
    MTP_REQUIRE(rows && out && ws, "mtp_blas_order_rowsums: null pointer");
    MTP_REQUIRE(l >= 1 && l <= KPP_MAX_L && n >= 1, "mtp_blas_order_rowsums: bad shape l=%d n=%lld", l, (long long)n);
    if (ws_bytes < (size_t)mtp_blas_order_rowsums_workspace_bytes(n, l)) {
        mtp_set_error("mtp_blas_order_rowsums: workspace of %zu bytes, need %lld", ws_bytes,
                      (long long)mtp_blas_order_rowsums_workspace_bytes(n, l));
        return MTP_E_WORKSPACE;
    }
    hipStream_t s = mtp_stream(stream);
    float *part = static_cast<float *>(ws);
    if (l == 1) {
        hipLaunchKernelGGL(k_kpp_pot_sdot, dim3(1), dim3(64), 0, s, rows, n, (size_t)0, out);
        MTP_LAUNCH_CHECK();
        return MTP_OK;
    }
    const int64_t nb = kpp_pot_blocks(n);
    if (nb > 0) {
        hipLaunchKernelGGL(k_kpp_pot_blocks, dim3((unsigned)nb, (unsigned)l, 1), dim3(64), 0, s, rows, n, l, (size_t)0, part, nb,
                           (const double *)nullptr, 0.0, (size_t)0, nb, (size_t)0);
        MTP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_kpp_pot_combine, dim3(1), dim3(64), 0, s, rows, n, l, part, nb, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ---- the same orders for rows that are split over ranks at multiples of 4096 points (pccs_accel.seeding.
// kmeans_plusplus_sharded): every rank forms the block sums of its piece, the pieces' block sums are gathered and
// added in order.  Oracle: oracle.c orc_sgemv_t_block_sums / orc_sdot_carry (checked piecewise == whole on the CPU). ----
extern "C" int64_t mtp_blas_order_blocks(int64_t n) { return n > 0 ? kpp_pot_blocks(n) : 0; }

// part[b][j][nb] = the sum of the n & 3 trailing points of row j of restart b
__global__ void k_blas_tail_sums(const float *__restrict__ rows, int64_t n, int64_t total_rows, float *part, int64_t nb) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total_rows) return;
    const int tail_n = (int)(n & 3);
    const float *tail = rows + (size_t)r * n + (n - tail_n);
    float t = 0.f;
    if (tail_n == 3) t = (tail[0] + tail[1]) + tail[2];
    else if (tail_n == 2) t = tail[0] + tail[1];
    else if (tail_n == 1) t = tail[0];
    part[(size_t)r * (nb + 1) + nb] = t;
}

extern "C" int mtp_blas_order_block_sums_f32(const float *rows, int l, int64_t n, int batch, const double *pot64, double eps,
                                             float *part, void *stream) {
    MTP_REQUIRE(part && (rows || n == 0), "mtp_blas_order_block_sums: null pointer");
    MTP_REQUIRE(l >= 1 && l <= KPP_MAX_L && n >= 0 && batch >= 1 && batch <= 65535,
                "mtp_blas_order_block_sums: bad shape l=%d n=%lld batch=%d", l, (long long)n, batch);
    hipStream_t s = mtp_stream(stream);
    const int64_t nb = mtp_blas_order_blocks(n);
    if (nb > 0) {
        hipLaunchKernelGGL(k_kpp_pot_blocks, dim3((unsigned)nb, (unsigned)l, (unsigned)batch), dim3(64), 0, s, rows, n, l,
                           (size_t)l * n * 4, part, nb, pot64, eps, (size_t)l * (nb + 1) * 4, nb + 1, (size_t)l * 8);
        MTP_LAUNCH_CHECK();
    }
    const int64_t total_rows = (int64_t)batch * l;
    hipLaunchKernelGGL(k_blas_tail_sums, dim3((unsigned)((total_rows + 63) / 64)), dim3(64), 0, s, rows, n, total_rows, part, nb);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// out[r] = ((0 + part[r][0]) + part[r][1]) + ... in float.  One wave per row: 64 entries are fetched with one
// coalesced load (the next 64 are on their way meanwhile) and added one after the other from the lanes' registers.
__global__ __launch_bounds__(64) void k_blas_combine(const float *__restrict__ part, int64_t m, float *out) {
    const float *p = part + (size_t)blockIdx.x * m;
    const int lane = threadIdx.x;
    float y = 0.f;
    float v = lane < m ? p[lane] : 0.f;
    for (int64_t j = 0; j < m; j += 64) {
        const float vn = (j + 64 + lane < m) ? p[j + 64 + lane] : 0.f;
#pragma unroll
        for (int t = 0; t < 64; ++t)  // entries beyond m are +0: y + 0 == y
            y = y + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t));
        v = vn;
    }
    if (lane == 0) out[blockIdx.x] = y;
}

extern "C" int mtp_blas_order_combine_f32(const float *part, int64_t m, int64_t total_rows, float *out, void *stream) {
    MTP_REQUIRE(part && out && m >= 1 && total_rows >= 1 && total_rows <= 0x7fffffff, "mtp_blas_order_combine: bad arguments");
    hipLaunchKernelGGL(k_blas_combine, dim3((unsigned)total_rows), dim3(64), 0, mtp_stream(stream), part, m, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// sdot, piecewise: one wave per row; lane e = 16 a + l is accumulator a, lane l
__global__ __launch_bounds__(64) void k_blas_sdot_carry(const float *__restrict__ x0, int64_t row_stride, int64_t main_local,
                                                        int64_t left_over, float *acc0, int finish, float *out) {
    const float *x = x0 + (size_t)blockIdx.x * row_stride;
    float *accp = acc0 + (size_t)blockIdx.x * 64;
    const int lane = threadIdx.x, a = lane >> 4, l = lane & 15;
    float acc = accp[lane];
    int64_t i = 0;
    for (; i + 512 <= main_local; i += 512) {  // eight independent loads in flight per lane
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[i + 64 * e + lane];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = acc + v[e];
    }
    for (; i < main_local; i += 64) acc = acc + x[i + lane];
    accp[lane] = acc;
    if (!finish) return;
    const int64_t chunk = left_over & ~(int64_t)31;  // 0 or 32
    float w = acc + __shfl_down(acc, 8, 16);  // lanes l < 8 of each accumulator: 16 -> 8
    if (chunk && l < 8) w = w + x[main_local + 8 * a + l];
    const int l7 = lane & 7;
    const float c = ((__shfl(w, l7) + __shfl(w, 16 + l7)) + __shfl(w, 32 + l7)) + __shfl(w, 48 + l7);
    const float h = c + __shfl_down(c, 4, 8);
    const float t = h + __shfl_down(h, 1, 8);
    const float y = t + __shfl_down(t, 2, 8);
    if (lane == 0) {
        double dot = (double)y;
        for (int64_t jj = chunk; jj < left_over; ++jj) dot += (double)x[main_local + jj];
        out[blockIdx.x] = (float)dot;
    }
}

extern "C" int mtp_blas_order_sdot_carry_f32(const float *x, int64_t row_stride, int batch, int64_t main_local,
                                             int64_t left_over, float *acc, int finish, float *out, void *stream) {
    MTP_REQUIRE(acc && batch >= 1 && (x || (main_local == 0 && (!finish || left_over == 0))),
                "mtp_blas_order_sdot_carry: null pointer");
    MTP_REQUIRE(main_local >= 0 && main_local % 64 == 0 && left_over >= 0 && left_over < 64 && (!finish || out),
                "mtp_blas_order_sdot_carry: main_local must be a multiple of 64 and left_over < 64");
    hipLaunchKernelGGL(k_blas_sdot_carry, dim3((unsigned)batch), dim3(64), 0, mtp_stream(stream), x, row_stride, main_local,
                       left_over, acc, finish, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// One workgroup of 1024 threads.  (1) potentials of the n_prev candidate rows = sums of their block
// sums; best = first argmin (float32 compare like np.argmin on the float32 potentials); record the
// chosen centre.  (2) if n_new > 0: draw positions for the next n_new candidates.
template <int KPP_Q>
__global__ __launch_bounds__(1024) void k_kpp_select(const float *__restrict__ X, int64_t n, int d, int n_prev,
                                                     int n_new, const double *__restrict__ u /* [n_new] */,
                                                     const float *rows_cur, const double *bsum_cur, int64_t nblk,
                                                     double *prefix, int64_t *cand, double *cand_x, double *cand_xx,
                                                     int32_t *state, float *pot_out, int64_t *index_out,
                                                     size_t bstride, size_t u_stride /* doubles */,
                                                     size_t index_stride /* int64s */, int ltile,
                                                     const float *pot_part /* NULL: double sums */, int64_t pot_nb,
                                                     int pot_tail, double *pot64, double pot_eps,
                                                     const double *__restrict__ yy) {
    {
        const int b = blockIdx.x;
        if (pot_part != nullptr) pot_part = kpp_at(pot_part, bstride, b);
        pot64 = kpp_at(pot64, bstride, b);
        if (u != nullptr) u += u_stride * b;
        rows_cur = kpp_at(rows_cur, bstride, b);
        bsum_cur = kpp_at(bsum_cur, bstride, b);
        prefix = kpp_at(prefix, bstride, b);
        cand = kpp_at(cand, bstride, b);
        cand_x = kpp_at(cand_x, bstride, b);
        cand_xx = kpp_at(cand_xx, bstride, b);
        state = kpp_at(state, bstride, b);
        pot_out = kpp_at(pot_out, bstride, b);
        index_out += index_stride * b;
    }
    __shared__ double red[16];
    __shared__ double red_l[KPP_MAX_L][16];
    __shared__ float pots[KPP_MAX_L];
    __shared__ float s_part[KPP_PART_LDS];
    __shared__ double s_prefix[KPP_PREFIX_LDS];
    __shared__ int s_sel;
    __shared__ double s_rand[KPP_MAX_L];
    __shared__ int s_blk[KPP_MAX_L];
    __shared__ double s_carry;
    __shared__ double s_pot64[KPP_MAX_L];
    __shared__ double s_u[KPP_MAX_L];
    __shared__ int64_t s_cand[KPP_MAX_L];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // The kernel is a chain of dependent trips to memory (launch-bound: one workgroup per restart), so everything that
    // does not depend on WHICH candidate wins is requested here, at the start, side by side: the double potentials of
    // the contender test, the draws, the candidates' ids, the trailing points of the float32 sums, and -- when the
    // block sums fit one pass (nblk <= 1024) and the candidate tile is the small one -- this thread's block sum of
    // EVERY candidate row, of which (2a) keeps the winner's.  What remains dependent: the winner's points inside the
    // drawn blocks (2b), then the drawn rows of X.
    constexpr int PRE_L = 8;
    const bool pre_bs = n_new > 0 && nblk <= 1024 && n_prev <= PRE_L;
    double bsv[PRE_L];
#pragma unroll
    for (int l = 0; l < PRE_L; ++l) bsv[l] = (pre_bs && l < n_prev && tid < nblk) ? bsum_cur[(size_t)l * nblk + tid] : 0.0;
    if (tid < n_prev) {
        s_pot64[tid] = pot64[tid];
        s_cand[tid] = cand[tid];
    }
    if (tid >= 64 && tid < 64 + n_new) s_u[tid - 64] = u[tid - 64];
    float tailv[3] = {0.f, 0.f, 0.f};
    if (pot_part != nullptr && tid < n_prev && pot_tail > 0) {
        const float *tail = rows_cur + (size_t)tid * n + (n - pot_tail);
        for (int e = 0; e < pot_tail; ++e) tailv[e] = tail[e];
    }

    // (1) potentials: per candidate a thread-strided partial, a wave tree, then the 16 wave sums in order
    // (one barrier for all candidates; the additions and their order are those of a per-candidate loop)
    if (pot_part == nullptr) {
        for (int l = 0; l < n_prev; ++l) {
            double v = 0.0;
            for (int64_t b = tid; b < nblk; b += 1024) v += bsum_cur[(size_t)l * nblk + b];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) red_l[l][wave] = v;
        }
        __syncthreads();
        if (tid < n_prev) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w) t += red_l[tid][w];
            pots[tid] = (float)t;
        }
    } else {
        // the block sums come in through LDS, fetched by all threads at once: walked straight from global memory
        // the chain below paid an L1 round trip per block (2 us of this kernel's 9.5 at 18 blocks)
        const bool part_lds = (int64_t)n_prev * pot_nb <= KPP_PART_LDS;
        if (part_lds)
            for (int e = tid; e < n_prev * (int)pot_nb; e += 1024) s_part[e] = pot_part[e];
        __syncthreads();
        if (tid < n_prev) {
        // OpenBLAS order (see k_kpp_pot_blocks): y += block sum, block by block, then the n & 3 trailing points;
        // rows that cannot hold the smallest potential were not summed and are out of the argmin
        float y = __builtin_inff();
        if (n_prev == 1 || kpp_contender(s_pot64, n_prev, tid, pot_eps)) {
            y = 0.f;
            if (part_lds)
                for (int b = 0; b < (int)pot_nb; ++b) y = y + s_part[tid * (int)pot_nb + b];
            else
                for (int64_t b = 0; b < pot_nb; ++b) y = y + pot_part[(size_t)tid * pot_nb + b];
            if (pot_tail == 3) y = y + ((tailv[0] + tailv[1]) + tailv[2]);
            else if (pot_tail == 2) y = y + (tailv[0] + tailv[1]);
            else if (pot_tail == 1) y = y + tailv[0];
        }
        pots[tid] = y;
        }
    }
    __syncthreads();
    if (tid < KPP_MAX_L) pot64[tid] = 0.0;  // the next distance kernel adds its rows' block sums here
    if (tid == 0) {
        int best = 0;
        for (int l = 1; l < n_prev; ++l)
            if (pots[l] < pots[best]) best = l;
        s_sel = best;
        state[0] = best;
        pot_out[0] = pots[best];
        index_out[0] = s_cand[best];
        for (int l = 0; l < n_new; ++l) {
            s_rand[l] = s_u[l] * (double)pots[best];  // random_state.uniform(size=L) * current_pot
            s_blk[l] = 0;
        }
        s_carry = 0.0;
    }
    __syncthreads();
    if (n_new == 0) return;
    const int sel = s_sel;
    const double *bs = bsum_cur + (size_t)sel * nblk;
    const float *row = rows_cur + (size_t)sel * n;

    // (2a) inclusive prefix of the block sums, chunks of 1024 with a running carry; for every
    // candidate count the blocks whose prefix is still below its draw (= index of its block)
    for (int64_t b0 = 0; b0 < nblk; b0 += 1024) {
        const int64_t b = b0 + tid;
        double v = 0.0;
        if (pre_bs) {  // this thread's block sum of every row came in at the start: keep the winner's
#pragma unroll
            for (int l = 0; l < PRE_L; ++l) v = sel == l ? bsv[l] : v;
        } else if (b < nblk) {
            v = bs[b];
        }
        // wave inclusive scan
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (lane == 63) red[wave] = v;
        __syncthreads();
        double woff = 0.0;
        for (int w = 0; w < wave; ++w) woff += red[w];
        const double carry = s_carry;
        v += woff + carry;
        if (b < nblk) {
            if (nblk <= KPP_PREFIX_LDS) s_prefix[b] = v;  // read back below: no trip through global memory
            else prefix[b] = v;
        }
        for (int l = 0; l < n_new; ++l) {  // one LDS atomic per wave and candidate
            const unsigned long long below = __ballot(b < nblk && v < s_rand[l]);
            if (lane == 0 && below) atomicAdd(&s_blk[l], (int)__popcll(below));
        }
        __syncthreads();
        if (tid == 1023) s_carry = v;
        __syncthreads();
    }
    // (2b) one wave per candidate: scan its block's points (KPP_Q per lane) for the first cumulative
    // sum >= draw (np.searchsorted side='left'); past the end -> clip to n-1
    for (int l = wave; l < n_new; l += 16) {
        const int blk = s_blk[l];
        int64_t id = n - 1;
        if (blk < nblk) {
            const double start = blk > 0 ? (nblk <= KPP_PREFIX_LDS ? s_prefix[blk - 1] : prefix[blk - 1]) : 0.0;
            constexpr int PB = 64 * KPP_Q;  // points per block sum
            const int64_t p0 = (int64_t)blk * PB + lane * KPP_Q;
            double loc[KPP_Q];
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < KPP_Q; ++q) {
                const int64_t i = p0 + q;
                s += (i < n) ? (double)row[i] : 0.0;
                loc[q] = s;
            }
            double incl = s;
            for (int o = 1; o < 64; o <<= 1) {
                const double t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const double excl = start + (incl - s);
            int first = KPP_Q;
#pragma unroll
            for (int q = KPP_Q - 1; q >= 0; --q)
                if (excl + loc[q] >= s_rand[l]) first = q;
            const unsigned long long mask = __ballot(first < KPP_Q);
            if (mask) {
                const int wl = (int)__builtin_ctzll(mask);
                const int fq = __shfl(first, wl);
                const int64_t cid = (int64_t)blk * PB + wl * KPP_Q + fq;
                id = cid < n ? cid : n - 1;
            } else {
                // rounding put the draw past this block's end: first point of the next block
                const int64_t cid = (int64_t)(blk + 1) * PB;
                id = cid < n ? cid : n - 1;
            }
        }
        if (lane == 0) cand[l] = id;
        // gather the candidate row in double and its squared norm
        const float *x = X + id * d;
        for (int k = lane; k < d; k += 64) cand_x[(size_t)k * ltile + l] = (double)x[k];
        // |x|^2: the candidate is row `id` of X, whose norm k_kpp_rownorm has already formed in NumPy's einsum order --
        // the very bits the distance kernel adds as |y|^2 for that point, as in sklearn
        if (lane == 0) cand_xx[l] = yy[id];
    }
}

__global__ __launch_bounds__(64) void k_kpp_first(const float *__restrict__ X, int d, int64_t center_id,
                                                  const int64_t *center_ids, int64_t *cand, double *cand_x,
                                                  double *cand_xx, size_t bstride, int ltile,
                                                  const double *__restrict__ yy) {
    const int lane = threadIdx.x;
    if (center_ids != nullptr) center_id = center_ids[blockIdx.x];
    cand = kpp_at(cand, bstride, blockIdx.x);
    cand_x = kpp_at(cand_x, bstride, blockIdx.x);
    cand_xx = kpp_at(cand_xx, bstride, blockIdx.x);
    const float *x = X + center_id * d;
    for (int k = lane; k < d; k += 64) cand_x[(size_t)k * ltile] = (double)x[k];
    if (lane == 0) {
        cand_xx[0] = yy[center_id];  // the row's norm in NumPy's einsum order (k_kpp_rownorm)
        cand[0] = center_id;
    }
}

__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ X, int d, const int64_t *idx, int k,
                                                     float *out) {
    idx += (size_t)blockIdx.y * k;
    out += (size_t)blockIdx.y * k * d;
    const int j = blockIdx.x;
    const float *x = X + idx[j] * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(size_t)j * d + c] = x[c];
}

static int kpp_run(const float *X, int64_t n, int d, int k, int l, int batch, int64_t center_id,
                   const int64_t *center_ids_dev, const double *u, int pot_order, int64_t *indices_out,
                   float *centers_out, void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(pot_order == MTP_POT_F64 || pot_order == MTP_POT_OPENBLAS, "mtp_kmeans_plusplus: unknown pot_order %d",
                pot_order);
    MTP_REQUIRE(X && indices_out && centers_out && ws, "mtp_kmeans_plusplus: null pointer");
    MTP_REQUIRE(n >= 1 && d >= 1 && k >= 1 && k <= n, "mtp_kmeans_plusplus: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    MTP_REQUIRE(batch >= 1 && batch <= 65535, "mtp_kmeans_plusplus: batch must be in [1, 65535] (got %d)", batch);
    MTP_REQUIRE(l >= 1 && l <= KPP_MAX_L, "mtp_kmeans_plusplus: n_local_trials must be in [1, %d] (got %d)", KPP_MAX_L, l);
    MTP_REQUIRE(center_ids_dev != nullptr || (center_id >= 0 && center_id < n),
                "mtp_kmeans_plusplus: first centre index out of range");
    MTP_REQUIRE(k == 1 || u != nullptr, "mtp_kmeans_plusplus: uniform draws missing");
    MTP_REQUIRE((size_t)(l <= 8 ? 8 : 16) * d * 8 <= 60 * 1024, "mtp_kmeans_plusplus: d too large for the candidate tile");
    KppWs w = kpp_carve(ws, n, d, l, batch);
    if (ws_bytes < w.bytes) {
        mtp_set_error("mtp_kmeans_plusplus: workspace of %zu bytes, need %zu", ws_bytes, w.bytes);
        return MTP_E_WORKSPACE;
    }
    hipStream_t s = mtp_stream(stream);
    const bool tile = kpp_tile_path(d, l);
    const int pb = tile ? KPP_TP : KPP_PB;  // points per block sum
    const int64_t nblk = (n + pb - 1) / pb;
    MTP_REQUIRE(nblk <= 0x7fffffff, "mtp_kmeans_plusplus: n too large");
    hipLaunchKernelGGL(k_kpp_rownorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, n, d, w.yy);
    MTP_LAUNCH_CHECK();
    const int ltile = kpp_ltile(l);
    hipLaunchKernelGGL(k_kpp_first, dim3(batch), dim3(64), 0, s, X, d, center_id, center_ids_dev, w.cand, w.cand_x,
                       w.cand_xx, w.stride, ltile, w.yy);
    MTP_LAUNCH_CHECK();
    MTP_HIP(hipMemset2DAsync(w.pot64, w.stride, 0, KPP_MAX_L * 8, batch, s));
    // closest distances to the first centre: one "candidate", no previous row
    int rc = launch_kpp_dist(X, n, d, 1, w, nullptr, 0, w.rows, w.bsum, nblk, batch, ltile, s);
    if (rc != MTP_OK) return rc;
    int cur = 0, n_prev = 1;
    const size_t u_stride = (size_t)(k - 1) * l;
    const double pot_eps = (1100.0 + (double)kpp_pot_blocks(n)) * (1.0 / 8388608.0);  // see kpp_contender
    for (int c = 1; c <= k; ++c) {
        // select among the previous candidates -> centre c-1; draw the candidates for centre c
        const int n_new = c < k ? l : 0;
        float *rows_cur = w.rows + (size_t)cur * l * n;
        double *bsum_cur = w.bsum + (size_t)cur * l * nblk;
        const float *pot_part = nullptr;
        int64_t pot_nb = 0;
        int pot_tail = 0;
        if (pot_order == MTP_POT_OPENBLAS) {
            pot_part = w.pot_part;
            if (n_prev == 1) {  // the first centre: [1, n] @ [n] goes to sdot
                hipLaunchKernelGGL(k_kpp_pot_sdot, dim3(batch), dim3(64), 0, s, rows_cur, n, w.stride, w.pot_part);
                pot_nb = 1;
            } else {
                pot_nb = kpp_pot_blocks(n);
                pot_tail = (int)(n & 3);
                if (pot_nb > 0)
                    hipLaunchKernelGGL(k_kpp_pot_blocks, dim3((unsigned)pot_nb, (unsigned)n_prev, (unsigned)batch), dim3(64), 0, s,
                                       rows_cur, n, n_prev, w.stride, w.pot_part, pot_nb, w.pot64, pot_eps, w.stride, pot_nb, w.stride);
            }
            MTP_LAUNCH_CHECK();
        }
        if (tile)
            hipLaunchKernelGGL(k_kpp_select<KPP_TP / 64>, dim3(batch), dim3(1024), 0, s, X, n, d, n_prev, n_new,
                               c < k ? u + (size_t)(c - 1) * l : nullptr, rows_cur, bsum_cur, nblk, w.prefix, w.cand,
                               w.cand_x, w.cand_xx, w.state, w.pot, indices_out + (c - 1), w.stride, u_stride, (size_t)k,
                               ltile, pot_part, pot_nb, pot_tail, w.pot64, pot_eps, w.yy);
        else
            hipLaunchKernelGGL(k_kpp_select<KPP_PB / 64>, dim3(batch), dim3(1024), 0, s, X, n, d, n_prev, n_new,
                               c < k ? u + (size_t)(c - 1) * l : nullptr, rows_cur, bsum_cur, nblk, w.prefix, w.cand,
                               w.cand_x, w.cand_xx, w.state, w.pot, indices_out + (c - 1), w.stride, u_stride, (size_t)k,
                               ltile, pot_part, pot_nb, pot_tail, w.pot64, pot_eps, w.yy);
        MTP_LAUNCH_CHECK();
        if (c == k) break;
        float *rows_nxt = w.rows + (size_t)(cur ^ 1) * l * n;
        double *bsum_nxt = w.bsum + (size_t)(cur ^ 1) * l * nblk;
        rc = launch_kpp_dist(X, n, d, l, w, rows_cur, 1, rows_nxt, bsum_nxt, nblk, batch, ltile, s);
        if (rc != MTP_OK) return rc;
        cur ^= 1;
        n_prev = l;
    }
    hipLaunchKernelGGL(k_gather_rows, dim3(k, batch), dim3(128), 0, s, X, d, indices_out, k, centers_out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_kmeans_plusplus_f32(const float *X, int64_t n, int d, int k, int l, int64_t center_id,
                                       const double *u, int pot_order, int64_t *indices_out, float *centers_out,
                                       void *ws, size_t ws_bytes, void *stream) {
    return kpp_run(X, n, d, k, l, 1, center_id, nullptr, u, pot_order, indices_out, centers_out, ws, ws_bytes, stream);
}

extern "C" int mtp_kmeans_plusplus_batched_f32(const float *X, int64_t n, int d, int k, int l, int batch,
                                               const int64_t *center_ids, const double *u, int pot_order,
                                               int64_t *indices_out, float *centers_out, void *ws, size_t ws_bytes,
                                               void *stream) {
    MTP_REQUIRE(center_ids != nullptr, "mtp_kmeans_plusplus_batched: center_ids is NULL");
    return kpp_run(X, n, d, k, l, batch, 0, center_ids, u, pot_order, indices_out, centers_out, ws, ws_bytes, stream);
}


// ------------------------------------------------------------------------------------------
// Step-level pieces for seeding over row-sharded points (pccs_accel.seeding.kmeans_plusplus_sharded):
// every rank holds a block of rows; the candidate rows of a step are exchanged by the host side
// (they are L rows per restart), and each rank computes their distances to ITS points here.
// ------------------------------------------------------------------------------------------

// candidate rows (float, [batch][n_cand][d]) -> the kernels' operand form: double, feature-major
// [d][8] per restart, and |x|^2 summed exactly as k_kpp_select / k_kpp_first do (NumPy's einsum order,
// kpp_sqnorm_einsum_f64), so a one-rank run carries the fused path's bits.
__global__ __launch_bounds__(64) void k_kpp_stage_cands(const float *__restrict__ cand, int d, int n_cand,
                                                        double *cand_x, double *cand_xx) {
    const int b = blockIdx.y, l = blockIdx.x, lane = threadIdx.x;
    const float *x = cand + ((size_t)b * n_cand + l) * d;
    double *cx = cand_x + (size_t)b * d * 8;
    __shared__ double s_sq[KPP_NORM_LDS];
    const bool via_lds = d <= KPP_NORM_LDS;
    for (int k = lane; k < d; k += 64) {
        const double v = (double)x[k];
        cx[(size_t)k * 8 + l] = v;
        if (via_lds) s_sq[k] = v * v;
    }
    __builtin_amdgcn_wave_barrier();
    double xx = 0.0;
    if (lane < 2) xx = via_lds ? kpp_sqnorm_einsum_f64_lane_sq(s_sq, d, lane) : kpp_sqnorm_einsum_f64_lane(x, d, lane);
    xx += __shfl_xor(xx, 1);
    if (lane == 0) cand_xx[(size_t)b * 8 + l] = xx;
}

// potentials of the candidate rows = sums of their block sums, in k_kpp_select's order
__global__ __launch_bounds__(1024) void k_kpp_potentials(const double *__restrict__ bsum, int64_t nblk, int n_cand,
                                                         double *pot) {
    __shared__ double red_l[8][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *bs = bsum + (size_t)blockIdx.x * 8 * nblk;
    for (int l = 0; l < n_cand; ++l) {
        double v = 0.0;
        for (int64_t b = tid; b < nblk; b += 1024) v += bs[(size_t)l * nblk + b];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red_l[l][wave] = v;
    }
    __syncthreads();
    if (tid < n_cand) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red_l[tid][w];
        pot[(size_t)blockIdx.x * n_cand + tid] = t;
    }
}

struct KppCandWs {
    double *cand_x;   // [batch][d][8]
    double *cand_xx;  // [batch][8]
    double *bsum;     // [batch][8][nblk]
    int32_t *zero;    // one int = 0: "the closest-so-far row is row 0 of its array"
    size_t bytes;
};

static KppCandWs kpp_cand_carve(void *ws, int64_t n, int d, int batch) {
    KppCandWs w;
    const int64_t nblk = (n + KPP_TP - 1) / KPP_TP;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = ws ? static_cast<char *>(ws) + off : nullptr;
        off += kpp_align(bytes);
        return p;
    };
    w.cand_x = static_cast<double *>(take((size_t)batch * d * 8 * 8));
    w.cand_xx = static_cast<double *>(take((size_t)batch * 8 * 8));
    w.bsum = static_cast<double *>(take((size_t)batch * 8 * nblk * 8));
    w.zero = static_cast<int32_t *>(take(64));
    w.bytes = off;
    return w;
}

extern "C" int64_t mtp_kpp_candidate_dists_workspace_bytes(int64_t n, int d, int batch) {
    if (n <= 0 || d <= 0 || batch <= 0) return 0;
    return (int64_t)kpp_cand_carve(nullptr, n, d, batch).bytes;
}

extern "C" int mtp_kpp_row_sqnorm_f64(const float *X, int64_t n, int d, double *yy, void *stream) {
    MTP_REQUIRE(n >= 0 && d >= 1, "mtp_kpp_row_sqnorm_f64: bad shape");
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(X && yy, "mtp_kpp_row_sqnorm_f64: null pointer");
    hipLaunchKernelGGL(k_kpp_rownorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mtp_stream(stream), X, n, d, yy);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_kpp_candidate_dists_f32(const float *X, int64_t n, int d, const double *yy, const float *cand,
                                           int batch, int n_cand, const float *closest, float *rows_out,
                                           double *pot_out, void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(n >= 0 && d >= 1 && batch >= 1 && batch <= 65535, "mtp_kpp_candidate_dists: bad shape n=%lld d=%d batch=%d",
                (long long)n, d, batch);
    MTP_REQUIRE(n_cand >= 1 && n_cand <= 8, "mtp_kpp_candidate_dists: 1..8 candidates per restart (got %d)", n_cand);
    MTP_REQUIRE(d % 16 == 0, "mtp_kpp_candidate_dists: d must be a multiple of 16 (got %d)", d);
    MTP_REQUIRE(cand && pot_out && ws, "mtp_kpp_candidate_dists: null pointer");
    hipStream_t s = mtp_stream(stream);
    if (n == 0) {  // a rank without rows: zero potentials
        MTP_HIP(hipMemsetAsync(pot_out, 0, (size_t)batch * n_cand * sizeof(double), s));
        return MTP_OK;
    }
    MTP_REQUIRE(X && yy && rows_out, "mtp_kpp_candidate_dists: null pointer");
    KppCandWs w = kpp_cand_carve(ws, n, d, batch);
    if (ws_bytes < w.bytes) {
        mtp_set_error("mtp_kpp_candidate_dists: workspace of %zu bytes, need %zu", ws_bytes, w.bytes);
        return MTP_E_WORKSPACE;
    }
    const int64_t nblk = (n + KPP_TP - 1) / KPP_TP;
    MTP_REQUIRE(nblk <= 0x7fffffff, "mtp_kpp_candidate_dists: n too large");
    MTP_HIP(hipMemsetAsync(w.zero, 0, 64, s));
    // candidates beyond n_cand keep whatever the tile holds: their products are never read
    hipLaunchKernelGGL(k_kpp_stage_cands, dim3(n_cand, batch), dim3(64), 0, s, cand, d, n_cand, w.cand_x, w.cand_xx);
    MTP_LAUNCH_CHECK();
    KppStrides st;
    st.cand_x = (size_t)d * 8 * sizeof(double);
    st.cand_xx = 8 * sizeof(double);
    st.rows_prev = (size_t)n * sizeof(float);
    st.sel = 0;
    st.rows_out = (size_t)n_cand * n * sizeof(float);
    st.bsum = (size_t)8 * nblk * sizeof(double);
    st.pot64 = 0;
    launch_kpp_dist_mfma_any(X, n, d, n_cand, w.cand_x, w.cand_xx, yy, closest, w.zero, closest != nullptr ? 1 : 0, rows_out,
                             w.bsum, nblk, st, batch, nullptr, s);
    MTP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_kpp_potentials, dim3(batch), dim3(1024), 0, s, w.bsum, nblk, n_cand, pot_out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}


// ------------------------------------------------------------------------------------------
// counts[b, l] = #{ i : closest[b, 0] + ... + closest[b, i] < thresholds[b, l] }  (sums in double)
// = np.searchsorted(stable_cumsum(closest[b]), thresholds[b, l]) on this rank's rows.  With the
// thresholds shifted by the lower ranks' totals, the sum of the ranks' counts is the candidate id
// sklearn draws (cluster/_kmeans.py:234-238) -- without materialising the cumulative sum.
// Two launches: block sums (128 points each), then per restart one workgroup: prefix of the block
// sums, block of the crossing, one in-block scan per threshold (k_kpp_select's scheme).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_kpp_block_sums(const float *__restrict__ closest, int64_t n, int64_t nblk,
                                                        double *bsum) {
    __shared__ double part[2];
    const float *row = closest + (size_t)blockIdx.y * n;
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    double v = i < n ? (double)row[i] : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) bsum[(size_t)blockIdx.y * nblk + blockIdx.x] = part[0] + part[1];
}

__global__ __launch_bounds__(1024) void k_kpp_count_below(const float *__restrict__ closest, int64_t n, int64_t nblk,
                                                          const double *__restrict__ bsum, double *prefix,
                                                          const double *__restrict__ thresholds, int l_count,
                                                          int64_t *counts) {
    __shared__ double red[16];
    __shared__ double s_thr[KPP_MAX_L];
    __shared__ int s_blk[KPP_MAX_L];
    __shared__ double s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = closest + (size_t)blockIdx.x * n;
    const double *bs = bsum + (size_t)blockIdx.x * nblk;
    prefix += (size_t)blockIdx.x * nblk;
    thresholds += (size_t)blockIdx.x * l_count;
    counts += (size_t)blockIdx.x * l_count;
    if (tid < l_count) {
        s_thr[tid] = thresholds[tid];
        s_blk[tid] = 0;
    }
    if (tid == 0) s_carry = 0.0;
    __syncthreads();
    // inclusive prefix of the block sums, chunks of 1024 with a running carry; per threshold the
    // number of blocks whose prefix is still below it (= index of the block of the crossing)
    for (int64_t b0 = 0; b0 < nblk; b0 += 1024) {
        const int64_t b = b0 + tid;
        double v = b < nblk ? bs[b] : 0.0;
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (lane == 63) red[wave] = v;
        __syncthreads();
        double woff = 0.0;
        for (int w = 0; w < wave; ++w) woff += red[w];
        v += woff + s_carry;
        if (b < nblk) prefix[b] = v;
        for (int l = 0; l < l_count; ++l) {  // one LDS atomic per wave and threshold
            const unsigned long long below = __ballot(b < nblk && v < s_thr[l]);
            if (lane == 0 && below) atomicAdd(&s_blk[l], (int)__popcll(below));
        }
        __syncthreads();
        if (tid == 1023) s_carry = v;
        __syncthreads();
    }
    // one wave per threshold: two points per lane of its block, first cumulative sum >= threshold
    for (int l = wave; l < l_count; l += 16) {
        const int blk = s_blk[l];
        int64_t cnt = n;  // every cumulative sum is below the threshold
        if (blk < nblk) {
            const double start = blk > 0 ? prefix[blk - 1] : 0.0;
            const int64_t p0 = (int64_t)blk * 128 + lane * 2;
            const double a0 = p0 < n ? (double)row[p0] : 0.0;
            const double a1 = p0 + 1 < n ? (double)row[p0 + 1] : 0.0;
            const double s2 = a0 + a1;
            double incl = s2;
            for (int o = 1; o < 64; o <<= 1) {
                const double t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const double excl = start + (incl - s2);
            int first = 2;
            if (excl + s2 >= s_thr[l]) first = 1;
            if (excl + a0 >= s_thr[l]) first = 0;
            const unsigned long long mask = __ballot(first < 2);
            int64_t cid;
            if (mask) {
                const int wl = (int)__builtin_ctzll(mask);
                cid = (int64_t)blk * 128 + wl * 2 + __shfl(first, wl);
            } else {
                cid = (int64_t)(blk + 1) * 128;  // rounding put the threshold past this block's end
            }
            cnt = cid < n ? cid : n;
        }
        if (lane == 0) counts[l] = cnt;
    }
}

extern "C" int64_t mtp_kpp_count_below_workspace_bytes(int64_t n, int batch) {
    if (n <= 0 || batch <= 0) return 0;
    const int64_t nblk = (n + 127) / 128;
    return (int64_t)(2 * kpp_align((size_t)batch * nblk * sizeof(double)));
}

extern "C" int mtp_kpp_count_below_f32(const float *closest, int64_t n, int batch, const double *thresholds, int l,
                                       int64_t *counts, void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(n >= 0 && batch >= 1 && batch <= 65535, "mtp_kpp_count_below: bad shape n=%lld batch=%d", (long long)n, batch);
    MTP_REQUIRE(l >= 1 && l <= KPP_MAX_L, "mtp_kpp_count_below: 1..%d thresholds per restart (got %d)", KPP_MAX_L, l);
    MTP_REQUIRE(thresholds && counts, "mtp_kpp_count_below: null pointer");
    hipStream_t s = mtp_stream(stream);
    if (n == 0) {
        MTP_HIP(hipMemsetAsync(counts, 0, (size_t)batch * l * sizeof(int64_t), s));
        return MTP_OK;
    }
    MTP_REQUIRE(closest && ws, "mtp_kpp_count_below: null pointer");
    const int64_t nblk = (n + 127) / 128;
    MTP_REQUIRE(nblk <= 0x7fffffff, "mtp_kpp_count_below: n too large");
    const size_t need = (size_t)mtp_kpp_count_below_workspace_bytes(n, batch);
    if (ws_bytes < need) {
        mtp_set_error("mtp_kpp_count_below: workspace of %zu bytes, need %zu", ws_bytes, need);
        return MTP_E_WORKSPACE;
    }
    double *bsum = static_cast<double *>(ws);
    double *prefix = reinterpret_cast<double *>(static_cast<char *>(ws) + kpp_align((size_t)batch * nblk * sizeof(double)));
    hipLaunchKernelGGL(k_kpp_block_sums, dim3((unsigned)nblk, (unsigned)batch), dim3(128), 0, s, closest, n, nblk, bsum);
    MTP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_kpp_count_below, dim3(batch), dim3(1024), 0, s, closest, n, nblk, bsum, prefix, thresholds, l, counts);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
