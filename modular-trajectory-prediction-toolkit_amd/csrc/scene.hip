// Modality-loss soft labels on gfx950: the direct consumer of the
// cluster labels on the reference's default classifier path (PCCSNet/train_steps.py:368-371).
//
// Replaces
//   PCCSNet/utils/data_process.py:125-131  get_speed_and_angle  (cmath.polar of a float32 offset)
//   PCCSNet/utils/data_process.py:134-160  process_scene: for every sample i the samples j of the
//       same scene with ||last_i - last_j|| <= rad and a similar motion (both standing still, or
//       0.9 s_i < s_j < 1.1 s_i and a heading within 0.1 pi) -- an O(n^2)-per-scene Python loop,
//       "~120 min" in PCCSNet/README.md:102
//   PCCSNet/utils/data_process.py:172-187, :212-218  gen_prob / get_scene_data: histogram of the
//       neighbours' clusters (+ the sample's own) / (len(neighbours) + 1) -> dense [N, K] float64
//
// Arithmetic follows the reference's dtypes: positions are float32 (main.py:81 `.float()`), their
// difference and np.linalg.norm (dx*dx + dy*dy, sqrt; rounded separately, -ffp-contract=off) stay
// float32 and are compared with float32(rad) (NumPy 2 weak-scalar promotion); speed, angle and all
// their comparisons are double.  The speed is glibc's hypot restated (hypot_glibc below: the reference's bits on a
// glibc >= 2.35 host); atan2 comes from the device math library, which may differ from glibc's in the last bit:
// a neighbour can flip only if a heading difference sits within ~1e-15 relative of 0.1 pi (tests classify any
// mismatch that way).
//
// One 64-lane wave per sample, lanes over the candidates j of its scene (their 24 bytes come from
// L2: a scene's attributes are a few hundred KB).  The neighbour lists are produced as CSR in two
// passes over the pairs (count, then fill in ascending j via ballot prefix), so every list is in
// the reference's order without any sorting.
#include <cmath>

#include "mtp_internal.h"

// hypot(x, y) as glibc >= 2.35 computes it on x86-64 (sysdeps/ieee754/dbl-64/e_hypot.c, the build without
// __FP_FAST_FMA): sqrt(ax^2 + ay^2) followed by one correction step in plain double arithmetic -- every operation
// correctly rounded and none contracted (-ffp-contract=off), so the speed carries the bits of the reference's
// cmath.polar(...)[0] (Python hands the modulus to libm's hypot).  Established against cmath.polar itself on 3e5
// float32-valued pairs (0 mismatches); the device library's own hypot differs in the last bit about once in 20, which is
// enough to move a neighbour across `0.9 s_i < s_j < 1.1 s_i` when the ratio sits on the boundary (found by
// tests/fuzz_gpu_parity.py).  An older glibc (fdlibm's hypot) may differ from this in the last bit: unpinned there.
__device__ inline double hypot_glibc(double x, double y) {
    if (!(isfinite(x) && isfinite(y))) return (isinf(x) || isinf(y)) ? (double)__builtin_inf() : x + y;
    x = fabs(x);
    y = fabs(y);
    double ax = x < y ? y : x, ay = x < y ? x : y;
    constexpr double SCALE = 0x1p-600, LARGE_VAL = 0x1p+511, TINY_VAL = 0x1p-459, EPS = 0x1p-54;
    auto kernel = [](double ax, double ay) -> double {
        double h = sqrt(ax * ax + ay * ay);
        double t1, t2;
        if (h <= 2.0 * ay) {
            const double delta = h - ay;
            t1 = ax * (2.0 * delta - ax);
            t2 = (delta - 2.0 * (ax - ay)) * delta;
        } else {
            const double delta = h - ax;
            t1 = 2.0 * delta * (ax - 2.0 * ay);
            t2 = (4.0 * delta - ay) * ay + delta * delta;
        }
        h -= (t1 + t2) / (2.0 * h);
        return h;
    };
    if (ax > LARGE_VAL) {
        if (ay <= ax * EPS) return ax + ay;
        return kernel(ax * SCALE, ay * SCALE) / SCALE;
    }
    if (ay < TINY_VAL) {
        if (ax >= TINY_VAL / EPS || ax == 0.0) return ax + ay;
        return kernel(ax / SCALE, ay / SCALE) * SCALE;
    }
    if (ay <= ax * EPS) return ax + ay;
    return kernel(ax, ay);
}

__global__ __launch_bounds__(256) void k_speed_angle(const float *__restrict__ trj, int64_t n, int tlen, int t,
                                                     float *__restrict__ last, double *__restrict__ sa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = trj + (size_t)i * tlen * 2;
    const float x0 = p[0], y0 = p[1], xt = p[2 * t], yt = p[2 * t + 1];
    const float dxf = xt - x0, dyf = yt - y0;  // float32 subtraction, then complex(double, double)
    const double dx = (double)dxf, dy = (double)dyf;
    last[2 * i] = xt;
    last[2 * i + 1] = yt;
    sa[2 * i] = hypot_glibc(dx, dy);
    sa[2 * i + 1] = atan2(dy, dx);
}

__device__ __forceinline__ bool scene_pair(float xi, float yi, double si, double ai, double smin, double smax,
                                           float xj, float yj, double sj, double aj, float rad, double two_pi,
                                           double ang_thr) {
    const float dx = xi - xj, dy = yi - yj;
    const float dist = sqrtf(dx * dx + dy * dy);
    if (!(dist <= rad)) return false;
    if (sj == 0.0 && si == 0.0) return true;
    if (smin < sj && sj < smax) {
        const double d = fabs(ai - aj);
        const double o = two_pi - d;
        const double m = o < d ? o : d;  // np.min([d, 2 pi - d])
        return m < ang_thr;
    }
    return false;
}

// FILL = false: counts[i] = number of neighbours.  FILL = true: ids[offsets[i] ...] = neighbour ids.
template <bool FILL>
__global__ __launch_bounds__(256) void k_scene_neighbors(const float *__restrict__ last, const double *__restrict__ sa,
                                                         const int32_t *__restrict__ scene_of,
                                                         const int64_t *__restrict__ split_marks, int64_t n, float rad,
                                                         double two_pi, double ang_thr, int64_t *__restrict__ counts,
                                                         const int64_t *__restrict__ offsets,
                                                         int64_t *__restrict__ ids) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int sc = scene_of[i];
    const int64_t lo = split_marks[sc], hi = split_marks[sc + 1];
    const float xi = last[2 * i], yi = last[2 * i + 1];
    const double si = sa[2 * i], ai = sa[2 * i + 1];
    const double smax = si * 1.1, smin = si * 0.9;
    int64_t cnt = 0;
    const int64_t base = FILL ? offsets[i] : 0;
    for (int64_t j0 = lo; j0 < hi; j0 += 64) {
        const int64_t j = j0 + lane;
        bool ok = false;
        if (j < hi && j != i) {
            const float2 pj = *reinterpret_cast<const float2 *>(last + 2 * j);
            const double2 qj = *reinterpret_cast<const double2 *>(sa + 2 * j);
            ok = scene_pair(xi, yi, si, ai, smin, smax, pj.x, pj.y, qj.x, qj.y, rad, two_pi, ang_thr);
        }
        const unsigned long long mask = __ballot(ok);
        if (FILL && ok) ids[base + cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = j;
        cnt += __builtin_popcountll(mask);
    }
    if (!FILL && lane == 0) counts[i] = cnt;
}

extern "C" int mtp_speed_angle_f32(const float *trj, int64_t n, int tlen, int t, float *last, double *speed_angle,
                                   void *stream) {
    MTP_REQUIRE(n >= 0 && tlen >= 1 && t >= 0 && t < tlen, "mtp_speed_angle: bad shape n=%lld tlen=%d t=%d", (long long)n,
                tlen, t);
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(trj && last && speed_angle, "mtp_speed_angle: null pointer");
    hipLaunchKernelGGL(k_speed_angle, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mtp_stream(stream), trj, n, tlen, t,
                       last, speed_angle);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_scene_neighbors_f32(const float *last, const double *speed_angle, const int32_t *scene_of,
                                       const int64_t *split_marks, int64_t n, double rad, int64_t *counts,
                                       const int64_t *offsets, int64_t *ids, void *stream) {
    MTP_REQUIRE(n >= 0, "mtp_scene_neighbors: bad n");
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(last && speed_angle && scene_of && split_marks, "mtp_scene_neighbors: null pointer");
    MTP_REQUIRE((counts != nullptr) != (offsets != nullptr && ids != nullptr),
                "mtp_scene_neighbors: pass counts (count pass) or offsets + ids (fill pass)");
    MTP_REQUIRE((n + 3) / 4 <= 0x7fffffff, "mtp_scene_neighbors: n too large");
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    const double two_pi = M_PI * 2, ang_thr = M_PI * 0.1;  // np.pi * 2, np.pi * 0.1
    hipStream_t s = mtp_stream(stream);
    if (counts != nullptr)
        hipLaunchKernelGGL(k_scene_neighbors<false>, grid, block, 0, s, last, speed_angle, scene_of, split_marks, n,
                           (float)rad, two_pi, ang_thr, counts, nullptr, nullptr);
    else
        hipLaunchKernelGGL(k_scene_neighbors<true>, grid, block, 0, s, last, speed_angle, scene_of, split_marks, n,
                           (float)rad, two_pi, ang_thr, nullptr, offsets, ids);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// dense[i, c] = (#{neighbours of i in cluster c} + [c == labels[i]]) / (len(neighbours of i) + 1).
// One wave per sample; the K-bin histogram lives in LDS.  Rows are written whole (zeros included).
__global__ __launch_bounds__(256) void k_soft_labels(const int64_t *__restrict__ offsets, const int64_t *__restrict__ ids,
                                                     const int32_t *__restrict__ labels, int64_t n, int k,
                                                     double *__restrict__ dense) {
    extern __shared__ int hist_all[];  // 4 waves x k ints
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *hist = hist_all + (size_t)wave * k;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    const bool live = i < n;
    for (int c = lane; c < k; c += 64) hist[c] = 0;
    __syncthreads();
    int64_t b = 0, e = 0;
    if (live) {
        b = offsets[i];
        e = offsets[i + 1];
        // labels outside [0, k) are rejected on the host; the guard keeps a bad one out of LDS
        if (lane == 0 && (unsigned)labels[i] < (unsigned)k) atomicAdd(&hist[labels[i]], 1);
        for (int64_t q = b + lane; q < e; q += 64) {
            const int l = labels[ids[q]];
            if ((unsigned)l < (unsigned)k) atomicAdd(&hist[l], 1);
        }
    }
    __syncthreads();
    if (live) {
        const double denom = (double)(e - b + 1);
        double *row = dense + (size_t)i * k;
        for (int c = lane; c < k; c += 64) {
            const int h = hist[c];
            row[c] = h ? (double)h / denom : 0.0;
        }
    }
}

extern "C" int mtp_soft_labels_f64(const int64_t *offsets, const int64_t *ids, const int32_t *labels, int64_t n, int k,
                                   double *dense, void *stream) {
    MTP_REQUIRE(n >= 0 && k >= 1 && k <= 8192, "mtp_soft_labels: bad shape n=%lld k=%d", (long long)n, k);
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(offsets && labels && dense, "mtp_soft_labels: null pointer");
    MTP_REQUIRE((n + 3) / 4 <= 0x7fffffff, "mtp_soft_labels: n too large");
    MTP_SET_MAX_LDS_ONCE(&k_soft_labels, 4 * 8192 * 4);
    hipLaunchKernelGGL(k_soft_labels, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)4 * k * sizeof(int),
                       mtp_stream(stream), offsets, ids, labels, n, k, dense);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
