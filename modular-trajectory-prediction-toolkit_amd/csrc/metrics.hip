// Best-of-K ADE / FDE on gfx950.
//
// Replaces, for a whole batch of agents in one launch,
//   PCCSNet/utils/metrics.py:4-23        displacement_error / final_displacement_error
//   PCCSNet/train_steps.py:219-232       per-sample ADE/FDE, three selection modes, running mean
//   StimulusVerification/verify_predictions.py:238-247  (float64, first best_of of 200 candidates)
//
// HBM-bound: every candidate trajectory (t*2 values) is read exactly once; the per-candidate
// (ADE, FDE) pair goes through LDS only, the min-over-K is an 8-lane lexicographic
// (value, index) reduction so the first minimum wins like torch.argmin / np.argmin.
// Arithmetic per candidate follows metrics.py: d = gt - pred; d*d; x^2 + y^2; sqrt (correctly
// rounded); sum over t in time order; "/ pred_len" as a true division.  Built with
// -ffp-contract=off so no product is fused into the add.
#include "mtp_internal.h"

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
    typedef float4 type;
};
template <>
struct Vec4<double> {
    typedef double4 type;
};

template <typename T>
__device__ __forceinline__ T tsqrt(T v);
// sqrtf / sqrt are correctly rounded here (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is NOT -- it maps to the native
// approximate instruction unless OCML_BASIC_ROUNDED_OPERATIONS is defined.
template <>
__device__ __forceinline__ float tsqrt<float>(float v) {
    return sqrtf(v);
}
template <>
__device__ __forceinline__ double tsqrt<double>(double v) {
    return sqrt(v);
}

// One candidate: returns (sum_t dist_t, dist_T).  VEC = true when t is even and the
// trajectories are 4-element aligned, so two time steps travel per load.  DISP = true: `p` holds
// the network's per-step displacements and the positions are their running sum in time order,
// p_j = d_j + p_{j-1} in T arithmetic -- transform_trajectory (PCCSNet/utils/data_process.py:111-122)
// fused into the scoring pass, so the decoded trajectories are never written.
template <typename T, bool VEC, bool DISP>
__device__ __forceinline__ void candidate_dists(const T *__restrict__ p, const T *__restrict__ g, int t, T &sum,
                                                T &last) {
    T s = 0, l = 0;
    T cx = 0, cy = 0;
    if (VEC) {
        typedef typename Vec4<T>::type V;
        const V *pv = reinterpret_cast<const V *>(p);
        const V *gv = reinterpret_cast<const V *>(g);
        const int half = t >> 1;
#pragma unroll 6
        for (int q = 0; q < half; ++q) {
            V a = pv[q];
            const V b = gv[q];
            if (DISP) {
                a.x = a.x + cx;
                a.y = a.y + cy;
                a.z = a.z + a.x;
                a.w = a.w + a.y;
                cx = a.z;
                cy = a.w;
            }
            const T dx0 = b.x - a.x, dy0 = b.y - a.y, dx1 = b.z - a.z, dy1 = b.w - a.w;
            const T d0 = tsqrt<T>(dx0 * dx0 + dy0 * dy0);
            l = tsqrt<T>(dx1 * dx1 + dy1 * dy1);
            s = (s + d0) + l;
        }
    } else {
        for (int q = 0; q < t; ++q) {
            T ax = p[2 * q], ay = p[2 * q + 1];
            if (DISP) {
                ax = ax + cx;
                ay = ay + cy;
                cx = ax;
                cy = ay;
            }
            const T dx = g[2 * q] - ax, dy = g[2 * q + 1] - ay;
            l = tsqrt<T>(dx * dx + dy * dy);
            s = s + l;
        }
    }
    sum = s;
    last = l;
}

// The same arithmetic on a candidate whose HALF step pairs are already in registers (the software-pipelined path of
// k_best_of_k: the loads of the NEXT candidate are in flight while this one is scored).
template <typename T, bool DISP, int HALF>
__device__ __forceinline__ void candidate_dists_regs(const typename Vec4<T>::type (&pv)[HALF],
                                                     const typename Vec4<T>::type *__restrict__ gv, T &sum, T &last) {
    typedef typename Vec4<T>::type V;
    T s = 0, l = 0;
    T cx = 0, cy = 0;
#pragma unroll
    for (int q = 0; q < HALF; ++q) {
        V a = pv[q];
        const V b = gv[q];
        if (DISP) {
            a.x = a.x + cx;
            a.y = a.y + cy;
            a.z = a.z + a.x;
            a.w = a.w + a.y;
            cx = a.z;
            cy = a.w;
        }
        const T dx0 = b.x - a.x, dy0 = b.y - a.y, dx1 = b.z - a.z, dy1 = b.w - a.w;
        const T d0 = tsqrt<T>(dx0 * dx0 + dy0 * dy0);
        l = tsqrt<T>(dx1 * dx1 + dy1 * dy1);
        s = (s + d0) + l;
    }
    sum = s;
    last = l;
}

template <typename T>
__device__ __forceinline__ void lexmin(T &v, int &i, T ov, int oi) {
    if (ov < v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// grid-stride over groups of `apb` agents; 256 threads.  order (NULL or [n, ostride] int32):
// candidate c of agent a is pred[a, order[a, c]] -- the re-ranked candidate list of
// verify_predictions.py:277-280 scored without materialising the re-ordered tensor; the
// reported indices are positions c in that list.
template <typename T, bool VEC, bool DISP, int HALF = 0>
__global__ __launch_bounds__(256) void k_best_of_k(const T *__restrict__ pred, const T *__restrict__ gt, int64_t n,
                                                   int k, int kstride, int t, T ade_div, int mode, int apb,
                                                   const int32_t *__restrict__ order, int ostride,
                                                   T *min_ade, T *min_fde, int32_t *idx, int32_t *idx2,
                                                   double *sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s_ade = reinterpret_cast<T *>(smem_raw);
    T *s_fde = s_ade + (size_t)apb * k;
    const int tid = threadIdx.x;
    const int64_t n_groups = (n + apb - 1) / apb;
    const size_t tstride = (size_t)t * 2;
    double acc_a = 0.0, acc_f = 0.0;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t a0 = grp * apb;
        const int na = (int)min((int64_t)apb, n - a0);
        // phase 1: per-candidate (ADE, FDE) -> LDS
        if constexpr (HALF > 0) {
            // t = 2 * HALF known at compile time (the reference predicts 12 steps): a candidate is HALF 16-byte (float)
            // loads, all issued at once, and the candidate 256 places on is requested BEFORE this one is scored -- two
            // candidates per lane in flight, so the memory pipe never runs dry while the square roots are refined
            typedef typename Vec4<T>::type V;
            const int total = na * k;
            auto cand_ptr = [&](int c) -> const V * {
                const int a = c / k, cand = c - a * k;
                const int src = order ? order[(size_t)(a0 + a) * ostride + cand] : cand;
                return reinterpret_cast<const V *>(pred + ((size_t)(a0 + a) * kstride + src) * tstride);
            };
            auto score = [&](int c, const V (&buf)[HALF]) {
                const int a = c / k;
                T s, l;
                candidate_dists_regs<T, DISP, HALF>(buf, reinterpret_cast<const V *>(gt + (size_t)(a0 + a) * tstride), s, l);
                s_ade[c] = s / ade_div;
                s_fde[c] = l;
            };
            V b0[HALF], b1[HALF];
            int c = tid;
            if (c < total) {
                const V *p = cand_ptr(c);
#pragma unroll
                for (int q = 0; q < HALF; ++q) b0[q] = p[q];
            }
            while (c < total) {
                if (c + 256 < total) {
                    const V *p = cand_ptr(c + 256);
#pragma unroll
                    for (int q = 0; q < HALF; ++q) b1[q] = p[q];
                }
                score(c, b0);
                c += 256;
                if (c >= total) break;
                if (c + 256 < total) {
                    const V *p = cand_ptr(c + 256);
#pragma unroll
                    for (int q = 0; q < HALF; ++q) b0[q] = p[q];
                }
                score(c, b1);
                c += 256;
            }
        } else
        for (int c = tid; c < na * k; c += 256) {
            const int a = c / k, cand = c - a * k;
            const int src = order ? order[(size_t)(a0 + a) * ostride + cand] : cand;
            const T *p = pred + ((size_t)(a0 + a) * kstride + src) * tstride;
            const T *g = gt + (size_t)(a0 + a) * tstride;
            T s, l;
            candidate_dists<T, VEC, DISP>(p, g, t, s, l);
            s_ade[c] = s / ade_div;
            s_fde[c] = l;
        }
        __syncthreads();
        // phase 2: 8 lanes per agent, lexicographic (value, index) minimum
        for (int a = tid >> 3; a < na; a += 32) {
            const int sub = tid & 7;
            T va = (T)INFINITY, vf = (T)INFINITY;
            int ia = 0x7fffffff, jf = 0x7fffffff;
            for (int cand = sub; cand < k; cand += 8) {
                const T ea = s_ade[a * k + cand], ef = s_fde[a * k + cand];
                if (ea < va || ia == 0x7fffffff) { va = ea; ia = cand; }
                if (ef < vf || jf == 0x7fffffff) { vf = ef; jf = cand; }
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const T ova = __shfl_xor(va, o), ovf = __shfl_xor(vf, o);
                const int oia = __shfl_xor(ia, o), ojf = __shfl_xor(jf, o);
                lexmin<T>(va, ia, ova, oia);
                lexmin<T>(vf, jf, ovf, ojf);
            }
            if (sub == 0) {
                T ra, rf;
                int i1, i2;
                if (mode == 0) { ra = va; rf = s_fde[a * k + ia]; i1 = ia; i2 = ia; }
                else if (mode == 1) { ra = s_ade[a * k + jf]; rf = vf; i1 = jf; i2 = jf; }
                else { ra = va; rf = vf; i1 = ia; i2 = jf; }
                const int64_t ag = a0 + a;
                if (min_ade) min_ade[ag] = ra;
                if (min_fde) min_fde[ag] = rf;
                if (idx) idx[ag] = i1;
                if (idx2) idx2[ag] = i2;
                acc_a += (double)ra;
                acc_f += (double)rf;
            }
        }
        __syncthreads();
    }
    if (sums != nullptr) {
        __shared__ double pa[4], pf[4];
        for (int o = 32; o > 0; o >>= 1) {
            acc_a += __shfl_xor(acc_a, o);
            acc_f += __shfl_xor(acc_f, o);
        }
        if ((tid & 63) == 0) {
            pa[tid >> 6] = acc_a;
            pf[tid >> 6] = acc_f;
        }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&sums[0], pa[0] + pa[1] + pa[2] + pa[3]);
            atomicAdd(&sums[1], pf[0] + pf[1] + pf[2] + pf[3]);
        }
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_ade_fde(const T *__restrict__ pred, const T *__restrict__ gt, int64_t n, int k,
                                                 int kstride, int t, T ade_div, T *ade, T *fde) {
    const size_t tstride = (size_t)t * 2;
    const int64_t total = n * k;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = c / k;
        const int cand = (int)(c - a * k);
        T s, l;
        candidate_dists<T, VEC, false>(pred + ((size_t)a * kstride + cand) * tstride, gt + (size_t)a * tstride, t, s, l);
        ade[c] = s / ade_div;
        fde[c] = l;
    }
}

template <typename T>
static int best_of_k_impl(const T *pred, const T *gt, int64_t n, int k, int kstride, int t, T ade_div, int mode,
                          bool disp, const int32_t *order, int ostride, T *min_ade, T *min_fde, int32_t *idx,
                          int32_t *idx2, double *sums, void *stream) {
    MTP_REQUIRE(n >= 0 && k >= 1 && kstride >= k && t >= 1, "mtp_best_of_k: bad shape n=%lld k=%d kstride=%d t=%d",
                (long long)n, k, kstride, t);
    MTP_REQUIRE(mode >= 0 && mode <= 2, "mtp_best_of_k: mode must be 0, 1 or 2 (got %d)", mode);
    MTP_REQUIRE(order == nullptr || ostride >= k, "mtp_best_of_k: order stride %d < k=%d", ostride, k);
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(pred && gt, "mtp_best_of_k: null input");
    const size_t per_agent = 2 * (size_t)k * sizeof(T);
    MTP_REQUIRE(per_agent <= 64 * 1024, "mtp_best_of_k: k=%d too large for the LDS candidate buffer", k);
    int apb = (int)((48 * 1024) / per_agent);
    if (apb > 32) apb = 32;
    if (apb < 1) apb = 1;
    const size_t lds = (size_t)apb * per_agent;
    const int64_t n_groups = (n + apb - 1) / apb;
    const bool vec = (t % 2 == 0) && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(gt)) % (4 * sizeof(T)) == 0);
    // The software-pipelined instantiation (float, t = 12: the reference's pred_len) runs one workgroup per group of agents:
    // 111 registers let a CU hold 4 of them, and handing the groups out one by one keeps every CU fed to the end
    // (1.64 ms as a persistent grid of 2048, 1.60 ms this way at 1M x 100 x 12).  The other instantiations keep the
    // persistent grid (two atomics per workgroup on the two sums).
    // Short candidate lists (the re-ranked first 20: 2.5 candidates per lane and group) have nothing to pipeline and
    // measured 12 % slower this way: they keep the plain loop.
    const bool piped = vec && t == 12 && sizeof(T) == 4 && order == nullptr && (int64_t)apb * k >= 4 * 256;
    const int64_t gmax = piped ? 0x7fffffff : 256 * 8;
    const unsigned grid = (unsigned)(n_groups < gmax ? n_groups : gmax);
    hipStream_t s = mtp_stream(stream);
#define MTP_BOK_LAUNCH(VEC, DISP, ...)                                                                              \
    hipLaunchKernelGGL((k_best_of_k<T, VEC, DISP, ##__VA_ARGS__>), dim3(grid), dim3(256), lds, s, pred, gt, n, k, kstride, t, \
                       ade_div, mode, apb, order, ostride, min_ade, min_fde, idx, idx2, sums)
    if constexpr (sizeof(T) == 4) {
        if (piped && disp) {
            MTP_BOK_LAUNCH(true, true, 6);
            MTP_LAUNCH_CHECK();
            return MTP_OK;
        }
        if (piped) {
            MTP_BOK_LAUNCH(true, false, 6);
            MTP_LAUNCH_CHECK();
            return MTP_OK;
        }
    }
    if (vec && disp)
        MTP_BOK_LAUNCH(true, true);
    else if (vec)
        MTP_BOK_LAUNCH(true, false);
    else if (disp)
        MTP_BOK_LAUNCH(false, true);
    else
        MTP_BOK_LAUNCH(false, false);
#undef MTP_BOK_LAUNCH
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_best_of_k_f32(const float *pred, const float *gt, int64_t n, int k, int kstride, int t, float ade_div,
                                 int mode, float *min_ade, float *min_fde, int32_t *idx, int32_t *idx2, double *sums,
                                 void *stream) {
    return best_of_k_impl<float>(pred, gt, n, k, kstride, t, ade_div, mode, false, nullptr, 0, min_ade, min_fde, idx,
                                 idx2, sums, stream);
}

extern "C" int mtp_best_of_k_f64(const double *pred, const double *gt, int64_t n, int k, int kstride, int t,
                                 double ade_div, int mode, double *min_ade, double *min_fde, int32_t *idx,
                                 int32_t *idx2, double *sums, void *stream) {
    return best_of_k_impl<double>(pred, gt, n, k, kstride, t, ade_div, mode, false, nullptr, 0, min_ade, min_fde, idx,
                                  idx2, sums, stream);
}

extern "C" int mtp_best_of_k_disp_f32(const float *disp, const float *gt, int64_t n, int k, int kstride, int t,
                                      float ade_div, int mode, float *min_ade, float *min_fde, int32_t *idx,
                                      int32_t *idx2, double *sums, void *stream) {
    return best_of_k_impl<float>(disp, gt, n, k, kstride, t, ade_div, mode, true, nullptr, 0, min_ade, min_fde, idx,
                                 idx2, sums, stream);
}

extern "C" int mtp_best_of_k_reranked_f32(const float *pred, const float *gt, const int32_t *order, int ostride,
                                          int64_t n, int k, int kstride, int t, float ade_div, int mode,
                                          float *min_ade, float *min_fde, int32_t *idx, int32_t *idx2, double *sums,
                                          void *stream) {
    MTP_REQUIRE(order != nullptr, "mtp_best_of_k_reranked: order is NULL");
    return best_of_k_impl<float>(pred, gt, n, k, kstride, t, ade_div, mode, false, order, ostride, min_ade, min_fde,
                                 idx, idx2, sums, stream);
}

extern "C" int mtp_best_of_k_reranked_f64(const double *pred, const double *gt, const int32_t *order, int ostride,
                                          int64_t n, int k, int kstride, int t, double ade_div, int mode,
                                          double *min_ade, double *min_fde, int32_t *idx, int32_t *idx2,
                                          double *sums, void *stream) {
    MTP_REQUIRE(order != nullptr, "mtp_best_of_k_reranked: order is NULL");
    return best_of_k_impl<double>(pred, gt, n, k, kstride, t, ade_div, mode, false, order, ostride, min_ade, min_fde,
                                  idx, idx2, sums, stream);
}

extern "C" int mtp_ade_fde_f32(const float *pred, const float *gt, int64_t n, int k, int kstride, int t, float ade_div,
                               float *ade, float *fde, void *stream) {
    MTP_REQUIRE(n >= 0 && k >= 1 && kstride >= k && t >= 1, "mtp_ade_fde: bad shape");
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(pred && gt && ade && fde, "mtp_ade_fde: null pointer");
    const int64_t total = n * k;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool vec = (t % 2 == 0) && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(gt)) % 16 == 0);
    hipStream_t s = mtp_stream(stream);
    if (vec)
        hipLaunchKernelGGL((k_ade_fde<float, true>), dim3((unsigned)blocks), dim3(256), 0, s, pred, gt, n, k, kstride, t,
                           ade_div, ade, fde);
    else
        hipLaunchKernelGGL((k_ade_fde<float, false>), dim3((unsigned)blocks), dim3(256), 0, s, pred, gt, n, k, kstride,
                           t, ade_div, ade, fde);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
