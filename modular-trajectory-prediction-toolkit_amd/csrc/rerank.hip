// Score-sorted NMS re-ranking of stored prediction sets on gfx950.
//
// Replaces the per-agent Python loop of StimulusVerification/verify_predictions.py:251-277:
//   score  = verification_scores[i][:M] // bin_size                       (NumPy floor_divide)
//   key    = float32(score) - float32(j) / (2 M)                          (torch float32)
//   order  = argsort(key, descending)                                     (ties cannot occur while
//            |score| < 2^23 / M; here they go to the lower index)
//   greedy NMS over the sorted candidates on the FINAL predicted point: candidate p is kept iff
//   fewer than best_of are kept so far and no kept candidate's final point is closer than
//   nms_thresh (np.linalg.norm: dx*dx + dy*dy, sqrt, in the predictions' dtype); the list becomes
//   kept ++ dropped, both in sorted order.
// Output: order[a, pos] = index (into the agent's ORIGINAL candidate list) of the candidate that
// ends up at position pos.  mtp_gather_candidates_* materialises the re-ordered tensor like the
// reference does; mtp_best_of_k_reranked_* (metrics.hip) scores through `order` without it.
//
// One agent per 64-lane workgroup.  Everything an agent needs -- M scores, M final points -- is
// read once (HBM-bound: M*(sizeof score + 2*sizeof T) bytes in, 4*M out; the final points are one
// 2*sizeof(T) piece of every T*2*sizeof(T)-byte candidate, so the fetched sectors are what counts).
// The sort is a rank-by-counting over LDS (M^2/64 broadcast reads per lane); the greedy pass is
// at most best_of rounds of "update every later candidate's distance to the newest kept one, then
// ballot for the first survivor".
#include "mtp_internal.h"

static constexpr int RERANK_MAX_M = 1024;

// NumPy's floor_divide for floating types (numpy/core/src/npymath: npy_divmod).
template <typename S>
__device__ __forceinline__ S np_floor_divide(S a, S b) {
    if (b == (S)0) return a / b;
    S mod = fmod(a, b);
    S div = (a - mod) / b;
    if (mod != (S)0) {
        if ((b < (S)0) != (mod < (S)0)) div -= (S)1;
    }
    if (div != (S)0) {
        S fl = floor(div);
        if (div - fl > (S)0.5) fl += (S)1;
        return fl;
    }
    return copysign((S)0, a / b);
}
__device__ __forceinline__ float tsqrt_(float v) { return sqrtf(v); }
__device__ __forceinline__ double tsqrt_(double v) { return sqrt(v); }

template <typename T, typename S>
__global__ __launch_bounds__(64) void k_nms_rerank(const T *__restrict__ pred, const S *__restrict__ scores, int64_t n,
                                                   int mtot, int t, int m, int best_of, S bin_size, T nms_thresh,
                                                   int32_t *__restrict__ order, int32_t *__restrict__ n_selected) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    T *fx = reinterpret_cast<T *>(raw);          // [m] final point of the candidate at sorted position p
    T *fy = fx + m;
    T *mind = fy + m;                            // [m] distance to the nearest kept candidate so far
    float *key = reinterpret_cast<float *>(mind + m);  // [m]
    int *sidx = reinterpret_cast<int *>(key + m);      // [m] sorted position -> original index
    int *slot = sidx + m;                              // [m] >= 0: position in the kept list; -1: not kept
    const int lane = threadIdx.x;
    const int64_t a = blockIdx.x;
    const S *sc = scores + (size_t)a * mtot;
    const T *pa = pred + (size_t)a * mtot * t * 2;

    for (int c = lane; c < m; c += 64) {
        const S q = np_floor_divide<S>(sc[c], bin_size);
        float kf = (float)q - (float)c / (float)(2 * m);
        if (kf != kf) kf = -__builtin_inff();  // NaN scores sort last (the reference's order is unspecified)
        key[c] = kf;
    }
    __syncthreads();
    // rank by counting: descending key, ties to the lower index
    for (int c = lane; c < m; c += 64) {
        const float kc = key[c];
        int r = 0;
        for (int o = 0; o < m; ++o) {
            const float ko = key[o];
            r += (ko > kc) || (ko == kc && o < c);
        }
        sidx[r] = c;
    }
    __syncthreads();
    for (int p = lane; p < m; p += 64) {
        const T *fp = pa + ((size_t)sidx[p] * t + (t - 1)) * 2;
        fx[p] = fp[0];
        fy[p] = fp[1];
        mind[p] = (T)INFINITY;
        slot[p] = -1;
    }
    __syncthreads();
    // greedy pass: position 0 is always kept (selected_idx = [0])
    int cur = 0, n_sel = 1;
    if (lane == 0) slot[0] = 0;
    while (n_sel < best_of) {
        const T sx = fx[cur], sy = fy[cur];
        int next = -1;
        for (int p0 = 0; p0 < m && next < 0; p0 += 64) {
            const int p = p0 + lane;
            bool ok = false;
            if (p < m && p > cur) {
                const T dx = sx - fx[p], dy = sy - fy[p];
                const T d = tsqrt_(dx * dx + dy * dy);
                T md = mind[p];
                md = d < md ? d : md;
                mind[p] = md;
                ok = !(md < nms_thresh);
            }
            const unsigned long long mask = __ballot(ok);
            if (mask) next = p0 + (int)__builtin_ctzll(mask);
        }
        if (next < 0) break;
        // the search stopped at next's chunk: candidates in later chunks still need this round's
        // update against `cur`
        for (int p0 = (next / 64 + 1) * 64; p0 < m; p0 += 64) {
            const int p = p0 + lane;
            if (p < m) {
                const T dx = sx - fx[p], dy = sy - fy[p];
                const T d = tsqrt_(dx * dx + dy * dy);
                const T md = mind[p];
                mind[p] = d < md ? d : md;
            }
        }
        if (lane == 0) slot[next] = n_sel;
        cur = next;
        ++n_sel;
        __syncthreads();
    }
    __syncthreads();
    // new order = kept (in pick order) ++ dropped (ascending sorted position)
    int32_t *oa = order + (size_t)a * m;
    int base = 0;
    for (int p0 = 0; p0 < m; p0 += 64) {
        const int p = p0 + lane;
        const bool valid = p < m;
        const int sl = valid ? slot[p] : 0;
        const bool dropped = valid && sl < 0;
        const unsigned long long mask = __ballot(dropped);
        if (valid) {
            const int below = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            const int pos = dropped ? n_sel + base + below : sl;
            oa[pos] = sidx[p];
        }
        base += __builtin_popcountll(mask);
    }
    if (n_selected != nullptr && lane == 0) n_selected[a] = n_sel;
}

template <typename T, typename S>
static int nms_rerank_impl(const T *pred, const S *scores, int64_t n, int mtot, int t, int max_samples, int best_of,
                           double bin_size, double nms_thresh, int32_t *order, int32_t *n_selected, void *stream) {
    MTP_REQUIRE(n >= 0 && mtot >= 1 && t >= 1, "mtp_nms_rerank: bad shape n=%lld mtot=%d t=%d", (long long)n, mtot, t);
    MTP_REQUIRE(max_samples >= 1 && max_samples <= mtot && max_samples <= RERANK_MAX_M,
                "mtp_nms_rerank: max_samples=%d must be in [1, min(mtot=%d, %d)]", max_samples, mtot, RERANK_MAX_M);
    MTP_REQUIRE(best_of >= 1, "mtp_nms_rerank: best_of must be >= 1 (got %d)", best_of);
    MTP_REQUIRE(n <= 0x7fffffff, "mtp_nms_rerank: n too large");
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(pred && scores && order, "mtp_nms_rerank: null pointer");
    const size_t lds = (size_t)max_samples * (3 * sizeof(T) + 12);
    hipLaunchKernelGGL((k_nms_rerank<T, S>), dim3((unsigned)n), dim3(64), lds, mtp_stream(stream), pred, scores, n, mtot,
                       t, max_samples, best_of, (S)bin_size, (T)nms_thresh, order, n_selected);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_nms_rerank_f32(const float *pred, const void *scores, int scores_f64, int64_t n, int mtot, int t,
                                  int max_samples, int best_of, double bin_size, double nms_thresh, int32_t *order,
                                  int32_t *n_selected, void *stream) {
    if (scores_f64)
        return nms_rerank_impl<float, double>(pred, static_cast<const double *>(scores), n, mtot, t, max_samples,
                                              best_of, bin_size, nms_thresh, order, n_selected, stream);
    return nms_rerank_impl<float, float>(pred, static_cast<const float *>(scores), n, mtot, t, max_samples, best_of,
                                         bin_size, nms_thresh, order, n_selected, stream);
}

extern "C" int mtp_nms_rerank_f64(const double *pred, const void *scores, int scores_f64, int64_t n, int mtot, int t,
                                  int max_samples, int best_of, double bin_size, double nms_thresh, int32_t *order,
                                  int32_t *n_selected, void *stream) {
    if (scores_f64)
        return nms_rerank_impl<double, double>(pred, static_cast<const double *>(scores), n, mtot, t, max_samples,
                                               best_of, bin_size, nms_thresh, order, n_selected, stream);
    return nms_rerank_impl<double, float>(pred, static_cast<const float *>(scores), n, mtot, t, max_samples, best_of,
                                          bin_size, nms_thresh, order, n_selected, stream);
}

// out[a, p] = pred[a, order[a, p]] for p < m, out[a, p] = pred[a, p] for m <= p < mtot
// (base_model_predictions_copy[i, :max_samples] = ...[new_order], verify_predictions.py:277).
// One 16-byte piece per thread; out must not alias pred.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_candidates(const T *__restrict__ pred,
                                                           const int32_t *__restrict__ order, int64_t n, int mtot, int m,
                                                           int cand_vecs /* 16-byte pieces per candidate */,
                                                           T *__restrict__ out) {
    typedef float4 V;  // 16 bytes, whatever T is
    const int64_t total = n * mtot * cand_vecs;
    const V *src = reinterpret_cast<const V *>(pred);
    V *dst = reinterpret_cast<V *>(out);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = e / cand_vecs;
        const int piece = (int)(e - c * cand_vecs);
        const int64_t a = c / mtot;
        const int p = (int)(c - a * mtot);
        const int from = p < m ? order[a * m + p] : p;
        dst[e] = src[(a * mtot + from) * cand_vecs + piece];
    }
}

template <typename T>
static int gather_impl(const T *pred, const int32_t *order, int64_t n, int mtot, int m, int t, T *out, void *stream) {
    MTP_REQUIRE(n >= 0 && mtot >= 1 && m >= 0 && m <= mtot && t >= 1, "mtp_gather_candidates: bad shape");
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(pred && order && out && pred != out, "mtp_gather_candidates: null or aliasing pointers");
    const size_t cand_bytes = (size_t)t * 2 * sizeof(T);
    MTP_REQUIRE(cand_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(pred) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(out) % 16 == 0,
                "mtp_gather_candidates: a candidate must be a multiple of 16 bytes and the tensors 16-byte aligned");
    const int cand_vecs = (int)(cand_bytes / 16);
    const int64_t total = n * mtot * cand_vecs;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_gather_candidates<T>, dim3((unsigned)blocks), dim3(256), 0, mtp_stream(stream), pred, order, n,
                       mtot, m, cand_vecs, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

extern "C" int mtp_gather_candidates_f32(const float *pred, const int32_t *order, int64_t n, int mtot, int m, int t,
                                         float *out, void *stream) {
    return gather_impl<float>(pred, order, n, mtot, m, t, out, stream);
}
extern "C" int mtp_gather_candidates_f64(const double *pred, const int32_t *order, int64_t n, int mtot, int m, int t,
                                         double *out, void *stream) {
    return gather_impl<double>(pred, order, n, mtot, m, t, out, stream);
}
