// Decode glue around the best-of-K evaluation on gfx950.
//
// Replaces, for all candidates of all agents in one launch,
//   PCCSNet/utils/data_process.py:111-122   transform_trajectory: in-place prefix sum over time
//   PCCSNet/train_steps.py:253-255          gen_prediction: rotate back with the per-agent 2x2
//                                           matrix (torch.bmm) and add the last observed position
// The output layout [n, k, t, 2] float32 is the wire format of all_output_<ds>_top<k>.npy
// (train_steps.py:257-262), which StimulusVerification/verify_predictions.py:52-54 reads back.
//
// HBM-bound: N*K*T*2*4 bytes read + the same written.  One thread per candidate; the prefix
// sum is sequential in time (p_j = d_j + p_{j-1}, float adds in the reference's order, so the
// cumulative positions are bit-identical to torch's).  A candidate is T*8 bytes (96 at T = 12), so
// per-thread 16-byte accesses would be strided by that and every store a partial line (measured
// 1.6 TB/s): k_decode_staged moves a workgroup's 256 candidates through LDS instead -- global
// loads and stores are linear 16 bytes per lane, each thread picks its own candidate out of LDS
// (rows padded to an odd number of 16-byte slots: conflict-free ds_read_b128).
#include "mtp_internal.h"

__device__ __forceinline__ void decode_step(float x, float y, float &cx, float &cy, bool has_rot, float m00, float m01,
                                            float m10, float m11, bool has_last, float lx, float ly, float &ox,
                                            float &oy) {
    // running sum, then out = M @ p (+ last_obs); products and sums rounded separately
    cx = x + cx;
    cy = y + cy;
    ox = cx;
    oy = cy;
    if (has_rot) {
        ox = m00 * cx + m01 * cy;
        oy = m10 * cx + m11 * cy;
    }
    if (has_last) {
        ox = ox + lx;
        oy = oy + ly;
    }
}

// t even, 16-byte aligned tensors.  cv = t/2 16-byte slots per candidate, LDS row stride rs (odd).
// CV > 0: cv as a compile-time constant (6 for the reference's 12 steps) -- a thread's CV loads of a tile are all issued
// before the first of them is used, and the grid is one workgroup per tile (the dispatcher keeps every CU fed to the
// end); CV = 0: any cv, persistent grid.
template <int CV>
__global__ __launch_bounds__(256) void k_decode_staged(const float *disp, int64_t total /* n*k */, int k, int cv_rt, int rs,
                                                       const float *__restrict__ rot,
                                                       const float *__restrict__ last_obs, float *out) {
    extern __shared__ __attribute__((aligned(16))) float4 tile[];  // 256 * rs slots
    const int cv = CV > 0 ? CV : cv_rt;
    const int tid = threadIdx.x;
    const int64_t n_tiles = (total + 255) / 256;
    for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int64_t c0 = tl * 256;
        const int nc = (int)min((int64_t)256, total - c0);
        const int n_slots = nc * cv;
        const float4 *src = reinterpret_cast<const float4 *>(disp) + c0 * cv;
        float4 *dst = reinterpret_cast<float4 *>(out) + c0 * cv;
        if constexpr (CV > 0) {
            float4 v[CV];
#pragma unroll
            for (int i = 0; i < CV; ++i) {
                const int e = tid + 256 * i;
                v[i] = src[e < n_slots ? e : n_slots - 1];
            }
#pragma unroll
            for (int i = 0; i < CV; ++i) {
                const int e = tid + 256 * i;
                const int c = e / CV, q = e - c * CV;
                if (e < n_slots) tile[c * rs + q] = v[i];
            }
        } else {
            for (int e = tid; e < n_slots; e += 256) {
                const int c = e / cv, q = e - c * cv;
                tile[c * rs + q] = src[e];
            }
        }
        __syncthreads();
        if (tid < nc) {
            const int64_t a = (c0 + tid) / k;
            float m00 = 1.f, m01 = 0.f, m10 = 0.f, m11 = 1.f, lx = 0.f, ly = 0.f;
            if (rot != nullptr) {
                const float4 m = *reinterpret_cast<const float4 *>(rot + a * 4);
                m00 = m.x; m01 = m.y; m10 = m.z; m11 = m.w;
            }
            if (last_obs != nullptr) {
                lx = last_obs[a * 2];
                ly = last_obs[a * 2 + 1];
            }
            float cx = 0.f, cy = 0.f;
            float4 *row = tile + tid * rs;
#pragma unroll
            for (int q = 0; q < (CV > 0 ? CV : cv); ++q) {
                const float4 v = row[q];
                float4 w;
                decode_step(v.x, v.y, cx, cy, rot != nullptr, m00, m01, m10, m11, last_obs != nullptr, lx, ly, w.x, w.y);
                decode_step(v.z, v.w, cx, cy, rot != nullptr, m00, m01, m10, m11, last_obs != nullptr, lx, ly, w.z, w.w);
                row[q] = w;
            }
        }
        __syncthreads();
        if constexpr (CV > 0) {
#pragma unroll
            for (int i = 0; i < CV; ++i) {
                const int e = tid + 256 * i;
                const int c = e / CV, q = e - c * CV;
                if (e < n_slots) dst[e] = tile[c * rs + q];
            }
        } else {
            for (int e = tid; e < n_slots; e += 256) {
                const int c = e / cv, q = e - c * cv;
                dst[e] = tile[c * rs + q];
            }
        }
        __syncthreads();
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_decode_trajectories(const float *disp, int64_t n, int k, int t,
                                                             const float *__restrict__ rot,
                                                             const float *__restrict__ last_obs, float *out) {
    const int64_t total = n * k;
    const size_t ts = (size_t)t * 2;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = c / k;
        float m00 = 1.f, m01 = 0.f, m10 = 0.f, m11 = 1.f, lx = 0.f, ly = 0.f;
        if (rot != nullptr) {
            const float4 m = *reinterpret_cast<const float4 *>(rot + a * 4);
            m00 = m.x; m01 = m.y; m10 = m.z; m11 = m.w;
        }
        if (last_obs != nullptr) {
            lx = last_obs[a * 2];
            ly = last_obs[a * 2 + 1];
        }
        const float *p = disp + (size_t)c * ts;
        float *o = out + (size_t)c * ts;
        float cx = 0.f, cy = 0.f;
        auto emit = [&](float x, float y, float &ox, float &oy) {
            // running sum, then out = M @ p (+ last_obs); products and sums rounded separately
            cx = x + cx;
            cy = y + cy;
            ox = cx;
            oy = cy;
            if (rot != nullptr) {
                ox = m00 * cx + m01 * cy;
                oy = m10 * cx + m11 * cy;
            }
            if (last_obs != nullptr) {
                ox = ox + lx;
                oy = oy + ly;
            }
        };
        if (VEC) {
            const float4 *pv = reinterpret_cast<const float4 *>(p);
            float4 *ov = reinterpret_cast<float4 *>(o);
            const int half = t >> 1;
#pragma unroll 6
            for (int q = 0; q < half; ++q) {
                const float4 v = pv[q];
                float4 w;
                emit(v.x, v.y, w.x, w.y);
                emit(v.z, v.w, w.z, w.w);
                ov[q] = w;
            }
        } else {
            for (int q = 0; q < t; ++q) {
                float ox, oy;
                emit(p[2 * q], p[2 * q + 1], ox, oy);
                o[2 * q] = ox;
                o[2 * q + 1] = oy;
            }
        }
    }
}

extern "C" int mtp_decode_trajectories_f32(const float *disp, int64_t n, int k, int t, const float *rot,
                                           const float *last_obs, float *out, void *stream) {
    MTP_REQUIRE(n >= 0 && k >= 1 && t >= 1, "mtp_decode_trajectories: bad shape n=%lld k=%d t=%d", (long long)n, k, t);
    if (n == 0) return MTP_OK;
    MTP_REQUIRE(disp && out, "mtp_decode_trajectories: null pointer");
    MTP_REQUIRE(rot == nullptr || reinterpret_cast<uintptr_t>(rot) % 16 == 0,
                "mtp_decode_trajectories: rotation matrices must be 16-byte aligned");
    const int64_t total = n * k;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool vec = (t % 2 == 0) &&
                     ((reinterpret_cast<uintptr_t>(disp) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    hipStream_t s = mtp_stream(stream);
    const int cv = t / 2, rs = (cv % 2 == 0) ? cv + 1 : cv;
    const size_t lds = (size_t)256 * rs * 16;
    if (vec && lds <= 64 * 1024) {
        int64_t tiles = (total + 255) / 256;
        if (cv == 6 && tiles <= 0x7fffffff) {
            hipLaunchKernelGGL(k_decode_staged<6>, dim3((unsigned)tiles), dim3(256), lds, s, disp, total, k, cv, rs, rot,
                               last_obs, out);
        } else {
            if (tiles > 256 * 8) tiles = 256 * 8;
            hipLaunchKernelGGL(k_decode_staged<0>, dim3((unsigned)tiles), dim3(256), lds, s, disp, total, k, cv, rs, rot,
                               last_obs, out);
        }
    } else if (vec)
        hipLaunchKernelGGL(k_decode_trajectories<true>, dim3((unsigned)blocks), dim3(256), 0, s, disp, n, k, t, rot,
                           last_obs, out);
    else
        hipLaunchKernelGGL(k_decode_trajectories<false>, dim3((unsigned)blocks), dim3(256), 0, s, disp, n, k, t, rot,
                           last_obs, out);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
