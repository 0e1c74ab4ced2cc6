// The segmented centroid-sum kernel (included by update.hip and by tools/accumulate_tune.hip).
//
// grid = (row chunks, cluster ranges).  A workgroup owns clusters [k0, k0+kr) in LDS and scans the
// labels of its row chunk; rows whose label is in range are fetched (16 B per lane, LPR lanes
// per row, 64/LPR rows per wave instruction) and added in LDS; the partials are flushed with
// global double atomics.  LDS layout per cluster row: element 4c+e stored at e*(d/4) + c, so
// consecutive lanes hit consecutive accumulators for a fixed e.
#pragma once
#include "mtp_internal.h"

// one 16-byte piece of a row; NT: a non-temporal load (the four scalar loads merge into one global_load_dwordx4 nt)
template <bool NT>
__device__ inline float4 acc_load16(const float *p) {
    if constexpr (NT) {
        float4 v;
        v.x = __builtin_nontemporal_load(p);
        v.y = __builtin_nontemporal_load(p + 1);
        v.z = __builtin_nontemporal_load(p + 2);
        v.w = __builtin_nontemporal_load(p + 3);
        return v;
    } else {
        return *reinterpret_cast<const float4 *>(p);
    }
}

// ACC = double (product) or float (tuning only): the LDS accumulator type.  G = groups of
// 64/LPR rows whose loads are issued together before their LDS adds.  DO_ADD = false skips the
// LDS adds (tuning ablation: load-only time).
// DT: the feature size as a compile-time constant (0: any d, taken from the argument) -- row addresses and LDS
// offsets then fold into immediates and shifts.
// NT: the rows are fetched with non-temporal loads (a single run: every row is read once, by the one workgroup whose
// cluster range holds its label; batched restarts read the rows once per restart and keep the plain loads).
template <int LPR, int G, typename ACC, bool DO_ADD, int DT = 0, bool NT = false>
__global__ __launch_bounds__(1024) void k_accumulate(const float *__restrict__ X, int64_t n, int d_rt,
                                                     const int32_t *__restrict__ labels, int k, int kr,
                                                     int64_t rows_per_chunk, double *sums, double *counts,
                                                     const int32_t *state, MtpBatchStride bs) {
    const int d = DT ? DT : d_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char acc_raw[];  // kr*d ACC, then kr ints
    ACC *acc = reinterpret_cast<ACC *>(acc_raw);
    {
        const size_t b = blockIdx.z;  // restart
        labels += b * bs.labels;
        sums += b * bs.red;
        counts += b * bs.red;
        if (state != nullptr) state += b * bs.flags;
    }
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    const int k0 = blockIdx.y * kr;
    const int kn = min(kr, k - k0);
    int *cnt = reinterpret_cast<int *>(acc + (size_t)kr * d);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int e = tid; e < kn * d; e += nthreads) acc[e] = (ACC)0;
    for (int e = tid; e < kn; e += nthreads) cnt[e] = 0;
    __syncthreads();

    constexpr int RPW = 64 / LPR;  // rows per wave instruction
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int sub = lane / LPR, c0 = lane % LPR;
    const int dq = d / 4;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r_end = min(n, r_begin + rows_per_chunk);
    const int64_t stride = (int64_t)nwaves * 64;

    // Software pipeline over "trips" (up to G groups of RPW matching rows): the row loads of
    // trip i+1 are issued before the LDS adds of trip i, so a wave always has loads in flight.
    int64_t base = r_begin + (int64_t)wave * 64;
    int lab_next = -1;
    if (base + lane < r_end) lab_next = labels[base + lane];
    int rel = -1;
    unsigned long long m = 0;
    bool first = true;
    float4 vc[G];
    int rc[G];
    const float *pc[G];
    bool have_cur = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rc[g] = -1;
        pc[g] = nullptr;
        vc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (;;) {
        // advance to the next batch of 64 labels that has a row for this cluster range
        while (m == 0 && base < r_end) {
            if (!first) base += stride;
            first = false;
            if (base >= r_end) break;
            rel = lab_next - k0;
            lab_next = -1;  // the batch after this one travels while this one's rows are fetched
            if (base + stride + lane < r_end) lab_next = labels[base + stride + lane];
            const bool hit = (rel >= 0) && (rel < kn);
            m = __ballot(hit);
            if (hit) atomicAdd(&cnt[rel], 1);
        }
        const bool have_nxt = (m != 0);
        float4 vn[G];
        int rn[G];
        const float *pn[G];
        if (have_nxt) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                int b = -1;
#pragma unroll
                for (int q = 0; q < RPW; ++q) {
                    const int bit = m ? (int)__builtin_ctzll(m) : -1;
                    if (m) m &= m - 1;
                    if (q == sub) b = bit;
                }
                const int r = __shfl(rel, b < 0 ? 0 : b);
                rn[g] = (b >= 0 && c0 < dq) ? r : -1;
                pn[g] = X + (base + (b < 0 ? 0 : b)) * d;
                vn[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rn[g] >= 0) vn[g] = acc_load16<NT>(pn[g] + 4 * c0);
            }
        }
        if (have_cur) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (rc[g] >= 0) {
                    if constexpr (DO_ADD) {
                        ACC *a = acc + (size_t)rc[g] * d + c0;
                        atomicAdd(a, (ACC)vc[g].x);
                        atomicAdd(a + dq, (ACC)vc[g].y);
                        atomicAdd(a + 2 * dq, (ACC)vc[g].z);
                        atomicAdd(a + 3 * dq, (ACC)vc[g].w);
                        // rows wider than LPR*4 floats: remaining 16-byte chunks
                        for (int c = c0 + LPR; c < dq; c += LPR) {
                            const float4 w = reinterpret_cast<const float4 *>(pc[g])[c];
                            atomicAdd(a + (c - c0), (ACC)w.x);
                            atomicAdd(a + (c - c0) + dq, (ACC)w.y);
                            atomicAdd(a + (c - c0) + 2 * dq, (ACC)w.z);
                            atomicAdd(a + (c - c0) + 3 * dq, (ACC)w.w);
                        }
                    } else {
#if defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("" ::"v"(vc[g].x), "v"(vc[g].y), "v"(vc[g].z), "v"(vc[g].w));
#endif
                    }
                }
            }
        }
        if (!have_nxt) break;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            vc[g] = vn[g];
            rc[g] = rn[g];
            pc[g] = pn[g];
        }
        have_cur = true;
    }
    __syncthreads();
    // flush in the order of the global array (consecutive lanes add to consecutive doubles; LDS element
    // (row, e*dq + c) is column 4c+e), every row chunk of a cluster range starting at a different place so that the
    // workgroups of a range, which finish together, do not queue up on the same addresses
    {
        const int kd = kn * d;
        const int start = (int)((int64_t)blockIdx.x * kd / gridDim.x);
        double *out = sums + (size_t)k0 * d;
        for (int j = tid; j < kd; j += nthreads) {
            int gidx = j + start;
            if (gidx >= kd) gidx -= kd;
            const int row = gidx / d, col = gidx % d;
            const double v = (double)acc[(size_t)row * d + (col & 3) * dq + (col >> 2)];
            if (v != 0.0) atomicAdd(&out[gidx], v);
        }
    }
    for (int idx = tid; idx < kn; idx += nthreads)
        if (cnt[idx] != 0) atomicAdd(&counts[k0 + idx], (double)cnt[idx]);
}

// The same sums when ONE workgroup holds all k clusters (k*d doubles fit the LDS: the reference's K = 200 at D = 96,
// get_modality): every row of the chunk is added, so the ballot / bit-scan / shuffle that picks a range's rows out
// of a batch of labels is not needed.  A wave takes batches of 64 consecutive rows; trip t of a batch adds rows
// t*G*RPW .. +G*RPW-1, their labels come out of the batch's label register with v_readlane.
// Where the time goes at 73 568 x 96, K = 200, 20 restarts in one launch (12 chunks x 20 = 240 workgroups, one per
// CU; builds with parts switched off): 37 us with nothing but the label scan and the loop's instructions, +13 us
// with the row loads, +20 us with the LDS adds (a ds_add_f64 instruction occupies the CU's LDS for 9 clocks
// whatever its active lanes, tools/lds_atomic_rate), +22 us for the flush.
template <int LPR, int G, int DT = 0>
__global__ __launch_bounds__(1024) void k_accumulate_all(const float *__restrict__ X, int64_t n, int d_rt,
                                                         const int32_t *__restrict__ labels, int k,
                                                         int64_t rows_per_chunk, double *sums, double *counts,
                                                         const int32_t *state, MtpBatchStride bs) {
    const int d = DT ? DT : d_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char acc_raw[];  // k*d doubles, then k ints
    double *acc = reinterpret_cast<double *>(acc_raw);
    // 1-D grid of chunks x restarts.  Workgroup i runs on XCD i % 8: the (chunk, restart) pairs are dealt out chunk-major
    // in eight consecutive runs, one per XCD, so that the restarts that add the same rows share an L2 -- the rows come
    // from HBM / the memory-side cache once per XCD instead of once per restart.
    int chunk;
    {
        const int total = gridDim.x, i = blockIdx.x, x = i & 7, q = total >> 3, r = total & 7;
        const int w = x * q + min(x, r) + (i >> 3);
        chunk = w / bs.batch;
        const size_t b = w % bs.batch;  // restart
        labels += b * bs.labels;
        sums += b * bs.red;
        counts += b * bs.red;
        if (state != nullptr) state += b * bs.flags;
    }
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    int *cnt = reinterpret_cast<int *>(acc + (size_t)k * d);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int e = tid; e < k * d; e += nthreads) acc[e] = 0.0;
    for (int e = tid; e < k; e += nthreads) cnt[e] = 0;
    __syncthreads();

    constexpr int RPW = 64 / LPR;   // rows per wave instruction
    constexpr int TR = G * RPW;     // rows per trip
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int sub = lane / LPR, c0 = lane % LPR;
    const int dq = d / 4;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = min(n, r_begin + rows_per_chunk);
    const int64_t stride = (int64_t)nwaves * 64;

    float4 vc[G];
    int rc[G];
    int ic[G];  // the row, counted from the chunk's first (only rows wider than LPR*4 floats are fetched again)
    bool have_cur = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rc[g] = -1;
        ic[g] = 0;
        vc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto add_cur = [&]() {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (rc[g] >= 0) {
                double *a = acc + (size_t)rc[g] * d + c0;
                atomicAdd(a, (double)vc[g].x);
                atomicAdd(a + dq, (double)vc[g].y);
                atomicAdd(a + 2 * dq, (double)vc[g].z);
                atomicAdd(a + 3 * dq, (double)vc[g].w);
                if constexpr (LPR == 64) {  // rows wider than LPR*4 floats (the launcher picks LPR >= d/4 up to 64)
                    for (int c = c0 + LPR; c < dq; c += LPR) {
                        const float4 w = reinterpret_cast<const float4 *>(X + (r_begin + ic[g]) * d)[c];
                        atomicAdd(a + (c - c0), (double)w.x);
                        atomicAdd(a + (c - c0) + dq, (double)w.y);
                        atomicAdd(a + (c - c0) + 2 * dq, (double)w.z);
                        atomicAdd(a + (c - c0) + 3 * dq, (double)w.w);
                    }
                }
            }
        }
    };

    int64_t base = r_begin + (int64_t)wave * 64;
    int lab_next = -1;
    if (base + lane < r_end) lab_next = labels[base + lane];
    for (; base < r_end; base += stride) {
        int lab = lab_next;
        lab_next = -1;  // the next batch's labels travel while this one's rows are added
        if (base + stride + lane < r_end) lab_next = labels[base + stride + lane];
        if ((unsigned)lab < (unsigned)k) atomicAdd(&cnt[lab], 1);
        else lab = -1;  // beyond the chunk, or not a cluster: skipped
        const int rows_here = (int)min((int64_t)64, r_end - base);
        for (int t0 = 0; t0 < rows_here; t0 += TR) {
            float4 vn[G];
            int rn[G];
            int in[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                int r = -1;
#pragma unroll
                for (int q = 0; q < RPW; ++q) {
                    const int rq = __builtin_amdgcn_readlane(lab, (t0 + g * RPW + q) & 63);
                    if (q == sub) r = rq;
                }
                if (t0 + g * RPW + sub >= 64) r = -1;
                rn[g] = (c0 < dq) ? r : -1;
                in[g] = (int)(base - r_begin) + t0 + g * RPW + sub;
                vn[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rn[g] >= 0) vn[g] = reinterpret_cast<const float4 *>(X + (r_begin + in[g]) * d)[c0];
            }
            if (have_cur) add_cur();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                vc[g] = vn[g];
                rc[g] = rn[g];
                ic[g] = in[g];
            }
            have_cur = true;
        }
    }
    if (have_cur) add_cur();
    __syncthreads();
    // flush in the order of the global array (consecutive lanes add to consecutive doubles; LDS element
    // (row, e*dq + c) is column 4c+e), every chunk of a restart starting at a different place so that the workgroups
    // of a restart, which finish together, do not queue up on the same addresses
    {
        const int kd = k * d;
        const int chunks = gridDim.x / bs.batch;
        const int start = (int)((int64_t)chunk * kd / (chunks > 0 ? chunks : 1));
        for (int j = tid; j < kd; j += nthreads) {
            int gidx = j + start;
            if (gidx >= kd) gidx -= kd;
            const int row = gidx / d, col = gidx % d;
            const double v = acc[(size_t)row * d + (col & 3) * dq + (col >> 2)];
            if (v != 0.0) atomicAdd(&sums[gidx], v);
        }
    }
    for (int idx = tid; idx < k; idx += nthreads)
        if (cnt[idx] != 0) atomicAdd(&counts[idx], (double)cnt[idx]);
}
