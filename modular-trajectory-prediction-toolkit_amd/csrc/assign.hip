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
#include "mtp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte LDS-DMA: LDS destination = wave-uniform byte address (M0) + lane*16, the global source
// is per lane.  Written as inline asm on purpose: hipcc treats the builtin as a pending LDS
// write and puts s_waitcnt vmcnt(0) in front of the next ds_read, which would serialise the
// prefetch of stage t+1 with the MFMAs of stage t.  The asm form is invisible to its vmcnt
// bookkeeping; the kernel waits for its own DMA (dma_wait) before the barrier that publishes a
// stage.  M0 is saved/restored inside the statement (it is compiler-reserved).
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr_uniform)
        : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int D, int PT>
__global__ __launch_bounds__(256, 2) void k_assign_mfma(const float *__restrict__ X, int64_t n,
                                                        const float *__restrict__ staged, int n_tiles,
                                                        int32_t *__restrict__ labels, double *n_changed,
                                                        const int32_t *state) {
    constexpr int R = D / 4;                 // 16-byte slots per centroid row
    constexpr int TILE_F = 32 * D + 32;      // floats per 32-centroid tile image
    constexpr int STAGE_F = MTP_CTW * TILE_F;
    constexpr int STAGE_SLOTS = STAGE_F / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * STAGE_F floats

    if (state != nullptr && *state != MTP_STATE_RUNNING) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t pbase = ((int64_t)blockIdx.x * 4 + wave) * (32 * PT);

    // ---- B operands: this wave's points, k = 2s + h for step s, kept in registers ----
    float xr[PT][D / 2];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int64_t row = pbase + pt * 32 + j;
        if (row >= n) row = n - 1;  // tail rows are computed and discarded
        const float4 *xp = reinterpret_cast<const float4 *>(X + row * D);
#pragma unroll
        for (int q = 0; q < D / 4; ++q) {
            const float4 v = xp[q];
            xr[pt][2 * q] = h ? v.y : v.x;
            xr[pt][2 * q + 1] = h ? v.w : v.z;
            // half of every 16-byte load is dropped: bound the loads in flight so the
            // prologue does not spill
            if (q % 8 == 7) asm volatile("" ::: "memory");
        }
    }

    float best[PT];
    int bidx[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        best[pt] = __builtin_inff();
        bidx[pt] = 0;
    }

    const int n_stages = n_tiles / MTP_CTW;
    const int wave_slot0 = wave * 64;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto stage = [&](int t) {
        const float *g = staged + (size_t)t * STAGE_F;
        const unsigned l = smem_base + (unsigned)((t & 1) * STAGE_F * 4);
        for (int s0 = wave_slot0; s0 < STAGE_SLOTS; s0 += 256) {
            if (s0 + lane < STAGE_SLOTS)
                glds16(g + (size_t)(s0 + lane) * 4, __builtin_amdgcn_readfirstlane(l + (unsigned)s0 * 16));
        }
    };

    const int swz = mtp_swz(j, R);
    stage(0);
    for (int t = 0; t < n_stages; ++t) {
        dma_wait();       // this wave's share of stage t has landed ...
        __syncthreads();  // ... and so has everyone else's; all waves are done with the other buffer
        if (t + 1 < n_stages) stage(t + 1);
        const float *sb = smem + (t & 1) * STAGE_F;
#pragma unroll
        for (int ct = 0; ct < MTP_CTW; ++ct) {
            const float *tb = sb + ct * TILE_F;
            const float4 *arow = reinterpret_cast<const float4 *>(tb) + j * R;
            f32x16 acc[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
            // A operands one u-step ahead of the MFMAs that use them; the compiler fence keeps it
            // from hoisting all D/8 ds_read_b128 to the top (64 VGPRs) and spilling.
            float4 a = arow[h ^ swz];
#pragma unroll
            for (int u = 0; u < D / 8; ++u) {
                float4 an = a;
                if (u + 1 < D / 8) an = arow[(2 * (u + 1) + h) ^ swz];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[pt][4 * u + 0], acc[pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[pt][4 * u + 1], acc[pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[pt][4 * u + 2], acc[pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[pt][4 * u + 3], acc[pt], 0, 0, 0);
                asm volatile("" ::: "memory");
                a = an;
            }
            // epilogue: d = c2 - 2*dot (one rounding), strict '<' in ascending centroid order
            const int cbase = (t * MTP_CTW + ct) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c2v = *reinterpret_cast<const float4 *>(tb + 32 * D + 8 * g + 4 * h);
                const float c2a[4] = {c2v.x, c2v.y, c2v.z, c2v.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        const float dist = fmaf(-2.0f, acc[pt][4 * g + r], c2a[r]);
                        const bool lt = dist < best[pt];
                        best[pt] = lt ? dist : best[pt];
                        bidx[pt] = lt ? (cbase + 8 * g + r) : bidx[pt];
                    }
                }
            }
        }
    }

    // merge the two lane halves (rows 4h.. interleave): lexicographic (dist, index) minimum
    int changed = 0;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const float ob = __shfl_xor(best[pt], 32);
        const int oi = __shfl_xor(bidx[pt], 32);
        const bool take = (ob < best[pt]) || (ob == best[pt] && oi < bidx[pt]);
        const int lab = take ? oi : bidx[pt];
        const int64_t row = pbase + pt * 32 + j;
        if (h == 0 && row < n) {
            const int old = labels[row];
            labels[row] = lab;
            changed += (old != lab);
        }
    }
    if (n_changed != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
        if (lane == 0 && changed != 0) atomicAdd(n_changed, (double)changed);
    }
}

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
                                                      float *c2, const int32_t *state) {
    if (state != nullptr && *state != MTP_STATE_RUNNING) return;
    const int j = blockIdx.x;
    const bool valid = j < k;
    __shared__ float c2s;
    if (threadIdx.x == 0) {
        c2s = valid ? mtp_sqnorm_einsum_order(C + (size_t)j * d, d) : __builtin_inff();
        if (valid && c2 != nullptr) c2[j] = c2s;
    }
    __syncthreads();
    if (staged != nullptr) mtp_write_staged_row(staged, j, d, C + (size_t)(valid ? j : 0) * d, valid, c2s, threadIdx.x, 64);
}

int mtp_launch_stage_centers(const float *C, int k, int d, float *staged, float *c2, const int32_t *state,
                             hipStream_t s) {
    const bool mfma = mtp_mfma_dim(d);
    const int rows = mfma ? mtp_n_tiles(k) * 32 : k;
    hipLaunchKernelGGL(k_stage_centers, dim3(rows), dim3(64), 0, s, C, k, d, mfma ? staged : nullptr, c2, state);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

template <int D>
static int launch_assign_mfma(const float *X, int64_t n, const float *staged, int k, int32_t *labels,
                              double *n_changed, const int32_t *state, hipStream_t s) {
    constexpr int PT = 2;
    const size_t lds = 2 * (size_t)MTP_CTW * mtp_tile_floats(D) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        MTP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign_mfma<D, PT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int64_t pts_per_block = 4 * 32 * PT;
    const int64_t blocks = (n + pts_per_block - 1) / pts_per_block;
    MTP_REQUIRE(blocks <= 0x7fffffff, "mtp_assign: n too large");
    hipLaunchKernelGGL((k_assign_mfma<D, PT>), dim3((unsigned)blocks), dim3(256), lds, s, X, n, staged, mtp_n_tiles(k),
                       labels, n_changed, state);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

int mtp_launch_assign(const float *X, int64_t n, int d, const float *C, const float *staged, const float *c2, int k,
                      int32_t *labels, double *n_changed, const int32_t *state, hipStream_t s) {
    if (n == 0) return MTP_OK;
    if (mtp_mfma_dim(d)) {
        MTP_REQUIRE(staged != nullptr, "mtp_assign: staged centroid image missing for d=%d", d);
        switch (d) {
            case 64: return launch_assign_mfma<64>(X, n, staged, k, labels, n_changed, state, s);
            case 96: return launch_assign_mfma<96>(X, n, staged, k, labels, n_changed, state, s);
            case 128: return launch_assign_mfma<128>(X, n, staged, k, labels, n_changed, state, s);
        }
    }
    MTP_REQUIRE(d >= 1 && d <= 1024, "mtp_assign: generic kernel supports 1 <= d <= 1024 (got %d)", d);
    MTP_REQUIRE(C != nullptr && c2 != nullptr, "mtp_assign: generic kernel needs C and c2");
    const size_t lds = 32 * (size_t)d * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        MTP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign_generic),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024 * 4));
        attr_set = true;
    }
    const int64_t blocks = (n + 255) / 256;
    MTP_REQUIRE(blocks <= 0x7fffffff, "mtp_assign: n too large");
    hipLaunchKernelGGL(k_assign_generic, dim3((unsigned)blocks), dim3(256), lds, s, X, n, d, C, c2, k, labels,
                       n_changed, state);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}
