// The fp32-MFMA E-step kernel (included by assign.hip and by tools/assign_tune.hip).
#pragma once
#include <type_traits>

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
// keeps a value live without cost (ablation builds); the asm only exists in the device pass
__device__ __forceinline__ void keep_alive(const f32x16 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#else
    (void)v;
#endif
}
__device__ __forceinline__ void pin2(float &a, float &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(a), "+v"(b));
#endif
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Template parameters (LG: 16-byte point loads in flight in the prologue, must divide D/4; PRIO:
// raised wave priority during the MFMA run of a tile): D feature size, PT 32-point tiles per wave, CTW 32-centroid tiles per LDS
// stage, WPB waves per workgroup, WPS waves per SIMD the register budget is sized for (launch bound), EPI = false drops
// the bias/argmin epilogue (tuning ablation only; the product always instantiates EPI = true).
template <int D, int PT, int CTW, int WPB, int WPS, bool EPI, int LG = 8, bool PRIO = false>
__global__ __launch_bounds__(64 * WPB, WPS) void k_assign_mfma(const float *__restrict__ X, int64_t n,
                                                        const float *__restrict__ staged, int n_tiles,
                                                        int32_t *__restrict__ labels, double *n_changed,
                                                        const int32_t *state, MtpBatchStride bs) {
    // n_tiles: tiles that hold centroids (ceil(k / 32)); the image is padded to whole stages, and
    // the tiles of the last stage beyond n_tiles are skipped
    constexpr int R = D / 4;                 // 16-byte slots per centroid row
    constexpr int TILE_F = 32 * (D + 4) + 32;  // floats per 32-centroid tile image (rows padded by one slot)
    constexpr int STAGE_F = CTW * TILE_F;
    constexpr int STAGE_SLOTS = STAGE_F / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * STAGE_F floats

    // restarts batched along blockIdx.y: every restart has its own centroid image, labels and loop state
    {
        const size_t b = blockIdx.y;
        staged += b * bs.staged;
        labels += b * bs.labels;
        if (n_changed != nullptr) n_changed += b * bs.red;
        if (state != nullptr) state += b * bs.flags;
    }
    if (mtp_skip(state, bs.skip_mode)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t pbase = ((int64_t)blockIdx.x * WPB + wave) * (32 * PT);

    const int n_stages = (n_tiles + CTW - 1) / CTW;
    const int wave_slot0 = wave * 64;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto stage = [&](int t) {
        const float *g = staged + (size_t)t * STAGE_F;
        const unsigned l = smem_base + (unsigned)((t & 1) * STAGE_F * 4);
        for (int s0 = wave_slot0; s0 < STAGE_SLOTS; s0 += 64 * WPB) {
            if (s0 + lane < STAGE_SLOTS)
                glds16(g + (size_t)(s0 + lane) * 4, __builtin_amdgcn_readfirstlane(l + (unsigned)s0 * 16));
        }
    };
    stage(0);  // the first centroid stage travels while the points are fetched

    // The labels of the iteration before (only compared at the very end) are requested now, so that the workgroup does
    // not end on a dependent load; inline asm because the compiler would sink a plain load to its use.
    // The load is unconditional (labels is always a valid [n] array; without n_changed the value is simply not used), so
    // the register is written by the asm on every path: no PHI and hence no compiler-made copy of a register whose load
    // is still in flight.  The load has landed at the first dma_wait() below (vmcnt(0) before the first stage barrier);
    // tools/check_assign_asm.py disassembles every instantiation and fails the build if the destination register is
    // touched between the load and that first vmcnt(0).
    int old_lab[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int64_t row = pbase + pt * 32 + j;
        if (row >= n) row = n - 1;
        asm volatile("global_load_dword %0, %1, off" : "=v"(old_lab[pt]) : "v"(labels + row) : "memory");
    }

    // ---- B operands: this wave's points, k = 2s + h for step s, kept in registers ----
    // Lanes j and j + 32 hold the same point.  Lane half h fetches the 16-byte slots 2m + h of the row -- each byte of
    // the row once per wave, D/8 loads per lane, in one or two round trips -- and the halves then trade what the other
    // needs with v_permlane32_swap: half 0 ends up with the even features, half 1 with the odd ones.  (Every lane
    // fetching the whole row and dropping half of it took D/4 loads in groups of 8: three to four dependent round
    // trips before the first MFMA, a fifth of a workgroup's life at the reference's K = 200.)
    constexpr int SLOTS = D / 8;                    // slots per lane
    constexpr int SG = SLOTS <= 12 ? SLOTS : 8;     // in flight at once (raw registers: 4 per slot)
    static_assert(SLOTS % SG == 0, "slot groups");
    float xr[PT][D / 2];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int64_t row = pbase + pt * 32 + j;
        if (row >= n) row = n - 1;  // tail rows are computed and discarded
        const float4 *xp = reinterpret_cast<const float4 *>(X + row * D);
#pragma unroll
        for (int m0 = 0; m0 < SLOTS; m0 += SG) {
            float4 v[SG];
#pragma unroll
            for (int g = 0; g < SG; ++g) v[g] = xp[2 * (m0 + g) + h];
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const int m = m0 + g;
                // half 0 holds slot 2m (features 8m .. 8m+3), half 1 slot 2m+1 (8m+4 .. 8m+7).  swap(a, b): the upper
                // lanes' a and the lower lanes' b change places.
                const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g].x), __float_as_uint(v[g].y), false, false);
                const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g].z), __float_as_uint(v[g].w), false, false);
                xr[pt][4 * m + 0] = __uint_as_float(xy[0]);  // feature 8m + h
                xr[pt][4 * m + 1] = __uint_as_float(zw[0]);  // 8m + 2 + h
                xr[pt][4 * m + 2] = __uint_as_float(xy[1]);  // 8m + 4 + h
                xr[pt][4 * m + 3] = __uint_as_float(zw[1]);  // 8m + 6 + h
            }
        }
    }

    float best[PT];
    int breg[PT], btile[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        best[pt] = __builtin_inff();
        breg[pt] = 0;
        btile[pt] = 0;
    }

    for (int t = 0; t < n_stages; ++t) {
        dma_wait();       // this wave's share of stage t has landed ...
        __syncthreads();  // ... and so has everyone else's; all waves are done with the other buffer
        if (t + 1 < n_stages) stage(t + 1);
        const float *sb = smem + (t & 1) * STAGE_F;
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            if (CTW > 1 && t * CTW + ct >= n_tiles) break;  // padding tile of the last stage (uniform)
            const float *tb = sb + ct * TILE_F;
            const float4 *arow = reinterpret_cast<const float4 *>(tb) + j * (R + 1) + h;
            f32x16 acc[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
            // A operands one u-step ahead of the MFMAs that use them (written as `an = next; mfma(a); a = an`
            // the two sets were merged into one register quad and every read was waited for at once).
            float4 a = arow[0];
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            {
                static_assert((D / 8) % 2 == 0, "u-steps are taken two at a time");
                // two explicit operand sets, alternating: the ds_read of step u+1 is issued BEFORE the four
                // MFMAs of step u (scheduling barriers pin the order; the register allocator cannot merge
                // the sets because both are live across the MFMAs), so its latency passes behind them
                float4 b = a;
#pragma unroll
                for (int u = 0; u < D / 8; u += 2) {
                    b = arow[2 * (u + 1)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[pt][4 * u + 0], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[pt][4 * u + 1], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[pt][4 * u + 2], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[pt][4 * u + 3], acc[pt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u + 2 < D / 8) a = arow[2 * (u + 2)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, xr[pt][4 * u + 4], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, xr[pt][4 * u + 5], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, xr[pt][4 * u + 6], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, xr[pt][4 * u + 7], acc[pt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (!EPI) {  // ablation: keep the accumulators alive, skip the epilogue
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) keep_alive(acc[pt]);
                continue;
            }
            // epilogue: d = c2 - 2*dot (one rounding), strict '<' in ascending centroid order.
            // The running argmin is kept as (tile, accumulator register): the register number
            // is an inline constant of the select, the tile is updated once per tile.
            float before[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) before[pt] = best[pt];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c2v = *reinterpret_cast<const float4 *>(tb + 32 * (D + 4) + 8 * g + 4 * h);
                const float c2a[4] = {c2v.x, c2v.y, c2v.z, c2v.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        const float dist = fmaf(-2.0f, acc[pt][4 * g + r], c2a[r]);
                        const bool lt = dist < best[pt];
                        best[pt] = lt ? dist : best[pt];
                        breg[pt] = lt ? (4 * g + r) : breg[pt];
                    }
                }
            }
            const int tile = t * CTW + ct;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) btile[pt] = (best[pt] < before[pt]) ? tile : btile[pt];
        }
    }

    // merge the two lane halves (rows 4h.. interleave): lexicographic (dist, index) minimum
    int changed = 0;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        // accumulator register r of lane half h is centroid row (r&3) + 8(r>>2) + 4h of its tile
        const int mine = btile[pt] * 32 + (breg[pt] & 3) + 8 * (breg[pt] >> 2) + 4 * h;
        const float ob = __shfl_xor(best[pt], 32);
        const int oi = __shfl_xor(mine, 32);
        const bool take = (ob < best[pt]) || (ob == best[pt] && oi < mine);
        const int lab = take ? oi : mine;
        const int64_t row = pbase + pt * 32 + j;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(old_lab[pt])::"memory");  // long landed (first dma_wait); the only consumer
        if (h == 0 && row < n) {
            labels[row] = lab;
            changed += (old_lab[pt] != lab);
        }
    }
    if (n_changed != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
        if (lane == 0 && changed != 0) atomicAdd(n_changed, (double)changed);
    }
}

// ------------------------------------------------------------------------------------------
// Restart-grouped E-step for the batched fit (KMeans(K, n_init=20) at the reference's K = 200 / 400).
//
// With one restart per workgroup (k_assign_mfma, blockIdx.y = restart) a workgroup at K = 200 lives for 7 centroid
// tiles, an eighth of which (24 of 224 rows) is padding, and its fixed costs -- fetching its points, the first
// centroid stage, the closing label write -- are paid once per 7 tiles.  Here a workgroup keeps its points in
// registers and scores them against a GROUP of R restarts whose centres form ONE image: row r * kp + j is centre
// j of the group's r-th restart (kp = k rounded up to 8).  R * kp rows are ceil(R * kp / 32) tiles -- 25 for
// R = 4, K = 200, no padding at all -- and the fixed costs are paid once per group.  A restart boundary falls on a
// multiple of 8 rows, which in the 32x32 accumulator layout is a boundary between register quads (rows 8q .. 8q+7
// are registers 4q .. 4q+3 of both lane halves): the running argmin is closed and re-opened between two quads,
// uniformly for the whole wave.  The arithmetic per (point, centroid) and the ascending order of the comparisons
// within a restart are those of k_assign_mfma: same labels (tests/test_gpu_batched.py compares the two).
// Restarts that have stopped keep their rows in the image; tiles that hold only such rows are skipped (no DMA, no
// MFMA), a tile shared with a live restart is computed and the stopped restart's quads ignored.
// ------------------------------------------------------------------------------------------
template <int D, int WPB>
__global__ __launch_bounds__(64 * WPB, 4) void k_assign_mfma_grouped(const float *__restrict__ X, int64_t n,
                                                                    const float *__restrict__ images, int kp, int R,
                                                                    int group_tiles, size_t image_floats,
                                                                    int32_t *__restrict__ labels, double *n_changed,
                                                                    const int32_t *state, MtpBatchStride bs) {
    constexpr int CTW = 2;
    constexpr int RS = D / 4;                  // 16-byte slots per centroid row
    constexpr int TILE_F = 32 * (D + 4) + 32;
    constexpr int STAGE_F = CTW * TILE_F;
    constexpr int STAGE_SLOTS = STAGE_F / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * STAGE_F floats

    const int b0 = blockIdx.y * R;                                  // first restart of this group
    const int rg = min(R, bs.batch - b0);                           // restarts in this group
    const float *staged = images + (size_t)blockIdx.y * image_floats;
    unsigned live = 0;                                              // wave-uniform: bit r = restart b0 + r still runs
    for (int r = 0; r < rg; ++r)
        if (!mtp_skip(state != nullptr ? state + (size_t)(b0 + r) * bs.flags : nullptr, bs.skip_mode)) live |= 1u << r;
    if (live == 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t pbase = ((int64_t)blockIdx.x * WPB + wave) * 32;
    const int64_t row = min(pbase + j, n - 1);  // tail rows are computed and discarded

    // a tile is worth computing if one of the restarts with rows in it is live: bit `tile` of tmask (the host keeps
    // group_tiles <= 64).  Walked restart by restart, no division.
    unsigned long long tmask = 0;
    for (int r = 0, lo = 0; r < rg; ++r, lo += kp) {
        if (!((live >> r) & 1u)) continue;
        const int t_lo = lo >> 5, t_hi = (lo + kp - 1) >> 5;
        tmask |= (t_hi - t_lo >= 63 ? ~0ull : ((1ull << (t_hi - t_lo + 1)) - 1)) << t_lo;
    }
    auto tile_live = [&](int tile) -> bool { return tile < group_tiles && ((tmask >> tile) & 1ull); };
    const int n_stages = (group_tiles + CTW - 1) / CTW;
    auto next_live_stage = [&](int t) -> int {
        while (t < n_stages && !(tile_live(CTW * t) || tile_live(CTW * t + 1))) ++t;
        return t;
    };
    const int wave_slot0 = wave * 64;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto stage = [&](int t, int buf) {
        const float *g = staged + (size_t)t * STAGE_F;
        const unsigned l = smem_base + (unsigned)(buf * STAGE_F * 4);
        for (int s0 = wave_slot0; s0 < STAGE_SLOTS; s0 += 64 * WPB) {
            if (s0 + lane < STAGE_SLOTS)
                glds16(g + (size_t)(s0 + lane) * 4, __builtin_amdgcn_readfirstlane(l + (unsigned)s0 * 16));
        }
    };
    int t = next_live_stage(0);
    int buf = 0;
    stage(t, buf);  // the first centroid stage travels while the points are fetched

    // The label of the iteration before is only compared when a restart is closed.  It is requested one restart
    // ahead -- the first live restart's here, the next live one's each time a restart is closed -- so that a close
    // does not wait for memory (the stage loop's vmcnt(0) waits have long covered the load by then).
    auto next_live_restart = [&](int r) -> int {
        while (r < rg && !((live >> r) & 1u)) ++r;
        return r;
    };
    int old_r = next_live_restart(0);  // the restart old_lab belongs to (uniform)
    int old_lab = (n_changed != nullptr && old_r < rg) ? labels[(size_t)(b0 + old_r) * bs.labels + row] : -1;

    // B operands: this wave's 32 points, as in k_assign_mfma (each half fetches every second 16-byte slot, the halves
    // trade with v_permlane32_swap)
    constexpr int SLOTS = D / 8;
    constexpr int SG = SLOTS <= 12 ? SLOTS : 8;
    static_assert(SLOTS % SG == 0, "slot groups");
    float xr[D / 2];
    {
        const float4 *xp = reinterpret_cast<const float4 *>(X + row * D);
#pragma unroll
        for (int m0 = 0; m0 < SLOTS; m0 += SG) {
            float4 v[SG];
#pragma unroll
            for (int g = 0; g < SG; ++g) v[g] = xp[2 * (m0 + g) + h];
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const int m = m0 + g;
                const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g].x), __float_as_uint(v[g].y), false, false);
                const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g].z), __float_as_uint(v[g].w), false, false);
                xr[4 * m + 0] = __uint_as_float(xy[0]);
                xr[4 * m + 1] = __uint_as_float(zw[0]);
                xr[4 * m + 2] = __uint_as_float(xy[1]);
                xr[4 * m + 3] = __uint_as_float(zw[1]);
            }
        }
    }

    float best = __builtin_inff();
    int breg = 0, btile = 0;
    int cur_r = 0;          // restart whose rows are being compared (uniform)
    int cur_end = kp;       // first image row of the next restart
    bool open = false;      // rows of cur_r have been compared

    // close restart r: merge the lane halves (lexicographic (dist, index) minimum = sklearn's strict '<' / first
    // index), write the labels, count the changed ones
    auto close_restart = [&](int r) {
        const int mine = btile * 32 + (breg & 3) + 8 * (breg >> 2) + 4 * h - r * kp;
        const float ob = __shfl_xor(best, 32);
        const int oi = __shfl_xor(mine, 32);
        const bool take = (ob < best) || (ob == best && oi < mine);
        const int lab = take ? oi : mine;
        int32_t *lp = labels + (size_t)(b0 + r) * bs.labels;
        const bool mine_to_write = h == 0 && pbase + j < n;
        if (mine_to_write) lp[pbase + j] = lab;
        if (n_changed != nullptr) {  // r == old_r: restarts are closed in ascending order, live ones only
            const int changed = __builtin_popcountll(__ballot(mine_to_write && old_lab != lab));
            if (lane == 0 && changed != 0) atomicAdd(n_changed + (size_t)(b0 + r) * bs.red, (double)changed);
            old_r = next_live_restart(r + 1);
            if (old_r < rg) old_lab = labels[(size_t)(b0 + old_r) * bs.labels + row];
        }
    };

    while (t < n_stages) {
        const int t_next = next_live_stage(t + 1);
        dma_wait();       // this wave's share of the stage has landed ...
        __syncthreads();  // ... and so has everyone else's; all waves are done with the other buffer
        if (t_next < n_stages) stage(t_next, buf ^ 1);
        const float *sb = smem + buf * STAGE_F;
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            const int tile = t * CTW + ct;
            if (!tile_live(tile)) continue;  // uniform
            const float *tb = sb + ct * TILE_F;
            const float4 *arow = reinterpret_cast<const float4 *>(tb) + j * (RS + 1) + h;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float4 a = arow[0];
            {
                static_assert((D / 8) % 2 == 0, "u-steps are taken two at a time");
                float4 b = a;
#pragma unroll
                for (int u = 0; u < D / 8; u += 2) {
                    b = arow[2 * (u + 1)];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[4 * u + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[4 * u + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[4 * u + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[4 * u + 3], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u + 2 < D / 8) a = arow[2 * (u + 2)];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, xr[4 * u + 4], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, xr[4 * u + 5], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, xr[4 * u + 6], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, xr[4 * u + 7], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // epilogue: d = c2 - 2*dot (one rounding), strict '<' in ascending row order; between two register quads
            // (8 image rows) the restart may change
            float before = best;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rowbase = tile * 32 + 8 * g;
                if (rowbase >= cur_end) {  // uniform, at most once per restart
                    btile = (best < before) ? tile : btile;
                    while (rowbase >= cur_end) {
                        if (open) close_restart(cur_r);
                        open = false;
                        ++cur_r;
                        cur_end += kp;
                    }
                    // re-open at the restart's own first row (a live restart's first tile is live, so this quad IS that
                    // row): a point that no distance beats -- NaN input -- still gets a label inside its restart's
                    // range (0) and not a negative one
                    best = __builtin_inff();
                    breg = 4 * g;
                    btile = tile;
                    before = best;
                }
                if (cur_r < rg && ((live >> cur_r) & 1u)) {
                    open = true;
                    const float4 c2v = *reinterpret_cast<const float4 *>(tb + 32 * (D + 4) + 8 * g + 4 * h);
                    const float c2a[4] = {c2v.x, c2v.y, c2v.z, c2v.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dist = fmaf(-2.0f, acc[4 * g + r], c2a[r]);
                        const bool lt = dist < best;
                        best = lt ? dist : best;
                        breg = lt ? (4 * g + r) : breg;
                    }
                }
            }
            btile = (best < before) ? tile : btile;
        }
        t = t_next;
        buf ^= 1;
    }
    if (open) close_restart(cur_r);
}
