// E-step kernel variants that were measured and NOT kept (DESIGN.md 3, K1+K2 ablations).  Tuner-only:
// instantiated by tools/assign_tune.hip and tools/assign_trace.hip, never by the product library.
#pragma once
#include "assign_kernel.h"

// ------------------------------------------------------------------------------------------
// Software-pipelined variant (one 32-point tile per wave, two centroid tiles per LDS stage): the
// bias / argmin epilogue of tile i is issued BETWEEN the MFMAs of tile i+1 (two accumulator sets),
// so its ~64 vector ops run while the matrix pipe works instead of leaving the wave's share of the
// pipe idle.  ||c||^2 of all tiles is copied to a persistent LDS array first, because the stage
// buffer a deferred epilogue would read it from is already being refilled by then.  Same
// arithmetic in the same order per (point, centroid) as k_assign_mfma: same labels.
// Dynamic LDS: 2 * STAGE_F floats + 32 * n_tiles floats.
// ------------------------------------------------------------------------------------------
template <int D, int WPB, int WPS, int LG = 8>
__global__ __launch_bounds__(64 * WPB, WPS) void k_assign_mfma_pipe(const float *__restrict__ X, int64_t n,
                                                             const float *__restrict__ staged, int n_tiles,
                                                             int32_t *__restrict__ labels, double *n_changed,
                                                             const int32_t *state, MtpBatchStride bs) {
    constexpr int CTW = 2;
    constexpr int R = D / 4;
    constexpr int U = D / 8;                 // u-steps (4 MFMAs each) per tile
    constexpr int TILE_F = 32 * (D + 4) + 32;
    constexpr int STAGE_F = CTW * TILE_F;
    constexpr int STAGE_SLOTS = STAGE_F / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *c2all = smem + 2 * STAGE_F;
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
    const int64_t pbase = ((int64_t)blockIdx.x * WPB + wave) * 32;

    float xr[D / 2];
    {
        int64_t row = pbase + j;
        if (row >= n) row = n - 1;
        const float4 *xp = reinterpret_cast<const float4 *>(X + row * D);
#pragma unroll
        for (int q0 = 0; q0 < D / 4; q0 += LG) {
            float4 v[LG];
#pragma unroll
            for (int g = 0; g < LG; ++g) v[g] = xp[q0 + g];
#pragma unroll
            for (int g = 0; g < LG; ++g) {
                const int q = q0 + g;
                xr[2 * q] = h ? v[g].y : v[g].x;
                xr[2 * q + 1] = h ? v[g].w : v[g].z;
                pin2(xr[2 * q], xr[2 * q + 1]);
            }
        }
    }
    for (int e = tid; e < n_tiles * 32; e += 64 * WPB)
        c2all[e] = staged[(size_t)(e >> 5) * TILE_F + 32 * (D + 4) + (e & 31)];

    float best = __builtin_inff();
    int breg = 0, btile = 0;

    const int n_stages = n_tiles / CTW;
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
    // one accumulator register of a finished tile: d = c2 - 2*dot, strict '<' in ascending row order
    auto epi_reg = [&](const f32x16 &acc, int r, const float4 &c2v) {
        const float c2 = (r & 3) == 0 ? c2v.x : (r & 3) == 1 ? c2v.y : (r & 3) == 2 ? c2v.z : c2v.w;
        const float dist = fmaf(-2.0f, acc[r], c2);
        const bool lt = dist < best;
        best = lt ? dist : best;
        breg = lt ? r : breg;
    };
    // MFMAs of the tile at tb into `cur`; when have_prev, the epilogue of `prev` (tile index
    // prev_tile) is spread over the u-steps, 16 / U registers per step
    auto tile_pass = [&](const float *tb, f32x16 &cur, const f32x16 &prev, int prev_tile, auto have_prev) {
        const float4 *arow = reinterpret_cast<const float4 *>(tb) + j * (R + 1) + h;
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[r] = 0.f;
        float before = best;
        float4 c2v = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 a = arow[0];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 an = a;
            if (u + 1 < U) an = arow[2 * (u + 1)];
            cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[4 * u + 0], cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[4 * u + 1], cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[4 * u + 2], cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[4 * u + 3], cur, 0, 0, 0);
            if constexpr (decltype(have_prev)::value) {
#pragma unroll
                for (int r = (16 * u) / U; r < (16 * (u + 1)) / U; ++r) {
                    if ((r & 3) == 0)
                        c2v = *reinterpret_cast<const float4 *>(c2all + prev_tile * 32 + 8 * (r >> 2) + 4 * h);
                    epi_reg(prev, r, c2v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a = an;
        }
        if constexpr (decltype(have_prev)::value) btile = (best < before) ? prev_tile : btile;
    };

    f32x16 accA, accB;
    stage(0);
    for (int t = 0; t < n_stages; ++t) {
        dma_wait();
        __syncthreads();
        if (t + 1 < n_stages) stage(t + 1);
        const float *sb = smem + (t & 1) * STAGE_F;
        if (t == 0)
            tile_pass(sb, accA, accB, 0, std::false_type{});
        else
            tile_pass(sb, accA, accB, 2 * t - 1, std::true_type{});
        tile_pass(sb + TILE_F, accB, accA, 2 * t, std::true_type{});
    }
    {   // the last tile's epilogue has nothing left to hide behind
        const float before = best;
        const int tile = n_tiles - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 c2v = *reinterpret_cast<const float4 *>(c2all + tile * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_reg(accB, 4 * g + r, c2v);
        }
        btile = (best < before) ? tile : btile;
    }

    const int mine = btile * 32 + (breg & 3) + 8 * (breg >> 2) + 4 * h;
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(mine, 32);
    const bool take = (ob < best) || (ob == best && oi < mine);
    const int lab = take ? oi : mine;
    const int64_t row = pbase + j;
    int changed = 0;
    if (h == 0 && row < n) {
        const int old = labels[row];
        labels[row] = lab;
        changed = (old != lab);
    }
    if (n_changed != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
        if (lane == 0 && changed != 0) atomicAdd(n_changed, (double)changed);
    }
}

// ------------------------------------------------------------------------------------------
// Persistent E-step with a work queue and half-size units at the end -- TUNER ONLY (tools/assign_tune,
// tools/assign_trace); the product launches k_assign_mfma.  Kept because it is the measured answer to "would a
// persistent grid remove the tail of a short launch?": it does not (DESIGN.md, K2 ablations of round 2).
//
// What the plain grid loses on a launch of a few rounds (N = 9.0 ... 10.25 rounds of 512 workgroups): the
// hardware places the last, partial round two workgroups per CU on some CUs and none on others, so the launch
// time moves in steps of half a round, and N = 1.25 M (9.54 rounds) pays for 10.0.  Variants tried here:
//   * equal static shares per workgroup (per CU, in quads of 4 wave-tiles): 2 % slower at N = 10 M -- of the
//     two workgroups on a CU the older one gets 58 % of the matrix pipe, finishes early and leaves the other
//     alone, and the XCDs differ by 1-4 % in speed, which only a dynamic hand-out absorbs;
//   * one ticket per CU pair (workgroups g and g + gridDim.x / 2 share a CU, tools/hwid_probe): the pair
//     balances, the CUs do not (end times spread over 2 %);
//   * this kernel: one global ticket; big units = 2 quads = 8 wave-tiles of 32 points (a workgroup of the
//     plain kernel), and for the last small_per_restart quads of every restart small units = 1 quad, where
//     waves w and w + 4 (they share a SIMD) take the same 32 points and every other centroid tile each and
//     merge their (distance, index) minima through LDS.  Same speed as the plain grid at N = 10 M (88.2 vs
//     88.9 %), within +-1 % of it at N = 1.18 ... 1.34 M, 7 % slower at the reference's K = 200 (the ticket
//     is on the critical path of a 36 us unit): what is in flight when the queue runs dry is one unit of the
//     SLOWER workgroup of a pair either way.
// The centroid stream continues across units (stage 0 of the next unit is fetched during the last stage of
// this one).  *ticket is zero at entry; the workgroup that draws the last past-the-end ticket zeroes it.
// Same arithmetic per (point, centroid) in the same order as k_assign_mfma, lexicographic (distance, index)
// merge: same labels (the tuner checks).
// ------------------------------------------------------------------------------------------
struct MtpAssignQueue {
    unsigned *ticket;
    int big_per_restart;    // 2-quad units per restart; they cover quads [0, 2 big)
    int small_per_restart;  // 1-quad units per restart: quads [2 big, 2 big + small) reach the end of the points
};

template <int D, int WPS, int LG = 8, bool TRACE = false>
__global__ __launch_bounds__(512, WPS) void k_assign_mfma_queue(const float *__restrict__ X, int64_t n,
                                                                const float *__restrict__ staged0, int n_tiles,
                                                                int32_t *__restrict__ labels0, double *n_changed0,
                                                                const int32_t *state0, MtpBatchStride bs,
                                                                MtpAssignQueue q, long long *trace = nullptr) {
    constexpr int CTW = 2;  // the small units split a stage's two tiles between a wave pair
    constexpr int WPB = 8;
    constexpr int R = D / 4;
    constexpr int TILE_F = 32 * (D + 4) + 32;
    constexpr int STAGE_F = CTW * TILE_F;
    constexpr int STAGE_SLOTS = STAGE_F / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * STAGE_F floats
    __shared__ unsigned s_ticket[2];
    __shared__ float s_best[4][32];
    __shared__ int s_lab[4][32];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n_stages = (n_tiles + CTW - 1) / CTW;
    const int wave_slot0 = wave * 64;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const int batch = bs.batch;

    auto runs = [&](int b) { return !(state0 != nullptr && mtp_skip(state0 + (size_t)b * bs.flags, bs.skip_mode)); };
    auto stage = [&](int b, int t, int buf) {
        const float *g = staged0 + (size_t)b * bs.staged + (size_t)t * STAGE_F;
        const unsigned l = smem_base + (unsigned)(buf * STAGE_F * 4);
        for (int s0 = wave_slot0; s0 < STAGE_SLOTS; s0 += 64 * WPB) {
            if (s0 + lane < STAGE_SLOTS)
                glds16(g + (size_t)(s0 + lane) * 4, __builtin_amdgcn_readfirstlane(l + (unsigned)s0 * 16));
        }
    };

    // the queue: big units of every running restart (restart-major), then the small units likewise
    int n_run = 0;
    for (int b = 0; b < batch; ++b) n_run += runs(b) ? 1 : 0;
    const unsigned n_big = (unsigned)q.big_per_restart * n_run;
    const unsigned n_units = n_big + (unsigned)q.small_per_restart * n_run;
    // ticket -> (restart, first quad, kind: 2 = big, 1 = small, 0 = past the end)
    int cur_b = -1, cur_act = -1;  // cursor over the running restarts
    auto decode = [&](unsigned tk, int &b, int64_t &quad0, int &kind) {
        if (tk >= n_units) {
            kind = 0;
            return;
        }
        int want;
        if (tk < n_big) {
            want = (int)(tk / (unsigned)q.big_per_restart);
            quad0 = (int64_t)(tk - (unsigned)want * q.big_per_restart) * 2;
            kind = 2;
        } else {
            const unsigned v = tk - n_big;
            want = (int)(v / (unsigned)q.small_per_restart);
            quad0 = (int64_t)q.big_per_restart * 2 + (v - (unsigned)want * q.small_per_restart);
            kind = 1;
        }
        if (want < cur_act) cur_b = cur_act = -1;  // the small units start again at the first restart
        while (cur_act < want) {
            ++cur_b;
            if (runs(cur_b)) ++cur_act;
        }
        b = cur_b;
    };
    // one lane; the last of the gridDim.x past-the-end tickets puts the counter back to zero
    auto draw = [&]() -> unsigned {
        const unsigned tk = atomicAdd(q.ticket, 1u);
        if (tk == n_units + gridDim.x - 1) atomicExch(q.ticket, 0u);
        return tk;
    };

    if (tid == 0) s_ticket[0] = draw();
    __syncthreads();
    int slot = 0;
    int b = 0, kind = 0;
    int64_t quad0 = 0;
    decode(s_ticket[0], b, quad0, kind);
    int buf = 0;                // LDS buffer of the next stage to be consumed
    bool staged_ahead = false;  // stage 0 of this unit is already on its way into `buf`
    int trace_n = 0;
    while (kind > 0) {
        if constexpr (TRACE) {
            if (tid == 0 && trace_n < 126) trace[(size_t)blockIdx.x * 128 + 2 + trace_n++] = wall_clock64();
        }
        if (tid == 0) s_ticket[slot ^ 1] = draw();  // published by this unit's first barrier

        const bool split = kind == 1;  // wave pair (w, w + 4) shares the tile: wave-half hs takes centroid tile hs
        const int hs = wave >> 2;
        const int64_t pbase = split ? (quad0 * 4 + (wave & 3)) * 32 : ((quad0 + hs) * 4 + (wave & 3)) * 32;
        const bool active = pbase < n;  // wave-uniform

        float xr[D / 2];
        if (active) {
            int64_t row = pbase + j;
            if (row >= n) row = n - 1;
            const float4 *xp = reinterpret_cast<const float4 *>(X + row * D);
#pragma unroll
            for (int q0 = 0; q0 < D / 4; q0 += LG) {
                float4 v[LG];
#pragma unroll
                for (int g = 0; g < LG; ++g) v[g] = xp[q0 + g];
#pragma unroll
                for (int g = 0; g < LG; ++g) {
                    const int qq = q0 + g;
                    xr[2 * qq] = h ? v[g].y : v[g].x;
                    xr[2 * qq + 1] = h ? v[g].w : v[g].z;
                    pin2(xr[2 * qq], xr[2 * qq + 1]);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < D / 2; ++e) xr[e] = 0.f;
        }
        float best = __builtin_inff();
        int breg = 0, btile = 0;

        if (!staged_ahead) stage(b, 0, buf);
        staged_ahead = false;
        int nb = 0, nkind = 0;
        int64_t nquad0 = 0;
        for (int t = 0; t < n_stages; ++t) {
            dma_wait();
            __syncthreads();
            if (t + 1 < n_stages) {
                stage(b, t + 1, buf ^ 1);
            } else {  // last stage of this unit: start on the next unit's centroids
                decode(s_ticket[slot ^ 1], nb, nquad0, nkind);
                if (nkind > 0) {
                    stage(nb, 0, buf ^ 1);
                    staged_ahead = true;
                }
            }
            const float *sb = smem + buf * STAGE_F;
            buf ^= 1;
            if (!active) continue;
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                if (t * CTW + ct >= n_tiles) break;
                if (split && ct != hs) continue;
                const float *tb = sb + ct * TILE_F;
                const float4 *arow = reinterpret_cast<const float4 *>(tb) + j * (R + 1) + h;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                float4 a = arow[0];
                static_assert((D / 8) % 2 == 0, "u-steps are taken two at a time");
                float4 bq = a;
#pragma unroll
                for (int u = 0; u < D / 8; u += 2) {
                    bq = arow[2 * (u + 1)];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[4 * u + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[4 * u + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[4 * u + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[4 * u + 3], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u + 2 < D / 8) a = arow[2 * (u + 2)];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.x, xr[4 * u + 4], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.y, xr[4 * u + 5], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.z, xr[4 * u + 6], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.w, xr[4 * u + 7], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float before = best;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
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
                btile = (best < before) ? (t * CTW + ct) : btile;
            }
        }
        // merge the two lane halves: lexicographic (distance, index) minimum
        int lab;
        {
            const int mine = btile * 32 + (breg & 3) + 8 * (breg >> 2) + 4 * h;
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(mine, 32);
            const bool take = (ob < best) || (ob == best && oi < mine);
            lab = take ? oi : mine;
            best = take ? ob : best;
        }
        if (split) {  // workgroup-uniform: the upper wave of every pair hands its minimum to the lower one
            if (hs == 1 && h == 0) {
                s_best[wave & 3][j] = best;
                s_lab[wave & 3][j] = lab;
            }
            __syncthreads();
            if (hs == 0) {
                const float ob = s_best[wave][j];
                const int oi = s_lab[wave][j];
                const bool take = (ob < best) || (ob == best && oi < lab);
                lab = take ? oi : lab;
            }
        }
        if (active && (!split || hs == 0)) {
            const int64_t row = pbase + j;
            int32_t *labels = labels0 + (size_t)b * bs.labels;
            int changed = 0;
            if (h == 0 && row < n) {
                const int old = labels[row];
                labels[row] = lab;
                changed = (old != lab);
            }
            if (n_changed0 != nullptr) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
                if (lane == 0 && changed != 0) atomicAdd(n_changed0 + (size_t)b * bs.red, (double)changed);
            }
        }
        // no barrier here: the next unit's first stage barrier orders the reuse of the LDS buffers (a stage
        // is only overwritten one barrier after every wave has finished reading it), of s_ticket and s_best
        slot ^= 1;
        b = nb;
        quad0 = nquad0;
        kind = nkind;
    }
    if constexpr (TRACE) {
        if (tid == 0) {
            trace[(size_t)blockIdx.x * 128] = wall_clock64();
            trace[(size_t)blockIdx.x * 128 + 1] =
                ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
}

// ------------------------------------------------------------------------------------------
// "Direct" variant: no LDS, no barriers, no LDS-DMA.  The centroid operands come straight from global memory
// (L1 / L2: the image is 0.5 MB and every wave of the chip walks it front to back) with plain 16-byte loads, PF
// u-steps ahead of the MFMAs that use them; the image is laid out so that a wave's load is one contiguous KB:
//   tile t, u-step u, lane l  ->  float4 at ((t * (8 D + 8)) + u * 64 + l),   then the tile's 32 ||c||^2.
// The waves of a workgroup are independent (the workgroup only exists to share a launch).  Same arithmetic in the
// same order per (point, centroid) as k_assign_mfma: same labels.
// ------------------------------------------------------------------------------------------
template <int D>
__host__ __device__ constexpr int mtp_direct_tile_f4() { return 8 * D + 8; }  // float4 per tile: 32*D/4 operands + 32/4 norms

// staged image (mtp_internal.h layout) -> direct layout; one block per tile
template <int D>
__global__ __launch_bounds__(256) void k_direct_image(const float *__restrict__ staged, float *__restrict__ out) {
    constexpr int R = D / 4, U = D / 8, TILE_F = 32 * (D + 4) + 32;
    const float4 *src = reinterpret_cast<const float4 *>(staged + (size_t)blockIdx.x * TILE_F);
    float4 *dst = reinterpret_cast<float4 *>(out) + (size_t)blockIdx.x * mtp_direct_tile_f4<D>();
    for (int e = threadIdx.x; e < U * 64; e += blockDim.x) {
        const int u = e >> 6, l = e & 63, j = l & 31, h = l >> 5;
        dst[e] = src[j * (R + 1) + 2 * u + h];
    }
    const float *c2 = staged + (size_t)blockIdx.x * TILE_F + 32 * (D + 4);
    float *c2o = reinterpret_cast<float *>(dst + U * 64);
    for (int e = threadIdx.x; e < 32; e += blockDim.x) c2o[e] = c2[e];
}

template <int D, int WPB, int PF, bool EPI = true>
__global__ __launch_bounds__(64 * WPB, 4) void k_assign_mfma_direct(const float *__restrict__ X, int64_t n,
                                                                   const float *__restrict__ img, int n_tiles,
                                                                   int32_t *__restrict__ labels) {
    constexpr int U = D / 8;
    static_assert(U % PF == 0, "the operand ring must divide the u-steps of a tile");
    constexpr int TQ = mtp_direct_tile_f4<D>();
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t pbase = ((int64_t)blockIdx.x * WPB + wave) * 32;
    const float4 *img4 = reinterpret_cast<const float4 *>(img);  // uniform base; lanes add their own 16 bytes

    // the first PF operand quads travel while the points are fetched
    float4 a[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) a[q] = img4[q * 64 + lane];

    constexpr int SLOTS = D / 8;
    constexpr int SG = SLOTS <= 12 ? SLOTS : 8;
    float xr[D / 2];
    {
        int64_t row = pbase + j;
        if (row >= n) row = n - 1;
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
    for (int tile = 0; tile < n_tiles; ++tile) {
        const float4 *tp = img4 + (size_t)tile * TQ;           // uniform
        const int nt = tile + 1 < n_tiles ? tile + 1 : tile;  // beyond the last tile: a harmless re-read
        const float4 *np = img4 + (size_t)nt * TQ;
        float4 c2v[4];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 cur = a[u % PF];
            __builtin_amdgcn_sched_barrier(0);
            a[u % PF] = (u + PF < U) ? tp[(u + PF) * 64 + lane] : np[(u + PF - U) * 64 + lane];
            if (EPI && u == U - 2) {
                // this tile's norms, two u-steps before the epilogue (lane half h needs rows 8 g + 4 h .. + 3 of quad g)
#pragma unroll
                for (int g = 0; g < 4; ++g) c2v[g] = tp[U * 64 + 2 * g + h];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x, xr[4 * u + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.y, xr[4 * u + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.z, xr[4 * u + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.w, xr[4 * u + 3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!EPI) {
            keep_alive(acc);
            continue;
        }
        const float before = best;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float c2a[4] = {c2v[g].x, c2v[g].y, c2v[g].z, c2v[g].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dist = fmaf(-2.0f, acc[4 * g + r], c2a[r]);
                const bool lt = dist < best;
                best = lt ? dist : best;
                breg = lt ? (4 * g + r) : breg;
            }
        }
        btile = (best < before) ? tile : btile;
    }
    const int mine = btile * 32 + (breg & 3) + 8 * (breg >> 2) + 4 * h;
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(mine, 32);
    const bool take = (ob < best) || (ob == best && oi < mine);
    const int64_t row = pbase + j;
    if (h == 0 && row < n) labels[row] = take ? oi : mine;
}
