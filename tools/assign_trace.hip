// Development probe (not part of the product): per-pass time line of the persistent E-step kernel.
// Build: make -C tools assign_trace     Run: tools/assign_trace [N] [K]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "assign_variants.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ void k_fill(float *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    }
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int k = argc > 2 ? atoi(argv[2]) : 1024;
    constexpr int D = 128;
    float *X, *C, *staged, *c2;
    int32_t *labels;
    long long *trace;
    CK(hipMalloc(&X, n * D * 4));
    CK(hipMalloc(&C, (size_t)k * D * 4));
    CK(hipMalloc(&staged, (size_t)mtp_n_tiles(k) * mtp_tile_floats(D) * 4));
    CK(hipMalloc(&c2, k * 4));
    CK(hipMalloc(&labels, n * 4));
    k_fill<<<4096, 256>>>(X, (size_t)n * D, 1);
    k_fill<<<256, 256>>>(C, (size_t)k * D, 7);
    if (mtp_launch_stage_centers(C, k, D, staged, c2, nullptr, 0) != 0) { printf("stage failed\n"); return 1; }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = 2 * prop.multiProcessorCount;
    CK(hipMalloc(&trace, (size_t)grid * 128 * 8));
    const size_t lds = 2 * (size_t)2 * mtp_tile_floats(D) * sizeof(float);
    auto kern = &k_assign_mfma_queue<D, 4, 8, true>;
    unsigned *tickets;
    CK(hipMalloc(&tickets, 256));
    CK(hipMemset(tickets, 0, 256));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int small_per_wg = argc > 3 ? atoi(argv[3]) : 2;
    const int64_t T4 = (n + 127) / 128;
    int64_t small = (int64_t)small_per_wg * grid;
    if (small > T4) small = T4;
    MtpAssignQueue q;
    q.ticket = tickets;
    q.big_per_restart = (int)((T4 - small) / 2);
    q.small_per_restart = (int)(T4 - 2 * (int64_t)q.big_per_restart);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(trace, 0, (size_t)grid * 128 * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, X, n, staged, (k + 31) / 32, labels, nullptr, nullptr,
                           mtp_no_batch(), q, trace);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> h((size_t)grid * 128);
    CK(hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = h[2];
    for (int g = 0; g < grid; ++g) t0 = std::min(t0, h[(size_t)g * 128 + 2]);
    // per workgroup: start, end, passes; per XCD: mean pass duration in the first / middle / last third
    double xs[8][3] = {}, xn[8][3] = {};
    long long end_max = 0, end_min = 1LL << 62, start_max = 0;
    std::vector<double> ends;
    for (int g = 0; g < grid; ++g) {
        const long long *t = &h[(size_t)g * 128];
        int np = 0;
        while (np < 126 && t[2 + np] != 0) ++np;
        const int xcc = (int)(t[1] >> 32) & 7;
        for (int p = 0; p + 1 < np; ++p) {
            const int third = p * 3 / (np - 1);
            xs[xcc][third] += (t[3 + p] - t[2 + p]) * 0.01;
            xn[xcc][third] += 1;
        }
        end_max = std::max(end_max, t[0] - t0);
        end_min = std::min(end_min, t[0] - t0);
        start_max = std::max(start_max, t[2] - t0);
        ends.push_back((t[0] - t0) * 0.01);
    }
    printf("N=%lld K=%d grid=%d: last start %.1f us, first end %.1f us, last end %.1f us\n", (long long)n, k, grid,
           start_max * 0.01, end_min * 0.01, end_max * 0.01);
    for (int x = 0; x < 8; ++x)
        printf("  xcc %d: mean pass us, first third %.2f  middle %.2f  last %.2f\n", x, xs[x][0] / xn[x][0],
               xs[x][1] / xn[x][1], xs[x][2] / xn[x][2]);
    std::sort(ends.begin(), ends.end());
    printf("  end times us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f\n", ends[0], ends[grid / 10], ends[grid / 2],
           ends[grid * 9 / 10], ends[grid - 1]);
    // pass durations of workgroup 0 and its CU partner
    for (int g : {0, grid / 2, 1, grid / 2 + 1}) {
        const long long *t = &h[(size_t)g * 128];
        printf("  wg %d (xcc %d hw 0x%05x) pass starts us:", g, (int)(t[1] >> 32) & 7, (unsigned)(t[1] & 0xfff00));
        for (int p = 0; p < 14 && t[2 + p] != 0; ++p) printf(" %.1f", (t[2 + p] - t0) * 0.01);
        printf(" ... end %.1f\n", (t[0] - t0) * 0.01);
    }
    return 0;
}
