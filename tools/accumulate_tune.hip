// Development micro-benchmark (not part of the product): variants of the centroid-sum kernel.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#include "accumulate_kernel.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __host__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void k_fill(float *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = ((int)(hash32((unsigned)i * 2654435761u + seed) >> 8) - (1 << 23)) * (1.0f / (1 << 23));
}
__global__ void k_labels(int32_t *p, size_t n, int k, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = hash32((unsigned)i + seed) % k;
}

struct Cfg { size_t lds_bytes; int threads; int max_wg; };

template <int LPR, int G, typename ACC, bool DO_ADD, int DT = 0, bool NT = false>
void launch_v(const float *X, int64_t n, int d, const int32_t *labels, int k, double *sums, double *counts, Cfg c) {
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_accumulate<LPR, G, ACC, DO_ADD, DT, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        set = true;
    }
    int kr = (int)(c.lds_bytes / ((size_t)d * sizeof(ACC) + 4));
    if (kr > k) kr = k;
    const int n_ranges = (k + kr - 1) / kr;
    kr = (k + n_ranges - 1) / n_ranges;
    const size_t lds = (size_t)kr * d * sizeof(ACC) + (size_t)kr * 4;
    int64_t n_chunks = c.max_wg / n_ranges;
    if (n_chunks < 1) n_chunks = 1;
    int64_t rpc = (n + n_chunks - 1) / n_chunks;
    rpc = (rpc + 63) / 64 * 64;
    n_chunks = (n + rpc - 1) / rpc;
    hipLaunchKernelGGL((k_accumulate<LPR, G, ACC, DO_ADD, DT, NT>), dim3((unsigned)n_chunks, (unsigned)n_ranges), dim3(c.threads),
                       lds, 0, X, n, d, labels, k, kr, rpc, sums, counts, nullptr, mtp_no_batch());
    CK(hipGetLastError());
}

struct Variant {
    const char *name;
    void (*launch)(const float *, int64_t, int, const int32_t *, int, double *, double *, Cfg);
    Cfg cfg;
    bool check;
};

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int k = argc > 2 ? atoi(argv[2]) : 1024;
    const int d = 128;
    float *X;
    int32_t *labels;
    double *sums, *counts;
    CK(hipMalloc(&X, n * d * 4));
    CK(hipMalloc(&labels, n * 4));
    CK(hipMalloc(&sums, (size_t)k * d * 8));
    CK(hipMalloc(&counts, k * 8));
    k_fill<<<4096, 256>>>(X, (size_t)n * d, 1);
    k_labels<<<4096, 256>>>(labels, n, k, 5);
    CK(hipDeviceSynchronize());
    const Cfg big = {152 * 1024, 1024, 256}, half = {76 * 1024, 1024, 512}, half512 = {76 * 1024, 512, 512};
    const Cfg r8 = {132 * 1024, 1024, 256};  // K = 1024, D = 128: 8 cluster ranges of 128 x 32 row chunks = 256 workgroups
    std::vector<Variant> vs = {
        {"f64 G4 1024thr 152KB (product)", launch_v<32, 4, double, true>, big, true},
        {"f64 G4 d=128 const", launch_v<32, 4, double, true, 128>, big, true},
        {"f64 G4 d=128 const, nt rows", launch_v<32, 4, double, true, 128, true>, big, true},
        {"f64 G4 d=128 nt, 8 ranges x 32", launch_v<32, 4, double, true, 128, true>, r8, true},
        {"f64 G8 d=128 const, nt rows", launch_v<32, 8, double, true, 128, true>, big, true},
        {"f64 G4 no-add d=128 nt (load only)", launch_v<32, 4, double, false, 128, true>, big, false},
        {"f64 G8 1024thr 152KB", launch_v<32, 8, double, true>, big, true},
        {"f64 G8 no-add (load only)", launch_v<32, 8, double, false>, big, false},
        {"f64 G8 1024thr 76KB x2/CU", launch_v<32, 8, double, true>, half, true},
        {"f32 G8 1024thr 152KB", launch_v<32, 8, float, true>, big, true},
        {"f32 G8 1024thr 76KB x2/CU", launch_v<32, 8, float, true>, half, true},
        {"f32 G4 512thr 76KB x2/CU", launch_v<32, 4, float, true>, half512, true},
        {"f32 G8 no-add 76KB x2/CU", launch_v<32, 8, float, false>, half, false},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> ref((size_t)k * d), got((size_t)k * d);
    const int rounds = 4;
    std::vector<std::vector<float>> ms(vs.size());
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < vs.size(); ++v) {
            CK(hipMemset(sums, 0, (size_t)k * d * 8));
            CK(hipMemset(counts, 0, k * 8));
            CK(hipEventRecord(e0, 0));
            vs[v].launch(X, n, d, labels, k, sums, counts, vs[v].cfg);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms[v].push_back(t);
            if (r == 0 && vs[v].check) {
                CK(hipMemcpy(got.data(), sums, (size_t)k * d * 8, hipMemcpyDeviceToHost));
                if (v == 0) ref = got;
                double worst = 0;
                for (size_t i = 0; i < ref.size(); ++i) worst = fmax(worst, fabs(got[i] - ref[i]) / fmax(1.0, fabs(ref[i])));
                if (worst > 1e-3) printf("  !! %s: max rel diff %.3g vs product\n", vs[v].name, worst);
            }
        }
    const double bytes = (double)n * d * 4 + (double)n * 4 + (double)k * d * 8 + k * 8;
    for (size_t v = 0; v < vs.size(); ++v) {
        float mn = 1e30f;
        for (int r = 1; r < rounds; ++r) mn = ms[v][r] < mn ? ms[v][r] : mn;
        printf("%-34s min %7.3f ms  %6.0f GB/s (%4.1f%% of 8000)\n", vs[v].name, mn, bytes / (mn * 1e-3) / 1e9,
               bytes / (mn * 1e-3) / 1e9 / 80);
    }
    return 0;
}
