// Development probe: what a read-only stream over a large buffer gets on this GPU, by load style.
//   tools/hbm_read_probe [GiB]
// Variants: plain 16-byte loads (U loads in flight per lane), the same non-temporal, LDS-DMA (global_load_lds_dwordx4)
// into a ring that nobody reads, and 96-byte-per-lane rows (the candidate layout of k_best_of_k: 6 x 16 B per lane).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_plain(const float4 *__restrict__ p, size_t n4, float *out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) {
                const float *q = reinterpret_cast<const float *>(p + i + u * stride);
                v[u].x = __builtin_nontemporal_load(q);
                v[u].y = __builtin_nontemporal_load(q + 1);
                v[u].z = __builtin_nontemporal_load(q + 2);
                v[u].w = __builtin_nontemporal_load(q + 3);
            } else {
                v[u] = p[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// contiguous per workgroup: each workgroup owns a contiguous span (what a "tile" kernel does), U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_span(const float4 *__restrict__ p, size_t n4, float *out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    float acc = 0.f;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) {
                const float *q = reinterpret_cast<const float *>(p + i + u * 256);
                v[u].x = __builtin_nontemporal_load(q);
                v[u].y = __builtin_nontemporal_load(q + 1);
                v[u].z = __builtin_nontemporal_load(q + 2);
                v[u].w = __builtin_nontemporal_load(q + 3);
            } else {
                v[u] = p[i + u * 256];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// one lane = one 96-byte row (6 x 16 B), like a candidate trajectory of k_best_of_k
template <bool NT>
__global__ __launch_bounds__(256) void k_rows96(const float4 *__restrict__ p, size_t rows, float *out) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float4 *q = p + r * 6;
    float4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        if (NT) {
            const float *f = reinterpret_cast<const float *>(q + u);
            v[u].x = __builtin_nontemporal_load(f);
            v[u].y = __builtin_nontemporal_load(f + 1);
            v[u].z = __builtin_nontemporal_load(f + 2);
            v[u].w = __builtin_nontemporal_load(f + 3);
        } else {
            v[u] = q[u];
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 12345.678f) out[0] = acc;
}

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr_uniform) {
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
// LDS-DMA: every wave streams its share into a private 4 KB ring slot set, DEPTH instructions in flight, nothing reads it
template <int DEPTH>
__global__ __launch_bounds__(256) void k_dma(const float4 *__restrict__ p, size_t n4, float *out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    const unsigned wbase = base + wave * DEPTH * 1024;
    size_t i = lo + wave * 64 + lane;
    int slot = 0;
    for (; i < hi; i += 256) {
        glds16(p + i, __builtin_amdgcn_readfirstlane(wbase + slot * 1024));
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        if (slot == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 12345.678f) out[0] = 1.f;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 9.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / 4096 * 4096;
    const size_t n4 = bytes / 16;
    float4 *p;
    float *out;
    CK(hipMalloc(&p, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(p, 1, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0 && t < best) best = t;
        }
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
        fflush(stdout);
    };
    for (int wpc : {4, 8, 16}) {
        const int grid = cus * wpc;
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride U=4, %d wg/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL((k_plain<4, false>), dim3(grid), dim3(256), 0, 0, p, n4, out); });
        snprintf(nm, sizeof nm, "grid-stride U=8, %d wg/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL((k_plain<8, false>), dim3(grid), dim3(256), 0, 0, p, n4, out); });
        snprintf(nm, sizeof nm, "grid-stride U=8 nt, %d wg/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL((k_plain<8, true>), dim3(grid), dim3(256), 0, 0, p, n4, out); });
    }
    for (int spans : {8, 64, 512}) {
        const int grid = cus * spans;
        char nm[96];
        snprintf(nm, sizeof nm, "spans U=6, %d spans/CU", spans);
        time(nm, [&] { hipLaunchKernelGGL((k_span<6, false>), dim3(grid), dim3(256), 0, 0, p, n4, out); });
        snprintf(nm, sizeof nm, "spans U=6 nt, %d spans/CU", spans);
        time(nm, [&] { hipLaunchKernelGGL((k_span<6, true>), dim3(grid), dim3(256), 0, 0, p, n4, out); });
    }
    {
        const size_t rows = n4 / 6;
        const unsigned grid = (unsigned)((rows + 255) / 256);
        time("96-byte rows, one per lane", [&] { hipLaunchKernelGGL((k_rows96<false>), dim3(grid), dim3(256), 0, 0, p, rows, out); });
        time("96-byte rows, one per lane, nt", [&] { hipLaunchKernelGGL((k_rows96<true>), dim3(grid), dim3(256), 0, 0, p, rows, out); });
    }
    for (int wpc : {2, 4, 8}) {
        const int grid = cus * wpc;
        char nm[96];
        snprintf(nm, sizeof nm, "LDS-DMA depth 8, %d wg/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL((k_dma<8>), dim3(grid), dim3(256), 4 * 8 * 1024, 0, p, n4, out); });
        if (wpc <= 4) {
            snprintf(nm, sizeof nm, "LDS-DMA depth 16, %d wg/CU", wpc);
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16 * 1024));
            time(nm, [&] { hipLaunchKernelGGL((k_dma<16>), dim3(grid), dim3(256), 4 * 16 * 1024, 0, p, n4, out); });
        }
    }
    return 0;
}
