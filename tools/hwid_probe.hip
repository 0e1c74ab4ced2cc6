// Development probe (not part of the product): where do the workgroups of a 2-per-CU grid land?  Records
// HW_ID / XCC_ID of every wave of a 512-thread, 68 KB-LDS workgroup grid, then prints how many distinct
// CUs were used, how many workgroups each held at once, which blockIdx pairs shared a CU and how the
// waves of a workgroup map to SIMDs.   Build: make -C tools hwid_probe    Run: tools/hwid_probe [grid]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ __launch_bounds__(512) void k_probe(unsigned *out, int spin) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID [3:0]
    const long long t0 = wall_clock64();
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    smem[threadIdx.x] = acc;
    __syncthreads();
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = hw;
        o[1] = xcc;
        o[2] = (unsigned)t0;
        o[3] = (unsigned)(t1 - t0) + (smem[0] == 12345.f);
    }
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 512;
    const int spin = argc > 2 ? atoi(argv[2]) : 20000;
    unsigned *d;
    CK(hipMalloc(&d, (size_t)grid * 8 * 4 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 68 * 1024, 0, d, spin);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> by_cu;
    int simd_rr = 0, simd_other = 0;
    for (int b = 0; b < grid; ++b) {
        const unsigned hw0 = h[(size_t)b * 32], xcc = h[(size_t)b * 32 + 1] & 15;
        const unsigned key = (xcc << 8) | ((hw0 >> 8) & 0xff);
        by_cu[key].push_back(b);
        const int s0 = (hw0 >> 4) & 3;
        for (int w = 0; w < 8; ++w) {
            const unsigned hw = h[((size_t)b * 8 + w) * 4];
            if ((((hw >> 8) & 0xff) != ((hw0 >> 8) & 0xff))) printf("block %d wave %d on another CU\n", b, w);
            if ((int)((hw >> 4) & 3) == ((s0 + w) & 3)) ++simd_rr; else ++simd_other;
        }
    }
    printf("grid %d: %zu distinct CUs; waves on SIMD (first + w) %% 4: %d, elsewhere: %d\n", grid, by_cu.size(), simd_rr, simd_other);
    std::map<size_t, int> hist;
    for (auto &kv : by_cu) hist[kv.second.size()]++;
    for (auto &kv : hist) printf("  CUs that held %zu workgroups: %d\n", kv.first, kv.second);
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 24) break;
        printf("  xcc %u cu-key 0x%02x:", kv.first >> 8, kv.first & 0xff);
        for (int b : kv.second) printf(" %d(simd0=%u)", b, (h[(size_t)b * 32] >> 4) & 3);
        printf("\n");
    }
    // first 16 blocks: raw
    for (int b = 0; b < 16 && b < grid; ++b)
        printf("  block %d: xcc %u hw 0x%08x simds", b, h[(size_t)b * 32 + 1] & 15, h[(size_t)b * 32]);
    printf("\n");
    for (int b = 0; b < 4 && b < grid; ++b) {
        printf("  block %d wave simds:", b);
        for (int w = 0; w < 8; ++w) printf(" %u", (h[((size_t)b * 8 + w) * 4] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
