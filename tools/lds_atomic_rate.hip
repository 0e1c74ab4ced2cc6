// LDS atomic-add rates on gfx950 (development probe): how many lane-atomics per clock a CU sustains for
// double / 64-bit integer / float / 32-bit integer adds, for the access shape of k_accumulate (a wave
// instruction = two rows of 24..32 consecutive accumulators at unrelated offsets).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_rate.hip -o /tmp/lds_atomic_rate && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

// W consecutive accumulators per "row", 64 / RW rows per wave instruction (RW >= W: lanes c >= W of a row idle)
template <typename T, int RW, int W>
__global__ __launch_bounds__(1024) void k_rate(const int *__restrict__ offs, int iters, int span, T *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    T *acc = reinterpret_cast<T *>(raw);
    for (int e = threadIdx.x; e < span; e += blockDim.x) acc[e] = (T)0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWS = 64 / RW;
    const int sub = lane / RW, c = lane % RW;
    T v = (T)(lane + 1);
    for (int i = 0; i < iters; ++i) {
        const int base = offs[((wave * iters + i) * ROWS + sub) & 65535];
        T *a = acc + base + c;
        if (c < W) {
            atomicAdd(a, v);
            atomicAdd(a + W, v);
            atomicAdd(a + 2 * W, v);
            atomicAdd(a + 3 * W, v);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc[0] + acc[span - 1];
}

template <typename T, int RW, int W>
static void run(const char *name, const int *d_offs, int span_elems) {
    const int iters = 2000, blocks = 256;
    T *sink;
    hipMalloc(&sink, blocks * sizeof(T));
    const size_t lds = (size_t)span_elems * sizeof(T);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_rate<T, RW, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<T, RW, W>), dim3(blocks), dim3(1024), lds, 0, d_offs, iters, span_elems, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 16.0 * 4 * iters;  // one workgroup per CU
    const double lanes = (64 / RW) * W;
    printf("%-4s %2d rows x %2d lanes: %8.3f ms  %6.2f clk per instruction and CU  %6.2f lane-atomics/clk/CU (2.4 GHz)\n", name,
           64 / RW, W, ms, ms * 1e-3 * 2.4e9 / instr_per_cu, instr_per_cu * lanes / (ms * 1e-3 * 2.4e9));
    hipFree(sink);
}

int main() {
    std::vector<int> offs(65536);
    uint32_t s = 12345;
    const int span = 150 * 1024 / 8;  // elements: 150 KB of doubles; the 32-bit types use half the bytes
    for (auto &o : offs) {
        s = s * 1664525u + 1013904223u;
        o = (int)((s >> 8) % (uint32_t)(span - 256)) & ~1;
    }
    int *d_offs;
    hipMalloc(&d_offs, offs.size() * 4);
    hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
    run<double, 64, 64>("f64", d_offs, span);
    run<double, 32, 32>("f64", d_offs, span);
    run<double, 32, 24>("f64", d_offs, span);
    run<double, 16, 16>("f64", d_offs, span);
    run<double, 8, 8>("f64", d_offs, span);
    run<unsigned long long, 32, 32>("u64", d_offs, span);
    run<unsigned long long, 8, 8>("u64", d_offs, span);
    run<unsigned int, 32, 32>("u32", d_offs, span);
    run<float, 32, 32>("f32", d_offs, span);
    hipFree(d_offs);
    return 0;
}
