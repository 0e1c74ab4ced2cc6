// Development probe: can the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) and the fp32 vector pipe
// (v_pk_fma_f32) of a SIMD work at the same time?  Workgroups of 512 threads (2 waves per SIMD), 2 per CU:
// mode 0 = every wave MFMA, mode 1 = every wave packed FMA, mode 2 = odd waves MFMA / even waves FMA
// (same per-wave loop lengths as modes 0/1).  Prints time and TFLOP/s per mode.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float run_mfma(int iters, float a, float b) {
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    return s;
}

__device__ __forceinline__ float run_fma(int iters, float a, float b) {
    f32x2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x2){(float)i, 1.0f};
    f32x2 av = {a, a + 1.0f}, bv = {b, b - 1.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(av, bv, acc[i]);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    return s;
}

// per wave and iteration: MFMA 16 x 4096 flop = 65536; FMA 64 pk_fma x 64 lanes x 4 flop = 16384
__global__ __launch_bounds__(512, 2) void k_probe(float *out, int mode, int it_mfma, int it_fma, float a, float b) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1));
    float s;
    if (do_mfma)
        s = run_mfma(it_mfma, a + threadIdx.x, b);
    else
        s = run_fma(it_fma, a + threadIdx.x, b);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, (size_t)512 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 512, it_mfma = 20000, it_fma = 80000;  // equal flop per wave
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            k_probe<<<grid, 512>>>(out, mode, it_mfma, it_fma, 1.0f, 2.0f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double waves = (double)grid * 8;
            const double flop = waves * (double)it_mfma * 65536.0;  // every wave does the same flop count
            if (rep)
                printf("mode %d (%s): %.3f ms  %.1f TFLOP/s\n", mode,
                       mode == 0 ? "all MFMA" : mode == 1 ? "all packed FMA" : "half MFMA / half packed FMA", ms,
                       flop / ms / 1e9);
        }
    }
    return 0;
}
