// Development probe: sustained rate of v_mfma_f64_16x16x4_f64 (and of v_fma_f64 for comparison) with every
// SIMD busy: WAVES waves per SIMD, 8 independent accumulators per wave, no memory traffic in the loop.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, b, acc[i]);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the multiplier in a SCALAR register (uniform kernel argument): v_fmac_f64 v, s, v
template <int NACC>
__global__ __launch_bounds__(256) void k_fma_sgpr(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "s"(a0), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double *out;
    hipMalloc(&out, (size_t)256 * 16 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            k_mfma<8><<<grid, 256>>>(out, iters, 1.0, 2.0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 8 * 2048.0;
            if (rep) printf("mfma_f64_16x16x4: %d waves/SIMD  %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            k_fma<16><<<grid, 256>>>(out, iters, 1.0, 2.0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 256 * iters * 16 * 2.0;
            if (rep) printf("v_fma_f64:        %d waves/SIMD  %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            k_fma_sgpr<16><<<grid, 256>>>(out, iters, 1.0, 2.0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 256 * iters * 16 * 2.0;
            if (rep) printf("v_fmac_f64 sgpr:  %d waves/SIMD  %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
