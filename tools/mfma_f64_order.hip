// Development probe (not part of the product): is v_mfma_f64_16x16x4_f64 bit-for-bit a k-ordered
// chain of double fma per output element?  Computes C = A * B (16 x K by K x 16, K = 96) with
// the matrix instruction, feeding k in ascending groups of four, and with a sequential
// fma(a, b, acc) loop per element; prints the number of elements whose bits differ, also for the
// alternatives (pairwise within a group, unfused multiply-add).  Build: make -C tools mfma_f64_order
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int K = 96;

__global__ void k_probe(const double *A /*[16][K]*/, const double *B /*[K][16]*/, double *C_mfma, double *C_chain,
                        double *C_pair) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[j * K + k0 + g], B[(k0 + g) * 16 + j], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C_mfma[(4 * r + g) * 16 + j] = acc[r];  // register r of lane group g = row 4 r + g
    // references: every lane computes 4 elements (i = 4 g + r, j)
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        double c = 0.0, p = 0.0;
        for (int k = 0; k < K; ++k) c = fma(A[i * K + k], B[k * 16 + j], c);
        for (int k0 = 0; k0 < K; k0 += 4) {  // (k0 + k1) + (k2 + k3) products fused pairwise, then added
            const double t0 = fma(A[i * K + k0 + 1], B[(k0 + 1) * 16 + j], A[i * K + k0] * B[k0 * 16 + j]);
            const double t1 = fma(A[i * K + k0 + 3], B[(k0 + 3) * 16 + j], A[i * K + k0 + 2] * B[(k0 + 2) * 16 + j]);
            p += t0 + t1;
        }
        C_chain[i * 16 + j] = c;
        C_pair[i * 16 + j] = p;
    }
}

int main() {
    std::vector<double> A(16 * K), B(K * 16), Cm(256), Cc(256), Cp(256);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        return ((double)(x >> 11) / 9007199254740992.0 - 0.5) * 2.0;
    };
    int total_diff_chain = 0, total_diff_pair = 0;
    double *dA, *dB, *dCm, *dCc, *dCp;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8);
    hipMalloc(&dCm, 256 * 8); hipMalloc(&dCc, 256 * 8); hipMalloc(&dCp, 256 * 8);
    for (int trial = 0; trial < 64; ++trial) {
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
        k_probe<<<1, 64>>>(dA, dB, dCm, dCc, dCp);
        hipMemcpy(Cm.data(), dCm, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(Cc.data(), dCc, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(Cp.data(), dCp, 256 * 8, hipMemcpyDeviceToHost);
        for (int e = 0; e < 256; ++e) {
            total_diff_chain += memcmp(&Cm[e], &Cc[e], 8) != 0;
            total_diff_pair += memcmp(&Cm[e], &Cp[e], 8) != 0;
        }
    }
    printf("v_mfma_f64_16x16x4_f64 vs sequential fma chain: %d of %d elements differ\n", total_diff_chain, 64 * 256);
    printf("v_mfma_f64_16x16x4_f64 vs pairwise-in-group sum:  %d of %d elements differ\n", total_diff_pair, 64 * 256);
    return total_diff_chain != 0;
}
