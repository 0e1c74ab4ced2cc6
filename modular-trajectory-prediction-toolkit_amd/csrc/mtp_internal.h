// Internal declarations shared by the HIP translation units of libmtp_accel.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "mtp_accel.h"

void mtp_set_error(const char *fmt, ...);

#define MTP_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            mtp_set_error(__VA_ARGS__); \
            return MTP_E_ARG;           \
        }                               \
    } while (0)

#define MTP_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            mtp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MTP_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

#define MTP_LAUNCH_CHECK() MTP_HIP(hipGetLastError())

constexpr int MTP_MAX_DEVICES = 64;  // per-device host state (function attributes, symbol addresses) is kept for this many

// Compute units of the current device (cached per device).
inline int mtp_device_cu_count(int *cus) {
    static std::atomic<int> cached[MTP_MAX_DEVICES];
    int dev = 0;
    MTP_HIP(hipGetDevice(&dev));
    const bool track = dev >= 0 && dev < MTP_MAX_DEVICES;
    int c = track ? cached[dev].load(std::memory_order_acquire) : 0;
    if (c == 0) {
        MTP_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
        if (c < 1) c = 1;
        if (track) cached[dev].store(c, std::memory_order_release);
    }
    *cus = c;
    return MTP_OK;
}

// hipFuncSetAttribute acts on the CURRENT device's function object: done once per device (not once per
// process), safe from several host threads.  `func` in parentheses when it holds template commas.
#define MTP_SET_MAX_LDS_ONCE(func, bytes)                                                                       \
    do {                                                                                                        \
        static std::atomic<unsigned char> done_[MTP_MAX_DEVICES];                                               \
        int dev_ = 0;                                                                                           \
        MTP_HIP(hipGetDevice(&dev_));                                                                           \
        const bool track_ = dev_ >= 0 && dev_ < MTP_MAX_DEVICES;                                                \
        if (!track_ || !done_[dev_].load(std::memory_order_acquire)) {                                          \
            MTP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(func),                                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));             \
            if (track_) done_[dev_].store(1, std::memory_order_release);                                        \
        }                                                                                                       \
    } while (0)

// ------------------------------------------------------------------------------------------
// Staged centroid image (what the MFMA E-step streams through LDS).
//
// Centroids are grouped in tiles of 32 rows.  One tile image = 32*(R+1) 16-byte slots + 8 slots
// of ||c||^2, R = d/4.  Row i's slot p = 2u + h holds the four floats C[i][8u + h + 2e], e=0..3:
// the A operands of lane (i, h) for MFMA k-steps 4u..4u+3 (v_mfma_f32_32x32x2_f32 takes
// k = 2s + h from lane half h at step s, so the dot product runs in natural k order).
// Rows are R+1 slots apart (one pad slot): the 16-lane groups of a ds_read_b128 then hit
// (i*(R+1) + p) mod 16 = (c*i + p) mod 16 with c odd for R = 16, 24, 32 -> 16 distinct 16-byte
// bank slots, and every read address is one per-lane base + a compile-time offset.
// ------------------------------------------------------------------------------------------
constexpr int MTP_TILE_PAD = 4;  // the image holds a multiple of 4 tiles (any CTW in {1, 2, 4} divides it)

__host__ __device__ inline bool mtp_mfma_dim(int d) { return d == 64 || d == 96 || d == 128; }
__host__ __device__ inline int mtp_tile_floats(int d) { return 32 * (d + 4) + 32; }
__host__ __device__ inline int mtp_n_tiles(int k) { return ((k + 32 * MTP_TILE_PAD - 1) / (32 * MTP_TILE_PAD)) * MTP_TILE_PAD; }

// The sklearn Lloyd "red" buffer layout: sums[k*d], counts[k], n_changed, spare
__host__ __device__ inline int64_t mtp_red_len(int k, int d) { return (int64_t)k * d + k + 2; }

// Restart batching (blockIdx.y / .z = restart b): element strides from one restart's buffer to the
// next; all zero for a single run.  skip_mode 0: a kernel runs only while the restart's state is
// RUNNING; 1: it runs unless the state is STRICT (the final E-step of sklearn _kmeans.py:736-748).
struct MtpBatchStride {
    size_t staged;  // floats
    size_t labels;  // int32s
    size_t red;     // doubles
    size_t flags;   // int32s
    size_t c0, c1;  // floats: the two centre ping-pong buffers
    size_t kvec;    // floats: c2 / shift arrays
    int skip_mode;
    int batch;
    // Restart-grouped centroid images (k_assign_mfma_grouped): group_r > 0 -> restart b's centre j is row
    // (b % group_r) * kp + j of the image of group b / group_r; the images are image_floats apart and `staged`
    // points at the first.  group_r == 0: one image per restart, `staged` strided by .staged.
    int group_r;
    int kp;
    size_t image_floats;
    int b_first;  // restart index of blockIdx 0 for the image lookup (a single restart of a batch handled alone)
};
static inline MtpBatchStride mtp_no_batch() { return MtpBatchStride{0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0}; }

// where restart b's centre j goes: (image base, row within the image)
__device__ __forceinline__ float *mtp_image_of(float *staged, const MtpBatchStride &bs, size_t b, int j, int *row) {
    if (bs.group_r > 0) {
        b += (size_t)bs.b_first;
        *row = (int)(b % (size_t)bs.group_r) * bs.kp + j;
        return staged + (b / (size_t)bs.group_r) * bs.image_floats;
    }
    *row = j;
    return staged + b * bs.staged;
}

__device__ __forceinline__ bool mtp_skip(const int32_t *state, int skip_mode) {
    if (state == nullptr) return false;
    const int st = *state;
    return skip_mode == 0 ? st != MTP_STATE_RUNNING : st == MTP_STATE_STRICT;
}

static inline hipStream_t mtp_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- launchers implemented across the .hip files (all enqueue only) ----
int mtp_launch_stage_centers(const float *C, int k, int d, float *staged, float *c2, const int32_t *state,
                             hipStream_t s, MtpBatchStride bs = mtp_no_batch());
int mtp_launch_clear_images(float *images, int n_images, int tiles, int d, size_t image_floats, hipStream_t s);
int mtp_pick_restart_group(int64_t n, int k, int batch, int cus);
int mtp_launch_assign(const float *X, int64_t n, int d, const float *C, const float *staged, const float *c2, int k,
                      int32_t *labels, double *n_changed, const int32_t *state, hipStream_t s,
                      MtpBatchStride bs = mtp_no_batch());
// sums_mode: 0 = double sums (schedule-independent, the default) or MTP_LLOYD_SUMS_SKLEARN_1T = float32 row-order
// chains, what sklearn forms with one thread (the bit-exact comparison mode)
int mtp_launch_accumulate(const float *X, int64_t n, int d, const int32_t *labels, int k, double *sums,
                          double *counts, const int32_t *state, hipStream_t s, MtpBatchStride bs = mtp_no_batch(),
                          int sums_mode = 0);
// flag_parity = 0: C_old / C_new as given.  1 (batched runs): C_old is buffer 0 of the restart, C_new
// buffer 1, swapped when the restart's n_iter is odd -- restarts drift apart when one of them pauses.
int mtp_launch_mstep(double *red, const float *C_old, float *C_new, int k, int d, double tol, int max_iter,
                     int relocation_done, float *staged, float *c2, float *shift, int32_t *flags, double *red_backup,
                     hipStream_t s, MtpBatchStride bs = mtp_no_batch(), int flag_parity = 0, int sums_mode = 0);
int mtp_launch_inertia(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                       float *per_point, hipStream_t s);
int mtp_launch_inertia_batched(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                               float *per_point, int batch, size_t c_stride, size_t lab_stride, hipStream_t s);

// sklearn's one-thread float32 inertia (per-point float distances added in row order); per_point_scratch: batch * n floats
int mtp_launch_inertia_chain_f32(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                                 float *per_point_scratch, int batch, size_t c_stride, size_t lab_stride, hipStream_t s);
int mtp_launch_apply_relocation(double *red, int k, int d, int m, const float *far_val, const int32_t *far_lab,
                                const float *far_rows, int sums_mode, hipStream_t s);

// ||row||^2 in NumPy einsum("ij,ij->i") float order (see oracle.c orc_row_sqnorm_f32): four lane
// accumulators, product rounded then added, blocks of 16 consumed vector 3,2,1,0, tail vectors in
// order, (l0 + l1) + (l2 + l3).  Compiled with -ffp-contract=off so a*a + l stays two roundings.
__device__ inline float mtp_sqnorm_einsum_order(const float *a, int d) {
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    int e = 0;
    for (; e + 16 <= d; e += 16) {
        for (int v = 3; v >= 0; --v) {
            const float *p = a + e + 4 * v;
            l0 = l0 + p[0] * p[0];
            l1 = l1 + p[1] * p[1];
            l2 = l2 + p[2] * p[2];
            l3 = l3 + p[3] * p[3];
        }
    }
    for (; e < d; e += 4) {
        l0 = l0 + a[e] * a[e];
        if (e + 1 < d) l1 = l1 + a[e + 1] * a[e + 1];
        if (e + 2 < d) l2 = l2 + a[e + 2] * a[e + 2];
        if (e + 3 < d) l3 = l3 + a[e + 3] * a[e + 3];
    }
    return (l0 + l1) + (l2 + l3);
}

// Write one centroid row (values in `row`, LDS or global) into the staged image.  Called by a
// whole block: thread t < R writes slot t.  j = global centroid index, valid = j < k.
__device__ inline void mtp_write_staged_row(float *staged, int j, int d, const float *row, bool valid, float c2v,
                                            int tid, int nthreads) {
    const int R = d / 4;
    const int tile = j >> 5, i = j & 31;
    float *tb = staged + (size_t)tile * mtp_tile_floats(d);
    for (int p = tid; p < R; p += nthreads) {
        const int u = p >> 1, h = p & 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = make_float4(row[8 * u + h], row[8 * u + h + 2], row[8 * u + h + 4], row[8 * u + h + 6]);
        reinterpret_cast<float4 *>(tb)[i * (R + 1) + p] = v;
    }
    if (tid == 0) {
        reinterpret_cast<float4 *>(tb)[i * (R + 1) + R] = make_float4(0.f, 0.f, 0.f, 0.f);  // pad slot
        tb[32 * (d + 4) + i] = valid ? c2v : __builtin_inff();
    }
}
