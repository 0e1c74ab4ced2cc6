// C-ABI entry points of libmtp_accel.so (see include/mtp_accel.h) and the single-GPU Lloyd driver.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "mtp_internal.h"

static thread_local char g_err[512] = "";

void mtp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mtp_last_error(void) { return g_err; }
extern "C" int mtp_version(void) { return 100; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int64_t mtp_staged_centers_floats(int k, int d) {
    if (k <= 0 || d <= 0 || !mtp_mfma_dim(d)) return 0;
    return (int64_t)mtp_n_tiles(k) * mtp_tile_floats(d);
}

extern "C" int64_t mtp_assign_workspace_bytes(int k, int d) {
    if (k <= 0 || d <= 0) return 0;
    return (int64_t)(align_up((size_t)mtp_staged_centers_floats(k, d) * 4, 256) + align_up((size_t)k * 4, 256));
}

// Lloyd workspace carve (all offsets multiples of 256 bytes)
struct LloydWs {
    double *red;
    float *c_alt, *staged, *c2, *shift;
    int32_t *flags;
    float *far_val;
    int64_t *far_idx;
    int32_t *far_lab;
    float *far_rows;
    size_t bytes;
};
static LloydWs carve_lloyd(void *ws, int k, int d) {
    LloydWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = ws ? static_cast<char *>(ws) + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    w.red = static_cast<double *>(take((size_t)mtp_red_len(k, d) * 8));
    w.c_alt = static_cast<float *>(take((size_t)k * d * 4));
    w.staged = static_cast<float *>(take((size_t)mtp_staged_centers_floats(k, d) * 4 + 16));
    w.c2 = static_cast<float *>(take((size_t)k * 4));
    w.shift = static_cast<float *>(take((size_t)k * 4));
    w.flags = static_cast<int32_t *>(take(MTP_FLAG_WORDS * 4));
    // one far point per empty cluster, and up to k - 1 clusters can be empty at once
    // (sklearn _k_means_common.pyx:167-211 relocates every one of them)
    w.far_val = static_cast<float *>(take((size_t)k * 4));
    w.far_idx = static_cast<int64_t *>(take((size_t)k * 8));
    w.far_lab = static_cast<int32_t *>(take((size_t)k * 4));
    w.far_rows = static_cast<float *>(take((size_t)k * d * 4));
    w.bytes = off;
    return w;
}

extern "C" int64_t mtp_lloyd_workspace_bytes(int k, int d) {
    if (k <= 0 || d <= 0) return 0;
    return (int64_t)carve_lloyd(nullptr, k, d).bytes;
}

// dist_scratch of mtp_far_points_f32: n per-point distances + the <= 1024 slice winners of each argmax pass
extern "C" int64_t mtp_far_points_scratch_bytes(int64_t n) {
    if (n < 0) return 0;
    return (int64_t)(align_up((size_t)n * 4, 256) + 1024 * 4 + 1024 * 8);
}

extern "C" int mtp_assign_f32(const float *X, int64_t n, int d, const float *C, int k, const float *c2, int32_t *labels,
                              double *n_changed, void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(X && C && labels, "mtp_assign: null pointer");
    MTP_REQUIRE(n >= 0 && d >= 1 && k >= 1, "mtp_assign: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    if ((size_t)mtp_assign_workspace_bytes(k, d) > ws_bytes || ws == nullptr) {
        mtp_set_error("mtp_assign: workspace of %zu bytes, need %lld", ws_bytes, (long long)mtp_assign_workspace_bytes(k, d));
        return MTP_E_WORKSPACE;
    }
    hipStream_t s = mtp_stream(stream);
    float *staged = static_cast<float *>(ws);
    float *c2w = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)mtp_staged_centers_floats(k, d) * 4, 256));
    if (c2 != nullptr && mtp_mfma_dim(d)) {
        mtp_set_error("mtp_assign: caller-provided c2 is only honoured by the generic (non-MFMA) kernel; pass NULL");
        return MTP_E_ARG;
    }
    // stage C (and ||c||^2 unless the caller brought its own)
    int rc = mtp_launch_stage_centers(C, k, d, staged, c2w, nullptr, s);
    if (rc != MTP_OK) return rc;
    return mtp_launch_assign(X, n, d, C, staged, c2 ? c2 : c2w, k, labels, n_changed, nullptr, s);
}

// ---- the one entry point that allocates nothing and takes no workspace (include/mtp_accel.h) ----
// mtp_assign_f32's scratch -- the staged centroid image and ||c||^2 -- lives in a file-scope device
// array instead of a caller workspace.  In host code `g_noalloc_scratch` is only a stand-in for the
// symbol, NOT a device address: the address is looked up once per device with hipGetSymbolAddress
// (each GPU holds its own copy of the array) and only that pointer is ever handed to a kernel.
static __device__ __align__(256) unsigned char g_noalloc_scratch[MTP_NOALLOC_SCRATCH_BYTES];

static std::mutex g_noalloc_mutex;
static void *g_noalloc_ptr[MTP_MAX_DEVICES] = {};  // per device: where that GPU's copy of the array lives

static int noalloc_scratch_ptr(void **out) {
    int dev = -1;
    MTP_HIP(hipGetDevice(&dev));
    MTP_REQUIRE(dev >= 0 && dev < MTP_MAX_DEVICES, "mtp_assign_noalloc: device ordinal %d outside [0, %d)", dev,
                MTP_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_ptr[dev] == nullptr) {
        void *p = nullptr;
        MTP_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch)));  // the CURRENT device's copy
        MTP_REQUIRE(p != nullptr && (reinterpret_cast<uintptr_t>(p) & 255) == 0,
                    "mtp_assign_noalloc: scratch symbol resolved to %p", p);
        g_noalloc_ptr[dev] = p;
    }
    *out = g_noalloc_ptr[dev];
    return MTP_OK;
}

extern "C" int mtp_assign_noalloc_f32(const float *X, int64_t n, int d, const float *C, int k, int32_t *labels,
                                      double *n_changed, void *stream) {
    MTP_REQUIRE(X && C && labels, "mtp_assign_noalloc: null pointer");
    MTP_REQUIRE(n >= 0 && d >= 1 && k >= 1, "mtp_assign_noalloc: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    const int64_t need = mtp_assign_workspace_bytes(k, d);
    if (need > (int64_t)MTP_NOALLOC_SCRATCH_BYTES) {
        mtp_set_error("mtp_assign_noalloc: k=%d, d=%d needs %lld bytes of scratch, the built-in buffer holds %d "
                      "(use mtp_assign_f32 with a caller workspace)", k, d, (long long)need, MTP_NOALLOC_SCRATCH_BYTES);
        return MTP_E_WORKSPACE;
    }
    void *scratch = nullptr;
    int rc = noalloc_scratch_ptr(&scratch);
    if (rc != MTP_OK) return rc;
    // Every byte the E-step reads of the scratch is rewritten by the staging kernel of this call (all
    // rows of every padded tile and all k norms), so nothing left by an earlier, larger call is seen.
    return mtp_assign_f32(X, n, d, C, k, nullptr, labels, n_changed, scratch, MTP_NOALLOC_SCRATCH_BYTES, stream);
}

extern "C" int mtp_centroid_sums_f32(const float *X, int64_t n, int d, const int32_t *labels, int k, double *sums,
                                     double *counts, void *stream) {
    MTP_REQUIRE(X && labels && sums && counts, "mtp_centroid_sums: null pointer");
    MTP_REQUIRE(n >= 0 && d >= 1 && k >= 1, "mtp_centroid_sums: bad shape");
    return mtp_launch_accumulate(X, n, d, labels, k, sums, counts, nullptr, mtp_stream(stream));
}

extern "C" int mtp_inertia_f32(const float *X, int64_t n, int d, const float *C, const int32_t *labels, double *inertia,
                               float *per_point, void *stream) {
    MTP_REQUIRE(X && C && labels && (inertia || per_point), "mtp_inertia: null pointer");
    return mtp_launch_inertia(X, n, d, C, labels, inertia, per_point, mtp_stream(stream));
}

// ---- step-level Lloyd API ----
extern "C" int mtp_lloyd_begin_f32(const float *C, int k, int d, float *staged, float *c2, double *red, int32_t *flags,
                                   void *stream) {
    MTP_REQUIRE(C && c2 && red && flags && k >= 1 && d >= 1, "mtp_lloyd_begin: bad arguments");
    MTP_REQUIRE(!mtp_mfma_dim(d) || staged, "mtp_lloyd_begin: staged image buffer missing");
    hipStream_t s = mtp_stream(stream);
    MTP_HIP(hipMemsetAsync(red, 0, (size_t)mtp_red_len(k, d) * 8, s));
    MTP_HIP(hipMemsetAsync(flags, 0, MTP_FLAG_WORDS * 4, s));
    return mtp_launch_stage_centers(C, k, d, staged, c2, nullptr, s);
}

extern "C" int mtp_lloyd_assign_f32(const float *X, int64_t n, int d, const float *C, const float *staged,
                                    const float *c2, int k, int32_t *labels, double *red, const int32_t *flags,
                                    void *stream) {
    MTP_REQUIRE(X && C && c2 && labels && red && flags, "mtp_lloyd_assign: null pointer");
    return mtp_launch_assign(X, n, d, C, staged, c2, k, labels, red + (size_t)k * d + k, flags + MTP_FLAG_STATE,
                             mtp_stream(stream));
}

extern "C" int mtp_lloyd_accumulate_f32(const float *X, int64_t n, int d, const int32_t *labels, int k, double *red,
                                        const int32_t *flags, void *stream) {
    MTP_REQUIRE(X && labels && red && flags, "mtp_lloyd_accumulate: null pointer");
    return mtp_launch_accumulate(X, n, d, labels, k, red, red + (size_t)k * d, flags + MTP_FLAG_STATE,
                                 mtp_stream(stream));
}

extern "C" int mtp_lloyd_estep_f32(const float *X, int64_t n, int d, const float *C, const float *staged, const float *c2,
                                   int k, int32_t *labels, double *red, const int32_t *flags, void *stream) {
    int rc = mtp_lloyd_assign_f32(X, n, d, C, staged, c2, k, labels, red, flags, stream);
    if (rc != MTP_OK) return rc;
    return mtp_lloyd_accumulate_f32(X, n, d, labels, k, red, flags, stream);
}

extern "C" int mtp_lloyd_mstep_f32(double *red, const float *C_old, float *C_new, int k, int d, double tol, int max_iter,
                                   int relocation_done, float *staged, float *c2, float *shift, int32_t *flags,
                                   double *red_backup, void *stream) {
    MTP_REQUIRE(red && C_old && C_new && c2 && shift && flags && C_old != C_new, "mtp_lloyd_mstep: bad arguments");
    return mtp_launch_mstep(red, C_old, C_new, k, d, tol, max_iter, relocation_done, staged, c2, shift, flags,
                            red_backup, mtp_stream(stream));
}

// ---- RCCL, bound at run time ----
// The library does not link librccl: it looks the five entry points up in the librccl.so the process already
// holds (PyTorch-ROCm ships one and loads it with torch; a second copy in the process would mean a second set of
// device state), else in the system's.  A box without RCCL can still use everything that is not sharded.
typedef struct ncclComm *mtp_nccl_comm_t;
struct MtpNcclId {
    char internal[128];  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
};
namespace {
struct Rccl {
    int (*GetUniqueId)(MtpNcclId *) = nullptr;
    int (*CommInitRank)(mtp_nccl_comm_t *, int, MtpNcclId, int) = nullptr;
    int (*CommDestroy)(mtp_nccl_comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, mtp_nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, mtp_nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclSum = 0, kNcclChar = 0, kNcclDouble = 8;  // rccl.h: ncclSum, ncclChar, ncclFloat64

const Rccl &rccl() {
    static Rccl r = [] {
        Rccl q;
        void *h = nullptr;
        // MTP_RCCL_LIB=<path>: bind the six entry points from that library instead (read once, at the first
        // use).  For collective libraries that are not called librccl.so, and for tests/stub_nccl -- a
        // shared-memory stand-in that lets several processes on ONE GPU run the sharded loop, which RCCL
        // refuses.  No fallback when the path does not load: the caller asked for that library.
        const char *override_path = getenv("MTP_RCCL_LIB");
        if (override_path != nullptr && override_path[0] != '\0') {
            h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
            if (h == nullptr) fprintf(stderr, "libmtp_accel: MTP_RCCL_LIB=%s: %s\n", override_path, dlerror());
        } else {
            h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
            if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
            if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        }
        if (h == nullptr) return q;
        q.GetUniqueId = reinterpret_cast<decltype(q.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        q.CommInitRank = reinterpret_cast<decltype(q.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        q.AllReduce = reinterpret_cast<decltype(q.AllReduce)>(dlsym(h, "ncclAllReduce"));
        q.AllGather = reinterpret_cast<decltype(q.AllGather)>(dlsym(h, "ncclAllGather"));
        q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.AllGather && q.GetErrorString;
        return q;
    }();
    return r;
}
}  // namespace

#define MTP_NCCL(expr)                                                                                        \
    do {                                                                                                      \
        const int e_ = (expr);                                                                                \
        if (e_ != 0) {                                                                                        \
            mtp_set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(e_), __FILE__, __LINE__);     \
            return MTP_E_COMM;                                                                                \
        }                                                                                                     \
    } while (0)

struct mtp_comm {
    mtp_nccl_comm_t nccl;
    int rank, world, device;
};

extern "C" int mtp_comm_unique_id(void *id_out) {
    MTP_REQUIRE(id_out != nullptr, "mtp_comm_unique_id: null pointer");
    if (!rccl().ok) {
        mtp_set_error("mtp_comm_unique_id: librccl.so not found in this process or on the library path");
        return MTP_E_COMM;
    }
    MTP_NCCL(rccl().GetUniqueId(static_cast<MtpNcclId *>(id_out)));
    return MTP_OK;
}

extern "C" int mtp_comm_create(int rank, int world, const void *id, mtp_comm **out) {
    MTP_REQUIRE(id != nullptr && out != nullptr && world >= 1 && rank >= 0 && rank < world, "mtp_comm_create: bad arguments");
    if (!rccl().ok) {
        mtp_set_error("mtp_comm_create: librccl.so not found in this process or on the library path");
        return MTP_E_COMM;
    }
    MtpNcclId uid;
    memcpy(&uid, id, sizeof(uid));
    mtp_comm *c = new mtp_comm{nullptr, rank, world, 0};
    hipError_t he = hipGetDevice(&c->device);
    if (he != hipSuccess) {
        delete c;
        mtp_set_error("mtp_comm_create: hipGetDevice failed: %s", hipGetErrorString(he));
        return MTP_E_HIP;
    }
    const int e = rccl().CommInitRank(&c->nccl, world, uid, rank);  // collective: every rank calls it, on its own GPU
    if (e != 0) {
        mtp_set_error("mtp_comm_create: ncclCommInitRank failed: %s", rccl().GetErrorString(e));
        delete c;
        return MTP_E_COMM;
    }
    *out = c;
    return MTP_OK;
}

extern "C" int mtp_comm_destroy(mtp_comm *c) {
    if (c == nullptr) return MTP_OK;
    const int e = rccl().CommDestroy(c->nccl);
    delete c;
    if (e != 0) {
        mtp_set_error("mtp_comm_destroy: ncclCommDestroy failed: %s", rccl().GetErrorString(e));
        return MTP_E_COMM;
    }
    return MTP_OK;
}

// Global top-m of the ranks' local far points (all-gathered, W * m entries): descending distance, ties to the
// lowest global row index, entries without a point (index < 0) last -- what one GPU's search over all points
// would return.  One block; rank by counting (the lists are short, and empty clusters are rare).
__global__ __launch_bounds__(1024) void k_far_merge(const float *gval, const int64_t *gidx, const int32_t *glab,
                                                    const float *grows, int total, int m, int d, float *val,
                                                    int64_t *idx, int32_t *lab, float *rows, int32_t *peer_failed) {
    // a rank whose far-point search failed still joins the all-gathers, with the value kFarFailed in its list:
    // every rank sees it in the same gathered data and leaves the loop with an error instead of hanging
    __shared__ int s_failed;
    if (threadIdx.x == 0) s_failed = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x)
        if (gval[e] == -2.f) s_failed = 1;
    __syncthreads();
    if (threadIdx.x == 0) *peer_failed = s_failed;
    for (int q = threadIdx.x; q < m; q += blockDim.x) {  // "no candidate" unless a ranked entry lands here
        val[q] = -1.f;
        idx[q] = -1;
        lab[q] = -1;
    }
    for (int e = threadIdx.x; e < m * d; e += blockDim.x) rows[e] = 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int64_t ie = gidx[e];
        if (ie < 0) continue;
        const float ve = gval[e];
        int rank = 0;
        for (int j = 0; j < total; ++j) {
            const int64_t ij = gidx[j];
            if (ij >= 0 && (gval[j] > ve || (gval[j] == ve && ij < ie))) ++rank;
        }
        if (rank < m) {
            val[rank] = ve;
            idx[rank] = ie;
            lab[rank] = glab[e];
            for (int c = 0; c < d; ++c) rows[(size_t)rank * d + c] = grows[(size_t)e * d + c];
        }
    }
}

__global__ void k_fill_no_far_points(float *val, int64_t *idx, int32_t *lab, float *rows, int m, int d, float mark) {
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < m * d; q += gridDim.x * blockDim.x) {
        rows[q] = 0.f;
        if (q < m) {
            val[q] = mark;  // -1: this rank holds no points; -2: its search failed (k_far_merge)
            idx[q] = -1;
            lab[q] = -1;
        }
    }
}

extern "C" int mtp_far_merge_f32(const float *gval, const int64_t *gidx, const int32_t *glab, const float *grows, int total,
                                 int m, int d, float *val, int64_t *idx, int32_t *lab, float *rows, int32_t *peer_failed,
                                 void *stream) {
    MTP_REQUIRE(gval && gidx && glab && grows && val && idx && lab && rows && peer_failed, "mtp_far_merge: null pointer");
    MTP_REQUIRE(total >= 0 && m >= 1 && d >= 1, "mtp_far_merge: bad shape total=%d m=%d d=%d", total, m, d);
    hipLaunchKernelGGL(k_far_merge, dim3(1), dim3(1024), 0, mtp_stream(stream), gval, gidx, glab, grows, total, m, d, val, idx,
                       lab, rows, peer_failed);
    MTP_LAUNCH_CHECK();
    return MTP_OK;
}

// ---- whole single-GPU run: sklearn _kmeans_single_lloyd (cluster/_kmeans.py:624-752) ----
// Points sharded over the ranks of `cm` (NULL: one GPU): every rank runs the same loop on its rows; per
// iteration ONE all-reduce (sum) of the packed double buffer red = [k*d sums | k counts | n_changed | spare],
// after which every rank finishes the M-step identically -- centres, norms and loop state stay replicated bit
// for bit, no broadcast.  Empty clusters: the ranks' local top-m far points are all-gathered and merged
// (k_far_merge).  Iterations enqueued behind a pause are no-ops, but their all-reduces are not: the paused
// M-step parks red in red_backup and zeroes it, so those all-reduces add zeros (mtp_lloyd_mstep_f32).
struct ShardWs {
    double *red_backup;
    float *g_val;
    int64_t *g_idx;
    int32_t *g_lab;
    float *g_rows;
    double *inertia;
    size_t bytes;
};
static ShardWs carve_shard(void *ws, int k, int d, int world) {
    ShardWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = ws ? static_cast<char *>(ws) + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    w.red_backup = static_cast<double *>(take((size_t)mtp_red_len(k, d) * 8));
    w.g_val = static_cast<float *>(take((size_t)world * k * 4));
    w.g_idx = static_cast<int64_t *>(take((size_t)world * k * 8));
    w.g_lab = static_cast<int32_t *>(take((size_t)world * k * 4));
    w.g_rows = static_cast<float *>(take((size_t)world * k * d * 4));
    w.inertia = static_cast<double *>(take(8));
    w.bytes = off;
    return w;
}

extern "C" int64_t mtp_lloyd_sharded_workspace_bytes(int k, int d, int64_t n_local, int world) {
    if (k <= 0 || d <= 0 || n_local < 0 || world <= 0) return 0;
    return (int64_t)(carve_lloyd(nullptr, k, d).bytes + (size_t)mtp_far_points_scratch_bytes(n_local) +
                     carve_shard(nullptr, k, d, world).bytes);
}

static int lloyd_run(const mtp_comm *cm, const float *X, int64_t n, int64_t row_offset, int d, float *C_inout, int k,
                     int max_iter, double tol_abs, int check_every, int32_t *labels_out, double *inertia_out,
                     int *n_iter_out, float *phase_ms_out, int options, void *ws, size_t ws_bytes, void *stream) {
    LloydWs w = carve_lloyd(ws, k, d);
    const size_t scratch_bytes = (size_t)mtp_far_points_scratch_bytes(n);
    float *dist_scratch = reinterpret_cast<float *>(static_cast<char *>(ws) + w.bytes);
    ShardWs sw{};
    size_t need = w.bytes + scratch_bytes;
    if (cm != nullptr) {
        sw = carve_shard(static_cast<char *>(ws) + need, k, d, cm->world);
        need += sw.bytes;
    }
    if (ws_bytes < need) {
        mtp_set_error("mtp_lloyd: workspace of %zu bytes, need %zu", ws_bytes, need);
        return MTP_E_WORKSPACE;
    }
    hipStream_t s = mtp_stream(stream);
    if (check_every < 1) check_every = 1;
    const int64_t red_len = mtp_red_len(k, d);
    // MTP_LLOYD_SUMS_SKLEARN_1T: centroid sums, the total shift and the inertia in the float32 orders of a one-thread
    // sklearn run (row-order chains): one rank only -- a chain does not split over ranks without taking turns
    const int sums_mode = options & MTP_LLOYD_SUMS_SKLEARN_1T;
    MTP_REQUIRE(sums_mode == 0 || cm == nullptr || cm->world == 1,
                "mtp_lloyd_sharded: MTP_LLOYD_SUMS_SKLEARN_1T needs all points on one rank (world = %d)", cm ? cm->world : 1);
    MTP_REQUIRE(sums_mode == 0 || k <= 8192, "mtp_lloyd: MTP_LLOYD_SUMS_SKLEARN_1T supports k <= 8192 (NumPy's pairwise "
                "order of the total shift is only restated up to its buffer size), got k = %d", k);

    int rc = mtp_lloyd_begin_f32(C_inout, k, d, w.staged, w.c2, w.red, w.flags, stream);
    if (rc != MTP_OK) return rc;
    if (n > 0) MTP_HIP(hipMemsetAsync(labels_out, 0xff, (size_t)n * 4, s));  // labels_old = -1 (_kmeans.py:685)

    // optional per-phase timing (bench.py): events around the E-step / the sums / the all-reduce / the M-step finish.
    // The events are created here, before anything is timed (5 per iteration, for the first kMaxTimedIters iterations).
    constexpr int kMarks = 5, kMaxTimedIters = 1024;
    std::vector<hipEvent_t> ev;
    struct EvGuard {
        std::vector<hipEvent_t> &v;
        ~EvGuard() {
            for (hipEvent_t e : v) (void)hipEventDestroy(e);
        }
    } ev_guard{ev};
    size_t ev_used = 0;
    if (phase_ms_out != nullptr) {
        const int timed = max_iter < kMaxTimedIters ? max_iter : kMaxTimedIters;
        ev.reserve((size_t)timed * kMarks);
        for (int i = 0; i < timed * kMarks; ++i) {
            hipEvent_t e;
            MTP_HIP(hipEventCreate(&e));
            ev.push_back(e);
        }
    }
    auto mark = [&]() -> int {
        if (ev_used >= ev.size()) return MTP_OK;
        MTP_HIP(hipEventRecord(ev[ev_used++], s));
        return MTP_OK;
    };

    // one iteration: E-step, local sums, [all-reduce], M-step finish (pauses on an empty cluster)
    auto iteration = [&](const float *c_old, float *c_new) -> int {
        int r = mark();
        if (r != MTP_OK) return r;
        if (n > 0) {
            r = mtp_lloyd_assign_f32(X, n, d, c_old, w.staged, w.c2, k, labels_out, w.red, w.flags, stream);
            if (r != MTP_OK) return r;
        }
        r = mark();
        if (r != MTP_OK) return r;
        if (n > 0) {
            r = mtp_launch_accumulate(X, n, d, labels_out, k, w.red, w.red + (size_t)k * d, w.flags + MTP_FLAG_STATE, s,
                                      mtp_no_batch(), sums_mode);
            if (r != MTP_OK) return r;
        }
        r = mark();
        if (r != MTP_OK) return r;
        if (cm != nullptr) MTP_NCCL(rccl().AllReduce(w.red, w.red, (size_t)red_len, kNcclDouble, kNcclSum, cm->nccl, s));
        r = mark();
        if (r != MTP_OK) return r;
        r = mtp_launch_mstep(w.red, c_old, c_new, k, d, tol_abs, max_iter, 0, w.staged, w.c2, w.shift, w.flags,
                             cm != nullptr ? sw.red_backup : nullptr, s, mtp_no_batch(), 0, sums_mode);
        if (r != MTP_OK) return r;
        return mark();
    };

    // centres ping-pong between C_inout (cur = 0) and the workspace copy (cur = 1)
    float *cbuf[2] = {C_inout, w.c_alt};
    int cur = 0;           // buffer holding the centres the next E-step reads
    int enq_iters = 0;     // iterations enqueued so far
    int32_t hflags[MTP_FLAG_WORDS];
    for (;;) {
        const int budget = (max_iter - enq_iters) < check_every ? (max_iter - enq_iters) : check_every;
        for (int i = 0; i < budget; ++i) {
            rc = iteration(cbuf[cur], cbuf[cur ^ 1]);
            if (rc != MTP_OK) return rc;
            cur ^= 1;
            ++enq_iters;
        }
        MTP_HIP(hipMemcpyAsync(hflags, w.flags, sizeof(hflags), hipMemcpyDeviceToHost, s));
        MTP_HIP(hipStreamSynchronize(s));
        const int done_iters = hflags[MTP_FLAG_N_ITER];
        if (hflags[MTP_FLAG_STATE] == MTP_STATE_RELOCATE) {
            // iteration `done_iters` (0-based) stopped before its M-step: its E-step ran, labels and
            // red are valid.  The centres it read are in parity (done_iters & 1).
            const int old = done_iters & 1;
            const int m = hflags[MTP_FLAG_N_EMPTY];  // < k: at least one cluster holds points
            int far_rc = MTP_OK;
            if (n > 0) {
                far_rc = mtp_far_points_f32(X, n, d, cbuf[old], labels_out, m, row_offset, w.far_val, w.far_idx, w.far_lab,
                                            w.far_rows, dist_scratch, stream);
                if (far_rc != MTP_OK && cm == nullptr) return far_rc;
            }
            if (n == 0 || far_rc != MTP_OK) {
                // no points here -- or the search failed: the other ranks are about to enter the all-gathers, so
                // this rank joins them with a marked list and everyone returns an error together (k_far_merge)
                hipLaunchKernelGGL(k_fill_no_far_points, dim3(64), dim3(256), 0, s, w.far_val, w.far_idx, w.far_lab, w.far_rows,
                                   m, d, far_rc != MTP_OK ? -2.f : -1.f);
                MTP_LAUNCH_CHECK();
            }
            if (cm != nullptr) {
                MTP_HIP(hipMemcpyAsync(w.red, sw.red_backup, (size_t)red_len * 8, hipMemcpyDeviceToDevice, s));
                MTP_NCCL(rccl().AllGather(w.far_val, sw.g_val, (size_t)m * 4, kNcclChar, cm->nccl, s));
                MTP_NCCL(rccl().AllGather(w.far_idx, sw.g_idx, (size_t)m * 8, kNcclChar, cm->nccl, s));
                MTP_NCCL(rccl().AllGather(w.far_lab, sw.g_lab, (size_t)m * 4, kNcclChar, cm->nccl, s));
                MTP_NCCL(rccl().AllGather(w.far_rows, sw.g_rows, (size_t)m * d * 4, kNcclChar, cm->nccl, s));
                hipLaunchKernelGGL(k_far_merge, dim3(1), dim3(1024), 0, s, sw.g_val, sw.g_idx, sw.g_lab, sw.g_rows,
                                   cm->world * m, m, d, w.far_val, w.far_idx, w.far_lab, w.far_rows,
                                   w.flags + MTP_FLAG_PEER_FAILED);
                MTP_LAUNCH_CHECK();
                int32_t peer_failed = 0;  // relocations are rare and already a host sync point: one more word
                MTP_HIP(hipMemcpyAsync(&peer_failed, w.flags + MTP_FLAG_PEER_FAILED, 4, hipMemcpyDeviceToHost, s));
                MTP_HIP(hipStreamSynchronize(s));
                if (far_rc != MTP_OK) return far_rc;  // this rank's own error (mtp_last_error() still names it)
                if (peer_failed != 0) {
                    mtp_set_error("mtp_lloyd_sharded: the far-point search failed on another rank; all ranks stop");
                    return MTP_E_COMM;
                }
            }
            rc = mtp_launch_apply_relocation(w.red, k, d, m, w.far_val, w.far_lab, w.far_rows, sums_mode, s);
            if (rc != MTP_OK) return rc;
            MTP_HIP(hipMemsetAsync(w.flags + MTP_FLAG_STATE, 0, 4, s));
            rc = mtp_launch_mstep(w.red, cbuf[old], cbuf[old ^ 1], k, d, tol_abs, max_iter, 1, w.staged, w.c2, w.shift,
                                  w.flags, nullptr, s, mtp_no_batch(), 0, sums_mode);
            if (rc != MTP_OK) return rc;
            cur = old ^ 1;
            enq_iters = done_iters + 1;
            if (enq_iters < max_iter) continue;  // keep iterating (state decides below otherwise)
            MTP_HIP(hipMemcpyAsync(hflags, w.flags, sizeof(hflags), hipMemcpyDeviceToHost, s));
            MTP_HIP(hipStreamSynchronize(s));
        }
        if (hflags[MTP_FLAG_STATE] != MTP_STATE_RUNNING && hflags[MTP_FLAG_STATE] != MTP_STATE_RELOCATE) break;
        if (enq_iters >= max_iter) break;  // cannot happen: the last M-step sets MAXITER
    }
    const int n_iter = hflags[MTP_FLAG_N_ITER];
    const int fin = n_iter & 1;  // centres after n_iter M-steps live in parity n_iter & 1
    if (hflags[MTP_FLAG_STATE] != MTP_STATE_STRICT && n > 0 && !(options & MTP_LLOYD_NO_FINAL_ESTEP)) {
        // rerun the E-step so that the labels match the final centres (_kmeans.py:736-748).
        // staged / c2 already describe the final centres (written by the last M-step).
        rc = mtp_launch_assign(X, n, d, cbuf[fin], w.staged, w.c2, k, labels_out, nullptr, nullptr, s);
        if (rc != MTP_OK) return rc;
    }
    if (fin != 0) MTP_HIP(hipMemcpyAsync(C_inout, w.c_alt, (size_t)k * d * 4, hipMemcpyDeviceToDevice, s));
    if (inertia_out != nullptr) {
        MTP_HIP(hipMemsetAsync(inertia_out, 0, 8, s));
        if (sums_mode != 0) {  // sklearn's float32 chain; the far-point scratch (n floats) holds the per-point distances
            rc = mtp_launch_inertia_chain_f32(X, n, d, cbuf[fin], labels_out, inertia_out, dist_scratch, 1, 0, 0, s);
            if (rc != MTP_OK) return rc;
        } else if (n > 0) {
            rc = mtp_launch_inertia(X, n, d, cbuf[fin], labels_out, inertia_out, nullptr, s);
            if (rc != MTP_OK) return rc;
        }
        if (cm != nullptr) MTP_NCCL(rccl().AllReduce(inertia_out, inertia_out, 1, kNcclDouble, kNcclSum, cm->nccl, s));
    }
    if (n_iter_out != nullptr) *n_iter_out = n_iter;
    MTP_HIP(hipStreamSynchronize(s));
    if (phase_ms_out != nullptr) {
        // averages over the iterations that ran (an attempt enqueued behind a pause or the end is a no-op)
        double acc[kMarks - 1] = {0, 0, 0, 0};
        int cnt = 0;
        for (size_t e = 0; e + kMarks <= ev_used && cnt < n_iter; e += kMarks, ++cnt) {
            for (int ph = 0; ph < kMarks - 1; ++ph) {
                float ms = 0.f;
                MTP_HIP(hipEventElapsedTime(&ms, ev[e + ph], ev[e + ph + 1]));
                acc[ph] += ms;
            }
        }
        for (int ph = 0; ph < kMarks - 1; ++ph) phase_ms_out[ph] = cnt ? (float)(acc[ph] / cnt) : 0.f;
    }
    return MTP_OK;
}

extern "C" int mtp_lloyd_f32(const float *X, int64_t n, int d, float *C_inout, int k, int max_iter, double tol_abs,
                             int check_every, int32_t *labels_out, double *inertia_out, int *n_iter_out, int options,
                             void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(X && C_inout && labels_out && ws, "mtp_lloyd: null pointer");
    MTP_REQUIRE(n >= k && k >= 1 && d >= 1 && max_iter >= 1, "mtp_lloyd: bad shape n=%lld k=%d d=%d max_iter=%d",
                (long long)n, k, d, max_iter);
    return lloyd_run(nullptr, X, n, 0, d, C_inout, k, max_iter, tol_abs, check_every, labels_out, inertia_out, n_iter_out,
                     nullptr, options, ws, ws_bytes, stream);
}

extern "C" int mtp_lloyd_sharded_f32(const mtp_comm *comm, const float *X_local, int64_t n_local, int64_t row_offset,
                                     int64_t n_total, int d, float *C_inout, int k, int max_iter, double tol_abs,
                                     int check_every, int32_t *labels_out, double *inertia_out, int *n_iter_out,
                                     float *phase_ms_out, int options, void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(comm != nullptr || (row_offset == 0 && n_local == n_total),
                "mtp_lloyd_sharded: without a communicator the rank must hold all %lld points", (long long)n_total);
    MTP_REQUIRE(C_inout && ws && (n_local == 0 || (X_local && labels_out)), "mtp_lloyd_sharded: null pointer");
    MTP_REQUIRE(n_local >= 0 && row_offset >= 0 && row_offset + n_local <= n_total && n_total >= k && k >= 1 && d >= 1 &&
                    max_iter >= 1,
                "mtp_lloyd_sharded: bad shape n_local=%lld offset=%lld n_total=%lld k=%d d=%d max_iter=%d", (long long)n_local,
                (long long)row_offset, (long long)n_total, k, d, max_iter);
    if (comm != nullptr) {
        int dev = -1;
        MTP_HIP(hipGetDevice(&dev));
        MTP_REQUIRE(dev == comm->device, "mtp_lloyd_sharded: the communicator was created on device %d, current device is %d",
                    comm->device, dev);
    }
    return lloyd_run(comm, X_local, n_local, row_offset, d, C_inout, k, max_iter, tol_abs, check_every, labels_out,
                     inertia_out, n_iter_out, phase_ms_out, options, ws, ws_bytes, stream);
}

// ---- n_init restarts at once: sklearn KMeans.fit's loop over inits (cluster/_kmeans.py:1495-1532) ----
// Every restart runs _kmeans_single_lloyd on the same points from its own initial centres; the
// restarts share each launch (blockIdx.y / .z = restart), keep their own loop state and stop
// on their own.  At reference scale (N ~ 7e4, K = 200) one restart cannot fill the GPU and an
// iteration is launch-bound; 20 restarts per launch are what makes the fit fast.
// the restart-grouped centroid images (k_assign_mfma_grouped), whatever the grouping: ceil(batch / R) images of
// ceil(R * kp / 32) tiles rounded up to 2, an image holding at most 64 tiles -- bounded by
// batch * (ceil(kp / 32) + 2) + 66 tiles
static size_t batched_image_bytes(int k, int d, int batch) {
    if (!mtp_mfma_dim(d)) return 0;
    const int kp = (k + 7) / 8 * 8;
    return align_up(((size_t)batch * (size_t)((kp + 31) / 32 + 2) + 66) * mtp_tile_floats(d) * 4, 256);
}

extern "C" int64_t mtp_lloyd_batched_workspace_bytes(int k, int d, int batch, int64_t n) {
    if (k <= 0 || d <= 0 || batch <= 0 || n < 0) return 0;
    // per restart the Lloyd buffers; the far-point scratch; the packed flags; batch * n floats of per-point distances
    // (the float32 inertia chains of MTP_LLOYD_SUMS_SKLEARN_1T)
    return (int64_t)(carve_lloyd(nullptr, k, d).bytes * (size_t)batch + (size_t)mtp_far_points_scratch_bytes(n) +
                     align_up((size_t)batch * MTP_FLAG_WORDS * 4, 256) + align_up((size_t)batch * (size_t)n * 4, 256) +
                     batched_image_bytes(k, d, batch));
}

// the restarts' flag words, gathered into one contiguous block for a single plain copy to the host
__global__ void k_collect_flags(const int32_t *flags, size_t stride_words, int batch, int32_t *out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < batch * MTP_FLAG_WORDS) out[e] = flags[(size_t)(e / MTP_FLAG_WORDS) * stride_words + e % MTP_FLAG_WORDS];
}

extern "C" int mtp_lloyd_batched_f32(const float *X, int64_t n, int d, float *C_inout, int k, int batch, int max_iter,
                                     double tol_abs, int check_every, int32_t *labels_out, double *inertia_out,
                                     int *n_iter_out, int options, void *ws, size_t ws_bytes, void *stream) {
    MTP_REQUIRE(X && C_inout && labels_out && ws, "mtp_lloyd_batched: null pointer");
    MTP_REQUIRE(n >= k && k >= 1 && d >= 1 && max_iter >= 1, "mtp_lloyd_batched: bad shape n=%lld k=%d d=%d max_iter=%d",
                (long long)n, k, d, max_iter);
    MTP_REQUIRE(batch >= 1 && batch <= 1024, "mtp_lloyd_batched: batch must be in [1, 1024] (got %d)", batch);
    MTP_REQUIRE(mtp_mfma_dim(d), "mtp_lloyd_batched: d must be 64, 96 or 128 (got %d); run the restarts one by one", d);
    const LloydWs w = carve_lloyd(ws, k, d);  // restart 0; restart b sits b * w.bytes further
    const size_t need = (size_t)mtp_lloyd_batched_workspace_bytes(k, d, batch, n);
    if (ws_bytes < need) {
        mtp_set_error("mtp_lloyd_batched: workspace of %zu bytes, need %zu", ws_bytes, need);
        return MTP_E_WORKSPACE;
    }
    float *dist_scratch = reinterpret_cast<float *>(static_cast<char *>(ws) + w.bytes * (size_t)batch);
    int32_t *flags_packed = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(dist_scratch) + (size_t)mtp_far_points_scratch_bytes(n));
    float *per_point = reinterpret_cast<float *>(reinterpret_cast<char *>(flags_packed) + align_up((size_t)batch * MTP_FLAG_WORDS * 4, 256));
    float *images = reinterpret_cast<float *>(reinterpret_cast<char *>(per_point) + align_up((size_t)batch * (size_t)n * 4, 256));
    const int sums_mode = options & MTP_LLOYD_SUMS_SKLEARN_1T;
    MTP_REQUIRE(sums_mode == 0 || k <= 8192, "mtp_lloyd_batched: MTP_LLOYD_SUMS_SKLEARN_1T supports k <= 8192, got k = %d", k);
    hipStream_t s = mtp_stream(stream);
    if (check_every < 1) check_every = 1;
    // restarts that share a centroid image and a workgroup (assign_kernel.h, k_assign_mfma_grouped); MTP_RESTART_GROUP
    // overrides the choice (0 / 1: one image per restart) -- the labels do not depend on it
    int cus = 0;
    if (const int r = mtp_device_cu_count(&cus)) return r;
    int group_r = mtp_pick_restart_group(n, k, batch, cus);
    if (const char *e = getenv("MTP_RESTART_GROUP")) {
        group_r = atoi(e);
        if (group_r <= 1) group_r = 0;
        if (group_r > batch) group_r = batch;
        if (group_r > 8) group_r = 8;
        while (group_r > 1 && (group_r * ((k + 7) / 8 * 8) + 31) / 32 > 64) --group_r;
        if (group_r <= 1) group_r = 0;
    }
    MtpBatchStride bs = mtp_no_batch();
    bs.staged = w.bytes / 4;
    bs.labels = (size_t)n;
    bs.red = w.bytes / 8;
    bs.flags = w.bytes / 4;
    bs.c0 = (size_t)k * d;
    bs.c1 = w.bytes / 4;
    bs.kvec = w.bytes / 4;
    bs.skip_mode = 0;
    bs.batch = batch;
    float *staged = w.staged;  // what the E-step / M-step kernels get as `staged`: restart 0's image, or the group images
    if (group_r > 0) {
        bs.group_r = group_r;
        bs.kp = (k + 7) / 8 * 8;
        const int tiles = ((group_r * bs.kp + 31) / 32 + 1) / 2 * 2;  // whole 2-tile stages
        bs.image_floats = (size_t)tiles * mtp_tile_floats(d);
        const int n_groups = (batch + group_r - 1) / group_r;
        MTP_REQUIRE((size_t)n_groups * bs.image_floats * 4 <= batched_image_bytes(k, d, batch), "mtp_lloyd_batched: image area (internal error)");
        staged = images;
        int rci = mtp_launch_clear_images(images, n_groups, tiles, d, bs.image_floats, s);
        if (rci != MTP_OK) return rci;
    }
    auto at = [&](auto *p, int b) { return reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + w.bytes * (size_t)b); };

    // begin: zero red + flags of every restart, stage the initial centres, labels_old = -1
    MTP_HIP(hipMemset2DAsync(w.red, w.bytes, 0, (size_t)mtp_red_len(k, d) * 8, batch, s));
    MTP_HIP(hipMemset2DAsync(w.flags, w.bytes, 0, MTP_FLAG_WORDS * 4, batch, s));
    int rc = mtp_launch_stage_centers(C_inout, k, d, staged, w.c2, nullptr, s, bs);
    if (rc != MTP_OK) return rc;
    MTP_HIP(hipMemsetAsync(labels_out, 0xff, (size_t)n * 4 * batch, s));

    std::vector<int32_t> hflags((size_t)batch * MTP_FLAG_WORDS);
    auto read_flags = [&]() -> int {
        const int words = batch * MTP_FLAG_WORDS;
        hipLaunchKernelGGL(k_collect_flags, dim3((words + 255) / 256), dim3(256), 0, s, w.flags, w.bytes / 4, batch,
                           flags_packed);
        MTP_LAUNCH_CHECK();
        MTP_HIP(hipMemcpyAsync(hflags.data(), flags_packed, (size_t)words * 4, hipMemcpyDeviceToHost, s));
        MTP_HIP(hipStreamSynchronize(s));
        return MTP_OK;
    };
    const int max_rounds = 2 * (max_iter / check_every + 1) + 2 * max_iter + 8;
    for (int round = 0;; ++round) {
        MTP_REQUIRE(round < max_rounds, "mtp_lloyd_batched: the restarts did not finish (internal error)");
        for (int i = 0; i < check_every; ++i) {
            // centres of a restart: buffer (n_iter & 1); the MFMA E-step only reads the staged image
            rc = mtp_launch_assign(X, n, d, nullptr, staged, w.c2, k, labels_out, w.red + (size_t)k * d + k, w.flags, s, bs);
            if (rc != MTP_OK) return rc;
            rc = mtp_launch_accumulate(X, n, d, labels_out, k, w.red, w.red + (size_t)k * d, w.flags, s, bs, sums_mode);
            if (rc != MTP_OK) return rc;
            rc = mtp_launch_mstep(w.red, C_inout, w.c_alt, k, d, tol_abs, max_iter, 0, staged, w.c2, w.shift, w.flags,
                                  nullptr, s, bs, 1, sums_mode);
            if (rc != MTP_OK) return rc;
        }
        rc = read_flags();
        if (rc != MTP_OK) return rc;
        bool any_live = false;
        for (int b = 0; b < batch; ++b) {
            const int32_t *f = &hflags[(size_t)b * MTP_FLAG_WORDS];
            if (f[MTP_FLAG_STATE] == MTP_STATE_RELOCATE) {
                // this restart paused before its M-step: relocate, then redo its M-step alone
                const int old = f[MTP_FLAG_N_ITER] & 1;
                const int m = f[MTP_FLAG_N_EMPTY];
                float *c0 = C_inout + (size_t)b * k * d, *c1 = at(w.c_alt, b);
                rc = mtp_far_points_f32(X, n, d, old ? c1 : c0, labels_out + (size_t)b * n, m, 0, at(w.far_val, b),
                                        at(w.far_idx, b), at(w.far_lab, b), at(w.far_rows, b), dist_scratch, stream);
                if (rc != MTP_OK) return rc;
                rc = mtp_launch_apply_relocation(at(w.red, b), k, d, m, at(w.far_val, b), at(w.far_lab, b), at(w.far_rows, b),
                                                 sums_mode, s);
                if (rc != MTP_OK) return rc;
                MTP_HIP(hipMemsetAsync(at(w.flags, b) + MTP_FLAG_STATE, 0, 4, s));
                MtpBatchStride one = bs;
                one.batch = 1;
                one.b_first = b;  // grouped images: the restart's rows sit in its group's image
                rc = mtp_launch_mstep(at(w.red, b), c0, c1, k, d, tol_abs, max_iter, 1, group_r > 0 ? staged : at(w.staged, b),
                                      at(w.c2, b), at(w.shift, b), at(w.flags, b), nullptr, s, one, 1, sums_mode);
                if (rc != MTP_OK) return rc;
                any_live = true;
            } else if (f[MTP_FLAG_STATE] == MTP_STATE_RUNNING) {
                any_live = true;
            }
        }
        if (getenv("MTP_DEBUG")) {
            fprintf(stderr, "[mtp_lloyd_batched] round %d states:", round);
            for (int b = 0; b < batch; ++b) fprintf(stderr, " %d/%d", hflags[(size_t)b * MTP_FLAG_WORDS], hflags[(size_t)b * MTP_FLAG_WORDS + 1]);
            fprintf(stderr, "\n");
        }
        if (!any_live) break;
    }
    // labels must match the final centres: one more E-step for every restart that did not stop on
    // "labels unchanged" (_kmeans.py:736-748); staged already describes the final centres
    MtpBatchStride fin = bs;
    fin.skip_mode = 1;
    rc = mtp_launch_assign(X, n, d, nullptr, staged, w.c2, k, labels_out, nullptr, w.flags, s, fin);
    if (rc != MTP_OK) return rc;
    if (inertia_out != nullptr) MTP_HIP(hipMemsetAsync(inertia_out, 0, 8 * (size_t)batch, s));
    for (int b = 0; b < batch; ++b) {
        const int n_iter = hflags[(size_t)b * MTP_FLAG_WORDS + MTP_FLAG_N_ITER];
        float *c0 = C_inout + (size_t)b * k * d;
        if (n_iter & 1) MTP_HIP(hipMemcpyAsync(c0, at(w.c_alt, b), (size_t)k * d * 4, hipMemcpyDeviceToDevice, s));
        if (n_iter_out != nullptr) n_iter_out[b] = n_iter;
    }
    if (inertia_out != nullptr) {  // all restarts in one launch
        rc = sums_mode != 0
                 ? mtp_launch_inertia_chain_f32(X, n, d, C_inout, labels_out, inertia_out, per_point, batch, (size_t)k * d, (size_t)n, s)
                 : mtp_launch_inertia_batched(X, n, d, C_inout, labels_out, inertia_out, nullptr, batch, (size_t)k * d, (size_t)n, s);
        if (rc != MTP_OK) return rc;
    }
    MTP_HIP(hipStreamSynchronize(s));
    return MTP_OK;
}
