// Development micro-benchmark (not part of the product): times variants of the MFMA E-step kernel
// on one GPU, interleaved in one process, and checks that every variant with the epilogue
// produces identical labels.  Build: make -C tools   Run (GPU box): tools/assign_tune [N] [K]
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

static MtpBatchStride g_bs = mtp_no_batch();  // restarts: same points, own centroid image and labels

// filter: substrings separated by '|'
static bool name_matches(const char *name, const char *filter) {
    char buf[256];
    strncpy(buf, filter, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = 0;
    for (char *tok = strtok(buf, "|"); tok != nullptr; tok = strtok(nullptr, "|"))
        if (strstr(name, tok)) return true;
    return false;
}

struct Variant {
    const char *name;
    void (*launch)(const float *, int64_t, const float *, int, int32_t *, hipStream_t);
    bool epi;
};

template <int D, int PT, int CTW, int WPB, int WPS, bool EPI, int LG = 8, bool PRIO = false>
void launch_v(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    const size_t lds = 2 * (size_t)CTW * mtp_tile_floats(D) * sizeof(float);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign_mfma<D, PT, CTW, WPB, WPS, EPI, LG, PRIO>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set = true;
    }
    const int64_t ppb = WPB * 32 * PT;
    hipLaunchKernelGGL((k_assign_mfma<D, PT, CTW, WPB, WPS, EPI, LG, PRIO>), dim3((unsigned)((n + ppb - 1) / ppb), g_bs.batch), dim3(64 * WPB), lds, s, X,
                       n, staged, (k + 31) / 32, labels, nullptr, nullptr, g_bs);
    CK(hipGetLastError());
}

// the product's batched launch as the fit issues it: per-restart loop state read first, changed labels counted
static double *g_changed = nullptr;  // [batch] (stride g_bs.red)
static int32_t *g_state = nullptr;   // [batch] (stride g_bs.flags), all RUNNING
template <int D, bool STATE = true, bool CHANGED = true>
void launch_flags(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    const size_t lds = 2 * (size_t)2 * mtp_tile_floats(D) * sizeof(float);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign_mfma<D, 1, 2, 8, 4, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set = true;
    }
    hipLaunchKernelGGL((k_assign_mfma<D, 1, 2, 8, 4, true>), dim3((unsigned)((n + 255) / 256), g_bs.batch), dim3(512), lds, s, X, n,
                       staged, (k + 31) / 32, labels, CHANGED ? g_changed : nullptr, STATE ? g_state : nullptr, g_bs);
    CK(hipGetLastError());
}

template <int D, int WPB, int WPS, int LG = 8>
void launch_pipe(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    const size_t lds = 2 * (size_t)2 * mtp_tile_floats(D) * sizeof(float) + (size_t)mtp_n_tiles(k) * 32 * sizeof(float);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign_mfma_pipe<D, WPB, WPS, LG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set = true;
    }
    const int64_t ppb = WPB * 32;
    hipLaunchKernelGGL((k_assign_mfma_pipe<D, WPB, WPS, LG>), dim3((unsigned)((n + ppb - 1) / ppb), g_bs.batch), dim3(64 * WPB), lds, s, X, n,
                       staged, mtp_n_tiles(k), labels, nullptr, nullptr, g_bs);
    CK(hipGetLastError());
}

// The plain grid in two launches: whole rounds of 512 workgroups, then the rest with workgroups of WT waves,
// WT chosen so that at most one workgroup per CU is needed (the hardware spreads <= 256 workgroups one per CU).
template <int D, int WT>
void launch_tail(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    launch_v<D, 1, 2, WT, 4, true>(X, n, staged, k, labels, s);
}
template <int D>
void launch_split(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    const int cus = 256, slots = 2 * cus;
    const int64_t blocks = (n + 255) / 256;
    const int64_t full = blocks / slots * slots;
    const int64_t n_main = full * 256 < n ? full * 256 : n;
    const int64_t n_tail = n - n_main;
    if (n_main > 0) launch_v<D, 1, 2, 8, 4, true>(X, n_main, staged, k, labels, s);
    if (n_tail <= 0) return;
    const int64_t waves = (n_tail + 31) / 32;
    const int w = (int)((waves + cus - 1) / cus);
    const float *Xt = X + n_main * D;
    int32_t *lt = labels + n_main;
    if (w <= 4) launch_tail<D, 4>(Xt, n_tail, staged, k, lt, s);
    else if (w <= 8) launch_tail<D, 8>(Xt, n_tail, staged, k, lt, s);
    else if (w <= 12) launch_tail<D, 12>(Xt, n_tail, staged, k, lt, s);
    else launch_tail<D, 16>(Xt, n_tail, staged, k, lt, s);
}

// the direct (no-LDS) variant: its image is made from the staged one on first use
static float *g_direct = nullptr;
template <int D, int WPB, int PF, bool EPI = true>
void launch_direct(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    const int tiles = mtp_n_tiles(k);
    if (g_direct == nullptr) {
        CK(hipMalloc(&g_direct, (size_t)tiles * mtp_direct_tile_f4<D>() * 16));
        hipLaunchKernelGGL(k_direct_image<D>, dim3(tiles), dim3(256), 0, s, staged, g_direct);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    const int64_t ppb = WPB * 32;
    hipLaunchKernelGGL((k_assign_mfma_direct<D, WPB, PF, EPI>), dim3((unsigned)((n + ppb - 1) / ppb)), dim3(64 * WPB), 0, s, X, n,
                       g_direct, (k + 31) / 32, labels);
    CK(hipGetLastError());
}

static unsigned *g_tickets = nullptr;
// SMALL: small units per workgroup at the end of the queue (-1: everything in small units)
template <int D, int WPS, int SMALL>
void launch_queue(const float *X, int64_t n, const float *staged, int k, int32_t *labels, hipStream_t s) {
    const size_t lds = 2 * (size_t)2 * mtp_tile_floats(D) * sizeof(float);
    static bool set = false;
    static int cus = 0;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign_mfma_queue<D, WPS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        cus = prop.multiProcessorCount;
        set = true;
    }
    const int grid = 2 * cus;
    const int64_t T4 = (n + 127) / 128;
    int64_t small = SMALL < 0 ? T4 : ((int64_t)SMALL * grid + g_bs.batch - 1) / g_bs.batch;
    if (small > T4) small = T4;
    MtpAssignQueue q;
    q.ticket = g_tickets;
    q.big_per_restart = (int)((T4 - small) / 2);
    q.small_per_restart = (int)(T4 - 2 * (int64_t)q.big_per_restart);
    hipLaunchKernelGGL((k_assign_mfma_queue<D, WPS>), dim3(grid), dim3(512), lds, s, X, n, staged, (k + 31) / 32, labels,
                       nullptr, nullptr, g_bs, q, nullptr);
    CK(hipGetLastError());
}

template <int D>
int run(int64_t n, int k, const char *filter, int batch) {
    float *X, *C, *staged, *c2;
    int32_t *labels, *ref;
    CK(hipMalloc(&X, n * D * 4));
    CK(hipMalloc(&C, (size_t)k * D * 4));
    const size_t image = (size_t)mtp_n_tiles(k) * mtp_tile_floats(D);
    CK(hipMalloc(&staged, image * 4 * batch));
    CK(hipMalloc(&c2, k * 4));
    CK(hipMalloc(&labels, n * 4 * batch));
    CK(hipMalloc(&ref, n * 4 * batch));
    CK(hipMalloc(&g_tickets, 256));
    CK(hipMemset(g_tickets, 0, 256));
    k_fill<<<4096, 256>>>(X, (size_t)n * D, 1);
    k_fill<<<256, 256>>>(C, (size_t)k * D, 7);
    for (int b = 0; b < batch; ++b)  // every restart gets the same centroids (the timing does not care)
        if (mtp_launch_stage_centers(C, k, D, staged + b * image, c2, nullptr, 0) != 0) { printf("stage failed\n"); return 1; }
    g_bs = mtp_no_batch();
    g_bs.batch = batch;
    g_bs.staged = image;
    g_bs.labels = (size_t)n;
    g_bs.red = 512;    // doubles between the restarts' changed counters
    g_bs.flags = 512;  // int32s between the restarts' loop states
    CK(hipMalloc(&g_changed, (size_t)batch * 512 * 8));
    CK(hipMalloc(&g_state, (size_t)batch * 512 * 4));
    CK(hipMemset(g_changed, 0, (size_t)batch * 512 * 8));
    {
        std::vector<int32_t> st((size_t)batch * 512, MTP_STATE_RUNNING);
        CK(hipMemcpy(g_state, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs = {
        {"PT1 CTW2 WPB8 WPS4 (one launch)", launch_v<D, 1, 2, 8, 4, true>, true},
        {"SPLIT rounds + tail (product)", launch_split<D>, true},
        {"PT1 CTW2 WPB8 WPS4 with loop state + changed count", launch_flags<D>, true},
        {"PT1 CTW2 WPB8 WPS4 with loop state only", launch_flags<D, true, false>, true},
        {"PT1 CTW2 WPB8 WPS4 with changed count only", launch_flags<D, false, true>, true},
        {"QUEUE small 0", launch_queue<D, 4, 0>, true},
        {"QUEUE small 1/wg", launch_queue<D, 4, 1>, true},
        {"QUEUE small 2/wg", launch_queue<D, 4, 2>, true},
        {"QUEUE small 3/wg", launch_queue<D, 4, 3>, true},
        {"QUEUE small 4/wg", launch_queue<D, 4, 4>, true},
        {"QUEUE all small", launch_queue<D, 4, -1>, true},
        {"PIPE WPB8 WPS4", launch_pipe<D, 8, 4>, true},
        {"PT1 CTW2 WPB8 WPS4 LG16", launch_v<D, 1, 2, 8, 4, true, 16>, true},
        {"PT1 CTW2 WPB8 WPS4 LG16 PRIO", launch_v<D, 1, 2, 8, 4, true, 16, true>, true},
        {"PT1 CTW4 WPB16 WPS4", launch_v<D, 1, 4, 16, 4, true>, true},
        {"PT1 CTW1 WPB8 WPS4", launch_v<D, 1, 1, 8, 4, true>, true},
        {"PT1 CTW2 WPB8 WPS4 no-epi", launch_v<D, 1, 2, 8, 4, false>, false},
        {"PT1 CTW1 WPB4 WPS4", launch_v<D, 1, 1, 4, 4, true>, true},
        {"PT1 CTW2 WPB16 WPS4", launch_v<D, 1, 2, 16, 4, true>, true},
        {"DIRECT WPB4 PF2", launch_direct<D, 4, 2>, true},
        {"DIRECT WPB4 PF4", launch_direct<D, 4, 4>, true},
        {"DIRECT WPB8 PF4", launch_direct<D, 8, 4>, true},
        {"DIRECT WPB4 PF4 no-epi", launch_direct<D, 4, 4, false>, false},
    };
    if (filter != nullptr) {  // keep the variants whose name contains it (the product variant always stays)
        std::vector<Variant> keep;
        for (size_t v = 0; v < vs.size(); ++v)
            if (v == 0 || name_matches(vs[v].name, filter)) keep.push_back(vs[v]);
        vs = keep;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = 4;
    std::vector<std::vector<float>> ms(vs.size());
    vs[0].launch(X, n, staged, k, ref, 0);
    CK(hipDeviceSynchronize());
    const int64_t nl = n * batch;
    std::vector<int32_t> href(nl), hl(nl);
    CK(hipMemcpy(href.data(), ref, nl * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < rounds; ++r) {
        for (size_t v = 0; v < vs.size(); ++v) {
            CK(hipMemset(labels, 0xff, nl * 4));
            CK(hipEventRecord(e0, 0));
            vs[v].launch(X, n, staged, k, labels, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms[v].push_back(t);
            if (r == 0 && vs[v].epi) {
                CK(hipMemcpy(hl.data(), labels, nl * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (int64_t i = 0; i < nl; ++i) bad += hl[i] != href[i];
                if (bad) printf("  !! %s: %zu label mismatches vs product variant\n", vs[v].name, bad);
            }
        }
    }
    const double flop = 2.0 * n * k * D * batch;
    for (size_t v = 0; v < vs.size(); ++v) {
        float mn = 1e30f, sum = 0;
        for (int r = 1; r < rounds; ++r) { mn = ms[v][r] < mn ? ms[v][r] : mn; sum += ms[v][r]; }
        const float avg = sum / (rounds - 1);
        printf("%-30s min %7.3f ms  avg %7.3f ms  %6.1f TFLOP/s (%4.1f%% of 157.3)\n", vs[v].name, mn, avg,
               flop / (mn * 1e-3) / 1e12, flop / (mn * 1e-3) / 1e12 / 157.3 * 100);
    }
    return 0;
}

// tools/assign_tune [N] [K] [name filter | -] [D: 64 | 96 | 128] [restarts]
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int k = argc > 2 ? atoi(argv[2]) : 1024;
    const char *filter = argc > 3 && strcmp(argv[3], "-") != 0 ? argv[3] : nullptr;
    const int d = argc > 4 ? atoi(argv[4]) : 128;
    const int batch = argc > 5 ? atoi(argv[5]) : 1;
    if (d == 96) return run<96>(n, k, filter, batch);
    if (d == 64) return run<64>(n, k, filter, batch);
    return run<128>(n, k, filter, batch);
}
