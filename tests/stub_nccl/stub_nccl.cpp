// TEST INFRASTRUCTURE, not part of the product: a stand-in for the six RCCL entry points libmtp_accel.so binds
// at run time (csrc/capi.hip, rccl()), so that 2-3 PROCESSES sharing ONE GPU can run mtp_lloyd_sharded_f32 --
// RCCL itself refuses two ranks on one device, and the round's GPU box has one.  Selected with
// MTP_RCCL_LIB=<path of this .so>.
//
// Exchange: POSIX shared memory + a sense-reversing barrier.  A collective drains the caller's stream, copies the
// send buffer to this rank's slot of the segment, meets the others, combines the slots in RANK ORDER (so every
// rank forms the same bits) and copies the result back to the device.  Host-synchronous where RCCL is
// stream-ordered: same results, no overlap -- this is for correctness tests, never for timing.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t kSlotBytes = 16u << 20;  // per rank; the largest message of the tests is ~2 MB
constexpr int kMaxWorld = 16;
constexpr int kOk = 0, kUnhandledHip = 1, kSystemError = 2, kInvalidArgument = 4;

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    int world;
};

struct Comm {
    Header *hdr;
    char *slots;
    size_t map_bytes;
    int rank, world;
    char name[64];
};

struct UniqueId {
    char internal[128];
};

size_t segment_bytes(int world) { return 4096 + (size_t)world * kSlotBytes; }

bool barrier(Comm *c) {
    Header *h = c->hdr;
    const int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spin = 0;; ++spin) {
        if (h->generation.load(std::memory_order_acquire) != gen) return true;
        if ((spin & 1023) == 1023) {
            usleep(50);
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 100) {  // a rank that left the collective sequence: fail, do not hang
                fprintf(stderr, "[stub_nccl] rank %d: barrier timed out\n", c->rank);
                return false;
            }
        }
    }
}

size_t dtype_size(int dt) {
    switch (dt) {
        case 0: case 1: return 1;   // ncclInt8 / ncclChar, ncclUint8
        case 2: case 3: case 7: return 4;   // int32, uint32, float32
        case 4: case 5: case 8: return 8;   // int64, uint64, float64
        default: return 0;
    }
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(int e) {
    switch (e) {
        case kOk: return "no error";
        case kUnhandledHip: return "stub_nccl: HIP call failed";
        case kSystemError: return "stub_nccl: shared-memory segment / barrier failed";
        case kInvalidArgument: return "stub_nccl: invalid argument (dtype, op or message larger than a slot)";
        default: return "stub_nccl: unknown error";
    }
}

int ncclGetUniqueId(UniqueId *id) {
    if (id == nullptr) return kInvalidArgument;
    memset(id, 0, sizeof(*id));
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/mtp_stub_nccl_%d_%ld_%ld", (int)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return kOk;
}

int ncclCommInitRank(Comm **out, int world, UniqueId id, int rank) {
    if (out == nullptr || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/') return kInvalidArgument;
    Comm *c = new Comm{};
    c->rank = rank;
    c->world = world;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = segment_bytes(world);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);   // whoever comes first creates it (zero-filled)
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
        perror("[stub_nccl] shm_open/ftruncate");
        delete c;
        return kSystemError;
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        perror("[stub_nccl] mmap");
        delete c;
        return kSystemError;
    }
    c->hdr = static_cast<Header *>(p);
    c->slots = static_cast<char *>(p) + 4096;
    c->hdr->attached.fetch_add(1);
    if (!barrier(c)) return kSystemError;       // everyone is attached ...
    if (rank == 0) shm_unlink(c->name);         // ... so the name can go: the segment dies with its last mapping
    *out = c;
    return kOk;
}

int ncclCommDestroy(Comm *c) {
    if (c == nullptr) return kOk;
    munmap(c->hdr, c->map_bytes);
    delete c;
    return kOk;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, Comm *c, hipStream_t s) {
    const size_t es = dtype_size(dtype);
    if (c == nullptr || op != 0 || (dtype != 7 && dtype != 8) || count * es > kSlotBytes) return kInvalidArgument;
    const size_t bytes = count * es;
    if (hipStreamSynchronize(s) != hipSuccess) return kUnhandledHip;
    if (bytes && hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledHip;
    if (!barrier(c)) return kSystemError;
    std::vector<char> acc(bytes);
    if (dtype == 8) {
        double *a = reinterpret_cast<double *>(acc.data());
        for (size_t i = 0; i < count; ++i) a[i] = 0.0;
        for (int r = 0; r < c->world; ++r) {
            const double *q = reinterpret_cast<const double *>(c->slots + (size_t)r * kSlotBytes);
            for (size_t i = 0; i < count; ++i) a[i] += q[i];
        }
    } else {
        float *a = reinterpret_cast<float *>(acc.data());
        for (size_t i = 0; i < count; ++i) a[i] = 0.f;
        for (int r = 0; r < c->world; ++r) {
            const float *q = reinterpret_cast<const float *>(c->slots + (size_t)r * kSlotBytes);
            for (size_t i = 0; i < count; ++i) a[i] += q[i];
        }
    }
    if (!barrier(c)) return kSystemError;  // everyone has read every slot: they may be overwritten
    if (bytes && hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandledHip;
    return kOk;
}

int ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, Comm *c, hipStream_t s) {
    const size_t es = dtype_size(dtype);
    if (c == nullptr || es == 0 || sendcount * es > kSlotBytes) return kInvalidArgument;
    const size_t bytes = sendcount * es;
    if (hipStreamSynchronize(s) != hipSuccess) return kUnhandledHip;
    if (bytes && hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledHip;
    if (!barrier(c)) return kSystemError;
    std::vector<char> all(bytes * (size_t)c->world);
    for (int r = 0; r < c->world; ++r) memcpy(all.data() + (size_t)r * bytes, c->slots + (size_t)r * kSlotBytes, bytes);
    if (!barrier(c)) return kSystemError;
    if (bytes && hipMemcpy(recv, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return kUnhandledHip;
    return kOk;
}

}  // extern "C"
