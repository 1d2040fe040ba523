// TEST TRANSPORT -- not a product path and never loaded unless MVS_CCL_LIBRARY names it.
//
// RCCL refuses a communicator whose ranks share one device, and a gpurun box has one MI355X, so the engine's multi-rank
// exchange (mvs_engine_exchange: count all-gather, in-place block broadcasts, commit of the union) could only ever run
// there with world = 1.  This library exports the eight nccl* entry points the engine binds (mvs_engine.cpp, struct Rccl)
// and moves the bytes through a POSIX shared-memory segment between PROCESSES OF ONE HOST: every collective is executed
// eagerly and synchronously (stream sync, device -> segment, barrier, segment -> device, barrier), group calls are no-ops.
// tests/test_gpu_dist.py points MVS_CCL_LIBRARY at it to run 2 and 3 engine ranks on the one GPU through the real C ABI.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>

namespace {
constexpr size_t kArea = 64u << 20;  // payload bytes per step; larger transfers go in steps
struct Header {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
};
struct Comm {
    Header* hdr = nullptr;
    unsigned char* area = nullptr;
    size_t mapped = 0;
    int rank = 0, world = 1;
    char name[64] = {0};
};
size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
// A rank that never arrives (it died, or the ranks disagree about the sequence of collectives) must not leave the others
// spinning on a test box: the wait gives up after kBarrierSeconds and the collective returns an error.
constexpr double kBarrierSeconds = 120.0;
double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
bool barrier(Comm* c) {
    const uint32_t gen = c->hdr->generation.load(std::memory_order_acquire);
    if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        c->hdr->arrived.store(0, std::memory_order_relaxed);
        c->hdr->generation.fetch_add(1, std::memory_order_release);
        return true;
    }
    const double t0 = now_s();
    for (unsigned spins = 0; c->hdr->generation.load(std::memory_order_acquire) == gen; ++spins) {
        sched_yield();
        if ((spins & 1023u) == 1023u && now_s() - t0 > kBarrierSeconds) return false;
    }
    return true;
}
bool ok(hipError_t e) { return e == hipSuccess; }
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<int> serial{0};
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/mvs_loopback_%d_%d", (int)getpid(), serial.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    Comm* c = new Comm;
    c->rank = rank; c->world = nranks;
    memcpy(c->name, id.internal, sizeof c->name - 1);
    c->mapped = 4096 + kArea;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);  // a fresh segment reads as zeros: the barrier starts at rest
    if (fd < 0 || ftruncate(fd, (off_t)c->mapped) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->mapped, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr = reinterpret_cast<Header*>(p);
    c->area = reinterpret_cast<unsigned char*>(p) + 4096;
    if (!barrier(c)) { munmap(p, c->mapped); delete c; return ncclSystemError; }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    (void)barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->hdr, c->mapped);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t b = count * type_bytes(t);
    if (!c || !b || b * (size_t)c->world > kArea) return ncclInvalidArgument;
    if (!ok(hipStreamSynchronize(st))) return ncclUnhandledCudaError;
    if (!ok(hipMemcpy(c->area + b * (size_t)c->rank, send, b, hipMemcpyDeviceToHost))) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (!ok(hipMemcpy(recv, c->area, b * (size_t)c->world, hipMemcpyHostToDevice))) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t st) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t b = count * type_bytes(t);
    if (!c || !type_bytes(t) || root < 0 || root >= c->world) return ncclInvalidArgument;
    if (!ok(hipStreamSynchronize(st))) return ncclUnhandledCudaError;
    for (size_t off = 0; off < b; off += kArea) {
        const size_t n = b - off < kArea ? b - off : kArea;
        if (c->rank == root) {
            if (!ok(hipMemcpy(c->area, static_cast<const unsigned char*>(send) + off, n, hipMemcpyDeviceToHost))) return ncclUnhandledCudaError;
        }
        if (!barrier(c)) return ncclSystemError;
        if (c->rank != root) {
            if (!ok(hipMemcpy(static_cast<unsigned char*>(recv) + off, c->area, n, hipMemcpyHostToDevice))) return ncclUnhandledCudaError;
        } else if (send != recv) {
            if (!ok(hipMemcpy(static_cast<unsigned char*>(recv) + off, static_cast<const unsigned char*>(send) + off, n, hipMemcpyDeviceToDevice))) return ncclUnhandledCudaError;
        }
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c || !count) return ncclInvalidArgument;
    *count = c->world;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "success";
        case ncclUnhandledCudaError: return "loopback transport: HIP error";
        case ncclSystemError: return "loopback transport: shared-memory segment, or a rank did not reach a collective in time";
        case ncclInvalidArgument: return "loopback transport: invalid argument";
        default: return "loopback transport: error";
    }
}

}  // extern "C"
