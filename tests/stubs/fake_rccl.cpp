// TEST-ONLY stand-in for librccl (selected with MCALF_RCCL_LIB): just enough of the NCCL API for
// mcalf_comm_* / mcalf_loglike_gatherv_device to run their N > 1 branch between two PROCESSES THAT SHARE ONE GPU
// (a GPU box of this pool has one card and RCCL refuses two ranks on one device).  Point-to-point messages travel
// through a POSIX shared-memory file: ncclSend synchronises the stream, copies the block to the (src, dst) mailbox
// and publishes a sequence number; ncclRecv polls for it (bounded: ncclSystemError after 60 s, never a hang) and
// copies it to the device.  Not a performance model and not shipped: it exercises control flow, offsets, the
// ragged counts and the NaN-block error path of the library.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat64 = 8 } ncclDataType_t;
struct ncclUniqueId { char internal[128]; };

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kBoxDoubles = 1 << 20;      // 8 MB per mailbox
struct Box {
    std::atomic<uint64_t> sent, taken;
    uint64_t count;
    double data[kBoxDoubles];
};
struct Shm {
    std::atomic<int> ready;
    Box box[kMaxRanks][kMaxRanks];           // [src][dst]
};
struct Comm {
    Shm* shm;
    int nranks, rank;
    char name[64];
};
struct Pending { bool send; void* buf; size_t count; int peer; Comm* c; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;


template <typename F>
bool poll(F ok) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!ok()) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

ncclResult_t do_send(const Pending& p) {
    Box& b = p.c->shm->box[p.c->rank][p.peer];
    if (p.count > kBoxDoubles) return ncclInvalidArgument;
    if (!poll([&] { return b.taken.load() == b.sent.load(); })) return ncclSystemError;      // previous message consumed
    if (hipStreamSynchronize(p.st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(b.data, p.buf, p.count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    b.count = p.count;
    b.sent.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t do_recv(const Pending& p) {
    Box& b = p.c->shm->box[p.peer][p.c->rank];
    if (!poll([&] { return b.sent.load() > b.taken.load(); })) return ncclSystemError;
    if (b.count != p.count) return ncclInvalidArgument;
    if (hipStreamSynchronize(p.st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(p.buf, b.data, p.count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    b.taken.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t run(const Pending& p) { return p.send ? do_send(p) : do_recv(p); }
}  // namespace

extern "C" {
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
        case ncclSystemError: return "peer did not show up within 60 s (fake rccl)";
        case ncclInvalidArgument: return "invalid argument (fake rccl)";
        default: return "internal error (fake rccl)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "/mcalf_fake_rccl_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(Comm** out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->nranks = nranks; c->rank = rank;
    std::snprintf(c->name, sizeof c->name, "%s", id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
    } else if (!poll([&] { return (fd = shm_open(c->name, O_RDWR, 0600)) >= 0; })) {
        return ncclSystemError;
    }
    void* m = MAP_FAILED;
    if (!poll([&] { return (m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0)) != MAP_FAILED; })) return ncclSystemError;
    close(fd);
    c->shm = static_cast<Shm*>(m);           // a fresh shm file is zero-filled: every counter starts at 0
    c->shm->ready.fetch_add(1);
    if (!poll([&] { return c->shm->ready.load() >= nranks; })) return ncclSystemError;        // collective, like the real one
    *out = c;
    return ncclSuccess;
}

static ncclResult_t release(Comm* c) {
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(Comm* c) { return release(c); }
ncclResult_t ncclCommAbort(Comm* c) { return release(c); }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    for (const Pending& p : g_pending)
        if (r == ncclSuccess) r = run(p);
    g_pending.clear();
    return r;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, Comm* c, hipStream_t st) {
    if (dt != ncclFloat64 || !c || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    Pending p{true, const_cast<void*>(buf), count, peer, c, st};
    if (g_depth > 0) { g_pending.push_back(p); return ncclSuccess; }
    return run(p);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, Comm* c, hipStream_t st) {
    if (dt != ncclFloat64 || !c || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    Pending p{false, buf, count, peer, c, st};
    if (g_depth > 0) { g_pending.push_back(p); return ncclSuccess; }
    return run(p);
}
}
