// TEST-ONLY stand-in for librccl (selected with MCALF_RCCL_LIB): just enough of the NCCL API for
// mcalf_comm_* / mcalf_loglike_gatherv_device to run their N > 1 branch between two PROCESSES THAT SHARE ONE GPU
// (a GPU box of this pool has one card and RCCL refuses two ranks on one device).  Point-to-point messages travel
// through a POSIX shared-memory file, one mailbox per (src, dst) with a sequence number.
//
// Like the real library the calls are ASYNCHRONOUS and STREAM-ORDERED: ncclSend / ncclRecv (and ncclGroupEnd) only
// enqueue work on the caller's stream and return --
//   send:  [host function: wait until the mailbox is free]  D2H copy into a page-locked block
//          [host function: block -> mailbox, publish the sequence number]
//   recv:  [host function: wait for the message (and FAKE_RCCL_DELAY_US more, so that "the data is there as soon as the
//          call returns" can never be what a test relies on), mailbox -> page-locked block]  H2D copy
//          [host function: mark the mailbox free]
// -- so what the caller's buffers hold, and when, depends on the stream / event chain the library builds around the
// exchange, as it does with RCCL.  Every wait is bounded (60 s; the message is then dropped, an error is latched and
// returned by the next call): never a hang.  FAKE_RCCL_SYNC=1 selects the older blocking form (the stream is synchronised
// inside every call), which messages too large for the page-locked blocks use as well.
// Not a performance model and not shipped: it exercises control flow, offsets, ragged counts, ordering and the
// NaN-block error path of the library.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
constexpr int kRing = 32;                    // page-locked blocks of the asynchronous form, used round robin
constexpr size_t kStageDoubles = 1 << 16;
struct Comm {
    Shm* shm;
    int nranks, rank;
    char name[64];
    double* stage[kRing];
    unsigned next_stage;
    std::atomic<int> async_err;
    bool sync;
    long delay_us;
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

// ---- the asynchronous form: host functions in stream order (they run on the runtime's callback thread and call no HIP) ----
struct Op { Comm* c; Box* box; double* stage; size_t count; };

void cb_send_wait(void* v) {
    Op* op = static_cast<Op*>(v);
    if (!poll([&] { return op->box->taken.load() == op->box->sent.load(); })) op->c->async_err.store((int)ncclSystemError);
}
void cb_send_publish(void* v) {
    Op* op = static_cast<Op*>(v);
    if (op->c->async_err.load() == 0) {
        std::memcpy(op->box->data, op->stage, op->count * sizeof(double));
        op->box->count = op->count;
        op->box->sent.fetch_add(1);
    }
    delete op;
}
void cb_recv_wait(void* v) {
    Op* op = static_cast<Op*>(v);
    if (!poll([&] { return op->box->sent.load() > op->box->taken.load(); })) { op->c->async_err.store((int)ncclSystemError); return; }
    if (op->c->delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(op->c->delay_us));
    if (op->box->count != op->count) { op->c->async_err.store((int)ncclInvalidArgument); return; }
    std::memcpy(op->stage, op->box->data, op->count * sizeof(double));
}
void cb_recv_done(void* v) {
    Op* op = static_cast<Op*>(v);
    if (op->c->async_err.load() == 0) op->box->taken.fetch_add(1);
    delete op;
}

ncclResult_t enqueue(const Pending& p) {
    Comm* c = p.c;
    if (c->async_err.load() != 0) return (ncclResult_t)c->async_err.load();
    Op* op = new Op{c, p.send ? &c->shm->box[c->rank][p.peer] : &c->shm->box[p.peer][c->rank], c->stage[c->next_stage++ % kRing], p.count};
    const size_t bytes = p.count * sizeof(double);
    bool ok;
    if (p.send)
        ok = hipLaunchHostFunc(p.st, cb_send_wait, op) == hipSuccess &&
             hipMemcpyAsync(op->stage, p.buf, bytes, hipMemcpyDeviceToHost, p.st) == hipSuccess &&
             hipLaunchHostFunc(p.st, cb_send_publish, op) == hipSuccess;
    else
        ok = hipLaunchHostFunc(p.st, cb_recv_wait, op) == hipSuccess &&
             hipMemcpyAsync(p.buf, op->stage, bytes, hipMemcpyHostToDevice, p.st) == hipSuccess &&
             hipLaunchHostFunc(p.st, cb_recv_done, op) == hipSuccess;
    return ok ? ncclSuccess : ncclUnhandledCudaError;      // (on failure `op` may leak: the test fails anyway)
}

ncclResult_t run(const Pending& p) {
    if (!p.c->sync && p.count <= kStageDoubles) return enqueue(p);
    return p.send ? do_send(p) : do_recv(p);
}
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
    const char* e = std::getenv("FAKE_RCCL_SYNC");
    c->sync = e && *e && *e != '0';
    e = std::getenv("FAKE_RCCL_DELAY_US");
    c->delay_us = e ? std::atol(e) : 0;
    c->next_stage = 0;
    c->async_err.store(0);
    for (int k = 0; k < kRing; ++k) {
        c->stage[k] = nullptr;
        if (!c->sync && hipHostMalloc((void**)&c->stage[k], kStageDoubles * sizeof(double), hipHostMallocDefault) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    c->shm->ready.fetch_add(1);
    if (!poll([&] { return c->shm->ready.load() >= nranks; })) return ncclSystemError;        // collective, like the real one
    *out = c;
    return ncclSuccess;
}

static ncclResult_t release(Comm* c) {
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();                         // (no host function of ours is left on any stream)
    for (int k = 0; k < kRing; ++k)
        if (c->stage[k]) (void)hipHostFree(c->stage[k]);
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
