// Test double of librccl for the engine group's result gather (ocrs_amd/csrc/group.cpp binds librccl with dlopen;
// OCRS_RCCL_LIB names this library instead).  TEST INFRASTRUCTURE, never shipped or measured.
//
// Why: a one-GPU box can only build a one-rank RCCL communicator, and RCCL refuses the same device twice, so the
// G > 1 logic of gather_rccl (slot offsets, length prefixes, per-member streams and syncs, several gathers in flight)
// could never run there.  This double implements exactly the six entry points group.cpp binds, with the semantics of
// the real calls, on top of HIP copies — and accepts a communicator with repeated devices (the marker symbol
// ocrs_rccl_stub_accepts_duplicate_devices tells group.cpp so):
//   ncclCommInitAll   one "world" of ndev ranks, rank r on devlist[r]
//   ncclGroupStart / ncclGroupEnd   calls between them are queued per host thread and issued together at the end
//   ncclAllGather     rank r's recv buffer receives every rank's send buffer at offset rank * count, on rank r's stream,
//                     after the producing rank's stream has reached the call; a send buffer may be reused on its own
//                     stream once every rank has read it (stream-ordered, like the real collective)
// Failure injection: OCRS_RCCL_STUB_FAIL_INIT=1 makes ncclCommInitAll fail; OCRS_RCCL_STUB_FAIL_GATHER=n makes the
// n-th ncclAllGather call of the process fail.  Counters: ocrs_rccl_stub_stats().
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
}

namespace {
struct World {
    int n = 0;
    std::vector<int> devices;
};
struct Op {
    const void* send; void* recv; size_t bytes; ncclComm* comm; hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
std::atomic<uint64_t> g_inits{0}, g_gathers{0}, g_groups{0}, g_max_ranks{0}, g_bytes{0};
}  // namespace

struct ncclComm {
    std::shared_ptr<World> world;
    int rank = 0, device = 0;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(d);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Issues one set of all-gather calls (one per rank of one world).
ncclResult_t issue(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    World* w = ops[0].comm->world.get();
    const int n = w->n;
    std::vector<const Op*> by_rank(n, nullptr);
    for (const Op& op : ops) {
        if (op.comm->world.get() != w || op.bytes != ops[0].bytes) return ncclInvalidUsage;
        if (by_rank[op.comm->rank]) return ncclInvalidUsage;
        by_rank[op.comm->rank] = &op;
    }
    for (int r = 0; r < n; r++)
        if (!by_rank[r]) return ncclInvalidUsage;   // a collective needs every rank
    std::vector<hipEvent_t> produced(n, nullptr), consumed(n, nullptr);
    ncclResult_t rc = ncclSuccess;
    for (int r = 0; r < n && rc == ncclSuccess; r++) {   // "rank r's stream has reached the call"
        DeviceGuard g(w->devices[r]);
        if (hipEventCreateWithFlags(&produced[r], hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(produced[r], by_rank[r]->stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
    }
    for (int r = 0; r < n && rc == ncclSuccess; r++) {
        DeviceGuard g(w->devices[r]);
        const Op& me = *by_rank[r];
        for (int s = 0; s < n && rc == ncclSuccess; s++) {
            if (s != r && hipStreamWaitEvent(me.stream, produced[s], 0) != hipSuccess) rc = ncclUnhandledCudaError;
            if (rc == ncclSuccess &&
                hipMemcpyAsync(static_cast<char*>(me.recv) + (size_t)s * me.bytes, by_rank[s]->send, me.bytes, hipMemcpyDefault, me.stream) !=
                    hipSuccess)
                rc = ncclUnhandledCudaError;
        }
        if (rc == ncclSuccess && (hipEventCreateWithFlags(&consumed[r], hipEventDisableTiming) != hipSuccess ||
                                  hipEventRecord(consumed[r], me.stream) != hipSuccess))
            rc = ncclUnhandledCudaError;
    }
    for (int s = 0; s < n && rc == ncclSuccess; s++) {   // a send buffer is free on its stream once every rank has read it
        DeviceGuard g(w->devices[s]);
        for (int r = 0; r < n; r++)
            if (r != s && hipStreamWaitEvent(by_rank[s]->stream, consumed[r], 0) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    // events may be destroyed while work that references them is pending (HIP defers the release)
    for (int r = 0; r < n; r++) {
        if (produced[r]) (void)hipEventDestroy(produced[r]);
        if (consumed[r]) (void)hipEventDestroy(consumed[r]);
    }
    if (rc != ncclSuccess) (void)hipGetLastError();
    uint64_t prev = g_max_ranks.load();
    while ((uint64_t)n > prev && !g_max_ranks.compare_exchange_weak(prev, (uint64_t)n)) {
    }
    g_bytes += (uint64_t)n * n * ops[0].bytes;
    return rc;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int ocrs_rccl_stub_accepts_duplicate_devices = 1;

// out[5] = {communicator sets created, all-gather calls, group sections, largest world, bytes moved}
__attribute__((visibility("default"))) void ocrs_rccl_stub_stats(uint64_t out[5]) {
    out[0] = g_inits.load(); out[1] = g_gathers.load(); out[2] = g_groups.load(); out[3] = g_max_ranks.load(); out[4] = g_bytes.load();
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist) {
    if (!comm || ndev <= 0) return ncclInvalidArgument;
    const char* f = getenv("OCRS_RCCL_STUB_FAIL_INIT");
    if (f && *f && *f != '0') return ncclSystemError;
    auto w = std::make_shared<World>();
    w->n = ndev;
    for (int r = 0; r < ndev; r++) w->devices.push_back(devlist ? devlist[r] : r);
    for (int r = 0; r < ndev; r++) {
        comm[r] = new ncclComm();
        comm[r]->world = w;
        comm[r]->rank = r;
        comm[r]->device = w->devices[r];
    }
    g_inits++;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled cuda error (stub)";
        case ncclSystemError: return "unhandled system error (stub)";
        case ncclInternalError: return "internal error (stub)";
        case ncclInvalidArgument: return "invalid argument (stub)";
        case ncclInvalidUsage: return "invalid usage (stub)";
    }
    return "unknown result code (stub)";
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() {
    t_depth++;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    g_groups++;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return issue(ops);
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount,
                                                                  ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    if (!sendbuff || !recvbuff || !comm) return ncclInvalidArgument;
    if (datatype != ncclUint8 && datatype != ncclInt8) return ncclInvalidArgument;   // all group.cpp sends
    const uint64_t k = ++g_gathers;
    const char* f = getenv("OCRS_RCCL_STUB_FAIL_GATHER");
    if (f && *f && (uint64_t)atoll(f) == k) return ncclInternalError;
    Op op{sendbuff, recvbuff, sendcount, comm, stream};
    if (t_depth > 0) {
        t_ops.push_back(op);
        return ncclSuccess;
    }
    std::vector<Op> ops{op};
    return issue(ops);   // outside a group only a one-rank world can complete
}

}  // extern "C"
