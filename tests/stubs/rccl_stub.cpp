// TEST INFRASTRUCTURE, not a collective library: a stand-in for librccl that tests/test_gpu_group_rccl_branch.py builds and hands
// to libbirdnet_hip.so through BN_RCCL_LIB, so that the RCCL branch of csrc/group.cpp (ncclCommInitAll, the grouped in-place
// ncclAllGather per slab, ncclCommDestroy -- the dlsym'd ABI, the dtype codes, the send-offset arithmetic, the stream ordering and
// the error paths) EXECUTES on a one-GPU box.  It proves nothing about xGMI.  The entry points follow the declarations of
// /opt/rocm/include/rccl/rccl.h (ncclResult_t = int, 0 = success; ncclDataType_t: ncclUint32 = 3, ncclFloat32 = 7);
// the all-gather is performed with device-to-device copies in the stream order the real call promises: rank r's receive
// buffer is written on rank r's stream, after an event recorded on the SENDER's stream when the group closes.
//
//   g++ -shared -fPIC -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stubs/rccl_stub.cpp -L/opt/rocm/lib -lamdhip64 -o <tmp>/librccl_stub.so
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {
struct World;
struct Comm {
    World *world;
    int rank, dev;
};
struct World {
    std::vector<Comm *> comms;
    int alive;
};
struct Pending {
    const void *send;
    void *recv;
    size_t count;
    int dtype;
    Comm *comm;
    hipStream_t stream;
};
std::mutex g_mu;
int g_depth = 0;
std::vector<Pending> g_pending;
// counters the test reads back (rccl_stub_counters)
uint64_t g_counters[8] = {0};  // 0 init_all, 1 allgather calls, 2 groups closed, 3 comm destroys, 4 in-place calls, 5 f32 calls, 6 u32 calls, 7 errors
int g_fail_next = 0;           // rccl_stub_fail_next(n): the n-th ncclAllGather from now returns ncclInternalError (3)

size_t dtype_size(int dt) {
    switch (dt) {
        case 0: case 1: return 1;          // int8 / uint8
        case 2: case 3: case 7: return 4;  // int32, uint32, float32
        case 4: case 5: case 8: return 8;  // int64, uint64, float64
        case 6: case 9: return 2;          // float16, bfloat16
        default: return 0;
    }
}

int run_group() {
    // every rank of a world must have posted exactly one call per collective, in rank order of arrival per world
    std::vector<Pending> calls;
    calls.swap(g_pending);
    size_t i = 0;
    while (i < calls.size()) {
        World *w = calls[i].comm->world;
        const size_t n = w->comms.size();
        if (i + n > calls.size()) return 5;  // ncclInvalidUsage: a rank is missing from the group
        std::vector<const Pending *> by_rank(n, nullptr);
        for (size_t j = 0; j < n; j++) {
            const Pending &p = calls[i + j];
            if (p.comm->world != w || by_rank[(size_t)p.comm->rank]) return 5;
            by_rank[(size_t)p.comm->rank] = &p;
        }
        const size_t bytes = by_rank[0]->count * dtype_size(by_rank[0]->dtype);
        for (size_t r = 0; r < n; r++)
            if (by_rank[r]->count != by_rank[0]->count || by_rank[r]->dtype != by_rank[0]->dtype) return 4;  // ncclInvalidArgument
        // the sender's data is ready in ITS stream's order: one event per sender, every receiving stream waits for it
        std::vector<hipEvent_t> ready(n);
        for (size_t s = 0; s < n; s++) {
            if (hipSetDevice(by_rank[s]->comm->dev) != hipSuccess) return 1;
            if (hipEventCreateWithFlags(&ready[s], hipEventDisableTiming) != hipSuccess) return 1;
            if (hipEventRecord(ready[s], by_rank[s]->stream) != hipSuccess) return 1;
        }
        for (size_t d = 0; d < n; d++) {
            if (hipSetDevice(by_rank[d]->comm->dev) != hipSuccess) return 1;
            for (size_t s = 0; s < n; s++) {
                char *dst = static_cast<char *>(by_rank[d]->recv) + s * bytes;
                if (s == d && dst == by_rank[s]->send) continue;  // in place: the rank's own slab is already where it belongs
                if (s != d && hipStreamWaitEvent(by_rank[d]->stream, ready[s], 0) != hipSuccess) return 1;
                if (hipMemcpyAsync(dst, by_rank[s]->send, bytes, hipMemcpyDeviceToDevice, by_rank[d]->stream) != hipSuccess) return 1;
            }
        }
        for (size_t s = 0; s < n; s++) (void)hipEventDestroy(ready[s]);  // (destruction is deferred by the runtime until the event has completed)
        i += n;
    }
    g_counters[2]++;
    return 0;
}
}  // namespace

extern "C" {

int ncclCommInitAll(void **comms, int ndev, const int *devlist) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!comms || ndev <= 0) return 4;
    World *w = new World();
    w->alive = ndev;
    for (int r = 0; r < ndev; r++) {
        Comm *c = new Comm{w, r, devlist ? devlist[r] : r};
        w->comms.push_back(c);
        comms[r] = c;
    }
    g_counters[0]++;
    return 0;
}

int ncclCommDestroy(void *comm) {
    std::lock_guard<std::mutex> lk(g_mu);
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 4;
    World *w = c->world;
    g_counters[3]++;
    if (--w->alive == 0) {
        for (Comm *x : w->comms) delete x;
        delete w;
    }
    return 0;
}

int ncclGroupStart() {
    std::lock_guard<std::mutex> lk(g_mu);
    g_depth++;
    return 0;
}

int ncclGroupEnd() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    int prev = -1;
    (void)hipGetDevice(&prev);
    const int res = run_group();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (res) g_counters[7]++;
    return res;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    Comm *c = static_cast<Comm *>(comm);
    if (!c || !send || !recv || dtype_size(dtype) == 0) {
        g_counters[7]++;
        return 4;
    }
    if (g_fail_next > 0 && --g_fail_next == 0) {
        g_counters[7]++;
        return 3;  // ncclInternalError
    }
    g_counters[1]++;
    g_counters[dtype == 7 ? 5 : dtype == 3 ? 6 : 7]++;
    // rccl.h: "In-place operation will happen if sendbuff == recvbuff + rank * sendcount" (in elements)
    if (send == static_cast<const char *>(recv) + (size_t)c->rank * count * dtype_size(dtype)) g_counters[4]++;
    g_pending.push_back(Pending{send, recv, count, dtype, c, stream});
    if (g_depth == 0) {  // an ungrouped call from one thread per rank would block in the real library; not what group.cpp does
        g_pending.pop_back();
        g_counters[7]++;
        return 5;
    }
    return 0;
}

const char *ncclGetErrorString(int r) {
    switch (r) {
        case 0: return "no error";
        case 1: return "unhandled cuda error (stub: a HIP call failed)";
        case 3: return "internal error (stub: injected)";
        case 4: return "invalid argument";
        case 5: return "invalid usage";
        default: return "unknown result code";
    }
}

// ---- test hooks (not part of rccl.h)
void rccl_stub_counters(uint64_t *out8) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < 8; i++) out8[i] = g_counters[i];
}
void rccl_stub_fail_next(int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_fail_next = n;
}
}
