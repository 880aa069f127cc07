// Process-wide gate between graph captures and the runtime calls that may not overlap one.
//
// A stream capture of this library holds kernel launches only, but on ROCm 7.x it was still invalidated now and then
// ("operation failed due to a previous error during capture") when ANOTHER host thread of the process -- a sibling rank
// of a bn_group uploading its slice of the recording -- allocated, freed or synchronously copied device memory while
// the capture was open, whatever the capture mode.  So the two are made mutually exclusive: a capture takes the gate
// exclusively (which also serialises captures), every allocation / free / synchronous copy / memset / stream or event
// creation of the library takes it shared.  Kernel launches, asynchronous copies, graph replays and stream waits of
// other threads are ordinary stream work and stay ungated.  A caller's own HIP calls are outside the library's reach:
// include/birdnet_hip.h says what not to do during the first step of a new batch size.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <shared_mutex>

namespace bn {
void note_launch_device(int dev);  // kernels.hip
// every device selection of the library: the runtime call + the thread-local note the launchers read
inline hipError_t use_device(int dev) {
    note_launch_device(dev);
    return ::hipSetDevice(dev);
}
inline std::shared_mutex &capture_gate() {
    static std::shared_mutex m;
    return m;
}
namespace gated {
struct Shared {
    std::shared_lock<std::shared_mutex> lk{capture_gate()};
};
template <class T>
inline hipError_t Malloc(T **p, size_t n) { Shared g; return ::hipMalloc(reinterpret_cast<void **>(p), n); }
inline hipError_t Free(void *p) { Shared g; return ::hipFree(p); }
template <class T>
inline hipError_t HostMalloc(T **p, size_t n, unsigned flags) { Shared g; return ::hipHostMalloc(reinterpret_cast<void **>(p), n, flags); }
inline hipError_t HostFree(void *p) { Shared g; return ::hipHostFree(p); }
inline hipError_t Memcpy(void *dst, const void *src, size_t n, hipMemcpyKind k) { Shared g; return ::hipMemcpy(dst, src, n, k); }
inline hipError_t Memset(void *dst, int v, size_t n) { Shared g; return ::hipMemset(dst, v, n); }
inline hipError_t StreamCreateWithFlags(hipStream_t *s, unsigned flags) { Shared g; return ::hipStreamCreateWithFlags(s, flags); }
inline hipError_t StreamDestroy(hipStream_t s) { Shared g; return ::hipStreamDestroy(s); }
inline hipError_t EventCreate(hipEvent_t *e) { Shared g; return ::hipEventCreate(e); }
inline hipError_t EventCreateWithFlags(hipEvent_t *e, unsigned flags) { Shared g; return ::hipEventCreateWithFlags(e, flags); }
inline hipError_t EventDestroy(hipEvent_t e) { Shared g; return ::hipEventDestroy(e); }
inline hipError_t DeviceSynchronize() { Shared g; return ::hipDeviceSynchronize(); }
inline hipError_t GraphExecDestroy(hipGraphExec_t g_) { Shared g; return ::hipGraphExecDestroy(g_); }
}  // namespace gated
}  // namespace bn
