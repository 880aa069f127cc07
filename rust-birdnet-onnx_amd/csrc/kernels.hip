// Hand-written gfx950 (CDNA4, wave64) kernels of the inference plan.
//
//  gemm_mfma_kernel   pointwise conv / conv1d framing / MatMul / FC on the matrix
//                     cores with exact-f32 MFMA (v_mfma_f32_32x32x2_f32), LDS
//                     staged, fused bias + activation + residual epilogue
//  dwconv_kernel      NHWC depthwise KxK, 4 channels per lane (float4), fused
//                     bias + activation
//  conv_direct_kernel NHWC direct convolution (stem conv, grouped conv fallback)
//  reduce_*_kernel    strided reductions (global pooling, per-segment min/max)
//  elt_*_kernel       strided elementwise / broadcast / copy
//
// All tensors are f32, the batch is the outermost dimension.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include <atomic>

#include "kernels.h"
#include "plan_rules.h"
#include "hip_gate.h"
#include "bf16x3.h"
#include "device_common.h"

namespace bn {

namespace {
thread_local std::string g_launch_error;
thread_local bool g_launch_failed = false;
thread_local std::string g_launch_error_out;
}  // namespace
void launch_error(const char *msg) {
    if (!g_launch_failed) {
        g_launch_failed = true;
        g_launch_error = msg;
    }
}
const char *take_launch_error() {
    if (!g_launch_failed) return nullptr;
    g_launch_failed = false;
    g_launch_error_out = g_launch_error;
    return g_launch_error_out.c_str();
}
// ---- per-device preparation -----------------------------------------------------------------------------------
// Everything a launcher would otherwise have to ask the runtime for on its first launch -- the opt-in to more than
// 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize, per function AND device) and the device's CU
// count -- is done ONCE per device by prepare_device(), which bn_model_load / bn_ctx_create / the stand-alone top-K
// entry points call before anything is launched.  Nothing but kernel launches is then issued between
// hipStreamBeginCapture and hipStreamEndCapture: a hipFuncSetAttribute inside a capture (the first launch of the
// 80 KB frame_fold_kernel of a plan) is what invalidated one graph capture in six when several ranks shared a device.
namespace {
std::mutex g_prep_mu;
std::atomic<uint64_t> g_prepared_mask{0};  // device ordinals < 64 that prepare_device() has completed on
std::atomic<int> g_cu_count[64];
std::atomic<float *> g_zero_page[64];
std::vector<const void *> &dyn_lds_kernels() {
    static std::vector<const void *> v;
    return v;
}
}  // namespace
void register_dynamic_lds_kernel(const void *kernel) { dyn_lds_kernels().push_back(kernel); }
void register_kernels_hip();   // kernels.hip (below)
void register_stft_kernels();  // stft.hip
void register_topk_kernels();  // topk.hip
void register_gemm_dma_kernels();  // gemm_dma.hip
void register_mbmap_kernels();     // mbmap.hip
void register_mbmap_ws_kernels();  // mbmap_ws.hip
void register_gemm_dma3_kernels();  // gemm_dma3.hip

bool prepare_device(int dev) {
    if (dev < 0 || dev >= 64) return false;
    if (g_prepared_mask.load(std::memory_order_acquire) & (1ull << dev)) return true;
    std::lock_guard<std::mutex> lk(g_prep_mu);
    if (g_prepared_mask.load(std::memory_order_acquire) & (1ull << dev)) return true;
    if (dyn_lds_kernels().empty()) {
        register_kernels_hip();
        register_stft_kernels();
        register_topk_kernels();
        register_gemm_dma_kernels();
        register_mbmap_kernels();
        register_mbmap_ws_kernels();
        register_gemm_dma3_kernels();
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return false;
    if (cur != dev && hipSetDevice(dev) != hipSuccess) return false;
    bool ok = true;
    for (const void *k : dyn_lds_kernels()) {
        // the limit is on static + dynamic LDS together: a kernel with static __shared__ arrays gets the remainder
        hipFuncAttributes fa{};
        size_t stat = 0;
        if (hipFuncGetAttributes(&fa, k) == hipSuccess) stat = fa.sharedSizeBytes;
        else (void)hipGetLastError();
        const int dyn = (int)(160 * 1024 - std::min<size_t>(stat, 96 * 1024));
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
        }
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    g_cu_count[dev].store(v, std::memory_order_relaxed);
    {  // a page of zeros: the source of LDS-DMA lanes whose chunk lies in a row's padding (mbmap.hip)
        float *z = nullptr;
        // only the opt-in padded-k small-map kernels read it (launch_mbmap refuses them on a null page): a failure here must not
        // take the device away from every other plan
        if (gated::Malloc(&z, 4096) != hipSuccess || gated::Memset(z, 0, 4096) != hipSuccess || gated::DeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            if (z) (void)gated::Free(z);
            z = nullptr;
        }
        g_zero_page[dev].store(z, std::memory_order_relaxed);
    }
    if (cur != dev && cur >= 0) (void)bn::use_device(cur);  // (restores the thread's launch-device note with it)
    if (ok) g_prepared_mask.fetch_or(1ull << dev, std::memory_order_release);
    return ok;
}
// the launch-time side: no runtime call at all.  A launcher that needs the opt-in on a device nobody prepared
// (a programming error in the C ABI layer) refuses the launch instead of asking the runtime mid-capture.
static thread_local int t_launch_dev = -1;
void note_launch_device(int dev) { t_launch_dev = dev; }
bool ensure_dynamic_lds(const void *kernel, size_t bytes) {
    (void)kernel;
    if (bytes <= 64 * 1024) return true;  // within the default limit
    if (bytes > 160 * 1024) return false;
    const uint64_t m = g_prepared_mask.load(std::memory_order_acquire);
    const int dev = t_launch_dev;
    // a thread that never went through the C ABI's device selection (the stand-alone probes under tools/) is answered for
    // "some prepared device"; everything the library launches itself has noted its device
    return dev >= 0 && dev < 64 ? (m >> dev) & 1ull : m != 0;
}
const float *device_zero_page() {
    const uint64_t m = g_prepared_mask.load(std::memory_order_acquire);
    const int dev = t_launch_dev;
    if (dev >= 0 && dev < 64 && ((m >> dev) & 1ull)) return g_zero_page[dev].load(std::memory_order_relaxed);
    for (int d = 0; d < 64; d++)
        if (m & (1ull << d)) return g_zero_page[d].load(std::memory_order_relaxed);
    return nullptr;
}
// live inference contexts per device (capi.cpp counts them): kernels whose grid is a trade between one launch's latency and the work per
// block (mbmap.hip's chunks per block) size it for a device that is SHARED when several contexts keep batches in flight
static std::atomic<int> g_ctx_count[64];
void device_context_count_add(int dev, int delta) {
    if (dev >= 0 && dev < 64) g_ctx_count[dev].fetch_add(delta, std::memory_order_relaxed);
}
static std::atomic<int> g_sharing_mode{0};  // bn_set_sharing_mode: 0 = alone (default), 1 = shared, -1 = by the count of live contexts
void device_sharing_mode(int mode) { g_sharing_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
int device_context_count() {
    const int dev = t_launch_dev;
    int n = (dev >= 0 && dev < 64) ? g_ctx_count[dev].load(std::memory_order_relaxed) : 1;
    n = n < 1 ? 1 : n;
    const int mode = g_sharing_mode.load(std::memory_order_relaxed);
    return mode == 0 ? 1 : (mode == 1 && n < 2 ? 2 : n);
}
int device_cu_count() {
    const uint64_t m = g_prepared_mask.load(std::memory_order_acquire);
    const int dev = t_launch_dev;
    if (dev >= 0 && dev < 64 && ((m >> dev) & 1ull)) return g_cu_count[dev].load(std::memory_order_relaxed);
    for (int d = 0; d < 64; d++)
        if (m & (1ull << d)) return g_cu_count[d].load(std::memory_order_relaxed);
    return 256;
}

namespace {

// ------------------------------------------------------------------ elementwise chains
struct EltPtrs {
    const float *b[ELT_MAX_STAGES];
};

// Unsigned division by a launch-invariant divisor d (1 <= d < 2^31) of n < 2^31:
//   s = ceil(log2 d), m = ceil(2^(31+s) / d)  ->  n / d == umulhi(n, m) >> (s - 1)      (d >= 2)
// (m*d - 2^(31+s) < d <= 2^s and n < 2^31, so the error term never reaches the next quotient.)
struct EltDiv {
    uint32_t mul[ELT_MAX_DIMS], shift[ELT_MAX_DIMS];  // shift = s - 1; mul == 0 marks d == 1
};
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t mul, uint32_t shift) {
    return mul ? (__umulhi(n, mul) >> shift) : n;
}

// generic strided form: 4 elements per thread, flat index -> multi-index with multiply-shift
// divisions (no hardware integer division: the per-element div/mod chains made this kernel
// VALU-bound).  grid: (ceil(per_sample / 1024) capped, batch)
__global__ __launch_bounds__(256) void elt_strided_kernel(EltDesc d, EltDiv dv, float *__restrict__ out, const float *__restrict__ a, EltPtrs bp) {
    const int64_t bidx = blockIdx.y;
    const uint32_t per = (uint32_t)d.per_sample;
    float *o = out + bidx * d.bo;
    const float *pa = a + bidx * d.ba;
    for (uint32_t i0 = blockIdx.x * 1024u + threadIdx.x; i0 < per; i0 += gridDim.x * 1024u) {
        int64_t oo[4], oa[4], ob[ELT_MAX_STAGES][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * 256u;
            uint32_t rem = i < per ? i : per - 1;
            oo[u] = 0; oa[u] = 0;
#pragma unroll
            for (int s = 0; s < ELT_MAX_STAGES; s++) ob[s][u] = 0;
#pragma unroll
            for (int k = ELT_MAX_DIMS - 1; k >= 0; k--) {
                if (k < d.nd) {
                    const uint32_t q = fast_div(rem, dv.mul[k], dv.shift[k]);
                    const uint32_t ix = rem - q * (uint32_t)d.size[k];
                    rem = q;
                    oo[u] += (int64_t)ix * d.so[k];
                    oa[u] += (int64_t)ix * d.sa[k];
#pragma unroll
                    for (int s = 0; s < ELT_MAX_STAGES; s++)
                        if (s < d.nstages) ob[s][u] += (int64_t)ix * d.st[s].sb[k];
                }
            }
        }
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = pa[oa[u]];
#pragma unroll
        for (int s = 0; s < ELT_MAX_STAGES; s++) {
            if (s < d.nstages) {
                const EltStage &st = d.st[s];
                if (st.bin != BIN_NONE) {
                    const float *pb = bp.b[s] + bidx * st.bb;
                    float w[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) w[u] = pb[ob[s][u]];
                    bin_array<4>(st.bin, st.bsq, v, w);
                }
                act_array_all<4>(st.act, st.p0, st.p1, v);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + u * 256u < per) o[oo[u]] = v[u];
    }
}

// out and a dense over the whole index space (any nd, collapsed to a flat range); every stage
// operand is dense like them (mode 1), one scalar per sample (mode 0), or periodic (mode 2: it
// varies only over the trailing dims, e.g. a per-channel vector under a channels-last tensor).
struct EltFlat {
    int32_t a_scalar;  // the primary operand is one value per sample (fills)
    int32_t mode[ELT_MAX_STAGES];
    uint32_t period[ELT_MAX_STAGES];  // mode 2: elements; 1, 2 or a multiple of 4
};
__global__ __launch_bounds__(256) void elt_flat4_kernel(EltDesc d, EltFlat f, float *__restrict__ out, const float *__restrict__ a, EltPtrs bp) {
    const int64_t bidx = blockIdx.y;
    const uint32_t per4 = (uint32_t)(d.per_sample >> 2);
    float4 *o = reinterpret_cast<float4 *>(out + bidx * d.bo);
    const float4 *pa = reinterpret_cast<const float4 *>(a + bidx * d.ba);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < per4; i += gridDim.x * 256u) {
        float v[4];
        if (f.a_scalar) {
            v[0] = v[1] = v[2] = v[3] = a[bidx * d.ba];
        } else {
            const float4 v4 = pa[i];
            v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        }
#pragma unroll
        for (int s = 0; s < ELT_MAX_STAGES; s++) {
            if (s < d.nstages) {
                const EltStage &st = d.st[s];
                if (st.bin != BIN_NONE) {
                    const float *pb = bp.b[s] + bidx * st.bb;
                    float w[4];
                    if (f.mode[s] == 1) {
                        const float4 w4 = reinterpret_cast<const float4 *>(pb)[i];
                        w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                    } else if (f.mode[s] == 0 || f.period[s] == 1) {
                        w[0] = w[1] = w[2] = w[3] = pb[0];
                    } else if (f.period[s] == 2) {
                        w[0] = w[2] = pb[0];
                        w[1] = w[3] = pb[1];
                    } else {
                        const float4 w4 = reinterpret_cast<const float4 *>(pb)[i % (f.period[s] >> 2)];
                        w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                    }
                    bin_array<4>(st.bin, st.bsq, v, w);
                }
                act_array_all<4>(st.act, st.p0, st.p1, v);
            }
        }
        o[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ float red_init(int op) {
    switch (op) {
        case RED_MAX: return -INFINITY;
        case RED_MIN: return INFINITY;
        case RED_PROD: return 1.0f;
        default: return 0.0f;
    }
}
__device__ __forceinline__ float red_step(int op, float acc, float x) {
    switch (op) {
        case RED_MAX: return fmaxf(acc, x);
        case RED_MIN: return fminf(acc, x);
        case RED_PROD: return acc * x;
        case RED_L2:
        case RED_SUMSQ: return acc + x * x;
        default: return acc + x;
    }
}
__device__ __forceinline__ float red_merge(int op, float a, float b) {
    switch (op) {
        case RED_MAX: return fmaxf(a, b);
        case RED_MIN: return fminf(a, b);
        case RED_PROD: return a * b;
        default: return a + b;
    }
}
__device__ __forceinline__ float red_finish(int op, float acc, int64_t n) {
    if (op == RED_MEAN) return acc / (float)n;
    if (op == RED_L2) return sqrtf(acc);
    return acc;
}
__device__ __forceinline__ int64_t red_offset(const ReduceDesc &d, uint32_t r) {
    int64_t off = 0;
#pragma unroll
    for (int k = 2; k >= 0; k--) {
        if (k < d.nr) {
            const uint32_t sz = (uint32_t)d.rsize[k];
            off += (int64_t)(r % sz) * d.rin[k];
            r /= sz;
        }
    }
    return off;
}
__device__ __forceinline__ void kept_offsets(const ReduceDesc &d, uint32_t kidx, int64_t &in_off,
                                             int64_t &out_off) {
    in_off = 0;
    out_off = 0;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        if (k < d.nk) {
            const uint32_t sz = (uint32_t)d.ksize[k];
            const uint32_t idx = kidx % sz;
            kidx /= sz;
            in_off += (int64_t)idx * d.kin[k];
            out_off += (int64_t)idx * d.kout[k];
        }
    }
}

// kept index on threadIdx.x (coalesced along the innermost kept dim), the reduced
// range split over threadIdx.y.  block (64, 16); grid (ceil(kept/64), batch)
__global__ __launch_bounds__(1024) void reduce_inner_kept_kernel(ReduceDesc d, float *__restrict__ out,
                                                                 const float *__restrict__ in) {
    __shared__ float part[16][64];
    const int64_t bidx = blockIdx.y;
    const uint32_t kidx = blockIdx.x * 64u + threadIdx.x;
    const bool valid = kidx < (uint32_t)d.kept;
    float acc = red_init(d.op);
    int64_t in_off = 0, out_off = 0;
    if (valid) {
        kept_offsets(d, kidx, in_off, out_off);
        const float *p = in + bidx * d.bi + in_off;
        for (uint32_t r = threadIdx.y; r < (uint32_t)d.red; r += 16u) acc = red_step(d.op, acc, p[red_offset(d, r)]);
    }
    part[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && valid) {
        for (int y = 1; y < 16; y++) acc = red_merge(d.op, acc, part[y][threadIdx.x]);
        out[bidx * d.bo + out_off] = red_finish(d.op, acc, d.red);
    }
}

// one block per (kept index, sample): threads stride over the reduced range (float4 when the
// reduced dim is contiguous and aligned).  block = blockDim.x (256 or 1024); grid (kept, batch)
__global__ __launch_bounds__(1024) void reduce_row_kernel(ReduceDesc d, float *__restrict__ out,
                                                          const float *__restrict__ in, int vec4) {
    __shared__ float part[16];
    const int64_t bidx = blockIdx.y;
    const uint32_t nt = blockDim.x;
    int64_t in_off, out_off;
    kept_offsets(d, blockIdx.x, in_off, out_off);
    const float *p = in + bidx * d.bi + in_off;
    float acc = red_init(d.op);
    if (vec4) {
        const float4 *p4 = reinterpret_cast<const float4 *>(p);
        const uint32_t n4 = (uint32_t)(d.red >> 2);
        // four loads in flight per lane (independent accumulators); exact for min/max, fixed order for sums
        float a1 = red_init(d.op), a2 = a1, a3 = a1;
        uint32_t r = threadIdx.x;
        for (; r + 3 * nt < n4; r += 4 * nt) {
            const float4 v0 = p4[r], v1 = p4[r + nt], v2 = p4[r + 2 * nt], v3 = p4[r + 3 * nt];
            acc = red_step(d.op, red_step(d.op, red_step(d.op, red_step(d.op, acc, v0.x), v0.y), v0.z), v0.w);
            a1 = red_step(d.op, red_step(d.op, red_step(d.op, red_step(d.op, a1, v1.x), v1.y), v1.z), v1.w);
            a2 = red_step(d.op, red_step(d.op, red_step(d.op, red_step(d.op, a2, v2.x), v2.y), v2.z), v2.w);
            a3 = red_step(d.op, red_step(d.op, red_step(d.op, red_step(d.op, a3, v3.x), v3.y), v3.z), v3.w);
        }
        for (; r < n4; r += nt) {
            const float4 v = p4[r];
            acc = red_step(d.op, red_step(d.op, red_step(d.op, red_step(d.op, acc, v.x), v.y), v.z), v.w);
        }
        acc = red_merge(d.op, red_merge(d.op, acc, a1), red_merge(d.op, a2, a3));
        for (uint32_t t = (n4 << 2) + threadIdx.x; t < (uint32_t)d.red; t += nt) acc = red_step(d.op, acc, p[t]);
    } else if (d.nr == 1) {
        const int64_t st = d.rin[0];
        for (uint32_t r = threadIdx.x; r < (uint32_t)d.red; r += nt) acc = red_step(d.op, acc, p[(int64_t)r * st]);
    } else {
        for (uint32_t r = threadIdx.x; r < (uint32_t)d.red; r += nt) acc = red_step(d.op, acc, p[red_offset(d, r)]);
    }
    for (int off = 32; off > 0; off >>= 1) acc = red_merge(d.op, acc, __shfl_down(acc, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < (nt >> 6); w++) acc = red_merge(d.op, acc, part[w]);
        out[bidx * d.bo + out_off] = red_finish(d.op, acc, d.red);
    }
}

// ------------------------------------------------------------------ MFMA GEMM
// Block tile 128 x BN, K step 32, 256 threads = 4 waves stacked along M.  Each
// wave owns a 32 x BN strip: BN/32 accumulators of v_mfma_f32_32x32x2_f32.
// LDS rows are padded to 36 floats so the b128 fragment reads are conflict free.
//
// Fragment trick: one ds_read_b128 per operand feeds four MFMAs.  The MFMA sums
// over two k-slots (lane>>5); slot h of MFMA j carries k = 8g + 4h + j, the same
// assignment for A and B, so every k of the 8-wide group is used exactly once.
#ifndef GEMM_BK_VALUE
#define GEMM_BK_VALUE 32
#endif
constexpr int GEMM_BM = 128, GEMM_BK = GEMM_BK_VALUE, GEMM_LD = GEMM_BK + 4;
static_assert(GEMM_BM == 128 && GEMM_BK == 32, "gemm_use_splitk (kernels.h) restates the tile sizes");
#ifndef GEMM_PF
#define GEMM_PF 1
#endif

// Row addressing is resolved once per thread before the K loop: aoff[i] is the element offset of
// the i-th row this thread stages (or -1 past the end), soff[i] the offset of its sample's gate.
__device__ __forceinline__ void row_split(const GemmDesc &d, int64_t r, int64_t &b, int64_t &m) {
    // rows and row counts fit 32 bits for every supported batch; 32-bit division is ~5x cheaper
    const uint32_t bb = (uint32_t)r / (uint32_t)d.rows;
    b = bb;
    m = r - (int64_t)bb * d.rows;
}

// Every staging load is UNCONDITIONAL and UNMASKED: out-of-range rows / columns read a clamped
// (valid) address and whatever lands there is simply never used -- rows past the end and weight
// rows past N only feed output elements the epilogue does not store, and columns past K are
// excluded by the K-tail step (mfma_ktile_partial multiplies only the 8-wide groups that hold
// data; a ragged K % 8 is zeroed explicitly in that one step).  Guarding or masking a load makes
// hipcc wait for it on the spot, which serialises the memory latency of the whole tile.
template <int AVEC, int ROWS_PER_PASS, int ITERS>
__device__ __forceinline__ void a_offsets(const GemmDesc &d, int64_t total_rows, int64_t row0, int row_first,
                                          int64_t (&aoff)[ITERS], int64_t (&soff)[ITERS]) {
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        int64_t r = row0 + row_first + i * ROWS_PER_PASS;
        r = r < total_rows ? r : total_rows - 1;
        int64_t b, m;
        row_split(d, r, b, m);
        aoff[i] = b * d.a_bs + m * d.lda;
        soff[i] = b * d.s_bs;
    }
}

// Squeeze-excite gate values for the same elements load_a_regs fetches.  They are multiplied in when
// the tile is written to LDS: multiplying right after the load would put an s_waitcnt behind every
// single load.
template <int AVEC, int ITERS>
__device__ __forceinline__ void load_gate_regs(const GemmDesc &d, const float *__restrict__ scale, const int64_t (&soff)[ITERS], int k,
                                               float (&regs)[ITERS * AVEC]) {
    const int kc = k < d.K ? k : 0;
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const float *sp = scale + soff[i] + kc;
        if constexpr (AVEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(sp);
            regs[i * 4] = t.x; regs[i * 4 + 1] = t.y; regs[i * 4 + 2] = t.z; regs[i * 4 + 3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < AVEC; j++) regs[i * AVEC + j] = sp[j];
        }
    }
}

template <int AVEC, int ITERS>
__device__ __forceinline__ void load_a_regs(const GemmDesc &d, const float *__restrict__ A, const int64_t (&aoff)[ITERS], int k,
                                            float (&regs)[ITERS * AVEC]) {
    const int kc = k < d.K ? k : 0;  // K % AVEC == 0 for AVEC > 1, so a vector never straddles K
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const float *p = A + aoff[i] + kc;
        if constexpr (AVEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p);
            regs[i * 4] = t.x; regs[i * 4 + 1] = t.y; regs[i * 4 + 2] = t.z; regs[i * 4 + 3] = t.w;
        } else if constexpr (AVEC == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(p);
            regs[i * 2] = t.x; regs[i * 2 + 1] = t.y;
        } else {
            regs[i] = *p;
        }
    }
}

// Folded framing rows (GemmDesc::fold): column c of the folded operand is x[1 + c] + sign * x[fold_n - 1 - c] of
// the frame.  One of the two pairs always starts at an odd element: both are loaded as 4-byte aligned dwordx2
// (the hardware's unaligned mode handles it) and combined when the tile is written to LDS.
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));
template <int ITERS>
__device__ __forceinline__ void load_a_fold_regs(const GemmDesc &d, const float *__restrict__ A, const int64_t (&aoff)[ITERS], int k,
                                                 float (&fwd)[ITERS * 2], float (&rev)[ITERS * 2]) {
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const float *p = A + aoff[i];
        const float2u f = *reinterpret_cast<const float2u *>(p + 1 + k);
        const float2u r = *reinterpret_cast<const float2u *>(p + (d.fold_n - 2 - k));
        fwd[i * 2] = f.x; fwd[i * 2 + 1] = f.y;
        rev[i * 2] = r.y; rev[i * 2 + 1] = r.x;  // mirrored order
    }
}

template <int WVEC, int ROWS_PER_PASS, int ITERS>
__device__ __forceinline__ void load_w_regs(const GemmDesc &d, const float *__restrict__ W, int n_first, int k, float (&regs)[ITERS * WVEC]) {
    const int kc = k < d.K ? k : 0;
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int n = n_first + i * ROWS_PER_PASS;
        const float *p = W + (int64_t)(n < d.N ? n : d.N - 1) * d.K + kc;
        if constexpr (WVEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p);
            regs[i * 4] = t.x; regs[i * 4 + 1] = t.y; regs[i * 4 + 2] = t.z; regs[i * 4 + 3] = t.w;
        } else {
            regs[i] = *p;
        }
    }
}

// zero the staged elements whose column index is >= K (ragged K tail only)
template <int VEC, int N>
__device__ __forceinline__ void zero_past_k(int k_col, int K, float (&regs)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++)
        if (k_col + (i % VEC) >= K) regs[i] = 0.0f;
}

template <int VEC, int ROWS_PER_PASS, int ITERS>
__device__ __forceinline__ void store_tile_regs(float *__restrict__ T, int row_first, int col, const float (&regs)[ITERS * VEC]) {
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        float *p = T + (row_first + i * ROWS_PER_PASS) * GEMM_LD + col;
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(p) = make_float4(regs[i * 4], regs[i * 4 + 1], regs[i * 4 + 2], regs[i * 4 + 3]);
        else if constexpr (VEC == 2) *reinterpret_cast<float2 *>(p) = make_float2(regs[i * 2], regs[i * 2 + 1]);
        else *p = regs[i];
    }
}

// ng = number of 8-wide k groups of this K step that hold real data, nt = number of 32-wide N
// tiles of this block that exist; padded groups / tiles are skipped (both are wave-uniform).
template <int NT>
__device__ __forceinline__ void mfma_ktile_full(const float *__restrict__ ap, const float *__restrict__ wp, floatx16 (&acc)[NT]) {
#pragma unroll
    for (int g = 0; g < GEMM_BK / 8; g++) {
        const float4 a4 = *reinterpret_cast<const float4 *>(ap + 8 * g);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float4 b4 = *reinterpret_cast<const float4 *>(wp + t * 32 * GEMM_LD + 8 * g);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[t], 0, 0, 0);
        }
    }
}

// K tail: only the first ng 8-wide groups of the step hold data (wave-uniform).
template <int NT>
__device__ __forceinline__ void mfma_ktile_partial(const float *__restrict__ ap, const float *__restrict__ wp, floatx16 (&acc)[NT], int ng) {
    for (int g = 0; g < ng; g++) {
        const float4 a4 = *reinterpret_cast<const float4 *>(ap + 8 * g);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float4 b4 = *reinterpret_cast<const float4 *>(wp + t * 32 * GEMM_LD + 8 * g);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[t], 0, 0, 0);
        }
    }
}

// Activation over a whole accumulator tile with ONE dispatch on the (wave-uniform) code.
template <int NT, class F>
__device__ __forceinline__ void map_tile(floatx16 (&acc)[NT], F f) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = f(acc[t][r]);
}
// activation over a small register array with ONE dispatch on the (uniform) code
template <int N>
__device__ __forceinline__ void act_array(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_NONE) return;
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_SIGMOID) map_array<N>(v, [](float x) { return net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
    else if (act == ACT_HSIGMOID) map_array<N>(v, [=](float x) { return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f); });
    else if (act == ACT_LEAKY) map_array<N>(v, [=](float x) { return x >= 0.0f ? x : p0 * x; });
    else if (act == ACT_TANH) map_array<N>(v, [](float x) { return tanhf(x); });
}

template <int NT>
__device__ __forceinline__ void act_tile(int act, float p0, float p1, floatx16 (&acc)[NT]) {
    if (act == ACT_NONE) return;
    if (act == ACT_RELU) map_tile<NT>(acc, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_tile<NT>(acc, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_tile<NT>(acc, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_SIGMOID) map_tile<NT>(acc, [](float x) { return net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_tile<NT>(acc, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
    else if (act == ACT_HSIGMOID) map_tile<NT>(acc, [=](float x) { return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f); });
    else if (act == ACT_LEAKY) map_tile<NT>(acc, [=](float x) { return x >= 0.0f ? x : p0 * x; });
    else if (act == ACT_TANH) map_tile<NT>(acc, [](float x) { return tanhf(x); });
}

// every unary stage code (the elementwise kernels' set) over an accumulator tile: absorbed elementwise chains
template <int NT>
__device__ __forceinline__ void act_tile_all(int act, float p0, float p1, floatx16 (&acc)[NT]) {
    switch (act) {
        case ACT_EXP: map_tile<NT>(acc, [](float x) { return net_exp(x); }); return;
        case ACT_LOG: map_tile<NT>(acc, [](float x) { return net_log(x); }); return;
        case ACT_SQRT: map_tile<NT>(acc, [](float x) { return sqrtf(x); }); return;
        case ACT_ABS: map_tile<NT>(acc, [](float x) { return fabsf(x); }); return;
        case ACT_NEG: map_tile<NT>(acc, [](float x) { return -x; }); return;
        case ACT_RECIP: map_tile<NT>(acc, [](float x) { return 1.0f / x; }); return;
        case ACT_POW: map_tile<NT>(acc, [=](float x) { return net_pow(x, p0); }); return;
        case ACT_AFFINE: map_tile<NT>(acc, [=](float x) { return p0 * x + p1; }); return;
        case ACT_MAXC: map_tile<NT>(acc, [=](float x) { return fmaxf(x, p0); }); return;
        case ACT_MINC: map_tile<NT>(acc, [=](float x) { return fminf(x, p0); }); return;
        case ACT_RSUB: map_tile<NT>(acc, [=](float x) { return p0 - x; }); return;
        case ACT_RDIV: map_tile<NT>(acc, [=](float x) { return p0 / x; }); return;
        case ACT_SQUARE: map_tile<NT>(acc, [](float x) { return x * x; }); return;
        case ACT_FLOOR: map_tile<NT>(acc, [](float x) { return floorf(x); }); return;
        case ACT_CEIL: map_tile<NT>(acc, [](float x) { return ceilf(x); }); return;
        case ACT_ERF: map_tile<NT>(acc, [](float x) { return erff(x); }); return;
        case ACT_SOFTPLUS: map_tile<NT>(acc, [](float x) { return log1pf(expf(x)); }); return;
        default: act_tile<NT>(act, p0, p1, acc); return;
    }
}

// Shared epilogue: bias, activation, residual, store.  Lane (lr, lh) of a wave holds, in
// acc[t][reg], C[rbase + (reg&3) + 8*(reg>>2) + 4*lh][n0 + 32*t + lr].  Output (and residual) rows
// of one launch are contiguous across samples (ldc == N-stride of a dense [rows*batch, ldc]
// array) whenever c_bs == rows*ldc, which the planner guarantees for its own allocations; then no
// per-row division is needed.
//
// POST (GemmDesc::npost / out_strided): an elementwise chain of unary stages that followed the GEMM in the plan is
// applied to the accumulators, and the result is stored through that chain's output view (element (m, n) of sample
// b at C + b*c_bs + m*out_rs + n*out_cs: transposed / flipped / channel-interleaved targets such as the
// spectrogram image), so the dense GEMM result and the separate strided copy never touch memory.
template <int NT, bool POST = false>
__device__ __forceinline__ void gemm_epilogue(const GemmDesc &d, float *__restrict__ C, const float *__restrict__ bias,
                                              const float *__restrict__ res, floatx16 (&acc)[NT], int64_t rbase, int64_t total_rows,
                                              int n0, int lr, int lh, int reg_lo, int reg_hi) {
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + t * 32 + lr;
        bv[t] = (d.has_bias && n < d.N) ? bias[n] : 0.0f;
    }
    if (d.has_bias) {
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] += bv[t];
    }
    act_tile<NT>(d.act, d.p0, d.p1, acc);
    if constexpr (POST) {
        for (int sidx = 0; sidx < d.npost; sidx++) act_tile_all<NT>(d.post_act[sidx], d.post_p0[sidx], d.post_p1[sidx], acc);
        if (d.out_strided) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int64_t r = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (reg >= reg_lo && reg < reg_hi && r < total_rows) {
                    int64_t b, m;
                    row_split(d, r, b, m);
                    float *crow = C + b * d.c_bs + m * d.out_rs;
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const int n = n0 + t * 32 + lr;
                        if (n < d.N) crow[(int64_t)n * d.out_cs] = acc[t][reg];
                    }
                }
            }
            return;
        }
    }
    const bool flat_c = d.c_bs == d.rows * d.ldc;
    const bool flat_r = !d.has_res || d.r_bs == d.rows * d.ldr;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int64_t r = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (reg >= reg_lo && reg < reg_hi && r < total_rows) {
            float *crow;
            const float *rrow = nullptr;
            if (flat_c && flat_r) {
                crow = C + r * d.ldc;
                if (d.has_res) rrow = res + r * d.ldr;
            } else {
                int64_t b, m;
                row_split(d, r, b, m);
                crow = C + b * d.c_bs + m * d.ldc;
                if (d.has_res) rrow = res + b * d.r_bs + m * d.ldr;
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int n = n0 + t * 32 + lr;
                if (n < d.N) {
                    float v = acc[t][reg];
                    if (d.has_res) v += rrow[n];
                    crow[n] = v;
                }
            }
        }
    }
}

template <int BN, int AVEC, int WVEC, bool GATED, bool FOLD = false, bool POST = false>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmDesc d, float *__restrict__ C,
                                                        const float *__restrict__ A,
                                                        const float *__restrict__ W,
                                                        const float *__restrict__ bias,
                                                        const float *__restrict__ res,
                                                        const float *__restrict__ scale, int64_t total_rows) {
    constexpr int NT = BN / 32;
    // staging geometry: a pass of 256 threads covers 256*VEC/32 rows of 32 floats
    constexpr int A_RPP = 256 * AVEC / GEMM_BK, A_IT = GEMM_BM / A_RPP;
    constexpr int W_RPP = 256 * WVEC / GEMM_BK, W_IT = BN / W_RPP;
    __shared__ __align__(16) float As[GEMM_BM * GEMM_LD];
    __shared__ __align__(16) float Ws[BN * GEMM_LD];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * GEMM_BM;
    const int n0 = blockIdx.y * BN;
    const int a_row = tid / (GEMM_BK / AVEC), a_col = (tid % (GEMM_BK / AVEC)) * AVEC;
    const int w_row = tid / (GEMM_BK / WVEC), w_col = (tid % (GEMM_BK / WVEC)) * WVEC;

    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;

    int64_t aoff[A_IT], soff[A_IT];
    a_offsets<AVEC, A_RPP, A_IT>(d, total_rows, row0, a_row, aoff, soff);
    // PF K tiles are kept in flight in registers: a load round trip costs ~2 us on this part while
    // the MFMAs of one K step take 0.2-0.9 us, so a one-deep prefetch leaves the waves waiting
    constexpr int PF = GEMM_PF;
    static_assert(!FOLD || (AVEC == 2 && !GATED), "folded rows are staged as float pairs, ungated");
    float ra[PF][A_IT * AVEC];
    float rm[PF][FOLD ? A_IT * AVEC : 1];  // mirrored half of a folded row
    float rg[PF][GATED ? A_IT * AVEC : 1];
    const float fold_sign = (float)d.fold;
    float rw[PF][W_IT * WVEC];
    const int ksteps = (d.K + GEMM_BK - 1) / GEMM_BK;
#pragma unroll
    for (int p = 0; p < PF; p++)
        if (p < ksteps) {
            if constexpr (FOLD) load_a_fold_regs<A_IT>(d, A, aoff, p * GEMM_BK + a_col, ra[p], rm[p]);
            else load_a_regs<AVEC, A_IT>(d, A, aoff, p * GEMM_BK + a_col, ra[p]);
            if constexpr (GATED) load_gate_regs<AVEC, A_IT>(d, scale, soff, p * GEMM_BK + a_col, rg[p]);
            load_w_regs<WVEC, W_RPP, W_IT>(d, W, n0 + w_row, p * GEMM_BK + w_col, rw[p]);
        }
    const bool wave_active = row0 + wave * 32 < total_rows;
    const float *ap = As + (wave * 32 + lr) * GEMM_LD + 4 * lh, *wp = Ws + lr * GEMM_LD + 4 * lh;
    const int kfull = d.K / GEMM_BK;  // K steps with all 32 columns present
    for (int ks0 = 0; ks0 < kfull; ks0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int ks = ks0 + p;
            if (ks < kfull) {  // block-uniform
                __syncthreads();  // previous tile fully consumed
                if constexpr (GATED) {
#pragma unroll
                    for (int i = 0; i < A_IT * AVEC; i++) ra[p][i] *= rg[p][i];
                }
                if constexpr (FOLD) {
#pragma unroll
                    for (int i = 0; i < A_IT * AVEC; i++) ra[p][i] = fmaf(fold_sign, rm[p][i], ra[p][i]);
                }
                store_tile_regs<AVEC, A_RPP, A_IT>(As, a_row, a_col, ra[p]);
                store_tile_regs<WVEC, W_RPP, W_IT>(Ws, w_row, w_col, rw[p]);
                __syncthreads();
                if (ks + PF < ksteps) {  // refill this slot with the tile PF steps ahead
                    if constexpr (FOLD) load_a_fold_regs<A_IT>(d, A, aoff, (ks + PF) * GEMM_BK + a_col, ra[p], rm[p]);
                    else load_a_regs<AVEC, A_IT>(d, A, aoff, (ks + PF) * GEMM_BK + a_col, ra[p]);
                    if constexpr (GATED) load_gate_regs<AVEC, A_IT>(d, scale, soff, (ks + PF) * GEMM_BK + a_col, rg[p]);
                    load_w_regs<WVEC, W_RPP, W_IT>(d, W, n0 + w_row, (ks + PF) * GEMM_BK + w_col, rw[p]);
                }
                mfma_ktile_full<NT>(ap, wp, acc);  // waves past the last row multiply clamped rows; their results are never stored
            }
        }
    }
    if (kfull < ksteps) {  // K tail: zero-filled past K, only the 8-wide groups holding data are multiplied
        const int slot = kfull % PF;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PF; p++)
            if (p == slot) {
                if constexpr (GATED) {
#pragma unroll
                    for (int i = 0; i < A_IT * AVEC; i++) ra[p][i] *= rg[p][i];
                }
                if (d.K % 8) {  // ragged K: the last multiplied group reaches past K
                    zero_past_k<AVEC>(kfull * GEMM_BK + a_col, d.K, ra[p]);
                    zero_past_k<WVEC>(kfull * GEMM_BK + w_col, d.K, rw[p]);
                }
                store_tile_regs<AVEC, A_RPP, A_IT>(As, a_row, a_col, ra[p]);
                store_tile_regs<WVEC, W_RPP, W_IT>(Ws, w_row, w_col, rw[p]);
            }
        __syncthreads();
        mfma_ktile_partial<NT>(ap, wp, acc, (d.K - kfull * GEMM_BK + 7) / 8);
    }
    if (!wave_active) return;
    gemm_epilogue<NT, POST>(d, C, bias, res, acc, row0 + wave * 32, total_rows, n0, lr, lh, 0, 16);
}

// ------------------------------------------------------------------ folded framing GEMM, signal resident in LDS
// The folded framing GEMM above re-reads every frame row from L2 once per K step and N tile: 530 MB of L2 -> L1
// traffic per launch at batch 32 (6.8 TB/s; matrix pipe 40 % busy, PMC).  Frames overlap (hop 278 of 2048 taps), so
// the 64 frames of a block span only 63*hop + L samples (<= 80 KB): this kernel loads that span into LDS ONCE and
// builds each folded 64 x 32 operand tile from it (LDS -> LDS, double buffered), while the filter tile streams
// through registers as before.  2*WN waves: wave w owns rows 32*(w&1).. and columns 32*(w>>1)..; one barrier per K
// step.  Same K order, fragment layout and fold expression as gemm_mfma_kernel<.., FOLD>: bit-identical results.
struct FrameDesc {
    int32_t rows, N, K, L, hop;   // per-sample frames, outputs, folded taps (L/2), filter length, frame hop
    int32_t tiles;                // 64-row tiles per sample
    int32_t span;                 // floats of signal one block keeps: 63*hop + L
    int32_t vec4;                 // span loads as float4 (16-byte aligned sample rows)
    int64_t a_bs, ldc, c_bs;
    float sign;
    int32_t has_bias;
    // absorbed elementwise chain (planner rule E) and the consumer's view: element (row m, column n) of sample b at C + b c_bs + m out_rs + n out_cs
    int32_t npost, out_strided;
    int32_t post_act[4];
    float post_p0[4], post_p1[4];
    int64_t out_rs, out_cs;
};
constexpr int FRAME_BM = 64;
static_assert(FRAME_BM == FRAME_BM_RULE && GEMM_BK == GEMM_BK_RULE && GEMM_LD == GEMM_LD_RULE, "plan_rules.h restates the framing GEMM tile sizes");
// One shared copy of the compact stage dispatch (device_common.h) for a wave's 16 accumulators, by value: registers in, registers out.
// (The generic gemm_epilogue inlines libm for every stage code: behind this kernel it took 256 registers and scratch.)
__device__ __noinline__ floatx16 act_small16(int act, float p0, float p1, floatx16 a) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = a[r];
    act_small<16>(act, p0, p1, v);
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = v[r];
    return a;
}
// PAIR (planner rule J): the product that consumes the rows of this one -- the mel filter bank of the spectrogram branch, K2 = N <= 128
// spectrum bins -> N2 <= 128 bands, with its absorbed epilogue chain and output view -- runs behind the K loop on the block's own 64 x N
// spectrum tile (accumulators -> LDS as 32-deep K tiles of the second product, filter rows staged step by step, the shared
// gemm_epilogue): the spectrum rows are neither written nor read back and the launch that did it is gone.
template <bool PAIR>
__global__ __launch_bounds__(PAIR ? 512 : 640) void frame_fold_kernel(FrameDesc d, float *__restrict__ C, const float *__restrict__ A,
                                                         const float *__restrict__ W, const float *__restrict__ bias, GemmDesc d2,
                                                         float *__restrict__ C2, const float *__restrict__ W2, const float *__restrict__ bias2) {
    extern __shared__ __align__(16) float frame_lds[];
    const int T = blockDim.x, tid = threadIdx.x;
    const int BN = (T >> 7) * 32;  // WN wave columns of 32 outputs
    float *sig = frame_lds;
    float *As = sig + ((d.span + 3) & ~3);
    float *Ws = As + 2 * FRAME_BM * GEMM_LD;
    const int b = blockIdx.x / d.tiles, rt = blockIdx.x - b * d.tiles;
    const int row0 = rt * FRAME_BM;
    const int rows_here = min(FRAME_BM, d.rows - row0);
    const float *src = A + (int64_t)b * d.a_bs + (int64_t)row0 * d.hop;
    const int count = (rows_here - 1) * d.hop + d.L;
    if (d.vec4) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(sig);
        const int n4 = count >> 2;
        for (int i = tid; i < n4; i += T) d4[i] = s4[i];
        for (int i = (n4 << 2) + tid; i < count; i += T) sig[i] = src[i];
    } else {
        for (int i = tid; i < count; i += T) sig[i] = src[i];
    }
    // filter tile staging: BN rows x 32 taps = BN * 8 float4, up to 2 per thread (BN <= 160, T = 4 * BN)
    const int wq = tid & 7, wr = tid >> 3;  // float4 column, row of this thread's first filter vector
    const int wrows_per_pass = T >> 3;      // = BN / 2
    int n0 = 0;
    auto w_ptr = [&](int pass, int k0) {
        int n = n0 + wr + pass * wrows_per_pass;
        n = n < d.N ? n : d.N - 1;
        return reinterpret_cast<const float4 *>(W + (int64_t)n * d.K + k0 + 4 * wq);
    };
    float4 rw0, rw1;
    const int ksteps = d.K / GEMM_BK;
    // operand staging: a thread owns up to 4 (row, column pair) slots of the 64 x 32 tile, fixed over the K loop, so the
    // row offsets are resolved once; per K step it reads the forward and mirrored pairs and writes one float2
    // (measured: pairs beat the conflict-free one-element-per-lane mapping, 49 vs 54 us -- instruction count, not banks)
    constexpr int SLOTS = 4;
    int fwd_off[SLOTS], rev_off[SLOTS], dst_off[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
        const int p = tid + i * T;
        const int r = (p >> 4) & (FRAME_BM - 1), cp = (p & 15) * 2;
        const int re = r < rows_here ? r : rows_here - 1;
        fwd_off[i] = re * d.hop + 1 + cp;
        rev_off[i] = re * d.hop + d.L - 2 - cp;
        dst_off[i] = p < FRAME_BM * 16 ? r * GEMM_LD + cp : -1;
    }
    auto stage_a = [&](int ks, float *dst) {
        const int k0 = ks * GEMM_BK;
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            if (dst_off[i] >= 0) {
                const float *f = sig + fwd_off[i] + k0, *m = sig + rev_off[i] - k0;
                *reinterpret_cast<float2 *>(dst + dst_off[i]) = make_float2(fmaf(d.sign, m[1], f[0]), fmaf(d.sign, m[0], f[1]));
            }
        }
    };
    auto store_w = [&](float *dst) {
        *reinterpret_cast<float4 *>(dst + wr * GEMM_LD + 4 * wq) = rw0;
        *reinterpret_cast<float4 *>(dst + (wr + wrows_per_pass) * GEMM_LD + 4 * wq) = rw1;
    };
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    __syncthreads();  // signal span complete
    // N tiles of this row tile: with grid.y == 1 one block walks all of them and the span is loaded once
    for (int ny = blockIdx.y; ny * BN < d.N; ny += gridDim.y) {
    n0 = ny * BN;
    rw0 = *w_ptr(0, 0); rw1 = *w_ptr(1, 0);
    stage_a(0, As);
    store_w(Ws);
    if (ksteps > 1) { rw0 = *w_ptr(0, GEMM_BK); rw1 = *w_ptr(1, GEMM_BK); }
    __syncthreads();
    floatx16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][r] = 0.0f;
    for (int ks = 0; ks < ksteps; ks++) {
        const int cur = ks & 1;
        float *An = As + (cur ^ 1) * FRAME_BM * GEMM_LD, *Wn = Ws + (cur ^ 1) * BN * GEMM_LD;
        if (ks + 1 < ksteps) {
            stage_a(ks + 1, An);
            store_w(Wn);
            if (ks + 2 < ksteps) { rw0 = *w_ptr(0, (ks + 2) * GEMM_BK); rw1 = *w_ptr(1, (ks + 2) * GEMM_BK); }
        }
        const float *ap = As + cur * FRAME_BM * GEMM_LD + (wm * 32 + lr) * GEMM_LD + 4 * lh;
        const float *wp = Ws + cur * BN * GEMM_LD + (wn * 32 + lr) * GEMM_LD + 4 * lh;
        mfma_ktile_full<1>(ap, wp, acc);
        // LDS-only rendezvous: the operand tiles of step ks + 1 are in place.  (__syncthreads would also wait for the filter
        // rows of step ks + 2, requested a moment ago: they are only needed after the next step's multiplications.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int n = n0 + wn * 32 + lr;
    if constexpr (PAIR) {
        // (one N tile covers the whole spectrum row: the launcher guarantees N <= BN and grid.y == 1)
        // the spectrum tile (+ bias) as K tiles of the second product: S[ks = wn][row][k = lr]; columns n >= N hold copies of the
        // last filter row's output and meet zero taps below
        float *S = As;                                    // [BN / 32][64][GEMM_LD]  (both tile buffers are free: the K loop ended on a barrier)
        float *W2s = S + (BN >> 5) * FRAME_BM * GEMM_LD;  // [<= 128][GEMM_LD]
        {
            const float bv = (d.has_bias && n < d.N) ? bias[n] : 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                S[(wn * FRAME_BM + r) * GEMM_LD + lr] = acc[0][reg] + bv;
            }
        }
        const int n2tiles = (d2.N + 31) >> 5, k2steps = (d2.K + 31) >> 5;
        floatx16 acc2[1];
#pragma unroll
        for (int r = 0; r < 16; r++) acc2[0][r] = 0.0f;
        for (int ks2 = 0; ks2 < k2steps; ks2++) {
            // filter rows of this step: W2 is [N2][K2], K2 contiguous; rows / taps past the end are zero
            for (int e = tid; e < n2tiles * 32 * 32; e += T) {
                const int n2 = e >> 5, kk = e & 31, k = ks2 * 32 + kk;
                W2s[n2 * GEMM_LD + kk] = (n2 < d2.N && k < d2.K) ? W2[(int64_t)n2 * d2.K + k] : 0.0f;
            }
            __syncthreads();  // (first step: the spectrum tile is complete as well)
            if (wn < n2tiles)
                mfma_ktile_full<1>(S + (ks2 * FRAME_BM + wm * 32 + lr) * GEMM_LD + 4 * lh, W2s + (wn * 32 + lr) * GEMM_LD + 4 * lh, acc2);
            __syncthreads();
        }
        if (wn < n2tiles) {
            const int64_t rb = (int64_t)b * d2.rows + row0;
            // epilogue of the second product: bias, activation, the absorbed chain, the consumer's view (gemm_epilogue's semantics with
            // the compact stage functions; the planner fuses only chains made of those)
            const int n2 = wn * 32 + lr;
            floatx16 a2 = acc2[0];
            if (d2.has_bias) {
                const float bv2 = n2 < d2.N ? bias2[n2] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; r++) a2[r] += bv2;
            }
            if (d2.act != ACT_NONE) a2 = act_small16(d2.act, d2.p0, d2.p1, a2);
            if (0 < d2.npost) a2 = act_small16(d2.post_act[0], d2.post_p0[0], d2.post_p1[0], a2);
            if (1 < d2.npost) a2 = act_small16(d2.post_act[1], d2.post_p0[1], d2.post_p1[1], a2);
            if (2 < d2.npost) a2 = act_small16(d2.post_act[2], d2.post_p0[2], d2.post_p1[2], a2);
            if (3 < d2.npost) a2 = act_small16(d2.post_act[3], d2.post_p0[3], d2.post_p1[3], a2);
            if (n2 < d2.N) {
                const int64_t rs = d2.out_strided ? d2.out_rs : d2.ldc, cs = d2.out_strided ? d2.out_cs : 1;
                float *cb2 = C2 + (int64_t)b * d2.c_bs + (int64_t)n2 * cs;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    if (r < rows_here) cb2[(int64_t)(row0 + r) * rs] = a2[reg];
                }
            }
            (void)rb;
        }
    } else if (d.npost || d.out_strided) {  // bias, the absorbed chain (compact stage functions), the consumer's view
        floatx16 a2 = acc[0];
        if (d.has_bias) {
            const float bv = n < d.N ? bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) a2[r] += bv;
        }
        if (0 < d.npost) a2 = act_small16(d.post_act[0], d.post_p0[0], d.post_p1[0], a2);
        if (1 < d.npost) a2 = act_small16(d.post_act[1], d.post_p0[1], d.post_p1[1], a2);
        if (2 < d.npost) a2 = act_small16(d.post_act[2], d.post_p0[2], d.post_p1[2], a2);
        if (3 < d.npost) a2 = act_small16(d.post_act[3], d.post_p0[3], d.post_p1[3], a2);
        if (n < d.N) {
            const int64_t rs = d.out_strided ? d.out_rs : d.ldc, cs = d.out_strided ? d.out_cs : 1;
            float *cb = C + (int64_t)b * d.c_bs + (int64_t)n * cs;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (r < rows_here) cb[(int64_t)(row0 + r) * rs] = a2[reg];
            }
        }
    } else if (n < d.N) {
        const float bv = d.has_bias ? bias[n] : 0.0f;
        float *cb = C + (int64_t)b * d.c_bs + (int64_t)row0 * d.ldc + n;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (r < rows_here) cb[(int64_t)r * d.ldc] = acc[0][reg] + bv;
        }
    }
    // (the last K step ended with a barrier: every wave is done with both tile buffers)
    }
}

// LDS addresses as 32-bit byte offsets (ds_* instructions take base register + immediate; generic pointers cost an add each)
typedef __attribute__((address_space(3))) float lds_float;
typedef float lds_f2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lds_f2v lds_float2;
__device__ __forceinline__ uint32_t lds_offset_of(const float *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)p; }
__device__ __forceinline__ float lds_ld(uint32_t byte_off) { return *(const lds_float *)(uintptr_t)byte_off; }
__device__ __forceinline__ float2 lds_ld2(uint32_t byte_off) {
    const lds_f2v v = *(const lds_float2 *)(uintptr_t)byte_off;
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void lds_st2(uint32_t byte_off, float2 v) { *(lds_float2 *)(uintptr_t)byte_off = lds_f2v{v.x, v.y}; }

// The block's signal span into LDS, with the absorbed per-sample chain (FramePre) applied on the way when there is one: the normalised
// segment is never written.  Literal indices into the by-value descriptor (a run-time index would move it to scratch memory).
__device__ __forceinline__ void frame_load_span(float *sig, const float *__restrict__ src, int count, bool vec4, int tid, int T, const FramePre &pre, int64_t b) {
    if (pre.n > 0) {
        static_assert(ELT_MAX_STAGES == 4, "the stages below are spelled out");
        PreChain c{pre.n, {pre.bin[0], pre.bin[1], pre.bin[2], pre.bin[3]}, {pre.act[0], pre.act[1], pre.act[2], pre.act[3]}, {0.f, 0.f, 0.f, 0.f},
                   {pre.p0[0], pre.p0[1], pre.p0[2], pre.p0[3]}, {pre.p1[0], pre.p1[1], pre.p1[2], pre.p1[3]}};
        if (0 < c.n && c.bin[0] != BIN_NONE) c.sc[0] = pre.sc[0][b * pre.bb[0]];
        if (1 < c.n && c.bin[1] != BIN_NONE) c.sc[1] = pre.sc[1][b * pre.bb[1]];
        if (2 < c.n && c.bin[2] != BIN_NONE) c.sc[2] = pre.sc[2][b * pre.bb[2]];
        if (3 < c.n && c.bin[3] != BIN_NONE) c.sc[3] = pre.sc[3][b * pre.bb[3]];
        // PB float4 per thread and trip: the loads of a trip are in flight together and every stage is dispatched once for 4 PB values
        // (one float4 per trip made the chain cost more than the launch it replaces: a load round trip and the stage dispatch per value)
        constexpr int PB = 8;
        const int n4 = vec4 ? count >> 2 : 0;
        for (int i0 = tid; i0 < n4; i0 += PB * T) {
            float v[4 * PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const float4 x = reinterpret_cast<const float4 *>(src)[min(i0 + j * T, n4 - 1)];
                v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
            }
            pre_chain<4 * PB>(c, v);
#pragma unroll
            for (int j = 0; j < PB; j++)
                if (i0 + j * T < n4) reinterpret_cast<float4 *>(sig)[i0 + j * T] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        }
        for (int i = (n4 << 2) + tid; i < count; i += T) {
            float v[1] = {src[i]};
            pre_chain<1>(c, v);
            sig[i] = v[0];
        }
        return;
    }
    if (vec4) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(sig);
        const int n4 = count >> 2;
        for (int i = tid; i < n4; i += T) d4[i] = s4[i];
        for (int i = (n4 << 2) + tid; i < count; i += T) sig[i] = src[i];
    } else {
        for (int i = tid; i < count; i += T) sig[i] = src[i];
    }
}

// ------------------------------------------------------------------ half-folded framing GEMM, round-4 form
// frame_fold_kernel<false>'s job with the structure that paid for the quarter fold below: compile-time block shape (WN wave columns), LDS
// addresses as 32-bit byte offsets, the signal reads of the NEXT step's operand tile issued before this step's matrix instructions and
// finished behind them, nothing conditional around the matrix instructions.  KS = 2: the 8-wide k groups of every step are split between
// two sets of 2 WN waves (groups 0, 1 | groups 2, 3) whose partial tiles are added through LDS at the end, slice 0 + slice 1 -- for
// THREE wave columns (N = 65 .. 96: v2.4's 96 merged mel filters) six waves leave two of the four SIMDs with one wave and the block
// runs at the pace of the two that have two; twelve waves balance (3 per SIMD).  KS is a function of N alone (frame_fold_kslices), so a
// segment's bits do not depend on the batch; with KS = 1 the k order, fragment layout and fold expression are those of
// gemm_mfma_kernel<.., FOLD> and of frame_fold_kernel (bit-identical, tested), with KS = 2 the sum is cut once more.
template <int NG>
__device__ __forceinline__ void mfma_ktile_groups(const float *__restrict__ ap, const float *__restrict__ wp, floatx16 &acc) {
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const float4 a4 = *reinterpret_cast<const float4 *>(ap + 8 * g);
        const float4 b4 = *reinterpret_cast<const float4 *>(wp + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
}
template <int WN, int KS>
__global__ __launch_bounds__(128 * WN * KS) void frame_foldh_kernel(FrameDesc d, float *__restrict__ C, const float *__restrict__ A, const float *__restrict__ W,
                                                                    const float *__restrict__ bias, FramePre pre) {
    extern __shared__ __align__(16) float frame_lds[];
    constexpr int T = 128 * WN * KS, BN = 32 * WN, TILE = FRAME_BM * GEMM_LD, SLOTS = (FRAME_BM * 16 + T - 1) / T;
    constexpr int WPASS = (8 * BN + T - 1) / T;  // float4 of the filter tile per thread (2, or 1 with two K slices)
    static_assert(8 * BN == WPASS * T && WPASS <= 2, "the filter tile is one or two whole passes");
    const int tid = threadIdx.x;
    float *sig = frame_lds;
    float *As = sig + ((d.span + 3) & ~3);  // [2][64][GEMM_LD]
    float *Ws = As + 2 * TILE;              // [2][BN][GEMM_LD]
    const int b = blockIdx.x / d.tiles, rt = blockIdx.x - b * d.tiles;
    const int row0 = rt * FRAME_BM;
    const int rows_here = min(FRAME_BM, d.rows - row0);
    const float *src = A + (int64_t)b * d.a_bs + (int64_t)row0 * d.hop;
    const int count = (rows_here - 1) * d.hop + d.L;
    frame_load_span(sig, src, count, d.vec4 != 0, tid, T, pre, b);
    const int wq = tid & 7, wr = tid >> 3;
    constexpr int WROWS = T >> 3;
    const int ksteps = d.K / GEMM_BK;
    const uint32_t sig0 = lds_offset_of(sig), as0 = lds_offset_of(As);
    const int cp = (tid & 15) * 2;
    uint32_t a_fw0[SLOTS], a_rv0[SLOTS], a_dst[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
        const int p = tid + i * T;
        const int r = (p >> 4) & (FRAME_BM - 1);
        const int re = r < rows_here ? r : rows_here - 1;
        const uint32_t f = sig0 + 4u * (uint32_t)(re * d.hop);
        a_fw0[i] = f + 4u * (1 + cp);            // x[1 + c], x[2 + c]
        a_rv0[i] = f + 4u * (d.L - 2 - cp);      // x[L - 2 - c], x[L - 1 - c]
        a_dst[i] = as0 + 4u * (uint32_t)(r * GEMM_LD + cp);
    }
    constexpr bool LAST_PARTIAL = (FRAME_BM * 16) % T != 0;
    const bool last_on = !LAST_PARTIAL || tid + (SLOTS - 1) * T < FRAME_BM * 16;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int kh = wave / (2 * WN), wv = wave - kh * 2 * WN;  // K slice, wave inside the slice
    const int wm = wv & 1, wn = wv >> 1;
    const float sign = d.sign;
    __syncthreads();  // signal span complete
    for (int ny = blockIdx.y; ny * BN < d.N; ny += gridDim.y) {
        const int n0 = ny * BN;
        // (scalars, not arrays: a register array captured by a lambda moves to scratch memory)
        const float *wrow0 = W + (int64_t)min(n0 + wr, d.N - 1) * d.K + 4 * wq;
        const float *wrow1 = W + (int64_t)min(n0 + wr + WROWS, d.N - 1) * d.K + 4 * wq;  // (WPASS == 2)
        float4 rw0, rw1 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto load_w = [&](int k0) {
            rw0 = *reinterpret_cast<const float4 *>(wrow0 + k0);
            if constexpr (WPASS == 2) rw1 = *reinterpret_cast<const float4 *>(wrow1 + k0);
        };
        uint32_t a_fw[SLOTS], a_rv[SLOTS];
#pragma unroll
        for (int i = 0; i < SLOTS; i++) { a_fw[i] = a_fw0[i]; a_rv[i] = a_rv0[i]; }
        float2 xf[SLOTS], xr[SLOTS];
        auto load_a = [&]() {
#pragma unroll
            for (int i = 0; i < SLOTS; i++) {
                xf[i] = make_float2(lds_ld(a_fw[i]), lds_ld(a_fw[i] + 4));
                xr[i] = make_float2(lds_ld(a_rv[i]), lds_ld(a_rv[i] + 4));
            }
        };
        auto finish_a = [&](uint32_t buf_off) {
#pragma unroll
            for (int i = 0; i < SLOTS; i++) {
                if (i < SLOTS - 1 || last_on) lds_st2(a_dst[i] + buf_off, make_float2(fmaf(sign, xr[i].y, xf[i].x), fmaf(sign, xr[i].x, xf[i].y)));
                a_fw[i] += 4 * GEMM_BK;
                a_rv[i] -= 4 * GEMM_BK;
            }
        };
        auto store_w = [&](float *dst) {
            *reinterpret_cast<float4 *>(dst + wr * GEMM_LD + 4 * wq) = rw0;
            if constexpr (WPASS == 2) *reinterpret_cast<float4 *>(dst + (wr + WROWS) * GEMM_LD + 4 * wq) = rw1;
        };
        load_w(0);
        load_a();
        finish_a(0);
        store_w(Ws);
        load_w(min(1, ksteps - 1) * GEMM_BK);
        __syncthreads();
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        const float *ap0 = As + (wm * 32 + lr) * GEMM_LD + 4 * lh + (KS == 2 ? 16 * kh : 0);
        const float *wp0 = Ws + (wn * 32 + lr) * GEMM_LD + 4 * lh + (KS == 2 ? 16 * kh : 0);
        for (int ks = 0; ks < ksteps; ks++) {
            const int cur = ks & 1;
            load_a();  // step ks + 1; behind the last step the streams read one step further inside the span's neighbourhood: see below
            mfma_ktile_groups<4 / KS>(ap0 + cur * TILE, wp0 + cur * BN * GEMM_LD, acc);
            finish_a((uint32_t)((cur ^ 1) * TILE * 4));
            store_w(Ws + (cur ^ 1) * BN * GEMM_LD);
            load_w(min(ks + 2, ksteps - 1) * GEMM_BK);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if constexpr (KS == 2) {  // slice 1's partial tile through LDS (both tile buffers are free: the loop ended on a barrier)
            floatx16 *xch = reinterpret_cast<floatx16 *>(As);  // [2 WN waves][64 lanes]
            if (kh == 1) xch[wv * 64 + lane] = acc;
            __syncthreads();
            if (kh == 0) {
                const floatx16 o = xch[wv * 64 + lane];
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] += o[r];
            }
        }
        const int n = n0 + wn * 32 + lr;
        if (kh == 0) {
            if (d.npost || d.out_strided) {  // bias, the absorbed chain (compact stage functions), the consumer's view
                floatx16 a2 = acc;
                if (d.has_bias) {
                    const float bv = n < d.N ? bias[n] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; r++) a2[r] += bv;
                }
                if (0 < d.npost) a2 = act_small16(d.post_act[0], d.post_p0[0], d.post_p1[0], a2);
                if (1 < d.npost) a2 = act_small16(d.post_act[1], d.post_p0[1], d.post_p1[1], a2);
                if (2 < d.npost) a2 = act_small16(d.post_act[2], d.post_p0[2], d.post_p1[2], a2);
                if (3 < d.npost) a2 = act_small16(d.post_act[3], d.post_p0[3], d.post_p1[3], a2);
                if (n < d.N) {
                    const int64_t rs = d.out_strided ? d.out_rs : d.ldc, cs = d.out_strided ? d.out_cs : 1;
                    float *cb = C + (int64_t)b * d.c_bs + (int64_t)n * cs;
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) {
                        const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        if (r < rows_here) cb[(int64_t)(row0 + r) * rs] = a2[reg];
                    }
                }
            } else if (n < d.N) {
                const float bv = d.has_bias ? bias[n] : 0.0f;
                float *cb = C + (int64_t)b * d.c_bs + (int64_t)row0 * d.ldc + n;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    if (r < rows_here) cb[(int64_t)r * d.ldc] = acc[reg] + bv;
                }
            }
        }
        if constexpr (KS == 2) __syncthreads();  // the exchange buffer is the next N tile's first operand tile
    }
}

// ------------------------------------------------------------------ quarter-folded framing GEMM (round 4)
// A bank of windowed COSINES (the real part of an STFT: v2.4's 127 mel-live bins of a 2048-point transform) needs a quarter of the filter
// length per output, not half.  With y[n] = w[n] x[n] (w symmetric about the frame centre, w[0] = 0) and ye[n] = y[n] + y[L-n]:
//     X[k] = sum_{n=0}^{L/2} ye[n] cos(2 pi k n / L),        cos(2 pi k (L/2 - n) / L) = (-1)^k cos(2 pi k n / L)
//  =>  even k:  sum_{n<L/4} S[n] cos(2 pi k n / L) + ye[L/4] cos(pi k / 2),   S[n] = ye[n] + ye[L/2-n]
//      odd  k:  sum_{n<L/4} D[n] cos(2 pi k n / L),                            D[n] = ye[n] - ye[L/2-n]
// (n = 0 pairs ye[0] = 0 with the centre tap ye[L/2] = y[L/2]: table entries wa[0] = 0, wb[0] = w[L/2] / 2 say so.)  The window cannot stay
// inside the filter rows as in the half fold -- w[n] != w[L/2-n] -- so it multiplies the signal: S[n] = wa[n] a + wb[n] b, D[n] = wa[n] a -
// wb[n] b with a = x[n] + x[L-n], b = x[L/2-n] + x[L/2+n]; the filter rows are the planner's pure cosines (double -> f32) times the row's
// amplitude.  Same structure as frame_fold_kernel: the block's signal span in LDS once, per 32-tap step TWO operand tiles (S and D) built
// LDS -> LDS and one filter tile streamed through registers, one barrier per step; wave (w & 1, w >> 1) owns rows 32 (w & 1).. of the
// columns 32 (w >> 1).. and reads the S tile if its columns are even bins (column < n_even), else the D tile.  L / 128 full steps + one
// step of a single 8-wide group that carries tap L/4.  Half the matrix instructions of the half fold for ~3x its (small) staging
// arithmetic.  Arithmetic differs from the half fold in the last place (window on the signal, cosines rounded once, another pairing) --
// the same kind of difference as between the half fold and the plain convolution, far inside the path's 2e-4 tolerance; both forms are
// held against the oracle (tests/test_gpu_ops.py, test_gpu_models.py with BN_CONVFOLD2=0 and default).
struct Frame2Desc {
    int32_t rows, N, K, L, hop;  // per-sample frames, columns (both groups, padded), filter row length (L/4 + 32), frame length, hop
    int32_t tiles, span, vec4;   // 64-row tiles per sample; floats of signal a block keeps: 63 * hop + L; span loads as float4
    int32_t n_even;              // columns of the even-bin group
    int32_t has_bias;
    int64_t a_bs, ldc, c_bs;
};
// WN wave columns of 32 outputs (N == 32 WN), 2 WN waves; a thread owns SLOTS (row, column pair) slots of the two 64 x 32 operand tiles.
// Schedule of a step: the signal / table reads of the NEXT step's operand tiles are issued, this step's matrix instructions run over them,
// then the next tiles are finished (12 vector operations per slot) and written -- nothing conditional around the matrix instructions
// (a branch there makes the compiler copy the accumulators twice per step), the single-group tail step sits behind the loop.
template <int WN>
__global__ __launch_bounds__(128 * WN) void frame_fold2_kernel(Frame2Desc d, float *__restrict__ C, const float *__restrict__ A, const float *__restrict__ W,
                                                               const float *__restrict__ bias, const float *__restrict__ wtab, const int32_t *__restrict__ colmap,
                                                               FramePre pre) {
    extern __shared__ __align__(16) float frame_lds[];
    constexpr int T = 128 * WN, BN = 32 * WN, TILE = FRAME_BM * GEMM_LD, SLOTS = (FRAME_BM * 16 + T - 1) / T;
    const int tid = threadIdx.x;
    float *sig = frame_lds;
    float *tab = sig + ((d.span + 4 + 3) & ~3);  // [2][K]: wa | wb
    float *As = tab + 2 * d.K;                   // [2 buffers][S | D][64][GEMM_LD]
    float *Ws = As + 4 * TILE;                   // [2 buffers][BN][GEMM_LD]
    const int b = blockIdx.x / d.tiles, rt = blockIdx.x - b * d.tiles;
    const int row0 = rt * FRAME_BM;
    const int rows_here = min(FRAME_BM, d.rows - row0);
    const float *src = A + (int64_t)b * d.a_bs + (int64_t)row0 * d.hop;
    const int count = (rows_here - 1) * d.hop + d.L;
    frame_load_span(sig, src, count, d.vec4 != 0, tid, T, pre, b);
    if (tid < 4) sig[count + tid] = 0.0f;  // tap 0 pairs x[0] with "x[L]" under a zero coefficient: a finite value, not whatever LDS held
    for (int i = tid; i < 2 * d.K; i += T) tab[i] = wtab[i];
    const int wq = tid & 7, wr = tid >> 3;  // filter tile staging: BN rows x 32 taps = BN * 8 float4, two per thread (T = 4 * BN)
    constexpr int WROWS = T >> 3;           // = BN / 2
    const int nfull = d.L / (4 * GEMM_BK);  // full steps; step nfull carries tap L/4 in its first 8-wide group
    const float *wrow0 = W + (int64_t)wr * d.K + 4 * wq, *wrow1 = wrow0 + (int64_t)WROWS * d.K;
    float4 rw0, rw1;
    // operand staging: byte addresses of the slot's four signal streams at step 0 (forward streams advance 128 B per step, mirrored ones
    // retreat), of the thread's table entries (every slot of a thread has the same column pair) and of its destination
    uint32_t a_fw[SLOTS], a_rv[SLOTS], a_hm[SLOTS], a_hp[SLOTS], a_dst[SLOTS];
    const uint32_t sig0 = lds_offset_of(sig), as0 = lds_offset_of(As);
    const int cp = (tid & 15) * 2;
    uint32_t a_wa = lds_offset_of(tab) + 4u * cp, a_wb = a_wa + 4u * d.K;
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
        const int p = tid + i * T;
        const int r = (p >> 4) & (FRAME_BM - 1);
        const int re = r < rows_here ? r : rows_here - 1;
        const uint32_t f = sig0 + 4u * (uint32_t)(re * d.hop);
        a_fw[i] = f + 4u * cp;                       // x[n], x[n+1]
        a_rv[i] = f + 4u * (d.L - cp - 1);           // x[L-n-1], x[L-n]
        a_hm[i] = f + 4u * (d.L / 2 - cp - 1);       // x[L/2-n-1], x[L/2-n]
        a_hp[i] = f + 4u * (d.L / 2 + cp);           // x[L/2+n], x[L/2+n+1]
        a_dst[i] = as0 + 4u * (uint32_t)(r * GEMM_LD + cp);
    }
    constexpr bool LAST_PARTIAL = (FRAME_BM * 16) % T != 0;  // the last slot exists only for the first threads
    const bool last_on = !LAST_PARTIAL || tid + (SLOTS - 1) * T < FRAME_BM * 16;
    float2 xf[SLOTS], xr[SLOTS], xm[SLOTS], xp[SLOTS], cwa, cwb;
    auto load_a = [&]() {  // the reads of one step's slots (two neighbouring floats per stream: one ds_read2_b32 each)
        cwa = lds_ld2(a_wa); cwb = lds_ld2(a_wb);
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            xf[i] = make_float2(lds_ld(a_fw[i]), lds_ld(a_fw[i] + 4));
            xr[i] = make_float2(lds_ld(a_rv[i]), lds_ld(a_rv[i] + 4));
            xm[i] = make_float2(lds_ld(a_hm[i]), lds_ld(a_hm[i] + 4));
            xp[i] = make_float2(lds_ld(a_hp[i]), lds_ld(a_hp[i] + 4));
        }
    };
    auto finish_a = [&](uint32_t buf_off) {  // buf_off: byte offset of the destination buffer (S tile; the D tile follows it); then the streams move on
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            if (i < SLOTS - 1 || last_on) {
                const float a0 = xf[i].x + xr[i].y, a1 = xf[i].y + xr[i].x;
                const float b0 = xm[i].y + xp[i].x, b1 = xm[i].x + xp[i].y;
                const float ya0 = cwa.x * a0, ya1 = cwa.y * a1, yb0 = cwb.x * b0, yb1 = cwb.y * b1;
                lds_st2(a_dst[i] + buf_off, make_float2(ya0 + yb0, ya1 + yb1));
                lds_st2(a_dst[i] + buf_off + 4u * TILE, make_float2(ya0 - yb0, ya1 - yb1));
            }
            a_fw[i] += 4 * GEMM_BK; a_hp[i] += 4 * GEMM_BK;
            a_rv[i] -= 4 * GEMM_BK; a_hm[i] -= 4 * GEMM_BK;
        }
        a_wa += 4 * GEMM_BK; a_wb += 4 * GEMM_BK;
    };
    auto store_w = [&](float *dst) {
        *reinterpret_cast<float4 *>(dst + wr * GEMM_LD + 4 * wq) = rw0;
        *reinterpret_cast<float4 *>(dst + (wr + WROWS) * GEMM_LD + 4 * wq) = rw1;
    };
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int kind = wn * 32 < d.n_even ? 0 : 1;  // wave-uniform: S or D tile
    rw0 = *reinterpret_cast<const float4 *>(wrow0); rw1 = *reinterpret_cast<const float4 *>(wrow1);
    __syncthreads();  // signal span and window tables complete
    load_a();
    finish_a(0);
    store_w(Ws);
    rw0 = *reinterpret_cast<const float4 *>(wrow0 + GEMM_BK); rw1 = *reinterpret_cast<const float4 *>(wrow1 + GEMM_BK);  // (nfull >= 2)
    __syncthreads();
    floatx16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][r] = 0.0f;
    const float *ap0 = As + kind * TILE + (wm * 32 + lr) * GEMM_LD + 4 * lh;
    const float *wp0 = Ws + (wn * 32 + lr) * GEMM_LD + 4 * lh;
    for (int ks = 0; ks < nfull; ks++) {
        const int cur = ks & 1;
        load_a();  // step ks + 1 (<= nfull: the tail step's tiles are staged like any other)
        mfma_ktile_full<1>(ap0 + cur * 2 * TILE, wp0 + cur * BN * GEMM_LD, acc);
        finish_a((uint32_t)((cur ^ 1) * 2 * TILE * 4));
        store_w(Ws + (cur ^ 1) * BN * GEMM_LD);
        {  // filter rows of step ks + 2; behind the last one the same rows again (never used)
            const int k2 = min(ks + 2, nfull) * GEMM_BK;
            rw0 = *reinterpret_cast<const float4 *>(wrow0 + k2); rw1 = *reinterpret_cast<const float4 *>(wrow1 + k2);
        }
        // LDS-only rendezvous (the filter rows requested a moment ago are only needed after the next step's products)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    mfma_ktile_partial<1>(ap0 + (nfull & 1) * 2 * TILE, wp0 + (nfull & 1) * BN * GEMM_LD, acc, 1);
    const int col = wn * 32 + lr;
    const int n = colmap[col];  // output channel of this lane's column (-1: padding)
    if (n >= 0) {
        const float bv = d.has_bias ? bias[n] : 0.0f;
        float *cb = C + (int64_t)b * d.c_bs + (int64_t)row0 * d.ldc + n;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int r = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (r < rows_here) cb[(int64_t)r * d.ldc] = acc[0][reg] + bv;
        }
    }
}

// Half-height form (round 4): 32 frames per block, WN waves (one per 32-column group), under 80 KB of LDS -- TWO blocks share a CU, each
// with barriers of its own, so one block's span load, staging and barrier waits run under the other's matrix instructions.  With one
// wave row every filter element is used by exactly one wave: the filter fragments come STRAIGHT from global memory (L2) into registers,
// one step ahead, from a copy the planner packs in fragment order ([column group][step][8-wide k group][lane][4]: every load is one
// coalesced KiB) -- no filter tile in LDS, no store / read of it, half the LDS traffic of a step.  Same products in the same order as the
// 64-row form: bit-identical (tested).  GemmDesc::fold_wpk says which layout W has.
template <int WN>
__global__ __launch_bounds__(64 * WN, 2) void frame_fold2p_kernel(Frame2Desc d, float *__restrict__ C, const float *__restrict__ A, const float *__restrict__ Wp,
                                                                  const float *__restrict__ bias, const float *__restrict__ wtab, const int32_t *__restrict__ colmap,
                                                                  FramePre pre) {
    extern __shared__ __align__(16) float frame_lds[];
    constexpr int BM = 32, T = 64 * WN, TILE = BM * GEMM_LD, SLOTS = (BM * 16 + T - 1) / T;
    const int tid = threadIdx.x;
    float *sig = frame_lds;
    float *tab = sig + ((d.span + 4 + 3) & ~3);  // [2][K]: wa | wb
    float *As = tab + 2 * d.K;                   // [2 buffers][S | D][32][GEMM_LD]
    const int b = blockIdx.x / d.tiles, rt = blockIdx.x - b * d.tiles;
    const int row0 = rt * BM;
    const int rows_here = min(BM, d.rows - row0);
    const float *src = A + (int64_t)b * d.a_bs + (int64_t)row0 * d.hop;
    const int count = (rows_here - 1) * d.hop + d.L;
    frame_load_span(sig, src, count, d.vec4 != 0, tid, T, pre, b);
    if (tid < 4) sig[count + tid] = 0.0f;
    for (int i = tid; i < 2 * d.K; i += T) tab[i] = wtab[i];
    const int nfull = d.L / (4 * GEMM_BK), nsteps = nfull + 1;
    uint32_t a_fw[SLOTS], a_rv[SLOTS], a_hm[SLOTS], a_hp[SLOTS], a_dst[SLOTS];
    const uint32_t sig0 = lds_offset_of(sig), as0 = lds_offset_of(As);
    const int cp = (tid & 15) * 2;
    uint32_t a_wa = lds_offset_of(tab) + 4u * cp, a_wb = a_wa + 4u * d.K;
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
        const int p = tid + i * T;
        const int r = (p >> 4) & (BM - 1);
        const int re = r < rows_here ? r : rows_here - 1;
        const uint32_t f = sig0 + 4u * (uint32_t)(re * d.hop);
        a_fw[i] = f + 4u * cp;
        a_rv[i] = f + 4u * (d.L - cp - 1);
        a_hm[i] = f + 4u * (d.L / 2 - cp - 1);
        a_hp[i] = f + 4u * (d.L / 2 + cp);
        a_dst[i] = as0 + 4u * (uint32_t)(r * GEMM_LD + cp);
    }
    constexpr bool LAST_PARTIAL = (BM * 16) % T != 0;
    const bool last_on = !LAST_PARTIAL || tid + (SLOTS - 1) * T < BM * 16;
    float2 xf[SLOTS], xr[SLOTS], xm[SLOTS], xp[SLOTS], cwa, cwb;
    auto load_a = [&]() {
        cwa = lds_ld2(a_wa); cwb = lds_ld2(a_wb);
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            xf[i] = make_float2(lds_ld(a_fw[i]), lds_ld(a_fw[i] + 4));
            xr[i] = make_float2(lds_ld(a_rv[i]), lds_ld(a_rv[i] + 4));
            xm[i] = make_float2(lds_ld(a_hm[i]), lds_ld(a_hm[i] + 4));
            xp[i] = make_float2(lds_ld(a_hp[i]), lds_ld(a_hp[i] + 4));
        }
    };
    auto finish_a = [&](uint32_t buf_off) {
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            if (i < SLOTS - 1 || last_on) {
                const float a0 = xf[i].x + xr[i].y, a1 = xf[i].y + xr[i].x;
                const float b0 = xm[i].y + xp[i].x, b1 = xm[i].x + xp[i].y;
                const float ya0 = cwa.x * a0, ya1 = cwa.y * a1, yb0 = cwb.x * b0, yb1 = cwb.y * b1;
                lds_st2(a_dst[i] + buf_off, make_float2(ya0 + yb0, ya1 + yb1));
                lds_st2(a_dst[i] + buf_off + 4u * TILE, make_float2(ya0 - yb0, ya1 - yb1));
            }
            a_fw[i] += 4 * GEMM_BK; a_hp[i] += 4 * GEMM_BK;
            a_rv[i] -= 4 * GEMM_BK; a_hm[i] -= 4 * GEMM_BK;
        }
        a_wa += 4 * GEMM_BK; a_wb += 4 * GEMM_BK;
    };
    const int wn = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int kind = wn * 32 < d.n_even ? 0 : 1;
    // this wave's filter fragments: [wn][step][g][lane] float4 (scalars, not arrays: see frame_foldh_kernel)
    const float4 *wf = reinterpret_cast<const float4 *>(Wp) + ((int64_t)wn * nsteps * 4) * 64 + lane;
    float4 w0 = wf[0], w1 = wf[64], w2 = wf[128], w3 = wf[192];
    float4 n0 = w0, n1 = w1, n2 = w2, n3 = w3;
    __syncthreads();  // signal span and window tables complete
    load_a();
    finish_a(0);
    __syncthreads();
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const float *ap0 = As + kind * TILE + lr * GEMM_LD + 4 * lh;
    for (int ks = 0; ks < nfull; ks++) {
        const int cur = ks & 1;
        {  // the fragments of step ks + 1 (<= nfull: the tail step's are there like any other)
            const float4 *wn_ = wf + (int64_t)(ks + 1) * 256;
            n0 = wn_[0]; n1 = wn_[64]; n2 = wn_[128]; n3 = wn_[192];
        }
        load_a();
        const float *ap = ap0 + cur * 2 * TILE;
        {
            const float4 a0 = *reinterpret_cast<const float4 *>(ap), a1 = *reinterpret_cast<const float4 *>(ap + 8);
            const float4 a2 = *reinterpret_cast<const float4 *>(ap + 16), a3 = *reinterpret_cast<const float4 *>(ap + 24);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w0.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w1.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w1.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w1.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, w2.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, w2.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, w2.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, w2.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.x, w3.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.y, w3.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.z, w3.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.w, w3.w, acc, 0, 0, 0);
        }
        finish_a((uint32_t)((cur ^ 1) * 2 * TILE * 4));
        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    {  // tail step: one 8-wide group (tap L/4)
        const float4 a0 = *reinterpret_cast<const float4 *>(ap0 + (nfull & 1) * 2 * TILE);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w0.w, acc, 0, 0, 0);
    }
    const int col = wn * 32 + lr;
    const int n = colmap[col];
    if (n >= 0) {
        const float bv = d.has_bias ? bias[n] : 0.0f;
        float *cb = C + (int64_t)b * d.c_bs + (int64_t)row0 * d.ldc + n;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (r < rows_here) cb[(int64_t)r * d.ldc] = acc[reg] + bv;
        }
    }
}

// ---- the quarter fold on the bf16 matrix pipe with f32-complete products (round 5; bf16x3.h has the arithmetic) --------------------
// frame_fold2p_kernel's structure -- 32 frames per block, their signal span and the two window tables resident in LDS, one wave per
// 32-column group, filter fragments straight from global memory one step ahead -- with the S / D operand tiles written as three exact
// bf16 planes each ([buffer][S | D][plane][32 rows][80 B]: 64 B of k + 16 of padding make the b128 fragment reads conflict free) and
// 2 x 6 v_mfma_f32_32x32x16_bf16 per 32-tap step in place of 16 exact-f32 instructions: 384 instead of 1024 matrix cycles, and the
// staging arithmetic (fold, window, split: ~35 vector instructions per slot) issues in their shadow instead of beside them.  The
// planner packs the filters as [column group][step][16-wide k group][plane][lane][8 bf16] (GemmDesc::fold_wpk == 2).  All K / 32 steps
// are ordinary ones: the tables and the filter rows are zero past tap L / 4.
// Arithmetic: per output one chain -- steps ascending, per step two 16-deep groups, per group the six partial products smallest terms
// first.  Other bits than the f32 forms (BN_FRAME2_B3=0 keeps those), the same tolerance of the oracle.
template <int WN>
__global__ __launch_bounds__(64 * WN, 2) void frame_fold2q_kernel(Frame2Desc d, float *__restrict__ C, const float *__restrict__ A, const b3_u32x4 *__restrict__ Wq,
                                                                  const float *__restrict__ bias, const float *__restrict__ wtab, const int32_t *__restrict__ colmap,
                                                                  FramePre pre) {
    extern __shared__ __align__(16) float frame_lds[];
    constexpr int BM = 32, T = 64 * WN, SLOTS = (BM * 16 + T - 1) / T;
    constexpr uint32_t PITCH = 80, PT = BM * PITCH, BUF = 6 * PT;  // bytes: row pitch, one plane tile, one buffer (S | D x three planes)
    const int tid = threadIdx.x;
    float *sig = frame_lds;
    float *tab = sig + ((d.span + 4 + 3) & ~3);  // [2][K]: wa | wb
    float *As = tab + 2 * d.K;                   // [2 buffers][S | D][3 planes][32][80 B]
    const int b = blockIdx.x / d.tiles, rt = blockIdx.x - b * d.tiles;
    const int row0 = rt * BM;
    const int rows_here = min(BM, d.rows - row0);
    const float *src = A + (int64_t)b * d.a_bs + (int64_t)row0 * d.hop;
    const int count = (rows_here - 1) * d.hop + d.L;
    frame_load_span(sig, src, count, d.vec4 != 0, tid, T, pre, b);
    if (tid < 4) sig[count + tid] = 0.0f;
    for (int i = tid; i < 2 * d.K; i += T) tab[i] = wtab[i];
    const int nsteps = d.K / GEMM_BK;
    uint32_t a_fw[SLOTS], a_rv[SLOTS], a_hm[SLOTS], a_hp[SLOTS], a_dst[SLOTS];
    const uint32_t sig0 = lds_offset_of(sig), as0 = lds_offset_of(As);
    const int cp = (tid & 15) * 2;
    uint32_t a_wa = lds_offset_of(tab) + 4u * cp, a_wb = a_wa + 4u * d.K;
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
        const int p = tid + i * T;
        const int r = (p >> 4) & (BM - 1);
        const int re = r < rows_here ? r : rows_here - 1;
        const uint32_t f = sig0 + 4u * (uint32_t)(re * d.hop);
        a_fw[i] = f + 4u * cp;
        a_rv[i] = f + 4u * (d.L - cp - 1);
        a_hm[i] = f + 4u * (d.L / 2 - cp - 1);
        a_hp[i] = f + 4u * (d.L / 2 + cp);
        a_dst[i] = as0 + (uint32_t)r * PITCH + 2u * cp;
    }
    constexpr bool LAST_PARTIAL = (BM * 16) % T != 0;
    const bool last_on = !LAST_PARTIAL || tid + (SLOTS - 1) * T < BM * 16;
    float2 xf[SLOTS], xr[SLOTS], xm[SLOTS], xp[SLOTS], cwa, cwb;
    auto load_a = [&]() {
        cwa = lds_ld2(a_wa); cwb = lds_ld2(a_wb);
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            xf[i] = make_float2(lds_ld(a_fw[i]), lds_ld(a_fw[i] + 4));
            xr[i] = make_float2(lds_ld(a_rv[i]), lds_ld(a_rv[i] + 4));
            xm[i] = make_float2(lds_ld(a_hm[i]), lds_ld(a_hm[i] + 4));
            xp[i] = make_float2(lds_ld(a_hp[i]), lds_ld(a_hp[i] + 4));
        }
    };
    auto top = [](float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); };
    auto pack = [](float hi16, float lo16) { return __builtin_amdgcn_perm(__float_as_uint(hi16), __float_as_uint(lo16), 0x07060302u); };
    auto st32 = [](uint32_t off, uint32_t v) { *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)off = v; };
    auto finish_a = [&](uint32_t buf_off) {
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            if (i < SLOTS - 1 || last_on) {
                const float a0 = xf[i].x + xr[i].y, a1 = xf[i].y + xr[i].x;
                const float b0 = xm[i].y + xp[i].x, b1 = xm[i].x + xp[i].y;
                const float ya0 = cwa.x * a0, ya1 = cwa.y * a1, yb0 = cwb.x * b0, yb1 = cwb.y * b1;
                const float v[4] = {ya0 + yb0, ya1 + yb1, ya0 - yb0, ya1 - yb1};  // S pair, D pair
                float r1[4], r2[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    r1[q] = v[q] - top(v[q]);
                    r2[q] = r1[q] - top(r1[q]);
                }
                const uint32_t dst = a_dst[i] + buf_off;
                st32(dst, pack(v[1], v[0]));
                st32(dst + PT, pack(r1[1], r1[0]));
                st32(dst + 2 * PT, pack(r2[1], r2[0]));
                st32(dst + 3 * PT, pack(v[3], v[2]));
                st32(dst + 4 * PT, pack(r1[3], r1[2]));
                st32(dst + 5 * PT, pack(r2[3], r2[2]));
            }
            a_fw[i] += 4 * GEMM_BK; a_hp[i] += 4 * GEMM_BK;
            a_rv[i] -= 4 * GEMM_BK; a_hm[i] -= 4 * GEMM_BK;
        }
        a_wa += 4 * GEMM_BK; a_wb += 4 * GEMM_BK;
    };
    const int wn = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int kind = wn * 32 < d.n_even ? 0 : 1;
    // this wave's filter fragments: [wn][step][g][plane][lane] x 16 bytes
    const b3_u32x4 *wf = Wq + ((int64_t)wn * nsteps * 6) * 64 + lane;
    b3_u32x4 w[6], nw[6];
#pragma unroll
    for (int q = 0; q < 6; q++) w[q] = wf[q * 64];
    __syncthreads();  // signal span and window tables complete
    load_a();
    finish_a(0);
    __syncthreads();
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const uint32_t ap0 = as0 + (uint32_t)kind * 3u * PT + (uint32_t)lr * PITCH + 16u * (uint32_t)lh;
    auto ld128 = [](uint32_t off) { return *(const __attribute__((address_space(3))) b3_u32x4 *)(uintptr_t)off; };
    auto mm32 = [](const b3_u32x4 &a, const b3_u32x4 &wv, const floatx16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b3_bf16x8, a), __builtin_bit_cast(b3_bf16x8, wv), c, 0, 0, 0);
    };
    for (int ks = 0; ks < nsteps; ks++) {
        const int cur = ks & 1;
        {  // the fragments of the next step (the last step re-reads its own: loads of this loop are unconditional)
            const b3_u32x4 *wn_ = wf + (int64_t)min(ks + 1, nsteps - 1) * 384;
#pragma unroll
            for (int q = 0; q < 6; q++) nw[q] = wn_[q * 64];
        }
        if (ks + 1 < nsteps) load_a();
        const uint32_t ap = ap0 + (uint32_t)cur * BUF;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const b3_u32x4 ah = ld128(ap + 32u * g), am = ld128(ap + 32u * g + PT), al = ld128(ap + 32u * g + 2 * PT);
            const b3_u32x4 &wh = w[3 * g], &wm = w[3 * g + 1], &wl = w[3 * g + 2];
            acc = mm32(ah, wl, acc);
            acc = mm32(al, wh, acc);
            acc = mm32(am, wm, acc);
            acc = mm32(ah, wm, acc);
            acc = mm32(am, wh, acc);
            acc = mm32(ah, wh, acc);
        }
        if (ks + 1 < nsteps) finish_a((uint32_t)(cur ^ 1) * BUF);
#pragma unroll
        for (int q = 0; q < 6; q++) w[q] = nw[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int col = wn * 32 + lr;
    const int n = colmap[col];
    if (n >= 0) {
        const float bv = d.has_bias ? bias[n] : 0.0f;
        float *cb = C + (int64_t)b * d.c_bs + (int64_t)row0 * d.ldc + n;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (r < rows_here) cb[(int64_t)r * d.ldc] = acc[reg] + bv;
        }
    }
}

// ------------------------------------------------------------------ small-M GEMM: intra-block split-K
// When the output has few 128-row tiles (late CNN stages, FC head) the kernel above leaves most
// CUs idle.  Here a block owns one 32 x BN tile and its 4 waves split K between them (k-steps
// interleaved), each wave staging its own operand slices through a private LDS region with no
// block barrier in the main loop; the four partial accumulators are then summed through LDS in a
// fixed order (deterministic) and the epilogue is shared between the waves.
#ifndef SPLITK_PF
#define SPLITK_PF 1  // k-slices per wave in flight (register prefetch depth of the split-K kernel)
#endif
template <int BN, int AVEC, int WVEC, bool GATED>
__global__ __launch_bounds__(256) void gemm_splitk_kernel(GemmDesc d, float *__restrict__ C,
                                                          const float *__restrict__ A,
                                                          const float *__restrict__ W,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ res,
                                                          const float *__restrict__ scale, int64_t total_rows) {
    constexpr int NT = BN / 32;
    constexpr int STAGE = (32 + BN) * GEMM_LD;                 // floats of operand staging per wave
    constexpr int RED = 32 * BN;                               // floats of one wave's accumulator tile
    constexpr int LDS_FLOATS = (4 * STAGE > 4 * RED) ? 4 * STAGE : 4 * RED;
    __shared__ __align__(16) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int n0 = blockIdx.y * BN;
    float *As = lds + wave * STAGE;
    float *Ws = As + 32 * GEMM_LD;

    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;

    // a pass of one wave (64 lanes) covers 64*VEC/32 rows of 32 floats
    constexpr int A_RPP = 64 * AVEC / GEMM_BK, A_IT = 32 / A_RPP;
    constexpr int W_RPP = 64 * WVEC / GEMM_BK, W_IT = BN / W_RPP;
    const int a_row = lane / (GEMM_BK / AVEC), a_col = (lane % (GEMM_BK / AVEC)) * AVEC;
    const int w_row = lane / (GEMM_BK / WVEC), w_col = (lane % (GEMM_BK / WVEC)) * WVEC;
    int64_t aoff[A_IT], soff[A_IT];
    a_offsets<AVEC, A_RPP, A_IT>(d, total_rows, row0, a_row, aoff, soff);
    const int ksteps = (d.K + GEMM_BK - 1) / GEMM_BK;
    constexpr int PF = SPLITK_PF;
    float ra[PF][A_IT * AVEC];
    float rg[PF][GATED ? A_IT * AVEC : 1];
    float rw[PF][W_IT * WVEC];
#pragma unroll
    for (int p = 0; p < PF; p++)
        if (wave + 4 * p < ksteps) {
            load_a_regs<AVEC, A_IT>(d, A, aoff, (wave + 4 * p) * GEMM_BK + a_col, ra[p]);
            if constexpr (GATED) load_gate_regs<AVEC, A_IT>(d, scale, soff, (wave + 4 * p) * GEMM_BK + a_col, rg[p]);
            load_w_regs<WVEC, W_RPP, W_IT>(d, W, n0 + w_row, (wave + 4 * p) * GEMM_BK + w_col, rw[p]);
        }
    auto sync_wave = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // (no K tail: only shapes with K % 32 == 0 are routed here, see gemm_use_splitk)
    for (int ks0 = wave; ks0 < ksteps; ks0 += 4 * PF) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int ks = ks0 + 4 * p;
            if (ks < ksteps) {  // wave-uniform
                const int k0 = ks * GEMM_BK;
                if constexpr (GATED) {
#pragma unroll
                    for (int i = 0; i < A_IT * AVEC; i++) ra[p][i] *= rg[p][i];
                }
                store_tile_regs<AVEC, A_RPP, A_IT>(As, a_row, a_col, ra[p]);
                store_tile_regs<WVEC, W_RPP, W_IT>(Ws, w_row, w_col, rw[p]);
                if (ks + 4 * PF < ksteps) {  // refill this slot with the slice PF rounds ahead
                    load_a_regs<AVEC, A_IT>(d, A, aoff, k0 + 4 * PF * GEMM_BK + a_col, ra[p]);
                    if constexpr (GATED) load_gate_regs<AVEC, A_IT>(d, scale, soff, k0 + 4 * PF * GEMM_BK + a_col, rg[p]);
                    load_w_regs<WVEC, W_RPP, W_IT>(d, W, n0 + w_row, k0 + 4 * PF * GEMM_BK + w_col, rw[p]);
                }
                sync_wave();  // the staging region is private to this wave; LDS ops of one wave complete in order
                mfma_ktile_full<NT>(As + lr * GEMM_LD + 4 * lh, Ws + lr * GEMM_LD + 4 * lh, acc);
                sync_wave();  // reads done before the next slice overwrites the region
            }
        }
    }
    // cross-wave reduction in a fixed order: red[wave][t*16+reg][lane]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) lds[wave * RED + (t * 16 + reg) * 64 + lane] = acc[t][reg];
    __syncthreads();
    // wave w finishes accumulator registers 4w .. 4w+3 of every N tile: gather the four partial
    // sums (fixed order) back into those registers, then the shared epilogue stores them
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int e = (t * 16 + reg) * 64 + lane;
            if ((reg >> 2) == wave) acc[t][reg] = ((lds[e] + lds[RED + e]) + lds[2 * RED + e]) + lds[3 * RED + e];
        }
    gemm_epilogue<NT>(d, C, bias, res, acc, row0, total_rows, n0, lr, lh, wave * 4, wave * 4 + 4);
}

// ------------------------------------------------------------------ squeeze-excite
// stage 1: block = CV*R threads (CV = C/4 channel vectors, R row lanes); thread (r, cv) walks rows
// r, r+R, ... of its split with fully coalesced float4 loads.  grid (splits, batch)
__global__ void gap_partial_kernel(GapDesc d, float *__restrict__ partial, const float *__restrict__ in, int R) {
    extern __shared__ __align__(16) float4 gsm[];
    const int CV = d.C >> 2;
    const int cv = threadIdx.x % CV, r = threadIdx.x / CV;
    const int64_t b = blockIdx.y;
    const int64_t rows_per = (d.HW + d.splits - 1) / d.splits;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per;
    const int64_t r1 = r0 + rows_per < d.HW ? r0 + rows_per : d.HW;
    const float4 *p = reinterpret_cast<const float4 *>(in + b * d.in_bs) + cv;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t row = r0 + r; row < r1; row += R) {
        const float4 v = p[row * CV];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    gsm[r * CV + cv] = acc;
    __syncthreads();
    if (r == 0) {
        for (int y = 1; y < R; y++) {
            const float4 v = gsm[y * CV + cv];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4 *>(partial + b * d.out_bs + (int64_t)blockIdx.x * d.C)[cv] = acc;
    }
}

// stage 2 (squeeze finish + reduce FC + excite FC + gate activation) in ONE launch:
// grid (blocks per sample group, ceil(batch / G)), 1024 threads.  A block serves G samples (round 3): the two weight
// matrices are the same for every sample, so each weight a block loads feeds G multiply-adds -- at batch 128 the launch
// moved 150 MB of weights from L2 to redo, per 256-channel slice of every sample, a 221 KB product; a quarter of that with
// G = 4.  Every block of a group redoes the cheap first half (the C-vector sums and the Cr hidden units, 16 waves in
// parallel) and then produces its own 256-channel slices of the gate; the work is spread over several CUs per group
// because one CU streams weights at only ~10 B/clk.  W1 is [Cr][C]; W2T is [Cr][C] (transposed at plan time).
// Weights do not depend on the data: the first W1 chunk of every wave and the W2T column of the block's first slice are
// requested BEFORE the squeeze sums are read, so their round trips overlap instead of following one another (the
// launch is a chain of dependent loads: 7.8 us for 0.6 MFLOP).
// Every sample's arithmetic (which partial sums meet in which order, the lane-strided dot products, the shuffle trees,
// the four-way split of the excite sum) is the one of the G = 1 kernel: a gate's bits do not depend on G or the batch.
template <int G>
__global__ __launch_bounds__(1024) void se_fc_kernel(SeFcDesc d, float *__restrict__ gate, const float *__restrict__ partial,
                                                     const float *__restrict__ w1, const float *__restrict__ b1,
                                                     const float *__restrict__ w2t, const float *__restrict__ b2, int batch) {
    constexpr int SE_NPRE = G == 1 ? 8 : 6;  // W2T values of a slice requested ahead, per thread (the rest of a column is loaded in the loop)
    extern __shared__ __align__(16) float ssm[];  // s[G][C] | h[G][Cr] | red[G * 1024]
    float *hid = ssm + G * d.C;
    float *red = hid + G * d.Cr;
    const int64_t bq = (int64_t)blockIdx.y * G;
    const int ng = (int)min((int64_t)G, (int64_t)batch - bq);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int nslices = (d.C + 255) / 256;
    const bool vec = (d.C & 3) == 0;
    const int CV = d.C >> 2;

    // ---- requests that do not depend on the data
    // (clamped addresses, not predicated loads: a select on a loaded value pins an s_waitcnt behind the load)
    float wpre[SE_NPRE];
    {
        const int c = min((int)blockIdx.x * 256 + cl, d.C - 1);
#pragma unroll
        for (int i = 0; i < SE_NPRE; i++) wpre[i] = w2t[(int64_t)min(q + 4 * i, d.Cr - 1) * d.C + c];
    }
    constexpr bool W1PRE = G == 1;  // (with four samples' accumulators the sixteen extra registers spill)
    float4 w1pre[4];
    if constexpr (W1PRE) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = wave + 16 * r < d.Cr ? wave + 16 * r : d.Cr - 1;
            w1pre[r] = reinterpret_cast<const float4 *>(w1 + (int64_t)j * d.C)[vec ? min(lane, CV - 1) : 0];  // (C >= 4 floats: in bounds either way)
        }
    }

    // ---- squeeze finish: P = 1024 / C thread groups share the splits (group p takes p, p+P, ...),
    // the P results are added in a fixed order (deterministic)
    const int P = d.C >= 1024 ? 1 : 1024 / d.C;
    if (d.splits == 1) {
        for (int i = threadIdx.x; i < ng * d.C; i += 1024) {
            const int g = i / d.C, c = i - g * d.C;
            ssm[i] = ((partial[(bq + g) * d.in_bs + c] + 0.f) + (0.f + 0.f)) * d.inv_hw;  // the P == 1 expression with one split
        }
    } else {
        for (int g = 0; g < ng; g++) {
            const float *pp = partial + (bq + g) * d.in_bs;
            float *sg = ssm + g * d.C;
            if (P == 1) {
                for (int c = threadIdx.x; c < d.C; c += 1024) {
                    // four partials in flight (independent accumulators, combined in a fixed order)
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    int sp = 0;
                    for (; sp + 3 < d.splits; sp += 4) {
                        a0 += pp[(int64_t)sp * d.C + c];
                        a1 += pp[(int64_t)(sp + 1) * d.C + c];
                        a2 += pp[(int64_t)(sp + 2) * d.C + c];
                        a3 += pp[(int64_t)(sp + 3) * d.C + c];
                    }
                    for (; sp < d.splits; sp++) a0 += pp[(int64_t)sp * d.C + c];
                    sg[c] = ((a0 + a1) + (a2 + a3)) * d.inv_hw;
                }
            } else {
                const int p = threadIdx.x / d.C, c = threadIdx.x - p * d.C;
                if (p < P) {
                    float a0 = 0.f, a1 = 0.f;
                    int sp = p;
                    for (; sp + P < d.splits; sp += 2 * P) {
                        a0 += pp[(int64_t)sp * d.C + c];
                        a1 += pp[(int64_t)(sp + P) * d.C + c];
                    }
                    if (sp < d.splits) a0 += pp[(int64_t)sp * d.C + c];
                    red[p * d.C + c] = a0 + a1;
                }
                __syncthreads();
                if ((int)threadIdx.x < d.C) {
                    float acc = red[threadIdx.x];
                    for (int t = 1; t < P; t++) acc += red[t * d.C + threadIdx.x];
                    sg[threadIdx.x] = acc * d.inv_hw;
                }
                if (g + 1 < ng) __syncthreads();  // red is reused by the next sample
            }
        }
    }
    __syncthreads();
    // ---- hidden units: wave w owns rows w, w+16, w+32, w+48 (then +64 ...) and streams them TOGETHER, so
    // four independent weight loads are in flight per step instead of one dependent chain per row
    // (at C=1152, Cr=48 the row-at-a-time loop cost ~25 us of pure load latency)
    if (vec) {
        const float4 *s4 = reinterpret_cast<const float4 *>(ssm);
        for (int j0 = wave; j0 < d.Cr; j0 += 64) {
            float a[4][G];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int g = 0; g < G; g++) a[r][g] = 0.f;
            const float4 *wr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = j0 + 16 * r < d.Cr ? j0 + 16 * r : d.Cr - 1;
                wr[r] = reinterpret_cast<const float4 *>(w1 + (int64_t)j * d.C);
            }
            int cv = lane;
            if (W1PRE && j0 == wave && cv < CV) {  // the chunk requested before the squeeze sums
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const float4 sv = s4[g * CV + cv];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a[r][g] = fmaf(sv.x, w1pre[r].x, a[r][g]);
                        a[r][g] = fmaf(sv.y, w1pre[r].y, a[r][g]);
                        a[r][g] = fmaf(sv.z, w1pre[r].z, a[r][g]);
                        a[r][g] = fmaf(sv.w, w1pre[r].w, a[r][g]);
                    }
                }
                cv += 64;
            }
#pragma unroll(G == 1 ? 2 : 1)
            for (; cv < CV; cv += 64) {
                float4 wv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) wv[r] = wr[r][cv];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const float4 sv = s4[g * CV + cv];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a[r][g] = fmaf(sv.x, wv[r].x, a[r][g]);
                        a[r][g] = fmaf(sv.y, wv[r].y, a[r][g]);
                        a[r][g] = fmaf(sv.z, wv[r].z, a[r][g]);
                        a[r][g] = fmaf(sv.w, wv[r].w, a[r][g]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = j0 + 16 * r;
                const float bj = (b1 && j < d.Cr) ? b1[j] : 0.f;
#pragma unroll
                for (int g = 0; g < G; g++) {
                    float acc = a[r][g];
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                    if (lane == 0 && j < d.Cr) hid[g * d.Cr + j] = act_apply(d.act1, acc + bj, d.p0_1, d.p1_1);
                }
            }
        }
    } else {
        for (int g = 0; g < ng; g++)
            for (int j = wave; j < d.Cr; j += 16) {
                const float *wr = w1 + (int64_t)j * d.C;
                float acc = 0.f;
                for (int c = lane; c < d.C; c += 64) acc = fmaf(ssm[g * d.C + c], wr[c], acc);
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                if (lane == 0) hid[g * d.Cr + j] = act_apply(d.act1, acc + (b1 ? b1[j] : 0.f), d.p0_1, d.p1_1);
            }
    }
    __syncthreads();
    // ---- excite: 256-channel slices, 4 thread groups split the Cr terms (fixed-order combine).  A block takes the
    // slices blockIdx.x, blockIdx.x + gridDim.x, ...: the launcher caps the blocks per group so that a large
    // batch does not redo the first half once per slice (the result does not depend on that split)
    for (int slice = blockIdx.x; slice < nslices; slice += gridDim.x) {
        const int c = slice * 256 + cl;
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; g++) acc[g] = 0.f;
        if (c < d.C) {
#pragma unroll
            for (int i = 0; i < SE_NPRE; i++) {
                const int j = q + 4 * i;
                if (j < d.Cr) {
#pragma unroll
                    for (int g = 0; g < G; g++) acc[g] = fmaf(hid[g * d.Cr + j], wpre[i], acc[g]);
                }
            }
#pragma unroll 4
            for (int j = q + 4 * SE_NPRE; j < d.Cr; j += 4) {
                const float w = w2t[(int64_t)j * d.C + c];
#pragma unroll
                for (int g = 0; g < G; g++) acc[g] = fmaf(hid[g * d.Cr + j], w, acc[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) red[(g * 4 + q) * 256 + cl] = acc[g];
        {   // the next slice's column, requested before the combine
            const int cn = (slice + (int)gridDim.x) * 256 + cl;
            if (slice + (int)gridDim.x < nslices) {
                const int cc = min(cn, d.C - 1);
#pragma unroll
                for (int i = 0; i < SE_NPRE; i++) wpre[i] = w2t[(int64_t)min(q + 4 * i, d.Cr - 1) * d.C + cc];
            }
        }
        __syncthreads();
        // thread group q finishes samples q, q + 4, ...
        if (c < d.C) {
            const float bc = b2 ? b2[c] : 0.f;
#pragma unroll
            for (int g0 = 0; g0 < G; g0 += 4) {
                const int g = g0 + q;
                if (g < ng) {
                    const float *rg = red + g * 1024;
                    const float v = ((rg[cl] + rg[256 + cl]) + rg[512 + cl]) + rg[768 + cl] + bc;
                    gate[(bq + g) * d.out_bs + c] = act_apply(d.act2, v, d.p0_2, d.p1_2);
                }
            }
        }
        __syncthreads();  // red is reused by the next slice
    }
}

// ------------------------------------------------------------------ squeeze-excite tail (SeTail, kernels.h)
// Called by EVERY thread of a block after the block's squeeze partials have been stored with se_store().  Returns in
// all but the last block of the sample.  `sm` is dynamic LDS the caller no longer needs (se_tail_lds_bytes).
__device__ __forceinline__ void se_store(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // sc1: write-through
__device__ __forceinline__ float se_load(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // sc1: past the L1
__device__ void se_tail(const SeTail &t, int64_t b, const float *__restrict__ partial_sample, float *__restrict__ sm) {
    const int nt = blockDim.x, tid = threadIdx.x;
    const SeFcDesc &d = t.se;
    // publish: every storing wave drains its stores, the block meets, ONE lane takes the ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *flag = reinterpret_cast<unsigned *>(sm);
    if (tid == 0) {
        uint32_t *cnt = t.counter + b * t.cnt_bs;
        const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = ticket == (unsigned)t.nblocks - 1u ? 1u : 0u;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
    __syncthreads();  // flag read by everyone before the region is reused
    float *sq = sm, *hid = sm + d.C, *red = hid + d.Cr;
    // squeeze finish (sc1 loads: the other blocks' stores bypassed their L2 lines): G thread groups share the partial rows
    // (group g takes rows g, g + G, ...: eight loads in flight per thread), the G sums are added in group order --
    // the order of every sum is fixed, whichever block happens to be last
    if (d.C <= nt) {
        const int G = max(1, min(nt / d.C, d.splits));
        const int g = tid / d.C, c = tid - g * d.C;
        if (g < G) {
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int sp = g;
            for (; sp + 7 * G < d.splits; sp += 8 * G) {
#pragma unroll
                for (int u = 0; u < 8; u++) a[u] += se_load(partial_sample + (int64_t)(sp + u * G) * d.C + c);
            }
            for (int u = 0; sp < d.splits; sp += G, u++) a[u & 7] += se_load(partial_sample + (int64_t)sp * d.C + c);
            red[g * d.C + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        __syncthreads();
        if (tid < d.C) {
            float acc = red[tid];
            for (int q = 1; q < G; q++) acc += red[q * d.C + tid];
            sq[tid] = acc * d.inv_hw;
        }
    } else {  // more channels than threads: columns tid, tid + nt, ...
        for (int cc = tid; cc < d.C; cc += nt) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int sp = 0;
            for (; sp + 3 < d.splits; sp += 4) {
                a0 += se_load(partial_sample + (int64_t)sp * d.C + cc);
                a1 += se_load(partial_sample + (int64_t)(sp + 1) * d.C + cc);
                a2 += se_load(partial_sample + (int64_t)(sp + 2) * d.C + cc);
                a3 += se_load(partial_sample + (int64_t)(sp + 3) * d.C + cc);
            }
            for (; sp < d.splits; sp++) a0 += se_load(partial_sample + (int64_t)sp * d.C + cc);
            sq[cc] = ((a0 + a1) + (a2 + a3)) * d.inv_hw;
        }
    }
    __syncthreads();
    // hidden units: a wave per row (rows w, w + nw, ...), four rows streamed together
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    for (int j0 = wave; j0 < d.Cr; j0 += 4 * nw) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const float *wr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = j0 + nw * r < d.Cr ? j0 + nw * r : d.Cr - 1;
            wr[r] = t.w1 + (int64_t)j * d.C;
        }
        for (int c = lane; c < d.C; c += 64) {
            const float sv = sq[c];
#pragma unroll
            for (int r = 0; r < 4; r++) a[r] = fmaf(sv, wr[r][c], a[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float acc = a[r];
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
            const int j = j0 + nw * r;
            if (lane == 0 && j < d.Cr) hid[j] = act_apply(d.act1, acc + (t.b1 ? t.b1[j] : 0.f), d.p0_1, d.p1_1);
        }
    }
    __syncthreads();
    // excite: thread per channel, the Cr terms in index order
    for (int c = tid; c < d.C; c += nt) {
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < d.Cr; j++) acc = fmaf(hid[j], t.w2t[(int64_t)j * d.C + c], acc);
        t.gate[b * d.out_bs + c] = act_apply(d.act2, acc + (t.b2 ? t.b2[c] : 0.f), d.p0_2, d.p1_2);
    }
}

// ------------------------------------------------------------------ fused expand + depthwise
// One block = one output tile (TOH x TOW pixels), ALL mid channels in chunks of 32:
//   0. the input halo tile ((TOH-1)*S+K) x ((TOW-1)*S+K) pixels x Cin is staged in LDS ONCE
//      (Cin <= 48: the whole K extent fits).  Pixels outside the image are zero rows and carry a
//      validity flag Vs[pixel] = 0 (1 inside).
//   per chunk of 32 mid channels:
//   1. expand on the matrix cores: rows = halo pixels (A from LDS), cols = 32 filters whose
//      B operands come STRAIGHT from global memory into registers (lane = filter, one float4 per
//      8-wide K group), prefetched one chunk ahead; the accumulators START at bias * Vs[pixel], so
//      a pixel outside the image expands to exactly act(0) = 0 -- which is what the depthwise
//      conv's zero padding of the EXPANDED tensor needs -- with no mask or bias add in the
//      epilogue and no extra K group (the planner only fuses activations with act(0) == 0);
//      interior tiles skip the flag look-ups; the m-tiles rotate over the waves from chunk to
//      chunk so the odd tile does not always load one SIMD
//   2. activation, written to LDS [pixel][32]
//   3. depthwise K x K from LDS: lane = channel (conflict free), each lane slides the window
//      along PPG consecutive pixels of one output row, so every LDS value is read once per row
//      tap instead of once per output; bias + activation, NHWC store
//   after the loop: per-tile channel sums for a following squeeze-excite (fixed order)
// The expanded tensor never exists in HBM.  grid (tiles, 1, batch), 256 threads, dynamic LDS;
// two barriers per chunk.
constexpr int MB_MAX_NG = 6;  // Cin <= 48
// In-kernel phase stamps for tools/mb_probe.cpp (which compiles this file with -DBN_MB_STAMPS); nothing in the product build.
#ifdef BN_MB_STAMPS
__device__ unsigned long long *bn_mb_stamps = nullptr;  // [blocks][32]: slot 0 wall clock (100 MHz), slots 1.. shader clock
#define MB_STAMP(i)                                                                                                              \
    do {                                                                                                                         \
        if (bn_mb_stamps && threadIdx.x == 0) {                                                                                  \
            unsigned long long *sp_ = bn_mb_stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 32;                         \
            if ((i) == 0) sp_[0] = wall_clock64();                                                                               \
            if ((i) == 31) sp_[31] = wall_clock64();                                                                             \
            else sp_[(i) + 1] = __builtin_amdgcn_s_memtime();                                                                    \
        }                                                                                                                        \
    } while (0)
#else
#define MB_STAMP(i)
#endif
template <int K, int S, bool IM2COL>
__global__ __launch_bounds__(256) void mbconv_expand_dw_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                               const float *__restrict__ w1, const float *__restrict__ b1,
                                                               const float *__restrict__ w2, const float *__restrict__ b2,
                                                               float *__restrict__ gap, SeTail tail) {
    constexpr int TOH = S == 1 ? 8 : 4, TOW = S == 1 ? 16 : 8;
    constexpr int IHT = (TOH - 1) * S + K, IWT = (TOW - 1) * S + K, HP = IHT * IWT;
    constexpr int MT = (HP + 31) / 32, MP = MT * 32;
    constexpr int TPW = (MT + 3) / 4;   // m-tiles per wave
    constexpr int PPG = TOH * TOW / 8;  // consecutive output pixels (of one row) per lane group
    constexpr int SEG = TOW / PPG;      // lane groups per output row
    constexpr int IWS = (PPG - 1) * S + K;  // input columns one lane group touches
    static_assert(TOW % PPG == 0, "a lane group must stay inside one output row");
    extern __shared__ __align__(16) float msm[];
    const int ng = (d.Cin + 7) / 8;  // 8-wide K groups holding data
    const int KS = ng * 8 + 4;       // LDS row stride (floats): (KS/4) is odd -> conflict-free b128 reads
    float *Xs = msm;             // [MP][KS]
    float *Es = Xs + MP * KS;    // [MP][32]
    float *Vs = Es + MP * 32;    // [MP] 1.0 for halo pixels inside the image, else 0.0
    float *red = Vs + MP;        // [nchunks][8][32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int ty = blockIdx.x / d.tiles_x, tx = blockIdx.x - ty * d.tiles_x;
    const int oh0 = ty * TOH, ow0 = tx * TOW;
    const int ih0 = oh0 * S - d.pt, iw0 = ow0 * S - d.pl;
    const int64_t b = blockIdx.z;
    const float *xin = in + b * d.in_bs;
    const int CV = d.Cin >> 2;  // float4 per pixel (Cin % 4 == 0)
    const int nchunks = (d.C + 31) / 32;
    // the whole halo lies inside the image (interior tiles): no validity look-ups in the expand
    const bool all_valid = ih0 >= 0 && ih0 + IHT <= d.H && iw0 >= 0 && iw0 + IWT <= d.W;

    // expand filters of a chunk: lane (lr, lh) holds columns 8g + 4lh .. +3 of filter c0 + lr for
    // every K group g.  w1 is the planner's padded repack [C][ng*8] = weights | zeros, so these are
    // plain loads with nothing depending on them until the matrix instructions.
    MB_STAMP(0);
    float4 bw[MB_MAX_NG], bnx[MB_MAX_NG];
    auto fetch_b = [&](float4 (&dst)[MB_MAX_NG], int c0) {
        const int n = c0 + lr < d.C ? c0 + lr : d.C - 1;
        const float *wr = w1 + (int64_t)n * (ng * 8) + 4 * lh;
#pragma unroll
        for (int g = 0; g < MB_MAX_NG; g++)
            if (g < ng) dst[g] = *reinterpret_cast<const float4 *>(wr + 8 * g);
    };
    fetch_b(bw, 0);

    // ---- 0. stage the halo tile
    if constexpr (IM2COL) {
        // stem: every halo pixel of the first conv's OUTPUT map becomes one im2col row of k1*k1*Cin1 input values
        // (column = (ky*k1 + kx)*Cin1 + c); taps outside the input image are the conv's own zero padding, pixels
        // outside the output map are zero rows with Vs = 0.  Clamped loads, selection at LDS-store time.
        const int KK = d.k1 * d.k1;
        const float inv_kk = 1.0f / (float)KK, inv_k1 = 1.0f / (float)d.k1;
        for (int r = tid; r < MP; r += 256) {
            const int iy = r / IWT, ix = r - iy * IWT;
            const int ih = ih0 + iy, iw = iw0 + ix;
            const bool ok = r < HP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
            Vs[r] = ok ? 1.0f : 0.0f;
            for (int k = d.Cin; k < ng * 8; k++) Xs[r * KS + k] = 0.0f;  // K padding
        }
        for (int it0 = tid; it0 < MP * KK; it0 += 256 * 4) {
            float v[4][4];
            bool okv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int it = it0 + i * 256;
                const int r = (int)(((float)it + 0.5f) * inv_kk), t = it - r * KK;  // exact for it < 2^14 (no integer division by a run-time value)
                const int ky = (int)(((float)t + 0.5f) * inv_k1), kx = t - ky * d.k1;
                const int iy = r / IWT, ix = r - iy * IWT;
                const int ih = ih0 + iy, iw = iw0 + ix;
                const int y = ih * d.s1 + ky - d.pt1, x = iw * d.s1 + kx - d.pl1;
                okv[i] = it < MP * KK && r < HP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W && y >= 0 && y < d.H1 && x >= 0 && x < d.W1;
                const int yc = y < 0 ? 0 : (y >= d.H1 ? d.H1 - 1 : y), xc = x < 0 ? 0 : (x >= d.W1 ? d.W1 - 1 : x);
                const float *px = xin + ((int64_t)yc * d.W1 + xc) * d.Cin1;
#pragma unroll
                for (int cc = 0; cc < 4; cc++) v[i][cc] = px[cc < d.Cin1 ? cc : 0];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int it = it0 + i * 256;
                if (it < MP * KK) {
                    const int r = (int)(((float)it + 0.5f) * inv_kk), t = it - r * KK;  // exact for it < 2^14 (no integer division by a run-time value)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
                        if (cc < d.Cin1) Xs[r * KS + t * d.Cin1 + cc] = okv[i] ? v[i][cc] : 0.0f;
                }
            }
        }
    } else {
        // 1x1 expand: PSTEP pixels per pass, lane = (pixel, float4 of its channels);
        // four passes' loads (clamped addresses, never predicated) are in flight together,
        // pixels outside the image are zeroed when the values go to LDS
        const int PSTEP = 256 / CV;
        const int p0 = tid / CV, cv = tid - p0 * CV;
        const bool lane_on = p0 < PSTEP;
        for (int r0 = p0; r0 < MP; r0 += 4 * PSTEP) {
            float4 xv[4];
            bool okv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i * PSTEP;
                const int iy = r / IWT, ix = r - iy * IWT;
                const int ih = ih0 + iy, iw = iw0 + ix;
                okv[i] = r < HP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
                const int ihc = ih < 0 ? 0 : (ih >= d.H ? d.H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= d.W ? d.W - 1 : iw);
                xv[i] = *reinterpret_cast<const float4 *>(xin + ((int64_t)ihc * d.W + iwc) * d.Cin + (lane_on ? cv * 4 : 0));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i * PSTEP;
                if (lane_on && r < MP) {
                    *reinterpret_cast<float4 *>(Xs + r * KS + cv * 4) = okv[i] ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cv == 0) {  // validity flag + zero K padding of this row
                        Vs[r] = okv[i] ? 1.0f : 0.0f;
                        if (CV & 1) *reinterpret_cast<float4 *>(Xs + r * KS + d.Cin) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
    }
    MB_STAMP(1);  // staging issued
    // depthwise role: lane = channel, group g8 = PPG consecutive pixels of output row oy
    const int c = tid & 31, g8 = tid >> 5;
    const int oy = g8 / SEG, ox0 = (g8 - oy * SEG) * PPG;
    const float *ebase = Es + ((oy * S) * IWT + ox0 * S) * 32 + c;
    float *ob = out + b * d.out_bs + ((int64_t)(oh0 + oy) * d.OW + ow0 + ox0) * d.C;
    const bool row_ok = oh0 + oy < d.OH;
    const bool seg_full = row_ok && ow0 + ox0 + PPG <= d.OW;

    for (int ch = 0; ch < nchunks; ch++) {
        const int c0 = ch * 32;
        // depthwise weights / bias of this chunk (consumed after the expand)
        const int cg = c0 + c;
        const bool cact = cg < d.C;
        float wd[K * K];
#pragma unroll
        for (int q = 0; q < K * K; q++) wd[q] = w2[q * d.C + (cact ? cg : d.C - 1)];
        const float bias2 = d.has_bias2 ? b2[cact ? cg : d.C - 1] : 0.0f;
        const float bv = d.has_bias1 ? b1[c0 + lr < d.C ? c0 + lr : d.C - 1] : 0.0f;
        if (ch + 1 < nchunks) fetch_b(bnx, c0 + 32);  // next chunk's filters in flight during this chunk
        __syncthreads();  // Es of the previous chunk consumed; first pass: Xs complete
        if (ch < 5) MB_STAMP(2 + 4 * ch);  // barrier A passed

        // ---- 1+2. expand -> Es
        const int wrole = (wave - ch) & 3;
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int mt = wrole + 4 * t;
            if (mt < MT) {
                // the accumulators start at the bias for pixels inside the image and at 0 outside: a pixel
                // outside the image then expands to act(0) = 0, which is what the depthwise conv's zero
                // padding of the EXPANDED tensor needs, with no mask in the epilogue and no extra K group
                floatx16 acc[1];
                if (all_valid) {
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[0][r] = bv;
                } else {
                    const float *vp = Vs + mt * 32 + 4 * lh;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[0][r] = bv * vp[(r & 3) + 8 * (r >> 2)];
                }
                const float *ap = Xs + (mt * 32 + lr) * KS + 4 * lh;
#pragma unroll
                for (int g = 0; g < MB_MAX_NG; g++)
                    if (g < ng) {
                        const float4 a4 = *reinterpret_cast<const float4 *>(ap + 8 * g);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bw[g].x, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bw[g].y, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bw[g].z, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bw[g].w, acc[0], 0, 0, 0);
                    }
                act_tile<1>(d.act1, d.p0_1, d.p1_1, acc);
                float *ep = Es + (mt * 32 + 4 * lh) * 32 + lr;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) ep[((reg & 3) + 8 * (reg >> 2)) * 32] = acc[0][reg];
            }
        }
        if (ch < 5) MB_STAMP(3 + 4 * ch);  // this wave's expand done
        __syncthreads();
        if (ch < 5) MB_STAMP(4 + 4 * ch);  // barrier B passed
        // ---- 3. depthwise from LDS, window sliding along the row
        float ov[PPG];
#pragma unroll
        for (int q = 0; q < PPG; q++) ov[q] = bias2;
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
#pragma unroll
            for (int ix = 0; ix < IWS; ix++) {
                const float v = ebase[(ky * IWT + ix) * 32];
#pragma unroll
                for (int kx = 0; kx < K; kx++)
                    if (ix - kx >= 0 && (ix - kx) % S == 0 && (ix - kx) / S < PPG)
                        ov[(ix - kx) / S] = fmaf(v, wd[ky * K + kx], ov[(ix - kx) / S]);
            }
        }
        act_array<PPG>(d.act2, d.p0_2, d.p1_2, ov);
        float sum = 0.0f;
        if (seg_full && c0 + 32 <= d.C) {  // block-uniform per lane group: no per-pixel predicates
#pragma unroll
            for (int q = 0; q < PPG; q++) {
                ob[(int64_t)q * d.C + cg] = ov[q];
                sum += ov[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPG; q++) {
                if (cact && row_ok && ow0 + ox0 + q < d.OW) {
                    ob[(int64_t)q * d.C + cg] = ov[q];
                    sum += ov[q];
                }
            }
        }
        red[(ch * 8 + g8) * 32 + c] = sum;
        if (ch < 5) MB_STAMP(5 + 4 * ch);  // depthwise + stores issued
#pragma unroll
        for (int g = 0; g < MB_MAX_NG; g++) bw[g] = bnx[g];
    }
    if (d.has_gap) {
        __syncthreads();
        for (int cc = tid; cc < d.C; cc += 256) {
            const float *rp = red + (cc >> 5) * 256 + (cc & 31);
            float t = rp[0];
#pragma unroll
            for (int y = 1; y < 8; y++) t += rp[y * 32];
            se_store(gap + b * d.gap_bs + (int64_t)blockIdx.x * d.C + cc, t);
        }
        if (tail.on) se_tail(tail, b, gap + b * d.gap_bs, msm);
    }
    MB_STAMP(30);
    MB_STAMP(31);
}

// ------------------------------------------------------------------ depthwise conv
// one lane = one output pixel x 4 channels.  grid (ceil(OH*OW*C4/256), batch)
template <int VEC>
__global__ __launch_bounds__(256) void dwconv_kernel(DwDesc d, float *__restrict__ out,
                                                     const float *__restrict__ in,
                                                     const float *__restrict__ w,
                                                     const float *__restrict__ bias) {
    const int64_t bidx = blockIdx.y;
    const uint32_t CV = (uint32_t)d.C / VEC;
    const uint32_t total = (uint32_t)d.OH * d.OW * CV;
    const float *ip = in + bidx * d.in_bs;
    float *op = out + bidx * d.out_bs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t cv = i % CV;
        const uint32_t pix = i / CV;
        const int ow = pix % d.OW, oh = pix / d.OW;
        const int c = cv * VEC;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) acc[j] = d.has_bias ? bias[c + j] : 0.0f;
        const int ih0 = oh * d.sh - d.pt, iw0 = ow * d.sw - d.pl;
        for (int ky = 0; ky < d.kh; ky++) {
            const int ih = ih0 + ky * d.dh;
            if (ih < 0 || ih >= d.H) continue;
            for (int kx = 0; kx < d.kw; kx++) {
                const int iw = iw0 + kx * d.dw;
                if (iw < 0 || iw >= d.W) continue;
                const float *px = ip + ((int64_t)ih * d.W + iw) * d.C + c;
                const float *pw = w + (ky * d.kw + kx) * d.C + c;
                if constexpr (VEC == 4) {
                    const float4 x = *reinterpret_cast<const float4 *>(px);
                    const float4 k = *reinterpret_cast<const float4 *>(pw);
                    acc[0] = fmaf(x.x, k.x, acc[0]);
                    acc[1] = fmaf(x.y, k.y, acc[1]);
                    acc[2] = fmaf(x.z, k.z, acc[2]);
                    acc[3] = fmaf(x.w, k.w, acc[3]);
                } else {
                    acc[0] = fmaf(px[0], pw[0], acc[0]);
                }
            }
        }
        float *po = op + (int64_t)pix * d.C + c;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4 *>(po) = make_float4(act_apply(d.act, acc[0], d.p0, d.p1), act_apply(d.act, acc[1], d.p0, d.p1),
                                                          act_apply(d.act, acc[2], d.p0, d.p1), act_apply(d.act, acc[3], d.p0, d.p1));
        } else {
            po[0] = act_apply(d.act, acc[0], d.p0, d.p1);
        }
    }
}

// Tiled depthwise conv: lane (r, cv) of a block computes TW adjacent output pixels (one row) for
// channels 4cv..4cv+3, loading each needed input column once per kernel row (NC = (TW-1)*S + KW
// float4 loads feed TW*KW float4 FMAs).  Optionally emits the per-block channel sums of the
// activated output (the squeeze of a squeeze-excite) with a fixed-order LDS reduction.
// block = CV * RPB lanes, grid (nblk, batch)
template <int KW, int S, int TW>
__global__ void dwconv_tiled_kernel(DwDesc d, float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ w,
                                    const float *__restrict__ bias, float *__restrict__ gap) {
    extern __shared__ __align__(16) float4 dsm[];
    constexpr int NC = (TW - 1) * S + KW;
    const int CV = d.C >> 2;
    const int cv = threadIdx.x % CV, r = threadIdx.x / CV;
    const int64_t b = blockIdx.y;
    const int OWT = (d.OW + TW - 1) / TW;
    const int tile = blockIdx.x * d.rpb + r;
    const bool live = tile < d.OH * OWT;
    const int oh = live ? tile / OWT : 0;
    const int ow0 = live ? (tile - oh * OWT) * TW : 0;
    const int c = cv * 4;
    float4 acc[TW];
    const float4 bz = d.has_bias ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < TW; p++) acc[p] = bz;
    if (live) {
        const float *ip = in + b * d.in_bs + c;
        const int iw0 = ow0 * S - d.pl;
        for (int ky = 0; ky < d.kh; ky++) {
            const int ih = oh * d.sh - d.pt + ky;
            if (ih < 0 || ih >= d.H) continue;
            float4 wv[KW];
#pragma unroll
            for (int kx = 0; kx < KW; kx++) wv[kx] = *reinterpret_cast<const float4 *>(w + (ky * KW + kx) * d.C + c);
            float4 xv[NC];
            const float *rowp = ip + (int64_t)ih * d.W * d.C;
#pragma unroll
            for (int j = 0; j < NC; j++) {
                const int iw = iw0 + j;
                xv[j] = (iw >= 0 && iw < d.W) ? *reinterpret_cast<const float4 *>(rowp + (int64_t)iw * d.C) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int p = 0; p < TW; p++)
#pragma unroll
                for (int kx = 0; kx < KW; kx++) {
                    const float4 x = xv[p * S + kx];
                    acc[p].x = fmaf(x.x, wv[kx].x, acc[p].x);
                    acc[p].y = fmaf(x.y, wv[kx].y, acc[p].y);
                    acc[p].z = fmaf(x.z, wv[kx].z, acc[p].z);
                    acc[p].w = fmaf(x.w, wv[kx].w, acc[p].w);
                }
        }
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        float *op = out + b * d.out_bs + ((int64_t)oh * d.OW + ow0) * d.C + c;
#pragma unroll
        for (int p = 0; p < TW; p++) {
            if (ow0 + p < d.OW) {
                float4 v = acc[p];
                v.x = act_apply(d.act, v.x, d.p0, d.p1);
                v.y = act_apply(d.act, v.y, d.p0, d.p1);
                v.z = act_apply(d.act, v.z, d.p0, d.p1);
                v.w = act_apply(d.act, v.w, d.p0, d.p1);
                *reinterpret_cast<float4 *>(op + (int64_t)p * d.C) = v;
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
    }
    if (d.has_gap) {
        dsm[r * CV + cv] = sum;
        __syncthreads();
        if (r == 0) {
            for (int y = 1; y < d.rpb; y++) {
                const float4 v = dsm[y * CV + cv];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            reinterpret_cast<float4 *>(gap + b * d.gap_bs + (int64_t)blockIdx.x * d.C)[cv] = sum;
        }
    }
}

// Pipelined variant of the kernel above (512 threads): waves 0-3 only expand, waves 4-7 only run the depthwise
// conv, one chunk behind, from the other half of a double-buffered Es -- the matrix-core phase of chunk c+1 and the
// vector-ALU phase of chunk c overlap inside the block, with ONE barrier per chunk.  Same arithmetic, same order of
// every sum, so the results are bit-identical to mbconv_expand_dw_kernel.
template <int K, int S, bool IM2COL>
__global__ __launch_bounds__(512) void mbconv_pipe_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                               const float *__restrict__ w1, const float *__restrict__ b1,
                                                               const float *__restrict__ w2, const float *__restrict__ b2,
                                                               float *__restrict__ gap, SeTail tail) {
    constexpr int TOH = S == 1 ? 8 : 4, TOW = S == 1 ? 16 : 8;
    constexpr int IHT = (TOH - 1) * S + K, IWT = (TOW - 1) * S + K, HP = IHT * IWT;
    constexpr int MT = (HP + 31) / 32, MP = MT * 32;
    constexpr int TPW = (MT + 3) / 4;   // m-tiles per wave
    constexpr int PPG = TOH * TOW / 8;  // consecutive output pixels (of one row) per lane group
    constexpr int SEG = TOW / PPG;      // lane groups per output row
    constexpr int IWS = (PPG - 1) * S + K;  // input columns one lane group touches
    static_assert(TOW % PPG == 0, "a lane group must stay inside one output row");
    extern __shared__ __align__(16) float msm[];
    const int ng = (d.Cin + 7) / 8;  // 8-wide K groups holding data
    const int KS = ng * 8 + 4;       // LDS row stride (floats): (KS/4) is odd -> conflict-free b128 reads
    float *Xs = msm;             // [MP][KS]
    float *Esb = Xs + MP * KS;   // [2][MP][32]: the expand waves fill one buffer while the depthwise waves read the other
    float *Vs = Esb + 2 * MP * 32;  // [MP] 1.0 for halo pixels inside the image, else 0.0
    float *red = Vs + MP;        // [nchunks][8][32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const bool is_exp = wave < 4;  // waves 0-3: expand (matrix cores); waves 4-7: depthwise (vector ALU)
    const int ew = wave & 3;
    const int ty = blockIdx.x / d.tiles_x, tx = blockIdx.x - ty * d.tiles_x;
    const int oh0 = ty * TOH, ow0 = tx * TOW;
    const int ih0 = oh0 * S - d.pt, iw0 = ow0 * S - d.pl;
    const int64_t b = blockIdx.z;
    const float *xin = in + b * d.in_bs;
    const int CV = d.Cin >> 2;  // float4 per pixel (Cin % 4 == 0)
    const int nchunks = (d.C + 31) / 32;
    // the whole halo lies inside the image (interior tiles): no validity look-ups in the expand
    const bool all_valid = ih0 >= 0 && ih0 + IHT <= d.H && iw0 >= 0 && iw0 + IWT <= d.W;

    // expand filters of a chunk: lane (lr, lh) holds columns 8g + 4lh .. +3 of filter c0 + lr for
    // every K group g.  w1 is the planner's padded repack [C][ng*8] = weights | zeros, so these are
    // plain loads with nothing depending on them until the matrix instructions.
    float4 bw[MB_MAX_NG], bnx[MB_MAX_NG];
    auto fetch_b = [&](float4 (&dst)[MB_MAX_NG], int c0) {
        const int n = c0 + lr < d.C ? c0 + lr : d.C - 1;
        const float *wr = w1 + (int64_t)n * (ng * 8) + 4 * lh;
#pragma unroll
        for (int g = 0; g < MB_MAX_NG; g++)
            if (g < ng) dst[g] = *reinterpret_cast<const float4 *>(wr + 8 * g);
    };
    if (is_exp) fetch_b(bw, 0);

    // ---- 0. stage the halo tile
    if constexpr (IM2COL) {
        // stem: every halo pixel of the first conv's OUTPUT map becomes one im2col row of k1*k1*Cin1 input values
        // (column = (ky*k1 + kx)*Cin1 + c); taps outside the input image are the conv's own zero padding, pixels
        // outside the output map are zero rows with Vs = 0.  Clamped loads, selection at LDS-store time.
        const int KK = d.k1 * d.k1;
        const float inv_kk = 1.0f / (float)KK, inv_k1 = 1.0f / (float)d.k1;
        for (int r = tid; r < MP; r += 512) {
            const int iy = r / IWT, ix = r - iy * IWT;
            const int ih = ih0 + iy, iw = iw0 + ix;
            const bool ok = r < HP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
            Vs[r] = ok ? 1.0f : 0.0f;
            for (int k = d.Cin; k < ng * 8; k++) Xs[r * KS + k] = 0.0f;  // K padding
        }
        for (int it0 = tid; it0 < MP * KK; it0 += 512 * 4) {
            float v[4][4];
            bool okv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int it = it0 + i * 512;
                const int r = (int)(((float)it + 0.5f) * inv_kk), t = it - r * KK;  // exact for it < 2^14 (no integer division by a run-time value)
                const int ky = (int)(((float)t + 0.5f) * inv_k1), kx = t - ky * d.k1;
                const int iy = r / IWT, ix = r - iy * IWT;
                const int ih = ih0 + iy, iw = iw0 + ix;
                const int y = ih * d.s1 + ky - d.pt1, x = iw * d.s1 + kx - d.pl1;
                okv[i] = it < MP * KK && r < HP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W && y >= 0 && y < d.H1 && x >= 0 && x < d.W1;
                const int yc = y < 0 ? 0 : (y >= d.H1 ? d.H1 - 1 : y), xc = x < 0 ? 0 : (x >= d.W1 ? d.W1 - 1 : x);
                const float *px = xin + ((int64_t)yc * d.W1 + xc) * d.Cin1;
#pragma unroll
                for (int cc = 0; cc < 4; cc++) v[i][cc] = px[cc < d.Cin1 ? cc : 0];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int it = it0 + i * 512;
                if (it < MP * KK) {
                    const int r = (int)(((float)it + 0.5f) * inv_kk), t = it - r * KK;  // exact for it < 2^14 (no integer division by a run-time value)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
                        if (cc < d.Cin1) Xs[r * KS + t * d.Cin1 + cc] = okv[i] ? v[i][cc] : 0.0f;
                }
            }
        }
    } else {
        // 1x1 expand: PSTEP pixels per pass, lane = (pixel, float4 of its channels);
        // four passes' loads (clamped addresses, never predicated) are in flight together,
        // pixels outside the image are zeroed when the values go to LDS
        const int PSTEP = 512 / CV;
        const int p0 = tid / CV, cv = tid - p0 * CV;
        const bool lane_on = p0 < PSTEP;
        for (int r0 = p0; r0 < MP; r0 += 4 * PSTEP) {
            float4 xv[4];
            bool okv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i * PSTEP;
                const int iy = r / IWT, ix = r - iy * IWT;
                const int ih = ih0 + iy, iw = iw0 + ix;
                okv[i] = r < HP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
                const int ihc = ih < 0 ? 0 : (ih >= d.H ? d.H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= d.W ? d.W - 1 : iw);
                xv[i] = *reinterpret_cast<const float4 *>(xin + ((int64_t)ihc * d.W + iwc) * d.Cin + (lane_on ? cv * 4 : 0));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i * PSTEP;
                if (lane_on && r < MP) {
                    *reinterpret_cast<float4 *>(Xs + r * KS + cv * 4) = okv[i] ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cv == 0) {  // validity flag + zero K padding of this row
                        Vs[r] = okv[i] ? 1.0f : 0.0f;
                        if (CV & 1) *reinterpret_cast<float4 *>(Xs + r * KS + d.Cin) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
    }
    // depthwise role: lane = channel, group g8 = PPG consecutive pixels of output row oy
    const int c = tid & 31, g8 = (tid & 255) >> 5;
    const int oy = g8 / SEG, ox0 = (g8 - oy * SEG) * PPG;
    const int eoff = ((oy * S) * IWT + ox0 * S) * 32 + c;
    float *ob = out + b * d.out_bs + ((int64_t)(oh0 + oy) * d.OW + ow0 + ox0) * d.C;
    const bool row_ok = oh0 + oy < d.OH;
    const bool seg_full = row_ok && ow0 + ox0 + PPG <= d.OW;

    // operands of a chunk are fetched one iteration ahead by the group that consumes them
    float wd[K * K], wdn[K * K], bias2 = 0.f, bias2n = 0.f, bv = 0.f, bvn = 0.f;
    auto fetch_dw = [&](int c0) {
        const int cgn = c0 + c < d.C ? c0 + c : d.C - 1;
#pragma unroll
        for (int q = 0; q < K * K; q++) wdn[q] = w2[q * d.C + cgn];
        bias2n = d.has_bias2 ? b2[cgn] : 0.0f;
    };
    if (is_exp) bvn = d.has_bias1 ? b1[lr < d.C ? lr : d.C - 1] : 0.0f;
    else fetch_dw(0);
    __syncthreads();  // Xs / Vs complete

    // software pipeline, one barrier per iteration: iteration `it` expands chunk `it` into buffer it & 1 (waves 0-3)
    // while the depthwise conv of chunk it - 1 runs from the other buffer (waves 4-7)
    for (int it = 0; it <= nchunks; it++) {
        if (is_exp) {
            if (it < nchunks) {
                const int c0 = it * 32;
                bv = bvn;
                if (it + 1 < nchunks) {
                    fetch_b(bnx, c0 + 32);
                    bvn = d.has_bias1 ? b1[c0 + 32 + lr < d.C ? c0 + 32 + lr : d.C - 1] : 0.0f;
                }
                float *Es = Esb + (it & 1) * MP * 32;
                const int wrole = (ew - it) & 3;
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    const int mt = wrole + 4 * t;
                    if (mt < MT) {
                        floatx16 acc[1];
                        if (all_valid) {
#pragma unroll
                            for (int r = 0; r < 16; r++) acc[0][r] = bv;
                        } else {
                            const float *vp = Vs + mt * 32 + 4 * lh;
#pragma unroll
                            for (int r = 0; r < 16; r++) acc[0][r] = bv * vp[(r & 3) + 8 * (r >> 2)];
                        }
                        const float *ap = Xs + (mt * 32 + lr) * KS + 4 * lh;
#pragma unroll
                        for (int g = 0; g < MB_MAX_NG; g++)
                            if (g < ng) {
                                const float4 a4 = *reinterpret_cast<const float4 *>(ap + 8 * g);
                                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bw[g].x, acc[0], 0, 0, 0);
                                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bw[g].y, acc[0], 0, 0, 0);
                                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bw[g].z, acc[0], 0, 0, 0);
                                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bw[g].w, acc[0], 0, 0, 0);
                            }
                        act_tile<1>(d.act1, d.p0_1, d.p1_1, acc);
                        float *ep = Es + (mt * 32 + 4 * lh) * 32 + lr;
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) ep[((reg & 3) + 8 * (reg >> 2)) * 32] = acc[0][reg];
                    }
                }
#pragma unroll
                for (int g = 0; g < MB_MAX_NG; g++) bw[g] = bnx[g];
            }
        } else if (it >= 1) {
            const int ch = it - 1;
            const int c0 = ch * 32;
            const int cg = c0 + c;
            const bool cact = cg < d.C;
#pragma unroll
            for (int q = 0; q < K * K; q++) wd[q] = wdn[q];
            bias2 = bias2n;
            if (ch + 1 < nchunks) fetch_dw(c0 + 32);
            const float *ebase = Esb + (ch & 1) * MP * 32 + eoff;
            float ov[PPG];
#pragma unroll
            for (int q = 0; q < PPG; q++) ov[q] = bias2;
#pragma unroll
            for (int ky = 0; ky < K; ky++) {
#pragma unroll
                for (int ix = 0; ix < IWS; ix++) {
                    const float v = ebase[(ky * IWT + ix) * 32];
#pragma unroll
                    for (int kx = 0; kx < K; kx++)
                        if (ix - kx >= 0 && (ix - kx) % S == 0 && (ix - kx) / S < PPG)
                            ov[(ix - kx) / S] = fmaf(v, wd[ky * K + kx], ov[(ix - kx) / S]);
                }
            }
            act_array<PPG>(d.act2, d.p0_2, d.p1_2, ov);
            float sum = 0.0f;
            if (seg_full && c0 + 32 <= d.C) {
#pragma unroll
                for (int q = 0; q < PPG; q++) {
                    ob[(int64_t)q * d.C + cg] = ov[q];
                    sum += ov[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < PPG; q++) {
                    if (cact && row_ok && ow0 + ox0 + q < d.OW) {
                        ob[(int64_t)q * d.C + cg] = ov[q];
                        sum += ov[q];
                    }
                }
            }
            red[(ch * 8 + g8) * 32 + c] = sum;
        }
        __syncthreads();
    }
    if (d.has_gap) {
        for (int cc = tid; cc < d.C; cc += 512) {
            const float *rp = red + (cc >> 5) * 256 + (cc & 31);
            float t = rp[0];
#pragma unroll
            for (int y = 1; y < 8; y++) t += rp[y * 32];
            se_store(gap + b * d.gap_bs + (int64_t)blockIdx.x * d.C + cc, t);
        }
        if (tail.on) se_tail(tail, b, gap + b * d.gap_bs, msm);
    }
}


// Depthwise K x K conv of a whole [H*W][32-channel] slab held in LDS (after a barrier): lane =
// channel, 8 lane groups take segments of 8 consecutive pixels of an output row and slide the
// window along them; bias + activation, NHWC store, complete per-channel sums (fixed order).
template <int K, int S>
__device__ __forceinline__ void dw_map_from_lds(const float *In, float *red, const float (&wd)[K * K], float bz, int H, int W, int OH, int OW,
                                                int pt, int pl, int C, int act, float p0, float p1, float *__restrict__ ob_sample, int cg,
                                                bool cact, float *__restrict__ gap_sample) {
    constexpr int PPG = 8, IWS = (PPG - 1) * S + K;
    const int c = threadIdx.x & 31, g8 = threadIdx.x >> 5;
    const int nsx = (OW + PPG - 1) / PPG;
    const int nseg = OH * nsx;
    float *ob = ob_sample + cg;
    float sum = 0.0f;
    for (int seg = g8; seg < nseg; seg += 8) {
        const int oy = seg / nsx, ox0 = (seg - oy * nsx) * PPG;
        const int ix0 = ox0 * S - pl;
        float ov[PPG];
#pragma unroll
        for (int q = 0; q < PPG; q++) ov[q] = bz;
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            const int iy = oy * S - pt + ky;
            if (iy >= 0 && iy < H) {
                const float *rp = In + iy * W * 32 + c;
#pragma unroll
                for (int ix = 0; ix < IWS; ix++) {
                    const int xg = ix0 + ix;
                    const float v = (xg >= 0 && xg < W) ? rp[xg * 32] : 0.0f;
#pragma unroll
                    for (int kx = 0; kx < K; kx++)
                        if (ix - kx >= 0 && (ix - kx) % S == 0 && (ix - kx) / S < PPG)
                            ov[(ix - kx) / S] = fmaf(v, wd[ky * K + kx], ov[(ix - kx) / S]);
                }
            }
        }
        act_array<PPG>(act, p0, p1, ov);
#pragma unroll
        for (int q = 0; q < PPG; q++) {
            if (cact && ox0 + q < OW) {
                ob[((int64_t)oy * OW + ox0 + q) * C] = ov[q];
                sum += ov[q];
            }
        }
    }
    if (gap_sample) {
        red[g8 * 32 + c] = sum;
        __syncthreads();
        if (g8 == 0 && cact) {
            float t = red[c];
#pragma unroll
            for (int y = 1; y < 8; y++) t += red[y * 32 + c];
            se_store(gap_sample + cg, t);
        }
    }
}

// Depthwise conv of a SMALL feature map (H*W <= 768): one block = (32 channels, one sample).
//   1. the whole [H*W][32-channel] slab goes to LDS with coalesced float4 loads, 8 in flight
//   2. lane = channel (conflict-free LDS reads), 8 lane groups take segments of 8 consecutive
//      pixels of an output row and slide the K x K window along them
//   3. bias + activation, NHWC store (128 B per pixel), and the COMPLETE per-channel sums of the
//      activated output for a following squeeze-excite (fixed order; no partials to add up)
// grid (ceil(C/32), batch), 256 threads, dynamic LDS (H*W*32 + 256 floats).
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_map_kernel(DwDesc d, float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ gap, SeTail tail) {
    extern __shared__ __align__(16) float dmsm[];
    const int HW = d.H * d.W;
    float *In = dmsm;
    float *red = dmsm + HW * 32;
    const int c0 = blockIdx.x * 32;
    const int64_t b = blockIdx.y;
    const int tid = threadIdx.x;
    {
        const int cq = tid & 7, p0 = tid >> 3;
        const bool qok = c0 + cq * 4 < d.C;
        const float *ip = in + b * d.in_bs + c0 + (qok ? cq * 4 : 0);
        for (int pb = p0; pb < HW; pb += 32 * 8) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = pb + i * 32;
                v[i] = *reinterpret_cast<const float4 *>(ip + (int64_t)(p < HW ? p : HW - 1) * d.C);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = pb + i * 32;
                if (p < HW) *reinterpret_cast<float4 *>(In + p * 32 + cq * 4) = qok ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const int c = tid & 31;
    const int cg = c0 + c;
    const bool cact = cg < d.C;
    const int cc = cact ? cg : d.C - 1;
    float wd[K * K];
#pragma unroll
    for (int q = 0; q < K * K; q++) wd[q] = w[q * d.C + cc];
    const float bz = d.has_bias ? bias[cc] : 0.0f;
    __syncthreads();
    dw_map_from_lds<K, S>(In, red, wd, bz, d.H, d.W, d.OH, d.OW, d.pt, d.pl, d.C, d.act, d.p0, d.p1, out + b * d.out_bs, cg, cact,
                          d.has_gap ? gap + b * d.gap_bs : nullptr);
    if (d.has_gap && tail.on) se_tail(tail, b, gap + b * d.gap_bs, dmsm);
}

// ---- the same launch with the MAP SIZE AT COMPILE TIME (round 5): dwconv_map_kernel is vector-ALU bound on its bookkeeping -- PMC, Perch's
// 32 x 8 x 816 5x5 launches: 37.1 M vector instructions for 10.4 M multiply-adds, i.e. exactly the launch's 60 us at full issue rate --
// because with run-time H / W every input value carries a bounds check and an address computation.  Here the slab sits in LDS WITH its
// zero padding ([H + K - 1][W + K - 1][32 channels]), a lane group owns a band of R output rows x CW output columns, and every (input
// element, tap) -> output relation is resolved at compile time: the loads are ds_reads at immediate offsets from one per-lane base, every
// input element of the band is read once, nothing is predicated.  Multiply-adds per output in dwconv_map_kernel's order (bias, then taps
// (ky, kx) ascending); the padding's zeros are multiplied where dwconv_map_kernel skips them, which changes no value.
// Lane groups: (OH / R) * (OW / CW) == 8.  grid (ceil(C/32), batch), 256 threads.
template <int K, int S, int H, int W, int R, int CW>
__global__ __launch_bounds__(256) void dwconv_mapt_kernel(DwDesc d, float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ gap, SeTail tail) {
    constexpr int PT = (K - 1) / 2, OH = (H + 2 * PT - K) / S + 1, OW = (W + 2 * PT - K) / S + 1;
    constexpr int HP = H + K - 1, WP = W + K - 1, HW = H * W;
    constexpr int IR = (R - 1) * S + K, IC = (CW - 1) * S + K;  // input rows / columns a band reads
    static_assert(OH % R == 0 && OW % CW == 0 && (OH / R) * (OW / CW) == 8, "eight lane groups cover the output map");
    extern __shared__ __align__(16) float dmsm[];
    float *In = dmsm;                 // [HP][WP][32]
    float *red = dmsm + HP * WP * 32; // [8][32]
    const int c0 = blockIdx.x * 32;
    const int64_t b = blockIdx.y;
    const int tid = threadIdx.x;
    // the padding frame: zero (rows above / below, columns left / right of the map)
    for (int i = tid; i < HP * WP * 8; i += 256) {
        const int q = i & 7, px = i >> 3, y = px / WP, x = px - y * WP;
        if (y < PT || y >= PT + H || x < PT || x >= PT + W) *reinterpret_cast<float4 *>(In + px * 32 + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const int cq = tid & 7, p0 = tid >> 3;
        const bool qok = c0 + cq * 4 < d.C;
        const float *ip = in + b * d.in_bs + c0 + (qok ? cq * 4 : 0);
        for (int pb = p0; pb < HW; pb += 32 * 8) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = pb + i * 32;
                v[i] = *reinterpret_cast<const float4 *>(ip + (int64_t)(p < HW ? p : HW - 1) * d.C);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = pb + i * 32;
                const int y = p / W, x = p - y * W;
                if (p < HW) *reinterpret_cast<float4 *>(In + ((y + PT) * WP + x + PT) * 32 + cq * 4) = qok ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const int c = tid & 31, g8 = tid >> 5;
    const int cg = c0 + c;
    const bool cact = cg < d.C;
    const int cc = cact ? cg : d.C - 1;
    float wd[K * K];
#pragma unroll
    for (int q = 0; q < K * K; q++) wd[q] = w[q * d.C + cc];
    const float bz = d.has_bias ? bias[cc] : 0.0f;
    __syncthreads();
    constexpr int NCS = OW / CW;  // column strips
    const int oy0 = (g8 / NCS) * R, ox0 = (g8 % NCS) * CW;
    const float *base = In + ((oy0 * S) * WP + ox0 * S) * 32 + c;  // padded coordinates: output (oy, ox) reads rows oy S .. + K - 1, columns ox S .. + K - 1
    float ov[R * CW];
#pragma unroll
    for (int q = 0; q < R * CW; q++) ov[q] = bz;
#pragma unroll
    for (int iy = 0; iy < IR; iy++)
#pragma unroll
        for (int ix = 0; ix < IC; ix++) {
            const float v = base[(iy * WP + ix) * 32];
#pragma unroll
            for (int ky = 0; ky < K; ky++)
#pragma unroll
                for (int kx = 0; kx < K; kx++)
                    if (iy - ky >= 0 && (iy - ky) % S == 0 && (iy - ky) / S < R && ix - kx >= 0 && (ix - kx) % S == 0 && (ix - kx) / S < CW)
                        ov[((iy - ky) / S) * CW + (ix - kx) / S] = fmaf(v, wd[ky * K + kx], ov[((iy - ky) / S) * CW + (ix - kx) / S]);
        }
    act_array<R * CW>(d.act, d.p0, d.p1, ov);
    float sum = 0.0f;
    if (cact) {
        float *ob = out + b * d.out_bs + cg;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int q = 0; q < CW; q++) {
                ob[((int64_t)(oy0 + r) * OW + ox0 + q) * d.C] = ov[r * CW + q];
                sum += ov[r * CW + q];
            }
    }
    if (d.has_gap) {
        float *gap_sample = gap + b * d.gap_bs;
        red[g8 * 32 + c] = sum;
        __syncthreads();
        if (g8 == 0 && cact) {
            float t = red[c];
#pragma unroll
            for (int y = 1; y < 8; y++) t += red[y * 32 + c];
            se_store(gap_sample + cg, t);
        }
        if (tail.on) se_tail(tail, b, gap_sample, dmsm);
    }
}

// Fused expand + depthwise for a SMALL feature map (H*W <= 768): one block = (32 mid channels,
// one sample), no halo, nothing recomputed.
//   1. the 32 expand filters of the block go to LDS (Ws[32][K+pad], zero K padding)
//   2. expand on the matrix cores: rows = the map's pixels in m-tiles of 32, A operands straight
//      from global memory (lane = pixel, one float4 per 8-wide K group, pieces of 8 groups with the
//      next piece in flight), B from LDS; the m-tiles rotate over the waves with the block index
//   3. bias + activation -> Es[H*W][32] in LDS
//   4. depthwise from LDS (dw_map_from_lds) with the complete squeeze sums
// grid (ceil(C/32), batch), 256 threads, dynamic LDS.
template <int K, int S>
__global__ __launch_bounds__(256) void mbconv_map_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                         const float *__restrict__ w1, const float *__restrict__ b1,
                                                         const float *__restrict__ w2, const float *__restrict__ b2,
                                                         float *__restrict__ gap) {
    extern __shared__ __align__(16) float mmsm[];
    const int HW = d.H * d.W;
    const int MT = (HW + 31) / 32;
    const int ng = (d.Cin + 7) / 8;
    const int KS = ng * 8 + 4;
    float *Ws = mmsm;                 // [32][KS]
    float *Es = Ws + 32 * KS;         // [MT*32][32]
    float *red = Es + MT * 32 * 32;   // [8][32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int c0 = blockIdx.x * 32;
    const int64_t b = blockIdx.y;
    const float *xin = in + b * d.in_bs;
    const int CV = d.Cin >> 2;
    // ---- 1. filters -> LDS (rows past C repeat the last filter: their outputs are never stored)
    for (int f = tid; f < 32 * (KS >> 2); f += 256) {
        const int n = f / (KS >> 2), q = f - n * (KS >> 2);
        const int nn = c0 + n < d.C ? c0 + n : d.C - 1;
        const float4 v = *reinterpret_cast<const float4 *>(w1 + (int64_t)nn * d.Cin + 4 * (q < CV ? q : 0));
        *reinterpret_cast<float4 *>(Ws + n * KS + 4 * q) = q < CV ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bv = d.has_bias1 ? b1[c0 + lr < d.C ? c0 + lr : d.C - 1] : 0.0f;
    // depthwise weights (consumed after the expand)
    const int c = tid & 31;
    const int cg = c0 + c;
    const bool cact = cg < d.C;
    const int cc = cact ? cg : d.C - 1;
    float wd[K * K];
#pragma unroll
    for (int q = 0; q < K * K; q++) wd[q] = w2[q * d.C + cc];
    const float bz = d.has_bias2 ? b2[cc] : 0.0f;
    __syncthreads();
    // ---- 2+3. expand -> Es
    const int wrole = (wave + blockIdx.x) & 3;
    const int npiece = (ng + 7) / 8;
    for (int mt = wrole; mt < MT; mt += 4) {
        const int row = mt * 32 + lr;
        const float *ar = xin + (int64_t)(row < HW ? row : HW - 1) * d.Cin;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        float4 an[8];
        auto fetch_a = [&](float4 (&dst)[8], int piece) {
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int slot = 2 * (piece * 8 + g) + lh;  // float4 slot inside the row; the K tail re-reads slot 0 (x 0)
                dst[g] = *reinterpret_cast<const float4 *>(ar + 4 * (slot < CV ? slot : 0));
            }
        };
        fetch_a(an, 0);
        for (int piece = 0; piece < npiece; piece++) {
            float4 ac[8];
#pragma unroll
            for (int g = 0; g < 8; g++) ac[g] = an[g];
            if (piece + 1 < npiece) fetch_a(an, piece + 1);
            const float *wp = Ws + lr * KS + 4 * lh + 64 * piece;
#pragma unroll
            for (int g = 0; g < 8; g++) {
                if (piece * 8 + g < ng) {
                    const float4 b4 = *reinterpret_cast<const float4 *>(wp + 8 * g);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[g].x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[g].y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[g].z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[g].w, b4.w, acc, 0, 0, 0);
                }
            }
        }
        floatx16 at[1] = {acc};
#pragma unroll
        for (int reg = 0; reg < 16; reg++) at[0][reg] += bv;
        act_tile<1>(d.act1, d.p0_1, d.p1_1, at);
        float *ep = Es + (mt * 32 + 4 * lh) * 32 + lr;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) ep[((reg & 3) + 8 * (reg >> 2)) * 32] = at[0][reg];
    }
    __syncthreads();
    // ---- 4. depthwise + squeeze
    dw_map_from_lds<K, S>(Es, red, wd, bz, d.H, d.W, d.OH, d.OW, d.pt, d.pl, d.C, d.act2, d.p0_2, d.p1_2, out + b * d.out_bs, cg, cact,
                          d.has_gap ? gap + b * d.gap_bs : nullptr);
}

// ------------------------------------------------------------------ direct conv
// one lane = one output element (oc fastest).  grid (ceil(OH*OW*Cout/256), batch)
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvDesc d, float *__restrict__ out,
                                                          const float *__restrict__ in,
                                                          const float *__restrict__ w,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ res) {
    const int64_t bidx = blockIdx.y;
    const uint32_t total = (uint32_t)d.OH * d.OW * d.Cout;
    const int cpg = d.Cin / d.groups, opg = d.Cout / d.groups;
    const float *ip = in + bidx * d.in_bs;
    float *op = out + bidx * d.out_bs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int oc = i % d.Cout;
        const uint32_t pix = i / d.Cout;
        const int ow = pix % d.OW, oh = pix / d.OW;
        const int g = oc / opg;
        float acc = d.has_bias ? bias[oc] : 0.0f;
        const int ih0 = oh * d.sh - d.pt, iw0 = ow * d.sw - d.pl;
        for (int ky = 0; ky < d.kh; ky++) {
            const int ih = ih0 + ky * d.dh;
            if (ih < 0 || ih >= d.H) continue;
            for (int kx = 0; kx < d.kw; kx++) {
                const int iw = iw0 + kx * d.dw;
                if (iw < 0 || iw >= d.W) continue;
                const float *px = ip + ((int64_t)ih * d.W + iw) * d.Cin + g * cpg;
                const float *pw = w + (int64_t)((ky * d.kw + kx) * cpg) * d.Cout + oc;
                for (int ci = 0; ci < cpg; ci++) acc = fmaf(px[ci], pw[(int64_t)ci * d.Cout], acc);
            }
        }
        acc = act_apply(d.act, acc, d.p0, d.p1);
        if (d.has_res) acc += res[bidx * d.out_bs + i];
        op[i] = acc;
    }
}

// Dense conv with a small weight tensor (stem conv): weights [kh][kw][Cin][Cout] staged in LDS,
// one lane = one output pixel x 4 output channels.  grid (ceil(OH*OW*Cout/4/256), batch)
__global__ __launch_bounds__(256) void conv_small_kernel(ConvDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                         const float *__restrict__ w, const float *__restrict__ bias,
                                                         const float *__restrict__ res) {
    extern __shared__ __align__(16) float wsm[];
    const int wn = d.kh * d.kw * d.Cin * d.Cout;
    for (int i = threadIdx.x * 4; i < wn; i += 1024) *reinterpret_cast<float4 *>(wsm + i) = *reinterpret_cast<const float4 *>(w + i);
    __syncthreads();
    const int64_t bidx = blockIdx.y;
    const uint32_t OC4 = (uint32_t)d.Cout >> 2;
    const uint32_t total = (uint32_t)d.OH * d.OW * OC4;
    const float *ip = in + bidx * d.in_bs;
    float *op = out + bidx * d.out_bs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t o4 = i % OC4;
        const uint32_t pix = i / OC4;
        const int ow = pix % d.OW, oh = pix / d.OW;
        const int oc = o4 * 4;
        float4 acc = d.has_bias ? *reinterpret_cast<const float4 *>(bias + oc) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int ih0 = oh * d.sh - d.pt, iw0 = ow * d.sw - d.pl;
        for (int ky = 0; ky < d.kh; ky++) {
            const int ih = ih0 + ky * d.dh;
            if (ih < 0 || ih >= d.H) continue;
            for (int kx = 0; kx < d.kw; kx++) {
                const int iw = iw0 + kx * d.dw;
                if (iw < 0 || iw >= d.W) continue;
                const float *px = ip + ((int64_t)ih * d.W + iw) * d.Cin;
                const float *pw = wsm + ((ky * d.kw + kx) * d.Cin) * d.Cout + oc;
                for (int ci = 0; ci < d.Cin; ci++) {
                    const float x = px[ci];
                    const float4 k4 = *reinterpret_cast<const float4 *>(pw + ci * d.Cout);
                    acc.x = fmaf(x, k4.x, acc.x);
                    acc.y = fmaf(x, k4.y, acc.y);
                    acc.z = fmaf(x, k4.z, acc.z);
                    acc.w = fmaf(x, k4.w, acc.w);
                }
            }
        }
        acc.x = act_apply(d.act, acc.x, d.p0, d.p1);
        acc.y = act_apply(d.act, acc.y, d.p0, d.p1);
        acc.z = act_apply(d.act, acc.z, d.p0, d.p1);
        acc.w = act_apply(d.act, acc.w, d.p0, d.p1);
        const int64_t o = (int64_t)pix * d.Cout + oc;
        if (d.has_res) {
            const float4 r4 = *reinterpret_cast<const float4 *>(res + bidx * d.out_bs + o);
            acc.x += r4.x; acc.y += r4.y; acc.z += r4.z; acc.w += r4.w;
        }
        *reinterpret_cast<float4 *>(op + o) = acc;
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline unsigned cap_blocks(int64_t want, int64_t cap) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, cap)); }

}  // namespace

// every kernel of this file that may be launched with more than 64 KB of dynamic LDS (prepare_device opts them in)
void register_kernels_hip() {
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold_kernel<false>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold_kernel<true>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_foldh_kernel<2, 1>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_foldh_kernel<3, 1>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_foldh_kernel<4, 1>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_foldh_kernel<5, 1>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_foldh_kernel<2, 2>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_foldh_kernel<3, 2>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2_kernel<2>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2_kernel<3>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2_kernel<4>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2_kernel<5>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2q_kernel<2>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2q_kernel<3>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2q_kernel<4>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2q_kernel<5>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2p_kernel<2>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2p_kernel<3>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2p_kernel<4>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(frame_fold2p_kernel<5>));
#define BN_REG_KS(KERNEL)                                                     \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<3, 1>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<3, 2>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<5, 1>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<5, 2>));
#define BN_REG_KSI(KERNEL)                                                           \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<3, 1, false>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<3, 2, false>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<5, 1, false>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<5, 2, false>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<3, 1, true>));  \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<3, 2, true>));  \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<5, 1, true>));  \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(KERNEL<5, 2, true>));
    BN_REG_KS(mbconv_map_kernel)
    BN_REG_KS(dwconv_map_kernel)
#define BN_REG_DWT(K, S, H, W, R, CW) register_dynamic_lds_kernel(reinterpret_cast<const void *>(dwconv_mapt_kernel<K, S, H, W, R, CW>));
    BN_REG_DWT(3, 1, 32, 8, 4, 8) BN_REG_DWT(5, 1, 32, 8, 4, 8) BN_REG_DWT(5, 2, 32, 8, 2, 4) BN_REG_DWT(3, 2, 32, 8, 2, 4)
    BN_REG_DWT(3, 1, 16, 4, 2, 4) BN_REG_DWT(5, 1, 16, 4, 2, 4)
    BN_REG_DWT(3, 1, 8, 32, 1, 32) BN_REG_DWT(5, 1, 8, 32, 1, 32) BN_REG_DWT(5, 2, 8, 32, 1, 8) BN_REG_DWT(3, 2, 8, 32, 1, 8)
#undef BN_REG_DWT
    BN_REG_KSI(mbconv_pipe_kernel)
    BN_REG_KSI(mbconv_expand_dw_kernel)
#undef BN_REG_KS
#undef BN_REG_KSI
}

void launch_eltwise(hipStream_t s, const EltDesc &d, float *out, const float *a, const float *const (&b)[ELT_MAX_STAGES], int64_t batch) {
    if (batch <= 0 || d.per_sample <= 0) return;
    EltPtrs bp;
    for (int k = 0; k < ELT_MAX_STAGES; k++) bp.b[k] = b[k];
    // dense (row-major over all dims) out and a -> flat float4 form
    EltFlat f{};
    f.a_scalar = 1;
    for (int k = 0; k < d.nd; k++) f.a_scalar = f.a_scalar && (d.size[k] == 1 || d.sa[k] == 0);
    bool vec4 = d.per_sample % 4 == 0 && d.bo % 4 == 0 && aligned16(out) && (f.a_scalar || (d.ba % 4 == 0 && aligned16(a)));
    {
        int64_t run = 1;
        for (int k = d.nd - 1; k >= 0 && vec4; k--) {
            if (d.size[k] != 1 && (d.so[k] != run || (!f.a_scalar && d.sa[k] != run))) vec4 = false;
            run *= d.size[k];
        }
    }
    for (int k = 0; k < d.nstages && vec4; k++) {
        const EltStage &st = d.st[k];
        if (st.bin == BIN_NONE) continue;
        // operand: scalar per sample, or dense (row-major) over the trailing dims t..nd-1 and
        // broadcast over the outer ones -> period = product of the trailing sizes
        int t = d.nd;
        for (int q = 0; q < d.nd; q++)
            if (d.size[q] != 1 && st.sb[q] != 0) { t = q; break; }
        if (t == d.nd) { f.mode[k] = 0; continue; }
        int64_t run = 1;
        bool ok = true;
        for (int q = d.nd - 1; q >= t; q--) {
            if (d.size[q] != 1 && st.sb[q] != run) ok = false;
            run *= d.size[q];
        }
        if (!ok) { vec4 = false; break; }
        const int64_t period = run;
        if (period == d.per_sample) {
            f.mode[k] = 1;
            vec4 = st.bb % 4 == 0 && aligned16(b[k]);
        } else {
            f.mode[k] = 2;
            f.period[k] = (uint32_t)period;
            vec4 = (period == 1 || period == 2 || (period % 4 == 0 && st.bb % 4 == 0 && aligned16(b[k])));
        }
    }
    if (vec4) {
        dim3 grid(cap_blocks((d.per_sample / 4 + 255) / 256, 4096), (unsigned)batch);
        hipLaunchKernelGGL(elt_flat4_kernel, grid, dim3(256), 0, s, d, f, out, a, bp);
    } else {
        EltDiv dv{};
        for (int k = 0; k < d.nd; k++) {
            const uint64_t dd = (uint64_t)d.size[k];
            if (dd <= 1) { dv.mul[k] = 0; dv.shift[k] = 0; continue; }
            uint32_t sh = 0;
            while ((1ull << sh) < dd) sh++;
            dv.mul[k] = (uint32_t)(((1ull << (31 + sh)) + dd - 1) / dd);
            dv.shift[k] = sh - 1;
        }
        dim3 grid(cap_blocks((d.per_sample + 1023) / 1024, 4096), (unsigned)batch);
        hipLaunchKernelGGL(elt_strided_kernel, grid, dim3(256), 0, s, d, dv, out, a, bp);
    }
}

// min and max of contiguous chunks in one pass (ReduceDesc::pair): grid (chunks, batch), 1024 threads, float4 loads.
// Both are exact and independent of evaluation order, so the bits equal those of the two separate launches.
__global__ __launch_bounds__(1024) void minmax_chunks_kernel(ReduceDesc d, float *__restrict__ out_min, float *__restrict__ out_max,
                                                             const float *__restrict__ in) {
    __shared__ float pmin[16], pmax[16];
    const int64_t b = blockIdx.y;
    const float4 *p4 = reinterpret_cast<const float4 *>(in + b * d.bi + (int64_t)blockIdx.x * d.kin[0]);
    const uint32_t n4 = (uint32_t)(d.red >> 2);
    float lo0 = INFINITY, lo1 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY;
    uint32_t r = threadIdx.x;
    for (; r + 1024 < n4; r += 2048) {
        const float4 u = p4[r], v = p4[r + 1024];
        lo0 = fminf(fminf(lo0, fminf(u.x, u.y)), fminf(u.z, u.w)); hi0 = fmaxf(fmaxf(hi0, fmaxf(u.x, u.y)), fmaxf(u.z, u.w));
        lo1 = fminf(fminf(lo1, fminf(v.x, v.y)), fminf(v.z, v.w)); hi1 = fmaxf(fmaxf(hi1, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    if (r < n4) {
        const float4 u = p4[r];
        lo0 = fminf(fminf(lo0, fminf(u.x, u.y)), fminf(u.z, u.w)); hi0 = fmaxf(fmaxf(hi0, fmaxf(u.x, u.y)), fmaxf(u.z, u.w));
    }
    float lo = fminf(lo0, lo1), hi = fmaxf(hi0, hi1);
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_down(lo, off)); hi = fmaxf(hi, __shfl_down(hi, off)); }
    if ((threadIdx.x & 63) == 0) { pmin[threadIdx.x >> 6] = lo; pmax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) { lo = fminf(lo, pmin[w]); hi = fmaxf(hi, pmax[w]); }
        out_min[b * d.bo + blockIdx.x] = lo;
        out_max[b * d.bo2 + blockIdx.x] = hi;
    }
}

void launch_reduce(hipStream_t s, const ReduceDesc &d, float *out, const float *in, int64_t batch, float *out2) {
    if (batch <= 0) return;
    if (d.pair) {  // planner guarantees: one kept dim of contiguous chunks, chunk length % 4 == 0, 16-byte aligned rows
        if (!out2 || d.nk != 1 || d.nr != 1 || d.rin[0] != 1 || d.red % 4 || d.kin[0] % 4 || d.bi % 4 || !aligned16(in) || d.kout[0] != 1) {
            launch_error("paired min/max reduction: the input rows must be 16-byte aligned contiguous chunks (device input pointers must be 16-byte aligned)");
            return;
        }
        hipLaunchKernelGGL(minmax_chunks_kernel, dim3((unsigned)d.kept, (unsigned)batch), dim3(1024), 0, s, d, out, out2, in);
        return;
    }
    if (d.inner_kept) {
        dim3 grid((unsigned)((d.kept + 63) / 64), (unsigned)batch);
        hipLaunchKernelGGL(reduce_inner_kept_kernel, grid, dim3(64, 16), 0, s, d, out, in);
    } else {
        dim3 grid((unsigned)d.kept, (unsigned)batch);
        bool kept_aligned = true;
        for (int k = 0; k < d.nk; k++) kept_aligned = kept_aligned && d.kin[k] % 4 == 0;
        const int vec4 = d.nr == 1 && d.rin[0] == 1 && d.bi % 4 == 0 && aligned16(in) && kept_aligned;
        hipLaunchKernelGGL(reduce_row_kernel, grid, dim3(d.red >= 8192 ? 1024 : 256), 0, s, d, out, in, vec4);
    }
}

template <int BN, bool SPLITK>
static void launch_gemm_bn(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias,
                           const float *res, const float *scale, int64_t total_rows) {
    constexpr int BM = SPLITK ? 32 : GEMM_BM;
    dim3 grid((unsigned)((total_rows + BM - 1) / BM), (unsigned)((d.N + BN - 1) / BN));
    const bool w4 = (d.K % 4 == 0) && aligned16(W);
    int avec = 1;
    const bool s4 = !d.has_scale || (d.s_bs % 4 == 0 && aligned16(scale));
    if constexpr (!SPLITK) {
        if (d.fold) {  // planner guarantees: K % 32 == 0 (no K tail), 16-byte aligned W, no gate
            if (d.K % GEMM_BK || d.fold_n != 2 * d.K || !w4 || d.has_scale) {
                launch_error("folded GEMM launched with an unsupported layout");
                return;
            }
            hipLaunchKernelGGL((gemm_mfma_kernel<BN, 2, 4, false, true>), grid, dim3(256), 0, s, d, C, A, W, bias, res, scale, total_rows);
            return;
        }
    }
    if (d.K % 4 == 0 && d.lda % 4 == 0 && d.a_bs % 4 == 0 && aligned16(A) && s4) avec = 4;
    else if (d.K % 2 == 0 && d.lda % 2 == 0 && d.a_bs % 2 == 0 && (reinterpret_cast<uintptr_t>(A) & 7u) == 0) avec = 2;
    if constexpr (!SPLITK && BN == 32) {
        if (d.npost || d.out_strided) {  // absorbed elementwise chain: ungated 32-wide tiles only (gemm_accepts_post)
            if (d.has_scale || d.has_res || d.fold) {
                launch_error("GEMM post stages launched with an unsupported layout");
                return;
            }
#define BN_LAUNCH_POST(AV, WV) hipLaunchKernelGGL((gemm_mfma_kernel<32, AV, WV, false, false, true>), grid, dim3(256), 0, s, d, C, A, W, bias, res, scale, total_rows)
            if (w4) {
                if (avec == 4) BN_LAUNCH_POST(4, 4);
                else if (avec == 2) BN_LAUNCH_POST(2, 4);
                else BN_LAUNCH_POST(1, 4);
            } else {
                if (avec == 4) BN_LAUNCH_POST(4, 1);
                else if (avec == 2) BN_LAUNCH_POST(2, 1);
                else BN_LAUNCH_POST(1, 1);
            }
#undef BN_LAUNCH_POST
            return;
        }
    }
#define BN_LAUNCH2(AV, WV, G)                                                                                                  \
    do {                                                                                                                       \
        if constexpr (SPLITK) hipLaunchKernelGGL((gemm_splitk_kernel<BN, AV, WV, G>), grid, dim3(256), 0, s, d, C, A, W, bias, res, scale, total_rows); \
        else hipLaunchKernelGGL((gemm_mfma_kernel<BN, AV, WV, G>), grid, dim3(256), 0, s, d, C, A, W, bias, res, scale, total_rows); \
    } while (0)
#define BN_LAUNCH(AV, WV)                   \
    do {                                    \
        if (d.has_scale) BN_LAUNCH2(AV, WV, true); \
        else BN_LAUNCH2(AV, WV, false);     \
    } while (0)
    if (w4) {
        if (avec == 4) BN_LAUNCH(4, 4);
        else if (avec == 2) BN_LAUNCH(2, 4);
        else BN_LAUNCH(1, 4);
    } else {
        if (avec == 4) BN_LAUNCH(4, 1);
        else if (avec == 2) BN_LAUNCH(2, 1);
        else BN_LAUNCH(1, 1);
    }
#undef BN_LAUNCH
#undef BN_LAUNCH2
}

static void launch_gemm_tiled(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias,
                              const float *res, const float *scale, int64_t total_rows) {
    const int64_t mblocks = (total_rows + GEMM_BM - 1) / GEMM_BM;
    // widest N tile that still fills the 256 CUs a couple of times over
    // N tile: the least padded of {128, 96, 64, 32} that still fills the 256 CUs a couple of times over
    auto waste = [&](int bn) { return (int64_t)((d.N + bn - 1) / bn) * bn - d.N; };
    static const int force_bn = getenv("BN_FORCE_BN") ? atoi(getenv("BN_FORCE_BN")) : 0;  // experiments only
    if (force_bn == 128) return launch_gemm_bn<128, false>(s, d, C, A, W, bias, res, scale, total_rows);
    if (force_bn == 96) return launch_gemm_bn<96, false>(s, d, C, A, W, bias, res, scale, total_rows);
    if (force_bn == 64) return launch_gemm_bn<64, false>(s, d, C, A, W, bias, res, scale, total_rows);
    if (force_bn == 32) return launch_gemm_bn<32, false>(s, d, C, A, W, bias, res, scale, total_rows);
    // Measured on MI355X (tools/gemm_bench): 32-wide N tiles win or tie almost everywhere -- more,
    // smaller blocks hide the load latency better than wider tiles save operand re-reads.  Wider
    // tiles only pay once the grid is several thousand blocks deep (high-resolution expand convs).
    if (d.fold) {
        static const int fold_bn_wide = getenv("BN_FOLD_BN") ? atoi(getenv("BN_FOLD_BN")) : 0;  // experiments only
        static const int fold_bn_narrow = getenv("BN_FOLD_BN_NARROW") ? atoi(getenv("BN_FOLD_BN_NARROW")) : fold_bn_wide;
        const int fold_bn = d.N > 200 ? fold_bn_wide : fold_bn_narrow;
        if (fold_bn == 128) return launch_gemm_bn<128, false>(s, d, C, A, W, bias, res, scale, total_rows);
        if (fold_bn == 96) return launch_gemm_bn<96, false>(s, d, C, A, W, bias, res, scale, total_rows);
        if (fold_bn == 64) return launch_gemm_bn<64, false>(s, d, C, A, W, bias, res, scale, total_rows);
        if (fold_bn == 32) return launch_gemm_bn<32, false>(s, d, C, A, W, bias, res, scale, total_rows);
        // folded rows cost two loads per staged element, so the wider tile (operand rows shared by two N tiles) wins
        // with several contexts in flight: 64 / 64 41.35 k seg/s, 32 / 32 40.0 k, 96 / 64 41.1 k, 64 / 128 41.2 k
        if (d.N > 32) return launch_gemm_bn<64, false>(s, d, C, A, W, bias, res, scale, total_rows);
        return launch_gemm_bn<32, false>(s, d, C, A, W, bias, res, scale, total_rows);
    }
    static const int deep_bn = getenv("BN_DEEPK_BN") ? atoi(getenv("BN_DEEPK_BN")) : 0;  // experiments: N tile for K >= 1024
    if (deep_bn && d.K >= 1024) {
        if (deep_bn == 128) return launch_gemm_bn<128, false>(s, d, C, A, W, bias, res, scale, total_rows);
        if (deep_bn == 96) return launch_gemm_bn<96, false>(s, d, C, A, W, bias, res, scale, total_rows);
        if (deep_bn == 64) return launch_gemm_bn<64, false>(s, d, C, A, W, bias, res, scale, total_rows);
    }
    // (deep K, the DFT framing convs: 64-wide N tiles give +1% with four contexts in flight but cost 8% of the
    // kernel's own time -- 98 -> 110 us, 105 -> 94 TF/s -- so the 32-wide tiles stay; BN_DEEPK_BN=64 to compare)
    const int64_t blocks32 = mblocks * ((d.N + 31) / 32);
    if (blocks32 > 6000 && d.N > 64 && waste(96) < waste(128) && waste(96) <= waste(64)) launch_gemm_bn<96, false>(s, d, C, A, W, bias, res, scale, total_rows);
    else if (blocks32 > 6000 && d.N > 64 && waste(128) <= waste(64)) launch_gemm_bn<128, false>(s, d, C, A, W, bias, res, scale, total_rows);
    else if (blocks32 > 6000 && d.N > 32) launch_gemm_bn<64, false>(s, d, C, A, W, bias, res, scale, total_rows);
    else launch_gemm_bn<32, false>(s, d, C, A, W, bias, res, scale, total_rows);
}

static void launch_gemm_splitk(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias,
                               const float *res, const float *scale, int64_t total_rows) {
    {
        // few output tiles and a deep K: 32-row tiles with the 4 waves of a block splitting K
        const int64_t m32 = (total_rows + 31) / 32;
        static const int force_bn = getenv("BN_FORCE_SPLITK_BN") ? atoi(getenv("BN_FORCE_SPLITK_BN")) : 0;  // experiments only
        if (force_bn == 96 && d.N > 64) return launch_gemm_bn<96, true>(s, d, C, A, W, bias, res, scale, total_rows);
        if (force_bn >= 64 && d.N > 32) return launch_gemm_bn<64, true>(s, d, C, A, W, bias, res, scale, total_rows);
        if (force_bn == 32) return launch_gemm_bn<32, true>(s, d, C, A, W, bias, res, scale, total_rows);
        if (d.N > 32 && m32 * ((d.N + 63) / 64) >= 512) launch_gemm_bn<64, true>(s, d, C, A, W, bias, res, scale, total_rows);
        else launch_gemm_bn<32, true>(s, d, C, A, W, bias, res, scale, total_rows);
    }
}

// pair != nullptr: the fused second product (frame_fold_kernel<true>); the caller has checked frame_fold_pair_ok
static bool launch_frame_fold(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias, int64_t batch,
                              const GemmDesc *pair = nullptr, float *C2 = nullptr, const float *W2 = nullptr, const float *bias2 = nullptr,
                              const FramePre *pre = nullptr) {
    if (!frame_fold_shape_ok(d, W)) return false;
    FramePre pre_v{};
    if (pre) pre_v = *pre;
    if (pre_v.n > 0 && pair) return false;  // (only the round-4 kernels carry the chain; BN_FRAMEH=0 leaves launches WITH a chain on them)
    FrameDesc f{};
    f.rows = (int32_t)d.rows; f.N = d.N; f.K = d.K; f.L = d.fold_n; f.hop = (int32_t)d.lda;
    f.tiles = (int32_t)((d.rows + FRAME_BM - 1) / FRAME_BM);
    f.span = (FRAME_BM - 1) * f.hop + f.L;
    f.vec4 = d.a_bs % 4 == 0 && (FRAME_BM * (int64_t)f.hop) % 4 == 0 && aligned16(A);
    f.a_bs = d.a_bs; f.ldc = d.ldc; f.c_bs = d.c_bs;
    f.sign = (float)d.fold; f.has_bias = d.has_bias;
    f.npost = d.npost; f.out_strided = d.out_strided; f.out_rs = d.out_rs; f.out_cs = d.out_cs;
    for (int q = 0; q < 4; q++) { f.post_act[q] = d.post_act[q]; f.post_p0[q] = d.post_p0[q]; f.post_p1[q] = d.post_p1[q]; }
    if (pair && (d.npost || d.out_strided)) return false;  // (the fused second product carries its own chain)
    // wave columns: the N tile (32 * WN) with the least padding among 128 and 160 (fewer, wider tiles on a tie)
    static const int force_wn = getenv("BN_FRAME_WN") ? atoi(getenv("BN_FRAME_WN")) : 0;  // experiments only
    auto padded = [&](int bn) { return (d.N + bn - 1) / bn * bn; };
    int wn = padded(160) <= padded(128) ? 5 : 4;
    if (d.N <= 96) wn = (d.N + 31) / 32 < 2 ? 2 : (d.N + 31) / 32;
    // a handful of row tiles (predict's single segment: 8): narrow N tiles give the chip more blocks -- the tile width does not
    // enter any output's arithmetic (47 -> 27 us for one segment)
    // K slices (round 4): from the tile width N ALONE would get -- three wave columns leave the SIMDs unbalanced, see frame_foldh_kernel --
    // so that a segment's bits do not depend on the batch-dependent choices around it
    const int ks = (!pair && wn == 3 && env_int("BN_FRAME_KS", 2) == 2) ? 2 : 1;
    if ((int64_t)f.tiles * batch <= 32 && d.N > 64) wn = 2;
    if (force_wn >= 2 && force_wn <= 5) wn = force_wn;
    if (pair) wn = std::max(2, (d.N + 31) / 32);  // one N tile holds the whole spectrum row (N <= 128)
    const int bn = 32 * wn;
    const size_t lds = (size_t)(((f.span + 3) & ~3) + 2 * FRAME_BM * GEMM_LD + 2 * bn * GEMM_LD) * sizeof(float);
    if (lds > 160 * 1024) return false;
    if (ks == 2 && wn > 3) return false;  // (a forced tile width the sliced instances do not cover)
    const void *fn = pair ? reinterpret_cast<const void *>(frame_fold_kernel<true>)
                   : ks == 2 ? (wn == 2 ? reinterpret_cast<const void *>(frame_foldh_kernel<2, 2>) : reinterpret_cast<const void *>(frame_foldh_kernel<3, 2>))
                   : wn == 2 ? reinterpret_cast<const void *>(frame_foldh_kernel<2, 1>)
                   : wn == 3 ? reinterpret_cast<const void *>(frame_foldh_kernel<3, 1>)
                   : wn == 4 ? reinterpret_cast<const void *>(frame_foldh_kernel<4, 1>) : reinterpret_cast<const void *>(frame_foldh_kernel<5, 1>);
    if (!ensure_dynamic_lds(fn, lds)) return false;
    // enough row tiles to fill the chip: one block walks all N tiles of its rows (span loaded once); else spread them
    const int64_t row_blocks = (int64_t)f.tiles * batch;
    const int walk_env = getenv("BN_FRAME_WALK") ? atoi(getenv("BN_FRAME_WALK")) : -1;  // tests / experiments
    const bool walk = walk_env >= 0 ? walk_env != 0 : row_blocks >= 256;
    dim3 grid((unsigned)row_blocks, (walk || pair) ? 1u : (unsigned)((d.N + bn - 1) / bn));
    GemmDesc none{};
    if (pair) hipLaunchKernelGGL(frame_fold_kernel<true>, grid, dim3(128 * wn), lds, s, f, C, A, W, bias, *pair, C2, W2, bias2);
    else if (env_int("BN_FRAMEH", 1) == 0 && ks == 1 && pre_v.n == 0)  // the round-3 form of the same launch (A/B; bit-identical)
        hipLaunchKernelGGL(frame_fold_kernel<false>, grid, dim3(128 * wn), lds, s, f, C, A, W, bias, none, nullptr, nullptr, nullptr);
    else if (ks == 2 && wn == 2) hipLaunchKernelGGL((frame_foldh_kernel<2, 2>), grid, dim3(512), lds, s, f, C, A, W, bias, pre_v);
    else if (ks == 2) hipLaunchKernelGGL((frame_foldh_kernel<3, 2>), grid, dim3(768), lds, s, f, C, A, W, bias, pre_v);
    else if (wn == 2) hipLaunchKernelGGL((frame_foldh_kernel<2, 1>), grid, dim3(256), lds, s, f, C, A, W, bias, pre_v);
    else if (wn == 3) hipLaunchKernelGGL((frame_foldh_kernel<3, 1>), grid, dim3(384), lds, s, f, C, A, W, bias, pre_v);
    else if (wn == 4) hipLaunchKernelGGL((frame_foldh_kernel<4, 1>), grid, dim3(512), lds, s, f, C, A, W, bias, pre_v);
    else hipLaunchKernelGGL((frame_foldh_kernel<5, 1>), grid, dim3(640), lds, s, f, C, A, W, bias, pre_v);
    return true;
}

void launch_gemm_fold_pair(hipStream_t s, const GemmDesc &d, const GemmDesc &d2, float *C2, const float *A, const float *W, const float *bias,
                           const float *W2, const float *bias2, int64_t batch) {
    if (batch <= 0) return;
    if (!frame_fold_pair_ok(d, d2) || !launch_frame_fold(s, d, nullptr, A, W, bias, batch, &d2, C2, W2, bias2))
        launch_error("folded framing GEMM + fused product: shape outside what the planner may fuse");
}

void launch_gemm_fold2(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias, const float *wtab,
                       const int32_t *colmap, int64_t batch, const FramePre *pre) {
    if (batch <= 0) return;
    FramePre pre_v{};
    if (pre) pre_v = *pre;
    if (!frame_fold2_shape_ok(d, W) || !wtab || !colmap) {
        launch_error("quarter-folded framing GEMM: shape outside what the planner may emit");
        return;
    }
    Frame2Desc f{};
    f.rows = (int32_t)d.rows; f.N = d.N; f.K = d.K; f.L = d.fold_n; f.hop = (int32_t)d.lda;
    f.n_even = d.fold_ne; f.has_bias = d.has_bias;
    f.a_bs = d.a_bs; f.ldc = d.ldc; f.c_bs = d.c_bs;
    if (d.fold_wpk == 2) {  // (round 5) the same blocks on the bf16 matrix pipe, filters as three bf16 planes in fragment order (frame_fold2q_kernel)
        f.tiles = (int32_t)((d.rows + 31) / 32);
        f.span = 31 * f.hop + f.L;
        f.vec4 = d.a_bs % 4 == 0 && (32 * (int64_t)f.hop) % 4 == 0 && aligned16(A);
        const size_t ldsq = frame_fold2q_lds_bytes(d);
        const dim3 gridq((unsigned)((int64_t)f.tiles * batch));
        const b3_u32x4 *Wq = reinterpret_cast<const b3_u32x4 *>(W);
#define FOLD2Q_GO(WN)                                                                                                       \
    do {                                                                                                                    \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(frame_fold2q_kernel<WN>), ldsq)) {                           \
            launch_error("quarter-folded framing GEMM: the device refused the dynamic-LDS opt-in");                         \
            return;                                                                                                         \
        }                                                                                                                   \
        hipLaunchKernelGGL(frame_fold2q_kernel<WN>, gridq, dim3(64 * WN), ldsq, s, f, C, A, Wq, bias, wtab, colmap, pre_v); \
    } while (0)
        if (d.N == 64) FOLD2Q_GO(2);
        else if (d.N == 96) FOLD2Q_GO(3);
        else if (d.N == 128) FOLD2Q_GO(4);
        else FOLD2Q_GO(5);
#undef FOLD2Q_GO
        return;
    }
    if (d.fold_wpk) {  // half-height blocks, filter fragments packed by the planner (frame_fold2p_kernel)
        f.tiles = (int32_t)((d.rows + 31) / 32);
        f.span = 31 * f.hop + f.L;
        f.vec4 = d.a_bs % 4 == 0 && (32 * (int64_t)f.hop) % 4 == 0 && aligned16(A);
        const size_t ldsp = frame_fold2p_lds_bytes(d);
        const dim3 gridp((unsigned)((int64_t)f.tiles * batch));
#define FOLD2P_GO(WN)                                                                                                      \
    do {                                                                                                                   \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(frame_fold2p_kernel<WN>), ldsp)) {                          \
            launch_error("quarter-folded framing GEMM: the device refused the dynamic-LDS opt-in");                        \
            return;                                                                                                        \
        }                                                                                                                  \
        hipLaunchKernelGGL(frame_fold2p_kernel<WN>, gridp, dim3(64 * WN), ldsp, s, f, C, A, W, bias, wtab, colmap, pre_v); \
    } while (0)
        if (d.N == 64) FOLD2P_GO(2);
        else if (d.N == 96) FOLD2P_GO(3);
        else if (d.N == 128) FOLD2P_GO(4);
        else FOLD2P_GO(5);
#undef FOLD2P_GO
        return;
    }
    f.tiles = (int32_t)((d.rows + FRAME_BM - 1) / FRAME_BM);
    f.span = (FRAME_BM - 1) * f.hop + f.L;
    f.vec4 = d.a_bs % 4 == 0 && (FRAME_BM * (int64_t)f.hop) % 4 == 0 && aligned16(A);
    const size_t lds = frame_fold2_lds_bytes(d);
    const dim3 grid((unsigned)((int64_t)f.tiles * batch));
#define FOLD2_GO(WN)                                                                                                   \
    do {                                                                                                               \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(frame_fold2_kernel<WN>), lds)) {                        \
            launch_error("quarter-folded framing GEMM: the device refused the dynamic-LDS opt-in");                    \
            return;                                                                                                    \
        }                                                                                                              \
        hipLaunchKernelGGL(frame_fold2_kernel<WN>, grid, dim3(128 * WN), lds, s, f, C, A, W, bias, wtab, colmap, pre_v); \
    } while (0)
    if (d.N == 64) FOLD2_GO(2);
    else if (d.N == 96) FOLD2_GO(3);
    else if (d.N == 128) FOLD2_GO(4);
    else FOLD2_GO(5);
#undef FOLD2_GO
}

void launch_gemm(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias,
                 const float *res, const float *scale, int64_t batch, const FramePre *pre) {
    if (batch <= 0) return;
    if (pre && pre->n > 0) {  // (planner rule G moves a chain into a folded GEMM only where frame_fold_post_ok holds)
        if (!(d.fold == 1 || d.fold == -1) || !launch_frame_fold(s, d, C, A, W, bias, batch, nullptr, nullptr, nullptr, nullptr, pre))
            launch_error("framing GEMM with an absorbed signal chain: the LDS-resident kernel refused the launch and no other kernel carries it");
        return;
    }
    if (d.fold == 2) {  // (its operands do not fit this signature)
        launch_error("quarter-folded framing GEMM launched without its window tables");
        return;
    }
    const int64_t total_rows = batch * d.rows;
    if (d.fold && launch_frame_fold(s, d, C, A, W, bias, batch)) return;
    if (d.fold && (d.npost || d.out_strided)) {  // (planner rule E fuses a chain into a folded GEMM only where frame_fold_post_ok holds)
        launch_error("folded framing GEMM with an absorbed chain: the LDS-resident kernel refused the launch and no other kernel carries both");
        return;
    }
    if (d.w3) {  // (round 5) the weights are the planner's three-plane bf16 image: only gemm_dma3_kernel reads that
        if (!(d.w3 == 2 ? launch_gemm_b3(s, d, C, A, W, bias, res, scale, batch) : launch_gemm_dma3(s, d, C, A, W, bias, res, scale, batch))) launch_error("GEMM with bf16x3 weights: the LDS-DMA kernel refused the launch and no other kernel reads that weight form");
        return;
    }
    if (d.npost || d.out_strided) return launch_gemm_bn<32, false>(s, d, C, A, W, bias, res, scale, total_rows);
    if (launch_gemm_dma(s, d, C, A, W, bias, res, scale, batch)) return;  // LDS-DMA kernel (gemm_dma.hip) where its tiles fit the shape
    if (d.gap) {
        launch_error("GEMM with the row mean in its epilogue: the LDS-DMA kernel refused the launch and no other kernel pools");
        return;
    }
    if (gemm_use_splitk(d)) launch_gemm_splitk(s, d, C, A, W, bias, res, scale, total_rows);
    else launch_gemm_tiled(s, d, C, A, W, bias, res, scale, total_rows);
}

void launch_gap_partial(hipStream_t s, const GapDesc &d, float *partial, const float *in, int64_t batch) {
    if (batch <= 0) return;
    const int CV = d.C / 4;
    int R = 256 / CV;
    if (R < 1) R = 1;
    dim3 grid((unsigned)d.splits, (unsigned)batch);
    hipLaunchKernelGGL(gap_partial_kernel, grid, dim3(CV * R), (size_t)CV * R * sizeof(float4), s, d, partial, in, R);
}

void launch_se_fc(hipStream_t s, const SeFcDesc &d, float *gate, float *hidden, const float *partial, const float *w1,
                  const float *b1, const float *w2, const float *b2, int64_t batch) {
    (void)hidden;
    if (batch <= 0) return;
    // samples per block: 1 by default.  BN_SEFC_G=4 lets a block serve four samples (a quarter of the weight traffic:
    // the wide layers at batch 128 take 17 us instead of 21) -- but with four samples' accumulators there is no room to
    // keep as many weight loads in flight, a launch at batch 32 takes 15 us instead of 9.5, and with four contexts the
    // two cancel (55.2 - 56.1 k against 55.6 k segments/s, one box): kept as a tested option, not the default
    const int force_g = getenv("BN_SEFC_G") ? atoi(getenv("BN_SEFC_G")) : 0;  // (read per call: the tests flip it)
    const int G = force_g == 4 ? 4 : 1;
    const int64_t groups = (batch + G - 1) / G;
    // blocks per group: one per 256-channel slice, capped so that the whole launch stays near two blocks per CU
    const int64_t nslices = (d.C + 255) / 256;
    const int64_t per_group = std::max<int64_t>(1, std::min<int64_t>(nslices, 512 / std::max<int64_t>(groups, 1)));
    const size_t lds = (size_t)(G * (d.C + d.Cr) + G * 1024) * sizeof(float);
    if (G == 4)
        hipLaunchKernelGGL(se_fc_kernel<4>, dim3((unsigned)per_group, (unsigned)groups), dim3(1024), lds, s, d, gate, partial, w1, b1, w2, b2, (int)batch);
    else
        hipLaunchKernelGGL(se_fc_kernel<1>, dim3((unsigned)per_group, (unsigned)groups), dim3(1024), lds, s, d, gate, partial, w1, b1, w2, b2, (int)batch);
}

void launch_conv(hipStream_t s, const ConvDesc &d, float *out, const float *in, const float *w, const float *bias,
                 const float *res, int64_t batch) {
    if (batch <= 0) return;
    const int64_t wfloats = (int64_t)d.kh * d.kw * d.Cin * d.Cout;
    if (d.groups == 1 && d.Cout % 4 == 0 && wfloats <= 12288 && d.out_bs % 4 == 0 && aligned16(out) && aligned16(w) &&
        (!d.has_res || aligned16(res))) {
        const int64_t total = (int64_t)d.OH * d.OW * (d.Cout / 4);
        dim3 grid(cap_blocks((total + 255) / 256, 8192), (unsigned)batch);
        hipLaunchKernelGGL(conv_small_kernel, grid, dim3(256), (size_t)wfloats * sizeof(float), s, d, out, in, w, bias, res);
        return;
    }
    const int64_t total = (int64_t)d.OH * d.OW * d.Cout;
    dim3 grid(cap_blocks((total + 255) / 256, 8192), (unsigned)batch);
    hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), 0, s, d, out, in, w, bias, res);
}

void launch_mbconv(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2,
                   const float *b2, float *gap, int64_t batch, const SeTail *tailp) {
    if (batch <= 0) return;
    if (d.row_mode && !(tailp && tailp->on)) {
        launch_mbconv_row(s, d, out, in, w1, b1, w2, b2, gap, batch);
        return;
    }
    SeTail tail{};
    if (tailp && !d.whole_map) tail = *tailp;
    tail.nblocks = d.tiles_x * d.tiles_y;
    size_t lds = mbconv_lds_bytes(d);
    if (tail.on) lds = std::max(lds, se_tail_lds_bytes(tail.se));
    if (d.whole_map == 2) {
        if (!launch_mbmap(s, d, out, in, w1, b1, w2, b2, gap, batch)) launch_error("whole-map MBConv: operands are not 16-byte aligned");
        return;
    }
    if (d.whole_map) {
        dim3 gridm((unsigned)((d.C + 31) / 32), (unsigned)batch);
#define MBM_LAUNCH(K, S)                                                                                                                           \
    do {                                                                                                                                         \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(mbconv_map_kernel<K, S>), lds)) { launch_error("kernel needs more LDS than the device grants"); break; } \
        hipLaunchKernelGGL((mbconv_map_kernel<K, S>), gridm, dim3(256), lds, s, d, out, in, w1, b1, w2, b2, gap);                                \
    } while (0)
        if (d.k == 3 && d.s == 1) MBM_LAUNCH(3, 1);
        else if (d.k == 3 && d.s == 2) MBM_LAUNCH(3, 2);
        else if (d.k == 5 && d.s == 1) MBM_LAUNCH(5, 1);
        else MBM_LAUNCH(5, 2);
#undef MBM_LAUNCH
        return;
    }
    dim3 grid((unsigned)(d.tiles_x * d.tiles_y), 1, (unsigned)batch);
    // The pipelined variant pays when the plain kernel fits only ONE block per CU (5x5 stride-1 tiles: 83 KB of LDS,
    // four waves per CU): its eight waves overlap the matrix-core and vector phases inside the block (47 -> 40 us).
    // Where two or more plain blocks fit, those already overlap each other and the plain kernel is faster.
    // BN_MBPIPE=0 never, =1 always (results are bit-identical either way).
    const int pipe_mode = getenv("BN_MBPIPE") ? atoi(getenv("BN_MBPIPE")) : -1;  // read per launch (launches are captured once per graph)
    size_t plds = mbconv_pipe_lds_bytes(d);
    if (tail.on) plds = std::max(plds, se_tail_lds_bytes(tail.se));
    const bool pipe = pipe_mode == 1 || (pipe_mode == -1 && mbconv_lds_bytes(d) > 80 * 1024);
    if (pipe && d.C > 32 && plds <= 160 * 1024) {
#define MBP_LAUNCH2(K, S, IM)                                                                                                    \
    do {                                                                                                                         \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(mbconv_pipe_kernel<K, S, IM>), plds)) { launch_error("kernel needs more LDS than the device grants"); break; } \
        hipLaunchKernelGGL((mbconv_pipe_kernel<K, S, IM>), grid, dim3(512), plds, s, d, out, in, w1, b1, w2, b2, gap, tail);     \
    } while (0)
#define MBP_LAUNCH(K, S)                       \
    do {                                       \
        if (d.k1 > 0) MBP_LAUNCH2(K, S, true); \
        else MBP_LAUNCH2(K, S, false);         \
    } while (0)
        if (d.k == 3 && d.s == 1) MBP_LAUNCH(3, 1);
        else if (d.k == 3 && d.s == 2) MBP_LAUNCH(3, 2);
        else if (d.k == 5 && d.s == 1) MBP_LAUNCH(5, 1);
        else MBP_LAUNCH(5, 2);
#undef MBP_LAUNCH
#undef MBP_LAUNCH2
        return;
    }
#define MB_LAUNCH2(K, S, IM)                                                                                                     \
    do {                                                                                                                         \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(mbconv_expand_dw_kernel<K, S, IM>), lds)) { launch_error("kernel needs more LDS than the device grants"); break; } \
        hipLaunchKernelGGL((mbconv_expand_dw_kernel<K, S, IM>), grid, dim3(256), lds, s, d, out, in, w1, b1, w2, b2, gap, tail); \
    } while (0)
#define MB_LAUNCH(K, S)                  \
    do {                                 \
        if (d.k1 > 0) MB_LAUNCH2(K, S, true); \
        else MB_LAUNCH2(K, S, false);    \
    } while (0)
    if (d.k == 3 && d.s == 1) MB_LAUNCH(3, 1);
    else if (d.k == 3 && d.s == 2) MB_LAUNCH(3, 2);
    else if (d.k == 5 && d.s == 1) MB_LAUNCH(5, 1);
    else MB_LAUNCH(5, 2);
#undef MB_LAUNCH2
#undef MB_LAUNCH
}

void launch_dwconv(hipStream_t s, const DwDesc &d, float *out, const float *in, const float *w, const float *bias, float *gap,
                   int64_t batch, const SeTail *tailp) {
    if (batch <= 0) return;
    if (d.tiled == 2) {
        dim3 grid((unsigned)((d.C + 31) / 32), (unsigned)batch);
        SeTail tail{};
        if (tailp) tail = *tailp;
        tail.nblocks = (d.C + 31) / 32;
        size_t lds = ((size_t)d.H * d.W * 32 + 256) * sizeof(float);
        if (tail.on) lds = std::max(lds, se_tail_lds_bytes(tail.se));
        // (round 5) the map sizes of the late stages of BirdNET v3.0 (8 x 32) and Perch (32 x 8, 16 x 4) at compile time
#define DWT_LAUNCH(K, S, H, W, R, CW)                                                                                                             \
    do {                                                                                                                                         \
        size_t ldst = ((size_t)(H + K - 1) * (W + K - 1) * 32 + 256) * sizeof(float);                                                            \
        if (tail.on) ldst = std::max(ldst, se_tail_lds_bytes(tail.se));                                                                          \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(dwconv_mapt_kernel<K, S, H, W, R, CW>), ldst)) { launch_error("kernel needs more LDS than the device grants"); break; } \
        hipLaunchKernelGGL((dwconv_mapt_kernel<K, S, H, W, R, CW>), grid, dim3(256), ldst, s, d, out, in, w, bias, gap, tail);                   \
        return;                                                                                                                                  \
    } while (0)
        if (d.mapt && d.kh == d.kw && d.sh == d.sw && d.dh == 1 && d.dw == 1 && d.pt == (d.kh - 1) / 2 && d.pl == (d.kw - 1) / 2 &&
            d.OH == (d.H + 2 * d.pt - d.kh) / d.sh + 1 && d.OW == (d.W + 2 * d.pl - d.kw) / d.sw + 1) {
            const int k = d.kw, st = d.sw;
            if (d.H == 32 && d.W == 8) {
                if (k == 3 && st == 1) DWT_LAUNCH(3, 1, 32, 8, 4, 8);
                if (k == 5 && st == 1) DWT_LAUNCH(5, 1, 32, 8, 4, 8);
                if (k == 5 && st == 2) DWT_LAUNCH(5, 2, 32, 8, 2, 4);
                if (k == 3 && st == 2) DWT_LAUNCH(3, 2, 32, 8, 2, 4);
            } else if (d.H == 16 && d.W == 4) {
                if (k == 3 && st == 1) DWT_LAUNCH(3, 1, 16, 4, 2, 4);
                if (k == 5 && st == 1) DWT_LAUNCH(5, 1, 16, 4, 2, 4);
            } else if (d.H == 8 && d.W == 32) {
                if (k == 3 && st == 1) DWT_LAUNCH(3, 1, 8, 32, 1, 32);
                if (k == 5 && st == 1) DWT_LAUNCH(5, 1, 8, 32, 1, 32);
                if (k == 5 && st == 2) DWT_LAUNCH(5, 2, 8, 32, 1, 8);
                if (k == 3 && st == 2) DWT_LAUNCH(3, 2, 8, 32, 1, 8);
            }
        }
#undef DWT_LAUNCH
#define DWM_LAUNCH(K, S)                                                                                                                           \
    do {                                                                                                                                         \
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(dwconv_map_kernel<K, S>), lds)) { launch_error("kernel needs more LDS than the device grants"); break; } \
        hipLaunchKernelGGL((dwconv_map_kernel<K, S>), grid, dim3(256), lds, s, d, out, in, w, bias, gap, tail);                                  \
    } while (0)
        if (d.kw == 3 && d.sw == 1) DWM_LAUNCH(3, 1);
        else if (d.kw == 3 && d.sw == 2) DWM_LAUNCH(3, 2);
        else if (d.kw == 5 && d.sw == 1) DWM_LAUNCH(5, 1);
        else DWM_LAUNCH(5, 2);
#undef DWM_LAUNCH
        return;
    }
    if (d.tiled) {
        const int CV = d.C / 4;
        dim3 grid((unsigned)d.nblk, (unsigned)batch), block((unsigned)(CV * d.rpb));
        const size_t lds = d.has_gap ? (size_t)CV * d.rpb * sizeof(float4) : 0;
#define DW_LAUNCH(KW, S) hipLaunchKernelGGL((dwconv_tiled_kernel<KW, S, 4>), grid, block, lds, s, d, out, in, w, bias, gap)
        if (d.kw == 3 && d.sw == 1) DW_LAUNCH(3, 1);
        else if (d.kw == 3 && d.sw == 2) DW_LAUNCH(3, 2);
        else if (d.kw == 5 && d.sw == 1) DW_LAUNCH(5, 1);
        else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
        return;
    }
    const bool v4 = d.C % 4 == 0 && d.in_bs % 4 == 0 && d.out_bs % 4 == 0 && aligned16(in) && aligned16(out) && aligned16(w);
    if (v4) {
        const int64_t total = (int64_t)d.OH * d.OW * (d.C / 4);
        dim3 grid(cap_blocks((total + 255) / 256, 8192), (unsigned)batch);
        hipLaunchKernelGGL(dwconv_kernel<4>, grid, dim3(256), 0, s, d, out, in, w, bias);
    } else {
        const int64_t total = (int64_t)d.OH * d.OW * d.C;
        dim3 grid(cap_blocks((total + 255) / 256, 8192), (unsigned)batch);
        hipLaunchKernelGGL(dwconv_kernel<1>, grid, dim3(256), 0, s, d, out, in, w, bias);
    }
}

// ------------------------------------------------------------------ pooling
// MaxPool / AveragePool, NHWC: one lane = one output pixel x VEC channels.  Max ignores padding taps;
// average divides by the in-image tap count unless count_include_pad.  grid (ceil(total/256) capped, batch)
namespace {
template <int VEC>
__global__ __launch_bounds__(256) void pool_kernel(PoolDesc d, float *__restrict__ out, const float *__restrict__ in) {
    const int64_t b = blockIdx.y;
    const int CV = d.C / VEC;
    const uint32_t total = (uint32_t)d.OH * d.OW * CV;
    const float *ip = in + b * d.in_bs;
    float *op = out + b * d.out_bs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int cv = i % CV;
        const uint32_t pix = i / CV;
        const int ow = pix % d.OW, oh = pix / d.OW;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = d.is_max ? -INFINITY : 0.0f;
        int taps = 0;
        for (int ky = 0; ky < d.kh; ky++) {
            const int ih = oh * d.sh - d.pt + ky;
            if (ih < 0 || ih >= d.H) continue;
            for (int kx = 0; kx < d.kw; kx++) {
                const int iw = ow * d.sw - d.pl + kx;
                if (iw < 0 || iw >= d.W) continue;
                const float *px = ip + ((int64_t)ih * d.W + iw) * d.C + cv * VEC;
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v] = d.is_max ? fmaxf(acc[v], px[v]) : acc[v] + px[v];
                taps++;
            }
        }
        const float div = d.is_max ? 1.0f : (float)(d.count_include_pad ? d.kh * d.kw : (taps > 0 ? taps : 1));
        float *po = op + (int64_t)pix * d.C + cv * VEC;
#pragma unroll
        for (int v = 0; v < VEC; v++) po[v] = d.is_max ? acc[v] : acc[v] / div;
    }
}
}  // namespace
void launch_pool(hipStream_t s, const PoolDesc &d, float *out, const float *in, int64_t batch) {
    if (batch <= 0) return;
    if (d.C % 4 == 0) {
        dim3 grid(cap_blocks(((int64_t)d.OH * d.OW * (d.C / 4) + 255) / 256, 8192), (unsigned)batch);
        hipLaunchKernelGGL(pool_kernel<4>, grid, dim3(256), 0, s, d, out, in);
    } else {
        dim3 grid(cap_blocks(((int64_t)d.OH * d.OW * d.C + 255) / 256, 8192), (unsigned)batch);
        hipLaunchKernelGGL(pool_kernel<1>, grid, dim3(256), 0, s, d, out, in);
    }
}

// ------------------------------------------------------------------ recording -> windows
// chunk_audio on the device (reference src/bin/birdnet-analyze.rs:707-743) fused with the WAV
// sample conversion (:683-687, f32::from(s) / 32768.0 -- a division by a power of two, exact):
// window w of the launch starts at sample (first + w) * step; samples past the end of the recording
// are 0.  One lane = 4 consecutive samples of a window (float4 store; S % 4 == 0 checked by the
// caller), reads are 2- or 4-byte loads at arbitrary alignment.  grid (ceil(S/1024), count)
namespace {
template <class T>
__global__ __launch_bounds__(256) void windows_kernel(float *__restrict__ dst, const T *__restrict__ src, uint64_t n_samples, uint64_t first_start,
                                                     uint64_t step, uint32_t S) {
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i >= S) return;
    const uint64_t pos = first_start + (uint64_t)blockIdx.y * step + i;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint64_t q = pos + u;
        if (q < n_samples) {
            if constexpr (sizeof(T) == 2) v[u] = (float)src[q] * (1.0f / 32768.0f);
            else v[u] = (float)src[q];
        } else {
            v[u] = 0.0f;
        }
    }
    *reinterpret_cast<float4 *>(dst + (uint64_t)blockIdx.y * S + i) = make_float4(v[0], v[1], v[2], v[3]);
}
}  // namespace
void launch_windows(hipStream_t s, float *dst, const void *src, int32_t is_i16, uint64_t n_samples, uint64_t first_start, uint64_t step, uint32_t S,
                    uint32_t count) {
    if (count == 0 || S == 0) return;
    dim3 grid((S / 4 + 255) / 256, count);
    if (is_i16) hipLaunchKernelGGL(windows_kernel<int16_t>, grid, dim3(256), 0, s, dst, static_cast<const int16_t *>(src), n_samples, first_start, step, S);
    else hipLaunchKernelGGL(windows_kernel<float>, grid, dim3(256), 0, s, dst, static_cast<const float *>(src), n_samples, first_start, step, S);
}

// ------------------------------------------------------------------ polyphase resampler
// y[n] = sum_j table[(n*M) % L][j] * x[(n*M) / L + j - (T/2 - 1)]   (zero outside the recording)
// with a host-built windowed-sinc table (L phases x T taps, each phase normalised to unit DC gain).
// One lane = one output sample; the phase rows are tiny and cache-resident, the input reads of a
// wave overlap almost entirely.  The source is int16 (/32768, exact) or f32.
namespace {
template <class T_>
__global__ __launch_bounds__(256) void resample_kernel(float *__restrict__ dst, const T_ *__restrict__ src, const float *__restrict__ table,
                                                      uint64_t n_src, uint64_t n_dst, uint32_t L, uint32_t M, uint32_t T) {
    const uint64_t n = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (n >= n_dst) return;
    const uint64_t pos = n * M;
    const uint64_t base = pos / L;
    const uint32_t phase = (uint32_t)(pos - base * L);
    const float *row = table + (uint64_t)phase * T;
    const int64_t first = (int64_t)base - (int64_t)(T / 2 - 1);
    float acc = 0.0f;
    for (uint32_t j = 0; j < T; j++) {
        const int64_t q = first + j;
        float x = 0.0f;
        if (q >= 0 && (uint64_t)q < n_src) {
            if constexpr (sizeof(T_) == 2) x = (float)src[q] * (1.0f / 32768.0f);
            else x = (float)src[q];
        }
        acc = fmaf(row[j], x, acc);
    }
    dst[n] = acc;
}
}  // namespace
void launch_resample(hipStream_t s, float *dst, const void *src, int32_t is_i16, const float *table, uint64_t n_src, uint64_t n_dst, uint32_t L, uint32_t M,
                     uint32_t T) {
    if (n_dst == 0) return;
    dim3 grid((unsigned)((n_dst + 255) / 256));
    if (is_i16) hipLaunchKernelGGL(resample_kernel<int16_t>, grid, dim3(256), 0, s, dst, static_cast<const int16_t *>(src), table, n_src, n_dst, L, M, T);
    else hipLaunchKernelGGL(resample_kernel<float>, grid, dim3(256), 0, s, dst, static_cast<const float *>(src), table, n_src, n_dst, L, M, T);
}

// empty launch: calibrates the event-to-event overhead of bn_ctx_time_kernels
namespace {
__global__ void null_kernel() {}
}  // namespace
void launch_null(hipStream_t s) { hipLaunchKernelGGL(null_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace bn
