// top_k_predictions on the device: one wavefront (64 lanes) per logits row.
//
// Reference behaviour being reproduced (file:line under the reference tree):
//   src/postprocess.rs:8-35   ScoreEntry ordering = reversed f32::total_cmp
//   src/postprocess.rs:40-87  push every logit into a BinaryHeap capped at k,
//                             sigmoid the survivors in Vec order, filter
//                             `confidence >= min`, stable sort descending
//   src/postprocess.rs:91-93  sigmoid(x) = 1 / (1 + exp(-x))
//
// The result must be identical to the reference INCLUDING which index survives
// a tie and the order of equal confidences, both of which are decided by Rust
// std's BinaryHeap arrangement.  So the kernel runs the very same heap
// algorithm (sift_up / sift_down_to_bottom with the Hole idiom) on an LDS
// resident heap, and gets its speed from skipping, 64 logits at a time, the
// pushes that are provably no-ops:
//
//   push(x) followed by pop() leaves the heap array bit-for-bit unchanged when
//   (1) key(x) < key(root) strictly, and
//   (2) along the fixed root->slot-k path every on-path node is strictly
//       smaller than the sibling of its on-path child (and than slot k-1 when
//       k is even).
//   Under (1) x sifts up to the root, shifting the path down one level; pop()
//   then drops x, and sift_down_to_bottom walks the same path back because (2)
//   makes the on-path child the unique minimum at every level.
//
// Condition (2) only involves O(log k) fixed slots, so it is re-evaluated by
// lane 0 after every real heap update.  On typical logits ~k*ln(n/k) of the n
// elements are real updates; fully tied rows degrade to the serial algorithm
// but stay exact.
//
// exp(): Rust's f32::exp is the platform libm expf.  glibc 2.35's expf (the
// ARM optimized-routines algorithm: 32-entry 2^(i/32) table, cubic in double
// precision) is restated here operation for operation, in the fused form its
// x86-64 FMA build executes, so confidences are bit-identical to the
// reference on an FMA-capable host.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace bn {
namespace {

__device__ __forceinline__ uint32_t total_key(uint32_t b) {
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__constant__ uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// glibc 2.35 expf, FMA build.  Compiled with -ffp-contract=off so only the
// explicit fma() calls fuse.
__device__ float expf_glibc(float x) {
    const uint32_t ix = __float_as_uint(x);
    const uint32_t abstop = (ix >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {  // |x| >= 88 or NaN
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);  // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                         // underflow
    }
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double xd = (double)x;
    double kd = fma(InvLn2N, xd, Shift);
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - Shift;
    const double r = fma(InvLn2N, xd, -kd);
    const uint64_t t = kExp2fTab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double z = fma(C0, r, C1);
    const double r2 = r * r;
    double y = fma(r, C2, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (float)y;
}

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf_glibc(-x)); }

struct Heap {
    uint32_t *key;  // total_cmp keys
    uint32_t *idx;
};

// heap_le(a, b) <=> ScoreEntry a <= b <=> key(a) >= key(b)
__device__ __forceinline__ void sift_up(Heap h, uint32_t start, uint32_t pos) {
    const uint32_t ek = h.key[pos], ei = h.idx[pos];
    while (pos > start) {
        const uint32_t parent = (pos - 1) >> 1;
        if (ek >= h.key[parent]) break;
        h.key[pos] = h.key[parent];
        h.idx[pos] = h.idx[parent];
        pos = parent;
    }
    h.key[pos] = ek;
    h.idx[pos] = ei;
}

__device__ __forceinline__ void sift_down_to_bottom(Heap h, uint32_t end, uint32_t pos) {
    const uint32_t start = pos;
    const uint32_t ek = h.key[pos], ei = h.idx[pos];
    uint32_t child = 2 * pos + 1;
    const uint32_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim) {
        child += (h.key[child] >= h.key[child + 1]) ? 1u : 0u;
        h.key[pos] = h.key[child];
        h.idx[pos] = h.idx[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        h.key[pos] = h.key[child];
        h.idx[pos] = h.idx[child];
        pos = child;
    }
    h.key[pos] = ek;
    h.idx[pos] = ei;
    sift_up(h, start, pos);
}

// Condition (2) of the header comment for a full heap of k entries.
__device__ bool path_is_strict(Heap h, uint32_t k) {
    uint32_t node = k;  // slot the next push lands in
    while (node > 0) {
        const uint32_t parent = (node - 1) >> 1;
        const uint32_t sib = (node & 1u) ? node + 1 : node - 1;
        // sibling of the on-path child; slot k itself is not in the heap yet
        if (sib < k && !(h.key[parent] < h.key[sib])) return false;
        node = parent;
    }
    return true;
}

// LDS layout: key[k+1] | idx[k+1] | conf[k] (conf reuses nothing: kept separate for clarity)
__global__ __launch_bounds__(64) void topk_kernel(const float *__restrict__ logits, int64_t n,
                                                  uint32_t k, int has_min, float min_conf,
                                                  int64_t k_stride, uint32_t *__restrict__ idx_out,
                                                  float *__restrict__ conf_out,
                                                  uint32_t *__restrict__ count_out,
                                                  const uint32_t *__restrict__ flags) {
    extern __shared__ __align__(16) uint32_t lds[];
    if (flags && flags[blockIdx.x] == 0) return;  // the fast kernel already produced this row
    Heap h{lds, lds + (k + 1)};
    float *conf = reinterpret_cast<float *>(lds + 2 * (size_t)(k + 1));
    uint32_t *oidx = lds + 2 * (size_t)(k + 1) + k;
    __shared__ uint32_t s_strict, s_root, s_cnt;

    const int64_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const float *x = logits + row * n;

    // ---- phase 1: the first k pushes (no pops) ----
    if (lane == 0) {
        for (uint32_t i = 0; i < k; i++) {
            h.key[i] = total_key(__float_as_uint(x[i]));
            h.idx[i] = i;
            sift_up(h, 0, i);
        }
        s_strict = path_is_strict(h, k) ? 1u : 0u;
        s_root = h.key[0];
    }
    __syncthreads();

    // ---- phase 2: remaining elements, 64 per step, skipping provable no-ops ----
    for (int64_t base = k; base < n; base += 64) {
        const int64_t i = base + lane;
        const bool valid = i < n;
        const uint32_t myk = valid ? total_key(__float_as_uint(x[i])) : 0u;
        uint64_t pending = __ballot(valid);
        while (pending) {
            const uint32_t strict = s_strict, root = s_root;
            const bool mine = (pending >> lane) & 1ull;
            const uint64_t cand = __ballot(mine && (!strict || myk >= root));
            if (!cand) break;
            const int l = __ffsll((long long)cand) - 1;  // lowest index first: sequential order
            const uint32_t xk = __shfl(myk, l);
            if (lane == 0) {
                // heap.push(x); heap.pop()
                h.key[k] = xk;
                h.idx[k] = (uint32_t)(base + l);
                sift_up(h, 0, k);
                // pop: item = data.pop(); swap(item, data[0]); sift_down_to_bottom(0)
                h.key[0] = h.key[k];
                h.idx[0] = h.idx[k];
                sift_down_to_bottom(h, k, 0);
                s_strict = path_is_strict(h, k) ? 1u : 0u;
                s_root = h.key[0];
            }
            __syncthreads();
            pending &= ~((2ull << l) - 1ull);  // lanes <= l are done
        }
        __syncthreads();
    }

    // ---- phase 3: sigmoid in Vec order, filter, stable sort descending ----
    for (uint32_t j = lane; j < k; j += 64) {
        const uint32_t kk = h.key[j];
        const uint32_t bits = (kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk;
        conf[j] = sigmoid_ref(__uint_as_float(bits));
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t m = 0;
        for (uint32_t j = 0; j < k; j++) {
            const float c = conf[j];
            if (has_min && !(c >= min_conf)) continue;
            conf[m] = c;  // m <= j: in-place compaction
            oidx[m] = h.idx[j];
            m++;
        }
        s_cnt = m;
    }
    __syncthreads();
    const uint32_t m = s_cnt;
    uint32_t *io = idx_out + row * k_stride;
    float *co = conf_out + row * k_stride;
    if (m <= 64) {
        // insertion sort, exactly what slice::sort_by runs for len <= 20 (and the
        // oracle for every len): move left while strictly greater.
        if (lane == 0) {
            for (uint32_t i = 1; i < m; i++) {
                const float tc = conf[i];
                const uint32_t ti = oidx[i];
                uint32_t j = i;
                while (j > 0 && tc > conf[j - 1]) {
                    conf[j] = conf[j - 1];
                    oidx[j] = oidx[j - 1];
                    j--;
                }
                conf[j] = tc;
                oidx[j] = ti;
            }
        }
        __syncthreads();
        for (uint32_t j = lane; j < m; j += 64) {
            io[j] = oidx[j];
            co[j] = conf[j];
        }
    } else {
        // stable rank sort (identical to any stable sort when confidences are
        // totally ordered, the only case std::sort_by specifies for len > 20)
        for (uint32_t i = lane; i < m; i += 64) {
            const float ci = conf[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; j++) {
                const float cj = conf[j];
                rank += (cj > ci || (j < i && !(ci > cj))) ? 1u : 0u;
            }
            io[rank] = oidx[i];
            co[rank] = ci;
        }
    }
    // slots past the count are defined (zero): callers that ship whole [rows, k] blocks never see stale memory
    for (uint32_t j = m + lane; j < k; j += 64) {
        io[j] = 0u;
        co[j] = 0.0f;
    }
    if (lane == 0) count_out[row] = m;
}


constexpr int FAST_CAP = 1024;  // candidates (keys >= the lower bound of the (k+1)-th largest) the fast path ranks in LDS

// Second half of the fast path, shared by both scan variants: rank the compacted candidates, check that the result is
// decided by the values alone, write it or flag the row for the exact kernel.
__device__ __forceinline__ void topk_fast_finish(uint32_t cnt, uint32_t lane, int64_t row, uint32_t k, int has_min, float min_conf, int64_t k_stride,
                                                 uint32_t *ckey, uint32_t *cidx, uint32_t *skey, uint32_t *sidx, float *sconf,
                                                 uint32_t *__restrict__ idx_out, float *__restrict__ conf_out, uint32_t *__restrict__ count_out,
                                                 uint32_t *__restrict__ flags) {
    if (cnt > FAST_CAP) {  // heavy ties around the threshold: exact kernel
        if (lane == 0) flags[row] = 1;
        return;
    }
    __syncthreads();
    for (uint32_t c = lane; c < cnt; c += 64) {
        const uint32_t kc = ckey[c], ic = cidx[c];
        uint32_t r = 0;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t kj = ckey[j];
            r += (kj > kc || (kj == kc && cidx[j] < ic)) ? 1u : 0u;
        }
        if (r <= k) {
            skey[r] = kc;
            sidx[r] = ic;
        }
    }
    __syncthreads();
    bool bad = false;
    float c0 = 0.f;
    if (lane <= k) {
        const uint32_t kk = skey[lane];
        const uint32_t bits = (kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk;
        const float v = __uint_as_float(bits);
        c0 = sigmoid_ref(v);
        sconf[lane] = c0;
        if (lane < k) bad = (kk == skey[lane + 1]) || (v != v);  // adjacent equal keys / NaN
    }
    __syncthreads();
    if (lane + 1 < k) bad = bad || (c0 == sconf[lane + 1]);      // equal confidences among the survivors
    if (__ballot(bad)) {
        if (lane == 0) flags[row] = 1;
        return;
    }
    const bool keep = lane < k && (!has_min || c0 >= min_conf);  // a prefix: confidences are descending
    const uint64_t km = __ballot(keep);
    if (lane < k) {  // slots past the count are defined (zero)
        idx_out[row * k_stride + lane] = keep ? sidx[lane] : 0u;
        conf_out[row * k_stride + lane] = keep ? c0 : 0.0f;
    }
    if (lane == 0) {
        count_out[row] = (uint32_t)__popcll(km);
        flags[row] = 0;
    }
}

// ---------------------------------------------------------------------------
// Fast path.  When the k+1 largest keys of a row are pairwise distinct and the k
// confidences are pairwise distinct and not NaN, the reference's result does not
// depend on the heap's internal arrangement at all: it is the k largest logits
// in descending order (sigmoid is monotone), cut where `confidence >= min`
// stops holding.  This kernel computes exactly that, one wavefront per row:
//   pass 1  per-lane maximum; the (k+1)-th largest of the 64 lane maxima is a lower
//           bound T0 of the row's (k+1)-th largest key
//   pass 2  ballot-compact every element with key >= T0 into LDS (typically < 100)
//   rank    each candidate counts the candidates that beat it -> sorted top k+1
// and raises flags[row] = 1 for every row it cannot decide (ties, NaN, too many
// candidates); those rows are then redone by the exact heap kernel above.
__global__ __launch_bounds__(64) void topk_fast_kernel(const float *__restrict__ logits, int64_t n, uint32_t k, int has_min,
                                                       float min_conf, int64_t k_stride, uint32_t *__restrict__ idx_out,
                                                       float *__restrict__ conf_out, uint32_t *__restrict__ count_out,
                                                       uint32_t *__restrict__ flags) {
    __shared__ uint32_t ckey[FAST_CAP], cidx[FAST_CAP];
    __shared__ uint32_t skey[64], sidx[64];
    __shared__ float sconf[64];
    const int64_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t *x = reinterpret_cast<const uint32_t *>(logits + row * n);

    uint32_t mk = 0;
    {
        // four loads in flight per lane (the row is read twice, both passes are latency-bound)
        int64_t i = lane;
        uint32_t m1 = 0, m2 = 0, m3 = 0;
        for (; i + 192 < n; i += 256) {
            const uint32_t k0 = total_key(x[i]), k1 = total_key(x[i + 64]), k2 = total_key(x[i + 128]), k3 = total_key(x[i + 192]);
            mk = k0 > mk ? k0 : mk;
            m1 = k1 > m1 ? k1 : m1;
            m2 = k2 > m2 ? k2 : m2;
            m3 = k3 > m3 ? k3 : m3;
        }
        for (; i < n; i += 64) {
            const uint32_t kk = total_key(x[i]);
            mk = kk > mk ? kk : mk;
        }
        m1 = m1 > m2 ? m1 : m2;
        mk = mk > m3 ? mk : m3;
        mk = mk > m1 ? mk : m1;
    }
    // rank of this lane's maximum among the 64 (ties by lane id): the lane of rank k holds T0
    uint32_t rank = 0;
    for (int l = 0; l < 64; l++) {
        const uint32_t o = __shfl(mk, l);
        rank += (o > mk || (o == mk && (uint32_t)l < lane)) ? 1u : 0u;
    }
    const uint64_t who = __ballot(rank == k);
    const uint32_t T0 = __shfl(mk, __ffsll((long long)who) - 1);

    uint32_t cnt = 0;
    for (int64_t base = 0; base < n; base += 256) {
        // element order (base+lane, base+64+lane, ...) is preserved: candidates stay in index order
        uint32_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = base + 64 * u + lane;
            kk[u] = i < n ? total_key(x[i]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = base + 64 * u + lane;
            const bool pred = i < n && kk[u] >= T0;
            const uint64_t m = __ballot(pred);
            const uint32_t pos = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pred && pos < FAST_CAP) {
                ckey[pos] = kk[u];
                cidx[pos] = (uint32_t)i;
            }
            cnt += (uint32_t)__popcll(m);
        }
    }
    topk_fast_finish(cnt, lane, row, k, has_min, min_conf, k_stride, ckey, cidx, skey, sidx, sconf, idx_out, conf_out, count_out, flags);
}

// The same fast path with the WHOLE ROW IN REGISTERS (n <= 64 * NCH): every load of the row is in flight at once, so
// the scan costs one memory round trip instead of two passes of ~26 dependent iterations (22.8 -> ~5 us for 32 rows of
// 6522 logits, the longest latency-only launch of a step).  Candidates are compacted in the same index order as above,
// so the two variants give identical results.
template <int NCH>
__global__ __launch_bounds__(64) void topk_fast_reg_kernel(const float *__restrict__ logits, int64_t n, uint32_t k, int has_min,
                                                           float min_conf, int64_t k_stride, uint32_t *__restrict__ idx_out,
                                                           float *__restrict__ conf_out, uint32_t *__restrict__ count_out,
                                                           uint32_t *__restrict__ flags) {
    __shared__ uint32_t ckey[FAST_CAP], cidx[FAST_CAP];
    __shared__ uint32_t skey[64], sidx[64];
    __shared__ float sconf[64];
    const int64_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t *x = reinterpret_cast<const uint32_t *>(logits + row * n);
    uint32_t kk[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int64_t i = 64 * j + lane;
        kk[j] = x[i < n ? i : n - 1];  // clamped, never predicated: all NCH loads issue back to back
    }
    uint32_t mk = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        kk[j] = 64 * j + (int64_t)lane < n ? total_key(kk[j]) : 0u;
        mk = kk[j] > mk ? kk[j] : mk;
    }
    uint32_t rank = 0;
    for (int l = 0; l < 64; l++) {
        const uint32_t o = __shfl(mk, l);
        rank += (o > mk || (o == mk && (uint32_t)l < lane)) ? 1u : 0u;
    }
    const uint64_t who = __ballot(rank == k);
    const uint32_t T0 = __shfl(mk, __ffsll((long long)who) - 1);
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int64_t i = 64 * j + lane;
        const bool pred = i < n && kk[j] >= T0;
        const uint64_t m = __ballot(pred);
        if (m) {  // wave-uniform: most groups of 64 hold no candidate
            const uint32_t pos = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pred && pos < FAST_CAP) {
                ckey[pos] = kk[j];
                cidx[pos] = (uint32_t)i;
            }
            cnt += (uint32_t)__popcll(m);
        }
    }
    topk_fast_finish(cnt, lane, row, k, has_min, min_conf, k_stride, ckey, cidx, skey, sidx, sconf, idx_out, conf_out, count_out, flags);
}

// Results to the host without the copy engines: the blocks store straight into pinned (device-mapped, coherent) host
// memory, 16 bytes per lane where both ends are 16-byte aligned.  Up to three regions per launch (logits, embeddings,
// packed top-K rows).  hipMemcpyAsync would hand each region to an SDMA engine picked per call; with several
// contexts copying at once the runtime brings further engines up lazily, a few ms each -- measured as three rounds of
// 4 concurrent steps at 6 ms instead of 3 ms after every process start (tools/warmup_profile.py) -- and every copy is
// a host API call of its own.
__global__ void __launch_bounds__(256) copy_out_kernel(CopyOut c) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nth = gridDim.x * 256u;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        if (r >= c.n) break;
        const uint32_t words = c.words[r];
        const uint32_t *src = static_cast<const uint32_t *>(c.src[r]);
        uint32_t *dst = static_cast<uint32_t *>(c.dst[r]);
        uint32_t done = 0;
        if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
            const uint32_t n4 = words >> 2;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
            uint4 *d4 = reinterpret_cast<uint4 *>(dst);
            for (uint32_t i = tid; i < n4; i += nth) d4[i] = s4[i];
            done = n4 << 2;
        }
        for (uint32_t i = done + tid; i < words; i += nth) dst[i] = src[i];
    }
}

}  // namespace

void launch_copy_out(hipStream_t s, const CopyOut &c) {
    uint64_t words = 0;
    for (int r = 0; r < c.n && r < 3; r++) words += c.words[r];
    if (words == 0) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(64, (words / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(copy_out_kernel, dim3(blocks), dim3(256), 0, s, c);
}

// device -> device with the same kernel over the whole chip (a caller's batch into the context's input buffer): the runtime's
// hipMemcpyAsync is a command of its own kind on the queue; a kernel of the library's own is ordinary stream work like the plan
// behind it (BN_INPUT_MEMCPY=1 restores the runtime copy)
void launch_copy_dev(hipStream_t s, void *dst, const void *src, uint64_t bytes) {
    if (bytes == 0) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(2048, (bytes / 16 + 255) / 256 + 1);
    uint64_t done = 0;
    while (done < bytes) {  // 32-bit word counts per launch: 8 GiB pieces
        const uint64_t piece = std::min<uint64_t>(bytes - done, (uint64_t)0x7ffffff0u * 4);
        CopyOut c{};
        c.n = 1;
        c.dst[0] = static_cast<char *>(dst) + done;
        c.src[0] = static_cast<const char *>(src) + done;
        c.words[0] = (uint32_t)(piece / 4);
        hipLaunchKernelGGL(copy_out_kernel, dim3(blocks), dim3(256), 0, s, c);
        done += piece;
    }
}

size_t topk_lds_bytes(int64_t n, int64_t k) {
    (void)n;
    size_t b = (size_t)(2 * (k + 1) + 2 * k) * 4;
    return b <= 150 * 1024 ? b : 0;
}

void register_topk_kernels() { register_dynamic_lds_kernel(reinterpret_cast<const void *>(topk_kernel)); }

void launch_topk(hipStream_t s, const float *logits, int64_t rows, int64_t n, int64_t k,
                 int has_min, float min_conf, int64_t k_stride, uint32_t *idx, float *conf,
                 uint32_t *count, uint32_t *flags) {
    if (rows <= 0 || k <= 0 || n <= 0) return;
    // fast path needs k+1 lane maxima; BN_TOPK_EXACT=1 forces the exact heap kernel (tests)
    if (flags && (k > 62 || n < 64 || getenv("BN_TOPK_EXACT"))) flags = nullptr;
    if (flags && n <= 64 * 112 && !getenv("BN_TOPK_TWOPASS"))
        hipLaunchKernelGGL(topk_fast_reg_kernel<112>, dim3((unsigned)rows), dim3(64), 0, s, logits, n, (uint32_t)k, has_min, min_conf, k_stride, idx,
                           conf, count, flags);
    else if (flags)
        hipLaunchKernelGGL(topk_fast_kernel, dim3((unsigned)rows), dim3(64), 0, s, logits, n, (uint32_t)k, has_min, min_conf, k_stride, idx,
                           conf, count, flags);
    const size_t lds = topk_lds_bytes(n, k);
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(topk_kernel), lds)) {
        launch_error("top-K heap needs more LDS than the device grants");
        return;
    }
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)rows), dim3(64), lds, s, logits, n, (uint32_t)k,
                       has_min, min_conf, k_stride, idx, conf, count, (const uint32_t *)flags);
}

}  // namespace bn
