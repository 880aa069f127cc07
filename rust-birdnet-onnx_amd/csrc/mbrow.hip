// Fused MBConv (expand 1x1 / stem k1 x k1 conv -> depthwise K x K -> squeeze partials), row-streaming form.
//
// One WAVE = one unit of work = (sample, band of output rows, strip of output columns, chunk of 32 mid channels); the
// four waves of a block are four consecutive units (chunk fastest, so they read the same input rows) and never
// synchronise: no LDS, no barrier.
//
//   * The wave walks down the strip's halo rows.  For each row it expands 32 consecutive halo pixels x 32 channels
//     with ONE 32x32 accumulator tile (v_mfma_f32_32x32x2_f32; A operands straight from global memory / L2 into
//     registers, loaded one row ahead; B operands = the chunk's 32 filters, resident in registers for the whole unit).
//     The tile's row index m is mapped to the pixel x(m) = 16*((m>>2)&1) + 4*(m>>3) + (m&3), so that the C layout
//     (lane (c, lh) holds rows 8*(r>>2) + 4*lh + (r&3), r = 0..15, of column c) leaves every lane with SIXTEEN
//     CONSECUTIVE PIXELS of one channel: x = 16*lh + r.
//   * The last K expanded rows stay in registers (K x 16 values per lane + the K - S values of the neighbouring half
//     row that the window of the left half reaches into, fetched with v_permlane32_swap).  As soon as a row completes
//     an output row, the lane runs the K x K depthwise window over its 16 / S outputs -- vertical taps are other
//     registers, horizontal taps are neighbouring registers -- applies bias + activation and stores NHWC (lanes
//     c = 0..31 of a half wave write 128 contiguous bytes per pixel).
//   * The expanded tensor never exists outside registers; the input is re-read once per chunk from L2.
//
// Arithmetic order is the one of mbconv_expand_dw_kernel (kernels.hip): accumulators start at the expand bias inside the
// image and at 0 outside (a pixel outside the image expands to act(0) = 0, the zero padding the depthwise conv needs),
// K groups ascending, depthwise taps (ky, kx) ascending from bias2 -- the two kernels produce the same bits; only the
// squeeze partials are cut differently (per band x strip instead of per 2-D tile).
//
// Halo recompute: a strip's 32 halo pixels yield (32 - K) / S + 1 outputs (30 / 28 / 15 / 14), a band of toh output rows
// needs (toh - 1) * S + K halo rows; the matrix pipe (15 % busy in the tiled kernel) absorbs both.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "bf16x3.h"
#include "device_common.h"
#include "kernels.h"
#include "plan_rules.h"

namespace bn {
namespace {

// activations the row kernel carries: mbconv_row_act_supported (kernels.h); each call site is replicated per row slot, so
// the set is kept small and the planner sends every other code to the tiled kernel
template <int N>
__device__ __forceinline__ void row_act(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
}

// value of the same register in the lane 32 places up (lanes 0..31 receive lanes 32..63; the upper half keeps its own)
__device__ __forceinline__ float upper_half(float v) {
    const unsigned u = __float_as_uint(v);
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[1]);
}

typedef float float2_t __attribute__((ext_vector_type(2)));


#ifdef BN_MB_STAMPS  // tools/mb_probe.cpp: shader-clock stamps of the first main-loop steps of every unit
__device__ unsigned long long *bn_row_stamps = nullptr;  // [units][64]
#define ROW_STAMP()                                                                                                  \
    do {                                                                                                             \
        if (bn_row_stamps && lane == 0 && nst < 64) bn_row_stamps[(size_t)u * 64 + nst] = __builtin_amdgcn_s_memtime(); \
        nst++;                                                                                                       \
    } while (0)
#else
#define ROW_STAMP()
#endif

template <int K, int S>
struct RowCfg {
    static constexpr int NOUT = 16 / S;             // outputs per lane and row
    static constexpr int OUTW = (32 - K) / S + 1;   // outputs per strip and row
    static constexpr int EXT = K - S;               // values of the neighbouring half row a left-half lane needs
    static constexpr int XW = 16 + EXT;
};

// depthwise window over the K rows, NEWEST = slot of the row that completed last (oldest = NEWEST + 1 mod K)
template <int K, int S, int NEWEST, int XW_, int KK_, int NOUT>
__device__ __forceinline__ void dw_rows(const float (&rows)[K][XW_], const float (&wd)[KK_], float bias2, float (&ov)[NOUT]) {
#pragma unroll
    for (int q = 0; q < NOUT; q++) ov[q] = bias2;
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
        constexpr int base = NEWEST + 1;
        const int slot = (base + ky) % K;  // folds: ky is unrolled
#pragma unroll
        for (int ix = 0; ix < (NOUT - 1) * S + K; ix++) {
            const float v = rows[slot][ix];
#pragma unroll
            for (int kx = 0; kx < K; kx++)
                if (ix - kx >= 0 && (ix - kx) % S == 0 && (ix - kx) / S < NOUT) ov[(ix - kx) / S] = fmaf(v, wd[ky * K + kx], ov[(ix - kx) / S]);
        }
    }
}

// Compile-time activation (ACT >= 0: both activations of the block are this code) or run-time dispatch (ACT < 0).
template <int ACT, int N>
__device__ __forceinline__ void row_act_t(int act, float p0, float p1, float (&v)[N]) {
    if constexpr (ACT == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if constexpr (ACT == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else row_act<N>(act, p0, p1, v);
}

// Schedule of one unit (nrows halo rows, J = 0 .. nrows; step J expands row J and emits the output row that row J - 1
// completed, so that inside ONE basic block the matrix instructions of row J and the depthwise arithmetic over rows
// J - K .. J - 1 are independent and the scheduler interleaves them):
//   prologue  J = 0 .. K-1        expand only
//   main      J = K .. nrows      unrolled 2K times: slot = i % K, operand buffer = (K + i) & 1, stride-2 blocks emit on
//                                 even i; J = nrows is the drain step (its expansion is discarded)
// Row and column masks (halo outside the image) are applied by rarely taken uniform branches AFTER the block.
// Waves per SIMD the instance is compiled for: two (256 registers each) wherever everything fits; the instances whose live state does not
// (5 x 5 windows with SiLU / run-time activations or five and more K groups; run-time activations with six K groups or an im2col stem) get
// one wave's 512 registers instead of spilling to scratch memory -- tests/test_build_hygiene.py keeps the build free of scratch
template <int K, int NG, int S, bool IM2COL, int ACT, bool B3 = false>
constexpr int mbrow_waves_per_simd() {
    if (B3 && K == 5 && NG >= 5) return 1;  // (three filter planes + the split's temporaries: 36 - 40 bytes of scratch at 256 registers)
    if (K == 5 && (ACT != ACT_RELU || (S == 1 && NG >= 5))) return 1;
    if (ACT != ACT_RELU && ACT != ACT_SILU && (NG >= 6 || IM2COL)) return 1;
    if (IM2COL && ACT == ACT_SILU && NG >= 4) return 1;
    return 2;
}
// B3 (round 5): the expand conv as six exact-bf16 partial products per 16-deep k group on the bf16 matrix pipe (bf16x3.h) -- 1x1 expands with
// Cin % 8 == 0 only.  Lane (pixel lr, lh) then feeds k = 16 G + 8 lh .. + 7 of group G (two float4 loads, split in registers: ~44 vector
// instructions per group and halo row, which issue in the matrix instructions' shadow), the filters are split once per unit.
template <int K, int S, int NG, bool IM2COL, int ACT, bool TR = false, bool B3 = false>
__global__ __launch_bounds__(256, (mbrow_waves_per_simd<K, NG, S, IM2COL, ACT, B3>())) void mbconv_row_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                            const float *__restrict__ w1, const float *__restrict__ b1,
                                                            const float *__restrict__ w2, const float *__restrict__ b2,
                                                            float *__restrict__ gap, int total_units) {
    using Cfg = RowCfg<K, S>;
    constexpr int NOUT = Cfg::NOUT, OUTW = Cfg::OUTW, XW = Cfg::XW, EXT = Cfg::EXT;
    constexpr int QBOTH = OUTW - NOUT;  // outputs q < QBOTH exist in both half rows, the rest only in the left one
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int u = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (u >= total_units) return;
    const int nchunks = (d.C + 31) >> 5;
    int t = u / nchunks;
    const int ch = u - t * nchunks;
    const int strip = t % d.tiles_x;
    t /= d.tiles_x;
    const int band = t % d.tiles_y;
    const int64_t b = t / d.tiles_y;

    const int oy0 = band * d.toh;
    const int rows_out = min(d.toh, d.OH - oy0);
    const int nrows = (rows_out - 1) * S + K;  // halo rows this unit walks
    const int iy0 = oy0 * S - d.pt;
    const int ox0 = strip * OUTW;
    const int ix0 = ox0 * S - d.pl;
    // every halo column inside the image / every output column inside the output map: the unit runs without masks
    const bool cols_in = ix0 >= 0 && ix0 + 32 <= d.W;
    const bool outs_in = ox0 + OUTW <= d.OW && ch * 32 + 32 <= d.C;  // a full chunk in an interior strip

    // ---- matrix-operand role: lane lr feeds accumulator row m = lr = pixel xm of the halo row
    const int xm = 16 * ((lr >> 2) & 1) + 4 * (lr >> 3) + (lr & 3);
    const int ixc = min(max(ix0 + xm, 0), d.W - 1);  // clamped: the loads never leave the row, masked columns are zeroed after the act
    constexpr int ngr = NG;                           // K groups: the launcher picks the instance with NG == ceil(Cin / 8)
    static_assert(!B3 || !IM2COL, "the bf16 form is for 1x1 expands");
    constexpr int NG16 = (NG + 1) / 2;        // B3: 16-deep k groups; an odd NG leaves the last group's upper half (lh == 1) as padding
    constexpr bool B3_PAD = B3 && (NG & 1);
    float4 bw[B3 ? 1 : NG];
    b3_u32x4 bwh[B3 ? NG16 : 1], bwm[B3 ? NG16 : 1], bwl[B3 ? NG16 : 1];
    {
        const int n = min(ch * 32 + lr, d.C - 1);
        if constexpr (B3) {
            const float *wr = w1 + (int64_t)n * (ngr * 8);
#pragma unroll
            for (int G = 0; G < NG16; G++) {
                const bool pad = B3_PAD && G == NG16 - 1 && lh == 1;
                const float *p = wr + (pad ? 0 : 16 * G + 8 * lh);
                b3_floatx4 lo4 = *reinterpret_cast<const b3_floatx4 *>(p), hi4 = *reinterpret_cast<const b3_floatx4 *>(p + 4);
                if (pad) lo4 = hi4 = b3_floatx4{0.f, 0.f, 0.f, 0.f};
                split3(lo4, hi4, bwh[G], bwm[G], bwl[G]);
            }
        } else {
            const float *wr = w1 + (int64_t)n * (ngr * 8) + 4 * lh;
#pragma unroll
            for (int g = 0; g < NG; g++) bw[g] = *reinterpret_cast<const float4 *>(wr + 8 * g);
        }
    }
    // ---- accumulator / depthwise role: lane (c, lh) owns channel cg, halo pixels 16*lh .. 16*lh + 15
    const int cg = ch * 32 + lr;
    const bool cact = cg < d.C;
    const int cgc = cact ? cg : d.C - 1;
    unsigned cmask = 0;  // bit r: halo pixel 16*lh + r lies inside the image
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int ix = ix0 + 16 * lh + r;
        cmask |= (ix >= 0 && ix < d.W) ? (1u << r) : 0u;
    }
    unsigned omask = 0;  // bit q: output q of this lane exists (inside the strip, inside the map, a real channel)
#pragma unroll
    for (int q = 0; q < NOUT; q++) omask |= (cact && NOUT * lh + q < OUTW && ox0 + NOUT * lh + q < d.OW) ? (1u << q) : 0u;
    const float bv = d.has_bias1 ? b1[cgc] : 0.0f;
    const float bias2 = d.has_bias2 ? b2[cgc] : 0.0f;
    float wd[K * K];  // [kernel row tap][kernel column tap]; transposed: the map's (kx, ky)
    constexpr bool tr = TR;  // compile-time: the second addressing form costs the plain instances nothing (3 x 3, 1x1-expand blocks only)
    if (tr) {
#pragma unroll
        for (int q = 0; q < K * K; q++) wd[q] = w2[((q % K) * K + q / K) * d.C + cgc];
    } else {
#pragma unroll
        for (int q = 0; q < K * K; q++) wd[q] = w2[q * d.C + cgc];
    }

    // ---- A operand addressing.  plain: X[iy][ixc][8g + 4lh .. +3], a group that is channel padding (Cin % 8 == 4, last
    // group, upper half) re-reads group 0 and is zeroed; stem: im2col column k = 8g + 4lh + j -> tap (ky, kx), channel cc
    const float *xin = in + b * d.in_bs;
    const bool pad_lane = !IM2COL && (d.Cin & 7) != 0 && lh == 1;  // its last group lies past the pixel's channels
    const bool any_pad = !IM2COL && (d.Cin & 7) != 0;
    unsigned a_off[IM2COL ? NG * 4 : 1];  // stem: element offset from the pixel's top-left tap (0 = safe dummy)
    unsigned im_ky = 0, im_ok = 0;        // stem: 2 bits of ky per element (k1 <= 4), validity bit per element
    if constexpr (IM2COL) {
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 8 * g + 4 * lh + j;
                const int tap = k / d.Cin1, cc = k - tap * d.Cin1;
                const int ky = tap / d.k1, kx = tap - ky * d.k1;
                const int x = ixc * d.s1 - d.pl1 + kx;
                const bool ok = k < d.Cin && x >= 0 && x < d.W1;
                a_off[g * 4 + j] = ok ? (unsigned)((ky * d.W1 + kx) * d.Cin1 + cc) : 0u;
                im_ky |= (unsigned)(ok ? ky : 0) << (2 * (g * 4 + j));
                im_ok |= ok ? (1u << (g * 4 + j)) : 0u;
            }
    } else {
        a_off[0] = 0;
    }
    // stem: every lane's every REAL column (k < K) has its tap inside the image horizontally -> the rows that also lie
    // inside vertically load without predicates; columns k >= K read the dummy element against zero weights
    bool im_fast = false;
    if constexpr (IM2COL) {
        bool mine = true;
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int j = 0; j < 4; j++) mine = mine && (((im_ok >> (g * 4 + j)) & 1u) || 8 * g + 4 * lh + j >= d.Cin);
        im_fast = __all(mine);
    }
    // ... and with a 2-channel image (the v2.4 spectrogram pair) whose samples are 8-byte aligned, the columns (k, k + 1) of a tap
    // are one dwordx2: half the load instructions of the launch's busiest path (a dummy column pair reads elements 0, 1)
    bool im_pair = false;
    if constexpr (IM2COL) {
        bool pairs = im_fast && d.Cin1 == 2 && (d.in_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(in) & 7u) == 0;
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const unsigned ok0 = (im_ok >> (g * 4 + j)) & 1u, ok1 = (im_ok >> (g * 4 + j + 1)) & 1u;
                const bool real = ok0 && ok1 && a_off[g * 4 + j + 1] == a_off[g * 4 + j] + 1u && (a_off[g * 4 + j] & 1u) == 0;
                pairs = pairs && (real || (!ok0 && !ok1));  // a padding pair (k >= K) reads elements 0, 1 against zero weights
            }
        im_pair = __all(pairs);
    }

    // (transposed: kernel pixel (r, c) is map pixel (y = c, x = r) of a map that is d.H wide -- a row step is one pixel, a column
    // step one map row)
    const int64_t a_rs = tr ? (int64_t)d.Cin : (int64_t)d.W * d.Cin;   // floats per input row
    const int a_px = tr ? d.H * d.Cin : d.Cin;                          // floats per input column step
    const unsigned a_lane4 = 4u * (unsigned)(ixc * a_px + (B3 ? 8 : 4) * lh);  // byte offset of this lane's pixel + K half within the row
    const unsigned a_last4 = a_lane4 + (pad_lane ? 0u : 32u * (NG - 1));  // channel-padding lanes re-read group 0 (then zeroed)
    constexpr int NA = B3 ? 2 * NG16 : NG;  // float4 operands per halo row and lane
    const bool b3_pad_lane = B3_PAD && lh == 1;
    float4 abuf[2][NA];
#pragma unroll
    for (int g = 0; g < NA; g++) abuf[0][g] = abuf[1][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    // operands of halo row J (clamped into the image: rows outside it are zeroed after the expansion)
#define MBROW_LOAD_A(dst, J)                                                                                                    \
    do {                                                                                                                        \
        const int iy_ = min(max(iy0 + (J), 0), d.H - 1);                                                                        \
        if (MBROW_DBG(8)) break;                                                                                                \
        if constexpr (IM2COL) {                                                                                                 \
            const int y0_ = iy_ * d.s1 - d.pt1;                                                                                 \
            const int x0_ = ixc * d.s1 - d.pl1;                                                                                 \
            const float *px_ = xin + ((int64_t)y0_ * d.W1 + x0_) * d.Cin1;                                                      \
            if (im_pair && y0_ >= 0 && y0_ + d.k1 <= d.H1) { /* two channels per tap: columns (k, k + 1) are neighbours in memory */ \
                _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                \
                    const float2 lo_ = *reinterpret_cast<const float2 *>(px_ + a_off[g * 4]);                                   \
                    const float2 hi_ = *reinterpret_cast<const float2 *>(px_ + a_off[g * 4 + 2]);                               \
                    dst[g] = make_float4(lo_.x, lo_.y, hi_.x, hi_.y);                                                           \
                }                                                                                                               \
            } else if (im_fast && y0_ >= 0 && y0_ + d.k1 <= d.H1) {                                                             \
                _Pragma("unroll") for (int g = 0; g < NG; g++)                                                                  \
                    dst[g] = make_float4(px_[a_off[g * 4]], px_[a_off[g * 4 + 1]], px_[a_off[g * 4 + 2]], px_[a_off[g * 4 + 3]]); \
            } else {                                                                                                            \
                _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                \
                    float v_[4];                                                                                                \
                    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                             \
                        const int y_ = y0_ + (int)((im_ky >> (2 * (g * 4 + j))) & 3u);                                          \
                        const bool ok_ = ((im_ok >> (g * 4 + j)) & 1u) && y_ >= 0 && y_ < d.H1;                                 \
                        v_[j] = ok_ ? px_[a_off[g * 4 + j]] : 0.0f;                                                             \
                    }                                                                                                           \
                    dst[g] = make_float4(v_[0], v_[1], v_[2], v_[3]);                                                           \
                }                                                                                                               \
            }                                                                                                                   \
        } else if constexpr (B3) { /* k = 16 G + 8 lh .. + 7: two float4 per group; a padded upper half re-reads group 0 and is zeroed */ \
            const char *prow_ = reinterpret_cast<const char *>(xin + (int64_t)iy_ * a_rs);                                      \
            _Pragma("unroll") for (int G = 0; G < NG16; G++) {                                                                  \
                const unsigned o_ = a_lane4 + ((G == NG16 - 1 && b3_pad_lane) ? 0u : 64u * G);                                  \
                dst[2 * G] = *reinterpret_cast<const float4 *>(prow_ + o_);                                                     \
                dst[2 * G + 1] = *reinterpret_cast<const float4 *>(prow_ + o_ + 16);                                            \
            }                                                                                                                   \
            if (B3_PAD) {                                                                                                       \
                if (b3_pad_lane) dst[2 * NG16 - 2] = dst[2 * NG16 - 1] = make_float4(0.f, 0.f, 0.f, 0.f);                       \
            }                                                                                                                   \
        } else {                                                                                                                \
            const char *prow_ = reinterpret_cast<const char *>(xin + (int64_t)iy_ * a_rs); /* wave-uniform row base */          \
            _Pragma("unroll") for (int g = 0; g < NG - 1; g++) dst[g] = *reinterpret_cast<const float4 *>(prow_ + 32 * g + a_lane4); \
            dst[NG - 1] = *reinterpret_cast<const float4 *>(prow_ + a_last4);                                                   \
            if (any_pad) {                                                                                                      \
                if (pad_lane) dst[NG - 1] = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
            }                                                                                                                   \
        }                                                                                                                       \
    } while (0)

    float rows[K][XW];
#pragma unroll
    for (int s_ = 0; s_ < K; s_++)
#pragma unroll
        for (int x = 0; x < XW; x++) rows[s_][x] = 0.0f;

    float *obase = out + b * d.out_bs;  // wave-uniform base, lanes add 32-bit offsets
    const int o_px = tr ? d.OH * d.C : d.C;                           // floats per output column step
    const unsigned ocol4 = 4u * (unsigned)((ox0 + NOUT * lh) * o_px + cgc);  // this lane's byte offset within an output row
    const int64_t o_rs = tr ? (int64_t)d.C : (int64_t)d.OW * d.C;     // floats per output row
    const size_t o_ps = 4 * (size_t)o_px;                                   // bytes per output pixel
    int nst = 0;
    (void)nst;
    float2_t sum2 = {0.0f, 0.0f};  // squeeze partial of this lane, even / odd outputs
    floatx16 acc, bvec;           // bvec: the expand bias in every accumulator slot, the C operand of a row's first matrix instruction
#pragma unroll
    for (int r = 0; r < 16; r++) bvec[r] = bv;

    // matrix instructions of one halo row (operands in A)
#ifdef BN_MBROW_DBG  // tools/mb_probe.cpp phase experiments (the branches defeat the interleaving, so timings shift)
#define MBROW_DBG(bit) ((d.row_mode >> 8) & (bit))
#else
#define MBROW_DBG(bit) 0
#endif
#define MBROW_EXPAND(A)                                                                                                         \
    do {                                                                                                                        \
        if (MBROW_DBG(1)) {                                                                                                     \
            acc = bvec;                                                                                                         \
            break;                                                                                                              \
        }                                                                                                                       \
        if constexpr (B3) {                                                                                                     \
            _Pragma("unroll") for (int G = 0; G < NG16; G++) {                                                                  \
                b3_u32x4 xh_, xm_, xl_;                                                                                         \
                split3(b3_floatx4{A[2 * G].x, A[2 * G].y, A[2 * G].z, A[2 * G].w},                                              \
                       b3_floatx4{A[2 * G + 1].x, A[2 * G + 1].y, A[2 * G + 1].z, A[2 * G + 1].w}, xh_, xm_, xl_);              \
                acc = mm32_bf16(xh_, bwl[G], G == 0 ? bvec : acc);                                                              \
                acc = mm32_bf16(xl_, bwh[G], acc);                                                                              \
                acc = mm32_bf16(xm_, bwm[G], acc);                                                                              \
                acc = mm32_bf16(xh_, bwm[G], acc);                                                                              \
                acc = mm32_bf16(xm_, bwh[G], acc);                                                                              \
                acc = mm32_bf16(xh_, bwh[G], acc);                                                                              \
            }                                                                                                                   \
        } else {                                                                                                                \
            _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                    \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g].x, bw[g].x, g == 0 ? bvec : acc, 0, 0, 0);                      \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g].y, bw[g].y, acc, 0, 0, 0);                                      \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g].z, bw[g].z, acc, 0, 0, 0);                                      \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g].w, bw[g].w, acc, 0, 0, 0);                                      \
            }                                                                                                                   \
        }                                                                                                                       \
    } while (0)
    // activation of the expanded row J into slot SLOT (+ masks, + the neighbouring half's first EXT values)
#define MBROW_COMMIT(SLOT, J)                                                                                                   \
    do {                                                                                                                        \
        float e_[16];                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; r++) e_[r] = acc[r];                                                          \
        row_act_t<ACT, 16>(d.act1, d.p0_1, d.p1_1, e_);                                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; r++) rows[SLOT][r] = e_[r];                                                   \
        const int iyj_ = iy0 + (J);                                                                                             \
        if (iyj_ < 0 || iyj_ >= d.H) { /* a halo row above / below the image: the depthwise conv's zero padding */             \
            _Pragma("unroll") for (int r = 0; r < 16; r++) rows[SLOT][r] = 0.0f;                                                \
        } else if (!cols_in) {                                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; r++) rows[SLOT][r] = ((cmask >> r) & 1u) ? rows[SLOT][r] : 0.0f;          \
        }                                                                                                                       \
        _Pragma("unroll") for (int x = 0; x < EXT; x++) rows[SLOT][16 + x] = upper_half(rows[SLOT][x]);                         \
    } while (0)
    // output row JO (relative to the unit's first halo row) from the K rows whose newest sits in slot NEWEST
#define MBROW_EMIT(NEWEST, JO)                                                                                                  \
    do {                                                                                                                        \
        float ov[NOUT];                                                                                                         \
        if (MBROW_DBG(2)) {                                                                                                     \
            _Pragma("unroll") for (int q = 0; q < NOUT; q++) ov[q] = rows[NEWEST][q];                                           \
        } else                                                                                                                  \
            dw_rows<K, S, NEWEST>(rows, wd, bias2, ov);                                                                         \
        row_act_t<ACT, NOUT>(d.act2, d.p0_2, d.p1_2, ov);                                                                       \
        char *orow_ = reinterpret_cast<char *>(obase + (int64_t)(oy0 + (JO) / S) * o_rs); /* wave-uniform */                    \
        if (MBROW_DBG(4)) {                                                                                                     \
            _Pragma("unroll") for (int q = 0; q < NOUT; q += 2) sum2 += float2_t{ov[q], ov[q + 1]};                             \
        } else if (outs_in) {                                                                                                   \
            _Pragma("unroll") for (int q = 0; q + 1 < QBOTH; q += 2) sum2 += float2_t{ov[q], ov[q + 1]};                        \
            if constexpr (QBOTH & 1) sum2[0] += ov[QBOTH - 1];                                                                  \
            if (lh == 0) {                                                                                                      \
                _Pragma("unroll") for (int q = QBOTH; q + 1 < NOUT; q += 2) sum2 += float2_t{ov[q], ov[q + 1]};                 \
                if constexpr ((NOUT - QBOTH) & 1) sum2[1] += ov[NOUT - 1];                                                      \
            }                                                                                                                   \
            _Pragma("unroll") for (int q = 0; q < QBOTH; q++)                                                                   \
                *reinterpret_cast<float *>(orow_ + (size_t)q * o_ps + ocol4) = ov[q];                                           \
            if (lh == 0) {                                                                                                      \
                _Pragma("unroll") for (int q = QBOTH; q < NOUT; q++)                                                            \
                    *reinterpret_cast<float *>(orow_ + (size_t)q * o_ps + ocol4) = ov[q];                                       \
            }                                                                                                                   \
        } else {                                                                                                                \
            _Pragma("unroll") for (int q = 0; q < NOUT; q++)                                                                    \
                if ((omask >> q) & 1u) {                                                                                        \
                    *reinterpret_cast<float *>(orow_ + (size_t)q * o_ps + ocol4) = ov[q];                                       \
                    sum2[q & 1] += ov[q];                                                                                       \
                }                                                                                                               \
        }                                                                                                                       \
    } while (0)

    // ---- prologue: rows 0 .. K-1 (operand buffer = J & 1)
    ROW_STAMP();
    MBROW_LOAD_A(abuf[0], 0);
#define MBROW_PRO(J)                                  \
    do {                                              \
        MBROW_LOAD_A(abuf[((J) + 1) & 1], (J) + 1);   \
        MBROW_EXPAND(abuf[(J) & 1]);                  \
        MBROW_COMMIT((J) % K, (J));                   \
    } while (0)
    MBROW_PRO(0);
    MBROW_PRO(1);
    MBROW_PRO(2);
    if constexpr (K > 3) {
        MBROW_PRO(3);
        MBROW_PRO(4);
    }
#undef MBROW_PRO
    // ---- main: J = K + m * 2K + i
#define MBROW_MAIN(I)                                                                            \
    do {                                                                                         \
        const int j_ = jb + (I);                                                                 \
        if (j_ <= nrows) {                                                                       \
            ROW_STAMP();                                                                         \
            MBROW_LOAD_A(abuf[(K + (I) + 1) & 1], j_ + 1);                                       \
            MBROW_EXPAND(abuf[(K + (I)) & 1]);                                                   \
            if constexpr (S == 1 || ((I) & 1) == 0) MBROW_EMIT(((I) + K - 1) % K, j_ - K);       \
            ROW_STAMP();                                                                         \
            MBROW_COMMIT((I) % K, j_);                                                           \
        }                                                                                        \
    } while (0)
    for (int jb = K; jb <= nrows; jb += 2 * K) {
        {
            MBROW_MAIN(0);
            MBROW_MAIN(1);
            MBROW_MAIN(2);
            MBROW_MAIN(3);
            MBROW_MAIN(4);
            MBROW_MAIN(5);
            if constexpr (K > 3) {
                MBROW_MAIN(6);
                MBROW_MAIN(7);
                MBROW_MAIN(8);
                MBROW_MAIN(9);
            }
        }
    }
    ROW_STAMP();
#undef MBROW_MAIN
#undef MBROW_EMIT
#undef MBROW_COMMIT
#undef MBROW_EXPAND
#undef MBROW_LOAD_A
    if (d.has_gap) {
        const float sum = sum2[0] + sum2[1];
        const float other = upper_half(sum);  // left half + right half, in that order
        if (lh == 0 && cact) gap[b * d.gap_bs + (int64_t)(band * d.tiles_x + strip) * d.C + cg] = sum + other;
    }
}

}  // namespace

static void launch_mbconv_row_impl(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2,
                                   const float *b2, float *gap, int64_t batch);

void launch_mbconv_row(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2,
                       const float *b2, float *gap, int64_t batch) {
    if (batch <= 0) return;
    const int64_t total = batch * d.tiles_x * d.tiles_y * ((d.C + 31) / 32);
    if (!mbconv_row_supported(d) || d.toh <= 0 || total > 0x7fffffff || (d.row_tr && (d.k1 > 0 || d.k != 3))) {
        launch_error("row-streaming MBConv: shape outside the instantiated set");
        return;
    }
    if (d.row_tr) {  // the kernel sees the transposed map (plan terms -> kernel terms); tiles_x / tiles_y / toh are kernel terms already
        MbDesc t = d;
        std::swap(t.H, t.W);
        std::swap(t.OH, t.OW);
        std::swap(t.pt, t.pl);
        return launch_mbconv_row_impl(s, t, out, in, w1, b1, w2, b2, gap, batch);
    }
    launch_mbconv_row_impl(s, d, out, in, w1, b1, w2, b2, gap, batch);
}

static void launch_mbconv_row_impl(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2,
                                   const float *b2, float *gap, int64_t batch) {
    const int64_t total = batch * d.tiles_x * d.tiles_y * ((d.C + 31) / 32);
    const dim3 grid((unsigned)((total + 3) / 4));
    const int ng = (d.Cin + 7) / 8;
    const int actc = d.act1 == d.act2 && (d.act1 == ACT_RELU || d.act1 == ACT_SILU) ? d.act1 : -1;
#ifdef BN_MBROW_FEW  // tools/mb_probe.cpp: only the ReLU instances (a third of the compile time)
#define ROW_LAUNCH(K, S, NG, IM) hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, IM, ACT_RELU>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total)
#else
#define ROW_LAUNCH(K, S, NG, IM)                                                                                                           \
    do {                                                                                                                                   \
        if (actc == ACT_RELU) hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, IM, ACT_RELU>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else if (actc == ACT_SILU) hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, IM, ACT_SILU>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, IM, -1>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
    } while (0)
#endif
    // (round 5) the expand on the bf16 matrix pipe: the planner's choice (MbDesc::row_b3), 1x1 expands with whole 8-channel groups
#define ROW_LAUNCH3(K, S, NG)                                                                                                              \
    do {                                                                                                                                   \
        if (actc == ACT_RELU) hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, false, ACT_RELU, false, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else if (actc == ACT_SILU) hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, false, ACT_SILU, false, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else hipLaunchKernelGGL((mbconv_row_kernel<K, S, NG, false, -1, false, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
    } while (0)
#define ROW_NG(K, S)                                    \
    do {                                                \
        if (d.row_b3 && d.Cin % 8 == 0) {               \
            if (ng <= 2) ROW_LAUNCH3(K, S, 2);          \
            else if (ng == 3) ROW_LAUNCH3(K, S, 3);     \
            else if (ng == 4) ROW_LAUNCH3(K, S, 4);     \
            else if (ng == 5) ROW_LAUNCH3(K, S, 5);     \
            else ROW_LAUNCH3(K, S, 6);                  \
        } else if (ng <= 2) ROW_LAUNCH(K, S, 2, false); \
        else if (ng == 3) ROW_LAUNCH(K, S, 3, false);   \
        else if (ng == 4) ROW_LAUNCH(K, S, 4, false);   \
        else if (ng == 5) ROW_LAUNCH(K, S, 5, false);   \
        else ROW_LAUNCH(K, S, 6, false);                \
    } while (0)
    // transposed streaming (3 x 3 blocks with a 1x1 expand): the same instances with the second addressing form
#define ROW_LAUNCH_T(S, NG)                                                                                                                \
    do {                                                                                                                                   \
        if (actc == ACT_RELU) hipLaunchKernelGGL((mbconv_row_kernel<3, S, NG, false, ACT_RELU, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else if (actc == ACT_SILU) hipLaunchKernelGGL((mbconv_row_kernel<3, S, NG, false, ACT_SILU, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else hipLaunchKernelGGL((mbconv_row_kernel<3, S, NG, false, -1, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
    } while (0)
#define ROW_LAUNCH_T3(S, NG)                                                                                                               \
    do {                                                                                                                                   \
        if (actc == ACT_RELU) hipLaunchKernelGGL((mbconv_row_kernel<3, S, NG, false, ACT_RELU, true, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else if (actc == ACT_SILU) hipLaunchKernelGGL((mbconv_row_kernel<3, S, NG, false, ACT_SILU, true, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
        else hipLaunchKernelGGL((mbconv_row_kernel<3, S, NG, false, -1, true, true>), grid, dim3(256), 0, s, d, out, in, w1, b1, w2, b2, gap, (int)total); \
    } while (0)
#define ROW_NG_T(S)                             \
    do {                                        \
        if (d.row_b3 && d.Cin % 8 == 0) {       \
            if (ng <= 2) ROW_LAUNCH_T3(S, 2);   \
            else if (ng == 3) ROW_LAUNCH_T3(S, 3); \
            else if (ng == 4) ROW_LAUNCH_T3(S, 4); \
            else if (ng == 5) ROW_LAUNCH_T3(S, 5); \
            else ROW_LAUNCH_T3(S, 6);           \
        } else                                  \
        if (ng <= 2) ROW_LAUNCH_T(S, 2);        \
        else if (ng == 3) ROW_LAUNCH_T(S, 3);   \
        else if (ng == 4) ROW_LAUNCH_T(S, 4);   \
        else if (ng == 5) ROW_LAUNCH_T(S, 5);   \
        else ROW_LAUNCH_T(S, 6);                \
    } while (0)
    if (d.row_tr) {
        if (d.s == 1) ROW_NG_T(1);
        else ROW_NG_T(2);
        return;
    }
    if (d.k1 > 0) {
        if (d.s == 1) {
            if (ng <= 2) ROW_LAUNCH(3, 1, 2, true);
            else if (ng == 3) ROW_LAUNCH(3, 1, 3, true);
            else ROW_LAUNCH(3, 1, 4, true);
        } else {
            if (ng <= 2) ROW_LAUNCH(3, 2, 2, true);
            else if (ng == 3) ROW_LAUNCH(3, 2, 3, true);
            else ROW_LAUNCH(3, 2, 4, true);
        }
    } else if (d.k == 3 && d.s == 1) ROW_NG(3, 1);
    else if (d.k == 3 && d.s == 2) ROW_NG(3, 2);
    else if (d.k == 5 && d.s == 1) ROW_NG(5, 1);
    else ROW_NG(5, 2);
#undef ROW_NG_T
#undef ROW_LAUNCH_T3
#undef ROW_LAUNCH_T
#undef ROW_NG
#undef ROW_LAUNCH3
#undef ROW_LAUNCH
}

}  // namespace bn
