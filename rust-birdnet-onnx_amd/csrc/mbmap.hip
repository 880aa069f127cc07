// Fused MBConv front half for SMALL feature maps (6x32, 3x16: the late stages of the BirdNET / EfficientNet stacks):
// expand 1x1 conv (+bias+act) -> depthwise K x K (+bias+act) -> complete squeeze sums, ONE launch, the expanded
// tensor never leaves the CU.  Replaces a GEMM launch + a depthwise launch and the HBM / L2 round trip between them.
//
// A block owns one sample's WHOLE map and a group of mid channels, which it walks in chunks of NC (32 / 64):
//
//   * the sample's input [H*W][Cin] is fetched ONCE per block into LDS by global_load_lds_dwordx4 (it is one dense
//     block of memory); the chunk's expand filters [NC][Cin] likewise, double buffered -- the next chunk's filters
//     arrive while this chunk is computed.  No staging instruction runs on the vector ALU: on this part the exact-f32
//     matrix instructions and the f32 vector ALU are one resource (tools/mfma_valu_probe.cpp), and the depthwise
//     arithmetic already has to share it;
//   * expand: v_mfma_f32_16x16x4_f32 with the FILTERS as the A operand, so a lane ends up with four consecutive
//     channels of one pixel and writes its activated tile into the chunk image Es with one ds_write_b128 per tile; the
//     accumulators start at the expand bias; fragments of the next 16-wide k group are read while this group multiplies;
//   * Es is [H][W + K - 1][NC] with zero columns left and right: the depthwise window needs no horizontal bound
//     checks (vertical ones are wave-uniform branches); lane = channel (conflict-free LDS reads, 256 / 128 contiguous
//     bytes per stored pixel), every lane group slides the window along PPG consecutive outputs of a row;
//   * the squeeze sums are complete per (sample, channel): the excite kernel adds nothing up (splits = 1).
//
// Arithmetic order per output: expand = bias + k ascending in 16-wide groups (k-slot j of a group: k = 16 g + 4 q + j),
// depthwise = bias2 + taps (ky, kx) ascending -- independent of the batch and of the channel grouping.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "device_common.h"
#include "kernels.h"

namespace bn {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MM_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define MM_GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

__host__ __device__ constexpr int mm_kib(int floats) { return (floats + 255) & ~255; }  // LDS-DMA writes whole 1-KiB pieces

// dense global -> LDS copy of `floats` floats (a multiple of 4) in 1-KiB pieces, piece p by wave p % NWAVES; lanes past
// the end re-read the last 16 bytes into the region's padding (every region is padded to whole pieces)
template <int NWAVES>
__device__ __forceinline__ void mm_copy(float *lds_dst, const float *gsrc, int floats, int wave, int lane) {
    const int n16 = floats >> 2;
    for (int c0 = wave * 64; c0 < n16; c0 += NWAVES * 64) {
        int c = c0 + lane;
        c = c < n16 ? c : n16 - 1;
        __builtin_amdgcn_global_load_lds(MM_GLB_PTR(gsrc + 4 * c), MM_LDS_PTR(lds_dst + 4 * c0), 16, 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void mm_act(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
}

// K x K depthwise, stride S; MW x NW 16x16 tiles per wave, WM x WN waves: the map has exactly 16 MW WM pixels, a chunk
// 16 NW WN channels; PPG outputs per window slide
template <int K, int S, int MW, int NW, int WM, int WN, int PPG>
__global__ __launch_bounds__(64 * WM * WN) void mbmap_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                             const float *__restrict__ w1, const float *__restrict__ b1,
                                                             const float *__restrict__ w2, const float *__restrict__ b2,
                                                             float *__restrict__ gap, int nch) {
    constexpr int NWAVES = WM * WN, T = 64 * NWAVES, HW = 16 * MW * WM, NC = 16 * NW * WN, NG = T / NC;
    constexpr int IWS = (PPG - 1) * S + K;
    extern __shared__ __align__(1024) float mm_lds[];
    const int Cin = d.Cin, WP = d.W + K - 1;
    float *Xs = mm_lds;                                   // [HW][Cin]
    float *Ws = Xs + mm_kib(HW * Cin);                    // [2][NC][Cin]
    const int wsz = mm_kib(NC * Cin);
    float *Es = Ws + 2 * wsz;                             // [H][WP][NC], columns < pl and >= pl + W stay zero
    float *red = Es + mm_kib(d.H * WP * NC);              // [NG][NC]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lc = lane & 15, lq = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    const int64_t b = blockIdx.y;
    const int cbase = blockIdx.x * nch * NC;              // first mid channel of this block
    const int nchunks = min(nch, (d.C - cbase + NC - 1) / NC);

    // ---- prologue: the sample's input and the first filter chunk on their way, the chunk image zeroed meanwhile
    mm_copy<NWAVES>(Xs, in + b * d.in_bs, HW * Cin, wave, lane);
    mm_copy<NWAVES>(Ws, w1 + (int64_t)cbase * Cin, min(NC, d.C - cbase) * Cin, wave, lane);
    for (int i = tid * 4; i < d.H * WP * NC; i += T * 4) *reinterpret_cast<floatx4 *>(Es + i) = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment offsets: this wave's pixels (B operand rows of Xs) and channels (A operand rows of the filter chunk)
    int xrow[MW], epix[MW], wrow[NW];
#pragma unroll
    for (int mt = 0; mt < MW; mt++) {
        const int m = (wm * MW + mt) * 16 + lc;
        xrow[mt] = m * Cin + 4 * lq;
        const int y = m / d.W, x = m - y * d.W;
        epix[mt] = (y * WP + x + d.pl) * NC;
    }
#pragma unroll
    for (int nt = 0; nt < NW; nt++) wrow[nt] = ((wn * NW + nt) * 16 + lc) * Cin + 4 * lq;
    const int G = Cin >> 4;  // 16-wide k groups (Cin % 16 == 0: checked by the planner)

    // depthwise mapping: lane = channel, NG lane groups over the output segments
    const int c = tid % NC, grp = tid / NC;
    const int nsx = (d.OW + PPG - 1) / PPG, nseg = d.OH * nsx;

    for (int ch = 0; ch < nchunks; ch++) {
        const int c0 = cbase + ch * NC;
        float *Wc = Ws + (ch & 1) * wsz;
        // filters of this chunk (and, first time, the input) have landed; every wave is done with the previous chunk
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // per-lane constants of the chunk: expand bias of the lane's four channels per n-tile, depthwise filter of channel c
        floatx4 bias4[NW];
#pragma unroll
        for (int nt = 0; nt < NW; nt++) {
            const int n = c0 + (wn * NW + nt) * 16 + 4 * lq;
            bias4[nt] = (d.has_bias1 && n < d.C) ? *reinterpret_cast<const floatx4 *>(b1 + n) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
        const int cg = c0 + c;
        const bool cact = cg < d.C;
        const int cc = cact ? cg : d.C - 1;
        float wd[K * K];
#pragma unroll
        for (int q = 0; q < K * K; q++) wd[q] = w2[q * d.C + cc];
        const float bz = d.has_bias2 ? b2[cc] : 0.0f;
        floatx4 acc[MW][NW];
#pragma unroll
        for (int mt = 0; mt < MW; mt++)
#pragma unroll
            for (int nt = 0; nt < NW; nt++) acc[mt][nt] = bias4[nt];
        // the next chunk's filters start moving now (their buffer was last read two barriers ago)
        if (ch + 1 < nchunks) mm_copy<NWAVES>(Ws + ((ch + 1) & 1) * wsz, w1 + (int64_t)(c0 + NC) * Cin, min(NC, d.C - c0 - NC) * Cin, wave, lane);

        // ---- expand: D[channel][pixel] += W[channel][k] X[pixel][k]
        floatx4 xa[MW], wa[NW], xb[MW], wb[NW];
#pragma unroll
        for (int mt = 0; mt < MW; mt++) xa[mt] = *reinterpret_cast<const floatx4 *>(Xs + xrow[mt]);
#pragma unroll
        for (int nt = 0; nt < NW; nt++) wa[nt] = *reinterpret_cast<const floatx4 *>(Wc + wrow[nt]);
        for (int g = 0; g < G; g += 2) {
            if (g + 1 < G) {
#pragma unroll
                for (int mt = 0; mt < MW; mt++) xb[mt] = *reinterpret_cast<const floatx4 *>(Xs + xrow[mt] + 16 * (g + 1));
#pragma unroll
                for (int nt = 0; nt < NW; nt++) wb[nt] = *reinterpret_cast<const floatx4 *>(Wc + wrow[nt] + 16 * (g + 1));
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int nt = 0; nt < NW; nt++)
#pragma unroll
                    for (int mt = 0; mt < MW; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][j], xa[mt][j], acc[mt][nt], 0, 0, 0);
            if (g + 1 < G) {
                if (g + 2 < G) {
#pragma unroll
                    for (int mt = 0; mt < MW; mt++) xa[mt] = *reinterpret_cast<const floatx4 *>(Xs + xrow[mt] + 16 * (g + 2));
#pragma unroll
                    for (int nt = 0; nt < NW; nt++) wa[nt] = *reinterpret_cast<const floatx4 *>(Wc + wrow[nt] + 16 * (g + 2));
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int nt = 0; nt < NW; nt++)
#pragma unroll
                        for (int mt = 0; mt < MW; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[nt][j], xb[mt][j], acc[mt][nt], 0, 0, 0);
            }
        }
        {
            float v[MW * NW * 4];
#pragma unroll
            for (int mt = 0; mt < MW; mt++)
#pragma unroll
                for (int nt = 0; nt < NW; nt++)
#pragma unroll
                    for (int i = 0; i < 4; i++) v[(mt * NW + nt) * 4 + i] = acc[mt][nt][i];
            mm_act<MW * NW * 4>(d.act1, d.p0_1, d.p1_1, v);
#pragma unroll
            for (int mt = 0; mt < MW; mt++)
#pragma unroll
                for (int nt = 0; nt < NW; nt++)
                    *reinterpret_cast<floatx4 *>(Es + epix[mt] + (wn * NW + nt) * 16 + 4 * lq) =
                        floatx4{v[(mt * NW + nt) * 4], v[(mt * NW + nt) * 4 + 1], v[(mt * NW + nt) * 4 + 2], v[(mt * NW + nt) * 4 + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the chunk image is complete
        asm volatile("" ::: "memory");

        // ---- depthwise + squeeze
        float sum = 0.0f;
        float *ob = out + b * d.out_bs + cg;
        for (int seg = grp; seg < nseg; seg += NG) {
            const int oy = seg / nsx, ox0 = (seg - oy * nsx) * PPG;
            float ov[PPG];
#pragma unroll
            for (int q = 0; q < PPG; q++) ov[q] = bz;
#pragma unroll
            for (int ky = 0; ky < K; ky++) {
                const int iy = oy * S - d.pt + ky;
                if (iy >= 0 && iy < d.H) {  // uniform over the lane group's wave (a wave holds whole groups)
                    const float *rp = Es + (iy * WP + ox0 * S) * NC + c;
#pragma unroll
                    for (int ix = 0; ix < IWS; ix++) {
                        const float val = rp[ix * NC];
#pragma unroll
                        for (int kx = 0; kx < K; kx++)
                            if (ix - kx >= 0 && (ix - kx) % S == 0 && (ix - kx) / S < PPG) ov[(ix - kx) / S] = fmaf(val, wd[ky * K + kx], ov[(ix - kx) / S]);
                    }
                }
            }
            mm_act<PPG>(d.act2, d.p0_2, d.p1_2, ov);
            if (cact) {
                float *op = ob + (int64_t)(oy * d.OW + ox0) * d.C;
#pragma unroll
                for (int q = 0; q < PPG; q++)
                    if (ox0 + q < d.OW) {
                        op[(int64_t)q * d.C] = ov[q];
                        sum += ov[q];
                    }
            }
        }
        if (d.has_gap) {
            red[grp * NC + c] = sum;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (grp == 0 && cact) {
                float t = red[c];
#pragma unroll
                for (int y = 1; y < NG; y++) t += red[y * NC + c];
                gap[b * d.gap_bs + cg] = t;
            }
        }
    }
}

template <int MW, int NW, int WM, int WN>
size_t cfg_lds(const MbDesc &d) {
    constexpr int HW = 16 * MW * WM, NC = 16 * NW * WN, NG = 64 * WM * WN / NC;
    return (size_t)(mm_kib(HW * d.Cin) + 2 * mm_kib(NC * d.Cin) + mm_kib(d.H * (d.W + d.k - 1) * NC) + NG * NC) * sizeof(float);
}

inline bool mm_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// Which configuration takes this block (0 = none).  Per-sample quantities only.
//   1: 192-pixel map, chunks of 64 channels, 8 waves      2: 192-pixel map, chunks of 32, 8 waves (wider inputs)
//   3: 48-pixel map, chunks of 64 channels, 4 waves
int mbmap_config(const MbDesc &d) {
    static const int mode = getenv("BN_MBMAP2") ? atoi(getenv("BN_MBMAP2")) : 1;
    if (mode == 0) return 0;
    if (d.k1 > 0 || !((d.k == 3 || d.k == 5) && (d.s == 1 || d.s == 2))) return 0;
    if (d.Cin % 16 || d.Cin < 16 || d.C % 4 || d.in_bs % 4 || d.W % 4) return 0;
    if (d.pl < 0 || d.pl > d.k - 1 || d.pt < 0) return 0;
    if (d.OH != (d.H + 2 * d.pt - d.k) / d.s + 1 && d.OH != (d.H + d.s - 1) / d.s) return 0;  // "same" or symmetric padding
    if ((d.OW - 1) * d.s + d.k > d.W + d.k - 1) return 0;                                      // the padded row holds every tap
    if (!mbconv_row_act_supported(d.act1) || !mbconv_row_act_supported(d.act2)) return 0;
    const size_t cap = 160 * 1024;
    const int hw = d.H * d.W;
    if (hw == 192) {
        if (cfg_lds<3, 2, 4, 2>(d) <= cap) return 1;
        if (cfg_lds<3, 1, 4, 2>(d) <= cap) return 2;
    } else if (hw == 48) {
        if (cfg_lds<3, 1, 1, 4>(d) <= cap) return 3;
    }
    return 0;
}

// chunks of channels one block walks (the input is fetched once per block): enough blocks per sample to fill the chip
// at batch 32 with four contexts, few enough that the input fetch is amortised over at least two chunks
int mbmap_chunks_per_block(const MbDesc &d, int cfg) {
    static const int force = getenv("BN_MBMAP2_NCH") ? atoi(getenv("BN_MBMAP2_NCH")) : 0;
    if (force > 0) return force;
    const int nc = cfg == 2 ? 32 : 64;
    const int chunks = (d.C + nc - 1) / nc;
    return cfg == 2 ? (chunks >= 12 ? 3 : 2) : 2;
}

void register_mbmap_kernels() {
#define MM_REG(K, S, MW, NW, WM, WN, PPG) register_dynamic_lds_kernel(reinterpret_cast<const void *>(mbmap_kernel<K, S, MW, NW, WM, WN, PPG>));
#define MM_REG_KS(MW, NW, WM, WN, P1, P2) \
    MM_REG(3, 1, MW, NW, WM, WN, P1) MM_REG(5, 1, MW, NW, WM, WN, P1) MM_REG(3, 2, MW, NW, WM, WN, P2) MM_REG(5, 2, MW, NW, WM, WN, P2)
    MM_REG_KS(3, 2, 4, 2, 8, 4)
    MM_REG_KS(3, 1, 4, 2, 4, 4)
    MM_REG_KS(3, 1, 1, 4, 4, 4)
#undef MM_REG_KS
#undef MM_REG
}

bool launch_mbmap(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                  float *gap, int64_t batch) {
    const int cfg = mbmap_config(d);
    if (!cfg || !mm_al16(in) || !mm_al16(w1) || !mm_al16(b1)) return false;
    const int nch = mbmap_chunks_per_block(d, cfg);
#define MM_GO(K, S, MW, NW, WM, WN, PPG)                                                                                              \
    do {                                                                                                                              \
        constexpr int NC = 16 * NW * WN;                                                                                              \
        dim3 grid((unsigned)((d.C + nch * NC - 1) / (nch * NC)), (unsigned)batch);                                                    \
        const size_t lds_ = cfg_lds<MW, NW, WM, WN>(d);                                                                               \
        hipLaunchKernelGGL((mbmap_kernel<K, S, MW, NW, WM, WN, PPG>), grid, dim3(64 * WM * WN), lds_, s, d, out, in, w1, b1, w2, b2, gap, nch); \
    } while (0)
#define MM_GO_KS(MW, NW, WM, WN, P1, P2)                         \
    do {                                                         \
        if (d.k == 3 && d.s == 1) MM_GO(3, 1, MW, NW, WM, WN, P1);      \
        else if (d.k == 5 && d.s == 1) MM_GO(5, 1, MW, NW, WM, WN, P1); \
        else if (d.k == 3) MM_GO(3, 2, MW, NW, WM, WN, P2);             \
        else MM_GO(5, 2, MW, NW, WM, WN, P2);                           \
    } while (0)
    if (cfg == 1) MM_GO_KS(3, 2, 4, 2, 8, 4);
    else if (cfg == 2) MM_GO_KS(3, 1, 4, 2, 4, 4);
    else MM_GO_KS(3, 1, 1, 4, 4, 4);
#undef MM_GO_KS
#undef MM_GO
    return true;
}

}  // namespace bn
