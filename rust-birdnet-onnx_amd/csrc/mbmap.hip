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
//   * Es is [H][W + K - 1][NC + 4] with zero columns left and right (no horizontal bound checks; the 4 floats of
//     padding per pixel make the tile stores conflict free); depthwise: lane = channel (conflict-free LDS reads, 256 /
//     128 contiguous bytes per stored pixel), lane group = a strip of PPG output columns over ALL rows: every input
//     row of the strip is read from LDS once and feeds the accumulators of the <= K output rows it touches.  The map
//     size and the padding are template parameters, so the whole phase is straight-line code -- no division, no
//     branch, every (input row, tap) -> output row relation resolved at compile time (the first version walked
//     row segments with run-time bounds: 5 - 7 us per chunk against 3 us for the expand it follows);
//   * the squeeze sums are complete per (sample, channel): the excite kernel adds nothing up (splits = 1).
//
// Round 4 -- three generalisations, all in this one kernel (plan_rules.h, mbmap_shape):
//   * BANDS (NB = 2, HM = 8: BirdNET v3.0's 8 x 32 stage, Perch's 32 x 8 one walked transposed): a sample's map is cut into NB bands
//     of OHM / NB output rows; a band is a block of its own (blockIdx.z) that loads the H = 6 input rows its outputs reach.  Those are
//     REAL rows of the map (the first loaded row gy0 is clamped into the map), so nothing is padded at run time: which (input
//     row, tap row) pairs feed which output row is still resolved at compile time, per band (ROFF = PT + gy0 - band * OHB * S);
//     the squeeze sums are partial per band (gap[b][band][C], the excite kernel adds the NB partials in band order);
//   * TRANSPOSED maps (d.map_tr, run time): the kernel's row index is the map's x -- only the input gather, the output
//     address and the order of the depthwise taps change;
//   * PADDED k (d.cin_pad > d.Cin, run time): rows of the LDS images are cin_pad floats, the chunks past Cin of an input row
//     come from a page of zeros (the LDS-DMA source address is per lane), the planner pads the filter rows with zeros.
//
// Arithmetic order per output: expand = bias + k ascending in 16-wide groups (k-slot j of a group: k = 16 g + 4 q + j),
// depthwise = bias2 + taps (ky, kx) ascending -- independent of the batch and of the channel grouping.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "bf16x3.h"
#include "device_common.h"
#include "kernels.h"
#include "plan_rules.h"

namespace bn {
namespace {

#include "mbmap_common.h"

// K x K depthwise, stride S; MW x NW 16x16 tiles per wave, WM x WN waves per K slice, KSP K slices (KSP = 2: the k groups of
// the expand are split between two sets of waves whose partial tiles are added through the chunk image -- twice the
// waves for the same LDS, used where a map of 48 pixels gives four waves too little to hide anything); the map has
// exactly H W = 16 MW WM pixels, a chunk 16 NW WN channels
//
// NSW > 0 (round 5): the expand on the BF16 matrix pipe with f32-complete products (bf16x3.h).  A wave keeps the three bf16 planes of ITS
// pixels' input rows in registers for the whole block (MW NSW fragments of 32-deep k, read from an LDS image that exists only during the
// prologue, split once), the planner stores the filters in fragment order (pack_mbmap_w3f: a chunk is one dense copy) and a filter
// fragment is split as it is read -- 44 vector instructions against the 6 MW matrix instructions it feeds, on the other pipe.  NSW =
// 32-deep steps per wave (all of them, or this K slice's half).  Arithmetic per expanded value: bias + steps ascending, inside a step the
// six partial products in bf16x3.h's fixed order; the depthwise half is unchanged.
// MEASURED (tools/mbmap_phases.py, v2.4 at batch 32, one context; exact-f32 -> bf16x3): the expand phase of the 6 x 32 x 112 launch
// 9.1 -> 4.5 us, of the 3 x 16 x 192 one 7.5 -> 3.2 us; the launches 24.1 -> 22.0 and 18.3 -> 16.4 us -- the prologue grew by 2.2 us
// (the split of the block's input: 96 values per lane, 5.5 vector instructions each, both waves of a SIMD), which three chunks per
// block do not amortise well; marginal cost of the ten launches per 32 more segments 106 -> 76 us; four contexts 64.6 -> 65.7 k
// segments/s (v3.0 at batch 64: 53.6 -> 54.5 k).  More chunks per block (BN_MBMAP2_NCH=6) buy another 1.5 % with four contexts and
// cost 130 us of the one-context chain: not taken.
template <int K, int S, int MW, int NW, int WM, int WN, int KSP, int H, int W, bool SWZ16, int NB = 1, int HM = H, int NSW = 0>
__global__ __launch_bounds__(64 * WM * WN * KSP) void mbmap_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in,
                                                                   const float *__restrict__ w1, const float *__restrict__ b1,
                                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                                   float *__restrict__ gap, int nch, uint32_t inv_ch, const float *__restrict__ zpage) {
    constexpr int WPS = WM * WN, NWAVES = WPS * KSP, T = 64 * NWAVES, HW = 16 * MW * WM, NC = 16 * NW * WN, NG = T / NC;
    constexpr bool B3 = NSW > 0;
    constexpr int NST = NSW * KSP;  // 32-deep k steps of the whole product (B3)
    static_assert(H * W == HW, "the band is exactly the pixels of the wave tiles");
    static_assert(KSP == 1 || (KSP == 2 && (SWZ16 || B3)), "the K split walks the 4-group blocks of the SWZ16 layout");
    static_assert(MW * NSW <= 12, "the input planes of a wave: 12 registers per fragment");
    static_assert((W & (W - 1)) == 0 && NB >= 1 && NB <= 2 && (NB > 1 || HM == H), "bands: one or two, of a map HM rows high");
    constexpr int PT = (K - 1) / 2;  // padding on every side (checked by mbmap_shape)
    constexpr int OHM = (HM + 2 * PT - K) / S + 1, OH = OHM / NB, OW = (W + 2 * PT - K) / S + 1;  // OH: output rows of ONE band
    static_assert(OHM % NB == 0 && mm_bands_ok(K, S, H, HM, NB), "a band's outputs reach only the rows its block loads");
    static_assert(OW % NG == 0, "one strip of output columns per lane group");
    constexpr int PPG = OW / NG, IWS = (PPG - 1) * S + K, WP = W + K - 1;
    constexpr int EP = NC + 4;  // floats per pixel of the chunk image: 4 of padding make the tile stores conflict free
    extern __shared__ __align__(1024) float mm_lds[];
    const int Cin = d.cin_pad, CH = Cin >> 2;             // floats / chunks per LDS row (the padded k; == d.Cin unless the planner padded)
    const int tr = d.map_tr;
    float *Xs = mm_lds;                                   // [HW][Cin]                      (B3: absent)
    float *Ws = B3 ? mm_lds : Xs + mm_kib(HW * Cin);      // [2][NC][Cin]                   (B3: [2][NC / 16][NST][128 chunks])
    const int wsz = B3 ? mm_kib(NC * 32 * NST) : mm_kib(NC * Cin);
    float *Es = Ws + 2 * wsz;                             // [H][WP][EP], columns < PT and >= PT + W stay zero
    float *red = Es + mm_kib(H * WP * EP);                // [NG][NC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar registers, scalar branches
    const int lc = lane & 15, lq = lane >> 4;
    const int kh = wave / WPS, w4 = wave % WPS;           // K slice, wave inside the slice
    const int wm = w4 % WM, wn = w4 / WM;
    const int64_t b = blockIdx.y;
    const int band = NB > 1 ? (int)blockIdx.z : 0;        // (block-uniform)
    const int gy0 = NB > 1 ? min(max(band * OH * S - PT, 0), HM - H) : 0;  // first map row of the band's image
    const int cbase = blockIdx.x * nch * NC;              // first mid channel of this block
    const int nchunks = min(nch, (d.C - cbase + NC - 1) / NC);

    if (d.dbg & 32) return;  // (tools/mbmap_phases.py: the cost of the empty launch)
    // ---- prologue: the sample's input and the first filter chunk on their way, the padding of the chunk image zeroed
    b3_u32x4 xh[B3 ? MW : 1][B3 ? NSW : 1], xm[B3 ? MW : 1][B3 ? NSW : 1], xl[B3 ? MW : 1][B3 ? NSW : 1];
    floatx4 raw[B3 ? MW : 1][B3 ? NSW : 1][2];
    if constexpr (B3) {
        // The sample's input goes through LDS once: the f32 form's dense copy (whole cache lines, 1 KiB per wave instruction) into an image
        // that ALIASES the second filter buffer, the chunk image and the squeeze partials -- it is dead before any of them is written.
        // Lane (c, q) of fragment (mt, s) then reads k groups 2 s' and 2 s' + 1 (s' = kh NSW + s) of pixel 16 (wm MW + mt) + c exactly as the
        // f32 form reads them (same swizzle, conflict free): its eight values are k = 32 s' + 4 q .. + 3 and 32 s' + 16 + 4 q .. + 3 -- the k
        // order inside a step is free as long as the filters follow it (pack_mbmap_w3f does).  (Loading the fragments straight from global
        // memory -- one pixel row per lane -- cost 3 us per launch: 64 separate 16-byte requests per instruction.)
        float *Xi = mm_lds + wsz;
        mm_copy_in<NWAVES, SWZ16, W, HM>(Xi, in + b * d.in_bs, zpage, HW, CH, d.Cin >> 2, d.Cin, inv_ch, gy0, tr, wave, lane);
        mm_copy_w3<NWAVES>(Ws, w1 + (int64_t)(cbase / 16) * (NST * 512), NC / 16 * NST, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int G16 = Cin >> 4;
#pragma unroll
        for (int mt = 0; mt < MW; mt++) {
            const int m = (wm * MW + mt) * 16 + (lane & 15);
            const int sw = mm_swz<SWZ16>(m);
#pragma unroll
            for (int st = 0; st < NSW; st++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int g = 2 * (kh * NSW + st) + h;
                    const int off = SWZ16 ? m * Cin + 64 * (g >> 2) + 16 * ((g & 3) ^ (sw >> 2)) + 4 * ((lane >> 4) ^ (sw & 3))
                                          : m * Cin + 16 * g + 4 * ((lane >> 4) ^ sw);
                    raw[mt][st][h] = (g < G16 && !(d.dbg & 8)) ? *reinterpret_cast<const floatx4 *>(Xi + off) : floatx4{0.f, 0.f, 0.f, 0.f};
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has its fragments: the image's space is free for the chunk image and the filter ring
        asm volatile("" ::: "memory");
    } else {
        mm_copy_in<NWAVES, SWZ16, W, HM>(Xs, in + b * d.in_bs, zpage, HW, CH, d.Cin >> 2, d.Cin, inv_ch, gy0, tr, wave, lane);
        mm_copy<NWAVES, SWZ16>(Ws, w1 + (int64_t)cbase * Cin, min(NC, d.C - cbase), CH, inv_ch, wave, lane);
    }
    // the K - 1 padding columns of every row of the chunk image are zero and stay zero (the expand writes the interior)
    for (int i = tid; i < H * (K - 1) * (EP / 4); i += T) {
        const int q4 = i % (EP / 4), pc = (i / (EP / 4)) % (K - 1), y = i / ((EP / 4) * (K - 1));
        const int xcol = pc < PT ? pc : W + pc;
        *reinterpret_cast<floatx4 *>(Es + (y * WP + xcol) * EP + 4 * q4) = floatx4{0.f, 0.f, 0.f, 0.f};
    }

    // fragment offsets: this wave's pixels (B operand rows of Xs) and channels (A operand rows of the filter chunk).
    // SWZ16: group gg of a 256-byte block sits in chunk slots 4 (gg ^ (swz >> 2)) + (lq ^ (swz & 3)); a wave reads NGG
    // groups per block (all four, or its slice's two); else one offset, the k group adds 16 floats
    constexpr int NGG = SWZ16 ? 4 / KSP : 1;
    int xrow[MW][NGG], epix[MW], wrow[NW][NGG];
#pragma unroll
    for (int mt = 0; mt < MW; mt++) {
        const int m = (wm * MW + mt) * 16 + lc;
        const int sw = mm_swz<SWZ16>(m);
#pragma unroll
        for (int gi = 0; gi < NGG; gi++) xrow[mt][gi] = m * Cin + 16 * ((NGG * kh + gi) ^ (sw >> 2)) + 4 * (lq ^ (sw & 3));
        const int y = m / W, x = m - y * W;
        epix[mt] = (y * WP + x + PT) * EP;
    }
#pragma unroll
    for (int nt = 0; nt < NW; nt++) {
        const int r = (wn * NW + nt) * 16 + lc;
        const int sw = mm_swz<SWZ16>(r);
#pragma unroll
        for (int gi = 0; gi < NGG; gi++) wrow[nt][gi] = r * Cin + 16 * ((NGG * kh + gi) ^ (sw >> 2)) + 4 * (lq ^ (sw & 3));
    }
    const int G = Cin >> 4;  // 16-wide k groups (Cin % 16 == 0; SWZ16: Cin % 64 == 0, so G % 4 == 0)

    // depthwise mapping: lane = channel, lane group = strip of PPG output columns
    const int c = tid % NC, grp = tid / NC;
    const int ox0 = grp * PPG;

    // per-lane constants of a chunk (expand bias of the lane's four channels per n-tile, depthwise filter and bias of
    // channel c), fetched one chunk AHEAD: their load latency is hidden behind the previous chunk
    struct ChunkConst {
        floatx4 bias4[NW];
        float wd[K * K], bz;
    };
    auto fetch_bias = [&](ChunkConst &cc_, int c0) {
#pragma unroll
        for (int nt = 0; nt < NW; nt++) {
            const int n = c0 + (wn * NW + nt) * 16 + 4 * lq;
            cc_.bias4[nt] = (d.has_bias1 && kh == 0 && n < d.C) ? *reinterpret_cast<const floatx4 *>(b1 + n) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto fetch_dw = [&](ChunkConst &cc_, int c0) {
        // (uniform base + unsigned 32-bit lane offset: the form the compiler can issue as a scalar-base load -- with a signed index it
        // sign-extended and added 64 bits on the vector ALU for every tap, 110 of the chunk's 800 vector instructions)
        const unsigned cl = (unsigned)min(c0 + c, d.C - 1);
#pragma unroll
        for (int q = 0; q < K * K; q++) cc_.wd[q] = (w2 + (size_t)(tr ? (q % K) * K + q / K : q) * (size_t)d.C)[cl];  // kernel tap (ky, kx) = map tap (kx, ky) when transposed
        cc_.bz = d.has_bias2 ? b2[cl] : 0.0f;
    };
    // f32 form: everything one chunk ahead.  bf16x3 form: the expand bias one chunk ahead, the depthwise constants of a chunk at ITS start
    // (they are first used behind the expand phase, which hides the loads just as well) -- one set of K K + 1 registers instead of two
    // beside the input planes
    auto fetch = [&](ChunkConst &cc_, int c0) {
        fetch_bias(cc_, c0);
        if constexpr (!B3) fetch_dw(cc_, c0);
    };
    ChunkConst nxt;
    fetch(nxt, cbase);
    if constexpr (B3) {  // the split of the input fragments runs while the chunk constants requested above are on their way
#pragma unroll
        for (int mt = 0; mt < MW; mt++)
#pragma unroll
            for (int st = 0; st < NSW; st++) {
                if (d.dbg & 16) {
                    xh[mt][st] = xm[mt][st] = xl[mt][st] = __builtin_bit_cast(b3_u32x4, raw[mt][st][0] + raw[mt][st][1]);
                } else {
                    split3(raw[mt][st][0], raw[mt][st][1], xh[mt][st], xm[mt][st], xl[mt][st]);
                }
            }
    }
    // the first filter chunk and the input have landed, the padding is written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if (d.dbg & 64) return;  // (... of launch + prologue)
    for (int ch = 0; ch < nchunks; ch++) {
        const int c0 = cbase + ch * NC;
        float *Wc = Ws + (ch & 1) * wsz;
        ChunkConst cur = nxt;
        if constexpr (B3) fetch_dw(cur, c0);
        const int cg = c0 + c;
        const bool cact = cg < d.C;
        // the next chunk's filters and constants start moving now (that filter buffer was last read two barriers ago)
        if (ch + 1 < nchunks) {
            if constexpr (B3) mm_copy_w3<NWAVES>(Ws + ((ch + 1) & 1) * wsz, w1 + (int64_t)((c0 + NC) / 16) * (NST * 512), NC / 16 * NST, wave, lane);
            else mm_copy<NWAVES, SWZ16>(Ws + ((ch + 1) & 1) * wsz, w1 + (int64_t)(c0 + NC) * Cin, min(NC, d.C - c0 - NC), CH, inv_ch, wave, lane);
            fetch(nxt, c0 + NC);
        }
        floatx4 acc[MW][NW];
#pragma unroll
        for (int mt = 0; mt < MW; mt++)
#pragma unroll
            for (int nt = 0; nt < NW; nt++) acc[mt][nt] = cur.bias4[nt];  // (slices past the first start at zero: fetch)

        // ---- expand: D[channel][pixel] += W[channel][k] X[pixel][k]; the fragments of the next group are read while this
        // group multiplies (two register sets)
        floatx4 xa[MW], wa[NW], xb[MW], wb[NW];
        auto rd = [&](floatx4 (&xf)[MW], floatx4 (&wf)[NW], int blk, int gi) {  // gi: compile-time at every call site
#pragma unroll
            for (int mt = 0; mt < MW; mt++) xf[mt] = *reinterpret_cast<const floatx4 *>(Xs + xrow[mt][gi] + blk);
#pragma unroll
            for (int nt = 0; nt < NW; nt++) wf[nt] = *reinterpret_cast<const floatx4 *>(Wc + wrow[nt][gi] + blk);
        };
        auto mm = [&](const floatx4 (&xf)[MW], const floatx4 (&wf)[NW]) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int nt = 0; nt < NW; nt++)
#pragma unroll
                    for (int mt = 0; mt < MW; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][j], xf[mt][j], acc[mt][nt], 0, 0, 0);
        };
        if (d.dbg & 1) {
        } else if constexpr (B3) {
            // step s, channel tile nt: the filter fragment (two lane-linear reads, fetched one fragment ahead), split, six products per pixel tile
            floatx4 wr[2][2];
            auto rdw = [&](floatx4 (&r)[2], int f) {  // f = st * NW + nt, compile time at every call site
                const float *wb = Wc + (((wn * NW + f % NW) * NST + kh * NSW + f / NW) * 128 + lane) * 4;
                r[0] = *reinterpret_cast<const floatx4 *>(wb);
                r[1] = *reinterpret_cast<const floatx4 *>(wb + 256);
            };
            rdw(wr[0], 0);
#pragma unroll
            for (int f = 0; f < NSW * NW; f++) {
                if (f + 1 < NSW * NW) rdw(wr[(f + 1) & 1], f + 1);
                b3_u32x4 wh, wmid, wl;
                split3(wr[f & 1][0], wr[f & 1][1], wh, wmid, wl);
#pragma unroll
                for (int mt = 0; mt < MW; mt++) acc[mt][f % NW] = mm6(wh, wmid, wl, xh[mt][f / NW], xm[mt][f / NW], xl[mt][f / NW], acc[mt][f % NW]);
            }
        } else if constexpr (SWZ16) {
            // G % 4 == 0: one 256-byte block of four groups per trip; this slice takes NGG of them, two at a time
            for (int g = 0; g < G; g += 4) {
#pragma unroll
                for (int gi = 0; gi < NGG; gi += 2) {
                    rd(xa, wa, 16 * g, gi);
                    rd(xb, wb, 16 * g, gi + 1);
                    mm(xa, wa);
                    mm(xb, wb);
                }
            }
        } else {
            int g = 0;
            for (; g + 1 < G; g += 2) {
                rd(xa, wa, 16 * g, 0);
                rd(xb, wb, 16 * g + 16, 0);
                mm(xa, wa);
                mm(xb, wb);
            }
            if (g < G) {
                rd(xa, wa, 16 * g, 0);
                mm(xa, wa);
            }
        }
        if constexpr (KSP == 2) {
            // the second slice parks its partial tile in the chunk image, the first adds it (slice 0 + slice 1) and goes on
            if (kh == 1) {
#pragma unroll
                for (int mt = 0; mt < MW; mt++)
#pragma unroll
                    for (int nt = 0; nt < NW; nt++) *reinterpret_cast<floatx4 *>(Es + epix[mt] + (wn * NW + nt) * 16 + 4 * lq) = acc[mt][nt];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kh == 0) {
#pragma unroll
                for (int mt = 0; mt < MW; mt++)
#pragma unroll
                    for (int nt = 0; nt < NW; nt++) acc[mt][nt] += *reinterpret_cast<const floatx4 *>(Es + epix[mt] + (wn * NW + nt) * 16 + 4 * lq);
            }
        }
        if (kh == 0) {
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
        // the chunk image is complete -- and this is where every wave waits for the NEXT chunk's filters (requested a whole
        // expand phase ago): only loads are outstanding here.  Waiting for them at the end of the chunk, behind the depthwise
        // phase's result stores, made every chunk wait for a store round trip as well (vmcnt counts both).
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        // ---- depthwise + squeeze: all OH x PPG outputs of the strip in registers, input rows read once each.  ROFF = PT + gy0 -
        // band * OH * S places the band's outputs on its image rows: output row oy (of the band) takes image row oy * S + ky - ROFF
        // where that row exists (compile time; for a whole map ROFF = PT and the skipped rows are the zero padding)
        float sum = 0.0f;
        auto dw_phase = [&](auto roff_c) {
            constexpr int ROFF = decltype(roff_c)::value;
            float ov[OH][PPG];
#pragma unroll
            for (int oy = 0; oy < OH; oy++)
#pragma unroll
                for (int q = 0; q < PPG; q++) ov[oy][q] = cur.bz;
            const float *rp0 = Es + (ox0 * S) * EP + c;
#pragma unroll
            for (int iy = 0; iy < H; iy++) {
                bool used = false;  // (compile time) an image row no output of the band reaches is not read
#pragma unroll
                for (int ky = 0; ky < K; ky++) {
                    const int t = iy + ROFF - ky;
                    used = used || (t >= 0 && t % S == 0 && t / S < OH);
                }
                if (!used) continue;
                float val[IWS];
#pragma unroll
                for (int ix = 0; ix < IWS; ix++) val[ix] = rp0[(iy * WP + ix) * EP];
#pragma unroll
                for (int ky = 0; ky < K; ky++) {
                    const int t = iy + ROFF - ky;  // = oy * S for the output row this (image row, tap row) pair feeds
                    if (t >= 0 && t % S == 0 && t / S < OH) {  // compile time after unrolling
#pragma unroll
                        for (int q = 0; q < PPG; q++)
#pragma unroll
                            for (int kx = 0; kx < K; kx++) ov[t / S][q] = fmaf(val[q * S + kx], cur.wd[ky * K + kx], ov[t / S][q]);
                    }
                }
            }
            // output element (row, column) of the kernel's geometry: + row * o_rs + column * o_cs floats (transposed: the map's (x, y))
            float *ob = out + b * d.out_bs;                                  // uniform
            const unsigned o_cs = tr ? (unsigned)(OHM * d.C) : (unsigned)d.C;
            const unsigned o_rs = tr ? (unsigned)d.C : (unsigned)(OW * d.C);
            const unsigned olane = (unsigned)cg + (unsigned)ox0 * o_cs;        // this lane's channel + strip offset
#pragma unroll
            for (int oy = 0; oy < OH; oy++) {
                float r[PPG];
#pragma unroll
                for (int q = 0; q < PPG; q++) r[q] = ov[oy][q];
                mm_act<PPG>(d.act2, d.p0_2, d.p1_2, r);
                if (cact) {
#pragma unroll
                    for (int q = 0; q < PPG; q++) {
                        if (!(d.dbg & 4)) (ob + (size_t)((unsigned)(band * OH + oy) * o_rs + (unsigned)q * o_cs))[olane] = r[q];
                        sum += r[q];
                    }
                }
            }
        };
        if (!(d.dbg & 2)) {
            if constexpr (NB == 1) {
                dw_phase(std::integral_constant<int, PT>{});
            } else {
                constexpr int G1 = (OH * S - PT) < 0 ? 0 : ((OH * S - PT) > HM - H ? HM - H : (OH * S - PT));  // gy0 of band 1
                if (band == 0) dw_phase(std::integral_constant<int, PT>{});               // gy0 = 0
                else dw_phase(std::integral_constant<int, PT + G1 - OH * S>{});
            }
        }
        if (d.has_gap) red[grp * NC + c] = sum;
        // ONE barrier ends the chunk: the squeeze partials are written and every wave is done reading the chunk image (LDS
        // only: the result stores stay in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (d.has_gap && grp == 0 && cact) {
            float t = red[c];
#pragma unroll
            for (int y = 1; y < NG; y++) t += red[y * NC + c];
            gap[b * d.gap_bs + (int64_t)band * d.C + cg] = t;  // (read before the next chunk writes `red`: that happens behind its own barrier)
        }
    }
}

inline bool mm_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// chunks of channels one block walks (the input is fetched once per block): the fewest blocks that still give every CU
// one -- the blocks are LDS-bound to one per CU, so more blocks than CUs means a second round (measured, batch 32:
// 1152 channels as 288 blocks 32 us, as 192 blocks 25 us) and fewer blocks amortise the input fetch better.  The
// grouping does not enter the arithmetic (squeeze sums are complete per channel inside a block).
// "Every CU" is the device's when ONE context runs on it.  With several contexts in flight a launch gets a share of the CUs and runs
// its blocks in rounds anyway, while every block pays the prologue (with the bf16x3 forms: input image, fragment reads, the split -- 40 %
// of a block at three chunks); sized for HALF the CUs the v2.4 step with four contexts gains 5 % (65.5 -> 68.6 k segments/s; a third:
// 68.7 k, a quarter: 69.0 k) and one launch alone loses 30 % (160 -> 209 us for the ten launches; a third: 255 us) -- so the share is 2
// as soon as a second live context exists on the device (capi.cpp counts them; BN_MBMAP_SHARE=n fixes it, 1 = the latency form).
int mbmap_chunks_per_block(const MbDesc &d, const MbmapShape &sh, int64_t batch) {
    const int force = getenv("BN_MBMAP2_NCH") ? atoi(getenv("BN_MBMAP2_NCH")) : 0;
    const int nc = (!d.map_ws && (sh.cfg == 1 || sh.cfg == 3)) ? 64 : 32;
    const int chunks = (d.C + nc - 1) / nc;
    if (force > 0) return std::min(force, chunks);
    const int share_env = getenv("BN_MBMAP_SHARE") ? atoi(getenv("BN_MBMAP_SHARE")) : 0;
    const int share = share_env > 0 ? share_env : std::min(device_context_count(), 2);
    const int64_t ncu = std::max<int64_t>(1, device_cu_count() / share);
    return (int)std::max<int64_t>(1, std::min<int64_t>(chunks, (chunks * batch * sh.bands + ncu - 1) / ncu));
}

void register_mbmap_kernels() {
#define MM_REG(K, S, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW) \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(mbmap_kernel<K, S, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW>));
#define MM_REG_KS(MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW)                                                          \
    MM_REG(3, 1, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW) MM_REG(5, 1, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW) \
    MM_REG(3, 2, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW) MM_REG(5, 2, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW)
    MM_REG_KS(3, 2, 4, 2, 1, 6, 32, false, 1, 6, 0)
    MM_REG_KS(3, 1, 4, 2, 1, 6, 32, false, 1, 6, 0)
    MM_REG_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 0)
    MM_REG(3, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 0) MM_REG(5, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 0)
    MM_REG_KS(3, 1, 4, 2, 1, 6, 32, false, 2, 8, 0)                                                          // cfg 5: 8 x 32 in two bands
    MM_REG(3, 1, 2, 1, 2, 2, 1, 4, 16, false, 1, 4, 0) MM_REG(5, 1, 2, 1, 2, 2, 1, 4, 16, false, 1, 4, 0)  // cfg 6
    // bf16x3 expand (plan_rules.h mbmap_b3_steps)
    MM_REG_KS(3, 2, 4, 2, 1, 6, 32, false, 1, 6, 2) MM_REG_KS(3, 2, 4, 2, 1, 6, 32, false, 1, 6, 3)
    MM_REG_KS(3, 1, 4, 2, 1, 6, 32, false, 1, 6, 3) MM_REG_KS(3, 1, 4, 2, 1, 6, 32, false, 1, 6, 4)
    MM_REG_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 2) MM_REG_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 3) MM_REG_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 4)
    MM_REG(3, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 2) MM_REG(5, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 2)
    MM_REG(3, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 3) MM_REG(5, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 3)
    MM_REG(3, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 4) MM_REG(5, 1, 2, 1, 2, 2, 2, 4, 16, true, 1, 4, 4)
#undef MM_REG_KS
#undef MM_REG
}

bool launch_mbmap(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                  float *gap, int64_t batch) {
    const MbmapShape sh = mbmap_shape(d);
    const float *zpage = device_zero_page();
    if (!sh.cfg || !mm_al16(in) || !mm_al16(w1) || !mm_al16(b1) || !zpage) return false;
    // the plan was built under the same rules: a descriptor whose padding / transposition / bands disagree with them is refused
    if (d.cin_pad != sh.cin_pad || d.map_tr != sh.tr || d.map_bands != sh.bands) return false;
    const int nsw = d.map_b3;
    if (nsw && nsw != mbmap_b3_steps(d, sh)) return false;
    if (d.map_ws) return d.map_ws == mbmap_ws_steps(d, sh) && mm_al16(in) && launch_mbmap_ws(s, d, out, in, w1, b1, w2, b2, gap, batch, mbmap_chunks_per_block(d, sh, batch));
    MbDesc dd = d;
    dd.dbg = getenv("BN_MM_DBG") ? atoi(getenv("BN_MM_DBG")) : 0;
    const int nch = mbmap_chunks_per_block(d, sh, batch);
    const int nst = nsw * (sh.cfg >= 3 ? 2 : 1);
    const uint32_t inv_ch = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(sh.cin_pad / 4)) + 1u;  // slot -> row of the swizzled copies
    MbDesc lds_d = d;  // LDS sizes: padded rows, the kernel's geometry (transposed maps: H <-> W), a band's rows
    lds_d.Cin = sh.cin_pad;
    if (sh.tr) std::swap(lds_d.H, lds_d.W);
    if (sh.bands > 1) lds_d.H = 6;
#define MM_GO(K, S, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW)                                                                                \
    do {                                                                                                                              \
        constexpr int NC = 16 * NW * WN;                                                                                              \
        dim3 grid((unsigned)((d.C + nch * NC - 1) / (nch * NC)), (unsigned)batch, (unsigned)NB);                                      \
        const size_t lds_ = NSW ? mbmap_lds_bytes_b3(lds_d, nst, NW, WM, WN, KSP) : mbmap_lds_bytes(lds_d, MW, NW, WM, WN, KSP);      \
        hipLaunchKernelGGL((mbmap_kernel<K, S, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW>), grid, dim3(64 * WM * WN * KSP), lds_, s, dd, out, in, w1, \
                           b1, w2, b2, gap, nch, inv_ch, zpage);                                                                      \
    } while (0)
#define MM_GO_KS(MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW)                                    \
    do {                                                                                       \
        if (d.k == 3 && d.s == 1) MM_GO(3, 1, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW);      \
        else if (d.k == 5 && d.s == 1) MM_GO(5, 1, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW); \
        else if (d.k == 3) MM_GO(3, 2, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW);             \
        else MM_GO(5, 2, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW);                           \
    } while (0)
#define MM_GO_K31(MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW)                     \
    do {                                                                         \
        if (d.k == 3) MM_GO(3, 1, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW);    \
        else MM_GO(5, 1, MW, NW, WM, WN, KSP, H, W, SW, NB, HM, NSW);             \
    } while (0)
    if (nsw) {
        if (sh.cfg == 1 && nsw == 2) MM_GO_KS(3, 2, 4, 2, 1, 6, 32, false, 1, 6, 2);
        else if (sh.cfg == 1) MM_GO_KS(3, 2, 4, 2, 1, 6, 32, false, 1, 6, 3);
        else if (sh.cfg == 2 && nsw == 3) MM_GO_KS(3, 1, 4, 2, 1, 6, 32, false, 1, 6, 3);
        else if (sh.cfg == 2) MM_GO_KS(3, 1, 4, 2, 1, 6, 32, false, 1, 6, 4);
        else if (sh.cfg == 3 && nsw == 2) MM_GO_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 2);
        else if (sh.cfg == 3 && nsw == 3) MM_GO_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 3);
        else if (sh.cfg == 3) MM_GO_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 4);
        else if (nsw == 2) MM_GO_K31(2, 1, 2, 2, 2, 4, 16, true, 1, 4, 2);
        else if (nsw == 3) MM_GO_K31(2, 1, 2, 2, 2, 4, 16, true, 1, 4, 3);
        else MM_GO_K31(2, 1, 2, 2, 2, 4, 16, true, 1, 4, 4);
    } else if (sh.cfg == 1) MM_GO_KS(3, 2, 4, 2, 1, 6, 32, false, 1, 6, 0);
    else if (sh.cfg == 2) MM_GO_KS(3, 1, 4, 2, 1, 6, 32, false, 1, 6, 0);
    else if (sh.cfg == 3) MM_GO_KS(3, 1, 1, 4, 2, 3, 16, true, 1, 3, 0);
    else if (sh.cfg == 4) MM_GO_K31(2, 1, 2, 2, 2, 4, 16, true, 1, 4, 0);
    else if (sh.cfg == 5) MM_GO_KS(3, 1, 4, 2, 1, 6, 32, false, 2, 8, 0);
    else MM_GO_K31(2, 1, 2, 2, 1, 4, 16, false, 1, 4, 0);
#undef MM_GO_K31
#undef MM_GO_KS
#undef MM_GO
    return true;
}

}  // namespace bn
