// Fused MBConv front half for the small maps (6 x 32; 8 x 32 in two bands; 3 x 16 and 4 x 16 with Cin = 128 / 192), WAVE-SPECIALISED (round 5): expand 1x1 conv (+bias+act) on the bf16 matrix pipe (bf16x3.h)
// and depthwise K x K (+bias+act) + squeeze sums on the vector ALU run in the SAME phase, on different waves and different chunks.
//
// mbmap.hip walks a block's channel chunks in two phases per chunk -- every wave expands, barrier, every wave runs the depthwise window,
// barrier -- so the matrix pipe idles through the depthwise phase and the vector ALU through the expand (tools/mbmap_phases.py: of the
// 22 us of the 6 x 32 x 112 launch at batch 32, 4.5 are expand, 5.3 depthwise, 4.8 the loop's waits and barriers).  Here a block is
// eight waves of two kinds:
//   * waves 0-3 (one per SIMD) EXPAND chunk p: each owns three pixel tiles (48 of the map's 192 pixels) and both 16-channel tiles of
//     the 32-channel chunk; the three bf16 planes of its pixels' input rows sit in registers for the whole block (read from an LDS
//     image that exists only in the prologue, split once), the filter chunk arrives as bf16 planes in fragment order by LDS-DMA (pack_mbmap_w3p;
//     these waves issue the copy of chunk p + 1 at the start of phase p and are the only ones that wait for it -- they store nothing to
//     memory, so their vmcnt wait never includes a result store's round trip); the activated tiles go
//     into chunk image p mod 2;
//   * waves 4-7 (the other wave of each SIMD) run the DEPTHWISE window of chunk p - 1 from chunk image (p - 1) mod 2: lane = channel,
//     lane group = strip of output columns over all rows, exactly mbmap.hip's phase (same taps, same order, same bits), with eight lane
//     groups instead of sixteen; they store the results and the squeeze sums.
// One barrier per phase; nchunks + 1 phases.  The matrix instructions of one wave and the vector instructions of the other share a
// SIMD's issue but not its pipes.
// MEASURED (tools/mbmap_phases.py ... 0 BN_MBMAP_WS; v2.4, one context, mbmap.hip's bf16x3 form -> this kernel): the 6 x 32 x 112 launches
// 22.2 -> 20.6 us at batch 32 and 60.0 -> 47.4 us at batch 128 (marginal cost per 32 segments 12.6 -> 8.9 us; the exact-f32 form: 16),
// the 6 x 32 x 80 ones 14.2 -> 14.1 and 33.2 -> 28.6 us; four contexts +1 %.  A phase was then bound by the vector ALU alone -- the
// depthwise taps (600 FMA per lane and chunk), the two SiLU's transcendentals and 352 instructions of filter split -- so the split
// moved to the planner (pack_mbmap_w3p): the 112-channel launches 20.0 / 45.8 us (marginal 8.6), four contexts 65.3 -> 66.8 k
// segments/s (+2.3 % over mbmap.hip's bf16x3 form, 0.478 - 0.481 ms per step).  The 3 x 16 / 4 x 16 instances (one channel tile and two
// pixel tiles per expand wave, all K steps: no K slices to add up): v2.4's four launches 17.2 -> 15.5 us at batch 32, 40 -> 37 at
// batch 128, four contexts unchanged; v3.0's four 4 x 16 launches at batch 64 136 -> 103 us, 55.1 -> 56.2 k segments/s.  The banded 8 x 32
// instances replace v3.0's six expand GEMM + depthwise pairs (the exact-f32 banded form of round 4 had lost to them): 651 + 130 us of GEMM and
// depthwise launches become 488 + 206 us of GEMM and small-map launches, 56.0 -> 58.7 k segments/s; v3.0 runs no separate depthwise launch.
// TRANSPOSED bands (d.map_tr: the kernel's rows are the map's columns) with the input rows padded to whole 16-wide groups in LDS (d.cin_pad)
// serve Perch's ten 32 x 8 blocks (K = 96, and K = 136 in five steps: 252 registers): GEMM 2926 -> 2256 us, depthwise 793 -> 319 us,
// small-map launches + 718 us of its 5.5 ms chain; 23.2 -> 24.8 k segments/s at batch 128; 100 -> 90 launches.  Its six 16 x 4 blocks with
// K = 232 take the one-pixel-tile-per-wave form (EM = 2): GEMM - 270 us, depthwise - 150 us, small-map launches + 375 us; 24.25 -> 24.8 k on
// one box; 84 launches, two depthwise launches left (3 % of the chain).
//
// Arithmetic per value: identical to mbmap.hip's bf16x3 form (expand = bias + 32-deep steps ascending, six partial products per step
// in bf16x3.h's order; depthwise = bias2 + taps ascending): the same result bits.  The squeeze sum of a channel adds the partials of
// EIGHT column strips (in strip order) where mbmap.hip's 32-channel configuration adds sixteen -- the same terms, associated differently;
// against its 64-channel configuration (eight strips as well) everything is bit-identical (tests/test_gpu_ops.py).
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

// H x W = 6 x 32: an expand wave owns three pixel tiles and both channel tiles of a chunk.  3 x 16 and 4 x 16 (Cin % 64 == 0; round 5, the
// maps mbmap.hip walks with two K slices): an expand wave owns ONE channel tile and two pixel tiles (tiles 0-1 or 2-3; the 48-pixel map
// has no tile 3), so its input planes are 2 NSW fragments -- NSW up to 6 (Cin = 192) in 144 registers.
// NB = 2, HM = 8 (BirdNET v3.0's 8 x 32 stage): the map is cut into two bands of OHM / 2 output rows, a band is a block of its own
// (blockIdx.z) over the H = 6 REAL map rows its outputs reach (first row gy0 clamped into the map, mbmap.hip's scheme: which image row
// feeds which output row through which tap row is compile time per band); the squeeze sums are partial per band.
// EM = 2 (Perch's 16 x 4 maps, walked transposed as 4 x 16, K = 232 in eight steps): an expand wave owns ONE pixel tile and both channel
// tiles -- 96 registers of input planes at eight steps -- and the input image takes the four-chunk swizzle (SWZ16 = false: K % 64 != 0).
template <int K, int S, int NSW, int H, int W, int NB = 1, int HM = H, int EM = (W == 16 ? 1 : 0), bool SWZ16 = (W == 16)>
__global__ __launch_bounds__(512) void mbmap_ws_kernel(MbDesc d, float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ w1,
                                                       const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
                                                       float *__restrict__ gap, int nch, uint32_t inv_ch, const float *__restrict__ zpage) {
    constexpr int HW = H * W, NT = HW / 16, NC = 32, MW = EM == 0 ? 3 : (EM == 1 ? 2 : 1), NW = EM == 1 ? 1 : 2, EWV = 4, TD = 256, NGD = TD / NC;
    static_assert(EM == 0 ? W == 32 : (W == 16 && (EM == 1 || NT <= 4)), "tile ownership of the expand waves");
    static_assert((H == 6 && W == 32) || ((H == 3 || H == 4) && W == 16), "compiled map sizes");
    static_assert(MW * NSW <= 15, "the input planes of an expand wave: 12 registers per fragment");
    constexpr int PT = (K - 1) / 2, OHM = (HM + 2 * PT - K) / S + 1, OH = OHM / NB, OW = (W + 2 * PT - K) / S + 1;  // OH: output rows of ONE band
    static_assert(NB == 1 ? HM == H : (NB == 2 && H == 6 && W == 32 && OHM % 2 == 0), "bands: two, of six rows, of a map HM rows high");
    static_assert(OW % NGD == 0, "one strip of output columns per lane group");
    constexpr int PPG = OW / NGD, IWS = (PPG - 1) * S + K, WP = W + K - 1, EP = NC + 4;
    constexpr int WSZ = (NC / 16) * NSW * 3 * 256, ESZ = mm_kib(H * WP * EP), RSZ = NGD * NC;
    extern __shared__ __align__(1024) float ws_lds[];
    float *Ws = ws_lds;                 // [2][NC / 16][NSW][3 planes][64 lanes][8 bf16]
    float *Es = Ws + 2 * WSZ;           // [2][H][WP][EP], columns < PT and >= PT + W stay zero
    float *red = Es + 2 * ESZ;          // [2][NGD][NC]
    float *Xi = ws_lds + WSZ;           // prologue only: the input image [HW][Cin], over everything behind the first filter buffer
    const int Cin = d.cin_pad, CH = Cin >> 2;  // floats / chunks per row of the input image (the padded k: chunks past d.Cin come from the page of zeros)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool expander = wave < EWV;   // (wave-uniform)
    const int lc = lane & 15, lq = lane >> 4;
    const int64_t b = blockIdx.y;
    const int band = NB > 1 ? (int)blockIdx.z : 0;                          // (block-uniform)
    const int gy0 = NB > 1 ? min(max(band * OH * S - PT, 0), HM - H) : 0;   // first map row of the band's image
    const int cbase = blockIdx.x * nch * NC;
    const int nchunks = min(nch, (d.C - cbase + NC - 1) / NC);

    // ---- prologue, all waves: the sample's input image and the first filter chunk
    const int tr = d.map_tr;  // (run time) the kernel's rows are the map's columns: input gather, tap order and output address follow
    mm_copy_in<8, SWZ16, W, HM>(Xi, in + b * d.in_bs, zpage, HW, CH, d.Cin >> 2, d.Cin, inv_ch, gy0, tr, wave, lane);
    mm_copy_lin<8>(Ws, w1 + (int64_t)(cbase / 16) * (NSW * 768), WSZ / 256, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (from here the two kinds of waves run separate code -- the barriers pair up: 2 more in the prologue, then one per phase)
    auto zero_padding = [&]() {  // the K - 1 padding columns of every row of both chunk images are zero and stay zero
        for (int i = tid; i < 2 * H * (K - 1) * (EP / 4); i += 512) {
            const int q4 = i % (EP / 4), pc = (i / (EP / 4)) % (K - 1), y = (i / ((EP / 4) * (K - 1))) % H, im = i / ((EP / 4) * (K - 1) * H);
            const int xcol = pc < PT ? pc : W + pc;
            *reinterpret_cast<floatx4 *>(Es + im * ESZ + (y * WP + xcol) * EP + 4 * q4) = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (expander) {
        // =================================================================== expand waves
        // lane (c, q) of fragment (mt, s): k groups 2 s and 2 s + 1 of pixel 16 (mt0 + mt) + c, read as mbmap.hip's f32 form reads them
        const int nt0 = EM == 1 ? (wave & 1) : 0, mt0 = EM == 0 ? 3 * wave : (EM == 1 ? 2 * (wave >> 1) : wave);  // (wave-uniform)
        floatx4 raw[MW][NSW][2];
        {
            const int G16 = Cin >> 4;
#pragma unroll
            for (int mt = 0; mt < MW; mt++) {
                const int m = min(mt0 + mt, NT - 1) * 16 + lc;  // (a tile past the map repeats the last one: never multiplied, never stored)
                const int sw = mm_swz<SWZ16>(m);
#pragma unroll
                for (int st = 0; st < NSW; st++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int g = 2 * st + h;
                        const int off = SWZ16 ? m * Cin + 64 * (g >> 2) + 16 * ((g & 3) ^ (sw >> 2)) + 4 * (lq ^ (sw & 3)) : m * Cin + 16 * g + 4 * (lq ^ sw);
                        raw[mt][st][h] = g < G16 ? *reinterpret_cast<const floatx4 *>(Xi + off) : floatx4{0.f, 0.f, 0.f, 0.f};
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the image's space is free
        asm volatile("" ::: "memory");
        zero_padding();
        b3_u32x4 xh[MW][NSW], xm[MW][NSW], xl[MW][NSW];
        int epix[MW];
#pragma unroll
        for (int mt = 0; mt < MW; mt++) {
            const int m = min(mt0 + mt, NT - 1) * 16 + lc;
            const int y = m / W, x = m - y * W;
            epix[mt] = (y * WP + x + PT) * EP + nt0 * 16;
        }
        auto fetch_bias = [&](floatx4 (&bz)[NW], int c0) {
#pragma unroll
            for (int nt = 0; nt < NW; nt++) {
                const int n = c0 + (nt0 + nt) * 16 + 4 * lq;
                bz[nt] = (d.has_bias1 && n < d.C) ? *reinterpret_cast<const floatx4 *>(b1 + n) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        };
        floatx4 nbias[NW];
        fetch_bias(nbias, cbase);
#pragma unroll
        for (int mt = 0; mt < MW; mt++)
#pragma unroll
            for (int st = 0; st < NSW; st++) split3(raw[mt][st][0], raw[mt][st][1], xh[mt][st], xm[mt][st], xl[mt][st]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        for (int p = 0; p < nchunks; p++) {
            const int c0 = cbase + p * NC;
            const float *Wc = Ws + (p & 1) * WSZ;
            float *Ec = Es + (p & 1) * ESZ;
            floatx4 acc[MW][NW];
#pragma unroll
            for (int mt = 0; mt < MW; mt++)
#pragma unroll
                for (int nt = 0; nt < NW; nt++) acc[mt][nt] = nbias[nt];
            if (p + 1 < nchunks) {  // (that filter buffer was last read in phase p - 1)
                mm_copy_lin<EWV>(Ws + ((p + 1) & 1) * WSZ, w1 + (int64_t)((c0 + NC) / 16) * (NSW * 768), WSZ / 256, wave, lane);
                fetch_bias(nbias, c0 + NC);
            }
            b3_u32x4 wr[2][3];
            auto rdw = [&](b3_u32x4 (&r)[3], int f) {  // f = st * NW + nt, compile time at every call site: the three planes of the fragment
                const float *wb = Wc + (((nt0 + f % NW) * NSW + f / NW) * 3 * 64 + lane) * 4;
#pragma unroll
                for (int pp = 0; pp < 3; pp++) r[pp] = *reinterpret_cast<const b3_u32x4 *>(wb + pp * 256);
            };
            rdw(wr[0], 0);
#pragma unroll
            for (int f = 0; f < NSW * NW; f++) {
                if (f + 1 < NSW * NW) rdw(wr[(f + 1) & 1], f + 1);
#pragma unroll
                for (int mt = 0; mt < MW; mt++)
                    if (NT % MW == 0 || mt0 + mt < NT)  // (wave-uniform; compile-time true for the maps whose tiles divide evenly)
                        acc[mt][f % NW] = mm6(wr[f & 1][0], wr[f & 1][1], wr[f & 1][2], xh[mt][f / NW], xm[mt][f / NW], xl[mt][f / NW], acc[mt][f % NW]);
            }
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
                    if (NT % MW == 0 || mt0 + mt < NT)
                        *reinterpret_cast<floatx4 *>(Ec + epix[mt] + nt * 16 + 4 * lq) =
                        floatx4{v[(mt * NW + nt) * 4], v[(mt * NW + nt) * 4 + 1], v[(mt * NW + nt) * 4 + 2], v[(mt * NW + nt) * 4 + 3]};
            // chunk image p is complete, filter chunk p + 1 has landed (only loads are outstanding on these waves)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        return;  // (a finished wave leaves the work-group's barrier count: the depthwise waves' last barrier is among themselves)
    }

    // ======================================================================= depthwise waves
    __builtin_amdgcn_s_barrier();  // (the expand waves have read their input fragments: the image's space is free)
    asm volatile("" ::: "memory");
    zero_padding();
    const int td = tid - 64 * EWV;
    const int c = td % NC, grp = td / NC;
    const int ox0 = grp * PPG;
    struct DwConst {
        float wd[K * K], bz;
    };
    auto fetch_dw = [&](DwConst &cc_, int c0) {
        const unsigned cl = (unsigned)min(c0 + c, d.C - 1);
#pragma unroll
        for (int q = 0; q < K * K; q++) cc_.wd[q] = (w2 + (size_t)(tr ? (q % K) * K + q / K : q) * (size_t)d.C)[cl];  // kernel tap (ky, kx) = map tap (kx, ky) when transposed
        cc_.bz = d.has_bias2 ? b2[cl] : 0.0f;
    };
    DwConst nxt;
    fetch_dw(nxt, cbase);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (the prologue's third barrier)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // phase 0: the first chunk image is being written
    asm volatile("" ::: "memory");
    auto flush_gap = [&](int q) {  // the squeeze sums of chunk q, complete per (sample, channel): its partials were written a barrier ago
        const int cgq = cbase + q * NC + c;
        if (d.has_gap && grp == 0 && cgq < d.C) {
            const float *r = red + (q & 1) * RSZ;
            float t = r[c];
#pragma unroll
            for (int y = 1; y < NGD; y++) t += r[y * NC + c];
            gap[b * d.gap_bs + (int64_t)band * d.C + cgq] = t;
        }
    };
    for (int q = 0; q < nchunks; q++) {
        const int c0 = cbase + q * NC;
        const float *Ec = Es + (q & 1) * ESZ;
        const DwConst cur = nxt;
        if (q + 1 < nchunks) fetch_dw(nxt, c0 + NC);
        if (q > 0) flush_gap(q - 1);
        const int cg = c0 + c;
        const bool cact = cg < d.C;
        float sum = 0.0f;
        auto dw_phase = [&](auto roff_c) {
            constexpr int ROFF = decltype(roff_c)::value;  // image row of output row oy through tap row ky: oy * S + ky - ROFF
            // (Two output columns per instruction -- v_pk_fma_f32 on pairs read from LDS with ds_read2, bit-identical chains -- measured no
            // faster: 20.0 -> 20.3 us at batch 32, 45.8 -> 47.9 at batch 128 for the 112-channel launches.  The phase is not bound by the FMA
            // issue rate; the scalar form with its 8 reads per image row stays.)
            float ov[OH][PPG];
#pragma unroll
            for (int oy = 0; oy < OH; oy++)
#pragma unroll
                for (int x = 0; x < PPG; x++) ov[oy][x] = cur.bz;
            const float *rp0 = Ec + (ox0 * S) * EP + c;
#pragma unroll
            for (int iy = 0; iy < H; iy++) {
                bool used = false;  // (compile time) an image row no output reaches is not read
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
                    if (t >= 0 && t % S == 0 && t / S < OH) {
#pragma unroll
                        for (int x = 0; x < PPG; x++)
#pragma unroll
                            for (int kx = 0; kx < K; kx++) ov[t / S][x] = fmaf(val[x * S + kx], cur.wd[ky * K + kx], ov[t / S][x]);
                    }
                }
            }
            float *ob = out + b * d.out_bs;
            // output element (row, column) of the kernel's geometry: + row * o_rs + column * o_cs floats (transposed: the map's (x, y))
            const unsigned o_cs = tr ? (unsigned)(OHM * d.C) : (unsigned)d.C, o_rs = tr ? (unsigned)d.C : (unsigned)(OW * d.C);
            const unsigned olane = (unsigned)cg + (unsigned)ox0 * o_cs;
#pragma unroll
            for (int oy = 0; oy < OH; oy++) {
                float r[PPG];
#pragma unroll
                for (int x = 0; x < PPG; x++) r[x] = ov[oy][x];
                mm_act<PPG>(d.act2, d.p0_2, d.p1_2, r);
                if (cact) {
#pragma unroll
                    for (int x = 0; x < PPG; x++) {
                        (ob + (size_t)((unsigned)(band * OH + oy) * o_rs + (unsigned)x * o_cs))[olane] = r[x];
                        sum += r[x];
                    }
                }
            }
        };
        if constexpr (NB == 1) {
            dw_phase(std::integral_constant<int, PT>{});
        } else {
            constexpr int G1 = (OH * S - PT) < 0 ? 0 : ((OH * S - PT) > HM - H ? HM - H : (OH * S - PT));  // gy0 of band 1
            if (band == 0) dw_phase(std::integral_constant<int, PT>{});
            else dw_phase(std::integral_constant<int, PT + G1 - OH * S>{});
        }
        if (d.has_gap) red[(q & 1) * RSZ + grp * NC + c] = sum;
        // ends phase q + 1: this chunk image is read, the partials are written (LDS only: the result stores stay in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    flush_gap(nchunks - 1);
}

}  // namespace

size_t mbmap_ws_lds_bytes(const MbDesc &d, int nsw) {
    const int wsz = 2 * nsw * 3 * 256, esz = mm_kib(d.H * (d.W + d.k - 1) * 36), ring = 2 * wsz + 2 * esz + 2 * 8 * 32, pro = wsz + mm_kib(d.H * d.W * d.Cin);
    return (size_t)std::max(ring, pro) * sizeof(float);
}

void register_mbmap_ws_kernels() {
#define WS_REG(K, S, NSW, H, W) register_dynamic_lds_kernel(reinterpret_cast<const void *>(mbmap_ws_kernel<K, S, NSW, H, W>));
#define WS_REGB(K, S, NSW) register_dynamic_lds_kernel(reinterpret_cast<const void *>(mbmap_ws_kernel<K, S, NSW, 6, 32, 2, 8>));
#define WS_REG_KS(NSW, H, W) WS_REG(3, 1, NSW, H, W) WS_REG(5, 1, NSW, H, W) WS_REG(3, 2, NSW, H, W) WS_REG(5, 2, NSW, H, W)
    WS_REG_KS(2, 6, 32) WS_REG_KS(3, 6, 32) WS_REG_KS(4, 6, 32)
    WS_REG_KS(4, 3, 16) WS_REG_KS(6, 3, 16)
    WS_REG(3, 1, 4, 4, 16) WS_REG(5, 1, 4, 4, 16) WS_REG(3, 1, 6, 4, 16) WS_REG(5, 1, 6, 4, 16)
    WS_REGB(3, 1, 3) WS_REGB(5, 1, 3) WS_REGB(3, 2, 3) WS_REGB(5, 2, 3) WS_REGB(3, 1, 4) WS_REGB(5, 1, 4) WS_REGB(3, 2, 4) WS_REGB(5, 2, 4)  // 8 x 32 in two bands
    WS_REGB(3, 1, 5) WS_REGB(5, 1, 5) WS_REGB(3, 2, 5) WS_REGB(5, 2, 5)                                                                      // (Perch: K = 136)
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(mbmap_ws_kernel<3, 1, 8, 4, 16, 1, 4, 2, false>));  // 4 x 16, K = 232 / 256 (Perch, transposed)
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(mbmap_ws_kernel<5, 1, 8, 4, 16, 1, 4, 2, false>));
#undef WS_REGB
#undef WS_REG_KS
#undef WS_REG
}

// d.map_ws = 32-deep steps (plan_rules.h mbmap_ws_steps); w1 = pack_mbmap_w3p's image
bool launch_mbmap_ws(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                     float *gap, int64_t batch, int nch) {
    const int nsw = d.map_ws;
    const float *zpage = device_zero_page();
    // (the kernel's geometry: a transposed map's rows are its columns)
    const int kh_ = d.map_tr ? d.W : d.H, kw_ = d.map_tr ? d.H : d.W;
    const bool banded = kh_ == 8 && kw_ == 32 && d.map_bands == 2;
    const bool deep4 = kh_ == 4 && kw_ == 16 && nsw == 8 && d.s == 1 && d.map_bands == 1;  // one pixel tile per expand wave
    const bool big = !d.map_tr && d.H == 6 && d.W == 32, small3 = !d.map_tr && d.H == 3 && d.W == 16, small4 = !d.map_tr && d.H == 4 && d.W == 16 && !deep4;
    if (!zpage || d.cin_pad % 16 || d.Cin % 4 || d.cin_pad < d.Cin || (d.Cin + 31) / 32 != nsw || (d.map_tr && !banded && !deep4)) return false;
    if (!(deep4 || big ? (deep4 || (nsw >= 2 && nsw <= 4)) : banded ? (nsw >= 3 && nsw <= 5) : ((small3 || (small4 && d.s == 1)) && d.Cin % 64 == 0 && (nsw == 4 || nsw == 6)))) return false;
    const uint32_t inv_ch = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(d.cin_pad / 4)) + 1u;
    MbDesc ld = d;  // LDS sizes follow the block's image: a band's six rows
    ld.Cin = d.cin_pad;
    if (banded) { ld.H = 6; ld.W = 32; }
    if (deep4) { ld.H = 4; ld.W = 16; }
    const size_t lds = mbmap_ws_lds_bytes(ld, nsw);
    dim3 grid((unsigned)((d.C + nch * 32 - 1) / (nch * 32)), (unsigned)batch, banded ? 2u : 1u);
#define WS_GO(K, S, NSW, H, W) hipLaunchKernelGGL((mbmap_ws_kernel<K, S, NSW, H, W>), grid, dim3(512), lds, s, d, out, in, w1, b1, w2, b2, gap, nch, inv_ch, zpage)
#define WS_GO_KS(NSW, H, W)                                  \
    do {                                                     \
        if (d.k == 3 && d.s == 1) WS_GO(3, 1, NSW, H, W);    \
        else if (d.k == 5 && d.s == 1) WS_GO(5, 1, NSW, H, W); \
        else if (d.k == 3) WS_GO(3, 2, NSW, H, W);           \
        else WS_GO(5, 2, NSW, H, W);                         \
    } while (0)
    if (deep4) {
        if (d.k == 3) hipLaunchKernelGGL((mbmap_ws_kernel<3, 1, 8, 4, 16, 1, 4, 2, false>), grid, dim3(512), lds, s, d, out, in, w1, b1, w2, b2, gap, nch, inv_ch, zpage);
        else hipLaunchKernelGGL((mbmap_ws_kernel<5, 1, 8, 4, 16, 1, 4, 2, false>), grid, dim3(512), lds, s, d, out, in, w1, b1, w2, b2, gap, nch, inv_ch, zpage);
    } else if (banded) {
#define WS_GOB(K, S, NSW) hipLaunchKernelGGL((mbmap_ws_kernel<K, S, NSW, 6, 32, 2, 8>), grid, dim3(512), lds, s, d, out, in, w1, b1, w2, b2, gap, nch, inv_ch, zpage)
#define WS_GOB_KS(NSW)                               \
    do {                                             \
        if (d.k == 3 && d.s == 1) WS_GOB(3, 1, NSW); \
        else if (d.k == 5 && d.s == 1) WS_GOB(5, 1, NSW); \
        else if (d.k == 3) WS_GOB(3, 2, NSW);        \
        else WS_GOB(5, 2, NSW);                      \
    } while (0)
        if (nsw == 3) WS_GOB_KS(3);
        else if (nsw == 4) WS_GOB_KS(4);
        else WS_GOB_KS(5);
#undef WS_GOB_KS
#undef WS_GOB
    } else if (big) {
        if (nsw == 2) WS_GO_KS(2, 6, 32);
        else if (nsw == 3) WS_GO_KS(3, 6, 32);
        else WS_GO_KS(4, 6, 32);
    } else if (small3) {
        if (nsw == 4) WS_GO_KS(4, 3, 16);
        else WS_GO_KS(6, 3, 16);
    } else {
        if (nsw == 4 && d.k == 3) WS_GO(3, 1, 4, 4, 16);
        else if (nsw == 4) WS_GO(5, 1, 4, 4, 16);
        else if (d.k == 3) WS_GO(3, 1, 6, 4, 16);
        else WS_GO(5, 1, 6, 4, 16);
    }
#undef WS_GO_KS
#undef WS_GO
    return true;
}

}  // namespace bn
