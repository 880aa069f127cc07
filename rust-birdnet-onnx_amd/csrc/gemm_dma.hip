// 1x1-conv / MatMul GEMM with direct-to-LDS operand staging (gfx950 only).
//
// Why a second GEMM kernel.  On this part the exact-f32 matrix instructions and the f32 vector ALU are ONE resource:
// tools/mfma_valu_probe.cpp times a wave of v_mfma_f32_* and a wave of v_fma_f32 on the same SIMD at the SUM of their
// solo times, in one instruction stream and in two.  Every vector instruction of a GEMM kernel is therefore paid in
// full out of the matrix pipe's time, and gemm_mfma_kernel / gemm_splitk_kernel (kernels.hip) spend 9 of them per
// matrix instruction on global -> VGPR -> LDS staging, address arithmetic and a 16-store-per-tile epilogue (PMC,
// profiles/r02_v4_pmc_mfma.json).  This kernel has no staging instruction at all:
//
//   * both operands arrive by global_load_lds_dwordx4 (LDS-DMA): a wave instruction moves 8 rows x 128 B straight
//     into a [rows][32 floats] stage image; the per-lane SOURCE address carries the XOR swizzle
//     (16-byte chunk c of row r lands in slot c ^ ((r >> 1) & 7)) that makes the ds_read_b128 fragment reads
//     conflict free, the per-stage advance is a scalar add on the uniform base;
//   * a 3-deep ring of stages, one raw s_barrier and one counted s_waitcnt vmcnt per 32-deep K step, the next two
//     stages in flight across the barrier;
//   * v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand: lane (c, q) then holds four CONSECUTIVE output channels
//     4q..4q+3 of row c, so bias, residual and result move as one dwordx4 per 16x16 tile instead of four dwords;
//   * a block owns TR rows of ONE sample (TR divides the rows of a sample) x up to 128 output channels: the
//     squeeze-excite gate of the sample sits in LDS and multiplies the activation fragment (4 v_mul per 16 k);
//     tiles are 64 x N (N <= 128) or 48 x 64: 16-24 flop per byte fetched from L2 instead of 8 for the 32 x 32
//     tiles of the split-K kernel, whose time followed its staged bytes.
//
// Arithmetic: every output element is one accumulation chain over k in a fixed order (k-slot j of a 16-wide group
// takes k = 16g + 4q + j), bias added after the sum, activation, then the residual -- independent of the batch size,
// so a segment's bits do not depend on what shares its batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "device_common.h"
#include "kernels.h"
#include "plan_rules.h"

namespace bn {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void gd_act(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
    else if (act == ACT_SIGMOID) map_array<N>(v, [](float x) { return net_sigmoid(x); });
    else if (act == ACT_HSIGMOID) map_array<N>(v, [=](float x) { return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f); });
}

#ifdef BN_GD_STAMPS  // tools/gemm_stamps.cpp: shader-clock stamps of block (0, 0)'s waves around every K step (diagnostic build only)
__device__ unsigned long long bn_gd_stamps[8][96][3];  // [wave][iteration][before the wait | behind the barrier + refill | behind the matrix instructions]
#define GD_STAMP(IT, WHICH)                                                                                                \
    do {                                                                                                                   \
        if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && lane == 0 && (IT) < 96) bn_gd_stamps[wave][(IT)][(WHICH)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GD_STAMP(IT, WHICH)
#endif

#define GD_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GD_GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// MTW x NTW 16x16 tiles per wave, WM x WN waves per K slice (2 or 4), KS K slices (slice ks takes the 32-deep K
// steps ks, ks + KS, ... through a ring of D stages of its own; the slices' partial tiles are summed through LDS in
// slice order at the end): block = 64 WM WN KS threads, tile = (16 MTW WM) rows x (16 NTW WN) channels.  The tile shape
// does not enter any output element's arithmetic (only KS does), so the launcher may pick it by the batch size.
// GATE: 0 = plain, 1 = the squeeze-excite gate of the sample is read from memory, 2 = ... is computed in the prologue from
// the squeeze partial sums (every block of a sample redoes the two small excite products: one launch less per block,
// 7 - 8 us of the step's serial chain each; the order of every sum in it is fixed and does not depend on the block shape)
template <int MTW, int NTW, int WM, int WN, int KS, int D, int GATE>
__global__ __launch_bounds__(64 * WM * WN * KS) void gemm_dma_kernel(GemmDesc d, float *__restrict__ C, const float *__restrict__ A, const float *__restrict__ W,
                                                            const float *__restrict__ bias, const float *__restrict__ res,
                                                            const float *__restrict__ scale, int tiles_per_sample, int gate_floats, SeInline sei) {
    constexpr bool GATED = GATE != 0;
    constexpr int WPS = WM * WN;  // waves per K slice
    static_assert(WPS == 2 || WPS == 4, "two or four waves per K slice");
    constexpr int TR = 16 * MTW * WM, BN = 16 * NTW * WN;
    constexpr int XP = TR / 8, WP = BN / 8, PIECES = XP + WP;  // 1-KiB pieces (8 rows x 128 B) of one stage
    constexpr int NP = (PIECES + WPS - 1) / WPS;               // pieces every wave issues per stage
    constexpr int STAGE_FLOATS = (TR + BN) * 32;
    extern __shared__ __align__(1024) float gd_lds[];
    float *gate = gd_lds + KS * D * STAGE_FLOATS;  // [gate_floats] (GATED)

    // everything derived from the wave index is wave-uniform: said so (readfirstlane), it lives in scalar registers and the
    // per-slice branches below are scalar branches (left to itself the compiler carries them per lane: exec-mask
    // branches, a copy of every accumulator per K step)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / WPS, w4 = wave % WPS;
    const int lc = lane & 15, lq = lane >> 4;
    const int wm = w4 % WM, wn = w4 / WM;
    const int b = blockIdx.x / tiles_per_sample, rt = blockIdx.x - b * tiles_per_sample;
    const int n0 = blockIdx.y * BN;
    const int K = d.K;
    const float *Xb = A + (int64_t)b * d.a_bs + (int64_t)rt * TR * d.lda;  // this tile's rows (lda == K: checked by the launcher)
    float *ring = gd_lds + ks * D * STAGE_FLOATS;

    // ---- per-lane source offsets of this wave's pieces (elements, relative to Xb / W), swizzle on the source side
    const int prow = lane >> 3, pslot = lane & 7;
    uint32_t off[NP];
    bool is_w[NP];
    uint32_t dst[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        int p = w4 + WPS * j;
        if (p >= PIECES) p = w4 % PIECES;  // a wave without a piece of its own repeats one (same bytes, same place)
        const bool w_img = p >= XP;
        const int row = 8 * (w_img ? p - XP : p) + prow;  // row inside its image
        const int chunk = pslot ^ ((row >> 1) & 7);       // source chunk that belongs in this slot
        int grow = row;
        if (w_img) {
            grow = n0 + row;
            grow = grow < d.N ? grow : d.N - 1;
        }
        is_w[j] = w_img;
        off[j] = (uint32_t)grow * (uint32_t)K + 4u * (uint32_t)chunk;
        dst[j] = (uint32_t)(p * 256);  // floats from the stage base
    }
    // K steps: nfs full 32-deep ones, then (K % 32 == 16) one HALF step whose stage covers columns K-32 .. K-1 -- memory
    // that exists, same lane offsets -- and whose second 16-wide group alone is multiplied.  Step s belongs to slice
    // s % KS; a slice's i-th step is ks + i KS.
    const int nfs = K >> 5;
    const bool half_tail = (K & 31) != 0;
    const int nst = nfs + (half_tail ? 1 : 0);
    const int nmine = (nst - ks + KS - 1) / KS;
    const int niter = (nst + KS - 1) / KS;  // block-uniform loop count (the barriers)
    const int nmain = nfs / KS;             // leading iterations in which EVERY slice has a full step
    auto issue = [&](int i) {               // this slice's i-th step into ring slot i % D
        const int s = ks + i * KS;
        float *sb = ring + (i % D) * STAGE_FLOATS;
        const int k0 = (half_tail && s == nst - 1) ? K - 32 : 32 * s;
        const float *xk = Xb + k0, *wk = W + k0;  // uniform bases; the lane offsets are 32-bit
#pragma unroll
        for (int j = 0; j < NP; j++)
            __builtin_amdgcn_global_load_lds(GD_GLB_PTR((is_w[j] ? wk : xk) + off[j]), GD_LDS_PTR(sb + dst[j]), 16, 0, 0);
    };

    // ---- prologue: the sample's gate (older than every stage piece, so the first counted wait covers it), D - 1 steps
    if constexpr (GATE == 1) {
        const float *gsrc = scale + (int64_t)b * d.s_bs;
        const int n16 = K >> 2;
        for (int c0 = wave * 64; c0 < gate_floats / 4; c0 += 64 * WPS * KS) {
            int c = c0 + lane;
            c = c < n16 ? c : n16 - 1;
            __builtin_amdgcn_global_load_lds(GD_GLB_PTR(gsrc + 4 * c), GD_LDS_PTR(gate + 4 * c0), 16, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < D - 1; i++)
        if (i < nmine) issue(i);

    if constexpr (GATE == 2) {
        // the excite products of this block's sample, while the first K steps are on their way.  Fixed orders: squeeze =
        // splits ascending; hidden unit j = one wave, lane l sums c = l, l + 64, ... then a 6-step butterfly; gate c = one
        // lane, j ascending.  None of it depends on how many waves the block has.
        constexpr int T = 64 * WPS * KS, NWV = WPS * KS;
        const SeFcDesc &se = sei.se;
        float *sbuf = gate + gate_floats, *hbuf = sbuf + ((se.C + 3) & ~3);
        const float *pp = sei.partial + (int64_t)b * se.in_bs;
        for (int c = tid; c < se.C; c += T) {
            float a = 0.0f;
            for (int sp = 0; sp < se.splits; sp++) a += pp[(int64_t)sp * se.C + c];
            sbuf[c] = a * se.inv_hw;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int j = wave; j < se.Cr; j += NWV) {
            const float *wr = sei.w1 + (int64_t)j * se.C;
            float a = 0.0f;
            for (int c = lane; c < se.C; c += 64) a = fmaf(wr[c], sbuf[c], a);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) {
                float hv[1] = {a + (sei.b1 ? sei.b1[j] : 0.0f)};
                gd_act<1>(se.act1, se.p0_1, se.p1_1, hv);
                hbuf[j] = hv[0];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int c = tid; c < se.C; c += T) {
            float a = 0.0f;
            for (int j = 0; j < se.Cr; j++) a = fmaf(sei.w2t[(int64_t)j * se.C + c], hbuf[j], a);
            float gv[1] = {a + (sei.b2 ? sei.b2[c] : 0.0f)};
            gd_act<1>(se.act2, se.p0_2, se.p1_2, gv);
            gate[c] = gv[0];
        }
        // (visible to every wave behind the barrier of the first turn() below, which also waits lgkmcnt(0))
    }

    floatx4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (floats inside a stage): activation rows of this wave's m-tiles, weight rows of its n-tiles
    int xoff[MTW][2], woff[NTW][2];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        const int r = (wm * MTW + mt) * 16 + lc;
#pragma unroll
        for (int g = 0; g < 2; g++) xoff[mt][g] = r * 32 + 4 * ((4 * g + lq) ^ ((r >> 1) & 7));
    }
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int r = (wn * NTW + nt) * 16 + lc;
#pragma unroll
        for (int g = 0; g < 2; g++) woff[nt][g] = TR * 32 + r * 32 + 4 * ((4 * g + lq) ^ ((r >> 1) & 7));
    }

    // one 16-wide k group of the stage at sb: fragments, gate, matrix instructions
    auto group = [&](const float *sb, int g, int kcol) {
        floatx4 xf[MTW], wf[NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) xf[mt] = *reinterpret_cast<const floatx4 *>(sb + xoff[mt][g]);
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) wf[nt] = *reinterpret_cast<const floatx4 *>(sb + woff[nt][g]);
        if constexpr (GATED) {  // the gate runs along k: either operand carries it -- the one with fewer fragments
            const floatx4 gf = *reinterpret_cast<const floatx4 *>(gate + kcol + 4 * lq);
            if constexpr (NTW < MTW) {
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) wf[nt] *= gf;
            } else {
#pragma unroll
                for (int mt = 0; mt < MTW; mt++) xf[mt] *= gf;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                for (int mt = 0; mt < MTW; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][j], xf[mt][j], acc[mt][nt], 0, 0, 0);
    };
    // the wait + barrier that make this slice's step i readable, then the refill of the slot read in iteration i - 1.
    // Step i has landed once all but the pieces of this slice's YOUNGER steps in flight are done (in-order completion):
    // min(D - 2, steps left) x NP of them.  lgkmcnt(0): this wave's fragment reads of step i - 1 are done before any
    // wave refills that slot.
    auto turn = [&](int i) {
        if (nmine - 1 - i >= D - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 2) * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + D - 1 < nmine) issue(i + D - 1);
    };
    static_assert(D == 3, "the counted waits above are written for a ring of three");

    // main loop: every slice multiplies a full step, nothing conditional around the matrix instructions (a branch there
    // makes the compiler copy all accumulators through VGPRs every iteration)
    // Round 4, in-kernel stamps (tools/gemm_stamps.cpp, -DBN_GD_STAMPS; gpurun_out/r4k/stamps*.txt): a K-step pair of this loop takes
    // ~4 800 cycles for 3 584 cycles of matrix instructions per SIMD.  The difference is the ISSUE of the LDS-DMA pieces: ~100 - 170 cycles
    // per global_load_lds_dwordx4 (1 KiB), 12 per SIMD and K-step pair, during which that SIMD multiplies nothing.  Behind the barrier
    // (here) they show as wait time -- the first slice's waves, older, win the matrix pipe, finish 2 100 cycles early and wait; the
    // second slice's waves spend ~1 000 cycles issuing.  Spread between the matrix instructions of the step (one piece behind each k
    // slot's block, sched_barrier-pinned) the wait share falls from 0.44 / 0.24 to 0.17 / 0.08 of the loop AND THE LOOP TAKES AS LONG
    // (48 880 against 48 172 cycles at K = 672; 113 456 against 114 700 at K = 1392): the cost moves, it does not overlap.  What a
    // step pays is bytes staged per multiply-add -- (TR + BN) * 128 B per TR * BN * 32 -- i.e. the tile shape, which the LDS bounds
    // (two K slices x three stages: TR + BN <= 208 rows).  The simple order stays.
    for (int i = 0; i < nmain; i++) {
        GD_STAMP(i, 0);
        turn(i);
        GD_STAMP(i, 1);
        const float *sb = ring + (i % D) * STAGE_FLOATS;
        const int kcol = 32 * (ks + i * KS);
        group(sb, 0, kcol);
        group(sb, 1, kcol + 16);
        GD_STAMP(i, 2);
    }
    // ragged end: at most two more iterations in which some slices have a full step, one the half step, some none
    for (int i = nmain; i < niter; i++) {
        turn(i);
        if (i < nmine) {
            const int s = ks + i * KS;
            const float *sb = ring + (i % D) * STAGE_FLOATS;
            if (half_tail && s == nst - 1) {
                group(sb, 1, K - 16);
            } else {
                group(sb, 0, 32 * s);
                group(sb, 1, 32 * s + 16);
            }
        }
    }

    // ---- bias and residual of the finishing slice are requested BEFORE the slices meet (clamped addresses, no predicates): their
    // round trip runs under the exchange below instead of behind it
    floatx4 bpre[NTW], rpre[MTW][NTW];
    if (ks == 0) {
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const int n = min(n0 + (wn * NTW + nt) * 16 + 4 * lq, d.N - 4);
            bpre[nt] = d.has_bias ? *reinterpret_cast<const floatx4 *>(bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MTW; mt++) {
                const int64_t m = (int64_t)rt * TR + (wm * MTW + mt) * 16 + lc;
                rpre[mt][nt] = d.has_res ? *reinterpret_cast<const floatx4 *>(res + (int64_t)b * d.r_bs + m * d.ldr + n) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    // ---- the slices' partial tiles, summed in slice order (fixed: slice 0 + slice 1 (+ slice 2 + slice 3))
    if constexpr (KS > 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every fragment read of the rings is done: they become the exchange buffer
        asm volatile("" ::: "memory");
        floatx4 *xch = reinterpret_cast<floatx4 *>(gd_lds);  // [KS - 1][WPS waves][MTW * NTW][64 lanes]
        if (ks > 0) {
#pragma unroll
            for (int mt = 0; mt < MTW; mt++)
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) xch[(((ks - 1) * WPS + w4) * (MTW * NTW) + mt * NTW + nt) * 64 + lane] = acc[mt][nt];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ks > 0) return;
#pragma unroll
        for (int q = 1; q < KS; q++)
#pragma unroll
            for (int mt = 0; mt < MTW; mt++)
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) acc[mt][nt] += xch[(((q - 1) * WPS + w4) * (MTW * NTW) + mt * NTW + nt) * 64 + lane];
    }

    // ---- epilogue: lane (lc, lq) holds channels nb + 4 lq .. + 3 of row lc of each tile
    float v[MTW * NTW * 4];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int n = n0 + (wn * NTW + nt) * 16 + 4 * lq;
        (void)n;
        const floatx4 bv = bpre[nt];  // (columns n >= N are never stored)
#pragma unroll
        for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[(mt * NTW + nt) * 4 + i] = acc[mt][nt][i] + bv[i];
    }
    gd_act<MTW * NTW * 4>(d.act, d.p0, d.p1, v);
    if constexpr (WM == 1) {
        if (d.gap) {  // the sample's mean over its rows (all TR of them sit in this wave): m-tiles ascending, then the 16 rows of a tile by a
                      // fixed butterfly over the lanes lc -- one dwordx4 per n-tile and lane group instead of TR rows
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) {
                floatx4 sm = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < MTW; mt++)
#pragma unroll
                    for (int i = 0; i < 4; i++) sm[i] += v[(mt * NTW + nt) * 4 + i];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                    for (int i = 0; i < 4; i++) sm[i] += __shfl_xor(sm[i], o);
                const int n = n0 + (wn * NTW + nt) * 16 + 4 * lq;
                if (lc == 0 && n < d.N) {
                    const float inv_rows = (float)TR;
                    *reinterpret_cast<floatx4 *>(C + (int64_t)b * d.c_bs + n) = floatx4{sm[0] / inv_rows, sm[1] / inv_rows, sm[2] / inv_rows, sm[3] / inv_rows};
                }
            }
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        const int64_t m = (int64_t)rt * TR + (wm * MTW + mt) * 16 + lc;
        float *crow = C + (int64_t)b * d.c_bs + m * d.ldc;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const int n = n0 + (wn * NTW + nt) * 16 + 4 * lq;
            if (n < d.N) {
                floatx4 o = floatx4{v[(mt * NTW + nt) * 4], v[(mt * NTW + nt) * 4 + 1], v[(mt * NTW + nt) * 4 + 2], v[(mt * NTW + nt) * 4 + 3]};
                if (d.has_res) o += rpre[mt][nt];
                *reinterpret_cast<floatx4 *>(crow + n) = o;
            }
        }
    }
}

// ---- streaming form (round 4): ONE K slice, no gate, a block walks `tpb` CONSECUTIVE row tiles of the batch against the same
// channel tile and its ring never drains between them.
// Why: the expand convs of the late stages (K 64 .. 232, N 480 .. 1392: short K, wide N) are two to eight K steps per tile -- with one
// tile per block, prologue (first stages from L2) and epilogue (bias, activation, store) are as long as the product between them, which
// is why round 3 left these shapes on the tiled kernel.  Here step i of the block's step sequence (tile i / nst, K step i % nst) is
// issued D - 1 steps ahead whatever tile it belongs to, so the next tile's operands arrive while this tile's epilogue runs; at
// 74 KB of LDS two blocks share a CU and cover each other's barriers.  Same fragment layout, same k order (k-slot j of a 16-wide
// group: k = 16 g + 4 q + j, groups ascending), same epilogue as gemm_dma_kernel<.., KS = 1, GATE = 0>: the bits of an output do not
// depend on tpb or on the tile shape.
template <int MTW, int NTW, int WM, int WN, int D>
__global__ __launch_bounds__(64 * WM * WN) void gemm_dma_stream_kernel(GemmDesc d, float *__restrict__ C, const float *__restrict__ A, const float *__restrict__ W,
                                                                        const float *__restrict__ bias, const float *__restrict__ res, int tiles_per_sample,
                                                                        int total_tiles, int tpb) {
    constexpr int WPS = WM * WN;
    static_assert(WPS == 2 || WPS == 4, "two or four waves");
    constexpr int TR = 16 * MTW * WM, BN = 16 * NTW * WN;
    constexpr int XP = TR / 8, WP = BN / 8, PIECES = XP + WP;
    constexpr int NP = (PIECES + WPS - 1) / WPS;
    constexpr int STAGE_FLOATS = (TR + BN) * 32;
    extern __shared__ __align__(1024) float gd_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lq = lane >> 4;
    const int wm = w4 % WM, wn = w4 / WM;
    const int n0 = blockIdx.y * BN;
    const int K = d.K;
    const int t_first = blockIdx.x * tpb;
    const int ntiles = min(tpb, total_tiles - t_first);
    if (ntiles <= 0) return;

    const int prow = lane >> 3, pslot = lane & 7;
    uint32_t off[NP];
    bool is_w[NP];
    uint32_t dst[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        int p = w4 + WPS * j;
        if (p >= PIECES) p = w4 % PIECES;
        const bool w_img = p >= XP;
        const int row = 8 * (w_img ? p - XP : p) + prow;
        const int chunk = pslot ^ ((row >> 1) & 7);
        int grow = row;
        if (w_img) {
            grow = n0 + row;
            grow = grow < d.N ? grow : d.N - 1;
        }
        is_w[j] = w_img;
        off[j] = (uint32_t)grow * (uint32_t)K + 4u * (uint32_t)chunk;
        dst[j] = (uint32_t)(p * 256);
    }
    const int nfs = K >> 5;
    const bool half_tail = (K & 31) != 0;
    const int nst = nfs + (half_tail ? 1 : 0);
    const int total_steps = ntiles * nst;
    // the issuing side's position in the step sequence (tile, K step, ring slot), advanced by one per issue()
    int is_t = 0, is_s = 0, is_slot = 0;
    auto tile_rows = [&](int t) -> const float * {  // first activation row of the block's t-th tile (uniform)
        const int g = t_first + t;
        const int b = g / tiles_per_sample, rt = g - b * tiles_per_sample;
        return A + (int64_t)b * d.a_bs + (int64_t)rt * TR * d.lda;
    };
    const float *is_x = tile_rows(0);
    auto issue = [&]() {
        float *sb = gd_lds + is_slot * STAGE_FLOATS;
        const int k0 = (half_tail && is_s == nst - 1) ? K - 32 : 32 * is_s;
        const float *xk = is_x + k0, *wk = W + k0;
#pragma unroll
        for (int j = 0; j < NP; j++)
            __builtin_amdgcn_global_load_lds(GD_GLB_PTR((is_w[j] ? wk : xk) + off[j]), GD_LDS_PTR(sb + dst[j]), 16, 0, 0);
        is_slot = is_slot + 1 == D ? 0 : is_slot + 1;
        if (++is_s == nst) {
            is_s = 0;
            if (++is_t < ntiles) is_x = tile_rows(is_t);
        }
    };
    int issued = 0;
#pragma unroll
    for (int i = 0; i < D - 1; i++)
        if (issued < total_steps) { issue(); issued++; }

    int xoff[MTW][2], woff[NTW][2];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        const int r = (wm * MTW + mt) * 16 + lc;
#pragma unroll
        for (int g = 0; g < 2; g++) xoff[mt][g] = r * 32 + 4 * ((4 * g + lq) ^ ((r >> 1) & 7));
    }
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int r = (wn * NTW + nt) * 16 + lc;
#pragma unroll
        for (int g = 0; g < 2; g++) woff[nt][g] = TR * 32 + r * 32 + 4 * ((4 * g + lq) ^ ((r >> 1) & 7));
    }
    floatx4 acc[MTW][NTW];
    auto group = [&](const float *sb, int g) {
        floatx4 xf[MTW], wf[NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) xf[mt] = *reinterpret_cast<const floatx4 *>(sb + xoff[mt][g]);
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) wf[nt] = *reinterpret_cast<const floatx4 *>(sb + woff[nt][g]);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                for (int mt = 0; mt < MTW; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][j], xf[mt][j], acc[mt][nt], 0, 0, 0);
    };
    // wait + barrier that make step `done` (counted over the whole block) readable, then the refill of the slot read one step ago
    int done = 0, rd_slot = 0;
    auto turn = [&]() {
        if (total_steps - 1 - done >= D - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 2) * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < total_steps) { issue(); issued++; }
    };
    static_assert(D == 3, "the counted waits are written for a ring of three");

    // the channel tile's bias: the same for every row tile of the block
    floatx4 bpre[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int n = min(n0 + (wn * NTW + nt) * 16 + 4 * lq, d.N - 4);
        bpre[nt] = d.has_bias ? *reinterpret_cast<const floatx4 *>(bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
    }

    for (int t = 0; t < ntiles; t++) {
#pragma unroll
        for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
        // the residual rows of this tile are requested before its product (clamped addresses): their round trip runs under it
        const int g_t = t_first + t;
        const int b = g_t / tiles_per_sample, rt = g_t - b * tiles_per_sample;
        floatx4 rpre[MTW][NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const int n = min(n0 + (wn * NTW + nt) * 16 + 4 * lq, d.N - 4);
#pragma unroll
            for (int mt = 0; mt < MTW; mt++) {
                const int64_t m = (int64_t)rt * TR + (wm * MTW + mt) * 16 + lc;
                rpre[mt][nt] = d.has_res ? *reinterpret_cast<const floatx4 *>(res + (int64_t)b * d.r_bs + m * d.ldr + n) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // (the residual loads above are younger than the stage pieces in flight: the counted waits below still cover the stages,
        // because vmcnt completes in order and waits for "all but the N youngest" -- so the residual, when there is one, is simply
        // waited for along with them: has_res launches pay that, the expand convs this kernel is for have none)
        for (int s_ = 0; s_ < nfs; s_++) {
            if (d.has_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            turn();
            const float *sb = gd_lds + rd_slot * STAGE_FLOATS;
            group(sb, 0);
            group(sb, 1);
            rd_slot = rd_slot + 1 == D ? 0 : rd_slot + 1;
            done++;
        }
        if (half_tail) {
            if (d.has_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            turn();
            group(gd_lds + rd_slot * STAGE_FLOATS, 1);
            rd_slot = rd_slot + 1 == D ? 0 : rd_slot + 1;
            done++;
        }
        // ---- epilogue of the tile (the next tile's first stages are already in flight)
        float v[MTW * NTW * 4];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const floatx4 bv = bpre[nt];
#pragma unroll
            for (int mt = 0; mt < MTW; mt++)
#pragma unroll
                for (int i = 0; i < 4; i++) v[(mt * NTW + nt) * 4 + i] = acc[mt][nt][i] + bv[i];
        }
        gd_act<MTW * NTW * 4>(d.act, d.p0, d.p1, v);
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) {
            const int64_t m = (int64_t)rt * TR + (wm * MTW + mt) * 16 + lc;
            float *crow = C + (int64_t)b * d.c_bs + m * d.ldc;
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) {
                const int n = n0 + (wn * NTW + nt) * 16 + 4 * lq;
                if (n < d.N) {
                    floatx4 o = floatx4{v[(mt * NTW + nt) * 4], v[(mt * NTW + nt) * 4 + 1], v[(mt * NTW + nt) * 4 + 2], v[(mt * NTW + nt) * 4 + 3]};
                    if (d.has_res) o += rpre[mt][nt];
                    *reinterpret_cast<floatx4 *>(crow + n) = o;
                }
            }
        }
    }
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int MTW, int NTW, int WM, int WN, int KS, int D>
void launch_cfg(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias, const float *res, const float *scale,
                int64_t batch, const SeInline *se) {
    constexpr int TR = 16 * MTW * WM, BN = 16 * NTW * WN;
    const int tps = (int)(d.rows / TR);
    const int gate_floats = d.has_scale ? (d.K + 1023) / 1024 * 1024 : 0;
    const size_t lds = gemm_dma_lds_bytes(d, MTW, NTW, WM, WN, KS, D, se ? se->se.Cr : 0);
    dim3 grid((unsigned)(batch * tps), (unsigned)((d.N + BN - 1) / BN));
    SeInline none{};
    if (d.se_inline && se)
        hipLaunchKernelGGL((gemm_dma_kernel<MTW, NTW, WM, WN, KS, D, 2>), grid, dim3(64 * WM * WN * KS), lds, s, d, C, A, W, bias, res, scale, tps, gate_floats, *se);
    else if (d.has_scale)
        hipLaunchKernelGGL((gemm_dma_kernel<MTW, NTW, WM, WN, KS, D, 1>), grid, dim3(64 * WM * WN * KS), lds, s, d, C, A, W, bias, res, scale, tps, gate_floats, none);
    else
        hipLaunchKernelGGL((gemm_dma_kernel<MTW, NTW, WM, WN, KS, D, 0>), grid, dim3(64 * WM * WN * KS), lds, s, d, C, A, W, bias, res, scale, tps, gate_floats, none);
}

}  // namespace

#ifdef BN_GD_STAMPS
void gemm_dma_read_stamps(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(bn_gd_stamps), sizeof(unsigned long long) * 8 * 96 * 3); }
#endif

void register_gemm_dma_kernels() {
#define GD_REG1(MTW, NTW, WM, WN, KS)                                                                               \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma_kernel<MTW, NTW, WM, WN, KS, 3, 0>));       \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma_kernel<MTW, NTW, WM, WN, KS, 3, 1>));       \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma_kernel<MTW, NTW, WM, WN, KS, 3, 2>));
#define GD_REG(MTW, NTW, WM, WN) GD_REG1(MTW, NTW, WM, WN, 1) GD_REG1(MTW, NTW, WM, WN, 2)
    GD_REG(1, 1, 4, 1) GD_REG(1, 1, 2, 1) GD_REG(1, 2, 4, 1) GD_REG(1, 3, 4, 1) GD_REG(1, 4, 4, 1) GD_REG(1, 5, 4, 1) GD_REG(1, 6, 4, 1) GD_REG(1, 7, 4, 1) GD_REG(1, 8, 4, 1)
    GD_REG(1, 2, 2, 1) GD_REG(1, 3, 2, 1) GD_REG(1, 4, 2, 1) GD_REG(1, 5, 2, 1) GD_REG(1, 6, 2, 1) GD_REG(1, 7, 2, 1) GD_REG(1, 8, 2, 1)
    GD_REG(3, 1, 1, 2) GD_REG(3, 1, 1, 4) GD_REG(3, 2, 1, 4)
#define GD_REG_S(NTW) \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma_stream_kernel<1, NTW, 4, 1, 3>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma_stream_kernel<1, NTW, 2, 1, 3>)); \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma_stream_kernel<2, NTW, 4, 1, 3>));
    GD_REG_S(4) GD_REG_S(5) GD_REG_S(6) GD_REG_S(7) GD_REG_S(8)
#undef GD_REG_S
#undef GD_REG
#undef GD_REG1
}

bool launch_gemm_dma(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias, const float *res, const float *scale,
                     int64_t batch, const SeInline *se) {
    const int shape = gemm_dma_shape(d);
    if (d.gap && (shape != 2 || !gemm_gap_shape_ok(d))) return false;  // (the caller reports it: no other kernel pools in its epilogue)
    if (!shape || !al16(A) || !al16(W) || !al16(C) || (d.has_res && !al16(res)) || (d.has_bias && !al16(bias)) ||
        (d.has_scale && !d.se_inline && !al16(scale)))
        return false;
    if (d.se_inline && !se) return false;
    if (shape == 3) {
        // streaming form: the fewest channel tiles of at most 128 (whole 16-channel tiles, evenly sized); 64-row tiles where a
        // sample's rows allow; tiles per block so that the launch is about two blocks per CU deep (two fit a CU: 74 KB each) --
        // neither enters the arithmetic
        int ntw = 8;  // channel tile of 64 .. 128 with the least padding over the layer (ties: the wider tile)
        {
            int64_t best = -1;
            for (int c = 8; c >= 4; c--) {
                const int64_t padded = (int64_t)((d.N + 16 * c - 1) / (16 * c)) * 16 * c;
                if (best < 0 || padded < best) { best = padded; ntw = c; }
            }
        }
        const bool r64 = d.rows % 64 == 0;
        const bool r128 = d.rows % 128 == 0 && getenv("BN_GEMMSTREAM_TR") && atoi(getenv("BN_GEMMSTREAM_TR")) == 128;  // experiment: 128-row tiles measured slower (191 against 125 us)
        const int tr = r128 ? 128 : r64 ? 64 : 32;
        const int tps = (int)(d.rows / tr);
        const int64_t tiles = batch * tps, nbn = (d.N + 16 * ntw - 1) / (16 * ntw);
        const int64_t want_blocks = 2 * (int64_t)device_cu_count();
        const int force_tpb = getenv("BN_GEMMSTREAM_TPB") ? atoi(getenv("BN_GEMMSTREAM_TPB")) : 0;  // tests / experiments
        // (whole rounds: grid.x * nbn must not exceed what is resident at once, or a handful of left-over blocks run a round of their own --
        // measured: 516 blocks on 512 slots 180 us, 474 blocks 125 us)
        const int64_t gx = std::max<int64_t>(1, want_blocks / nbn);
        const int tpb = force_tpb > 0 ? force_tpb : (int)std::max<int64_t>(1, (tiles + gx - 1) / gx);
        dim3 grid((unsigned)((tiles + tpb - 1) / tpb), (unsigned)nbn);
        const size_t lds = (size_t)3 * (tr + 16 * ntw) * 32 * sizeof(float);
#define GD_STREAM(NTW)                                                                                                                         \
    do {                                                                                                                                       \
        if (r128) hipLaunchKernelGGL((gemm_dma_stream_kernel<2, NTW, 4, 1, 3>), grid, dim3(256), lds, s, d, C, A, W, bias, res, tps, (int)tiles, tpb); \
        else if (r64) hipLaunchKernelGGL((gemm_dma_stream_kernel<1, NTW, 4, 1, 3>), grid, dim3(256), lds, s, d, C, A, W, bias, res, tps, (int)tiles, tpb); \
        else hipLaunchKernelGGL((gemm_dma_stream_kernel<1, NTW, 2, 1, 3>), grid, dim3(128), lds, s, d, C, A, W, bias, res, tps, (int)tiles, tpb);      \
    } while (0)
        switch (ntw) {
            case 4: GD_STREAM(4); break;
            case 5: GD_STREAM(5); break;
            case 6: GD_STREAM(6); break;
            case 7: GD_STREAM(7); break;
            default: GD_STREAM(8); break;
        }
#undef GD_STREAM
        return true;
    }
    const int ks = gemm_dma_kslices(d, se ? se->se.Cr : 0);  // decided for the layer, not for the tile (plan_rules.h)
    // Tile shape by the size of the launch: big tiles (16 - 24 flop per byte staged from L2) as soon as they give 64 blocks,
    // smaller ones below that.  With four contexts in flight a step is bound by the SUM of its launches' marginal costs,
    // not by any launch's own latency (three contexts already reach 94 % of four), so the efficient tile wins even when
    // one launch alone leaves CUs idle: measured 54.6 k -> 55.5 k segments/s against a threshold of 192 blocks.
    // BN_GEMMDMA_MINBLOCKS moves the line.
    const int64_t min_blocks = getenv("BN_GEMMDMA_MINBLOCKS") ? atoll(getenv("BN_GEMMDMA_MINBLOCKS")) : (device_context_count() > 1 ? 1 : 64);  // (a shared device: the efficient tile always, +0.7 %)
#define GD_GO(MTW, NTW, WM, WN)                                                                  \
    do {                                                                                         \
        if (ks == 2) launch_cfg<MTW, NTW, WM, WN, 2, 3>(s, d, C, A, W, bias, res, scale, batch, se); \
        else launch_cfg<MTW, NTW, WM, WN, 1, 3>(s, d, C, A, W, bias, res, scale, batch, se);         \
    } while (0)
#define GD_GO_N(WM)                                 \
    do {                                            \
        switch (ntw) {                              \
            case 1: GD_GO(1, 1, WM, 1); break;      \
            case 2: GD_GO(1, 2, WM, 1); break;      \
            case 3: GD_GO(1, 3, WM, 1); break;      \
            case 4: GD_GO(1, 4, WM, 1); break;      \
            case 5: GD_GO(1, 5, WM, 1); break;      \
            case 6: GD_GO(1, 6, WM, 1); break;      \
            case 7: GD_GO(1, 7, WM, 1); break;      \
            default: GD_GO(1, 8, WM, 1); break;     \
        }                                           \
    } while (0)
    if (shape == 1) {
        // the fewest channel blocks of at most 128, evenly sized, whole 16-channel tiles
        const int nb = (d.N + 127) / 128;
        const int ntw = ((d.N + nb - 1) / nb + 15) / 16;
        const bool big = d.rows % 64 == 0 && batch * (d.rows / 64) * nb >= min_blocks;
        if (big) GD_GO_N(4);
        else GD_GO_N(2);
    } else {
        const int64_t tiles = batch * (d.rows / 48);
        if (d.N > 512 && tiles * ((d.N + 127) / 128) >= min_blocks) GD_GO(3, 2, 1, 4);
        else if (tiles * ((d.N + 63) / 64) >= min_blocks) GD_GO(3, 1, 1, 4);
        else GD_GO(3, 1, 1, 2);
    }
#undef GD_GO_N
#undef GD_GO
    return true;
}

}  // namespace bn
