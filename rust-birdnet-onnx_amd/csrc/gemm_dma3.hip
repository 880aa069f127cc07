// 1x1-conv / MatMul GEMM on the BF16 matrix pipe with f32-complete products: "f32 by three exact bf16 terms" (gfx950 only; round 5).
//
// Why.  gfx950 has no f32 matrix datapath of its own: v_mfma_f32_16x16x4_f32 runs on the f32 vector ALU (tools/mfma_valu_probe.cpp: a
// wave of it and a wave of v_fma_f32 take the SUM of their solo times) at 1/16 of the bf16 matrix rate, and every vector instruction of
// the kernel is paid out of the same cycles.  The bf16 pipe is separate: v_mfma_f32_16x16x32_bf16 retires 8192 multiply-adds in 16
// cycles and holds the SIMD's vector issue for only 8 of them (MI355X_MICROARCH.md, cycle constants).  An f32 number is EXACTLY the sum
// of three bf16 numbers -- hi = its top 8 significand bits (the f32 with the low 16 bits cleared), mid = the same of the remainder
// x - hi (exact), lo = x - hi - mid (8 significant bits left: a bf16 number) -- so
//     x w = (xh + xm + xl)(wh + wm + wl) = xh wh + xh wm + xm wh + xm wm + xh wl + xl wh   + [xm wl + xl wm + xl wl]
// and the six products kept here are each EXACT in the accumulator's f32 (8 x 8 significand bits); of the three dropped ones
// xm wl and xl wm are below 2^-22 |x w| each (|mid| < 2^-7 |x|, |lo| < 2^-15 |x|) and xl wl below 2^-30 |x w|.  A product therefore carries
// a relative error below 2^-21 in the worst case and 2^-24 in the root mean square (tests/test_host_logic.py restates the arithmetic in
// numpy) -- the size of the rounding the exact-f32 instruction commits per product as well (it rounds x w + acc); sums are f32 either way,
// and the measured error of a K-deep dot product against double precision equals the exact-f32 kernel's (tools/gemm3_bench).  Six bf16 instructions of
// 32-deep k replace eight f32 instructions of 4-deep k per 16 x 16 tile and K step: 96 against 256 matrix cycles, and the split's
// vector instructions (two AND, two SUB, 1.5 PERM per activation value) issue in the matrix instructions' shadow instead of beside them.
// Weights are split once, by the planner (plan_rules.h, pack_w3): three bf16 planes [plane][N][Kp], k permuted inside every 32-deep step
// into the order the activation fragment reads deliver.
//
// Everything else is gemm_dma.hip's structure: both operands by global_load_lds_dwordx4 into a ring of stages (activations f32, 8 rows x
// 128 B per piece; weights bf16, 16 rows x 64 B per piece and plane; XOR swizzles on the per-lane SOURCE address so that the ds_read_b128
// fragment reads are conflict free), one raw barrier and one counted wait per K step, weights as the A operand so that lane (c, q) holds
// four consecutive output channels of row c, K slices summed through LDS in slice order, the squeeze-excite gate multiplied into the
// activation fragment (f32, before the split -- the product the f32 kernel rounds too), bias / activation / residual / pooled epilogue.
//
// Arithmetic: an output element is one accumulation chain -- K steps ascending (per slice), inside a step the six partial products in
// the fixed order (wl xh, wh xl, wm xm, wm xh, wh xm, wh xh), each a 32-deep instruction -- independent of the tile shape and of the
// batch.  The bits differ from the exact-f32 kernels' (another summation order and the dropped 2^-24 terms); against the oracle the
// results sit inside the same tolerance, an order of magnitude below it (tests/test_gpu_ops.py::test_gemm_bf16x3_*).
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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void g3_act(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
    else if (act == ACT_SIGMOID) map_array<N>(v, [](float x) { return net_sigmoid(x); });
    else if (act == ACT_HSIGMOID) map_array<N>(v, [=](float x) { return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f); });
}

#define G3_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define G3_GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// the eight f32 values of a lane's k group -> three vectors of eight bf16 (hi, mid, lo), exactly: x = hi + mid + lo
__device__ __forceinline__ void split3(const floatx4 &a, const floatx4 &b, u32x4 &hi, u32x4 &mid, u32x4 &lo) {
    float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r1[j] = x[j] - __uint_as_float(__float_as_uint(x[j]) & 0xffff0000u);    // exact: the low 16 significand bits
        r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xffff0000u);  // exact: at most 8 significant bits are left
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {  // bf16 element 2p in the low half, 2p + 1 in the high half: the top 16 bits of each f32
        hi[p] = __builtin_amdgcn_perm(__float_as_uint(x[2 * p + 1]), __float_as_uint(x[2 * p]), 0x07060302u);
        mid[p] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * p + 1]), __float_as_uint(r1[2 * p]), 0x07060302u);
        lo[p] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * p + 1]), __float_as_uint(r2[2 * p]), 0x07060302u);
    }
}

__device__ __forceinline__ floatx4 mm(const u32x4 &w, const u32x4 &x, const floatx4 &acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// MTW x NTW 16x16 tiles per wave, WM x WN waves per K slice, KS K slices (slice ks takes the 32-deep K steps ks, ks + KS, ...
// through a ring of D stages of its own), block = 64 WM WN KS threads, tile = (16 MTW WM) rows x (16 NTW WN) channels.
// Stage image (bytes): [TR rows x 128: f32 activations, chunk c of row r in slot c ^ ((r >> 1) & 7)]
//                      [3 planes x BN rows x 64: bf16 weights, chunk c of row r in slot c ^ (-(r >> 2) & 3): conflict free for ds_read_b128's
//                       four non-contiguous 16-lane groups (gemm_b3.hip has the derivation)]
// GATE as in gemm_dma_kernel: 0 plain, 1 gate read from memory, 2 gate computed in the prologue from the squeeze partial sums.
template <int MTW, int NTW, int WM, int WN, int KS, int D, int GATE>
__global__ __launch_bounds__(64 * WM * WN * KS) void gemm_dma3_kernel(GemmDesc d, float *__restrict__ C, const float *__restrict__ A, const uint16_t *__restrict__ W3,
                                                             const float *__restrict__ bias, const float *__restrict__ res,
                                                             const float *__restrict__ scale, int tiles_per_sample, int gate_floats, SeInline sei) {
    constexpr bool GATED = GATE != 0;
    constexpr int WPS = WM * WN;
    static_assert(WPS == 2 || WPS == 4 || WPS == 8, "two, four or eight waves per K slice");
    static_assert(D == 2 || D == 3, "ring of two or three stages");
    constexpr int TR = 16 * MTW * WM, BN = 16 * NTW * WN;
    constexpr int XP = TR / 8, WPP = BN / 16, PIECES = XP + 3 * WPP;  // 1-KiB pieces of one stage: activations, then three weight planes
    constexpr int NP = (PIECES + WPS - 1) / WPS;
    constexpr int X_BYTES = TR * 128, P_BYTES = BN * 64, STAGE_BYTES = X_BYTES + 3 * P_BYTES;
    extern __shared__ __align__(1024) float g3_lds[];
    char *lds = reinterpret_cast<char *>(g3_lds);
    float *gate = reinterpret_cast<float *>(lds + KS * D * STAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / WPS, w4 = wave % WPS;
    const int lc = lane & 15, lq = lane >> 4;
    const int wm = w4 % WM, wn = w4 / WM;
    const int b = blockIdx.x / tiles_per_sample, rt = blockIdx.x - b * tiles_per_sample;
    const int n0 = blockIdx.y * BN;
    const int K = d.K;
    const int Kp = (K + 31) & ~31;  // the planes' row pitch (bf16 elements)
    const float *Xb = A + (int64_t)b * d.a_bs + (int64_t)rt * TR * d.lda;
    char *ring = lds + ks * D * STAGE_BYTES;

    // ---- per-lane source offsets of this wave's pieces (activations: floats relative to Xb; weights: bf16 elements relative to W3)
    uint32_t off[NP];
    bool is_w[NP];
    uint32_t dst[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        int p = w4 + WPS * j;
        if (p >= PIECES) p = w4 % PIECES;  // a wave without a piece of its own repeats one (same bytes, same place)
        const bool w_img = p >= XP;
        is_w[j] = w_img;
        if (!w_img) {
            const int row = 8 * p + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            off[j] = (uint32_t)row * (uint32_t)K + 4u * (uint32_t)chunk;
            dst[j] = (uint32_t)(p * 1024);
        } else {
            const int pw = p - XP, plane = pw / WPP, rblk = pw - plane * WPP;
            const int r16 = lane >> 2;
            const int chunk = (lane & 3) ^ ((0 - (r16 >> 2)) & 3);
            int grow = n0 + 16 * rblk + r16;
            grow = grow < d.N ? grow : d.N - 1;
            off[j] = ((uint32_t)plane * (uint32_t)d.N + (uint32_t)grow) * (uint32_t)Kp + 8u * (uint32_t)chunk;
            dst[j] = (uint32_t)(X_BYTES + plane * P_BYTES + rblk * 1024);
        }
    }
    const int nfs = K >> 5;
    const bool half_tail = (K & 31) != 0;
    const int nst = nfs + (half_tail ? 1 : 0);
    const int nmine = (nst - ks + KS - 1) / KS;
    const int niter = (nst + KS - 1) / KS;
    const int nmain = nfs / KS;
    auto issue = [&](int i) {
        const int s = ks + i * KS;
        char *sb = ring + (i % D) * STAGE_BYTES;
        const float *xk = Xb + ((half_tail && s == nst - 1) ? K - 32 : 32 * s);  // the half step re-reads columns K-32 .. K-1 (memory that exists)
        const uint16_t *wk = W3 + 32 * s;                                       // ... the planes hold their own (zero-padded) last step
#pragma unroll
        for (int j = 0; j < NP; j++) {
            if (is_w[j]) __builtin_amdgcn_global_load_lds(G3_GLB_PTR(wk + off[j]), G3_LDS_PTR(sb + dst[j]), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(G3_GLB_PTR(xk + off[j]), G3_LDS_PTR(sb + dst[j]), 16, 0, 0);
        }
    };

    if constexpr (GATE == 1) {
        const float *gsrc = scale + (int64_t)b * d.s_bs;
        const int n16 = K >> 2;
        for (int c0 = wave * 64; c0 < gate_floats / 4; c0 += 64 * WPS * KS) {
            int c = c0 + lane;
            c = c < n16 ? c : n16 - 1;
            __builtin_amdgcn_global_load_lds(G3_GLB_PTR(gsrc + 4 * c), G3_LDS_PTR(gate + 4 * c0), 16, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < D - 1; i++)
        if (i < nmine) issue(i);

    if constexpr (GATE == 2) {  // the excite products of this block's sample (fixed orders; see gemm_dma_kernel)
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
                g3_act<1>(se.act1, se.p0_1, se.p1_1, hv);
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
            g3_act<1>(se.act2, se.p0_2, se.p1_2, gv);
            gate[c] = gv[0];
        }
    }

    floatx4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes inside a stage)
    int xoff[MTW][2], woff[NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
        const int r = (wm * MTW + mt) * 16 + lc;
#pragma unroll
        for (int g = 0; g < 2; g++) xoff[mt][g] = r * 128 + 16 * ((4 * g + lq) ^ ((r >> 1) & 7));
    }
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int r = (wn * NTW + nt) * 16 + lc;
        woff[nt] = X_BYTES + r * 64 + 16 * (lq ^ ((0 - (lc >> 2)) & 3));
    }

    // one 32-deep K step of the stage at sb.  kcol: first column of the stage (gate index); TAIL: the stage holds columns K-32 .. K-1
    // and only its second half (the last 16 columns) is new -- the first half is multiplied as zeros
    auto step = [&](const char *sb, int kcol, bool tail) {
        u32x4 xh[MTW], xm[MTW], xl[MTW];
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) {
            floatx4 x0 = *reinterpret_cast<const floatx4 *>(sb + xoff[mt][0]);
            floatx4 x1 = *reinterpret_cast<const floatx4 *>(sb + xoff[mt][1]);
            if constexpr (GATED) {
                x0 *= *reinterpret_cast<const floatx4 *>(gate + kcol + 4 * lq);
                x1 *= *reinterpret_cast<const floatx4 *>(gate + kcol + 16 + 4 * lq);
            }
            if (tail) x0 = floatx4{0.f, 0.f, 0.f, 0.f};
            split3(x0, x1, xh[mt], xm[mt], xl[mt]);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const u32x4 wh = *reinterpret_cast<const u32x4 *>(sb + woff[nt]);
            const u32x4 wmd = *reinterpret_cast<const u32x4 *>(sb + woff[nt] + P_BYTES);
            const u32x4 wl = *reinterpret_cast<const u32x4 *>(sb + woff[nt] + 2 * P_BYTES);
#pragma unroll
            for (int mt = 0; mt < MTW; mt++) {
                floatx4 a = acc[mt][nt];
                a = mm(wl, xh[mt], a);
                a = mm(wh, xl[mt], a);
                a = mm(wmd, xm[mt], a);
                a = mm(wmd, xh[mt], a);
                a = mm(wh, xm[mt], a);
                a = mm(wh, xh[mt], a);
                acc[mt][nt] = a;
            }
        }
    };
    auto turn = [&](int i) {
        if (D == 3 && nmine - 1 - i >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 2) * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + D - 1 < nmine) issue(i + D - 1);
    };

    for (int i = 0; i < nmain; i++) {
        turn(i);
        step(ring + (i % D) * STAGE_BYTES, 32 * (ks + i * KS), false);
    }
    for (int i = nmain; i < niter; i++) {  // ragged end: some slices have a full step, one the half step, some none
        turn(i);
        if (i < nmine) {
            const int s = ks + i * KS;
            const char *sb = ring + (i % D) * STAGE_BYTES;
            if (half_tail && s == nst - 1) step(sb, K - 32, true);
            else step(sb, 32 * s, false);
        }
    }

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

    if constexpr (KS > 1) {  // the slices' partial tiles, summed in slice order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        floatx4 *xch = reinterpret_cast<floatx4 *>(g3_lds);  // [KS - 1][WPS waves][MTW * NTW][64 lanes]
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

    float v[MTW * NTW * 4];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const floatx4 bv = bpre[nt];
#pragma unroll
        for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[(mt * NTW + nt) * 4 + i] = acc[mt][nt][i] + bv[i];
    }
    g3_act<MTW * NTW * 4>(d.act, d.p0, d.p1, v);
    if constexpr (WM == 1) {
        if (d.gap) {  // the sample's mean over its rows (all TR of them sit in this wave): same order as gemm_dma_kernel's
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
                    const float rows = (float)TR;
                    *reinterpret_cast<floatx4 *>(C + (int64_t)b * d.c_bs + n) = floatx4{sm[0] / rows, sm[1] / rows, sm[2] / rows, sm[3] / rows};
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

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int MTW, int NTW, int WM, int WN, int KS, int D>
void launch_cfg3(hipStream_t s, const GemmDesc &d, float *C, const float *A, const uint16_t *W3, const float *bias, const float *res, const float *scale,
                 int64_t batch, const SeInline *se) {
    constexpr int TR = 16 * MTW * WM, BN = 16 * NTW * WN;
    const int tps = (int)(d.rows / TR);
    const int gate_floats = d.has_scale ? (d.K + 1023) / 1024 * 1024 : 0;
    const size_t lds = gemm_dma3_lds_bytes(d, MTW, NTW, WM, WN, KS, D, se ? se->se.Cr : 0);
    dim3 grid((unsigned)(batch * tps), (unsigned)((d.N + BN - 1) / BN));
    SeInline none{};
    if (d.se_inline && se)
        hipLaunchKernelGGL((gemm_dma3_kernel<MTW, NTW, WM, WN, KS, D, 2>), grid, dim3(64 * WM * WN * KS), lds, s, d, C, A, W3, bias, res, scale, tps, gate_floats, *se);
    else if (d.has_scale)
        hipLaunchKernelGGL((gemm_dma3_kernel<MTW, NTW, WM, WN, KS, D, 1>), grid, dim3(64 * WM * WN * KS), lds, s, d, C, A, W3, bias, res, scale, tps, gate_floats, none);
    else
        hipLaunchKernelGGL((gemm_dma3_kernel<MTW, NTW, WM, WN, KS, D, 0>), grid, dim3(64 * WM * WN * KS), lds, s, d, C, A, W3, bias, res, scale, tps, gate_floats, none);
}

}  // namespace

void register_gemm_dma3_kernels() {
#define G3_REG1(MTW, NTW, WM, WN, KS, D)                                                                              \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma3_kernel<MTW, NTW, WM, WN, KS, D, 0>));        \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma3_kernel<MTW, NTW, WM, WN, KS, D, 1>));        \
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(gemm_dma3_kernel<MTW, NTW, WM, WN, KS, D, 2>));
#define G3_REG(MTW, NTW, WM, WN) G3_REG1(MTW, NTW, WM, WN, 1, 3) G3_REG1(MTW, NTW, WM, WN, 2, 2)
    G3_REG(1, 1, 4, 1) G3_REG(1, 2, 4, 1) G3_REG(1, 3, 4, 1) G3_REG(1, 4, 4, 1) G3_REG(1, 5, 4, 1) G3_REG(1, 6, 4, 1) G3_REG(1, 7, 4, 1) G3_REG(1, 8, 4, 1)
    G3_REG(1, 1, 2, 1) G3_REG(1, 2, 2, 1) G3_REG(1, 3, 2, 1) G3_REG(1, 4, 2, 1) G3_REG(1, 5, 2, 1) G3_REG(1, 6, 2, 1) G3_REG(1, 7, 2, 1) G3_REG(1, 8, 2, 1)
    G3_REG(3, 1, 1, 2) G3_REG(3, 1, 1, 4) G3_REG(3, 2, 1, 4)
#undef G3_REG
#undef G3_REG1
}

// d.w3 launches only: W3 is the planner's three-plane bf16 image of the layer's weights (pack_w3).  Returns false (nothing launched)
// when the shape or a pointer's alignment rules the kernel out -- there is no other kernel for a packed layer, the caller reports it.
bool launch_gemm_dma3(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W3f, const float *bias, const float *res, const float *scale,
                      int64_t batch, const SeInline *se) {
    const int shape = gemm_dma_shape(d);
    if (!d.w3 || (shape != 1 && shape != 2)) return false;
    if (d.gap && (shape != 2 || !gemm_gap_shape_ok(d))) return false;
    if (!al16(A) || !al16(W3f) || !al16(C) || (d.has_res && !al16(res)) || (d.has_bias && !al16(bias)) || (d.has_scale && !d.se_inline && !al16(scale))) return false;
    if (d.se_inline && !se) return false;
    const uint16_t *W3 = reinterpret_cast<const uint16_t *>(W3f);
    const int ks = gemm_dma3_kslices(d, se ? se->se.Cr : 0);  // decided for the layer (it enters the summation order), not for the tile
    const int64_t min_blocks = getenv("BN_GEMMDMA_MINBLOCKS") ? atoll(getenv("BN_GEMMDMA_MINBLOCKS")) : (device_context_count() > 1 ? 1 : 64);  // (a shared device: the efficient tile always, +0.7 %)
#define G3_GO(MTW, NTW, WM, WN)                                                                      \
    do {                                                                                             \
        if (ks == 2) launch_cfg3<MTW, NTW, WM, WN, 2, 2>(s, d, C, A, W3, bias, res, scale, batch, se); \
        else launch_cfg3<MTW, NTW, WM, WN, 1, 3>(s, d, C, A, W3, bias, res, scale, batch, se);         \
    } while (0)
#define G3_GO_N(WM)                                 \
    do {                                            \
        switch (ntw) {                              \
            case 1: G3_GO(1, 1, WM, 1); break;      \
            case 2: G3_GO(1, 2, WM, 1); break;      \
            case 3: G3_GO(1, 3, WM, 1); break;      \
            case 4: G3_GO(1, 4, WM, 1); break;      \
            case 5: G3_GO(1, 5, WM, 1); break;      \
            case 6: G3_GO(1, 6, WM, 1); break;      \
            case 7: G3_GO(1, 7, WM, 1); break;      \
            default: G3_GO(1, 8, WM, 1); break;     \
        }                                           \
    } while (0)
    if (shape == 1) {
        const int nb = (d.N + 127) / 128;
        const int ntw = ((d.N + nb - 1) / nb + 15) / 16;
        const bool big = d.rows % 64 == 0 && batch * (d.rows / 64) * nb >= min_blocks;
        if (big) G3_GO_N(4);
        else G3_GO_N(2);
    } else {
        const int64_t tiles = batch * (d.rows / 48);
        if (d.N > 512 && tiles * ((d.N + 127) / 128) >= min_blocks) G3_GO(3, 2, 1, 4);
        else if (tiles * ((d.N + 63) / 64) >= min_blocks) G3_GO(3, 1, 1, 4);
        else G3_GO(3, 1, 1, 2);
    }
#undef G3_GO_N
#undef G3_GO
    return true;
}

}  // namespace bn
