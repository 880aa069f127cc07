// Windowed-DFT filter bank as a real FFT (gfx950, wave64).
//
// The exporters write the STFT of the audio front end as a strided 1-D convolution whose filters are
// window[n] * cos(2 pi k n / L) and -window[n] * sin(2 pi k n / L) (reference: the graph ORT executes at
// src/classifier.rs:637-639, 721-723, 851-853).  Evaluated as a matrix product that is
// O(L) multiplications per output bin; this kernel computes the same numbers with ONE real FFT per frame:
//
//   block = (sample, TPB consecutive frames)
//   1. the PCM span the frames cover ((TPB-1)*hop + L samples; frames overlap, hop << L) is read ONCE from HBM with
//      coalesced 16-byte loads into LDS; an absorbed per-sample normalisation chain (min-max scaling of the v2.4
//      graph) is applied on the way
//   2. a wave transforms 1024/M frames at a time (M = L/2): the windowed frame is packed as M complex numbers
//      z[m] = y[2m] + i y[2m+1], transformed IN PLACE in LDS by decimation-in-frequency passes (radix 4, one radix-2
//      pass first when log2 M is odd) down to 16-point sub-problems, which every lane finishes in registers; the
//      result sits in digit-reversed order, which costs nothing because
//   3. only the bins the mel filter bank uses are produced: output c is a fixed linear combination
//      a*Z[k] + b*Z*[M-k] ("untangling" the packed transform, folded with the filter's own amplitude and phase)
//      of two LDS positions, both precomputed at plan time
//   4. optionally the mel filter bank (sparse: triangles), its compression chain and the layout copy into the CNN's
//      input image follow in the same launch (spectrum and mel rows stay in LDS)
//
// Per frame: 5 M log2 M flops instead of 2 L nbins; LDS traffic ~ 80 KB.  Numerics: the FFT's rounding error grows
// with log2 L instead of L; the filters themselves are taken as exactly window x cosine (the plan-time detector
// accepts a bank only if every tap agrees within f32 rounding).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "device_common.h"
#include "kernels.h"
#include "plan_rules.h"

namespace bn {
namespace {

// complex slots of one wave's transform buffer: SLOTS / 16 blocks of 16 + 2 slots of padding each.  SLOTS = 1024 with 8 waves per
// block, or 512 with 16 waves (M <= 512: one frame per wave at a time) -- the same LDS either way, but four waves per SIMD
// instead of two to cover the LDS round trips and barriers the kernel is bound by

__device__ __forceinline__ int phys(int i) { return i + 2 * (i >> 4); }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

// forward radix-4 butterfly (omega_4 = -i): v_p = sum_m u_m omega_4^(m p)
__device__ __forceinline__ void bfly4(float2 &u0, float2 &u1, float2 &u2, float2 &u3) {
    const float2 t0 = make_float2(u0.x + u2.x, u0.y + u2.y), t1 = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 t2 = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 t3 = make_float2(u1.y - u3.y, -(u1.x - u3.x));  // -i (u1 - u3)
    u0 = make_float2(t0.x + t2.x, t0.y + t2.y);
    u2 = make_float2(t0.x - t2.x, t0.y - t2.y);
    u1 = make_float2(t1.x + t3.x, t1.y + t3.y);
    u3 = make_float2(t1.x - t3.x, t1.y - t3.y);
}

// forward radix-5 butterfly (omega_5 = e^(-2 pi i / 5)): v_p = sum_m u_m omega_5^(m p)
//   t1 = u1 + u4, t2 = u2 + u3, t3 = u1 - u4, t4 = u2 - u3;  a1 = u0 + c1 t1 + c2 t2, a2 = u0 + c2 t1 + c1 t2;
//   b1 = s1 t3 + s2 t4, b2 = s2 t3 - s1 t4;  v1 = a1 - i b1, v4 = a1 + i b1, v2 = a2 - i b2, v3 = a2 + i b2
__device__ __forceinline__ void bfly5(float2 &u0, float2 &u1, float2 &u2, float2 &u3, float2 &u4) {
    constexpr float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;  // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;   // sin(2 pi / 5), sin(4 pi / 5)
    const float2 t1 = make_float2(u1.x + u4.x, u1.y + u4.y), t2 = make_float2(u2.x + u3.x, u2.y + u3.y);
    const float2 t3 = make_float2(u1.x - u4.x, u1.y - u4.y), t4 = make_float2(u2.x - u3.x, u2.y - u3.y);
    const float2 a1 = make_float2(u0.x + C1 * t1.x + C2 * t2.x, u0.y + C1 * t1.y + C2 * t2.y);
    const float2 a2 = make_float2(u0.x + C2 * t1.x + C1 * t2.x, u0.y + C2 * t1.y + C1 * t2.y);
    const float2 b1 = make_float2(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y);
    const float2 b2 = make_float2(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
    u0 = make_float2(u0.x + t1.x + t2.x, u0.y + t1.y + t2.y);
    u1 = make_float2(a1.x + b1.y, a1.y - b1.x);  // a1 - i b1
    u4 = make_float2(a1.x - b1.y, a1.y + b1.x);
    u2 = make_float2(a2.x + b2.y, a2.y - b2.x);
    u3 = make_float2(a2.x - b2.y, a2.y + b2.x);
}

// one in-place DIF pass over the wave's 1024 slots: sub-problems of size n, radix R, twiddles tw[(p-1)*q + j].
// Two butterflies per lane are in flight at a time (register budget: 256 per lane at 8 waves per CU).
template <int R, int SLOTS>
__device__ __forceinline__ void strided_pass(float2 *__restrict__ x, const float2 *__restrict__ tw, int logn, int lane) {
    constexpr int LR = R == 4 ? 2 : 1;
    const int lq = logn - LR, q = 1 << lq;
    constexpr int PER_LANE = SLOTS / R / 64, INFL = PER_LANE < 4 ? PER_LANE : 4;
#pragma unroll 1
    for (int h = 0; h < PER_LANE; h += INFL) {
        float2 u[INFL][R];
        int base[INFL], j[INFL];
#pragma unroll
        for (int t = 0; t < INFL; t++) {
            // which butterflies a lane takes is free: with q = 16 a half wave reads two 64-blocks, and under the 2-per-16 padding
            // neighbouring blocks b, b + 1 overlap on the banks (72 b mod 32) while b, b + 2 do not -- lanes 16-31 take block b + 2
            const int g0 = lane + 64 * (h + t);
            const int g = lq == 4 ? ((g0 & ~0x30) | ((g0 & 0x10) << 1) | ((g0 & 0x20) >> 1)) : g0;
            j[t] = g & (q - 1);
            base[t] = ((g >> lq) << logn) + j[t];
#pragma unroll
            for (int m = 0; m < R; m++) u[t][m] = x[phys(base[t] + m * q)];
        }
#pragma unroll
        for (int t = 0; t < INFL; t++) {
            if constexpr (R == 4) {
                const float2 w1 = tw[j[t]], w2 = tw[q + j[t]], w3 = tw[2 * q + j[t]];
                bfly4(u[t][0], u[t][1], u[t][2], u[t][3]);
                x[phys(base[t])] = u[t][0];
                x[phys(base[t] + q)] = cmul(u[t][1], w1);
                x[phys(base[t] + 2 * q)] = cmul(u[t][2], w2);
                x[phys(base[t] + 3 * q)] = cmul(u[t][3], w3);
            } else {
                const float2 w1 = tw[j[t]];
                x[phys(base[t])] = make_float2(u[t][0].x + u[t][1].x, u[t][0].y + u[t][1].y);
                x[phys(base[t] + q)] = cmul(make_float2(u[t][0].x - u[t][1].x, u[t][0].y - u[t][1].y), w1);
            }
        }
    }
}

// the last two radix-4 passes (n = 16, n = 4) of one 16-slot block, in registers
__device__ __forceinline__ void block16(float2 *__restrict__ blk) {
    float2 x[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float4 v = *reinterpret_cast<const float4 *>(blk + 2 * i);
        x[2 * i] = make_float2(v.x, v.y);
        x[2 * i + 1] = make_float2(v.z, v.w);
    }
    // n = 16, q = 4: element j + 4p <- v_p * omega_16^(j p)
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R2 = 0.70710678118654752440f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        bfly4(x[j], x[j + 4], x[j + 8], x[j + 12]);
        if (j == 1) {
            x[5] = cmul(x[5], make_float2(C1, -S1));
            x[9] = cmul(x[9], make_float2(R2, -R2));
            x[13] = cmul(x[13], make_float2(S1, -C1));
        } else if (j == 2) {
            x[6] = cmul(x[6], make_float2(R2, -R2));
            x[10] = make_float2(x[10].y, -x[10].x);  // omega_16^4 = -i
            x[14] = cmul(x[14], make_float2(-R2, -R2));
        } else if (j == 3) {
            x[7] = cmul(x[7], make_float2(S1, -C1));
            x[11] = cmul(x[11], make_float2(-R2, -R2));
            x[15] = cmul(x[15], make_float2(-C1, S1));  // omega_16^9
        }
    }
    // n = 4, q = 1: no twiddles
#pragma unroll
    for (int b4 = 0; b4 < 4; b4++) bfly4(x[4 * b4], x[4 * b4 + 1], x[4 * b4 + 2], x[4 * b4 + 3]);
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<float4 *>(blk + 2 * i) = make_float4(x[2 * i].x, x[2 * i].y, x[2 * i + 1].x, x[2 * i + 1].y);
}

// One shared copy of the stage dispatch for the mel phase's four-value arrays (by value: registers in, registers out).  Inlined
// at its five call sites the dispatch -- a compare chain over every stage code, each with a four-element body -- made the
// compression chain of 4 values cost 2.4 us per tile (instruction fetch, not arithmetic: 9.5 of the kernel's 61 us).
struct Vals4 { float x[4]; };
__device__ __noinline__ Vals4 act_small4(int act, float p0, float p1, Vals4 v) {
    act_small<4>(act, p0, p1, v.x);
    return v;
}

// contiguous global -> LDS copy without registers (global_load_lds_dwordx4: each lane names its 16 source bytes, the
// wave writes 1 KiB at a wave-uniform LDS address).  `floats` is rounded up to whole 16-byte chunks; lanes past the
// end re-read the last chunk into the region's padding.  Completion: s_waitcnt vmcnt(0) by every wave, then a barrier.
template <int NW>
__device__ __forceinline__ void async_copy(float *lds_dst, const float *gsrc, int floats, int wave, int lane) {
    const int n16 = (floats + 3) >> 2;
    for (int c0 = wave * 64; c0 < n16; c0 += NW * 64) {
        int c = c0 + lane;
        c = c < n16 ? c : n16 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + 4 * c),
                                         (__attribute__((address_space(3))) void *)(lds_dst + 4 * c0), 16, 0, 0);
    }
}

constexpr int SPAN_FLOATS = 8192;  // a tile's signal span is staged in registers: SPAN_FLOATS / (4 threads) float4 per thread

// first DIF pass fused with the frame load: element i of frame fi is (w[2i] x[2i], w[2i+1] x[2i+1]) straight from the span
template <int R, int SLOTS>
__device__ __forceinline__ void first_pass(float2 *__restrict__ x, const float2 *__restrict__ tw, const float *__restrict__ sig,
                                           const float2 *__restrict__ wnd, int M, int logM, int hop, int frame0, int rows_here, int lane) {
    constexpr int LR = R == 4 ? 2 : 1;
    const int q = M >> LR;
    constexpr int PER_LANE = SLOTS / R / 64, INFL = 2;
#pragma unroll 1
    for (int h = 0; h < PER_LANE; h += INFL) {
        float2 u[INFL][R];
        int base[INFL], j[INFL];
#pragma unroll
        for (int t = 0; t < INFL; t++) {
            const int g = lane + 64 * (h + t);
            const int fi = g >> (logM - LR);
            j[t] = g & (q - 1);
            base[t] = (fi << logM) + j[t];
            int fr = frame0 + fi;
            fr = fr < rows_here ? fr : rows_here - 1;
            const float *sp = sig + fr * hop;
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int i = j[t] + m * q;
                const float2 w2 = wnd[i];
                u[t][m] = make_float2(sp[2 * i] * w2.x, sp[2 * i + 1] * w2.y);
            }
        }
#pragma unroll
        for (int t = 0; t < INFL; t++) {
            if constexpr (R == 4) {
                const float2 w1 = tw[j[t]], w2 = tw[q + j[t]], w3 = tw[2 * q + j[t]];
                bfly4(u[t][0], u[t][1], u[t][2], u[t][3]);
                x[phys(base[t])] = u[t][0];
                x[phys(base[t] + q)] = cmul(u[t][1], w1);
                x[phys(base[t] + 2 * q)] = cmul(u[t][2], w2);
                x[phys(base[t] + 3 * q)] = cmul(u[t][3], w3);
            } else {
                const float2 w1 = tw[j[t]];
                x[phys(base[t])] = make_float2(u[t][0].x + u[t][1].x, u[t][0].y + u[t][1].y);
                x[phys(base[t] + q)] = cmul(make_float2(u[t][0].x - u[t][1].x, u[t][0].y - u[t][1].y), w1);
            }
        }
    }
}

// the same for a frame of M = 5 q complex points (q a power of two: Perch's L = 640 bank is 5 x 64): radix 5 first, the rest of the
// transform then runs on power-of-two sub-problems of size q with the passes above.  F frames per wave pass, frame fi at slot fi M.
template <int SLOTS>
__device__ __forceinline__ void first_pass5(float2 *__restrict__ x, const float2 *__restrict__ tw, const float *__restrict__ sig,
                                            const float2 *__restrict__ wnd, int M, int logq, int F, int hop, int frame0, int rows_here, int lane) {
    const int q = 1 << logq, total = F << logq;
#pragma unroll 1
    for (int g = lane; g < total; g += 64) {
        const int fi = g >> logq, j = g & (q - 1);
        int fr = frame0 + fi;
        fr = fr < rows_here ? fr : rows_here - 1;
        const float *sp = sig + fr * hop;
        float2 u[5];
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int i = j + m * q;
            const float2 w2 = wnd[i];
            u[m] = make_float2(sp[2 * i] * w2.x, sp[2 * i + 1] * w2.y);
        }
        const float2 w1 = tw[j], w2 = tw[q + j], w3 = tw[2 * q + j], w4 = tw[3 * q + j];
        bfly5(u[0], u[1], u[2], u[3], u[4]);
        const int base = fi * M + j;
        x[phys(base)] = u[0];
        x[phys(base + q)] = cmul(u[1], w1);
        x[phys(base + 2 * q)] = cmul(u[2], w2);
        x[phys(base + 3 * q)] = cmul(u[3], w3);
        x[phys(base + 4 * q)] = cmul(u[4], w4);
    }
}

// Persistent blocks: a block walks the (sample, frame tile) work items blockIdx.x, + gridDim.x, ...; the tables are
// copied to LDS once, and the NEXT tile's signal span is fetched into registers while the current tile is transformed.
// PRE: the launch carries an absorbed prologue chain (false: none of its state exists -- the kernel keeps far more launch-uniform
// values than there are scalar registers, and every one it does not need is one fewer reloaded from a spill lane in the hot loops)
// MELM: the absorbed mel bank -- 0 none, 1 the sparse walk, 2 16 x 16 tiles on the matrix cores, -1 decided at run time (the instances with
// a prologue chain, which are big as it is)
template <int NW, int SLOTS, bool PRE, int MELM>
__global__ __launch_bounds__(NW * 64) void stft_kernel(FftDesc d, StftPtrs p, int total_tiles, int tiles_per_sample) {
    constexpr int SPAN_R = SPAN_FLOATS / (4 * NW * 64);
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const StftLds lay = stft_layout(d, NW, SLOTS);
    float *sig = lds + lay.sig;
    float2 *tw = reinterpret_cast<float2 *>(lds + lay.tw);
    float2 *wbuf = reinterpret_cast<float2 *>(lds + lay.wbuf) + wave * stft_wbuf_slots(SLOTS);
    float2 *wnd = reinterpret_cast<float2 *>(lds + lay.window);
    float *otab = lds + lay.otab;
    float *mstart = lds + lay.mstart;
    float2 *ment = reinterpret_cast<float2 *>(lds + lay.ment);
    float *spec = lds + lay.spec;
    const int M = d.M, logM = d.logM, hop = d.hop, nout = d.nout, nmel = MELM == 0 ? 0 : d.nmel;
    const int F = SLOTS == 1024 ? d.F : (SLOTS >> d.logM);  // frames per wave pass (M = 5 q: three frames of 320 slots)
    const int dbg = d.dbg;

    // ---- tables -> LDS, once per block
    async_copy<NW>(lds + lay.tw, reinterpret_cast<const float *>(p.tw), 2 * d.tw_count, wave, lane);
    async_copy<NW>(lds + lay.window, p.window, d.L, wave, lane);
    const int power = d.power, planar = d.otab_planar;
    // row form: [nout][ostride]; plane form (the planner's default): [nout][4] at 0, [nout][2] at 4 nout, [nout][4] at 6 nout
    const int ostride = planar ? 4 : (d.otab_stride > 0 ? d.otab_stride : 8);
    const int o1 = planar ? 4 * nout : 4, os1 = planar ? 2 : ostride, o2 = planar ? 6 * nout : 8;
    async_copy<NW>(otab, p.otab, planar ? (power ? 10 : 6) * nout : ostride * nout, wave, lane);
    const int mel_mode = MELM < 0 ? d.mel_mode : (MELM == 2 ? 1 : 0), SS = d.spec_stride > 0 ? d.spec_stride : nout;
    if (nmel && mel_mode == 0) {
        async_copy<NW>(mstart, p.mstart, nmel + 1, wave, lane);
        async_copy<NW>(lds + lay.ment, p.mcol, 2 * d.mel_nnz, wave, lane);
    }
    if (nmel && SS > nout)  // the padding columns of the spectrum rows meet zero filter taps: they must hold numbers (written once)
        for (int i = tid; i < d.tpb * (SS - nout); i += NW * 64) spec[(i / (SS - nout)) * SS + nout + i % (SS - nout)] = 0.0f;

    // (literal indices into the descriptor arrays everywhere: a loop the compiler does not unroll would index the
    // kernel argument dynamically and move ALL of it into scratch memory)
    static_assert(ELT_MAX_STAGES == 4, "the stages below are spelled out");
    PreChain chain{PRE ? d.npre : 0, {d.pre_bin[0], d.pre_bin[1], d.pre_bin[2], d.pre_bin[3]}, {d.pre_act[0], d.pre_act[1], d.pre_act[2], d.pre_act[3]},
                   {0.f, 0.f, 0.f, 0.f}, {d.pre_p0[0], d.pre_p0[1], d.pre_p0[2], d.pre_p0[3]}, {d.pre_p1[0], d.pre_p1[1], d.pre_p1[2], d.pre_p1[3]}};
    const int64_t bb0 = d.pre_bb[0], bb1 = d.pre_bb[1], bb2 = d.pre_bb[2], bb3 = d.pre_bb[3];
    const float *pre0 = p.pre[0], *pre1 = p.pre[1], *pre2 = p.pre[2], *pre3 = p.pre[3];
    const bool use0 = PRE && 0 < chain.n && chain.bin[0] != BIN_NONE, use1 = PRE && 1 < chain.n && chain.bin[1] != BIN_NONE;
    const bool use2 = PRE && 2 < chain.n && chain.bin[2] != BIN_NONE, use3 = PRE && 3 < chain.n && chain.bin[3] != BIN_NONE;
    const int npost = d.npost;
    const int po_a0 = d.post_act[0], po_a1 = d.post_act[1], po_a2 = d.post_act[2], po_a3 = d.post_act[3];
    const float po_p00 = d.post_p0[0], po_p01 = d.post_p0[1], po_p02 = d.post_p0[2], po_p03 = d.post_p0[3];
    const float po_p10 = d.post_p1[0], po_p11 = d.post_p1[1], po_p12 = d.post_p1[2], po_p13 = d.post_p1[3];
    // log2 of the sub-problem sizes of the strided passes after the first (each a quarter of its predecessor)
    const int ln1 = 31 - __builtin_clz((unsigned)max(d.pass_n[1], 1)), ln2 = ln1 - 2, ln3 = ln1 - 4;  // (pass_n[1] = M / first radix)
    const int pt0 = d.pass_tw[0], pt1 = d.pass_tw[1], pt2 = d.pass_tw[2], pt3 = d.pass_tw[3];
    const int npass = d.npass, first_radix = d.pass_r[0];

    // the span of a tile: up to SPAN_R float4 per thread, clamped loads, written to LDS (through the chain) later.
    // (macros, not lambdas: a by-reference closure would put the staged registers into scratch memory)
    float4 sr[SPAN_R];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int tpb = d.tpb, frames = d.frames, L = d.L, a_vec4 = d.a_vec4, pad_l = d.pad_l, in_len = d.in_len;
    const int64_t a_bs = d.a_bs;
#define BN_TILE_GEOM(TILE)                                              \
    const int64_t b = (TILE) / tiles_per_sample;                        \
    const int t0 = ((TILE) - (int)b * tiles_per_sample) * tpb;          \
    const int rows_here = min(tpb, frames - t0);                        \
    const int count = (rows_here - 1) * hop + L;
#define BN_ISSUE_SPAN(TILE)                                                                                              \
    do {                                                                                                                 \
        BN_TILE_GEOM(TILE)                                                                                               \
        const float *src = p.in + b * a_bs + (int64_t)t0 * hop - pad_l;                                                  \
        const int n4 = (count + 3) >> 2;                                                                                 \
        /* zero-padded signal (planner: absorb_pad_into_fft): a tile that touches the padding loads element by element */ \
        const int lo_ = t0 * hop - pad_l;                                                                                \
        const bool edge_ = in_len > 0 && (lo_ < 0 || lo_ + 4 * n4 > in_len);                                             \
        _Pragma("unroll") for (int k = 0; k < SPAN_R; k++) {                                                             \
            int c = tid + k * NW * 64;                                                                                   \
            c = c < n4 ? c : n4 - 1;                                                                                     \
            if (dbg & 32) {                                                                                              \
                sr[k] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
            } else if (edge_) {                                                                                          \
                const int e = lo_ + 4 * c;                                                                               \
                const float *sx = p.in + b * a_bs;                                                                       \
                float v_[4];                                                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                          \
                    const int r_ = e + j;                                                                                \
                    const float x_ = sx[min(max(r_, 0), in_len - 1)];                                                    \
                    v_[j] = (r_ >= 0 && r_ < in_len) ? x_ : 0.0f;                                                        \
                }                                                                                                        \
                sr[k] = make_float4(v_[0], v_[1], v_[2], v_[3]);                                                         \
            } else if (a_vec4) {                                                                                         \
                sr[k] = reinterpret_cast<const float4 *>(src)[c];                                                        \
            } else { /* unaligned rows: element loads, the last chunk clamped element by element */                     \
                const int e = 4 * c;                                                                                     \
                sr[k] = make_float4(src[e], src[min(e + 1, count - 1)], src[min(e + 2, count - 1)], src[min(e + 3, count - 1)]); \
            }                                                                                                            \
        }                                                                                                                \
        if (use0) s0 = pre0[b * bb0];                                                                                    \
        if (use1) s1 = pre1[b * bb1];                                                                                    \
        if (use2) s2 = pre2[b * bb2];                                                                                    \
        if (use3) s3 = pre3[b * bb3];                                                                                    \
    } while (0)
#define BN_WRITE_SPAN(TILE)                                                                       \
    do {                                                                                          \
        BN_TILE_GEOM(TILE)                                                                        \
        (void)b;                                                                                  \
        const int n4 = (count + 3) >> 2;                                                          \
        chain.sc[0] = s0; chain.sc[1] = s1; chain.sc[2] = s2; chain.sc[3] = s3;                   \
        float v[4 * SPAN_R];                                                                      \
        _Pragma("unroll") for (int k = 0; k < SPAN_R; k++) {                                      \
            v[4 * k] = sr[k].x; v[4 * k + 1] = sr[k].y; v[4 * k + 2] = sr[k].z; v[4 * k + 3] = sr[k].w; \
        }                                                                                         \
        if (PRE && !(dbg & 16)) pre_chain<4 * SPAN_R>(chain, v);                                  \
        _Pragma("unroll") for (int k = 0; k < SPAN_R; k++) {                                      \
            const int c = tid + k * NW * 64;                                                      \
            if (c < n4) reinterpret_cast<float4 *>(sig)[c] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]); \
        }                                                                                         \
    } while (0)

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    BN_ISSUE_SPAN(tile);
    BN_WRITE_SPAN(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the table copies
    __syncthreads();
    if (dbg & 8) return;

    // (the wave <-> band tile assignment is the same for every frame tile: filter tiles, group indices and bias are loaded ONCE per
    // block and stay in registers -- per tile they were three dependent L2 round trips behind the barrier, 5 - 6 us of a 16 us tile)
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    constexpr int MPRE = 8;
    const bool mel_mfma = nmel && mel_mode == 1 && !(dbg & 4);
    const int ntile = (nmel + 15) >> 4;
    const float *mt = p.mstart;
    const float4 *mp = reinterpret_cast<const float4 *>(p.mcol);
    float4 apre[MPRE];
    int gpre[MPRE];
    int pg0 = 0, pg1 = 0;
    float bpre[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // mel bias of this lane's four bands
    if (mel_mfma && wave < ntile) {
        const int tl = __builtin_amdgcn_readfirstlane(wave);
        pg0 = (int)mt[tl];
        pg1 = (int)mt[tl + 1];
#pragma unroll
        for (int k = 0; k < MPRE; k++) {
            const int gg = min(pg0 + k, pg1 - 1);
            gpre[k] = (int)mt[ntile + 1 + gg];
            apre[k] = mp[(int64_t)gg * 64 + lane];
        }
        if (d.mel_has_bias) {
#pragma unroll
            for (int j = 0; j < 4; j++) bpre[j] = p.mel_bias[min(16 * tl + 4 * (lane >> 4) + j, nmel - 1)];
        }
    }
    const int fstride = M + (M >> 3);
    for (; tile < total_tiles;) {
        const int next = tile + gridDim.x;
        if (next < total_tiles) BN_ISSUE_SPAN(next);
        BN_TILE_GEOM(tile)
        (void)count;
        const int groups = (rows_here + F - 1) / F;
        for (int g = wave; g < groups; g += NW) {
            if (!(dbg & 1)) {
                if (first_radix == 5) first_pass5<SLOTS>(wbuf, tw + pt0, sig, wnd, M, 31 - __builtin_clz((unsigned)(M / 5)), F, hop, g * F, rows_here, lane);
                else if (first_radix == 2) first_pass<2, SLOTS>(wbuf, tw + pt0, sig, wnd, M, logM, hop, g * F, rows_here, lane);
                else first_pass<4, SLOTS>(wbuf, tw + pt0, sig, wnd, M, logM, hop, g * F, rows_here, lane);
                wave_sync();
                if (1 < npass) { strided_pass<4, SLOTS>(wbuf, tw + pt1, ln1, lane); wave_sync(); }
                if (2 < npass) { strided_pass<4, SLOTS>(wbuf, tw + pt2, ln2, lane); wave_sync(); }
                if (3 < npass) { strided_pass<4, SLOTS>(wbuf, tw + pt3, ln3, lane); wave_sync(); }
                if (SLOTS == 1024 || lane < SLOTS / 16) block16(wbuf + 18 * lane);
                wave_sync();
            }
            if (dbg & 2) continue;
            // the kept bins: out[c] = alpha Re Z[k] + beta Im Z[k] + gamma Re Z[M-k] + delta Im Z[M-k]
            // (four bins per lane in flight: the table entry, then the two transform slots it names, are dependent LDS reads
            // -- one bin at a time this loop was five round trips of ~250 cycles per frame, 8 of the kernel's 67 us)
            for (int rep = 0; rep < ((dbg & 64) ? 4 : 1); rep++)
            for (int fi = 0; fi < F; fi++) {
                const int t = g * F + fi;
                const float2 *zf = wbuf + fi * fstride;
                constexpr int BI = SLOTS == 1024 ? 4 : 2;  // bins per lane in flight (twice the waves hold half as many each)
                for (int c0 = lane; c0 < nout; c0 += 64 * BI) {
                    float4 e0[BI];
                    float2 e1[BI], za[BI], zb[BI];
#pragma unroll
                    for (int k = 0; k < BI; k++) {
                        const int c = min(c0 + 64 * k, nout - 1);
                        e0[k] = *reinterpret_cast<const float4 *>(otab + ostride * c);
                        e1[k] = *reinterpret_cast<const float2 *>(otab + o1 + os1 * c);
                    }
#pragma unroll
                    for (int k = 0; k < BI; k++) {
                        za[k] = zf[(int)e0[k].x];
                        zb[k] = zf[(int)e0[k].y];
                    }
#pragma unroll
                    for (int k = 0; k < BI; k++) {
                        const int c = c0 + 64 * k;
                        if (c < nout) {
                            float v = e0[k].z * za[k].x + e0[k].w * za[k].y + e1[k].x * zb[k].x + e1[k].y * zb[k].y;
                            if (power) {  // launch-uniform: the second linear form of the bin, then u^2 + v^2 (+ sqrt)
                                const float4 e2 = *reinterpret_cast<const float4 *>(otab + o2 + ostride * c);
                                const float w = e2.x * za[k].x + e2.y * za[k].y + e2.z * zb[k].x + e2.w * zb[k].y;
                                const float uu = v * v, ww = w * w;
                                v = uu + ww;
                                if (power == 2) v = sqrtf(v);
                            }
                            if (d.has_bias) v += p.bias[c];
                            if (nmel) spec[t * SS + c] = v;
                            else if (t < rows_here) p.out[b * d.c_bs + (int64_t)(t0 + t) * d.ldc + c] = v;
                        }
                    }
                }
            }
            wave_sync();  // the transform buffer is rewritten by this wave's next group
        }
        // Mel filter bank on the matrix cores (mel_mode 1): D[band][frame] = sum over the kept 16-bin tiles of the band tile's
        // row, A = a filter tile in fragment order straight from global memory (one coalesced dwordx4 per lane), B = the
        // spectrum rows in LDS (k-slot j of a bin group: bin 16 g + 4 q + j, so one ds_read_b128 per lane feeds four matrix
        // instructions).  A wave owns a tile of 16 bands x the tile's 16 frames; a lane ends up with 4 bands of one frame and
        // runs the compression chain on those four values ONCE.  (The sparse (column, weight) walk it replaces was a chain of
        // dependent LDS reads per (band, frame): 19 of the kernel's 58 us for 10 000 multiply-adds per tile.)  The filter
        __syncthreads();  // every wave is done with the span; the tile's spectrum rows are complete
        if (mel_mfma) {
            const int ln = lane & 15, lq = lane >> 4;
            float *ob = p.out + b * d.c_bs;
            const float *srow = spec + ln * SS + 4 * lq;
            for (int tl0 = wave; tl0 < ntile; tl0 += NW) {
                const int tl = __builtin_amdgcn_readfirstlane(tl0);
                const int g0 = (int)mt[tl], g1 = (int)mt[tl + 1];
                floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                int gi = g0;
                if (tl0 == wave) {  // the requested tiles
                    float4 bv[MPRE];
#pragma unroll
                    for (int k = 0; k < MPRE; k++) bv[k] = *reinterpret_cast<const float4 *>(srow + 16 * gpre[k]);
#pragma unroll
                    for (int k = 0; k < MPRE; k++) {
                        if (pg0 + k < pg1) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(apre[k].x, bv[k].x, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(apre[k].y, bv[k].y, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(apre[k].z, bv[k].z, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(apre[k].w, bv[k].w, acc, 0, 0, 0);
                        }
                    }
                    gi = g0 + MPRE;
                }
                for (; gi < g1; gi += 4) {
                    float4 av[4], bv[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int gg = min(gi + k, g1 - 1);
                        const int g = (int)mt[ntile + 1 + gg];
                        av[k] = mp[(int64_t)gg * 64 + lane];
                        bv[k] = *reinterpret_cast<const float4 *>(srow + 16 * g);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (gi + k < g1) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k].x, bv[k].x, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k].y, bv[k].y, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k].z, bv[k].z, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k].w, bv[k].w, acc, 0, 0, 0);
                        }
                    }
                }
                const int m0 = 16 * tl + 4 * lq;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = acc[j] + (tl0 == wave ? bpre[j] : ((d.mel_has_bias && m0 + j < nmel) ? p.mel_bias[m0 + j] : 0.0f));
                if (!(dbg & 128)) {
                    Vals4 w4 = {{v[0], v[1], v[2], v[3]}};
                    if (d.mel_act != ACT_NONE) w4 = act_small4(d.mel_act, d.mel_p0, d.mel_p1, w4);
                    if (0 < npost) w4 = act_small4(po_a0, po_p00, po_p10, w4);
                    if (1 < npost) w4 = act_small4(po_a1, po_p01, po_p11, w4);
                    if (2 < npost) w4 = act_small4(po_a2, po_p02, po_p12, w4);
                    if (3 < npost) w4 = act_small4(po_a3, po_p03, po_p13, w4);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = w4.x[j];
                }
                if (ln < rows_here && !(dbg & 256)) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (m0 + j < nmel) ob[(int64_t)(t0 + ln) * d.out_rs + (int64_t)(m0 + j) * d.out_cs] = v[j];
                }
            }
        }
        if (nmel && mel_mode == 0 && !(dbg & 4)) {
            // Mel filter bank over the whole tile: one work item per (band, frame), frames fastest -- the 16 lanes of a band
            // read the same (column, weight) entry (an LDS broadcast) and 16 different spectrum rows (row stride nout is
            // odd: conflict-free), sum the band's entries in index order, run the compression chain on the one value and
            // store it; the 16 frames of a band are neighbours in the target view.  (The first version gave four threads a
            // band and 16 frames each: 650 instructions per thread, a quarter of the lanes idle, as long as the transform.)
            const int ltpb = 31 - __builtin_clz((unsigned)tpb);  // frames per tile: a power of two (16, or 8 where the LDS is short)
            float *ob = p.out + b * d.c_bs;
            for (int i0 = 0; i0 < (nmel << ltpb); i0 += NW * 64) {
                const int i = i0 + tid;
                const int m = i >> ltpb, t = i & (tpb - 1);
                const bool live = m < nmel && t < rows_here;
                const int e0 = live ? (int)mstart[m] : 0, e1 = live ? (int)mstart[m + 1] : 0;
                const float *sp = spec + (live ? t : 0) * SS;
                // Eight entries of the band in flight: an entry (column, weight) and the spectrum value it names are two
                // dependent LDS reads, and a band has up to ~30 entries -- two at a time the phase was a chain of ~15 round
                // trips per work item, 30 of the kernel's 67 us for 10 000 multiply-adds per tile.  Same sums as before:
                // entries at even offsets from the band's first feed a0, odd ones a1, combined once at the end.
                float a0 = 0.0f, a1 = 0.0f;
                constexpr int MI = SLOTS == 1024 ? 8 : 4;  // entries per lane in flight (even: the a0 / a1 assignment stays)
                for (int e = e0; e < ((dbg & 512) ? e0 : e1); e += MI) {
                    float2 cc[MI];
                    float sv[MI];
#pragma unroll
                    for (int k = 0; k < MI; k++) cc[k] = ment[e + k < e1 ? e + k : e0];
#pragma unroll
                    for (int k = 0; k < MI; k++) sv[k] = sp[(int)cc[k].x];
#pragma unroll
                    for (int k = 0; k < MI; k += 2) {
                        if (e + k < e1) a0 = fmaf(sv[k], cc[k].y, a0);
                        if (e + k + 1 < e1) a1 = fmaf(sv[k + 1], cc[k + 1].y, a1);
                    }
                }
                float acc[1] = {(a0 + a1) + ((live && d.mel_has_bias) ? p.mel_bias[m] : 0.0f)};
                if (!(dbg & 128)) {
                act_small<1>(d.mel_act, d.mel_p0, d.mel_p1, acc);
                if (0 < npost) act_small<1>(po_a0, po_p00, po_p10, acc);
                if (1 < npost) act_small<1>(po_a1, po_p01, po_p11, acc);
                if (2 < npost) act_small<1>(po_a2, po_p02, po_p12, acc);
                if (3 < npost) act_small<1>(po_a3, po_p03, po_p13, acc);
                }
                if (live && !(dbg & 256)) ob[(int64_t)(t0 + t) * d.out_rs + (int64_t)m * d.out_cs] = acc[0];
            }
        }
        if (next < total_tiles) BN_WRITE_SPAN(next);
        // next span in place, spectrum rows read: an LDS-only rendezvous (__syncthreads would also wait for the tile's result
        // stores to complete -- a full write round trip per tile, 4.4 of the kernel's 58 us)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tile = next;
    }
#undef BN_TILE_GEOM
#undef BN_ISSUE_SPAN
#undef BN_WRITE_SPAN
}


}  // namespace

void register_stft_kernels() {
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(stft_kernel<8, 1024, false, 0>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(stft_kernel<8, 1024, false, 1>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(stft_kernel<8, 1024, false, 2>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(stft_kernel<8, 1024, true, -1>));
    register_dynamic_lds_kernel(reinterpret_cast<const void *>(stft_kernel<16, 512, true, -1>));
}

void launch_stft(hipStream_t s, const FftDesc &d, const StftPtrs &p, int64_t batch) {
    if (batch <= 0) return;
    const bool wide = stft_wide(d);
    const size_t lds = stft_lds_bytes(d, 0);
    const int span = (d.tpb - 1) * d.hop + d.L;
    const bool pre = d.npre > 0;
    const int melm = d.nmel == 0 ? 0 : (d.mel_mode == 1 ? 2 : 1);
    const void *fn = wide  ? reinterpret_cast<const void *>(stft_kernel<16, 512, true, -1>)
                     : pre ? reinterpret_cast<const void *>(stft_kernel<8, 1024, true, -1>)
                     : melm == 0 ? reinterpret_cast<const void *>(stft_kernel<8, 1024, false, 0>)
                     : melm == 1 ? reinterpret_cast<const void *>(stft_kernel<8, 1024, false, 1>)
                                 : reinterpret_cast<const void *>(stft_kernel<8, 1024, false, 2>);
    if (lds > 160 * 1024 || span > SPAN_FLOATS || !ensure_dynamic_lds(fn, lds)) {
        launch_error("STFT kernel: the frame span does not fit the LDS / the staging registers");
        return;
    }
    FftDesc dd = d;
    dd.dbg = getenv("BN_STFT_DBG") ? atoi(getenv("BN_STFT_DBG")) : 0;
    // float4 span loads need 16-byte aligned tile starts (base pointer, batch stride, a tile's first sample) and must
    // not run past the sample row: the last chunk of a span is rounded up to 4 floats
    dd.a_vec4 = (reinterpret_cast<uintptr_t>(p.in) & 15u) == 0 && d.a_bs % 4 == 0 && ((int64_t)d.tpb * d.hop) % 4 == 0 &&
                (d.in_len > 0 ? d.pad_l % 4 == 0  // (padded: interior tiles lie inside the sample row by construction, edge tiles load by element)
                              : (int64_t)(d.frames - 1) * d.hop + d.L + 3 <= d.a_bs);
    const int tps = (d.frames + d.tpb - 1) / d.tpb;
    const int64_t total = (int64_t)tps * batch;
    const int ncu = device_cu_count();  // asked once per device by prepare_device(), never inside a stream capture
    const unsigned grid = (unsigned)std::min<int64_t>(total, ncu);  // one block per CU (LDS-bound), persistent over its tiles
    if (wide) hipLaunchKernelGGL((stft_kernel<16, 512, true, -1>), dim3(grid), dim3(16 * 64), lds, s, dd, p, (int)total, tps);
    else if (pre) hipLaunchKernelGGL((stft_kernel<8, 1024, true, -1>), dim3(grid), dim3(8 * 64), lds, s, dd, p, (int)total, tps);
    else if (melm == 0) hipLaunchKernelGGL((stft_kernel<8, 1024, false, 0>), dim3(grid), dim3(8 * 64), lds, s, dd, p, (int)total, tps);
    else if (melm == 1) hipLaunchKernelGGL((stft_kernel<8, 1024, false, 1>), dim3(grid), dim3(8 * 64), lds, s, dd, p, (int)total, tps);
    else hipLaunchKernelGGL((stft_kernel<8, 1024, false, 2>), dim3(grid), dim3(8 * 64), lds, s, dd, p, (int)total, tps);
}

}  // namespace bn
