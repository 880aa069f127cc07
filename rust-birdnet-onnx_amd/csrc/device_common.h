// Device helpers shared by the kernel translation units (kernels.hip, stft.hip): the logistic / exp / log / pow
// forms of the network path and the one-dispatch-per-stage unary / binary stage functions.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels.h"

namespace bn {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Logistic function inside the network (SE gates, SiLU): hardware exp2 and reciprocal
// (v_exp_f32 / v_rcp_f32, ~1 ulp each; the argument scaling adds |x| * 2^-24 relative), 6
// instructions instead of ~30 for the IEEE expf + division.  Saturates correctly: x -> -inf gives
// rcp(inf) = 0, x -> +inf gives rcp(1) = 1.  The CONFIDENCE sigmoid of the post-processing
// (topk.hip) does not use this: it is bit-exact against the reference's f32 sigmoid.
__device__ __forceinline__ float net_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
// exp / log / pow of the front end's dynamic-range compression (power spectrum -> x^p, log-mel):
// hardware exp2 / log2 (v_exp_f32 / v_log_f32, ~1 ulp each).  pow(x, p) = exp2(p * log2 x) for
// x > 0 has a relative error of about |p * log2 x| * 2^-23 (<= 1e-5 over the 1e-30..1e30 range a
// spectrogram can span), against ~150 instructions for the correctly rounded powf -- the two Pow
// chains of the v2.4 front end were VALU-bound on it.  Non-positive bases keep the libm path
// (signs, zeros, NaN rules).
__device__ __forceinline__ float net_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float net_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309417f; }
__device__ __forceinline__ float net_pow(float x, float p) {
    return x > 0.0f ? __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)) : powf(x, p);
}

template <int N, class F>
__device__ __forceinline__ void map_array(float (&v)[N], F f) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = f(v[i]);
}

__device__ __forceinline__ float act_apply(int act, float x, float p0, float p1) {
    switch (act) {
        case ACT_NONE: return x;
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_CLIP: return fminf(fmaxf(x, p0), p1);
        case ACT_SIGMOID: return net_sigmoid(x);
        case ACT_SILU: return x * net_sigmoid(x);
        case ACT_HSIGMOID: return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f);
        case ACT_HSWISH: return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
        case ACT_LEAKY: return x >= 0.0f ? x : p0 * x;
        case ACT_TANH: return tanhf(x);
        case ACT_EXP: return net_exp(x);
        case ACT_LOG: return net_log(x);
        case ACT_SQRT: return sqrtf(x);
        case ACT_ABS: return fabsf(x);
        case ACT_NEG: return -x;
        case ACT_RECIP: return 1.0f / x;
        case ACT_POW: return net_pow(x, p0);
        case ACT_AFFINE: return p0 * x + p1;
        case ACT_MAXC: return fmaxf(x, p0);
        case ACT_MINC: return fminf(x, p0);
        case ACT_RSUB: return p0 - x;
        case ACT_RDIV: return p0 / x;
        case ACT_SQUARE: return x * x;
        case ACT_FLOOR: return floorf(x);
        case ACT_CEIL: return ceilf(x);
        case ACT_ERF: return erff(x);
        case ACT_SOFTPLUS: return log1pf(expf(x));
        case ACT_GTC: return x > p0 ? 1.0f : 0.0f;
        case ACT_LTC: return x < p0 ? 1.0f : 0.0f;
        case ACT_GEC: return x >= p0 ? 1.0f : 0.0f;
        case ACT_LEC: return x <= p0 ? 1.0f : 0.0f;
        case ACT_EQC: return x == p0 ? 1.0f : 0.0f;
        case ACT_NEZ: return x != 0.0f ? 1.0f : 0.0f;
        case ACT_TRUNC: return truncf(x);
        case ACT_ROUND: return rintf(x);
        default: return x;
    }
}

// Stage ops over a small register array with ONE dispatch on the (launch-uniform) op code: the
// per-element switch of act_apply / bin_apply costs a branch tree per element per stage.
template <int N>
__device__ __forceinline__ void act_array_all(int act, float p0, float p1, float (&v)[N]) {
    switch (act) {
        case ACT_NONE: return;
        case ACT_RELU: map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); }); return;
        case ACT_CLIP: map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); }); return;
        case ACT_SIGMOID: map_array<N>(v, [](float x) { return net_sigmoid(x); }); return;
        case ACT_SILU: map_array<N>(v, [](float x) { return x * net_sigmoid(x); }); return;
        case ACT_HSIGMOID: map_array<N>(v, [=](float x) { return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f); }); return;
        case ACT_HSWISH: map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); }); return;
        case ACT_LEAKY: map_array<N>(v, [=](float x) { return x >= 0.0f ? x : p0 * x; }); return;
        case ACT_TANH: map_array<N>(v, [](float x) { return tanhf(x); }); return;
        case ACT_EXP: map_array<N>(v, [](float x) { return net_exp(x); }); return;
        case ACT_LOG: map_array<N>(v, [](float x) { return net_log(x); }); return;
        case ACT_SQRT: map_array<N>(v, [](float x) { return sqrtf(x); }); return;
        case ACT_ABS: map_array<N>(v, [](float x) { return fabsf(x); }); return;
        case ACT_NEG: map_array<N>(v, [](float x) { return -x; }); return;
        case ACT_RECIP: map_array<N>(v, [](float x) { return 1.0f / x; }); return;
        case ACT_POW: map_array<N>(v, [=](float x) { return net_pow(x, p0); }); return;
        case ACT_AFFINE: map_array<N>(v, [=](float x) { return p0 * x + p1; }); return;
        case ACT_MAXC: map_array<N>(v, [=](float x) { return fmaxf(x, p0); }); return;
        case ACT_MINC: map_array<N>(v, [=](float x) { return fminf(x, p0); }); return;
        case ACT_RSUB: map_array<N>(v, [=](float x) { return p0 - x; }); return;
        case ACT_RDIV: map_array<N>(v, [=](float x) { return p0 / x; }); return;
        case ACT_SQUARE: map_array<N>(v, [](float x) { return x * x; }); return;
        case ACT_FLOOR: map_array<N>(v, [](float x) { return floorf(x); }); return;
        case ACT_CEIL: map_array<N>(v, [](float x) { return ceilf(x); }); return;
        case ACT_ERF: map_array<N>(v, [](float x) { return erff(x); }); return;
        case ACT_SOFTPLUS: map_array<N>(v, [](float x) { return log1pf(expf(x)); }); return;
        case ACT_GTC: map_array<N>(v, [=](float x) { return x > p0 ? 1.0f : 0.0f; }); return;
        case ACT_LTC: map_array<N>(v, [=](float x) { return x < p0 ? 1.0f : 0.0f; }); return;
        case ACT_GEC: map_array<N>(v, [=](float x) { return x >= p0 ? 1.0f : 0.0f; }); return;
        case ACT_LEC: map_array<N>(v, [=](float x) { return x <= p0 ? 1.0f : 0.0f; }); return;
        case ACT_EQC: map_array<N>(v, [=](float x) { return x == p0 ? 1.0f : 0.0f; }); return;
        case ACT_NEZ: map_array<N>(v, [](float x) { return x != 0.0f ? 1.0f : 0.0f; }); return;
        case ACT_TRUNC: map_array<N>(v, [](float x) { return truncf(x); }); return;
        case ACT_ROUND: map_array<N>(v, [](float x) { return rintf(x); }); return;
        default: return;
    }
}
template <int N, class F>
__device__ __forceinline__ void zip_array(float (&v)[N], const float (&w)[N], F f) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = f(v[i], w[i]);
}
template <int N>
__device__ __forceinline__ void bin_array(int bin, int bsq, float (&v)[N], float (&w)[N]) {
    if (bsq) map_array<N>(w, [](float x) { return x * x; });
    switch (bin) {
        case BIN_ADD: zip_array<N>(v, w, [](float a, float b) { return a + b; }); return;
        case BIN_SUB: zip_array<N>(v, w, [](float a, float b) { return a - b; }); return;
        case BIN_MUL: zip_array<N>(v, w, [](float a, float b) { return a * b; }); return;
        case BIN_DIV: zip_array<N>(v, w, [](float a, float b) { return a / b; }); return;
        case BIN_POW: zip_array<N>(v, w, [](float a, float b) { return net_pow(a, b); }); return;
        case BIN_MAX: zip_array<N>(v, w, [](float a, float b) { return fmaxf(a, b); }); return;
        case BIN_MIN: zip_array<N>(v, w, [](float a, float b) { return fminf(a, b); }); return;
        case BIN_GT: zip_array<N>(v, w, [](float a, float b) { return a > b ? 1.0f : 0.0f; }); return;
        case BIN_LT: zip_array<N>(v, w, [](float a, float b) { return a < b ? 1.0f : 0.0f; }); return;
        case BIN_GE: zip_array<N>(v, w, [](float a, float b) { return a >= b ? 1.0f : 0.0f; }); return;
        case BIN_LE: zip_array<N>(v, w, [](float a, float b) { return a <= b ? 1.0f : 0.0f; }); return;
        case BIN_EQ: zip_array<N>(v, w, [](float a, float b) { return a == b ? 1.0f : 0.0f; }); return;
        case BIN_NE: zip_array<N>(v, w, [](float a, float b) { return a != b ? 1.0f : 0.0f; }); return;
        case BIN_SELA: zip_array<N>(v, w, [](float a, float b) { return b != 0.0f ? a : 0.0f; }); return;
        case BIN_SELB: zip_array<N>(v, w, [](float a, float b) { return b != 0.0f ? 0.0f : a; }); return;
        default: return;
    }
}

}  // namespace
// Stage functions of the absorbed chains, COMPACT on purpose: the generic act_array_all / bin_array dispatchers inline
// libm (tanhf, erff, powf, ...) for every instantiation -- with nine stage slots that made this kernel 676 KB of code,
// and walking through it ran at instruction-fetch speed (a 4-stage chain over 16 floats per thread cost 19 us per
// tile).  Only codes with a few-instruction body are accepted here (kernels.h, stft_stage_supported: the planner
// absorbs nothing else), dispatched by compare chains on the launch-uniform code.
__device__ __forceinline__ float pow_compact(float x, float p) {
    // x^p without libm: exp2(p log2 |x|) (hardware exp2 / log2, ~1e-6 relative), the sign and the special cases by hand
    const float ax = fabsf(x);
    float r = __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(ax));
    if (x < 0.0f) {
        const float fl = floorf(p);
        if (fl != p) r = __builtin_nanf("");                 // negative base, non-integer exponent
        else if (fl * 0.5f != floorf(fl * 0.5f)) r = -r;     // odd integer exponent keeps the sign
    }
    if (p == 0.0f) r = 1.0f;
    return r;
}
// ONE dispatch per stage for a whole register array (the code is launch-uniform): a compare chain per element would
// be re-evaluated, operands reloaded, for every value.
template <int N>
__device__ __forceinline__ void act_small(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_NONE) return;
    if (act == ACT_AFFINE) map_array<N>(v, [=](float x) { return p0 * x + p1; });
    else if (act == ACT_SQUARE) map_array<N>(v, [](float x) { return x * x; });
    else if (act == ACT_POW) map_array<N>(v, [=](float x) { return pow_compact(x, p0); });
    else if (act == ACT_LOG) map_array<N>(v, [](float x) { return net_log(x); });
    else if (act == ACT_EXP) map_array<N>(v, [](float x) { return net_exp(x); });
    else if (act == ACT_SQRT) map_array<N>(v, [](float x) { return sqrtf(x); });
    else if (act == ACT_ABS) map_array<N>(v, [](float x) { return fabsf(x); });
    else if (act == ACT_MAXC) map_array<N>(v, [=](float x) { return fmaxf(x, p0); });
    else if (act == ACT_MINC) map_array<N>(v, [=](float x) { return fminf(x, p0); });
    else if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_NEG) map_array<N>(v, [](float x) { return -x; });
    else if (act == ACT_RECIP) map_array<N>(v, [](float x) { return 1.0f / x; });
    else if (act == ACT_RSUB) map_array<N>(v, [=](float x) { return p0 - x; });
    else if (act == ACT_RDIV) map_array<N>(v, [=](float x) { return p0 / x; });
    else if (act == ACT_SIGMOID) map_array<N>(v, [](float x) { return net_sigmoid(x); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_LEAKY) map_array<N>(v, [=](float x) { return x >= 0.0f ? x : p0 * x; });
    else if (act == ACT_FLOOR) map_array<N>(v, [](float x) { return floorf(x); });
    else if (act == ACT_CEIL) map_array<N>(v, [](float x) { return ceilf(x); });
}

// binary stage against ONE scalar for the whole array (the absorbed per-sample chains: stft.hip, the framing GEMMs of kernels.hip)
template <int N>
__device__ __forceinline__ void bin_small(int bin, float b, float (&v)[N]) {
    if (bin == BIN_ADD) map_array<N>(v, [=](float a) { return a + b; });
    else if (bin == BIN_SUB) map_array<N>(v, [=](float a) { return a - b; });
    else if (bin == BIN_MUL) map_array<N>(v, [=](float a) { return a * b; });
    else if (bin == BIN_DIV) {
        // one divisor for the whole array: 1 / b once, then q = a r corrected by one residual step, which is what the
        // hardware's division sequence computes minus its scaling for denormal / overflowing quotients (3 instructions per
        // element instead of ~10; the absorbed chain divides the whole segment by max - min)
        const float r = 1.0f / b;
        map_array<N>(v, [=](float a) {
            const float q = a * r;
            return fmaf(fmaf(-q, b, a), r, q);
        });
    }
    else if (bin == BIN_MAX) map_array<N>(v, [=](float a) { return fmaxf(a, b); });
    else if (bin == BIN_MIN) map_array<N>(v, [=](float a) { return fminf(a, b); });
}

// The absorbed per-sample chain: up to four stages  v = act(bin(v, scalar)).  All indices are literals so that the
// fields stay in registers.
struct PreChain {
    int n, bin[4], act[4];
    float sc[4], p0[4], p1[4];
};
template <int S, int N>
__device__ __forceinline__ void pre_stage(const PreChain &c, float (&v)[N]) {
    if (S < c.n) {
        bin_small<N>(c.bin[S], c.sc[S], v);
        act_small<N>(c.act[S], c.p0[S], c.p1[S], v);
    }
}
template <int N>
__device__ __forceinline__ void pre_chain(const PreChain &c, float (&v)[N]) {
    pre_stage<0, N>(c, v);
    pre_stage<1, N>(c, v);
    pre_stage<2, N>(c, v);
    pre_stage<3, N>(c, v);
}

}  // namespace bn
