// f32 as three exact bf16 terms, device side (gemm_dma3.hip, gemm_b3.hip): x = hi + mid + lo with hi = x's top 8 significand bits (the f32
// with its low 16 bits cleared), mid the same of the remainder x - hi (exact), lo = x - hi - mid (at most 8 significant bits are left: a
// bf16 number).  The six partial products kept by the kernels (everything but mid lo, lo mid, lo lo) are each exact in f32; the dropped
// ones sum to less than 2^-21 |x w| in the worst case, 2^-24 |x w| in the root mean square (|mid| < 2^-7 |x|, |lo| < 2^-15 |x|).  Host side and the weight packers: plan_rules.h (split_bf16x3, pack_w3, pack_w3f).
#pragma once
#include <hip/hip_runtime.h>

namespace bn {
namespace {

typedef float b3_floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int b3_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));

// eight f32 values (a = elements 0..3, b = 4..7) -> three vectors of eight bf16, element e in bits [16 e, 16 e + 15]
__device__ __forceinline__ void split3(const b3_floatx4 &a, const b3_floatx4 &b, b3_u32x4 &hi, b3_u32x4 &mid, b3_u32x4 &lo) {
    float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r1[j] = x[j] - __uint_as_float(__float_as_uint(x[j]) & 0xffff0000u);
        r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xffff0000u);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {  // v_perm_b32: the top halves of two f32 side by side
        hi[p] = __builtin_amdgcn_perm(__float_as_uint(x[2 * p + 1]), __float_as_uint(x[2 * p]), 0x07060302u);
        mid[p] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * p + 1]), __float_as_uint(r1[2 * p]), 0x07060302u);
        lo[p] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * p + 1]), __float_as_uint(r2[2 * p]), 0x07060302u);
    }
}

__device__ __forceinline__ b3_floatx4 mm_bf16(const b3_u32x4 &w, const b3_u32x4 &x, const b3_floatx4 &acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b3_bf16x8, w), __builtin_bit_cast(b3_bf16x8, x), acc, 0, 0, 0);
}
typedef float b3_floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ b3_floatx16 mm32_bf16(const b3_u32x4 &a, const b3_u32x4 &b, const b3_floatx16 &acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b3_bf16x8, a), __builtin_bit_cast(b3_bf16x8, b), acc, 0, 0, 0);
}
// the six partial products of one 16 x 16 tile and 32-deep k, smallest terms first (fixed order: part of every output's arithmetic)
__device__ __forceinline__ b3_floatx4 mm6(const b3_u32x4 &wh, const b3_u32x4 &wm, const b3_u32x4 &wl, const b3_u32x4 &xh, const b3_u32x4 &xm,
                                          const b3_u32x4 &xl, b3_floatx4 a) {
    a = mm_bf16(wl, xh, a);
    a = mm_bf16(wh, xl, a);
    a = mm_bf16(wm, xm, a);
    a = mm_bf16(wm, xh, a);
    a = mm_bf16(wh, xm, a);
    a = mm_bf16(wh, xh, a);
    return a;
}

}  // namespace
}  // namespace bn
