// Does an exact-f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4) share its issue / ALU time with f32 VALU work?
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_valu_probe tools/mfma_valu_probe.cpp && tools/mfma_valu_probe
// Four timed kernels on every CU (one 512-thread block per CU = 2 waves per SIMD, or 256 = 1 wave per SIMD):
//   A  every wave: NM matrix instructions                      (matrix pipe alone)
//   B  every wave: NV independent v_fma_f32                    (vector ALU alone)
//   C  every wave: both, interleaved in ONE instruction stream (in-order issue of one wave)
//   D  waves 0-3 run A's loop, waves 4-7 run B's loop          (two waves of one SIMD, one matrix, one vector)
// If the two units were independent, C and D would take max(A, B); if they are one resource, A + B.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// E  every wave: NV independent v_pk_fma_f32 (two f32 FMAs per lane per instruction) -- does the packed form really retire
//    twice the FMAs per issue slot, alone and beside exact-f32 MFMAs?   F = E's loop on waves 4-7, A's loop on waves 0-3.
template <int MODE>
__global__ __launch_bounds__(512) void probe_pk(float *out, int iters, int nv_per_m) {
    const int wave = threadIdx.x >> 6;
    floatx16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    floatx2 a2 = {a, a + 1}, b2 = {b, b + 1};
    floatx2 v0 = a2, v1 = b2, v2 = a2 + b2, v3 = a2 - b2, v4 = a2 * 2, v5 = b2 * 2, v6 = a2 + 1, v7 = b2 + 1;
    const bool do_m = MODE == 1 && wave < 4;
    const bool do_v = MODE == 0 || (MODE == 1 && wave >= 4);
    for (int i = 0; i < iters; i++) {
        if (do_m) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
        if (do_v) {
            for (int j = 0; j < nv_per_m; j++) {
                v0 = __builtin_elementwise_fma(v0, b2, a2); v1 = __builtin_elementwise_fma(v1, b2, a2);
                v2 = __builtin_elementwise_fma(v2, b2, a2); v3 = __builtin_elementwise_fma(v3, b2, a2);
                v4 = __builtin_elementwise_fma(v4, b2, a2); v5 = __builtin_elementwise_fma(v5, b2, a2);
                v6 = __builtin_elementwise_fma(v6, b2, a2); v7 = __builtin_elementwise_fma(v7, b2, a2);
            }
        }
    }
    floatx2 s2 = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    float s = s2.x + s2.y;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE>
float run_pk(int threads, int iters, int nv, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_pk<MODE>), dim3(256), dim3(threads), 0, 0, d, iters, nv);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_pk<MODE>), dim3(256), dim3(threads), 0, 0, d, iters, nv);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f;
}

template <int MODE, int SHAPE>
__global__ __launch_bounds__(512) void probe(float *out, int iters, int nv_per_m) {
    const int wave = threadIdx.x >> 6;
    floatx16 acc0 = {0}, acc1 = {0};
    floatx4 q0 = {0}, q1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * 2, v5 = b * 2, v6 = a + 1, v7 = b + 1;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int i = 0; i < iters; i++) {
        if (do_m) {
            if (SHAPE == 32) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            } else {
                q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, q1, 0, 0, 0);
                q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, q1, 0, 0, 0);
            }
        }
        if (do_v) {
            for (int j = 0; j < nv_per_m; j++) {
                v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a);
                v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a); v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
    for (int r = 0; r < 4; r++) s += q0[r] + q1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int SHAPE>
float run(int threads, int iters, int nv, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, SHAPE>), dim3(256), dim3(threads), 0, 0, d, iters, nv);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, SHAPE>), dim3(256), dim3(threads), 0, 0, d, iters, nv);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f;
}

int main() {
    float *d;
    hipMalloc(&d, 4096);
    const int iters = 20000;
    for (int nv : {1, 2, 4}) {
        // per iteration: 2 x 32x32x2 (128 cycles of matrix pipe) or 4 x 16x16x4 (128 cycles); 8*nv v_fma (4 cycles each alone)
        printf("nv_per_iter=%d (8*nv v_fma per 128 matrix cycles)\n", 8 * nv);
        for (int threads : {256, 512}) {
            float a32 = run<0, 32>(threads, iters, nv, d), b = run<1, 32>(threads, iters, nv, d), c32 = run<2, 32>(threads, iters, nv, d);
            float a16 = run<0, 16>(threads, iters, nv, d), c16 = run<2, 16>(threads, iters, nv, d);
            printf("  threads=%d  A(32x32x2)=%.1f us  A(16x16x4)=%.1f us  B(valu)=%.1f us  C32(same wave)=%.1f us  C16=%.1f us", threads, a32, a16, b, c32, c16);
            if (threads == 512) {
                float d32 = run<3, 32>(threads, iters, nv, d), d16 = run<3, 16>(threads, iters, nv, d);
                printf("  D32(split waves)=%.1f us  D16=%.1f us", d32, d16);
            }
            printf("\n");
        }
        printf("  packed: E(v_pk_fma_f32 alone) 256 thr=%.1f us  512 thr=%.1f us   F(MFMA waves 0-3 + pk waves 4-7)=%.1f us\n", run_pk<0>(256, iters, nv, d),
               run_pk<0>(512, iters, nv, d), run_pk<1>(512, iters, nv, d));
    }
    return 0;
}
