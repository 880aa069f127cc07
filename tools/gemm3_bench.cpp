// Stand-alone A/B of the exact-f32 LDS-DMA GEMM (gemm_dma.hip) against the bf16x3 form (gemm_dma3.hip) on the GEMM shapes of the three
// benchmark models, with both checked against a double-precision host product on sampled outputs.
// Build: make tools/gemm3_bench   (links the in-tree objects; never part of the library)
//   tools/gemm3_bench [batch=32] [iters=50]          SHAPE="rows K N gate res act" for one custom shape
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"
#include "plan_rules.h"

using namespace bn;
namespace bn { bool prepare_device(int dev); }

struct Shape { const char *name; int64_t rows; int K, N; int act, gate, res; };

int main(int argc, char **argv) {
    const int64_t batch = argc > 1 ? atoll(argv[1]) : 32;
    const int iters = argc > 2 ? atoi(argv[2]) : 50;
    std::vector<Shape> shapes = {
        {"v24 project 96->24 gate", 3072, 96, 24, 0, 1, 0},      {"v24 project 144->40 gate", 768, 144, 40, 0, 1, 0},
        {"v24 project 240->40 gate res", 768, 240, 40, 0, 1, 1}, {"v24 project 240->80 gate", 192, 240, 80, 0, 1, 0},
        {"v24 project 480->80 gate res", 192, 480, 80, 0, 1, 1}, {"v24 project 480->112 gate", 192, 480, 112, 0, 1, 0},
        {"v24 project 672->112 gate res", 192, 672, 112, 0, 1, 1}, {"v24 project 672->192 gate", 48, 672, 192, 0, 1, 0},
        {"v24 project 1152->192 gate res", 48, 1152, 192, 0, 1, 1}, {"v24 project 1152->320 gate", 48, 1152, 320, 0, 1, 0},
        {"v24 head 320->1024 relu", 48, 320, 1024, 1, 0, 0},     {"perch-like 1392->232 gate res", 64, 1392, 232, 0, 1, 1},
        {"perch expand 232->1392 silu", 64, 232, 1392, 4, 0, 0},  {"perch expand 136->816 silu", 256, 136, 816, 4, 0, 0},
        {"perch expand 96->576 silu", 256, 96, 576, 4, 0, 0},     {"v30 expand 112->672 silu", 256, 112, 672, 4, 0, 0}, {"v30 expand 80->480 silu", 256, 80, 480, 4, 0, 0},  {"tail K=144 (half step) 144->24", 3072, 144, 24, 0, 1, 1},
    };
    if (getenv("SHAPE")) {
        long long r_; int k_, n_, g_, rs_, a_;
        if (sscanf(getenv("SHAPE"), "%lld %d %d %d %d %d", &r_, &k_, &n_, &g_, &rs_, &a_) == 6) shapes = {{"custom", r_, k_, n_, a_, g_, rs_}};
    }
    hipSetDevice(0);
    prepare_device(0);
    note_launch_device(0);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double tot[3] = {0, 0, 0};
    for (auto &s : shapes) {
        GemmDesc d{};
        d.rows = s.rows; d.K = s.K; d.N = s.N; d.lda = s.K; d.a_bs = s.rows * s.K;
        d.ldc = s.N; d.c_bs = s.rows * s.N; d.ldr = s.N; d.r_bs = s.rows * s.N;
        d.act = s.act; d.has_bias = 1; d.has_res = s.res; d.has_scale = s.gate; d.s_bs = (s.K + 3) / 4 * 4;
        const int shape = gemm_dma_shape(d);
        const size_t a_elems = (size_t)d.a_bs * batch, w_elems = (size_t)s.K * s.N, c_elems = (size_t)d.c_bs * batch, s_elems = (size_t)d.s_bs * batch;
        std::vector<float> hA(a_elems), hW(w_elems), hR(c_elems), hB(s.N), hS(s_elems), hC[3];
        uint32_t seed = 12345;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xffffff) / 16777216.0f - 0.5f; };
        for (auto &v : hA) v = 2.0f * rnd() * (1.0f + 3.0f * (rnd() > 0.4f));  // activations of mixed magnitude, all 24 significand bits in use
        for (auto &v : hW) v = rnd() / std::sqrt((float)s.K) * 4.0f;
        for (auto &v : hR) v = rnd();
        for (auto &v : hB) v = rnd();
        for (auto &v : hS) v = 0.5f + rnd();  // gate in (0, 1)
        std::vector<float> w3 = pack_w3(hW.data(), s.N, s.K), w3f = pack_w3f(hW.data(), s.N, s.K);
        float *A, *W, *W3, *W3F, *C, *R, *B, *S;
        hipMalloc(&W3F, w3f.size() * 4); hipMemcpy(W3F, w3f.data(), w3f.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&A, a_elems * 4); hipMalloc(&W, w_elems * 4); hipMalloc(&W3, w3.size() * 4); hipMalloc(&C, c_elems * 4); hipMalloc(&R, c_elems * 4);
        hipMalloc(&B, (size_t)s.N * 4); hipMalloc(&S, s_elems * 4);
        hipMemcpy(A, hA.data(), a_elems * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), w_elems * 4, hipMemcpyHostToDevice);
        hipMemcpy(W3, w3.data(), w3.size() * 4, hipMemcpyHostToDevice); hipMemcpy(R, hR.data(), c_elems * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), (size_t)s.N * 4, hipMemcpyHostToDevice); hipMemcpy(S, hS.data(), s_elems * 4, hipMemcpyHostToDevice);
        double us[3] = {0, 0, 0}, err[3] = {0, 0, 0}, rel[3] = {0, 0, 0};
        for (int form = 0; form < 3; form++) {
            GemmDesc df = d;
            df.w3 = form;
            auto go = [&]() { if (form == 0 && !shape) { launch_gemm(st, df, C, A, W, B, s.res ? R : nullptr, s.gate ? S : nullptr, batch); return true; }  // (the tiled / split-K kernels)
                              if (form == 1 && !shape) return false;
                              return form == 2 ? launch_gemm_b3(st, df, C, A, W3F, B, s.res ? R : nullptr, s.gate ? S : nullptr, batch)
                                   : form ? launch_gemm_dma3(st, df, C, A, W3, B, s.res ? R : nullptr, s.gate ? S : nullptr, batch) : launch_gemm_dma(st, df, C, A, W, B, s.res ? R : nullptr, s.gate ? S : nullptr, batch); };
            hipMemset(C, 0xff, c_elems * 4);
            if ((form == 2 && !gemm_b3_shape_ok(d)) || !go()) { us[form] = -1; continue; }
            for (int i = 0; i < 5; i++) go();
            hipEventRecord(e0, st);
            for (int i = 0; i < iters; i++) go();
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[form] = ms * 1000.0 / iters;
            hC[form].resize(c_elems);
            hipMemcpy(hC[form].data(), C, c_elems * 4, hipMemcpyDeviceToHost);
            // sampled outputs against a double-precision product of the SAME f32 operands (gate product rounded to f32 as both kernels do)
            for (int t = 0; t < 4000; t++) {
                seed = seed * 1664525u + 1013904223u;
                const int64_t bb = seed % batch, m = (seed >> 8) % s.rows, n = (seed >> 16) % s.N;
                double acc = 0, mag = 0;
                for (int k = 0; k < s.K; k++) {
                    float x = hA[(size_t)(bb * d.a_bs + m * s.K + k)];
                    if (s.gate) x = x * hS[(size_t)(bb * d.s_bs + k)];
                    acc += (double)x * (double)hW[(size_t)n * s.K + k];
                    mag += std::fabs((double)x * (double)hW[(size_t)n * s.K + k]);
                }
                acc += hB[n];
                if (s.act == ACT_RELU) acc = std::max(acc, 0.0);
                else if (s.act == ACT_SILU) acc = acc / (1.0 + std::exp(-acc));
                if (s.res) acc += hR[(size_t)(bb * d.r_bs + m * s.N + n)];
                const double got = hC[form][(size_t)(bb * d.c_bs + m * s.N + n)];
                err[form] = std::max(err[form], std::fabs(got - acc));
                rel[form] = std::max(rel[form], std::fabs(got - acc) / (mag + 1e-30));  // relative to the sum of |terms|: the scale rounding errors live on
            }
        }
        const double macs = (double)s.rows * batch * s.K * s.N;
        printf("%-32s rows=%5lld K=%5d N=%5d shape=%d | f32 %7.1f us %6.1f TF err %.2e (%.1e) | dma3 %7.1f us %6.1f TF err %.2e (%.1e) | b3 %7.1f us %6.1f TF err %.2e (%.1e) | x%.2f x%.2f\n", s.name,
               (long long)s.rows, s.K, s.N, shape, us[0], 2 * macs / us[0] / 1e6, err[0], rel[0], us[1], 2 * macs / us[1] / 1e6, err[1], rel[1], us[2], 2 * macs / us[2] / 1e6, err[2], rel[2],
               us[0] / us[1], us[0] / us[2]);
        if (us[0] > 0 && us[2] > 0) { tot[0] += us[0]; tot[1] += us[1] > 0 ? us[1] : us[0]; tot[2] += us[2]; }
        hipFree(W3F);
        hipFree(A); hipFree(W); hipFree(W3); hipFree(C); hipFree(R); hipFree(B); hipFree(S);
    }
    printf("TOTAL f32 %.1f us, dma3 %.1f us (x%.2f), b3 %.1f us (x%.2f)\n", tot[0], tot[1], tot[0] / tot[1], tot[2], tot[0] / tot[2]);
    return 0;
}
