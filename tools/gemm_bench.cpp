// Stand-alone timing of launch_gemm on the benchmark model's GEMM shapes (batch 32).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Irust-birdnet-onnx_amd/csrc tools/gemm_bench.cpp rust-birdnet-onnx_amd/csrc/kernels.o -o tools/gemm_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"

using namespace bn;

struct Shape {
    const char *name;
    int64_t rows;
    int K, N;
    int64_t lda, a_bs;  // 0 => dense
    int act, gate, res;
};

int main(int argc, char **argv) {
    const int64_t batch = argc > 1 ? atoll(argv[1]) : 32;
    const int iters = argc > 3 ? atoi(argv[3]) : 50;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    std::vector<Shape> shapes = {
        {"dft2048 (conv1d hop278)", 511, 2048, 127, 278, 144000, 0, 0, 0},
        {"dft1024 (conv1d hop280)", 511, 1024, 309, 280, 144000, 0, 0, 0},
        {"mel 127->96", 511, 127, 96, 0, 0, 0, 0, 0},
        {"mel 309->96", 511, 309, 96, 0, 0, 0, 0, 0},
        {"project 32->16 gate", 12288, 32, 16, 0, 0, 0, 1, 0},
        {"expand 16->96 relu", 12288, 16, 96, 0, 0, 1, 0, 0},
        {"project 96->24 gate", 3072, 96, 24, 0, 0, 0, 1, 0},
        {"expand 24->144 relu", 3072, 24, 144, 0, 0, 1, 0, 0},
        {"project 144->24 gate res", 3072, 144, 24, 0, 0, 0, 1, 1},
        {"project 144->40 gate", 768, 144, 40, 0, 0, 0, 1, 0},
        {"expand 40->240 relu", 768, 40, 240, 0, 0, 1, 0, 0},
        {"project 240->40 gate res", 768, 240, 40, 0, 0, 0, 1, 1},
        {"project 240->80 gate", 192, 240, 80, 0, 0, 0, 1, 0},
        {"expand 80->480 relu", 192, 80, 480, 0, 0, 1, 0, 0},
        {"project 480->80 gate res", 192, 480, 80, 0, 0, 0, 1, 1},
        {"expand 112->672 relu", 192, 112, 672, 0, 0, 1, 0, 0},
        {"project 672->112 gate res", 192, 672, 112, 0, 0, 0, 1, 1},
        {"project 672->192 gate", 48, 672, 192, 0, 0, 0, 1, 0},
        {"expand 192->1152 relu", 48, 192, 1152, 0, 0, 1, 0, 0},
        {"project 1152->192 gate res", 48, 1152, 192, 0, 0, 0, 1, 1},
        {"project 1152->320 gate", 48, 1152, 320, 0, 0, 0, 1, 0},
        {"head 320->1024 relu", 48, 320, 1024, 0, 0, 1, 0, 0},
        {"fc 1024->6522", 1, 1024, 6522, 0, 0, 0, 0, 0},
    };
    if (getenv("SHAPE")) {  // SHAPE="rows K N gate res act": one custom shape instead of the model's list
        long long r_; int k_, n_, g_, rs_, a_;
        if (sscanf(getenv("SHAPE"), "%lld %d %d %d %d %d", &r_, &k_, &n_, &g_, &rs_, &a_) == 6) shapes = {{"custom", r_, k_, n_, 0, 0, a_, g_, rs_}};
    }
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double total = 0;
    int sidx = -1;
    for (auto &s : shapes) {
        sidx++;
        if (only >= 0 && sidx != only) continue;
        GemmDesc d{};
        d.rows = s.rows; d.K = s.K; d.N = s.N;
        d.lda = s.lda ? s.lda : s.K;
        d.a_bs = s.a_bs ? s.a_bs : s.rows * d.lda;
        d.ldc = s.N; d.c_bs = s.rows * s.N; d.ldr = s.N; d.r_bs = s.rows * s.N;
        d.act = s.act; d.has_bias = 1; d.has_res = s.res; d.has_scale = s.gate; d.s_bs = (s.K + 3) / 4 * 4;
        size_t a_elems = (size_t)d.a_bs * batch + 4096, w_elems = (size_t)s.K * s.N, c_elems = (size_t)d.c_bs * batch;
        float *A, *W, *C, *R, *B, *S;
        hipMalloc(&A, a_elems * 4); hipMalloc(&W, w_elems * 4); hipMalloc(&C, c_elems * 4); hipMalloc(&R, c_elems * 4);
        hipMalloc(&B, (size_t)s.N * 4); hipMalloc(&S, (size_t)d.s_bs * batch * 4);
        std::vector<float> h(std::max({a_elems, w_elems, c_elems}));
        for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
        hipMemcpy(A, h.data(), a_elems * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), w_elems * 4, hipMemcpyHostToDevice);
        hipMemcpy(R, h.data(), c_elems * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, h.data(), (size_t)s.N * 4, hipMemcpyHostToDevice);
        hipMemcpy(S, h.data(), (size_t)d.s_bs * batch * 4, hipMemcpyHostToDevice);
        for (int i = 0; i < 5; i++) launch_gemm(st, d, C, A, W, B, s.res ? R : nullptr, s.gate ? S : nullptr, batch);
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; i++) launch_gemm(st, d, C, A, W, B, s.res ? R : nullptr, s.gate ? S : nullptr, batch);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1000.0 / iters;
        double macs = (double)s.rows * batch * s.K * s.N;
        double bytes = 4.0 * ((double)s.rows * batch * (s.K + s.N * (s.res ? 2 : 1)) + (double)s.K * s.N);
        printf("%-28s rows=%6lld K=%5d N=%5d  %8.1f us  %7.2f TF/s  %7.1f GB/s\n", s.name, (long long)s.rows, s.K, s.N, us, 2 * macs / us / 1e6, bytes / us / 1e3);
        total += us;
        hipFree(A); hipFree(W); hipFree(C); hipFree(R); hipFree(B); hipFree(S);
    }
    printf("TOTAL (one of each) %.1f us\n", total);
    return 0;
}
