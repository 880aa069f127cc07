// Where the cycles of one LDS-DMA GEMM block go: shader-clock stamps around every K step of one block's eight waves.
// Build (diagnostic objects, not the library's):
//   S=rust-birdnet-onnx_amd/csrc; hipcc -O3 -std=c++17 -fPIC -Iinclude --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DBN_GD_STAMPS -c $S/gemm_dma.hip -o /tmp/gd_stamps.o
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -I$S tools/gemm_stamps.cpp /tmp/gd_stamps.o $S/kernels.o $S/stft.o $S/topk.o $S/mbrow.o $S/mbmap.o -o tools/gemm_stamps
//   tools/gemm_stamps <batch> "<rows> <K> <N> <gate> <res> <act>"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace bn {
void gemm_dma_read_stamps(unsigned long long *out);
}
using namespace bn;

int main(int argc, char **argv) {
    const int64_t batch = argc > 1 ? atoll(argv[1]) : 32;
    long long rows = 192;
    int K = 672, N = 112, gate = 1, res = 1, act = 0;
    if (argc > 2) sscanf(argv[2], "%lld %d %d %d %d %d", &rows, &K, &N, &gate, &res, &act);
    if (!prepare_device(0)) { fprintf(stderr, "prepare_device failed\n"); return 1; }
    note_launch_device(0);
    GemmDesc d{};
    d.rows = rows; d.K = K; d.N = N; d.lda = K; d.a_bs = rows * K; d.ldc = N; d.c_bs = rows * N; d.ldr = N; d.r_bs = rows * N;
    d.act = act; d.has_bias = 1; d.has_res = res; d.has_scale = gate; d.s_bs = (K + 3) / 4 * 4;
    size_t a_elems = (size_t)d.a_bs * batch + 4096, w_elems = (size_t)K * N, c_elems = (size_t)d.c_bs * batch;
    float *A, *W, *C, *R, *B, *S;
    hipMalloc(&A, a_elems * 4); hipMalloc(&W, w_elems * 4); hipMalloc(&C, c_elems * 4); hipMalloc(&R, c_elems * 4);
    hipMalloc(&B, (size_t)N * 4); hipMalloc(&S, (size_t)d.s_bs * batch * 4);
    std::vector<float> h(std::max({a_elems, w_elems, c_elems}));
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(A, h.data(), a_elems * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), w_elems * 4, hipMemcpyHostToDevice);
    hipMemcpy(R, h.data(), c_elems * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    hipMemcpy(S, h.data(), (size_t)d.s_bs * batch * 4, hipMemcpyHostToDevice);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) launch_gemm(st, d, C, A, W, B, res ? R : nullptr, gate ? S : nullptr, batch);
    hipEventRecord(e0, st);
    const int iters = 50;
    for (int i = 0; i < iters; i++) launch_gemm(st, d, C, A, W, B, res ? R : nullptr, gate ? S : nullptr, batch);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / iters, macs = (double)rows * batch * K * N;
    printf("rows=%lld K=%d N=%d gate=%d res=%d batch=%lld: %.1f us  %.2f TF/s\n", rows, K, N, gate, res, (long long)batch, us, 2 * macs / us / 1e6);
    std::vector<unsigned long long> s((size_t)8 * 96 * 3);
    gemm_dma_read_stamps(s.data());
    auto at = [&](int w, int it, int k) { return s[((size_t)w * 96 + it) * 3 + k]; };
    const int iterations = std::min(96, (K / 32) / 2);  // two K slices: every wave takes every second step
    printf("wave: per iteration [wait+barrier+refill | fragment reads + matrix instructions] in shader cycles (last launch, middle block)\n");
    for (int w = 0; w < 8; w++) {
        printf("w%d:", w);
        unsigned long long tw = 0, tc = 0;
        for (int it = 0; it < iterations; it++) {
            const unsigned long long a = at(w, it, 1) - at(w, it, 0), b = at(w, it, 2) - at(w, it, 1);
            if (it < 12) printf(" [%llu|%llu]", a, b);
            tw += a; tc += b;
        }
        printf("  sum wait %llu compute %llu (wait share %.2f); loop span %llu\n", tw, tc, (double)tw / (double)(tw + tc),
               at(w, iterations - 1, 2) - at(w, 0, 0));
    }
    // skew: arrival (stamp 0) of each wave at iteration 4 relative to the first
    unsigned long long first = ~0ull;
    for (int w = 0; w < 8; w++) first = std::min(first, at(w, 4, 0));
    printf("arrival at the wait of iteration 4, relative to the first wave:");
    for (int w = 0; w < 8; w++) printf(" %llu", at(w, 4, 0) - first);
    printf("\nleave (behind barrier) of iteration 4:");
    first = ~0ull;
    for (int w = 0; w < 8; w++) first = std::min(first, at(w, 4, 1));
    for (int w = 0; w < 8; w++) printf(" %llu", at(w, 4, 1) - first);
    printf("\n");
    return 0;
}
