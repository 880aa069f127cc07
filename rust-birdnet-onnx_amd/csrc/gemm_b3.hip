// 1x1-conv / MatMul GEMM on the bf16 matrix pipe with f32-complete products, REGISTER-staged (gfx950 only; round 5).
//
// The arithmetic is bf16x3.h's: every f32 operand is the exact sum of three bf16 numbers, six of the nine partial products are kept
// (each exact in the f32 accumulator; the dropped ones are below 2^-24 of the product), so a product carries the error of one f32
// rounding -- what the exact-f32 matrix instruction commits as well -- at 96 instead of 256 matrix cycles per 16 x 16 tile and 32-deep
// k, on a pipe the vector ALU does not share.
//
// Why not the LDS-DMA structure of gemm_dma.hip / gemm_dma3.hip.  That structure exists to keep VECTOR instructions out of an exact-f32
// kernel, where they are paid out of the matrix time; its price is the issue cost of the DMA pieces (100 - 170 cycles per 1-KiB piece,
// profiles/r04_gemm_dma_stamps_*.txt), which the bf16 form with its 1.5 x larger weight image made the longest item of a K step
// (gemm_dma3_kernel: 1.2 x over the f32 kernel where the matrix time fell 2.7 x; tools/gemm3_bench).  On the bf16 pipe vector
// instructions issue in the matrix instructions' shadow (8 of every 16 cycles are free), so operands can go through registers again:
//
//   * weights: the planner packs them in FRAGMENT order (plan_rules.h, pack_w3f: [16-channel tile][K step][plane][lane][8 bf16]); a
//     wave owns whole channel tiles, so every weight element is needed by exactly one wave and goes global -> VGPR directly -- one
//     coalesced 1-KiB load per (tile, step, plane), PF steps ahead, no LDS, no barrier;
//   * activations: each thread loads 8 consecutive k of one row (32 B; a wave covers 16 rows x 128 B), multiplies the squeeze-excite
//     gate in (f32), splits ONCE for the whole block and writes the three bf16 planes to LDS ([plane][row][64 B], 16-byte chunk c of
//     row r in slot c ^ (-(r >> 2) & 3): ds_read_b128 is served in four NON-contiguous 16-lane groups -- {0-3, 12-15, 20-27}, ... -- so a
//     fragment read puts rows {0-3, 12-15} at chunk q and rows {4-11} at chunk q ^ 1 on the bank row at once; the first layout, c ^ ((r >> 2) & 3),
//     was 2-way conflicted, PMC 41 - 44 % of the LDS cycles); double-buffered, ONE raw barrier per K step;
//   * a block is TR = 16 MT rows of the BATCH's row matrix (tiles may span samples: the gate is per row) x NW channel tiles, one wave
//     per channel tile: 6 MT matrix instructions, 3 MT fragment reads and 3 weight loads per wave and K step.
//
// Arithmetic: an output element is one accumulation chain -- K steps ascending, inside a step the six partial products in mm6's order
// -- independent of the tile shape, the batch and the prefetch depth.  (Measured and dropped, tools/gemm3_bench: four K steps of prefetch
// instead of two -- 217 -> 238 us over the models' shapes at batch 32, 514 -> 631 us at batch 128; two K steps per barrier -- 224 -> 246 us over the models' shapes at batch 32, 523 ->
// 662 us at batch 128; 128-row tiles -- slower at every size.)  Same bits as gemm_dma3_kernel's K-slice-free form; other bits
// than the exact-f32 kernels (another summation order, the dropped 2^-24 terms), inside the same tolerance of the oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "bf16x3.h"
#include "device_common.h"
#include "kernels.h"
#include "plan_rules.h"

namespace bn {
namespace {

typedef b3_floatx4 floatx4;
typedef b3_u32x4 u32x4;

template <int N>
__device__ __forceinline__ void gb_act(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
    else if (act == ACT_SIGMOID) map_array<N>(v, [](float x) { return net_sigmoid(x); });
    else if (act == ACT_HSIGMOID) map_array<N>(v, [=](float x) { return fminf(fmaxf(p0 * x + p1, 0.0f), 1.0f); });
}

// MT 16-row tiles per block (every wave multiplies all of them), NW waves of NTW 16-channel tiles each.  NTW = 2 (round 5, the large
// products of v3.0 / Perch): an activation fragment read from LDS feeds twelve matrix instructions instead of six -- with one tile per wave
// the eight waves of a block read 96 KB of planes per K step for 768 matrix cycles per SIMD: exactly the LDS pipe's 768 cycles.
template <int MT, int NW, bool GATED, int PF, int NTW = 1>
__global__ __launch_bounds__(64 * NW) void gemm_b3_kernel(GemmDesc d, float *__restrict__ C, const float *__restrict__ A, const u32x4 *__restrict__ W3F,
                                                           const float *__restrict__ bias, const float *__restrict__ res, const float *__restrict__ scale,
                                                           int64_t total_rows, int nt16, int nst4, int nst) {
    static_assert(PF == 2 || PF == 4, "K steps of both operands in flight");
    constexpr int TR = 16 * MT, T = 64 * NW, SLOTS = 4 * TR, XS = (SLOTS + T - 1) / T;
    constexpr int PLANE_BYTES = TR * 64, BUF_BYTES = 3 * PLANE_BYTES;
    extern __shared__ __align__(1024) float gb_lds[];
    char *lds = reinterpret_cast<char *>(gb_lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TR;
    const int K = d.K;

    // ---- loader slots: slot = (row, 8-wide k chunk)
    const float *xsrc[XS];
    const float *gsrc[XS];
    int wofs[XS], kc[XS];
    bool slot_ok[XS];
#pragma unroll
    for (int j = 0; j < XS; j++) {
        const int slot = tid + T * j;
        slot_ok[j] = slot < SLOTS;
        const int s2 = slot_ok[j] ? slot : 0;
        const int row = s2 >> 2;
        kc[j] = s2 & 3;
        int64_t g = row0 + row;
        g = g < total_rows ? g : total_rows - 1;
        const int64_t b = g / d.rows, m = g - b * d.rows;
        xsrc[j] = A + b * d.a_bs + m * d.lda;
        gsrc[j] = GATED ? scale + b * d.s_bs : nullptr;
        wofs[j] = row * 64 + 16 * (kc[j] ^ ((0 - (row >> 2)) & 3));
    }
    const u32x4 *wsrc[NTW];  // this wave's channel tiles (a padding tile repeats the last one, stores nothing)
#pragma unroll
    for (int jt = 0; jt < NTW; jt++)
        wsrc[jt] = W3F + (int64_t)min(((int)blockIdx.y * NW + wave) * NTW + jt, nt16 - 1) * nst4 * 192 + lane;  // (nst4 K steps per tile, zeros past K)

    floatx4 xr[PF][XS][2], gr[PF][XS][2];
    u32x4 wr[PF][NTW][3];
    auto load_x = [&](int s, int u) {
#pragma unroll
        for (int j = 0; j < XS; j++) {
            const int k = 32 * s + 8 * kc[j];
            const int ks = k + 8 <= K ? k : 0;  // (the half step's upper chunks: any address that exists; zeroed in write_x)
            xr[u][j][0] = *reinterpret_cast<const floatx4 *>(xsrc[j] + ks);
            xr[u][j][1] = *reinterpret_cast<const floatx4 *>(xsrc[j] + ks + 4);
            if constexpr (GATED) {
                gr[u][j][0] = *reinterpret_cast<const floatx4 *>(gsrc[j] + ks);
                gr[u][j][1] = *reinterpret_cast<const floatx4 *>(gsrc[j] + ks + 4);
            }
        }
    };
    auto load_w = [&](int s, int u) {
#pragma unroll
        for (int jt = 0; jt < NTW; jt++)
#pragma unroll
            for (int p = 0; p < 3; p++) wr[u][jt][p] = wsrc[jt][(s * 3 + p) * 64];
    };
    auto write_x = [&](int s, int u, int buf) {
#pragma unroll
        for (int j = 0; j < XS; j++) {
            floatx4 x0 = xr[u][j][0], x1 = xr[u][j][1];
            if constexpr (GATED) {
                x0 *= gr[u][j][0];
                x1 *= gr[u][j][1];
            }
            if (32 * s + 8 * kc[j] + 8 > K) {
                x0 = floatx4{0.f, 0.f, 0.f, 0.f};
                x1 = floatx4{0.f, 0.f, 0.f, 0.f};
            }
            u32x4 h, m, l;
            split3(x0, x1, h, m, l);
            if (slot_ok[j]) {
                char *dst = lds + buf * BUF_BYTES + wofs[j];
                *reinterpret_cast<u32x4 *>(dst) = h;
                *reinterpret_cast<u32x4 *>(dst + PLANE_BYTES) = m;
                *reinterpret_cast<u32x4 *>(dst + 2 * PLANE_BYTES) = l;
            }
        }
    };

    floatx4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int jt = 0; jt < NTW; jt++) acc[mt][jt] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int aoff = lc * 64 + 16 * (lq ^ ((0 - (lc >> 2)) & 3));
    auto compute = [&](int u, int buf) {
        const char *ab = lds + buf * BUF_BYTES + aoff;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const u32x4 ah = *reinterpret_cast<const u32x4 *>(ab + mt * 1024);
            const u32x4 am = *reinterpret_cast<const u32x4 *>(ab + mt * 1024 + PLANE_BYTES);
            const u32x4 al = *reinterpret_cast<const u32x4 *>(ab + mt * 1024 + 2 * PLANE_BYTES);
#pragma unroll
            for (int jt = 0; jt < NTW; jt++) acc[mt][jt] = mm6(wr[u][jt][0], wr[u][jt][1], wr[u][jt][2], ah, am, al, acc[mt][jt]);
        }
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- the K loop.  nst is EVEN (pack_w3f pads the planes with a zero step) and every load of the loop is issued unconditionally, with
    // its step index clamped to the last one: the number of loads in flight at any point is then the same on every path, which is what
    // lets the compiler count its vmcnt waits -- with conditional prefetches it has to drain them all (vmcnt(0)) in front of every use,
    // i.e. one L2 round trip per K step (seen in the first build's ISA; 0.6 us per step).  The re-loaded last step and the planes written
    // behind the last step are never read.
    const int last = nst - 1;
#pragma unroll
    for (int u = 0; u < PF; u++) {
        load_x(min(u, last), u);
        load_w(min(u, last), u);
    }
    write_x(0, 0, 0);
    load_x(min(PF, last), 0);
    sync();
    for (int i0 = 0; i0 < nst; i0 += PF) {  // (nst is a multiple of PF: the launcher pads the run with zero steps the planes hold)
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int i = i0 + u;
            const int u1 = (u + 1) % PF;
            write_x(min(i + 1, last), u1, (u + 1) & 1);  // the other buffer: every wave is past its reads of step i - 1 (previous barrier)
            load_x(min(i + 1 + PF, last), u1);
            __builtin_amdgcn_sched_barrier(0);  // (keeps the prefetch in front of the products: left alone the compiler sinks it behind them)
            compute(u, u & 1);
            load_w(min(i + PF, last), u);
            sync();
        }
    }

    // ---- epilogue: lane (lc, lq) holds channels n .. n + 3 of row lc of each m-tile, per channel tile
#pragma unroll
    for (int jt = 0; jt < NTW; jt++) {
        const int n = (((int)blockIdx.y * NW + wave) * NTW + jt) * 16 + 4 * lq;
        if (n >= d.N) continue;  // (a padding tile / the padding lanes of the last tile)
        const floatx4 bv = d.has_bias ? *reinterpret_cast<const floatx4 *>(bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
        float v[MT * 4];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[mt * 4 + i] = acc[mt][jt][i] + bv[i];
        gb_act<MT * 4>(d.act, d.p0, d.p1, v);
        if (d.gap) {  // (launcher: TR == rows, one block per sample) the sample's mean over its rows: m-tiles ascending, then a fixed butterfly over the 16 rows
            floatx4 sm = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < 4; i++) sm[i] += v[mt * 4 + i];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int i = 0; i < 4; i++) sm[i] += __shfl_xor(sm[i], o);
            if (lc == 0) {
                const float rows = (float)TR;
                *reinterpret_cast<floatx4 *>(C + (int64_t)blockIdx.x * d.c_bs + n) = floatx4{sm[0] / rows, sm[1] / rows, sm[2] / rows, sm[3] / rows};
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int64_t g = row0 + mt * 16 + lc;
            if (g < total_rows) {
                const int64_t b = g / d.rows, m = g - b * d.rows;
                floatx4 o = floatx4{v[mt * 4], v[mt * 4 + 1], v[mt * 4 + 2], v[mt * 4 + 3]};
                if (d.has_res) o += *reinterpret_cast<const floatx4 *>(res + b * d.r_bs + m * d.ldr + n);
                *reinterpret_cast<floatx4 *>(C + b * d.c_bs + m * d.ldc + n) = o;
            }
        }
    }
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int MT, int NW, int NTW = 1>
void launch_b3(hipStream_t s, const GemmDesc &d, float *C, const float *A, const u32x4 *W3F, const float *bias, const float *res, const float *scale,
               int64_t total_rows, int nt16, int nst4, int nb) {
    constexpr int TR = 16 * MT;
    dim3 grid((unsigned)((total_rows + TR - 1) / TR), (unsigned)nb);
    const size_t lds = 2 * 3 * (size_t)TR * 64;
    const int nst = (d.K + 31) / 32;
    // two K steps of both operands in flight (PF = 4 is instantiable: measured slower at every size, tools/gemm3_bench)
    const int nrun = (nst + 1) & ~1;
    if (d.has_scale) hipLaunchKernelGGL((gemm_b3_kernel<MT, NW, true, 2, NTW>), grid, dim3(64 * NW), lds, s, d, C, A, W3F, bias, res, scale, total_rows, nt16, nst4, nrun);
    else hipLaunchKernelGGL((gemm_b3_kernel<MT, NW, false, 2, NTW>), grid, dim3(64 * NW), lds, s, d, C, A, W3F, bias, res, scale, total_rows, nt16, nst4, nrun);
}

}  // namespace

// d.w3 == 2 launches: W3F = pack_w3f's fragment-order image.  Returns false (nothing launched) when a pointer's alignment rules the
// kernel out -- there is no other kernel for a packed layer, the caller reports it.
bool launch_gemm_b3(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W3F, const float *bias, const float *res, const float *scale,
                    int64_t batch) {
    if (d.w3 != 2 || d.fold || d.npost || d.out_strided || d.se_inline || d.K % 8 || d.lda % 4 || d.N % 4 || (d.gap && d.rows != 48)) return false;
    if (!al16(A) || !al16(W3F) || !al16(C) || (d.has_res && !al16(res)) || (d.has_bias && !al16(bias)) || (d.has_scale && !al16(scale))) return false;
    const int nt16 = (d.N + 15) / 16, nst = ((d.K + 31) / 32 + 3) & ~3;  // (the planes' steps per tile: pack_w3f pads to a multiple of four)
    const int nb = (nt16 + 7) / 8;                 // channel blocks of at most 8 tiles (128 channels), evenly sized
    const int nw = (nt16 + nb - 1) / nb;           // waves = channel tiles per block
    const int64_t total_rows = batch * d.rows;
    const u32x4 *W = reinterpret_cast<const u32x4 *>(W3F);
    // rows per block (it does not enter the arithmetic): a pooled epilogue needs the sample's 48 rows in one block; otherwise 64-row tiles
    // once they give every CU a block (half the weight traffic of 32-row tiles), 32-row tiles below that -- the launch is then a chain of
    // K steps per block, and twice the blocks is the only parallelism left (tools/gemm3_bench sweeps: 128-row tiles lose at every size)
    // That is the rule for ONE context on the device.  With several in flight the launch shares the CUs, the blocks of other launches are
    // the parallelism, and the weight traffic decides: 64-row tiles throughout gain 1.9 % on the v2.4 step with four contexts (68.8 -> 70.1 k
    // segments/s) and cost the one-context chain 65 us -- so they are taken as soon as a second live context exists (kernels.h).
    const int64_t cus = device_cu_count();
    const int force_mt = env_int("BN_GEMMB3_MT", 0);  // tests / experiments
    int mt = (device_context_count() > 1 || (total_rows + 63) / 64 * nb >= cus) ? 4 : 2;
    if (d.gap) mt = 3;
    else if (force_mt == 2 || force_mt == 3 || force_mt == 4) mt = force_mt;
#define GB_GO(MT)                                                                                          \
    do {                                                                                                   \
        switch (nw) { /* N > 48 (gemm_b3_shape_ok) => at least four channel tiles per block */             \
            case 4: launch_b3<MT, 4>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb); break;   \
            case 5: launch_b3<MT, 5>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb); break;   \
            case 6: launch_b3<MT, 6>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb); break;   \
            case 7: launch_b3<MT, 7>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb); break;   \
            default: launch_b3<MT, 8>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb); break;  \
        }                                                                                                  \
    } while (0)
    if (nw < 4) return false;
    // two channel tiles per wave: the large products (64-row tiles, at least 16 channel tiles, enough blocks to fill the device twice over)
    const int ntw_env = env_int("BN_GEMMB3_NTW", 0);
    if (mt == 4 && !d.gap && nt16 >= 16 && ntw_env != 1 && (ntw_env == 2 || (total_rows + 63) / 64 * ((nt16 + 15) / 16) >= 2 * cus)) {
        const int nb2 = (nt16 + 15) / 16;                  // channel blocks of at most 16 tiles (256 channels), evenly sized, two tiles per wave
        const int nw2 = ((nt16 + nb2 - 1) / nb2 + 1) / 2;  // waves per block
        switch (nw2) {
            case 4: launch_b3<4, 4, 2>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb2); return true;
            case 5: launch_b3<4, 5, 2>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb2); return true;
            case 6: launch_b3<4, 6, 2>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb2); return true;
            case 7: launch_b3<4, 7, 2>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb2); return true;
            case 8: launch_b3<4, 8, 2>(s, d, C, A, W, bias, res, scale, total_rows, nt16, nst, nb2); return true;
            default: break;  // (fewer than eight tiles per block cannot happen with nt16 >= 16)
        }
    }
    if (mt == 2) GB_GO(2);
    else if (mt == 3) GB_GO(3);
    else GB_GO(4);
#undef GB_GO
    return true;
}

}  // namespace bn
