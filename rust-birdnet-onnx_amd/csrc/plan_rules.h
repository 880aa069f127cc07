// Shape rules the planner (engine.cpp, host C++) and the launchers (*.hip) must agree on: which kernel a descriptor may
// take and how much LDS that kernel carves up for it.  They are pure functions of per-sample quantities (and of the
// documented BN_* switches, read per call because the tests flip them), defined ONCE here and included by both sides --
// so the host-only sanitizer build of the planner (tools/asan_plan.cpp, tests/test_planner_sanitizers.py) links without
// any HIP object and without restating a rule.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.h"

#if defined(__HIPCC__)
#define BN_HD __host__ __device__
#else
#define BN_HD
#endif

namespace bn {

inline bool ptr_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ---- tiled GEMM family (kernels.hip) ------------------------------------------------------------------------------
constexpr int GEMM_BK_RULE = 32, GEMM_LD_RULE = GEMM_BK_RULE + 4, FRAME_BM_RULE = 64;

// Folded framing GEMM from an LDS-resident signal span (frame_fold_kernel): per-sample quantities and, when given, the
// filter pointer's alignment.  BN_FRAMELDS=0 disables.
inline bool frame_fold_shape_ok(const GemmDesc &d, const float *W) {
    if (env_int("BN_FRAMELDS", 1) == 0) return false;
    if (d.has_res || d.has_scale || d.act != ACT_NONE || d.K % GEMM_BK_RULE || d.fold_n != 2 * d.K || d.K % 4 || (W && !ptr_aligned16(W))) return false;
    if (d.lda <= 0 || d.lda > 4096 || d.rows < 32 || d.c_bs < 0) return false;
    return true;
}

// ... with an absorbed elementwise chain / the consumer's output view (planner rule E, round 4): only where the LDS-resident kernel is
// certain to take the launch -- the generic folded GEMM has no such epilogue -- i.e. whatever tile width the launcher picks fits the LDS
inline bool frame_fold_post_ok(const GemmDesc &d) {
    if (!frame_fold_shape_ok(d, nullptr)) return false;
    const int64_t span = (int64_t)(FRAME_BM_RULE - 1) * d.lda + d.fold_n;
    const int wn = d.N <= 96 ? std::max(2, (d.N + 31) / 32) : 5;  // (the launcher's widest choice)
    return (size_t)(((span + 3) & ~3) + 2 * FRAME_BM_RULE * GEMM_LD_RULE + 2 * 32 * wn * GEMM_LD_RULE) * sizeof(float) <= 160 * 1024;
}

// Quarter-folded framing GEMM (frame_fold2_kernel, GemmDesc::fold == 2): per-sample quantities and, when given, the filter pointer's
// alignment.  One N tile holds every column (two to five wave columns of 32), the block keeps the signal span, both window tables, two
// operand tiles (S and D) and the filter tile double-buffered in LDS.  BN_FRAMELDS=0 / BN_CONVFOLD2=0 disable (the planner then keeps
// the half fold).
inline size_t frame_fold2_lds_bytes(const GemmDesc &d) {
    const int64_t span = (int64_t)(FRAME_BM_RULE - 1) * d.lda + d.fold_n + 4;  // (+ the one element behind the last frame that tap 0 pairs with)
    return (size_t)(((span + 3) & ~3) + 2 * d.K + 4 * FRAME_BM_RULE * GEMM_LD_RULE + 2 * d.N * GEMM_LD_RULE) * sizeof(float);
}
// half-height form (frame_fold2p_kernel): 32-row span, both window tables, two S | D operand tile pairs; taken where two blocks fit a CU
inline size_t frame_fold2p_lds_bytes(const GemmDesc &d) {
    const int64_t span = (int64_t)31 * d.lda + d.fold_n + 4;
    return (size_t)(((span + 3) & ~3) + 2 * d.K + 4 * 32 * GEMM_LD_RULE) * sizeof(float);
}
// ... on the bf16 matrix pipe (frame_fold2q_kernel): the operand tiles are three bf16 planes each, rows of 80 bytes
inline size_t frame_fold2q_lds_bytes(const GemmDesc &d) {
    const int64_t span = (int64_t)31 * d.lda + d.fold_n + 4;
    return (size_t)(((span + 3) & ~3) + 2 * d.K) * sizeof(float) + (size_t)2 * 6 * 32 * 80;
}
inline bool frame_fold2q_ok(const GemmDesc &d) { return env_int("BN_FRAME2_B3", 1) != 0 && env_int("BN_GEMM3", 2) != 0 && frame_fold2q_lds_bytes(d) <= 80 * 1024; }
inline bool frame_fold2p_ok(const GemmDesc &d) { return env_int("BN_FRAME2_WPK", 1) != 0 && frame_fold2p_lds_bytes(d) <= 80 * 1024; }
inline bool frame_fold2_shape_ok(const GemmDesc &d, const float *W) {
    if (env_int("BN_FRAMELDS", 1) == 0 || env_int("BN_CONVFOLD2", 1) == 0) return false;
    if (d.fold != 2 || d.has_res || d.has_scale || d.act != ACT_NONE || d.npost || d.out_strided || d.se_inline) return false;
    if (d.fold_n < 256 || d.fold_n % 128 || d.K != d.fold_n / 4 + GEMM_BK_RULE || (W && !ptr_aligned16(W))) return false;
    if (d.N % 32 || d.N < 64 || d.N > 160 || d.fold_ne % 32 || d.fold_ne < 0 || d.fold_ne > d.N) return false;
    if (d.lda <= 0 || d.lda > 4096 || d.rows < 32 || d.c_bs < 0) return false;
    return frame_fold2_lds_bytes(d) <= 160 * 1024;
}

// Planner rule J and its launcher agree through this: the folded framing GEMM `d` (its LDS-resident kernel) followed by a plain
// product over its rows (frame_fold_kernel<true>).
// Opt-in (BN_FRAMEPAIR=1): correct, one launch and the spectrum's round trip less -- and slower: behind the K loop the block's eight
// waves stage the second product's filter rows and run its epilogue with the CU to themselves, 68.4 us against 50 + 19 at batch 32 and
// 63 against 56 us of marginal cost, where the separate launch spreads the same work over the chip beside the other contexts' kernels
inline bool frame_fold_pair_ok(const GemmDesc &d, const GemmDesc &d2) {
    if (env_int("BN_FRAMEPAIR", 0) != 1) return false;
    if (!d.fold || !frame_fold_shape_ok(d, nullptr)) return false;
    if (d.N > 128 || d.ldc != d.N) return false;
    const int wn = std::max(2, (d.N + 31) / 32);
    const int64_t span = (int64_t)(FRAME_BM_RULE - 1) * d.lda + d.fold_n;
    if ((size_t)(((span + 3) & ~3) + 2 * FRAME_BM_RULE * GEMM_LD_RULE + 2 * 32 * wn * GEMM_LD_RULE) * sizeof(float) > 160 * 1024) return false;
    // the second product's tiles must fit the two tile buffers: K tiles of the spectrum + one step of its filter rows
    const int n2pad = (d2.N + 31) / 32 * 32;
    if (wn * FRAME_BM_RULE * GEMM_LD_RULE + n2pad * GEMM_LD_RULE > 2 * FRAME_BM_RULE * GEMM_LD_RULE + 2 * 32 * wn * GEMM_LD_RULE) return false;
    if (d2.N > 32 * wn || d2.N < 1 || d2.K != d.N || d2.lda != d2.K || d2.rows != d.rows || d2.a_bs != d.c_bs) return false;
    if (d2.fold || d2.has_scale || d2.has_res || d2.se_inline) return false;
    bool stages = stft_act_supported(d2.act);  // the compact stage functions only
    for (int q = 0; q < d2.npost && q < 4; q++) stages = stages && stft_act_supported(d2.post_act[q]);
    return stages && d2.npost <= 4;
}

// ---- LDS-DMA GEMM (gemm_dma.hip) ----------------------------------------------------------------------------------
// epilogue activations the kernel carries (one dispatch on the launch-uniform code); other codes keep the older kernels
BN_HD inline bool gemm_dma_act_ok(int act) {
    return act == ACT_NONE || act == ACT_RELU || act == ACT_CLIP || act == ACT_SILU || act == ACT_HSWISH || act == ACT_SIGMOID || act == ACT_HSIGMOID;
}

// floats of LDS a (tile, K-slice count, ring depth) configuration needs for this layer
inline size_t gemm_dma_lds_bytes(const GemmDesc &d, int mtw, int ntw, int wm, int wn, int ks, int depth, int se_cr = 0) {
    const int tr = 16 * mtw * wm, bn = 16 * ntw * wn;
    const int gate_floats = d.has_scale ? (d.K + 1023) / 1024 * 1024 : 0;  // whole 1-KiB pieces
    const int se_floats = d.se_inline ? ((d.K + 3) & ~3) + ((se_cr + 3) & ~3) : 0;  // squeeze means + hidden units
    return std::max((size_t)(ks * depth * (tr + bn) * 32 + gate_floats + se_floats), (size_t)((ks - 1) * wm * wn * mtw * ntw * 256)) * sizeof(float);
}

// which block family the LDS-DMA kernel would take for this GEMM (0 = not eligible; 1 / 2: one tile per block, 64- / 48-row tiles;
// 3: the streaming form): per-sample quantities only
inline int gemm_dma_shape(const GemmDesc &d) {
    const int mode = env_int("BN_GEMMDMA", 1);
    if (mode == 0) return 0;
    if (d.fold || d.npost || d.out_strided || d.lda != d.K || d.K % 16 || d.K < 32 || d.N % 4 || d.N < 16 || !gemm_dma_act_ok(d.act)) return 0;
    if (d.ldc % 4 || d.c_bs % 4 || d.a_bs % 4 || (d.has_res && (d.ldr % 4 || d.r_bs % 4)) || (d.has_scale && d.s_bs % 4)) return 0;
    if ((int64_t)d.rows * d.K >= ((int64_t)1 << 30) || (int64_t)d.N * d.K >= ((int64_t)1 << 30)) return 0;  // 32-bit lane offsets
    if (d.has_scale && d.K > 8192) return 0;
    // Where it pays (measured, batch 32 and 128, tools/kernel_table.py): deep products with few output channels -- the
    // project convs and the head conv.  Short-K, wide-N expands are bound by their output stores and their launch, not by
    // staging: the tiled kernel keeps them (mode 2 sends every eligible shape here, for tests).
    // ... and the project convs of the big feature maps (few output channels, K 32 .. 144): memory-bound either way, but
    // the tiled kernel spends 17 - 27 vector instructions per matrix instruction on them, and at four contexts every
    // vector instruction is taken from the budget the other contexts' matrix work needs (BN_GEMMDMA_SMALLN=0 keeps
    // them on the tiled kernel)
    const bool small_n = d.N <= 32 && d.has_scale && env_int("BN_GEMMDMA_SMALLN", 1) != 0;
    // Round 4: the expand convs of the late stages (K 64 .. 127 here, N >= 128, no gate, no residual) CAN take the streaming form of the
    // kernel (gemm_dma_stream_kernel: a block walks consecutive row tiles, its ring never drains) -- family 3.  Opt-in (BN_GEMMSTREAM=1):
    // measured EQUAL to the tiled kernel on these shapes (BirdNET v3.0, K = 112 -> N = 672: 37.4 - 38.0 against 37.5 us at batch 64, 125
    // against 124 us at batch 256, i.e. 79 TF/s either way; K = 80 -> 480: 76 against 82 us at batch 256, 26.0 against 25.6 at 64) --
    // removing the per-tile prologue does not move a launch that already runs at the rate every f32-MFMA kernel of this library
    // reaches at saturation.  The default plan is unchanged (and so are its bits).
    if (env_int("BN_GEMMSTREAM", 0) != 0 && !d.has_scale && !d.has_res && !d.se_inline && d.K >= 64 && d.K < 128 && d.N >= 128 && d.rows % 32 == 0) return 3;
    if (mode != 2 && d.K < 128 && !small_n) return 0;
    if (d.rows % 32 == 0) return 1;  // 64- or 32-row tiles x up to 128 channels, waves along the rows
    if (d.rows % 48 == 0) return 2;  // 48-row tiles x 32 / 64 / 128 channels, waves along the channels
    return 0;
}

// K slices per block: a property of the layer's SHAPE (it enters the summation order).  Deep products (project convs:
// K 240 .. 1152, head conv) run as two interleaved slices -- unless two slices of the LARGEST tile the launcher may pick
// for this layer would not fit the LDS: the tile follows the batch, so "fall back to one slice for the big tile only"
// would make a segment's bits depend on its batch (gated layers with K > 3072).
inline int gemm_dma_kslices(const GemmDesc &d, int se_cr = 0) {
    const int force = env_int("BN_GEMMDMA_KS", 0);
    int ks = (force == 1 || force == 2) ? force : (d.K >= 192 ? 2 : 1);
    if (ks == 2) {
        const size_t cap = 156 * 1024;
        const bool fits = d.rows % 32 == 0 ? gemm_dma_lds_bytes(d, 1, 8, 4, 1, 2, 3, se_cr) <= cap   // 64 x 128, the widest shape-1 tile
                                           : gemm_dma_lds_bytes(d, 3, 2, 1, 4, 2, 3, se_cr) <= cap;  // 48 x 128
        if (!fits) ks = 1;
    }
    return ks;
}

// ---- the same GEMMs on the bf16 matrix pipe with f32-complete products (gemm_dma3.hip, round 5) ------------------------------------
// bytes of LDS of a (tile, K-slice count, ring depth) configuration: activations f32 (128 B per row and stage), weights three bf16 planes
// (192 B per row and stage)
inline size_t gemm_dma3_lds_bytes(const GemmDesc &d, int mtw, int ntw, int wm, int wn, int ks, int depth, int se_cr = 0) {
    const int tr = 16 * mtw * wm, bn = 16 * ntw * wn;
    const int gate_floats = d.has_scale ? (d.K + 1023) / 1024 * 1024 : 0;
    const int se_floats = d.se_inline ? ((d.K + 3) & ~3) + ((se_cr + 3) & ~3) : 0;
    return std::max((size_t)ks * depth * ((size_t)tr * 128 + (size_t)bn * 192) + (size_t)(gate_floats + se_floats) * 4,
                    (size_t)(ks - 1) * wm * wn * mtw * ntw * 256 * sizeof(float));
}
// K slices of the layer (they enter the summation order: a property of the SHAPE, never of the batch): deep products run as two
// interleaved slices through rings of TWO stages each (three would not fit the widest tile), unless even that does not fit
inline int gemm_dma3_kslices(const GemmDesc &d, int se_cr = 0) {
    const int force = env_int("BN_GEMM3_KS", 0);
    int ks = (force == 1 || force == 2) ? force : (d.K >= 192 ? 2 : 1);
    if (ks == 2) {
        const size_t cap = 156 * 1024;
        const bool fits = d.rows % 32 == 0 ? gemm_dma3_lds_bytes(d, 1, 8, 4, 1, 2, 2, se_cr) <= cap : gemm_dma3_lds_bytes(d, 3, 2, 1, 4, 2, 2, se_cr) <= cap;
        if (!fits) ks = 1;
    }
    return ks;
}
// which GEMMs take the form: every one-tile-per-block LDS-DMA shape (BN_GEMM3=0 keeps the exact-f32 kernel)
inline bool gemm_dma3_wanted(const GemmDesc &d) {
    if (env_int("BN_GEMM3", 1) == 0 || d.w3) return false;
    const int shape = gemm_dma_shape(d);
    return shape == 1 || shape == 2;
}
// x = hi + mid + lo exactly, each the f32 of a bf16 number: hi = x with its low 16 bits cleared, mid the same of x - hi, lo the rest
inline void split_bf16x3(float x, uint16_t &hi, uint16_t &mid, uint16_t &lo) {
    auto top = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); u &= 0xffff0000u; float r; std::memcpy(&r, &u, 4); return r; };
    auto bits = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return (uint16_t)(u >> 16); };
    const float h = top(x), r1 = x - h, m = top(r1), r2 = r1 - m;
    hi = bits(h); mid = bits(m); lo = bits(r2);
}
// The weights [N][K] (K contiguous) as three bf16 planes [plane][N][Kp], Kp = K rounded up to 32, returned in a vector of floats (a bit
// container: 3 N Kp / 2 of them).  Inside every 32-deep step position 8 q + j holds k offset 4 q + j (j < 4) or 16 + 4 q + (j - 4): the
// order in which a lane's two 16-byte activation reads (chunks q and 4 + q of the stage row) deliver its eight values.  K % 32 == 16:
// the kernel's last stage re-reads columns K-32 .. K-1 and multiplies its first half as zeros, so the last step holds column
// K - 16 + 4 q + (j - 4) at position 8 q + j (j >= 4) and zeros at j < 4.
inline std::vector<float> pack_w3(const float *W, int64_t N, int64_t K) {
    const int64_t Kp = (K + 31) & ~(int64_t)31, nfs = K / 32;
    std::vector<uint16_t> img((size_t)(3 * N * Kp), 0);
    for (int64_t n = 0; n < N; n++)
        for (int64_t s = 0; s < Kp / 32; s++)
            for (int64_t q = 0; q < 4; q++)
                for (int64_t j = 0; j < 8; j++) {
                    int64_t k;
                    if (s < nfs) k = 32 * s + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
                    else k = j < 4 ? -1 : K - 16 + 4 * q + (j - 4);
                    if (k < 0 || k >= K) continue;
                    uint16_t h, m, l;
                    split_bf16x3(W[n * K + k], h, m, l);
                    const size_t at = (size_t)(n * Kp + 32 * s + 8 * q + j);
                    img[at] = h;
                    img[(size_t)(N * Kp) + at] = m;
                    img[(size_t)(2 * N * Kp) + at] = l;
                }
    std::vector<float> out((img.size() + 1) / 2, 0.0f);
    std::memcpy(out.data(), img.data(), img.size() * sizeof(uint16_t));
    return out;
}

// ---- ... register-staged (gemm_b3.hip).  Its own shape rule: the kernel walks the batch's row matrix in 8-float chunks and has no
// staging constraint on K or on the rows of a sample, so it also takes what the LDS-DMA kernels leave to the tiled one -- the expand
// convs of the late stages (K 80 .. 232, N 480 .. 1392; Perch's K = 136 / 232 are not multiples of 16).  Not where the squeeze-excite
// products ride in the GEMM's prologue (gemm_dma3_kernel carries those), not below four channel tiles: one wave per 16-channel tile
// gives a block too few waves there, and those layers (the project convs of the large maps, N = 16 .. 48) are bound by their activation
// traffic, not by the product.  Per-sample quantities only.
inline bool gemm_b3_shape_ok(const GemmDesc &d) {
    if (env_int("BN_GEMM3", 2) < 2) return false;
    if (d.fold || d.npost || d.out_strided || d.se_inline || d.K % 8 || d.K < 32 || d.N % 4 || d.N <= 48 || !gemm_dma_act_ok(d.act)) return false;
    if (d.lda % 4 || d.lda < d.K || d.ldc % 4 || d.c_bs % 4 || d.a_bs % 4 || (d.has_res && (d.ldr % 4 || d.r_bs % 4)) || (d.has_scale && d.s_bs % 4)) return false;
    if (d.gap && !(gemm_gap_shape_ok(d) && d.rows == 48)) return false;
    if ((int64_t)d.N * d.K >= ((int64_t)1 << 30)) return false;
    if (d.rows < 8) return false;  // one row per sample (the FC heads): a GEMV bound by its weight bytes -- 4 B per weight beat 6, the split-K kernel keeps it
    // the expand convs only where the product is big enough to matter; tiny ones keep their kernels (and their bits)
    return d.K >= 64 || d.has_scale;
}
// The weights [N][K] as three bf16 planes in FRAGMENT order: [16-channel tile][K step][plane][lane][8 bf16] -- lane (c, q) of the tile's
// wave holds k = 32 step + 8 q .. + 7 of channel 16 tile + c, i.e. one coalesced 1-KiB load per (tile, step, plane).  Channels past N
// and k past K -- including up to three whole padding steps -- are zeros.  Returned in a vector of floats (a bit container).
inline std::vector<float> pack_w3f(const float *W, int64_t N, int64_t K) {
    const int64_t nt16 = (N + 15) / 16, nst = ((K + 31) / 32 + 3) & ~(int64_t)3;  // a multiple of FOUR K steps (the kernel's loop is unrolled by its prefetch depth, unguarded)
    std::vector<uint16_t> img((size_t)(nt16 * nst * 3 * 64 * 8), 0);
    for (int64_t t = 0; t < nt16; t++)
        for (int64_t s = 0; s < nst; s++)
            for (int64_t lane = 0; lane < 64; lane++) {
                const int64_t n = 16 * t + (lane & 15), q = lane >> 4;
                if (n >= N) continue;
                for (int64_t j = 0; j < 8; j++) {
                    const int64_t k = 32 * s + 8 * q + j;
                    if (k >= K) continue;
                    uint16_t h, m, l;
                    split_bf16x3(W[n * K + k], h, m, l);
                    const size_t at = (size_t)((((t * nst + s) * 3) * 64 + lane) * 8 + j);
                    img[at] = h;
                    img[at + 64 * 8] = m;
                    img[at + 2 * 64 * 8] = l;
                }
            }
    std::vector<float> out(img.size() / 2, 0.0f);
    std::memcpy(out.data(), img.data(), img.size() * sizeof(uint16_t));
    return out;
}

// the largest channel count whose excite products a GEMM block computes for itself (BN_SEGEMM_MAXC; 0 = never)
inline int gemm_dma_se_max_channels() { return env_int("BN_SEGEMM_MAXC", 768); }

// ---- tiled fused MBConv (kernels.hip) -----------------------------------------------------------------------------
// dynamic LDS of mbconv_expand_dw_kernel for this shape
inline size_t mbconv_lds_bytes(const MbDesc &d) {
    if (d.whole_map) {
        const int mt = (d.H * d.W + 31) / 32, ks = (d.Cin + 7) / 8 * 8 + 4;
        return (size_t)(32 * ks + mt * 32 * 32 + 8 * 32) * sizeof(float);
    }
    const int toh = d.s == 1 ? 8 : 4, tow = d.s == 1 ? 16 : 8;
    const int hp = ((toh - 1) * d.s + d.k) * ((tow - 1) * d.s + d.k);
    const int mp = (hp + 31) / 32 * 32;
    const int ks = (d.Cin + 7) / 8 * 8 + 4;
    const int nchunks = (d.C + 31) / 32;
    return (size_t)(mp * ks + mp * 32 + mp + nchunks * 8 * 32) * sizeof(float);
}
// dynamic LDS of the pipelined variant (Es double-buffered)
inline size_t mbconv_pipe_lds_bytes(const MbDesc &d) {
    const int toh = d.s == 1 ? 8 : 4, tow = d.s == 1 ? 16 : 8;
    const int hp = ((toh - 1) * d.s + d.k) * ((tow - 1) * d.s + d.k);
    const int mp = (hp + 31) / 32 * 32;
    const int ks = (d.Cin + 7) / 8 * 8 + 4;
    const int nchunks = (d.C + 31) / 32;
    return (size_t)(mp * ks + 2 * mp * 32 + mp + nchunks * 8 * 32) * sizeof(float);
}

// ---- small-map MBConv (mbmap.hip) ---------------------------------------------------------------------------------
BN_HD constexpr int mm_kib(int floats) { return (floats + 255) & ~255; }  // LDS-DMA writes whole 1-KiB pieces

// LDS of the <MW, NW, WM, WN, KSP> block for this layer: input image + two filter chunks + the expanded chunk image + partial sums
inline size_t mbmap_lds_bytes(const MbDesc &d, int mw, int nw, int wm, int wn, int ksp = 1) {
    const int hw = 16 * mw * wm, nc = 16 * nw * wn, ng = 64 * wm * wn * ksp / nc;
    return (size_t)(mm_kib(hw * d.Cin) + 2 * mm_kib(nc * d.Cin) + mm_kib(d.H * (d.W + d.k - 1) * (nc + 4)) + ng * nc) * sizeof(float);
}

// Which configuration takes this block (cfg 0 = none).  Per-sample quantities only.  BN_MBMAP2=0 disables.
//   1: 192-pixel map (6 x 32), chunks of 64 channels, 8 waves      2: the same, chunks of 32 (wider inputs)
//   3: 48-pixel map (3 x 16), chunks of 64 channels, 8 waves (two K slices)
//   4: 64-pixel map (4 x 16), chunks of 32 channels, 8 waves (two K slices; Cin % 64 == 0)
//   5: 256-pixel map (8 x 32) as TWO BANDS of 4 output rows (2 at stride 2), each band a block of its own that loads the 6 input rows
//      its outputs reach (the band's halo rows are real rows of the map: nothing is padded or recomputed except the expand
//      of the 4 rows the two bands share); chunks of 32 channels, 8 waves            (round 4: BirdNET v3.0's 8 x 32 stage)
//   6: 64-pixel map (4 x 16), chunks of 32 channels, 4 waves, for Cin % 64 in {16, 48}
// Round 4 also lets every configuration run TRANSPOSED (tr: the kernel's rows are the map's columns -- Perch's maps are 32 x 8
// and 16 x 4) and with the input rows PADDED in LDS to whole 16-wide k groups (cin_pad > Cin: the missing chunks are read from
// a page of zeros, the planner pads the filter rows; Perch's Cin = 136 / 232).
// MEASURED (round 4, one box per comparison, four contexts): correct (op tests + the full-size models against the oracle) and a
// shorter launch chain -- BirdNET v3.0 at batch 64: 1766 against 1810 us of launches, Perch at batch 128: 7036 against 7125 -- but
// LOWER throughput where those models run saturated: v3.0 48.2 k against 49.0 k segments/s, Perch 19.55 k (all), 19.78 k (cfg 6
// only) against 20.13 k.  A band expands 6 rows for 4 (1.5 x the expand work of the layer), and even without recompute (cfg 6)
// the fused block's marginal cost per batch is above GEMM + whole-map depthwise, as round 1 found for its whole-map kernel.
// Which kernel a layer takes may not depend on the batch (a segment's bits must not), so the round-4 configurations are
// OPT-IN: BN_MBMAP3=1 (BN_MBMAP_BANDS=0 then keeps the banded one off).  Default: round 3's set (cfg 1-4, plain).
struct MbmapShape {
    int cfg = 0;
    int bands = 1;    // blocks per (sample, channel group) along the map's rows; the squeeze sums are partial per band
    int cin_pad = 0;  // floats per input / filter row in LDS (multiple of 16)
    int tr = 0;
};
inline MbmapShape mbmap_shape(const MbDesc &d) {
    MbmapShape none, sh;
    if (env_int("BN_MBMAP2", 1) == 0) return none;
    const bool r4 = env_int("BN_MBMAP3", 0) != 0;
    if (d.k1 > 0 || !((d.k == 3 || d.k == 5) && (d.s == 1 || d.s == 2))) return none;
    if (d.Cin % 4 || d.Cin < 16 || d.C % 4 || d.in_bs % 4) return none;
    const int pad = (d.k - 1) / 2;  // the kernels are compiled for symmetric "same" padding
    if (d.pt != pad || d.pl != pad || d.OH != (d.H + 2 * pad - d.k) / d.s + 1 || d.OW != (d.W + 2 * pad - d.k) / d.s + 1) return none;
    if (!mbconv_row_act_supported(d.act1) || !mbconv_row_act_supported(d.act2)) return none;
    // the kernel's geometry: R rows x Wc columns, Wc in {32, 16}; a tall narrow map is walked transposed
    int R = d.H, Wc = d.W;
    // round 5: a 32 x 8 map is walked transposed by DEFAULT where the banded wave-specialised kernel serves it (Perch: K = 96 / 136 -> 3 / 5
    // steps of 32; any K % 16 == 0 -- its fragment reads happen once per block, so no padded k for the bank pattern)
    const int nst_ws = (d.Cin + 31) / 32;
    const bool ws_on = env_int("BN_MBMAP_WS", 1) != 0 && env_int("BN_MBMAP_WS_BANDS", 1) != 0 && env_int("BN_MBMAP_B3", 1) != 0 && env_int("BN_GEMM3", 2) != 0;
    const bool ws_tr = !r4 && ws_on && d.W == 8 && d.H == 32 && d.Cin % 8 == 0 && nst_ws >= 3 && nst_ws <= 5 && env_int("BN_MBMAP_WS_TR", 1) != 0;
    // ... and a 16 x 4 map with K in eight steps of 32 (Perch: K = 232) by its one-pixel-tile-per-wave form
    const bool ws_deep = !r4 && ws_on && d.W == 4 && d.H == 16 && d.Cin % 8 == 0 && d.Cin % 64 != 0 && nst_ws == 8 && d.s == 1 &&
                         env_int("BN_MBMAP_WS_DEEP", 1) != 0;
    if ((r4 && d.W < 16 && (d.H == 32 || d.H == 16)) || ws_tr || (ws_deep && d.W == 4)) { R = d.W; Wc = d.H; sh.tr = 1; }
    if (Wc % 4) return none;
    sh.cin_pad = (d.Cin + 15) & ~15;
    if (!r4 && !ws_tr && !ws_deep && sh.cin_pad != d.Cin) return none;
    MbDesc p = d;
    p.Cin = sh.cin_pad;  // LDS sizes follow the padded rows
    p.H = R; p.W = Wc;
    const size_t cap = 160 * 1024;
    const int cls = sh.cin_pad % 64;
    const bool c1648 = cls == 16 || cls == 48;
    // round 5: the 8 x 32 map in two bands is taken by DEFAULT where the wave-specialised kernel serves it (mbmap_ws.hip: not transposed,
    // 3 or 4 steps of 32, no padded k) -- the exact-f32 banded form stays opt-in
    const bool ws_bands = !r4 && ws_on && (ws_tr || (!sh.tr && d.Cin % 16 == 0 && (nst_ws == 3 || nst_ws == 4)));
    // (the row swizzle each configuration is compiled with: see mm_swz)
    if (R == 6 && Wc == 32 && c1648) {
        if (mbmap_lds_bytes(p, 3, 2, 4, 2) <= cap) sh.cfg = 1;
        else if (mbmap_lds_bytes(p, 3, 1, 4, 2) <= cap) sh.cfg = 2;
    } else if (R == 3 && Wc == 16 && cls == 0) {
        if (mbmap_lds_bytes(p, 3, 1, 1, 4, 2) <= cap) sh.cfg = 3;
    } else if (R == 4 && Wc == 16 && cls == 0 && d.s == 1) {  // BirdNET v3.0's last stage (5 s segments: one more row than v2.4's 3 x 16)
        if (mbmap_lds_bytes(p, 2, 1, 2, 2, 2) <= cap) sh.cfg = 4;  // 32-channel chunks (Cin = 192: two filter chunks of 64 would not fit), eight waves
    } else if (ws_deep && R == 4 && Wc == 16) {
        sh.cfg = 6;  // (mbmap_ws.hip; no exact-f32 counterpart by default)
    } else if (r4 && R == 4 && Wc == 16 && c1648 && d.s == 1) {
        if (mbmap_lds_bytes(p, 2, 1, 2, 2, 1) <= cap) sh.cfg = 6;
    } else if (ws_bands && R == 8 && Wc == 32) {
        sh.cfg = 5; sh.bands = 2;  // (mbmap_ws.hip: its LDS carve-up fits for every K it is compiled for; no padded k)
    } else if (r4 && R == 8 && Wc == 32 && env_int("BN_MBMAP_BANDS", 1) != 0) {
        // rows whose length is 0 or 32 mod 64 floats (Perch: Cin = 96) would meet the fragment reads' bank pattern: one more k group
        // of zeros moves them into a class the compiled swizzle serves (96 -> 112: a sixth more expand work, still ahead of the
        // GEMM + depthwise pair it replaces)
        if (!c1648) sh.cin_pad += 16;
        p.Cin = sh.cin_pad;
        p.H = 6;  // a band's rows
        if ((sh.cin_pad % 64 == 16 || sh.cin_pad % 64 == 48) && mbmap_lds_bytes(p, 3, 1, 4, 2) <= cap) { sh.cfg = 5; sh.bands = 2; }
    }
    return sh.cfg ? sh : none;
}
inline int mbmap_config(const MbDesc &d) { return mbmap_shape(d).cfg; }

// The small-map kernel's expand on the bf16 matrix pipe (mbmap.hip, NSW > 0): 32-deep k steps per wave, or 0 where the f32 form stays.
// Compiled: cfg 1 with 2 / 3 steps (Cin <= 96), cfg 2 with 3 / 4 (Cin <= 128), cfg 3 / 4 (two K slices) with 2 / 3 / 4 per slice
// (Cin = 128 / 192 / 256).  BN_MBMAP_B3=0 (or BN_GEMM3=0) keeps the exact-f32 instruction.
inline int mbmap_b3_steps(const MbDesc &d, const MbmapShape &sh) {
    if (!sh.cfg || sh.cfg > 4 || sh.tr || sh.bands != 1 || sh.cin_pad != d.Cin || d.Cin % 8) return 0;
    if (env_int("BN_MBMAP_B3", 1) == 0 || env_int("BN_GEMM3", 2) == 0) return 0;
    const int nst = (d.Cin + 31) / 32, ksp = sh.cfg >= 3 ? 2 : 1;
    if (nst % ksp) return 0;
    const int nsw = nst / ksp;
    const bool ok = sh.cfg == 1 ? (nsw == 2 || nsw == 3) : sh.cfg == 2 ? (nsw == 3 || nsw == 4) : (nsw >= 2 && nsw <= 4);
    return ok ? nsw : 0;
}
// the expand filters [C][K] in the order that form's filter chunk has in LDS: [tile of 16 channels][32-deep step][h][q][c][4 floats] holds
// channel 16 tile + c, k = 32 step + 16 h + 4 q .. + 3 (the k order of the kernel's input fragments; zeros past C and K; tiles padded to whole 64-channel chunks), so a chunk is ONE dense
// block of memory for the global -> LDS copy and a wave's fragment of a step is two lane-linear 16-byte reads
inline std::vector<float> pack_mbmap_w3f(const float *w, int64_t C, int64_t K) {
    const int64_t nst = (K + 31) / 32, tiles = (C + 63) / 64 * 4;
    std::vector<float> out((size_t)(tiles * nst * 512), 0.0f);
    for (int64_t t = 0; t < tiles; t++)
        for (int64_t st = 0; st < nst; st++)
            for (int64_t h = 0; h < 2; h++)
                for (int64_t q = 0; q < 4; q++)
                    for (int64_t c = 0; c < 16; c++)
                        for (int64_t j = 0; j < 4; j++) {
                            const int64_t row = 16 * t + c, k = 32 * st + 16 * h + 4 * q + j;
                            if (row < C && k < K) out[(size_t)((((t * nst + st) * 128) + h * 64 + q * 16 + c) * 4 + j)] = w[row * K + k];
                        }
    return out;
}
// ... and for the 6 x 32 maps (cfg 1 / 2) the wave-specialised kernel (mbmap_ws.hip: expand of chunk p and depthwise of chunk p - 1 in the
// same phase, chunks of 32 channels): 32-deep steps of the whole product, or 0.  BN_MBMAP_WS=0 keeps mbmap.hip; BN_MBMAP_WS_SMALL=0 keeps it for
// the 3 x 16 / 4 x 16 maps only.
inline int mbmap_ws_steps(const MbDesc &d, const MbmapShape &sh) {
    if (env_int("BN_MBMAP_WS", 1) == 0 || env_int("BN_MBMAP_B3", 1) == 0 || env_int("BN_GEMM3", 2) == 0) return 0;
    const int nst = (d.Cin + 31) / 32;
    if (sh.cfg == 6)  // 4 x 16 (Perch: transposed 16 x 4) with eight steps: one pixel tile per expand wave
        return (sh.bands == 1 && sh.cin_pad % 16 == 0 && nst == 8 && d.s == 1 && d.Cin % 64 != 0 && env_int("BN_MBMAP_WS_DEEP", 1) != 0) ? nst : 0;
    if (sh.cfg == 5)  // 8 x 32 in two bands of six rows: the 6 x 32 kernel per band
        return (sh.bands == 2 && (sh.tr || sh.cin_pad == d.Cin) && sh.cin_pad % 16 == 0 && nst >= 3 && nst <= (sh.tr ? 5 : 4) && env_int("BN_MBMAP_WS_BANDS", 1) != 0 &&
                (!sh.tr || env_int("BN_MBMAP_WS_TR", 1) != 0)) ? nst : 0;
    if (mbmap_b3_steps(d, sh) == 0) return 0;
    if (sh.cfg <= 2) return nst;                                               // 6 x 32: 2 .. 4 steps (mbmap_b3_steps)
    return (env_int("BN_MBMAP_WS_SMALL", 1) != 0 && (nst == 4 || nst == 6)) ? nst : 0;  // 3 x 16 / 4 x 16: Cin = 128 / 192
}
// the same filters as three bf16 planes for mbmap_ws.hip, whose expand waves share their SIMD's vector ALU with the depthwise waves and
// should not spend it on splitting filters: [tile][step][plane hi | mid | lo][lane (q, c)][8 bf16], element e of lane (q, c) = k
// 32 step + 16 (e / 4) + 4 q + e % 4 -- a fragment is three lane-linear 16-byte reads.  Returned in floats (a bit container).
inline std::vector<float> pack_mbmap_w3p(const float *w, int64_t C, int64_t K) {
    const int64_t nst = (K + 31) / 32, tiles = (C + 63) / 64 * 4;
    std::vector<uint16_t> img((size_t)(tiles * nst * 3 * 64 * 8), 0);
    for (int64_t t = 0; t < tiles; t++)
        for (int64_t st = 0; st < nst; st++)
            for (int64_t q = 0; q < 4; q++)
                for (int64_t c = 0; c < 16; c++)
                    for (int64_t e = 0; e < 8; e++) {
                        const int64_t row = 16 * t + c, k = 32 * st + 16 * (e / 4) + 4 * q + e % 4;
                        if (row >= C || k >= K) continue;
                        uint16_t pl[3];
                        split_bf16x3(w[row * K + k], pl[0], pl[1], pl[2]);
                        for (int64_t pp = 0; pp < 3; pp++) img[(size_t)(((((t * nst + st) * 3 + pp) * 64) + q * 16 + c) * 8 + e)] = pl[pp];
                    }
    std::vector<float> out(img.size() / 2);
    std::memcpy(out.data(), img.data(), img.size() * 2);
    return out;
}
inline size_t mbmap_lds_bytes_b3(const MbDesc &d, int nst, int nw, int wm, int wn, int ksp) {  // d in the kernel's geometry (H, W, k)
    // two filter buffers + chunk image + squeeze partials; the input image of the prologue lies over everything behind the first buffer
    const int nc = 16 * nw * wn, ng = 64 * wm * wn * ksp / nc, wsz = mm_kib(nc * 32 * nst);
    const int ring = 2 * wsz + mm_kib(d.H * (d.W + d.k - 1) * (nc + 4)) + ng * nc, pro = wsz + mm_kib(d.H * d.W * d.Cin);
    return (size_t)(ring > pro ? ring : pro) * sizeof(float);
}

// ---- FFT front end (stft.hip) -------------------------------------------------------------------------------------
// LDS carve-up (floats), shared by the kernel and stft_lds_bytes.  Table regions are whole KiB: the asynchronous
// global -> LDS copies write 1 KiB per wave instruction.
struct StftLds {
    int sig, tw, wbuf, window, otab, mstart, ment, spec, mel, total;
};
BN_HD constexpr int stft_wbuf_slots(int slots) { return slots + slots / 8; }
BN_HD inline int stft_kib(int floats) { return (floats + 255) & ~255; }
BN_HD inline StftLds stft_layout(const FftDesc &d, int nw, int slots) {
    StftLds l;
    int o = 0;
    l.sig = o; o += ((d.tpb - 1) * d.hop + d.L + 3) & ~3;
    l.wbuf = o; o += 2 * nw * stft_wbuf_slots(slots);
    l.tw = o; o += stft_kib(2 * d.tw_count);
    l.window = o; o += stft_kib(d.L);
    l.otab = o; o += stft_kib(d.otab_planar ? (d.power ? 10 : 6) * d.nout : (d.otab_stride > 0 ? d.otab_stride : 8) * d.nout);
    const bool csr = d.nmel && d.mel_mode == 0;  // (MFMA mode reads its tiles from global memory / L2: nothing of the bank in LDS)
    l.mstart = o; o += csr ? stft_kib(d.nmel + 1) : 0;
    l.ment = o; o += csr ? stft_kib(2 * d.mel_nnz) : 0;  // (column, value) pairs
    l.spec = o; o += d.nmel ? d.tpb * (d.spec_stride > 0 ? d.spec_stride : d.nout) : 0;  // the tile's spectrum rows (mel fusion only)
    l.mel = o;
    l.total = o;
    return l;
}
// 16 waves x 512 slots where a frame fits 512 complex points (opt-in BN_STFT_NW=16: measured slower -- 128-register cap at
// 16 waves: 156 B of scratch per lane; 74.6 vs 69.1 us), else 8 x 1024
inline bool stft_wide(const FftDesc &d) {
    if (env_int("BN_STFT_NW", 8) != 16) return false;
    return d.M <= 512 && d.tpb % (512 / d.M) == 0;
}
inline size_t stft_lds_bytes(const FftDesc &d, int /*nwaves*/) {
    return (size_t)(stft_wide(d) ? stft_layout(d, 16, 512) : stft_layout(d, 8, 1024)).total * sizeof(float);
}

}  // namespace bn
