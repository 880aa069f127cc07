// Launch descriptors shared by the planner (engine.cpp, host C++) and the HIP
// kernels (kernels.hip, topk.hip).  All tensors are f32.  "Batch" is always the
// outermost dimension and is the only run-time-variable size.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdlib>

namespace bn {

// Epilogue / unary function codes.
enum Act : int32_t {
    ACT_NONE = 0,
    ACT_RELU,
    ACT_CLIP,      // p0 = lo, p1 = hi
    ACT_SIGMOID,
    ACT_SILU,      // x * sigmoid(x)
    ACT_HSIGMOID,  // max(0, min(1, p0*x + p1))
    ACT_HSWISH,    // x * max(0, min(1, x/6 + 0.5))
    ACT_LEAKY,     // x >= 0 ? x : p0*x
    ACT_TANH,
    ACT_EXP,
    ACT_LOG,
    ACT_SQRT,
    ACT_ABS,
    ACT_NEG,
    ACT_RECIP,
    ACT_POW,       // pow(x, p0)
    ACT_AFFINE,    // p0*x + p1
    ACT_MAXC,      // max(x, p0)
    ACT_MINC,      // min(x, p0)
    ACT_RSUB,      // p0 - x
    ACT_RDIV,      // p0 / x
    ACT_SQUARE,
    ACT_FLOOR,
    ACT_CEIL,
    ACT_ERF,
    ACT_SOFTPLUS,
    // (round 5) comparison results are f32 0.0 / 1.0 ("bool" tensors of the ONNX graph); NaN compares false like IEEE / ORT
    ACT_GTC,       // x > p0 ? 1 : 0
    ACT_LTC,       // x < p0 ? 1 : 0
    ACT_GEC,       // x >= p0 ? 1 : 0
    ACT_LEC,       // x <= p0 ? 1 : 0
    ACT_EQC,       // x == p0 ? 1 : 0
    ACT_NEZ,       // x != 0 ? 1 : 0   (Cast to bool)
    ACT_TRUNC,     // toward zero       (Cast to an integer type; values stay f32)
    ACT_ROUND,     // to nearest, ties to even (ONNX Round)
};

enum BinOp : int32_t { BIN_NONE = 0, BIN_ADD, BIN_SUB, BIN_MUL, BIN_DIV, BIN_POW, BIN_MAX, BIN_MIN,
                       // (round 5) comparisons -> 0.0 / 1.0; the two halves of Where(cond, a, b) = SELA(a, cond) + SELB(b, cond)
                       BIN_GT, BIN_LT, BIN_GE, BIN_LE, BIN_EQ, BIN_NE,
                       BIN_SELA,   // b != 0 ? a : +0
                       BIN_SELB }; // b != 0 ? +0 : a
enum RedOp : int32_t { RED_SUM = 0, RED_MEAN, RED_MAX, RED_MIN, RED_PROD, RED_L2, RED_SUMSQ };

constexpr int ELT_MAX_DIMS = 5;  // per-sample loop dims (batch is separate)

// Elementwise chain over a per-sample index space of nd dims (outermost first):
//   v = a[b, i...];  for each stage s: v = act_s(bin_s(v, operand_s[b, i...]));  out[b, i...] = v
// Strides are in elements; 0 broadcasts.  Chains are built at plan time by fusing consecutive
// elementwise launches (normalisation, power-law compression, BatchNorm, layout copies).
constexpr int ELT_MAX_STAGES = 4;
struct EltStage {
    int32_t bin;  // BinOp (BIN_NONE => no second operand)
    int32_t act;  // applied after bin
    float p0, p1;
    int64_t sb[ELT_MAX_DIMS];
    int64_t bb;   // batch stride of the operand (0 for constants)
    int32_t bsq;  // 1: the operand is squared before the binary op (|z|^2 = re*re + im*im in one launch)
    int32_t reserved;
};
struct EltDesc {
    int32_t nd;
    int64_t size[ELT_MAX_DIMS];
    int64_t so[ELT_MAX_DIMS], sa[ELT_MAX_DIMS];
    int64_t bo, ba;      // batch strides
    int64_t per_sample;  // product of size[]
    int32_t nstages;
    EltStage st[ELT_MAX_STAGES];
};

// Generic reduction: kept dims (<=4, per sample) x reduced dims (<=3).
struct ReduceDesc {
    int32_t nk, nr;
    int64_t ksize[4], kin[4], kout[4];
    int64_t rsize[3], rin[3];
    int64_t bi, bo;  // batch strides
    int64_t kept, red;  // products
    int32_t op;
    int32_t inner_kept;  // 1: some kept dim has input stride 1 (threads map to kept index)
    // pair = 1: a min reduction and a max reduction over the same contiguous chunks run as ONE pass over the input
    // (op is RED_MIN and lands in out, the max lands in out2 with batch stride bo2); planner rule, chunk stage only
    int32_t pair;
    int64_t bo2;
};

// C[r, n] = act(sum_k A[r, k] * W[n, k] + bias[n]) (+ res[r, n]) where row
// r = b * rows + m addresses A at b*a_bs + m*lda (rows may overlap: conv1d
// framing) and C / res at b*c_bs + m*ldc.  W is [N][K], K contiguous.
struct GemmDesc {
    int64_t rows;  // per sample
    int32_t K, N;
    int64_t lda, a_bs, ldc, c_bs, ldr, r_bs;
    int32_t act;
    float p0, p1;
    int32_t has_bias, has_res;
    int32_t has_scale;    // A is multiplied by scale[b, k] on load (squeeze-excite gate)
    int64_t s_bs;         // batch stride of scale
    // Folded framing rows (filters that are symmetric / antisymmetric about their centre, e.g. windowed DFT
    // bases): fold = +1 / -1, fold_n = the filter length.  Column c < K = fold_n / 2 of the operand is then
    // x[1 + c] + fold * x[fold_n - 1 - c] of the frame, and W holds the first-half taps 1 .. fold_n/2.
    // fold = 2 (round 4, "quarter fold"; frame_fold2_kernel): a bank of WINDOWED COSINES whose window is symmetric about the frame centre.
    // With y = window * x, ye[n] = y[n] + y[L-n]:  cos(2 pi k (L/2 - n) / L) = (-1)^k cos(2 pi k n / L), so the EVEN bins need only
    // S[n] = ye[n] + ye[L/2-n] and the odd ones D[n] = ye[n] - ye[L/2-n], n < L/4 (+ the lone tap n = L/4 for the even bins): a
    // quarter of the filter length per output.  W is [N][K] with K = fold_n / 4 + 32 (pure cosines x the row's amplitude, zero past
    // tap L/4), the first fold_ne columns are the even-bin group (a multiple of 32), the rest the odd-bin group; the launch takes the
    // window tables and the column -> output channel map as two more operands (launch_gemm_fold2).
    int32_t fold, fold_n;
    int32_t fold_ne;
    // fold == 2: W is packed in fragment order for the half-height kernel (frame_fold2p_kernel: [column group of 32][step][8-wide k group]
    // [lane][4 floats], lane (lr, lh) holding W[32 group + lr][32 step + 8 g + 4 lh .. + 3]); decided by the planner from per-sample
    // quantities (frame_fold2p_ok): two such blocks fit a CU's LDS
    int32_t fold_wpk;
    // Absorbed elementwise chain (planner rule E): unary stages applied after act, and the consumer's output view.
    int32_t npost, out_strided;
    int32_t post_act[4];
    float post_p0[4], post_p1[4];
    int64_t out_rs, out_cs;  // element (m, n) of sample b is stored at C + b*c_bs + m*out_rs + n*out_cs
    // squeeze-excite computed in the GEMM's own prologue (planner rule I, gemm_dma.hip): the gate is not read from
    // memory, every block derives it from the squeeze partial sums of its sample
    int32_t se_inline;
    // (round 4, planner: fuse_gap_into_gemm) the mean over ALL rows of a sample is what leaves the launch: C is [batch][N] (c_bs = N) and
    // holds  sum_m act(row m) / rows.  Only the LDS-DMA kernel's 48-row tiles with rows == 48 (one block sees every row of the sample for its
    // channels, one wave every row of its channels): the head conv of v2.4 in front of its GlobalAveragePool.
    int32_t gap;
    // (round 5, planner: pack_bf16x3_weights) W is the layer's weights as three bf16 planes [plane][N][Kp] (Kp = K rounded up to 32, k
    // permuted inside every 32-deep step; plan_rules.h, pack_w3): the launch takes gemm_dma3_kernel -- f32-complete products from six
    // bf16 matrix instructions per 32-deep k (gemm_dma3.hip).  Only where gemm_dma_shape is 1 or 2; there is no other kernel for this form.
    // w3 == 2: the planes in FRAGMENT order (pack_w3f) for the register-staged kernel of the same arithmetic (gemm_b3.hip).
    int32_t w3;
};
inline bool gemm_gap_shape_ok(const GemmDesc &d) { return d.rows == 48 && !d.has_res && !d.npost && !d.out_strided && !d.fold; }

// Direct NHWC convolution, weights [kh][kw][cin/groups][cout].
struct ConvDesc {
    int32_t H, W, Cin, OH, OW, Cout;
    int32_t kh, kw, sh, sw, pt, pl, dh, dw, groups;
    int32_t act;
    float p0, p1;
    int32_t has_bias, has_res;
    int64_t in_bs, out_bs;
};

// Depthwise NHWC convolution, weights [kh][kw][C]; optionally accumulates the
// per-(sample, channel) sum of the activated output (squeeze of squeeze-excite).
struct DwDesc {
    int32_t H, W, C, OH, OW;
    int32_t kh, kw, sh, sw, pt, pl, dh, dw;
    int32_t act;
    float p0, p1;
    int32_t has_bias;
    int64_t in_bs, out_bs;
    // tiled kernel (kernel 3/5, stride 1/2, no dilation, C % 4 == 0): each lane produces `tw`
    // adjacent output pixels of 4 channels; a block is (C/4) x rpb lanes = rpb pixel tiles.
    int32_t tiled, tw, rpb, nblk;
    // squeeze of a following squeeze-excite: per-block channel sums -> gap[b][nblk][C]
    int32_t has_gap;
    int64_t gap_bs;
    // (round 5) tiled == 2: take the instance with the map size at compile time (dwconv_mapt_kernel) where one exists -- the planner's
    // choice (BN_DWMAPT), same values as dwconv_map_kernel
    int32_t mapt;
};

// Fused MBConv front half: expand 1x1 conv (+bias+act) -> depthwise KxK (+bias+act) [+ SE squeeze],
// the expanded tensor lives only in LDS.  x is NHWC [H][W][Cin]; expand weights [C][Cin];
// depthwise weights [k][k][C]; output NHWC [OH][OW][C].
struct MbDesc {
    int32_t H, W, Cin, C, OH, OW;
    int32_t k, s, pt, pl;
    int32_t act1, act2;
    float p0_1, p1_1, p0_2, p1_2;
    int32_t has_bias1, has_bias2;
    int64_t in_bs, out_bs;
    int32_t tiles_x, tiles_y;  // output tiles of (8x16 at stride 1, 4x8 at stride 2)
    int32_t whole_map;         // 1: small feature map, one block = (32 mid channels, whole map); w1 is [C][Cin]; 2: mbmap.hip (input resident in LDS)
    // first conv is a k1 x k1 convolution with few input channels (stem) instead of a 1x1 expand: the halo tile is
    // staged as im2col rows, Cin = k1*k1*Cin1 (column order (ky, kx, c)); H, W are its OUTPUT map (what the
    // depthwise conv reads), H1 x W1 x Cin1 the image it reads with stride s1 and padding (pt1, pl1).  k1 == 0: 1x1.
    int32_t k1, s1, pt1, pl1, H1, W1, Cin1;
    int32_t has_gap;
    int64_t gap_bs;
    // row-streaming form (mbrow.hip): tiles_x = strips of mbconv_row_outw(k, s) output columns, tiles_y = bands of
    // toh output rows; 0 = the tiled kernels above
    int32_t row_mode, toh;
    // row-streaming form, transposed: the kernel's rows are the map's COLUMNS (strips then run across the map's height).  For
    // maps that are tall and narrow (Perch: 125 x 32, 63 x 16) a 30-column strip is half empty while the height fills
    // several; H/W, OH/OW, pt/pl, tiles_x/tiles_y and toh stay in MAP terms in the plan, the launcher swaps them for the kernel
    int32_t row_tr;
    int32_t dbg;  // mbmap.hip experiments (BN_MM_DBG bit mask, set by the launcher): skip phases to time the rest
    // mbmap.hip, round 4 (plan_rules.h, MbmapShape; all 0 for the other kernels): bands of the map a sample is cut into (one block
    // each, squeeze sums partial per band: tiles_y = bands), the kernel's rows are the map's columns, floats per input / filter row
    // in LDS (Cin rounded up to 16: the planner pads w1's rows to it, the missing input chunks are read from a page of zeros)
    int32_t map_bands, map_tr, cin_pad;
    // (round 5) row-streaming form: the expand conv on the bf16 matrix pipe with f32-complete products (bf16x3.h): 1x1 expands with
    // Cin % 8 == 0; decided by the planner (BN_MBROW_B3, BN_GEMM3), part of the block's arithmetic like any kernel choice
    int32_t row_b3;
    int32_t map_ws;    // 6 x 32 maps: the wave-specialised kernel (mbmap_ws.hip) with this many 32-deep steps, 0 = mbmap.hip
    int32_t map_b3;    // small-map kernel: 32-deep bf16x3 steps per wave (plan_rules.h mbmap_b3_steps), 0 = exact-f32 expand
};
// MaxPool / AveragePool over an NHWC tensor (1-D pooling = H == 1).
struct PoolDesc {
    int32_t H, W, C, OH, OW;
    int32_t kh, kw, sh, sw, pt, pl;
    int32_t is_max;             // 1 = MaxPool, 0 = AveragePool
    int32_t count_include_pad;  // AveragePool: divide by kh*kw instead of the number of in-image taps
    int64_t in_bs, out_bs;
};
void launch_pool(hipStream_t s, const PoolDesc &d, float *out, const float *in, int64_t batch);

// Row-streaming form (mbrow.hip): which blocks it takes (planner and launcher agree through these), outputs per strip
inline bool mbconv_row_act_supported(int act) { return act == ACT_NONE || act == ACT_RELU || act == ACT_CLIP || act == ACT_SILU || act == ACT_HSWISH; }
inline bool mbconv_row_supported(const MbDesc &d) {
    const int ng = (d.Cin + 7) / 8;
    if (d.whole_map) return false;
    if (!((d.k == 3 || d.k == 5) && (d.s == 1 || d.s == 2))) return false;
    if (!mbconv_row_act_supported(d.act1) || !mbconv_row_act_supported(d.act2)) return false;
    if ((int64_t)d.OH * d.OW * d.C >= ((int64_t)1 << 31) || (int64_t)d.W * d.Cin >= ((int64_t)1 << 30)) return false;  // 32-bit lane offsets
    if (d.row_tr && ((int64_t)d.H * d.W * d.Cin >= ((int64_t)1 << 30) || d.k1 > 0 || d.k != 3)) return false;  // transposed: a column step is a map row; 3 x 3 instances only
    if (d.k1 > 0) return ng >= 2 && ng <= 4 && d.k == 3 && d.k1 <= 4 && d.Cin1 >= 1;
    return d.Cin % 4 == 0 && ng >= 2 && ng <= 6;
}
inline int mbconv_row_outw(int k, int s) { return (32 - k) / s + 1; }
void launch_mbconv_row(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2,
                       const float *b2, float *gap, int64_t batch);
// whole-map form with the input resident in LDS (mbmap.hip): configuration this block takes (0 = none; per-sample
// quantities only), and the launch (false = not eligible, nothing launched)
bool launch_mbmap_ws(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                     float *gap, int64_t batch, int nch);
bool launch_mbmap(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1, const float *b1, const float *w2, const float *b2,
                  float *gap, int64_t batch);
struct SeTail;
void launch_mbconv(hipStream_t s, const MbDesc &d, float *out, const float *in, const float *w1,
                   const float *b1, const float *w2, const float *b2, float *gap, int64_t batch, const SeTail *tail = nullptr);

// Squeeze-excite, stage 1: per-(sample, split) channel sums of an NHWC tensor [HW][C]
// -> partial [splits][C].  Stage 2 (SeFcDesc) sums the splits in a fixed order (deterministic).
struct GapDesc {
    int64_t HW;
    int32_t C, splits;
    int64_t in_bs, out_bs;
};
// Squeeze-excite, stage 2, one workgroup per sample:
//   s = mean over HW (from the partial sums); h = act1(W1 s + b1); gate = act2(W2 h + b2)
// W1 is [Cr][C], W2 is [C][Cr].
struct SeFcDesc {
    int32_t C, Cr, splits;
    float inv_hw;
    int32_t act1, act2;
    float p0_1, p1_1, p0_2, p1_2;
    int64_t in_bs, out_bs;
};

// Squeeze-excite inside the consuming GEMM (GemmDesc::se_inline): the excite shapes and operands
struct SeInline {
    SeFcDesc se;
    const float *partial, *w1, *b1, *w2t, *b2;  // squeeze partial sums [batch][splits][C]; W1 [Cr][C]; W2 transposed [Cr][C]
};

// Squeeze-excite finished INSIDE the kernel that produced the squeeze sums (depthwise / fused MBConv launches): every
// block publishes its partial sums write-through, takes a ticket on a per-sample counter, and the block that draws
// the last ticket of its sample reduces the partials and runs both excite products -- no separate launch, no
// release fence (the sums are the only hand-off and they are stored with sc1 / read with sc1, MI355X guide
// "Valid forms", counter row).  counter[b] must be 0 on entry; the last block resets it.  se.splits = blocks per
// sample of the producing launch.
struct SeTail {
    int32_t on;
    SeFcDesc se;
    const float *w1, *b1, *w2t, *b2;
    float *gate;
    uint32_t *counter;  // one word per sample, stride cnt_bs words
    int64_t cnt_bs;
    int32_t nblocks;    // blocks per sample of the hosting launch (set by the launcher)
};
// dynamic LDS the tail needs (floats): s[C] | h[Cr] | 1024 scratch
inline size_t se_tail_lds_bytes(const SeFcDesc &d) { return (size_t)(d.C + d.Cr + 2048 + 8) * sizeof(float); }
void launch_gap_partial(hipStream_t s, const GapDesc &d, float *partial, const float *in, int64_t batch);
// hidden: scratch [batch][Cr]
void launch_se_fc(hipStream_t s, const SeFcDesc &d, float *gate, float *hidden, const float *partial,
                  const float *w1, const float *b1, const float *w2, const float *b2, int64_t batch);

void launch_eltwise(hipStream_t s, const EltDesc &d, float *out, const float *a,
                    const float *const (&b)[ELT_MAX_STAGES], int64_t batch);
void launch_reduce(hipStream_t s, const ReduceDesc &d, float *out, const float *in, int64_t batch, float *out2 = nullptr);
// Which of the two GEMM kernels runs is decided from per-sample quantities only, so that the
// summation order of every output element -- and with it the result bits -- does not depend on
// how many segments share a batch (a shard's last, shorter batch matches the single-GPU run).
// Per-sample scalar chain applied to the signal while a framing GEMM loads its span into LDS (planner rule G "pre", the same rule that fills
// FftDesc::npre): up to four stages  v = act(bin(v, the sample's scalar)) -- v2.4's min-max normalisation of the segment, which then is never
// written.  sc[k]: operand of stage k (batch stride bb[k] elements; unused where bin[k] is BIN_NONE); the plan keeps the operands in PlanOp::eb.
struct FramePre {
    int32_t n, bin[ELT_MAX_STAGES], act[ELT_MAX_STAGES];
    float p0[ELT_MAX_STAGES], p1[ELT_MAX_STAGES];
    int64_t bb[ELT_MAX_STAGES];
    const float *sc[ELT_MAX_STAGES];
};

// (host logic shared by the launcher and the planner)
inline bool gemm_use_splitk(const GemmDesc &d) {
    static const int min_k = getenv("BN_SPLITK_MINK") ? atoi(getenv("BN_SPLITK_MINK")) : 256;
    static const int max_rows = getenv("BN_SPLITK_MAXROWS") ? atoi(getenv("BN_SPLITK_MAXROWS")) : 256;
    // deep K, few output tiles per sample (measured: pays below ~8 tiles of 128x32 per sample);
    // the split-K kernel has no K-tail step
    const double tiles = (double)d.rows / 128 * ((d.N + 31) / 32);
    return !d.fold && d.K >= min_k && d.rows <= max_rows && d.K % 32 == 0 && tiles < 8.0;
}
// true when the launch would run a kernel variant that supports npost / out_strided
inline bool gemm_accepts_post(const GemmDesc &d) { return !d.fold && !d.has_scale && !d.has_res && !gemm_use_splitk(d); }
// Folded framing GEMM with the product over its rows fused behind it (planner rule J; kernels.hip, frame_fold_kernel<true>)
void launch_gemm_fold_pair(hipStream_t s, const GemmDesc &d, const GemmDesc &d2, float *C2, const float *A, const float *W, const float *bias,
                           const float *W2, const float *bias2, int64_t batch);
void launch_gemm(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W,
                 const float *bias, const float *res, const float *scale, int64_t batch, const FramePre *pre = nullptr);
// quarter-folded framing GEMM (GemmDesc::fold == 2): wtab = [2][K] window tables (wa | wb), colmap = [N] output channel of every
// column (-1: padding), bias indexed by output channel.  The planner emits it only where frame_fold2_shape_ok (plan_rules.h) holds; a
// launch outside that is an error (there is no other kernel for this operand form).
void launch_gemm_fold2(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias, const float *wtab,
                       const int32_t *colmap, int64_t batch, const FramePre *pre = nullptr);
// LDS-DMA GEMM (gemm_dma.hip): 0 = not eligible, 1 = 64-row tiles, 2 = 48-row tiles (per-sample quantities only);
// launch_gemm_dma returns false (nothing launched) when the shape or a pointer's alignment rules it out.  BN_GEMMDMA=0 disables.
bool launch_gemm_dma(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W, const float *bias, const float *res,
                     const float *scale, int64_t batch, const SeInline *se = nullptr);
// the same launches with GemmDesc::w3 (gemm_dma3.hip): W3 = pack_w3's image of the layer's weights
bool launch_gemm_dma3(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W3, const float *bias, const float *res,
                      const float *scale, int64_t batch, const SeInline *se = nullptr);
// GemmDesc::w3 == 2 (gemm_b3.hip): W3F = pack_w3f's fragment-order image; operands through registers, one wave per 16-channel tile
bool launch_gemm_b3(hipStream_t s, const GemmDesc &d, float *C, const float *A, const float *W3F, const float *bias, const float *res,
                    const float *scale, int64_t batch);
// the largest channel count whose excite products a GEMM block computes for itself (BN_SEGEMM_MAXC; 0 = never)
void launch_conv(hipStream_t s, const ConvDesc &d, float *out, const float *in, const float *w,
                 const float *bias, const float *res, int64_t batch);
void launch_dwconv(hipStream_t s, const DwDesc &d, float *out, const float *in, const float *w,
                   const float *bias, float *gap, int64_t batch, const SeTail *tail = nullptr);

// top-K + sigmoid + filter + stable sort, bit-exact with the reference's
// BinaryHeap semantics (topk.hip).  idx/conf/count are device buffers with row
// stride k_stride.
// `flags` is a device scratch of `rows` uint32 (may be NULL: exact kernel only).
void launch_topk(hipStream_t s, const float *logits, int64_t rows, int64_t n, int64_t k,
                 int has_min, float min_conf, int64_t k_stride, uint32_t *idx, float *conf,
                 uint32_t *count, uint32_t *flags);
// LDS bytes the top-K kernel needs for (n, k); 0 if it cannot run (too large).
// recording -> f32 windows (chunk_audio + i16/32768 conversion); S % 4 == 0
void launch_windows(hipStream_t s, float *dst, const void *src, int32_t is_i16, uint64_t n_samples, uint64_t first_start, uint64_t step, uint32_t S,
                    uint32_t count);
// polyphase FIR resampler: table [L][T], output n reads phase (n*M)%L at source position (n*M)/L
void launch_resample(hipStream_t s, float *dst, const void *src, int32_t is_i16, const float *table, uint64_t n_src, uint64_t n_dst, uint32_t L, uint32_t M,
                     uint32_t T);
// Windowed-DFT filter bank as a real FFT (stft.hip).  Frames of L samples (L = 2 M, M a power of two in 64..1024) at
// `hop`, per sample `frames` of them; a wave transforms F = 1024 / M frames at a time in place in LDS: `npass`
// strided decimation-in-frequency passes (sub-problem size pass_n, radix pass_r, twiddles at tw + pass_tw), then
// 16-point blocks in registers.  Output c (< nout) = otab[c] . (Re Z[k], Im Z[k], Re Z[M-k], Im Z[M-k]) read at
// two precomputed buffer positions.  Optional stages around it: a per-sample scalar elementwise chain on the signal
// (npre), a sparse mel filter bank (nmel rows in CSR form) with the GEMM-style post chain and strided store.
struct FftDesc {
    int32_t L, M, logM, hop, frames, nout;
    int32_t tpb, F;  // frames per block, frames per wave pass
    int32_t npass, pass_n[4], pass_r[4], pass_tw[4], tw_count;
    int64_t a_bs;
    int32_t a_vec4;  // set by the launcher
    int32_t dbg;     // experiments (BN_STFT_DBG bit mask, set by the launcher): skip phases to time the rest
    int32_t npre, pre_bin[ELT_MAX_STAGES], pre_act[ELT_MAX_STAGES];
    float pre_p0[ELT_MAX_STAGES], pre_p1[ELT_MAX_STAGES];
    int64_t pre_bb[ELT_MAX_STAGES];
    int64_t ldc, c_bs;
    int32_t has_bias;
    int32_t nmel, mel_nnz, mel_has_bias, mel_act;
    float mel_p0, mel_p1;
    int32_t npost, post_act[4];
    float post_p0[4], post_p1[4];
    int64_t out_rs, out_cs;  // mel element (t, m) of sample b is stored at out + b*c_bs + t*out_rs + m*out_cs
    // power spectrum mode (the planner folds  re^2 + im^2 [-> sqrt]  behind a cos | sin bank into the launch): output c is
    // f(u^2 + v^2) with u, v the two linear forms of bin c (otab row: positions, 4 coefficients of u, 4 of v); 0 = off,
    // 1 = u^2 + v^2, 2 = sqrt(u^2 + v^2).  otab_stride: floats per otab row (8, or 12 in power mode)
    int32_t power, otab_stride;
    // mel filter bank on the matrix cores (mel_mode 1, tpb == 16): the bank is cut into tiles of 16 bands x 16 bins, only
    // the tiles that hold a non-zero are kept (triangular filters: a diagonal stripe), each stored in the fragment order of
    // v_mfma_f32_16x16x4_f32; mstart = [ceil(nmel/16) + 1 tile-row starts | bin-group index of every kept tile] (as floats),
    // mcol = the kept tiles, 256 floats each.  spec_stride: floats per spectrum row in LDS (nout in CSR mode; in MFMA mode
    // 16 ceil(nout/16) + 8: whole bin groups, and = 8 mod 16 makes the ds_read_b128 fragment reads conflict free)
    int32_t mel_mode, mel_groups, spec_stride;
    int32_t otab_planar;  // 1: the untangle table is stored as planes [nout][4] | [nout][2] | [nout][4 (power mode)] instead of rows
    // (round 5, planner: absorb_pad_into_fft) the frames are cut from the ZERO-PADDED signal [pad_l zeros | in_len samples | zeros]: the
    // span load reads sample p - pad_l for padded position p and 0 outside [0, in_len); in_len == 0: no padding (a_bs >= the span)
    int32_t pad_l, in_len;
};
struct StftPtrs {
    float *out;
    const float *in;
    const float *window;  // [L]
    const float2 *tw;     // [tw_count]
    const float *otab;    // [nout][otab_stride]: position of Z[k], position of Z[M-k] (as floats), 4 coefficients, 2 pad (power mode: + 4 coefficients)
    const float *bias;    // [nout] or NULL
    const float *pre[ELT_MAX_STAGES];
    const float *mstart, *mcol, *mval, *mel_bias;  // CSR of the mel filter bank: row starts, (column, value) pairs in mcol (indices stored as floats); mval unused
};
// stage codes stft_kernel implements (a compact subset: no libm bodies); the planner absorbs only chains made of these
inline bool stft_act_supported(int act) { return act != ACT_TANH && act != ACT_ERF && act != ACT_SOFTPLUS && act != ACT_HSIGMOID && act != ACT_HSWISH && act < ACT_GTC; }
inline bool stft_bin_supported(int bin) { return bin != BIN_POW && bin < BIN_GT; }
void launch_stft(hipStream_t s, const FftDesc &d, const StftPtrs &p, int64_t batch);

void launch_null(hipStream_t s);  // empty kernel (timing calibration)

// A launcher that is handed a layout its kernel cannot take (a caller's device pointer without the required
// alignment, a descriptor the planner should never have produced) launches NOTHING and records a message for the
// calling thread instead of aborting the host process; the C ABI turns it into BN_ERR_INVALID_ARG / BN_ERR_BACKEND.
void launch_error(const char *msg);
// message of the first launch_error since the last call on this thread (NULL if none); clears it
const char *take_launch_error();
// Per-device preparation (kernels.hip): opts every kernel that may use more than 64 KB of dynamic LDS in on `dev`
// and caches the device's CU count, ONCE, outside any stream capture.  The C ABI calls it wherever a device is first
// used (bn_model_load, bn_ctx_create, the stand-alone top-K entry points); launchers never call the runtime for it.
bool prepare_device(int dev);
void register_dynamic_lds_kernel(const void *kernel);  // used by each .hip file's register_*_kernels()
// The device the calling thread's launches go to: noted (thread-local, no runtime call) by the C ABI wherever it selects a device,
// so that the two launch-time questions below are answered for THAT device (hipFuncSetAttribute is per device: ADVICE r3).
void note_launch_device(int dev);
// launch-time check only (no runtime call): true when `bytes` fits and the thread's launch device was prepared
bool ensure_dynamic_lds(const void *kernel, size_t bytes);
int device_cu_count();
void device_context_count_add(int dev, int delta);  // capi.cpp: a context was created (+1) / destroyed (-1) on the device
int device_context_count();                        // live contexts on the launching thread's device (>= 1; bn_set_sharing_mode overrides: 1 / >= 2)
void device_sharing_mode(int mode);
const float *device_zero_page();  // 4 KiB of zeros on the thread's launch device (allocated by prepare_device)
size_t topk_lds_bytes(int64_t n, int64_t k);

// device -> pinned host memory by a kernel's own stores (topk.hip): up to three regions of 32-bit words per launch;
// dst are the DEVICE addresses of pinned host buffers (hipHostGetDevicePointer)
struct CopyOut {
    void *dst[3];
    const void *src[3];
    uint32_t words[3];
    int n;
};
void launch_copy_out(hipStream_t s, const CopyOut &c);
void launch_copy_dev(hipStream_t s, void *dst, const void *src, uint64_t bytes);  // device -> device, bytes % 4 == 0

}  // namespace bn
