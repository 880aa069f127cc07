// Helpers shared by the small-map MBConv kernels (mbmap.hip, mbmap_ws.hip): LDS image swizzles, the dense global -> LDS copies, the
// activation dispatch.  Included inside namespace bn { namespace { ... } } of each translation unit.
#pragma once
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MM_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define MM_GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))


// Row swizzle of the [rows][Cin] LDS images (input map, filter chunk).  A ds_read_b128 fragment read touches 16 rows at
// one k position; with a row stride of Cin floats they fall on 64 / gcd(64, Cin mod 64) ... banks: 2-way conflicts for
// Cin = 80 / 112, 8-way for Cin = 192 (measured: 45 - 83 % of the LDS cycles of the first version were conflicts).  The
// LDS-DMA destination is lane-linear, so the permutation goes on the SOURCE address (16-byte chunk c of row r is
// stored in slot c ^ swz(r)) and on the read:
//   SWZ16 = false (Cin mod 64 in {16, 48}):  swz(r) = 3 ((r >> 3) & 1)      -- inside each 64-byte k group
//   SWZ16 = true  (Cin mod 64 == 0):         swz(r) = r & 15                 -- inside each 256-byte block of 4 k groups
// both conflict free for the lane groups of ds_read_b128 (MI355X guide, LDS table).
template <bool SWZ16>
__device__ __forceinline__ int mm_swz(int r) { return SWZ16 ? (r & 15) : 3 * ((r >> 3) & 1); }

// dense global -> LDS copy of `rows` rows of CH 16-byte chunks in 1-KiB pieces, piece p by wave p % NWAVES, with the
// row swizzle; inv_ch = floor(2^32 / CH) + 1 (slot -> row by one multiply-high); lanes past the end re-read the last
// chunk into the region's padding (every region is padded to whole pieces)
template <int NWAVES, bool SWZ16>
__device__ __forceinline__ void mm_copy(float *lds_dst, const float *gsrc, int rows, int CH, uint32_t inv_ch, int wave, int lane) {
    const int n16 = rows * CH;
    for (int c0 = wave * 64; c0 < n16; c0 += NWAVES * 64) {
        int sl = c0 + lane;
        sl = sl < n16 ? sl : n16 - 1;
        const int r = (int)__umulhi((uint32_t)sl, inv_ch);
        const int c = sl - r * CH;
        const int src = r * CH + (c ^ mm_swz<SWZ16>(r));
        __builtin_amdgcn_global_load_lds(MM_GLB_PTR(gsrc + 4 * src), MM_LDS_PTR(lds_dst + 4 * c0), 16, 0, 0);
    }
}

// every real map row the outputs of band b reach lies inside the H rows the band's block loads (first row gy0, clamped into the map)
constexpr bool mm_bands_ok(int K, int S, int H, int HM, int NB) {
    const int PT = (K - 1) / 2, OHM = (HM + 2 * PT - K) / S + 1, OH = OHM / NB;
    for (int b = 0; b < NB; b++) {
        const int first = b * OH * S - PT, last = (b * OH + OH - 1) * S + K - 1 - PT;
        const int lo = first < 0 ? 0 : first, hi = last > HM - 1 ? HM - 1 : last;
        int gy0 = NB > 1 ? first : 0;
        gy0 = gy0 < 0 ? 0 : (gy0 > HM - H ? HM - H : gy0);
        if (lo < gy0 || hi >= gy0 + H) return false;
    }
    return true;
}

// the input image of a block: rows = pixels of the band (row-major in the KERNEL's geometry, W pixels per row), CHP chunks per LDS
// row of which the first CHS exist in memory; transposed maps gather (kernel pixel (r, c) is map pixel (y = c, x = gy0 + r) of a
// map that is HM wide), padding chunks read the page of zeros
template <int NWAVES, bool SWZ16, int W, int HM>
__device__ __forceinline__ void mm_copy_in(float *lds_dst, const float *gsrc, const float *zpage, int rows, int CHP, int CHS, int Cin, uint32_t inv_ch,
                                           int gy0, int tr, int wave, int lane) {
    const int n16 = rows * CHP;
    for (int c0 = wave * 64; c0 < n16; c0 += NWAVES * 64) {
        int sl = c0 + lane;
        sl = sl < n16 ? sl : n16 - 1;
        const int r = (int)__umulhi((uint32_t)sl, inv_ch);
        const int c = sl - r * CHP;
        const int cl = c ^ mm_swz<SWZ16>(r);
        const int pr = r / W, pc = r - pr * W;  // W is a power of two
        const int pix = tr ? pc * HM + gy0 + pr : gy0 * W + r;
        const float *src = cl < CHS ? gsrc + (size_t)pix * (size_t)Cin + 4 * cl : zpage;
        __builtin_amdgcn_global_load_lds(MM_GLB_PTR(src), MM_LDS_PTR(lds_dst + 4 * c0), 16, 0, 0);
    }
}

// bf16x3 form (NSW > 0): the filter chunk is stored by the planner in FRAGMENT ORDER (plan_rules.h pack_mbmap_w3f) -- piece (tile, step)
// = 128 chunks of 16 bytes, [h][q][c]: channel 16 tile + c, k = 32 step + 16 h + 4 q .. + 3 -- so the copy of a chunk is one dense block
// (1 KiB per wave instruction, whole cache lines) and a wave's fragment of a step is two lane-linear, conflict-free ds_read_b128.  (A
// first version gathered the pieces from the [C][K] rows on the source side of the copy: 16 half-used lines per instruction, and the
// launch skeleton -- everything but expand and depthwise -- went from 7.6 to 13.6 us at batch 32.)
template <int NWAVES>
__device__ __forceinline__ void mm_copy_w3(float *lds_dst, const float *gsrc, int npieces, int wave, int lane) {
    for (int i = wave; i < 2 * npieces; i += NWAVES)
        __builtin_amdgcn_global_load_lds(MM_GLB_PTR(gsrc + i * 256 + lane * 4), MM_LDS_PTR(lds_dst + i * 256), 16, 0, 0);
}

// dense copy of `kib` KiB, piece i by wave i % NWAVES
template <int NWAVES>
__device__ __forceinline__ void mm_copy_lin(float *lds_dst, const float *gsrc, int kib, int wave, int lane) {
    for (int i = wave; i < kib; i += NWAVES)
        __builtin_amdgcn_global_load_lds(MM_GLB_PTR(gsrc + i * 256 + lane * 4), MM_LDS_PTR(lds_dst + i * 256), 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void mm_act(int act, float p0, float p1, float (&v)[N]) {
    if (act == ACT_RELU) map_array<N>(v, [](float x) { return fmaxf(x, 0.0f); });
    else if (act == ACT_CLIP) map_array<N>(v, [=](float x) { return fminf(fmaxf(x, p0), p1); });
    else if (act == ACT_SILU) map_array<N>(v, [](float x) { return x * net_sigmoid(x); });
    else if (act == ACT_HSWISH) map_array<N>(v, [](float x) { return x * fminf(fmaxf(x * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f); });
}

