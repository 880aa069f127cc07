// Phase timeline of the fused MBConv kernel on the benchmark model's five shapes (batch 32): the kernel is compiled
// with -DBN_MB_STAMPS, thread 0 of every block records the shader clock at each phase boundary.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBN_MB_STAMPS -Irust-birdnet-onnx_amd/csrc -Iinclude tools/mb_probe.cpp -o tools/mb_probe
#include "../rust-birdnet-onnx_amd/csrc/kernels.hip"
#include "../rust-birdnet-onnx_amd/csrc/mbrow.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace bn;

struct Shape { const char *name; int H, W, Cin, C, k, s; int k1; };

int main(int argc, char **argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 32;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    std::vector<Shape> shapes = {{"stem 96x512x2 -> 48x256x32 (im2col K=18) dw3 s1", 48, 256, 18, 32, 3, 1, 3},
                                 {"48x256x16 -> 96 dw3 s2", 48, 256, 16, 96, 3, 2, 0},
                                 {"24x128x24 -> 144 dw3 s1", 24, 128, 24, 144, 3, 1, 0},
                                 {"24x128x24 -> 144 dw5 s2", 24, 128, 24, 144, 5, 2, 0},
                                 {"12x64x40 -> 240 dw5 s1", 12, 64, 40, 240, 5, 1, 0}};
    for (size_t si = 0; si < shapes.size(); si++) {
        if (only >= 0 && (int)si != only) continue;
        const Shape &sh = shapes[si];
        MbDesc d{};
        d.H = sh.H; d.W = sh.W; d.Cin = sh.Cin; d.C = sh.C; d.k = sh.k; d.s = sh.s;
        d.pt = d.pl = (sh.k - 1) / 2;
        d.OH = (sh.H + sh.s - 1) / sh.s; d.OW = (sh.W + sh.s - 1) / sh.s;
        if (sh.s == 2) { d.pt = d.pl = (sh.k - 1) / 2 - 0; }
        d.act1 = ACT_RELU; d.act2 = ACT_RELU; d.has_bias1 = d.has_bias2 = 1;
        const int toh = sh.s == 1 ? 8 : 4, tow = sh.s == 1 ? 16 : 8;
        d.tiles_y = (d.OH + toh - 1) / toh; d.tiles_x = (d.OW + tow - 1) / tow;
        d.has_gap = 1; d.gap_bs = (int64_t)d.tiles_x * d.tiles_y * d.C;
        size_t in_elems;
        if (sh.k1) { d.k1 = 3; d.s1 = 2; d.pt1 = d.pl1 = 0; d.H1 = 2 * sh.H + 1; d.W1 = 2 * sh.W + 1; d.Cin1 = 2; in_elems = (size_t)d.H1 * d.W1 * d.Cin1; }
        else in_elems = (size_t)sh.H * sh.W * sh.Cin;
        d.in_bs = (int64_t)in_elems; d.out_bs = (int64_t)d.OH * d.OW * d.C;
        const int ng = (sh.Cin + 7) / 8;
        std::vector<float> hin(in_elems * batch), hw1((size_t)sh.C * ng * 8, 0.f), hb1(sh.C), hw2((size_t)sh.k * sh.k * sh.C), hb2(sh.C);
        unsigned r = 12345;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto &v : hin) v = rnd();
        for (int c = 0; c < sh.C; c++) for (int k = 0; k < sh.Cin; k++) hw1[(size_t)c * ng * 8 + k] = rnd();
        for (auto &v : hb1) v = rnd();
        for (auto &v : hw2) v = rnd();
        for (auto &v : hb2) v = rnd();
        float *din, *dout, *dw1, *db1, *dw2, *db2, *dgap;
        unsigned long long *dst;
        const size_t nblk = (size_t)d.tiles_x * d.tiles_y * batch;
        hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, (size_t)d.out_bs * batch * 4); hipMalloc(&dw1, hw1.size() * 4); hipMalloc(&db1, hb1.size() * 4);
        hipMalloc(&dw2, hw2.size() * 4); hipMalloc(&db2, hb2.size() * 4); hipMalloc(&dgap, (size_t)d.gap_bs * batch * 4); hipMalloc(&dst, nblk * 32 * 8);
        hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db1, hb1.data(), hb1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice);
        hipMemset(dst, 0, nblk * 32 * 8);
        hipStream_t st; hipStreamCreate(&st);
        setenv("BN_MBPIPE", "0", 1);
        unsigned long long *nullp = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(bn_mb_stamps), &nullp, sizeof(nullp));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; i++) launch_mbconv(st, d, dout, din, dw1, db1, dw2, db2, dgap, batch);
        hipEventRecord(e0, st);
        for (int i = 0; i < 20; i++) launch_mbconv(st, d, dout, din, dw1, db1, dw2, db2, dgap, batch);
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyToSymbol(HIP_SYMBOL(bn_mb_stamps), &dst, sizeof(dst));
        launch_mbconv(st, d, dout, din, dw1, db1, dw2, db2, dgap, batch);
        hipStreamSynchronize(st);
        if (const char *why = take_launch_error()) printf("launch refused: %s\n", why);
        std::vector<unsigned long long> hs(nblk * 32);
        hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost);
        const int nch = std::min(5, (sh.C + 31) / 32);
        unsigned long long w0 = ~0ull, w1 = 0;
        double ph[32] = {0}; double total = 0;
        for (size_t b = 0; b < nblk; b++) {
            const unsigned long long *s_ = &hs[b * 32];
            w0 = std::min(w0, s_[0]); w1 = std::max(w1, s_[31]);
            // shader-clock slots: [1]=start [2]=staging issued, per chunk: [3+4c]=barrier A, [4+4c]=expand done, [5+4c]=barrier B, [6+4c]=dw done, [31? no 30+1]
            ph[0] += (double)(s_[2] - s_[1]);
            ph[1] += (double)(s_[3] - s_[2]);
            for (int c = 0; c < nch; c++) {
                ph[2] += (double)(s_[4 + 4 * c] - s_[3 + 4 * c]);   // expand (thread 0's wave)
                ph[3] += (double)(s_[5 + 4 * c] - s_[4 + 4 * c]);   // wait at barrier B
                ph[4] += (double)(s_[6 + 4 * c] - s_[5 + 4 * c]);   // depthwise + store issue
                if (c + 1 < nch) ph[5] += (double)(s_[3 + 4 * (c + 1)] - s_[6 + 4 * c]);  // loads for next chunk + barrier A
            }
            total += (double)(s_[31 - 0] == 0 ? 0 : 0);
            total += (double)((sh.C + 31) / 32 <= 5 ? 0 : 0);
        }
        double blk = 0;
        for (size_t b = 0; b < nblk; b++) blk += (double)(hs[b * 32 + 31] - hs[b * 32 + 0]) * 10.0;  // ns (100 MHz wall clock)
        printf("%-50s %7.1f us/launch | grid %zu blocks, lds %zu B | kernel span %.1f us, mean block residency %.1f us => ~%.1f blocks/CU in flight\n", sh.name, ms * 1000 / 20, nblk,
               mbconv_lds_bytes(d), (double)(w1 - w0) / 100.0, blk / nblk / 1000.0, blk / nblk / 1000.0 * nblk / ((double)(w1 - w0) / 100.0) / 256.0);
        printf("   cycles per block (mean, first %d chunks): staging issue %.0f | wait staging+barrier %.0f | expand %.0f | barrier B wait %.0f | depthwise %.0f | next-chunk loads+barrier A %.0f\n",
               nch, ph[0] / nblk, ph[1] / nblk, ph[2] / nblk, ph[3] / nblk, ph[4] / nblk, ph[5] / nblk);
        // ---- row-streaming kernel on the same operands: bit-identical output, its own squeeze tiling
        {
            MbDesc r = d;
            r.row_mode = 1 | ((getenv("DBG") ? atoi(getenv("DBG")) : 0) << 8);
            r.toh = getenv("TOH") ? atoi(getenv("TOH")) : 8;
            if (r.toh > d.OH) r.toh = d.OH;
            r.tiles_x = (d.OW + mbconv_row_outw(d.k, d.s) - 1) / mbconv_row_outw(d.k, d.s);
            r.tiles_y = (d.OH + r.toh - 1) / r.toh;
            r.gap_bs = (int64_t)r.tiles_x * r.tiles_y * d.C;
            float *dout2, *dgap2;
            hipMalloc(&dout2, (size_t)d.out_bs * batch * 4); hipMalloc(&dgap2, (size_t)r.gap_bs * batch * 4);
            hipMemset(dout2, 0xff, (size_t)d.out_bs * batch * 4);
            if (!mbconv_row_supported(r)) printf("   row kernel: shape not supported\n");
            else {
                const long long nun = (long long)batch * r.tiles_x * r.tiles_y * ((d.C + 31) / 32);
                unsigned long long *drs;
                hipMalloc(&drs, (size_t)nun * 64 * 8); hipMemset(drs, 0, (size_t)nun * 64 * 8);
                hipMemcpyToSymbol(HIP_SYMBOL(bn_row_stamps), &drs, sizeof(drs));
                launch_mbconv_row(st, r, dout2, din, dw1, db1, dw2, db2, dgap2, batch);
                hipStreamSynchronize(st);
                {
                    std::vector<unsigned long long> hs2((size_t)nun * 64);
                    hipMemcpy(hs2.data(), drs, hs2.size() * 8, hipMemcpyDeviceToHost);
                    // stamps: [0] unit start, then per main step: (top, after emit), last: end
                    double pro = 0, blk = 0, com = 0, tot = 0; long long nb = 0, nu2 = 0;
                    for (long long uu = 0; uu < nun; uu++) {
                        const unsigned long long *q = &hs2[uu * 64];
                        int n = 0; while (n < 64 && q[n]) n++;
                        if (n < 4) continue;
                        nu2++;
                        pro += (double)(q[1] - q[0]); tot += (double)(q[n - 1] - q[0]);
                        for (int k2 = 1; k2 + 2 < n; k2 += 2) { blk += (double)(q[k2 + 1] - q[k2]); com += (double)(q[k2 + 2] - q[k2 + 1]); nb++; }
                    }
                    printf("   row kernel stamps: unit lifetime %.0f cycles; prologue (setup + first K rows) %.0f; per main step: load+expand+emit %.0f, commit(+loop) %.0f  (%lld steps/unit)\n",
                           tot / nu2, pro / nu2, blk / nb, com / nb, nb / nu2);
                }
                unsigned long long *nullq = nullptr;
                hipMemcpyToSymbol(HIP_SYMBOL(bn_row_stamps), &nullq, sizeof(nullq));
                hipFree(drs);
                for (int i = 0; i < 3; i++) launch_mbconv_row(st, r, dout2, din, dw1, db1, dw2, db2, dgap2, batch);
                hipEventRecord(e0, st);
                for (int i = 0; i < 20; i++) launch_mbconv_row(st, r, dout2, din, dw1, db1, dw2, db2, dgap2, batch);
                hipEventRecord(e1, st);
                hipError_t err = hipStreamSynchronize(st);
                float ms2; hipEventElapsedTime(&ms2, e0, e1);
                std::vector<float> a((size_t)d.out_bs * batch), c((size_t)d.out_bs * batch);
                hipMemcpy(a.data(), dout, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dout2, c.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0, first = 0; double maxd = 0;
                for (size_t i = 0; i < a.size(); i++) if (memcmp(&a[i], &c[i], 4)) { if (!bad) first = i; bad++; maxd = std::max(maxd, (double)fabsf(a[i] - c[i])); }
                // squeeze sums: total over tiles per (sample, channel) agrees with the tiled kernel's within rounding
                std::vector<float> g1((size_t)d.gap_bs * batch), g2((size_t)r.gap_bs * batch);
                hipMemcpy(g1.data(), dgap, g1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(g2.data(), dgap2, g2.size() * 4, hipMemcpyDeviceToHost);
                double gmax = 0;
                for (int bb = 0; bb < batch; bb++) for (int c2 = 0; c2 < d.C; c2++) {
                    double s1 = 0, s2 = 0;
                    for (int t2 = 0; t2 < d.tiles_x * d.tiles_y; t2++) s1 += g1[(size_t)bb * d.gap_bs + (size_t)t2 * d.C + c2];
                    for (int t2 = 0; t2 < r.tiles_x * r.tiles_y; t2++) s2 += g2[(size_t)bb * r.gap_bs + (size_t)t2 * d.C + c2];
                    gmax = std::max(gmax, fabs(s1 - s2) / (fabs(s1) + 1.0));
                }
                printf("   row kernel (toh %d, %d x %d tiles, %lld units): %7.1f us/launch  [%s]  outputs differing from the tiled kernel: %zu of %zu (first %zu, max |d| %.3g), squeeze rel diff %.2g\n",
                       r.toh, r.tiles_x, r.tiles_y, (long long)batch * r.tiles_x * r.tiles_y * ((d.C + 31) / 32), ms2 * 1000 / 20, hipGetErrorString(err), bad, a.size(), first, maxd, gmax);
            }
            hipFree(dout2); hipFree(dgap2);
        }
        hipFree(din); hipFree(dout); hipFree(dw1); hipFree(db1); hipFree(dw2); hipFree(db2); hipFree(dgap); hipFree(dst);
    }
    return 0;
}
