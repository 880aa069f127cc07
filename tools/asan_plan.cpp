// Host-side sanitizer check of the ONNX reader + planner (no device needed):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
//       -Irust-birdnet-onnx_amd/csrc tools/asan_plan.cpp rust-birdnet-onnx_amd/csrc/{onnx_proto,engine,detect}.cpp -o /tmp/asan_plan
//   /tmp/asan_plan model.onnx [more.onnx ...]      (also feeds truncated / bit-flipped copies of each file)
// The kernel-side helpers the planner calls are restated here so that no HIP object is linked.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "engine.h"

namespace bn {
size_t mbconv_lds_bytes(const MbDesc &d) {
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
size_t topk_lds_bytes(int64_t, int64_t) { return 1; }
size_t stft_lds_bytes(const FftDesc &d, int nw) {  // stft.hip's carve-up, table regions rounded to whole KiB
    auto kib = [](int f) { return (f + 255) & ~255; };
    int o = (((d.tpb - 1) * d.hop + d.L + 3) & ~3) + 2 * nw * (1024 + 128) + kib(2 * d.tw_count) + kib(d.L) + kib(8 * d.nout);
    if (d.nmel) o += kib(d.nmel + 1) + kib(2 * d.mel_nnz) + d.tpb * d.nout;
    return (size_t)o * sizeof(float);
}
}  // namespace bn

static int plan_bytes(const std::vector<uint8_t> &bytes, bool quiet) {
    try {
        bn::OnnxModel m = bn::parse_onnx(bytes.data(), bytes.size());
        std::vector<int> all;
        for (size_t k = 0; k < m.outputs.size(); k++) all.push_back((int)k);
        auto p = bn::build_plan(m, all);
        if (!quiet) printf("  ok: %zu launches, arena %lld floats/sample\n", p->ops.size(), (long long)p->arena_elems);
        return 0;
    } catch (const std::exception &e) {
        if (!quiet) printf("  refused: %s\n", e.what());
        return 1;
    }
}

int main(int argc, char **argv) {
    for (int a = 1; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        printf("%s (%zu bytes)\n", argv[a], bytes.size());
        plan_bytes(bytes, false);
        // malformed variants must be refused (or planned) without touching memory they do not own
        int refused = 0, total = 0;
        unsigned rng = 12345;
        for (size_t cut : {(size_t)0, (size_t)1, bytes.size() / 7, bytes.size() / 3, bytes.size() / 2, bytes.size() - 1}) {
            std::vector<uint8_t> t(bytes.begin(), bytes.begin() + std::min(cut, bytes.size()));
            refused += plan_bytes(t, true);
            total++;
        }
        const int n_mut = getenv("ASAN_PLAN_MUTATIONS") ? atoi(getenv("ASAN_PLAN_MUTATIONS")) : 1500;
        for (int k = 0; k < n_mut; k++) {
            std::vector<uint8_t> t = bytes;
            for (int q = 0; q < 4; q++) {
                rng = rng * 1664525u + 1013904223u;
                const size_t pos = (size_t)(rng >> 4) % std::min<size_t>(t.size(), 4096 + (k % 2 ? t.size() : 0));  // headers and anywhere
                rng = rng * 1664525u + 1013904223u;
                t[pos] ^= (uint8_t)(1u << ((rng >> 8) & 7));
            }
            refused += plan_bytes(t, true);
            total++;
        }
        printf("  %d of %d malformed variants refused, none crashed\n", refused, total);
    }
    return 0;
}
