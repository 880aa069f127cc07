// Host-side sanitizer check of the ONNX reader + planner (no device needed):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
//       -Irust-birdnet-onnx_amd/csrc tools/asan_plan.cpp rust-birdnet-onnx_amd/csrc/{onnx_proto,engine,detect}.cpp -o /tmp/asan_plan
//   /tmp/asan_plan model.onnx [more.onnx ...]      (also feeds truncated / bit-flipped copies of each file)
// The shape rules the planner shares with the launchers live in csrc/plan_rules.h (inline, host C++), so nothing is restated here and
// no HIP object is linked.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "engine.h"

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
