#include "detect.h"

namespace bn {

uint32_t model_sample_rate(int mt) { return mt == BN_MODEL_BIRDNET_V24 ? 48000u : 32000u; }
float model_segment_duration(int mt) { return mt == BN_MODEL_BIRDNET_V24 ? 3.0f : 5.0f; }
uint64_t model_sample_count(int mt) { return mt == BN_MODEL_BIRDNET_V24 ? 144000u : 160000u; }
const char *model_type_name(int mt) {
    switch (mt) {
        case BN_MODEL_BIRDNET_V24: return "BirdNetV24";
        case BN_MODEL_BIRDNET_V30: return "BirdNetV30";
        case BN_MODEL_PERCH_V2: return "PerchV2";
        default: return "Unknown";
    }
}

namespace {
std::string shape_str(const std::vector<int64_t> &s) {
    std::string r = "[";
    for (size_t k = 0; k < s.size(); k++) r += (k ? ", " : "") + std::to_string(s[k]);
    return r + "]";
}
// detection.rs:149-163
bool sample_count_of(const std::vector<int64_t> &shape, uint64_t &out, std::string &reason) {
    int64_t v;
    if (shape.size() == 2) v = shape[1];
    else if (shape.size() == 3) v = shape[2];
    else {
        reason = "unexpected input shape: " + shape_str(shape);
        return false;
    }
    if (v < 0) {
        reason = "invalid sample count: " + std::to_string(v);
        return false;
    }
    out = (uint64_t)v;
    return true;
}
// detection.rs:166-174
bool last_dim_of(const std::vector<int64_t> &shape, uint64_t &out, std::string &reason) {
    if (shape.empty()) {
        reason = "empty output shape";
        return false;
    }
    if (shape.back() < 0) {
        reason = "invalid dimension: " + std::to_string(shape.back());
        return false;
    }
    out = (uint64_t)shape.back();
    return true;
}
void fill(bn_model_config &c, int mt, uint64_t sc, uint64_t ns, bool has_e, uint64_t ed) {
    c.model_type = mt;
    c.sample_rate = model_sample_rate(mt);
    c.segment_duration = model_segment_duration(mt);
    c.sample_count = sc;
    c.num_species = ns;
    c.has_embedding = has_e ? 1 : 0;
    c.embedding_dim = has_e ? ed : 0;
    // classifier.rs:917-934: which outputs carry logits / embeddings
    c.logits_output = mt == BN_MODEL_BIRDNET_V24 ? 0 : mt == BN_MODEL_BIRDNET_V30 ? 1 : 3;
    c.embedding_output = mt == BN_MODEL_BIRDNET_V24 ? -1 : 0;
}
}  // namespace

bool detect_model_type(const std::vector<int64_t> &in, const std::vector<std::vector<int64_t>> &outs, int override_type,
                       bn_model_config &cfg, std::string &reason) {
    uint64_t sc = 0;
    if (!sample_count_of(in, sc, reason)) return false;
    const size_t n_out = outs.size();
    if (override_type >= 0) {  // detection.rs:83-145
        const int mt = override_type;
        if (mt > BN_MODEL_PERCH_V2) {
            reason = "unknown model type override " + std::to_string(mt);
            return false;
        }
        const uint64_t expected = model_sample_count(mt);
        if (sc != expected) {
            reason = std::string("model type ") + model_type_name(mt) + " expects " + std::to_string(expected) + " samples, but model has " + std::to_string(sc);
            return false;
        }
        uint64_t ns = 0, ed = 0;
        if (mt == BN_MODEL_BIRDNET_V24) {
            if (n_out != 1) { reason = "`BirdNET` v2.4 expects 1 output, got " + std::to_string(n_out); return false; }
            if (!last_dim_of(outs[0], ns, reason)) return false;
            fill(cfg, mt, sc, ns, false, 0);
        } else if (mt == BN_MODEL_BIRDNET_V30) {
            if (n_out != 2) { reason = "`BirdNET` v3.0 expects 2 outputs, got " + std::to_string(n_out); return false; }
            if (!last_dim_of(outs[0], ed, reason) || !last_dim_of(outs[1], ns, reason)) return false;
            fill(cfg, mt, sc, ns, true, ed);
        } else {
            if (n_out != 4) { reason = "`Perch` v2 expects 4 outputs, got " + std::to_string(n_out); return false; }
            if (!last_dim_of(outs[0], ed, reason) || !last_dim_of(outs[3], ns, reason)) return false;
            fill(cfg, mt, sc, ns, true, ed);
        }
        return true;
    }
    // detection.rs:29-79
    if (sc == 144000 && n_out == 1) {
        uint64_t ns;
        if (!last_dim_of(outs[0], ns, reason)) return false;
        fill(cfg, BN_MODEL_BIRDNET_V24, 144000, ns, false, 0);
        return true;
    }
    if (sc == 160000 && n_out == 2) {
        uint64_t ed, ns;
        if (!last_dim_of(outs[0], ed, reason) || !last_dim_of(outs[1], ns, reason)) return false;
        fill(cfg, BN_MODEL_BIRDNET_V30, 160000, ns, true, ed);
        return true;
    }
    if (sc == 160000 && n_out == 4) {
        uint64_t ed, ns;
        if (!last_dim_of(outs[0], ed, reason) || !last_dim_of(outs[3], ns, reason)) return false;
        fill(cfg, BN_MODEL_PERCH_V2, 160000, ns, true, ed);
        return true;
    }
    reason = "unsupported model: " + std::to_string(sc) + " samples, " + std::to_string(n_out) + " outputs (expected 144000/1, 160000/2, or 160000/4)";
    return false;
}

}  // namespace bn
