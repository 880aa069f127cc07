// Host-side mirror of the reference's Classifier API (see include/birdnet_host.h).
// Everything numeric is delegated to the C ABI (bn_infer / bn_topk); this file
// keeps what the Rust shim keeps: validation order, error payloads, label
// lookup, result assembly, buffer/context ownership and locking.
#include "../../include/birdnet_host.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace birdnet {

uint32_t sample_rate(ModelType t) { return t == ModelType::BirdNetV24 ? 48000u : 32000u; }
float segment_duration(ModelType t) { return t == ModelType::BirdNetV24 ? 3.0f : 5.0f; }
size_t sample_count(ModelType t) { return t == ModelType::BirdNetV24 ? 144000u : 160000u; }
bool has_embeddings(ModelType t) { return t != ModelType::BirdNetV24; }
const char *as_str(ExecutionProviderInfo p) { return p == ExecutionProviderInfo::Rocm ? "ROCm" : "CPU"; }
const char *category(ExecutionProviderInfo p) { return p == ExecutionProviderInfo::Rocm ? "GPU" : "CPU"; }

namespace {

std::string last_backend_error() {
    char buf[1024];
    bn_last_error(buf, sizeof(buf));
    return buf;
}

// Rust's `{:?}` for std::time::Duration (core::time::Duration Debug): integral part, then
// the fractional digits with trailing zeros trimmed, in the largest unit that keeps >= 1.
std::string duration_debug(uint64_t ns) {
    auto fmt = [](uint64_t integer, uint64_t frac, uint64_t frac_div, const char *unit) {
        std::string s = std::to_string(integer);
        if (frac) {
            std::string f;
            for (uint64_t d = frac_div / 10; d >= 1; d /= 10) {
                f += (char)('0' + (frac / d) % 10);
                if (d == 1) break;
            }
            while (!f.empty() && f.back() == '0') f.pop_back();
            if (!f.empty()) s += "." + f;
        }
        return s + unit;
    };
    if (ns >= 1000000000ull) return fmt(ns / 1000000000ull, ns % 1000000000ull, 1000000000ull, "s");
    if (ns >= 1000000ull) return fmt(ns / 1000000ull, ns % 1000000ull, 1000000ull, "ms");
    if (ns >= 1000ull) return fmt(ns / 1000ull, ns % 1000ull, 1000ull, "\xc2\xb5s");
    return std::to_string(ns) + "ns";
}

Error input_size(size_t expected, size_t got) {
    return Error(Error::InputSize, "input size mismatch: expected " + std::to_string(expected) + " samples, got " + std::to_string(got), 0, expected, got);
}
Error batch_input_size(size_t index, size_t expected, size_t got) {
    return Error(Error::BatchInputSize,
                 "batch input size mismatch: segment " + std::to_string(index) + " has " + std::to_string(got) + " samples, expected " + std::to_string(expected),
                 index, expected, got);
}
Error inference(const std::string &m) { return Error(Error::Inference, "inference failed: " + m); }

Error from_status(bn_status st, const InferenceOptions &opt) {
    if (st == BN_ERR_TIMEOUT) {
        uint64_t ns = opt.timeout ? (uint64_t)opt.timeout->count() : 0;
        return Error(Error::Timeout, "inference timed out after " + duration_debug(ns), 0, 0, 0, ns);
    }
    if (st == BN_ERR_CANCELLED) return Error(Error::Cancelled, "inference was cancelled");
    return inference(last_backend_error());
}

}  // namespace

struct ClassifierInner {
    bn_model *model = nullptr;
    ModelConfig config{};
    bn_model_config raw_cfg{};
    std::vector<std::string> labels;
    ExecutionProviderInfo requested_provider = ExecutionProviderInfo::Cpu;
    size_t top_k = 10;
    std::optional<float> min_confidence;
    // Mutex<Session> of the reference (classifier.rs:435): predict/predict_batch serialise on one default context.
    std::mutex mu;
    bn_ctx *default_ctx = nullptr;
    size_t default_ctx_batch = 0;
    ~ClassifierInner() {
        if (default_ctx) bn_ctx_destroy(default_ctx);
        if (model) bn_model_free(model);
    }
};

BatchInferenceContext::~BatchInferenceContext() {
    if (ctx_) bn_ctx_destroy(ctx_);
}

const ModelConfig &Classifier::config() const { return inner_->config; }
const std::vector<std::string> &Classifier::labels() const { return inner_->labels; }
ExecutionProviderInfo Classifier::requested_provider() const { return inner_->requested_provider; }

namespace {

constexpr size_t kDefaultCtxCap = 1024;

// process_batch_outputs_from_flat (classifier.rs:872-911): slice rows, top-K, copy raw scores.
template <class InferFn>
std::vector<PredictionResult> run_on_ctx_with(ClassifierInner &in, bn_ctx *ctx, size_t n, const InferenceOptions &opt, InferFn infer) {
    // row lengths as planned (equal to num_species / embedding_dim for every 2-D output)
    size_t N = in.config.num_species, E = in.config.embedding_dim.value_or(0);
    {
        const float *dp = nullptr;
        size_t row = 0;
        if (bn_ctx_output_device(ctx, in.raw_cfg.logits_output, &dp, &row) == BN_OK) N = row;
        if (E && bn_ctx_output_device(ctx, in.raw_cfg.embedding_output, &dp, &row) == BN_OK) E = row;
    }
    std::vector<float> logits(n * N), emb(n * E);
    const volatile int32_t *cancel = opt.cancellation_token ? opt.cancellation_token->raw() : nullptr;
    const uint64_t timeout_ns = opt.timeout ? std::max<uint64_t>((uint64_t)opt.timeout->count(), 1) : 0;
    bn_status st = infer(logits.data(), E ? emb.data() : nullptr, cancel, timeout_ns);
    if (st != BN_OK) throw from_status(st, opt);
    const size_t k = std::min(in.top_k, N);
    std::vector<uint32_t> idx(n * std::max<size_t>(k, 1)), cnt(n);
    std::vector<float> conf(n * std::max<size_t>(k, 1));
    st = bn_topk(ctx, n, in.top_k, in.min_confidence ? 1 : 0, in.min_confidence.value_or(0.0f), std::max<size_t>(k, 1), idx.data(), conf.data(), cnt.data());
    if (st != BN_OK) throw from_status(st, opt);
    std::vector<PredictionResult> out(n);
    for (size_t i = 0; i < n; i++) {
        PredictionResult &r = out[i];
        r.model_type = in.config.model_type;
        r.raw_scores.assign(logits.begin() + i * N, logits.begin() + (i + 1) * N);
        if (E) r.embeddings = std::vector<float>(emb.begin() + i * E, emb.begin() + (i + 1) * E);
        for (size_t j = 0; j < cnt[i]; j++) {
            const size_t id = idx[i * std::max<size_t>(k, 1) + j];
            // labels.get(index) or "unknown_{index}" (postprocess.rs:69-72)
            r.predictions.push_back(Prediction{id < in.labels.size() ? in.labels[id] : "unknown_" + std::to_string(id), conf[i * std::max<size_t>(k, 1) + j], id});
        }
    }
    return out;
}

// results of one batch from flat host arrays (process_batch_outputs_from_flat, classifier.rs:872-911)
std::vector<PredictionResult> build_results(ClassifierInner &in, size_t n, size_t N, size_t E, size_t kstride, const std::vector<float> &logits,
                                            const std::vector<float> &emb, const std::vector<uint32_t> &idx, const std::vector<float> &conf,
                                            const std::vector<uint32_t> &cnt) {
    std::vector<PredictionResult> out(n);
    for (size_t i = 0; i < n; i++) {
        PredictionResult &r = out[i];
        r.model_type = in.config.model_type;
        r.raw_scores.assign(logits.begin() + i * N, logits.begin() + (i + 1) * N);
        if (E) r.embeddings = std::vector<float>(emb.begin() + i * E, emb.begin() + (i + 1) * E);
        for (size_t j = 0; j < cnt[i]; j++) {
            const size_t id = idx[i * kstride + j];
            r.predictions.push_back(Prediction{id < in.labels.size() ? in.labels[id] : "unknown_" + std::to_string(id), conf[i * kstride + j], id});
        }
    }
    return out;
}

// The host-slice call (prepare_input + run + extract_outputs + top-K, classifier.rs:826-867) as ONE submit and ONE
// collect: staging by the library's persistent pool, plan, top-K kernel and all device-to-host copies on the
// context's stream, a single wait.
std::vector<PredictionResult> run_on_ctx(ClassifierInner &in, bn_ctx *ctx, const float *const *segs, size_t n, const InferenceOptions &opt) {
    size_t N = in.config.num_species, E = in.config.embedding_dim.value_or(0);
    {
        const float *dp = nullptr;
        size_t row = 0;
        if (bn_ctx_output_device(ctx, in.raw_cfg.logits_output, &dp, &row) == BN_OK) N = row;
        if (E && bn_ctx_output_device(ctx, in.raw_cfg.embedding_output, &dp, &row) == BN_OK) E = row;
    }
    const volatile int32_t *cancel = opt.cancellation_token ? opt.cancellation_token->raw() : nullptr;
    const uint64_t timeout_ns = opt.timeout ? std::max<uint64_t>((uint64_t)opt.timeout->count(), 1) : 0;
    if (cancel && *cancel) throw from_status(BN_ERR_CANCELLED, opt);
    const size_t k = std::min(in.top_k, N), kstride = std::max<size_t>(k, 1);
    uint64_t ticket = 0;
    bn_status st = bn_infer_submit(ctx, segs, n, in.top_k, in.min_confidence ? 1 : 0, in.min_confidence.value_or(0.0f), &ticket);
    if (st != BN_OK) throw from_status(st, opt);
    std::vector<float> logits(n * N), emb(n * E), conf(n * kstride);
    std::vector<uint32_t> idx(n * kstride), cnt(n);
    st = bn_infer_collect(ctx, ticket, logits.data(), E ? emb.data() : nullptr, kstride, idx.data(), conf.data(), cnt.data(), cancel, timeout_ns);
    if (st != BN_OK) throw from_status(st, opt);
    return build_results(in, n, N, E, kstride, logits, emb, idx, conf, cnt);
}

bn_ctx *ensure_default_ctx(ClassifierInner &in, size_t n) {
    const size_t want = std::min(std::max<size_t>(n, 1), kDefaultCtxCap);
    if (!in.default_ctx || in.default_ctx_batch < want) {
        if (in.default_ctx) bn_ctx_destroy(in.default_ctx);
        in.default_ctx = nullptr;
        size_t cap = 1;
        while (cap < want) cap <<= 1;
        if (bn_ctx_create(in.model, cap, BN_CTX_DEFAULT, &in.default_ctx) != BN_OK) throw inference(last_backend_error());
        in.default_ctx_batch = cap;
    }
    return in.default_ctx;
}

}  // namespace

PredictionResult Classifier::predict(const float *segment, size_t len, const InferenceOptions &options) const {
    const size_t expected = inner_->config.sample_count;
    if (len != expected) throw input_size(expected, len);
    std::lock_guard<std::mutex> lk(inner_->mu);
    bn_ctx *ctx = ensure_default_ctx(*inner_, 1);
    const float *segs[1] = {segment};
    return std::move(run_on_ctx(*inner_, ctx, segs, 1, options)[0]);
}

std::vector<PredictionResult> Classifier::predict_batch(const float *const *segments, const size_t *lens, size_t n, const InferenceOptions &options) const {
    if (n == 0) return {};
    const size_t expected = inner_->config.sample_count;
    for (size_t i = 0; i < n; i++)
        if (lens[i] != expected) throw batch_input_size(i, expected, lens[i]);
    std::lock_guard<std::mutex> lk(inner_->mu);
    bn_ctx *ctx = ensure_default_ctx(*inner_, n);
    std::vector<PredictionResult> out;
    out.reserve(n);
    for (size_t off = 0; off < n; off += kDefaultCtxCap) {  // one device batch unless n exceeds the default context cap
        const size_t m = std::min(kDefaultCtxCap, n - off);
        auto part = run_on_ctx(*inner_, ctx, segments + off, m, options);
        for (auto &r : part) out.push_back(std::move(r));
    }
    return out;
}

std::unique_ptr<BatchInferenceContext> Classifier::create_batch_context(size_t max_batch_size) const {
    // batch_context.rs:107-114
    if (inner_->config.model_type == ModelType::PerchV2)
        throw inference("BatchInferenceContext does not yet support PerchV2 models. Use predict_batch() instead.");
    std::unique_ptr<BatchInferenceContext> c(new BatchInferenceContext());
    if (bn_ctx_create(inner_->model, std::max<size_t>(max_batch_size, 1), BN_CTX_DEFAULT, &c->ctx_) != BN_OK)
        throw inference("failed to create IoBinding: " + last_backend_error());
    c->max_batch_size_ = max_batch_size;
    c->sample_count_ = inner_->config.sample_count;
    c->model_type_ = inner_->config.model_type;
    return c;
}

std::unique_ptr<BatchInferenceContext> Classifier::create_native_batch_context(size_t max_batch_size, bool all_outputs) const {
    std::unique_ptr<BatchInferenceContext> c(new BatchInferenceContext());
    if (bn_ctx_create(inner_->model, std::max<size_t>(max_batch_size, 1), all_outputs ? BN_CTX_ALL_OUTPUTS : BN_CTX_DEFAULT, &c->ctx_) != BN_OK)
        throw inference("failed to create IoBinding: " + last_backend_error());
    c->max_batch_size_ = max_batch_size;
    c->sample_count_ = inner_->config.sample_count;
    c->model_type_ = inner_->config.model_type;
    return c;
}

std::vector<float> BatchInferenceContext::read_output(int index, size_t batch, size_t *row_elems) const {
    const float *dp = nullptr;
    size_t row = 0;
    if (bn_ctx_output_device(ctx_, index, &dp, &row) != BN_OK) throw inference(last_backend_error());
    if (row_elems) *row_elems = row;
    std::vector<float> out(batch * row);
    if (batch && bn_ctx_read_output(ctx_, index, batch, out.data()) != BN_OK) throw inference(last_backend_error());
    return out;
}

std::vector<PredictionResult> Classifier::predict_batch_with_context(BatchInferenceContext &ctx, const float *const *segments, const size_t *lens, size_t n,
                                                                     const InferenceOptions &options) const {
    if (n == 0) return {};
    // prepare_input (batch_context.rs:188-211): batch limit first, then per-segment sizes
    if (n > ctx.max_batch_size_) throw inference("batch size " + std::to_string(n) + " exceeds context max " + std::to_string(ctx.max_batch_size_));
    for (size_t i = 0; i < n; i++)
        if (lens[i] != ctx.sample_count_) throw batch_input_size(i, ctx.sample_count_, lens[i]);
    return run_on_ctx(*inner_, ctx.ctx_, segments, n, options);
}

// ---- recording-level path (birdnet-analyze.rs:556-600, 683-687, 707-743) ----
Recording::Recording(const void *pcm, size_t n_samples, int32_t format, int device) : n_samples_(n_samples) {
    bn_recording *r = nullptr;
    if (bn_recording_create(device, pcm, n_samples, format, &r) != BN_OK) throw inference("failed to upload recording: " + last_backend_error());
    rec_ = std::shared_ptr<bn_recording>(r, [](bn_recording *p) { bn_recording_free(p); });
}

std::vector<ChunkResult> Classifier::predict_recording(BatchInferenceContext &ctx, const Recording &rec, float overlap_secs, size_t first_chunk, size_t count,
                                                       const InferenceOptions &options) const {
    const ModelConfig &cfg = inner_->config;
    // chunk_audio: overlap_samples = (overlap * sample_rate) as usize; step = segment - overlap_samples
    const size_t overlap_samples = (size_t)(overlap_secs * (float)cfg.sample_rate);
    if (overlap_secs < 0.0f || overlap_samples >= cfg.sample_count) throw inference("overlap must be shorter than the segment duration");
    const size_t step = cfg.sample_count - overlap_samples;
    const size_t total = bn_chunk_count(rec.n_samples(), step);
    if (first_chunk > total) throw inference("first chunk " + std::to_string(first_chunk) + " is past the " + std::to_string(total) + " chunks of the recording");
    const size_t n = std::min(count, total - first_chunk);
    std::vector<ChunkResult> out;
    out.reserve(n);
    for (size_t off = 0; off < n; off += ctx.max_batch_size_) {
        const size_t m = std::min(ctx.max_batch_size_, n - off);
        const size_t first = first_chunk + off;
        auto part = run_on_ctx_with(*inner_, ctx.ctx_, m, options, [&](float *logits, float *emb, const volatile int32_t *cancel, uint64_t timeout_ns) {
            return bn_infer_windows(ctx.ctx_, rec.raw(), step, first, m, logits, emb, cancel, timeout_ns);
        });
        for (size_t i = 0; i < m; i++) {
            // start_time = pos as f32 / sample_rate as f32 (birdnet-analyze.rs:736)
            const float t = (float)((first + i) * step) / (float)cfg.sample_rate;
            out.push_back(ChunkResult{t, std::move(part[i])});
        }
    }
    return out;
}

Classifier ClassifierBuilder::build() {
    if (!model_path_) throw Error(Error::ModelPathRequired, "model path required");
    if (!labels_ && !labels_path_) throw Error(Error::LabelsRequired, "labels required (provide path or vec)");
    auto in = std::make_shared<ClassifierInner>();
    bn_status st = bn_model_load(model_path_->c_str(), device_, model_type_ ? (int32_t)*model_type_ : -1, &in->model);
    if (st == BN_ERR_MODEL_DETECTION) throw Error(Error::ModelDetection, "model detection failed: " + last_backend_error());
    if (st != BN_OK) throw Error(Error::ModelLoad, "failed to load model: " + last_backend_error());
    bn_model_get_config(in->model, &in->raw_cfg);
    const bn_model_config &c = in->raw_cfg;
    in->config.model_type = (ModelType)c.model_type;
    in->config.sample_rate = c.sample_rate;
    in->config.segment_duration = c.segment_duration;
    in->config.sample_count = (size_t)c.sample_count;
    in->config.num_species = (size_t)c.num_species;
    if (c.has_embedding) in->config.embedding_dim = (size_t)c.embedding_dim;
    in->labels = labels_ ? *labels_ : load_labels_from_file(*labels_path_, in->config.model_type);
    if (in->labels.size() != in->config.num_species)
        throw Error(Error::LabelCount, "label count mismatch: model expects " + std::to_string(in->config.num_species) + ", got " + std::to_string(in->labels.size()), 0,
                    in->config.num_species, in->labels.size());
    in->requested_provider = provider_;
    in->top_k = top_k_;
    in->min_confidence = min_confidence_;
    Classifier cl;
    cl.inner_ = std::move(in);
    return cl;
}

// ---- labels (labels.rs:22-95) ----
namespace {
std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
std::string lower(std::string s) {
    for (auto &ch : s) ch = (char)tolower((unsigned char)ch);
    return s;
}
bool looks_like_header(const std::string &v) {
    const std::string l = lower(v);
    auto ends_with = [&](const char *suf) { size_t n = strlen(suf); return l.size() >= n && l.compare(l.size() - n, n, suf) == 0; };
    return l == "label" || l == "species" || l == "name" || l == "class" || l == "common_name" || l == "scientific_name" || l.rfind("inat", 0) == 0 ||
           ends_with("_fsd50k");
}
}  // namespace

std::vector<std::string> parse_text_labels(const std::string &content) {
    std::vector<std::string> out;
    std::istringstream ss(content);
    std::string line;
    while (std::getline(ss, line)) {
        std::string t = trim(line);
        if (!t.empty()) out.push_back(t);
    }
    return out;
}

std::vector<std::string> parse_csv_labels(const std::string &content) {
    // first column of every record; RFC-4180 quoting for that column
    std::vector<std::string> out;
    bool first_row = true;
    size_t p = 0;
    const size_t n = content.size();
    while (p < n) {
        std::string field;
        bool any = false;
        if (content[p] == '"') {
            p++;
            while (p < n) {
                if (content[p] == '"') {
                    if (p + 1 < n && content[p + 1] == '"') { field += '"'; p += 2; }
                    else { p++; break; }
                } else field += content[p++];
            }
            any = true;
        }
        while (p < n && content[p] != ',' && content[p] != '\n' && content[p] != '\r') { field += content[p++]; any = true; }
        // skip the rest of the record
        bool in_q = false;
        while (p < n && (in_q || content[p] != '\n')) {
            if (content[p] == '"') in_q = !in_q;
            p++;
        }
        if (p < n) p++;
        if (!any && field.empty()) continue;  // csv crate skips empty lines
        std::string label = trim(field);
        if (first_row && looks_like_header(label)) { first_row = false; continue; }
        first_row = false;
        if (!label.empty()) out.push_back(label);
    }
    return out;
}

// ---- JSON labels (labels.rs:95-121), with serde_json's acceptance rules for the three shapes ----
namespace {
struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    std::string str;
    std::vector<JsonValue> items;                             // Array
    std::vector<std::pair<std::string, JsonValue>> members;   // Object (duplicates: the last one wins, as in serde)
};
struct JsonParser {
    const std::string &s;
    size_t p = 0;
    int depth = 0;
    explicit JsonParser(const std::string &src) : s(src) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++; }
    [[noreturn]] void bad() { throw std::runtime_error("invalid JSON"); }
    static void utf8(std::string &o, uint32_t cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    uint32_t hex4() {
        if (p + 4 > s.size()) bad();
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            const char ch = s[p++];
            v <<= 4;
            if (ch >= '0' && ch <= '9') v |= (uint32_t)(ch - '0');
            else if (ch >= 'a' && ch <= 'f') v |= (uint32_t)(ch - 'a' + 10);
            else if (ch >= 'A' && ch <= 'F') v |= (uint32_t)(ch - 'A' + 10);
            else bad();
        }
        return v;
    }
    std::string string() {
        if (p >= s.size() || s[p] != '"') bad();
        p++;
        std::string o;
        while (true) {
            if (p >= s.size()) bad();
            const unsigned char ch = (unsigned char)s[p++];
            if (ch == '"') break;
            if (ch < 0x20) bad();
            if (ch != '\\') { o += (char)ch; continue; }
            if (p >= s.size()) bad();
            const char e = s[p++];
            switch (e) {
                case '"': o += '"'; break;
                case '\\': o += '\\'; break;
                case '/': o += '/'; break;
                case 'b': o += '\b'; break;
                case 'f': o += '\f'; break;
                case 'n': o += '\n'; break;
                case 'r': o += '\r'; break;
                case 't': o += '\t'; break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF) {  // surrogate pair
                        if (p + 2 > s.size() || s[p] != '\\' || s[p + 1] != 'u') bad();
                        p += 2;
                        const uint32_t lo = hex4();
                        if (lo < 0xDC00 || lo > 0xDFFF) bad();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) bad();
                    utf8(o, cp);
                    break;
                }
                default: bad();
            }
        }
        return o;
    }
    JsonValue value() {
        if (++depth > 128) bad();
        ws();
        if (p >= s.size()) bad();
        JsonValue v;
        const char ch = s[p];
        if (ch == '"') { v.kind = JsonValue::String; v.str = string(); }
        else if (ch == '[') {
            v.kind = JsonValue::Array;
            p++;
            ws();
            if (p < s.size() && s[p] == ']') p++;
            else
                while (true) {
                    v.items.push_back(value());
                    ws();
                    if (p < s.size() && s[p] == ',') { p++; continue; }
                    if (p < s.size() && s[p] == ']') { p++; break; }
                    bad();
                }
        } else if (ch == '{') {
            v.kind = JsonValue::Object;
            p++;
            ws();
            if (p < s.size() && s[p] == '}') p++;
            else
                while (true) {
                    ws();
                    std::string key = string();
                    ws();
                    if (p >= s.size() || s[p] != ':') bad();
                    p++;
                    v.members.emplace_back(std::move(key), value());
                    ws();
                    if (p < s.size() && s[p] == ',') { p++; continue; }
                    if (p < s.size() && s[p] == '}') { p++; break; }
                    bad();
                }
        } else if (s.compare(p, 4, "true") == 0) { v.kind = JsonValue::Bool; p += 4; }
        else if (s.compare(p, 5, "false") == 0) { v.kind = JsonValue::Bool; p += 5; }
        else if (s.compare(p, 4, "null") == 0) { v.kind = JsonValue::Null; p += 4; }
        else if (ch == '-' || (ch >= '0' && ch <= '9')) {
            v.kind = JsonValue::Number;
            const size_t st = p;
            if (s[p] == '-') p++;
            while (p < s.size() && ((s[p] >= '0' && s[p] <= '9') || s[p] == '.' || s[p] == 'e' || s[p] == 'E' || s[p] == '+' || s[p] == '-')) p++;
            if (p == st || (s[st] == '-' && p == st + 1)) bad();
        } else bad();
        depth--;
        return v;
    }
    JsonValue document() {
        JsonValue v = value();
        ws();
        if (p != s.size()) bad();  // trailing characters
        return v;
    }
};
// Vec<String>: every element a string
bool as_string_array(const JsonValue &v, std::vector<std::string> &out) {
    if (v.kind != JsonValue::Array) return false;
    out.clear();
    for (const auto &e : v.items) {
        if (e.kind != JsonValue::String) return false;
        out.push_back(e.str);
    }
    return true;
}
const JsonValue *member(const JsonValue &o, const char *key) {
    const JsonValue *hit = nullptr;
    for (const auto &kv : o.members)
        if (kv.first == key) hit = &kv.second;
    return hit;
}
}  // namespace

std::vector<std::string> parse_json_labels(const std::string &content) {
    JsonValue doc;
    bool ok = true;
    try {
        doc = JsonParser(content).document();
    } catch (const std::exception &) {
        ok = false;
    }
    std::vector<std::string> labels;
    if (ok) {
        if (as_string_array(doc, labels)) return labels;                       // ["a", "b"]
        if (doc.kind == JsonValue::Object) {                                    // {"labels": [...]} (other keys ignored)
            const JsonValue *l = member(doc, "labels");
            if (l && as_string_array(*l, labels)) return labels;
        }
        if (doc.kind == JsonValue::Array) {                                     // [{"name"|"label"|"species": ...}, ...]
            bool shape = true;
            labels.clear();
            for (const auto &e : doc.items) {
                if (e.kind != JsonValue::Object) { shape = false; break; }
                const JsonValue *pick = nullptr;
                for (const char *key : {"name", "label", "species"}) {
                    const JsonValue *m = member(e, key);
                    if (!m || m->kind == JsonValue::Null) continue;            // Option<String>: missing or null => None
                    if (m->kind != JsonValue::String) { shape = false; break; }  // wrong type: the whole document fails
                    if (!pick) pick = m;
                }
                if (!shape) break;
                if (pick) labels.push_back(pick->str);
            }
            if (shape && !labels.empty()) return labels;
        }
    }
    throw Error(Error::LabelParse,
                "failed to parse labels: unrecognized JSON format: expected array of strings, {labels: [...]}, or [{name: ...}]");
}

std::vector<std::string> parse_labels(const std::string &content, LabelFormat format) {
    switch (format) {
        case LabelFormat::Text: return parse_text_labels(content);
        case LabelFormat::Csv: return parse_csv_labels(content);
        default: return parse_json_labels(content);
    }
}

std::vector<std::string> load_labels_from_file(const std::string &path, ModelType t) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(Error::LabelLoad, "failed to load labels from " + path + ": No such file or directory (os error 2)");
    std::stringstream ss;
    ss << f.rdbuf();
    return t == ModelType::BirdNetV24 ? parse_text_labels(ss.str()) : parse_csv_labels(ss.str());
}

std::vector<Chunk> chunk_plan(size_t n_samples, size_t segment_samples, float overlap_secs, uint32_t rate) {
    // birdnet-analyze.rs:707-743
    const float prod = overlap_secs * (float)rate;
    const size_t overlap = prod > 0.0f ? (size_t)prod : 0;
    const size_t step = segment_samples > overlap ? segment_samples - overlap : 0;
    std::vector<Chunk> out;
    if (step == 0) return out;
    for (size_t pos = 0; pos < n_samples; pos += step) out.push_back(Chunk{pos, (float)pos / (float)rate});
    return out;
}

}  // namespace birdnet
