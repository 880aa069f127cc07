// Flat C ABI (bnh_*) over the C++ host mirror, for the ctypes test/bench harness.
#include <cstdio>
#include <cstring>

#include "../../include/birdnet_host.h"

using namespace birdnet;

struct bnh_classifier {
    Classifier cl;
};
struct bnh_context {
    std::unique_ptr<BatchInferenceContext> ctx;
};
struct bnh_results {
    std::vector<PredictionResult> v;
};

namespace {

int32_t set_err(bnh_error *err, const Error &e) {
    const int32_t kind = (int32_t)e.kind + 1;
    if (err) {
        err->kind = kind;
        err->index = e.index;
        err->expected = e.expected;
        err->got = e.got;
        err->duration_ns = e.duration_ns;
        err->latitude = e.latitude;
        err->longitude = e.longitude;
        err->month = e.month;
        err->day = e.day;
        snprintf(err->message, sizeof(err->message), "%s", e.what());
    }
    return kind;
}
int32_t set_other(bnh_error *err, const char *what) {
    if (err) {
        memset(err, 0, sizeof(*err));
        err->kind = BNH_ERR_OTHER;
        snprintf(err->message, sizeof(err->message), "%s", what);
    }
    return BNH_ERR_OTHER;
}
// timeout_ns < 0 => None.  A non-NULL cancel flag becomes a CancellationToken aliasing it, so
// "cancel from another thread" is observed while the batch runs.
InferenceOptions make_opts(int64_t timeout_ns, const volatile int32_t *cancel) {
    InferenceOptions o;
    if (timeout_ns >= 0) o.timeout = std::chrono::nanoseconds(timeout_ns);
    if (cancel) o.cancellation_token = CancellationToken::alias(const_cast<volatile int32_t *>(cancel));
    return o;
}

template <class F>
int32_t guarded(bnh_error *err, F &&f) {
    try {
        f();
        if (err) memset(err, 0, sizeof(*err));
        return BNH_OK;
    } catch (const Error &e) {
        return set_err(err, e);
    } catch (const std::exception &e) {
        return set_other(err, e.what());
    }
}

}  // namespace

extern "C" {

int32_t bnh_classifier_build(const char *model_path, const char *labels_path, const char *const *labels, size_t n_labels, int32_t model_type,
                             int64_t top_k, int32_t has_min, float min_conf, int32_t device, bnh_classifier **out, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        ClassifierBuilder b;
        if (model_path) b.model_path(model_path);
        if (labels) {
            std::vector<std::string> l;
            for (size_t i = 0; i < n_labels; i++) l.emplace_back(labels[i]);
            b.labels(std::move(l));
        } else if (labels_path) b.labels_path(labels_path);
        if (model_type >= 0) b.model_type((ModelType)model_type);
        if (top_k >= 0) b.top_k((size_t)top_k);
        else b.top_k((size_t)-1);  // usize::MAX
        if (has_min) b.min_confidence(min_conf);
        b.with_rocm(device);
        auto *c = new bnh_classifier{b.build()};
        *out = c;
    });
}

void bnh_classifier_free(bnh_classifier *c) { delete c; }

void bnh_classifier_config(const bnh_classifier *c, bn_model_config *out) {
    const ModelConfig &m = c->cl.config();
    memset(out, 0, sizeof(*out));
    out->model_type = (int32_t)m.model_type;
    out->sample_rate = m.sample_rate;
    out->segment_duration = m.segment_duration;
    out->sample_count = m.sample_count;
    out->num_species = m.num_species;
    out->has_embedding = m.embedding_dim ? 1 : 0;
    out->embedding_dim = m.embedding_dim.value_or(0);
    out->logits_output = m.model_type == ModelType::BirdNetV24 ? 0 : m.model_type == ModelType::BirdNetV30 ? 1 : 3;
    out->embedding_output = m.model_type == ModelType::BirdNetV24 ? -1 : 0;
}

const char *bnh_classifier_provider(const bnh_classifier *c) { return as_str(c->cl.requested_provider()); }
size_t bnh_classifier_label_count(const bnh_classifier *c) { return c->cl.labels().size(); }
const char *bnh_classifier_label(const bnh_classifier *c, size_t i) { return i < c->cl.labels().size() ? c->cl.labels()[i].c_str() : nullptr; }

int32_t bnh_predict(const bnh_classifier *c, const float *segment, size_t len, int64_t timeout_ns, const volatile int32_t *cancel, bnh_results **out,
                    bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        InferenceOptions o = make_opts(timeout_ns, cancel);
        auto r = std::make_unique<bnh_results>();
        r->v.push_back(c->cl.predict(segment, len, o));
        *out = r.release();
    });
}

int32_t bnh_predict_batch(const bnh_classifier *c, const float *const *segments, const size_t *lens, size_t n, int64_t timeout_ns,
                          const volatile int32_t *cancel, bnh_results **out, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        InferenceOptions o = make_opts(timeout_ns, cancel);
        auto r = std::make_unique<bnh_results>();
        r->v = c->cl.predict_batch(segments, lens, n, o);
        *out = r.release();
    });
}

int32_t bnh_create_batch_context(const bnh_classifier *c, size_t max_batch, bnh_context **out, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        auto x = std::make_unique<bnh_context>();
        x->ctx = c->cl.create_batch_context(max_batch);
        *out = x.release();
    });
}

int32_t bnh_create_native_batch_context(const bnh_classifier *c, size_t max_batch, int32_t all_outputs, bnh_context **out, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        auto x = std::make_unique<bnh_context>();
        x->ctx = c->cl.create_native_batch_context(max_batch, all_outputs != 0);
        *out = x.release();
    });
}

size_t bnh_context_read_output(const bnh_context *ctx, int32_t index, size_t batch, float *out, size_t cap, size_t *row_elems, bnh_error *err) {
    size_t need = 0;
    guarded(err, [&] {
        size_t row = 0;
        std::vector<float> v = ctx->ctx->read_output(index, batch, &row);
        if (row_elems) *row_elems = row;
        need = v.size();
        if (out) memcpy(out, v.data(), std::min(cap, need) * sizeof(float));
    });
    return need;
}

void bnh_context_free(bnh_context *ctx) { delete ctx; }
size_t bnh_context_max_batch_size(const bnh_context *ctx) { return ctx->ctx->max_batch_size(); }
size_t bnh_context_sample_count(const bnh_context *ctx) { return ctx->ctx->sample_count(); }
size_t bnh_context_input_buffer_capacity(const bnh_context *ctx) { return ctx->ctx->input_buffer_capacity(); }
size_t bnh_context_input_buffer_bytes(const bnh_context *ctx) { return ctx->ctx->input_buffer_bytes(); }
int32_t bnh_context_model_type(const bnh_context *ctx) { return (int32_t)ctx->ctx->model_type(); }

int32_t bnh_predict_batch_with_context(const bnh_classifier *c, bnh_context *ctx, const float *const *segments, const size_t *lens, size_t n,
                                       int64_t timeout_ns, const volatile int32_t *cancel, bnh_results **out, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        InferenceOptions o = make_opts(timeout_ns, cancel);
        auto r = std::make_unique<bnh_results>();
        r->v = c->cl.predict_batch_with_context(*ctx->ctx, segments, lens, n, o);
        *out = r.release();
    });
}

int32_t bnh_predict_recording(const bnh_classifier *c, bnh_context *ctx, const void *pcm, size_t n_samples, int32_t format, float overlap_secs,
                              size_t first_chunk, size_t count, int64_t timeout_ns, const volatile int32_t *cancel, bnh_results **out, float *start_times,
                              size_t times_cap, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        InferenceOptions o = make_opts(timeout_ns, cancel);
        Recording rec(pcm, n_samples, format, 0);
        auto chunks = c->cl.predict_recording(*ctx->ctx, rec, overlap_secs, first_chunk, count, o);
        auto r = std::make_unique<bnh_results>();
        for (size_t i = 0; i < chunks.size(); i++) {
            if (start_times && i < times_cap) start_times[i] = chunks[i].start_time;
            r->v.push_back(std::move(chunks[i].result));
        }
        *out = r.release();
    });
}

size_t bnh_results_len(const bnh_results *r) { return r ? r->v.size() : 0; }
int32_t bnh_result_model_type(const bnh_results *r, size_t i) { return (int32_t)r->v[i].model_type; }
size_t bnh_result_n_predictions(const bnh_results *r, size_t i) { return r->v[i].predictions.size(); }
const char *bnh_result_species(const bnh_results *r, size_t i, size_t j) { return r->v[i].predictions[j].species.c_str(); }
float bnh_result_confidence(const bnh_results *r, size_t i, size_t j) { return r->v[i].predictions[j].confidence; }
size_t bnh_result_index(const bnh_results *r, size_t i, size_t j) { return r->v[i].predictions[j].index; }
size_t bnh_result_raw_scores(const bnh_results *r, size_t i, const float **data) {
    *data = r->v[i].raw_scores.data();
    return r->v[i].raw_scores.size();
}
size_t bnh_result_embeddings(const bnh_results *r, size_t i, const float **data) {
    *data = nullptr;
    if (!r->v[i].embeddings) return 0;
    *data = r->v[i].embeddings->data();
    return r->v[i].embeddings->size();
}
void bnh_results_free(bnh_results *r) { delete r; }

// ---- range filter ----
struct bnh_range_filter {
    RangeFilter f;
    std::vector<std::string> labels;
};

float bnh_calculate_week(uint32_t month, uint32_t day) { return calculate_week(month, day); }
int32_t bnh_validate_coordinates(float latitude, float longitude, bnh_error *err) {
    return guarded(err, [&] { validate_coordinates(latitude, longitude); });
}
int32_t bnh_validate_date(uint32_t month, uint32_t day, bnh_error *err) {
    return guarded(err, [&] { validate_date(month, day); });
}

int32_t bnh_range_filter_build(const char *model_path, const char *labels_path, const char *const *labels, size_t n_labels, float threshold, int32_t device,
                               bnh_range_filter **out, bnh_error *err) {
    if (out) *out = nullptr;
    return guarded(err, [&] {
        RangeFilterBuilder b = RangeFilter::builder();
        std::vector<std::string> l;
        if (model_path) b.model_path(model_path);
        if (labels) {
            for (size_t i = 0; i < n_labels; i++) l.emplace_back(labels[i]);
            b.labels(l);
        } else if (labels_path) b.labels_path(labels_path);
        if (threshold >= 0.0f) b.threshold(threshold);
        b.with_rocm(device);
        auto r = std::make_unique<bnh_range_filter>();
        r->f = b.build();
        if (labels) r->labels = std::move(l);
        else if (labels_path) r->labels = load_labels_from_file(labels_path, ModelType::BirdNetV24);
        *out = r.release();
    });
}

void bnh_range_filter_free(bnh_range_filter *f) { delete f; }
const char *bnh_range_filter_label(const bnh_range_filter *f, size_t i) { return i < f->labels.size() ? f->labels[i].c_str() : nullptr; }

int32_t bnh_range_filter_predict(const bnh_range_filter *f, float latitude, float longitude, uint32_t month, uint32_t day, uint32_t *idx_out, float *score_out,
                                 size_t cap, size_t *n_out, bnh_error *err) {
    if (n_out) *n_out = 0;
    return guarded(err, [&] {
        auto v = f->f.predict(latitude, longitude, month, day);
        if (n_out) *n_out = v.size();
        for (size_t i = 0; i < v.size() && i < cap; i++) {
            if (idx_out) idx_out[i] = (uint32_t)v[i].index;
            if (score_out) score_out[i] = v[i].score;
        }
    });
}

size_t bnh_filter_predictions(const char *const *pred_species, const float *pred_conf, size_t n_pred, const char *const *loc_species, const float *loc_score,
                              size_t n_loc, float threshold, int32_t rerank, uint32_t *keep_pos, float *conf_out) {
    std::vector<Prediction> preds;
    for (size_t i = 0; i < n_pred; i++) preds.push_back(Prediction{pred_species[i], pred_conf[i], i});  // index carries the input position
    std::vector<LocationScore> loc;
    for (size_t i = 0; i < n_loc; i++) loc.push_back(LocationScore{loc_species[i], loc_score[i], i});
    auto out = filter_predictions(preds, loc, threshold, rerank != 0);
    for (size_t i = 0; i < out.size(); i++) {
        keep_pos[i] = (uint32_t)out[i].index;
        conf_out[i] = out[i].confidence;
    }
    return out.size();
}

size_t bnh_parse_labels(const char *content, int32_t csv, char *out, size_t cap) {
    std::vector<std::string> l;
    try {
        l = csv ? parse_csv_labels(content ? content : "") : parse_text_labels(content ? content : "");
    } catch (...) {
        return 0;
    }
    std::string joined;
    for (size_t i = 0; i < l.size(); i++) joined += (i ? "\n" : "") + l[i];
    if (out && cap) snprintf(out, cap, "%s", joined.c_str());
    return joined.size() + 1;
}

size_t bnh_parse_labels_format(const char *content, int32_t format, char *out, size_t cap, bnh_error *err) {
    std::string joined;
    const int32_t rc = guarded(err, [&] {
        const auto l = parse_labels(content ? content : "", (LabelFormat)format);
        for (size_t i = 0; i < l.size(); i++) joined += (i ? "\x1f" : "") + l[i];
    });
    if (rc != BNH_OK) return 0;
    if (out && cap) {
        const size_t n = std::min(cap - 1, joined.size());
        memcpy(out, joined.data(), n);
        out[n] = 0;
    }
    return joined.size() + 1;
}

size_t bnh_chunk_plan(size_t n_samples, size_t segment_samples, float overlap_secs, uint32_t sample_rate, uint64_t *starts, float *start_times,
                      size_t cap) {
    auto v = chunk_plan(n_samples, segment_samples, overlap_secs, sample_rate);
    for (size_t i = 0; i < v.size() && i < cap; i++) {
        if (starts) starts[i] = v[i].start;
        if (start_times) start_times[i] = v[i].start_time;
    }
    return v.size();
}

}  // extern "C"
