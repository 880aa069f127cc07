// Host-side mirror of the reference's RangeFilter (src/rangefilter.rs, see include/birdnet_host.h).
// The meta model ((lat, lon, week) -> per-species prior) runs on the MI355X through the same engine as
// the audio models (BN_MODEL_GENERIC); everything else here is what a Rust shim keeps: validation
// order and error payloads, week arithmetic, threshold / sort / filter / rerank.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "../../include/birdnet_host.h"

namespace birdnet {

namespace {

std::string backend_error() {
    char buf[1024];
    bn_last_error(buf, sizeof(buf));
    return buf;
}

// Rust `{}` of an f32: shortest decimal that round-trips, never in exponent form.
std::string f32_display(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    for (int p = 1; p <= 9; p++) {
        snprintf(buf, sizeof(buf), "%.*g", p, (double)v);
        if (strtof(buf, nullptr) == v) break;
    }
    if (strchr(buf, 'e') || strchr(buf, 'E')) {
        // expand: enough fractional digits for the shortest round trip, then trim
        for (int d = 0; d <= 60; d++) {
            snprintf(buf, sizeof(buf), "%.*f", d, (double)v);
            if (strtof(buf, nullptr) == v) break;
        }
    }
    return buf;
}

// f32::total_cmp as an integer key
uint32_t total_key(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

Error invalid_coordinates(float lat, float lon, const std::string &reason) {
    Error e(Error::InvalidCoordinates, "invalid coordinates: latitude: " + f32_display(lat) + ", longitude: " + f32_display(lon) + ", reason: " + reason);
    e.latitude = lat;
    e.longitude = lon;
    return e;
}
Error invalid_date(uint32_t month, uint32_t day, const std::string &reason) {
    Error e(Error::InvalidDate, "invalid date: month: " + std::to_string(month) + ", day: " + std::to_string(day) + ", reason: " + reason);
    e.month = month;
    e.day = day;
    return e;
}
Error rf_inference(const std::string &m) { return Error(Error::RangeFilterInference, "range filter inference failed: " + m); }

}  // namespace

float calculate_week(uint32_t month, uint32_t day) {
    const uint32_t weeks_from_months = (month - 1u) * 4u;
    const uint32_t week_in_month = (day - 1u) / 7u + 1u;
    return (float)(weeks_from_months + week_in_month);
}

void validate_coordinates(float latitude, float longitude) {
    if (!(latitude >= -90.0f && latitude <= 90.0f))
        throw invalid_coordinates(latitude, longitude, "latitude must be in range [-90, 90], got " + f32_display(latitude));
    if (!(longitude >= -180.0f && longitude <= 180.0f))
        throw invalid_coordinates(latitude, longitude, "longitude must be in range [-180, 180], got " + f32_display(longitude));
}

void validate_date(uint32_t month, uint32_t day) {
    if (month < 1 || month > 12) throw invalid_date(month, day, "month must be in range [1, 12], got " + std::to_string(month));
    if (day < 1 || day > 31) throw invalid_date(month, day, "day must be in range [1, 31], got " + std::to_string(day));
}

std::vector<Prediction> filter_predictions(const std::vector<Prediction> &predictions, const std::vector<LocationScore> &location_scores, float threshold,
                                           bool rerank) {
    std::unordered_map<std::string, float> location_map;  // a later duplicate overrides an earlier one (HashMap collect)
    for (const auto &s : location_scores) location_map[s.species] = s.score;
    std::vector<Prediction> out;
    for (const auto &p : predictions) {
        auto it = location_map.find(p.species);
        if (it == location_map.end()) {
            out.push_back(p);  // not in the meta model: keep unchanged
        } else if (it->second >= threshold) {
            out.push_back(Prediction{p.species, rerank ? p.confidence * it->second : p.confidence, p.index});
        }  // in the meta model, below the threshold: drop
    }
    if (rerank)
        std::stable_sort(out.begin(), out.end(), [](const Prediction &a, const Prediction &b) { return total_key(a.confidence) > total_key(b.confidence); });
    return out;
}

struct RangeFilterInner {
    bn_model *model = nullptr;
    bn_ctx *ctx = nullptr;
    std::mutex mu;  // Mutex<Session> (rangefilter.rs:389-393)
    std::vector<std::string> labels;
    float threshold = 0.01f;
    size_t n_out = 0;
    ~RangeFilterInner() {
        if (ctx) bn_ctx_destroy(ctx);
        if (model) bn_model_free(model);
    }
};

RangeFilterBuilder RangeFilter::builder() { return RangeFilterBuilder(); }
size_t RangeFilter::labels_count() const { return inner_->labels.size(); }
float RangeFilter::threshold() const { return inner_->threshold; }

RangeFilter RangeFilterBuilder::build() {
    if (!model_path_) throw Error(Error::ModelPathRequired, "model path required");
    if (!labels_ && !labels_path_) throw Error(Error::LabelsRequired, "labels required (provide path or vec)");
    auto in = std::make_shared<RangeFilterInner>();
    // labels file: text format, one label per line (rangefilter.rs:228-236)
    in->labels = labels_ ? *labels_ : load_labels_from_file(*labels_path_, ModelType::BirdNetV24);
    if (bn_model_load(model_path_->c_str(), device_, BN_MODEL_GENERIC, &in->model) != BN_OK) throw Error(Error::ModelLoad, "failed to load model: " + backend_error());
    bn_io_info io;
    bn_model_io_info(in->model, &io);
    if (io.n_outputs != 1) throw Error(Error::ModelDetection, "model detection failed: meta model expects 1 output, got " + std::to_string(io.n_outputs));
    bn_model_config cfg;
    bn_model_get_config(in->model, &cfg);
    in->n_out = (size_t)cfg.num_species;
    if (in->labels.size() != in->n_out)
        throw Error(Error::LabelCount, "label count mismatch: model expects " + std::to_string(in->n_out) + ", got " + std::to_string(in->labels.size()), 0, in->n_out,
                    in->labels.size());
    if (cfg.sample_count != 3) throw Error(Error::ModelDetection, "model detection failed: meta model input must be [1, 3] (latitude, longitude, week)");
    if (bn_ctx_create(in->model, 1, BN_CTX_DEFAULT, &in->ctx) != BN_OK) throw Error(Error::ModelLoad, "failed to load model: " + backend_error());
    in->threshold = threshold_;
    RangeFilter f;
    f.inner_ = std::move(in);
    return f;
}

std::vector<LocationScore> RangeFilter::predict(float latitude, float longitude, uint32_t month, uint32_t day) const {
    validate_coordinates(latitude, longitude);
    validate_date(month, day);
    const float input[3] = {latitude, longitude, calculate_week(month, day)};
    const float *segs[1] = {input};
    std::vector<float> data(inner_->n_out);
    {
        std::lock_guard<std::mutex> lk(inner_->mu);
        if (bn_infer(inner_->ctx, segs, 1, data.data(), nullptr, nullptr, 0) != BN_OK) throw rf_inference(backend_error());
    }
    std::vector<LocationScore> scores;
    for (size_t i = 0; i < data.size(); i++)
        if (data[i] >= inner_->threshold && i < inner_->labels.size()) scores.push_back(LocationScore{inner_->labels[i], data[i], i});
    std::stable_sort(scores.begin(), scores.end(), [](const LocationScore &a, const LocationScore &b) { return total_key(a.score) > total_key(b.score); });
    return scores;
}

std::vector<Prediction> RangeFilter::filter_predictions(const std::vector<Prediction> &predictions, const std::vector<LocationScore> &location_scores,
                                                        bool rerank) const {
    return birdnet::filter_predictions(predictions, location_scores, inner_->threshold, rerank);
}

std::vector<std::vector<Prediction>> RangeFilter::filter_batch_predictions(const std::vector<std::vector<Prediction>> &predictions_batch,
                                                                           const std::vector<LocationScore> &location_scores, bool rerank) const {
    std::vector<std::vector<Prediction>> out;
    out.reserve(predictions_batch.size());
    for (const auto &p : predictions_batch) out.push_back(birdnet::filter_predictions(p, location_scores, inner_->threshold, rerank));
    return out;
}

}  // namespace birdnet
