/*
 * birdnet_host.h -- host-side mirror of the reference's public inference API
 * in C++17, layered on the C ABI of birdnet_hip.h.
 *
 * The reference's host language is Rust and no Rust toolchain exists in the
 * build image, so the code a Rust maintainer would keep on their side of the
 * FFI (Classifier, ClassifierBuilder, BatchInferenceContext, InferenceOptions,
 * CancellationToken, Error, Prediction, PredictionResult) is written here in
 * C++ with the same names, argument meaning and error behaviour:
 *
 *   birdnet::Classifier              src/classifier.rs:436-867
 *   birdnet::ClassifierBuilder       src/classifier.rs:46-383
 *   birdnet::BatchInferenceContext   src/batch_context.rs:70-165
 *   birdnet::InferenceOptions        src/inference_options.rs:73-114
 *   birdnet::CancellationToken       src/inference_options.rs:24-47
 *   birdnet::Error                   src/error.rs:6-128 (variants on this path)
 *   birdnet::ModelConfig / Prediction / PredictionResult   src/types.rs:72-109
 *
 * The second half of this header is a flat C ABI over those classes (bnh_*),
 * used by the ctypes test/bench harness so that the parity tests exercise the
 * compiled host code rather than a Python re-implementation.
 */
#ifndef BIRDNET_HOST_H
#define BIRDNET_HOST_H

#include "birdnet_hip.h"

#ifdef __cplusplus
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace birdnet {

enum class ModelType { BirdNetV24 = 0, BirdNetV30 = 1, PerchV2 = 2 };
uint32_t sample_rate(ModelType t);
float segment_duration(ModelType t);
size_t sample_count(ModelType t);
bool has_embeddings(ModelType t);

/* ExecutionProviderInfo (src/types.rs:124-185): only the two values this build can report. */
enum class ExecutionProviderInfo { Cpu, Rocm };
const char *as_str(ExecutionProviderInfo p);
const char *category(ExecutionProviderInfo p);

struct ModelConfig {
    ModelType model_type;
    uint32_t sample_rate;
    float segment_duration;
    size_t sample_count;
    size_t num_species;
    std::optional<size_t> embedding_dim;
};

struct Prediction {
    std::string species;
    float confidence;
    size_t index;
};

struct PredictionResult {
    ModelType model_type;
    std::vector<Prediction> predictions;
    std::optional<std::vector<float>> embeddings;
    std::vector<float> raw_scores;
};

/* Error (src/error.rs): `kind` selects the variant, what() is the Display text. */
class Error : public std::runtime_error {
   public:
    enum Kind {
        InputSize,       /* expected, got */
        BatchInputSize,  /* index, expected, got */
        ModelDetection,
        LabelCount,      /* expected, got */
        ModelPathRequired,
        LabelsRequired,
        ModelLoad,
        LabelLoad,
        LabelParse,
        Inference,
        Timeout,         /* duration_ns */
        Cancelled,
        InvalidCoordinates,   /* latitude, longitude */
        InvalidDate,          /* month, day */
        RangeFilterInference,
    };
    Error(Kind k, const std::string &msg, size_t index = 0, size_t expected = 0, size_t got = 0, uint64_t duration_ns = 0)
        : std::runtime_error(msg), kind(k), index(index), expected(expected), got(got), duration_ns(duration_ns) {}
    Kind kind;
    size_t index, expected, got;
    uint64_t duration_ns;
    float latitude = 0.0f, longitude = 0.0f;  /* InvalidCoordinates */
    uint32_t month = 0, day = 0;              /* InvalidDate */
};

class CancellationToken {
   public:
    CancellationToken() : flag_(std::make_shared<std::atomic<int32_t>>(0)) {}
    /* A token that observes a caller-owned flag (FFI callers hand over a plain int32). */
    static CancellationToken alias(volatile int32_t *external) {
        CancellationToken t;
        t.flag_ = std::shared_ptr<std::atomic<int32_t>>(reinterpret_cast<std::atomic<int32_t> *>(const_cast<int32_t *>(external)),
                                                        [](std::atomic<int32_t> *) {});
        return t;
    }
    void cancel() const { flag_->store(1, std::memory_order_seq_cst); }
    bool is_cancelled() const { return flag_->load(std::memory_order_seq_cst) != 0; }
    const volatile int32_t *raw() const { return reinterpret_cast<const volatile int32_t *>(flag_.get()); }

   private:
    std::shared_ptr<std::atomic<int32_t>> flag_;
};

struct InferenceOptions {
    std::optional<std::chrono::nanoseconds> timeout;
    std::optional<CancellationToken> cancellation_token;
    static InferenceOptions with_timeout_of(std::chrono::nanoseconds d) {
        InferenceOptions o;
        o.timeout = d;
        return o;
    }
    InferenceOptions &with_timeout(std::chrono::nanoseconds d) {
        timeout = d;
        return *this;
    }
    InferenceOptions &with_cancellation_token(CancellationToken t) {
        cancellation_token = std::move(t);
        return *this;
    }
    bool needs_monitor() const { return timeout.has_value() || cancellation_token.has_value(); }
};

class Classifier;

/* BatchInferenceContext (src/batch_context.rs:70-165): not thread-safe, one per thread. */
class BatchInferenceContext {
   public:
    ~BatchInferenceContext();
    BatchInferenceContext(const BatchInferenceContext &) = delete;
    BatchInferenceContext &operator=(const BatchInferenceContext &) = delete;
    size_t max_batch_size() const { return max_batch_size_; }
    size_t sample_count() const { return sample_count_; }
    size_t input_buffer_capacity() const { return max_batch_size_ * sample_count_; }
    size_t input_buffer_bytes() const { return input_buffer_capacity() * sizeof(float); }
    ModelType model_type() const { return model_type_; }
    /* Native extension: graph output `index` of the last batch, row-major
     * [batch, row_elems] -- e.g. Perch's spatial embedding (1) and spectrogram (2), which the reference
     * discards.  Only for contexts made by create_native_batch_context(.., all_outputs = true). */
    std::vector<float> read_output(int index, size_t batch, size_t *row_elems = nullptr) const;

   private:
    friend class Classifier;
    BatchInferenceContext() = default;
    bn_ctx *ctx_ = nullptr;
    size_t max_batch_size_ = 0, sample_count_ = 0;
    ModelType model_type_ = ModelType::BirdNetV24;
};

struct ClassifierInner;

/* A mono recording uploaded to the device once, in its storage format: the native form of the
 * reference CLI's read_wav (i16 / 32768.0, src/bin/birdnet-analyze.rs:683-687) + chunk_audio
 * (:707-743).  Windows are cut on the device (bn_infer_windows). */
class Recording {
   public:
    /* format: BN_PCM_I16 or BN_PCM_F32; throws Error::Inference when the upload fails */
    Recording(const void *pcm, size_t n_samples, int32_t format, int device = 0);
    static Recording from_i16(const int16_t *pcm, size_t n_samples, int device = 0) { return Recording(pcm, n_samples, BN_PCM_I16, device); }
    static Recording from_f32(const float *pcm, size_t n_samples, int device = 0) { return Recording(pcm, n_samples, BN_PCM_F32, device); }
    size_t n_samples() const { return n_samples_; }
    const bn_recording *raw() const { return rec_.get(); }

   private:
    std::shared_ptr<bn_recording> rec_;
    size_t n_samples_ = 0;
};

/* One chunk of a recording: start time as chunk_audio reports it + the prediction. */
struct ChunkResult {
    float start_time;
    PredictionResult result;
};

class Classifier {
   public:
    const ModelConfig &config() const;
    const std::vector<std::string> &labels() const;
    ExecutionProviderInfo requested_provider() const;
    /* src/classifier.rs:610-643 */
    PredictionResult predict(const float *segment, size_t len, const InferenceOptions &options = {}) const;
    /* src/classifier.rs:676-727; segments[i] has lens[i] samples */
    std::vector<PredictionResult> predict_batch(const float *const *segments, const size_t *lens, size_t n,
                                                const InferenceOptions &options = {}) const;
    /* src/classifier.rs:777-792 (PerchV2 => Error::Inference, batch_context.rs:107-114) */
    std::unique_ptr<BatchInferenceContext> create_batch_context(size_t max_batch_size) const;
    /* Native extension: the same context for EVERY model family (the reference refuses PerchV2,
     * batch_context.rs:107-114; the native context has no such limit), optionally computing all graph
     * outputs so that read_output() can return the ones the reference drops. */
    std::unique_ptr<BatchInferenceContext> create_native_batch_context(size_t max_batch_size, bool all_outputs = false) const;
    /* src/classifier.rs:826-867 */
    std::vector<PredictionResult> predict_batch_with_context(BatchInferenceContext &ctx, const float *const *segments, const size_t *lens,
                                                             size_t n, const InferenceOptions &options = {}) const;
    /* The CLI's loop body (src/bin/birdnet-analyze.rs:556-600) over chunks [first_chunk, first_chunk+count)
     * of chunk_audio(recording, overlap): batches of ctx.max_batch_size() windows cut on the device.
     * count == SIZE_MAX means "to the end".  overlap >= segment duration => Error::Inference. */
    std::vector<ChunkResult> predict_recording(BatchInferenceContext &ctx, const Recording &rec, float overlap_secs, size_t first_chunk = 0,
                                               size_t count = (size_t)-1, const InferenceOptions &options = {}) const;

   private:
    friend class ClassifierBuilder;
    std::shared_ptr<ClassifierInner> inner_;
};

class ClassifierBuilder {
   public:
    ClassifierBuilder &model_path(std::string p) { model_path_ = std::move(p); return *this; }
    ClassifierBuilder &labels_path(std::string p) { labels_path_ = std::move(p); labels_.reset(); return *this; }
    ClassifierBuilder &labels(std::vector<std::string> l) { labels_ = std::move(l); labels_path_.reset(); return *this; }
    ClassifierBuilder &model_type(ModelType t) { model_type_ = t; return *this; }
    ClassifierBuilder &top_k(size_t k) { top_k_ = k; return *this; }
    ClassifierBuilder &min_confidence(float c) { min_confidence_ = c; return *this; }
    /* with_rocm() (src/classifier.rs:287-292) selects the native MI355X path; device = HIP ordinal. */
    ClassifierBuilder &with_rocm(int device = 0) { device_ = device; provider_ = ExecutionProviderInfo::Rocm; return *this; }
    Classifier build();  /* src/classifier.rs:334-383 */

   private:
    std::optional<std::string> model_path_, labels_path_;
    std::optional<std::vector<std::string>> labels_;
    std::optional<ModelType> model_type_;
    size_t top_k_ = 10;                    /* classifier.rs:72 */
    std::optional<float> min_confidence_;  /* classifier.rs:73 */
    int device_ = 0;
    ExecutionProviderInfo provider_ = ExecutionProviderInfo::Cpu; /* classifier.rs:71 */
};

/* labels.rs:22-81: text (one per line, trimmed, blanks dropped) for v2.4, CSV first column for v3.0/Perch. */
std::vector<std::string> load_labels_from_file(const std::string &path, ModelType t);
std::vector<std::string> parse_text_labels(const std::string &content);
std::vector<std::string> parse_csv_labels(const std::string &content);
/* labels.rs:95-121: ["a","b"]  |  {"labels": ["a","b"]}  |  [{"name"|"label"|"species": "a"}, ...]; anything else
 * => Error::LabelParse("unrecognized JSON format: ...") */
std::vector<std::string> parse_json_labels(const std::string &content);
enum class LabelFormat { Text = 0, Csv = 1, Json = 2 };  /* src/types.rs LabelFormat */
std::vector<std::string> parse_labels(const std::string &content, LabelFormat format);  /* labels.rs:33-39 */

/* ---- range filter (src/rangefilter.rs): location / date prior over species from a small meta model ---- */
struct LocationScore {  /* src/types.rs:111-120 */
    std::string species;
    float score;
    size_t index;
};
/* rangefilter.rs:77-81: 48-week year, week = (month-1)*4 + (day-1)/7 + 1 */
float calculate_week(uint32_t month, uint32_t day);
/* rangefilter.rs:91-133: throw Error::InvalidCoordinates / Error::InvalidDate */
void validate_coordinates(float latitude, float longitude);
void validate_date(uint32_t month, uint32_t day);
/* rangefilter.rs:333-386: species in the meta model with score >= threshold are kept (confidence *= score when
 * reranking), those below are dropped, species unknown to the meta model are kept unchanged; descending re-sort
 * when reranking (equal confidences keep their order; the reference's sort_unstable leaves that open). */
std::vector<Prediction> filter_predictions(const std::vector<Prediction> &predictions, const std::vector<LocationScore> &location_scores,
                                           float threshold, bool rerank);

struct RangeFilterInner;
class RangeFilterBuilder;
class RangeFilter {
   public:
    static RangeFilterBuilder builder();
    /* rangefilter.rs:435-502: validate, week, run the meta model on [lat, lon, week], keep scores >= threshold,
     * sort descending.  The model runs on the MI355X through the same engine (BN_MODEL_GENERIC). */
    std::vector<LocationScore> predict(float latitude, float longitude, uint32_t month, uint32_t day) const;
    std::vector<Prediction> filter_predictions(const std::vector<Prediction> &predictions, const std::vector<LocationScore> &location_scores,
                                               bool rerank) const;
    std::vector<std::vector<Prediction>> filter_batch_predictions(const std::vector<std::vector<Prediction>> &predictions_batch,
                                                                  const std::vector<LocationScore> &location_scores, bool rerank) const;
    size_t labels_count() const;
    float threshold() const;

   private:
    friend class RangeFilterBuilder;
    std::shared_ptr<RangeFilterInner> inner_;
};

class RangeFilterBuilder {  /* rangefilter.rs:142-277 */
   public:
    RangeFilterBuilder &model_path(std::string p) { model_path_ = std::move(p); return *this; }
    RangeFilterBuilder &labels_path(std::string p) { labels_path_ = std::move(p); labels_.reset(); return *this; }
    RangeFilterBuilder &labels(std::vector<std::string> l) { labels_ = std::move(l); labels_path_.reset(); return *this; }
    RangeFilterBuilder &from_classifier_labels(const std::vector<std::string> &l) { return labels(l); }
    RangeFilterBuilder &threshold(float t) { threshold_ = t; return *this; }
    RangeFilterBuilder &with_rocm(int device = 0) { device_ = device; return *this; }
    RangeFilter build();

   private:
    std::optional<std::string> model_path_, labels_path_;
    std::optional<std::vector<std::string>> labels_;
    float threshold_ = 0.01f;  /* rangefilter.rs:165 */
    int device_ = 0;
};

/* chunk_audio (src/bin/birdnet-analyze.rs:707-743): start sample + start time of every chunk. */
struct Chunk {
    size_t start;
    float start_time;
};
std::vector<Chunk> chunk_plan(size_t n_samples, size_t segment_samples, float overlap_secs, uint32_t sample_rate);

}  // namespace birdnet

extern "C" {
#endif /* __cplusplus */

/* ---------------- flat C ABI over the C++ mirror (test / bench harness) ---------------- */
typedef struct bnh_classifier bnh_classifier;
typedef struct bnh_context bnh_context;
typedef struct bnh_results bnh_results; /* Vec<PredictionResult> */

/* Error kinds, same order as birdnet::Error::Kind; 0 = success. */
enum {
    BNH_OK = 0,
    BNH_ERR_INPUT_SIZE = 1,
    BNH_ERR_BATCH_INPUT_SIZE = 2,
    BNH_ERR_MODEL_DETECTION = 3,
    BNH_ERR_LABEL_COUNT = 4,
    BNH_ERR_MODEL_PATH_REQUIRED = 5,
    BNH_ERR_LABELS_REQUIRED = 6,
    BNH_ERR_MODEL_LOAD = 7,
    BNH_ERR_LABEL_LOAD = 8,
    BNH_ERR_LABEL_PARSE = 9,
    BNH_ERR_INFERENCE = 10,
    BNH_ERR_TIMEOUT = 11,
    BNH_ERR_CANCELLED = 12,
    BNH_ERR_INVALID_COORDINATES = 13,
    BNH_ERR_INVALID_DATE = 14,
    BNH_ERR_RANGE_FILTER_INFERENCE = 15,
    BNH_ERR_OTHER = 16
};
typedef struct bnh_error {
    int32_t kind;
    uint64_t index, expected, got, duration_ns;
    char message[512];
    float latitude, longitude; /* InvalidCoordinates */
    uint32_t month, day;       /* InvalidDate */
} bnh_error;

/* Builder in one call.  labels: NULL => labels_path is used (either may be NULL => LabelsRequired).
 * model_type < 0 => auto; top_k: value of .top_k(); has_min/min_conf: .min_confidence(). */
int32_t bnh_classifier_build(const char *model_path, const char *labels_path, const char *const *labels, size_t n_labels,
                             int32_t model_type, int64_t top_k, int32_t has_min, float min_conf, int32_t device,
                             bnh_classifier **out, bnh_error *err);
void bnh_classifier_free(bnh_classifier *c);
void bnh_classifier_config(const bnh_classifier *c, bn_model_config *out);
const char *bnh_classifier_provider(const bnh_classifier *c);
size_t bnh_classifier_label_count(const bnh_classifier *c);
const char *bnh_classifier_label(const bnh_classifier *c, size_t i);

/* timeout_ns < 0 => None; cancel NULL => None (else a flag the caller may set from another thread). */
int32_t bnh_predict(const bnh_classifier *c, const float *segment, size_t len, int64_t timeout_ns, const volatile int32_t *cancel,
                    bnh_results **out, bnh_error *err);
int32_t bnh_predict_batch(const bnh_classifier *c, const float *const *segments, const size_t *lens, size_t n, int64_t timeout_ns,
                          const volatile int32_t *cancel, bnh_results **out, bnh_error *err);
int32_t bnh_create_batch_context(const bnh_classifier *c, size_t max_batch, bnh_context **out, bnh_error *err);
void bnh_context_free(bnh_context *ctx);
size_t bnh_context_max_batch_size(const bnh_context *ctx);
size_t bnh_context_sample_count(const bnh_context *ctx);
size_t bnh_context_input_buffer_capacity(const bnh_context *ctx);
size_t bnh_context_input_buffer_bytes(const bnh_context *ctx);
int32_t bnh_context_model_type(const bnh_context *ctx);
int32_t bnh_predict_batch_with_context(const bnh_classifier *c, bnh_context *ctx, const float *const *segments, const size_t *lens, size_t n,
                                       int64_t timeout_ns, const volatile int32_t *cancel, bnh_results **out, bnh_error *err);

/* Classifier::predict_recording over a host recording (uploaded once, windows cut on the device);
 * start_times[i] receives chunk i's start time for i < times_cap. */
int32_t bnh_create_native_batch_context(const bnh_classifier *c, size_t max_batch, int32_t all_outputs, bnh_context **out, bnh_error *err);
/* returns the number of floats needed (batch * row_elems); writes up to cap of them */
size_t bnh_context_read_output(const bnh_context *ctx, int32_t index, size_t batch, float *out, size_t cap, size_t *row_elems, bnh_error *err);
int32_t bnh_predict_recording(const bnh_classifier *c, bnh_context *ctx, const void *pcm, size_t n_samples, int32_t format, float overlap_secs,
                              size_t first_chunk, size_t count, int64_t timeout_ns, const volatile int32_t *cancel, bnh_results **out,
                              float *start_times, size_t times_cap, bnh_error *err);
size_t bnh_results_len(const bnh_results *r);
int32_t bnh_result_model_type(const bnh_results *r, size_t i);
size_t bnh_result_n_predictions(const bnh_results *r, size_t i);
const char *bnh_result_species(const bnh_results *r, size_t i, size_t j);
float bnh_result_confidence(const bnh_results *r, size_t i, size_t j);
size_t bnh_result_index(const bnh_results *r, size_t i, size_t j);
size_t bnh_result_raw_scores(const bnh_results *r, size_t i, const float **data);
/* returns 0 and leaves *data NULL when embeddings is None */
size_t bnh_result_embeddings(const bnh_results *r, size_t i, const float **data);
void bnh_results_free(bnh_results *r);

/* ---- range filter ---- */
typedef struct bnh_range_filter bnh_range_filter;
float bnh_calculate_week(uint32_t month, uint32_t day);
int32_t bnh_validate_coordinates(float latitude, float longitude, bnh_error *err);
int32_t bnh_validate_date(uint32_t month, uint32_t day, bnh_error *err);
/* labels: NULL => labels_path (either may be NULL => LabelsRequired); threshold < 0 => default 0.01 */
int32_t bnh_range_filter_build(const char *model_path, const char *labels_path, const char *const *labels, size_t n_labels, float threshold,
                               int32_t device, bnh_range_filter **out, bnh_error *err);
void bnh_range_filter_free(bnh_range_filter *f);
/* RangeFilter::predict: writes up to cap (index, score) pairs, sorted descending; *n_out = number of scores */
int32_t bnh_range_filter_predict(const bnh_range_filter *f, float latitude, float longitude, uint32_t month, uint32_t day, uint32_t *idx_out,
                                 float *score_out, size_t cap, size_t *n_out, bnh_error *err);
const char *bnh_range_filter_label(const bnh_range_filter *f, size_t i);
/* filter_predictions_impl on parallel arrays (species by name): returns the number of survivors, their positions
 * in the input in keep_pos and their confidences in conf_out */
size_t bnh_filter_predictions(const char *const *pred_species, const float *pred_conf, size_t n_pred, const char *const *loc_species,
                              const float *loc_score, size_t n_loc, float threshold, int32_t rerank, uint32_t *keep_pos, float *conf_out);

/* labels.rs parsers and chunk_audio, for host-logic tests */
size_t bnh_parse_labels(const char *content, int32_t csv, char *out, size_t cap); /* '\n'-joined; returns needed bytes */
/* parse_labels with a LabelFormat (0 text, 1 csv, 2 json): labels joined by '\x1f' (labels may contain newlines in
 * JSON); returns the needed bytes, or 0 with *err filled on Error::LabelParse */
size_t bnh_parse_labels_format(const char *content, int32_t format, char *out, size_t cap, bnh_error *err);
size_t bnh_chunk_plan(size_t n_samples, size_t segment_samples, float overlap_secs, uint32_t sample_rate, uint64_t *starts,
                      float *start_times, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* BIRDNET_HOST_H */
