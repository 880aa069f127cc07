/*
 * birdnet_hip.h -- C ABI of libbirdnet_hip.so, the MI355X-native (gfx950 / HIP)
 * replacement for the ONNX Runtime session behind the reference's
 * Classifier::predict / predict_batch / BatchInferenceContext path.
 *
 * The reference (tphakala/rust-birdnet-onnx, crate birdnet-onnx 2.0.0-rc.5) has
 * no FFI of its own; its seam is the safe `ort` crate API.  Every entry point
 * below names the `ort` call site (file:line under the reference tree) it
 * stands in for.  Plain pointers and sizes only; no C++/torch types.
 *
 * Threading: a bn_model is immutable after load and may be shared between
 * threads; a bn_ctx owns one HIP stream plus its buffers and must be used by
 * one thread at a time (same contract as BatchInferenceContext,
 * src/batch_context.rs:56-60).
 *
 * Errors: every call returns a bn_status; the message of the last failure on
 * the calling thread is available from bn_last_error().
 */
#ifndef BIRDNET_HIP_H
#define BIRDNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN_ABI_VERSION 2 /* 2: bn_model_get_cost / bn_ctx_get_stats take the caller's struct size; bn_ctx_get_stats, bn_group_get_stats */
#define BN_MAX_OUTPUTS 8
#define BN_MAX_RANK 6
#define BN_NAME_LEN 64

typedef struct bn_model bn_model; /* stands in for ort::session::Session (src/classifier.rs:340-350,435) */
typedef struct bn_ctx bn_ctx;     /* stands in for ort IoBinding + buffers (src/batch_context.rs:70-85) */

typedef enum bn_status {
    BN_OK = 0,
    BN_ERR_INVALID_ARG = 1,       /* NULL handle, B > max_batch, bad index ... */
    BN_ERR_BAD_SIZE = 2,          /* reserved: size checks live in the host shim (classifier.rs:612-618,688-696) */
    BN_ERR_TIMEOUT = 3,           /* -> Error::Timeout   (classifier.rs:568-573) */
    BN_ERR_CANCELLED = 4,         /* -> Error::Cancelled (classifier.rs:568-573) */
    BN_ERR_BACKEND = 5,           /* HIP runtime failure -> Error::Inference(String) */
    BN_ERR_MODEL_LOAD = 6,        /* unreadable / malformed .onnx -> Error::ModelLoad */
    BN_ERR_UNSUPPORTED_MODEL = 7, /* graph uses an operator/shape outside the native subset */
    BN_ERR_MODEL_DETECTION = 8,   /* -> Error::ModelDetection{reason} (detection.rs:15-145) */
    BN_ERR_NO_DEVICE = 9          /* no usable gfx950 device: the path never falls back to CPU */
} bn_status;

/* ModelType (src/types.rs:3-11) */
typedef enum bn_model_type {
    BN_MODEL_BIRDNET_V24 = 0,
    BN_MODEL_BIRDNET_V30 = 1,
    BN_MODEL_PERCH_V2 = 2,
    /* Not an audio model: no detection rules, output 0 is the result.  Used for the range filter's
     * meta model ((lat, lon, week) -> per-species prior, src/rangefilter.rs:451-496); only valid as
     * model_type_override.  sample_count = elements per input row, num_species = last dim of output 0. */
    BN_MODEL_GENERIC = 100
} bn_model_type;

/* Tensor metadata: what session.inputs()/outputs() + dtype().tensor_shape()
 * return (src/classifier.rs:387-420).  A dynamic dimension is reported as -1. */
typedef struct bn_io_info {
    int32_t input_rank;
    int64_t input_shape[BN_MAX_RANK];
    char input_name[BN_NAME_LEN];
    int32_t n_outputs;
    int32_t output_rank[BN_MAX_OUTPUTS];
    int64_t output_shape[BN_MAX_OUTPUTS][BN_MAX_RANK];
    char output_name[BN_MAX_OUTPUTS][BN_NAME_LEN];
} bn_io_info;

/* ModelConfig (src/types.rs:72-85) + which graph outputs carry logits /
 * embeddings (src/classifier.rs:917-934: v2.4 -> 0; v3.0 -> 1,0; Perch -> 3,0). */
typedef struct bn_model_config {
    int32_t model_type; /* bn_model_type */
    uint32_t sample_rate;
    float segment_duration;
    uint64_t sample_count;
    uint64_t num_species;
    int32_t has_embedding;
    uint64_t embedding_dim;
    int32_t logits_output;    /* graph output index of the logits */
    int32_t embedding_output; /* graph output index of the embeddings, -1 if none */
} bn_model_config;

/* Work the loaded graph costs per segment, from the engine's own graph walk
 * (denominators for roofline reporting). */
typedef struct bn_model_cost {
    double macs_mfma;          /* multiply-accumulates issued on matrix cores (1x1 conv / conv1d / FC) */
    double macs_valu;          /* multiply-accumulates on the vector ALU (depthwise, stem conv) */
    double weight_bytes;       /* resident parameter bytes after folding/pruning */
    double activation_bytes;   /* bytes of intermediates written to HBM per segment by the current plan */
    int32_t n_launches;        /* kernel launches per batch in the current plan */
    /* The front end's windowed-DFT filter banks (+ absorbed mel product) counted two ways, SURVEY.md 8(d):
     * multiply-accumulates of the matrix-product evaluation the exporter's graph spells out, and flops of the
     * FFT formulation the plan runs where it can (2.5 L log2 L per real frame + 2 per mel non-zero); both 0
     * when the graph has no such bank.  macs_valu contains fft_flops / 2 for banks that run as FFTs. */
    double dft_gemm_macs;
    double fft_flops;
    /* what the current plan spends on those banks (folded matrix product: half the taps; FFT: fft_flops / 2), and the
     * real-FFT flop count of the same frames (2.5 L log2 L each) whichever way they run: a roofline quoted on
     * 2 x (macs_mfma + macs_valu) counts the former ("flops performed"); replacing it by the latter gives the
     * FFT-normalised count SURVEY.md 8(d) prices the front end at. */
    double dft_performed_macs;
    double dft_fft_equiv_flops;
    /* (appended in round 4; callers pass their struct size) multiply-accumulates of macs_mfma that are RECOMPUTE: the fused
     * MBConv launches expand the halo rows / columns of neighbouring bands and strips again (and the padded k of an opt-in
     * configuration); 2 x (macs_mfma - recompute_macs + macs_valu) is the work the graph asks for. */
    double recompute_macs;
} bn_model_cost;

/* ---- version / device ------------------------------------------------- */
int32_t bn_abi_version(void);
/* Number of visible HIP devices whose arch is gfx950 (0 => every load fails with BN_ERR_NO_DEVICE). */
int32_t bn_device_count(void);

/* ---- model load: Session::builder()...commit_from_file(path)  (classifier.rs:340-350) ---- */
/* model_type_override: -1 for auto-detection, else a bn_model_type (ClassifierBuilder::model_type).
 * The BN_* environment switches (diagnostic A/B knobs) are read when the plan is built AND by the
 * launchers' shape checks: they must not change between bn_model_load and the last launch of that model's contexts.  The
 * plan forms of round 4 (quarter fold, folded GEMMs with absorbed chains, pooled epilogue) have no generic fallback kernel,
 * so a switch flipped in between is a launch error (BN_ERR_BACKEND), never a silently different result. */
bn_status bn_model_load(const char *onnx_path, int32_t device, int32_t model_type_override,
                        bn_model **out);
bn_status bn_model_load_buffer(const void *onnx_bytes, size_t len, int32_t device,
                               int32_t model_type_override, bn_model **out);
/* Drops the caller's reference.  Contexts created from the model keep it alive: the order of
 * bn_model_free and bn_ctx_destroy does not matter (the reference's BatchInferenceContext is an owned
 * value, src/batch_context.rs:70-85). */
void bn_model_free(bn_model *m);
/* HIP device ordinal the model was loaded on (device_id of the reference's GPU configs, cuda_config.rs:179-182); -1 for NULL. */
int32_t bn_model_device(const bn_model *m);
/* session.inputs()/outputs() metadata (classifier.rs:387-420) */
bn_status bn_model_io_info(const bn_model *m, bn_io_info *out);
/* detect_model_type() result for the loaded graph (detection.rs:15-145) */
bn_status bn_model_get_config(const bn_model *m, bn_model_config *out);
/* struct_size = sizeof(bn_model_cost) as the CALLER was compiled: at most that many bytes are written, so a caller
 * built against a shorter struct keeps working when fields are appended. */
bn_status bn_model_get_cost(const bn_model *m, bn_model_cost *out, size_t struct_size);

/* The same detection rules exposed on raw shapes, for shims that keep
 * detection on their side (detection.rs:15-80; override < 0 means None).
 * out_shapes is the concatenation of all output shapes, out_ranks their ranks. */
bn_status bn_detect_model_type(const int64_t *in_shape, size_t in_rank, const int64_t *out_shapes,
                               const size_t *out_ranks, size_t n_out, int32_t model_type_override,
                               bn_model_config *out);

/* ---- context: session.create_binding() + vec![0f32; max*S]  (batch_context.rs:102-133) ---- */
#define BN_CTX_DEFAULT 0u
#define BN_CTX_ALL_OUTPUTS 1u /* also compute graph outputs the reference discards (Perch 1,2) */
#define BN_CTX_NO_GRAPH 2u    /* launch kernels eagerly instead of replaying a captured hipGraph */
bn_status bn_ctx_create(bn_model *m, size_t max_batch, uint32_t flags, bn_ctx **out);
void bn_ctx_destroy(bn_ctx *c);
size_t bn_ctx_max_batch(const bn_ctx *c);
/* How the context's plans reached the stream so far.  A plan is captured into a hipGraph once per (batch size, input
 * buffer) and replayed; every LDS opt-in and allocation the launches need happens at bn_ctx_create, so a capture holds
 * kernel launches only.  Should a capture still fail, that batch runs launch by launch, the event is COUNTED here
 * (capture_fallbacks, with the runtime's message in last_fallback) and printed to stderr once per context;
 * BN_STRICT_GRAPH=1 in the environment makes it BN_ERR_BACKEND instead.  capture_fallbacks must read 0 in a
 * healthy process. */
typedef struct bn_ctx_stats {
    uint64_t captures;          /* hipStreamBeginCapture..EndCapture runs */
    uint64_t instantiates;      /* hipGraphInstantiate calls */
    uint64_t replays;           /* hipGraphLaunch calls */
    uint64_t eager_runs;        /* plans launched kernel by kernel (BN_CTX_NO_GRAPH, or a failed capture) */
    uint64_t capture_fallbacks; /* captures that did not become a graph */
    uint64_t evictions;         /* instantiated graphs dropped from the 16-entry cache */
    uint64_t cached_graphs;
    char last_fallback[192];
    uint64_t input_copies;      /* device-to-device copies of a caller's batch into the context's own input buffer */
} bn_ctx_stats;
bn_status bn_ctx_get_stats(const bn_ctx *c, bn_ctx_stats *out, size_t struct_size);
/* Bytes of device memory held by the context (activations arena + I/O buffers). */
size_t bn_ctx_device_bytes(const bn_ctx *c);

/*
 * The hot call: session.run_with_options / run_binding_with_options
 * (classifier.rs:637-639, 721-723, 851-853) together with the host staging of
 * prepare_input (batch_context.rs:188-226) and the output copies of
 * extract_outputs / extract_tensor_data (batch_context.rs:289-338,
 * classifier.rs:1062-1077).
 *
 *   segs        batch_size pointers to sample_count host floats each
 *   logits_out  host [batch_size * num_species]
 *   emb_out     host [batch_size * embedding_dim], or NULL
 *   cancel      optional flag polled while the batch runs (CancellationToken,
 *               inference_options.rs:24-47); non-zero => BN_ERR_CANCELLED
 *   timeout_ns  0 = none; exceeded => BN_ERR_TIMEOUT (RunOptions::terminate,
 *               classifier.rs:527-554).  Granularity is one launch group.
 * batch_size == 0 returns BN_OK and touches nothing (classifier.rs:681-683).
 */
bn_status bn_infer(bn_ctx *c, const float *const *segs, size_t batch_size, float *logits_out,
                   float *emb_out, const volatile int32_t *cancel, uint64_t timeout_ns);

/*
 * The same hot call split in two, so that the host staging and the PCIe upload of batch k+1 overlap the device
 * work of batch k (a synchronous call leaves the GPU idle while the caller's 576 KB per segment are copied and
 * uploaded, and the link idle while the plan runs):
 *
 *   bn_infer_submit   validates, copies the caller's slices into pinned staging (prepare_input,
 *                     batch_context.rs:188-226; a persistent pool of staging threads copies segment by segment
 *                     and every finished chunk goes on the wire while the next is still being copied), enqueues
 *                     the plan, the top-K kernel (top_k > 0: top_k_predictions, postprocess.rs:40-87, with the
 *                     same has_min / min_conf meaning as bn_topk) and the device-to-host copies of logits,
 *                     embeddings and top-K rows, and returns a ticket.  The caller's slices are no longer
 *                     referenced once it returns.
 *   bn_infer_collect  waits for that batch (cancel / timeout_ns as in bn_infer) and copies its results out:
 *                     logits_out [batch * num_species], emb_out [batch * embedding_dim] or NULL, and -- when
 *                     count_out is not NULL -- idx_out / conf_out [batch * k_stride] and count_out [batch] as
 *                     bn_topk writes them (k_stride >= min(top_k, num_species)).
 *
 * A context holds at most TWO submitted batches (a third submit before a collect is BN_ERR_INVALID_ARG);
 * batches complete in submission order.  More batches in flight = more contexts, each on its own stream.
 * A batch that timed out or was cancelled in bn_infer_collect is abandoned: its ticket is gone and the context
 * drains before its next use.  batch_size == 0 yields ticket 0, which collects to nothing.  Same threading rule
 * as every context call: one thread at a time per context.  Tickets share nothing with the synchronous entry points
 * (bn_infer_windows, bn_step_device, bn_step_windows) but the context's stream: each of the two slots owns its device
 * input and pinned staging, so those calls may be mixed with tickets in flight; they run in call order.
 */
bn_status bn_infer_submit(bn_ctx *c, const float *const *segs, size_t batch_size, size_t top_k,
                          int32_t has_min, float min_conf, uint64_t *ticket);
bn_status bn_infer_collect(bn_ctx *c, uint64_t ticket, float *logits_out, float *emb_out,
                           size_t k_stride, uint32_t *idx_out, float *conf_out, uint32_t *count_out,
                           const volatile int32_t *cancel, uint64_t timeout_ns);

/* Same computation with the batch already resident in HBM as one contiguous
 * [batch_size, sample_count] f32 array; outputs stay on the device.
 * d_pcm must be 16-byte aligned (the kernels read it as float4); a pointer that is
 * not is refused with BN_ERR_INVALID_ARG, nothing is launched.
 * Asynchronous on the context's stream unless `sync` is non-zero. */
bn_status bn_infer_device(bn_ctx *c, const float *d_pcm, size_t batch_size, int32_t sync);
/* The context's OWN device input buffer ([max_batch, sample_count] f32, 256-byte aligned).  The plan always reads its batch
 * from here: one hipGraph per batch size, however many buffers a caller cycles through.  bn_infer_device /
 * bn_step_device copy a batch that lives elsewhere in on the context's stream (device to device, ~10 us for 32 x 3 s);
 * a caller that produces its batch directly in this buffer and passes this pointer pays no copy.  The buffer is also
 * what bn_infer_windows / bn_step_windows fill: do not write it while such a call is in flight. */
bn_status bn_ctx_input_device(const bn_ctx *c, float **d_ptr, size_t *capacity_floats);
/* Device pointer and row length of graph output `index` after the last run
 * (row-major [batch, row_elems] f32). */
bn_status bn_ctx_output_device(const bn_ctx *c, int32_t index, const float **d_ptr,
                               size_t *row_elems);
/* Copy graph output `index` of the last run to host: [batch_size * row_elems]. */
bn_status bn_ctx_read_output(bn_ctx *c, int32_t index, size_t batch_size, float *host_out);
/* Block until the context's stream is idle (IoBinding::synchronize_outputs, batch_context.rs:276-281). */
bn_status bn_ctx_synchronize(bn_ctx *c);
/* The context's hipStream_t, for callers that order their own device work after it. */
void *bn_ctx_stream(const bn_ctx *c);
/* Mean device time per launch group of the last timed run is not kept here;
 * use bn_ctx_time_kernels to measure: runs the plan once for batch_size with a
 * HIP event pair around every launch on the context's stream and writes up to
 * cap (name, microseconds) pairs.  Returns the number of launches. */
size_t bn_ctx_time_kernels(bn_ctx *c, size_t batch_size, char (*names)[BN_NAME_LEN], float *usec,
                           double *macs, double *bytes, size_t cap);
/* The planner's cost figures per launch for batch_size (no device work): multiply-adds on the
 * matrix cores (a fused MBConv launch's expand conv included, with its halo / band recompute),
 * on the vector ALU, the recompute share of the first, and algorithmic bytes (activations in +
 * out + weights).  macs_mfma[k] + macs_valu[k] is what bn_ctx_time_kernels reports as macs[k];
 * summed over the launches they equal bn_model_get_cost's macs_mfma / macs_valu x batch_size.
 * Measurement only; the reference has no counterpart.  Returns the number of
 * launches; arrays may be NULL. */
size_t bn_ctx_launch_costs(const bn_ctx *c, size_t batch_size, double *macs_mfma, double *macs_valu,
                           double *macs_recompute, double *bytes, size_t cap);

/*
 * top_k_predictions (postprocess.rs:40-87) on the device-resident logits of
 * the last run: per row the K = min(top_k, num_species) entries the
 * reference's BinaryHeap keeps, sigmoid (postprocess.rs:91-93), the
 * `confidence >= min_confidence` filter and the stable descending sort.
 * Outputs are host arrays with row stride k_stride >= K:
 *   idx_out[b*k_stride + j], conf_out[b*k_stride + j] for j < count_out[b].
 * has_min == 0 <=> min_confidence == None.
 */
bn_status bn_topk(bn_ctx *c, size_t batch_size, size_t top_k, int32_t has_min, float min_conf,
                  size_t k_stride, uint32_t *idx_out, float *conf_out, uint32_t *count_out);
/* The same kernel on caller-provided device logits [rows, n] (any device buffer). */
bn_status bn_topk_device(int32_t device, const float *d_logits, size_t rows, size_t n, size_t top_k,
                         int32_t has_min, float min_conf, size_t k_stride, uint32_t *idx_out,
                         float *conf_out, uint32_t *count_out);
/* Host logits in, host results out (uploads, runs the kernel, downloads). */
bn_status bn_topk_host(int32_t device, const float *logits, size_t rows, size_t n, size_t top_k,
                       int32_t has_min, float min_conf, size_t k_stride, uint32_t *idx_out,
                       float *conf_out, uint32_t *count_out);

/*
 * One whole pass of the hot path over a device-resident batch, fully asynchronous
 * on the context's stream: the plan (bn_infer_device), the top-K kernel
 * (bn_topk), and the device-to-host copies of the logits rows (raw_scores,
 * classifier.rs:907,948) and of the top-K results into pinned buffers owned by
 * the context.  With sync == 0 the caller overlaps host work and calls
 * bn_ctx_synchronize() before reading bn_step_results().
 */
bn_status bn_step_device(bn_ctx *c, const float *d_pcm, size_t batch_size, size_t top_k,
                         int32_t has_min, float min_conf, int32_t sync);
/* Pinned host views of the last bn_step_device: logits [batch, num_species], idx/conf
 * [batch, k_stride], count [batch].  Valid until the next step on this context. */
bn_status bn_step_results(const bn_ctx *c, const float **logits, const uint32_t **idx,
                          const float **conf, const uint32_t **count, size_t *k_stride);

/*
 * Recording-level ingest: the caller side of the
 * hot path.  The reference CLI reads a 16-bit mono WAV, converts every sample
 * with `f32::from(s) / 32768.0` (src/bin/birdnet-analyze.rs:683-687), cuts it
 * into fixed-length windows with `chunk_audio` (:707-743: step = S -
 * floor(overlap * sample_rate), one window for every pos = k*step < len, the
 * tail zero-padded) and uploads each window as f32.  Here the recording is
 * uploaded ONCE in its storage format (i16: half the PCIe bytes; with overlap no
 * sample crosses the bus twice) and the windows of a batch are materialised on
 * the device, straight into the context's input buffer, by one small kernel in
 * front of the plan.  The conversion is a division by a power of two, so the
 * windows are bit-identical to the reference's.
 */
typedef struct bn_recording bn_recording;
#define BN_PCM_I16 0 /* int16_t mono, value / 32768.0 */
#define BN_PCM_F32 1 /* float mono, used as is */
bn_status bn_recording_create(int32_t device, const void *pcm, size_t n_samples, int32_t format,
                              bn_recording **out);
/* The same, returning at once: a thread of the recording's own uploads `pcm` chunk by chunk (32 MiB; BN_UPLOAD_CHUNK_MB) while
 * the caller already analyses the first windows -- bn_infer_windows / bn_step_windows / bn_recording_windows / bn_recording_read_f32
 * block (on the host) until the last sample they read has arrived, so a loop over the windows in time order overlaps the
 * upload of a long recording with its analysis (a 24 h recording on one GPU: 0.72 s -> the analysis time alone).
 * `pcm` must stay valid and unchanged until bn_recording_wait() has returned or the recording is freed (the reference CLI
 * keeps the whole file in memory for the run, src/bin/birdnet-analyze.rs:653-704). */
bn_status bn_recording_create_async(int32_t device, const void *pcm, size_t n_samples, int32_t format,
                                    bn_recording **out);
/* Blocks until the whole recording is on the device (BN_ERR_BACKEND if the upload failed). */
bn_status bn_recording_wait(const bn_recording *r);
/* Sample-rate conversion front end (the reference CLI refuses a WAV whose rate differs
 * from the model's, src/bin/birdnet-analyze.rs:447-455).  The recording is uploaded in its storage format and
 * converted ON THE DEVICE to f32 at dst_rate by a polyphase windowed-sinc FIR: L/M = dst/src reduced, cutoff
 * 0.5*min(1, L/M) of the source Nyquist band, `zero_crossings` sinc lobes per side (0 => 16) under a Kaiser
 * window (beta 8.6, ~ -90 dB), every phase normalised to unit DC gain.  The result holds
 * ceil(n_samples * L / M) samples and is used like any other recording.  src_rate == dst_rate is a plain upload.
 * There is no reference behaviour to match; the filter design above is the contract (oracle/resample.py). */
bn_status bn_recording_create_resampled(int32_t device, const void *pcm, size_t n_samples, int32_t format,
                                        uint32_t src_rate, uint32_t dst_rate, uint32_t zero_crossings,
                                        bn_recording **out);
/* The polyphase table the resampler uses, for inspection / tests: writes up to cap floats of [L][T] and
 * returns L*T; *L_out, *M_out, *T_out receive the factors.  Needs no device. */
size_t bn_resample_table(uint32_t src_rate, uint32_t dst_rate, uint32_t zero_crossings, float *table, size_t cap,
                         uint32_t *L_out, uint32_t *M_out, uint32_t *T_out);
/* Copy samples [first, first+count) of an f32 recording back to the host (tests, diagnostics). */
bn_status bn_recording_read_f32(const bn_recording *r, size_t first, size_t count, float *host_out);
void bn_recording_free(bn_recording *r);
size_t bn_recording_samples(const bn_recording *r);
/* Number of windows chunk_audio produces for n_samples at this step (0 for an
 * empty recording or step == 0). */
size_t bn_chunk_count(size_t n_samples, size_t step_samples);
/* chunk_audio on the device, copied back: windows [first, first+count) as host
 * f32 [count, segment_samples] (segment_samples % 4 == 0). */
bn_status bn_recording_windows(const bn_recording *r, size_t segment_samples, size_t step_samples,
                               size_t first_window, size_t count, float *host_out);
/* bn_infer over windows [first_window, first_window+count) of the recording
 * (count <= max_batch; the window length is the model's sample_count).  Same
 * outputs, cancel and timeout behaviour as bn_infer. */
bn_status bn_infer_windows(bn_ctx *c, const bn_recording *r, size_t step_samples,
                           size_t first_window, size_t count, float *logits_out, float *emb_out,
                           const volatile int32_t *cancel, uint64_t timeout_ns);

/* bn_step_device over windows of an uploaded recording: window kernel + plan + top-K + D2H of
 * logits and top-K into the context's pinned buffers, asynchronous unless sync != 0; read with
 * bn_step_results after bn_ctx_synchronize.  This is the loop body of a recording analysis
 * (src/bin/birdnet-analyze.rs:556-600) with several contexts in flight. */
bn_status bn_step_windows(bn_ctx *c, const bn_recording *r, size_t step_samples, size_t first_window,
                          size_t count, size_t top_k, int32_t has_min, float min_conf, int32_t sync);

/* Device view of the packed top-K rows of the last bn_step_device / bn_step_windows on this context:
 * [idx: m*k][conf: m*k (float bits)][count: m] for the m rows and k = min(top_k, num_species) of that step.
 * Valid until the next step; reading it must be ordered after the step on bn_ctx_stream(). */
bn_status bn_ctx_step_device_rows(const bn_ctx *c, const uint32_t **d_rows);

/*
 * Multi-GPU (BASELINE.json configs[4]).  The reference has no multi-device code -- its only
 * knob is device_id (src/cuda_config.rs:179-182, src/tensorrt_config.rs:273-276) -- so the contract is "identical to
 * running every window on one device": windows of chunk_audio (src/bin/birdnet-analyze.rs:707-743) are sharded by
 * contiguous range, rank r of R owns [r * ceil(G/R), min(G, (r+1) * ceil(G/R))) (bn_shard_range).
 *
 * bn_group_create takes one model replica per device (load the same file once per device with bn_model_load) and
 * creates contexts_per_device contexts (streams) of max_batch on each.  bn_group_analyze_recording cuts the recording
 * (host int16 / f32 samples) into windows of the model's sample_count at step_samples, lets every rank upload and
 * analyse only its own slice (one host thread per device inside the call), and assembles the results with ONE
 * all-gather of the [G, num_species] logits (logits_out != NULL) and one of the packed top-K rows (count_out != NULL)
 * -- RCCL ncclAllGather over xGMI when the ranks sit on distinct devices (librccl is loaded on first use), plain
 * device copies when they share one (tests on a single GPU).  Outputs are host arrays in window order:
 * logits_out [G * num_species], idx_out / conf_out [G * k_stride], count_out [G]; *n_windows_out = G.
 * A step of 0 (overlap >= segment) yields G = 0 as chunk_audio does.  Errors: bn_group_last_error().
 */
typedef struct bn_group bn_group;
bn_status bn_group_create(bn_model *const *models, const int32_t *devices, int32_t n, size_t max_batch,
                          int32_t contexts_per_device, bn_group **out);
void bn_group_destroy(bn_group *g);
int32_t bn_group_size(const bn_group *g);
int32_t bn_group_uses_rccl(const bn_group *g);
/* bn_ctx_get_stats summed over every context of the group (last_fallback: the most recent one); capture_fallbacks must be 0. */
bn_status bn_group_get_stats(const bn_group *g, bn_ctx_stats *out, size_t struct_size);
void bn_shard_range(size_t n_windows, int32_t rank, int32_t world, size_t *lo, size_t *hi);
bn_status bn_group_analyze_recording(bn_group *g, const void *pcm, size_t n_samples, int32_t format,
                                     size_t step_samples, size_t top_k, int32_t has_min, float min_conf,
                                     float *logits_out, size_t k_stride, uint32_t *idx_out, float *conf_out,
                                     uint32_t *count_out, size_t *n_windows_out);
size_t bn_group_last_error(char *buf, size_t cap);

/* Diagnostic, needs no device: parse the file, build the launch plan (all graph
 * outputs when all_outputs != 0, else logits + embeddings only) and write a
 * text description (one line per launch, then totals) into buf.  Returns the
 * number of bytes the full text needs (excluding the NUL); *status receives the
 * outcome of the parse / detection / planning steps. */
/* First contact with a model file, needs no device: opset, graph input / outputs, what detect_model_type decides (detection.rs:15-145),
 * every operator type with its node count and whether the lowering has a rule for it ("NOT MAPPED" + the first such node), and the
 * outcome of planning -- for every graph output and for the audio path (logits + embeddings) -- with the refusal's node and reason.
 * The reference loads whatever ONNX Runtime loads (classifier.rs:340-350); this says in one call what stands between a real export
 * and the native path.  *status: BN_OK, BN_ERR_UNSUPPORTED_MODEL (a plan was refused) or BN_ERR_MODEL_LOAD (unreadable file).
 * Returns the number of bytes the full text needs (excluding the NUL). */
size_t bn_model_survey(const char *onnx_path, char *buf, size_t cap, bn_status *status);

/* How launches size their grids where a choice exists between one launch's latency and the work per block (chunks per block of the
 * small-map MBConv kernels, row tiles of the 1x1-conv GEMMs).  BN_SHARING_ALONE (default): for a device the launch has to itself -- the
 * lowest latency of one batch.  BN_SHARING_SHARED: for a device whose CUs are shared by several batches in flight -- the reference's way
 * to throughput is several BatchInferenceContexts (classifier.rs:826-867); a launch then gets a share of the CUs whatever its grid, and
 * bigger blocks amortise their prologues and weight traffic (BirdNET v2.4, four contexts: +7 % segments/s; one batch alone: +13 % time).
 * BN_SHARING_AUTO: SHARED while more than one context is alive on the device.  Results do not depend on the mode (bit for bit);
 * process-wide; graphs captured under one form are re-captured under the other.  No ort counterpart: ORT has no such notion. */
#define BN_SHARING_AUTO (-1)
#define BN_SHARING_ALONE 0
#define BN_SHARING_SHARED 1
void bn_set_sharing_mode(int32_t mode);
size_t bn_plan_describe(const char *onnx_path, int32_t model_type_override, int32_t all_outputs,
                        char *buf, size_t cap, bn_status *status);

/* Message of the last failing call on this thread; returns its length. */
size_t bn_last_error(char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* BIRDNET_HIP_H */
