/*
 * oracle/detection.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's model-type detection and of the CLI's
 * segment chunking (the sharding unit of the multi-GPU configuration):
 *   /root/reference/src/detection.rs:15-80    detect_model_type (auto)
 *   /root/reference/src/detection.rs:83-145   build_config_with_override
 *   /root/reference/src/detection.rs:149-174  extract_sample_count / extract_last_dim
 *   /root/reference/src/types.rs:14-44        ModelType::{sample_rate,segment_duration,sample_count}
 *   /root/reference/src/bin/birdnet-analyze.rs:707-743  chunk_audio
 *
 * Pinned by the known-answer tests detection.rs:183-284 and types.rs:194-235,
 * replayed in tests/test_oracle_detection.py.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

enum { MT_BIRDNET_V24 = 0, MT_BIRDNET_V30 = 1, MT_PERCH_V2 = 2 };
enum { DET_OK = 0, DET_ERR = 1 };

typedef struct {
    int32_t model_type;
    uint32_t sample_rate;
    float segment_duration;
    uint64_t sample_count;
    uint64_t num_species;
    int32_t has_embedding;
    uint64_t embedding_dim;
} oracle_config_t;

static uint32_t mt_sample_rate(int mt) { return mt == MT_BIRDNET_V24 ? 48000u : 32000u; }
static float mt_duration(int mt) { return mt == MT_BIRDNET_V24 ? 3.0f : 5.0f; }
static uint64_t mt_samples(int mt) { return mt == MT_BIRDNET_V24 ? 144000u : 160000u; }

/* detection.rs:149-163: rank 2 -> shape[1]; rank 3 -> shape[2]; negative -> error */
static int extract_sample_count(const int64_t *shape, size_t rank, uint64_t *out) {
    int64_t v;
    if (rank == 2) v = shape[1];
    else if (rank == 3) v = shape[2];
    else return DET_ERR;
    if (v < 0) return DET_ERR;
    *out = (uint64_t)v;
    return DET_OK;
}

/* detection.rs:166-174 */
static int extract_last_dim(const int64_t *shape, size_t rank, uint64_t *out) {
    if (rank == 0) return DET_ERR;
    int64_t v = shape[rank - 1];
    if (v < 0) return DET_ERR;
    *out = (uint64_t)v;
    return DET_OK;
}

/*
 * out_shapes: concatenated dims; out_ranks[i] = rank of output i.
 * override_type < 0 means None.
 */
int oracle_detect_model_type(const int64_t *in_shape, size_t in_rank, const int64_t *out_shapes,
                             const size_t *out_ranks, size_t n_out, int override_type,
                             oracle_config_t *cfg) {
    uint64_t sc;
    if (extract_sample_count(in_shape, in_rank, &sc) != DET_OK) return DET_ERR;
    const int64_t *oshape[8];
    size_t off = 0;
    for (size_t i = 0; i < n_out && i < 8; i++) {
        oshape[i] = out_shapes + off;
        off += out_ranks[i];
    }
    memset(cfg, 0, sizeof(*cfg));
    if (override_type >= 0) {
        int mt = override_type;
        if (sc != mt_samples(mt)) return DET_ERR;
        uint64_t ns = 0, ed = 0;
        int has_e = 0;
        if (mt == MT_BIRDNET_V24) {
            if (n_out != 1) return DET_ERR;
            if (extract_last_dim(oshape[0], out_ranks[0], &ns)) return DET_ERR;
        } else if (mt == MT_BIRDNET_V30) {
            if (n_out != 2) return DET_ERR;
            if (extract_last_dim(oshape[0], out_ranks[0], &ed)) return DET_ERR;
            if (extract_last_dim(oshape[1], out_ranks[1], &ns)) return DET_ERR;
            has_e = 1;
        } else {
            if (n_out != 4) return DET_ERR;
            if (extract_last_dim(oshape[0], out_ranks[0], &ed)) return DET_ERR;
            if (extract_last_dim(oshape[3], out_ranks[3], &ns)) return DET_ERR;
            has_e = 1;
        }
        cfg->model_type = mt;
        cfg->sample_rate = mt_sample_rate(mt);
        cfg->segment_duration = mt_duration(mt);
        cfg->sample_count = sc;
        cfg->num_species = ns;
        cfg->has_embedding = has_e;
        cfg->embedding_dim = ed;
        return DET_OK;
    }
    if (sc == 144000 && n_out == 1) {
        uint64_t ns;
        if (extract_last_dim(oshape[0], out_ranks[0], &ns)) return DET_ERR;
        cfg->model_type = MT_BIRDNET_V24;
        cfg->sample_rate = 48000;
        cfg->segment_duration = 3.0f;
        cfg->sample_count = 144000;
        cfg->num_species = ns;
        return DET_OK;
    }
    if (sc == 160000 && (n_out == 2 || n_out == 4)) {
        uint64_t ed, ns;
        size_t pi = n_out == 2 ? 1 : 3;
        if (extract_last_dim(oshape[0], out_ranks[0], &ed)) return DET_ERR;
        if (extract_last_dim(oshape[pi], out_ranks[pi], &ns)) return DET_ERR;
        cfg->model_type = n_out == 2 ? MT_BIRDNET_V30 : MT_PERCH_V2;
        cfg->sample_rate = 32000;
        cfg->segment_duration = 5.0f;
        cfg->sample_count = 160000;
        cfg->num_species = ns;
        cfg->has_embedding = 1;
        cfg->embedding_dim = ed;
        return DET_OK;
    }
    return DET_ERR;
}

/*
 * chunk_audio (birdnet-analyze.rs:707-743).
 *   overlap_samples = (overlap_secs * sample_rate as f32) as usize   (f32 product, truncating)
 *   step = segment_samples.saturating_sub(overlap_samples); step == 0 -> no chunks
 *   one chunk for every pos < len, zero padded; start_time = pos as f32 / sample_rate as f32
 * Returns the number of chunks; when `starts` is non-NULL writes each chunk's
 * start sample and f32 start time (up to `cap` entries).
 */
size_t oracle_chunk_plan(size_t n_samples, size_t segment_samples, float overlap_secs,
                         uint32_t sample_rate, uint64_t *starts, float *start_times, size_t cap) {
    float prod = overlap_secs * (float)sample_rate;
    size_t overlap_samples = prod > 0.0f ? (size_t)prod : 0; /* `as usize` saturates; NaN -> 0 */
    size_t step = segment_samples > overlap_samples ? segment_samples - overlap_samples : 0;
    if (step == 0) return 0;
    size_t n = 0;
    for (size_t pos = 0; pos < n_samples; pos += step) {
        if (starts && n < cap) {
            starts[n] = pos;
            start_times[n] = (float)pos / (float)sample_rate;
        }
        n++;
    }
    return n;
}

/* Materialise chunk `c` (zero padded) into out[segment_samples]. */
void oracle_chunk_fill(const float *samples, size_t n_samples, size_t segment_samples,
                       uint64_t start, float *out) {
    size_t end = start + segment_samples < n_samples ? start + segment_samples : n_samples;
    size_t have = end > start ? end - start : 0;
    memcpy(out, samples + start, have * sizeof(float));
    memset(out + have, 0, (segment_samples - have) * sizeof(float));
}
