/*
 * oracle/postprocess.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's prediction post-processing:
 *   /root/reference/src/postprocess.rs:8-35   ScoreEntry ordering (reversed f32::total_cmp)
 *   /root/reference/src/postprocess.rs:40-87  top_k_predictions
 *   /root/reference/src/postprocess.rs:91-93  sigmoid = 1/(1+exp(-x))
 *
 * The reference drives Rust std's alloc::collections::BinaryHeap (MSRV 1.92,
 * Cargo.toml:5).  Which index survives a tie and in which order equal
 * confidences come out is decided by that container's sift_up /
 * sift_down_to_bottom algorithm and by Vec order of `into_iter()`, followed by
 * the stable `sort_by`.  Those std routines are third-party code that is not
 * in /root/reference; they are restated here from their published algorithm
 * (library/alloc/src/collections/binary_heap/mod.rs: push, pop, sift_up,
 * sift_down_to_bottom; library/core/src/slice/sort/shared/smallsort.rs:
 * insertion_sort_shift_left, which `sort_by` uses for len <= 20).
 *
 * Pinning: every known-answer test in postprocess.rs:101-331 is replayed
 * against this file by tests/test_oracle_postprocess.py.  Tie ORDER is not
 * pinned by any reference test (postprocess.rs:208-219 only checks counts), so
 * for ties this file is "restated std behaviour, unpinned".
 *
 * Rust's f32::exp lowers to the platform libm expf; this file calls the same
 * glibc expf, so confidences are bit-identical to the reference built on this
 * platform.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t index;
    float score;
} entry_t;

/* f32::total_cmp as an unsigned key: larger key == Greater. */
static inline uint32_t total_key(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* ScoreEntry::cmp(self, other) = other.score.total_cmp(self.score)
 * heap_le(a, b)  <=>  a <= b in ScoreEntry order  <=>  key(a.score) >= key(b.score) */
static inline int heap_le(const entry_t *a, const entry_t *b) {
    return total_key(a->score) >= total_key(b->score);
}

/* BinaryHeap::sift_up(start, pos) with the Hole idiom. */
static size_t sift_up(entry_t *d, size_t start, size_t pos) {
    entry_t elt = d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (heap_le(&elt, &d[parent])) break;
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = elt;
    return pos;
}

/* BinaryHeap::sift_down_to_bottom(pos) over d[0..end). */
static void sift_down_to_bottom(entry_t *d, size_t end, size_t pos) {
    size_t start = pos;
    entry_t elt = d[pos];
    size_t child = 2 * pos + 1;
    size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
    while (child <= lim) {
        /* child += (hole.get(child) <= hole.get(child + 1)) as usize */
        child += (size_t)heap_le(&d[child], &d[child + 1]);
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        d[pos] = d[child];
        pos = child;
    }
    d[pos] = elt;
    sift_up(d, start, pos);
}

float oracle_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/*
 * top_k_predictions without the label lookup (labels are a host-side string
 * join on `index`, postprocess.rs:69-72).
 *   has_min == 0  <=>  min_confidence == None
 * Writes up to min(top_k, n) (index, confidence) pairs sorted as the
 * reference sorts them; returns the count.  `scratch` must hold
 * (min(top_k,n)+1) entry_t.
 */
size_t oracle_top_k(const float *logits, size_t n, size_t top_k, int has_min, float min_conf,
                    uint32_t *idx_out, float *conf_out) {
    if (n == 0 || top_k == 0) return 0;
    size_t k = top_k < n ? top_k : n;
    entry_t *heap = (entry_t *)malloc((k + 1) * sizeof(entry_t));
    size_t len = 0;
    for (size_t i = 0; i < n; i++) {
        /* heap.push */
        heap[len].index = (uint32_t)i;
        heap[len].score = logits[i];
        len++;
        sift_up(heap, 0, len - 1);
        if (len > k) {
            /* heap.pop(): item = data.pop(); swap(item, data[0]); sift_down_to_bottom(0) */
            len--;
            entry_t item = heap[len];
            if (len > 0) {
                heap[0] = item; /* the old root is the popped (discarded) value */
                sift_down_to_bottom(heap, len, 0);
            }
        }
    }
    /* into_iter() walks the backing Vec in order; map sigmoid; filter conf >= min */
    size_t m = 0;
    for (size_t i = 0; i < len; i++) {
        float c = oracle_sigmoid(heap[i].score);
        if (has_min && !(c >= min_conf)) continue;
        idx_out[m] = heap[i].index;
        conf_out[m] = c;
        m++;
    }
    free(heap);
    /* sort_by(|a,b| b.conf.partial_cmp(a.conf).unwrap_or(Equal)), stable.
     * is_less(a, b) <=> a.conf > b.conf.  Insertion sort == what std runs for
     * len <= 20; for longer inputs any stable sort agrees whenever the
     * confidences are totally ordered (no NaN), which is the only case std
     * specifies. */
    for (size_t i = 1; i < m; i++) {
        uint32_t ti = idx_out[i];
        float tc = conf_out[i];
        size_t j = i;
        while (j > 0 && tc > conf_out[j - 1]) {
            idx_out[j] = idx_out[j - 1];
            conf_out[j] = conf_out[j - 1];
            j--;
        }
        idx_out[j] = ti;
        conf_out[j] = tc;
    }
    return m;
}

/* Batched convenience wrapper: rows of `n` logits, outputs padded to stride k_stride. */
void oracle_top_k_batch(const float *logits, size_t rows, size_t n, size_t top_k, int has_min,
                        float min_conf, size_t k_stride, uint32_t *idx_out, float *conf_out,
                        uint32_t *count_out) {
    for (size_t r = 0; r < rows; r++) {
        count_out[r] = (uint32_t)oracle_top_k(logits + r * n, n, top_k, has_min, min_conf,
                                              idx_out + r * k_stride, conf_out + r * k_stride);
    }
}

/* ScoreEntry ordering probe for the KATs at postprocess.rs:300-331:
 * returns -1 / 0 / +1 for a.cmp(b) == Less / Equal / Greater. */
int oracle_score_entry_cmp(float a, float b) {
    uint32_t ka = total_key(a), kb = total_key(b);
    /* self.cmp(other) = other.score.total_cmp(self.score) */
    return (kb > ka) - (kb < ka);
}

/* testutil.rs:110-121 random_logits(count, seed): LCG -> [-5, 5] */
void oracle_random_logits(float *out, size_t count, uint64_t seed) {
    uint64_t state = seed;
    for (size_t i = 0; i < count; i++) {
        state = state * 1103515245ull + 12345ull;
        float bits = (float)((state >> 16) & 0xFFFFu);
        out[i] = fmaf(bits, 10.0f / 65535.0f, -5.0f);
    }
}

/* testutil.rs:137-147 mock_embeddings(dim, seed): LCG -> [0, 1] */
void oracle_mock_embeddings(float *out, size_t dim, uint64_t seed) {
    uint64_t state = seed;
    for (size_t i = 0; i < dim; i++) {
        state = state * 1103515245ull + 12345ull;
        float bits = (float)((state >> 16) & 0xFFFFu);
        out[i] = bits / 65535.0f;
    }
}
