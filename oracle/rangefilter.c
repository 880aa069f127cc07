/*
 * oracle/rangefilter.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's location/date range filter host logic:
 *   /root/reference/src/rangefilter.rs:77-81    calculate_week (48-week year, 4 weeks per month)
 *   /root/reference/src/rangefilter.rs:91-133   validate_coordinates / validate_date
 *   /root/reference/src/rangefilter.rs:477-496  predict: threshold filter + descending sort of the scores
 *   /root/reference/src/rangefilter.rs:333-386  filter_predictions_impl (lookup by species, drop / keep /
 *                                               rerank, descending re-sort when reranking)
 * Species are integer ids here (the harness maps names to ids); the reference keys its lookup by the
 * species string, and a later duplicate in location_scores overrides an earlier one (HashMap collect).
 * Both sorts are `sort_unstable_by(total_cmp)` in the reference: the order of EQUAL keys is not
 * specified there; this restatement (and the product) keep equal keys in input order.
 *
 * Pinned by the reference's known-answer tests rangefilter.rs:586-935, replayed in
 * tests/test_oracle_rangefilter.py.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* rangefilter.rs:77-81 */
float oracle_calculate_week(uint32_t month, uint32_t day) {
    uint32_t weeks_from_months = (month - 1u) * 4u;
    uint32_t week_in_month = (day - 1u) / 7u + 1u;
    return (float)(weeks_from_months + week_in_month);
}

/* rangefilter.rs:91-107: 0 = ok, 1 = latitude out of [-90, 90], 2 = longitude out of [-180, 180]
 * (RangeInclusive::contains: NaN is outside) */
int oracle_validate_coordinates(float latitude, float longitude) {
    if (!(latitude >= -90.0f && latitude <= 90.0f)) return 1;
    if (!(longitude >= -180.0f && longitude <= 180.0f)) return 2;
    return 0;
}

/* rangefilter.rs:118-133: 0 = ok, 1 = month out of [1, 12], 2 = day out of [1, 31] */
int oracle_validate_date(uint32_t month, uint32_t day) {
    if (month < 1u || month > 12u) return 1;
    if (day < 1u || day > 31u) return 2;
    return 0;
}

/* f32::total_cmp key: monotone map of the bit pattern to an unsigned integer */
static uint32_t total_key(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* stable insertion sort, descending by total_cmp of key[] */
static void sort_desc(uint32_t *ids, float *key, size_t n) {
    for (size_t i = 1; i < n; i++) {
        uint32_t id = ids[i];
        float k = key[i];
        size_t j = i;
        while (j > 0 && total_key(key[j - 1]) < total_key(k)) {
            ids[j] = ids[j - 1];
            key[j] = key[j - 1];
            j--;
        }
        ids[j] = id;
        key[j] = k;
    }
}

/* rangefilter.rs:477-496: keep (i, score) with score >= threshold && i < n_labels, sort descending */
size_t oracle_location_scores(const float *scores, size_t n, size_t n_labels, float threshold, uint32_t *idx_out, float *score_out) {
    size_t m = 0;
    for (size_t i = 0; i < n; i++)
        if (scores[i] >= threshold && i < n_labels) {
            idx_out[m] = (uint32_t)i;
            score_out[m] = scores[i];
            m++;
        }
    sort_desc(idx_out, score_out, m);
    return m;
}

/* rangefilter.rs:333-386.  keep_pos[j] = position of the j-th surviving prediction in the input,
 * conf_out[j] its (possibly reranked) confidence. */
size_t oracle_filter_predictions(const uint32_t *pred_species, const float *pred_conf, size_t n_pred, const uint32_t *loc_species,
                                 const float *loc_score, size_t n_loc, float threshold, int rerank, uint32_t *keep_pos, float *conf_out) {
    size_t m = 0;
    for (size_t p = 0; p < n_pred; p++) {
        int found = 0;
        float score = 0.0f;
        for (size_t l = 0; l < n_loc; l++)  /* the last entry for a species wins */
            if (loc_species[l] == pred_species[p]) {
                found = 1;
                score = loc_score[l];
            }
        if (found && !(score >= threshold)) continue;             /* in the meta model, below the threshold: drop */
        keep_pos[m] = (uint32_t)p;
        conf_out[m] = (found && rerank) ? pred_conf[p] * score : pred_conf[p];  /* not in the meta model: unchanged */
        m++;
    }
    if (rerank) sort_desc(keep_pos, conf_out, m);
    return m;
}
