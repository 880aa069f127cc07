"""The oracle's restatement of the range filter host logic, pinned by the reference's own known-answer
tests (src/rangefilter.rs:586-935).  Inputs and asserted facts are restated as data."""
import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("month,day,week", [(1, 1, 1.0), (1, 8, 2.0), (2, 1, 5.0), (12, 31, 49.0)])   # rangefilter.rs:586-625
def test_calculate_week_kats(month, day, week):
    assert oracle.calculate_week(month, day) == week


def test_calculate_week_all_days_follow_the_formula():
    for m in range(1, 13):
        for d in range(1, 32):
            assert oracle.calculate_week(m, d) == float((m - 1) * 4 + (d - 1) // 7 + 1)


def test_validate_coordinates_kats():                                   # rangefilter.rs:627-650
    for lat, lon in ((45.0, -122.0), (0.0, 0.0), (-90.0, -180.0), (90.0, 180.0)):
        assert oracle.validate_coordinates(lat, lon) == 0
    assert oracle.validate_coordinates(91.0, 0.0) == 1
    assert oracle.validate_coordinates(0.0, 181.0) == 2
    assert oracle.validate_coordinates(float("nan"), 0.0) == 1           # RangeInclusive::contains(NaN) is false
    assert oracle.validate_coordinates(95.0, 200.0) == 1                 # latitude is checked first


def test_validate_date_kats():                                          # rangefilter.rs:652-690
    for m, d in ((1, 1), (6, 15), (12, 31)):
        assert oracle.validate_date(m, d) == 0
    assert oracle.validate_date(0, 1) == 1 and oracle.validate_date(13, 1) == 1
    assert oracle.validate_date(1, 0) == 2 and oracle.validate_date(1, 32) == 2


A, B, C_, D = 0, 1, 2, 3  # "Species A".."Species D"


def test_filter_predictions_above_threshold():                          # rangefilter.rs:707-761
    pos, conf = oracle.filter_predictions([A, B, C_], [0.8, 0.3, 0.05], [A, B, C_], [0.9, 0.02, 0.5], 0.03, False)
    assert pos.tolist() == [0, 2]                                       # B filtered (0.02 < 0.03); order kept
    assert conf.tolist() == [np.float32(0.8), np.float32(0.05)]


def test_filter_predictions_with_rerank():                              # rangefilter.rs:763-835
    pos, conf = oracle.filter_predictions([A, B, C_], [0.9, 0.8, 0.7], [A, B, C_], [0.5, 0.9, 0.6], 0.03, True)
    assert pos.tolist() == [1, 0, 2]                                    # B (0.72), A (0.45), C (0.42)
    assert np.allclose(conf, [0.72, 0.45, 0.42], atol=1e-3)
    assert conf[0] == np.float32(0.8) * np.float32(0.9)                 # plain f32 product


def test_filter_predictions_species_not_in_meta_model():                # rangefilter.rs:837-889
    pos, conf = oracle.filter_predictions([A, B, D], [0.8, 0.7, 0.9], [A, C_], [0.9, 0.8], 0.03, False)
    assert pos.tolist() == [0, 1, 2]
    assert conf.tolist() == [np.float32(0.8), np.float32(0.7), np.float32(0.9)]


def test_filter_batch_predictions():                                    # rangefilter.rs:904-935
    loc = ([A, B], [0.9, 0.05])
    assert len(oracle.filter_predictions([A], [0.8], *loc, 0.1, False)[0]) == 1
    assert len(oracle.filter_predictions([B], [0.6], *loc, 0.1, False)[0]) == 0


def test_location_scores_threshold_sort_and_label_bound():              # rangefilter.rs:477-496
    scores = np.array([0.5, 0.009, 0.01, 0.9, 0.2, 0.7], dtype=np.float32)
    idx, sc = oracle.location_scores(scores, n_labels=5, threshold=0.01)
    assert idx.tolist() == [3, 0, 4, 2]                                 # 0.009 below threshold; index 5 has no label
    assert sc.tolist() == [np.float32(0.9), np.float32(0.5), np.float32(0.2), np.float32(0.01)]
    idx, _ = oracle.location_scores(np.array([0.3, 0.3, 0.3], dtype=np.float32), 3, 0.01)
    assert idx.tolist() == [0, 1, 2]                                    # equal keys stay in input order (documented choice)
    assert len(oracle.location_scores(np.array([np.nan], dtype=np.float32), 1, 0.01)[0]) == 0   # NaN >= t is false


def test_duplicate_location_entries_last_one_wins():                    # HashMap collect (rangefilter.rs:340-343)
    pos, conf = oracle.filter_predictions([A], [0.5], [A, A], [0.9, 0.001], 0.01, False)
    assert len(pos) == 0
