"""Pins oracle/postprocess.c against the reference's own known-answer tests
(/root/reference/src/postprocess.rs:101-331, one test per reference #[test])."""
import math

import numpy as np

import oracle

INF = float("inf")
NAN = float("nan")


def labels(n, fmt="species_{}"):
    return [fmt.format(i) for i in range(n)]


def species(preds, labs):
    # postprocess.rs:69-72: labels.get(index) or "unknown_{index}"
    return [labs[i] if i < len(labs) else f"unknown_{i}" for i, _ in preds]


def test_sigmoid():  # postprocess.rs:102-107
    assert abs(oracle.sigmoid(0.0) - 0.5) < 1e-4
    assert oracle.sigmoid(10.0) > 0.99
    assert oracle.sigmoid(-10.0) < 0.01


def test_top_k_predictions_basic():  # :109-122
    p = oracle.top_k([0.1, 0.5, 0.9, 0.3, 0.7], 3)
    assert len(p) == 3
    assert p[0][1] >= p[1][1] >= p[2][1]
    assert species(p, labels(5))[0] == "species_2"
    assert [i for i, _ in p] == [2, 4, 1]


def test_top_k_with_min_confidence():  # :124-137
    p = oracle.top_k([-5.0, 0.0, 5.0], 10, 0.4)
    assert len(p) == 2
    assert all(c >= 0.4 for _, c in p)


def test_top_k_larger_than_input():  # :139-147
    assert len(oracle.top_k([0.1, 0.2], 100)) == 2


def test_top_k_empty_input():  # :149-153
    assert oracle.top_k(np.zeros(0, np.float32), 10) == []


def test_top_k_zero_k():  # :155-162
    assert oracle.top_k([0.1, 0.2, 0.3], 0) == []


def test_predictions_have_correct_indices():  # :164-174
    labs = ["zero", "one", "two"]
    p = oracle.top_k([0.1, 0.9, 0.5], 3)
    names = species(p, labs)
    assert p[names.index("one")][0] == 1


def test_sigmoid_infinity():  # :178-187
    eps = np.finfo(np.float32).eps
    assert abs(oracle.sigmoid(INF) - 1.0) < eps
    assert abs(oracle.sigmoid(-INF)) < eps


def test_sigmoid_nan():  # :189-194
    assert math.isnan(oracle.sigmoid(NAN))


def test_sigmoid_large_values():  # :196-205
    assert oracle.sigmoid(100.0) > 0.9999
    assert oracle.sigmoid(-100.0) < 0.0001


def test_top_k_all_equal_scores():  # :207-219
    p = oracle.top_k([0.5, 0.5, 0.5, 0.5], 2)
    assert len(p) == 2
    assert abs(p[0][1] - p[1][1]) < 1e-4


def test_top_k_negative_logits():  # :221-233
    p = oracle.top_k([-10.0, -5.0, -1.0, -20.0], 2)
    assert len(p) == 2
    assert p[0][1] >= p[1][1]
    assert p[0][0] == 2


def test_top_k_with_nan_values():  # :235-245
    p = oracle.top_k([1.0, NAN, 2.0, 0.5], 3)
    assert len(p) > 0
    # total_cmp ranks (positive) NaN above every number: it must be among the survivors
    assert 1 in [i for i, _ in p]


def test_min_confidence_zero():  # :247-256
    assert len(oracle.top_k([-10.0, 0.0, 10.0], 10, 0.0)) == 3


def test_min_confidence_one():  # :258-269
    assert len(oracle.top_k([-10.0, 0.0, 10.0], 10, 1.0)) == 0


def test_top_k_max_usize():  # :271-280
    assert len(oracle.top_k([0.1, 0.2, 0.3], 2**64 - 1)) == 3


def test_missing_labels():  # :282-298
    labs = ["a", "b"]
    p = oracle.top_k([0.1, 0.2, 0.3, 0.4], 4)
    assert len(p) == 4
    assert sum(s.startswith("unknown_") for s in species(p, labs)) == 2


def test_score_entry_ordering():  # :300-317
    assert oracle.score_entry_cmp(1.0, 2.0) > 0   # entry1 > entry2 in min-heap order
    assert oracle.score_entry_cmp(1.0, 1.0) == 0  # equal scores


def test_score_entry_with_nan():  # :319-331
    oracle.score_entry_cmp(1.0, NAN)
    assert oracle.score_entry_cmp(NAN, NAN) == 0


# ---- beyond the reference's KATs: properties that any correct restatement has ----

def test_matches_numpy_on_distinct_values():
    for seed in (1, 42, 12345):
        x = oracle.random_logits(6522, seed)
        x = x + np.arange(6522, dtype=np.float32) * np.float32(1e-6)  # mostly distinct
        uniq, cnt = np.unique(x, return_counts=True)
        p = oracle.top_k(x, 10)
        conf = np.array([c for _, c in p])
        assert np.all(conf[:-1] >= conf[1:])
        kth = np.sort(x)[-10]
        if np.sum(x >= kth) == 10:  # no boundary tie: survivor set is unique
            assert sorted(i for i, _ in p) == sorted(np.argsort(-x, kind="stable")[:10].tolist())


def test_random_logits_lcg():  # testutil.rs:110-121 + its tests :224-243
    a = oracle.random_logits(100, 42)
    b = oracle.random_logits(100, 42)
    assert np.array_equal(a, b)
    assert a.min() >= -5.0 and a.max() <= 5.0
    assert not np.array_equal(a, oracle.random_logits(100, 43))
    # first value by hand: state = 42*1103515245+12345; bits = (state>>16)&0xFFFF
    st = (42 * 1103515245 + 12345) & (2**64 - 1)
    bits = np.float32((st >> 16) & 0xFFFF)
    assert a[0] == np.float32(np.float64(bits) * np.float64(np.float32(10.0 / 65535.0)) - 5.0)
