"""GPU parity of the top-K / sigmoid kernel (row a9 of SURVEY.md 8): BIT-EXACT against the
oracle restatement of postprocess.rs:40-93 -- indices, their order, confidence bits and counts,
including ties, NaN and infinities."""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

INF, NAN = float("inf"), float("nan")


def check_rows(bn, logits, top_k, min_conf=None):
    """Both device paths must reproduce the oracle: the fast kernel (with its exact-kernel fallback
    for undecidable rows) and the exact BinaryHeap kernel alone (BN_TOPK_EXACT=1)."""
    import os
    _check_rows(bn, logits, top_k, min_conf)
    os.environ["BN_TOPK_EXACT"] = "1"
    try:
        _check_rows(bn, logits, top_k, min_conf)
    finally:
        del os.environ["BN_TOPK_EXACT"]


def _check_rows(bn, logits, top_k, min_conf=None):
    a = np.ascontiguousarray(logits, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    idx, conf, cnt = bn.topk_host(a, top_k, min_conf)
    for r in range(a.shape[0]):
        want = oracle.top_k(a[r], top_k, min_conf)
        got_i = idx[r, :cnt[r]].tolist()
        got_c = conf[r, :cnt[r]]
        assert cnt[r] == len(want), (r, cnt[r], len(want))
        assert got_i == [w[0] for w in want], (r, got_i, [w[0] for w in want])
        assert got_c.tobytes() == np.asarray([w[1] for w in want], dtype=np.float32).tobytes(), (r, got_c, want)


# ---- the reference's own known-answer inputs (postprocess.rs:101-331) through the kernel ----
@pytest.mark.parametrize("logits,k,min_conf", [
    ([0.1, 0.5, 0.9, 0.3, 0.7], 3, None),          # test_top_k_predictions_basic
    ([-5.0, 0.0, 5.0], 10, 0.4),                    # test_top_k_with_min_confidence
    ([0.1, 0.2], 100, None),                        # test_top_k_larger_than_input
    ([0.1, 0.9, 0.5], 3, None),                     # test_predictions_have_correct_indices
    ([0.5, 0.5, 0.5, 0.5], 2, None),                # test_top_k_all_equal_scores
    ([-10.0, -5.0, -1.0, -20.0], 2, None),          # test_top_k_negative_logits
    ([1.0, NAN, 2.0, 0.5], 3, None),                # test_top_k_with_nan_values
    ([-10.0, 0.0, 10.0], 10, 0.0),                  # test_min_confidence_zero
    ([-10.0, 0.0, 10.0], 10, 1.0),                  # test_min_confidence_one
    ([0.1, 0.2, 0.3], 2 ** 64 - 1, None),           # test_top_k_max_usize
    ([0.1, 0.2, 0.3, 0.4], 4, None),                # test_missing_labels
    ([INF, -INF, 0.0, 100.0, -100.0], 5, None),     # sigmoid edge values
])
def test_reference_kats(bn, logits, k, min_conf):
    check_rows(bn, logits, k, min_conf)


def test_zero_k_and_empty(bn):
    idx, conf, cnt = bn.topk_host(np.zeros((3, 5), np.float32), 0)
    assert cnt.tolist() == [0, 0, 0]                # test_top_k_zero_k


def test_basic_expectations(bn):
    idx, conf, cnt = bn.topk_host([0.1, 0.5, 0.9, 0.3, 0.7], 3)
    assert idx[0, :3].tolist() == [2, 4, 1] and cnt[0] == 3
    idx, conf, cnt = bn.topk_host([-10.0, 0.0, 10.0], 10, 1.0)
    assert cnt[0] == 0


@pytest.mark.parametrize("k", [1, 3, 10, 20, 21, 64, 65, 100, 500])
def test_random_logits_rows(bn, k):
    # testutil.rs:110-121 LCG rows at the real class counts; 16-bit quantised => plenty of exact ties
    rows = np.stack([oracle.random_logits(6522, 12345 + s) for s in range(24)])
    check_rows(bn, rows, k)
    check_rows(bn, rows[:8], k, 0.5)


def test_perch_width_rows(bn):
    rows = np.stack([oracle.random_logits(14795, 7 + s) for s in range(8)])
    check_rows(bn, rows, 10)
    check_rows(bn, rows, 10, 0.9)


def test_continuous_values_and_batch_sizes(bn):
    rng = np.random.default_rng(0)
    for b in (1, 2, 31, 32, 33, 128):
        check_rows(bn, rng.normal(-2.0, 3.0, size=(b, 6522)).astype(np.float32), 10, 0.1)


def test_ties_at_the_boundary_and_everywhere(bn):
    rng = np.random.default_rng(1)
    rows = []
    rows.append(np.zeros(6522, np.float32))                                   # all equal
    rows.append(np.full(6522, -3.5, np.float32))
    rows.append(rng.integers(-3, 4, size=6522).astype(np.float32))             # 7 distinct values
    rows.append(np.repeat(rng.normal(size=1087).astype(np.float32), 6))        # runs of 6 equal values
    r = rng.normal(size=6522).astype(np.float32); r[100:120] = r.max() + 1.0   # 20-way tie at the top
    rows.append(r)
    r = rng.normal(size=6522).astype(np.float32); r[::7] = 25.0               # sigmoid saturates to 1.0
    rows.append(r)
    r = np.sort(rng.normal(size=6522).astype(np.float32)); rows.append(r)       # ascending: every push enters
    rows.append(r[::-1].copy())                                               # descending: nothing enters
    rows.append(np.where(np.arange(6522) % 2 == 0, 0.0, -0.0).astype(np.float32))  # +0 / -0 differ under total_cmp
    for k in (1, 2, 3, 7, 8, 10, 16, 31, 64):
        check_rows(bn, np.stack(rows), k)
        check_rows(bn, np.stack(rows), k, 0.5)


def test_nan_and_infinities(bn):
    rng = np.random.default_rng(2)
    rows = []
    r = rng.normal(size=6522).astype(np.float32); r[[5, 77, 6000]] = NAN; rows.append(r)
    r = rng.normal(size=6522).astype(np.float32); r[[1, 2]] = INF; r[[3, 4]] = -INF; rows.append(r)
    rows.append(np.full(6522, NAN, np.float32))
    r = rng.normal(size=6522).astype(np.float32)
    r[10] = np.frombuffer(np.uint32(0xFFC00001).tobytes(), np.float32)[0]      # negative NaN ranks below -inf
    r[11] = np.frombuffer(np.uint32(0x7FC00123).tobytes(), np.float32)[0]      # payload NaN ranks above +inf
    rows.append(r)
    for k in (1, 3, 10, 20):
        check_rows(bn, np.stack(rows), k)
        check_rows(bn, np.stack(rows), k, 0.0)


def test_sigmoid_is_bit_exact_with_libm(bn):
    # one-element rows: confidence = sigmoid(logit); sweep dense + special regions
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-110, 110, 40000), rng.normal(0, 4, 40000), np.linspace(-20, 20, 8001),
                         [0.0, -0.0, 88.0, -88.0, 88.72, 88.73, -103.9, -104.0, 1e-30, -1e-30, 16.6, 17.0, 89.0]])
    xs = xs.astype(np.float32)[:, None]
    idx, conf, cnt = bn.topk_host(xs, 1)
    want = np.array([oracle.sigmoid(float(v)) for v in xs[:, 0]], dtype=np.float32)
    assert cnt.min() == 1
    mism = np.flatnonzero(conf[:, 0].view(np.uint32) != want.view(np.uint32))
    assert mism.size == 0, (mism[:10], xs[mism[:10], 0], conf[mism[:10], 0], want[mism[:10]])
