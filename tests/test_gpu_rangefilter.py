"""RangeFilter (reference src/rangefilter.rs:435-502) with the meta model on the MI355X: scores against the
fp32 oracle interpreter, threshold / label bound / descending order against the oracle's restatement."""
import numpy as np
import pytest

import oracle
from oracle import onnx_ref
from gpu_helpers import ATOL, RTOL, synth, write_model

pytestmark = pytest.mark.gpu


def test_range_filter_predict_matches_oracle(bn):
    n = 500
    data = synth.meta_model(num_species=n, hidden=48)
    labels = [f"Genus{i} species{i}_Common name {i}" for i in range(n)]
    rf = bn.RangeFilter.builder().model_path(write_model(data)).from_classifier_labels(labels).threshold(0.03).with_rocm(0).build()
    for lat, lon, month, day in ((60.17, 24.94, 6, 15), (-33.9, 151.2, 12, 31), (0.0, 0.0, 1, 1), (90.0, -180.0, 2, 29)):
        x = np.array([[lat, lon, oracle.calculate_week(month, day)]], dtype=np.float32)
        want = onnx_ref.run_model(data, x)["output"].reshape(-1)
        got = rf.predict(lat, lon, month, day)
        idx, sc = oracle.location_scores(want, n, 0.03)
        # scores within the fp32 tolerance; ordering / membership decided by the product's own scores
        dense = np.zeros(n, dtype=np.float32)
        for s in got:
            dense[s.index] = s.score
            assert s.species == labels[s.index] and s.score >= np.float32(0.03)
        clear = np.abs(want - 0.03) > 1e-3                      # away from the threshold the membership must agree
        assert ((dense > 0) == (want >= 0.03))[clear].all()
        keep = dense > 0
        assert np.all(np.abs(dense[keep] - want[keep]) <= ATOL + RTOL * np.abs(want[keep]))
        assert [s.score for s in got] == sorted([s.score for s in got], reverse=True)
        gi, gs = oracle.location_scores(dense, n, 0.03)          # the oracle's sort of the product's scores
        assert [s.index for s in got] == gi.tolist() and [np.float32(s.score) for s in got] == gs.tolist()
        assert 0 < len(got) < n and abs(len(got) - len(idx)) <= 2
    # end to end with filter_predictions
    scores = rf.predict(60.17, 24.94, 6, 15)
    preds = [bn.Prediction(labels[s.index], 0.5, s.index) for s in scores[:3]] + [bn.Prediction("not in the meta model", 0.4, 9999)]
    out = rf.filter_predictions(preds, scores, rerank=True)
    assert len(out) == 4 and [p.confidence for p in out] == sorted([p.confidence for p in out], reverse=True)
    assert rf.filter_batch_predictions([preds, preds[:1]], scores, False)[1][0].species == preds[0].species


def test_range_filter_build_and_predict_errors(bn):
    data = synth.meta_model(num_species=40, hidden=16)
    path = write_model(data)
    with pytest.raises(bn.Error) as e:                                   # rangefilter.rs:263-269
        bn.RangeFilter.builder().model_path(path).labels(["a"] * 39).build()
    assert e.value.kind == bn.ErrorKind.LabelCount and (e.value.expected, e.value.got) == (40, 39)
    with pytest.raises(bn.Error) as e:                                   # rangefilter.rs:253-259: exactly one output
        bn.RangeFilter.builder().model_path(write_model(synth.birdnet_v30(num_species=50, width=0.25, depth=0.25))).labels(["a"] * 50).build()
    assert e.value.kind == bn.ErrorKind.ModelDetection and "meta model expects 1 output, got 2" in str(e.value)
    with pytest.raises(bn.Error) as e:
        bn.RangeFilter.builder().model_path("/nonexistent/meta.onnx").labels(["a"]).build()
    assert e.value.kind == bn.ErrorKind.ModelLoad
    rf = bn.RangeFilter.builder().model_path(path).labels([f"s{i}" for i in range(40)]).build()
    assert rf.threshold == pytest.approx(0.01)
    with pytest.raises(bn.Error) as e:
        rf.predict(91.0, 0.0, 6, 15)
    assert e.value.kind == bn.ErrorKind.InvalidCoordinates
    with pytest.raises(bn.Error) as e:                                   # coordinates are validated before the date
        rf.predict(0.0, 181.0, 13, 1)
    assert e.value.kind == bn.ErrorKind.InvalidCoordinates
    with pytest.raises(bn.Error) as e:
        rf.predict(0.0, 0.0, 0, 1)
    assert e.value.kind == bn.ErrorKind.InvalidDate
