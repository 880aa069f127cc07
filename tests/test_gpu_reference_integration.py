"""The reference's integration suite (tests/integration_test.rs), test for test, run through the drop-in
`Classifier` / `RangeFilter` API on the MI355X.  Each test keeps the reference test's name and asserts what it
asserts (cited by line); the model files are the synthetic-weight ones with the reference's I/O contract, the
fixtures (silent segment, 0.5-amplitude sine) are restated from integration_test.rs:52-67."""
import threading

import numpy as np
import pytest

from gpu_helpers import synth, write_model

pytestmark = pytest.mark.gpu

MT = {"v24": (144000, 48000), "v30": (160000, 32000), "perch": (160000, 32000)}


def silent_segment(family):                      # integration_test.rs:52-54
    return np.zeros(MT[family][0], dtype=np.float32)


def sine_wave_segment(family, frequency):        # integration_test.rs:57-67
    n, sr = MT[family]
    t = np.arange(n, dtype=np.float32) / np.float32(sr)
    return (np.sin(np.float32(2.0 * np.pi) * np.float32(frequency) * t) * np.float32(0.5)).astype(np.float32)


@pytest.fixture(scope="module")
def fixtures(tmp_path_factory):
    d = tmp_path_factory.mktemp("fixtures")
    out = {}
    out["birdnet_v24.onnx"] = write_model(synth.birdnet_v24(num_species=400, width=0.5, depth=0.5, head=256))
    (d / "birdnet_v24_labels.txt").write_text("\n".join(f"Genus{i} species{i}_Common name {i}" for i in range(400)) + "\n")
    out["birdnet_v30.onnx"] = write_model(synth.birdnet_v30(num_species=300, width=0.35, depth=0.35))
    (d / "birdnet_v30_labels.csv").write_text("label\n" + "\n".join(f"species_{i},x" for i in range(300)) + "\n")
    out["perch_v2.onnx"] = write_model(synth.perch_v2(num_species=500, width=0.35, depth=0.25, emb=1536))
    (d / "perch_v2_labels.csv").write_text("inat2024_fsd50k\n" + "\n".join(f"class_{i}" for i in range(500)) + "\n")
    out["meta.onnx"] = write_model(synth.meta_model(num_species=400, hidden=32))
    out["dir"] = d
    return out


def v24(bn, fx):
    return bn.Classifier.builder().model_path(fx["birdnet_v24.onnx"]).labels_path(str(fx["dir"] / "birdnet_v24_labels.txt"))


def test_birdnet_v24_load(bn, fixtures):                        # :75-93
    config = v24(bn, fixtures).build().config()
    assert config.model_type == bn.ModelType.BirdNetV24
    assert config.sample_rate == 48000 and config.sample_count == 144000
    assert config.embedding_dim is None


def test_birdnet_v24_predict(bn, fixtures):                     # :97-122
    classifier = v24(bn, fixtures).top_k(10).build()
    result = classifier.predict(silent_segment("v24"))
    assert result.model_type == bn.ModelType.BirdNetV24
    assert len(result.predictions) <= 10
    assert result.embeddings is None
    assert len(result.raw_scores) > 0
    for a, b in zip(result.predictions, result.predictions[1:]):
        assert a.confidence >= b.confidence


def test_birdnet_v24_predict_batch(bn, fixtures):               # :126-151
    classifier = v24(bn, fixtures).build()
    segments = [silent_segment("v24"), sine_wave_segment("v24", 440.0), sine_wave_segment("v24", 1000.0)]
    results = classifier.predict_batch(segments)
    assert len(results) == 3
    for result in results:
        assert result.model_type == bn.ModelType.BirdNetV24 and result.embeddings is None


def test_birdnet_v24_wrong_input_size(bn, fixtures):            # :155-173
    classifier = v24(bn, fixtures).build()
    with pytest.raises(bn.Error) as e:
        classifier.predict(np.zeros(100000, dtype=np.float32))
    assert "input size mismatch" in str(e.value)


def test_birdnet_v30_load(bn, fixtures):                        # :181-200
    config = bn.Classifier.builder().model_path(fixtures["birdnet_v30.onnx"]).labels_path(str(fixtures["dir"] / "birdnet_v30_labels.csv")).build().config()
    assert config.model_type == bn.ModelType.BirdNetV30
    assert config.sample_rate == 32000 and config.sample_count == 160000
    assert config.embedding_dim is not None


def test_birdnet_v30_predict_with_embeddings(bn, fixtures):     # :204-225
    classifier = bn.Classifier.builder().model_path(fixtures["birdnet_v30.onnx"]).labels_path(str(fixtures["dir"] / "birdnet_v30_labels.csv")).build()
    result = classifier.predict(silent_segment("v30"))
    assert result.model_type == bn.ModelType.BirdNetV30
    assert result.embeddings is not None and len(result.embeddings) == 1024


def perch(bn, fx):
    return bn.Classifier.builder().model_path(fx["perch_v2.onnx"]).labels_path(str(fx["dir"] / "perch_v2_labels.csv"))


def test_perch_v2_load_with_override(bn, fixtures):             # :233-252
    config = perch(bn, fixtures).model_type(bn.ModelType.PerchV2).build().config()
    assert config.model_type == bn.ModelType.PerchV2 and config.sample_rate == 32000 and config.embedding_dim is not None


def test_perch_v2_predict(bn, fixtures):                        # :256-275
    result = perch(bn, fixtures).model_type(bn.ModelType.PerchV2).build().predict(silent_segment("perch"))
    assert result.model_type == bn.ModelType.PerchV2 and result.embeddings is not None


def test_perch_v2_auto_detection(bn, fixtures):                 # :279-309
    config = perch(bn, fixtures).build().config()
    assert config.model_type == bn.ModelType.PerchV2
    assert config.sample_rate == 32000 and config.segment_duration == 5.0 and config.sample_count == 160000
    assert config.embedding_dim is not None


def test_perch_v2_predict_real_model(bn, fixtures):             # :313-355
    classifier = perch(bn, fixtures).top_k(10).min_confidence(0.1).build()
    result = classifier.predict(silent_segment("perch"))
    assert result.model_type == bn.ModelType.PerchV2
    assert result.embeddings is not None and len(result.predictions) <= 10 and len(result.raw_scores) > 0
    for a, b in zip(result.predictions, result.predictions[1:]):
        assert a.confidence >= b.confidence
    for pred in result.predictions:
        assert pred.confidence >= np.float32(0.1)


def test_perch_v2_batch_predict(bn, fixtures):                  # :359-392
    classifier = perch(bn, fixtures).build()
    results = classifier.predict_batch([silent_segment("perch"), sine_wave_segment("perch", 440.0), sine_wave_segment("perch", 1000.0)])
    assert len(results) == 3
    for result in results:
        assert result.model_type == bn.ModelType.PerchV2 and result.embeddings is not None


def test_in_memory_labels(bn, fixtures):                        # :400-435
    label_count = v24(bn, fixtures).build().config().num_species
    labels = [f"Species_{i}" for i in range(label_count)]
    classifier = bn.Classifier.builder().model_path(fixtures["birdnet_v24.onnx"]).labels(labels).build()
    for pred in classifier.predict(silent_segment("v24")).predictions:
        assert pred.species.startswith("Species_")


def test_top_k_configuration(bn, fixtures):                     # :439-456
    assert len(v24(bn, fixtures).top_k(5).build().predict(silent_segment("v24")).predictions) <= 5


def test_min_confidence_configuration(bn, fixtures):            # :460-484
    for pred in v24(bn, fixtures).min_confidence(0.5).build().predict(silent_segment("v24")).predictions:
        assert pred.confidence >= 0.5


def test_classifier_is_send_sync(bn, fixtures):                 # :488-491 (Send + Sync: usable from any thread)
    classifier = v24(bn, fixtures).build()
    out = []
    th = threading.Thread(target=lambda: out.append(classifier.predict(silent_segment("v24")).model_type))
    th.start()
    th.join()
    assert out == [bn.ModelType.BirdNetV24]


def test_concurrent_predictions(bn, fixtures):                  # :495-529: 4 threads x 10 predictions on one classifier
    classifier = v24(bn, fixtures).build()
    errors = []

    def worker():
        try:
            segment = np.zeros(144000, dtype=np.float32)
            for _ in range(10):
                assert classifier.predict(segment).model_type == bn.ModelType.BirdNetV24
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)
    threads = [threading.Thread(target=worker) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_missing_model_path(bn):                                # :536-546
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().labels_path("labels.txt").build()
    assert "model path required" in str(e.value)


def test_missing_labels(bn):                                    # :549-554
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().model_path("model.onnx").build()
    assert "labels required" in str(e.value)


def test_nonexistent_model_file(bn):                            # :558-565
    with pytest.raises(bn.Error):
        bn.Classifier.builder().model_path("/nonexistent/model.onnx").labels_path("labels.txt").build()


def test_label_count_mismatch(bn, fixtures):                    # :569-591
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().model_path(fixtures["birdnet_v24.onnx"]).labels(["only_one"]).build()
    assert "label count mismatch" in str(e.value)


def range_filter(bn, fx, classifier=None):
    b = bn.RangeFilter.builder().model_path(fx["meta.onnx"])
    if classifier is not None:
        return b.from_classifier_labels(classifier.labels())
    return b.labels((fx["dir"] / "birdnet_v24_labels.txt").read_text().splitlines())


def test_range_filter_with_real_model(bn, fixtures):            # :595-654
    scores = range_filter(bn, fixtures).threshold(0.01).build().predict(60.1695, 24.9354, 6, 15)
    assert scores
    for a, b in zip(scores, scores[1:]):
        assert a.score >= b.score
    for s in scores:
        assert s.score >= np.float32(0.01)


def test_range_filter_invalid_inputs(bn, fixtures):             # :658-705
    rf = range_filter(bn, fixtures).build()
    for args in ((95.0, 0.0, 1, 1), (0.0, 190.0, 1, 1), (0.0, 0.0, 0, 1), (0.0, 0.0, 13, 1), (0.0, 0.0, 1, 0), (0.0, 0.0, 1, 32)):
        with pytest.raises(bn.Error):
            rf.predict(*args)


def test_range_filter_from_classifier_labels(bn, fixtures):     # :709-752
    classifier = v24(bn, fixtures).build()
    assert range_filter(bn, fixtures, classifier).threshold(0.01).build().predict(60.1695, 24.9354, 6, 15)


def test_range_filter_complete_workflow(bn, fixtures):          # :756-835
    classifier = v24(bn, fixtures).build()
    rf = range_filter(bn, fixtures, classifier).threshold(0.01).build()
    labels = classifier.labels()
    predictions = [bn.Prediction(labels[0], 0.8, 0), bn.Prediction(labels[1], 0.6, 1)]
    location_scores = rf.predict(60.1695, 24.9354, 6, 15)
    filtered = rf.filter_predictions(predictions, location_scores, False)
    assert len(filtered) <= len(predictions)
    filtered_batch = rf.filter_batch_predictions([list(predictions), predictions], location_scores, True)
    assert len(filtered_batch) == 2
    # beyond the reference's assertions: survivors are exactly the species at or above the threshold
    score_of = {s.species: s.score for s in location_scores}
    assert [p.species for p in filtered] == [p.species for p in predictions if p.species in score_of]
    for row in filtered_batch:
        for a, b in zip(row, row[1:]):
            assert a.confidence >= b.confidence
