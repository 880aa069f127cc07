"""GPU parity of the whole hot path (rows a1-a8 of SURVEY.md 8): synthetic-weight models on the
hypothesised topologies through the C ABI / C++ Classifier mirror vs the CPU oracle.

Tolerance (fp32, stated here as north_star asks): |gpu - oracle| <= 2e-4 + 2e-4*|oracle| on every
logit / embedding value, identical top-1 index, and top-K sets equal to the oracle's top-K of the
GPU's own logits (bit-exact post-processing)."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import oracle
from oracle import onnx_ref
from gpu_helpers import ATOL, RTOL, assert_close, write_model

pytestmark = pytest.mark.gpu
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")


def labels(n):
    return [f"Species_{i}" for i in range(n)]  # testutil.rs:70-72 mock_labels


@pytest.fixture(scope="module")
def v24_small():
    data = synth.birdnet_v24(num_species=500, width=0.5, depth=0.5, head=256)
    return data, write_model(data)


@pytest.fixture(scope="module")
def v24_full():
    data = synth.birdnet_v24()
    return data, write_model(data)


@pytest.fixture(scope="module")
def v30_small():
    data = synth.birdnet_v30(num_species=300, width=0.5, depth=0.34, emb=1024)
    return data, write_model(data)


@pytest.fixture(scope="module")
def perch_small():
    data = synth.perch_v2(num_species=700, width=0.35, depth=0.25, emb=192)
    return data, write_model(data)


def check_results(results, ref_logits, ref_emb, top_k, min_conf):
    for i, r in enumerate(results):
        assert_close(r.raw_scores, ref_logits[i], f"logits[{i}]")
        assert int(np.argmax(r.raw_scores)) == int(np.argmax(ref_logits[i])), "top-1 differs from the oracle"
        if ref_emb is not None:
            assert_close(r.embeddings, ref_emb[i], f"embeddings[{i}]")
        else:
            assert r.embeddings is None
        want = oracle.top_k(r.raw_scores, top_k, min_conf)
        assert [(p.index, np.float32(p.confidence).tobytes()) for p in r.predictions] == \
               [(w[0], np.float32(w[1]).tobytes()) for w in want]
        assert [p.species for p in r.predictions] == [f"Species_{w[0]}" for w in want]
        # structural assertions of the reference's integration tests (tests/integration_test.rs:111-119)
        assert len(r.predictions) <= top_k
        assert all(a.confidence >= b.confidence for a, b in zip(r.predictions, r.predictions[1:]))
        if min_conf is not None:
            assert all(p.confidence >= min_conf for p in r.predictions)


def test_v24_small_predict_and_batch(bn, v24_small):
    data, path = v24_small
    clf = bn.Classifier.builder().model_path(path).labels(labels(500)).top_k(3).min_confidence(0.1).with_rocm().build()
    cfg = clf.config()
    assert (cfg.model_type, cfg.sample_rate, cfg.sample_count, cfg.num_species, cfg.embedding_dim) == \
           (bn.ModelType.BirdNetV24, 48000, 144000, 500, None)
    assert clf.requested_provider() == "ROCm" and len(clf.labels()) == 500
    x = synth.synthetic_segments(5, 144000, 48000)
    x[3] = 0.0  # silent segment (tests/integration_test.rs:52-54)
    ref = onnx_ref.run_model(data, x)["output"]
    check_results([clf.predict(x[0])], ref[:1], None, 3, 0.1)
    check_results(clf.predict_batch(list(x)), ref, None, 3, 0.1)
    ctx = clf.create_batch_context(8)
    assert (ctx.max_batch_size(), ctx.sample_count(), ctx.input_buffer_capacity(), ctx.input_buffer_bytes(), ctx.model_type()) == \
           (8, 144000, 8 * 144000, 8 * 144000 * 4, bn.ModelType.BirdNetV24)
    res1 = clf.predict_batch_with_context(ctx, list(x))
    check_results(res1, ref, None, 3, 0.1)
    res2 = clf.predict_batch_with_context(ctx, list(x[:2]))  # context reuse with a smaller batch
    check_results(res2, ref[:2], None, 3, 0.1)
    assert clf.predict_batch([]) == [] and clf.predict_batch_with_context(ctx, []) == []


def test_v24_batch_composition_does_not_change_results(bn, v24_small):
    data, path = v24_small
    m = bn.Model(path)
    x = synth.synthetic_segments(9, 144000, 48000)
    a, _ = bn.Context(m, 16).infer(x)
    b = np.concatenate([bn.Context(m, 4).infer(x[i:i + 3])[0] for i in range(0, 9, 3)])
    c, _ = bn.Context(m, 9, bn.BN_CTX_NO_GRAPH).infer(x)
    assert a.tobytes() == b.tobytes() == c.tobytes()   # deterministic: same kernels, same per-sample order


def test_sharing_modes_change_grids_not_bits(bn, v24_full):
    """bn_set_sharing_mode: the forms for a device of the launch's own, for a shared device and by the count of live contexts give the same
    bytes (full-size v2.4: small-map MBConv blocks with more chunks, 64-row GEMM tiles, the LDS-DMA GEMMs' larger tile), each form is
    captured as its own graph, and the default is the form for a device of its own."""
    data, path = v24_full
    m = bn.Model(path)
    x = synth.synthetic_segments(24, 144000, 48000)
    ctx = bn.Context(m, 24)
    base = ctx.infer(x)[0].copy()
    g0 = ctx.stats()["instantiates"]
    try:
        bn.set_sharing_mode(bn.SHARING_SHARED)
        shared = ctx.infer(x)[0].copy()
        assert ctx.stats()["instantiates"] == g0 + 1          # a graph of its own for the other form
        assert ctx.infer(x)[0].tobytes() == shared.tobytes() and ctx.stats()["instantiates"] == g0 + 1
        bn.set_sharing_mode(bn.SHARING_AUTO)
        other = bn.Context(m, 24)                           # a second live context: AUTO = the shared form
        auto2 = ctx.infer(x)[0].copy()
        assert ctx.stats()["instantiates"] == g0 + 1
        other.close()
        auto1 = ctx.infer(x)[0].copy()                      # alone again: the first graph
        assert ctx.stats()["instantiates"] == g0 + 1
    finally:
        bn.set_sharing_mode(bn.SHARING_ALONE)
    assert base.tobytes() == shared.tobytes() == auto2.tobytes() == auto1.tobytes()
    ref = onnx_ref.run_model(data, x[:3])["output"]
    assert np.abs(base[:3] - ref).max() <= 2e-4 + 2e-4 * np.abs(ref).max()


def test_v24_full_size_model(bn, v24_full):
    data, path = v24_full
    clf = bn.Classifier.builder().model_path(path).labels(labels(6522)).with_rocm().build()
    assert clf.config().num_species == 6522
    x = synth.synthetic_segments(4, 144000, 48000)
    ref = onnx_ref.run_model(data, x)["output"]
    check_results(clf.predict_batch(list(x)), ref, None, 10, None)


def test_fft_front_end_whole_models(bn, v24_full, v30_small, monkeypatch):
    """BN_STFT=1: the windowed-DFT banks run as real FFTs (v2.4: both branches with the mel filter banks, compression
    chains and the min-max normalisation absorbed; v3.0: cos and sin blocks from ONE transform) -- same tolerance
    against the oracle's plain convolutions, same top-1, and top-K rows identical to the default plan's."""
    monkeypatch.setenv("BN_STFT", "1")
    data, path = v24_full
    assert bn.plan_describe(path).count(" FFT ") == 2
    clf = bn.Classifier.builder().model_path(path).labels(labels(6522)).with_rocm().build()
    x = synth.synthetic_segments(5, 144000, 48000)
    ref = onnx_ref.run_model(data, x)["output"]
    res = clf.predict_batch(list(x))
    check_results(res, ref, None, 10, None)
    data3, path3 = v30_small
    assert bn.plan_describe(path3).count(" FFT ") == 1 and "~" not in bn.plan_describe(path3)
    clf3 = bn.Classifier.builder().model_path(path3).labels(labels(300)).top_k(5).with_rocm().build()
    x3 = synth.synthetic_segments(3, 160000, 32000)
    out3 = onnx_ref.run_model(data3, x3)
    check_results(clf3.predict_batch(list(x3)), out3["output_1"], out3["output_0"], 5, None)
    # the all-matrix plan, the default (round 4: the 127-bin bank quarter-folded, the 309-bin bank merged with its mel product into 96
    # folded filters -- no FFT left in v2.4's plan) and the per-bank choice of round 3 (the 309-bin bank as an FFT, the other folded)
    for env, nfft in (({"BN_STFT": "0"}, 0), ({}, 0), ({"BN_CONVMERGE": "0"}, 1)):
        monkeypatch.delenv("BN_STFT", raising=False)
        monkeypatch.delenv("BN_CONVMERGE", raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        assert bn.plan_describe(path).count(" FFT ") == nfft
        other = bn.Classifier.builder().model_path(path).labels(labels(6522)).with_rocm().build().predict_batch(list(x))
        check_results(other, ref, None, 10, None)
        for a, b in zip(res, other):
            assert [p.index for p in a.predictions] == [p.index for p in b.predictions]


def test_v30_power_spectrum_folded_into_the_fft_launch(bn, v30_small, monkeypatch):
    """Round 3: behind v3.0's cos | sin bank the planner folds  sqrt(re^2 + im^2)  into the FFT launch (the mel bank stays a
    dense GEMM by default: with 513 bins its spectrum rows only fit the LDS in tiles of 8 frames, which is slower;
    BN_STFT_MEL=force takes it in as well -- the whole front end in ONE launch).  With the rule off (BN_STFT_POWER=0) the
    magnitude is its own elementwise launch again.  All three within the oracle's tolerance, same top-K order."""
    data, path = v30_small
    x = synth.synthetic_segments(5, 160000, 32000)
    out = onnx_ref.run_model(data, x)

    def run():
        return bn.Classifier.builder().model_path(path).labels(labels(300)).top_k(5).with_rocm().build().predict_batch(list(x))

    first = [l for l in bn.plan_describe(path).splitlines() if " FFT " in l]
    assert len(first) == 1 and "power=2" in first[0] and "mel=0" in first[0] and "Sqrt" in first[0], first
    res = run()
    check_results(res, out["output_1"], out["output_0"], 5, None)
    monkeypatch.setenv("BN_STFT_MEL", "force")
    one = [l for l in bn.plan_describe(path).splitlines() if " FFT " in l]
    assert "power=2" in one[0] and "tpb=8" in one[0] and "MatMul" in one[0] and "mel=0" not in one[0], one
    res1 = run()
    check_results(res1, out["output_1"], out["output_0"], 5, None)
    monkeypatch.delenv("BN_STFT_MEL")
    monkeypatch.setenv("BN_STFT_POWER", "0")
    desc0 = bn.plan_describe(path)
    assert "power=0" in desc0 and any(" ELT " in l and "Sqrt" in l for l in desc0.splitlines())
    res0 = run()
    check_results(res0, out["output_1"], out["output_0"], 5, None)
    for a, b, c in zip(res, res0, res1):
        assert [p.index for p in a.predictions] == [p.index for p in b.predictions] == [p.index for p in c.predictions]


def test_v30_embeddings_and_logits(bn, v30_small):
    data, path = v30_small
    clf = bn.Classifier.builder().model_path(path).labels(labels(300)).top_k(5).with_rocm().build()
    cfg = clf.config()
    assert (cfg.model_type, cfg.sample_rate, cfg.sample_count, cfg.num_species, cfg.embedding_dim) == \
           (bn.ModelType.BirdNetV30, 32000, 160000, 300, 1024)
    x = synth.synthetic_segments(3, 160000, 32000)
    out = onnx_ref.run_model(data, x)
    res = clf.predict_batch(list(x))
    check_results(res, out["output_1"], out["output_0"], 5, None)
    assert all(len(r.embeddings) == 1024 for r in res)   # tests/integration_test.rs:221-222
    ctx = clf.create_batch_context(4)
    check_results(clf.predict_batch_with_context(ctx, list(x)), out["output_1"], out["output_0"], 5, None)


def test_perch_outputs_and_context_refusal(bn, perch_small):
    data, path = perch_small
    clf = bn.Classifier.builder().model_path(path).labels(labels(700)).top_k(4).with_rocm().build()
    cfg = clf.config()
    assert (cfg.model_type, cfg.num_species, cfg.embedding_dim) == (bn.ModelType.PerchV2, 700, 192)
    x = synth.synthetic_segments(2, 160000, 32000)
    out = onnx_ref.run_model(data, x)
    check_results(clf.predict_batch(list(x)), out["label"], out["embedding"], 4, None)
    with pytest.raises(bn.Error) as e:      # batch_context.rs:107-114
        clf.create_batch_context(4)
    assert e.value.kind == bn.ErrorKind.Inference
    assert "BatchInferenceContext does not yet support PerchV2 models" in str(e.value)
    # the two outputs the reference computes and discards, on request
    m = bn.Model(path)
    ctx = bn.Context(m, 2, bn.BN_CTX_ALL_OUTPUTS)
    ctx.infer(x)
    assert_close(ctx.read_output(2, 2).reshape(out["spectrogram"].shape), out["spectrogram"], "spectrogram")
    assert_close(ctx.read_output(1, 2).reshape(out["spatial_embedding"].shape), out["spatial_embedding"], "spatial")
    with pytest.raises(bn.EngineError):
        bn.Context(m, 2).read_output(1, 1)   # not computed by a default context


def test_perch_in_the_native_batch_context(bn, perch_small):
    """the native context serves Perch too (same results as predict_batch, bit for bit) and can
    hand out the spectrogram / spatial embedding the reference discards."""
    data, path = perch_small
    clf = bn.Classifier.builder().model_path(path).labels(labels(700)).top_k(4).with_rocm().build()
    x = synth.synthetic_segments(3, 160000, 32000)
    out = onnx_ref.run_model(data, x)
    ctx = clf.create_native_batch_context(4, all_outputs=True)
    assert (ctx.model_type(), ctx.max_batch_size(), ctx.sample_count()) == (bn.ModelType.PerchV2, 4, 160000)
    got = clf.predict_batch_with_context(ctx, list(x))
    check_results(got, out["label"], out["embedding"], 4, None)
    ref = clf.predict_batch(list(x))
    for g, r in zip(got, ref):
        assert np.array_equal(np.asarray(g.raw_scores, np.float32).view(np.uint32), np.asarray(r.raw_scores, np.float32).view(np.uint32))
        assert [(p.index, np.float32(p.confidence)) for p in g.predictions] == [(p.index, np.float32(p.confidence)) for p in r.predictions]
    assert_close(ctx.read_output(2, 3).reshape(out["spectrogram"].shape), out["spectrogram"], "spectrogram")
    assert_close(ctx.read_output(1, 3).reshape(out["spatial_embedding"].shape), out["spatial_embedding"], "spatial")
    # the usual context checks still apply (batch_context.rs:188-211)
    with pytest.raises(bn.Error) as e:
        clf.predict_batch_with_context(ctx, list(synth.synthetic_segments(5, 160000, 32000)))
    assert "batch size 5 exceeds context max 4" in str(e.value)
    plain = clf.create_native_batch_context(2)
    with pytest.raises(bn.Error):
        plain.read_output(1, 1)  # not computed without all_outputs


def test_model_type_override(bn, perch_small, v24_small):
    _, ppath = perch_small
    clf = bn.Classifier.builder().model_path(ppath).labels(labels(700)).model_type(bn.ModelType.PerchV2).build()
    assert clf.config().model_type == bn.ModelType.PerchV2
    with pytest.raises(bn.Error) as e:      # detection.rs:88-97
        bn.Classifier.builder().model_path(ppath).labels(labels(700)).model_type(bn.ModelType.BirdNetV24).build()
    assert e.value.kind == bn.ErrorKind.ModelDetection and "expects 144000 samples" in str(e.value)


def test_error_behaviour(bn, v24_small, tmp_path):
    data, path = v24_small
    with pytest.raises(bn.Error) as e:      # classifier.rs:366-371
        bn.Classifier.builder().model_path(path).labels(labels(499)).build()
    assert e.value.kind == bn.ErrorKind.LabelCount and (e.value.expected, e.value.got) == (500, 499)
    assert str(e.value) == "label count mismatch: model expects 500, got 499"
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().model_path(str(tmp_path / "nope.onnx")).labels(labels(1)).build()
    assert e.value.kind == bn.ErrorKind.ModelLoad and str(e.value).startswith("failed to load model: ")
    lab = tmp_path / "labels.txt"
    lab.write_text("\n".join(labels(500)) + "\n\n")
    clf = bn.Classifier.builder().model_path(path).labels_path(str(lab)).build()
    assert clf.labels()[499] == "Species_499"
    with pytest.raises(bn.Error) as e:      # classifier.rs:612-618, error.rs Display
        clf.predict(np.zeros(1000, np.float32))
    assert e.value.kind == bn.ErrorKind.InputSize and (e.value.expected, e.value.got) == (144000, 1000)
    assert str(e.value) == "input size mismatch: expected 144000 samples, got 1000"
    good = np.zeros(144000, np.float32)
    with pytest.raises(bn.Error) as e:      # classifier.rs:688-696
        clf.predict_batch([good, np.zeros(5, np.float32), good])
    assert e.value.kind == bn.ErrorKind.BatchInputSize and (e.value.index, e.value.expected, e.value.got) == (1, 144000, 5)
    assert str(e.value) == "batch input size mismatch: segment 1 has 5 samples, expected 144000"
    ctx = clf.create_batch_context(2)
    with pytest.raises(bn.Error) as e:      # batch_context.rs:191-196
        clf.predict_batch_with_context(ctx, [good, good, good])
    assert e.value.kind == bn.ErrorKind.Inference and "batch size 3 exceeds context max 2" in str(e.value)
    with pytest.raises(bn.Error) as e:      # batch_context.rs:200-206
        clf.predict_batch_with_context(ctx, [good, np.zeros(7, np.float32)])
    assert e.value.kind == bn.ErrorKind.BatchInputSize and e.value.index == 1


def test_timeout_and_cancellation(bn, v24_full):
    data, path = v24_full
    clf = bn.Classifier.builder().model_path(path).labels(labels(6522)).build()
    x = synth.synthetic_segments(64, 144000, 48000)
    segs = list(x)
    clf.predict_batch(segs[:2])                                   # warm-up (graph capture)
    tok = bn.CancellationToken()
    tok.cancel()
    with pytest.raises(bn.Error) as e:                           # classifier.rs:533-540
        clf.predict_batch(segs, bn.InferenceOptions().with_cancellation_token(tok))
    assert e.value.kind == bn.ErrorKind.Cancelled and str(e.value) == "inference was cancelled"
    with pytest.raises(bn.Error) as e:                           # classifier.rs:542-549
        clf.predict_batch(segs, bn.InferenceOptions.with_timeout_of(1e-6))
    assert e.value.kind == bn.ErrorKind.Timeout and e.value.duration_ns == 1000
    assert str(e.value) == "inference timed out after 1µs"
    # a generous timeout / an un-cancelled token do not disturb the result, and the context recovers
    ref = clf.predict_batch(segs[:3])
    got = clf.predict_batch(segs[:3], bn.InferenceOptions.with_timeout_of(60.0).with_cancellation_token(bn.CancellationToken()))
    assert all(a.raw_scores.tobytes() == b.raw_scores.tobytes() for a, b in zip(ref, got))


def test_concurrent_predict_from_threads(bn, v24_small):
    # tests/integration_test.rs:495-529: 4 threads x 10 predicts on a shared classifier
    import threading
    data, path = v24_small
    clf = bn.Classifier.builder().model_path(path).labels(labels(500)).build()
    x = synth.synthetic_segments(4, 144000, 48000)
    want = [clf.predict(x[i]).raw_scores for i in range(4)]
    errs = []

    def work(i):
        try:
            for _ in range(10):
                assert clf.predict(x[i]).raw_scores.tobytes() == want[i].tobytes()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_size_independent_properties_at_bench_size(bn, v24_full):
    """BASELINE.json configs[1] size (batch 32): no oracle run needed -- permutation equivariance and
    duplicate-row equality are properties any correct per-segment path has."""
    data, path = v24_full
    m = bn.Model(path)
    ctx = bn.Context(m, 32)
    x = synth.synthetic_segments(32, 144000, 48000)
    x[7] = x[19]
    a, _ = ctx.infer(x)
    perm = np.random.default_rng(0).permutation(32)
    b, _ = ctx.infer(x[perm])
    assert a[perm].tobytes() == b.tobytes()
    assert a[7].tobytes() == a[19].tobytes()
    idx, conf, cnt = ctx.topk(32, 10, 0.01)
    for r in range(32):
        want = oracle.top_k(b[r], 10, 0.01)
        assert idx[r, :cnt[r]].tolist() == [w[0] for w in want]


@pytest.mark.parametrize("family,batch", [("v30", 64), ("perch", 128)])
def test_size_independent_properties_other_configs(bn, family, batch):
    """BASELINE.json configs[2] (BirdNET v3.0, batch 64, embeddings) and configs[3] (Perch v2, batch 128) at full
    model size: permutation equivariance and duplicate-row equality of logits AND embeddings, bit for bit; device
    top-K equals the oracle's top_k on the device logits."""
    data = synth.birdnet_v30() if family == "v30" else synth.perch_v2()
    m = bn.Model(write_model(data))
    assert m.config.sample_count == 160000 and m.config.has_embedding
    ctx = bn.Context(m, batch)
    x = synth.synthetic_segments(batch, 160000, 32000)
    x[5] = x[batch - 3]
    la, ea = ctx.infer(x)
    la, ea = la.copy(), ea.copy()
    perm = np.random.default_rng(1).permutation(batch)
    lb, eb = ctx.infer(x[perm])
    assert np.isfinite(la).all() and np.isfinite(ea).all()
    assert la[perm].tobytes() == lb.tobytes() and ea[perm].tobytes() == eb.tobytes()
    assert la[5].tobytes() == la[batch - 3].tobytes() and ea[5].tobytes() == ea[batch - 3].tobytes()
    assert ea.shape[1] == (1024 if family == "v30" else 1536)
    idx, conf, cnt = ctx.topk(batch, 5, None)
    for r in range(0, batch, 7):
        want = oracle.top_k(lb[r], 5)
        assert idx[r, :cnt[r]].tolist() == [w[0] for w in want]
        assert [np.float32(c).tobytes() for c in conf[r, :cnt[r]]] == [np.float32(w[1]).tobytes() for w in want]


def test_sharded_recording_equals_single_pass(bn, v24_small):
    """BASELINE.json configs[4] at reduced length: a continuous recording cut by chunk_audio rules,
    processed as R contiguous shards with a different batch size per shard, concatenated, must be
    bit-identical to one pass over all windows (what the RCCL all-gather reassembles)."""
    import torch
    dmod = importlib.import_module("rust-birdnet-onnx_amd.distributed")
    data, path = v24_small
    m = bn.Model(path)
    S, sr = 144000, 48000
    rec = synth.synthetic_segments(1, S * 9 + 5000, sr)[0]           # 9.03 windows -> 10 windows, last padded
    starts = dmod.chunk_starts(rec.shape[0], S, 0.0, sr)
    so, _ = oracle.chunk_plan(rec.shape[0], S, 0.0, sr)
    assert starts.tolist() == so.tolist() and len(starts) == 10
    ctx_all = bn.Context(m, 16)
    want, _ = ctx_all.infer(dmod.fill_windows(rec, starts, S))
    for world, batches in ((2, (3, 5)), (4, (1, 2, 3, 2))):
        parts = []
        for r in range(world):
            lo, hi = dmod.shard_range(len(starts), r, world)
            ctx = bn.Context(m, batches[r])
            rows = [ctx.infer(dmod.fill_windows(rec, starts[s:min(hi, s + batches[r])], S))[0] for s in range(lo, hi, batches[r])]
            parts.append(torch.from_numpy(np.concatenate(rows)) if rows else torch.zeros((0, want.shape[1])))
        got = torch.cat(parts).numpy()
        assert got.tobytes() == want.tobytes()
    # overlap > 0 re-uses samples between windows; still window-independent
    starts = dmod.chunk_starts(rec.shape[0], S, 1.5, sr)
    w = dmod.fill_windows(rec, starts, S)
    a, _ = ctx_all.infer(w[:16])
    b, _ = bn.Context(m, 4).infer(w[4:8])
    assert a[4:8].tobytes() == b.tobytes()


def test_group_api_sharded_recording_through_the_c_abi(bn, v24_small):
    """bn_group_*: the recording analysis of BASELINE.json configs[4] behind the C ABI.  One rank, and two / three ranks
    sharing device 0 (the one-GPU box: the gather then runs as device copies instead of ncclAllGather -- same slabs, same
    layout), with overlap so that shards share samples, must reproduce the single-context pass bit for bit: logits, top-K
    indices, confidence bits and counts; the ragged last shard and an int16 recording included."""
    data, path = v24_small
    S, sr = 144000, 48000
    pcm = (np.clip(synth.synthetic_segments(1, S * 6 + 7777, sr)[0], -1, 1) * 32767).astype(np.int16)
    step = S - int(1.0 * sr)                                         # 1 s overlap: windows share samples across shards
    m0 = bn.Model(path)
    ctx = bn.Context(m0, 4)
    rec = bn.Recording(pcm)
    G = rec.n_windows(step)
    assert G == bn.lib.bn_chunk_count(pcm.shape[0], step) and G >= 9
    want_l, want_i, want_c, want_n = [], [], [], []
    for f in range(0, G, 4):
        n = min(4, G - f)
        ctx.step_windows(rec, step, f, n, 5, 0.02)
        ctx.synchronize()
        lg, ix, cf, ct = ctx.step_results(n)
        want_l.append(lg); want_i.append(ix); want_c.append(cf); want_n.append(ct)
    want_l, want_i, want_c, want_n = np.concatenate(want_l), np.concatenate(want_i), np.concatenate(want_c), np.concatenate(want_n)
    st = ctx.stats()
    assert st["capture_fallbacks"] == 0 and st["eager_runs"] == 0 and st["replays"] == (G + 3) // 4, st
    for world, batch, nctx in ((1, 4, 2), (2, 3, 2), (3, 2, 1)):
        models = [bn.Model(path) for _ in range(world)]
        grp = bn.Group(models, max_batch=batch, contexts_per_device=nctx)
        assert grp.size() == world and not grp.uses_rccl()          # ranks share device 0
        lg, ix, cf, ct = grp.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=True)
        assert lg.tobytes() == want_l.tobytes(), world
        assert np.array_equal(ct, want_n)
        for r in range(G):
            assert np.array_equal(ix[r, :ct[r]], want_i[r, :ct[r]]) and cf[r, :ct[r]].tobytes() == want_c[r, :ct[r]].tobytes()
        # top-K rows only (80 B instead of 26 KB per window cross the collective), and a second call on the same group
        # (an explicit opt-in: the default carries the reference's raw_scores)
        none, ix2, cf2, ct2 = grp.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=False)
        assert none is None and np.array_equal(ct2, ct) and ix2.tobytes() == ix.tobytes() and cf2.tobytes() == cf.tobytes()
        gst = grp.stats()  # every step of every rank replayed a captured graph: no capture was lost, nothing ran eagerly
        assert gst["capture_fallbacks"] == 0 and gst["eager_runs"] == 0 and gst["replays"] > 0, gst
        assert gst["input_copies"] == 0, gst  # summed like the other counters; windows are cut straight into each context's own buffer
    # chunk_audio semantics at the boundary: a saturated step yields no windows; shard ranges tile [0, G)
    grp = bn.Group([bn.Model(path)], max_batch=2, contexts_per_device=1)
    lg, ix, cf, ct = grp.analyze_recording(pcm, 0, top_k=3, want_logits=True)
    assert lg.shape[0] == 0 and ct.shape[0] == 0
    lo, hi = C.c_size_t(), C.c_size_t()
    cover = []
    for r in range(8):
        bn.lib.bn_shard_range(28800, r, 8, C.byref(lo), C.byref(hi))
        cover.append((lo.value, hi.value))
    assert cover[0] == (0, 3600) and cover[-1] == (25200, 28800) and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    dmod = importlib.import_module("rust-birdnet-onnx_amd.distributed")
    assert [dmod.shard_range(13, r, 4) for r in range(4)] == [tuple(int(v) for v in _range(bn, 13, r, 4)) for r in range(4)]
    # a model of another device in the group is refused
    with pytest.raises(RuntimeError):
        h = C.c_void_p()
        hs, devs = (C.c_void_p * 1)(m0._h), (C.c_int32 * 1)(1)
        if bn.lib.bn_group_create(hs, devs, 1, 2, 1, C.byref(h)):
            raise RuntimeError(bn.group_last_error())


def _range(bn, G, r, world):
    lo, hi = C.c_size_t(), C.c_size_t()
    bn.lib.bn_shard_range(G, r, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def test_round3_kernels_tile_shape_and_grouping_do_not_enter_the_arithmetic(bn, v24_full, monkeypatch):
    """The LDS-DMA GEMM picks its block tile by the size of the launch and the whole-map MBConv kernel its channel
    grouping by the batch: neither may change a single bit (a shard's short last batch must equal the single pass).
    Forced extremes of both knobs, batch 5, full-size v2.4: logits bit-identical to the default plan's."""
    data, path = v24_full
    x = synth.synthetic_segments(5, 144000, 48000)
    base, _ = bn.Context(bn.Model(path), 5).infer(x)
    base = base.copy()
    # (BN_SEFC_G: samples per block of the squeeze-excite launch -- 1 everywhere, 4 everywhere incl. the ragged last group)
    for env in ({"BN_GEMMDMA_MINBLOCKS": "1"}, {"BN_GEMMDMA_MINBLOCKS": "100000000"}, {"BN_MBMAP2_NCH": "1"}, {"BN_MBMAP2_NCH": "7"},
                {"BN_SEFC_G": "1"}, {"BN_SEFC_G": "4"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got, _ = bn.Context(bn.Model(path), 5).infer(x)
        for k in env:
            monkeypatch.delenv(k)
        assert got.tobytes() == base.tobytes(), env
    # ... and batch 1 / 32 of the same segments (different tile shapes by the default rule) repeat the bits
    one = np.concatenate([bn.Context(bn.Model(path), 1).infer(x[i:i + 1])[0] for i in range(5)])
    assert one.tobytes() == base.tobytes()
    big = synth.synthetic_segments(32, 144000, 48000)
    big[:5] = x
    many, _ = bn.Context(bn.Model(path), 32).infer(big)
    assert many[:5].tobytes() == base.tobytes()


@pytest.mark.parametrize("env", [{"BN_GEMMDMA": "0"}, {"BN_GEMMDMA": "2"}, {"BN_MBMAP2": "0"}, {"BN_SEGEMM": "1"}, {"BN_GEMMDMA_KS": "1"},
                                 {"BN_STFT_MELMFMA": "0", "BN_CONVMERGE": "0"}, {"BN_STFT_NW": "16", "BN_CONVMERGE": "0"}, {"BN_MBROW_TOH": "8"}, {"BN_FRAMEPAIR": "1"},
                                 {"BN_CONVFOLD2": "0"}, {"BN_CONVMERGE": "0"}, {"BN_FRAME_PRE": "0"}, {"BN_FRAME_KS": "1"}, {"BN_GEMMGAP": "0"}, {"BN_FRAME2_WPK": "0"}, {"BN_FRAMEH": "0"},
                                 {"BN_GEMM3": "0"}, {"BN_GEMM3": "1"}, {"BN_FRAME2_B3": "0"}, {"BN_MBROW_B3": "0"}, {"BN_MBMAP_B3": "0"}, {"BN_MBMAP_WS": "0"}, {"BN_MBMAP_WS_SMALL": "0"}])
def test_round3_kernels_switched_off_and_on_against_the_oracle(bn, v24_full, monkeypatch, env):
    """Every round-3 rewrite has an off switch (and two opt-ins): the older kernels (BN_GEMMDMA=0, BN_MBMAP2=0), the
    LDS-DMA GEMM on every eligible shape (BN_GEMMDMA=2), the squeeze-excite products in the GEMM prologue (BN_SEGEMM=1),
    one K slice per block (BN_GEMMDMA_KS=1), the mel bank as a sparse walk on the vector ALU instead of 16 x 16 tiles on the
    matrix cores (BN_STFT_MELMFMA=0), the FFT kernel as 16 waves x 512 slots (BN_STFT_NW=16), the round-2 band height of the
    row-streaming MBConv (BN_MBROW_TOH=8) -- each within the network tolerance of the oracle, same top-1."""
    data, path = v24_full
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    desc = bn.plan_describe(path)
    if env.get("BN_GEMMDMA") == "0":
        assert "kernel=dma" not in desc
    if env.get("BN_MBMAP2") == "0":
        assert "tiles=1x1" not in desc
    if env.get("BN_SEGEMM") == "1":
        assert "se_inline=" in desc
    if env.get("BN_FRAMEPAIR") == "1":  # the mel product of the matrix-path branch behind the folded DFT conv, one launch
        assert " pair=" in desc and desc.count("MatMul:MatMul_11") == 1
    assert ("~quarter" in desc) == (not env.get("BN_FRAMEPAIR") and env.get("BN_CONVFOLD2") != "0"), desc  # (round 4) the 127-bin cosine bank
    if env.get("BN_CONVFOLD2") == "0":
        assert "~sym" in desc
    # (round 4, rule K) the head conv writes the pooled row: the LDS-DMA GEMM's 48-row tile holds a sample's whole map
    assert ("gap=1" in desc) == (env.get("BN_GEMMGAP") != "0" and env.get("BN_GEMMDMA") != "0"), desc
    # (round 4) the min-max normalisation rides in the framing launches' span load unless BN_FRAME_PRE=0 (or the opt-in pair rule) keeps it a launch
    assert (" ELT " in "".join(l for l in desc.splitlines() if "Sub:Sub_2" in l)) == (env.get("BN_FRAME_PRE") == "0" or env.get("BN_FRAMEPAIR") == "1"), desc
    # (round 4) the 309-bin bank and its mel product are one bank of 96 folded filters unless BN_CONVMERGE=0 keeps them apart (then: an FFT)
    assert (" FFT " in desc) == (env.get("BN_CONVMERGE") == "0"), desc
    if env.get("BN_STFT_MELMFMA") == "0":
        assert "(csr)" in desc and "(mfma)" not in desc
    elif env.get("BN_CONVMERGE") == "0":
        assert "(mfma)" in desc
    x = synth.synthetic_segments(3, 144000, 48000)
    got, _ = bn.Context(bn.Model(path), 3).infer(x)
    ref = onnx_ref.run_model(data, x)["output"]
    assert_close(got, ref, str(env))
    assert np.array_equal(got.argmax(1), ref.argmax(1))


# ---- BASELINE configs[2] / configs[3] at their OWN model size against the oracle ----------------------------
# The reduced-width fixtures above keep the map sizes but not the channel counts, so kernel instances only the full-size models
# reach (the 5 x 5 SiLU row-kernel instances at one wave per SIMD, Perch's 40-channel stem, its K = 24 project conv, the
# column-streaming blocks at 125 x 32, v3.0's 8 x 32 / 4 x 16 maps) met the oracle nowhere; permutation / duplicate-row
# properties cannot see a deterministic wrong answer.  Each switch set names the kernel family a disagreement would belong to.
FULL_SIZE_ENVS = [{}, {"BN_MBMAP_WS_BANDS": "0"}, {"BN_MBMAP_WS": "0", "BN_GEMMB3_NTW": "1"}, {"BN_MBROW": "0", "BN_GEMMDMA": "0"}, {"BN_MBMAP2": "0"}, {"BN_MBROW_TR": "0"}, {"BN_STFT": "0"}, {"BN_GEMMDMA": "2"}, {"BN_MBMAP3": "1"},
                  {"BN_GEMM3": "0"}, {"BN_GEMM3": "1"}, {"BN_STFT_PAD": "0", "BN_MBROW_B3": "0", "BN_DWMAPT": "0", "BN_MBMAP_B3": "0"}]
FULL_SIZE_IDS = ["default", "no_banded_small_map_kernels", "two_phase_small_map_kernel_one_tile_per_gemm_wave", "tiled_mbconv_and_gemm", "no_small_map_kernels", "no_column_streaming", "matrix_front_end", "dma_gemm_everywhere",
                  "round4_small_map_kernels", "exact_f32_gemms", "bf16x3_lds_dma_form_everywhere", "padded_copy_launches_f32_row_and_map_expand_runtime_size_depthwise"]


@pytest.fixture(scope="module")
def v30_full():
    data = synth.birdnet_v30()
    x = synth.synthetic_segments(3, 160000, 32000)
    x[1] = 0.0  # a silent segment (tests/integration_test.rs:52-54)
    return data, write_model(data), x, onnx_ref.run_model(data, x)


@pytest.fixture(scope="module")
def perch_full():
    data = synth.perch_v2()
    x = synth.synthetic_segments(2, 160000, 32000)
    return data, write_model(data), x, onnx_ref.run_model(data, x)


@pytest.mark.parametrize("env", FULL_SIZE_ENVS, ids=FULL_SIZE_IDS)
def test_v30_full_size_model_against_the_oracle(bn, v30_full, monkeypatch, env):
    """synth.birdnet_v30() as bench.py's `extra.v30` times it: logits and the 1024-d embeddings (detection.rs:44-56,
    classifier.rs:917-934: output 1 = logits, output 0 = embeddings) within the suite's tolerance of the oracle, identical top-1."""
    data, path, x, ref = v30_full
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    clf = bn.Classifier.builder().model_path(path).labels(labels(ref["output_1"].shape[1])).top_k(5).with_rocm().build()
    cfg = clf.config()
    assert (cfg.model_type, cfg.sample_count, cfg.embedding_dim) == (bn.ModelType.BirdNetV30, 160000, 1024)
    check_results(clf.predict_batch(list(x)), ref["output_1"], ref["output_0"], 5, None)
    check_results([clf.predict(x[2])], ref["output_1"][2:], ref["output_0"][2:], 5, None)


@pytest.mark.parametrize("env", FULL_SIZE_ENVS, ids=FULL_SIZE_IDS)
def test_perch_full_size_model_against_the_oracle(bn, perch_full, monkeypatch, env):
    """synth.perch_v2() as bench.py's `extra.perch` times it: logits [14795] + embedding [1536] (outputs 3 and 0,
    detection.rs:58-71), and through the native context the spectrogram [500,128] and the spatial embedding [16,4,1536]."""
    data, path, x, ref = perch_full
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    clf = bn.Classifier.builder().model_path(path).labels(labels(14795)).top_k(5).with_rocm().build()
    cfg = clf.config()
    assert (cfg.model_type, cfg.sample_count, cfg.num_species, cfg.embedding_dim) == (bn.ModelType.PerchV2, 160000, 14795, 1536)
    check_results(clf.predict_batch(list(x)), ref["label"], ref["embedding"], 5, None)
    ctx = clf.create_native_batch_context(2, all_outputs=True)
    check_results(clf.predict_batch_with_context(ctx, list(x)), ref["label"], ref["embedding"], 5, None)
    assert_close(ctx.read_output(2, 2).reshape(ref["spectrogram"].shape), ref["spectrogram"], "spectrogram")
    assert_close(ctx.read_output(1, 2).reshape(ref["spatial_embedding"].shape), ref["spatial_embedding"], "spatial")


# ---- the kernel INSTANCES configs[2] / configs[3] are measured on, tied to the oracle-checked bits -------------
# gemm_dma's tile (gemm_dma.hip: the 64-row tile once batch x tiles >= min_blocks), mbmap's channel grouping, se_fc's G and the
# FFT's frames per block all follow the batch, so batch 64 / 128 run other instances than the batch-3 / batch-2 contexts the
# oracle tests above hold to the oracle.  A segment's bits must not depend on the batch it rides in (classifier.rs:917-934
# slices rows of one batched run; detection.rs:44-71 fixes which outputs those are).
@pytest.mark.parametrize("family,batch", [("v30", 64), ("perch", 128)])
def test_bench_batch_instances_repeat_the_oracle_checked_bits(bn, family, batch, v30_full, perch_full, monkeypatch):
    data, path, x, ref = v30_full if family == "v30" else perch_full
    n = x.shape[0]
    lkey, ekey = ("output_1", "output_0") if family == "v30" else ("label", "embedding")
    small_l, small_e = bn.Context(bn.Model(path), n).infer(x)
    small_l, small_e = small_l.copy(), small_e.copy()
    assert_close(small_l, ref[lkey], family + " logits, small batch")       # the bits below are the oracle-checked ones
    assert_close(small_e, ref[ekey], family + " embeddings, small batch")
    assert np.array_equal(small_l.argmax(1), ref[lkey].argmax(1))
    big = synth.synthetic_segments(batch, 160000, 32000)
    big[:n] = x
    for env in ({}, {"BN_GEMMDMA_MINBLOCKS": "1"}, {"BN_GEMMDMA_MINBLOCKS": "100000000"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        l, e = bn.Context(bn.Model(path), batch).infer(big)
        for k in env:
            monkeypatch.delenv(k)
        assert l[:n].tobytes() == small_l.tobytes(), (family, batch, env, float(np.abs(l[:n] - small_l).max()))
        assert e[:n].tobytes() == small_e.tobytes(), (family, batch, env, float(np.abs(e[:n] - small_e).max()))
    # ... and the same rows at the END of the batch (another block / strip / group of every launch), default plan
    l2, e2 = bn.Context(bn.Model(path), batch).infer(np.roll(big, -n, axis=0))
    assert l2[batch - n:].tobytes() == small_l.tobytes() and e2[batch - n:].tobytes() == small_e.tobytes()


# ---- the same v2.4 network in the spellings an exporter may choose for its spectrogram ---------------------------
@pytest.mark.parametrize("front_end", ["dft", "stft"])
@pytest.mark.parametrize("canon", ["1", "0"])
def test_v24_front_end_dialects(bn, v24_small, front_end, canon, monkeypatch):
    """BirdNET v2.4 (reduced width, full-size front end) authored with ONNX DFT nodes behind tf.signal.frame's Reshape / Gather / Reshape
    and a window Mul ("dft"), or with opset-17 STFT nodes ("stft"), instead of Conv banks: Session::commit_from_file loads any of them
    (classifier.rs:340-350) and so does the planner.  With the node-level canonicalisation (default) the plan is the Conv dialect's plan,
    launch for launch; with BN_CANON_SPECTRO=0 the nodes go through lower_dft / lower_stft (cos | -sin banks on the framing kernels, the
    real half picked by a Gather view).  Both within the suite's tolerance of the oracle -- which evaluates DFT / STFT with torch.fft --
    with identical top-1, and within it of the Conv dialect's logits."""
    conv_data, conv_path = v24_small
    data = synth.birdnet_v24(num_species=500, width=0.5, depth=0.5, head=256, front_end=front_end)  # v24_small with another front end: same seed, same weights
    path = write_model(data)
    monkeypatch.setenv("BN_CANON_SPECTRO", canon)
    desc = bn.plan_describe(path)
    conv_desc = bn.plan_describe(conv_path)
    n_launch = lambda d: len([l for l in d.splitlines() if l[:3].strip().isdigit()])
    if canon == "1":
        assert n_launch(desc) == n_launch(conv_desc) and "~quarter" in desc and "~re" in desc, desc
        assert desc.splitlines()[-2].split("macs_mfma=")[1].split()[0] == conv_desc.splitlines()[-2].split("macs_mfma=")[1].split()[0]
    else:
        assert ("dft:DFT_" in desc) == (front_end == "dft") and ("stft:STFT_" in desc) == (front_end == "stft"), desc
    x = synth.synthetic_segments(3, 144000, 48000)
    x[1] = 0.0
    got, _ = bn.Context(bn.Model(path), 3).infer(x)
    ref = onnx_ref.run_model(data, x)["output"]
    assert_close(got, ref, f"{front_end} dialect, canon={canon}")
    assert np.array_equal(got.argmax(1), ref.argmax(1))
    conv_got, _ = bn.Context(bn.Model(conv_path), 3).infer(x)
    assert_close(got, conv_got, f"{front_end} dialect against the Conv dialect")
