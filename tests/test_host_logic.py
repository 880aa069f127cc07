"""CPU tests of the host-side logic behind the C ABI: detection rules, label parsing,
chunking, the ONNX reader and the launch planner (no GPU compute)."""
import ctypes as C
import importlib

import numpy as np
import pytest

import oracle

synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
writer = importlib.import_module("rust-birdnet-onnx_amd.onnx_writer")


def detect(bn, in_shape, out_shapes, override=-1):
    ins = np.asarray(in_shape, dtype=np.int64)
    flat = np.asarray([d for s in out_shapes for d in s] or [0], dtype=np.int64)
    ranks = (C.c_size_t * max(len(out_shapes), 1))(*[len(s) for s in out_shapes])
    cfg = bn.BnModelConfig()
    i64p = C.POINTER(C.c_int64)
    st = bn.lib.bn_detect_model_type(ins.ctypes.data_as(i64p), len(ins), flat.ctypes.data_as(i64p), ranks,
                                     len(out_shapes), override, C.byref(cfg))
    return (cfg if st == 0 else None), bn.last_error()


CASES = [
    ([1, 144000], [[1, 6522]], -1),
    ([1, 160000], [[1, 1024], [1, 1000]], -1),
    ([1, 160000], [[1, 1536], [1, 16, 4, 1536], [1, 500, 128], [1, 14795]], -1),
    ([1, 160000], [[1, 512], [1, 16, 4, 512], [1, 500, 128], [1, 500]], 2),
    ([1, 160000], [[1, 1024], [1, 1000]], 0),
    ([1, 100000], [[1, 1000]], -1),
    ([1, 1, 144000], [[1, 10]], -1),
    ([-1, 144000], [[-1, 10]], -1),
    ([144000], [[1, 10]], -1),
    ([1, 144000], [[1, 5], [1, 6]], 0),
    ([1, 160000], [[1, 5]], 1),
    ([1, 160000], [[1, 5], [1, 6]], 2),
    ([1, 144000], [[]], -1),
]


@pytest.mark.parametrize("in_shape,out_shapes,override", CASES)
def test_detection_matches_oracle(bn, in_shape, out_shapes, override):
    # detection.rs:183-284 KATs + error paths; the oracle is pinned by the same KATs
    got, msg = detect(bn, in_shape, out_shapes, override)
    want = oracle.detect_model_type(in_shape, out_shapes, None if override < 0 else override)
    assert (got is None) == (want is None), msg
    if got is not None:
        for f in ("model_type", "sample_rate", "segment_duration", "sample_count", "num_species", "has_embedding",
                  "embedding_dim"):
            assert getattr(got, f) == getattr(want, f), f
        assert got.logits_output == {0: 0, 1: 1, 2: 3}[got.model_type]
        assert got.embedding_output == (-1 if got.model_type == 0 else 0)


def test_detection_error_text(bn):
    _, msg = detect(bn, [1, 100000], [[1, 1000]])
    assert "unsupported model: 100000 samples, 1 outputs" in msg  # detection.rs:73-78, tested at :279-280
    _, msg = detect(bn, [1, 160000], [[1, 1024], [1, 1000]], 0)
    assert "BirdNetV24 expects 144000 samples, but model has 160000" in msg  # detection.rs:90-96


def test_parse_text_labels(bn):  # labels.rs:42-48
    assert bn.parse_labels("a\n  b  \n\n\nc\r\n", False) == ["a", "b", "c"]
    assert bn.parse_labels("", False) == []


def test_parse_csv_labels(bn):  # labels.rs:51-95
    assert bn.parse_labels("label,other\nfoo,1\nbar,2\n", True) == ["foo", "bar"]
    assert bn.parse_labels("inat2024_fsd50k\nx\ny\n", True) == ["x", "y"]
    assert bn.parse_labels("robin,1\nwren,2\n", True) == ["robin", "wren"]          # no header
    assert bn.parse_labels('"a, b",1\nc,2\n', True) == ["a, b", "c"]                  # quoted first column


def test_reference_label_files_parse_to_the_pinned_counts(bn):
    import os
    base = "/root/reference/data/labels"
    if not os.path.isdir(base):
        pytest.skip("reference tree not present (GPU box)")
    txt = open(os.path.join(base, "birdnet_v2.4", "BirdNET_GLOBAL_6K_V2.4_Labels_en_uk.txt"), encoding="utf-8").read()
    assert len(bn.parse_labels(txt, False)) == 6522
    csv = open(os.path.join(base, "perch_v2", "labels.csv"), encoding="utf-8").read()
    assert len(bn.parse_labels(csv, True)) == 14795


@pytest.mark.parametrize("n,seg,ov,sr", [(144000 * 3 + 10, 144000, 0.0, 48000), (300000, 144000, 1.5, 48000),
                                         (1000, 144000, 3.0, 48000), (1000, 144000, 4.0, 48000), (0, 144000, 0.0, 48000),
                                         (160000 * 7, 160000, 2.5, 32000), (999999, 144000, 2.999, 48000)])
def test_chunk_plan_matches_oracle(bn, n, seg, ov, sr):  # birdnet-analyze.rs:707-743
    s, t = bn.chunk_plan(n, seg, ov, sr)
    so, to = oracle.chunk_plan(n, seg, ov, sr)
    assert s.tolist() == so.tolist()
    assert t.tobytes() == to.tobytes()


def test_plan_of_v24_model(bn, tmp_path, monkeypatch):
    p = tmp_path / "m.onnx"
    p.write_bytes(synth.birdnet_v24(num_species=100, width=0.5, depth=0.5, head=128))
    # BN_STFT=1: both windowed-DFT banks are recognised and run as real FFTs, with the sparse mel filter banks, their
    # compression chains / layout copies and the min-max normalisation of the signal absorbed into the same launches
    monkeypatch.setenv("BN_STFT", "1")
    full = bn.plan_describe(str(p))
    fft = [l for l in full.splitlines() if " FFT " in l]
    assert len(fft) == 2 and all("mel=96" in l and "pre=4" in l and "post=3" in l for l in fft), fft
    assert " L=2048 hop=278 " in fft[0] and " L=1024 hop=280 " in fft[1]
    assert max(int(l.split("bins=")[1].split()[0]) for l in fft) < 400  # mel-dead bins are not even untangled
    assert "~" not in full and "Sub:Sub_2" in fft[0] and "Sub:Sub_2" in fft[1]
    tot = full.splitlines()[[i for i, l in enumerate(full.splitlines()) if l.startswith("TOTAL")][0]]
    fft_flops, dft_macs = float(tot.split("fft_flops=")[1].split()[0]), float(tot.split("dft_gemm_macs=")[1].split()[0])
    assert 3e7 < fft_flops < 6e7 and dft_macs > 5 * fft_flops  # ~42 MFLOP as FFTs vs the matrix-product count
    # the round-3 plan picks per bank by estimated cost: the L = 2048 bank keeps few live bins and stays a
    # folded matrix product, the L = 1024 bank with its ~300 live bins runs as an FFT with its mel bank absorbed; the
    # normalisation pass stayed a launch of its own because the matrix branch still read its output (BN_FRAME_PRE=0) -- since round 4
    # the folded framing GEMM applies the chain while it loads its span, like the FFT launch
    monkeypatch.delenv("BN_STFT")
    monkeypatch.setenv("BN_CONVMERGE", "0")
    monkeypatch.setenv("BN_CONVFOLD2", "0")
    monkeypatch.setenv("BN_FRAME_PRE", "0")
    auto = bn.plan_describe(str(p))
    afft = [l for l in auto.splitlines() if " FFT " in l]
    assert len(afft) == 1 and " L=1024 hop=280 " in afft[0] and "mel=96" in afft[0] and "pre=0" in afft[0], afft
    assert sum("~sym" in l and " K=1024 " in l for l in auto.splitlines()) == 1, auto
    assert sum(" ELT " in l and "Sub:Sub_2" in l for l in auto.splitlines()) == 1
    monkeypatch.delenv("BN_FRAME_PRE")
    auto = bn.plan_describe(str(p)).splitlines()
    assert sum(" FFT " in l and "pre=4" in l and "Sub:Sub_2" in l for l in auto) == 1 and sum("~sym" in l and " K=1024 " in l and "pre=4" in l for l in auto) == 1, auto
    assert not any(" ELT " in l and "Sub:Sub_2" in l for l in auto)
    # round 4, the default: the cosine-only L = 2048 bank quarter-folded (even bins against S, odd bins against D: L/4 + 1 taps each, K
    # padded to whole steps), the L = 1024 bank and the mel product behind it merged into 96 symmetric filters of 512 folded taps with
    # the compression chain in the launch -- no FFT, no mel launch for that branch
    monkeypatch.delenv("BN_CONVMERGE")
    monkeypatch.delenv("BN_CONVFOLD2")
    r4 = bn.plan_describe(str(p)).splitlines()
    assert not any(" FFT " in l for l in r4), r4
    assert sum("~quarter" in l and " K=544 " in l and "fold=2/2048" in l and "kernel=frame_fold2" in l for l in r4) == 1, r4
    assert sum("Conv_18~sym" in l and " K=512 " in l and " N=96 " in l and "post=3" in l and "kernel=frame_fold" in l for l in r4) == 1, r4
    assert sum("MatMul:" in l for l in r4) == 1  # (the 2048-point branch keeps its mel product)
    # ... and both framing launches apply the min-max normalisation while they load their spans: the normalised segment is never written
    assert sum(("~quarter" in l or "Conv_18~sym" in l) and "pre=4" in l and "Sub:Sub_2" in l for l in r4) == 2 and not any(" ELT " in l and "Sub:Sub_2" in l for l in r4), r4
    # BN_STFT=0 with the round-4 rules off: both banks as half-folded GEMMs
    monkeypatch.setenv("BN_CONVMERGE", "0")
    monkeypatch.setenv("BN_CONVFOLD2", "0")
    monkeypatch.setenv("BN_STFT", "0")
    text = bn.plan_describe(str(p))
    lines = text.splitlines()
    kinds = [l.split()[1] for l in lines if l[:3].strip().isdigit()]
    # Conv+BN+ReLU fused: no standalone BN/Relu launches beyond the explicit spectrogram BN (2 ELT ops)
    # (an expand 1x1 conv + depthwise pair may be fused into one MBCONV launch)
    # (the stem conv + its depthwise conv are one "stem:" MBCONV launch too, so no standalone CONV remains)
    assert kinds.count("DWCONV") + kinds.count("MBCONV") >= 7 and kinds.count("GEMM") + kinds.count("MBCONV") >= 20
    assert kinds.count("CONV") + sum("stem:" in l for l in lines) == 1
    assert "OUTPUT 0 output computed=1 row_elems=100" in text
    # mel filterbank zero rows pruned the DFT conv: 1025 -> <200 bins and 513 -> <400
    # ... and the Hann-windowed cosine bases are symmetric about the frame centre: folded GEMMs with half the taps
    gemm_n = [int(l.split("N=")[1].split()[0]) for l in lines if "~sym" in l and ((" K=1024 " in l and "lda=278" in l) or (" K=512 " in l and "lda=280" in l))]
    assert len(gemm_n) == 2 and max(gemm_n) < 400, gemm_n


def test_stft_absorption_switches(bn, tmp_path, monkeypatch):
    p = tmp_path / "m.onnx"
    p.write_bytes(synth.birdnet_v24(num_species=100, width=0.5, depth=0.5, head=128))
    monkeypatch.setenv("BN_STFT", "1")
    monkeypatch.setenv("BN_STFT_MEL", "0")
    lines = bn.plan_describe(str(p)).splitlines()
    assert sum(" FFT " in l and "mel=0" in l for l in lines) == 2 and sum("MatMul:MatMul_11" in l and " GEMM " in l for l in lines) == 1
    monkeypatch.delenv("BN_STFT_MEL")
    monkeypatch.setenv("BN_STFT_PRE", "0")
    lines = bn.plan_describe(str(p)).splitlines()
    assert sum(" FFT " in l and "pre=0" in l for l in lines) == 2 and sum(" ELT " in l and "Sub:Sub_2" in l for l in lines) == 1


def test_round3_planner_rules(bn, tmp_path, monkeypatch):
    """Planner rules of round 3, on the CPU (plan_describe needs no device): the mel bank of the FFT launch as 16 x 16 tiles for
    the matrix cores; the magnitude pass behind v3.0's cos | sin bank folded into the FFT launch (and the dense-ish mel bank
    left a GEMM unless forced); v3.0's 4 x 16 late stage through the LDS-resident whole-map MBConv kernel; balanced bands of the
    row-streaming MBConv; every rule with its switch."""
    p24 = tmp_path / "v24.onnx"
    p24.write_bytes(synth.birdnet_v24())
    monkeypatch.setenv("BN_CONVMERGE", "0")  # (round 4 merges this bank with its mel product: no FFT in the default v2.4 plan any more)
    d24 = bn.plan_describe(str(p24))
    fft = [l for l in d24.splitlines() if " FFT " in l]
    assert len(fft) == 1 and "mel=96(mfma)" in fft[0] and "power=0" in fft[0] and "tpb=16" in fft[0], fft
    rows = {l.split()[2]: int(l.split("rows=")[1].split()[0]) for l in d24.splitlines() if " MBCONV " in l and "rows=" in l}
    assert rows["stem:Conv_28+Conv_31"] == 8 and rows["mbconv:Conv_42+Conv_45"] == 12 and rows["mbconv:Conv_85+Conv_88"] == 12, rows
    monkeypatch.setenv("BN_STFT_MELMFMA", "0")
    assert "mel=96(csr)" in bn.plan_describe(str(p24))
    monkeypatch.delenv("BN_STFT_MELMFMA")
    p30 = tmp_path / "v30.onnx"
    p30.write_bytes(synth.birdnet_v30())
    d30 = bn.plan_describe(str(p30))
    fft = [l for l in d30.splitlines() if " FFT " in l]
    assert len(fft) == 1 and "bins=513" in fft[0] and "power=2" in fft[0] and "mel=0" in fft[0] and "Sqrt:" in fft[0], fft
    assert not any(" ELT " in l and "Sqrt:" in l for l in d30.splitlines())
    assert sum(" MBCONV " in l and "4x16x192->(1152)->4x16x1152" in l and "tiles=1x1" in l for l in d30.splitlines()) == 4, d30
    monkeypatch.setenv("BN_STFT_POWER", "0")
    d30b = bn.plan_describe(str(p30))
    assert "bins=1026" in d30b and any(" ELT " in l and "Sqrt:" in l for l in d30b.splitlines())
    monkeypatch.delenv("BN_STFT_POWER")
    monkeypatch.setenv("BN_STFT_MEL", "force")
    one = [l for l in bn.plan_describe(str(p30)).splitlines() if " FFT " in l]
    assert "power=2" in one[0] and "tpb=8" in one[0] and "mel=128(csr)" in one[0] and "MatMul:" in one[0], one
    monkeypatch.delenv("BN_STFT_MEL")
    monkeypatch.setenv("BN_MBMAP2", "0")
    assert not any(" MBCONV " in l and "4x16x192" in l for l in bn.plan_describe(str(p30)).splitlines())


def test_plan_without_folding(bn, tmp_path, monkeypatch):
    monkeypatch.setenv("BN_CONVFOLD", "0")
    p = tmp_path / "m.onnx"
    p.write_bytes(synth.birdnet_v24(num_species=100, width=0.5, depth=0.5, head=128))
    lines = bn.plan_describe(str(p)).splitlines()
    assert not any("~" in l for l in lines)
    gemm_n = [int(l.split("N=")[1].split()[0]) for l in lines if " K=2048 " in l or " K=1024 " in l and "lda=28" in l]
    assert len(gemm_n) == 2 and max(gemm_n) < 400, gemm_n
    # a looser tolerance than the bases' rounding noise is never needed; a zero tolerance only folds exact mirror images
    monkeypatch.delenv("BN_CONVFOLD")
    monkeypatch.setenv("BN_CONVFOLD_TOL", "0")
    monkeypatch.setenv("BN_CONVMERGE", "0")
    assert not any("~" in l or " FFT " in l for l in bn.plan_describe(str(p)).splitlines())
    # (round 4: the 96 merged filters of the 1024-point branch are made exact mirror images when they are rounded: they fold at tolerance zero)
    monkeypatch.delenv("BN_CONVMERGE")
    tilde = [l for l in bn.plan_describe(str(p)).splitlines() if "~" in l]
    assert len(tilde) == 1 and "Conv_18~sym" in tilde[0] and " N=96 " in tilde[0], tilde


def test_dead_outputs_are_not_planned(bn, tmp_path):
    p = tmp_path / "p.onnx"
    p.write_bytes(synth.perch_v2(num_species=50, width=0.25, depth=0.25, emb=64))
    main = bn.plan_describe(str(p))
    full = bn.plan_describe(str(p), all_outputs=True)
    assert "OUTPUT 1 spatial_embedding computed=0" in main and "OUTPUT 2 spectrogram computed=0" in main
    assert "OUTPUT 1 spatial_embedding computed=1" in full and "OUTPUT 2 spectrogram computed=1" in full
    assert "OUTPUT 0 embedding computed=1 row_elems=64" in main and "OUTPUT 3 label computed=1 row_elems=50" in main


def test_malformed_and_unsupported_models(bn, tmp_path):
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x00\x01garbage that is not a protobuf")
    with pytest.raises(bn.EngineError) as e:
        bn.plan_describe(str(bad))
    assert e.value.status == 6  # BN_ERR_MODEL_LOAD
    with pytest.raises(bn.EngineError):
        bn.plan_describe(str(tmp_path / "missing.onnx"))
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    y = g.node("LSTM", ["input"], outputs=["output"])
    g.add_output("output", [None, 10])
    un = tmp_path / "un.onnx"
    un.write_bytes(g.serialize())
    with pytest.raises(bn.EngineError) as e:
        bn.plan_describe(str(un))
    assert e.value.status == 7 and "LSTM" in str(e.value)  # BN_ERR_UNSUPPORTED_MODEL names the node


def test_builder_required_fields(bn):  # classifier.rs:336-337, tests :1105-1125
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().labels(["a"]).build()
    assert e.value.kind == bn.ErrorKind.ModelPathRequired and str(e.value) == "model path required"
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().model_path("x.onnx").build()
    assert e.value.kind == bn.ErrorKind.LabelsRequired and str(e.value) == "labels required (provide path or vec)"


def test_inference_options(bn):  # inference_options.rs:116-199
    o = bn.InferenceOptions()
    assert not o.needs_monitor() and o.timeout is None and o.cancellation_token is None
    assert bn.InferenceOptions.with_timeout_of(30).needs_monitor()
    t = bn.CancellationToken()
    assert not t.is_cancelled()
    o = bn.InferenceOptions().with_timeout(1.0).with_cancellation_token(t)
    t.cancel()
    assert o.cancellation_token.is_cancelled() and o.needs_monitor()


def test_model_type_constants(bn):  # types.rs:14-44, tests :194-235
    M = bn.ModelType
    assert (M.BirdNetV24.sample_rate(), M.BirdNetV24.segment_duration(), M.BirdNetV24.sample_count()) == (48000, 3.0, 144000)
    assert (M.BirdNetV30.sample_rate(), M.BirdNetV30.segment_duration(), M.BirdNetV30.sample_count()) == (32000, 5.0, 160000)
    assert (M.PerchV2.sample_rate(), M.PerchV2.sample_count()) == (32000, 160000)
    assert not M.BirdNetV24.has_embeddings() and M.BirdNetV30.has_embeddings() and M.PerchV2.has_embeddings()


def test_engine_chunk_count_matches_chunk_audio(bn):
    """bn_chunk_count (pure host arithmetic) == number of chunks of the reference's chunk_audio
    (src/bin/birdnet-analyze.rs:707-743) as restated by the oracle and by the host mirror."""
    import oracle
    for n in (0, 1, 999, 144000, 144001, 300000, 144000 * 7 + 3):
        for S, sr in ((144000, 48000), (160000, 32000)):
            for overlap in (0.0, 0.5, 1.5, 2.9):
                step = S - int(np.floor(np.float32(overlap) * np.float32(sr)))
                starts, _ = oracle.chunk_plan(n, S, overlap, sr)
                assert bn.lib.bn_chunk_count(n, step) == len(starts)
                assert len(bn.chunk_plan(n, S, overlap, sr)[0]) == len(starts)
    assert bn.lib.bn_chunk_count(1000, 0) == 0


# ---- range filter host logic (reference src/rangefilter.rs): the compiled C++ mirror against the oracle ----
def test_range_filter_week_and_validation(bn):
    import oracle
    for m in range(1, 13):
        for d in range(1, 32):
            assert bn.calculate_week(m, d) == oracle.calculate_week(m, d)
    for lat, lon in ((45.0, -122.0), (-90.0, -180.0), (90.0, 180.0), (91.0, 0.0), (0.0, 181.0), (-90.5, 10.0), (float("nan"), 0.0), (95.0, 200.0)):
        code = oracle.validate_coordinates(lat, lon)
        if code == 0:
            bn.validate_coordinates(lat, lon)
            continue
        with pytest.raises(bn.Error) as e:
            bn.validate_coordinates(lat, lon)
        assert e.value.kind == bn.ErrorKind.InvalidCoordinates
        assert ("latitude must be in range [-90, 90]" in str(e.value)) == (code == 1)
        assert ("longitude must be in range [-180, 180]" in str(e.value)) == (code == 2)
    with pytest.raises(bn.Error) as e:          # Display text of src/error.rs:72-73 with Rust's f32 formatting
        bn.validate_coordinates(95.0, 200.0)
    assert str(e.value) == "invalid coordinates: latitude: 95, longitude: 200, reason: latitude must be in range [-90, 90], got 95"
    assert (e.value.latitude, e.value.longitude) == (95.0, 200.0)
    with pytest.raises(bn.Error) as e:
        bn.validate_coordinates(-90.5, 0.25)
    assert "latitude: -90.5, longitude: 0.25" in str(e.value)
    for m, d in ((1, 1), (6, 15), (12, 31), (0, 1), (13, 1), (1, 0), (1, 32)):
        code = oracle.validate_date(m, d)
        if code == 0:
            bn.validate_date(m, d)
            continue
        with pytest.raises(bn.Error) as e:
            bn.validate_date(m, d)
        assert e.value.kind == bn.ErrorKind.InvalidDate and (e.value.month, e.value.day) == (m, d)
        assert (f"month must be in range [1, 12], got {m}" in str(e.value)) == (code == 1)
        assert (f"day must be in range [1, 31], got {d}" in str(e.value)) == (code == 2)
    with pytest.raises(bn.Error) as e:
        bn.validate_date(13, 32)
    assert str(e.value) == "invalid date: month: 13, day: 32, reason: month must be in range [1, 12], got 13"


def test_range_filter_filter_predictions_against_oracle(bn):
    import oracle
    rng = np.random.default_rng(3)
    names = [f"Species {i}" for i in range(12)]
    for trial in range(200):
        n_pred, n_loc = int(rng.integers(0, 9)), int(rng.integers(0, 10))
        ps = rng.integers(0, 12, n_pred)
        pc = rng.choice([0.05, 0.3, 0.5, 0.5, 0.8, 0.9], n_pred).astype(np.float32)
        ls = rng.integers(0, 12, n_loc)          # duplicates on purpose: the last entry wins
        lc = rng.choice([0.001, 0.02, 0.03, 0.5, 0.9, 1.0], n_loc).astype(np.float32)
        thr, rerank = float(rng.choice([0.01, 0.03, 0.1])), bool(rng.integers(0, 2))
        pos, conf = oracle.filter_predictions(ps, pc, ls, lc, thr, rerank)
        preds = [bn.Prediction(names[s], float(c), int(s)) for s, c in zip(ps, pc)]
        locs = [bn.LocationScore(names[s], float(c), int(s)) for s, c in zip(ls, lc)]
        got = bn.filter_predictions(preds, locs, thr, rerank)
        assert [(g.species, np.float32(g.confidence)) for g in got] == [(names[ps[p]], np.float32(c)) for p, c in zip(pos, conf)], trial
    # the reference's own KATs (rangefilter.rs:707-889) through the product
    P, L = bn.Prediction, bn.LocationScore
    out = bn.filter_predictions([P("Species A", 0.8, 0), P("Species B", 0.3, 1), P("Species C", 0.05, 2)],
                                [L("Species A", 0.9, 0), L("Species B", 0.02, 1), L("Species C", 0.5, 2)], 0.03, False)
    assert [p.species for p in out] == ["Species A", "Species C"]
    out = bn.filter_predictions([P("Species A", 0.8, 0), P("Species B", 0.7, 1), P("Species D", 0.9, 3)],
                                [L("Species A", 0.9, 0), L("Species C", 0.8, 2)], 0.03, False)
    assert [(p.species, np.float32(p.confidence), p.index) for p in out] == [("Species A", np.float32(0.8), 0), ("Species B", np.float32(0.7), 1),
                                                                            ("Species D", np.float32(0.9), 3)]


def test_range_filter_builder_errors_without_a_device(bn, tmp_path):
    with pytest.raises(bn.Error) as e:                                   # rangefilter.rs:692-697
        bn.RangeFilter.builder().build()
    assert e.value.kind == bn.ErrorKind.ModelPathRequired
    with pytest.raises(bn.Error) as e:                                   # rangefilter.rs:699-704
        bn.RangeFilter.builder().model_path("/tmp/model.onnx").build()
    assert e.value.kind == bn.ErrorKind.LabelsRequired
    with pytest.raises(bn.Error) as e:
        bn.RangeFilter.builder().model_path("/tmp/model.onnx").labels_path(str(tmp_path / "missing.txt")).build()
    assert e.value.kind == bn.ErrorKind.LabelLoad


def test_resampler_table_matches_the_oracle_design(bn):
    """Polyphase table of bn_recording_create_resampled (host arithmetic) == oracle/resample.py, bit for bit."""
    from oracle import resample as R
    for src, dst, zc in ((44100, 48000, 0), (48000, 32000, 0), (22050, 48000, 16), (16000, 32000, 8), (96000, 48000, 0), (44100, 32000, 12), (8000, 48000, 0)):
        tab, L, M, T = bn.resample_table(src, dst, zc)
        want, oL, oM, oT = R.make_table(src, dst, zc)
        assert (L, M, T) == (oL, oM, oT) and tab.shape == want.shape
        assert np.array_equal(tab.view(np.uint32), want.view(np.uint32))
        assert np.allclose(tab.sum(axis=1), 1.0, atol=2e-7)            # every phase has unit DC gain


def test_resampler_oracle_against_scipy_on_band_limited_signals():
    """The oracle's design is a sane resampler: a tone well inside both Nyquist bands comes out as the same tone
    (scipy's resample_poly as an independent implementation; loose tolerance, interior samples only)."""
    from scipy import signal

    from oracle import resample as R
    for src, dst in ((44100, 48000), (48000, 32000), (22050, 48000)):
        t = np.arange(src) / src
        x = (0.6 * np.sin(2 * np.pi * 1000.0 * t) + 0.3 * np.sin(2 * np.pi * 3300.0 * t + 0.4)).astype(np.float32)
        y = R.resample(x, src, dst)
        assert len(y) == -(-len(x) * dst // src)
        td = np.arange(len(y)) / dst
        ideal = 0.6 * np.sin(2 * np.pi * 1000.0 * td) + 0.3 * np.sin(2 * np.pi * 3300.0 * td + 0.4)
        mid = slice(2000, len(y) - 2000)
        assert np.abs(y[mid] - ideal[mid]).max() < 2e-4
        g = np.gcd(src, dst)
        ys = signal.resample_poly(x.astype(np.float64), dst // g, src // g, window=("kaiser", 8.6))
        assert np.abs(y[mid] - ys[mid]).max() < 5e-3
    # a tone above the new Nyquist is removed when down-sampling
    t = np.arange(48000) / 48000
    y = R.resample((0.8 * np.sin(2 * np.pi * 20000.0 * t)).astype(np.float32), 48000, 32000)
    assert np.abs(y[2000:-2000]).max() < 1e-3


@pytest.mark.parametrize("op,kwargs,extra,needle", [
    ("DFT", {"inverse": 1}, 0, "inverse DFT is outside the native subset"),      # (round 5: forward DFT, Where and the comparisons are mapped)
    ("DFT", {}, 0, "input must end in a dimension of 1"),
    ("Resize", {"mode": "nearest"}, 0, "resampling of feature maps"),
    ("GatherND", {}, 1, "gathers of activations are mapped only where"),
    ("NonMaxSuppression", {}, 0, "outside the native subset"),
])
def test_unmapped_exporter_nodes_are_refused_by_name(bn, tmp_path, op, kwargs, extra, needle):
    """Exporter dialects the path does not map are refused at load with the node's name and type,
    never run on some fallback: checked through the planner alone (bn_plan_describe needs no device)."""
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    ins = ["input"] + [g.const(np.ones((1,), dtype=np.float32)) for _ in range(extra)]
    y = g.node(op, ins, **kwargs)  # the writer names it "<op>_1"
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None, 144000])
    p = tmp_path / "m.onnx"
    p.write_bytes(g.serialize())
    with pytest.raises(bn.EngineError) as e:
        bn.plan_describe(str(p))
    msg = str(e.value)
    assert f"'{op}_1'" in msg and f"({op})" in msg and needle in msg, msg


def test_stft_node_plans_as_framing_convs(bn, tmp_path, monkeypatch):
    """An opset-17 STFT node with a periodic Hann window becomes one cos block and one sin block of folded framing GEMMs
    (and, under BN_STFT=1, one FFT launch); its [frames, bins, 2] result is a view, not a copy."""
    n, hop = 512, 128
    k = np.arange(n, dtype=np.float64)
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    s = g.node("STFT", ["input", g.const(np.array(hop, dtype=np.int64)), g.const((0.5 - 0.5 * np.cos(2 * np.pi * k / n)).astype(np.float32))])
    g.node("Identity", [s], outputs=["output"])
    frames = (144000 - n) // hop + 1
    g.add_output("output", [None, frames, n // 2 + 1, 2])
    p = tmp_path / "m.onnx"
    p.write_bytes(g.serialize())
    monkeypatch.setenv("BN_STFT", "0")
    text = bn.plan_describe(str(p))
    gemms = [l for l in text.splitlines() if " GEMM " in l and "stft:STFT_1" in l]
    assert len(gemms) == 2 and all("fold=" in l for l in gemms), text
    assert f"K={n // 2}" in gemms[0], text
    # default: 257 complex bins of a 512-point transform -- the FFT is estimated far cheaper and takes the node
    monkeypatch.delenv("BN_STFT")
    text = bn.plan_describe(str(p))
    assert text.count(" FFT ") == 1 and " GEMM " not in text, text


def test_tile_of_a_real_dimension_is_refused(bn, tmp_path):
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    y = g.node("Tile", ["input", g.const(np.array([1, 2], dtype=np.int64))])
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None, 288000])
    p = tmp_path / "m.onnx"
    p.write_bytes(g.serialize())
    with pytest.raises(bn.EngineError) as e:
        bn.plan_describe(str(p))
    assert "'Tile_1'" in str(e.value) and "only size-1 dimensions" in str(e.value)


# ---------------------------------------------------------------- round 5: planner-side facts that need no device
def test_spectrogram_dialects_reach_one_plan(bn, tmp_path, monkeypatch):
    """BirdNET v2.4 authored with Conv banks, with tf.signal.frame + window Mul + ONNX DFT, and with opset-17 STFT nodes: after the
    node-level canonicalisation the three files plan to the same launches (kind, kernel, shape and multiply-adds of every launch); with
    BN_CANON_SPECTRO=0 the DFT / STFT nodes are lowered directly instead (framing kernels over the strided view of the signal: no
    Gather / Reshape copy of the frames)."""
    def plan(fe):
        p = tmp_path / f"{fe}.onnx"
        p.write_bytes(synth.birdnet_v24(num_species=300, width=0.25, depth=0.25, head=128, front_end=fe))
        return bn.plan_describe(str(p))
    def shape(d):
        rows = [l.split() for l in d.splitlines() if l[:3].strip().isdigit()]
        return [(r[1], [t for t in r[3:] if not t.startswith("bytes=")]) for r in rows], d.splitlines()[-2]
    plans = {fe: plan(fe) for fe in ("conv", "dft", "stft")}
    assert shape(plans["conv"]) == shape(plans["dft"]) == shape(plans["stft"])
    assert "~re~quarter" in plans["dft"] and "~re~quarter" in plans["stft"] and "kernel=frame_fold2q" in plans["conv"]
    monkeypatch.setenv("BN_CANON_SPECTRO", "0")
    for fe in ("dft", "stft"):
        d = plan(fe)
        first = [l for l in d.splitlines() if " GEMM " in l][0]
        assert ("dft:DFT_" in d) == (fe == "dft") and ("stft:STFT_" in d) == (fe == "stft"), d
        assert "lda=278" in first and "pre=4" in first and "copy(Gather" not in d and "copy(Reshape" not in d, d


def test_bf16x3_error_bound_of_the_six_kept_products():
    """The arithmetic of csrc/bf16x3.h restated in numpy: x = hi + mid + lo exactly (three bf16 terms), the six kept partial products are
    each exact in f32 (8 x 8 significand bits), and what they leave out of x w (mid x lo, lo x mid, lo x lo) is below 2^-21 |x w| in the worst
    case (|mid| < 2^-7 |x|, |lo| < 2^-15 |x|) and 2^-24 |x w| in the root mean square -- the size of one f32 rounding per product."""
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    w = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    top = lambda v: (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    def split(v):
        h = top(v); r1 = v - h; m = top(r1); l = r1 - m
        assert np.array_equal(h + m + l, v) and not (l.view(np.uint32) & np.uint32(0xffff)).any()
        return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)
    xh, xm, xl = split(x)
    wh, wm, wl = split(w)
    kept = [wl * xh, wh * xl, wm * xm, wm * xh, wh * xm, wh * xh]
    for t in kept:  # each kept product is an f32 number
        assert np.array_equal(t.astype(np.float32).astype(np.float64), t)
    exact = x.astype(np.float64) * w.astype(np.float64)
    missing = np.abs(exact - sum(kept))
    assert (np.abs(xm) < 2.0 ** -7 * np.abs(xh)).all() and (np.abs(xl) < 2.0 ** -15 * np.abs(xh)).all()
    rel = missing / np.abs(exact)
    assert rel.max() <= 2.0 ** -21 and np.sqrt((rel ** 2).mean()) <= 1.1 * 2.0 ** -24
    assert np.array_equal(xm * wl + xl * wm + xl * wl, exact - sum(kept))


@pytest.mark.parametrize("op,kwargs,extra", [
    ("Elu", {"alpha": 0.5}, 0), ("Selu", {}, 0), ("Celu", {"alpha": 2.0}, 0), ("ThresholdedRelu", {}, 0), ("Softsign", {}, 0), ("Mish", {}, 0),
    ("Gelu", {}, 0), ("Gelu", {"approximate": "tanh"}, 0), ("Sign", {}, 0), ("Round", {}, 0), ("Sum", {}, 2), ("Mean", {}, 1),
    ("ReduceL1", {"axes": [1], "keepdims": 1}, 0), ("ReduceLogSum", {"axes": [1], "keepdims": 1}, 0), ("ReduceLogSumExp", {"axes": [1], "keepdims": 1}, 0),
    ("LayerNormalization", {"axis": -1}, 1),
])
def test_operators_written_out_by_the_planner_plan_without_refusal(bn, tmp_path, op, kwargs, extra):
    """Round 5: operators the planner writes out as the ones it maps (engine.cpp lower_composite) -- each plans as ELT / REDUCE launches
    carrying the node's name, and the first-contact survey lists the type as mapped."""
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    ins = ["input"] + [g.const(np.ones((1,), dtype=np.float32)) for _ in range(extra)]
    y = g.node(op, ins, **kwargs)
    if op.startswith("Reduce"):
        y = g.node("Mul", [y, g.const(np.ones((144000,), dtype=np.float32))])
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None, 144000])
    p = tmp_path / "m.onnx"
    p.write_bytes(g.serialize())
    text = bn.plan_describe(str(p))
    assert f"{op}_" in text and "TOTAL launches=" in text, text
    status, survey = bn.model_survey(str(p))
    assert status == 0 and "unmapped" not in survey.split(op)[1].splitlines()[0], survey


def test_size_of_an_activation_is_refused_and_of_a_constant_folds(bn, tmp_path):
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    s = g.node("Size", [g.const(np.ones((3, 5), dtype=np.float32))])
    y = g.node("Mul", ["input", g.node("Cast", [s], to=1)])
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None, 144000])
    p = tmp_path / "m.onnx"
    p.write_bytes(g.serialize())
    assert "p0=15" in bn.plan_describe(str(p)) or "ELT" in bn.plan_describe(str(p))
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    y = g.node("Mul", ["input", g.node("Cast", [g.node("Size", ["input"])], to=1)])
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None, 144000])
    p.write_bytes(g.serialize())
    with pytest.raises(bn.EngineError) as e:
        bn.plan_describe(str(p))
    assert "depends on the batch" in str(e.value)


def test_round5_small_map_forms_in_the_three_plans(bn, tmp_path, monkeypatch):
    """Planner facts of round 5, on the CPU: v2.4's ten small-map MBConv blocks, v3.0's ten (six of them 8 x 32 maps in two bands: no
    depthwise launch is left in its plan) and Perch's ten transposed 32 x 8 blocks take the wave-specialised kernel; Perch plans 90 launches
    (six more on the 16 x 4 maps' one-tile-per-wave form): 84 launches, 2 depthwise launches; every form has its switch."""
    def plan(blob, name):
        p = tmp_path / name
        p.write_bytes(blob)
        return bn.plan_describe(str(p))
    d24, d30, dpe = plan(synth.birdnet_v24(), "a.onnx"), plan(synth.birdnet_v30(), "b.onnx"), plan(synth.perch_v2(), "c.onnx")
    assert d24.count(",ws ") == 10 and d24.count("map=cfg") == 10
    assert d30.count(",ws ") == 10 and d30.count("cfg5,bands,ws ") == 6 and " DWCONV " not in d30
    assert dpe.count("cfg5,bands,transposed,ws ") == 10 and dpe.count("cfg6,transposed,ws ") == 6 and dpe.count(" DWCONV ") == 2 and "TOTAL launches=84 " in dpe
    assert "kpad=144" in dpe and "kpad=96" in dpe
    monkeypatch.setenv("BN_MBMAP_WS_TR", "0")
    assert plan(synth.perch_v2(), "c.onnx").count("map=cfg") == 6
    monkeypatch.setenv("BN_MBMAP_WS_DEEP", "0")
    assert "map=cfg" not in plan(synth.perch_v2(), "c.onnx")
    monkeypatch.delenv("BN_MBMAP_WS_TR")
    monkeypatch.delenv("BN_MBMAP_WS_DEEP")
    monkeypatch.setenv("BN_MBMAP_WS_BANDS", "0")
    d30b = plan(synth.birdnet_v30(), "b.onnx")
    assert d30b.count(" DWCONV ") == 6 and d30b.count(",ws ") == 4
    monkeypatch.delenv("BN_MBMAP_WS_BANDS")
    monkeypatch.setenv("BN_MBMAP_WS", "0")
    d24b = plan(synth.birdnet_v24(), "a.onnx")
    assert d24b.count(",b3 ") == 10 and ",ws " not in d24b
    monkeypatch.setenv("BN_MBMAP_B3", "0")
    d24c = plan(synth.birdnet_v24(), "a.onnx")
    assert d24c.count("map=cfg") == 10 and ",b3" not in d24c and ",ws" not in d24c
