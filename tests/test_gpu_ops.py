"""GPU parity of individual kernels: single-operator ONNX graphs through the C ABI vs the oracle."""
import numpy as np
import pytest

import oracle
from oracle import onnx_ref
from gpu_helpers import assert_close, op_graph, write_model

pytestmark = pytest.mark.gpu


def run_both(bn, data, batch=3, seed=0, all_outputs=False, scale=1.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((batch, 144000)) * scale).astype(np.float32)
    path = write_model(data)
    m = bn.Model(path)
    ctx = bn.Context(m, batch + 1, bn.BN_CTX_ALL_OUTPUTS if all_outputs else 0)
    logits, _ = ctx.infer(x)
    ref = onnx_ref.run_model(data, x)["output"]
    return logits.reshape(ref.shape), ref


def conv_case(cin, h, w, cout, k, stride, pad, groups=1, bias=True, act=None, dil=1, seed=0):
    rng = np.random.default_rng(seed)
    assert cin * h * w <= 144000
    wgt = (rng.standard_normal((cout, cin // groups, k, k)) / np.sqrt(cin // groups * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    oh = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        ins = [x, g.const(wgt)] + ([g.const(b)] if bias else [])
        y = g.node("Conv", ins, kernel_shape=[k, k], strides=[stride, stride], pads=[pad] * 4, group=groups,
                   dilations=[dil, dil])
        if act == "relu":
            y = g.node("Relu", [y])
        elif act == "silu":
            y = g.node("Mul", [y, g.node("Sigmoid", [y])])
        elif act == "relu6":
            y = g.node("Clip", [y, g.const(np.float32(0)), g.const(np.float32(6))])
        return y
    return op_graph(build, [cout, oh, ow])


@pytest.mark.parametrize("cin,h,w,cout", [(16, 12, 20, 96), (96, 12, 20, 24), (24, 7, 9, 144), (144, 6, 8, 40),
                                          (40, 5, 7, 33), (8, 3, 5, 7), (320, 3, 16, 128), (13, 4, 6, 130)])
def test_pointwise_conv_gemm(bn, cin, h, w, cout):
    got, ref = run_both(bn, conv_case(cin, h, w, cout, 1, 1, 0, act="relu"))
    assert_close(got, ref, f"1x1 conv {cin}->{cout}")


@pytest.mark.parametrize("cin,cout,act,h,w,expect", [(320, 1024, "relu", 3, 16, True), (128, 200, "silu", 3, 16, True), (192, 36, "relu", 6, 8, True),
                                                      (320, 128, "relu", 4, 16, False), (144, 96, None, 3, 16, True)])
def test_global_average_pool_in_the_gemm_epilogue(bn, cin, cout, act, h, w, expect, monkeypatch):
    """Planner rule K (round 4): GlobalAveragePool behind a 1x1 conv whose 48 rows per sample sit in one block of the LDS-DMA GEMM -- the
    epilogue writes the mean over the rows (v2.4's head conv + pool); ragged channel counts, no activation, a map of another size (64 rows:
    the rule steps aside).  Against the oracle, against the separate reduction launch, and bit-identical across batch sizes / tile widths."""
    rng = np.random.default_rng(cin + cout)

    def build(g, x):  # 1x1 (puts the map into the channels-last layout) -> the conv under test -> pool
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        x = g.node("Conv", [x, g.const((rng.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))], kernel_shape=[1, 1])
        y = g.node("Conv", [x, g.const((rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)),
                            g.const(rng.standard_normal(cout).astype(np.float32))], kernel_shape=[1, 1])
        if act == "relu":
            y = g.node("Relu", [y])
        elif act == "silu":
            y = g.node("Mul", [y, g.node("Sigmoid", [y])])
        p = g.node("GlobalAveragePool", [y])
        return g.node("Flatten", [p], axis=1)
    data = op_graph(build, [cout])
    text = bn.plan_describe(write_model(data))
    assert ("gap=1" in text and "GlobalAveragePool" in [l for l in text.splitlines() if "gap=1" in l][0]) == expect, text
    assert (" REDUCE " in text) == (not expect), text
    got, ref = run_both(bn, data, batch=3)
    assert_close(got, ref, f"pooled 1x1 conv {cin}->{cout}")
    if expect:
        big, _ = run_both(bn, data, batch=40)  # more blocks: another tile width (the width does not enter the arithmetic)
        assert np.array_equal(big[:3].view(np.uint32), got.view(np.uint32))
        monkeypatch.setenv("BN_GEMMGAP", "0")
        assert "gap=1" not in bn.plan_describe(write_model(data))
        sep, _ = run_both(bn, data, batch=3)
        assert_close(got, sep, "pooled epilogue vs separate reduction", atol=1e-5 * float(np.abs(sep).max()) + 1e-6, rtol=0)


@pytest.mark.parametrize("cin,h,w,cout,act,stream", [(80, 8, 32, 480, "silu", True), (112, 8, 32, 672, "silu", True), (96, 32, 8, 576, "silu", True),
                                                     (64, 6, 16, 128, "relu", True), (96, 4, 16, 1100, "relu6", True), (96, 5, 7, 200, "relu", False),
                                                     (48, 8, 32, 288, "silu", False)])
def test_streaming_expand_gemm(bn, cin, h, w, cout, act, stream, monkeypatch):
    """Round 4: the expand convs of the late stages (K 64 .. 127, N >= 128, no gate) on the streaming form of the LDS-DMA GEMM
    (gemm_dma_stream_kernel: a block walks consecutive row tiles -- across samples -- and its ring never drains).  Against the oracle;
    the number of tiles per block must not change a bit (1, 3 and 16 forced; 64- and 32-row tiles by the rows of a sample); the tiled
    kernel (BN_GEMMSTREAM=0) agrees within the tolerance."""
    rng = np.random.default_rng(cin + cout)

    def build(g, x):  # 1x1 (puts the map into the channels-last layout) -> the expand conv under test -> a project conv that reads it
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        x = g.node("Conv", [x, g.const((rng.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))], kernel_shape=[1, 1])
        y = g.node("Conv", [x, g.const((rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)),
                            g.const(rng.standard_normal(cout).astype(np.float32))], kernel_shape=[1, 1])
        if act == "relu":
            y = g.node("Relu", [y])
        elif act == "silu":
            y = g.node("Mul", [y, g.node("Sigmoid", [y])])
        else:
            y = g.node("Clip", [y, g.const(np.float32(0)), g.const(np.float32(6))])
        return g.node("Conv", [y, g.const((rng.standard_normal((24, cout, 1, 1)) / np.sqrt(cout)).astype(np.float32))], kernel_shape=[1, 1])
    data = op_graph(build, [24, h, w])
    assert "kernel=dma-stream" not in bn.plan_describe(write_model(data))  # opt-in (measured equal to the tiled kernel: plan_rules.h)
    monkeypatch.setenv("BN_GEMMSTREAM", "1")
    desc = bn.plan_describe(write_model(data))
    assert ("kernel=dma-stream" in desc) == stream, desc
    got, ref = run_both(bn, data, batch=5)
    assert_close(got, ref, f"expand {cin}->{cout} rows {h * w}")
    if not stream:
        return
    for tpb in ("1", "3", "16"):
        monkeypatch.setenv("BN_GEMMSTREAM_TPB", tpb)
        forced, _ = run_both(bn, data, batch=5)
        assert np.array_equal(forced.view(np.uint32), got.view(np.uint32)), tpb
    monkeypatch.delenv("BN_GEMMSTREAM_TPB")
    one, _ = run_both(bn, data, batch=1)
    assert np.array_equal(one.view(np.uint32), got[:1].view(np.uint32))
    monkeypatch.setenv("BN_GEMMSTREAM", "0")
    assert "kernel=dma-stream" not in bn.plan_describe(write_model(data))
    tiled, _ = run_both(bn, data, batch=5)
    assert_close(tiled, ref, "tiled kernel")
    assert_close(got, tiled, "streaming vs tiled")


@pytest.mark.parametrize("c,h,w,k,stride,act", [(32, 24, 30, 3, 1, "relu"), (96, 24, 30, 3, 2, "relu"),
                                                (144, 12, 17, 5, 2, "silu"), (240, 6, 9, 5, 1, "relu6"),
                                                (30, 9, 11, 3, 1, None), (8, 5, 5, 7, 1, None)])
def test_depthwise_conv(bn, c, h, w, k, stride, act):
    got, ref = run_both(bn, conv_case(c, h, w, c, k, stride, k // 2, groups=c, act=act))
    assert_close(got, ref, f"depthwise {c} k{k} s{stride}")


@pytest.mark.parametrize("cin,cout,k,stride,pad,groups,dil", [(2, 32, 3, 2, 1, 1, 1), (3, 8, 3, 1, 1, 1, 1),
                                                              (8, 16, 3, 1, 1, 2, 1), (4, 6, 5, 2, 2, 1, 1),
                                                              (6, 6, 3, 1, 2, 3, 2), (1, 4, 3, 1, 0, 1, 1)])
def test_direct_conv(bn, cin, cout, k, stride, pad, groups, dil):
    got, ref = run_both(bn, conv_case(cin, 20, 31, cout, k, stride, pad, groups=groups, dil=dil, act="relu"))
    assert_close(got, ref, f"conv {cin}->{cout} k{k} s{stride} g{groups}")


def test_conv_bn_residual_fusion(bn):
    rng = np.random.default_rng(5)
    c = 24

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(c * 10 * 12), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, c, 10, 12)])
        w0 = (rng.standard_normal((c, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
        x = g.node("Relu", [g.node("Conv", [x, g.const(w0)], kernel_shape=[1, 1])])   # makes an NHWC arena tensor
        w1 = (rng.standard_normal((c, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
        y = g.node("Conv", [x, g.const(w1)], kernel_shape=[1, 1])
        y = g.node("BatchNormalization", [y, g.const(rng.uniform(0.5, 1.5, c).astype(np.float32)),
                                          g.const(rng.standard_normal(c).astype(np.float32)),
                                          g.const(rng.standard_normal(c).astype(np.float32)),
                                          g.const(rng.uniform(0.5, 1.5, c).astype(np.float32))], epsilon=1e-3)
        return g.node("Add", [y, x])
    data = op_graph(build, [c, 10, 12])
    got, ref = run_both(bn, data)
    assert_close(got, ref, "conv+bn+residual")
    import tempfile
    text = bn.plan_describe(write_model(data))
    assert "res=1" in text and "bn." not in text  # BatchNorm and the residual Add were folded into the GEMM


@pytest.mark.parametrize("n_fft,hop,bins", [(2048, 278, 40), (1024, 280, 129), (640, 320, 65), (512, 1, 5)])
def test_conv1d_framing_as_gemm(bn, n_fft, hop, bins):
    rng = np.random.default_rng(6)
    w = (rng.standard_normal((bins, 1, n_fft)) / np.sqrt(n_fft)).astype(np.float32)
    L = 144000 if hop > 1 else 4000
    frames = (L - n_fft) // hop + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        if L != 144000:
            x = g.node("Slice", [x, i64(0), i64(L), i64(1), i64(1)])
        u = g.node("Unsqueeze", [x, i64(1)])
        return g.node("Conv", [u, g.const(w)], kernel_shape=[n_fft], strides=[hop])
    got, ref = run_both(bn, op_graph(build, [bins, frames]), batch=2)
    assert_close(got, ref, f"conv1d n_fft={n_fft} hop={hop}")


@pytest.mark.parametrize("n_fft,hop,kind,bias", [(2048, 278, "real", False), (1024, 280, "complex", True),
                                                 (640, 320, "complex", False), (256, 37, "sin-first", True)])
def test_conv1d_folded_dft_framing(bn, n_fft, hop, kind, bias, monkeypatch):
    """Windowed DFT filter banks (cos rows symmetric, sin rows antisymmetric about the frame centre, w[0] == 0) run
    as folded GEMMs with half the K; the result matches the plain convolution of the oracle and of the unfolded plan."""
    import importlib
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    rng = np.random.default_rng(8)
    w = synth.dft_basis(n_fft, "complex")
    half = w.shape[0] // 2
    if kind == "real":
        w = w[5:half - 3]
    elif kind == "sin-first":
        w = np.concatenate([w[half:half + 7], w[3:12], np.zeros((2, 1, n_fft), np.float32)], axis=0)
    w = np.ascontiguousarray(w * rng.uniform(0.5, 2.0, (w.shape[0], 1, 1)).astype(np.float32))
    b = rng.standard_normal(w.shape[0]).astype(np.float32)
    frames = (144000 - n_fft) // hop + 1

    def build(g, x):
        u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
        return g.node("Conv", [u, g.const(w)] + ([g.const(b)] if bias else []), kernel_shape=[n_fft], strides=[hop])
    data = op_graph(build, [w.shape[0], frames])
    pow2 = n_fft & (n_fft - 1) == 0 or n_fft == 640  # (640 = 2 x 5 x 64: radix 5 first, round 4)
    monkeypatch.setenv("BN_STFT", "1")
    text = bn.plan_describe(write_model(data))
    if pow2:
        # recognised as a windowed-DFT bank (bins, window and per-row amplitudes recovered from the taps): ONE real-FFT
        # launch for all rows, cos and sin blocks together
        assert text.count(" FFT ") == 1 and "~" not in text, text
        fft_out, ref = run_both(bn, data, batch=2)
        assert_close(fft_out, ref, f"stft n_fft={n_fft} {kind}", atol=2e-5 * float(np.abs(ref).max()), rtol=0)
        fft3, ref3 = run_both(bn, data, batch=3)  # 511 frames: the last block is ragged either way; odd batch
        assert_close(fft3, ref3, f"stft n_fft={n_fft} {kind} batch 3", atol=2e-5 * float(np.abs(ref3).max()), rtol=0)
    monkeypatch.setenv("BN_STFT", "0")   # the matrix-product path (the default picks per bank by estimated cost)
    text = bn.plan_describe(write_model(data))
    assert "~sym" in text or "~anti" in text, text
    assert text.count("~") == (1 if kind == "real" else 2), text
    got, ref = run_both(bn, data, batch=2)
    assert_close(got, ref, f"folded conv1d n_fft={n_fft} {kind}")
    if pow2:
        assert_close(fft_out, got, "stft vs folded GEMM", atol=2e-5 * float(np.abs(got).max()), rtol=0)
    # the LDS-resident-signal kernel and the generic folded GEMM use the same K order and fold expression
    monkeypatch.setenv("BN_FRAMELDS", "0")
    generic, _ = run_both(bn, data, batch=2)
    assert np.array_equal(got.view(np.uint32), generic.view(np.uint32))
    monkeypatch.delenv("BN_FRAMELDS")
    # ... and so do its two launch shapes (one block per N tile at small batches, one block walking all N tiles of its
    # rows once the row tiles alone fill the chip)
    for walk in ("0", "1"):
        monkeypatch.setenv("BN_FRAME_WALK", walk)
        shaped, _ = run_both(bn, data, batch=2)
        assert np.array_equal(got.view(np.uint32), shaped.view(np.uint32)), walk
    monkeypatch.delenv("BN_FRAME_WALK")
    monkeypatch.setenv("BN_CONVFOLD", "0")
    assert "~" not in bn.plan_describe(write_model(data))
    plain, _ = run_both(bn, data, batch=2)
    assert_close(got, plain, "folded vs unfolded plan", atol=2e-5 * float(np.abs(plain).max()), rtol=0)


@pytest.mark.parametrize("n_fft,hop,which,window", [(2048, 278, "low", "hann"), (1024, 280, "mixed", "tapered"), (256, 100, "even", "hann"),
                                                    (512, 37, "odd", "tapered"), (2048, 200, "wide", "hann")])
def test_conv1d_quarter_folded_cosine_bank(bn, n_fft, hop, which, window, monkeypatch):
    """A bank of windowed cosines only (the real part of an STFT, v2.4's first spectrogram branch) with a window symmetric about the frame
    centre runs as the QUARTER-folded framing GEMM (planner: emit_quarter_fold; kernels.hip: frame_fold2_kernel): even bins against
    S[n] = ye[n] + ye[L/2 - n], odd bins against D[n], L/4 + 1 taps per output.  Scrambled bin order, per-row gains, a dead row, DC and
    Nyquist rows, banks of even bins only / odd bins only / more than one wave column per group, a bias, ragged last row tile (batch 2
    and 3) -- against the oracle, against the half fold and against the plain convolution."""
    rng = np.random.default_rng(n_fft + hop)
    n = np.arange(n_fft, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    if window == "tapered":
        win = (0.54 - 0.46 * np.cos(2.0 * np.pi * n / n_fft)) * np.sin(np.pi * n / n_fft) ** 2
    half = n_fft // 2
    if which == "low":
        bins = np.arange(0, 127)                                  # v2.4: the mel-live bins of the 2048-point branch
    elif which == "mixed":
        bins = np.concatenate([rng.permutation(np.arange(1, half))[:70], [0, half]])
    elif which == "even":
        bins = rng.permutation(np.arange(0, half + 1, 2))[:40]
    elif which == "odd":
        bins = rng.permutation(np.arange(1, half, 2))[:33]
    else:
        bins = np.concatenate([rng.permutation(np.arange(0, half + 1, 2))[:60], rng.permutation(np.arange(1, half, 2))[:90]])  # five wave columns
        bins = rng.permutation(bins)
    rows = [win * np.cos(2.0 * np.pi * k * n / n_fft) * rng.uniform(0.3, 3.0) * rng.choice([-1.0, 1.0]) for k in bins]
    rows.insert(3, np.zeros(n_fft))                               # a dead row: its output is the bias
    w = np.ascontiguousarray(np.array(rows, dtype=np.float32)[:, None, :])
    b = rng.standard_normal(w.shape[0]).astype(np.float32)
    frames = (144000 - n_fft) // hop + 1

    def build(g, x):
        u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
        return g.node("Conv", [u, g.const(w), g.const(b)], kernel_shape=[n_fft], strides=[hop])
    data = op_graph(build, [w.shape[0], frames])
    monkeypatch.setenv("BN_STFT", "0")  # (the default would decide FFT / matrix product per bank by estimated cost)
    text = bn.plan_describe(write_model(data))
    assert "~quarter" in text and "kernel=frame_fold2" in text and f"fold=2/{n_fft}" in text, text
    got, ref = run_both(bn, data, batch=2)
    tol = 2e-5 * float(np.abs(ref).max())
    assert_close(got, ref, f"quarter-folded conv1d n_fft={n_fft} {which}", atol=tol, rtol=0)
    assert np.array_equal(got[:, 3], np.broadcast_to(b[3], got[:, 3].shape))  # the dead row
    got3, ref3 = run_both(bn, data, batch=3)
    assert_close(got3, ref3, f"quarter-folded conv1d n_fft={n_fft} {which} batch 3", atol=tol, rtol=0)
    assert np.array_equal(got3[:2].view(np.uint32), got.view(np.uint32))  # a segment's bits do not depend on the batch
    # the half-height kernel (filter fragments packed by the planner, straight from global memory) and the 64-row kernel: same products, same order
    packed = "kernel=frame_fold2p" in text
    monkeypatch.setenv("BN_FRAME2_WPK", "0")
    assert "kernel=frame_fold2p" not in bn.plan_describe(write_model(data))
    tall, _ = run_both(bn, data, batch=2)
    assert np.array_equal(tall.view(np.uint32), got.view(np.uint32)), packed
    monkeypatch.delenv("BN_FRAME2_WPK")
    monkeypatch.setenv("BN_CONVFOLD2", "0")
    text = bn.plan_describe(write_model(data))
    assert "~sym" in text and "~quarter" not in text, text
    halfold, _ = run_both(bn, data, batch=2)
    assert_close(got, halfold, "quarter fold vs half fold", atol=tol, rtol=0)
    monkeypatch.setenv("BN_CONVFOLD", "0")
    assert "~" not in bn.plan_describe(write_model(data))
    plain, _ = run_both(bn, data, batch=2)
    assert_close(got, plain, "quarter fold vs plain convolution", atol=tol, rtol=0)


@pytest.mark.parametrize("start,n_fft,hop,bins", [(3, 1024, 37, 90), (1, 2048, 278, 127), (2, 512, 101, 120)])
def test_quarter_and_half_fold_on_an_unaligned_signal_with_the_chain_in_the_span_load(bn, start, n_fft, hop, bins, monkeypatch):
    """The framing kernels' scalar span load: the signal is a view that starts 1 - 3 samples into the segment (no 16-byte alignment, the
    float4 path is off) and is normalised by the model's per-sample chain ((x - min) / (max - min + eps) - 0.5) * 2 that the planner moves into the span load
    (rule G) -- cosine-only bank (quarter fold, both block heights) and cos | sin bank (half folds) against the oracle."""
    import importlib
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    w = synth.dft_basis(n_fft, "complex")
    half = w.shape[0] // 2
    S = 100000
    frames = (S - n_fft) // hop + 1
    for kind in ("cos", "both"):
        ww = np.ascontiguousarray(w[2:2 + bins] if kind == "cos" else np.concatenate([w[2:2 + bins // 2], w[half + 2:half + 2 + bins // 2]], axis=0))

        def build(g, x):
            i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
            x = g.node("Slice", [x, i64(start), i64(start + S), i64(1), i64(1)])
            mn = g.node("ReduceMin", [x], axes=[1], keepdims=1)
            x1 = g.node("Sub", [x, mn])
            mx = g.node("ReduceMax", [x1], axes=[1], keepdims=1)
            x2 = g.node("Div", [x1, g.node("Add", [mx, g.const(np.float32(1e-6))])])
            # (centred like the model's chain: a segment with a large mean makes every low bin a sum of large cancelling terms whose f32
            # rounding -- in ANY order -- exceeds a tolerance scaled by the outputs)
            x3 = g.node("Mul", [g.node("Sub", [x2, g.const(np.float32(0.5))]), g.const(np.float32(2.0))])
            u = g.node("Unsqueeze", [x3, i64(1)])
            return g.node("Conv", [u, g.const(ww)], kernel_shape=[n_fft], strides=[hop])
        data = op_graph(build, [ww.shape[0], frames])
        monkeypatch.setenv("BN_STFT", "0")
        text = bn.plan_describe(write_model(data))
        assert " pre=" in text and ("~quarter" in text) == (kind == "cos"), text
        got, ref = run_both(bn, data, batch=2)
        tol = 2e-5 * float(np.abs(ref).max())
        assert_close(got, ref, f"unaligned {kind} n_fft={n_fft}", atol=tol, rtol=0)
        monkeypatch.setenv("BN_FRAME_PRE", "0")
        assert " pre=" not in bn.plan_describe(write_model(data))
        sep, _ = run_both(bn, data, batch=2)
        assert_close(got, sep, "chain in the span load vs its own launch", atol=tol, rtol=0)
        monkeypatch.delenv("BN_FRAME_PRE")
        if kind == "cos":
            monkeypatch.setenv("BN_FRAME2_WPK", "0")
            tall, _ = run_both(bn, data, batch=2)
            assert np.array_equal(tall.view(np.uint32), got.view(np.uint32))
            monkeypatch.delenv("BN_FRAME2_WPK")


@pytest.mark.parametrize("n_fft,hop,n_mels,fmin,fmax,bias,expect", [(1024, 280, 96, 500.0, 15000.0, False, True), (1024, 280, 40, 0.0, 24000.0, True, True),
                                                                    (2048, 278, 96, 0.0, 3000.0, False, False), (512, 160, 64, 1000.0, 20000.0, True, True)])
def test_framing_conv_merged_with_the_product_behind_it(bn, n_fft, hop, n_mels, fmin, fmax, bias, expect, monkeypatch):
    """Conv1D (real-part DFT bank) -> Transpose -> MatMul (mel bank) with nothing between them is one linear map of the frame: the planner
    (merge_framing_products) sums the filters -- n_mels symmetric rows, folded to half their taps, with the compression chain and the
    consumer's view in the launch's epilogue -- where that costs less than the bank plus the product (v2.4's 1024-point branch: yes; its
    127-bin 2048-point branch, which the quarter fold serves: no).  Against the oracle and against the unmerged plan."""
    import importlib
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    rng = np.random.default_rng(n_fft + n_mels)
    w = synth.dft_basis(n_fft, "real")
    mel = synth.mel_filterbank(n_fft // 2 + 1, n_mels, 48000, fmin, fmax).astype(np.float32)
    b = (rng.standard_normal(w.shape[0]) * 0.1).astype(np.float32)
    frames = (144000 - n_fft) // hop + 1

    def build(g, x):
        u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
        c = g.node("Conv", [u, g.const(w)] + ([g.const(b)] if bias else []), kernel_shape=[n_fft], strides=[hop])
        t = g.node("Transpose", [c], perm=[0, 2, 1])
        m = g.node("MatMul", [t, g.const(mel)])
        p = g.node("Pow", [m, g.const(np.float32(2.0))])
        return g.node("Pow", [p, g.const(np.float32(0.5))])  # (|m|: a two-stage chain whose slope stays bounded where m crosses zero)
    data = op_graph(build, [frames, n_mels])
    text = bn.plan_describe(write_model(data))
    merged = "MatMul" not in text
    assert merged == expect, text
    if merged:
        assert "~sym" in text and f"N={n_mels} " in text and f"K={n_fft // 2} " in text and "kernel=frame_fold" in text and "post=" in text, text
        assert text.count("\n") <= 5, text  # one launch (+ totals / outputs lines): the chain rides in its epilogue
    got, ref = run_both(bn, data, batch=2)
    tol = 2e-5 * float(np.abs(ref).max())
    assert_close(got, ref, f"merged framing conv n_fft={n_fft}", atol=tol, rtol=0)
    got3, ref3 = run_both(bn, data, batch=3)
    assert_close(got3, ref3, f"merged framing conv n_fft={n_fft} batch 3", atol=tol, rtol=0)
    assert np.array_equal(got3[:2].view(np.uint32), got.view(np.uint32))
    monkeypatch.setenv("BN_CONVMERGE", "1" if not merged else "0")  # the other form
    text2 = bn.plan_describe(write_model(data))
    assert ("MatMul" not in text2) == (not merged), text2
    other, _ = run_both(bn, data, batch=2)
    assert_close(got, other, "merged vs separate", atol=tol, rtol=0)


@pytest.mark.parametrize("n_fft,hop", [(128, 64), (256, 100), (512, 160), (1024, 320), (2048, 278), (640, 320), (640, 203)])
def test_stft_every_transform_size(bn, n_fft, hop, monkeypatch):
    """All supported frame lengths (one radix-2 pass first when log2 of the half length is odd; radix 5 first for Perch's
    L = 640 = 2 x 5 x 64, three frames per wave pass and tiles of 24 / 21 frames), a non-Hann window, rows in scrambled
    bin order with per-row gains and a bias."""
    rng = np.random.default_rng(n_fft)
    n = np.arange(n_fft, dtype=np.float64)
    win = 0.54 - 0.46 * np.cos(2.0 * np.pi * n / n_fft)  # periodic Hamming: w[0] != 0 ...
    win = win * np.sin(np.pi * n / n_fft) ** 2             # ... so taper it to the w[0] == 0 the folding rule asks for
    bins = rng.permutation(n_fft // 2 + 1)[:24]
    rows = []
    for i, k in enumerate(bins):
        ang = 2.0 * np.pi * k * n / n_fft
        rows.append(win * (np.cos(ang) if i % 3 else -np.sin(ang)) * rng.uniform(0.3, 3.0))
    w = np.ascontiguousarray(np.array(rows, dtype=np.float32)[:, None, :])
    b = rng.standard_normal(w.shape[0]).astype(np.float32)
    frames = (144000 - n_fft) // hop + 1

    def build(g, x):
        u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
        return g.node("Conv", [u, g.const(w), g.const(b)], kernel_shape=[n_fft], strides=[hop])
    data = op_graph(build, [w.shape[0], frames])
    monkeypatch.setenv("BN_STFT", "1")
    text = bn.plan_describe(write_model(data))
    assert text.count(" FFT ") == 1, text
    got, ref = run_both(bn, data, batch=2)
    assert_close(got, ref, f"stft n_fft={n_fft}", atol=2e-5 * float(np.abs(ref).max()), rtol=0)


@pytest.mark.parametrize("n_mels,bias", [(40, True), (96, False), (150, True)])
def test_stft_with_absorbed_mel_bank_sizes(bn, n_mels, bias, monkeypatch):
    """The mel filter bank inside the FFT launch on the matrix cores (16 x 16 tiles of the banded matrix) and as the sparse walk on the
    vector ALU, for band counts that leave a ragged last tile (40 = 2.5 tiles) and that give a wave more than one tile (150 bands = 10
    tiles over 8 waves), with a bias and a compression chain behind it -- both against the oracle and against each other."""
    import importlib
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    n_fft, hop, sr = 1024, 280, 48000
    frames = (144000 - n_fft) // hop + 1
    rng = np.random.default_rng(n_mels)
    mel = synth.mel_filterbank(n_fft // 2 + 1, n_mels, sr, 500.0, 15000.0).astype(np.float32)
    b = rng.standard_normal(n_mels).astype(np.float32)

    def build(g, x):
        u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
        c = g.node("Conv", [u, g.const(synth.dft_basis(n_fft, "real"))], kernel_shape=[n_fft], strides=[hop])
        t = g.node("Transpose", [c], perm=[0, 2, 1])
        m = g.node("MatMul", [t, g.const(mel)])
        if bias:
            m = g.node("Add", [m, g.const(b)])
        # (a well-conditioned chain: the power-law compression of the models amplifies the f32 noise of near-zero mel values a
        # thousandfold, which says nothing about the kernel; it is covered at model level)
        p = g.node("Add", [g.node("Mul", [g.node("Relu", [m]), g.const(np.float32(0.5))]), g.const(np.float32(0.25))])
        return g.node("Transpose", [p], perm=[0, 2, 1])
    data = op_graph(build, [n_mels, frames])
    monkeypatch.setenv("BN_STFT", "1")
    desc = bn.plan_describe(write_model(data))
    line = [l for l in desc.splitlines() if " FFT " in l]
    assert len(line) == 1 and f"mel={n_mels}(mfma)" in line[0] and "MatMul" in line[0], desc
    got, ref = run_both(bn, data, batch=3)
    tol = 2e-5 * float(np.abs(ref).max())
    assert_close(got, ref, f"stft + mel {n_mels} on the matrix cores", atol=tol, rtol=2e-4)
    monkeypatch.setenv("BN_STFT_MELMFMA", "0")
    assert f"mel={n_mels}(csr)" in bn.plan_describe(write_model(data))
    walk, _ = run_both(bn, data, batch=3)
    assert_close(walk, ref, f"stft + mel {n_mels} sparse walk", atol=tol, rtol=2e-4)
    assert_close(got, walk, "matrix cores vs sparse walk", atol=tol, rtol=2e-4)


@pytest.mark.parametrize("n_fft,hop,root", [(1024, 320, True), (512, 160, False), (256, 100, True), (640, 320, False), (640, 320, True)])
def test_stft_power_spectrum_folded_into_the_launch(bn, n_fft, hop, root, monkeypatch):
    """re^2 + im^2 (-> sqrt) behind a cos | sin bank: the planner folds it into the FFT launch (FftDesc::power 1 / 2), which then
    writes one value per bin; with the rule off the elementwise launch comes back.  Both against the oracle."""
    import importlib
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    nb = n_fft // 2 + 1
    frames = (144000 - n_fft) // hop + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        u = g.node("Unsqueeze", [x, i64(1)])
        c = g.node("Conv", [u, g.const(synth.dft_basis(n_fft, "complex"))], kernel_shape=[n_fft], strides=[hop])  # [B, 2 nb, frames]
        t = g.node("Transpose", [c], perm=[0, 2, 1])
        re = g.node("Slice", [t, i64(0), i64(nb), i64(2), i64(1)])
        im = g.node("Slice", [t, i64(nb), i64(2 * nb), i64(2), i64(1)])
        p = g.node("Add", [g.node("Mul", [re, re]), g.node("Mul", [im, im])])
        return g.node("Sqrt", [p]) if root else p
    data = op_graph(build, [frames, nb])
    monkeypatch.setenv("BN_STFT", "1")
    desc = bn.plan_describe(write_model(data))
    line = [l for l in desc.splitlines() if " FFT " in l]
    assert len(line) == 1 and f"power={2 if root else 1}" in line[0] and f"bins={nb} " in line[0] and " ELT " not in desc, desc
    got, ref = run_both(bn, data, batch=2)
    tol = 3e-5 * float(np.abs(ref).max())
    assert_close(got, ref, f"stft power n_fft={n_fft} root={root}", atol=tol, rtol=2e-4)
    monkeypatch.setenv("BN_STFT_POWER", "0")
    desc0 = bn.plan_describe(write_model(data))
    assert "power=0" in desc0 and " ELT " in desc0
    got0, _ = run_both(bn, data, batch=2)
    assert_close(got0, ref, f"stft + elementwise power n_fft={n_fft}", atol=tol, rtol=2e-4)


def test_conv1d_not_folded_when_not_symmetric(bn):
    """A filter bank that is symmetric except for one tap, or whose tap 0 is not zero, keeps the full-length GEMM."""
    import importlib
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    for breaker in ("tap", "w0"):
        w = synth.dft_basis(512, "real")[:9].copy()
        if breaker == "tap":
            w[4, 0, 100] += 1e-3
        else:
            w[:, 0, 0] = 0.01
        frames = (144000 - 512) // 256 + 1

        def build(g, x):
            u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
            return g.node("Conv", [u, g.const(w)], kernel_shape=[512], strides=[256])
        data = op_graph(build, [9, frames])
        assert "~" not in bn.plan_describe(write_model(data))
        got, ref = run_both(bn, data, batch=2)
        assert_close(got, ref, f"unfolded conv1d ({breaker})")


@pytest.mark.parametrize("length", [144000, 65536, 30000])
def test_whole_segment_min_max_bit_exact(bn, length, monkeypatch):
    """Min / max over a whole segment: long ranges are cut into chunks reduced by several blocks per sample plus a
    tiny second launch -- same bits as numpy (min / max do not depend on evaluation order) and as the one-pass plan."""
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        if length != 144000:
            x = g.node("Slice", [x, i64(0), i64(length), i64(1), i64(1)])
        mn = g.node("ReduceMin", [x], axes=[1], keepdims=1)
        mx = g.node("ReduceMax", [g.node("Sub", [x, mn])], axes=[1], keepdims=1)
        return g.node("Concat", [mn, mx], axis=1)
    data = op_graph(build, [2])
    text = bn.plan_describe(write_model(data))
    assert ("/chunks" in text) == (length >= 65536), text
    rng = np.random.default_rng(11)
    x = rng.standard_normal((5, 144000)).astype(np.float32)
    x[1, 143999] = -9.0
    x[2, 0] = 11.0
    x[3, length - 1] = 7.5
    path = write_model(data)
    got, _ = bn.Context(bn.Model(path), 8).infer(x)
    xs = x[:, :length]
    mn = xs.min(axis=1, keepdims=True)
    want = np.concatenate([mn, (xs - mn).max(axis=1, keepdims=True)], axis=1)
    assert np.array_equal(got.reshape(5, 2).view(np.uint32), want.view(np.uint32))
    monkeypatch.setenv("BN_REDUCE_SPLIT", "0")
    assert "/chunks" not in bn.plan_describe(path)
    one_pass, _ = bn.Context(bn.Model(path), 8).infer(x)
    assert np.array_equal(one_pass.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("variant", ["pow-flip-transpose", "affine-transpose", "exp-dense", "interleave"])
def test_matmul_with_absorbed_elementwise_chain(bn, variant, monkeypatch):
    """A chain of unary stages after a MatMul (and the layout copy behind it) runs in the GEMM epilogue: stages on
    the accumulators, store through the chain's output view.  Checked against the oracle and the unfused plan."""
    rng = np.random.default_rng(12)
    rows, k, n = 300, 77, 96
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(rows * k), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, rows, k)])
        y = g.node("MatMul", [x, g.const(w)])                                   # [B, rows, n]
        if variant == "pow-flip-transpose":
            y = g.node("Pow", [g.node("Pow", [y, g.const(np.float32(2.0))]), g.const(np.float32(0.3))])
            y = g.node("Slice", [y, i64(-1), i64(-(2 ** 62)), i64(2), i64(-1)])  # reverse the n axis
            return g.node("Transpose", [y], perm=[0, 2, 1])                      # [B, n, rows]
        if variant == "affine-transpose":
            y = g.node("Add", [g.node("Mul", [y, g.const(np.float32(1.5))]), g.const(np.float32(-0.25))])
            return g.node("Transpose", [g.node("Relu", [y])], perm=[0, 2, 1])
        if variant == "exp-dense":
            return g.node("Exp", [g.node("Mul", [y, g.const(np.float32(0.5))])])
        # two MatMul branches written into alternating channels of one image (the v2.4 spectrogram layout)
        y2 = g.node("Abs", [g.node("MatMul", [x, g.const(np.ascontiguousarray(w[:, ::-1]))])])
        a = g.node("Unsqueeze", [g.node("Transpose", [g.node("Sqrt", [g.node("Abs", [y])])], perm=[0, 2, 1]), i64(1)])
        b = g.node("Unsqueeze", [g.node("Transpose", [y2], perm=[0, 2, 1]), i64(1)])
        img = g.node("Concat", [a, b], axis=1)                                   # [B, 2, n, rows]
        # per-plane normalisation: rides on each plane's copy, i.e. ends up in the two GEMM epilogues as well
        img = g.node("BatchNormalization", [img, g.const(np.array([0.7, 1.3], np.float32)), g.const(np.array([0.1, -0.2], np.float32)),
                                            g.const(np.array([0.3, 0.5], np.float32)), g.const(np.array([1.5, 0.8], np.float32))], epsilon=1e-3)
        wc = (rng.standard_normal((4, 2, 3, 3)) / 4).astype(np.float32)
        return g.node("Conv", [img, g.const(wc)], kernel_shape=[3, 3], pads=[1, 1, 1, 1])
    shape = {"pow-flip-transpose": [n, rows], "affine-transpose": [n, rows], "exp-dense": [rows, n], "interleave": [4, n, rows]}[variant]
    data = op_graph(build, shape)
    text = bn.plan_describe(write_model(data))
    assert text.count(" post=") == (2 if variant == "interleave" else 1), text
    # nothing elementwise is left but the NHWC -> NCHW copy of the convolution result in the last variant
    assert sum(l.split()[1] == "ELT" for l in text.splitlines() if l[:3].strip().isdigit()) == (1 if variant == "interleave" else 0), text
    got, ref = run_both(bn, data, batch=3)
    assert_close(got, ref, f"absorbed chain {variant}")
    monkeypatch.setenv("BN_GEMMPOST", "0")
    assert " post=" not in bn.plan_describe(write_model(data))
    plain, _ = run_both(bn, data, batch=3)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32)), "same arithmetic either way"


def test_minmax_normalisation_reads_the_signal_once_less(bn, monkeypatch):
    """max(x - min) is taken as max(x) - min (exact: rounding is monotone), so x - min has one consumer left and
    fuses into the scaling chain; same bits as the plan without the rewrite."""
    def build(g, x):
        mn = g.node("ReduceMin", [x], axes=[1], keepdims=1)
        x1 = g.node("Sub", [x, mn])
        mx = g.node("ReduceMax", [x1], axes=[1], keepdims=1)
        return g.node("Div", [x1, g.node("Add", [mx, g.const(np.float32(1e-6))])])
    data = op_graph(build, [144000])
    path = write_model(data)
    text = bn.plan_describe(path)
    assert "/unshifted" in text and sum(l.split()[1] == "ELT" and "n=144000" in l for l in text.splitlines() if l[:3].strip().isdigit()) == 1, text
    got, ref = run_both(bn, data, batch=3, scale=0.3)
    assert_close(got, ref, "min-max normalisation", atol=1e-6, rtol=1e-6)
    # the chunk stages of the min and of the (now unshifted) max read the segment in ONE paired launch
    assert "/chunks+ReduceMax" in text, text
    monkeypatch.setenv("BN_REDUCE_PAIR", "0")
    assert "/chunks+ReduceMax" not in bn.plan_describe(path)
    unpaired, _ = run_both(bn, data, batch=3, scale=0.3)
    assert np.array_equal(got.view(np.uint32), unpaired.view(np.uint32))
    monkeypatch.delenv("BN_REDUCE_PAIR")
    monkeypatch.setenv("BN_REDUCE_SHIFT", "0")
    assert "/unshifted" not in bn.plan_describe(path)
    plain, _ = run_both(bn, data, batch=3, scale=0.3)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32))


def test_conv1d_with_padding(bn):
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((9, 1, 640)) / 25.0).astype(np.float32)

    def build(g, x):
        u = g.node("Unsqueeze", [x, g.const(np.array([1], dtype=np.int64))])
        return g.node("Conv", [u, g.const(w)], kernel_shape=[640], strides=[320], pads=[160, 160])
    got, ref = run_both(bn, op_graph(build, [9, 450]), batch=2)
    assert_close(got, ref, "padded conv1d")


@pytest.mark.parametrize("op,axis_last", [("Softmax", True), ("LogSoftmax", True), ("Softmax", False)])
def test_softmax(bn, op, axis_last):
    """Softmax / LogSoftmax along a non-batch axis (expanded into reduce + elementwise launches)."""
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        y = g.node("Mul", [x, g.const(np.float32(4.0))])
        if axis_last:
            return g.node(op, [y], axis=-1)
        return g.node(op, [y], axis=1)
    shape = [600, 240] if axis_last else [600, 240]
    got, ref = run_both(bn, op_graph(build, shape, in_reshape=shape), batch=3)
    assert_close(got, ref, f"{op} axis_last={axis_last}", atol=1e-5, rtol=2e-4)
    if op == "Softmax":
        assert np.allclose(ref.sum(axis=-1 if axis_last else 1), 1.0, atol=1e-4)


def test_split_views(bn):
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        a, b_, c = g.node("Split", [x, i64(5, 7, 12)], n_out=3, axis=1)
        y = g.node("Concat", [c, g.node("Relu", [a]), g.node("Neg", [b_])], axis=1)
        e, f = g.node("Split", [y], n_out=2, axis=2)        # equal halves
        return g.node("Sub", [e, f])
    got, ref = run_both(bn, op_graph(build, [24, 3000], in_reshape=[24, 6000]), batch=2)
    assert_close(got, ref, "split")


@pytest.mark.parametrize("op,c,h,w,k,stride,pad,cip", [("MaxPool", 32, 24, 40, 3, 2, 1, 0), ("MaxPool", 6, 17, 23, 2, 2, 0, 0),
                                                      ("AveragePool", 32, 24, 40, 3, 2, 1, 0), ("AveragePool", 16, 15, 31, 3, 1, 1, 1),
                                                      ("AveragePool", 5, 9, 9, 5, 3, 2, 0)])
def test_pooling(bn, op, c, h, w, k, stride, pad, cip):
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(c * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, c, h, w)])
        kw = dict(kernel_shape=[k, k], strides=[stride, stride], pads=[pad] * 4)
        if op == "AveragePool" and cip:
            kw["count_include_pad"] = 1
        return g.node(op, [x], **kw)
    got, ref = run_both(bn, op_graph(build, [c, oh, ow]), batch=2)
    assert_close(got, ref, f"{op} {c}x{h}x{w} k{k} s{stride} p{pad}")


def test_pooling_1d(bn):
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [x, i64(-1, 12, 12000)])
        return g.node("MaxPool", [g.node("AveragePool", [x], kernel_shape=[4], strides=[4])], kernel_shape=[3], strides=[2], pads=[1, 1])
    got, ref = run_both(bn, op_graph(build, [12, 1500]), batch=2)
    assert_close(got, ref, "pool1d")


@pytest.mark.parametrize("style", ["input", "attribute", "axes"])
def test_pad_constant(bn, style):
    """ONNX Pad (constant mode) on a [B, 6, 40, 600] view: spatial and channel pads, non-zero fill."""
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        if style == "attribute":
            y = g.node("Pad", [x], pads=[0, 1, 2, 3, 0, 0, 1, 5], value=-1.5)
        elif style == "input":
            y = g.node("Pad", [x, i64(0, 1, 2, 3, 0, 0, 1, 5), g.const(np.array(-1.5, dtype=np.float32))])
        else:
            y = g.node("Pad", [x, i64(2, 3, 1, 5), g.const(np.array(-1.5, dtype=np.float32)), i64(2, -1)])
        return g.node("Relu", [g.node("Mul", [y, g.const(np.array(-1.0, dtype=np.float32))])])
    c = 7 if style != "axes" else 6
    got, ref = run_both(bn, op_graph(build, [c, 43, 608], in_reshape=[6, 40, 600]), batch=3)
    assert_close(got, ref, f"pad {style}")
    assert np.count_nonzero(ref == 1.5) > 0  # the fill value is visible in the result


@pytest.mark.parametrize("k,n,style", [(1024, 6522, "gemm"), (96, 10, "matmul"), (130, 33, "matmul_bias"), (7, 5, "gemm_nt")])
def test_dense_head(bn, k, n, style):
    rng = np.random.default_rng(8)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(k), i64(1), i64(1)])
        if style == "gemm":
            return g.node("Gemm", [x, g.const(w), g.const(b)], transB=1)
        if style == "gemm_nt":
            return g.node("Gemm", [x, g.const(np.ascontiguousarray(w.T)), g.const(b)], alpha=0.5, beta=2.0)
        y = g.node("MatMul", [x, g.const(np.ascontiguousarray(w.T))])
        return g.node("Add", [y, g.const(b)]) if style == "matmul_bias" else y
    got, ref = run_both(bn, op_graph(build, [n]), batch=5)
    assert_close(got, ref, f"dense {k}->{n} {style}")


def test_reductions_and_broadcast(bn):
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [x, i64(-1, 30, 48, 100)])
        gap = g.node("GlobalAveragePool", [x])                          # [B,30,1,1]
        mx = g.node("ReduceMax", [x], axes=[1], keepdims=1)             # [B,1,48,100]
        mn = g.node("ReduceMin", [x], axes=[3], keepdims=1)             # [B,30,48,1]
        sm = g.node("ReduceSum", [x], axes=[2, 3], keepdims=1)          # [B,30,1,1]
        y = g.node("Add", [g.node("Mul", [x, gap]), mx])
        y = g.node("Sub", [y, mn])
        y = g.node("Div", [y, g.node("Add", [g.node("Abs", [sm]), g.const(np.float32(1.0))])])
        return g.node("ReduceMean", [y], axes=[2], keepdims=0)          # [B,30,100]
    got, ref = run_both(bn, op_graph(build, [30, 100]))
    assert_close(got, ref, "reductions", atol=1e-4, rtol=1e-4)


def test_views_concat_and_unary(bn):
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [x, i64(-1, 4, 60, 600)])
        t = g.node("Transpose", [x], perm=[0, 2, 3, 1])                                   # [B,60,600,4]
        rev = g.node("Slice", [t, i64(-1), i64(-(2 ** 62)), i64(2), i64(-1)])             # reverse axis 2
        ev = g.node("Slice", [rev, i64(0), i64(600), i64(2), i64(2)])                     # every 2nd -> 300
        a = g.node("Sigmoid", [ev])
        b = g.node("Tanh", [g.node("Slice", [t, i64(10), i64(310), i64(2), i64(1)])])     # [B,60,300,4]
        c = g.node("Concat", [a, b, g.node("Exp", [g.node("Neg", [g.node("Abs", [a])])])], axis=3)  # [B,60,300,12]
        c = g.node("Pow", [g.node("Abs", [c]), g.const(np.float32(0.3))])
        c = g.node("Log", [g.node("Add", [c, g.const(np.float32(1e-3))])])
        c = g.node("Max", [c, g.const(np.float32(-2.0))])
        c = g.node("LeakyRelu", [c], alpha=0.1)
        c = g.node("Transpose", [c], perm=[0, 3, 1, 2])                                   # [B,12,60,300]
        return g.node("Flatten", [g.node("ReduceMean", [c], axes=[3], keepdims=1)], axis=1)  # [B,720]
    got, ref = run_both(bn, op_graph(build, [720]))
    assert_close(got, ref, "views/concat/unary", atol=1e-4, rtol=1e-4)


def test_standalone_batchnorm_and_hard_activations(bn):
    rng = np.random.default_rng(9)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [x, i64(-1, 6, 40, 600)])
        y = g.node("BatchNormalization", [x, g.const(rng.uniform(0.5, 1.5, 6).astype(np.float32)),
                                          g.const(rng.standard_normal(6).astype(np.float32)),
                                          g.const(rng.standard_normal(6).astype(np.float32)),
                                          g.const(rng.uniform(0.5, 1.5, 6).astype(np.float32))])
        y = g.node("HardSwish", [y])
        y = g.node("HardSigmoid", [y], alpha=0.25, beta=0.4)
        y = g.node("Mul", [y, g.const(rng.standard_normal((1, 6, 1, 1)).astype(np.float32))])
        y = g.node("Add", [y, g.const(rng.standard_normal((600,)).astype(np.float32))])
        return g.node("ReduceMax", [y], axes=[2], keepdims=0)
    got, ref = run_both(bn, op_graph(build, [6, 600]))
    assert_close(got, ref, "bn/hard activations", atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("cin,h,w,cmid,k,stride,act", [(16, 24, 40, 96, 3, 2, "relu"), (24, 12, 40, 144, 3, 1, "relu"),
                                                       (24, 13, 37, 144, 5, 2, "silu"), (40, 9, 19, 240, 3, 2, "relu6"),
                                                       (8, 16, 32, 48, 5, 1, "relu"), (16, 7, 9, 40, 3, 1, None)])
@pytest.mark.parametrize("variant", ["tiled", "pipe", "map", "row"])
def test_fused_expand_depthwise(bn, cin, h, w, cmid, k, stride, act, variant):
    """expand 1x1 conv (+BN+act) -> depthwise KxK (+act): one MBCONV launch, the expanded tensor never in HBM.
    Kernels: output tiles with halo recompute in LDS ("tiled", its 512-thread "pipe" form), whole small maps per block ("map"),
    and the register-resident row-streaming form ("row", the default where it applies) -- which must reproduce the tiled
    kernel's output BIT FOR BIT (same accumulation order)."""
    rng = np.random.default_rng(11)
    assert cin * h * w <= 144000
    pad = k // 2
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1

    def activation(g, y):
        if act == "relu":
            return g.node("Relu", [y])
        if act == "silu":
            return g.node("Mul", [y, g.node("Sigmoid", [y])])
        if act == "relu6":
            return g.node("Clip", [y, g.const(np.float32(0)), g.const(np.float32(6))])
        return y

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        w0 = (rng.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        x = g.node("Conv", [x, g.const(w0)], kernel_shape=[1, 1])               # channels-last producer
        we = (rng.standard_normal((cmid, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        y = g.node("Conv", [x, g.const(we), g.const(rng.standard_normal(cmid).astype(np.float32))], kernel_shape=[1, 1])
        y = activation(g, y)
        wd = (rng.standard_normal((cmid, 1, k, k)) / k).astype(np.float32)
        z = g.node("Conv", [y, g.const(wd), g.const(rng.standard_normal(cmid).astype(np.float32))], kernel_shape=[k, k],
                   strides=[stride, stride], pads=[pad] * 4, group=cmid)
        return activation(g, z)
    data = op_graph(build, [cmid, oh, ow])
    import os
    os.environ["BN_MBFUSE"] = "force"   # fuse tiny feature maps with the tiled kernel too (read by the planner at model load)
    os.environ["BN_MBMAP"] = "1" if variant == "map" else "0"
    os.environ["BN_MBMAP_MAXHW"] = "1024"
    os.environ["BN_MBPIPE"] = "1" if variant == "pipe" else "0"   # pipelined 512-thread variant of the tiled kernel
    os.environ["BN_MBROW"] = "force" if variant == "row" else "0"   # force: narrow maps too (the planner leaves those to the tiled kernel)
    os.environ["BN_MBROW_TOH"] = "5"                              # several bands, a ragged last one
    os.environ["BN_MBROW_TR"] = "0"                               # row streaming along the map's rows (the bit-identical form)
    try:
        desc = bn.plan_describe(write_model(data))
        assert "MBCONV" in desc
        mb_rows = int([l for l in desc.splitlines() if " MBCONV " in l][0].rsplit("rows=", 1)[1].split("(")[0])  # ("(columns)": transposed streaming)
        if variant == "row":
            assert mb_rows == (min(5, oh) if cin >= 12 else 0), desc  # Cin <= 8 (one K group) stays with the tiled kernel
        else:
            assert mb_rows == 0, desc
            assert ("tiles=1x1" in desc) == (variant == "map" or (oh <= (8 if stride == 1 else 4) and ow <= (16 if stride == 1 else 8)))
        got, ref = run_both(bn, data)
        if variant == "row":
            # (round 5) the default row kernel runs its expand on the bf16 matrix pipe where Cin % 8 == 0 (bf16x3: other bits, the oracle's
            # tolerance -- asserted at the end); its exact-f32 form (BN_MBROW_B3=0) is the one that repeats the tiled kernel's bits
            os.environ["BN_MBROW_B3"] = "0"
            f32row, _ = run_both(bn, data)
            os.environ["BN_MBROW"] = "0"
            tiled, _ = run_both(bn, data)
            del os.environ["BN_MBROW_B3"]
            os.environ["BN_MBROW"] = "force"
            assert np.array_equal(f32row.view(np.uint32), tiled.view(np.uint32)), "row-streaming kernel differs from the tiled kernel"
            assert np.abs(got - f32row).max() <= 2e-5 * np.abs(ref).max()
            if k == 3:
                # ... and streaming along the map's COLUMNS (round 3: tall narrow maps; the depthwise taps then meet in (kx, ky)
                # order, so the bits may differ from the tiled kernel's: checked against the oracle)
                os.environ["BN_MBROW"], os.environ["BN_MBROW_TR"] = "force", "1"
                assert "(columns)" in bn.plan_describe(write_model(data))
                got_t, _ = run_both(bn, data)
                assert_close(got_t, ref, f"mbconv[row, transposed] {cin}->{cmid} k{k} s{stride}")
    finally:
        for key in ("BN_MBFUSE", "BN_MBMAP", "BN_MBMAP_MAXHW", "BN_MBPIPE", "BN_MBROW", "BN_MBROW_TOH", "BN_MBROW_TR", "BN_MBROW_B3"):
            os.environ.pop(key, None)
    assert_close(got, ref, f"mbconv[{variant}] {cin}->{cmid} k{k} s{stride}")


@pytest.mark.parametrize("cin,h,w,cout,k1,s1,k,stride,act", [(2, 96, 511, 32, 3, 2, 3, 1, "relu"), (1, 64, 129, 24, 5, 2, 3, 2, "relu6"),
                                                            (3, 70, 90, 40, 3, 1, 5, 1, "silu"), (2, 65, 131, 16, 3, 2, 5, 2, None)])
def test_fused_stem_conv_depthwise(bn, cin, h, w, cout, k1, s1, k, stride, act):
    """dense k1 x k1 conv with few input channels (+BN+act) -> depthwise KxK (+act): one launch (im2col rows in LDS),
    against the oracle; the unfused plan (BN_STEMFUSE=0) must agree too."""
    rng = np.random.default_rng(13)
    assert cin * h * w <= 144000
    p1, p2 = k1 // 2, k // 2
    h1, w1 = (h + 2 * p1 - k1) // s1 + 1, (w + 2 * p1 - k1) // s1 + 1
    oh, ow = (h1 + 2 * p2 - k) // stride + 1, (w1 + 2 * p2 - k) // stride + 1

    def activation(g, y):
        if act == "relu":
            return g.node("Relu", [y])
        if act == "silu":
            return g.node("Mul", [y, g.node("Sigmoid", [y])])
        if act == "relu6":
            return g.node("Clip", [y, g.const(np.float32(0)), g.const(np.float32(6))])
        return y

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, h, w, cin)])
        x = g.node("Transpose", [x], perm=[0, 3, 1, 2])                       # channels-last image, as the front end leaves it
        w0 = (rng.standard_normal((cout, cin, k1, k1)) / np.sqrt(cin * k1 * k1)).astype(np.float32)
        y = g.node("Conv", [x, g.const(w0), g.const(rng.standard_normal(cout).astype(np.float32))], kernel_shape=[k1, k1], strides=[s1, s1], pads=[p1] * 4)
        y = activation(g, y)
        wd = (rng.standard_normal((cout, 1, k, k)) / k).astype(np.float32)
        z = g.node("Conv", [y, g.const(wd), g.const(rng.standard_normal(cout).astype(np.float32))], kernel_shape=[k, k],
                   strides=[stride, stride], pads=[p2] * 4, group=cout)
        return activation(g, z)
    data = op_graph(build, [cout, oh, ow])
    import os
    os.environ["BN_MBFUSE"] = "force"   # fuse small feature maps too (the planner keeps those unfused by default)
    os.environ["BN_MBROW"] = "force"    # ... and take the row-streaming kernel whatever the strip utilisation
    try:
        desc = bn.plan_describe(write_model(data))
        assert "stem:" in desc, desc
        # 3x3 depthwise stems run in the row-streaming kernel (padding of the first conv included), the others tiled
        mb_rows = lambda t: int([l for l in t.splitlines() if " MBCONV " in l][0].rsplit("rows=", 1)[1])
        assert (mb_rows(desc) > 0) == (k == 3 and k1 <= 4 and k1 * k1 * cin > 8), desc  # mbconv_row_supported (kernels.h)
        got, ref = run_both(bn, data)
        os.environ["BN_MBROW"] = "0"
        assert mb_rows(bn.plan_describe(write_model(data))) == 0
        tiled, _ = run_both(bn, data)
        assert np.array_equal(got.view(np.uint32), tiled.view(np.uint32)), "row-streaming stem differs from the tiled kernel"
    finally:
        del os.environ["BN_MBFUSE"]
        os.environ.pop("BN_MBROW", None)
    assert_close(got, ref, f"stem {cin}->{cout} k1={k1} s1={s1} dw k{k} s{stride}")
    os.environ["BN_STEMFUSE"] = "0"
    try:
        assert "stem:" not in bn.plan_describe(write_model(data))
        got2, _ = run_both(bn, data)
    finally:
        del os.environ["BN_STEMFUSE"]
    assert_close(got2, ref, "unfused stem")


def test_pipelined_mbconv_is_bit_identical_to_the_plain_kernel(bn):
    """The planner / launcher may pick either MBConv variant per shape; results must not depend on that choice."""
    import os
    rng = np.random.default_rng(5)
    cin, h, w, cmid, k = 40, 12, 64, 240, 5

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        x = g.node("Conv", [x, g.const((rng.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))], kernel_shape=[1, 1])
        y = g.node("Relu", [g.node("Conv", [x, g.const((rng.standard_normal((cmid, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)),
                                            g.const(rng.standard_normal(cmid).astype(np.float32))], kernel_shape=[1, 1])])
        return g.node("Relu", [g.node("Conv", [y, g.const((rng.standard_normal((cmid, 1, k, k)) / k).astype(np.float32))], kernel_shape=[k, k],
                                      pads=[k // 2] * 4, group=cmid)])
    data = op_graph(build, [cmid, h, w])
    outs = []
    for mode, row in (("0", "0"), ("1", "0"), ("0", "force")):
        os.environ["BN_MBPIPE"] = mode
        os.environ["BN_MBROW"] = row
        os.environ["BN_MBFUSE"] = "force"
        os.environ["BN_MBROW_B3"] = "0"   # (the exact-f32 form of the row kernel: the one whose bits the tiled kernels repeat)
        try:
            assert "MBCONV" in bn.plan_describe(write_model(data))
            outs.append(run_both(bn, data, batch=3)[0].copy())
        finally:
            del os.environ["BN_MBPIPE"], os.environ["BN_MBFUSE"], os.environ["BN_MBROW"], os.environ["BN_MBROW_B3"]
    assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes()


# (last column: what the default plan runs the block on -- an mbmap.hip configuration (plan_rules.h, mbmap_shape) or None = GEMM + depthwise)
@pytest.mark.parametrize("cin,h,w,cmid,k,stride,expect", [
    (80, 6, 32, 480, 3, 1, "cfg1,ws "), (112, 6, 32, 672, 5, 2, "cfg2,ws "), (192, 3, 16, 1152, 5, 1, "cfg3,ws "),
    (192, 4, 16, 1152, 5, 1, "cfg4,ws "), (192, 4, 16, 1152, 3, 1, "cfg4,ws "), (192, 4, 16, 600, 5, 2, None),
    # round 5 -- the expand on the bf16 pipe: every compiled step count (Cin = 48 / 80: 1.5 / 2.5 steps, the half step zero-filled; 112: 3.5;
    # 128 / 256 in two K slices), ragged channel counts (a partial last chunk), Cin = 16 (half a step: stays on the exact-f32 form)
    (48, 6, 32, 288, 5, 1, "cfg1,ws "), (112, 6, 32, 672, 5, 1, "cfg2,ws "), (112, 6, 32, 600, 3, 1, "cfg2,ws "), (80, 6, 32, 40, 3, 2, "cfg1,ws "), (128, 3, 16, 776, 3, 2, "cfg3,ws "), (192, 3, 16, 200, 5, 2, "cfg3,ws "),
    (256, 3, 16, 1536, 5, 1, None), (128, 4, 16, 768, 5, 1, "cfg4,ws "), (256, 4, 16, 520, 3, 1, "cfg4,b3 "), (16, 6, 32, 96, 3, 1, "cfg1 "),
    (20, 5, 7, 72, 3, 1, None), (40, 12, 40, 100, 3, 2, None),
    # round 4 -- BirdNET v3.0's 8 x 32 stage in two bands (all four window / stride instances, both swizzle classes) ...
    # (round 5: not transposed, no padded k, 3 / 4 steps of 32 -> the wave-specialised kernel per band, taken by default)
    (80, 8, 32, 480, 3, 1, "cfg5,bands,ws "), (112, 8, 32, 672, 5, 1, "cfg5,bands,ws "), (112, 8, 32, 672, 5, 2, "cfg5,bands,ws "), (80, 8, 32, 252, 3, 2, "cfg5,bands,ws "),
    (48, 8, 32, 288, 3, 1, None),   # (1.5 steps: neither the banded ws kernel nor, by default, the exact-f32 banded form)
    # ... Perch's tall maps walked transposed, its Cin = 232 padded to 240 in LDS, the 64-pixel map with four waves ...
    (232, 16, 4, 1392, 5, 1, "cfg6,transposed,ws kpad=240"), (232, 16, 4, 700, 3, 1, "cfg6,transposed,ws kpad=240"), (48, 4, 16, 288, 5, 1, "cfg6 "),
    (96, 32, 8, 576, 3, 1, "cfg5,bands,transposed,ws kpad=96"), (96, 32, 8, 576, 5, 2, "cfg5,bands,transposed,ws kpad=96"),
    (96, 32, 8, 560, 5, 1, "cfg5,bands,transposed,ws kpad=96"),
    # ... Perch's K = 136 (rows padded to 144 in LDS, five steps of 32: 252 registers per expand wave)
    (136, 32, 8, 816, 5, 1, "cfg5,bands,transposed,ws kpad=144"), (136, 32, 8, 816, 5, 2, "cfg5,bands,transposed,ws kpad=144"), (136, 32, 8, 800, 3, 1, "cfg5,bands,transposed,ws kpad=144")])
def test_fused_expand_depthwise_small_maps(bn, cin, h, w, cmid, k, stride, expect):
    """The whole-map MBConv kernel at the late-stage shapes (K up to 192, ragged channel counts, K % 8 == 4)
    followed by a squeeze-excite that consumes its channel sums (complete, or one partial per band)."""
    rng = np.random.default_rng(12)
    pad = k // 2
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    cr = max(4, cmid // 24)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        w0 = (rng.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        x = g.node("Conv", [x, g.const(w0)], kernel_shape=[1, 1])
        we = (rng.standard_normal((cmid, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        y = g.node("Relu", [g.node("Conv", [x, g.const(we), g.const(rng.standard_normal(cmid).astype(np.float32))], kernel_shape=[1, 1])])
        wd = (rng.standard_normal((cmid, 1, k, k)) / k).astype(np.float32)
        z = g.node("Relu", [g.node("Conv", [y, g.const(wd), g.const(rng.standard_normal(cmid).astype(np.float32))], kernel_shape=[k, k],
                                   strides=[stride, stride], pads=[pad] * 4, group=cmid)])
        # squeeze-excite + projection
        s = g.node("GlobalAveragePool", [z])
        w1 = (rng.standard_normal((cr, cmid, 1, 1)) / np.sqrt(cmid)).astype(np.float32)
        w2 = (rng.standard_normal((cmid, cr, 1, 1)) / np.sqrt(cr)).astype(np.float32)
        e = g.node("Relu", [g.node("Conv", [s, g.const(w1), g.const(rng.standard_normal(cr).astype(np.float32))], kernel_shape=[1, 1])])
        e = g.node("Sigmoid", [g.node("Conv", [e, g.const(w2), g.const(rng.standard_normal(cmid).astype(np.float32))], kernel_shape=[1, 1])])
        zz = g.node("Mul", [z, e])
        wp = (rng.standard_normal((24, cmid, 1, 1)) / np.sqrt(cmid)).astype(np.float32)
        return g.node("Conv", [zz, g.const(wp)], kernel_shape=[1, 1])
    data = op_graph(build, [24, oh, ow])
    import os
    # (a) the round-1 whole-map kernel (opt-in BN_MBMAP=1; the LDS-resident form of round 3 switched off)
    os.environ["BN_MBMAP"], os.environ["BN_MBMAP2"] = "1", "0"
    try:
        desc = bn.plan_describe(write_model(data))
        assert "MBCONV" in desc and "tiles=1x1" in desc, desc
        got, ref = run_both(bn, data, batch=3)
    finally:
        del os.environ["BN_MBMAP"], os.environ["BN_MBMAP2"]
    assert_close(got, ref, f"mbconv map {cin}->{cmid} k{k} s{stride}")
    # (b) GEMM + whole-map depthwise + excite on the same graph
    os.environ["BN_MBMAP2"] = "0"
    try:
        assert "MBCONV" not in bn.plan_describe(write_model(data))
        got3, _ = run_both(bn, data, batch=3)
    finally:
        del os.environ["BN_MBMAP2"]
    assert_close(got3, ref, f"unfused {cin}->{cmid} k{k} s{stride}")
    # (c) the default plan: the LDS-resident whole-map kernel (mbmap.hip) wherever a configuration fits (192- and
    # 48-pixel maps with Cin % 16 == 0; BirdNET v3.0's 4 x 16 map), the unfused launches elsewhere
    round4 = bool(expect) and ",ws" not in expect and ("transposed" in expect or "cfg6" in expect or "bands" in expect)
    if round4:
        # the round-4 configurations (bands, transposed maps, padded k) are opt-in (measured slower where their models run saturated:
        # plan_rules.h); the default plan keeps GEMM + depthwise for these blocks
        assert "MBCONV" not in bn.plan_describe(write_model(data))
        os.environ["BN_MBMAP3"] = "1"
    try:
        desc = bn.plan_describe(write_model(data))
        assert ("MBCONV" in desc) == (expect is not None) and (expect is None or "map=" + expect in desc + " "), desc
        got2, _ = run_both(bn, data, batch=3)
        assert_close(got2, ref, f"default plan {cin}->{cmid} k{k} s{stride}")
        if expect and ("bands" in expect or "cfg6" in expect) and ",ws" in expect:
            # the banded wave-specialised form: same bytes whatever the batch (bands / chunks per block follow it), the switch of its own
            # gives the unfused plan, whose result it matches to a few roundings (another summation order in the expand, band-wise squeeze sums)
            one = run_both(bn, data, batch=1)[0][:1]
            many, _ = run_both(bn, data, batch=9)
            assert np.array_equal(one.view(np.uint32), many[:1].view(np.uint32)) and np.array_equal(got2.view(np.uint32), many[:3].view(np.uint32))
            sw = "BN_MBMAP_WS_DEEP" if "cfg6" in expect else "BN_MBMAP_WS_BANDS"
            os.environ[sw] = "0"
            try:
                assert "MBCONV" not in bn.plan_describe(write_model(data))
            finally:
                del os.environ[sw]
            assert np.abs(got2 - got3).max() <= 2e-5 * np.abs(got3).max()
        elif expect and (",b3" in expect or ",ws" in expect):
            # the exact-f32 expand of the same configuration (BN_MBMAP_B3=0), and the bits of the bf16x3 form do not depend on the batch a
            # segment rides in (chunks per block follow the batch)
            os.environ["BN_MBMAP_B3"] = "0"
            try:
                assert "map=" + expect.replace(",b3", "").replace(",ws", "") in bn.plan_describe(write_model(data)) + " "
                got4, _ = run_both(bn, data, batch=3)
            finally:
                del os.environ["BN_MBMAP_B3"]
            assert_close(got4, ref, f"exact-f32 expand {cin}->{cmid} k{k} s{stride}")
            if ",ws" in expect:
                # the wave-specialised kernel and mbmap.hip's bf16x3 form: the same expanded / depthwise bits; the squeeze sums add the column
                # strips' partials in 8 lane groups where mbmap.hip's 32-channel configuration has 16 (another association of the same terms)
                os.environ["BN_MBMAP_WS"] = "0"
                try:
                    assert "map=" + expect.replace(",ws", ",b3") in bn.plan_describe(write_model(data)) + " "
                    got5, _ = run_both(bn, data, batch=3)
                    if "cfg3" in expect or "cfg4" in expect:  # the small maps have a switch of their own
                        os.environ["BN_MBMAP_WS"], os.environ["BN_MBMAP_WS_SMALL"] = "1", "0"
                        assert "map=" + expect.replace(",ws", ",b3") in bn.plan_describe(write_model(data)) + " "
                finally:
                    del os.environ["BN_MBMAP_WS"]
                    os.environ.pop("BN_MBMAP_WS_SMALL", None)
                assert np.abs(got5 - got2).max() <= 2e-6 * np.abs(got2).max()
                if "cfg1" in expect:  # (64-channel chunks there: 8 lane groups as well)
                    assert np.array_equal(got5.view(np.uint32), got2.view(np.uint32))
            one = run_both(bn, data, batch=1)[0][:1]
            many, _ = run_both(bn, data, batch=9)
            assert np.array_equal(one.view(np.uint32), many[:1].view(np.uint32)) and np.array_equal(got2.view(np.uint32), many[:3].view(np.uint32))
        if round4:
            # batch composition: band / chunk grouping follows the batch, the bits must not
            one = np.concatenate([run_both(bn, data, batch=1)[0][:1]])
            many, _ = run_both(bn, data, batch=7)
            assert np.array_equal(one.view(np.uint32), many[:1].view(np.uint32)) and np.array_equal(got2.view(np.uint32), many[:3].view(np.uint32))
    finally:
        os.environ.pop("BN_MBMAP3", None)


@pytest.mark.parametrize("auto_pad,cin,cout,k,stride,groups", [("SAME_UPPER", 3, 16, 3, 2, 1), ("SAME_LOWER", 3, 16, 4, 2, 1), ("SAME_UPPER", 32, 32, 5, 2, 32),
                                                                ("SAME_LOWER", 32, 32, 3, 1, 32), ("VALID", 8, 24, 3, 2, 1), ("SAME_UPPER", 24, 40, 1, 1, 1)])
def test_conv_auto_pad(bn, auto_pad, cin, cout, k, stride, groups):
    """auto_pad (tf2onnx exports carry it instead of explicit pads): dense, depthwise and 1x1 convolutions against the
    oracle, whose auto_pad rule is itself checked against torch in tests/test_oracle_independent.py."""
    rng = np.random.default_rng(k + stride)
    h, w = 23, 30
    wgt = (rng.standard_normal((cout, cin // groups, k, k)) / np.sqrt(cin // groups * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    oh = (h - k) // stride + 1 if auto_pad == "VALID" else -(-h // stride)
    ow = (w - k) // stride + 1 if auto_pad == "VALID" else -(-w // stride)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, cin, h, w)])
        y = g.node("Conv", [x, g.const(wgt), g.const(b)], kernel_shape=[k, k], strides=[stride, stride], auto_pad=auto_pad, group=groups)
        return g.node("Relu", [y])
    got, ref = run_both(bn, op_graph(build, [cout, oh, ow]))
    assert_close(got, ref, f"conv auto_pad={auto_pad} k={k} s={stride} g={groups}")


# ---------------------------------------------------------------- exporter dialects (round 2)
def _hann(n, periodic=True):
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2 * np.pi * k / (n if periodic else n - 1))).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("nfft,hop,window,onesided,rank3", [
    (512, 128, "hann", 1, True),        # torch.stft-style export: [B, L, 1] signal, periodic Hann -> folded framing GEMMs
    (1024, 280, "hann_sym", 1, False),  # symmetric Hann: rows are not mirror images about N/2 -> plain framing GEMM
    (256, 64, None, 1, False),          # no window input: rectangular, frame_length given
    (128, 32, "hann", 0, True),         # two-sided
])
def test_stft_node_opset17(bn, nfft, hop, window, onesided, rank3, monkeypatch):
    """An opset-17 STFT node (what torch.onnx writes for torch.stft) is mapped to the framing kernels: magnitude and the
    raw [frames, bins, 2] tensor against the oracle, which evaluates the node with an FFT of the windowed frames."""
    L = 48000
    bins = nfft // 2 + 1 if onesided else nfft
    frames = (L - nfft) // hop + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(L), i64(1), i64(1)])
        if rank3:
            x = g.node("Unsqueeze", [x, i64(2)])
        ins = [x, g.const(np.array(hop, dtype=np.int64))]
        if window:
            ins.append(g.const(_hann(nfft, periodic=(window == "hann"))))
        else:
            ins += ["", g.const(np.array(nfft, dtype=np.int64))]
        s = g.node("STFT", ins, onesided=onesided)                     # [B, frames, bins, 2]
        re = g.node("Slice", [s, i64(0), i64(1), i64(3), i64(1)])
        im = g.node("Slice", [s, i64(1), i64(2), i64(3), i64(1)])
        p = g.node("Add", [g.node("Mul", [re, re]), g.node("Mul", [im, im])])
        mag = g.node("Sqrt", [g.node("Add", [p, g.const(np.array(1e-6, dtype=np.float32))])])
        return g.node("Concat", [s, mag], axis=3)                      # raw spectrum and magnitude side by side

    data = op_graph(build, [frames, bins, 3])
    # default plan: the planner picks FFT or matrix product per bank by estimated cost (every case here with a window
    # the FFT recognition accepts has enough bins for the FFT)
    got, ref = run_both(bn, data)
    # a bin is a sum of nfft products of O(1) terms: absolute error grows like sqrt(nfft) * 2^-24 * |frame|
    assert_close(got, ref, f"STFT n={nfft} hop={hop} window={window}", atol=2e-4 * np.sqrt(nfft / 256), rtol=2e-4)
    # the matrix-product path on its own
    monkeypatch.setenv("BN_STFT", "0")
    got, ref = run_both(bn, data)
    assert_close(got, ref, f"STFT as framing GEMMs n={nfft} hop={hop} window={window}", atol=2e-4 * np.sqrt(nfft / 256), rtol=2e-4)
    text = bn.plan_describe(write_model(data))
    assert " FFT " not in text
    assert ("fold=" in text) == (window == "hann"), text  # periodic Hann rows fold to half their taps, the others do not


@pytest.mark.gpu
def test_expand_and_nhwc_sandwich(bn):
    """Expand (broadcast against a constant shape, the way exporters spell tf.broadcast_to / x.expand) and an NHWC graph
    whose convolutions sit between Transpose pairs (tf2onnx): both against the oracle."""
    rng = np.random.default_rng(5)
    h, w, c, co = 12, 20, 8, 16
    wgt = (rng.standard_normal((co, c, 3, 3)) / np.sqrt(c * 9)).astype(np.float32)
    scale = rng.standard_normal((1, 1, 1, co)).astype(np.float32)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(h * w * c), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, h, w, c)])                          # NHWC activations
        y = g.node("Transpose", [x], perm=[0, 3, 1, 2])                       # -> NCHW for the Conv
        y = g.node("Conv", [y, g.const(wgt)], kernel_shape=[3, 3], pads=[1, 1, 1, 1])
        y = g.node("Transpose", [y], perm=[0, 2, 3, 1])                       # back to NHWC
        s = g.node("Expand", [g.const(scale), i64(1, h, w, co)])              # constant operand: folded at import
        y = g.node("Mul", [y, s])
        m = g.node("ReduceMean", [y], axes=[3], keepdims=1)                   # [B, h, w, 1]
        e = g.node("Expand", [m, i64(1, h, w, co)])                           # activation operand: runs on the device
        return g.node("Sub", [y, e])

    got, ref = run_both(bn, op_graph(build, [h, w, co]))
    assert_close(got, ref, "Expand + NHWC sandwich")


@pytest.mark.gpu
def test_stft_node_through_the_fft_kernel(bn, monkeypatch):
    """The same STFT node under BN_STFT=1: the synthesized cos | sin bank is recognised and runs as ONE FFT launch."""
    monkeypatch.setenv("BN_STFT", "1")
    nfft, hop, L = 1024, 256, 64000
    bins, frames = nfft // 2 + 1, (L - nfft) // hop + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(L), i64(1), i64(1)])
        return g.node("STFT", [x, g.const(np.array(hop, dtype=np.int64)), g.const(_hann(nfft))])

    data = op_graph(build, [frames, bins, 2])
    text = bn.plan_describe(write_model(data))
    assert " FFT " in text and " GEMM " not in text, text
    got, ref = run_both(bn, data)
    assert_close(got, ref, "STFT node as FFT", atol=4e-4, rtol=2e-4)


@pytest.mark.gpu
def test_prelu_tile_instance_norm(bn):
    """More exporter spellings (first-contact readiness): PRelu with per-channel and with scalar slopes, Tile of
    size-1 dimensions, InstanceNormalization -- each lowered to operators the planner already had, against the oracle
    (whose versions of these three are one torch call each)."""
    rng = np.random.default_rng(21)
    c, h, w = 12, 10, 25

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(c * h * w), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, c, h, w)])
        y = g.node("InstanceNormalization", [x, g.const(rng.uniform(0.5, 1.5, c).astype(np.float32)), g.const(rng.standard_normal(c).astype(np.float32))],
                   epsilon=1e-3)
        y = g.node("PRelu", [y, g.const(rng.uniform(0.05, 0.4, (c, 1, 1)).astype(np.float32))])
        y = g.node("PRelu", [y, g.const(np.array([0.2], dtype=np.float32))])
        m = g.node("ReduceMax", [y], axes=[3], keepdims=1)                     # [B, c, h, 1]
        t = g.node("Tile", [m, i64(1, 1, 1, w)])                               # broadcast back along the width
        return g.node("Sub", [y, t])

    got, ref = run_both(bn, op_graph(build, [c, h, w]), scale=3.0)
    assert_close(got, ref, "InstanceNormalization + PRelu + Tile")


# ---------------------------------------------------------------- exporter dialects (round 5)
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["frames_axis2", "opset20_default_axis", "zero_padded", "truncated", "two_sided", "leading_dims", "transposed_frames"])
def test_dft_node(bn, case):
    """ONNX DFT (what an opset-17+ exporter writes for an RFFT front end; classifier.rs:340-350 loads whatever ORT loads) is mapped onto
    the framing kernels like STFT: frames as rows of a framing convolution with a cos | -sin bank.  Axis as attribute (opset 17) and by
    default (opset 20: -2), dft_length longer (zero padding) and shorter (truncation) than the frames, two-sided, extra leading
    dimensions, and frames whose transformed axis is not innermost (copied first) -- raw [.., bins, 2] output against the oracle's
    torch.fft call."""
    from gpu_helpers import writer
    L, F = 256, 120
    opset = 20 if case == "opset20_default_axis" else 17
    nfft = {"zero_padded": 320, "truncated": 192}.get(case, L)
    onesided = 0 if case == "two_sided" else 1
    bins = nfft // 2 + 1 if onesided else nfft

    g = writer.GraphBuilder(opset=opset)
    g.add_input("input", [None, 144000])
    i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
    x = g.node("Slice", ["input", i64(0), i64(F * L), i64(1), i64(1)])
    if case == "leading_dims":
        x = g.node("Reshape", [x, i64(-1, 4, F // 4, L, 1)])
        axis, out_shape = 3, [4, F // 4, bins, 2]
    elif case == "transposed_frames":
        x = g.node("Reshape", [x, i64(-1, L, F)])
        x = g.node("Transpose", [x], perm=[0, 2, 1])                      # [B, F, L] with the samples of a frame F apart
        x = g.node("Unsqueeze", [x, i64(3)])
        axis, out_shape = 2, [F, bins, 2]
    else:
        x = g.node("Reshape", [x, i64(-1, F, L, 1)])
        axis, out_shape = 2, [F, bins, 2]
    ins = [x] + ([g.const(np.array(nfft, dtype=np.int64), scalar=True)] if nfft != L else [])
    if case == "opset20_default_axis":
        y = g.node("DFT", ins, onesided=onesided)                            # opset 20: axis is an input, default -2
    else:
        y = g.node("DFT", ins, axis=axis, onesided=onesided)
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None] + out_shape)
    data = g.serialize()
    text = bn.plan_describe(write_model(data))
    copies = [l for l in text.splitlines() if " copy(" in l and "copy(output" not in l]  # (the graph output itself is copied into its dense layout)
    assert ("dft:" in text) and (bool(copies) == (case == "transposed_frames")), text
    got, ref = run_both(bn, data)
    assert ref.shape[1:] == tuple(out_shape)
    assert_close(got, ref, f"DFT {case}", atol=2e-4 * np.sqrt(L / 256) * 4, rtol=2e-4)


@pytest.mark.gpu
def test_tf_signal_frame_gather_is_a_view_and_the_window_joins_the_dft_bank(bn):
    """tf.signal.stft as a TensorFlow export spells it: Reshape into sub-frames of gcd(L, hop) samples, Gather with the constant affine
    selector, Reshape to [frames, L], Mul by the window, DFT.  The Gather and both reshapes are views (no copy launch), the window is
    folded into the DFT bank (no multiply launch) and the bank folds like any windowed-DFT bank: ONE or two framing launches in the plan."""
    L, hop, S = 512, 120, 48000
    sub = int(np.gcd(L, hop))
    F = (S - L) // hop + 1

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(S), i64(1), i64(1)])
        xs = g.node("Reshape", [x, i64(0, S // sub, sub)])
        sel = np.arange(F, dtype=np.int64)[:, None] * (hop // sub) + np.arange(L // sub, dtype=np.int64)[None, :]
        fr = g.node("Reshape", [g.node("Gather", [xs, g.const(sel)], axis=1), i64(0, F, L)])
        w = g.node("Mul", [fr, g.const(_hann(L))])
        return g.node("DFT", [g.node("Unsqueeze", [w, i64(3)])], axis=2, onesided=1)

    data = op_graph(build, [F, L // 2 + 1, 2])
    text = bn.plan_describe(write_model(data))
    launches = [l for l in text.splitlines() if l[:3].strip().isdigit()]
    assert len(launches) <= 3 and not any(" ELT " in l and (("copy(" in l and "copy(output" not in l) or "Mul:" in l or "window.mul" in l) for l in launches), text
    assert (f"lda={hop} " in text and "fold=" in text) or f" FFT " in text and f"hop={hop} " in text, text  # overlapping rows straight from the signal: folded taps or an FFT
    got, ref = run_both(bn, data)
    assert_close(got, ref, "tf.signal.frame + window + DFT", atol=4e-4, rtol=2e-4)


@pytest.mark.gpu
def test_gather_that_is_not_affine_is_refused_and_a_pending_window_is_applied_for_other_consumers(bn):
    rng = np.random.default_rng(5)

    def bad(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [g.node("Slice", [x, i64(0), i64(4000), i64(1), i64(1)]), i64(0, 100, 40)])
        return g.node("Gather", [x, g.const(np.array([0, 1, 3, 7], dtype=np.int64))], axis=1)
    with pytest.raises(bn.EngineError):
        bn.Model(write_model(op_graph(bad, [4, 40])))
    assert "affine" in bn.last_error()

    # frames * window read by a DFT AND by something else: the window becomes an ordinary multiply
    L, F = 128, 50

    def both(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [g.node("Slice", [x, i64(0), i64(F * L), i64(1), i64(1)]), i64(0, F, L)])
        w = g.node("Mul", [x, g.const(_hann(L))])
        d = g.node("DFT", [g.node("Unsqueeze", [w, i64(3)])], axis=2, onesided=1)      # [B, F, 65, 2]
        e = g.node("ReduceSum", [g.node("Mul", [w, w]), i64(1, 2)], keepdims=1)         # second consumer of the windowed frames: [B, 1, 1]
        return g.node("Add", [d, g.node("Reshape", [e, i64(0, 1, 1, 1)])])
    data = op_graph(both, [F, L // 2 + 1, 2])
    got, ref = run_both(bn, data)
    assert_close(got, ref, "window with two consumers", atol=4e-4, rtol=2e-4)
    # strided, negative and scalar indices as views
    def views(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [g.node("Slice", [x, i64(0), i64(6000), i64(1), i64(1)]), i64(0, 100, 60)])
        a = g.node("Gather", [x, g.const(np.array([[90, 80], [70, 60], [50, 40]], dtype=np.int64))], axis=1)      # [B, 3, 2, 60], steps -20 / -10
        b = g.node("Gather", [a, g.const(np.array(-1, dtype=np.int64), scalar=True)], axis=2)                     # [B, 3, 60]
        return g.node("Gather", [b, g.const(np.arange(5, 60, 11, dtype=np.int64))], axis=2)                       # [B, 3, 5]
    got, ref = run_both(bn, op_graph(views, [3, 5]))
    assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_comparisons_logic_where_and_casts(bn):
    """Greater / Less / Equal / GreaterOrEqual / LessOrEqual against constants and tensors, Not / And / Or / Xor, Where with tensor and
    constant branches (incl. a branch holding inf where it is NOT selected), Cast to bool / int / float -- "bool" tensors are 0.0 / 1.0
    on the device; results equal the oracle's torch calls exactly (selection and comparison involve no rounding)."""
    rng = np.random.default_rng(11)
    c, n = 6, 500
    thr = rng.standard_normal((c, 1)).astype(np.float32)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        f32 = lambda v: g.const(np.array(v, dtype=np.float32))
        x = g.node("Reshape", [g.node("Slice", [x, i64(0), i64(2 * c * n), i64(1), i64(1)]), i64(0, 2, c, n)])
        a = g.node("Gather", [x, g.const(np.array(0, dtype=np.int64), scalar=True)], axis=1)        # [B, c, n]
        b = g.node("Gather", [x, g.const(np.array(1, dtype=np.int64), scalar=True)], axis=1)
        gt = g.node("Greater", [a, f32(0.25)])
        lt = g.node("Less", [f32(-0.5), b])                                                         # constant on the left: -0.5 < b
        ge = g.node("GreaterOrEqual", [a, b])
        le = g.node("LessOrEqual", [a, g.const(thr)])                                               # per-channel thresholds
        eq = g.node("Equal", [g.node("Floor", [a]), g.node("Floor", [b])])
        m1 = g.node("And", [gt, lt])
        m2 = g.node("Or", [g.node("Not", [ge]), le])
        m3 = g.node("Xor", [m1, eq])
        big = g.node("Div", [f32(1.0), g.node("Sub", [a, a])])                                      # 1 / 0 = inf (nan where a is nan): must not leak
        w1 = g.node("Where", [m1, a, b])
        w2 = g.node("Where", [m2, f32(0.0), w1])
        w3 = g.node("Where", [g.node("Cast", [m3], to=9), big, f32(-2.0)])                          # inf selected where m3, else -2
        w3 = g.node("Where", [m3, f32(7.0), w3])                                                    # ... and every inf replaced again
        ci = g.node("Cast", [g.node("Mul", [a, f32(3.7)])], to=7)                                   # float -> int64: toward zero
        cb = g.node("Cast", [g.node("Relu", [b])], to=9)                                            # float -> bool: != 0
        masks = g.node("Add", [g.node("Cast", [m2], to=1), g.node("Mul", [g.node("Cast", [m3], to=1), f32(2.0)])])
        outs = [w1, w2, w3, g.node("Cast", [ci], to=1), g.node("Cast", [cb], to=1), masks]
        return g.node("Concat", [g.node("Unsqueeze", [o, i64(1)]) for o in outs], axis=1)
    data = op_graph(build, [6, c, n])
    got, ref = run_both(bn, data)
    assert np.isfinite(ref).all()
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


@pytest.mark.gpu
def test_operators_lowered_as_compositions(bn):
    """Elu / Selu / Celu / ThresholdedRelu / Softsign / Mish / Gelu (both forms) / Sign / Round / Sum / Mean / ReduceL1 / ReduceLogSum /
    ReduceLogSumExp / LayerNormalization: written out by the planner as the operators it already maps (engine.cpp lower_composite), checked
    against the oracle's torch calls; the inputs span both signs, exact ties for Round and magnitudes at which a naive log-sum-exp
    overflows."""
    c, n = 5, 400
    rng = np.random.default_rng(21)
    scale = rng.standard_normal((n,)).astype(np.float32)
    bias = rng.standard_normal((n,)).astype(np.float32)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        f32 = lambda v: g.const(np.array(v, dtype=np.float32))
        x = g.node("Reshape", [g.node("Slice", [x, i64(0), i64(2 * c * n), i64(1), i64(1)]), i64(0, 2, c, n)])
        a = g.node("Gather", [x, g.const(np.array(0, dtype=np.int64), scalar=True)], axis=1)        # [B, c, n]
        b = g.node("Gather", [x, g.const(np.array(1, dtype=np.int64), scalar=True)], axis=1)
        outs = [g.node("Elu", [a], alpha=0.7), g.node("Selu", [a]), g.node("Celu", [b], alpha=1.5), g.node("ThresholdedRelu", [a], alpha=0.3),
                g.node("Softsign", [b]), g.node("Mish", [a]), g.node("Gelu", [a]), g.node("Gelu", [b], approximate="tanh"), g.node("Sign", [a]),
                g.node("Round", [g.node("Mul", [g.node("Round", [g.node("Mul", [a, f32(4.0)])]), f32(0.5)])]),   # multiples of 0.5: every other one a tie
                g.node("Sum", [a, b, a]), g.node("Mean", [a, b, b, a]),
                g.node("LayerNormalization", [a, g.const(scale), g.const(bias)], axis=-1, epsilon=1e-4),
                g.node("LayerNormalization", [b, g.const(np.ones((c, n), dtype=np.float32))], axis=1)]
        full = g.node("Concat", [g.node("Unsqueeze", [o, i64(1)]) for o in outs], axis=1)             # [B, 14, c, n]
        big = g.node("Mul", [a, f32(60.0)])                                                           # exp(60 * 3) overflows f32
        red = [g.node("ReduceL1", [a, i64(2)], keepdims=1), g.node("ReduceLogSum", [g.node("Abs", [b]), i64(2)], keepdims=1),
               g.node("ReduceLogSumExp", [big, i64(2)], keepdims=1),
               g.node("Unsqueeze", [g.node("ReduceLogSumExp", [b], axes=[2], keepdims=0), i64(2)])]
        small = g.node("Concat", [g.node("Unsqueeze", [o, i64(1)]) for o in red], axis=1)             # [B, 4, c, 1]
        return g.node("Concat", [full, g.node("Mul", [small, g.const(np.ones((n,), dtype=np.float32))])], axis=1)
    data = op_graph(build, [18, c, n])
    got, ref = run_both(bn, data)
    assert np.isfinite(ref).all()
    assert_close(got, ref, "composite operators", atol=2e-5, rtol=2e-5)
    assert np.array_equal(got[:, 8:10], ref[:, 8:10])    # Sign and Round involve no rounding of their own


# ---------------------------------------------------------------- GEMMs on the bf16 matrix pipe with f32-complete products (round 5)
@pytest.mark.gpu
@pytest.mark.parametrize("cin,h,w,cout,act,kernel", [
    (240, 6, 32, 80, "relu", "b3"),      # a late project-conv shape: 192 rows per sample, 15 K steps padded to 16
    (672, 6, 32, 112, None, "b3"),       # K = 672: 21 steps + the zero padding step; seven channel tiles = seven waves
    (1152, 3, 16, 320, "silu", "b3"),    # 48 rows per sample: row tiles span samples; three channel blocks of 7 / 7 / 6 tiles
    (136, 8, 32, 816, "silu", "b3"),     # an expand conv the tiled kernel used to keep: K % 16 == 8, N = 51 tiles
    (80, 5, 16, 100, "relu", "b3"),      # 80 rows per sample (no multiple of 32), N = 100: a partial last channel tile
    (144, 8, 16, 40, None, "dma3"),      # narrow project conv: the LDS-DMA form, K % 32 == 16 (half step)
    (256, 4, 16, 24, "relu", "dma3"),
])
def test_gemm_bf16x3_forms_against_the_oracle_and_the_exact_f32_kernels(bn, cin, h, w, cout, act, kernel, monkeypatch):
    """1x1 convs that take gemm_b3_kernel / gemm_dma3_kernel (exact three-way bf16 split of both operands, six partial products on the
    bf16 matrix pipe): the plan names the kernel, the result agrees with the oracle far inside the suite's tolerance -- to a few f32
    roundings of the sum, like the exact-f32 kernel the same layer gets under BN_GEMM3=0 -- and does not depend on the row tile, the batch
    it rides in or its position in the batch (bit for bit)."""
    rng0 = np.random.default_rng(cin + cout)
    wa = (rng0.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    wt = (rng0.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    wb = (rng0.standard_normal((8, cout, 1, 1)) / np.sqrt(cout)).astype(np.float32)

    def build(g, xin):  # 1x1 (brings the map into the channels-last layout) -> the conv under test, dense in and out -> 1x1 down to 8 channels
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        y = g.node("Reshape", [g.node("Slice", [xin, i64(0), i64(cin * h * w), i64(1), i64(1)]), i64(-1, cin, h, w)])
        y = g.node("Conv", [y, g.const(wa)], kernel_shape=[1, 1])
        y = g.node("Conv", [y, g.const(wt), g.const(rng0.standard_normal(cout).astype(np.float32))], kernel_shape=[1, 1])
        if act == "relu":
            y = g.node("Relu", [y])
        elif act == "silu":
            y = g.node("Mul", [y, g.node("Sigmoid", [y])])
        return g.node("Conv", [y, g.const(wb)], kernel_shape=[1, 1])
    data = op_graph(build, [8, h, w])
    path = write_model(data)
    text = bn.plan_describe(path)
    under_test = [l for l in text.splitlines() if f" K={cin} N={cout} " in l]
    assert len(under_test) == 1 and f"kernel={kernel}" in under_test[0], text
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((5, 144000))).astype(np.float32)
    ref = onnx_ref.run_model(data, x)["output"].reshape(5, -1)
    got = bn.Context(bn.Model(path), 5).infer(x)[0].reshape(5, -1).copy()
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-5 * scale, (np.abs(got - ref).max(), scale)
    monkeypatch.setenv("BN_GEMM3", "0")
    assert "kernel=b3" not in bn.plan_describe(path) and "kernel=dma3" not in bn.plan_describe(path)
    f32 = bn.Context(bn.Model(path), 5).infer(x)[0].reshape(5, -1).copy()
    monkeypatch.delenv("BN_GEMM3")
    assert np.abs(f32 - ref).max() <= 2e-5 * scale
    assert np.abs(got - f32).max() <= 2e-5 * scale
    # the tile and the batch do not enter the arithmetic
    one = np.concatenate([bn.Context(bn.Model(path), 1).infer(x[i:i + 1])[0].reshape(1, -1) for i in (0, 4)])
    assert one.tobytes() == got[[0, 4]].tobytes()
    if kernel == "b3":
        for mt in ("2", "4", "3"):  # rows per block: 32, 64, 48
            monkeypatch.setenv("BN_GEMMB3_MT", mt)
            alt = bn.Context(bn.Model(path), 5).infer(x)[0].reshape(5, -1)
            assert alt.tobytes() == got.tobytes(), mt
        monkeypatch.delenv("BN_GEMMB3_MT")
    big = np.tile(x, (7, 1))[:33]
    many = bn.Context(bn.Model(path), 33).infer(big)[0].reshape(33, -1)
    assert many[:5].tobytes() == got.tobytes() and many[30:33].tobytes() == got[[0, 1, 2]].tobytes()


@pytest.mark.gpu
def test_bf16x3_split_is_exact_and_the_weight_images_round_trip(bn):
    """The arithmetic's premise, on the host packer the planner uses (through the plan's constants nothing is observable, so the check runs
    on numpy's restatement of the same three masks / subtractions): x == hi + mid + lo exactly for every finite f32, each term a bf16
    number; and the device agrees with it -- a GEMM whose weights are powers of two times ones reproduces its input sums exactly."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(100000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 100000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.4e38, 1.2e-30], dtype=np.float32)])
    x = x[(np.abs(x) >= np.float32(1e-30)) | (x == 0)]  # (below 2^-100 the third term leaves bf16's normal range: not activations)
    top = lambda v: (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    hi = top(x); r1 = x - hi; mid = top(r1); lo = r1 - mid
    assert np.array_equal(hi + mid + lo, x) and np.array_equal((lo.view(np.uint32) & np.uint32(0xffff)), np.zeros_like(lo.view(np.uint32)))
    # identity-like product: out[n] = x[k == n] * 2^-3 (exact in every arithmetic that keeps 24 bits)
    cin = cout = 128

    def build(g, xin):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        y = g.node("Reshape", [g.node("Slice", [xin, i64(0), i64(cin * 64), i64(1), i64(1)]), i64(-1, cin, 8, 8)])
        eye = lambda f: g.const((np.eye(cin, dtype=np.float32) * np.float32(f)).reshape(cout, cin, 1, 1))
        y = g.node("Conv", [y, eye(1.0)], kernel_shape=[1, 1])        # into the channels-last layout
        y = g.node("Conv", [y, eye(0.125)], kernel_shape=[1, 1])      # dense in, dense out: the bf16x3 kernel
        return g.node("Conv", [y, eye(2.0)], kernel_shape=[1, 1])     # (its output view is the graph's NCHW: the tiled f32 kernel)
    data = op_graph(build, [cout, 8, 8])
    assert bn.plan_describe(write_model(data)).count("kernel=b3") == 2
    xs = (rng.standard_normal((2, 144000)) * 3.0).astype(np.float32)
    got = bn.Context(bn.Model(write_model(data)), 2).infer(xs)[0].reshape(2, -1)
    want = (xs[:, :cin * 64] * np.float32(0.25))
    assert np.array_equal(got, want)
