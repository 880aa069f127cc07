"""Seeded synthetic-weight model files on the hypothesised topologies.

No BirdNET / Perch .onnx file can be obtained offline, so tests and the
benchmark author their own model files: same I/O contract as the real exports
(input name/shape, output names/order/shapes: reference src/detection.rs:31-71,
src/batch_context.rs:222,249-261), plausible architecture, random weights.

* ``birdnet_v24``  two mel spectrograms (n_fft 2048/hop 278/0-3 kHz and n_fft
  1024/hop 280/0.5-15 kHz, 96x511 each, Hann window, real part of the STFT,
  power-law magnitude scaling) -> 2-channel image -> EfficientNet-B0-like
  MBConv stack (ReLU, squeeze-excite) -> 1024-d -> Dense(num_species).
* ``birdnet_v30``  one 128-mel log spectrogram at 32 kHz -> same family of
  backbone (SiLU) -> outputs (embeddings[1024], logits[N]).
* ``perch_v2``     power spectrogram (cos+sin DFT, n_fft 640, hop 320, SAME
  padding) -> 128 log-mels [500,128] -> EfficientNet-B3-like (SiLU, SE) ->
  outputs (embedding[1536], spatial_embedding[16,4,1536], spectrogram[500,128],
  logits[14795]).

All three are emitted in "exporter" style: NCHW Conv + BatchNormalization +
activation nodes, Transposes, Slices and MatMuls -- nothing is pre-fused, so the
engine's import-time fusion and layout handling are what gets exercised.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .onnx_writer import GraphBuilder

B0_STAGES = [(1, 16, 1, 1, 3), (6, 24, 2, 2, 3), (6, 40, 2, 2, 5), (6, 80, 3, 2, 3), (6, 112, 3, 1, 5),
             (6, 192, 4, 2, 5), (6, 320, 1, 1, 3)]
B3_STAGES = [(1, 24, 2, 1, 3), (6, 32, 3, 2, 3), (6, 48, 3, 2, 5), (6, 96, 5, 2, 3), (6, 136, 5, 1, 5),
             (6, 232, 6, 2, 5), (6, 384, 2, 1, 3)]


@dataclass
class BackboneSpec:
    stem: int = 32
    stages: list = field(default_factory=lambda: list(B0_STAGES))
    head: int = 1024
    act: str = "relu"      # relu | relu6 | silu
    se: bool = True
    bn_nodes: bool = True  # emit BatchNormalization nodes (else pre-folded conv bias)


def _round8(c: float) -> int:
    return max(8, int(c + 4) // 8 * 8)


def scaled_stages(stages, width: float, depth: float):
    return [(e, _round8(c * width), max(1, int(math.ceil(r * depth))), s, k) for e, c, r, s, k in stages]


def hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_bins: int, n_mels: int, sr: int, fmin: float, fmax: float) -> np.ndarray:
    """Triangular (HTK-mel) filterbank, [n_bins, n_mels], zero outside [fmin, fmax]."""
    freqs = np.linspace(0.0, sr / 2.0, n_bins)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_bins, n_mels), dtype=np.float64)
    for m in range(n_mels):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        up = (freqs - lo) / max(ce - lo, 1e-9)
        dn = (hi - freqs) / max(hi - ce, 1e-9)
        fb[:, m] = np.maximum(0.0, np.minimum(up, dn))
    return fb.astype(np.float32)


def dft_basis(n_fft: int, kind: str) -> np.ndarray:
    """Hann-windowed DFT basis rows, [bins(, x2), 1, n_fft] as a Conv1D weight."""
    n = np.arange(n_fft, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)  # periodic Hann (tf.signal.hann_window)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * k * n[None, :] / n_fft
    cos = np.cos(ang) * win
    if kind == "real":
        w = cos
    else:
        w = np.concatenate([cos, -np.sin(ang) * win], axis=0)
    return w[:, None, :].astype(np.float32)


class _Net:
    def __init__(self, g: GraphBuilder, rng: np.random.RandomState, spec: BackboneSpec):
        self.g, self.rng, self.spec = g, rng, spec

    def act(self, x: str) -> str:
        a = self.spec.act
        if a == "relu":
            return self.g.node("Relu", [x])
        if a == "relu6":
            return self.g.node("Clip", [x, self.g.const(np.float32(0.0)), self.g.const(np.float32(6.0))])
        s = self.g.node("Sigmoid", [x])
        return self.g.node("Mul", [x, s])

    def conv_bn(self, x: str, cin: int, cout: int, k: int, stride: int, groups: int = 1, act: bool = True,
                gain: float = 1.0) -> str:
        g, rng = self.g, self.rng
        fan_in = (cin // groups) * k * k
        w = (rng.standard_normal((cout, cin // groups, k, k)) * math.sqrt(2.0 / fan_in) * gain).astype(np.float32)
        pad = k // 2
        if self.spec.bn_nodes:
            y = g.node("Conv", [x, g.const(w, "w")], kernel_shape=[k, k], strides=[stride, stride],
                       pads=[pad, pad, pad, pad], group=groups)
            gamma = (1.0 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
            beta = (0.05 * rng.standard_normal(cout)).astype(np.float32)
            mean = (0.05 * rng.standard_normal(cout)).astype(np.float32)
            var = (1.0 + 0.1 * rng.rand(cout)).astype(np.float32)
            y = g.node("BatchNormalization", [y, g.const(gamma), g.const(beta), g.const(mean), g.const(var)],
                       epsilon=1e-3)
        else:
            b = (0.05 * rng.standard_normal(cout)).astype(np.float32)
            y = g.node("Conv", [x, g.const(w, "w"), g.const(b, "b")], kernel_shape=[k, k],
                       strides=[stride, stride], pads=[pad, pad, pad, pad], group=groups)
        return self.act(y) if act else y

    def se(self, x: str, c: int, c_red: int) -> str:
        g, rng = self.g, self.rng
        s = g.node("GlobalAveragePool", [x])
        w1 = (rng.standard_normal((c_red, c, 1, 1)) * math.sqrt(2.0 / c)).astype(np.float32)
        b1 = (0.05 * rng.standard_normal(c_red)).astype(np.float32)
        s = g.node("Conv", [s, g.const(w1), g.const(b1)], kernel_shape=[1, 1])
        s = self.act(s)
        w2 = (rng.standard_normal((c, c_red, 1, 1)) * math.sqrt(1.0 / c_red)).astype(np.float32)
        b2 = (0.5 + 0.05 * rng.standard_normal(c)).astype(np.float32)
        s = g.node("Conv", [s, g.const(w2), g.const(b2)], kernel_shape=[1, 1])
        s = g.node("Sigmoid", [s])
        return g.node("Mul", [x, s])

    def mbconv(self, x: str, cin: int, cout: int, expand: int, k: int, stride: int) -> str:
        inp = x
        mid = cin * expand
        if expand != 1:
            x = self.conv_bn(x, cin, mid, 1, 1)
        x = self.conv_bn(x, mid, mid, k, stride, groups=mid)
        if self.spec.se:
            x = self.se(x, mid, max(1, cin // 4))
        x = self.conv_bn(x, mid, cout, 1, 1, act=False, gain=0.7)
        if stride == 1 and cin == cout:
            x = self.g.node("Add", [x, inp])
        return x

    def backbone(self, x: str, cin: int) -> tuple[str, int]:
        sp = self.spec
        x = self.conv_bn(x, cin, sp.stem, 3, 2)
        c = sp.stem
        for expand, cout, repeats, stride, k in sp.stages:
            for r in range(repeats):
                x = self.mbconv(x, c, cout, expand, k, stride if r == 0 else 1)
                c = cout
        x = self.conv_bn(x, c, sp.head, 1, 1)
        return x, sp.head


def _dense(g: GraphBuilder, rng, x: str, cin: int, cout: int, style: str, out_name: str) -> str:
    w = (rng.standard_normal((cout, cin)) * (2.5 / math.sqrt(cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.5 - 2.0).astype(np.float32)
    if style == "gemm":
        return g.node("Gemm", [x, g.const(w, "fc_w"), g.const(b, "fc_b")], outputs=[out_name], transB=1)
    y = g.node("MatMul", [x, g.const(np.ascontiguousarray(w.T), "fc_w")])
    return g.node("Add", [y, g.const(b, "fc_b")], outputs=[out_name])


def _minmax_normalise(g: GraphBuilder, x: str) -> str:
    """x -> 2 * ((x - min) / (max(x - min) + 1e-6) - 0.5), per segment."""
    mn = g.node("ReduceMin", [x], axes=[1], keepdims=1)
    x1 = g.node("Sub", [x, mn])
    mx = g.node("ReduceMax", [x1], axes=[1], keepdims=1)
    x2 = g.node("Div", [x1, g.node("Add", [mx, g.const(np.float32(1e-6))])])
    x3 = g.node("Sub", [x2, g.const(np.float32(0.5))])
    return g.node("Mul", [x3, g.const(np.float32(2.0))])


def _frames_tf_signal(g: GraphBuilder, x: str, samples: int, n_fft: int, hop: int) -> str:
    """[B,S] -> [B, frames, n_fft] the way a TensorFlow export writes tf.signal.frame: the signal reshaped into sub-frames of
    gcd(n_fft, hop) samples, a Gather with the constant selector frame * (hop / sub) + j, and a Reshape that merges the sub-frames."""
    sub = math.gcd(n_fft, hop)
    assert samples % sub == 0
    frames = (samples - n_fft) // hop + 1
    xs = g.node("Reshape", [x, g.const(np.array([0, samples // sub, sub], dtype=np.int64))])
    sel = (np.arange(frames, dtype=np.int64)[:, None] * (hop // sub) + np.arange(n_fft // sub, dtype=np.int64)[None, :])
    gth = g.node("Gather", [xs, g.const(sel)], axis=1)                          # [B, frames, n_fft / sub, sub]
    return g.node("Reshape", [gth, g.const(np.array([0, frames, n_fft], dtype=np.int64))])


def _mel_branch_real(g: GraphBuilder, x3: str, sr: int, n_fft: int, hop: int, n_mels: int, fmin: float,
                     fmax: float, mag_scale: float, dialect: str = "conv", samples: int = 0) -> str:
    """[B,S] -> [B,1,n_mels,frames]: real STFT part -> mel -> ^2 -> ^(1/(1+e^mag_scale)) -> flip -> transpose.
    dialect "conv": the windowed DFT basis as a Conv1D weight (the default).  "dft": what an opset-17+ exporter writes for
    tf.signal.stft -- tf.signal.frame (Reshape / Gather / Reshape), the periodic Hann window as a Mul, an ONNX DFT node
    (onesided) and a Gather of its real part.  "stft": one opset-17 STFT node + the same Gather."""
    if dialect == "conv":
        u = g.node("Unsqueeze", [x3, g.const(np.array([1], dtype=np.int64))])
        c = g.node("Conv", [u, g.const(dft_basis(n_fft, "real"), "dft")], kernel_shape=[n_fft], strides=[hop])
        t = g.node("Transpose", [c], perm=[0, 2, 1])                         # [B, frames, bins]
    else:
        n = np.arange(n_fft, dtype=np.float64)
        hann = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)
        if dialect == "dft":
            fr = _frames_tf_signal(g, x3, samples, n_fft, hop)
            w = g.node("Mul", [fr, g.const(hann, "window")])
            u = g.node("Unsqueeze", [w, g.const(np.array([3], dtype=np.int64))])   # [B, frames, n_fft, 1]
            d = g.node("DFT", [u], axis=2, onesided=1)                              # [B, frames, bins, 2]
        else:
            d = g.node("STFT", [x3, g.const(np.array(hop, dtype=np.int64), scalar=True), g.const(hann, "window")], onesided=1)
        t = g.node("Gather", [d, g.const(np.array(0, dtype=np.int64), scalar=True)], axis=3)  # the real part, [B, frames, bins]
    m = g.node("MatMul", [t, g.const(mel_filterbank(n_fft // 2 + 1, n_mels, sr, fmin, fmax), "mel")])
    p = g.node("Pow", [m, g.const(np.float32(2.0))])
    q = g.node("Pow", [p, g.const(np.float32(1.0 / (1.0 + math.exp(mag_scale))))])
    r = g.node("Slice", [q, g.const(np.array([-1], dtype=np.int64)), g.const(np.array([-(2 ** 62)], dtype=np.int64)),
                         g.const(np.array([2], dtype=np.int64)), g.const(np.array([-1], dtype=np.int64))])  # reverse mel axis
    s = g.node("Transpose", [r], perm=[0, 2, 1])                             # [B, n_mels, frames]
    return g.node("Unsqueeze", [s, g.const(np.array([1], dtype=np.int64))])


def birdnet_v24(num_species: int = 6522, seed: int = 24, width: float = 1.0, depth: float = 1.0,
                head: int = 1024, se: bool = True, bn_nodes: bool = True, builder_out: list | None = None, front_end: str = "conv") -> bytes:
    """builder_out: if a list is given, the GraphBuilder is appended to it (tests read the raw constants from it).
    front_end: "conv" (windowed DFT basis as Conv weights), "dft" (tf.signal.frame + window Mul + ONNX DFT) or "stft" (ONNX STFT):
    three spellings of the same spectrogram, identical weights behind them."""
    rng = np.random.RandomState(seed)
    g = GraphBuilder("birdnet_v24_synth")
    if builder_out is not None:
        builder_out.append(g)
    S, sr = 144000, 48000
    g.add_input("input", [None, S])
    x3 = _minmax_normalise(g, "input")
    lo = _mel_branch_real(g, x3, sr, 2048, 278, 96, 0.0, 3000.0, 1.23, front_end, S)
    hi = _mel_branch_real(g, x3, sr, 1024, 280, 96, 500.0, 15000.0, 1.23, front_end, S)
    img = g.node("Concat", [lo, hi], axis=1)                                  # [B,2,96,511]
    gamma = np.array([0.02, 0.03], dtype=np.float32)
    img = g.node("BatchNormalization", [img, g.const(gamma), g.const(np.array([-0.5, -0.6], dtype=np.float32)),
                                        g.const(np.array([20.0, 15.0], dtype=np.float32)),
                                        g.const(np.array([400.0, 300.0], dtype=np.float32))], epsilon=1e-3)
    spec = BackboneSpec(stem=_round8(32 * width), stages=scaled_stages(B0_STAGES, width, depth), head=head,
                        act="relu", se=se, bn_nodes=bn_nodes)
    net = _Net(g, rng, spec)
    f, c = net.backbone(img, 2)
    p = g.node("GlobalAveragePool", [f])
    p = g.node("Flatten", [p], axis=1)
    _dense(g, rng, p, c, num_species, "gemm", "output")
    g.add_output("output", [None, num_species])
    return g.serialize()


def birdnet_v30(num_species: int = 1000, seed: int = 30, width: float = 1.0, depth: float = 1.0,
                emb: int = 1024, builder_out: list | None = None) -> bytes:
    rng = np.random.RandomState(seed)
    g = GraphBuilder("birdnet_v30_synth")
    if builder_out is not None:
        builder_out.append(g)
    S, sr, n_fft, hop, n_mels = 160000, 32000, 1024, 320, 128
    g.add_input("input", [None, S])
    u = g.node("Unsqueeze", ["input", g.const(np.array([1], dtype=np.int64))])
    c = g.node("Conv", [u, g.const(dft_basis(n_fft, "complex"), "dft")], kernel_shape=[n_fft], strides=[hop])
    bins = n_fft // 2 + 1
    i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
    re = g.node("Slice", [c, i64(0), i64(bins), i64(1), i64(1)])
    im = g.node("Slice", [c, i64(bins), i64(2 * bins), i64(1), i64(1)])
    pw = g.node("Add", [g.node("Mul", [re, re]), g.node("Mul", [im, im])])
    mag = g.node("Sqrt", [pw])
    t = g.node("Transpose", [mag], perm=[0, 2, 1])                             # [B, frames, bins]
    m = g.node("MatMul", [t, g.const(mel_filterbank(bins, n_mels, sr, 40.0, 15000.0), "mel")])
    lg = g.node("Log", [g.node("Add", [m, g.const(np.float32(1e-3))])])
    s = g.node("Transpose", [lg], perm=[0, 2, 1])                              # [B, n_mels, frames]
    img = g.node("Unsqueeze", [s, i64(1)])
    spec = BackboneSpec(stem=_round8(32 * width), stages=scaled_stages(B0_STAGES, width, depth), head=emb,
                        act="silu", se=True, bn_nodes=False)
    net = _Net(g, rng, spec)
    f, cch = net.backbone(img, 1)
    p = g.node("ReduceMean", [f], axes=[2, 3], keepdims=0)
    e = g.node("Identity", [p], outputs=["output_0"])
    _dense(g, rng, e, cch, num_species, "matmul", "output_1")
    g.add_output("output_0", [None, emb])
    g.add_output("output_1", [None, num_species])
    return g.serialize()


def perch_v2(num_species: int = 14795, seed: int = 2, width: float = 1.0, depth: float = 1.0,
             emb: int = 1536, builder_out: list | None = None) -> bytes:
    rng = np.random.RandomState(seed)
    g = GraphBuilder("perch_v2_synth")
    if builder_out is not None:
        builder_out.append(g)
    S, sr, n_fft, hop, n_mels = 160000, 32000, 640, 320, 128
    g.add_input("inputs", [None, S])
    i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
    u = g.node("Unsqueeze", ["inputs", i64(1)])
    pad = (n_fft - hop) // 2
    c = g.node("Conv", [u, g.const(dft_basis(n_fft, "complex"), "dft")], kernel_shape=[n_fft], strides=[hop],
               pads=[pad, pad])                                                # [B, 2*bins, 500]
    bins = n_fft // 2 + 1
    re = g.node("Slice", [c, i64(0), i64(bins), i64(1), i64(1)])
    im = g.node("Slice", [c, i64(bins), i64(2 * bins), i64(1), i64(1)])
    pw = g.node("Add", [g.node("Mul", [re, re]), g.node("Mul", [im, im])])
    t = g.node("Transpose", [pw], perm=[0, 2, 1])                              # [B, 500, bins]
    m = g.node("MatMul", [t, g.const(mel_filterbank(bins, n_mels, sr, 60.0, 16000.0), "mel")])
    lg = g.node("Log", [g.node("Max", [m, g.const(np.float32(1e-5))])])
    sp = g.node("Mul", [lg, g.const(np.float32(0.1))], outputs=["spectrogram"])  # [B, 500, 128]
    img = g.node("Unsqueeze", [sp, i64(1)])                                    # [B,1,500,128]
    spec = BackboneSpec(stem=_round8(40 * width), stages=scaled_stages(B3_STAGES, width, depth), head=emb,
                        act="silu", se=True, bn_nodes=True)
    net = _Net(g, rng, spec)
    f, cch = net.backbone(img, 1)                                              # [B,1536,16,4]
    spatial = g.node("Transpose", [f], perm=[0, 2, 3, 1], outputs=["spatial_embedding"])
    e = g.node("ReduceMean", [spatial], axes=[1, 2], keepdims=0, outputs=["embedding"])
    _dense(g, rng, e, cch, num_species, "gemm", "label")
    h = 500
    w = n_mels
    for _ in range(5):
        h, w = (h + 1) // 2, (w + 1) // 2
    g.add_output("embedding", [None, emb])
    g.add_output("spatial_embedding", [None, h, w, emb])
    g.add_output("spectrogram", [None, 500, n_mels])
    g.add_output("label", [None, num_species])
    return g.serialize()


def synthetic_segments(n: int, sample_count: int, sample_rate: int, first_index: int = 0) -> np.ndarray:
    """SURVEY.md 8(d) inputs: 0.5*sin(2*pi*f*t) + 0.05*LCG noise, f cycling over
    {440, 1000, 2500, 6000} Hz, LCG of reference src/testutil.rs:110-121 seeded
    12345 + segment index; every 32nd segment all zeros (reference
    tests/integration_test.rs:52-54 silent segment)."""
    out = np.empty((n, sample_count), dtype=np.float32)
    t = np.arange(sample_count, dtype=np.float64) / sample_rate
    freqs = (440.0, 1000.0, 2500.0, 6000.0)
    for k in range(n):
        gi = first_index + k
        if gi % 32 == 31:
            out[k] = 0.0
            continue
        # LCG noise in [-1, 1].  state_{j+1} = a*state_j + c (mod 2^64); the j-step map is
        # s -> M_j*s + A_j, built by doubling so no per-sample Python loop is needed.
        a, c = np.uint64(1103515245), np.uint64(12345)
        s0 = np.uint64(12345 + gi)
        mult = np.empty(sample_count, dtype=np.uint64)
        add = np.empty(sample_count, dtype=np.uint64)
        with np.errstate(over="ignore"):
            mult[0], add[0] = a, c
            filled = 1
            while filled < sample_count:
                take = min(filled, sample_count - filled)
                mult[filled:filled + take] = mult[:take] * mult[filled - 1]
                add[filled:filled + take] = mult[:take] * add[filled - 1] + add[:take]
                filled += take
            state = mult * s0 + add
        bits = ((state >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.float64)
        noise = bits * (2.0 / 65535.0) - 1.0
        out[k] = (0.5 * np.sin(2.0 * np.pi * freqs[gi % 4] * t) + 0.05 * noise).astype(np.float32)
    return out


def meta_model(num_species: int = 6522, hidden: int = 64, seed: int = 7) -> bytes:
    """Synthetic range-filter meta model with the I/O contract of reference src/rangefilter.rs:451-496:
    input f32 [1, 3] = (latitude, longitude, week), ONE output f32 [1, num_species] of probabilities (the Rust
    side applies no sigmoid).  Topology is a guess (the real file is not available offline): scaled inputs ->
    dense + ReLU -> dense + ReLU -> dense + Sigmoid."""
    rng = np.random.RandomState(seed)
    g = GraphBuilder("birdnet_meta_synth")
    g.add_input("input", [1, 3])
    scale = g.const(np.array([1.0 / 90.0, 1.0 / 180.0, 1.0 / 48.0], dtype=np.float32))
    x = g.node("Mul", ["input", scale])
    w1 = g.const((rng.randn(3, hidden) * 1.5).astype(np.float32))
    x = g.node("Relu", [g.node("Add", [g.node("MatMul", [x, w1]), g.const((rng.randn(hidden) * 0.3).astype(np.float32))])])
    w2 = g.const((rng.randn(hidden, hidden) / np.sqrt(hidden)).astype(np.float32))
    x = g.node("Relu", [g.node("Add", [g.node("MatMul", [x, w2]), g.const((rng.randn(hidden) * 0.3).astype(np.float32))])])
    w3 = g.const((rng.randn(hidden, num_species) * (3.0 / np.sqrt(hidden))).astype(np.float32))
    y = g.node("Sigmoid", [g.node("Add", [g.node("MatMul", [x, w3]), g.const((rng.randn(num_species) * 1.5 - 2.0).astype(np.float32))])])
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [1, num_species])
    return g.serialize()
