"""Independent formulations of the three synthetic model families with torch.stft / torch.nn.functional, built from the
RAW weights and the front-end PARAMETERS -- nothing here reads an ONNX file, and nothing goes through the ONNX writer
or either ONNX reader.  tests/test_oracle_independent.py checks oracle/onnx_ref.py (the restatement of the ONNX operator
specification that every GPU parity test leans on) against these: a shared misreading of ONNX semantics by the writer,
the product's planner and the oracle (pads order, BatchNormalization epsilon, Gemm transB, Slice with negative steps,
Transpose perms, Concat axes, keepdims, ...) would show up here as a disagreement.

This does NOT pin parity to the reference's real model files (nothing can): the topologies are the
hypothesised ones of rust-birdnet-onnx_amd/synth.py."""
import importlib
import math

import numpy as np
import torch
import torch.nn.functional as F

synth = importlib.import_module("rust-birdnet-onnx_amd.synth")


class Weights:
    """The float weight tensors of a GraphBuilder in creation order, front-end constants excluded."""

    def __init__(self, builder, dtype):
        self.items = [torch.from_numpy(np.array(a)).to(dtype) for _, hint, a in builder.const_values
                      if a.dtype == np.float32 and a.ndim >= 1 and hint not in ("dft", "mel")]
        self.pos = 0

    def take(self, *shape):
        t = self.items[self.pos]
        self.pos += 1
        assert tuple(t.shape) == tuple(shape), (self.pos - 1, tuple(t.shape), shape)
        return t

    def done(self):
        return self.pos == len(self.items)


def _act(x, kind):
    if kind == "relu":
        return F.relu(x)
    if kind == "relu6":
        return torch.clamp(x, 0.0, 6.0)
    return F.silu(x)


def _conv_bn(x, w, cin, cout, k, stride, groups, act, spec):
    wt = w.take(cout, cin // groups, k, k)
    if spec.bn_nodes:
        gamma, beta, mean, var = w.take(cout), w.take(cout), w.take(cout), w.take(cout)
        y = F.conv2d(x, wt, None, stride=stride, padding=k // 2, groups=groups)
        y = F.batch_norm(y, mean, var, gamma, beta, training=False, eps=1e-3)
    else:
        y = F.conv2d(x, wt, w.take(cout), stride=stride, padding=k // 2, groups=groups)
    return _act(y, spec.act) if act else y


def _se(x, w, c, cred, spec):
    s = F.adaptive_avg_pool2d(x, 1)
    s = _act(F.conv2d(s, w.take(cred, c, 1, 1), w.take(cred)), spec.act)
    s = torch.sigmoid(F.conv2d(s, w.take(c, cred, 1, 1), w.take(c)))
    return x * s


def backbone(x, w, cin, spec):
    x = _conv_bn(x, w, cin, spec.stem, 3, 2, 1, True, spec)
    c = spec.stem
    for expand, cout, repeats, stride, k in spec.stages:
        for r in range(repeats):
            inp, st = x, (stride if r == 0 else 1)
            mid = c * expand
            if expand != 1:
                x = _conv_bn(x, w, c, mid, 1, 1, 1, True, spec)
            x = _conv_bn(x, w, mid, mid, k, st, mid, True, spec)
            if spec.se:
                x = _se(x, w, mid, max(1, c // 4), spec)
            x = _conv_bn(x, w, mid, cout, 1, 1, 1, False, spec)
            if st == 1 and c == cout:
                x = x + inp
            c = cout
    return _conv_bn(x, w, c, spec.head, 1, 1, 1, True, spec), spec.head


def _stft(x, n_fft, hop):
    win = torch.hann_window(n_fft, periodic=True, dtype=x.dtype)
    return torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=win, center=False, return_complex=True)  # [B, bins, frames]


def _mel(n_fft, n_mels, sr, fmin, fmax, dtype):
    return torch.from_numpy(synth.mel_filterbank(n_fft // 2 + 1, n_mels, sr, fmin, fmax)).to(dtype)  # [bins, mels]


def birdnet_v24(x, builder, dtype, width=1.0, depth=1.0, head=1024, num_species=6522):
    x = torch.from_numpy(x).to(dtype)
    mn = x.amin(dim=1, keepdim=True)
    x1 = x - mn
    x = ((x1 / (x1.amax(dim=1, keepdim=True) + 1e-6)) - 0.5) * 2.0
    p = 1.0 / (1.0 + math.exp(1.23))
    planes = []
    for n_fft, hop, fmin, fmax in ((2048, 278, 0.0, 3000.0), (1024, 280, 500.0, 15000.0)):
        re = _stft(x, n_fft, hop).real.transpose(1, 2)                     # [B, frames, bins]
        m = re @ _mel(n_fft, 96, 48000, fmin, fmax, dtype)                  # [B, frames, mels]
        q = torch.pow(torch.pow(m, 2.0), p)
        planes.append(torch.flip(q, dims=[2]).transpose(1, 2).unsqueeze(1))  # [B, 1, mels (high first), frames]
    img = torch.cat(planes, dim=1)
    w = Weights(builder, dtype)
    gamma, beta, mean, var = w.take(2), w.take(2), w.take(2), w.take(2)
    img = F.batch_norm(img, mean, var, gamma, beta, training=False, eps=1e-3)
    spec = synth.BackboneSpec(stem=synth._round8(32 * width), stages=synth.scaled_stages(synth.B0_STAGES, width, depth), head=head,
                              act="relu", se=True, bn_nodes=True)
    f, c = backbone(img, w, 2, spec)
    pooled = f.mean(dim=(2, 3))
    out = F.linear(pooled, w.take(num_species, c), w.take(num_species))
    assert w.done()
    return {"output": out.numpy(), "_image": img.numpy()}


def birdnet_v30(x, builder, dtype, width=1.0, depth=1.0, emb=1024, num_species=1000):
    x = torch.from_numpy(x).to(dtype)
    mag = _stft(x, 1024, 320).abs().transpose(1, 2)                         # [B, frames, bins]
    lg = torch.log(mag @ _mel(1024, 128, 32000, 40.0, 15000.0, dtype) + 1e-3)
    img = lg.transpose(1, 2).unsqueeze(1)
    w = Weights(builder, dtype)
    spec = synth.BackboneSpec(stem=synth._round8(32 * width), stages=synth.scaled_stages(synth.B0_STAGES, width, depth), head=emb,
                              act="silu", se=True, bn_nodes=False)
    f, c = backbone(img, w, 1, spec)
    e = f.mean(dim=(2, 3))
    logits = e @ w.take(c, num_species) + w.take(num_species)              # MatMul style: the constant is stored [in, out]
    assert w.done()
    return {"output_0": e.numpy(), "output_1": logits.numpy(), "_image": img.numpy()}


def perch_v2(x, builder, dtype, width=1.0, depth=1.0, emb=1536, num_species=14795):
    x = torch.from_numpy(x).to(dtype)
    z = _stft(F.pad(x, (160, 160)), 640, 320)                               # SAME padding of the strided conv: (640 - 320) / 2 per side
    pw = (z.real ** 2 + z.imag ** 2).transpose(1, 2)                        # [B, 500, bins]
    m = pw @ _mel(640, 128, 32000, 60.0, 16000.0, dtype)
    spectro = torch.log(torch.clamp(m, min=1e-5)) * 0.1                     # [B, 500, 128]
    img = spectro.unsqueeze(1)
    w = Weights(builder, dtype)
    spec = synth.BackboneSpec(stem=synth._round8(40 * width), stages=synth.scaled_stages(synth.B3_STAGES, width, depth), head=emb,
                              act="silu", se=True, bn_nodes=True)
    f, c = backbone(img, w, 1, spec)
    spatial = f.permute(0, 2, 3, 1)
    e = spatial.mean(dim=(1, 2))
    label = F.linear(e, w.take(num_species, c), w.take(num_species))
    assert w.done()
    return {"embedding": e.numpy(), "spatial_embedding": spatial.numpy(), "spectrogram": spectro.numpy(), "label": label.numpy()}
