#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-facing entry points (not bench.py's `value`, which starts
with the batch resident in HBM):
  bn_infer          host f32 windows -> pinned staging -> H2D -> plan -> D2H of logits
  bn_infer_windows  i16 recording uploaded once (upload time included), windows cut on the device
One context, synchronous calls, batch 32, BirdNET-v2.4 synthetic model.   python tools/pcie_rate.py [n_batches]"""
import importlib
import os
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, S = 32, 144000
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(synth.birdnet_v24())
model = bn.Model(f.name)
ctx = bn.Context(model, B)
x = synth.synthetic_segments(B, S, 48000)
ctx.infer(x)
t0 = time.perf_counter()
for _ in range(nb):
    ctx.infer(x)
dt = time.perf_counter() - t0
print(f"bn_infer (host f32 windows):          {nb * B / dt:9.0f} segments/s  ({dt / nb * 1e3:.2f} ms per batch of {B}, {B * S * 4 / 1e6:.1f} MB H2D each)")

pcm = (np.clip(np.concatenate([x[i] for i in range(B)] * nb), -1, 1) * 32767).astype(np.int16)
t0 = time.perf_counter()
rec = bn.Recording(pcm)
up = time.perf_counter() - t0
G = rec.n_windows(S)
ctx.infer_windows(rec, S, 0, B)
t0 = time.perf_counter()
for first in range(0, G, B):
    ctx.infer_windows(rec, S, first, min(B, G - first))
dt = time.perf_counter() - t0
print(f"bn_infer_windows (i16 recording):     {G / (dt + up):9.0f} segments/s incl. upload ({up * 1e3:.1f} ms for {pcm.nbytes / 1e6:.0f} MB), "
      f"{G / dt:9.0f} segments/s after it ({dt / (G / B) * 1e3:.2f} ms per batch)")
os.unlink(f.name)
