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

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: F401,E402

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

# pipelined: bn_infer_submit / bn_infer_collect, two batches in flight on ONE context, then on four contexts
from collections import deque


def pipelined(ctxs, nb):
    qs = [deque() for _ in ctxs]
    for i in range(nb + 2 * len(ctxs)):
        if i == 2 * len(ctxs):
            t0 = time.perf_counter()
        q = qs[i % len(ctxs)]
        if len(q) == 2:
            ctxs[i % len(ctxs)].collect(q.popleft())
        q.append(ctxs[i % len(ctxs)].submit(x, 10, 0.1))
    for c, q in zip(ctxs, qs):
        while q:
            c.collect(q.popleft())
    return time.perf_counter() - t0


dt = pipelined([ctx], nb)
print(f"submit/collect, 1 context x 2 in flight: {nb * B / dt:9.0f} segments/s  ({dt / nb * 1e3:.2f} ms per batch, {nb * B * S * 4 / dt / 1e9:.1f} GB/s H2D)")
more = [ctx] + [bn.Context(model, B) for _ in range(3)]
dt = pipelined(more, 4 * nb)
print(f"submit/collect, 4 contexts x 2 in flight: {4 * nb * B / dt:9.0f} segments/s  ({dt / (4 * nb) * 1e3:.2f} ms per batch, {4 * nb * B * S * 4 / dt / 1e9:.1f} GB/s H2D)")
# raw link rate for reference: one pinned 64 MB buffer up and down
t = torch.empty(16 << 20, dtype=torch.float32).pin_memory()
d = torch.empty_like(t, device="cuda")
for _ in range(2):
    d.copy_(t, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d.copy_(t, non_blocking=True)
torch.cuda.synchronize()
print(f"pinned H2D link rate: {10 * t.numel() * 4 / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
del more

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
