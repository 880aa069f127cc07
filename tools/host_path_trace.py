#!/usr/bin/env python3
"""Per-call host times of the pipelined host-slice path (bn_infer_submit / bn_infer_collect), four contexts x two
batches in flight, from a cold start: shows warm-up effects.   python tools/host_path_trace.py [steps]"""
import importlib
import os
import sys
import tempfile
import time
from collections import deque

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: F401,E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 160
B, S = 32, 144000
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(synth.birdnet_v24())
model = bn.Model(f.name)
ctxs = [bn.Context(model, B) for _ in range(4)]
xs = [synth.synthetic_segments(B, S, 48000, first_index=k * B) for k in range(4)]
qs = [deque() for _ in ctxs]
sub, col = [], []
t_all = time.perf_counter()
for i in range(steps):
    c, q = ctxs[i % 4], qs[i % 4]
    tc = 0.0
    if len(q) == 2:
        t0 = time.perf_counter()
        c.collect(q.popleft())
        tc = time.perf_counter() - t0
    t0 = time.perf_counter()
    q.append(c.submit(xs[i % 4], 10, 0.1))
    sub.append(time.perf_counter() - t0)
    col.append(tc)
for c, q in zip(ctxs, qs):
    while q:
        c.collect(q.popleft())
dt = time.perf_counter() - t_all
print(f"{steps} steps in {dt * 1e3:.1f} ms ({steps * B / dt:.0f} segments/s incl. cold start); cpus {os.cpu_count()} affinity {len(os.sched_getaffinity(0))}")
for lo in range(0, steps, 16):
    s_, c_ = sub[lo:lo + 16], col[lo:lo + 16]
    print(f"steps {lo:4d}-{lo + len(s_) - 1:4d}: submit mean {np.mean(s_) * 1e3:6.3f} ms max {np.max(s_) * 1e3:6.3f}   collect mean {np.mean(c_) * 1e3:6.3f} ms max {np.max(c_) * 1e3:6.3f}")
os.unlink(f.name)
