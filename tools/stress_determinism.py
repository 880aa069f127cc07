#!/usr/bin/env python3
"""Stress: the same batch through four contexts in flight, many times; every result must equal the first one bit
for bit (guards the in-kernel squeeze-excite completion counters and anything else order-dependent).
    python tools/stress_determinism.py [iterations] [model v24|v30|perch] [batch]"""
import importlib
import os
import sys
import tempfile

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402,F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
family = sys.argv[2] if len(sys.argv) > 2 else "v24"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
make, S, sr = {"v24": (synth.birdnet_v24, 144000, 48000), "v30": (synth.birdnet_v30, 160000, 32000), "perch": (synth.perch_v2, 160000, 32000)}[family]
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(make())
model = bn.Model(f.name)
os.unlink(f.name)
ctxs = [bn.Context(model, B) for _ in range(4)]
x = torch.from_numpy(synth.synthetic_segments(B, S, sr)).cuda()
torch.cuda.synchronize()
ref = None
bad = 0
for it in range(iters):
    for c in ctxs:
        c.step_device(x.data_ptr(), B, 10, 0.05, sync=False)
    for k, c in enumerate(ctxs):
        c.synchronize()
        lg, ix, cf, ct = c.step_results(B)
        blob = lg.tobytes() + ix.tobytes() + cf.tobytes() + ct.tobytes()
        if ref is None:
            ref = blob
            assert np.isfinite(lg).all()
        elif blob != ref:
            bad += 1
            if bad < 5:
                d = np.frombuffer(blob[:lg.nbytes], np.float32) - np.frombuffer(ref[:lg.nbytes], np.float32)
                print(f"iteration {it} context {k}: mismatch, max |diff| {np.abs(d).max()}", flush=True)
print(f"{family} batch {B}: {iters} iterations x 4 contexts, mismatches: {bad}")
sys.exit(1 if bad else 0)
