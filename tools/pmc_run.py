#!/usr/bin/env python3
"""Run a few batches of a synthetic plan (BirdNET v2.4 by default) WITHOUT hipGraph capture, so that rocprofv3
--pmc sees one dispatch per plan op.   rocprofv3 --pmc ... -d out -- python3 tools/pmc_run.py [batch] [iters] [v24|v30|perch]"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = sys.argv[3] if len(sys.argv) > 3 else "v24"
make, S, SR = {"v24": (synth.birdnet_v24, 144000, 48000), "v30": (synth.birdnet_v30, 160000, 32000), "perch": (synth.perch_v2, 160000, 32000)}[which]
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(make())
model = bn.Model(f.name)
ctx = bn.Context(model, batch, flags=bn.BN_CTX_NO_GRAPH)
x = synth.synthetic_segments(batch, S, SR)
for _ in range(iters):
    ctx.infer(x)
os.unlink(f.name)
