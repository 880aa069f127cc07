#!/usr/bin/env python3
"""Run a few batches of the synthetic BirdNET-v2.4 plan WITHOUT hipGraph capture, so that rocprofv3
--pmc sees one dispatch per plan op.   rocprofv3 --pmc ... -d out -- python3 tools/pmc_run.py [batch] [iters]"""
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
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(synth.birdnet_v24())
model = bn.Model(f.name)
ctx = bn.Context(model, batch, flags=bn.BN_CTX_NO_GRAPH)
x = synth.synthetic_segments(batch, 144000, 48000)
for _ in range(iters):
    ctx.infer(x)
os.unlink(f.name)
