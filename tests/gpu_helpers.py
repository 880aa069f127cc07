"""Helpers shared by the GPU parity tests."""
import importlib
import os
import tempfile

import numpy as np

synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
writer = importlib.import_module("rust-birdnet-onnx_amd.onnx_writer")

# fp32 tolerance of the network path (north_star: "within a stated fp32 tolerance"):
# |gpu - oracle| <= ATOL + RTOL * |oracle|, plus identical top-1.  The oracle's own fp32-vs-fp64
# error on these graphs is ~5e-7, the summation orders differ (MFMA k-interleaving, fused BN).
ATOL, RTOL = 2e-4, 2e-4


def write_model(data: bytes) -> str:
    f = tempfile.NamedTemporaryFile(suffix=".onnx", delete=False)
    f.write(data)
    f.close()
    return f.name


def assert_close(got, want, what="", atol=ATOL, rtol=RTOL):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    lim = atol + rtol * np.abs(want.astype(np.float64))
    bad = err > lim
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} outside tolerance, max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} (want {want.flat[err.argmax()]:.6g})"
    return float(err.max())


def op_graph(build, out_shape, in_reshape=None):
    """Single-purpose graph: input [B,144000] -> optional Reshape -> build(g, x) -> output."""
    g = writer.GraphBuilder()
    g.add_input("input", [None, 144000])
    x = "input"
    if in_reshape is not None:
        x = g.node("Reshape", [x, g.const(np.array([-1] + list(in_reshape), dtype=np.int64))])
    y = build(g, x)
    g.node("Identity", [y], outputs=["output"])
    g.add_output("output", [None] + list(out_shape))
    return g.serialize()
