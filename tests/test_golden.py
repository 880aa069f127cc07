"""Committed fixtures: the reference's post-processing KAT inputs (CPU: oracle; GPU: HIP kernel) and
regression vectors of the network oracle (CPU: the oracle still reproduces them; GPU: the HIP path
matches them within the stated tolerance)."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")


def kats():
    return json.load(open(os.path.join(HERE, "topk_kats.json")))


@pytest.mark.parametrize("kat", kats(), ids=lambda k: k["name"])
def test_oracle_on_reference_kat_inputs(kat):
    got = oracle.top_k(np.asarray(kat["logits"], np.float32), kat["k"], kat["min"])
    assert len(got) == kat["len"]
    if kat["first"] is not None:
        assert got[0][0] == kat["first"]
    assert all(a[1] >= b[1] for a, b in zip(got, got[1:]))


@pytest.mark.gpu
@pytest.mark.parametrize("kat", kats(), ids=lambda k: k["name"])
def test_hip_on_reference_kat_inputs(bn, kat):
    idx, conf, cnt = bn.topk_host(np.asarray(kat["logits"], np.float32), kat["k"], kat["min"])
    assert cnt[0] == kat["len"]
    if kat["first"] is not None:
        assert idx[0, 0] == kat["first"]
    assert all(conf[0, j] >= conf[0, j + 1] for j in range(int(cnt[0]) - 1))


def _tiny():
    z = np.load(os.path.join(HERE, "v24_tiny_oracle.npz"))
    params = json.loads(str(z["params"]))
    x = synth.synthetic_segments(3, 144000, 48000)
    x[2] = 0.0
    return z, params, x


def test_network_oracle_reproduces_its_regression_vectors():
    from oracle import onnx_ref
    z, params, x = _tiny()
    y = onnx_ref.run_model(synth.birdnet_v24(**params), x)["output"]
    # torch CPU kernels may differ in summation order between builds/hosts: compare to the fp64 truth
    assert np.abs(y - z["logits_fp64"]).max() < 2e-5
    assert np.abs(z["logits_fp32"] - z["logits_fp64"]).max() < 2e-5


@pytest.mark.gpu
def test_hip_matches_golden_logits(bn, tmp_path):
    from gpu_helpers import assert_close
    z, params, x = _tiny()
    p = tmp_path / "tiny.onnx"
    p.write_bytes(synth.birdnet_v24(**params))
    logits, _ = bn.Context(bn.Model(str(p)), 4).infer(x)
    assert_close(logits, z["logits_fp32"], "golden logits")
    assert np.array_equal(np.argmax(logits, 1), np.argmax(z["logits_fp64"], 1))
