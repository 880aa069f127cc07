"""CPU side of the real-model gate (tests/test_gpu_real_model.py): the stand-alone input generator equals the package's, golden files are
validated, the environment gate skips silently, and the first-contact survey (bn_model_survey, no device needed) names every operator
type the lowering has no rule for and the node the planner refuses -- what a maintainer sees before any GPU run."""
import importlib
import os

import numpy as np
import pytest

import real_model
from gpu_helpers import op_graph, synth, write_model

bn = importlib.import_module("rust-birdnet-onnx_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gi():
    spec = importlib.util.spec_from_file_location("golden_inputs", os.path.join(ROOT, "tools", "golden_inputs.py"))
    gi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gi)
    return gi


def test_standalone_inputs_equal_the_package_generator():
    gi = _gi()
    for n, s, sr, first in ((3, 144000, 48000, 0), (2, 160000, 32000, 30)):
        assert gi.segments(n, s, sr, first).tobytes() == synth.synthetic_segments(n, s, sr, first).tobytes()


def test_gate_and_golden_validation(tmp_path, monkeypatch):
    for p in real_model.FAMILIES:
        monkeypatch.delenv(p + "_MODEL", raising=False)
        monkeypatch.delenv(p + "_GOLDEN", raising=False)
        assert real_model.configured(p) is None
    monkeypatch.setenv("PERCH_V2_MODEL", "/x.onnx")
    assert real_model.configured("PERCH_V2") is None            # the golden is missing: still skipped
    monkeypatch.setenv("PERCH_V2_GOLDEN", "/g.npz")
    assert real_model.configured("PERCH_V2") == ("/x.onnx", "/g.npz")
    np.savez(str(tmp_path / "junk.npz"), a=np.zeros(3))
    with pytest.raises(ValueError):
        real_model.load_golden(str(tmp_path / "junk.npz"))
    np.savez(str(tmp_path / "ok.npz"), inputs=np.zeros((2, 10), np.float32), output_names=np.array(["y"]), output_0=np.zeros((2, 4), np.float32))
    g = real_model.load_golden(str(tmp_path / "ok.npz"))
    assert g["names"] == ["y"] and g["outputs"][0].shape == (2, 4)
    np.savez(str(tmp_path / "rows.npz"), inputs=np.zeros((2, 10), np.float32), output_names=np.array(["y"]), output_0=np.zeros((3, 4), np.float32))
    with pytest.raises(ValueError):
        real_model.load_golden(str(tmp_path / "rows.npz"))
    src = open(os.path.join(ROOT, "tools", "dump_ort_golden.py")).read()
    assert len(src.splitlines()) <= 30 and "CPUExecutionProvider" in src and "golden_inputs" in src


def test_first_contact_survey_names_unmapped_operators_and_the_refusing_node():
    status, text = bn.model_survey(write_model(synth.birdnet_v24(num_species=300, width=0.25, depth=0.25, head=128)))
    assert status == 0 and "unmapped: 0 operator types" in text and "plan (logits + embeddings): OK" in text and "detected model_type=0" in text, text

    def build(g, x):  # an operator outside the subset (Resize) and one inside it whose ATTRIBUTES are refused (a non-affine Gather)
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Reshape", [g.node("Slice", [x, i64(0), i64(4000), i64(1), i64(1)]), i64(0, 1, 40, 100)])
        y = g.node("Resize", [x, "", g.const(np.array([1, 1, 2, 2], dtype=np.float32))], mode="nearest")
        return g.node("Gather", [y, g.const(np.array([0, 1, 3, 7], dtype=np.int64))], axis=2)
    status, text = bn.model_survey(write_model(op_graph(build, [1, 4, 200])))
    assert status == bn.BN_ERR_UNSUPPORTED_MODEL, text
    assert "op Resize" in text and "NOT MAPPED" in text and "unmapped: 1 operator types, 1 nodes" in text, text
    assert "REFUSED: node 'Resize_" in text, text  # (input [?, 144000] with one output still reads as a v2.4 by detection.rs's rules)
    with pytest.raises(bn.EngineError):
        bn.model_survey("/nonexistent/model.onnx")


def test_every_mapped_operator_type_really_has_a_rule():
    """op_type_mapped's list against the lowering: a one-node graph of each listed type must not be refused as "outside the native
    subset" (it may be refused for missing inputs -- that is a different message)."""
    import re
    src = open(os.path.join(ROOT, "rust-birdnet-onnx_amd", "csrc", "engine.cpp")).read()
    listed = re.findall(r'"(\w+)"', src[src.index("bool op_type_mapped"):src.index("IoMeta read_io_meta")])
    assert len(listed) > 70
    for t in listed:
        def build(g, x, t=t):
            return g.node(t, [x])
        try:
            bn.plan_describe(write_model(op_graph(build, [144000])), model_type=100)
            msg = ""
        except bn.EngineError:
            msg = bn.last_error()
        assert "operator is outside the native subset" not in msg, (t, msg)
