"""Real model files against ONNX Runtime's CPU outputs -- the only route from "parity unpinned" to "pinned" for the network
(VERDICT r4 item 2).  Mirrors the reference's own gating (tests/integration_test.rs:73-122 fixtures, :277-392 PERCH_V2_MODEL): set

    BIRDNET_V24_MODEL=/path/birdnet_v24.onnx  BIRDNET_V24_GOLDEN=/path/v24_golden.npz      (same for BIRDNET_V30_*, PERCH_V2_*)

with the golden written by tools/dump_ort_golden.py where onnxruntime exists; unset, the tests skip silently.  When the model is set but
the planner refuses it, the failure message is the full first-contact survey (every operator type, mapped or not, and the refusing node).
The plumbing itself is proven on every run by the last test: a synthetic model with a golden written by the oracle through the same files."""
import importlib
import os

import numpy as np
import pytest

import real_model
from gpu_helpers import synth, write_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prefix", sorted(real_model.FAMILIES))
def test_real_model_against_onnxruntime_golden(bn, prefix):
    conf = real_model.configured(prefix)
    if conf is None:
        pytest.skip(f"{prefix}_MODEL / {prefix}_GOLDEN not set")
    model_path, golden_path = conf
    status, survey = bn.model_survey(model_path)
    assert status == 0, "the planner refuses this model:\n" + survey
    report = real_model.compare(bn, model_path, real_model.load_golden(golden_path), prefix)
    print(prefix, report)


def test_the_gate_itself_with_a_synthetic_model_and_an_oracle_golden(bn, tmp_path, monkeypatch):
    """Same files, same code path as a maintainer's run, with the oracle standing in for onnxruntime: the golden is written in
    dump_ort_golden's format from tools/golden_inputs.py's segments (which must equal the package's generator bit for bit)."""
    from oracle import onnx_ref
    spec = importlib.util.spec_from_file_location("golden_inputs", os.path.join(os.path.dirname(__file__), "..", "tools", "golden_inputs.py"))
    gi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gi)
    data = synth.birdnet_v30(num_species=300, width=0.35, depth=0.35)
    path = write_model(data)
    x = gi.segments(3, 160000, 32000)
    assert x.tobytes() == synth.synthetic_segments(3, 160000, 32000).tobytes()
    ref = onnx_ref.run_model(data, x)
    names = list(ref)
    gpath = str(tmp_path / "golden.npz")
    np.savez_compressed(gpath, inputs=x, output_names=np.array(names), ort_version=np.array("oracle stand-in"),
                        **{f"output_{i}": ref[n] for i, n in enumerate(names)})
    monkeypatch.setenv("BIRDNET_V30_MODEL", path)
    monkeypatch.setenv("BIRDNET_V30_GOLDEN", gpath)
    assert real_model.configured("BIRDNET_V30") == (path, gpath) and real_model.configured("PERCH_V2") is None
    rep = real_model.compare(bn, path, real_model.load_golden(gpath), "BIRDNET_V30")
    assert rep["top1_equal"] and rep["logits_worst_excess"] <= 0 and "embeddings_max_abs_err" in rep
    # a golden that does NOT belong to the model must fail, not pass by accident
    bad = {k: v for k, v in np.load(gpath).items()}
    bad["output_1"] = bad["output_1"] + np.float32(0.01)
    np.savez_compressed(str(tmp_path / "bad.npz"), **bad)
    with pytest.raises(AssertionError):
        real_model.compare(bn, path, real_model.load_golden(str(tmp_path / "bad.npz")), "BIRDNET_V30")
