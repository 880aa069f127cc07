"""Shared by tests/test_gpu_real_model.py and tools/first_contact.py: a real model file + a golden file of ONNX Runtime CPU outputs
(tools/dump_ort_golden.py) against the HIP path.  This is how "parity unpinned" becomes "pinned" for the network:
the reference's own numeric path is ort's Session::run on CPU (src/classifier.rs:637-639, 721-723), and its integration tests are gated on
model files the same way (tests/integration_test.rs:73-122, 277-392)."""
from __future__ import annotations

import os

import numpy as np

FAMILIES = {  # env prefix -> (samples, outputs the reference reads: logits index, embedding index) per detection.rs:26-71 / classifier.rs:917-934
    "BIRDNET_V24": (144000, 0, None),
    "BIRDNET_V30": (160000, 1, 0),
    "PERCH_V2": (160000, 3, 0),
}
ATOL, RTOL = 2e-4, 2e-4  # the suite's network tolerance (tests/gpu_helpers.py)


def configured(prefix: str):
    """(model path, golden path) when both <PREFIX>_MODEL and <PREFIX>_GOLDEN are set, else None."""
    m, g = os.environ.get(prefix + "_MODEL"), os.environ.get(prefix + "_GOLDEN")
    return (m, g) if m and g else None


def load_golden(path: str) -> dict:
    z = np.load(path, allow_pickle=False)
    need = {"inputs", "output_names"}
    if not need <= set(z.files):
        raise ValueError(f"{path}: not a golden file of tools/dump_ort_golden.py (missing {sorted(need - set(z.files))})")
    names = [str(n) for n in z["output_names"]]
    outs = [np.asarray(z[f"output_{i}"], dtype=np.float32) for i in range(len(names))]
    x = np.asarray(z["inputs"], dtype=np.float32)
    if x.ndim != 2 or any(o.shape[0] != x.shape[0] for o in outs):
        raise ValueError(f"{path}: inputs must be [B, S] and every output must have B rows")
    return {"inputs": x, "names": names, "outputs": outs, "source": str(z["ort_version"]) if "ort_version" in z.files else "?"}


def compare(bn, model_path: str, golden: dict, prefix: str) -> dict:
    """Loads the model through Classifier::builder (labels synthesised to the model's own species count), runs the golden inputs and
    compares logits / embeddings with the golden outputs at the suite's tolerance; top-1 must be identical.  Returns the error figures;
    raises AssertionError on a mismatch."""
    samples, li, ei = FAMILIES[prefix]
    x = golden["inputs"]
    assert x.shape[1] == samples, f"{prefix}: golden inputs have {x.shape[1]} samples, the family has {samples}"
    m = bn.Model(model_path)
    cfg = m.config
    clf = bn.Classifier.builder().model_path(model_path).labels([f"Species_{i}" for i in range(cfg.num_species)]).top_k(5).with_rocm().build()
    res = clf.predict_batch(list(x))
    ref_l = golden["outputs"][cfg.logits_output].reshape(x.shape[0], -1)
    report = {"batch": int(x.shape[0]), "golden_source": golden["source"], "logits_output": int(cfg.logits_output)}
    got_l = np.stack([r.raw_scores for r in res])
    err = np.abs(got_l - ref_l)
    report["logits_max_abs_err"] = float(err.max())
    report["logits_worst_excess"] = float((err - (ATOL + RTOL * np.abs(ref_l))).max())
    report["top1_equal"] = bool(np.array_equal(got_l.argmax(1), ref_l.argmax(1)))
    assert cfg.logits_output == li, f"{prefix}: logits expected at output {li}, the library reads output {cfg.logits_output}"
    assert report["logits_worst_excess"] <= 0, f"{prefix}: logits differ from ONNX Runtime by {report['logits_max_abs_err']} (tolerance {ATOL} + {RTOL} |x|)"
    assert report["top1_equal"], f"{prefix}: top-1 differs from ONNX Runtime"
    for r, want in zip(res, ref_l):  # top-K rows are the reference's top_k_predictions of the golden logits wherever the logits agree in order
        assert r.predictions[0].index == int(np.argmax(want))
    if ei is not None:
        ref_e = golden["outputs"][cfg.embedding_output].reshape(x.shape[0], -1)
        got_e = np.stack([r.embeddings for r in res])
        e = np.abs(got_e - ref_e)
        report["embeddings_max_abs_err"] = float(e.max())
        assert (e <= ATOL + RTOL * np.abs(ref_e)).all(), f"{prefix}: embeddings differ from ONNX Runtime by {float(e.max())}"
    return report
