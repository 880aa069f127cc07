#!/usr/bin/env python3
"""Regenerates the committed golden fixtures.

* topk_kats.json      -- the inputs of the reference's own known-answer tests for the
                         post-processing path (reference src/postprocess.rs:101-331) with the facts
                         those tests assert (result length, first index, ordering).  Data only.
* v24_tiny_oracle.npz -- REGRESSION vectors (not reference output: no runnable reference exists,
                         SURVEY.md 8(c)): logits of the CPU oracle (oracle/onnx_ref.py, fp32) for a
                         seeded reduced-width synthetic BirdNET-v2.4-style model on three seeded
                         segments, plus the oracle's fp64 evaluation to bound its own rounding.
Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import torch
    from oracle import onnx_ref

    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    kats = [
        {"name": "test_top_k_predictions_basic", "logits": [0.1, 0.5, 0.9, 0.3, 0.7], "k": 3, "min": None, "len": 3, "first": 2},
        {"name": "test_top_k_with_min_confidence", "logits": [-5.0, 0.0, 5.0], "k": 10, "min": 0.4, "len": 2, "first": 2},
        {"name": "test_top_k_larger_than_input", "logits": [0.1, 0.2], "k": 100, "min": None, "len": 2, "first": 1},
        {"name": "test_top_k_zero_k", "logits": [0.1, 0.2, 0.3], "k": 0, "min": None, "len": 0, "first": None},
        {"name": "test_predictions_have_correct_indices", "logits": [0.1, 0.9, 0.5], "k": 3, "min": None, "len": 3, "first": 1},
        {"name": "test_top_k_all_equal_scores", "logits": [0.5, 0.5, 0.5, 0.5], "k": 2, "min": None, "len": 2, "first": None},
        {"name": "test_top_k_negative_logits", "logits": [-10.0, -5.0, -1.0, -20.0], "k": 2, "min": None, "len": 2, "first": 2},
        {"name": "test_min_confidence_zero", "logits": [-10.0, 0.0, 10.0], "k": 10, "min": 0.0, "len": 3, "first": 2},
        {"name": "test_min_confidence_one", "logits": [-10.0, 0.0, 10.0], "k": 10, "min": 1.0, "len": 0, "first": None},
        {"name": "test_top_k_max_usize", "logits": [0.1, 0.2, 0.3], "k": 2 ** 64 - 1, "min": None, "len": 3, "first": 2},
        {"name": "test_missing_labels", "logits": [0.1, 0.2, 0.3, 0.4], "k": 4, "min": None, "len": 4, "first": 3},
    ]
    with open(os.path.join(HERE, "topk_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)

    params = dict(num_species=64, seed=24, width=0.25, depth=0.25, head=64)
    data = synth.birdnet_v24(**params)
    x = synth.synthetic_segments(3, 144000, 48000)
    x[2] = 0.0
    g = onnx_ref.load_graph(data)
    y32 = onnx_ref.run_graph(g, x)["output"]
    y64 = onnx_ref.run_graph(g, x, dtype=torch.float64)["output"]
    np.savez_compressed(os.path.join(HERE, "v24_tiny_oracle.npz"), logits_fp32=y32, logits_fp64=y64.astype(np.float64),
                        params=json.dumps(params), segment_first_index=0)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
