#!/usr/bin/env python3
"""Per-launch table of one batch of the BirdNET-v2.4 plan: HIP-event time, MACs, bytes, TF/s, GB/s.

    python tools/kernel_table.py [--batch 32] [--model v24|v30|perch] [--reps 5] [--top 25]

Times come from bn_ctx_time_kernels (events around every launch on the context's stream)."""
import argparse
import importlib
import os
import sys
import tempfile

import torch  # noqa: F401  (first: one HIP runtime per process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="v24")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--top", type=int, default=200)
    a = ap.parse_args()
    blob = {"v24": synth.birdnet_v24, "v30": synth.birdnet_v30, "perch": synth.perch_v2}[a.model]()
    with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
        f.write(blob)
        path = f.name
    model = bn.Model(path)
    ctx = bn.Context(model, a.batch)
    acc = None
    for _ in range(a.reps + 1):
        rows = ctx.time_kernels(a.batch)
        if acc is None:
            acc = [[r[0], 0.0, r[2], r[3]] for r in rows]  # first pass = warm-up
        else:
            for x, r in zip(acc, rows):
                x[1] += r[1] / a.reps
    total = sum(x[1] for x in acc)
    print(f"{len(acc)} launches, {total:.1f} us per batch of {a.batch}  ({a.batch / total * 1e6:.0f} seg/s serial)")
    for nm, us, macs, byts in sorted(acc, key=lambda x: -x[1])[:a.top]:
        print(f"{us:9.1f} us  {2 * macs / us / 1e6 if us else 0:8.2f} TF/s  {byts / us / 1e3 if us else 0:8.1f} GB/s  {nm}")
    os.unlink(path)


if __name__ == "__main__":
    main()
