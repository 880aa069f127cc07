#!/usr/bin/env python3
"""Per-launch MARGINAL cost of one batch: (t(4B) - t(B)) / 3 with HIP events around every launch of one context
(bn_ctx_time_kernels), next to the time of the launch alone at B.  The marginal figure is what a launch costs once the
chip is full -- the regime the concurrent contexts of the headline number run in.

    python tools/marginal_table.py [--batch 32] [--model v24|v30|perch] [--reps 6] [--shared]"""
import argparse
import importlib
import os
import sys
import tempfile

import torch  # noqa: F401  (first: one HIP runtime per process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")


def timed(model, batch, reps):
    ctx = bn.Context(model, batch)
    acc = None
    for _ in range(reps + 1):
        rows = ctx.time_kernels(batch)
        if acc is None:
            acc = [[r[0], 0.0, r[2], r[3]] for r in rows]  # first pass = warm-up
        else:
            for x, r in zip(acc, rows):
                x[1] += r[1] / reps
    ctx.close()
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="v24")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--shared", action="store_true", help="the launches' forms for a shared device (bn_set_sharing_mode)")
    a = ap.parse_args()
    blob = {"v24": synth.birdnet_v24, "v30": synth.birdnet_v30, "perch": synth.perch_v2}[a.model]()
    with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
        f.write(blob)
        path = f.name
    model = bn.Model(path)
    bn.set_sharing_mode(bn.SHARING_SHARED if a.shared else bn.SHARING_ALONE)
    one = timed(model, a.batch, a.reps)
    four = timed(model, 4 * a.batch, a.reps)
    os.unlink(path)
    assert len(one) == len(four), (len(one), len(four))
    rows = []
    for x, y in zip(one, four):
        assert x[0] == y[0], (x[0], y[0])
        rows.append((x[0], x[1], (y[1] - x[1]) / 3.0, x[2], x[3]))
    t1 = sum(r[1] for r in rows)
    tm = sum(r[2] for r in rows)
    print(f"{len(rows)} launches: {t1:.1f} us alone at batch {a.batch}, {tm:.1f} us marginal  ({a.batch / tm * 1e6:.0f} seg/s at the marginal cost)")
    print("   alone  marginal   TF/s(marg)  GB/s(marg)  launch")
    for nm, us, mg, macs, byts in rows:
        print(f"{us:8.1f} {mg:9.1f} {2 * macs / mg / 1e6 if mg > 0 else 0:11.2f} {byts / mg / 1e3 if mg > 0 else 0:11.1f}  {nm}")


if __name__ == "__main__":
    main()
