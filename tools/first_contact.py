#!/usr/bin/env python3
"""First contact with a real model file: everything a maintainer needs to know in one command.

    python tools/first_contact.py birdnet_v24.onnx [--golden v24_golden.npz] [--batch 4] [--plan]

1. the survey (no GPU needed): opset, I/O shapes, detect_model_type's verdict, every operator type with its node count and whether the
   lowering has a rule for it, and the planner's verdict with the refusing node and reason (bn_model_survey);
2. with --plan: the launch plan, one line per kernel launch (bn_plan_describe);
3. on a machine with a gfx950 device: the HIP path on the SURVEY 8(d) inputs against the CPU oracle (oracle/onnx_ref.py, the ONNX
   operator specification in torch) -- max error, tolerance excess, top-1 agreement;
4. with --golden (written by tools/dump_ort_golden.py where onnxruntime exists): the same against ONNX Runtime's CPU outputs, i.e. against
   the reference's numeric path (src/classifier.rs:637-639).  Exit code 0 = everything that could be checked agrees.

The reference loads whatever ONNX Runtime loads (src/classifier.rs:340-350); this is the native path's answer to the same file."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("--golden")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--plan", action="store_true")
    args = ap.parse_args()
    bn = importlib.import_module("rust-birdnet-onnx_amd")
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    status, text = bn.model_survey(args.model)
    print(text, end="")
    if status:
        print("=> the planner refuses this file; the REFUSED lines above name the node.  Nothing was run.")
        return 2
    if args.plan:
        print(bn.plan_describe(args.model))
    if bn.device_count() < 1:
        print("=> plans accepted.  No gfx950 device here: the numeric checks need one (the native path has no CPU fallback).")
        return 0
    m = bn.Model(args.model)
    cfg = m.config
    x = synth.synthetic_segments(args.batch, cfg.sample_count, cfg.sample_rate)
    ctx = bn.Context(m, args.batch)
    logits, emb = ctx.infer(x)
    rc = 0
    from oracle import onnx_ref
    ref = onnx_ref.run_model(open(args.model, "rb").read(), x)
    names = list(ref)
    for what, got, idx in (("logits", logits, cfg.logits_output), ("embeddings", emb, cfg.embedding_output)):
        if idx is None or idx < 0 or got is None:
            continue
        want = ref[names[idx]].reshape(got.shape)
        err = np.abs(got - want)
        excess = float((err - (2e-4 + 2e-4 * np.abs(want))).max())
        print(f"HIP vs oracle, {what} (output {idx} '{names[idx]}'): max |diff| {float(err.max()):.3g}, tolerance excess {excess:.3g}" +
              (f", top-1 equal: {bool(np.array_equal(got.argmax(1), want.argmax(1)))}" if what == "logits" else ""))
        rc |= int(excess > 0)
    if args.golden:
        import real_model
        prefix = {0: "BIRDNET_V24", 1: "BIRDNET_V30", 2: "PERCH_V2"}[int(cfg.model_type)]
        try:
            print("HIP vs ONNX Runtime golden:", real_model.compare(bn, args.model, real_model.load_golden(args.golden), prefix))
        except AssertionError as e:
            print("HIP vs ONNX Runtime golden: MISMATCH:", e)
            rc |= 1
    print("=> " + ("agrees" if rc == 0 else "DISAGREES") + " within 2e-4 + 2e-4 |x|")
    return rc


if __name__ == "__main__":
    sys.exit(main())
