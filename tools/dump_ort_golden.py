#!/usr/bin/env python3
"""Writes the golden file that pins the HIP path to the REFERENCE's numeric path: ONNX Runtime on CPU (what Classifier::predict runs,
src/classifier.rs:637-639) on the SURVEY 8(d) inputs.  Run where onnxruntime and the model file exist -- never on the GPU box:

    python tools/dump_ort_golden.py birdnet_v24.onnx v24_golden.npz [--batch 4]
    BIRDNET_V24_MODEL=birdnet_v24.onnx BIRDNET_V24_GOLDEN=v24_golden.npz python -m pytest tests/test_gpu_real_model.py -m gpu

npz: inputs [B, S] f32, output_<i> for every graph output, output_names, ort_version."""
import importlib.util, os, sys
import numpy as np
import onnxruntime as ort

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("golden_inputs", os.path.join(here, "golden_inputs.py"))
gi = importlib.util.module_from_spec(spec); spec.loader.exec_module(gi)
model, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
sess = ort.InferenceSession(model, providers=["CPUExecutionProvider"])
inp = sess.get_inputs()[0]
samples = int(inp.shape[-1])
x = gi.segments(batch, samples, 48000 if samples == 144000 else 32000)
feed = x.reshape([batch] + [1] * (len(inp.shape) - 2) + [samples])      # rank-2 or rank-3 input (detection.rs:149-158)
names = [o.name for o in sess.get_outputs()]
vals = sess.run(names, {inp.name: feed})
np.savez_compressed(out, inputs=x, output_names=np.array(names), ort_version=np.array(ort.__version__),
                    **{f"output_{i}": np.asarray(v, dtype=np.float32) for i, v in enumerate(vals)})
print(f"{out}: {batch} x {samples} samples, outputs {[(n, v.shape) for n, v in zip(names, vals)]}")
