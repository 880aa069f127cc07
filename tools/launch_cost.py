import importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
bn = importlib.import_module("rust-birdnet-onnx_amd"); synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
p = tempfile.mktemp(suffix=".onnx"); open(p, "wb").write(synth.birdnet_v24()); m = bn.Model(p)
cs = [bn.Context(m, 32) for _ in range(4)]
x = torch.from_numpy(synth.synthetic_segments(32, 144000, 48000)).cuda()
ptrs = []
class DB:
    def __init__(s, ptr, shape): s.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f4", "version": 2}
for c in cs:
    ptr, cap = c.input_device(); torch.as_tensor(DB(ptr, (32, 144000)), device="cuda").copy_(x); ptrs.append(ptr)
torch.cuda.synchronize()
for r in range(5):
    for i, c in enumerate(cs): c.step_device(ptrs[i], 32, 10, 0.1, sync=False)
    for c in cs: c.synchronize()
ts = []
for r in range(20):
    for c in cs: c.synchronize()
    t0 = time.perf_counter()
    for i, c in enumerate(cs): c.step_device(ptrs[i], 32, 10, 0.1, sync=False)
    t1 = time.perf_counter()
    for c in cs: c.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e6 / 4, (t2 - t0) * 1e6))
print("host us per step_device call (4 back to back after idle):", round(np.median([a for a, b in ts]), 1), " round us:", round(np.median([b for a, b in ts]), 1))
