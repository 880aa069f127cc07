import importlib, os, sys, tempfile
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, numpy as np
sys.path.insert(0, os.getcwd())
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(synth.birdnet_v24(num_species=500, width=0.5, depth=0.5, head=128))
x = synth.synthetic_segments(4, 144000, 48000)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
base = None
for it in range(60):
    m = bn.Model(f.name)
    c = bn.Context(m, 4)
    c.infer(x)
    t = c.submit(x, 5, 0.1); c.collect(t)
    c2 = bn.Context(m, 2)
    d = torch.from_numpy(x[:2]).cuda()
    c2.step_device(d.data_ptr(), 2, 5, 0.1, sync=True)
    del c, c2, m, d
    if it == 4: base = free()
    if it in (4, 20, 40, 59): print(it, f"free {free():.0f} MiB")
print("leak MiB over 55 iterations:", round(base - free(), 1))
