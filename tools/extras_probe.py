import importlib, os, sys, tempfile, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
bn = importlib.import_module("rust-birdnet-onnx_amd"); synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

def load(fn):
    p = tempfile.mktemp(suffix=".onnx"); open(p, "wb").write(fn()); m = bn.Model(p); os.unlink(p); return m

def run(m, bsz, S, SR, nst=60, nwu=8, S_=4, own=True):
    cs = [bn.Context(m, bsz) for _ in range(S_)]
    x = torch.from_numpy(synth.synthetic_segments(bsz, S, SR)).cuda()
    ptrs = []
    class DB:
        def __init__(s, ptr, shape): s.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f4", "version": 2}
    for c in cs:
        if own:
            ptr, cap = c.input_device(); torch.as_tensor(DB(ptr, (bsz, S)), device="cuda").copy_(x); ptrs.append(ptr)
        else:
            ptrs.append(x.data_ptr())
    torch.cuda.synchronize()
    def go(n):
        for i in range(n):
            if i >= S_: cs[(i - S_) % S_].synchronize()
            cs[i % S_].step_device(ptrs[i % S_], bsz, 10, 0.1, sync=False)
        for c in cs: c.synchronize()
    go(nwu); torch.cuda.synchronize(); t = time.perf_counter(); go(nst); torch.cuda.synchronize(); d = time.perf_counter() - t
    return nst * bsz / d

m30 = load(synth.birdnet_v30)
print("A fresh v30:", round(run(m30, 64, 160000, 32000)))
print("A2 again v30:", round(run(m30, 64, 160000, 32000)))
m24 = load(synth.birdnet_v24)
print("v24:", round(run(m24, 32, 144000, 48000, nst=200, nwu=20)))
print("B v30 after v24 contexts:", round(run(m30, 64, 160000, 32000)))
c = bn.Context(m24, 32); x = synth.synthetic_segments(32, 144000, 48000)
for _ in range(30): c.collect(c.submit(x, 10, 0.1))
del c
print("C v30 after host leg:", round(run(m30, 64, 160000, 32000)))
big = bn.Context(m24, 128); big.infer(np.concatenate([x] * 4)); big.time_kernels(128); del big
print("D v30 after time_kernels:", round(run(m30, 64, 160000, 32000)))
print("E v30 foreign ptr:", round(run(m30, 64, 160000, 32000, own=False)))

# ---- idle contexts kept alive: do they slow the active ones?
keep = [bn.Context(m24, 32) for _ in range(4)]
xk = synth.synthetic_segments(32, 144000, 48000)
for c_ in keep:
    c_.infer(xk)
print("F v30 with 4 idle v24 contexts alive:", round(run(m30, 64, 160000, 32000)))
print("F2 v24 with 4 idle v24 contexts alive:", round(run(m24, 32, 144000, 48000, nst=200, nwu=20)))
del keep
import gc; gc.collect()
print("G v30 after freeing them:", round(run(m30, 64, 160000, 32000)))
