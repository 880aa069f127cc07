#!/usr/bin/env python3
"""Where do the first steps of a bench run go?  Runs the bench loop (4 contexts, batch 32, device-resident inputs)
as `--warmup W --steps K` for several (W, K) and prints ms/step of each timed region plus the host timestamps of the
step completions of the (5, 20) case.   python tools/warmup_profile.py"""
import importlib
import os
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

B, S = 32, 144000
with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
    f.write(synth.birdnet_v24())
model = bn.Model(f.name)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctxs = [bn.Context(model, B) for _ in range(NS)]
bufs = [torch.from_numpy(synth.synthetic_segments(B, S, 48000, first_index=b * B)).cuda() for b in range(4)]
torch.cuda.synchronize()


def run(W, K, stamps=None):
    def step(i):
        if i >= NS:
            ctxs[(i - NS) % NS].synchronize()
            if stamps is not None:
                stamps.append(time.perf_counter())
        if os.environ.get("PLAN_ONLY"):
            ctxs[i % NS].infer_device(bufs[i % 4].data_ptr(), B, sync=False)
        else:
            ctxs[i % NS].step_device(bufs[i % 4].data_ptr(), B, 10, 0.1, sync=False)

    def drain(n):
        for j in range(max(0, n - NS), n):
            ctxs[j % NS].synchronize()
            if stamps is not None:
                stamps.append(time.perf_counter())

    keep, stamps_ = stamps, None
    stamps = None
    for i in range(W):
        step(i)
    drain(W)
    torch.cuda.synchronize()
    stamps = keep
    t0 = time.perf_counter()
    if stamps is not None:
        stamps.append(t0)
    for i in range(K):
        step(i)
    drain(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


PRIME = int(os.environ.get("PRIME", "0"))
for c_ in ctxs:
    for k_ in range(PRIME):
        c_.step_device(bufs[ctxs.index(c_) % 4].data_ptr(), B, 10, 0.1, sync=True)
if os.environ.get("IDLE"):
    time.sleep(float(os.environ["IDLE"]))
CONC = int(os.environ.get("CONC", "0"))  # rounds of all contexts at once, before anything is timed
for k_ in range(CONC):
    for j_, c_ in enumerate(ctxs):
        c_.step_device(bufs[j_ % 4].data_ptr(), B, 10, 0.1, sync=False)
    for c_ in ctxs:
        c_.synchronize()
if os.environ.get("BURN"):  # unrelated full-chip load for that many ms, then straight into the run
    a_ = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    while (time.perf_counter() - t_) * 1e3 < float(os.environ["BURN"]):
        (a_ @ a_)
        torch.cuda.synchronize()
if os.environ.get("FIRST"):
    st = []
    w_, k_ = (int(v) for v in os.environ["FIRST"].split(","))
    print(f"very first run W={w_} K={k_}: {run(w_, k_, st):.4f} ms/step")
    print("  completion stamps (ms after t0):", " ".join(f"{(t - st[0]) * 1e3:.2f}" for t in st[1:]))
    sys.exit(0)
st = []
print(f"first run after context creation  W=5 K=20: {run(5, 20, st):.4f} ms/step")
print("  completion stamps (ms after t0):", " ".join(f"{(t - st[0]) * 1e3:.2f}" for t in st[1:]))
for W, K in ((5, 20), (5, 20), (0, 20), (0, 4), (0, 8), (0, 40), (0, 100), (0, 200), (5, 20), (0, 20)):
    print(f"W={W:3d} K={K:3d}: {run(W, K):.4f} ms/step")
time.sleep(0.5)
print(f"after 0.5 s idle, W=0 K=20: {run(0, 20):.4f} ms/step")
print(f"again,            W=0 K=20: {run(0, 20):.4f} ms/step")
os.unlink(f.name)
