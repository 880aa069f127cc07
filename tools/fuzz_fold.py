"""Randomised differential check of the folded framing convs (GPU): random filter lengths, hops, segment lengths, channel
blocks (cos / sin / both), bias, batch, sometimes a constant product (a random "mel" matrix) behind the bank -- against the oracle; the
LDS-resident half-fold kernel against the generic folded GEMM bit for bit where its K order is the same (one K slice); the quarter fold
(cosine-only banks) and the merged filters against the half fold / the unmerged plan within the tolerance.
    python tools/fuzz_fold.py [cases]        (60 cases by default, well under a minute)"""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
bn = importlib.import_module("rust-birdnet-onnx_amd")
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
from gpu_helpers import op_graph, write_model
from oracle import onnx_ref
rng = np.random.default_rng(123)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    L = int(rng.choice([128, 192, 256, 320, 512, 640, 1024, 2048]))
    hop = int(rng.integers(1, 700))
    Slen = int(rng.integers(L + hop * 3, 144001))
    frames = (Slen - L) // hop + 1
    if frames * 1 > 6000:
        Slen = L + hop * 5999; frames = 6000
    w = synth.dft_basis(L, "complex")
    half = w.shape[0] // 2
    lo = int(rng.integers(0, half - 2)); n1 = int(rng.integers(1, min(half - lo, 200)))
    kind = rng.choice(["cos", "sin", "both"])
    parts = []
    if kind in ("cos", "both"): parts.append(w[lo:lo + n1])
    if kind in ("sin", "both"): parts.append(w[half + lo:half + lo + n1])
    ww = np.ascontiguousarray(np.concatenate(parts, axis=0))
    bias = rng.standard_normal(ww.shape[0]).astype(np.float32) if rng.random() < 0.5 else None
    nmel = int(rng.integers(8, 120)) if rng.random() < 0.35 else 0  # a constant product behind the bank (the planner may merge the two)
    mel = (rng.random((ww.shape[0], nmel)) * (rng.random((ww.shape[0], nmel)) < 0.2)).astype(np.float32) if nmel else None
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        if Slen != 144000:
            x = g.node("Slice", [x, i64(0), i64(Slen), i64(1), i64(1)])
        u = g.node("Unsqueeze", [x, i64(1)])
        c = g.node("Conv", [u, g.const(ww)] + ([g.const(bias)] if bias is not None else []), kernel_shape=[L], strides=[hop])
        if not nmel:
            return c
        return g.node("MatMul", [g.node("Transpose", [c], perm=[0, 2, 1]), g.const(mel)])
    data = op_graph(build, [frames, nmel] if nmel else [ww.shape[0], frames])
    path = write_model(data)
    text = bn.plan_describe(path)
    B = int(rng.integers(1, 4))
    x = (rng.standard_normal((B, 144000)) * 0.5).astype(np.float32)
    got, _ = bn.Context(bn.Model(path), B).infer(x)
    ref = onnx_ref.run_model(data, x)["output"]
    err = np.abs(got.reshape(ref.shape) - ref).max()
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    # the round-3 forms of the same plan: half fold, no merging, the generic folded GEMM
    for k_, v_ in (("BN_FRAMELDS", "0"), ("BN_CONVFOLD2", "0"), ("BN_CONVMERGE", "0")):
        os.environ[k_] = v_
    gen, _ = bn.Context(bn.Model(path), B).infer(x)
    for k_ in ("BN_FRAMELDS", "BN_CONVFOLD2", "BN_CONVMERGE"):
        del os.environ[k_]
    lines = [l for l in text.splitlines() if "~" in l]
    quarter, merged = "~quarter" in text, bool(nmel) and "MatMul" not in text
    nn = [int(l.split(" N=")[1].split()[0]) for l in lines]
    sliced = any(64 < n_ <= 96 for n_ in nn)  # two K slices: another summation order than the generic kernel's
    exact = not (quarter or merged or sliced or nmel)
    same = np.array_equal(gen.view(np.uint32), got.view(np.uint32)) if exact else bool(np.abs(gen - got).max() <= tol)
    ok = err <= tol and same and ("~" in text or " FFT " in text or merged)  # (merged cos + sin rows are neither symmetric nor a DFT bank: a plain GEMM)
    bad += not ok
    print(f"{it:2d} L={L} hop={hop} S={Slen} frames={frames} N={ww.shape[0]} {kind} bias={bias is not None} mel={nmel} B={B} err={err:.2e} tol={tol:.2e} "
          f"{'bit-identical' if exact else 'close'}={same} quarter={quarter} merged={merged} slices={sliced} {'OK' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
