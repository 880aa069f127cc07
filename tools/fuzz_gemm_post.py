"""Randomised differential check of GEMM epilogue chains (GPU): MatMul of random shape followed by random unary stages and a
random output view (dense / transposed / flipped + transposed), against the oracle and against the unfused plan bit for bit.
    python tools/fuzz_gemm_post.py"""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: F401
bn = importlib.import_module("rust-birdnet-onnx_amd")
from gpu_helpers import op_graph, write_model
from oracle import onnx_ref
rng = np.random.default_rng(7)
bad = 0
for it in range(40):
    rows, k, n = int(rng.integers(1, 700)), int(rng.integers(1, 200)), int(rng.integers(1, 131))
    while rows * k > 144000:
        rows //= 2
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    stages = [str(s) for s in rng.choice(["relu", "abs_sqrt", "mul", "add", "exp", "square", "neg"], size=int(rng.integers(1, 4)))]
    layout = str(rng.choice(["dense", "transpose", "flip_transpose"]))
    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(rows * k), i64(1), i64(1)])
        x = g.node("Reshape", [x, i64(-1, rows, k)])
        y = g.node("MatMul", [x, g.const(w)])
        for s in stages:
            if s == "relu": y = g.node("Relu", [y])
            elif s == "abs_sqrt": y = g.node("Sqrt", [g.node("Abs", [y])])
            elif s == "mul": y = g.node("Mul", [y, g.const(np.float32(1.25))])
            elif s == "add": y = g.node("Add", [y, g.const(np.float32(-0.5))])
            elif s == "exp": y = g.node("Exp", [g.node("Mul", [y, g.const(np.float32(0.25))])])
            elif s == "square": y = g.node("Mul", [y, y])
            else: y = g.node("Neg", [y])
        if layout == "flip_transpose":
            y = g.node("Slice", [y, i64(-1), i64(-(2 ** 62)), i64(2), i64(-1)])
        if layout != "dense":
            y = g.node("Transpose", [y], perm=[0, 2, 1])
        return y
    data = op_graph(build, [rows, n] if layout == "dense" else [n, rows])
    path = write_model(data)
    text = bn.plan_describe(path)
    B = int(rng.integers(1, 4))
    x = (rng.standard_normal((B, 144000)) * 0.5).astype(np.float32)
    got, _ = bn.Context(bn.Model(path), B).infer(x)
    ref = onnx_ref.run_model(data, x)["output"]
    err = np.abs(got.reshape(ref.shape) - ref)
    # sqrt(|y|) has an unbounded slope at 0: f32 summation-order noise of ~1e-7 in y becomes ~3e-4 there
    lim = (5e-4 if "abs_sqrt" in stages else 2e-5) + 2e-5 * np.abs(ref)
    os.environ["BN_GEMMPOST"] = "0"
    plain, _ = bn.Context(bn.Model(path), B).infer(x)
    del os.environ["BN_GEMMPOST"]
    same = np.array_equal(plain.view(np.uint32), got.view(np.uint32))
    ok = bool((err <= lim).all()) and same
    bad += not ok
    print(f"{it:2d} rows={rows} K={k} N={n} {'+'.join(stages)} {layout} B={B} maxerr={err.max():.2e} same={same} absorbed={' post=' in text} {'OK' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
