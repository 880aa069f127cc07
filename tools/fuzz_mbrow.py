"""Randomised differential check of the row-streaming fused MBConv kernel (GPU): expand 1x1 (+bias +act) -> depthwise KxK (+act)
of random shape, channel count, stride, band height and batch -- against the oracle (tolerance) and against the tiled kernel BIT FOR
BIT (both follow the same accumulation order).  Also random 3x3 stems (dense k1 x k1 conv with few input channels in front).
    python tools/fuzz_mbrow.py [cases]"""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: F401
bn = importlib.import_module("rust-birdnet-onnx_amd")
from gpu_helpers import op_graph, write_model
from oracle import onnx_ref
rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
ACTS = ["relu", "silu", "relu6", "hswish", None]


def act_node(g, y, a):
    if a == "relu": return g.node("Relu", [y])
    if a == "silu": return g.node("Mul", [y, g.node("Sigmoid", [y])])
    if a == "relu6": return g.node("Clip", [y, g.const(np.float32(0)), g.const(np.float32(6))])
    if a == "hswish": return g.node("HardSwish", [y])
    return y


for it in range(cases):
    stem = it % 4 == 3
    k, s = int(rng.choice([3, 5])), int(rng.choice([1, 2]))
    a1, a2 = ACTS[int(rng.integers(0, 5))], ACTS[int(rng.integers(0, 5))]
    if stem:
        k = 3
        cin1, k1, s1 = int(rng.choice([1, 2, 3, 4])), int(rng.choice([3, 3, 4])), int(rng.choice([1, 2]))
        while k1 * k1 * cin1 <= 8 or k1 * k1 * cin1 > 32:
            cin1, k1 = int(rng.choice([1, 2, 3, 4])), int(rng.choice([3, 3, 4]))
        h, w = int(rng.integers(20, 70)), int(rng.integers(20, 140))
        while cin1 * h * w > 144000: h //= 2
        cmid = int(rng.choice([16, 24, 32, 40, 64]))
        p1 = int(rng.integers(0, k1 // 2 + 1))
        h1, w1 = (h + 2 * p1 - k1) // s1 + 1, (w + 2 * p1 - k1) // s1 + 1
        cin = cin1
    else:
        cin = 4 * int(rng.integers(3, 13))            # 12 .. 48: two to six K groups
        cmid = int(rng.choice([32, 48, 72, 96, 100, 144, 160, 240]))
        h1, w1 = int(rng.integers(3, 40)), int(rng.integers(8, 90))
        while cin * h1 * w1 > 144000: h1 //= 2
        h, w = h1, w1
    pad = k // 2
    oh, ow = (h1 + 2 * pad - k) // s + 1, (w1 + 2 * pad - k) // s + 1
    if oh < 1 or ow < 1: continue
    toh = int(rng.integers(1, 10))
    wts = {}
    if stem:
        wts["w0"] = (rng.standard_normal((cmid, cin1, k1, k1)) / np.sqrt(cin1 * k1 * k1)).astype(np.float32)
    else:
        wts["w0"] = (rng.standard_normal((cin, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        wts["we"] = (rng.standard_normal((cmid, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    wts["b1"] = rng.standard_normal(cmid).astype(np.float32)
    wts["wd"] = (rng.standard_normal((cmid, 1, k, k)) / k).astype(np.float32)
    wts["b2"] = rng.standard_normal(cmid).astype(np.float32)

    def build(g, x):
        i64 = lambda *v: g.const(np.array(v, dtype=np.int64))
        x = g.node("Slice", [x, i64(0), i64(cin * h * w), i64(1), i64(1)])
        if stem:
            x = g.node("Reshape", [x, i64(-1, h, w, cin)])
            x = g.node("Transpose", [x], perm=[0, 3, 1, 2])
            y = g.node("Conv", [x, g.const(wts["w0"]), g.const(wts["b1"])], kernel_shape=[k1, k1], strides=[s1, s1], pads=[p1] * 4)
        else:
            x = g.node("Reshape", [x, i64(-1, cin, h, w)])
            x = g.node("Conv", [x, g.const(wts["w0"])], kernel_shape=[1, 1])
            y = g.node("Conv", [x, g.const(wts["we"]), g.const(wts["b1"])], kernel_shape=[1, 1])
        y = act_node(g, y, a1)
        z = g.node("Conv", [y, g.const(wts["wd"]), g.const(wts["b2"])], kernel_shape=[k, k], strides=[s, s], pads=[pad] * 4, group=cmid)
        return act_node(g, z, a2)
    data = op_graph(build, [cmid, oh, ow])
    path = write_model(data)
    B = int(rng.integers(1, 4))
    x = (rng.standard_normal((B, 144000)) * 0.7).astype(np.float32)
    os.environ.update({"BN_MBFUSE": "force", "BN_MBMAP": "0", "BN_MBROW": "force", "BN_MBROW_TOH": str(toh)})
    text = bn.plan_describe(path)
    line = [l for l in text.splitlines() if " MBCONV " in l]
    rows = int(line[0].rsplit("rows=", 1)[1]) if line else -1
    got, _ = bn.Context(bn.Model(path), B).infer(x)
    os.environ["BN_MBROW"] = "0"
    tiled, _ = bn.Context(bn.Model(path), B).infer(x)
    for key in ("BN_MBFUSE", "BN_MBMAP", "BN_MBROW", "BN_MBROW_TOH"): del os.environ[key]
    ref = onnx_ref.run_model(data, x)["output"]
    err = np.abs(got.reshape(ref.shape) - ref)
    same = np.array_equal(got.view(np.uint32), tiled.view(np.uint32))
    ok = bool((err <= 2e-4 + 2e-4 * np.abs(ref)).all()) and (same or rows <= 0)
    bad += not ok
    print(f"{it:2d} {'stem' if stem else 'mb  '} {h1}x{w1}x{cin}->{cmid} k{k} s{s} {a1}/{a2} toh={toh} B={B} rows={rows} maxerr={err.max():.2e} same={same} {'OK' if ok else 'FAIL'}", flush=True)
    os.unlink(path)
print("failures:", bad)
sys.exit(1 if bad else 0)
