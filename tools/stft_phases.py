#!/usr/bin/env python3
"""Phase timing of the STFT launches (experiments): BN_STFT_DBG bit mask skips phases (1 transform, 2 outputs + mel,
4 frame load, 8 everything after the block prologue).  Prints the two launch times per mask."""
import importlib, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bn = importlib.import_module("rust-birdnet-onnx_amd")
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
        f.write(synth.birdnet_v24())
    ctx = bn.Context(bn.Model(f.name), 32)
    acc = None
    for r in range(6):
        rows = ctx.time_kernels(32)
        if r:
            acc = [a + b[1] / 5 for a, b in zip(acc, rows)] if acc else [b[1] / 5 for b in rows]
    print(os.environ.get("BN_STFT_DBG", "0"), " ".join(f"{n[:24]}={u:.1f}" for (n, _, _, _), u in zip(rows, acc) if "stft" in n), flush=True)
else:
    for mask in (sys.argv[1:] or ["0", "8", "7", "3", "2", "1", "6", "5"]):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, BN_STFT_DBG=mask))
