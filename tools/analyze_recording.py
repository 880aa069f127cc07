#!/usr/bin/env python3
"""BASELINE.json configs[4]: a long 48 kHz mono recording sharded by 3 s window across the GPUs of a
node, logits (or top-K rows) assembled by one all-gather over RCCL/xGMI.

    python tools/analyze_recording.py --hours 1                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/analyze_recording.py --hours 24 [--gather topk]         # one rank per GPU

The recording is synthetic int16 (tones + LCG noise, a silent stretch every 32nd window); every rank
uploads only the samples its windows touch and cuts the windows on the device.  Prints one JSON line
(rank 0): windows, wall time of the analysis (upload + inference + collective), segments/s, x realtime."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # four contexts, four hardware queues of their own (see bench.py)
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth_recording_i16(n_samples: int, sr: int, seed: int = 12345) -> np.ndarray:
    """One 96 s period (32 windows of 3 s: tones cycling over 4 frequencies + noise, the last window silent), tiled."""
    period = 32 * 3 * sr
    rng = np.random.default_rng(seed)
    t = np.arange(period, dtype=np.float64) / sr
    w = np.arange(period) // (3 * sr)
    f = np.array([440.0, 1000.0, 2500.0, 6000.0])[w % 4]
    x = 0.5 * np.sin(2 * np.pi * f * t) + 0.05 * rng.uniform(-1, 1, period)
    x[w == 31] = 0.0
    block = np.round(x * 32767.0).astype(np.int16)
    return np.tile(block, (n_samples + period - 1) // period)[:n_samples]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=float, default=1.0)
    ap.add_argument("--overlap", type=float, default=0.0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--gather", choices=["logits", "topk"], default="logits")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--device", type=int, default=None)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = int(os.environ.get("LOCAL_RANK", 0)) if a.device is None else a.device
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=a.backend)
    bn = importlib.import_module("rust-birdnet-onnx_amd")
    bn.set_sharing_mode(bn.SHARING_SHARED)  # several contexts per rank keep batches in flight: the launches' forms for a shared device
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    dmod = importlib.import_module("rust-birdnet-onnx_amd.distributed")
    with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
        f.write(synth.birdnet_v24())
    model = bn.Model(f.name, device=dev)
    os.unlink(f.name)
    sr = int(model.config.sample_rate)
    pcm = synth_recording_i16(int(a.hours * 3600 * sr), sr)
    # warm-up on a short prefix (graph capture, clocks), not timed
    ctxs = [bn.Context(model, a.batch) for _ in range(max(1, a.streams))]
    dmod.analyze_recording_sharded(bn, model, pcm[:sr * 3 * a.batch * a.streams], a.overlap, a.batch, a.streams, dist=None, ctxs=ctxs)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logits, idx, conf, cnt = dmod.analyze_recording_sharded(bn, model, pcm, a.overlap, a.batch, a.streams, dist=dist, gather=a.gather, ctxs=ctxs)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        G = len(cnt)
        print(json.dumps({"workload": f"BirdNET v2.4, {a.hours:g} h synthetic 48 kHz int16 recording, overlap {a.overlap:g} s", "n_gpus": world,
                          "windows": int(G), "seconds": round(dt, 3), "segments_per_s": round(G / dt, 1),
                          "x_realtime": round(a.hours * 3600 / dt, 1), "gather": a.gather,
                          "detections": int((cnt > 0).sum()), "top1_of_window_0": int(idx[0, 0]) if cnt[0] else None}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
