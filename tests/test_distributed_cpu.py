"""N>1 path on CPU: two gloo ranks shard a recording's windows, all-gather their rows, and every
rank must end up with exactly the single-process result."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dmod = importlib.import_module("rust-birdnet-onnx_amd.distributed")


def test_shard_ranges_cover_and_preserve_order():
    for n in (0, 1, 7, 8, 9, 28800, 28801):
        for world in (1, 2, 3, 8):
            spans = [dmod.shard_range(n, r, world) for r in range(world)]
            flat = [i for lo, hi in spans for i in range(lo, hi)]
            assert flat == list(range(n))
            assert all(hi - lo <= dmod.shard_capacity(n, world) for lo, hi in spans)
    # BASELINE.json configs[4]: 24 h at 48 kHz, overlap 0 -> 28 800 windows, 3 600 per GPU
    assert dmod.shard_range(28800, 7, 8) == (25200, 28800)


def test_chunk_starts_and_windows_match_oracle():
    n, S, sr = 144000 * 5 + 12345, 144000, 48000
    rec = np.random.default_rng(0).standard_normal(n).astype(np.float32)
    for ov in (0.0, 1.5, 2.25):
        starts = dmod.chunk_starts(n, S, ov, sr)
        so, _ = oracle.chunk_plan(n, S, ov, sr)
        assert starts.tolist() == so.tolist()
        w = dmod.fill_windows(rec, starts, S)
        for k in (0, len(starts) // 2, len(starts) - 1):
            assert np.array_equal(w[k], oracle.chunk_fill(rec, S, int(starts[k])))


def _fake_logits(x: np.ndarray) -> torch.Tensor:
    # deterministic stand-in for the device path: a function of the window content only
    t = torch.from_numpy(x)
    return torch.stack([t.sum(dim=1), t.abs().max(dim=1).values, t[:, 0], t[:, -1], (t * t).mean(dim=1)], dim=1)


def _worker(rank, world, port, n_windows, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("rust-birdnet-onnx_amd.distributed")
    S = 1000
    rec = np.random.default_rng(7).standard_normal(n_windows * S - 137).astype(np.float32)
    starts = np.arange(n_windows, dtype=np.int64) * S

    def windows_for(lo, hi):
        return d.fill_windows(rec, starts[lo:hi], S)
    got = d.analyze_sharded(windows_for, n_windows, _fake_logits, 5, batch=4, dist=dist)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), got.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_windows", [13, 16, 1])
def test_two_gloo_ranks_equal_single_process(tmp_path, n_windows):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_windows, str(tmp_path)), nprocs=2, join=True)
    S = 1000
    rec = np.random.default_rng(7).standard_normal(n_windows * S - 137).astype(np.float32)
    starts = np.arange(n_windows, dtype=np.int64) * S
    want = dmod.analyze_sharded(lambda lo, hi: dmod.fill_windows(rec, starts[lo:hi], S), n_windows, _fake_logits, 5, batch=4).numpy()
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == (n_windows, 5)
        assert got.tobytes() == want.tobytes()
