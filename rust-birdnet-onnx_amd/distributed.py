"""Segment-sharded analysis of a long recording across the GPUs of one node.

The path shards naturally: every fixed-length window is processed with no
cross-window state (reference src/bin/birdnet-analyze.rs:707-743 ``chunk_audio``,
src/classifier.rs:700-704).  One process per GPU (``torch.distributed``, backend
``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests):

* the global chunk plan (start sample of every window, reference semantics:
  ``step = S - floor(overlap*sr)``, one window for every ``pos < len``, zero-padded
  tail) is computed identically on every rank;
* rank ``r`` of ``R`` owns the contiguous range ``[r*ceil(G/R), min(G,(r+1)*ceil(G/R)))``
  so that concatenating the ranks' results restores time order;
* each rank runs its windows through its own context in batches and keeps the
  logits rows on its device;
* ONE collective at the end: ``all_gather_into_tensor`` of the ``[ceil(G/R), N]``
  logits slab (last rank zero-padded to equal size), trimmed to ``G`` rows.

Because the kernels' summation order does not depend on the batch size
(kernels.hip: ``gemm_use_splitk``), the gathered result is bit-identical to a
single-GPU run over the same windows whatever the batch composition.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range of rank `rank`: [lo, hi).  Every rank's capacity is ceil(n/world)."""
    per = (n_items + world - 1) // world if world > 0 else n_items
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi


def shard_capacity(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


def chunk_starts(n_samples: int, segment_samples: int, overlap_secs: float, sample_rate: int) -> np.ndarray:
    """Start sample of every window (host mirror of chunk_audio through the C++ shim)."""
    from . import chunk_plan  # compiled host mirror (bnh_chunk_plan)

    starts, _ = chunk_plan(n_samples, segment_samples, overlap_secs, sample_rate)
    return starts.astype(np.int64)


def fill_windows(samples: np.ndarray, starts: np.ndarray, segment_samples: int) -> np.ndarray:
    """Materialise zero-padded windows [len(starts), segment_samples] from a mono f32 recording."""
    out = np.zeros((len(starts), segment_samples), dtype=np.float32)
    n = samples.shape[0]
    for k, s in enumerate(starts):
        e = min(int(s) + segment_samples, n)
        if e > s:
            out[k, :e - int(s)] = samples[int(s):e]
    return out


def gather_rows(local_rows, n_items: int, dist=None):
    """All-gather per-rank row slabs (torch tensors [n_local, N]) into [n_items, N] on every rank.

    `dist` is torch.distributed (initialised) or None for a single process."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_rows[:n_items]
    world = dist.get_world_size()
    cap = shard_capacity(n_items, world)
    n_cols = local_rows.shape[1]
    slab = torch.zeros((cap, n_cols), dtype=local_rows.dtype, device=local_rows.device)
    slab[:local_rows.shape[0]] = local_rows
    out = torch.empty((world * cap, n_cols), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, slab)
    return out[:n_items]


def analyze_sharded(windows_for: Callable[[int, int], np.ndarray], n_windows: int, infer_rows: Callable[[np.ndarray], "object"],
                    n_cols: int, batch: int, dist=None, device: Optional[str] = None):
    """Run this rank's windows in batches and all-gather the logits.

    windows_for(lo, hi) -> f32 [hi-lo, S] windows of the global range (host);
    infer_rows(x)       -> torch tensor [len(x), n_cols] of logits for a batch (device or CPU).
    Returns the [n_windows, n_cols] logits on every rank."""
    import torch

    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    lo, hi = shard_range(n_windows, rank, world)
    rows = []
    for s in range(lo, hi, batch):
        e = min(hi, s + batch)
        rows.append(infer_rows(windows_for(s, e)))
    if rows:
        local = torch.cat(rows, dim=0)
    else:
        local = torch.zeros((0, n_cols), dtype=torch.float32, device=device or "cpu")
    return gather_rows(local, n_windows, dist)
