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


def shard_sample_range(n_samples: int, segment_samples: int, step_samples: int, lo: int, hi: int) -> Tuple[int, int]:
    """Samples [a, b) of the recording that windows [lo, hi) touch: a = lo*step, b = min(n, (hi-1)*step + S).
    A rank uploads only this slice; window g of the recording is window g - lo of the slice, and the
    zero padding of the recording's last window is reproduced by the slice ending at n."""
    if hi <= lo:
        return 0, 0
    return lo * step_samples, min(n_samples, (hi - 1) * step_samples + segment_samples)


def chunk_step(segment_samples: int, overlap_secs: float, sample_rate: int) -> int:
    """Window step of the reference's chunk_audio (src/bin/birdnet-analyze.rs:718-723):
    `overlap_samples = (overlap_secs * sample_rate as f32) as usize` -- a Rust float-to-usize cast truncates toward
    zero and SATURATES (negative and NaN give 0) -- then `segment_samples.saturating_sub(overlap_samples)`; a step
    of 0 means chunk_audio returns no windows."""
    prod = np.float32(overlap_secs) * np.float32(sample_rate)
    if not np.isfinite(prod):
        overlap = 0 if (np.isnan(prod) or prod < 0) else (1 << 64) - 1
    else:
        overlap = max(0, int(prod))
    return max(0, int(segment_samples) - overlap)


def analyze_recording_sharded(bn, model, samples: np.ndarray, overlap_secs: float, batch: int = 32, streams: int = 4, top_k: int = 10,
                              min_confidence: Optional[float] = None, dist=None, gather: str = "logits", ctxs=None):
    """BASELINE.json configs[4]: a long mono recording (int16 or float32), sharded by window across the
    ranks of one node.  Every rank uploads its slice once (bn_recording_create), cuts windows on the
    device and keeps `streams` contexts in flight (bn_step_windows); ONE collective at the end
    assembles the [G, N] logits and the [G, k] top-K rows (gather="logits", the default: 26 KB per window for BirdNET
    v2.4 -- the reference's PredictionResult carries `raw_scores` for every segment, classifier.rs:907,948) or, as an
    explicit opt-in, the top-K rows only (gather="topk": 80 B per window at k = 10; `raw_scores` are then None).

    Returns (logits or None, topk_idx, topk_conf, topk_count) for all G windows, in time order."""
    import torch

    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    cfg = model.config
    S, sr = int(cfg.sample_count), int(cfg.sample_rate)
    step = chunk_step(S, overlap_secs, sr)
    n = int(samples.shape[0])
    # chunk_audio returns no windows at all when the step saturates to 0 (birdnet-analyze.rs:720-723)
    G = (n + step - 1) // step if n > 0 and step > 0 else 0
    if G == 0:
        k0 = min(top_k, int(cfg.num_species))
        empty = np.empty((0, int(cfg.num_species)), dtype=np.float32) if gather == "logits" else None
        return empty, np.zeros((0, k0), dtype=np.uint32), np.zeros((0, k0), dtype=np.float32), np.zeros(0, dtype=np.uint32)
    lo, hi = shard_range(G, rank, world)
    a, b = shard_sample_range(n, S, step, lo, hi)
    # asynchronous upload: the first windows are analysed while the rest of this rank's slice is still crossing the bus (the step
    # calls block until the samples they read have arrived; the Recording keeps the array alive)
    rec = bn.Recording(np.ascontiguousarray(samples[a:b]), device=model.device, async_upload=True)
    if ctxs is None:  # callers analysing many recordings keep their contexts (arena + captured graphs)
        ctxs = [bn.Context(model, batch) for _ in range(max(1, streams))]
    N = ctxs[0].output_device(cfg.logits_output)[1]
    k = min(top_k, N)
    n_local = hi - lo
    # With RCCL the logits never leave the device before the collective: every finished step's rows are copied from the
    # context's device buffer into this rank's slab of the gather buffer ON THE CONTEXT'S STREAM (device to device), and
    # the slab is all-gathered in place.  (Round 1 went device -> host -> numpy -> device -> all_gather -> host.)
    on_gpu = dist is not None and dist.is_initialized() and dist.get_backend() == "nccl" and gather == "logits"  # (also a 1-rank rehearsal)
    cap = shard_capacity(G, world)
    slab = views = streams = None
    if on_gpu:
        class _DevView:  # zero-copy torch view of a context's device logits
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f4", "version": 2}

        torch.cuda.set_device(model.device)
        # torch.empty + zeroing only this rank's padding rows: a whole-buffer zero fill would run on torch's current
        # stream, which the contexts' non-blocking streams never synchronise with -- a late fill could wipe rows a
        # context had already copied in.  The padding fill is followed by a device-wide synchronize, so it is complete
        # before the first step is enqueued on any context stream.
        full = torch.empty((world * cap, N), dtype=torch.float32, device=torch.device("cuda", model.device))
        slab = full[rank * cap:(rank + 1) * cap]
        if n_local < cap:
            slab[n_local:].zero_()
        torch.cuda.synchronize(model.device)
        views = [torch.as_tensor(_DevView(c.output_device(cfg.logits_output)[0], (batch, N)), device=slab.device) for c in ctxs]
        streams = [torch.cuda.ExternalStream(c.stream()) for c in ctxs]
    logits = np.empty((n_local, N), dtype=np.float32) if gather == "logits" and not on_gpu else None
    idx = np.zeros((n_local, k), dtype=np.uint32)
    conf = np.zeros((n_local, k), dtype=np.float32)
    cnt = np.zeros(n_local, dtype=np.uint32)
    jobs = [(f, min(batch, n_local - f)) for f in range(0, n_local, batch)]

    def collect(j):
        f, m = jobs[j]
        c = ctxs[j % len(ctxs)]
        c.synchronize()
        lg, ix, cf, ct = c.step_results(m)
        if logits is not None:
            logits[f:f + m] = lg[:m]
        idx[f:f + m], conf[f:f + m], cnt[f:f + m] = ix[:m, :k], cf[:m, :k], ct[:m]

    for j, (f, m) in enumerate(jobs):
        if j >= len(ctxs):
            collect(j - len(ctxs))
        ctxs[j % len(ctxs)].step_windows(rec, step, f, m, top_k, min_confidence)
        if on_gpu:  # the step's rows -> this rank's slab, ordered behind the step on the context's own stream
            with torch.cuda.stream(streams[j % len(ctxs)]):
                slab[f:f + m].copy_(views[j % len(ctxs)][:m], non_blocking=True)
    for j in range(max(0, len(jobs) - len(ctxs)), len(jobs)):
        collect(j)

    if world == 1 and not on_gpu:
        return logits, idx, conf, cnt
    dev = torch.device("cuda", model.device) if dist.get_backend() == "nccl" else torch.device("cpu")

    def gather_np(arr, dtype):
        t = torch.from_numpy(arr.astype(dtype, copy=False).reshape(n_local, -1)).to(dev)
        return gather_rows(t, G, dist).cpu().numpy()
    g_idx = gather_np(idx.view(np.int32), np.int32).view(np.uint32)
    g_conf = gather_np(conf, np.float32)
    g_cnt = gather_np(cnt.view(np.int32).reshape(-1, 1), np.int32).view(np.uint32).reshape(-1)
    if on_gpu:
        for c in ctxs:
            c.synchronize()  # every slab copy has landed (`full` stays referenced here until the gather has read it)
        dist.all_gather_into_tensor(full, slab)
        g_logits = full[:G].cpu().numpy()
        # Teardown order, explicit (round 3's SIGSEGV): everything torch holds that names a context's stream or memory
        # -- the ExternalStream wrappers, the zero-copy views of the contexts' logits, the gather buffer the slab copies ran into on
        # those streams -- is released and the device drained BEFORE the contexts (locals of this frame when the caller passed none)
        # destroy their hipStream_t.  Round 3 additionally called full.record_stream(<context stream>): torch's caching allocator then
        # records an event on that stream when `full` is FREED -- at function exit, where CPython drops the parameter `ctxs` (an
        # early slot: bn_ctx_destroy -> hipStreamDestroy) before the later local `full` -- i.e. hipEventRecord on a destroyed stream,
        # the crash in gpurun_out/r3c/tests.log.  record_stream is not needed (every context is synchronised before the collective
        # reads the slab) and must never be used with a context's stream.
        del views, streams, slab, full
        torch.cuda.synchronize(model.device)
        # (rows of rank r sit at [r * cap, r * cap + n_r): with equal capacities the first G rows are exactly the windows in
        # time order, because only the LAST ranks can be short and their padding lies behind row G)
    else:
        g_logits = gather_np(logits, np.float32) if logits is not None else None
    return g_logits, g_idx, g_conf, g_cnt
