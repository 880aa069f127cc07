#!/usr/bin/env python3
"""Benchmark of the hot path: BirdNET-v2.4-style model, batch of 32 synthetic 48 kHz 3 s segments
per GPU (BASELINE.json configs[1]), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one whole pass of the hot path over one batch per GPU: the launch plan (front end,
CNN, head), the top-K kernel, and the device-to-host copies of the logits rows and top-K results
(bn_step_device).  With N > 1 the segments of a step are sharded contiguously over the ranks
(weak scaling: 32 per GPU) and the per-rank logits are all-gathered with RCCL.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  The roofline / per-kernel numbers
come from HIP events recorded on the context's stream around every launch (bn_ctx_time_kernels);
cpu_baseline times the CPU oracle (torch CPU, fp32) on a bounded sample on rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: f32-input MFMA peak (= vector peak)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="segments per GPU per step (BASELINE configs[1]: 32)")
    ap.add_argument("--streams", type=int, default=4, help="contexts (HIP streams) in flight per GPU")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host (bn_infer_submit / collect) measurement")
    ap.add_argument("--host-steps", type=int, default=120, help="minimum number of steps of the host-to-host leg")
    ap.add_argument("--host-warmup", type=int, default=96, help="minimum number of warm-up steps of the host-to-host leg")
    ap.add_argument("--cpu-sample", type=int, default=512, help="segments the CPU oracle is timed on (about 15-30 s of host work)")
    ap.add_argument("--own-buffer", action="store_true",
                    help="every step re-reads the batch placed in the context's own input buffer before the timed region (no per-step input "
                         "copy, input cache-resident): the round-3 protocol, reported as `own_buffer` next to the default rotating-input value")
    ap.add_argument("--rotate-mib", type=int, default=320, help="footprint of the rotating input batches (default: more than the 256 MiB Infinity Cache)")
    ap.add_argument("--dump-steps", action="store_true", help="keep step_done_ms for runs of any length")
    ap.add_argument("--no-saturated", action="store_true", help="skip roofline.saturated (the dominant family timed at 4x the batch)")
    ap.add_argument("--no-extras", action="store_true", help="skip latency_b1 (configs[0]) and the v3.0 b64 / Perch b128 lines (configs[2], [3])")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-launch timing table to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the N > 1 code path (process group, staging copy, all-gather) with a single rank")
    ap.add_argument("--model", choices=["v24", "v30", "perch"], default="v24",
                    help="v24 = the benchmark (BASELINE configs[1]); v30 / perch = configs[2] / [3], informational (use --batch 64 / 128)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    use_dist = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29577"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)

    # Four contexts need four hardware queues of their own: the ROCm runtime multiplexes streams onto
    # GPU_MAX_HW_QUEUES (default 4) queues per process and the null stream takes one, so with the default a
    # fourth context shares a queue with another and serialises behind it (measured: 30.4 k seg/s with 4
    # contexts on 4 queues, 37.3 k on 8).  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch

    dist = None
    dev = local_rank if args.device is None else args.device
    if use_dist:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(dev)
    local_rank = dev

    bn = importlib.import_module("rust-birdnet-onnx_amd")
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")

    MODELS = {"v24": (144000, 48000, 3.0, synth.birdnet_v24, "BirdNET v2.4", "3s@48kHz"),
              "v30": (160000, 32000, 5.0, synth.birdnet_v30, "BirdNET v3.0", "5s@32kHz"),
              "perch": (160000, 32000, 5.0, synth.perch_v2, "Perch v2", "5s@32kHz")}
    S, SR, SEC, make_model, model_name, seg_name = MODELS[args.model]
    B = args.batch
    model_bytes = make_model()  # full-size hypothesised topology, seeded synthetic weights
    with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f:
        f.write(model_bytes)
        path = f.name
    model = bn.Model(path, device=local_rank)
    os.unlink(path)
    cfg = model.config
    N = cfg.num_species
    ctxs = [bn.Context(model, B) for _ in range(max(1, args.streams))]

    # synthetic inputs, resident in HBM before the timed region: this rank's contiguous shard of
    # NBUF global batches (global segment index = (buffer * world + rank) * B + i).  The host copies are NOT kept:
    # with 4 x 18 MB of source arrays still alive the N > 1 path (staging copies + all-gather) ran a quarter slower
    # (single-rank RCCL rehearsal 36 k vs 48 k segments/s, same timed loop) -- the host leg below re-reads them.
    # Enough DISTINCT device buffers that the rotation's footprint exceeds the 256 MiB Infinity Cache: every timed step then reads
    # a batch that is not cache-resident from an earlier step (four 18 MB buffers, or one buffer per context re-read
    # every step, stay in the last-level cache).  The first NGEN are generated on the host, the rest are row rotations of those
    # made on the device (distinct addresses are what matters to the memory system; the values only need to be realistic).
    NGEN = 4
    NBUF = max(NGEN, -(-args.rotate_mib * 1024 * 1024 // (B * S * 4)))
    bufs = []
    for b in range(NGEN):
        x = synth.synthetic_segments(B, S, SR, first_index=(b * world + rank) * B)
        bufs.append(torch.from_numpy(x).cuda())
        del x
    for b in range(NGEN, NBUF):
        bufs.append(torch.roll(bufs[b % NGEN], shifts=b // NGEN, dims=0).contiguous())
    torch.cuda.synchronize()

    # Each context's batch is placed in the context's OWN device input buffer (bn_ctx_input_device) before the timed
    # region: the plan reads it from there with no copy, and the captured graph depends on the batch size alone.
    class _DevBuf:
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f4", "version": 2}

    own_ptr = []
    for j, c in enumerate(ctxs):
        ptr, cap = c.input_device()
        assert cap >= B * S
        torch.as_tensor(_DevBuf(ptr, (B, S)), device="cuda").copy_(bufs[j % NBUF])
        own_ptr.append(ptr)
    torch.cuda.synchronize()

    gathered = None
    if use_dist:
        gathered = torch.empty((world * B, N), dtype=torch.float32, device="cuda")

    class _DevView:  # zero-copy torch view of the context's device logits
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": "<f4", "version": 2}

    logit_views = []
    for c in ctxs:
        ptr, n = c.output_device(cfg.logits_output)
        logit_views.append(torch.as_tensor(_DevView(ptr, (B, n)), device="cuda") if use_dist else None)

    S_ = len(ctxs)
    # N > 1: a context's logits are copied to a small staging buffer and all-gathered from there, so the
    # context is free for its next batch as soon as the copy is done (same number of contexts in flight as N = 1)
    NGROUP = 3
    on_gpu_dist = use_dist and args.backend == "nccl"
    # The logits of one round of contexts (S_ steps) are staged side by side and gathered by ONE collective: every
    # step's logits are still all-gathered inside the timed region, with a quarter of the collective launches.
    stage = [torch.empty((S_ * B, N), dtype=torch.float32, device="cuda") for _ in range(NGROUP)] if on_gpu_dist else None
    if on_gpu_dist:
        gathered = torch.empty((world * S_ * B, N), dtype=torch.float32, device="cuda")
    # the staging copy of a context's logits is enqueued on THAT CONTEXT'S stream (no further stream: every extra
    # active queue costs throughput), so the host never waits for it and the context's next batch simply follows it
    # in stream order; a staging buffer is rewritten only after the collective that read it has finished
    ctx_streams = [torch.cuda.ExternalStream(c.stream()) for c in ctxs] if on_gpu_dist else None
    gather_done = [None] * NGROUP
    pending = []  # copy events of the group being filled

    def gather_group(g):
        for ev in pending:
            torch.cuda.current_stream().wait_event(ev)
        pending.clear()
        dist.all_gather_into_tensor(gathered, stage[g])
        gather_done[g] = torch.cuda.Event()
        gather_done[g].record()

    def finish(j, last=False):
        """Results of step j are complete on the host side; with N > 1 all-gather its logits."""
        c = ctxs[j % S_]
        c.synchronize()
        if use_dist:
            if on_gpu_dist:
                g, slot = (j // S_) % NGROUP, j % S_
                cs = ctx_streams[slot]
                with torch.cuda.stream(cs):
                    if gather_done[g] is not None:
                        cs.wait_event(gather_done[g])
                    stage[g][slot * B:(slot + 1) * B].copy_(logit_views[slot], non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(cs)
                pending.append(copied)
                if slot == S_ - 1 or last:
                    gather_group(g)
            else:  # rehearsal backends gather on the host
                parts = [torch.empty((B, N), dtype=torch.float32) for _ in range(world)]
                dist.all_gather(parts, logit_views[j % S_].cpu())

    # one HIP event per step, recorded on the step's own stream right behind its last kernel (the result store):
    # consecutive events give the completion-to-completion interval of the steps on the device clock (SURVEY 8(d):
    # median and p10 / p90 over the timed steps, next to the wall-clock mean that `value` is)
    ev_streams = [torch.cuda.ExternalStream(c.stream()) for c in ctxs]
    # a small pool of events, reused: a step's event is read (elapsed time since the base event) when the step's results
    # are consumed, i.e. after its context has been synchronised, and is then free for the step 2 S_ later
    ev_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * S_)]
    ev_base = torch.cuda.Event(enable_timing=True)
    done_ms = []
    timed = {"on": False, "first": 0}

    def step(i, record=False):
        # contexts are used round-robin; a context's previous results are consumed just before it is reused
        if i >= S_:
            finish(i - S_)
            if record and i - S_ >= timed["first"]:
                done_ms.append(ev_base.elapsed_time(ev_pool[(i - S_) % (2 * S_)]))
        # a FRESH batch every step: a foreign device pointer, copied into the context's input buffer by one device-to-device copy
        # on the context's stream INSIDE the timed region (--own-buffer: the batch already sits in the context's buffer, no copy)
        ctxs[i % S_].step_device(own_ptr[i % S_] if args.own_buffer else bufs[i % NBUF].data_ptr(), B, args.top_k, 0.1, sync=False)
        if record:
            ev_pool[i % (2 * S_)].record(ev_streams[i % S_])

    def drain(total):
        for j in range(max(0, total - S_), total):
            finish(j, last=(j == total - 1))

    def fence():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # ---- order of the legs: the per-launch HIP-event timing that `roofline` needs (one context, every launch bracketed by events on
    # the context's stream: 12 passes at the workload's batch, one uncounted pass in front) runs FIRST, then the W warm-up steps, then
    # the K timed steps.  It used to run last; measured per round of four steps, a process that starts on an idle GPU (sclk at its
    # 577 MHz idle level while the model is authored on the host) delivers 2.65 ms per round for its first ~8 rounds (~20 ms) and
    # 2.2 ms from then on in EITHER input protocol -- the clock governor's ramp, not the code under test -- and the driver's
    # `--steps 20 --warmup 5` run lies entirely inside that ramp.  The roofline leg is device work of this same benchmark; running it
    # first changes nothing in what W and K mean (W untimed steps, then exactly K timed steps between the fences).
    kernel_rows, kernel_rows4 = None, None
    if rank == 0:
        ctxs[0].infer(bufs[0].cpu().numpy())  # puts a real batch into the context's own input buffer
        # Launches whose grid is a trade between one launch's latency and the work per block come in two forms (bn_set_sharing_mode): the
        # per-launch table of `roofline` -- one context ALONE on the device -- is timed in the form a lone context runs (what `--streams 1`
        # runs, comparable with earlier rounds); the form the timed region runs with several contexts alive is timed next to it
        # (`roofline.shared_forms`; `saturated` and the marginal costs are of that form).
        def _timed_rows(npass):
            ctxs[0].time_kernels(B)  # (first pass: warm-up of the per-launch events, not counted)
            acc = ctxs[0].time_kernels(B)
            for _ in range(npass - 1):
                acc = [(a[0], a[1] + b[1], a[2], a[3]) for a, b in zip(acc, ctxs[0].time_kernels(B))]
            return [(n_, us / float(npass), m_, by_) for n_, us, m_, by_ in acc]
        NPASS = 12
        bn.set_sharing_mode(bn.SHARING_ALONE)
        kernel_rows = _timed_rows(NPASS)
        kernel_rows_shared = None
        if len(ctxs) > 1:
            bn.set_sharing_mode(bn.SHARING_SHARED)
            kernel_rows_shared = _timed_rows(6)
        bn.set_sharing_mode(bn.SHARING_AUTO)
        # (the same passes at four times the batch -- `roofline.saturated` -- run BEHIND the timed steps since round 4: measured on one box,
        # `--steps 20 --warmup 5` reads 58.9-59.0 k with that leg in front of the timed region and 60.8-61.3 k without it (tools/ab_legs.sh);
        # the timed steps' own ramp-up is what the passes above are for, a fifth context of 1.4 GB is not part of the workload)
        # (ctxs[0]'s own buffer was overwritten by infer(): put its batch back for --own-buffer and the own_buffer leg)
        torch.as_tensor(_DevBuf(own_ptr[0], (B, S)), device="cuda").copy_(bufs[0])
        torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    for i in range(args.warmup):
        step(i)
    drain(args.warmup)
    fence()
    ev_base.record(ev_streams[0])
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, record=True)
    drain(args.steps)
    fence()
    dt = time.perf_counter() - t0
    bn.set_sharing_mode(bn.SHARING_ALONE)  # (the legs behind the timed region -- own buffer, host to host, one segment -- run the default forms)
    for j in range(max(0, args.steps - S_), args.steps):  # the last S_ steps were consumed by drain()
        done_ms.append(ev_base.elapsed_time(ev_pool[j % (2 * S_)]))
    done_ms = np.array(sorted(done_ms))
    # with S_ contexts in flight the completions come in clusters, so the per-step time is taken over a sliding window
    # of S_ consecutive completions: (done[i + S_] - done[i]) / S_
    win = min(S_, max(1, len(done_ms) - 1))
    gaps = (done_ms[win:] - done_ms[:-win]) / win if len(done_ms) > win else np.array([dt * 1e3 / max(args.steps, 1)])
    step_stats = {"median_ms": round(float(np.median(gaps)), 4), "p10_ms": round(float(np.percentile(gaps, 10)), 4),
                  "p90_ms": round(float(np.percentile(gaps, 90)), 4), "n": int(len(gaps)), "window": int(win),
                  "what": "per-step time from one HIP event per step (recorded on the step's stream): completion time of step i + window minus that of step i, over the window, sorted completions of all contexts"}
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity of the last step's results (not timed): top-1 of the device top-K == argmax of the logits
    lg, ix, cf, ct = ctxs[(args.steps - 1) % len(ctxs)].step_results(B)
    assert np.isfinite(lg).all()
    if on_gpu_dist:  # ... and this rank's rows of the last all-gather are that step's logits
        slot = (args.steps - 1) % S_
        mine = gathered[(rank * S_ + slot) * B:(rank * S_ + slot + 1) * B].cpu().numpy()
        assert np.array_equal(mine.view(np.uint32), lg.reshape(B, -1).view(np.uint32)), "all-gather rows differ from the step's logits"
    for r in range(B):
        if ct[r]:
            assert ix[r, 0] == int(np.argmax(lg[r]))

    total_segments = args.steps * B * world
    value = total_segments / dt

    # the round-3 protocol next to it (never `value`): the same loop with each context re-reading the batch in its own buffer
    own_buffer = None
    if not args.own_buffer:  # (every world size since round 5: the round-to-round comparison must not depend on prose)
        def run_own(n):
            for i in range(n):
                if i >= S_:
                    ctxs[(i - S_) % S_].synchronize()
                ctxs[i % S_].step_device(own_ptr[i % S_], B, args.top_k, 0.1, sync=False)
            for c in ctxs:
                c.synchronize()

        bn.set_sharing_mode(bn.SHARING_SHARED if len(ctxs) > 1 else bn.SHARING_ALONE)  # (the timed region's forms)
        run_own(2 * S_)
        fence()
        t_own = time.perf_counter()
        run_own(args.steps)
        fence()
        d_own = time.perf_counter() - t_own
        bn.set_sharing_mode(bn.SHARING_ALONE)
        if use_dist:
            t = torch.tensor([d_own], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d_own = float(t.item())
        own_buffer = {"value": round(args.steps * B * world / d_own, 2), "unit": "segments/s", "ms_per_step": round(d_own / args.steps * 1e3, 4), "steps": args.steps,
                      "protocol": "r3-own-buffer" + ("" if world == 1 else " (no logits gather in this leg)"),
                      "what": "each context re-reads the batch pre-placed in its own input buffer: no per-step input copy, input cache-resident (BENCH_r03's protocol)"}

    # ---- the drop-in call, host to host (never `value`): the same batches as HOST f32 slices through
    # bn_infer_submit / bn_infer_collect -- what Classifier::predict_batch_with_context costs a caller
    # (reference src/batch_context.rs:188-226, src/bin/birdnet-analyze.rs:637-647 times exactly this, wall clock):
    # staging into pinned memory by the library's pool, PCIe upload, plan, top-K, logits + top-K rows back to
    # pageable host arrays.  Same contexts, two batches in flight per context.
    host_to_host = None
    if rank == 0 and world == 1 and not args.no_host_leg:
        from collections import deque

        hbufs = [b_.cpu().numpy() for b_ in bufs]
        hsteps = max(args.steps, args.host_steps)
        # this leg has its own warm-up: the host side (staging pool, pinned slots, page placement of the source
        # arrays) needs ~80 steps to settle on the pool's boxes (tools/host_path_trace.py: submit 0.8 ms -> 0.3 ms)
        hwarm = max(args.warmup, args.host_warmup)
        outstanding = [deque() for _ in ctxs]
        last = {}

        def hstep(i):
            q = outstanding[i % S_]
            if len(q) == 2:
                last[i % S_] = ctxs[i % S_].collect(q.popleft(), want_embeddings=True)
            q.append(ctxs[i % S_].submit(hbufs[i % NBUF], args.top_k, 0.1))

        def hdrain():
            for j, q in enumerate(outstanding):
                while q:
                    last[j] = ctxs[j].collect(q.popleft(), want_embeddings=True)

        for i in range(max(hwarm, 2 * S_)):
            hstep(i)
        hdrain()
        t1 = time.perf_counter()
        for i in range(hsteps):
            hstep(i)
        hdrain()
        hdt = time.perf_counter() - t1
        # same bits as the device-resident step of the same batch
        chk = (hsteps - 1) % S_
        ctxs[chk].step_device(bufs[(hsteps - 1) % NBUF].data_ptr(), B, args.top_k, 0.1, sync=True)
        dlg, dix, dcf, dct = ctxs[chk].step_results(B)
        hlg, _, hix, hcf, hct = last[chk]
        assert np.array_equal(hlg.view(np.uint32), dlg.view(np.uint32)), "host-slice logits differ from the device-resident step"
        assert np.array_equal(hct, dct) and all(np.array_equal(hix[r, :hct[r]], dix[r, :dct[r]]) for r in range(B))
        hv = hsteps * B / hdt
        host_to_host = {"value": round(hv, 1), "unit": "segments/s", "ms_per_step": round(hdt / hsteps * 1e3, 4), "steps": hsteps,
                        "warmup": max(hwarm, 2 * S_), "h2d_GBs": round(hv * S * 4 / 1e9, 2), "contexts": S_, "in_flight_per_context": 2,
                        "path": "host f32 slices -> bn_infer_submit (pool staging + PCIe) -> plan + top-K -> bn_infer_collect -> host logits + top-K rows"}

    out = {
        "metric": f"audio-segments/sec (batch) {model_name} {seg_name}",
        "value": round(value, 2),
        "unit": "segments/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "median_ms": step_stats["median_ms"],
        "p10_ms": step_stats["p10_ms"],
        "p90_ms": step_stats["p90_ms"],
        "step_interval_stats": step_stats,
        "step_done_ms": [round(float(v), 3) for v in done_ms] if (len(done_ms) <= 64 or args.dump_steps) else None,  # completion time of every timed step since the start of the timed region (short runs only)
        "higher_is_better": True,
        "protocol": "r4-own-buffer" if args.own_buffer else "r4-rotating-input",  # r4-rotating-input: a fresh device batch per step, one D2D copy inside the timed region (BENCH_r01-r03: own-buffer)
        "pre_warmup_device_passes": (13 + (7 if max(1, args.streams) > 1 else 0)) if rank == 0 else 0,  # per-launch HIP-event timing passes (roofline leg: 13 in the default forms, 7 in the shared ones) that run before the W warm-up steps
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{model_name} (synthetic-weights hypothesised topology), batch={B} synthetic {SR // 1000} kHz {SEC:g} s segments per GPU, " + ("the batch pre-placed in each context's own input buffer (re-read every step)" if args.own_buffer else f"inputs resident in HBM: every step takes the next of {NBUF} distinct device batches ({NBUF * B * S * 4 >> 20} MiB in rotation, more than the Infinity Cache) with one device-to-device copy into the context's buffer inside the timed region") + ", logits+top-10 copied to host",
            "global_batch": B * world,
            "segments_per_gpu_per_step": B,
            "streams_per_gpu": max(1, args.streams),
            "sharing_mode": "shared" if max(1, args.streams) > 1 else "alone",
            "num_species": int(N),
            "x_realtime": round(value * SEC, 1),
            "parallelism": f"segment-sharded x{world}" + ((f" + RCCL all-gather of logits (one collective per {S_} steps)" if args.backend == "nccl" else f" + {args.backend} all-gather of logits on the host (rehearsal backend)") if world > 1 else ""),
        },
    }

    def run_extras():
        """configs[2] / [3] in the same line.  Each runs as a CHILD process of this script (`--model v30 --batch 64`, `--model perch
        --batch 128`, 60 timed steps): inside this process, behind the v2.4 run, the same loop reads 10 % lower (39.4 k against
        44.0 k segments/s for v3.0, whether it runs before or after the timing blocks below) -- the in-process loop is kept as the
        fallback and says so in "mode"."""
        import subprocess

        extra = {}
        for key, bsz in (("v30", 64), ("perch", 128)):
            if os.environ.get("BN_BENCH_EXTRAS_INPROCESS") == "1":  # diagnosis of the in-process / child difference (tools/two_models.sh)
                extra = None
                break
            try:
                r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--model", key, "--batch", str(bsz), "--steps", "60", "--warmup", "8",
                                     "--streams", str(S_), "--no-extras", "--no-cpu-baseline", "--no-host-leg", "--no-saturated"],
                                    capture_output=True, text=True, timeout=600)
                line = [l for l in r_.stdout.splitlines() if l.startswith("{")][-1]
                j_ = json.loads(line)
                extra[key] = {"workload": j_["config"]["workload"], "value": j_["value"], "unit": j_["unit"], "ms_per_step": j_["ms_per_step"],
                              "median_ms": j_.get("median_ms"), "p10_ms": j_.get("p10_ms"), "p90_ms": j_.get("p90_ms"), "steps": j_["steps"],
                              "flops_performed_per_segment": j_["roofline"].get("flops_performed_per_segment"),
                              "frac_mfma_f32_whole_path": j_.get("whole_path_frac_mfma_f32"), "frac_mfma_f32_whole_path_8d_counted": j_.get("whole_path_frac_mfma_f32_8d_counted"),
                              "capture_fallbacks": j_.get("capture_fallbacks"),
                              "roofline": {k_: j_["roofline"].get(k_) for k_ in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches_per_step", "avg_launch_us", "share_of_step",
                                                                                "mfma_busy", "mfma_busy_source", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "pmc_refused")},
                              "kernel_families": j_.get("kernel_families"),
                              "device_us_per_step_sum_of_launches": j_.get("device_us_per_step_sum_of_launches"),
                              "mode": "child process: " + " ".join(r_.args[1:])}
            except Exception as e_:  # noqa: BLE001 -- fall back to the in-process loop below
                print(f"bench: child run of {key} failed ({e_}); measuring it in this process", file=sys.stderr)
                extra = None
                break
        if extra is not None:
            out["extra"] = extra
            return
        if args.model == "v24":
            extra = {}
            for key, bsz in (("v30", 64), ("perch", 128)):
                S2, SR2, SEC2, mk2, name2, seg2 = MODELS[key]
                with tempfile.NamedTemporaryFile(suffix=".onnx", delete=False) as f2:
                    f2.write(mk2())
                m2 = bn.Model(f2.name, device=local_rank)
                os.unlink(f2.name)
                cs2 = [bn.Context(m2, bsz) for _ in range(S_)]
                own2 = []
                for q, c_ in enumerate(cs2):
                    ptr2, cap2 = c_.input_device()
                    assert cap2 >= bsz * S2
                    torch.as_tensor(_DevBuf(ptr2, (bsz, S2)), device="cuda").copy_(torch.from_numpy(synth.synthetic_segments(bsz, S2, SR2, first_index=(q % 2) * bsz)))
                    own2.append(ptr2)
                torch.cuda.synchronize()
                nst, nwu = max(24, min(args.steps, 60)), 8

                def run2(n):
                    for i in range(n):
                        if i >= S_:
                            cs2[(i - S_) % S_].synchronize()
                        cs2[i % S_].step_device(own2[i % S_], bsz, args.top_k, 0.1, sync=False)
                    for c_ in cs2:
                        c_.synchronize()

                run2(nwu)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                run2(nst)
                torch.cuda.synchronize()
                d2 = time.perf_counter() - t2
                lg2 = cs2[(nst - 1) % S_].step_results(bsz)[0]
                assert np.isfinite(lg2).all()
                c2 = m2.cost()
                rows2 = cs2[0].time_kernels(bsz)
                for _ in range(2):
                    rows2 = [(a[0], a[1] + b_[1], a[2], a[3]) for a, b_ in zip(rows2, cs2[0].time_kernels(bsz))]
                sum2 = sum(r_[1] for r_ in rows2) / 3.0
                extra[key] = {"workload": f"{name2}, batch={bsz} synthetic {SR2 // 1000} kHz {SEC2:g} s segments, inputs resident in HBM",
                              "value": round(nst * bsz / d2, 1), "unit": "segments/s", "ms_per_step": round(d2 / nst * 1e3, 4), "steps": nst,
                              "flops_performed_per_segment": round(2.0 * (c2.macs_mfma + c2.macs_valu)),
                              "frac_mfma_f32_whole_path": round(2.0 * (c2.macs_mfma + c2.macs_valu) * nst * bsz / d2 / 1e12 / MFMA_F32_PEAK_TF, 4),
                              "capture_fallbacks": sum(c_.stats()["capture_fallbacks"] for c_ in cs2), "mode": "in-process",
                              "device_us_per_step_sum_of_launches": round(sum2, 1)}
                del cs2, m2
            out["extra"] = extra

    if own_buffer is not None:
        out["own_buffer"] = own_buffer
    if host_to_host is not None:
        out["host_to_host"] = host_to_host
    if rank == 0 and world == 1 and not args.no_extras:
        # ---- BASELINE configs[0]: ONE 3 s segment through predict's path, host slice in -> logits + top-K on the host
        # (bn_infer_submit + collect on a batch-1 context, what Classifier::predict costs a caller, classifier.rs:610-643),
        # and the same segment device-resident (bn_step_device, batch 1); median of 60 calls after 10 warm-up calls
        c1 = bn.Context(model, 1)
        x1 = bufs[0][:1].cpu().numpy()

        def med_ms(fn, reps=60, warm=10):
            ts = []
            for r_ in range(warm + reps):
                t_ = time.perf_counter()
                fn()
                if r_ >= warm:
                    ts.append((time.perf_counter() - t_) * 1e3)
            return float(np.median(ts)), float(np.percentile(ts, 10)), float(np.percentile(ts, 90))

        m_, p10_, p90_ = med_ms(lambda: c1.collect(c1.submit(x1, args.top_k, 0.1)))
        out["latency_b1_ms"] = round(m_, 4)
        d1 = bufs[0][:1].contiguous()
        md_, pd10_, pd90_ = med_ms(lambda: c1.step_device(d1.data_ptr(), 1, args.top_k, 0.1, sync=True))
        out["latency_b1"] = {"host_to_host_ms": {"median": round(m_, 4), "p10": round(p10_, 4), "p90": round(p90_, 4)},
                             "device_resident_ms": {"median": round(md_, 4), "p10": round(pd10_, 4), "p90": round(pd90_, 4)},
                             "launches": int(model.cost().n_launches), "capture_fallbacks": c1.stats()["capture_fallbacks"],
                             "what": "one segment: Classifier::predict's path (configs[0]) -- wall clock of one synchronous call, batch-1 context"}
        del c1
    if rank == 0:
        out["capture_fallbacks"] = sum(c.stats()["capture_fallbacks"] for c in ctxs)
        out["whole_path_frac_mfma_f32"] = None  # filled below from the plan's flop count
        # ---- per-kernel device times (HIP events on the context's stream), roofline of the dominant kernel
        rows = kernel_rows  # measured before the warm-up steps (see "order of the legs" above)
        # classify launches by the kernel that executes them (plan_describe gives the op kind per launch)
        desc = bn.plan_describe(path_for_describe(model_bytes))
        plan_lines = [l for l in desc.splitlines() if l[:3].strip().isdigit()]
        kind_of = [l.split()[1] for l in plan_lines]
        gemm_kernel_of = [("frame_fold2q_kernel" if " kernel=frame_fold2q" in l else "frame_fold2_kernel" if " kernel=frame_fold2" in l else "frame_fold_kernel" if " kernel=frame_fold" in l
                           else "gemm_b3_kernel" if " kernel=b3" in l else "gemm_dma3_kernel" if " kernel=dma3" in l else "gemm_dma_kernel" if " kernel=dma" in l
                           else "gemm_splitk_kernel" if " kernel=splitk" in l else "gemm_mfma_kernel")
                          if l.split()[1] == "GEMM" else None for l in plan_lines]
        fam_name = {"GEMM": "gemm_mfma_kernel", "DWCONV": "dwconv_kernel", "CONV": "conv_direct_kernel", "MBCONV": "mbconv_row_kernel",
                    "REDUCE": "reduce_kernel", "ELT": "elt_kernel", "GAP": "gap_partial_kernel", "SEFC": "se_fc_kernel", "POOL": "pool_kernel", "FFT": "stft_kernel"}
        # (the fused MBConv launches are two different kernels -- the row-streaming one, mbrow.hip, and the small-map one, mbmap.hip --
        # and are counted as two families since round 4: which family is "dominant" should not hang on a sum over unrelated kernels)
        fam = {}
        for (name, us, macs, byts), k, line in zip(rows, kind_of, plan_lines):
            fname = "mbmap_kernel" if k == "MBCONV" and " map=cfg" in line else fam_name[k]
            f_ = fam.setdefault(fname, {"us": 0.0, "macs": 0.0, "bytes": 0.0, "launches": 0})
            f_["us"] += us
            f_["macs"] += macs
            f_["bytes"] += byts
            f_["launches"] += 1
        total_us = sum(v["us"] for v in fam.values())
        cost = model.cost()
        dom = max(fam.items(), key=lambda kv: kv[1]["us"])
        dname, d = dom
        # the same launches by the kernel FUNCTION that runs them (the symbols rocprofv3 lists): `roofline.kernel` is the symbol with the
        # most device time inside the dominant family, `roofline.family` the family label the PMC summaries are keyed by
        symbol_of = [(gk_ if gk_ else (("mbmap_ws_kernel" if ",ws " in line else "mbmap_kernel") if k == "MBCONV" and " map=cfg" in line else fam_name[k]))
                     for gk_, k, line in zip(gemm_kernel_of, kind_of, plan_lines)]
        sym = {}
        for (name, us, macs, byts), sn, k, line in zip(rows, symbol_of, kind_of, plan_lines):
            fname = "mbmap_kernel" if k == "MBCONV" and " map=cfg" in line else fam_name[k]
            e_ = sym.setdefault(sn, {"us": 0.0, "macs": 0.0, "bytes": 0.0, "launches": 0, "family": fname})
            e_["us"] += us
            e_["macs"] += macs
            e_["bytes"] += byts
            e_["launches"] += 1
        dom_sym = max(((k_, v_) for k_, v_ in sym.items() if v_["family"] == dname), key=lambda kv: kv[1]["us"])
        tf = 2.0 * d["macs"] / (d["us"] * 1e-6) / 1e12
        gbs = d["bytes"] / (d["us"] * 1e-6) / 1e9
        frac_mfma, frac_hbm = tf / MFMA_F32_PEAK_TF, gbs / HBM_PEAK_GBS
        if frac_hbm >= frac_mfma:
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac_hbm, 4)}
        else:
            roof = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(frac_mfma, 4)}
        # HBM bytes per launch from the committed PMC passes (tools/profile_round.sh + profile_summary.py:
        # FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, gfx950 read correction applied)
        # A committed PMC summary is only quoted while it belongs to the newest profile set under profiles/ (tag rNN_vM of
        # the newest rNN_vM_kernel_stats_*.csv): a summary older than the kernels it is quoted for is refused.
        import glob
        import re

        def tag_key(t):
            m_ = re.match(r"r(\d+)_v(\d+)", t or "")
            return (int(m_.group(1)), int(m_.group(2))) if m_ else (-1, -1)

        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        newest = max([tag_key(os.path.basename(f_)) for f_ in glob.glob(os.path.join(prof_dir, "r*_kernel_stats_*.csv"))] or [(-1, -1)])
        stale = []
        traffic, traffic_src = None, None
        try:
            psuf = "" if args.model == "v24" else "_" + args.model  # per-model summaries: profiles/pmc_traffic_v30.json, ..._perch.json
            tj = json.load(open(os.path.join(prof_dir, f"pmc_traffic{psuf}.json")))
            if tag_key(tj.get("tag")) < newest:
                stale.append(f"profiles/pmc_traffic{psuf}.json (tag {tj.get('tag')}) is older than the newest kernel stats r{newest[0]:02d}_v{newest[1]}")
            elif tj.get("batch") == B and dname in tj.get("families", {}):
                traffic = tj["families"][dname]["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{tj['tag']}_pmc_traffic.json"
        except (OSError, ValueError, KeyError):
            pass
        # matrix-pipe utilisation of the same family from the committed SQ counter pass (profiles/pmc_mfma.json)
        mfma_busy, mfma_src = None, None
        try:
            mj = json.load(open(os.path.join(prof_dir, f"pmc_mfma{psuf}.json")))
            if tag_key(mj.get("tag")) < newest:
                stale.append(f"profiles/pmc_mfma{psuf}.json (tag {mj.get('tag')}) is older than the newest kernel stats r{newest[0]:02d}_v{newest[1]}")
            elif mj.get("batch") == B and dname in mj.get("families", {}):
                mfma_busy = mj["families"][dname]["mfma_busy"]
                mfma_src = f"profiles/{mj['tag']}_pmc_mfma.json"
        except (OSError, ValueError, KeyError):
            pass
        roof.update({"mfma_busy": mfma_busy, "mfma_busy_source": mfma_src})
        if stale:
            roof["pmc_refused"] = stale
        roof.update({"traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                     "kernel": dom_sym[0], "family": dname,
                     # (round 5) gemm_b3_kernel / gemm_dma3_kernel / frame_fold2q_kernel compute their f32 products as six exact bf16 partial products on the
                     # bf16 matrix pipe: `achieved` stays ALGORITHMIC f32 flops per second (against the f32 MFMA peak the exact-f32 kernels are held to);
                     # the bf16 pipe sees six times those flops against its own 2.5 PFLOP/s
                     "bf16x3": {"symbols": sorted(k_ for k_ in sym if k_ in ("gemm_b3_kernel", "gemm_dma3_kernel", "frame_fold2q_kernel")),
                                "kernel_is_bf16x3": dom_sym[0] in ("gemm_b3_kernel", "gemm_dma3_kernel", "frame_fold2q_kernel"),
                                "bf16_pipe_TFLOPs_of_kernel": round(6 * 2.0 * dom_sym[1]["macs"] / (dom_sym[1]["us"] * 1e-6) / 1e12, 1) if dom_sym[0] in ("gemm_b3_kernel", "gemm_dma3_kernel", "frame_fold2q_kernel") else None,
                                "bf16_pipe_peak_TFLOPs": 2500.0,
                                "what": "f32 operands split exactly into three bf16 terms, six of nine partial products kept (each exact in f32; the dropped ones sum to < 2^-21 of the product in the worst case, 2^-24 rms): "
                                        "error against a double-precision product equal to the exact-f32 kernel's (tools/gemm3_bench); dtype stays f32"},
                     "kernel_alone": {"launches_per_step": dom_sym[1]["launches"], "us_per_step": round(dom_sym[1]["us"], 1), "avg_launch_us": round(dom_sym[1]["us"] / dom_sym[1]["launches"], 2),
                                      "TFLOPs": round(2.0 * dom_sym[1]["macs"] / (dom_sym[1]["us"] * 1e-6) / 1e12, 2),
                                      "frac_mfma_f32": round(2.0 * dom_sym[1]["macs"] / (dom_sym[1]["us"] * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4)},
                     "family_members": {k_: v_["launches"] for k_, v_ in sym.items() if v_["family"] == dname},
                     "launches_per_step": d["launches"],
                     "avg_launch_us": round(d["us"] / d["launches"], 2), "share_of_step": round(d["us"] / total_us, 3),
                     "alt_frac": {"hbm": round(frac_hbm, 4), "mfma_f32": round(frac_mfma, 4)},
                     "flops_performed_per_segment": round(2.0 * (cost.macs_mfma + cost.macs_valu)),
                     "flops_fft_counted_per_segment": round(2.0 * (cost.macs_mfma + cost.macs_valu - cost.dft_performed_macs) + cost.dft_fft_equiv_flops),
                     # SURVEY.md 8(d)'s count: the front-end banks priced as real FFTs AND the halo / band recompute of the fused MBConv
                     # launches left out -- the work the graph asks for, not what the plan spends on it
                     "flops_8d_counted_per_segment": round(2.0 * (cost.macs_mfma - cost.recompute_macs + cost.macs_valu - cost.dft_performed_macs) + cost.dft_fft_equiv_flops),
                     "recompute_flops_per_segment": round(2.0 * cost.recompute_macs),
                     "flops_note": "achieved/frac use flops PERFORMED; flops_fft_counted prices the windowed-DFT banks at 2.5 L log2 L per frame "
                                   "(SURVEY.md 8(d)) instead of the folded matrix product the plan runs -- multiply value x flops_fft_counted for the FFT-normalised rate",
                     "flops_counted": "multiply-adds the launches perform (planner's walk after its rewrites: mel-dead DFT bins pruned, "
                                      "mirror-symmetric DFT bases folded to half their taps) -- not the exporter graph's nominal count"})
        # the same family where the launches are not latency-bound: one context at 4x the batch, and the MARGINAL cost of a
        # further batch of B, (t(4B) - t(B)) / 3 per launch -- what each of the concurrent contexts pays per step
        try:
            if args.no_saturated:
                raise RuntimeError("skipped (--no-saturated)")
            big = bn.Context(model, 4 * B)
            bn.set_sharing_mode(bn.SHARING_SHARED if kernel_rows_shared is not None else bn.SHARING_ALONE)
            big.infer(np.concatenate([bufs[0].cpu().numpy()] * 4))
            big.time_kernels(4 * B)
            kernel_rows4 = big.time_kernels(4 * B)
            for _ in range(5):
                kernel_rows4 = [(a[0], a[1] + b[1], a[2], a[3]) for a, b in zip(kernel_rows4, big.time_kernels(4 * B))]
            kernel_rows4 = [(n_, us / 6.0, m_, by_) for n_, us, m_, by_ in kernel_rows4]
            big.close()
            del big
            bn.set_sharing_mode(bn.SHARING_ALONE)
            rows4 = kernel_rows4
            rows_alone = rows
            rows = kernel_rows_shared if kernel_rows_shared is not None else rows  # (the marginal costs: both batch sizes in the timed region's form)
            if rows4 is None:
                raise RuntimeError("not measured")
            if len(rows4) == len(rows):
                sel = [i for i, k in enumerate(kind_of) if fam_name[k] == dname]
                us1, us4 = sum(rows[i][1] for i in sel), sum(rows4[i][1] for i in sel)
                macs1 = sum(rows[i][2] for i in sel)
                marg = (us4 - us1) / 3.0
                all_marg = (sum(r[1] for r in rows4) - sum(r[1] for r in rows)) / 3.0
                roof["saturated"] = {"batch": 4 * B, "family_us": round(us4, 1), "TFLOPs": round(2 * 4 * macs1 / (us4 * 1e-6) / 1e12, 2),
                                     "frac": round(2 * 4 * macs1 / (us4 * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4),
                                     "marginal_us_per_batch": round(marg, 1), "marginal_TFLOPs": round(2 * macs1 / (marg * 1e-6) / 1e12, 2),
                                     "marginal_frac": round(2 * macs1 / (marg * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4),
                                     "all_launches_marginal_us_per_batch": round(all_marg, 1),
                                     "what": f"the same launches timed at batch {4 * B} on one context, and (t({4 * B}) - t({B})) / 3 = the cost of one more batch of {B} "
                                             "once the chip is full -- the regime the concurrent contexts of the headline number run in"}
                if kernel_rows_shared is not None:
                    roof["shared_forms"] = {"family_us_alone": round(us1, 1), "frac_alone": round(2 * macs1 / (us1 * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4),
                                            "device_us_per_step_sum_of_launches": round(sum(r[1] for r in rows), 1),
                                            "what": "the grids the launches take while several contexts are alive on the device (bn_set_sharing_mode: 64-row GEMM tiles, "
                                                    "twice the chunks per small-map MBConv block) -- what the timed region runs; `frac`, `achieved` and the per-launch "
                                                    "table above are of the form one context ALONE runs; `saturated` is of this form"}
            rows = rows_alone
        except Exception as e:  # noqa: BLE001 -- informational block only
            bn.set_sharing_mode(bn.SHARING_ALONE)
            rows = kernel_rows
            roof["saturated"] = {"error": str(e)[:200]}
        out["roofline"] = roof
        out["whole_path_frac_mfma_f32"] = round(roof["flops_performed_per_segment"] * value / world / 1e12 / MFMA_F32_PEAK_TF, 4)
        out["whole_path_frac_mfma_f32_8d_counted"] = round(roof["flops_8d_counted_per_segment"] * value / world / 1e12 / MFMA_F32_PEAK_TF, 4)
        # the three longest single launches, each against its own bound (the family number above averages
        # 33 launches, most of them latency-bound 6x32 / 3x16 feature maps at batch 32)
        tops = []
        for (name, us, macs, byts), k in sorted(zip(rows, kind_of), key=lambda t: -t[0][1])[:3]:
            tfl, gb = 2.0 * macs / (us * 1e-6) / 1e12, byts / (us * 1e-6) / 1e9
            # the launch against ITS OWN bound: max(bytes / HBM peak, flops / f32-MFMA peak) -- the N = 16 / 24 / 40 project convs move
            # 8-20 flop per byte of activation and are not MFMA-bound at all
            t_hbm_us, t_mfma_us = byts / (HBM_PEAK_GBS * 1e9) * 1e6, 2.0 * macs / (MFMA_F32_PEAK_TF * 1e12) * 1e6
            tops.append({"launch": name, "kind": k, "us": round(us, 1), "TFLOPs": round(tfl, 1), "GBs": round(gb, 1),
                         "frac_mfma_f32": round(tfl / MFMA_F32_PEAK_TF, 3), "frac_hbm": round(gb / HBM_PEAK_GBS, 3),
                         "own_bound": "hbm" if t_hbm_us >= t_mfma_us else "mfma", "own_bound_us": round(max(t_hbm_us, t_mfma_us), 2),
                         "frac_of_own_bound": round(max(t_hbm_us, t_mfma_us) / us, 3)})
        out["roofline_top_launches"] = tops
        # the GEMM family by the kernel function that runs each launch (the names rocprofv3 reports)
        gk = {}
        for (name, us, macs, byts), kn in zip(rows, gemm_kernel_of):
            if kn is None:
                continue
            e_ = gk.setdefault(kn, {"us": 0.0, "macs": 0.0, "launches": 0})
            e_["us"] += us
            e_["macs"] += macs
            e_["launches"] += 1
        out["gemm_family_by_kernel"] = {k: {"us_per_step": round(v["us"], 1), "launches": v["launches"], "TFLOPs": round(2 * v["macs"] / (v["us"] * 1e-6) / 1e12, 2),
                                            "frac_mfma_f32": round(2 * v["macs"] / (v["us"] * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 3)} for k, v in gk.items()}
        out["kernel_families"] = {k: {"us_per_step": round(v["us"], 1), "launches": v["launches"],
                                      "TFLOPs": round(2 * v["macs"] / (v["us"] * 1e-6) / 1e12, 2),
                                      "GBs": round(v["bytes"] / (v["us"] * 1e-6) / 1e9, 1)} for k, v in fam.items()}
        out["kernel_symbols"] = {k: {"us_per_step": round(v["us"], 1), "launches": v["launches"], "family": v["family"],
                                     "TFLOPs": round(2 * v["macs"] / (v["us"] * 1e-6) / 1e12, 2)} for k, v in sym.items()}
        # every multiply-add of the plan belongs to exactly one launch (bn_ctx_time_kernels reports op.macs + the fused launches' expand
        # convs + squeeze-excite prologues since round 5): the families' flops per segment sum to flops_performed_per_segment
        out["kernel_families_flops_per_segment"] = round(2.0 * sum(v["macs"] for v in fam.values()) / B)
        out["device_us_per_step_sum_of_launches"] = round(total_us, 1)
        # HBM bytes of one whole step from the committed PMC passes (every launch of the plan, by family) against the step time
        try:
            tj = json.load(open(os.path.join(prof_dir, f"pmc_traffic{psuf}.json")))
            if tag_key(tj.get("tag")) >= newest and tj.get("batch") == B:
                steps_prof = float(tj.get("steps_profiled", 3))
                step_bytes = sum(v_["launches_profiled"] / steps_prof * v_["hbm_bytes_per_launch"] for v_ in tj["families"].values())
                out["hbm_bytes_per_step_pmc"] = round(step_bytes)
                out["hbm_frac_step"] = round(step_bytes / (dt / args.steps) / (HBM_PEAK_GBS * 1e9), 4)
        except (OSError, ValueError, KeyError):
            pass
        if args.kernel_table:
            for (name, us, macs, byts), k in sorted(zip(rows, kind_of), key=lambda t: -t[0][1])[:40]:
                print(f"{us:9.1f} us  {k:7s} {name:42s} {2 * macs / us / 1e6:8.2f} TF/s {byts / us / 1e3:8.1f} GB/s", file=sys.stderr)

        # ---- CPU baseline: the oracle (kind "port": torch CPU fp32 restatement of the graph -- ORT-CPU and
        # the model files do not exist offline), batches of 8 like the reference CLI's CPU default
        # (src/bin/birdnet-analyze.rs:38-39), with the same dead-bin pruning the GPU plan applies, at every host thread
        # the box gives this process and at ONE thread; bounded samples, rank 0, N = 1 only
        if world == 1 and not args.no_cpu_baseline:
            import torch as _t
            from oracle import onnx_ref

            g = onnx_ref.prune_dead_filter_rows(onnx_ref.load_graph(model_bytes))
            cpu_model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            quota = None  # a container's CPU share may be far below the CPUs its affinity mask shows
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    quota = max(1, int(float(q) / float(per) + 0.5))
            except (OSError, ValueError):
                try:
                    q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    if q > 0:
                        quota = max(1, int(q / per + 0.5))
                except (OSError, ValueError):
                    pass
            nproc = min(affinity, quota) if quota else affinity

            def timed(threads, budget_s, max_segments):
                _t.set_num_threads(threads)
                xs = synth.synthetic_segments(max_segments, S, SR)
                onnx_ref.run_graph(g, xs[:8])  # warm
                t1 = time.perf_counter()
                done = 0
                while done < max_segments and (done == 0 or time.perf_counter() - t1 < budget_s):
                    onnx_ref.run_graph(g, xs[done:done + 8])
                    done += len(xs[done:done + 8])
                return done, time.perf_counter() - t1

            # "every thread the box gives this process": a thread count above the share the scheduler actually grants
            # is slower, not faster (256 threads on a 16-CPU share: 0.1 segments/s), so the count is found by
            # doubling from 4 up to nproc while one batch of 8 keeps getting faster -- the best count is the baseline
            best_thr, best_rate, tried = 1, 0.0, []
            thr = min(4, nproc)
            while True:
                dn, tt = timed(thr, 0.0, 8)
                tried.append((thr, round(dn / tt, 1)))
                if dn / tt > best_rate:
                    best_thr, best_rate = thr, dn / tt
                if dn / tt < 0.8 * best_rate or thr >= nproc:
                    break
                thr = min(2 * thr, nproc)
            d_all, t_all = timed(best_thr, 12.0, args.cpu_sample)
            d_one, t_one = timed(1, 8.0, 64)
            _t.set_num_threads(best_thr)
            out["cpu_baseline"] = {"value": round(d_all / t_all, 3), "unit": "segments/s", "cores": int(best_thr), "kind": "port",
                                   "cpu_model": cpu_model, "cpus_visible": int(affinity), "cpu_quota": quota, "thread_counts_tried": tried,
                                   "single_thread": {"value": round(d_one / t_one, 3), "cores": 1, "sample": f"{d_one} segments, {t_one:.1f} s"},
                                   "sample": f"{d_all} segments (batches of 8) of the same synthetic workload through oracle/onnx_ref.py (torch CPU fp32, the "
                                             f"graph as exported minus the DFT rows no mel filter reads, as the GPU plan), {t_all:.1f} s on {best_thr} threads "
                                             f"(the fastest of the thread counts tried)"}
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extras and args.model == "v24":
            # teardown order (DESIGN.md 7, round 3's SIGSEGV): first everything torch holds that names a context's stream or memory
            # (ExternalStream wrappers, events recorded on them, zero-copy views), then a device drain, THEN the contexts
            import gc
            torch.cuda.synchronize()
            ev_pool.clear()
            ev_streams.clear()
            logit_views.clear()
            if ctx_streams is not None:
                ctx_streams.clear()
            gc.collect()
            torch.cuda.synchronize()
            for c_ in ctxs:
                c_.close()
            ctxs.clear()
            gc.collect()
            run_extras()
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


_describe_path = None


def path_for_describe(model_bytes: bytes) -> str:
    global _describe_path
    if _describe_path is None:
        f = tempfile.NamedTemporaryFile(suffix=".onnx", delete=False)
        f.write(model_bytes)
        f.close()
        _describe_path = f.name
    return _describe_path


if __name__ == "__main__":
    main()
