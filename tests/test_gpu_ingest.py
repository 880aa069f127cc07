"""Recording-level ingest on the device: i16 -> f32 conversion and chunk_audio
windows materialised by a kernel, against the oracle's restatement of the reference CLI
(src/bin/birdnet-analyze.rs:683-687 read_wav, :707-743 chunk_audio).  Bit-exact: integer / index work
plus a division by a power of two."""
import numpy as np
import pytest

import oracle
from gpu_helpers import synth, write_model

pytestmark = pytest.mark.gpu


def reference_windows(pcm_i16: np.ndarray, S: int, overlap: float, sr: int):
    f32 = (pcm_i16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)  # f32::from(s) / 32768.0
    starts, times = oracle.chunk_plan(len(f32), S, overlap, sr)
    return np.stack([oracle.chunk_fill(f32, S, int(st)) for st in starts]) if len(starts) else np.zeros((0, S), np.float32), starts


@pytest.mark.parametrize("n,S,overlap,sr", [(144000 * 3 + 777, 144000, 0.0, 48000), (500000, 144000, 1.5, 48000), (1000, 144000, 0.0, 48000),
                                            (160000 * 2, 160000, 2.5, 32000), (144000, 144000, 0.0, 48000), (144001, 144000, 2.9, 48000)])
def test_device_windows_match_chunk_audio(bn, n, S, overlap, sr):
    rng = np.random.default_rng(n)
    pcm = rng.integers(-32768, 32768, size=n, dtype=np.int16)
    pcm[:4] = [-32768, 32767, 0, -1]  # extremes of the conversion
    want, starts = reference_windows(pcm, S, overlap, sr)
    step = S - int(np.floor(np.float32(overlap) * np.float32(sr)))
    rec = bn.Recording(pcm)
    assert rec.n_windows(step) == len(starts)
    got = rec.windows(S, step, 0, len(starts))
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # a sub-range, and float input passed through untouched
    if len(starts) > 2:
        assert np.array_equal(rec.windows(S, step, 1, 2), want[1:3])
    recf = bn.Recording((pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32))
    assert np.array_equal(recf.windows(S, step, 0, len(starts)).view(np.uint32), want.view(np.uint32))


def test_empty_recording_and_bad_ranges(bn):
    rec = bn.Recording(np.zeros(0, dtype=np.int16))
    assert rec.n_windows(144000) == 0
    assert rec.windows(144000, 144000, 0, 0).shape == (0, 144000)
    rec = bn.Recording(np.zeros(1000, dtype=np.int16))
    with pytest.raises(bn.EngineError):
        rec.windows(144000, 144000, 0, 2)       # only one window exists
    with pytest.raises(bn.EngineError):
        rec.windows(144000, 0, 0, 1)            # overlap >= segment
    with pytest.raises(bn.EngineError):
        rec.windows(144002, 144002, 0, 1)       # not a multiple of 4


def test_infer_windows_equals_infer_on_host_windows(bn):
    """The whole path fed from an uploaded i16 recording gives the same BITS as bn_infer on host-made windows."""
    S, sr, overlap = 144000, 48000, 1.0
    rng = np.random.default_rng(5)
    t = np.arange(S * 4 + 12345) / sr
    pcm = np.clip(8000 * np.sin(2 * np.pi * 1800 * t) + rng.normal(0, 500, t.shape), -32768, 32767).astype(np.int16)
    want_windows, starts = reference_windows(pcm, S, overlap, sr)
    step = S - int(overlap * sr)
    path = write_model(synth.birdnet_v24(num_species=300, width=0.5))
    model = bn.Model(path)
    ctx = bn.Context(model, 4)
    rec = bn.Recording(pcm)
    G = len(starts)
    assert G == rec.n_windows(step) and G > 4
    got = []
    for first in range(0, G, 4):
        cnt = min(4, G - first)
        lg, _ = ctx.infer_windows(rec, step, first, cnt)
        got.append(lg.copy())
    got = np.concatenate(got)
    ref = np.concatenate([ctx.infer(want_windows[f:f + 4])[0].copy() for f in range(0, G, 4)])
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(bn.EngineError):
        ctx.infer_windows(rec, step, 0, 5)  # exceeds the context's max batch


def test_predict_recording_through_the_host_mirror(bn, tmp_path):
    """Classifier::predict_recording == predict_batch over chunk_audio windows made on the host (same bits, same
    top-K, start times as chunk_audio reports them)."""
    S, sr, overlap = 144000, 48000, 0.5
    rng = np.random.default_rng(9)
    t = np.arange(S * 5 + 999) / sr
    pcm = np.clip(6000 * np.sin(2 * np.pi * 3100 * t) + rng.normal(0, 900, t.shape), -32768, 32767).astype(np.int16)
    windows, starts = reference_windows(pcm, S, overlap, sr)
    _, times = oracle.chunk_plan(len(pcm), S, overlap, sr)
    path = write_model(synth.birdnet_v24(num_species=300, width=0.5))
    labels = [f"Genus species{i}_Common {i}" for i in range(300)]
    cl = bn.ClassifierBuilder().model_path(path).labels(labels).top_k(5).with_rocm(0).build()
    ctx = cl.create_batch_context(4)
    got = cl.predict_recording(ctx, pcm, overlap)
    assert len(got) == len(starts)
    ref = []
    for f in range(0, len(starts), 4):
        ref += cl.predict_batch_with_context(ctx, [w for w in windows[f:f + 4]])
    for (tm, g), r, t_ref in zip(got, ref, times):
        assert np.float32(tm) == np.float32(t_ref)
        assert np.array_equal(np.asarray(g.raw_scores, dtype=np.float32).view(np.uint32), np.asarray(r.raw_scores, dtype=np.float32).view(np.uint32))
        assert [(p.index, p.species, np.float32(p.confidence)) for p in g.predictions] == [(p.index, p.species, np.float32(p.confidence)) for p in r.predictions]
    # a sub-range and the overlap validation
    part = cl.predict_recording(ctx, pcm, overlap, first_chunk=2, count=3)
    assert [np.float32(a) for a, _ in part] == [np.float32(x) for x in times[2:5]]
    with pytest.raises(bn.Error) as e:
        cl.predict_recording(ctx, pcm, 3.0)
    assert e.value.kind == bn.ErrorKind.Inference


def _sharded_worker(rank, world, port, model_path, pcm_path, out_dir, overlap):
    import importlib as _il
    import os as _os
    import sys as _sys

    import torch  # noqa: F401  (first: one HIP runtime per process)
    import torch.distributed as dist

    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    _os.environ["MASTER_ADDR"] = "127.0.0.1"
    _os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bn_ = _il.import_module("rust-birdnet-onnx_amd")
    d = _il.import_module("rust-birdnet-onnx_amd.distributed")
    model = bn_.Model(model_path, device=0)
    pcm = np.load(pcm_path)
    logits, idx, conf, cnt = d.analyze_recording_sharded(bn_, model, pcm, overlap, batch=4, streams=2, top_k=5, dist=dist, gather="logits")
    np.savez(_os.path.join(out_dir, f"rank{rank}.npz"), logits=logits, idx=idx, conf=conf, cnt=cnt)
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [0.0, 1.0])
def test_sharded_recording_ingest_two_ranks_equal_single_pass(bn, tmp_path, overlap):
    """SURVEY 8(e) + 8(f)1: two ranks (gloo, both on this GPU) each upload their slice of an i16 recording, cut
    windows on the device, and all-gather; every rank holds exactly the single-process result, which itself equals
    bn_infer over chunk_audio windows made on the host."""
    import socket

    import torch.multiprocessing as mp

    dmod = __import__("importlib").import_module("rust-birdnet-onnx_amd.distributed")
    S, sr = 144000, 48000
    rng = np.random.default_rng(21)
    t = np.arange(S * 9 + 4321) / sr
    pcm = np.clip(7000 * np.sin(2 * np.pi * 2200 * t) + rng.normal(0, 700, t.shape), -32768, 32767).astype(np.int16)
    path = write_model(synth.birdnet_v24(num_species=300, width=0.5))
    model = bn.Model(path)
    want_logits, want_idx, want_conf, want_cnt = dmod.analyze_recording_sharded(bn, model, pcm, overlap, batch=4, streams=2, top_k=5, gather="logits")
    windows, starts = reference_windows(pcm, S, overlap, sr)
    ctx = bn.Context(model, 4)
    ref = np.concatenate([ctx.infer(windows[f:f + 4])[0].copy() for f in range(0, len(starts), 4)])
    assert want_logits.shape == ref.shape and np.array_equal(want_logits.view(np.uint32), ref.view(np.uint32))
    assert (want_cnt <= 5).all() and (want_idx[want_cnt > 0, 0] == np.argmax(ref, axis=1)[want_cnt > 0]).all()
    # slices: what each rank uploads covers exactly its windows
    step = S - int(overlap * sr)
    G = len(starts)
    for r in range(2):
        lo, hi = dmod.shard_range(G, r, 2)
        a, b = dmod.shard_sample_range(len(pcm), S, step, lo, hi)
        assert a == lo * step and b == min(len(pcm), (hi - 1) * step + S)
    np.save(tmp_path / "pcm.npy", pcm)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, path, str(tmp_path / "pcm.npy"), str(tmp_path), overlap), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["logits"].view(np.uint32), want_logits.view(np.uint32))
        assert np.array_equal(got["idx"], want_idx) and np.array_equal(got["cnt"], want_cnt)
        assert np.array_equal(got["conf"].view(np.uint32), want_conf.view(np.uint32))


def _rccl_rehearsal_worker(rank, port, model_path, pcm_path, out_dir):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    bn = __import__("importlib").import_module("rust-birdnet-onnx_amd")
    dmod = __import__("importlib").import_module("rust-birdnet-onnx_amd.distributed")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    try:
        model = bn.Model(model_path)
        lg, ix, cf, ct = dmod.analyze_recording_sharded(bn, model, np.load(pcm_path), 1.0, batch=4, streams=2, top_k=5, dist=dist, gather="logits")
        np.savez(os.path.join(out_dir, "rccl.npz"), logits=lg, idx=ix, conf=cf, cnt=ct)
    finally:
        dist.destroy_process_group()


def test_sharded_recording_device_resident_gather_over_rccl(bn, tmp_path):
    """The torchrun path with backend nccl (= RCCL): every step's logits rows go device to device into the rank's slab
    of the gather buffer on the context's stream and ONE all_gather_into_tensor runs in place -- rehearsed with a
    single-rank RCCL process group (the one-GPU box cannot host two RCCL ranks), bit-identical to the plain run."""
    import socket

    import torch.multiprocessing as mp

    dmod = __import__("importlib").import_module("rust-birdnet-onnx_amd.distributed")
    S, sr = 144000, 48000
    rng = np.random.default_rng(5)
    pcm = np.clip(rng.normal(0, 3000, S * 5 + 999), -32768, 32767).astype(np.int16)
    path = write_model(synth.birdnet_v24(num_species=300, width=0.5))
    want = dmod.analyze_recording_sharded(bn, bn.Model(path), pcm, 1.0, batch=4, streams=2, top_k=5, gather="logits")
    np.save(tmp_path / "pcm.npy", pcm)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_rehearsal_worker, args=(port, path, str(tmp_path / "pcm.npy"), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "rccl.npz")
    assert np.array_equal(got["logits"].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(got["idx"], want[1]) and np.array_equal(got["cnt"], want[3]) and got["conf"].tobytes() == want[2].tobytes()


@pytest.mark.parametrize("src,dst,dtype", [(44100, 48000, np.int16), (48000, 32000, np.int16), (22050, 48000, np.float32), (96000, 48000, np.int16),
                                           (44100, 32000, np.float32)])
def test_device_resampler_matches_the_oracle(bn, src, dst, dtype):
    """polyphase resampling on the device == oracle/resample.py (fp32 tolerance: the kernel
    accumulates in f32 with fmaf, the oracle in f64)."""
    from oracle import resample as R
    rng = np.random.default_rng(src + dst)
    n = src * 2 + 137
    t = np.arange(n) / src
    x = 0.5 * np.sin(2 * np.pi * 1500.0 * t) + 0.2 * np.sin(2 * np.pi * 5200.0 * t) + 0.05 * rng.standard_normal(n)
    pcm = np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16) if dtype == np.int16 else x.astype(np.float32)
    want = R.resample(pcm, src, dst)
    rec = bn.Recording(pcm, src_rate=src, dst_rate=dst)
    assert rec.n_samples == len(want) == -(-n * dst // src)
    got = rec.read_f32()
    assert np.abs(got - want).max() <= 3e-6, float(np.abs(got - want).max())
    # same rate: a plain upload, windows bit-identical to chunk_audio
    if dtype == np.int16:
        same = bn.Recording(pcm, src_rate=dst, dst_rate=dst)
        w, _ = reference_windows(pcm, 144000, 0.0, 48000)
        assert np.array_equal(same.windows(144000, 144000, 0, len(w)), w)


def test_resampled_recording_through_the_network(bn):
    """A 44.1 kHz int16 recording analysed by the 48 kHz model: device resampling + windows + plan == the same
    windows made on the host from the oracle's resampled signal, within the network tolerance."""
    from gpu_helpers import ATOL, RTOL
    from oracle import resample as R
    src, dst, S = 44100, 48000, 144000
    rng = np.random.default_rng(77)
    t = np.arange(src * 7) / src
    pcm = np.clip(9000 * np.sin(2 * np.pi * 2100 * t) + rng.normal(0, 600, t.shape), -32768, 32767).astype(np.int16)
    model = bn.Model(write_model(synth.birdnet_v24(num_species=300, width=0.5)))
    ctx = bn.Context(model, 4)
    rec = bn.Recording(pcm, src_rate=src, dst_rate=dst)
    G = rec.n_windows(S)
    got = np.concatenate([ctx.infer_windows(rec, S, f, min(4, G - f))[0].copy() for f in range(0, G, 4)])
    y = R.resample(pcm, src, dst)
    starts, _ = oracle.chunk_plan(len(y), S, 0.0, dst)
    win = np.stack([oracle.chunk_fill(y, S, int(s)) for s in starts])
    ref = np.concatenate([ctx.infer(win[f:f + 4])[0].copy() for f in range(0, G, 4)])
    assert got.shape == ref.shape
    assert np.all(np.abs(got - ref) <= ATOL + RTOL * np.abs(ref))
    assert (np.argmax(got, axis=1) == np.argmax(ref, axis=1)).all()


def test_asynchronous_upload_is_chunked_and_bit_identical(bn, monkeypatch):
    """bn_recording_create_async: the recording crosses the bus chunk by chunk on a thread of its own while the first windows are already
    being cut and analysed; a call that reads samples which have not arrived yet blocks until they have.  Tiny chunks (1 MiB: the 4.3 MB
    recording is five of them), windows consumed in time order and out of order, read-back: everything bit-identical to the synchronous
    upload.  Freeing the recording right away (upload still running) joins the thread."""
    S, sr = 144000, 48000
    rng = np.random.default_rng(21)
    pcm = rng.integers(-20000, 20000, size=S * 15 + 777, dtype=np.int16)
    step = S - sr
    path = write_model(synth.birdnet_v24(num_species=300, width=0.5))
    model = bn.Model(path)
    ctx = bn.Context(model, 4)
    sync = bn.Recording(pcm)
    G = sync.n_windows(step)
    want = np.concatenate([ctx.infer_windows(sync, step, f, min(4, G - f))[0].copy() for f in range(0, G, 4)])
    monkeypatch.setenv("BN_UPLOAD_CHUNK_MB", "1")
    rec = bn.Recording(pcm, async_upload=True)
    assert rec.n_windows(step) == G
    got = np.concatenate([ctx.infer_windows(rec, step, f, min(4, G - f))[0].copy() for f in range(0, G, 4)])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the LAST windows first on a fresh recording (the call waits for the final chunk), then the first ones
    rec2 = bn.Recording(pcm, async_upload=True)
    tail, _ = ctx.infer_windows(rec2, step, G - 3, 3)
    assert np.array_equal(tail.view(np.uint32), want[G - 3:].view(np.uint32))
    head, _ = ctx.infer_windows(rec2, step, 0, 4)
    assert np.array_equal(head.view(np.uint32), want[:4].view(np.uint32))
    rec2.wait()
    # windows on the host and the step path (top-K fused) read the same samples
    assert np.array_equal(rec2.windows(S, step, 2, 3), sync.windows(S, step, 2, 3))
    ctx.step_windows(rec2, step, 4, 4, 5, 0.02)
    ctx.synchronize()
    lg = ctx.step_results(4)[0]
    assert np.array_equal(lg.view(np.uint32), want[4:8].view(np.uint32))
    # f32 recordings and read-back through the same path
    f32 = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    recf = bn.Recording(f32, async_upload=True)
    assert np.array_equal(recf.read_f32(len(f32) - 1000, 1000), f32[-1000:])
    # free while the upload may still be running: joins, no crash; an empty recording has nothing to upload
    for _ in range(3):
        r3 = bn.Recording(pcm, async_upload=True)
        del r3
    assert bn.Recording(np.zeros(0, dtype=np.int16), async_upload=True).n_windows(step) == 0
    with pytest.raises(ValueError):  #  the combination used to drop async_upload silently
        bn.Recording(f32, src_rate=44100, dst_rate=48000, async_upload=True)
