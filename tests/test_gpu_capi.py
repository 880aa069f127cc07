"""Engine-level C ABI entry points not reached through the Classifier mirror: load from a buffer,
I/O metadata, cost report, device-resident inference, device/host top-K variants, step API."""
import ctypes as C
import importlib

import numpy as np
import pytest

import oracle
from gpu_helpers import write_model

pytestmark = pytest.mark.gpu
synth = importlib.import_module("rust-birdnet-onnx_amd.synth")


@pytest.fixture(scope="module")
def small():
    data = synth.birdnet_v30(num_species=200, width=0.5, depth=0.34, emb=1024)
    return data, write_model(data)


def test_load_buffer_io_info_config_cost(bn, small):
    data, path = small
    h = C.c_void_p()
    buf = C.create_string_buffer(data, len(data))
    assert bn.lib.bn_model_load_buffer(C.cast(buf, C.c_void_p), len(data), 0, -1, C.byref(h)) == 0
    io = bn.BnIoInfo()
    assert bn.lib.bn_model_io_info(h, C.byref(io)) == 0
    assert io.input_name == b"input" and io.input_rank == 2 and list(io.input_shape[:2]) == [-1, 160000]
    assert io.n_outputs == 2 and io.output_name[0].value == b"output_0" and io.output_name[1].value == b"output_1"
    assert list(io.output_shape[0][:2]) == [-1, 1024] and list(io.output_shape[1][:2]) == [-1, 200]
    cfg = bn.BnModelConfig()
    assert bn.lib.bn_model_get_config(h, C.byref(cfg)) == 0
    assert (cfg.model_type, cfg.sample_count, cfg.num_species, cfg.embedding_dim, cfg.logits_output, cfg.embedding_output) == \
           (1, 160000, 200, 1024, 1, 0)
    cost = bn.BnModelCost()
    assert bn.lib.bn_model_get_cost(h, C.byref(cost), C.sizeof(cost)) == 0
    assert cost.macs_mfma > 1e7 and cost.weight_bytes > 1e5 and cost.n_launches > 10
    bn.lib.bn_model_free(h)
    # garbage buffer -> BN_ERR_MODEL_LOAD with a message
    bad = C.create_string_buffer(b"\xff" * 64, 64)
    assert bn.lib.bn_model_load_buffer(C.cast(bad, C.c_void_p), 64, 0, -1, C.byref(h)) == 6
    assert bn.last_error()
    assert bn.lib.bn_model_load_buffer(C.cast(buf, C.c_void_p), len(data), 99, -1, C.byref(h)) == 9  # no such device


def test_device_resident_inference_and_step(bn, small):
    import torch
    data, path = small
    m = bn.Model(path)
    ctx = bn.Context(m, 8)
    x = synth.synthetic_segments(5, 160000, 32000)
    want_logits, want_emb = ctx.infer(x)
    xd = torch.from_numpy(x).cuda()
    ctx.infer_device(xd.data_ptr(), 5, sync=True)
    assert ctx.read_output(1, 5).tobytes() == want_logits.tobytes()
    assert ctx.read_output(0, 5).tobytes() == want_emb.tobytes()
    # whole-path step: plan + top-K + D2H, asynchronous then synchronised
    ctx.step_device(xd.data_ptr(), 5, top_k=7, min_confidence=0.05)
    ctx.synchronize()
    lg, ix, cf, ct = ctx.step_results(5)
    assert lg.tobytes() == want_logits.tobytes()
    for r in range(5):
        want = oracle.top_k(lg[r], 7, 0.05)
        assert ix[r, :ct[r]].tolist() == [w[0] for w in want]
        assert cf[r, :ct[r]].tobytes() == np.asarray([w[1] for w in want], np.float32).tobytes()
    # eager (no hipGraph) context gives the same bits
    e, _ = bn.Context(m, 8, bn.BN_CTX_NO_GRAPH).infer(x)
    assert e.tobytes() == want_logits.tobytes()
    # batch limits
    with pytest.raises(bn.EngineError):
        ctx.infer(np.zeros((9, 160000), np.float32))
    assert bn.lib.bn_ctx_max_batch(ctx._h) == 8 and bn.lib.bn_ctx_device_bytes(ctx._h) > 8 * 160000 * 4
    assert ctx.stream() != 0


def test_topk_device_pointer_variant(bn):
    import torch
    rows = np.stack([oracle.random_logits(6522, 900 + s) for s in range(6)])
    d = torch.from_numpy(rows).cuda()
    k = 10
    idx = np.zeros((6, k), np.uint32)
    conf = np.zeros((6, k), np.float32)
    cnt = np.zeros(6, np.uint32)
    u32p = C.POINTER(C.c_uint32)
    st = bn.lib.bn_topk_device(0, C.c_void_p(d.data_ptr()), 6, 6522, k, 1, C.c_float(0.2), k, idx.ctypes.data_as(u32p),
                               conf.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(u32p))
    assert st == 0
    for r in range(6):
        want = oracle.top_k(rows[r], k, 0.2)
        assert idx[r, :cnt[r]].tolist() == [w[0] for w in want]
        assert conf[r, :cnt[r]].tobytes() == np.asarray([w[1] for w in want], np.float32).tobytes()


def test_per_kernel_timing_report(bn, small):
    data, path = small
    ctx = bn.Context(bn.Model(path), 4)
    ctx.infer(synth.synthetic_segments(4, 160000, 32000))
    rows = ctx.time_kernels(4)
    assert len(rows) > 10 and all(us > 0 for _, us, _, _ in rows)
    assert sum(m for _, _, m, _ in rows) > 1e7
    #  every multiply-add of the plan belongs to exactly one launch: the fused MBConv launches carry their expand
    # convs, so the launches sum to the plan's macs_mfma + macs_valu, and bn_ctx_launch_costs splits the same figures by ALU
    cost = bn.Model(path).cost()
    costs = ctx.launch_costs(4)
    assert len(costs) == len(rows)
    assert sum(m for _, _, m, _ in rows) == pytest.approx(4 * (cost.macs_mfma + cost.macs_valu), rel=1e-9)
    assert sum(c[0] for c in costs) == pytest.approx(4 * cost.macs_mfma, rel=1e-9)
    assert sum(c[1] for c in costs) == pytest.approx(4 * cost.macs_valu, rel=1e-9)
    assert sum(c[2] for c in costs) == pytest.approx(4 * cost.recompute_macs, rel=1e-9)
    assert all(r[2] == pytest.approx(c[0] + c[1], rel=1e-12) and r[3] == c[3] for r, c in zip(rows, costs))
    fused = [c for c, l in zip(costs, [l for l in bn.plan_describe(path).splitlines() if l[:3].strip().isdigit()]) if l.split()[1] == "MBCONV"]
    assert fused and all(c[0] > 0 and c[1] > 0 for c in fused)  # expand conv on the matrix cores + depthwise taps on the vector ALU


def test_submit_collect_matches_the_synchronous_call_bit_for_bit(bn, small):
    """bn_infer_submit / bn_infer_collect (two batches in flight on one context, staging by the pool) give the
    bits of bn_infer + bn_topk; a third submit is refused; tickets are single-use."""
    data, path = small
    m = bn.Model(path)
    ctx = bn.Context(m, 8)
    xa = synth.synthetic_segments(8, 160000, 32000)
    xb = synth.synthetic_segments(5, 160000, 32000, first_index=40)
    la, ea = ctx.infer(xa)
    ia, ca, na = ctx.topk(8, 6, 0.02)
    lb, eb = ctx.infer(xb)
    ib, cb, nb = ctx.topk(5, 6, 0.02)
    for _ in range(3):  # slots are reused: every round must see fresh data
        ta = ctx.submit(xa, 6, 0.02)
        tb = ctx.submit([xb[i] for i in range(5)], 6, 0.02)  # a list of separate slices, as the Rust API hands them over
        with pytest.raises(bn.EngineError):
            ctx.submit(xa, 6, 0.02)
        assert bn.last_error().startswith("two batches are already in flight")
        l2, e2, i2, c2, n2 = ctx.collect(tb)  # out of order: waits for both
        l1, e1, i1, c1, n1 = ctx.collect(ta)
        assert l1.tobytes() == la.tobytes() and e1.tobytes() == ea.tobytes()
        assert l2.tobytes() == lb.tobytes() and e2.tobytes() == eb.tobytes()
        assert np.array_equal(n1, na) and np.array_equal(n2, nb)
        for r in range(8):
            assert np.array_equal(i1[r, :n1[r]], ia[r, :na[r]]) and c1[r, :n1[r]].tobytes() == ca[r, :na[r]].tobytes()
        for r in range(5):
            assert np.array_equal(i2[r, :n2[r]], ib[r, :nb[r]]) and c2[r, :n2[r]].tobytes() == cb[r, :nb[r]].tobytes()
        with pytest.raises(bn.EngineError):
            ctx.collect(ta)
    # top_k = 0: logits only; the empty batch is ticket 0 and collects to nothing
    t = ctx.submit(xb, 0)
    l3, e3, i3, c3, n3 = ctx.collect(t)
    assert l3.tobytes() == lb.tobytes() and i3 is None
    tk = C.c_uint64(7)
    assert bn.lib.bn_infer_submit(ctx._h, None, 0, 3, 0, C.c_float(0), C.byref(tk)) == 0 and tk.value == 0
    assert bn.lib.bn_infer_collect(ctx._h, 0, None, None, 0, None, None, None, None, 0) == 0


def test_submit_collect_timeout_then_reuse(bn, small):
    data, path = small
    m = bn.Model(path)
    ctx = bn.Context(m, 8)
    x = synth.synthetic_segments(8, 160000, 32000)
    want, _ = ctx.infer(x)
    t = ctx.submit(x, 3)
    with pytest.raises(bn.EngineError) as ei:
        ctx.collect(t, timeout_ns=1)
    assert ei.value.status == 3
    got = ctx.collect(ctx.submit(x, 3))[0]  # the context drains and is usable again
    assert got.tobytes() == want.tobytes()


def test_context_outlives_its_model_handle(bn, small):
    """bn_model_free before bn_ctx_destroy is legal: the context holds a reference."""
    data, path = small
    h, c = C.c_void_p(), C.c_void_p()
    assert bn.lib.bn_model_load(path.encode(), 0, -1, C.byref(h)) == 0
    assert bn.lib.bn_ctx_create(h, 4, 0, C.byref(c)) == 0
    m = bn.Model(path)
    ref = bn.Context(m, 4).infer(synth.synthetic_segments(2, 160000, 32000))[0]
    bn.lib.bn_model_free(h)
    x = np.ascontiguousarray(synth.synthetic_segments(2, 160000, 32000))
    f32p = C.POINTER(C.c_float)
    ptrs = (f32p * 2)(*[x[i].ctypes.data_as(f32p) for i in range(2)])
    out = np.empty((2, 200), dtype=np.float32)
    assert bn.lib.bn_infer(c, ptrs, 2, out.ctypes.data_as(f32p), None, None, 0) == 0
    assert out.tobytes() == ref.tobytes()
    bn.lib.bn_ctx_destroy(c)  # frees the model too


def test_misaligned_device_input_is_refused_not_fatal(bn):
    """A device pointer that is not 16-byte aligned (a window cut from a device-resident recording at an odd
    offset) must come back as BN_ERR_INVALID_ARG; round 1 aborted the process inside the paired min/max launcher."""
    import torch
    path = write_model(synth.birdnet_v24(num_species=50, width=0.25, depth=0.2, head=64))
    m = bn.Model(path)
    ctx = bn.Context(m, 2)
    buf = torch.zeros(2 * 144000 + 8, dtype=torch.float32, device="cuda")
    for off in (1, 2, 3):
        assert bn.lib.bn_infer_device(ctx._h, C.c_void_p(buf.data_ptr() + 4 * off), 2, 1) == 1
        assert "16-byte aligned" in bn.last_error()
        assert bn.lib.bn_step_device(ctx._h, C.c_void_p(buf.data_ptr() + 4 * off), 2, 3, 0, C.c_float(0), 1) == 1
    assert bn.lib.bn_infer_device(ctx._h, C.c_void_p(buf.data_ptr() + 16), 2, 1) == 0


def test_many_input_buffers_through_one_context(bn, small):
    """The plan always reads its batch from the context's own input buffer (a foreign device pointer is copied in on the
    stream), so the hipGraph cache is keyed by the batch size alone: a caller cycling through 64 device buffers and two
    batch sizes gets ONE capture + instantiate per batch size and the right answer every time; a batch written straight
    into bn_ctx_input_device runs without the copy."""
    import torch
    data, path = small
    m = bn.Model(path)
    ctx = bn.Context(m, 4)
    xs = [synth.synthetic_segments(4, 160000, 32000, first_index=10 * i) for i in range(6)]
    want = [ctx.infer(x)[0] for x in xs]
    base = ctx.stats()
    bufs = [torch.from_numpy(xs[i % 6]).cuda() for i in range(64)]
    for rnd in range(2):
        for i, d in enumerate(bufs):
            n = 4 if (i + rnd) % 3 else 3           # two batch sizes per buffer over the rounds
            ctx.infer_device(d.data_ptr(), n, sync=True)
            assert ctx.read_output(1, n).tobytes() == want[i % 6][:n].tobytes(), (rnd, i)
    st = ctx.stats()
    assert st["capture_fallbacks"] == 0 and st["evictions"] == 0
    assert st["instantiates"] - base["instantiates"] <= 2 and st["cached_graphs"] <= 3, st   # batch 4 (maybe cached by infer) and 3
    assert st["replays"] - base["replays"] == 128 and st["input_copies"] - base["input_copies"] == 128, st
    # zero-copy: the batch produced directly in the context's buffer
    ptr, cap = ctx.input_device()
    assert cap == 4 * 160000 and ptr % 256 == 0
    class _View:
        def __init__(self, p, shape):
            self.__cuda_array_interface__ = {"data": (p, False), "shape": shape, "typestr": "<f4", "version": 2}
    own = torch.as_tensor(_View(ptr, (4, 160000)), device="cuda")
    own.copy_(torch.from_numpy(xs[5]).cuda())
    torch.cuda.synchronize()
    before = ctx.stats()["input_copies"]
    ctx.infer_device(ptr, 4, sync=True)
    assert ctx.stats()["input_copies"] == before
    assert ctx.read_output(1, 4).tobytes() == want[5].tobytes()


def test_context_teardown_order_with_framework_objects_on_its_stream(bn, small):
    """Round 3's SIGSEGV: torch objects that name a context's stream -- an ExternalStream wrapper, events recorded on it,
    tensors copied on it -- are released and the device drained BEFORE the context destroys its hipStream_t; Context.close() destroys
    it now (not whenever the object is collected) and is safe to call twice.  The copies made on the stream are complete and correct."""
    import gc
    import torch
    data, path = small
    m = bn.Model(path)
    ctx = bn.Context(m, 2)
    x = synth.synthetic_segments(2, 160000, 32000)
    want, _ = ctx.infer(x)
    want = want.copy()
    ptr, n = ctx.output_device(m.config.logits_output)

    class _View:
        def __init__(self, p, shape):
            self.__cuda_array_interface__ = {"data": (p, False), "shape": shape, "typestr": "<f4", "version": 2}
    view = torch.as_tensor(_View(ptr, (2, n)), device="cuda")
    es = torch.cuda.ExternalStream(ctx.stream())
    staged = torch.empty((2, n), dtype=torch.float32, device="cuda")
    dx = torch.from_numpy(x).cuda()  # (kept alive until the device has been drained below)
    torch.cuda.synchronize()
    ctx.infer_device(dx.data_ptr(), 2, sync=False)
    with torch.cuda.stream(es):
        staged.copy_(view, non_blocking=True)  # ordered behind the step on the context's own stream
        ev = torch.cuda.Event()
        ev.record(es)
    ev.synchronize()
    assert staged.cpu().numpy().tobytes() == want.tobytes()
    # the order: wrappers / events / views first, drain, then the context
    del ev, es, view, staged, dx
    gc.collect()
    torch.cuda.synchronize()
    ctx.close()
    ctx.close()  # idempotent
    assert getattr(ctx, "_h", None) is None
    # the model outlives its contexts and serves a new one
    again, _ = bn.Context(m, 2).infer(x)
    assert again.tobytes() == want.tobytes()
