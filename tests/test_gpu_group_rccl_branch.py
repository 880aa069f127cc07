"""The RCCL branch of csrc/group.cpp EXECUTED on a one-GPU box.

`bn_group_create` takes the RCCL path only for ranks on pairwise distinct devices, which no box this repository has run on offers, so
`ncclCommInitAll` / the grouped in-place `ncclAllGather` per slab / `ncclCommDestroy` (group.cpp) had never run.  Here a test-only
stand-in for librccl (tests/stubs/rccl_stub.cpp: the rccl.h entry points implemented with device-to-device copies in the stream order
the real call promises) is built, named through BN_RCCL_LIB, and BN_GROUP_FORCE_RCCL=1 takes the branch for three ranks sharing
device 0.  What this checks: the dlsym'd ABI, the dtype codes (7 = f32 logits, 3 = u32 top-K rows), the in-place send offset
(send == recv + rank * count), one grouped collective per slab, stream ordering (results bit-identical to the copy path and to a
single pass), the error path, communicator teardown.  What it cannot check: xGMI, the real library (that stays with
test_group_on_distinct_devices_gathers_over_rccl, skipped below two GPUs).  Runs in a child process: the library resolves RCCL once."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes as C, importlib, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["BN_TEST_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BN_TEST_ROOT"], "tests"))
    from gpu_helpers import write_model
    bn = importlib.import_module("rust-birdnet-onnx_amd")
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    stub = C.CDLL(os.environ["BN_RCCL_LIB"], mode=C.RTLD_GLOBAL)   # the same handle the library dlopens (reference-counted)
    def counters():
        a = (C.c_uint64 * 8)(); stub.rccl_stub_counters(a); return list(a)
    path = write_model(synth.birdnet_v24(num_species=500, width=0.5, depth=0.5, head=256))
    rng = np.random.default_rng(3)
    pcm = (rng.standard_normal(144000 * 7 + 5000) * 4000).astype(np.int16)
    step = 144000 - 48000
    # the copy path (ranks share a device, RCCL not forced) is the reference
    os.environ["BN_GROUP_NO_RCCL"] = "1"
    g0 = bn.Group([bn.Model(path) for _ in range(3)], max_batch=3, contexts_per_device=2)
    assert not g0.uses_rccl()
    want = g0.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=True)
    del g0
    del os.environ["BN_GROUP_NO_RCCL"]
    assert counters()[0] == 0
    grp = bn.Group([bn.Model(path) for _ in range(3)], max_batch=3, contexts_per_device=2)
    assert grp.uses_rccl() and grp.size() == 3, "the forced RCCL branch was not taken"
    assert counters()[0] == 1                                        # one ncclCommInitAll for the three ranks
    got = grp.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=True)
    c = counters()
    # two slabs (logits, packed top-K rows) x three ranks, each slab one grouped collective, every call in place, dtype codes 7 / 3
    assert c[1] == 6 and c[2] == 2 and c[4] == 6 and c[5] == 3 and c[6] == 3 and c[7] == 0, c
    assert got[0].tobytes() == want[0].tobytes(), "logits gathered over the RCCL branch differ from the copy path"
    assert np.array_equal(got[3], want[3])
    for r in range(len(got[3])):
        n = got[3][r]
        assert np.array_equal(got[1][r, :n], want[1][r, :n]) and got[2][r, :n].tobytes() == want[2][r, :n].tobytes()
    # top-K rows only: one collective of three calls
    none, ix, cf, ct = grp.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=False)
    c2 = counters()
    assert none is None and c2[1] - c[1] == 3 and c2[6] - c[6] == 3 and c2[5] == c[5] and np.array_equal(ct, want[3])
    assert grp.stats()["capture_fallbacks"] == 0
    # error path: the second ncclAllGather of the next collective fails -> BN_ERR_BACKEND with the library's message, group still usable
    stub.rccl_stub_fail_next(2)
    try:
        grp.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=True)
        raise SystemExit("the injected ncclAllGather failure was swallowed")
    except RuntimeError as e:
        assert "ncclAllGather" in str(e) and "internal error" in str(e), str(e)
    again = grp.analyze_recording(pcm, step, top_k=5, min_confidence=0.02, want_logits=True)
    assert again[0].tobytes() == want[0].tobytes()
    del grp
    import gc; gc.collect()
    assert counters()[3] == 3, counters()                            # ncclCommDestroy once per rank
    print("RCCL_BRANCH_OK", counters())
''')


def test_rccl_branch_of_the_group_runs_against_a_stub_library(tmp_path):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("needs g++ and the HIP runtime library to build the stand-in")
    lib = str(tmp_path / "librccl_stub.so")
    r = subprocess.run(["g++", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "stubs", "rccl_stub.cpp"),
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, BN_RCCL_LIB=lib, BN_GROUP_FORCE_RCCL="1", BN_TEST_ROOT=ROOT, GPU_MAX_HW_QUEUES="8")
    env.pop("BN_GROUP_NO_RCCL", None)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_BRANCH_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
