"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/*.h declares, and refuses to compute without a gfx950 device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bnh?_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(bn):
    L = ctypes.CDLL(bn.LIB_PATH)
    names = declared("birdnet_hip.h") + declared("birdnet_host.h")
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # the ctypes table in the package covers exactly the declared entry points
    assert sorted(bn.ENGINE_SYMBOLS) == declared("birdnet_hip.h")
    assert sorted(bn.HOST_SYMBOLS) == declared("birdnet_host.h")


def test_abi_version(bn):
    assert bn.lib.bn_abi_version() == 2 == bn.BN_ABI_VERSION


def test_no_device_means_loud_failure_not_cpu_fallback(bn, tmp_path):
    if bn.device_count() > 0:
        pytest.skip("a gfx950 device is present")
    synth = __import__("importlib").import_module("rust-birdnet-onnx_amd.synth")
    p = tmp_path / "m.onnx"
    p.write_bytes(synth.birdnet_v24(num_species=16, width=0.25, depth=0.25, head=32))
    with pytest.raises(bn.EngineError) as e:
        bn.Model(str(p))
    assert e.value.status == 9  # BN_ERR_NO_DEVICE
    with pytest.raises(bn.Error) as e2:
        bn.Classifier.builder().model_path(str(p)).labels(["a"] * 16).build()
    assert e2.value.kind == bn.ErrorKind.ModelLoad
    with pytest.raises(bn.EngineError):
        bn.topk_host([[0.0, 1.0]], 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rust-birdnet-onnx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
