"""pytest configuration: registers the `gpu` marker and exposes repo-root imports."""
import importlib
import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: load it BEFORE libbirdnet_hip.so pulls in the system one,
    import torch  # noqa: F401  (two different libamdhip64 copies in one process cannot both see the GPU)
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _pkg():
    """The product package (directory name carries a hyphen, so importlib by string)."""
    return importlib.import_module("rust-birdnet-onnx_amd")


@pytest.fixture(scope="session")
def bn():
    return _pkg()
