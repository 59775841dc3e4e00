import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    # device_count() does not initialise the GPU on this image
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libcoxoracle.so, prefix coxo_) -- the checker, never the product."""
    from coxgraph_amd.capi import Engine
    lib = os.path.join(ROOT, "oracle", "libcoxoracle.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("cox_oracle.hpp", "cox_oracle_mesh.hpp", "cox_oracle_submap.hpp", "cox_oracle_projective.hpp", "cox_oracle_capi.cpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return Engine(lib, "coxo_")


@pytest.fixture(scope="session")
def hip():
    """The HIP engine through the C ABI; no fallback."""
    import coxgraph_amd
    return coxgraph_amd.load_engine()


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """On a GPU box let PyTorch create its HIP context before the engine does (the order bench.py uses)."""
    if _gpu_available():
        import torch
        torch.cuda.init()
        torch.zeros(1, device="cuda")
    yield
