"""coxgraph_amd -- MI355X-native TSDF fusion + submap registration behind coxgraph's seams.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
``include/coxgraph_hip.h``), ``host/`` (C++ adapters with the reference's own signatures) and a
ctypes binding used by the tests and ``bench.py``.  The engine is the in-tree shared library
``coxgraph_amd/lib/libcoxgraph_hip.so``; there is no CPU fallback: if it is missing,
:func:`load_engine` raises.
"""
import os

# Four stage streams per integrator (+ the caller's): more hardware queues than the runtime's default of 4, or stages share a
# queue and run back to back.  Only effective if the HIP runtime of this process is not initialised yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import capi  # noqa: E402
from .capi import Engine, Layer, Integrator, RegPoints, Registration, CoxError  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcoxgraph_hip.so")
_engine = None


def load_engine():
    """Load the HIP engine.  Fails loudly when the extension has not been built."""
    global _engine
    if _engine is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is deliberately no CPU fallback).")
        _engine = Engine(LIB_PATH, "cox_")
    return _engine
