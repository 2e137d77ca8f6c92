"""armada_amd — MI355X-native implementation of Armada's scheduling-round hot path.

The product is the C-ABI shared library ``armada_amd/csrc/libarmada_sched.so`` (hand-written HIP for
gfx950, declared in ``include/armada_sched.h``).  This Python package is the host-side harness around
it: the ctypes binding, a mirror of the reference's NodeDb / PreemptingQueueScheduler interface and the
synthetic workload generators of BASELINE.json.  There is no CPU fallback: loading fails loudly when
the HIP library has not been built.
"""
import os
import sys

from .binding import Config, Library, Scheduler, SchedError  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libarmada_sched.so")

_lib = None


def load_library() -> Library:
    """Load the HIP implementation (prefix ``asched_``). Raises if it has not been built."""
    global _lib
    if _lib is None:
        # PyTorch bundles its own HIP / HSA runtime; the library links /opt/rocm's.  In one process the runtime that initialises second finds no GPU, so when the
        # caller uses torch at all (device buffers for the collectives: sharded.py, queuehash.py, bench.py) torch's has to come up first.
        if "torch" in sys.modules:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.init()
            except Exception:
                pass
        # ASCHED_LIB_PATH: another build of the SAME HIP sources (e.g. -DASCHED_FASTPROF, the per-segment clock profile); never a different backend
        _lib = Library(os.environ.get("ASCHED_LIB_PATH") or LIB_PATH, "asched_")
    return _lib
