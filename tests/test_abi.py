"""CPU-side checks of the product library: it loads, exports the whole C ABI, and refuses to run without a gfx950."""
import ctypes
import os

import pytest

import armada_amd
from armada_amd.binding import ALL_SYMBOLS, OPTIONAL_SYMBOLS, Config, SchedError, Scheduler


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_symbol(hip_lib):
    assert sorted(hip_lib.exported()) == sorted(ALL_SYMBOLS)
    # and the header declares nothing the binding does not know about
    hdr = open(os.path.join(os.path.dirname(armada_amd.__file__), "..", "include", "armada_sched.h")).read()
    import re
    declared = set(re.findall(r"ASCHED_FN\((\w+)\)\(", hdr))
    assert declared == set(ALL_SYMBOLS)


def test_oracle_exports_every_symbol(oracle_lib):
    # the single-process checker has no communicator: the collectives that run on one are the product's (and the CPU build's, over an external transport)
    assert sorted(oracle_lib.exported()) == sorted(set(ALL_SYMBOLS) - OPTIONAL_SYMBOLS)


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback(hip_lib):
    """without a gfx950 device the product must fail loudly, not fall back to a CPU path"""
    cfg = Config(num_resources=2, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0, 1.0])
    with pytest.raises(SchedError):
        Scheduler(hip_lib, cfg)


def test_missing_library_is_loud(tmp_path):
    from armada_amd.binding import Library
    with pytest.raises(FileNotFoundError):
        Library(str(tmp_path / "nope.so"))
