"""pytest configuration.

`-m "not gpu"` (runs in the build container, no GPU): the CPU oracle against the reference's golden
vectors, host logic, and that the HIP library loads and exports every symbol of include/armada_sched.h.
`-m gpu` (runs on the MI355X box): parity of the HIP path against the oracle and the goldens, through the C ABI.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _oracle_path():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return path


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure only)."""
    from armada_amd.binding import Library
    return Library(_oracle_path(), "oracle_")


def _torch_cuda_first():
    """PyTorch ships its own HIP / HSA runtime next to /opt/rocm's, which the library links: in one process the runtime that initialises SECOND finds no GPU.  The
    tests that hand the library a torch tensor on the GPU need torch's to be the first (bench.py does the same by calling torch.cuda.set_device before loading the library)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def hip_lib():
    """The product: the HIP implementation.  No fallback — missing library is a hard failure.
    ASCHED_GPU_TESTS_DRY_RUN=1 (build container only) substitutes the CPU build of the device code so that the *test code* of the
    `-m gpu` tests can be exercised without a GPU; it proves nothing about the HIP library and is never set by the driver."""
    _torch_cuda_first()
    if os.environ.get("ASCHED_GPU_TESTS_DRY_RUN") == "1":
        from armada_amd.binding import Library
        here = os.path.join(ROOT, "tests", "hostsim")
        subprocess.check_call(["make", "-C", here])
        return Library(os.path.join(here, "libhostsim.so"), "asched_")
    import armada_amd
    return armada_amd.load_library()


@pytest.fixture(scope="session")
def hostsim_lib():
    """The device control code compiled for the CPU (tests/hostsim): logic checks without a GPU, never shipped."""
    from armada_amd.binding import Library
    here = os.path.join(ROOT, "tests", "hostsim")
    import fcntl
    with open(os.path.join(here, ".build.lock"), "w") as lock:   # pytest-xdist workers reach this at the same time: one rebuilds, the others wait (a half-written .so is "file too short")
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", here])
    return Library(os.path.join(here, "libhostsim.so"), "asched_")

