"""Bringing up the handle's communicator from a `torch.distributed` process group (the Python harness' control plane; a Go scheduler hands the
unique id around through its own).  Two ways, same entry points afterwards (`fit_select_batch_sharded`, `round_exchange`):

* `init_rccl(s, dist)`     — the product path on GPUs: rank 0 asks the LIBRARY for an ncclUniqueId, the process group only broadcasts those 128 bytes,
                             every rank calls asched_comm_init; from then on the all-reduces are ncclAllReduce calls the library itself enqueues on the
                             handle's stream (RCCL over xGMI).
* `init_external(s, dist)` — any backend torch.distributed offers (gloo in the CPU tests): the library calls back with the address of the words; the
                             transport wraps them in a tensor without copying when they are host memory (the CPU build of the tests) and stages them
                             through the host when they are device memory and the backend cannot reduce device tensors.
"""
from __future__ import annotations

import ctypes

import numpy as np

_OPS = None
_HIP = None


def hip_memcpy(dst: int, src: int, nbytes: int, kind: int) -> int:
    """hipMemcpy of the HIP runtime this process already holds (torch's, which the library binds to as well); kind: 1 H2D, 2 D2H, 3 D2D"""
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _HIP.hipMemcpy.restype = ctypes.c_int
    return int(_HIP.hipMemcpy(ctypes.c_void_p(dst), ctypes.c_void_p(src), nbytes, kind))


def _ops(dist):
    global _OPS
    if _OPS is None:
        _OPS = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}
    return _OPS


def init_rccl(s, dist, device=None):
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = s.comm_unique_id() if rank == 0 else bytes(128)
    t = torch.tensor(list(uid), dtype=torch.uint8, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(t, src=0)
    s.comm_init(bytes(t.cpu().tolist()), rank, world)


def init_external(s, dist, device_memory: bool = False):
    """device_memory: the library's buffers are GPU memory (the HIP library): staged through a pinned host tensor for host-only backends"""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()

    def allreduce(ptr: int, count: int, op: int) -> int:
        if count <= 0:
            return 0
        host_words = bool(op & 16)   # ASCHED_ALLREDUCE_HOST_WORDS: the exchange words of a sharded round's wide pass — host memory, and the round kernel is RUNNING (no device sync)
        op &= 15
        if not device_memory or host_words:
            arr = np.ctypeslib.as_array((ctypes.c_int64 * count).from_address(ptr))
            t = torch.from_numpy(arr)            # shares the library's words: reduced in place
            dist.all_reduce(t, op=_ops(dist)[op])
            return 0
        host = torch.empty(count, dtype=torch.int64).pin_memory()
        torch.cuda.synchronize()
        if hip_memcpy(host.data_ptr(), ptr, count * 8, 2) != 0:   # device -> host
            return 1
        dist.all_reduce(host, op=_ops(dist)[op])
        return 0 if hip_memcpy(ptr, host.data_ptr(), count * 8, 1) == 0 else 1   # host -> device

    s.comm_init_external(allreduce, rank, world)
