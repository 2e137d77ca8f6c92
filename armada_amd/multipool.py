"""Pool-per-GPU driver: the one axis along which the scheduling round shards exactly.

The reference builds one NodeDb per pool and schedules the pools one after another
(internal/scheduler/scheduling/scheduling_algo.go:165); pools share no node and no per-round state.  Here pool g is one
handle (`asched_config.device = g`) in process g; there is NO data-path collective — `torch.distributed` carries only the
barrier around the timed region and the max-over-ranks reduction of the elapsed time.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Tuple

from . import workloads as W


def pool_seed(base_seed: int, rank: int) -> int:
    """pools are different clusters: each rank draws its own pool from the generator"""
    return base_seed + rank


def timed_rounds(s, wl, steps: int, warmup: int, barrier: Callable[[], None], sync: Callable[[], None] = lambda: None) -> Tuple[List[float], List[float], object]:
    """`warmup` untimed + `steps` timed rounds of one pool.  round_prepare (input build) is outside the timed region."""
    lat, dev_ms, res = [], [], None
    timed_rounds.prepare_s = 0.0
    timed_rounds.control_ms = []   # per timed round: ms inside the persistent k_control launches (HIP events on the launch stream)
    for i in range(warmup + steps):
        tp = time.perf_counter()
        W.prepare(s, wl)
        timed_rounds.prepare_s += time.perf_counter() - tp
        barrier()
        t0 = time.perf_counter()
        res = s.schedule_round()
        sync()
        dt = time.perf_counter() - t0
        if i >= warmup:
            lat.append(dt)
            dev_ms.append(s.kernel_times()["round_ms"])
            timed_rounds.control_ms.append(s.round_timing()["control_ms"])
    barrier()
    return lat, dev_ms, res


def aggregate(world: int, steps: int, local_total_s: float, allreduce_max: Optional[Callable[[float], float]]) -> Tuple[float, float]:
    """whole-job rounds/s = pool-rounds of all ranks / slowest rank's time"""
    total = allreduce_max(local_total_s) if allreduce_max else local_total_s
    return world * steps / total, total
