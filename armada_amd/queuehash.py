"""BASELINE north_star's multi-GPU variant, behind this module only: queues hashed across ranks, ONE all-reduce of per-node committed resources.

**Approximate by construction** (DESIGN.md 7): in the reference every placement reads the node state all earlier placements of ALL queues
wrote, so queues scheduled side by side on replicas cannot see each other's binds.  What this mode does, per round:

  1. every rank holds a full replica of the pool (nodes, running jobs, every queue's allocation and demand: fair shares are the global ones) and runs
     the ordinary round with the queued lists of ITS queues only (queue q belongs to rank q mod G; the global rate limiter's burst is split evenly);
  2. ONE `all_reduce(SUM)` of the committed matrix — requests of the newly scheduled jobs per node, N x R int64 (3.2 MB at 100k nodes x 4 resources over
     RCCL / xGMI; gloo in the CPU tests) — plus the placement vector (job -> node + 1, every job is placed by at most one rank), the free matrix (MIN: the
     replicas differ where a rank's own jobs preempted something) and the preempted flags (MAX);
  3. nodes whose summed commitments exceed what was free are *conflicts*: their new jobs are re-admitted in a global deterministic order (queue index,
     position in the queue) while they fit; the rest are dropped from this round (they stay queued for the next one).

The result is a feasible assignment (no node oversubscribed at priority -2, checked), NOT the reference's: `compare()` counts the jobs whose outcome
differs from an exact round (the oracle / the single-GPU library) — that count is what the tests and DESIGN.md report; no "identical assignments" claim
is made for this mode.  The exact ways to use several GPUs are pools (multipool.py) and the node-sharded wide queries (sharded.py).
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import numpy as np

from . import workloads as W
from .binding import EVICTED_PRIORITY, Scheduler


def owner(queue: int, world: int) -> int:
    return queue % world


class QueueHashRound:
    def __init__(self, lib, wl: W.Workload, rank: int, world: int, dist=None, device: Optional[str] = None):
        self.wl, self.rank, self.world, self.dist, self.device = wl, rank, world, dist, device
        local = copy.copy(wl)
        local.queued = [list(q) if owner(i, world) == rank else [] for i, q in enumerate(wl.queued)]
        if not wl.rate_inf:
            local.global_burst = wl.global_burst // world + (1 if rank < wl.global_burst % world else 0)
        self.local = local
        self.s: Scheduler = W.load(lib, local)

    def _reduce(self, a: np.ndarray, op: str = "SUM") -> np.ndarray:
        if self.dist is None or self.world == 1:
            return a
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        if self.device:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return t.cpu().numpy()

    def run(self) -> Dict:
        wl, s = self.wl, self.s
        W.prepare(s, self.local)
        res = s.schedule_round()
        level0 = s.priorities.index(EVICTED_PRIORITY)
        alloc_after = s.get_nodes_alloc()[:, level0, :]                                   # [N][R] priority -2 allocatable after this rank's round
        new = [(int(j), int(n)) for j, n in res.scheduled.items() if wl.job_node[j] < 0]  # newly scheduled jobs (not rescheduled evicted ones)
        committed = np.zeros((wl.num_nodes, wl.job_req.shape[1]), dtype=np.int64)
        place = np.zeros(wl.num_jobs, dtype=np.int32)
        for j, n in new:
            committed[n] += wl.job_req[j]
            place[j] = n + 1
        free = alloc_after + committed                                                     # what was free before this rank's new jobs
        total = self._reduce(committed)                                                    # THE exchange of the round: per-node committed resources
        place = self._reduce(place)
        # the replicas differ where a rank's own jobs preempted or displaced something: every rank resolves against the same (most conservative) view
        free = self._reduce(free, "MIN")
        pre = np.zeros(wl.num_jobs, dtype=np.int32)
        for j in res.preempted:
            pre[int(j)] = 1
        pre = self._reduce(pre, "MAX")
        conflict = np.nonzero((total > free).any(axis=1))[0]
        final = {int(j): int(place[j]) - 1 for j in np.nonzero(place)[0]}
        dropped = []
        if len(conflict):
            cset = set(int(n) for n in conflict)
            qpos = {}
            for q, lst in enumerate(wl.queued):
                for p, j in enumerate(lst):
                    qpos[int(j)] = (q, p)
            by_node: Dict[int, list] = {}
            for j, n in final.items():
                if n in cset:
                    by_node.setdefault(n, []).append(j)
            for n, jobs in by_node.items():
                room = free[n].copy()
                for j in sorted(jobs, key=lambda x: qpos.get(x, (1 << 30, x))):
                    if (wl.job_req[j] <= room).all():
                        room -= wl.job_req[j]
                    else:
                        dropped.append(j); del final[j]
        used = np.zeros_like(committed)
        for j, n in final.items():
            used[n] += wl.job_req[j]
        assert (used <= free).all(), "queue-hash resolution left a node oversubscribed"
        return dict(scheduled=final, dropped=sorted(dropped), conflicts=int(len(conflict)), local_new=len(new), preempted=[int(j) for j in np.nonzero(pre)[0]])

    @staticmethod
    def compare(result: Dict, exact_scheduled: Dict[int, int], wl: W.Workload) -> Dict:
        """how far the approximate round is from an exact one: jobs scheduled by one and not the other, jobs scheduled by both on different nodes"""
        a = {j: n for j, n in result["scheduled"].items()}
        b = {int(j): int(n) for j, n in exact_scheduled.items() if wl.job_node[int(j)] < 0}
        both = set(a) & set(b)
        return dict(exact_new=len(b), approx_new=len(a), only_exact=len(set(b) - set(a)), only_approx=len(set(a) - set(b)),
                    same_node=sum(1 for j in both if a[j] == b[j]), other_node=sum(1 for j in both if a[j] != b[j]))

    def close(self):
        self.s.close()
