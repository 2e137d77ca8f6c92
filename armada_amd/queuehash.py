"""BASELINE north_star's multi-GPU variant, behind this module only: queues hashed across ranks, ONE all-reduce of per-node committed resources,
conflicts replayed in global order.

**Approximate by construction** (DESIGN.md 7): in the reference every placement reads the node state all earlier placements of ALL queues
wrote, so queues scheduled side by side on replicas cannot see each other's binds.  What this mode does, per round:

  1. every rank holds a full replica of the pool (nodes, running jobs, every queue's allocation and demand: fair shares are the global ones) and runs
     the ordinary round with the queued lists of ITS queues only (queue q belongs to rank q mod G; the global rate limiter's burst is split evenly);
  2. the library fills, ON THE DEVICE, one int64 buffer [N x R committed resources of this rank's newly scheduled jobs | one word per job: node, priority
     level, "preempted"] (asched_round_delta; 3.2 MB + 12 MB at 100k nodes x 4 resources x 1.5M jobs) and ONE `all_reduce(SUM)` runs on that buffer in
     place (RCCL over xGMI on the GPU box, gloo on host memory in the CPU tests) — the single exchange of the round;
  3. the library reads the reduced buffer on the device (asched_round_delta_resolve): a node whose summed commitments exceed what is free there once every
     rank's preemptions are applied is a CONFLICT; new jobs on other nodes are accepted as placed; new jobs on conflict nodes, and every member of a gang
     that has one there, are the replay set;
  4. ordered replay: the accepted state (running jobs + accepted placements - preempted jobs) is loaded and the replay set goes through one more ordinary
     round (global DRF order by `Less`, rate limiters holding what the accepted jobs left; a whole round because urgency preemption relies on the round's
     oversubscribed evictor) — on every rank identically, so no second exchange is needed.  What the replay cannot place stays queued for the next
     round ("dropped").

The result is a feasible assignment (no node oversubscribed at priority -2: the replay is an ordinary round on the accepted state, and accepted nodes
are conflict-free by construction), NOT the reference's: `compare()` counts the jobs whose outcome differs from an exact round (the oracle / the single-GPU
library) — that count is what the tests, `bench.py --mode queue-hash` and DESIGN.md report; no "identical assignments" claim is made for this mode.  The
exact ways to use several GPUs are pools (multipool.py) and the node-sharded wide queries (sharded.py).
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import numpy as np

from . import workloads as W
from .binding import Scheduler


def owner(queue: int, world: int) -> int:
    return queue % world


class QueueHashRound:
    def __init__(self, lib, wl: W.Workload, rank: int, world: int, dist=None, device: Optional[str] = None, replay: bool = True):
        self.wl, self.rank, self.world, self.dist, self.device, self.replay = wl, rank, world, dist, device, replay
        local = copy.copy(wl)
        local.queued = [list(q) if owner(i, world) == rank else [] for i, q in enumerate(wl.queued)]
        if not wl.rate_inf:
            local.global_burst = wl.global_burst // world + (1 if rank < wl.global_burst % world else 0)
        self.local = local
        self.s: Scheduler = W.load(lib, local)
        self.in_library = False
        self.timing: Dict[str, float] = {}

    def comm_init(self, transport: str = "rccl"):
        """the handle's own communicator ("rccl": ncclCommInitRank inside the library; "external": the library calls back into torch.distributed)"""
        from . import comm
        if transport == "rccl":
            comm.init_rccl(self.s, self.dist, device=self.device)
        else:
            comm.init_external(self.s, self.dist, device_memory=self.device is not None)
        self.in_library = True

    def _sync_device(self):
        if self.device is not None:
            import torch
            torch.cuda.synchronize()

    def run(self) -> Dict:
        import time
        import torch
        wl, s = self.wl, self.s
        t0 = time.perf_counter()
        W.prepare(s, self.local)
        res = s.schedule_round()
        t1 = time.perf_counter()
        # THE exchange of the round: the buffer is filled on the device and reduced in place
        if self.in_library:
            # delta -> ncclAllReduce(SUM) -> resolve as one stream-ordered sequence on the handle's stream (asched_round_exchange)
            summary, node, prio, rp = s.round_exchange()
        else:
            buf = torch.zeros(max(s.round_delta_words(), 1), dtype=torch.int64, device=self.device or "cpu")
            self._sync_device()      # torch's fill on its stream before the library's kernels on the handle's stream
            s.round_delta(buf.data_ptr())
            if self.dist is not None and self.world > 1:
                self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
                self._sync_device()  # ... and the collective before the library reads the reduced words (round-3 ADVICE: the ordering was implicit)
            summary, node, prio, rp = s.round_delta_resolve(buf.data_ptr())
        t2 = time.perf_counter()
        was_running = wl.job_node >= 0
        final = {int(j): int(node[j]) for j in np.nonzero(~was_running & (node >= 0))[0]}
        preempted = set(int(j) for j in np.nonzero(was_running & (node < 0))[0])
        dropped = [int(j) for j in np.nonzero(rp)[0]]
        replayed = 0
        if self.replay and summary["replay"]:
            # ordered replay on the accepted state: one ordinary QueueScheduler pass over the replay set in global DRF order (every rank runs the same one)
            W.set_jobs(s, wl, job_node=node.astype(np.int32), job_run_prio=prio.astype(np.int32))
            queued = [[int(j) for j in q if rp[j]] for q in wl.queued]
            acc_q = np.bincount(wl.job_queue[np.fromiter(final.keys(), dtype=np.int64, count=len(final))], minlength=wl.num_queues) if final else np.zeros(wl.num_queues, dtype=np.int64)
            nq = wl.num_queues
            s.round_prepare(wl.queue_weight, queued,
                            global_tokens=float(max(wl.global_burst - len(final), 0)), global_burst=wl.global_burst, global_rate_inf=wl.rate_inf,
                            queue_tokens=[float(max(wl.queue_burst - int(acc_q[q]), 0)) for q in range(nq)], queue_burst=[wl.queue_burst] * nq, queue_rate_inf=[wl.rate_inf] * nq)
            # a whole round, not the queue scheduler alone: a job placed by urgency preemption leaves its node oversubscribed until the round's own
            # oversubscribed evictor has run (pqs.go:160-200) — the accepted placements are running jobs of this round like any other
            r2 = s.schedule_round()
            for j, n in r2.scheduled.items():
                if rp[int(j)]:
                    final[int(j)] = int(n); replayed += 1
            for j in r2.preempted:
                if int(j) in final:
                    del final[int(j)]; dropped.append(int(j))     # an accepted placement the replay round preempted again: back to the queue
                else:
                    preempted.add(int(j))
            dropped = [j for j in dropped if j not in final]
        t3 = time.perf_counter()
        self.timing = dict(round_s=t1 - t0, exchange_s=t2 - t1, replay_s=t3 - t2)
        used = np.zeros((wl.num_nodes, wl.job_req.shape[1]), dtype=np.int64)
        if final:
            js = np.fromiter(final.keys(), dtype=np.int64, count=len(final)); ns = np.fromiter(final.values(), dtype=np.int64, count=len(final))
            np.add.at(used, ns, wl.job_req[js])
        return dict(scheduled=final, dropped=sorted(dropped), conflicts=int(summary["conflict_nodes"]), accepted=int(summary["accepted"]), replay_set=int(summary["replay"]),
                    replayed=replayed, local_new=sum(1 for j in res.scheduled if wl.job_node[int(j)] < 0), preempted=sorted(preempted), committed=used)

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
