"""TestEvict_JobsEvictedInFairshareOrder (preempting_queue_scheduler_test.go:47-123): after the node evictor has taken every job of three queues with equal
allocations, the evicted jobs sit in the NodeDb's evicted table in DRF order — A, B, C, A, B, C, A, B, C (queue name breaks the cost ties) — not queue by queue.

The ABI has no getter for that table; its order is what fair-share preemption reads: it keeps from being rescheduled the evicted jobs that come LAST in it
(nodedb.go:935-1043).  So the same set-up is run as a whole round with a fourth queue D whose one job needs k cpu of the full 9-cpu node: D has the lowest cost,
goes first, and must take exactly the last k entries of the table — the LAST job of C, then the last job of B.  An evicted table
filled queue by queue (A A A B B B C C C) would lose C's jobs first instead.  Oracle, CPU build, HIP library (-m gpu)."""
import os
import sys

import numpy as np
import pytest

import scenario

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

GI = 2 ** 30


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


@pytest.mark.parametrize("k", [1, 2])   # (a larger job would exceed D's fair-share budget and order after the evicted jobs: queue_scheduler.go:738-798)
def test_evicted_jobs_are_ordered_by_fair_share_across_queues(lib, k):
    cfg = F.TestSchedulingConfig()                      # protectedFractionOfFairShare 0: every queue's jobs are evictable
    pc = F.PriorityClass1   # (the Go test hands NewNodeEvictor a filter that takes every job; the round's own evictor takes preemptible ones: a preemptible class here)
    node = {"index": 1, "total": {"cpu": 9000, "memory": 256 * GI}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}
    running_jobs = [j for q in "ABC" for j in F.N1Cpu4GiJobs(q, pc, 3)]        # :64-71, jobsPerQueue = 3
    newcomer = F.Test1Cpu4GiJob("D", pc)
    newcomer["req"] = dict(newcomer["req"]); newcomer["req"]["cpu"] = 1000 * k
    jobs = running_jobs + [newcomer]
    c = scenario.Case(lib, cfg, [node])
    prio = cfg["priority_classes"][pc]["priority"]
    running = {i: (0, prio, i + 1) for i in range(9)}                           # WithNewRun(..., job.PriorityClass().Priority), leased in creation order
    c.set_jobs(jobs, {"A": 0, "B": 1, "C": 2, "D": 3}, running)
    demand = np.zeros((4, scenario.R), dtype=np.int64)
    for i, j in enumerate(jobs):
        demand["ABCD".index(j["queue"])] += np.array(scenario.vec(j["req"]), dtype=np.int64)
    c.sched.round_prepare([1.0] * 4, [[], [], [], [9]], name_rank=[0, 1, 2, 3], demand=demand)
    res = c.sched.schedule_round()
    c.no_oversubscription()
    assert res.num_evicted_phase1 == 9                                          # :103 every job evicted
    assert res.scheduled == {9: 0}
    # evicted-table order A0 B0 C0 A1 B1 C1 A2 B2 C2 (:119-121); the last k entries are lost
    table = [q * 3 + r for r in range(3) for q in range(3)]
    assert sorted(res.preempted) == sorted(table[9 - k:]), (k, res.preempted)
