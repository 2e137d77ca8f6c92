"""Fair-share preemption through the gang / queue scheduler marks its victims in the scheduling context — three stand-alone reference tests:

* TestGangScheduler_MarksPreemptedJobs (gang_scheduler_test.go:731-819) and TestQueueScheduler_PreemptedJobsGetMarkedInSctx
  (queue_scheduler_test.go:740-802): a node full of evicted priority-0 jobs of queue A; one priority-1 job of queue B must preempt exactly
  one of them (method ScheduledWithFairSharePreemption) and the victim is in sctx.PreemptedJobIds.
* TestGangScheduler_FairsharePreemption_PreemptsWholeGang (:825-936): the victim is a member of a two-node gang — the sibling on the other
  node is pulled in (preemptSiblingGangJobs), both are marked, no filler job is.

Restated through the ABI: the incumbents are bound, evicted and registered with the reference's evicted-table indexes
(asched_bind / asched_evict / asched_add_evicted), the incoming job goes through QueueScheduler.Schedule (asched_schedule_queues), whose
`preempted` list is sctx.PreemptedJobIds.  Run on the oracle, the CPU build of the device code and (-m gpu) the HIP library.
"""
import os
import sys

import numpy as np
import pytest

import scenario

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

METHOD_FAIRSHARE = 3


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


def _run(lib, nodes, placed, order, incoming):
    """placed: [(job, node)] incumbents; order: incumbents in evicted-table index order; incoming: the new job of queue B"""
    cfg = F.TestSchedulingConfig()
    jobs = [j for j, _ in placed] + [incoming]
    ident = {id(j): i for i, j in enumerate(jobs)}
    c = scenario.Case(lib, cfg, nodes)
    c.set_jobs(jobs, {"A": 0, "B": 1}, {})
    s = c.sched
    npc = len(c.pc_names)
    s.round_prepare([1.0, 1.0], [[], [ident[id(incoming)]]], name_rank=[0, 1], demand=np.zeros((2, scenario.R), dtype=np.int64),
                    allocated_by_pc=np.zeros((2, npc, scenario.R), dtype=np.int64), fairshare_preemption_tokens=100.0)
    node_of = {id(j): n for j, n in placed}
    for j, n in placed:                                   # CreateAndInsertWithJobDbJobsWithTxn: bound at the priority-class priority
        s.bind(ident[id(j)], n, cfg["priority_classes"][j["pc"]]["priority"])
    for idx, j in enumerate(order):                       # EvictJobsFromNode + AddEvictedJobSchedulingContextWithTxn(txn, idx, jctx)
        s.evict(ident[id(j)], node_of[id(j)])
        s.add_evicted(idx, ident[id(j)], node_of[id(j)])
    res = s.schedule_queues()
    return res, ident


def test_one_incumbent_is_preempted_and_marked(lib):
    node = F.Test32CpuNode(F.TestPriorities)
    incumbents = F.N1Cpu4GiJobs("A", F.PriorityClass0, 32)
    incoming = F.Test1Cpu4GiJob("B", F.PriorityClass1)
    res, ident = _run(lib, [node], [(j, 0) for j in incumbents], incumbents, incoming)
    new = ident[id(incoming)]
    assert res.scheduled == {new: 0} and res.scheduled_method[new] == METHOD_FAIRSHARE       # :793-797
    assert len(res.preempted) == 1 and next(iter(res.preempted)) in {ident[id(j)] for j in incumbents}   # "exactly one incumbent"


def test_preempting_a_gang_member_takes_the_whole_gang(lib):
    nodes = [F.Test32CpuNode(F.TestPriorities), F.Test32CpuNode(F.TestPriorities)]
    g1, g2 = F.WithGangJobDetails(F.N1Cpu4GiJobs("A", F.PriorityClass0, 2), "gang-1", 2, "")
    filler1, filler2 = F.N1Cpu4GiJobs("A", F.PriorityClass0, 31), F.N1Cpu4GiJobs("A", F.PriorityClass0, 31)
    node1_jobs, node2_jobs = [g1] + filler1, [g2] + filler2
    placed = [(j, 0) for j in node1_jobs] + [(j, 1) for j in node2_jobs]
    order = node2_jobs + filler1 + [g1]                   # :866-868: node2's jobs first, node1's filler, g1 last (highest index)
    incoming = F.Test1Cpu4GiJob("B", F.PriorityClass1)
    res, ident = _run(lib, nodes, placed, order, incoming)
    new = ident[id(incoming)]
    assert res.scheduled == {new: 0}, res.scheduled       # g1 has the highest index: it is preempted directly, on node 1
    assert set(res.preempted) == {ident[id(g1)], ident[id(g2)]}, "exactly the two gang members, no filler job"   # :905-922


@pytest.mark.parametrize("already_preempted", [False, True])
def test_preempted_job_is_not_rescheduled(lib, already_preempted):
    """nodedb_test.go:1103-1148 TestPreemptedJobIsNotRescheduled: SelectNodeForJobWithTxn refuses a job that carries PreemptionDetails (the failsafe of
    nodedb.go:538-544) although a node has room for it; a job that was never preempted schedules there.  The ABI has no setter for PreemptionDetails: the state
    is reached the way the scheduler reaches it — the job is the victim of a fair-share preemption applied by the gang scheduler (gang_scheduler.go:268-273) —
    and a second, empty node then has room for it."""
    nodes = [F.Test32CpuNode(F.TestPriorities), F.Test32CpuNode(F.TestPriorities)]
    incumbents = F.N1Cpu4GiJobs("A", F.PriorityClass0, 32)
    incoming = F.Test1Cpu4GiJob("B", F.PriorityClass1)
    fresh = F.Test1Cpu4GiJob("A", F.PriorityClass0)
    cfg = F.TestSchedulingConfig()
    # node 1 is held by one big non-preemptible-by-anyone job (non-preemptible priority class) while B is scheduled, so B has to preempt on node 0; it is unbound afterwards
    big = F.Test32Cpu256GiJob("A", F.PriorityClass3)
    jobs = incumbents + [incoming, fresh, big]
    ident = {id(j): i for i, j in enumerate(jobs)}
    c = scenario.Case(lib, cfg, nodes)
    c.set_jobs(jobs, {"A": 0, "B": 1}, {})
    s = c.sched
    npc = len(c.pc_names)
    s.round_prepare([1.0, 1.0], [[], [ident[id(incoming)]]], name_rank=[0, 1], demand=np.zeros((2, scenario.R), dtype=np.int64),
                    allocated_by_pc=np.zeros((2, npc, scenario.R), dtype=np.int64), fairshare_preemption_tokens=100.0)
    for j in incumbents:
        s.bind(ident[id(j)], 0, cfg["priority_classes"][j["pc"]]["priority"])
    s.bind(ident[id(big)], 1, cfg["priority_classes"][big["pc"]]["priority"])
    for idx, j in enumerate(incumbents):
        s.evict(ident[id(j)], 0)
        s.add_evicted(idx, ident[id(j)], 0)
    res = s.schedule_queues()
    assert res.scheduled == {ident[id(incoming)]: 0} and len(res.preempted) == 1
    victim = next(iter(res.preempted))
    s.unbind(ident[id(big)], 1)                                # node 1 is empty now
    job = victim if already_preempted else ident[id(fresh)]
    s.txn_begin()
    ok, pods, pre = s.schedule_many([job])
    s.txn_abort()
    assert len(pre) == 0                                       # :1135
    if already_preempted:
        assert not ok and pods[0].node < 0                     # :1143-1144 (no PodSchedulingContext is created)
    else:
        assert ok and pods[0].node == 1                        # :1138-1141
