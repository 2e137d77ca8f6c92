"""NodeDb-level operations restated from the reference's nodedb_test.go, through the C ABI (bind / evict / unbind / add_evicted /
schedule_many / get_alloc) on the oracle, on the device control code compiled for the CPU, and (-m gpu) on the HIP library.
The fixtures are the transcriptions in tests/golden/gofixtures.py; every test cites the lines it follows."""
import os
import sys

import numpy as np
import pytest

import scenario
from armada_amd.binding import SchedError

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

METHOD_FAIRSHARE = 3  # ASCHED_METHOD_FAIRSHARE


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


def _case(lib, nodes, jobs, running=None, **cfg_over):
    cfg = F.TestSchedulingConfig()
    cfg.update(cfg_over)
    c = scenario.Case(lib, cfg, nodes)
    queues = sorted({j["queue"] for j in jobs})
    c.set_jobs(jobs, {q: i for i, q in enumerate(queues)}, running or {})
    return c


def _level(c, prio):
    return list(c.sched.priorities).index(prio)


def test_node_binding_eviction_unbinding(lib):
    """nodedb_test.go:173-264 TestNodeBindingEvictionUnbinding: accounting after bind / evict / unbind round trips, and the calls that
    must fail (evict or unbind-then-bind of a job the node does not hold in that state)."""
    node = F.Test8GpuNode(F.TestPriorities)
    job = F.Test1GpuJob("A", F.PriorityClass0)
    c = _case(lib, [node], [job])
    s = c.sched
    req = np.array(scenario.vec(job["req"]), dtype=np.int64)
    prio = F.TEST_PRIORITY_CLASSES[F.PriorityClass0]["priority"]
    entry = s.get_alloc(0).copy()
    with pytest.raises(SchedError):          # :202-203 evicting a job that is not bound to the node
        s.evict(0, 0)
    s.bind(0, 0, prio)                       # :186
    bound = s.get_alloc(0).copy()
    lv = _level(c, prio)
    assert (bound[lv] == entry[lv] - req).all()          # :251-254 allocatable at the job's priority = total - request
    for l, p in enumerate(s.priorities):                 # preemptible job at priority 0: only the levels <= 0 are debited
        assert (bound[l] == (entry[l] - req if p <= prio else entry[l])).all()
    with pytest.raises(SchedError):          # :208-209 binding a job that is already bound
        s.bind(0, 0, prio)
    s.unbind(0, 0)                           # :189
    assert (s.get_alloc(0) == entry).all()   # :214 assertNodeAccountingEqual(entry, unboundNode)
    s.bind(0, 0, prio)
    s.evict(0, 0)                            # :195
    evicted = s.get_alloc(0).copy()
    ev = _level(c, -2)                       # an evicted job still holds its resources at the evicted priority only (node.go:449-474)
    for l, p in enumerate(s.priorities):
        assert (evicted[l] == (entry[l] - req if l == ev else entry[l])).all()
    with pytest.raises(SchedError):          # :211-212 evicting an evicted job
        s.evict(0, 0)
    s.bind(0, 0, prio)                       # :200 re-binding the evicted job
    assert (s.get_alloc(0) == bound).all()   # :217 assertNodeAccountingEqual(boundNode, evictedBoundNode)
    s.evict(0, 0)
    s.unbind(0, 0)                           # :198 unbinding an evicted job
    assert (s.get_alloc(0) == entry).all()   # :215-216


def test_bind_unbind_non_preemptible_releases_every_bucket(lib):
    """nodedb_test.go:1715-1745: a non-preemptible bind debits every priority bucket, the unbind restores each exactly."""
    node = F.Test8GpuNode(F.TestPriorities)
    job = F.Test1Cpu4GiJob("queue-a", F.PriorityClass2NonPreemptible)
    c = _case(lib, [node], [job])
    s = c.sched
    prio = F.TEST_PRIORITY_CLASSES[F.PriorityClass2NonPreemptible]["priority"]
    before = s.get_alloc(0).copy()
    s.bind(0, 0, prio)
    after = s.get_alloc(0)
    for l in range(len(s.priorities)):
        assert not (after[l] == before[l]).all(), f"bucket {s.priorities[l]} was not debited"
    s.unbind(0, 0)
    assert (s.get_alloc(0) == before).all()


@pytest.mark.parametrize("evicted_pc,new_pc,expect", [
    (F.PriorityClass2, F.PriorityClass1, False),   # cannot fair-share preempt higher-priority jobs
    (F.PriorityClass1, F.PriorityClass1, True),    # can fair-share preempt equal-priority jobs
    (F.PriorityClass0, F.PriorityClass1, True),    # can fair-share preempt lower-priority jobs
])
def test_fair_share_preemption_respects_priority_order(lib, evicted_pc, new_pc, expect):
    """nodedb_test.go:1025-1101: a full node whose 32 jobs are all evicted; with urgency preemption off, an incoming job is
    scheduled by fair-share preemption iff the evicted jobs' priority does not exceed its own."""
    node = F.Test32CpuNode(F.TestPriorities)
    evicted = F.N1Cpu4GiJobs("A", evicted_pc, 32)
    incoming = F.Test1Cpu4GiJob("B", new_pc)
    jobs = evicted + [incoming]
    eprio = F.TEST_PRIORITY_CLASSES[evicted_pc]["priority"]
    c = _case(lib, [node], jobs, disable_urgency=True)
    s = c.sched
    for i in range(32):                                  # CreateAndInsertWithJobDbJobsWithTxn(txn, evictedJob, node) :1062
        s.bind(i, 0, eprio)
    for i in range(32):                                  # EvictJobsFromNode + AddEvictedJobSchedulingContextWithTxn(txn, i, jctx) :1068-1078
        s.evict(i, 0)
        s.add_evicted(i, i, 0)
    s.txn_begin()
    ok, pods, _ = s.schedule_many([32])
    s.txn_abort()
    assert ok == expect
    if expect:
        assert pods[0].node == 0 and pods[0].method == METHOD_FAIRSHARE   # :1092-1095
    else:
        assert pods[0].node < 0                                            # :1097-1098
