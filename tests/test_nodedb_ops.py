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


METHOD_URGENCY = 4  # ASCHED_METHOD_URGENCY


@pytest.mark.parametrize("register_evicted,disable_fair,disable_urgency,expect,method", [
    (False, False, False, True, METHOD_URGENCY),     # urgency-based preemption by default
    (False, False, True, False, None),               # no urgency-based preemption when disabled
    (True, False, False, True, METHOD_FAIRSHARE),    # fair-share preemption by default
    (True, True, False, True, METHOD_URGENCY),       # falls through to urgency-based preemption when fair-share disabled
    (True, False, True, True, METHOD_FAIRSHARE),     # fair-share preemption when urgency disabled
    (True, True, True, False, None),                 # no preemption when both strategies disabled
])
def test_preemption_scheduling(lib, register_evicted, disable_fair, disable_urgency, expect, method):
    """nodedb_test.go:930-1023 TestPreemptionScheduling: a node filled with 32 priority-0 jobs and one incoming priority-1 job;
    which preemption strategy places it depends on whether the incumbents are registered as evicted and on the two switches."""
    node = F.Test32CpuNode(F.TestPriorities)
    bound = F.N1Cpu4GiJobs("A", F.PriorityClass0, 32)
    incoming = F.N1Cpu4GiJobs("B", F.PriorityClass1, 1)[0]
    c = _case(lib, [node], bound + [incoming], disable_fairshare=disable_fair, disable_urgency=disable_urgency)
    s = c.sched
    for i in range(32):                                  # :984-986
        s.bind(i, 0, 0)
    if register_evicted:                                 # :988-1001
        for i in range(32):
            s.evict(i, 0)
            s.add_evicted(i, i, 0)
    s.txn_begin()
    ok, pods, _ = s.schedule_many([32])
    s.txn_abort()
    assert ok == expect                                  # :1011
    if expect:
        assert pods[0].node == 0 and pods[0].method == method   # :1013-1016
    else:
        assert pods[0].node < 0


def test_eviction_bucket_values(lib):
    """nodedb_test.go:438-498 TestEviction: explicit AllocatableByPriority of a 32-cpu node holding a preemptible priority-0 job and a
    non-preemptible priority-3 job (debited at every level), before and after both are evicted."""
    node = F.Test32CpuNode(F.TestPriorities)
    jobs = [F.Test1Cpu4GiJob("queue-alice", F.PriorityClass0), F.Test1Cpu4GiJob("queue-alice", F.PriorityClass3)]
    c = _case(lib, [node], jobs)
    s = c.sched
    s.bind(0, 0, 0)
    s.bind(1, 0, 3)

    def cpu_mem(cpu, mem):
        return np.array(scenario.vec(F.rl({"cpu": cpu, "memory": mem})), dtype=np.int64)

    want = {-2: cpu_mem("30", "248Gi"), -1: cpu_mem("30", "248Gi"), 0: cpu_mem("30", "248Gi")}   # :463-473
    for p in (1, 2, 3, 28000, 29000, 30000):
        want[p] = cpu_mem("31", "252Gi")
    got = s.get_alloc(0)
    for l, p in enumerate(s.priorities):
        assert (got[l] == want[p]).all(), f"bound: priority {p}: {got[l]} != {want[p]}"
    s.evict(0, 0)
    s.evict(1, 0)
    want = {p: cpu_mem("32", "256Gi") for p in s.priorities}                                      # :488-498
    want[-2] = cpu_mem("30", "248Gi")
    got = s.get_alloc(0)
    for l, p in enumerate(s.priorities):
        assert (got[l] == want[p]).all(), f"evicted: priority {p}: {got[l]} != {want[p]}"


@pytest.mark.parametrize("disallowed,evicted,expect", [
    ([], False, True),          # scheduled - only using allowed resources
    (["cpu"], False, False),    # not scheduled - using disallowed resources
    (["cpu"], True, True),      # scheduled - reschedules even if using disallowed resources
])
def test_disallowed_job_resources(lib, disallowed, evicted, expect):
    """nodedb_test.go:745-798 TestDisallowedJobResources through SelectNodeForJobWithTxn: a job requesting a disallowed resource is
    refused unless it is an evicted job returning to its node."""
    node = F.Test32CpuNode(F.TestPriorities)
    job = F.Test1Cpu4GiJob("A", F.PriorityClass0)
    c = _case(lib, [node], [job], disallowed_resources=disallowed)
    s = c.sched
    s.txn_begin()
    pod, _ = s.select_node(0, 0 if evicted else -1)
    s.txn_abort()
    assert (pod.node == 0) == expect


def _gang_sibling_cases():
    g1 = F.WithGangJobDetails(F.N16Cpu128GiJobs("A", F.PriorityClass0, 3), "gang-1", 3, "")
    g2 = F.WithGangJobDetails(F.N32Cpu256GiJobs("A", F.PriorityClass0, 3), "gang-1", 2, "")
    filler = lambda: F.Test16Cpu128GiJob("A", F.PriorityClass0)  # noqa: E731
    return {
        # name: (per node: its jobs in eviction order, new jobs, preempted, preempted via sibling)     nodedb_test.go:1836-1899
        "preempt sibling on another node": ([[g2[1]], [g2[0]]], [F.Test32Cpu256GiJob("B", F.PriorityClass0)], [g2[0]], [g2[1]]),
        "preempt sibling on another node - 2 jobs on same node": ([[filler(), g1[1]], [g1[2], g1[0]]], [F.Test32Cpu256GiJob("B", F.PriorityClass0)],
                                                                  [g1[0], g1[2]], [g1[1]]),
        "preempted sibling - makes space for next scheduled job": ([[g1[1], filler()], [g1[2], g1[0]]],
                                                                   [F.Test32Cpu256GiJob("B", F.PriorityClass0), F.Test16Cpu128GiJob("B", F.PriorityClass0)],
                                                                   [g1[0], g1[2]], [g1[1]]),
    }


@pytest.mark.parametrize("name", list(_gang_sibling_cases()))
def test_fairshare_preemption_preempts_gang_siblings(lib, name):
    """nodedb_test.go:1811-1951 TestFairsharePreemption_PreemptGangSiblings: fair-share preemption of one gang member takes its evicted
    siblings on other nodes with it (preemptSiblingGangJobs, nodedb.go:468-525); every preempted job leaves the evicted table."""
    setup, new_jobs, exp_pre, exp_sib = _gang_sibling_cases()[name]
    nodes = [F.Test32CpuNode(F.TestPriorities) for _ in setup]
    jobs = [j for node_jobs in setup for j in node_jobs] + new_jobs
    ident = {id(j): i for i, j in enumerate(jobs)}
    c = _case(lib, nodes, jobs, disable_urgency=True)                 # :1907
    s = c.sched
    ev = 0
    for n, node_jobs in enumerate(setup):                             # :1909-1916
        for j in node_jobs:
            s.bind(ident[id(j)], n, 0)
        for j in node_jobs:
            s.evict(ident[id(j)], n)
            s.add_evicted(ev, ident[id(j)], n)
            ev += 1
    preempted = []
    for j in new_jobs:                                                # :1921-1937
        s.txn_begin()
        ok, pods, pre = s.schedule_many([ident[id(j)]])
        assert ok and pods[0].node >= 0
        preempted += pre
        s.txn_commit()
    assert sorted(preempted) == sorted(ident[id(j)] for j in exp_pre + exp_sib)   # :1938-1944 (the ABI does not tell the two causes apart)


_TOL_FOO = [{"key": "foo", "op": "Equal", "value": "foo", "effect": ""}]
_TAINT_FOO = [["foo", "foo", "NoSchedule"]]


@pytest.mark.parametrize("name,taints,labels,tolerations,selector,expect", [
    # nodematching_test.go:80-270 TestNodeSchedulingRequirementsMet, the cases without node affinity (which the ABI does not model)
    ("tolerated taints", _TAINT_FOO, {}, _TOL_FOO, {}, True),
    ("untolerated taints", _TAINT_FOO, {}, [], {}, False),
    ("matched node selector", [], {"bar": "bar"}, [], {"bar": "bar"}, True),
    ("unmatched node selector", [], {}, [], {"bar": "bar"}, False),
    ("tolerated taints and matched node selector", _TAINT_FOO, {"bar": "bar"}, _TOL_FOO, {"bar": "bar"}, True),
    ("untolerated taints and matched node selector", _TAINT_FOO, {"bar": "bar"}, [], {"bar": "bar"}, False),
    ("tolerated taints and unmatched node selector", _TAINT_FOO, {}, _TOL_FOO, {"bar": "bar"}, False),
])
def test_static_node_requirements(lib, name, taints, labels, tolerations, selector, expect):
    """StaticJobRequirementsMet (nodematching.go:161-190) through node selection on a one-node NodeDb: taints vs tolerations
    (empty effect and operator as in the reference's table), node selector vs labels."""
    node = F.Test32CpuNode(F.TestPriorities)
    node["taints"] = [list(t) for t in taints]
    node["labels"] = dict(labels)
    job = F.Test1Cpu4GiJob("A", F.PriorityClass0)
    job["tolerations"] = [dict(t) for t in tolerations]
    job["selector"] = dict(selector)
    c = _case(lib, [node], [job])
    s = c.sched
    s.txn_begin()
    pod, _ = s.select_node(0)
    s.txn_abort()
    assert (pod.node == 0) == expect


def test_remove_job_of_unbound_job_is_a_noop(lib):
    """internaltypes/node_test.go:357-364 TestNode_RemoveJob_AlreadyUnboundIsNoop, and :366-380: the cutoff stored at add decides which
    buckets a removal credits."""
    node = F.Test32CpuNode(F.TestPriorities)
    job = F.Test1Cpu4GiJob("queue-a", F.PriorityClass1)
    c = _case(lib, [node], [job])
    s = c.sched
    before = s.get_alloc(0).copy()
    s.unbind(0, 0)                                        # not bound: no error, nothing changes
    assert (s.get_alloc(0) == before).all()
    s.bind(0, 0, 1)
    after = s.get_alloc(0)
    req = np.array(scenario.vec(job["req"]), dtype=np.int64)
    for l, p in enumerate(s.priorities):                  # preemptible at priority 1: buckets <= 1 debited, the others untouched
        assert (after[l] == (before[l] - req if p <= 1 else before[l])).all()
    s.unbind(0, 0)
    assert (s.get_alloc(0) == before).all()


def test_nodedb_includes_cross_pool_priority(lib):
    """nodedb_test.go:33-46 TestNodeDbIncludesCrossPoolPriority: the priority axis starts EvictedPriority (-2), CrossPoolPriority (-1), then the real
    priorities in ascending order, and there is a plane for the cross-pool bucket"""
    c = _case(lib, F.N32CpuNodes(1, F.TestPriorities), [F.Test1Cpu4GiJob("A", F.PriorityClass0)])
    prios = list(c.sched.priorities)
    assert prios[0] == -2 and prios[1] == -1 and prios == sorted(prios) and len(set(prios)) == len(prios)
    assert c.sched.get_alloc(0).shape[0] == len(prios)


def test_select_node_for_pod_node_id_label_success(lib):
    """nodedb_test.go:96-119 TestSelectNodeForPod_NodeIdLabel_Success: a job whose context is pinned to a node (jctx.SetAssignedNode) is given exactly that
    node, although the first node in the order would fit as well"""
    nodes = F.N32CpuNodes(2, F.TestPriorities)
    c = _case(lib, nodes, F.N1Cpu4GiJobs("A", F.PriorityClass0, 1))
    res, preempted = c.sched.select_node(0, pinned_node=1)
    assert res.node == 1 and preempted == []
