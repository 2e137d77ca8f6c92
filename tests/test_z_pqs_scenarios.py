"""Two stand-alone PreemptingQueueScheduler scenarios of the reference, run as whole rounds.

* TestPreemptingQueueScheduler_RespectNodePodLimits (preempting_queue_scheduler_test.go:3196-3397, 5 cases; parameters in
  tests/golden/pqs_pod_limits_cases.json, table evaluated mechanically): `pods` is one more resource column with resolution 1, indexed
  (configuration.ApplyRespectNodePodLimits :627-653), every job requests one pod (jobdb.SetRespectNodePodLimits).  The harness's fourth
  column plays that role.  Expectations: number of preempted incumbents and of newly scheduled challengers, preempted jobs are
  incumbents, and with a free extra node the challenger lands there.
* TestPreemptingQueueScheduler_NonPreemptibleOverPack (:3430-3503): five non-preemptible priority-2 incumbents saturate a 5-cpu node; a
  priority-3 challenger must neither preempt them nor be packed on top of them.
"""
import copy

import numpy as np
import pytest

import scenario
from golden_io import ids, load

POD = load("pqs_pod_limits")
PODS = "test-floating-resource"   # the column that carries `pods` here
GI = 2**30


def run_round(lib, cfg, nodes, incumbents, challengers):
    c = scenario.Case(lib, cfg, nodes)
    pcs = cfg["priority_classes"]
    jobs = incumbents + challengers
    running = {i: (0, pcs[j["pc"]]["priority"], i + 1) for i, j in enumerate(incumbents)}   # WithNewRun(node, ..., j.PriorityClass().Priority)
    c.set_jobs(jobs, {"A": 0}, running)
    demand = np.zeros((1, scenario.R), dtype=np.int64)
    for j in challengers:                                   # the tests seed demand with the challengers' requests (:3357-3360, :3478-3480)
        demand[0] += np.array(scenario.vec(j["req"]), dtype=np.int64)
    queued = [c.sort_queued(jobs, list(range(len(incumbents), len(jobs))))]
    c.sched.round_prepare([1.0], queued, name_rank=[0], demand=demand)
    res = c.sched.schedule_round()
    c.no_oversubscription()
    return res


def job(pc, created, pods, gang=None):
    req = {"cpu": 1000, "memory": 4 * GI}
    if pods:
        req[PODS] = 1
    return {"created": created, "queue": "A", "pc": pc, "priority": 1000, "gang": gang, "tolerations": [], "selector": {}, "affinity": None, "req": req}


def run_pod_case(lib, case):
    cfg = copy.deepcopy(case["SchedulingConfig"])
    cfg["indexed_resources"] = cfg["indexed_resources"] + [[PODS, 1]]            # ensurePodsResourceType(config.IndexedResources)
    caps = [int(case["nodePodCapacity"])] + [int(x) for x in case.get("extraNodePodCapacities") or []]
    nodes = [{"index": i + 1, "total": {"cpu": 10000, "memory": 64 * GI, PODS: cap}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}
             for i, cap in enumerate(caps)]                                         # buildNode :3278-3284
    n_inc, n_ch = int(case.get("incumbentCount") or 0), int(case["challengerCount"])
    incumbents = [job(case.get("incumbentPriorityClass") or "priority-0", i + 1, True) for i in range(n_inc)]
    gang = {"id": "gang-1", "cardinality": n_ch, "uniformity": ""} if case.get("challengerIsGang") else None
    challengers = [job("priority-3", n_inc + i + 1, True, gang) for i in range(n_ch)]
    res = run_round(lib, cfg, nodes, incumbents, challengers)
    assert len(res.preempted) == int(case["expectedPreemptions"]), (res.preempted, res.scheduled)
    assert len(res.scheduled) == int(case["expectedNewlyScheduled"]), (res.preempted, res.scheduled)
    assert all(j < n_inc for j in res.preempted) and all(j >= n_inc for j in res.scheduled)
    if len(caps) > 1:
        assert all(n != 0 for n in res.scheduled.values()), "the challenger should land on an extra node, not the saturated one"
    return res


def run_overpack(lib):
    cfg = copy.deepcopy(POD[0]["SchedulingConfig"])
    nodes = [{"index": 1, "total": {"cpu": 5000, "memory": 64 * GI}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}]
    incumbents = [job("priority-2-non-preemptible", i + 1, False) for i in range(5)]
    challengers = [job("priority-3", 6, False)]
    res = run_round(lib, cfg, nodes, incumbents, challengers)
    assert not res.preempted, "no incumbent should be preempted (they are non-preemptible)"
    assert not res.scheduled, "the challenger should not be placed on a node already saturated by non-preemptible incumbents"
    return res


@pytest.mark.parametrize("case", POD, ids=ids(POD))
def test_pod_limits_oracle(oracle_lib, case):
    run_pod_case(oracle_lib, case)


@pytest.mark.parametrize("case", POD, ids=ids(POD))
def test_pod_limits_hostsim(hostsim_lib, oracle_lib, case):
    scenario.assert_same_round(run_pod_case(oracle_lib, case), run_pod_case(hostsim_lib, case))


@pytest.mark.gpu
@pytest.mark.parametrize("case", POD, ids=ids(POD))
def test_pod_limits_gpu(hip_lib, oracle_lib, case):
    scenario.assert_same_round(run_pod_case(oracle_lib, case), run_pod_case(hip_lib, case))


def test_non_preemptible_over_pack_oracle(oracle_lib):
    run_overpack(oracle_lib)


def test_non_preemptible_over_pack_hostsim(hostsim_lib, oracle_lib):
    scenario.assert_same_round(run_overpack(oracle_lib), run_overpack(hostsim_lib))


@pytest.mark.gpu
def test_non_preemptible_over_pack_gpu(hip_lib, oracle_lib):
    scenario.assert_same_round(run_overpack(oracle_lib), run_overpack(hip_lib))


def _releases_pod_slot(lib):
    """nodedb_test.go:315-387 TestNodeBindingEvictionUnbinding_ReleasesPodSlot: bind consumes the node's only pod slot, evict + unbind
    frees it, a second job can then bind (the pods column is an ordinary resource column of the NodeDb arithmetic)."""
    cfg = copy.deepcopy(POD[0]["SchedulingConfig"])
    cfg["indexed_resources"] = cfg["indexed_resources"] + [[PODS, 1]]
    node = {"index": 1, "total": {"cpu": 10000, "memory": 64 * GI, PODS: 1}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}
    mk = lambda i: {"created": i, "queue": "queue", "pc": "priority-0", "priority": 1, "gang": None, "tolerations": [], "selector": {}, "affinity": None,  # noqa: E731
                    "req": {"cpu": 1000, "memory": GI, PODS: 1}}
    c = scenario.Case(lib, cfg, [node])
    c.set_jobs([mk(1), mk(2)], {"queue": 0}, {})
    s = c.sched
    prio = cfg["priority_classes"]["priority-0"]["priority"]
    lv, col = list(s.priorities).index(prio), scenario.RES.index(PODS)
    s.bind(0, 0, prio)
    assert s.get_alloc(0)[lv, col] == 0, "bind should consume the pod slot"                 # :365-366
    s.evict(0, 0)
    s.unbind(0, 0)
    assert s.get_alloc(0)[lv, col] == 1, "evict+unbind should free the pod slot"            # :372-373
    s.bind(1, 0, prio)                                                                      # :376-377 second job binds after the slot is freed
    assert s.get_alloc(0)[lv, col] == 0
    return s.get_alloc(0).tolist()


def test_releases_pod_slot(oracle_lib, hostsim_lib):
    assert _releases_pod_slot(oracle_lib) == _releases_pod_slot(hostsim_lib)


@pytest.mark.gpu
def test_releases_pod_slot_gpu(oracle_lib, hip_lib):
    assert _releases_pod_slot(oracle_lib) == _releases_pod_slot(hip_lib)
