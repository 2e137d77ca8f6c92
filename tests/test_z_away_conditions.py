"""Conditional away node types: AwayNodeType.NodeTypes + Conditions (internal/common/types/scheduling.go:18-51; nodedb.go:632-675).

* tests/golden/nodedb_conditional_away_cases.json — the 4 cases of TestConditionalAwayNodeScheduling (nodedb_test.go:1236-1291): one
  node, one job of `armada-preemptible-away-conditional`, through SelectNodeForJobWithTxn.
* tests/golden/matches_conditions_cases.json — the 13 cases of TestMatchesConditions (nodedb_test.go:1150-1234).  matchesCondition is
  internal to the NodeDb, so each case is driven end to end: a priority class whose only away entry has no WellKnownNodeTypeName and one
  NodeTypes entry ("gpu") carrying the case's conditions, a gpu-tainted node, a job requesting the case's resources — the job lands
  (as an away job) exactly when the conditions match, because otherwise the entry contributes no taints (nodedb.go:703-706).
Run on the oracle, on the CPU build of the device code and (-m gpu) on the HIP library; static matching is host-side mask
construction (asched_host.inc jobs_set / rebuildMasks), so the device code is the same away path the other away tests cover.
"""
import copy

import pytest

import scenario
from golden_io import ids, load

COND, MATCH = load("nodedb_conditional_away"), load("matches_conditions")
AWAY = 5  # ASCHED_METHOD_AWAY


def run_conditional_away(lib, case):
    c = scenario.Case(lib, case["SchedulingConfig"], case["Nodes"])
    c.set_jobs(case["Jobs"], {case["Jobs"][0]["queue"]: 0}, {})
    pod, pre = c.sched.select_node(0)
    if case["ExpectScheduled"]:   # :1280-1284
        assert pod.node == 0 and pod.method == AWAY and not pre, pod
    else:                          # :1285-1287
        assert pod.node == -1, pod


def run_matches_condition(lib, case):
    cfg = copy.deepcopy(COND[0]["SchedulingConfig"])
    conds = [[c["Resource"], c["Operator"], int(c["Value"])] for c in case.get("conditions") or []]
    cfg["priority_classes"]["cond-test"] = {"priority": 30000, "preemptible": True, "away": [[29000, "", [["gpu", conds]]]]}
    node = {"index": 1, "total": {"cpu": 64000, "memory": 1024 * 2**30, "nvidia.com/gpu": 8000}, "taints": [["gpu", "true", "NoSchedule"]],
            "labels": {"gpu": "true"}, "used": {}, "unschedulable": False}
    job = {"created": 1, "queue": "A", "pc": "cond-test", "priority": 1000, "gang": None, "tolerations": [], "selector": {}, "affinity": None,
           "req": {k: v for k, v in (case.get("jobResources") or {}).items() if k in scenario.RES}}
    c = scenario.Case(lib, cfg, [node])
    c.set_jobs([job], {"A": 0}, {})
    pod, _ = c.sched.select_node(0)
    if case["expectMatch"]:
        assert pod.node == 0 and pod.method == AWAY and pod.scheduled_at_priority == 29000, pod
    else:
        assert pod.node == -1, pod


@pytest.mark.parametrize("case", COND, ids=ids(COND))
def test_conditional_away_oracle(oracle_lib, case):
    run_conditional_away(oracle_lib, case)


@pytest.mark.parametrize("case", COND, ids=ids(COND))
def test_conditional_away_hostsim(hostsim_lib, case):
    run_conditional_away(hostsim_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", COND, ids=ids(COND))
def test_conditional_away_gpu(hip_lib, case):
    run_conditional_away(hip_lib, case)


@pytest.mark.parametrize("case", MATCH, ids=ids(MATCH))
def test_matches_condition_oracle(oracle_lib, case):
    run_matches_condition(oracle_lib, case)


@pytest.mark.parametrize("case", MATCH, ids=ids(MATCH))
def test_matches_condition_hostsim(hostsim_lib, case):
    run_matches_condition(hostsim_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MATCH, ids=ids(MATCH))
def test_matches_condition_gpu(hip_lib, case):
    run_matches_condition(hip_lib, case)


def _round_with_conditional_away(lib):
    """a whole round in which jobs of the conditional class reach tainted nodes only through the away entry, others stay home"""
    import numpy as np
    cfg = copy.deepcopy(COND[0]["SchedulingConfig"])
    GI = 2**30
    nodes = []
    for i in range(12):
        kind = i % 3
        n = {"index": i + 1, "total": {"cpu": 16000, "memory": 128 * GI}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}
        if kind == 1:
            n["total"]["nvidia.com/gpu"] = 8000; n["taints"] = [["gpu", "true", "NoSchedule"]]; n["labels"] = {"gpu": "true"}
        elif kind == 2:
            n["taints"] = [["largeJobsOnly", "true", "NoSchedule"]]; n["labels"] = {"largeJobsOnly": "true"}
        nodes.append(n)
    rng = np.random.default_rng(5)
    jobs = []
    for i in range(120):
        req = {"cpu": int(rng.integers(1, 9)) * 1000, "memory": int(rng.integers(1, 33)) * GI}
        if i % 5 == 0:
            req["nvidia.com/gpu"] = 1000     # fails the "gpu == 0" condition: may only go away onto the large nodes
        jobs.append({"created": i + 1, "queue": f"q{i % 3}", "pc": ["armada-preemptible-away-conditional", "priority-1", "armada-preemptible-away"][i % 3],
                     "priority": 1000, "gang": None, "tolerations": [], "selector": {}, "affinity": None, "req": req})
    c = scenario.Case(lib, cfg, nodes)
    queues = ["q0", "q1", "q2"]
    c.set_jobs(jobs, {q: i for i, q in enumerate(queues)}, {})
    queued = [c.sort_queued(jobs, [i for i, j in enumerate(jobs) if j["queue"] == q]) for q in queues]
    c.sched.round_prepare([1.0, 1.0, 1.0], queued)
    return c.sched.schedule_round(), jobs, nodes


def test_round_with_conditional_away_hostsim_equals_oracle(oracle_lib, hostsim_lib):
    a, jobs, nodes = _round_with_conditional_away(oracle_lib)
    b, _, _ = _round_with_conditional_away(hostsim_lib)
    scenario.assert_same_round(a, b)
    on_gpu_nodes = [j for j, n in a.scheduled.items() if nodes[n]["taints"] and nodes[n]["taints"][0][0] == "gpu"]
    assert on_gpu_nodes and all("nvidia.com/gpu" not in jobs[j]["req"] for j in on_gpu_nodes if jobs[j]["pc"] == "armada-preemptible-away-conditional")


@pytest.mark.gpu
def test_round_with_conditional_away_gpu_equals_oracle(oracle_lib, hip_lib):
    a, _, _ = _round_with_conditional_away(oracle_lib)
    b, _, _ = _round_with_conditional_away(hip_lib)
    scenario.assert_same_round(a, b)
