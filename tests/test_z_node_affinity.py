"""Required node affinity (PodRequirements.GetAffinityNodeSelector; NodeAffinityRequirementsMet, nodematching.go:242-255).

The matching itself lives in k8s.io/component-helpers v0.32.11 (scheduling/corev1/nodeaffinity, not under /root/reference): terms are
ORed, the expressions of a term ANDed, a term without expressions matches nothing, In needs the label, NotIn / DoesNotExist accept a
missing label.  Pinned by the reference's own cases that use it — "nodeAffinity node notIn" (preempting_queue_scheduler_test.go:1773,
queue_scheduler_test.go:442) and "node affinity" (nodedb_test.go:598), which run with the other goldens — and by the rule table below,
one node-selection per rule, on the oracle, the CPU build of the device code and (-m gpu) the HIP library.  Affinity is part of the
requirement class, i.e. of the host-built static masks; the device code is untouched by it.
"""
import copy

import numpy as np
import pytest

import scenario
from golden_io import load

CFG = load("nodedb_conditional_away")[0]["SchedulingConfig"]
GI = 2**30


def node(i, labels):
    return {"index": i + 1, "total": {"cpu": 8000, "memory": 64 * GI}, "taints": [], "labels": dict(labels), "used": {}, "unschedulable": False}


def job(affinity, i=0, queue="A"):
    return {"created": i + 1, "queue": queue, "pc": "priority-0", "priority": 1000, "gang": None, "tolerations": [], "selector": {}, "affinity": affinity,
            "req": {"cpu": 1000, "memory": GI}}


NODES = [node(0, {}), node(1, {"zone": "a", "gen": "7"}), node(2, {"zone": "b", "disk": "ssd", "gen": "12"}), node(3, {"zone": "c", "disk": "hdd", "gen": "new"})]
# (name, affinity terms, nodes that satisfy it)
RULES = [
    ("nil selector matches every node", None, {0, 1, 2, 3}),
    ("no terms matches nothing", [], set()),
    ("empty term matches nothing", [[]], set()),
    ("In needs the label", [[["zone", "In", ["a", "b"]]]], {1, 2}),
    ("NotIn accepts a missing label", [[["zone", "NotIn", ["a", "b"]]]], {0, 3}),
    ("Exists", [[["disk", "Exists", []]]], {2, 3}),
    ("DoesNotExist", [[["disk", "DoesNotExist", []]]], {0, 1}),
    ("expressions of a term are ANDed", [[["zone", "In", ["b", "c"]], ["disk", "NotIn", ["hdd"]]]], {2}),
    ("terms are ORed", [[["zone", "In", ["a"]]], [["disk", "In", ["hdd"]]]], {1, 3}),
    ("an empty term next to a matching one", [[], [["zone", "In", ["c"]]]], {3}),
    ("value never seen on a node", [[["zone", "In", ["nowhere"]]]], set()),
    # Gt / Lt (k8s.io/apimachinery pkg/labels Requirement.Matches): integers on both sides, exactly one requirement value
    ("Gt compares integers, not strings", [[["gen", "Gt", ["9"]]]], {2}),                  # "12" > "9" as numbers ("12" < "9" as strings)
    ("Lt", [[["gen", "Lt", ["9"]]]], {1}),
    ("Gt is strict", [[["gen", "Gt", ["12"]]]], set()),
    ("a label that does not parse matches neither", [[["gen", "Gt", ["-100"]]]], {1, 2}),  # node 3 has gen=new; node 0 has no gen
    ("a requirement value that does not parse matches nothing", [[["gen", "Lt", ["many"]]]], set()),
    ("two requirement values match nothing", [[["gen", "Gt", ["1", "2"]]]], set()),
    ("Gt next to In", [[["gen", "Gt", ["5"]], ["zone", "In", ["a", "b"]]]], {1, 2}),
]


def eligible_nodes(lib, affinity):
    """which nodes the job may use: one single-node NodeDb per node (first fit returns one node; this asks about each)"""
    out = set()
    for i, n in enumerate(NODES):
        c = scenario.Case(lib, CFG, [n])
        c.set_jobs([job(affinity)], {"A": 0}, {})
        pod, _ = c.sched.select_node(0)
        if pod.node == 0:
            out.add(i)
    return out


@pytest.mark.parametrize("rule", RULES, ids=[r[0] for r in RULES])
def test_rules_oracle(oracle_lib, rule):
    assert eligible_nodes(oracle_lib, rule[1]) == rule[2]


@pytest.mark.parametrize("rule", RULES, ids=[r[0] for r in RULES])
def test_rules_hostsim(hostsim_lib, rule):
    assert eligible_nodes(hostsim_lib, rule[1]) == rule[2]


@pytest.mark.gpu
@pytest.mark.parametrize("rule", RULES, ids=[r[0] for r in RULES])
def test_rules_gpu(hip_lib, rule):
    assert eligible_nodes(hip_lib, rule[1]) == rule[2]


def _round(lib, seed):
    rng = np.random.default_rng(seed)
    zones, disks = ["a", "b", "c"], ["ssd", "hdd"]
    nodes = []
    for i in range(24):
        labels = {}
        if rng.random() < 0.8:
            labels["zone"] = str(rng.choice(zones))
        if rng.random() < 0.5:
            labels["disk"] = str(rng.choice(disks))
        if rng.random() < 0.7:
            labels["gen"] = str(int(rng.integers(1, 20)))
        nodes.append(node(i, labels))
    jobs = []
    for i in range(200):
        k = int(rng.integers(0, 8))
        aff = [None, [[["zone", "In", [str(rng.choice(zones))]]]], [[["zone", "NotIn", ["a"]], ["disk", "Exists", []]]],
               [[["disk", "DoesNotExist", []]], [["zone", "In", ["c"]]]], [[]], [[["disk", "In", ["ssd"]]]],
               [[["gen", "Gt", ["10"]]]], [[["gen", "Lt", ["6"]], ["zone", "NotIn", ["b"]]]]][k]
        j = job(copy.deepcopy(aff), i, queue=f"q{i % 2}")
        j["req"] = {"cpu": int(rng.integers(1, 4)) * 1000, "memory": int(rng.integers(1, 9)) * GI}
        jobs.append(j)
    c = scenario.Case(lib, CFG, nodes)
    c.set_jobs(jobs, {"q0": 0, "q1": 1}, {})
    queued = [c.sort_queued(jobs, [i for i, j in enumerate(jobs) if j["queue"] == q]) for q in ("q0", "q1")]
    c.sched.round_prepare([1.0, 2.0], queued)
    return c.sched.schedule_round(), jobs, nodes


@pytest.mark.parametrize("seed", range(3))
def test_round_hostsim_equals_oracle(oracle_lib, hostsim_lib, seed):
    a, jobs, nodes = _round(oracle_lib, seed)
    b, _, _ = _round(hostsim_lib, seed)
    scenario.assert_same_round(a, b)
    assert len(a.scheduled) > 20
    for j, n in a.scheduled.items():   # every placement honours the job's affinity (independent restatement of the rule)
        aff, lab = jobs[j]["affinity"], nodes[n]["labels"]
        if aff is None:
            continue
        def holds(e):
            k, op, vals = e
            return {"In": k in lab and lab[k] in vals, "NotIn": not (k in lab and lab[k] in vals), "Exists": k in lab, "DoesNotExist": k not in lab,
                    "Gt": k in lab and lab[k].isdigit() and int(lab[k]) > int(vals[0]), "Lt": k in lab and lab[k].isdigit() and int(lab[k]) < int(vals[0])}[op]
        assert any(t and all(holds(e) for e in t) for t in aff), (jobs[j], nodes[n])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_round_gpu_equals_oracle(oracle_lib, hip_lib, seed):
    a, _, _ = _round(oracle_lib, seed)
    b, _, _ = _round(hip_lib, seed)
    scenario.assert_same_round(a, b)
