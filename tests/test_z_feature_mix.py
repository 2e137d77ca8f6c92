"""Seeded rounds that mix everything the host-side tables encode — taints / tolerations, selectors, required node affinity, away node types
with and without conditions, a `pods`-like extra column, cordoned queues, per-queue per-class limits, gangs, running jobs — and compare
the whole round (assignments, preemptions, accounting, fair shares) between the oracle and the implementation under test.  The static
matching is host-side mask construction in the product and a per-node evaluation in the oracle: two independent implementations."""
import copy
import math

import numpy as np
import pytest

import scenario
from golden_io import load

CFG = load("nodedb_conditional_away")[0]["SchedulingConfig"]
GI = 2**30
EXTRA = "test-floating-resource"


def build(seed):
    rng = np.random.default_rng(7000 + seed)
    cfg = copy.deepcopy(CFG)
    cfg["protected_fraction_of_fair_share"] = float(rng.choice([0.0, 0.5, 1.0]))
    cfg["indexed_resources"] = cfg["indexed_resources"] + [[EXTRA, 1]]
    cfg["priority_classes"]["cond-away"] = {"priority": 30000, "preemptible": True,
                                            "away": [[29000, "", [["gpu", [["cpu", "<", 3]]], ["large", [["nvidia.com/gpu", "==", 0], ["cpu", ">", 100]]]]]]}
    if seed % 4 == 0:
        cfg["disable_gang_away"] = True
    nodes = []
    for i in range(int(rng.integers(8, 20))):
        kind = int(rng.integers(0, 4))
        n = {"index": i + 1, "total": {"cpu": 16000, "memory": 128 * GI, EXTRA: int(rng.integers(3, 12))}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}
        if kind == 1:
            n["total"]["nvidia.com/gpu"] = 8000; n["taints"] = [["gpu", "true", "NoSchedule"]]; n["labels"]["gpu"] = "true"
        elif kind == 2:
            n["taints"] = [["largeJobsOnly", "true", "NoSchedule"]]; n["labels"]["largeJobsOnly"] = "true"
        if rng.random() < 0.6:
            n["labels"]["zone"] = str(rng.choice(["a", "b", "c"]))
        nodes.append(n)
    queues = ["q0", "q1", "q2", "q3"]
    pcs = ["priority-0", "priority-1", "priority-2-non-preemptible", "armada-preemptible-away", "armada-preemptible-away-conditional", "cond-away"]
    jobs, running, free = [], {}, [16000] * len(nodes)
    slots = [n["total"][EXTRA] for n in nodes]
    created = 0

    def mk(q):
        nonlocal created
        created += 1
        req = {"cpu": int(rng.integers(1, 7)) * 1000, "memory": int(rng.integers(1, 17)) * GI, EXTRA: 1}
        tol, sel, aff = [], {}, None
        k = int(rng.integers(0, 8))
        if k == 1:
            req["nvidia.com/gpu"] = int(rng.integers(1, 5)) * 1000; tol = [{"key": "gpu", "op": "Equal", "value": "true", "effect": ""}]
        elif k == 2:
            tol = [{"key": "largeJobsOnly", "op": "Exists", "value": "", "effect": "NoSchedule"}]
        elif k == 3:
            sel = {"zone": str(rng.choice(["a", "b"]))}
        elif k == 4:
            aff = [[["zone", "NotIn", ["a"]]], [["gpu", "Exists", []], ["zone", "In", ["a", "c"]]]]
        elif k == 5:
            aff = [[["zone", "DoesNotExist", []]]]
        return {"created": created, "queue": q, "pc": str(rng.choice(pcs)), "priority": int(rng.integers(0, 3)), "gang": None, "tolerations": tol,
                "selector": sel, "affinity": aff, "req": req}
    for _ in range(int(rng.integers(10, 40))):           # running jobs, plain ones placed where they fit
        j = mk(str(rng.choice(queues)))
        j["tolerations"], j["selector"], j["affinity"] = [], {}, None
        j["req"].pop("nvidia.com/gpu", None)
        j["pc"] = str(rng.choice(pcs[:3]))
        n = int(rng.integers(0, len(nodes)))
        if nodes[n]["taints"] or free[n] < j["req"]["cpu"] or slots[n] < 1:
            continue
        free[n] -= j["req"]["cpu"]; slots[n] -= 1
        running[len(jobs)] = (n, cfg["priority_classes"][j["pc"]]["priority"], created)
        jobs.append(j)
    while len(jobs) < 150:
        q = str(rng.choice(queues))
        if rng.random() < 0.12:
            card = int(rng.integers(2, 5))
            proto = mk(q)
            for _ in range(card):
                created += 1
                g = copy.deepcopy(proto); g["created"] = created; g["gang"] = {"id": f"g{proto['created']}", "cardinality": card, "uniformity": ""}
                jobs.append(g)
        else:
            jobs.append(mk(q))
    cordoned = [0, 0, int(seed % 3 == 0), 0]
    frac = np.full((4, len(cfg["priority_classes"]), scenario.R), math.inf)
    pc_names = sorted(cfg["priority_classes"])
    frac[1, pc_names.index("priority-0"), scenario.RES.index("cpu")] = 0.1
    frac[3, :, scenario.RES.index("memory")] = 0.3
    return cfg, nodes, queues, jobs, running, cordoned, frac


def run(lib, seed):
    cfg, nodes, queues, jobs, running, cordoned, frac = build(seed)
    c = scenario.Case(lib, cfg, nodes)
    qidx = {q: i for i, q in enumerate(queues)}
    c.set_jobs(jobs, qidx, running)
    queued = [c.sort_queued(jobs, [i for i, j in enumerate(jobs) if i not in running and j["queue"] == q]) for q in queues]
    c.sched.round_prepare([1.0, 2.0, 0.5, 1.0], queued, name_rank=[2, 0, 3, 1], cordoned=cordoned, pc_resource_limit_fraction=frac,
                          global_tokens=60.0, global_burst=60, global_rate_inf=False)
    # (no "no bucket is negative" assertion here: the reference's own driver asserts it for its scenarios, pqs_test.go:2486-2492, but it is
    # not an invariant of a single round — jobs evicted from an oversubscribed node return after checking only their own priority level
    # (nodedb.go:897-906), so higher-priority returns can leave a lower level negative until the next round's eviction; seed 15 does)
    return c.sched.schedule_round()


@pytest.mark.parametrize("seed", range(24))
def test_feature_mix_hostsim_equals_oracle(hostsim_lib, oracle_lib, seed):
    a, b = run(oracle_lib, seed), run(hostsim_lib, seed)
    scenario.assert_same_round(a, b)


def test_feature_mix_is_not_trivial(oracle_lib):
    tot_s = tot_p = away = 0
    for seed in range(24):
        r = run(oracle_lib, seed)
        tot_s += len(r.scheduled); tot_p += len(r.preempted); away += sum(1 for m in r.scheduled_method.values() if m == 5)
    assert tot_s > 300 and tot_p > 10 and away > 5, (tot_s, tot_p, away)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_feature_mix_gpu_equals_oracle(hip_lib, oracle_lib, seed):
    scenario.assert_same_round(run(oracle_lib, seed), run(hip_lib, seed))
