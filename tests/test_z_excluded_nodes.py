"""PodSchedulingContext.NumExcludedNodesByReason (scheduling/context/pod.go:51; nodedb.go:538-630, 724-789, 881-928, 1118-1133) through asched_excluded_nodes,
for selections that end WITHOUT a node (include/armada_sched.h asched_excluded_reason).

What pins it to the reference:
  * its own assertions — the counts of an unsuccessful job add up to NumNodes (queue_scheduler_test.go:676-690: checked inside scenario.run_qs_case for all of the
    reference's QueueScheduler cases, and here on random rounds), a disallowed resource excludes every node under one reason (nodedb_test.go:780-796), a node-id
    selection that fails leaves exactly one reason (nodedb_test.go:122-146), the FIRST resource in factory order names the reason
    (nodematching_test.go:655-712 TestResourceRequirementsMet_RespectNodePodLimits);
  * the reason each failing case of the reference's requirement tables must give (TestNodeTypeSchedulingRequirementsMet nodematching_test.go:411-562,
    TestNodeSchedulingRequirementsMet :19-410 — the reference only asserts `reason != nil` there, "TODO: Test for specific reason"; the expected kinds below follow
    nodematching.go:127-267 line by line);
  * two independent implementations agreeing on random rounds: the oracle repeats the reference's iterator walk on failure; the product keeps a bit per node the
    iterator yielded plus the dynamic reasons (one wide pass on the device) and derives the static reasons on the host when asked.
"""
import copy
import os
import sys

import numpy as np
import pytest

import scenario
from golden_io import load

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

from armada_amd import workloads as W  # noqa: E402
from armada_amd.binding import SchedError  # noqa: E402

ERR_UNSUPPORTED = -3


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


def names(c, hist):
    """histogram with the interned ids turned back into the case's strings"""
    inv = {v: k for k, v in c.S.d.items()}
    out = []
    for kind, a, b, cc, req, avail, count in hist:
        if kind == "untolerated_taint":
            out.append((kind, inv[a], inv[b], cc, count))
        elif kind == "missing_label":
            out.append((kind, inv[a], count))
        elif kind == "unmatched_label":
            out.append((kind, inv[a], inv[b], inv[cc], count))
        elif kind == "insufficient_resources":
            out.append((kind, scenario.RES[a], req, avail, count))
        else:
            out.append((kind, count))
    return out


FOO_TAINT = [["foo", "foo", "NoSchedule"]]
FOO_TOL = [{"key": "foo", "op": "Equal", "value": "foo", "effect": ""}]
BAR_SEL = {"bar": "bar"}
NO_SCHEDULE = 1
# the failing cases of TestNodeTypeSchedulingRequirementsMet (nodematching_test.go:411-562) and the reason NodeTypeJobRequirementsMet gives for each (:127-139:
# tolerations first, then the selector; :216-243 an indexed label the type does not carry is "not set"): name, taints, labels, indexed labels, tolerations, selector
TYPE_CASES = [
    ("untolerated taints", FOO_TAINT, {}, None, [], {}, [("untolerated_taint", "foo", "foo", NO_SCHEDULE, 1)]),                               # :446-451
    ("unset indexed label", [], {}, ["bar"], FOO_TOL, BAR_SEL, [("missing_label", "bar", 1)]),                                              # :468-477
    ("different label value", [], {"bar": "baz"}, ["bar"], [], BAR_SEL, [("unmatched_label", "bar", "bar", "baz", 1)]),                     # :478-486
    ("untolerated taints and matched node selector", FOO_TAINT, {"bar": "bar"}, ["bar"], [], BAR_SEL, [("untolerated_taint", "foo", "foo", NO_SCHEDULE, 1)]),   # :505-513
    ("tolerated taints and different label value", FOO_TAINT, {"bar": "baz"}, ["bar"], FOO_TOL, BAR_SEL, [("unmatched_label", "bar", "bar", "baz", 1)]),        # :514-523
    ("tolerated taints and missing label", FOO_TAINT, {}, ["bar"], FOO_TOL, BAR_SEL, [("missing_label", "bar", 1)]),                        # :524-533
    # not in the reference's table: the label is not indexed, so the TYPE matches ("missing label" :487-494 expects success) and the NODE-level selector check names it (:161-175)
    ("missing label at node level", [], {}, None, [], BAR_SEL, [("missing_label", "bar", 1)]),
]


@pytest.mark.parametrize("name,taints,labels,ilabels,tolerations,selector,expect", TYPE_CASES, ids=[c[0] for c in TYPE_CASES])
def test_static_reasons_of_the_reference_requirement_cases(lib, name, taints, labels, ilabels, tolerations, selector, expect):
    cfg = F.TestSchedulingConfig()
    cfg["indexed_taints"] = sorted({t[0] for t in taints})
    cfg["indexed_node_labels"] = list(ilabels or [])
    node = F.Test32CpuNode(F.TestPriorities)
    node["taints"], node["labels"] = copy.deepcopy(taints), dict(labels)
    job = F.Test1Cpu4GiJob("A", F.PriorityClass0)
    job["tolerations"], job["selector"] = copy.deepcopy(tolerations), dict(selector)
    c = scenario.Case(lib, cfg, [node])
    c.set_jobs([job], {"A": 0}, {})
    pod, _ = c.sched.select_node(0)
    assert pod.node < 0
    assert names(c, c.sched.excluded_nodes(0)) == expect, name


NODE_CASES = load("node_requirements_met")
# TestNodeSchedulingRequirementsMet's failing cases (nodematching_test.go:19-410) through SelectNodeForJobWithTxn on a one-node NodeDb: what the ONE node is excluded for.
# Taints before selectors before affinity before resources (StaticJobRequirementsMet :161-190); a node whose indexed resources do not cover the request is never
# yielded by the iterator and is counted as "insufficient resources available" (nodedb.go:566-580), whatever JobRequirementsMet would have said about it.
NODE_EXPECT = {
    # required node affinity `bar In [bar]` on a node without labels: UnmatchedNodeSelector (:242-255); the tolerations have nothing to tolerate
    "nil taints and labels": "unmatched_affinity", "no taints or labels": "unmatched_affinity", "unmatched node affinity": "unmatched_affinity",
    "tolerated taints and unmatched node affinity": "unmatched_affinity",
    # taint foo=foo:NoSchedule without a toleration: UntoleratedTaint, before anything else is looked at (:162-165)
    "untolerated taints": "untolerated_taint", "untolerated taints and matched node affinity": "untolerated_taint", "untolerated taints and matched node selector": "untolerated_taint",
    # node selector bar=bar on a node that does not carry the label at all: MissingLabel (:230-240), not UnmatchedLabel
    "unmatched node selector": "missing_label", "tolerated taints and unmatched node selector": "missing_label",
    # the one node's cpu does not cover the request at any priority the selection tries: the iterator yields nothing (nodeiteration.go:318-382)
    "insufficient cpu": "implicit", "insufficient cpu at priority": "implicit",
}


@pytest.mark.parametrize("case", [c for c in NODE_CASES if not c["expectSuccess"]], ids=[c["name"] for c in NODE_CASES if not c["expectSuccess"]])
def test_reason_of_the_reference_node_requirement_cases(lib, case):
    n = case["node"]
    node = {"index": 1, "total": n["total"], "taints": n["taints"], "labels": n["labels"], "used": {}, "unschedulable": False}
    c = scenario.Case(lib, case["SchedulingConfig"], [node])
    s = c.sched
    if n.get("alloc_by_priority") is not None:
        total = np.array([scenario.vec(n["total"])], dtype=np.int64)
        abp = np.zeros((1, s.P, scenario.R), dtype=np.int64)
        for l, prio in enumerate(s.priorities):
            abp[0, l] = scenario.vec(n["alloc_by_priority"].get(str(prio), {}))
        s.nodes_upsert(total, total, index=[1], alloc_by_prio=abp, taints=[[]], labels=[[]])
    c.set_jobs([case["job"]], {"A": 0}, {})
    pod, _ = s.select_node(0)
    hist = s.excluded_nodes(0)
    if pod.node >= 0:   # (a case that only fails at the priority the table asks about: the whole selection finds the node at another one)
        assert hist == []
        return
    assert len(hist) == 1 and hist[0][-1] == 1, hist                       # one node, one reason: `reason != nil` (:405)
    assert hist[0][0] == NODE_EXPECT[case["name"]], (case["name"], hist)


def test_first_resource_in_factory_order_names_the_reason(lib):
    """nodematching_test.go:655-712: with cpu AND a later resource short, the reason names cpu; with only the later one short, that one.  Here through the pinned
    (evicted-job) selection, which runs DynamicJobRequirementsMet on exactly one node (nodedb.go:583-594, 897-920): the other nodes are implicit."""
    wl = W.small_random(n_nodes=8, n_jobs=40, n_queues=2, seed=3, occupied=0.0, gangs=0)
    s = W.load(lib, wl)
    W.set_jobs(s, wl)
    j = 0
    req = wl.job_req[j]
    lvl = list(s.priorities).index(int(wl.config.pc_priority[wl.job_pc[j]]))
    # node 2: everything short -> the first column; node 3: only the last non-zero column of the request short
    cols = [r for r in range(wl.job_req.shape[1]) if req[r] > 0]
    assert len(cols) >= 2
    for node, short in ((2, cols), (3, cols[-1:])):
        a = s.get_nodes_alloc([node])[0].copy()
        for r in short:
            a[:, r] = req[r] - 1
        s.node_upsert(node, a)
        pod, _ = s.select_node(j, node)
        assert pod.node < 0
        hist = s.excluded_nodes(j)
        assert hist == sorted(hist)
        d = {h[0]: h for h in hist}
        assert d["implicit"][-1] == wl.num_nodes - 1
        kind, col, _, _, required, available, count = d["insufficient_resources"]
        assert (col, required, available, count) == (short[0], int(req[short[0]]), int(req[short[0]]) - 1, 1), hist
        assert lvl >= 0
    # ...and a pinned selection that succeeds leaves nothing on record (nodedb_test.go:96-120: the histogram is empty)
    pod, _ = s.select_node(j, 5)
    assert pod.node == 5 and s.excluded_nodes(j) == []
    s.close()


def test_disallowed_resource_excludes_every_node(lib):
    """nodedb_test.go:780-796 (TestSelectNodeForJob_DisallowedJobResources): NumExcludedNodesByReason == {disallowedResourceRequested: NumNodes}"""
    wl = W.small_random(n_nodes=6, n_jobs=30, n_queues=2, seed=5, occupied=0.0, gangs=0)
    dis = np.zeros(wl.job_req.shape[1], dtype=np.uint8)
    dis[W.GPU] = 1
    wl.config.disallowed_resource = dis
    wl.job_req[:, W.GPU] = 0
    wl.job_req[1, W.GPU] = 1
    s = W.load(lib, wl)
    W.set_jobs(s, wl)
    pod, _ = s.select_node(1)
    assert pod.node < 0
    assert [(h[0], h[-1]) for h in s.excluded_nodes(1)] == [("disallowed_resource", wl.num_nodes)]
    pod, _ = s.select_node(0)
    assert pod.node >= 0 and s.excluded_nodes(0) == []   # a job that got a node: nothing on record (its histogram would depend on the iterator's order before the match)
    s.close()


def histograms(lib, wl, prepare=None):
    s = W.load(lib, wl)
    W.prepare(s, wl)
    if prepare:
        prepare(s)
    res = s.schedule_round()
    out = {}
    for j in range(wl.num_jobs):
        try:
            out[j] = s.excluded_nodes(j)
        except SchedError as e:
            out[j] = ("error", e.code if hasattr(e, "code") else str(e)[:24])
    s.close()
    return res, out


SEEDS = [dict(seed=1), dict(seed=2, away=True), dict(seed=3, ragged=True), dict(seed=5, offgrid=3), dict(seed=6, ragged=True, away=True), dict(seed=7, burst=(60, 25)),
         dict(seed=10, away=True, offgrid=3), dict(seed=12, ragged=True, away=True), dict(seed=15, ragged=True, offgrid=3)]


def check_against_oracle(lib, oracle_lib, wl, prepare=None):
    res, got = histograms(lib, wl, prepare)
    ores, exp = histograms(oracle_lib, wl, prepare)
    assert res.scheduled == ores.scheduled and res.preempted == ores.preempted
    assert got == exp, [(j, exp[j], got[j]) for j in exp if exp[j] != got[j]][:3]
    reasons = np.asarray(ores.job_unschedulable_reason)
    n = 0
    for j, h in exp.items():
        if h and h[0] != "error":
            n += 1
            assert reasons[j] != 0 or wl.job_gang[j] >= 0 or wl.job_node[j] >= 0, (j, h)
            if wl.job_gang[j] < 0:   # queue_scheduler_test.go:676-690 (gang members are exempt there too)
                total = sum(x[-1] for x in h)
                # (urgency preemption disabled: a gate-passed record counts a node the failed fair-preemption walk gave up for its static requirements ON TOP of its
                #  type's or the gate walk's count, nodedb.go:996-1006 — the sum may exceed NumNodes, and then there is no implicit rest; soak seed 101099)
                assert total == wl.num_nodes or (wl.config.disable_urgency_scheduling and total > wl.num_nodes), (j, h)
    return n, exp


@pytest.mark.parametrize("kw", SEEDS, ids=[str(k) for k in SEEDS])
def test_rounds_match_the_oracle_cpu_build(hostsim_lib, oracle_lib, kw):
    n, exp = check_against_oracle(hostsim_lib, oracle_lib, W.small_random(**kw))
    assert n > (0 if "burst" in kw else 20)   # (a rate-limited round stops early: few attempts fail)
    kinds = {x[0] for h in exp.values() if h and h[0] != "error" for x in h}
    assert "implicit" in kinds


@pytest.mark.gpu
@pytest.mark.parametrize("kw", SEEDS + [dict(seed=21, n_nodes=3000, n_jobs=9000, n_queues=12, gangs=12), dict(seed=22, n_nodes=700, n_jobs=4000, n_queues=6, ragged=True)],
                         ids=lambda k: str(k))
def test_rounds_match_the_oracle_gpu(hip_lib, oracle_lib, kw):
    n, _ = check_against_oracle(hip_lib, oracle_lib, W.small_random(**kw))
    assert n > (0 if "burst" in kw else 20)


# ---- a selection that FOUND a node at the job's priority and still ended without one (nodedb.go:747-789): only with urgency preemption disabled — with it on, the sweep
# reaches the gate's node at the latest.  The map then holds the gate's walk up to the node it stopped at, plus (fair-share preemption on) the static reasons of the
# nodes the failed walk over the evicted table found room on (:996-1006).  Until round 6 both implementations answered ASCHED_ERR_UNSUPPORTED here.
GATE_SEEDS = [dict(seed=31, occupied=0.9), dict(seed=32, occupied=1.0, away=True), dict(seed=33, occupied=0.9, ragged=True), dict(seed=34, occupied=1.0, offgrid=3),
              dict(seed=35, occupied=0.9, away=True, ragged=True)]
METHOD_URGENCY = 4


def gate_workload(kw, fair):
    wl = W.small_random(**kw)
    wl.config.disable_urgency_scheduling = True
    wl.config.disable_fairshare_scheduling = not fair
    return wl


def check_gate_passed(lib, oracle_lib, kw, fair):
    with_urgency = W.small_random(**kw)
    with_urgency.config.disable_fairshare_scheduling = not fair
    s = W.load(oracle_lib, with_urgency); W.prepare(s, with_urgency); r = s.schedule_round(); s.close()
    urgent = [j for j, m in r.scheduled_method.items() if m == METHOD_URGENCY]   # jobs that need the sweep: with it disabled their selection passes the gate and fails
    n, exp = check_against_oracle(lib, oracle_lib, gate_workload(kw, fair))
    assert not any(h and h[0] == "error" and h[1] == ERR_UNSUPPORTED for h in exp.values()), "a gate-passed record is produced, not refused"
    return len(urgent), n


@pytest.mark.parametrize("fair", [True, False], ids=["fair-preemption-on", "fair-preemption-off"])
@pytest.mark.parametrize("kw", GATE_SEEDS, ids=[str(k) for k in GATE_SEEDS])
def test_gate_passed_records_match_the_oracle_cpu_build(hostsim_lib, oracle_lib, kw, fair):
    check_gate_passed(hostsim_lib, oracle_lib, kw, fair)


def test_gate_passed_selections_occur_in_those_rounds(oracle_lib):
    assert sum(check_gate_passed(oracle_lib, oracle_lib, kw, False)[0] for kw in GATE_SEEDS) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("fair", [True, False], ids=["fair-preemption-on", "fair-preemption-off"])
@pytest.mark.parametrize("kw", GATE_SEEDS + [dict(seed=36, n_nodes=2000, n_jobs=8000, n_queues=10, occupied=0.95)], ids=lambda k: str(k))
def test_gate_passed_records_match_the_oracle_gpu(hip_lib, oracle_lib, kw, fair):
    check_gate_passed(hip_lib, oracle_lib, kw, fair)


def test_every_kind_of_reason_occurs_in_the_random_rounds(oracle_lib):
    seen = set()
    for kw in SEEDS:
        _, h = histograms(oracle_lib, W.small_random(**kw))
        seen |= {x[0] for v in h.values() if v and v[0] != "error" for x in v}
    assert {"implicit", "untolerated_taint", "missing_label", "unmatched_label", "insufficient_resources"} <= seen, seen


def test_record_limit_and_switch(lib, oracle_lib):
    wl = W.small_random(seed=4)
    _, full = histograms(lib, wl)
    with_record = [j for j, h in full.items() if h and h[0] != "error"]
    assert len(with_record) > 10
    _, off = histograms(lib, wl, lambda s: s.set_excluded_nodes(0))
    assert all(h == [] for h in off.values())
    _, two = histograms(lib, wl, lambda s: s.set_excluded_nodes(2))
    kept = [j for j in with_record if two[j] and two[j][0] != "error"]
    dropped = [j for j in with_record if two[j] and two[j][0] == "error"]
    # two records of attempts over the iterators are kept (the fixed-size outcomes — a pinned job's node, a disallowed resource — need none); the rest fail loudly
    # (ASCHED_ERR_UNSUPPORTED), not with a wrong histogram
    assert len(dropped) > 5 and len(kept) + len(dropped) == len(with_record)   # (which ones: compared with the oracle below — pinned outcomes need no record and are never dropped)
    assert all(two[j] == full[j] for j in kept)
    _, otwo = histograms(oracle_lib, wl, lambda s: s.set_excluded_nodes(2))
    assert {j: (h if not h or h[0] != "error" else "error") for j, h in two.items()} == {j: (h if not h or h[0] != "error" else "error") for j, h in otwo.items()}
