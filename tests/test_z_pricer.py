"""The indicative gang pricer of a market-driven pool (the last piece): asched_price_job_on_nodes = MinPriceNodeScheduler.Schedule for one job against every
node, asched_price_gang = GangPricer.Price (scheduling/pricer/node_scheduler.go, gang_pricer.go).

Golden cases, transcribed by hand with the tables' own expectations: pricer/node_scheduler_test.go — TestSchedule_NodeChecks (:33-88, 4 cases), TestSchedule_JobChecks
(:90-199, 4 cases), TestSchedule_PreemptsExpectedJobs (:208-323, all 7 cases, incl. the ORDER of the preempted jobs) — and pricer/gang_pricer_test.go
TestMarketBasedGangScheduler_Price (:17-213, all 11 cases).  Job ages: the Go fixtures lease the jobs one after the other, a later index is a younger job; here the
run timestamp grows with the index.  Then oracle == CPU build of the device code == HIP library on seeded workloads (every node's price and the gang prices)."""
import numpy as np
import pytest

from armada_amd import workloads as W
from armada_amd.binding import Config, Scheduler

GI = 2 ** 30
MEM, CPU = 0, 1
PCS = {"pc0": (0, 1), "pc1": (1, 1), "pc2": (2, 1), "pc2np": (2, 0), "pc3": (3, 0)}   # testfixtures.TestPriorityClasses
PC_NAMES = sorted(PCS)
L_CLUSTER, L_POOL, L_UNKNOWN, L_NODETYPE = 10, 11, 12, 13   # interned label keys: "cluster" and "pool" are indexed (testfixtures.TestIndexedNodeLabels), "unknown" is not
REASON_GANG, REASON_JOB = 10, 11                            # ASCHED_REASON_JOB_DOES_NOT_FIT / _GANG_DOES_NOT_FIT (armada_sched.h)
NOT_INDEXED, NO_NODES = 101, 102


def build(lib, nodes, jobs, gang=None):
    """nodes: [(cpu, mem GiB, tainted, {label key: value})]; jobs: [(queue, cpu, mem GiB, pc, node or -1, bid, lease index)]; gang = (member rows, uniformity label key)"""
    cfg = Config(num_resources=2, indexed_col=[CPU, MEM], indexed_resolution=[1000, GI], pc_priority=[PCS[n][0] for n in PC_NAMES],
                 pc_preemptible=[PCS[n][1] for n in PC_NAMES], drf_multiplier=[1.0, 1.0], indexed_taint_keys=None, indexed_label_keys=[L_CLUSTER, L_POOL, L_NODETYPE])
    total = np.array([[m * GI, c * 1000] for c, m, _, _ in nodes], dtype=np.int64).reshape(len(nodes), 2)
    s = Scheduler(lib, cfg)
    s.nodes_upsert(total, taints=[[(5, 1, 1)] if t else [] for _, _, t, _ in nodes], labels=[sorted(l.items()) for _, _, _, l in nodes])
    queues = sorted({j[0] for j in jobs})
    req = np.array([[j[2] * GI, j[1] * 1000] for j in jobs], dtype=np.int64).reshape(len(jobs), 2)
    gid = [-1] * len(jobs); card = [1] * len(jobs); uni = [-1] * len(jobs)
    if gang:
        for r in gang[0]:
            gid[r], card[r], uni[r] = 0, len(gang[0]), gang[1]
    sel = [j[7] if len(j) > 7 else None for j in jobs]
    classes = sorted({tuple(sorted(x.items())) if x else () for x in sel})
    s.jobs_set(req, queue=[queues.index(j[0]) for j in jobs], pc=[PC_NAMES.index(j[3]) for j in jobs], node=[j[4] for j in jobs],
               scheduled_at_priority=[PCS[j[3]][0] for j in jobs], run_timestamp=[(1000 + int(j[6])) * 1_000_000 for j in jobs], submit_time=list(range(len(jobs))),
               gang_id=gid, gang_cardinality=card, gang_uniformity_label=uni, bid_price=[float(j[5]) for j in jobs],
               req_class=[classes.index(tuple(sorted(x.items())) if x else ()) for x in sel], class_tolerations=[[] for _ in classes], class_selectors=[list(c) for c in classes])
    s.round_prepare([1.0] * len(queues), [[] for _ in queues])
    return s


def job(q, cpu, mem=0, pc="pc2", node=-1, bid=0.0, i=0, selector=None):
    return (q, cpu, mem, pc, node, bid, i, selector)


# ---- TestSchedule_NodeChecks (node_scheduler_test.go:33-88): Test1Cpu16GiJob("A", PriorityClass1) against one node
NODE_CASES = {
    "job matches node": ((32, 256, False, {}), None, True),
    "node has untolerated taints": ((32, 256, True, {}), None, False),
    "node doesn't match selector": ((32, 256, True, {}), {L_NODETYPE: 7}, False),
    "node too small": ((1, 5, False, {}), None, False),
}


def run_node_case(lib, name):
    node, sel, ok = NODE_CASES[name]
    s = build(lib, [node], [job("A", 1, 16, "pc1", selector=sel)])
    scores, victims = s.price_job_on_nodes(0, now_ms=5000, detail_node=0)
    s.close()
    assert scores[0] == (ok, 0, 0.0) and victims == []


# ---- TestSchedule_JobChecks (:90-199): a 10-cpu / 25Gi node holding one 8-cpu / 16Gi job of queue B
JOB_CASES = {
    "single job preemption costs existing job price": (0.5, job("A", 8, 16), True, 0.5),
    "single job preemption costs existing job price 2": (1.0, job("A", 8, 16), True, 1.0),
    "job with impossible node requirements returns error": (0.5, job("A", 8, 16, selector={L_NODETYPE: 7}), False, 0.0),
    "job with impossible resource requirements returns error": (0.5, job("A", 20, 30), False, 0.0),
}


def run_job_case(lib, name):
    bid, j, ok, price = JOB_CASES[name]
    s = build(lib, [(10, 25, False, {})], [job("B", 8, 16, node=0, bid=bid), j])
    scores, victims = s.price_job_on_nodes(1, now_ms=5000, detail_node=0)
    s.close()
    assert scores[0] == (ok, 1 if ok else 0, price) and victims == ([0] if ok else [])


# ---- TestSchedule_PreemptsExpectedJobs (:208-323): (cpu of the job to schedule, node cpu, jobs on the node (queue, cpu, bid, pc), victims in order, price)
B, C, D = "B", "C", "D"
PREEMPT_CASES = {
    "same price jobs, tie break on runtime": (8, 10, [(B, 4, 1.0, "pc2"), (B, 4, 1.0, "pc2")], [1, 0], 1.0),
    "multiple differently priced jobs": (8, 10, [(B, 2, 1.5, "pc2"), (B, 2, 1.2, "pc2"), (C, 2, 1.3, "pc2"), (C, 2, 0.8, "pc2")], [3, 1, 2], 1.3),
    "multiple differently priced jobs 2": (12, 18, [(B, 2, 1.5, "pc2")] * 5 + [(D, 2, 0.4, "pc2")] * 4, [8, 7, 6, 5, 4, 3], 1.5),
    "multiple differently priced jobs 3": (8, 10, [(B, 2, 0.2, "pc2"), (B, 4, 0.5, "pc2")], [0, 1], 0.5),
    "0 price": (3, 10, [(B, 2, 1.0, "pc2"), (B, 2, 0.0, "pc2"), (B, 4, 5.0, "pc2")], [1], 0.0),
    "0 price 2": (8, 10, [(B, 2, 1.0, "pc0"), (B, 2, 0.0, "pc0")], [1], 0.0),
    "priority class has no bearing on preemption order": (8, 10, [(B, 2, 0.1, "pc0"), (B, 1, 0.3, "pc2"), (B, 2, 0.8, "pc2"), (C, 2, 0.5, "pc2"), (C, 2, 0.45, "pc2"), (C, 1, 0.15, "pc2")],
                                                          [0, 5, 1, 4, 3], 0.5),
}


def run_preempt_case(lib, name):
    cpu, node_cpu, on_node, order, price = PREEMPT_CASES[name]
    jobs = [job(q, c, pc=pc, node=0, bid=b, i=i) for i, (q, c, b, pc) in enumerate(on_node)] + [job("A", cpu)]
    s = build(lib, [(node_cpu, 0, False, {})], jobs)
    scores, victims = s.price_job_on_nodes(len(jobs) - 1, now_ms=5000, detail_node=0)
    s.close()
    assert scores[0][0] and round(scores[0][2], 8) == price and victims == order, (scores, victims)


# ---- TestMarketBasedGangScheduler_Price (gang_pricer_test.go:17-213): 16-cpu / 128Gi nodes labelled cluster=..., jobs of B (1 cpu 4Gi at 2.0; one 16-cpu at 3.0) and C (1 cpu
#      4Gi at 0.5) on them; the gang: 16-cpu, 1-cpu or 8-cpu jobs of queue A.  (nodes: label value + the jobs on each; gang member sizes; label; schedulable, price, reason)
b1, c1, bL = (B, 1, 4, 2.0), (C, 1, 4, 0.5), (B, 16, 128, 3.0)
FOUR = [("c1", [b1]), ("c1", [c1]), ("c1", [b1, c1]), ("c1", [b1, c1])]
GANG_CASES = {
    "should schedule job on nodes with lowest cost": (FOUR, [16], L_CLUSTER, True, 0.5, 0),
    "should schedule job with zero cost if capacity is available": ([("c1", []), ("c1", [c1]), ("c1", [b1, c1]), ("c1", [b1, c1])], [1], L_CLUSTER, True, 0.0, 0),
    "should schedule gang with zero cost if capacity is available": (FOUR, [1, 1], L_CLUSTER, True, 0.0, 0),
    "should schedule gang with with lowest cost even if both scheduled on the same node": ([("c1", [bL])], [8, 8], L_CLUSTER, True, 3.0, 0),
    "should schedule gang on nodes with lowest cost": (FOUR, [16, 16], L_CLUSTER, True, 2.0, 0),
    "should schedule gang on nodes with lowest cost 2": ([("c1", [bL]), ("c1", [c1])], [16, 16], L_CLUSTER, True, 3.0, 0),
    "should schedule gang on nodes with lowest cost 3": ([("c1", [b1]), ("c1", [c1]), ("c2", [b1, c1]), ("c2", [b1, c1])], [16, 16], L_CLUSTER, True, 2.0, 0),
    "should not schedule over members of the scheduling gang": ([("c1", [b1])], [16, 16], L_CLUSTER, False, 0.0, REASON_GANG),
    "cannot schedule gang where no nodes have uniformity label": (FOUR, [16, 16], L_UNKNOWN, False, 0.0, NOT_INDEXED),
    "cannot schedule gang where no nodes have uniformity label 2": (FOUR, [16, 16], L_POOL, False, 0.0, NO_NODES),
    "not enough nodes matching desired uniformity label": ([("c1", [b1]), ("c2", [c1])], [16, 16], L_CLUSTER, False, 0.0, REASON_GANG),
}
VALUES = {"c1": 21, "c2": 22}


def run_gang_case(lib, name):
    nodes_spec, sizes, label, ok, price, reason = GANG_CASES[name]
    nodes, jobs = [], []
    for n, (cl, on) in enumerate(nodes_spec):
        nodes.append((16, 128, False, {L_CLUSTER: VALUES[cl]}))
        for (q, cpu, mem, bid) in on:
            jobs.append(job(q, cpu, mem, "pc1", node=n, bid=bid, i=len(jobs)))
    members = []
    for cpu in sizes:
        members.append(len(jobs)); jobs.append(job("A", cpu, cpu * 8 if cpu > 1 else 4, "pc1", i=len(jobs)))
    s = build(lib, nodes, jobs, gang=(members, label))
    before = s.get_nodes_alloc().copy()
    r = s.price_gang(members, now_ms=5000)
    assert (s.get_nodes_alloc() == before).all(), "pricing must leave the NodeDb as it was"
    s.close()
    assert r == dict(evaluated=True, schedulable=ok, price=price, reason=reason), r


ALL = ([("node", n) for n in NODE_CASES] + [("job", n) for n in JOB_CASES] + [("preempt", n) for n in PREEMPT_CASES] + [("gang", n) for n in GANG_CASES])
RUN = {"node": run_node_case, "job": run_job_case, "preempt": run_preempt_case, "gang": run_gang_case}


@pytest.mark.parametrize("kind,name", ALL, ids=[k + ":" + n for k, n in ALL])
def test_reference_cases_oracle(oracle_lib, kind, name):
    RUN[kind](oracle_lib, name)


@pytest.mark.parametrize("kind,name", ALL, ids=[k + ":" + n for k, n in ALL])
def test_reference_cases_cpu_build(hostsim_lib, kind, name):
    RUN[kind](hostsim_lib, name)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,name", ALL, ids=[k + ":" + n for k, n in ALL])
def test_reference_cases_gpu(hip_lib, kind, name):
    RUN[kind](hip_lib, name)


# ---- seeded: after a market round, every node's price for left-over jobs and the price of gangs with and without a uniformity label
def _differential(lib, oracle, seed, n_nodes=40, n_jobs=600):
    import test_z_market_round as M
    rng = np.random.default_rng(seed)
    wl, bids, cutoff = M.market_case(seed)
    out = []
    for l in (oracle, lib):
        s = W.load(l, wl); W.set_jobs(s, wl, bid_price=bids); W.prepare(s, wl)
        s.set_market(True, cutoff)
        res = s.schedule_round()
        s.set_market(False)
        left = [j for j in range(wl.num_jobs) if wl.job_node[j] < 0 and j not in res.scheduled]
        singles = [j for j in left if wl.job_gang[j] < 0][:10]
        r = [s.price_job_on_nodes(j, now_ms=300_000, detail_node=int(np.argmax([sc[1] for sc in s.price_job_on_nodes(j, now_ms=300_000)[0]]))) for j in singles]
        r += [s.price_gang([j], now_ms=300_000) for j in singles]
        gangs = sorted({int(wl.job_gang[j]) for j in left if wl.job_gang[j] >= 0})[:4]
        r += [s.price_gang([int(j) for j in np.nonzero(wl.job_gang == g)[0]], now_ms=300_000) for g in gangs]
        out.append(r)
        s.close()
    assert out[0] == out[1]
    return out[0]


@pytest.mark.parametrize("seed", range(7000, 7012))
def test_prices_cpu_build_equal_oracle(hostsim_lib, oracle_lib, seed):
    _differential(hostsim_lib, oracle_lib, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(7000, 7006))
def test_prices_gpu_equal_oracle(hip_lib, oracle_lib, seed):
    _differential(hip_lib, oracle_lib, seed)
