"""The experimental fairness optimiser's node scoring: PreemptingNodeScheduler.Schedule for one job against every node and the
candidate selection of FairnessOptimisingGangScheduler.scheduleOnNodes — asched_optimiser_schedule_job (round_opt.h, k_opt_score).

Golden cases: transcribed by hand from internal/scheduler/scheduling/optimiser/node_scheduler_test.go — TestSchedule_NodeChecks (:32-88, 4 cases),
TestSchedule_JobChecks (:90-200, 8 cases) and TestSchedule_PreemptsExpectedJobs (:256-417, all 7 cases) — with the table's own expectations
(schedulingCost, maximumQueueImpact, the ORDER of the preempted jobs).  Where the Go test hands the scheduling context a total that is larger
than the node (extraTotalResource, CpuMem("100", "2000Gi")), a second, tainted node the job cannot use carries the difference.  Job ages: the Go
fixtures lease the jobs one after the other (WithNewRun reads the clock), so a later index is a younger job; here run_timestamp grows with the index.
Then: oracle == CPU build of the device code == HIP library on seeded workloads, every node's score included.
"""
import numpy as np
import pytest

from armada_amd import workloads as W
from armada_amd.binding import Config, Scheduler

GI = 2 ** 30
MEM, CPU = 0, 1
PCS = {"pc0": (0, 1), "pc1": (1, 1), "pc2": (2, 1), "pc2np": (2, 0), "pc3": (3, 0)}   # testfixtures.TestPriorityClasses
PC_NAMES = sorted(PCS)


def build(lib, nodes, jobs, queues, extra_total=None, extra_alloc=None):
    """nodes: [(cpu, mem, taint)], jobs: [(queue, cpu, mem, pc, node or -1, lease_ms)], queues: [(name, priority factor)].
    Returns a prepared Scheduler (fair shares with unlimited demand, like setUpSctx :202-246)."""
    cfg = Config(num_resources=2, indexed_col=[CPU, MEM], indexed_resolution=[1000, GI], pc_priority=[PCS[n][0] for n in PC_NAMES],
                 pc_preemptible=[PCS[n][1] for n in PC_NAMES], drf_multiplier=[1.0, 1.0], indexed_taint_keys=None)
    nodes = list(nodes)
    if extra_total:   # the scheduling context's total is larger than the node: a tainted node nobody tolerates carries the rest
        nodes.append((extra_total[0], extra_total[1], True))
    total = np.array([[m, c] for c, m, _ in nodes], dtype=np.int64)
    total[:, CPU] *= 1000
    s = Scheduler(lib, cfg)
    s.nodes_upsert(total, taints=[[(5, 1, 1)] if t else [] for _, _, t in nodes])
    qn = [q for q, _ in queues]
    req = np.array([[m, c * 1000] for _, c, m, _, _, _ in jobs], dtype=np.int64).reshape(len(jobs), 2)
    s.jobs_set(req, queue=[qn.index(j[0]) for j in jobs], pc=[PC_NAMES.index(j[3]) for j in jobs], node=[j[4] for j in jobs],
               scheduled_at_priority=[PCS[j[3]][0] for j in jobs], run_timestamp=[int(j[5]) * 1_000_000 for j in jobs], submit_time=list(range(len(jobs))))
    npc = len(PC_NAMES)
    alloc = np.zeros((len(qn), npc, 2), dtype=np.int64)
    for i, j in enumerate(jobs):
        if j[4] >= 0:
            alloc[qn.index(j[0]), PC_NAMES.index(j[3])] += req[i]
    for (q, pc, vec) in (extra_alloc or []):      # a job scheduled earlier in this round: in the queue's allocation, not yet a run (sctx.AddJobSchedulingContext)
        alloc[qn.index(q), PC_NAMES.index(pc)] += np.array(vec, dtype=np.int64)
    demand = np.tile(np.array([100000 * GI, 10000 * 1000], dtype=np.int64), (len(qn), 1))   # unlimitedDemand
    s.round_prepare([1.0 / pf for _, pf in queues], [[] for _ in qn], allocated_by_pc=alloc, demand=demand)
    return s


QA, QB, QC, QD = ("A", 10), ("B", 10), ("C", 10), ("D", 5)


def cpu_job(q, cpu, node=0, pc="pc2", i=0):
    return (q, cpu, 0, pc, node, 1000 + i)   # later index = later lease = younger


# ---- TestSchedule_PreemptsExpectedJobs (:256-417): (job to schedule cpu, queues, node cpu, extra total cpu, jobs on node, expected preemption order,
#      schedulingCost, maximumQueueImpact)
PREEMPT_CASES = {
    "preempt jobs - multiple same queue": (8, [QA, QB], 10, 0, [("B", 4), ("B", 4)], [1, 0], 0.8, 1.0),
    "preempt jobs - multiple different queue": (8, [QA, QB, QC], 10, 0, [("B", 2), ("B", 2), ("C", 2), ("C", 2)], [3, 1, 2], 0.6, 1.0),
    "preempt jobs - mixed queue priorities": (12, [QA, QB, QD], 18, 82, [("B", 2)] * 3 + [("D", 2)] * 6, [8, 7, 2, 6, 5, 1], 0.12, 2.0 / 3),
    "preempt jobs - smallest first": (8, [QA, QB], 10, 0, [("B", 2), ("B", 4)], [0, 1], 0.6, 1.0),
    "preempting jobs above fairshare - 0 cost": (3, [QA, QB, QC], 10, 0, [("B", 2), ("B", 2), ("B", 4)], [1], 0.0, 0.25),
    "preempting jobs of lower priority - 0 cost": (8, [QA, QB, QC], 10, 0, [("B", 2, "pc0"), ("B", 2, "pc0")], [1], 0.0, 0.5),
    "preempt jobs - expected order": (8, [QA, QB, QC], 10, 0, [("B", 2, "pc0"), ("B", 1), ("B", 2), ("C", 2), ("C", 2), ("C", 1)], [0, 5, 1, 4, 3], 0.5, 1.0),
}


def r8(x):
    return round(x * 1e8) / 1e8   # roundFloatHighPrecision, as the Go assertions apply it


def run_preempt_case(lib, name):
    want_cpu, queues, node_cpu, extra, on_node, order, cost, impact = PREEMPT_CASES[name]
    jobs = [cpu_job(j[0], j[1], 0, j[2] if len(j) > 2 else "pc2", i) for i, j in enumerate(on_node)]
    jobs.append(("A", want_cpu, 0, "pc2", -1, 0))
    s = build(lib, [(node_cpu, 0, False)], jobs, queues, extra_total=(extra, 0) if extra else None)
    r = s.optimiser_schedule_job(len(jobs) - 1, min_improvement_pct=-1e9, now_ms=10_000, per_node=True)
    assert r["node"] == 0
    assert r["preempted"] == order, (name, r)
    assert r8(r["cost"]) == cost and r8(r["impact"]) == r8(impact), (name, r)
    s.close()
    return r


@pytest.mark.parametrize("name", list(PREEMPT_CASES))
def test_preempts_expected_jobs_oracle(oracle_lib, name):
    run_preempt_case(oracle_lib, name)


@pytest.mark.parametrize("name", list(PREEMPT_CASES))
def test_preempts_expected_jobs_cpu_build(hostsim_lib, oracle_lib, name):
    a, b = run_preempt_case(oracle_lib, name), run_preempt_case(hostsim_lib, name)
    assert a == b   # floats included: the same IEEE operations in the same order


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PREEMPT_CASES))
def test_preempts_expected_jobs_gpu(hip_lib, oracle_lib, name):
    a, b = run_preempt_case(oracle_lib, name), run_preempt_case(hip_lib, name)
    assert a == b


# ---- TestSchedule_JobChecks (:90-200): node 10 cpu / 25Gi, job to schedule A pc2 8 cpu / 16Gi, one existing job of queue B (8 cpu / 16Gi), sctx total 100 cpu / 2000Gi
JOB_CASES = {
    # name: (existing job's priority class, scheduled-at priority, scheduled in the current round?, gang?, maximumJobSizeToPreempt cpu, expect success)
    "preempts job - preempted job scheduled in current round": ("pc2", 2, True, False, None, True),
    "preempts job - preempted job scheduled in previous round": ("pc2", 2, False, False, None, True),
    "will not preempt non-preemptible jobs": ("pc2np", 2, True, False, None, False),
    "will not preempt gang jobs": ("pc2", 2, True, True, None, False),
    "will not preempt jobs larger than maximumJobSizeToPreempt": ("pc2", 2, True, False, 1, False),
    "will not preempt jobs scheduled at higher priority - preempted job scheduled in current round": ("pc2", 3, True, False, None, False),
    "will not preempt jobs scheduled at higher priority - preempted job scheduled in previous round": ("pc2", 3, False, False, None, False),
}


def run_job_case(lib, name):
    pc, sap, current_round, gang, max_cpu, ok = JOB_CASES[name]
    cfg_jobs = [("B", 8, 16 * GI, pc, -1 if current_round else 0, 1000), ("A", 8, 16 * GI, "pc2", -1, 0)]
    s = build(lib, [(10, 25 * GI, False)], cfg_jobs, [QA, QB], extra_total=(90, 1975 * GI),
              extra_alloc=[("B", pc, [16 * GI, 8000])] if current_round else None)
    if gang:   # WithGangAnnotationsJobs: the existing job is a gang member
        s.close()
        s = build_with_gang(lib, pc)
    if current_round:
        s.bind(0, 0, sap)     # the job was scheduled on the node earlier in this round, at `sap` (pctx.ScheduledAtPriority)
    elif sap != PCS[pc][0]:
        s.unbind(0, 0); s.bind(0, 0, sap)   # a run scheduled at another priority than its class's
    r = s.optimiser_schedule_job(1, min_improvement_pct=-1e9, max_job_size_to_preempt=None if max_cpu is None else [0, max_cpu * 1000], now_ms=10_000, per_node=True)
    sc = r["scores"][0]
    assert sc[0] == ok, (name, r)
    if ok:
        assert r["node"] == 0 and r["preempted"] == [0] and r["cost"] == 0.08 and r["impact"] == 1.0, (name, r)
    else:
        assert r["node"] == -1 and sc[1:] == (0, 0.0, 0.0)
    s.close()
    return r


def build_with_gang(lib, pc):
    cfg = Config(num_resources=2, indexed_col=[CPU, MEM], indexed_resolution=[1000, GI], pc_priority=[PCS[n][0] for n in PC_NAMES],
                 pc_preemptible=[PCS[n][1] for n in PC_NAMES], drf_multiplier=[1.0, 1.0], indexed_taint_keys=None)
    total = np.array([[25 * GI, 10000], [1975 * GI, 90000]], dtype=np.int64)
    s = Scheduler(lib, cfg)
    s.nodes_upsert(total, taints=[[], [(5, 1, 1)]])
    req = np.array([[16 * GI, 8000], [16 * GI, 8000]], dtype=np.int64)
    s.jobs_set(req, queue=[1, 0], pc=[PC_NAMES.index(pc), PC_NAMES.index("pc2")], node=[-1, -1], gang_id=[0, -1], gang_cardinality=[2, 1])
    alloc = np.zeros((2, len(PC_NAMES), 2), dtype=np.int64)
    alloc[1, PC_NAMES.index(pc)] = [16 * GI, 8000]
    demand = np.tile(np.array([100000 * GI, 10000 * 1000], dtype=np.int64), (2, 1))
    s.round_prepare([0.1, 0.1], [[], []], allocated_by_pc=alloc, demand=demand)
    return s


@pytest.mark.parametrize("name", list(JOB_CASES))
def test_job_checks_oracle(oracle_lib, name):
    run_job_case(oracle_lib, name)


@pytest.mark.parametrize("name", list(JOB_CASES))
def test_job_checks_cpu_build(hostsim_lib, oracle_lib, name):
    assert run_job_case(oracle_lib, name) == run_job_case(hostsim_lib, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(JOB_CASES))
def test_job_checks_gpu(hip_lib, oracle_lib, name):
    assert run_job_case(oracle_lib, name) == run_job_case(hip_lib, name)


# ---- TestSchedule_NodeChecks (:32-88): a 1 cpu / 16Gi job of queue A (pc1) against one empty node
NODE_CASES = {"job matches node": ((32, 256 * GI, False), True), "node has untolerated taints": ((32, 256 * GI, True), False), "node too small": ((1, 5 * GI, False), False)}


@pytest.mark.parametrize("name", list(NODE_CASES))
def test_node_checks(oracle_lib, hostsim_lib, name):
    node, ok = NODE_CASES[name]
    for lib in (oracle_lib, hostsim_lib):
        s = build(lib, [node], [("A", 1, 16 * GI, "pc1", -1, 0)], [QA])
        r = s.optimiser_schedule_job(0, per_node=True)
        assert r["scores"][0] == (ok, 0, 0.0, 0.0) and r["node"] == (0 if ok else -1) and r["preempted"] == []
        s.close()


# ---- seeded workloads after a round: every node's score and the selection, oracle vs the product code
def _differential(lib, oracle, seed, n_nodes=60, n_jobs=900):
    wl = W.small_random(n_nodes=n_nodes, n_jobs=n_jobs, n_queues=4, seed=seed, occupied=0.95, gangs=seed % 3)
    wl.job_run_ts = (np.arange(wl.num_jobs, dtype=np.int64) * 7919 % 100003) * 1_000_000   # distinct lease times
    out = []
    for l in (oracle, lib):
        s = W.load(l, wl); W.prepare(s, wl)
        r = s.schedule_round()
        left = [j for j in range(wl.num_jobs) if wl.job_node[j] < 0 and j not in r.scheduled and wl.job_gang[j] < 0][:12]
        res = []
        for j in left:
            for pct, ms in ((-1e9, None), (5.0, None), (-1e9, [8 * GI, 2000, 0, 0])):
                res.append(s.optimiser_schedule_job(j, min_improvement_pct=pct, max_job_size_to_preempt=ms, now_ms=200_000, per_node=True))
                short = s.optimiser_schedule_job(j, min_improvement_pct=pct, max_job_size_to_preempt=ms, now_ms=200_000)   # without per-node scores the HIP library selects on the device
                assert short == {k: res[-1][k] for k in short}, (j, short, {k: res[-1][k] for k in short})
        out.append(res)
        s.close()
    assert out[0] == out[1]
    return out[0]


@pytest.mark.parametrize("seed", range(6))
def test_scores_cpu_build_equal_oracle(hostsim_lib, oracle_lib, seed):
    res = _differential(hostsim_lib, oracle_lib, 700 + seed)
    assert any(r["preempted"] for r in res) and any(r["node"] >= 0 and not r["preempted"] for r in res) or seed > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_scores_gpu_equal_oracle(hip_lib, oracle_lib, seed):
    _differential(hip_lib, oracle_lib, 700 + seed)


@pytest.mark.gpu
def test_scores_gpu_equal_oracle_at_scale(hip_lib, oracle_lib):
    """20k nodes, 95% occupied: the wide kernel with every CU busy"""
    res = _differential(hip_lib, oracle_lib, 42, n_nodes=20_000, n_jobs=60_000)
    assert any(r["preempted"] for r in res)


# ---- nodes holding more preemptible candidates than the per-thread list of k_opt_score keeps (OPT_MAXJ = 48): scored again with the entry list in HBM
def _crowded_node_case(lib, per_node=(130, 20, 120)):
    """three 64-cpu nodes filled with 0.25-cpu jobs of four queues (130 / 20 / 120 of them: the first and third overflow the private list; the second node's jobs
    are not preemptible), one queued 60-cpu job: it needs more than a hundred victims wherever it goes"""
    nodes = [(64, 0, False)] * 3
    jobs = []
    for n, cnt in enumerate(per_node):
        for i in range(cnt):
            jobs.append(("ABCD"[(i * 7 + n) % 4], 0.25, 0, "pc3" if n == 1 else ("pc2" if i % 5 else "pc1"), n, 1000 + (i * 37 + n * 11) % 997))
    jobs.append(("A", 60, 0, "pc2", -1, 0))
    s = build(lib, nodes, jobs, [QA, QB, QC, QD])
    out = s.optimiser_schedule_job(len(jobs) - 1, min_improvement_pct=-1e9, now_ms=5000, per_node=True)
    s.close()
    return out


def test_more_candidates_than_the_private_list_cpu_build(hostsim_lib, oracle_lib):
    want, got = _crowded_node_case(oracle_lib), _crowded_node_case(hostsim_lib)
    assert got == want
    assert want["node"] >= 0 and len(want["preempted"]) > 48                  # the chosen node really needs more victims than OPT_MAXJ
    assert sum(1 for sc in want["scores"] if sc[1] > 48) >= 2


@pytest.mark.gpu
def test_more_candidates_than_the_private_list_gpu(hip_lib, oracle_lib):
    assert _crowded_node_case(hip_lib) == _crowded_node_case(oracle_lib)
