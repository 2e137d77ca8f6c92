"""jobdb.SchedulingOrderCompare (jobdb/comparison.go:49-107) at function level: the order the library pre-sorts every queue's jobs in
(asched_jobs_set: counting sort by queue, packed keys, buckets sorted on host threads) and the round takes evicted jobs in.

tests/golden/job_priority_comparer_cases.json: the 7 ordering cases of TestJobPriorityComparer (comparison_test.go:13-74; the two cases
about equal ids concern jobDb identity).  Each pair becomes a two-job queue; asched_scheduling_order must list the jobs in the order the
expected sign says.  A seeded sweep compares the library's order with the oracle's comparator sort on tie-rich random job tables."""
import numpy as np
import pytest

from armada_amd.binding import Config, Scheduler
from golden_io import ids, load

CASES = load("job_priority_comparer")


def handle(lib, pc_priorities):
    return Scheduler(lib, Config(num_resources=2, indexed_col=[0], indexed_resolution=[1], pc_priority=list(pc_priorities), pc_preemptible=[1] * len(pc_priorities),
                                 drf_multiplier=[1.0, 1.0]))


def run_case(lib, case):
    a, b = case["a"], case["b"]
    pcs = sorted({a["pcPriority"], b["pcPriority"]})
    s = handle(lib, pcs)
    s.nodes_upsert(np.array([[100, 100]], dtype=np.int64))
    jobs = sorted([a, b], key=lambda j: j["id"])           # table index order == id order (the final tie-break, comparison.go:99-105)
    s.jobs_set(np.ones((2, 2), dtype=np.int64), queue=[0, 0], pc=[pcs.index(j["pcPriority"]) for j in jobs], queue_priority=[j["priority"] for j in jobs],
               submit_time=[j["submittedTime"] for j in jobs], node=[0 if j["active"] else -1 for j in jobs], scheduled_at_priority=[j["pcPriority"] for j in jobs],
               run_timestamp=[j["activeRunTimestamp"] for j in jobs])
    order = s.scheduling_order(0)
    first = jobs[order[0]]["id"]
    assert first == ("a" if case["expected"] < 0 else "b"), (order, case)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_comparer_oracle(oracle_lib, case):
    run_case(oracle_lib, case)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_comparer_hostsim(hostsim_lib, case):
    run_case(hostsim_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_comparer_gpu(hip_lib, case):
    run_case(hip_lib, case)


def sweep(lib, oracle, m):
    rng = np.random.default_rng(m)
    pcs = [0, 1, 3, 30000]
    a, b = handle(lib, pcs), handle(oracle, pcs)
    node = rng.integers(-1, 2, size=m).astype(np.int32)      # a third queued, two thirds with an active run
    args = dict(queue=rng.integers(0, 5, size=m).astype(np.int32), pc=rng.integers(0, 4, size=m).astype(np.int32),
                queue_priority=rng.integers(0, 3, size=m).astype(np.uint32), submit_time=rng.integers(0, 4, size=m), node=node,
                scheduled_at_priority=np.zeros(m, np.int32), run_timestamp=rng.integers(0, 3, size=m))
    for s in (a, b):
        s.nodes_upsert(np.array([[10**9, 10**9]] * 2, dtype=np.int64))
        s.jobs_set(np.ones((m, 2), dtype=np.int64), **args)
    for q in range(6):
        assert a.scheduling_order(q) == b.scheduling_order(q), f"queue {q}"
    assert sum(len(a.scheduling_order(q)) for q in range(5)) == m


@pytest.mark.parametrize("m", [50, 3000, 70000])   # 70 000: above the size at which the buckets are sorted on several threads
def test_sweep_hostsim(hostsim_lib, oracle_lib, m):
    sweep(hostsim_lib, oracle_lib, m)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [50, 70000])
def test_sweep_gpu(hip_lib, oracle_lib, m):
    sweep(hip_lib, oracle_lib, m)
