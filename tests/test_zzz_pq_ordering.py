"""QueueCandidateGangIteratorPQ.Less (queue_scheduler.go:738-798) at function level.

tests/golden/pq_ordering_cases.json: the six sort.Sort tests of queue_scheduler_test.go:995-1164 (7 cases), items / pq flags / expected
order evaluated mechanically from the test bodies.  asched_pq_order sorts with the oracle's Less and, on the device builds, with the
round's own pqLess inside the auxiliary kernel — which also checks pair by pair that the packed lexicographic key the fast path orders
queues by gives the same verdicts.  A seeded sweep over random finite non-negative costs extends that equivalence
check beyond the reference's handful of items.
"""
import numpy as np
import pytest

from armada_amd.binding import Config, Scheduler
from golden_io import ids, load

CASES = load("pq_ordering")


def handle(lib):
    return Scheduler(lib, Config(num_resources=2, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0, 1.0]))


def to_items(case):
    names = sorted(i["queue"] for i in case["items"])
    return [dict(proposed_cost=i["proposedQueueCost"], current_cost=i["currentQueueCost"], budget=i["queueBudget"], item_size=i["itemSize"],
                 pc_priority=i["priorityClassPriority"], scheduling_priority=i["schedulingPriority"], name_rank=names.index(i["queue"])) for i in case["items"]]


def run_case(lib, case):
    order, agrees = handle(lib).pq_order(to_items(case), case["prioritiseLargerJobs"], case["compareSchedulingPriority"])
    assert [case["items"][k]["var"] for k in order] == case["expectedOrder"]
    assert agrees, "the fast path's packed key disagrees with Less on these items"


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_oracle(oracle_lib, case):
    run_case(oracle_lib, case)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_hostsim(hostsim_lib, case):
    run_case(hostsim_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_gpu(hip_lib, case):
    run_case(hip_lib, case)


def random_items(rng, n):
    """costs drawn from a small set so that ties on every comparison level occur"""
    vals = [0.0, 0.125, 0.25, 0.5, 0.75, 1.0, 1.5]
    return [dict(proposed_cost=float(rng.choice(vals)), current_cost=float(rng.choice(vals)), budget=float(rng.choice(vals)), item_size=float(rng.choice(vals)),
                 pc_priority=int(rng.choice([0, 1, 30000])), scheduling_priority=int(rng.choice([0, 29000, 30000])), name_rank=i) for i in range(n)]


def sweep(lib, oracle):
    rng = np.random.default_rng(11)
    a, b = handle(lib), handle(oracle)
    for t in range(60):
        items = random_items(rng, int(rng.integers(2, 33)))
        rng.shuffle(items)   # name ranks stay a permutation: the queue name is the last, always decisive, tie-break
        pl, cs = bool(t & 1), bool(t & 2)
        o1, agrees = a.pq_order(items, pl, cs)
        o2, _ = b.pq_order(items, pl, cs)
        assert o1 == o2, f"trial {t}: order differs from the oracle"
        assert agrees, f"trial {t}: packed key disagrees with Less"


def test_sweep_hostsim(hostsim_lib, oracle_lib):
    sweep(hostsim_lib, oracle_lib)


@pytest.mark.gpu
def test_sweep_gpu(hip_lib, oracle_lib):
    sweep(hip_lib, oracle_lib)


# ---- TestQueueCandidateGangIteratorPQ_HomeBeforeAway (queue_scheduler_test.go:698-716), by hand: two items, no table.  With preemptCrossPoolJobsFirst the
# home item sorts before the cross-pool ("away") item whatever their costs and priorities (Less :744-746); without it priority-class priority decides.
HOME = dict(proposed_cost=100.0, pc_priority=1, name_rank=0, away=0)
AWAY = dict(proposed_cost=1.0, pc_priority=1000, name_rank=1, away=1)   # CalculateAwayQueueName("q") = "q-away" sorts after "q"


def home_before_away(lib):
    s = handle(lib)
    for items, first in (([HOME, AWAY], 0), ([AWAY, HOME], 1)):
        order, _ = s.pq_order(items, False, True, preempt_cross_pool_jobs_first=True)
        assert order[0] == first                        # pq.Less(home, away) and !pq.Less(away, home)
        order, _ = s.pq_order(items, False, True, preempt_cross_pool_jobs_first=False)
        assert order[0] == 1 - first                    # flag off: the existing ordering applies and the away item wins (equal scheduling priorities, lower proposed cost)


def test_home_before_away_oracle(oracle_lib):
    home_before_away(oracle_lib)


def test_home_before_away_hostsim(hostsim_lib):
    home_before_away(hostsim_lib)


@pytest.mark.gpu
def test_home_before_away_gpu(hip_lib):
    home_before_away(hip_lib)
