"""TestCalculateTheoreticalShare and TestCalculateFairnessError (context/scheduling_test.go:248-336, 441-528): the two report figures a host derives from the
round's fair shares (armada_amd/constraints.py theoretical_share / fairness_error over asched_fair_shares and asched_drf_cost), with the Go tables' tolerances
(InDelta 1e-6 / 1e-5).  DominantResourceFairnessResourcesToConsider = [cpu]; cpu in millis."""
import pytest

from armada_amd import constraints as K
from armada_amd.binding import Config, Scheduler


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def sched(request):
    lib = request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])
    s = Scheduler(lib, Config(num_resources=1, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0]))
    yield s
    s.close()


def cpu(n):
    return [int(n * 1000)]


THEORETICAL = {   # name: (available, {queue: (weight, demand)}, base priority, expected)
    "Cluster Empty": (cpu(100), {}, 1.0, 1.0),
    "One user": (cpu(100), {"queueA": (1.0, cpu(1000))}, 1.0, 0.5),
    "Two users": (cpu(100), {"queueA": (1.0, cpu(1000)), "queueB": (1.0, cpu(1000))}, 1.0, 1.0 / 3),
    "One user with lower priority": (cpu(100), {"queueA": (0.5, cpu(1000))}, 1.0, 2.0 / 3),
    "One user with higher priority": (cpu(100), {"queueA": (2.0, cpu(1000))}, 1.0, 1.0 / 3),
    "One user with a low demand": (cpu(100), {"queueA": (1.0, cpu(1))}, 1.0, 0.99),
}


@pytest.mark.parametrize("name", list(THEORETICAL))
def test_calculate_theoretical_share(sched, name):
    total, queues, priority, want = THEORETICAL[name]
    names = sorted(queues)
    cds = [sched.drf_cost(queues[q][1], total) for q in names]                      # constrainedDemandShare (scheduling.go:267-270; AddQueueSchedulingContext is given demand = constrained demand)
    got = K.theoretical_share(sched, list(range(len(names))), [queues[q][0] for q in names], cds, priority, total)
    assert abs(got - want) <= 1e-6, (name, got, want)


FAIRNESS_ERROR = {   # name: ([(allocated cpu, DemandCappedAdjustedFairShare)], expected)
    "one queue, no error": ([(50, 0.5)], 0.0),
    "two queues, no error": ([(50, 0.5), (50, 0.5)], 0.0),
    "one queue with error": ([(40, 0.5)], 0.1),
    "two queues with error": ([(40, 0.5), (10, 0.5)], 0.5),
    "above fair share is not counted": ([(100, 0.5)], 0.0),
    "empty": ([], 0.0),
}


@pytest.mark.parametrize("name", list(FAIRNESS_ERROR))
def test_calculate_fairness_error(sched, name):
    rows, want = FAIRNESS_ERROR[name]
    got = K.fairness_error(sched, [cpu(a) for a, _ in rows], cpu(100), [s for _, s in rows])
    assert abs(got - want) <= 1e-5, (name, got, want)
