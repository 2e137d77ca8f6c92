"""The experimental fairness optimiser as part of the round (preempting_queue_scheduler.go:224-253, 666-710; optimising_queue_scheduler.go:58-180;
optimiser/gang_scheduler.go:45-254): asched_set_optimiser + asched_schedule_round.

The reference's own multi-round case (`TestPreemptingQueueScheduler/optimiser`, transcribed mechanically into tests/golden/pqs_cases.json) runs with the other PQS
goldens on the oracle, the CPU build and the HIP library.  Here: seeded crowded rounds with the optimiser on, oracle against the CPU build of the device code and
(-m gpu) the HIP library — jobs scheduled with method 6, their victims, the end-of-phase merge of the optimiser's lists into the round's (a job the optimiser
schedules, preempts, schedules again and preempts again comes out preempted, pqs.go:232-249), multi-member gangs (updateState between the members), limits.

Reference behaviour restated as it is: the round FAILS when a job that was scheduled earlier in the round is met again under a scheduling key the optimiser has
registered as unfeasible ("job already marked successful", context/queue.go:232-234 via queue_scheduler.go:398-413): oracle and product return ASCHED_ERR_INTERNAL
there.  A job scheduled in the round, evicted by the oversubscribed evictor, not rescheduled and then placed by the optimiser on ANOTHER node holds resources on two
nodes until the unbinding: the product keeps the old node's share as a "ghost" (dev.h optGhost)."""
import numpy as np
import pytest

import bench
from armada_amd import workloads as W
from armada_amd.binding import SchedError


def _case(seed):
    rng = np.random.default_rng(seed)
    wl = W.small_random(n_nodes=int(rng.integers(4, 60)), n_jobs=int(rng.integers(100, 1500)), n_queues=int(rng.integers(2, 8)), seed=seed,
                        occupied=float(rng.choice([0.8, 0.95, 1.0])), gangs=int(rng.integers(0, 5)),
                        burst=None if rng.random() < 0.5 else (int(rng.integers(20, 400)), int(rng.integers(10, 100))))
    wl.job_run_ts = (np.arange(wl.num_jobs, dtype=np.int64) * 7919 % 100003) * 1_000_000
    kw = dict(min_improvement_pct=float(rng.choice([0, 5, 50])), max_jobs_per_round=int(rng.choice([1, 5, 40])), now_ms=200_000,
              max_job_size_to_preempt=None if rng.random() < 0.6 else [64 * W.Gi, 4000, 0, 0], min_job_size_to_schedule=None if rng.random() < 0.7 else [0, 2000, 0, 0],
              max_resource_fraction_to_schedule=None if rng.random() < 0.6 else [0.2, 0.2, 1.0, 1.0])
    return wl, kw


def _gang_case(seed):
    """a full cluster, one preemptible priority class (no urgency preemption), every queue protected from eviction: what the main passes cannot place — multi-member
    gangs included — only the optimiser can"""
    import copy
    rng = np.random.default_rng(seed)
    wl = W.small_random(n_nodes=int(rng.integers(6, 40)), n_jobs=int(rng.integers(100, 600)), n_queues=int(rng.integers(2, 6)), seed=seed, occupied=1.0, gangs=int(rng.integers(3, 12)))
    wl.config = copy.copy(wl.config); wl.config.protected_fraction_of_fair_share = 1.0
    wl.job_pc[:] = 0
    return wl, dict(min_improvement_pct=0.0, max_jobs_per_round=60, now_ms=200_000)


def _run(lib, wl, kw):
    s = W.load(lib, wl); W.prepare(s, wl)
    s.set_optimiser(True, **kw)
    try:
        return s.schedule_round(), 0
    except SchedError as e:
        return None, e.code
    finally:
        s.close()


def compare(lib, oracle, seeds, case=None):
    used = errors = refused = 0
    for seed in seeds:
        wl, kw = (case or _case)(seed)
        want, wc = _run(oracle, wl, kw)
        got, gc = _run(lib, wl, kw)
        if gc == -2:           # corner (2) of the module docstring
            refused += 1
            continue
        assert wc == gc, (seed, wc, gc)
        if wc:
            errors += 1
            continue
        assert bench.round_diff(want, got) == [], (seed, bench.round_diff(want, got))
        used += sum(1 for j, m in want.scheduled_method.items() if m == 6 and (case is None or wl.job_gang[j] >= 0))
    return used, errors, refused


def test_optimiser_rounds_match_the_oracle(hostsim_lib, oracle_lib):
    used, errors, refused = compare(hostsim_lib, oracle_lib, range(160))
    assert used >= 25 and refused == 0, (used, errors, refused)


def test_optimiser_places_gangs(hostsim_lib, oracle_lib):
    """multi-member gangs through the optimiser: updateState between the members (the victims leave, the member is bound, the victims' queues get cheaper)"""
    used, errors, refused = compare(hostsim_lib, oracle_lib, range(120), case=_gang_case)
    assert used >= 8 and refused == 0, (used, errors, refused)


def test_optimiser_off_changes_nothing(hostsim_lib, oracle_lib):
    wl, _ = _case(7)
    a = W.load(hostsim_lib, wl); W.prepare(a, wl); r0 = a.schedule_round()
    W.prepare(a, wl); a.set_optimiser(True, max_jobs_per_round=0); r1 = a.schedule_round()     # MaximumJobsPerRound 0: the loop never runs
    W.prepare(a, wl); a.set_optimiser(False); r2 = a.schedule_round()
    assert bench.round_diff(r0, r1) == [] and bench.round_diff(r0, r2) == []
    a.close()


@pytest.mark.gpu
def test_optimiser_rounds_match_the_oracle_gpu(hip_lib, oracle_lib):
    used, errors, refused = compare(hip_lib, oracle_lib, range(60))
    assert used >= 8 and refused == 0, (used, errors, refused)
    used, errors, refused = compare(hip_lib, oracle_lib, range(60), case=_gang_case)
    assert used >= 4 and refused == 0, (used, errors, refused)


@pytest.mark.gpu
def test_optimiser_round_at_scale_gpu(hip_lib, oracle_lib):
    """2 000 nodes 95 % occupied, 20 000 queued: k_opt_score over every node for every optimiser candidate"""
    wl = W.config3(seed=11, n_nodes=2000, n_jobs=20_000, n_queues=16, occupied=0.95)
    wl.global_burst, wl.queue_burst = 4000, 400
    wl.job_run_ts = (np.arange(wl.num_jobs, dtype=np.int64) * 7919 % 100003) * 1_000_000
    kw = dict(min_improvement_pct=5.0, max_jobs_per_round=40, now_ms=200_000)
    want, wc = _run(oracle_lib, wl, kw)
    got, gc = _run(hip_lib, wl, kw)
    assert wc == gc
    if not wc:
        assert bench.round_diff(want, got) == []
