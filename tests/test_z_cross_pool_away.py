"""Cross-pool away jobs (asched_jobs.away, asched_config.preempt_cross_pool_jobs_first).

A running job whose latest run belongs to another pool (context.IsHomeJob false, context/util.go:9-16) is
  * bound at CrossPoolPriority (-1) whatever its run says, so any home job can urgency-preempt it first (bindJobToNodeInPlace, nodedb.go:1055-1068),
  * not held to this pool's floating-resource limits (context/scheduling.go:583-594; the reference case "floating resources - away jobs" of
    TestGangScheduler is in tests/golden/gang_scheduler_cases.json),
  * ordered after home gangs by Less when preemptCrossPoolJobsFirst is set (queue_scheduler.go:744-746), in the scheduling loop and in the
    eviction-order replay (pqs.go:603-606),
  * accounted against its queue's "<queue>-away" context: the caller passes that context's index as the job's queue.
The rounds here are compared job by job between the oracle and the CPU build of the device code (GPU: `-m gpu`).
"""
import copy

import numpy as np
import pytest

from armada_amd import workloads as W
from armada_amd.binding import CROSS_POOL_PRIORITY, EVICTED_PRIORITY
import bench


def with_away(seed, prefer_home, frac=0.3):
    rng = np.random.default_rng(seed)
    wl = W.small_random(n_nodes=int(rng.integers(6, 80)), n_jobs=int(rng.integers(100, 1500)), n_queues=int(rng.integers(2, 6)), seed=seed,
                        occupied=float(rng.choice([0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 4)), away=False)
    q = wl.num_queues
    # every queue gets an away context beside it (queues q .. 2q-1, no queued jobs, same weight); a share of the running jobs belong to another pool
    wl = copy.copy(wl)
    running = np.nonzero(np.asarray(wl.job_node) >= 0)[0]
    away = np.zeros(wl.num_jobs, dtype=np.uint8)
    pick = running[rng.random(len(running)) < frac]
    away[pick] = 1
    jq = np.asarray(wl.job_queue).copy()
    jq[pick] += q
    wl.job_queue = jq
    wl.job_away = away
    wl.queue_weight = list(wl.queue_weight) + list(wl.queue_weight)
    wl.queued = [list(x) for x in wl.queued] + [[] for _ in range(q)]
    cfg = copy.copy(wl.config)
    cfg.preempt_cross_pool_jobs_first = bool(prefer_home)
    wl.config = cfg
    return wl


def both(lib, oracle, wl):
    out = []
    for l in (lib, oracle):
        s = W.load(l, wl); W.prepare(s, wl)
        out.append(s.schedule_round())
        s.close()
    assert bench.round_diff(out[0], out[1]) == []
    return out[0]


@pytest.mark.parametrize("seed", range(40))
def test_rounds_with_away_jobs_equal_oracle(hostsim_lib, oracle_lib, seed):
    both(hostsim_lib, oracle_lib, with_away(810000 + seed, prefer_home=seed % 2 == 0))


def test_away_job_is_bound_at_cross_pool_priority(hostsim_lib, oracle_lib):
    """populateNodeDb binds a running away job at priority -1: allocatable drops at levels -2 and -1 only, whatever the run's priority is"""
    wl = with_away(810100, True, frac=1.0)
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl); W.prepare(s, wl)
        alloc = s.get_nodes_alloc()                  # [N][P][R]
        prios = list(s.priorities)
        base = np.asarray(wl.node_allocatable if wl.node_allocatable is not None else wl.node_total, dtype=np.int64)
        preemptible = list(wl.config.pc_preemptible)
        for lvl, p in enumerate(prios):
            want = base.copy()
            for j in np.nonzero(np.asarray(wl.job_node) >= 0)[0]:
                cutoff = CROSS_POOL_PRIORITY if preemptible[wl.job_pc[j]] else 2 ** 31 - 1   # priorityCutoffFor (nodedb.go:1329-1334) of priority -1
                if p <= cutoff:
                    want[wl.job_node[j]] -= wl.job_req[j]
            assert (alloc[:, lvl, :] == want).all(), (lib, p)
        s.close()


def test_home_gangs_order_before_away_gangs(oracle_lib, hostsim_lib):
    """preemptCrossPoolJobsFirst: with it, every evicted home job is rescheduled before any evicted away job gets its turn, so on a cluster the queued
    home jobs then fill, away jobs lose their place first; without it the plain cost order decides.  Both libraries agree with the oracle either way and
    the flag changes the outcome on this input."""
    a = both(hostsim_lib, oracle_lib, with_away(810200, True, frac=0.5))
    b = both(hostsim_lib, oracle_lib, with_away(810200, False, frac=0.5))
    assert set(a.preempted) != set(b.preempted) or a.scheduled != b.scheduled


@pytest.mark.gpu
def test_rounds_with_away_jobs_gpu(hip_lib, oracle_lib):
    for seed in range(8):
        both(hip_lib, oracle_lib, with_away(810000 + seed, prefer_home=seed % 2 == 0))
