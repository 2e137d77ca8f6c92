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


def test_away_jobs_do_not_take_the_round_off_the_fast_path(hostsim_lib, oracle_lib):
    """round 4: a job set with cross-pool away jobs used to clear the fast iteration for the whole round (round-3 review, weak #2).  The node evictor never takes an away
    job (pqs.go:102-104), so the first pass never meets one in a queue: it runs fast iterations; only the pass after the oversubscribed evictor is generic."""
    fast = generic = 0
    for seed in range(6):
        wl = with_away(810300 + seed, prefer_home=seed % 2 == 0, frac=[0.1, 0.3, 0.6][seed % 3])
        both(hostsim_lib, oracle_lib, wl)
        s = W.load(hostsim_lib, wl); W.prepare(s, wl); s.schedule_round(); st = s.round_stats(); s.close()
        fast += st["fast_iterations"]; generic += st["generic_iterations"]
    assert fast > generic > 0, (fast, generic)


@pytest.mark.gpu
def test_rounds_with_away_jobs_gpu(hip_lib, oracle_lib):
    for seed in range(8):
        both(hip_lib, oracle_lib, with_away(810000 + seed, prefer_home=seed % 2 == 0))


# ------------------------------------------------------------------------------------------------ the reference's own cross-pool tests, by hand
import os, sys  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402
import scenario  # noqa: E402


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


@pytest.mark.parametrize("name,pool_known,away,expected", [
    ("away when nodeDb pool differs", True, True, CROSS_POOL_PRIORITY),
    ("home when nodeDb pool matches run pool", True, False, None),
    ("home when nodeDb pool unset (cross-pool detection off)", False, True, None),
])
def test_bind_cross_pool_job_bucketing(lib, name, pool_known, away, expected):
    """nodedb_test.go:266-309 TestBindCrossPoolJobBucketing: BindJobToNode records CrossPoolPriority for a job whose run lives in another pool, its own
    priority otherwise — also when the NodeDb has not been told its pool (here: preempt_cross_pool_jobs_first off, scheduling_algo.go:759-764)"""
    cfg = F.TestSchedulingConfig()
    cfg["preempt_cross_pool_jobs_first"] = pool_known
    job = dict(F.Test1Cpu4GiJob("queue-a", F.PriorityClass0), away=away)
    c = scenario.Case(lib, cfg, [F.Test32CpuNode(F.TestPriorities)])
    c.set_jobs([job], {"queue-a": 0}, {})
    prio = F.TEST_PRIORITY_CLASSES[F.PriorityClass0]["priority"]
    c.sched.bind(0, 0, prio)
    got, ok = c.sched.get_scheduled_at_priority(0)     # priority, ok := nodeDb.GetScheduledAtPriority(job.Id())
    assert ok and got == (prio if expected is None else expected), name


@pytest.mark.parametrize("flag_on,preemptions,scheduled", [(True, 5, 5), (False, 0, 0)])
def test_cross_pool_preempted_first(lib, flag_on, preemptions, scheduled):
    """preempting_queue_scheduler_test.go:3504-3634 TestPreemptingQueueScheduler_CrossPoolPreemptedFirst: five preemptible priority-2 jobs of another pool
    fill a 5-cpu node; five queued home jobs of priority 0 preempt them when cross-pool jobs are bound at CrossPoolPriority (flag on) and cannot when the
    NodeDb does not know its pool (flag off).  Queue contexts as in the test: "A" with the home demand, "A-away" with the cross-pool allocation."""
    cfg = F.TestSchedulingConfig()
    cfg["preempt_cross_pool_jobs_first"] = flag_on
    node = F.TestNode(F.TestPriorities, {"cpu": "5", "memory": "64Gi"})
    cross = [dict(j, away=True, queue="A-away") for j in F.N1Cpu4GiJobs("A", F.PriorityClass2, 5)]
    home = F.N1Cpu4GiJobs("A", F.PriorityClass0, 5)
    jobs = cross + home
    c = scenario.Case(lib, cfg, [node])
    p2 = F.TEST_PRIORITY_CLASSES[F.PriorityClass2]["priority"]
    c.set_jobs(jobs, {"A": 0, "A-away": 1}, {i: (0, p2, i + 1) for i in range(5)})
    demand = np.zeros((2, scenario.R), dtype=np.int64)
    for j in home:
        demand[0] += np.array(scenario.vec(j["req"]), dtype=np.int64)
    c.sched.round_prepare([1.0, 1.0], [c.sort_queued(jobs, list(range(5, 10))), []], name_rank=[0, 1], demand=demand)
    res = c.sched.schedule_round()
    c.no_oversubscription()
    assert len(res.preempted) == preemptions and len(res.scheduled) == scheduled
    assert all(j < 5 for j in res.preempted), "only cross-pool jobs are preempted"
    assert all(j >= 5 for j in res.scheduled), "only home jobs are scheduled"


def test_oversubscribed_evictor_takes_the_cross_pool_jobs(lib):
    """preempting_queue_scheduler_test.go:168-220 TestEvictOversubscribed_CrossPoolJobsEvicted: 20 cross-pool and 20 home jobs (all priority 0) on a 32-cpu
    node.  Both debit the CrossPoolPriority bucket, which goes negative; only the home jobs debit priority 0, which does not: the oversubscribed evictor
    evicts exactly the cross-pool jobs.  The reference calls the evictor alone; here it runs inside a round (nothing is queued, protected fraction 1 keeps
    the balance evictor out of it): the 12 cross-pool jobs the node still has room for come back, the other 8 are the round's preemptions — no home job."""
    cfg = F.TestSchedulingConfig()
    cfg["preempt_cross_pool_jobs_first"] = True
    cfg["protected_fraction_of_fair_share"] = 100.0
    node = F.Test32CpuNode(F.TestPriorities)
    cross = [dict(j, away=True, queue="away-queue-away") for j in F.N1Cpu4GiJobs("away-queue", F.PriorityClass0, 20)]
    home = F.N1Cpu4GiJobs("home-queue", F.PriorityClass0, 20)
    jobs = cross + home
    c = scenario.Case(lib, cfg, [node])
    p0 = F.TEST_PRIORITY_CLASSES[F.PriorityClass0]["priority"]
    c.set_jobs(jobs, {"away-queue-away": 0, "home-queue": 1}, {i: (0, p0, i + 1) for i in range(40)})
    c.sched.round_prepare([1.0, 1.0], [[], []], name_rank=[0, 1])
    res = c.sched.schedule_round()
    assert len(res.scheduled) == 0
    assert len(res.preempted) == 8 and all(j < 20 for j in res.preempted)


@pytest.mark.parametrize("name,away,allocated,configured,expect", [
    ("nil floating resources", False, 5, False, False),
    ("gctx - within limits", False, 5, True, True),
    ("gctx - not within limits", False, 15, True, False),
    ("gctx - away - within limits", True, 5, True, True),
    ("gctx - away - not within limits", True, 15, True, True),
])
def test_is_within_floating_resource_limits(lib, name, away, allocated, configured, expect):
    """context/scheduling_test.go:529-592 TestIsWithinFloatingResourceLimits: a one-job gang that requests 1 of test-floating-resource (pool total 10) while the
    pool already holds `allocated` of it.  A home gang must stay within the pool's total; a gang from another pool is not checked here (the check ran on
    its own pool).  Without floating resources configured the request cannot be granted.  The reference calls the check with sctx.Allocated preset; here
    the allocation comes in as a queue's initial allocation and the gang goes through GangScheduler.Schedule, which adds it before it checks
    (gang_scheduler.go:100-143), so the limit is compared with allocated + 1 — same verdicts for 5 and 15 against 10."""
    cfg = F.TestSchedulingConfig()
    cfg["preempt_cross_pool_jobs_first"] = True
    cfg["floating_resources"] = {"test-floating-resource": {"pool": 10000 if configured else 0}}   # factory units: 10 whole units
    job = dict(F.Test1Cpu4GiJob("A", F.PriorityClass2), away=away)
    job["req"] = dict(job["req"], **{"test-floating-resource": 1000})
    if away:
        job["queue"] = "A-away"
    c = scenario.Case(lib, cfg, [F.Test32CpuNode(F.TestPriorities)])
    c.set_jobs([job], {"A": 0, "A-away": 1}, {})
    alloc = np.zeros((2, len(c.pc_names), scenario.R), dtype=np.int64)
    alloc[0, 0, scenario.RES.index("test-floating-resource")] = allocated * 1000
    c.sched.round_prepare([1.0, 1.0], [[], []], name_rank=[0, 1], demand=np.zeros((2, scenario.R), dtype=np.int64), allocated_by_pc=alloc)
    ok, reason, pods = c.sched.gang_schedule([0])
    assert ok == expect, (name, reason)
    if not ok:
        assert reason != 0


@pytest.mark.parametrize("home_context_exists,evicted", [(True, 0), (False, 0)])
def test_away_job_is_never_evicted_for_balancing(lib, home_context_exists, evicted):
    """pqs.go:101-104: the node evictor's job filter starts with `if job.LatestRun().Pool() != sch.schedulingContext.Pool { return false, "" }` — a cross-pool
    away job is never evicted in phase 1, whatever its queue's share and whether or not the pool has a context named after its queue (only urgency preemption
    and the oversubscribed evictor take it).  The 20 away jobs of queue "A-away" fill the node far above any fair share: none is evicted."""
    cfg = F.TestSchedulingConfig()
    cfg["preempt_cross_pool_jobs_first"] = True
    node = F.Test32CpuNode(F.TestPriorities)
    jobs = [dict(j, away=True, queue="A-away") for j in F.N1Cpu4GiJobs("A", F.PriorityClass0, 20)]
    c = scenario.Case(lib, cfg, [node])
    qidx = {"A": 0, "A-away": 1} if home_context_exists else {"A-away": 0}
    p0 = F.TEST_PRIORITY_CLASSES[F.PriorityClass0]["priority"]
    c.set_jobs(jobs, qidx, {i: (0, p0, i + 1) for i in range(20)})
    nq = len(qidx)
    c.sched.round_prepare([1.0] * nq, [[] for _ in range(nq)], name_rank=list(range(nq)))
    res = c.sched.schedule_round()
    assert res.num_evicted_phase1 == evicted
    assert len(res.preempted) == 0 and len(res.scheduled) == 0
