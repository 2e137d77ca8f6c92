"""Hard timeout / cancel and the soft time budgets of the scheduling round.

Reference: queue_scheduler_test.go:1482-1620 TestQueueSchedulerTimeouts (three cases: "global soft timeout exceeded", "queue soft timeout
exceeded", "global hard timeout: returns error"); queue_scheduler.go:105-112 (ctx.Done() at the top of every loop iteration, the round returns
ctx.Err()), :157 / :222-228 (time spent per gang of new jobs), constraints.go:159-169 (budget checks), scheduling_algo.go:130-134, 262-270
(maxSchedulingDuration wraps the context; nothing of a failed round is applied).

The Go table marks 50 jobs per queue `IsEvicted` and offers them before 100 new ones; here the same streams come out of a whole round: the 50
jobs of a queue run on the nodes, phase 1 evicts all of them (protectedFractionOfFairShare 0), and pass 1 sees [50 evicted, 100 queued] per
queue with every queue allocation starting from zero — the situation the Go test constructs by hand.  Expectations are the table's.
"""
import numpy as np
import pytest

import scenario
from armada_amd import workloads as W
from armada_amd.binding import ERR_TIMEOUT, Config, SchedError, Scheduler

Gi = 1024 ** 3


def _timeout_workload(global_ns=0, queue_ns=0, step_ns=0):
    cfg = Config(num_resources=4, indexed_col=[W.CPU, W.MEM, W.GPU], indexed_resolution=[1000, 128 * W.Mi, 1],
                 pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0, 1.0, 0.0, 1.0], prefer_large_job_ordering=True,
                 protected_fraction_of_fair_share=0.0, max_new_job_scheduling_duration_ns=global_ns,
                 max_new_job_scheduling_duration_per_queue_ns=queue_ns, clock_step_ns=step_ns)
    n_nodes = 10                                                    # testfixtures.N32CpuNodes(10, ...)
    node_total = np.tile(np.array([256 * Gi, 32000, 0, 0], dtype=np.int64), (n_nodes, 1))
    one = np.array([4 * Gi, 1000, 0, 0], dtype=np.int64)           # N1Cpu4GiJobs
    req, queue, node = [], [], []
    for q in (0, 1):                                                # queue A (weight 100), queue B (weight 1)
        for i in range(50):                                         # the 50 "evicted" jobs of the queue: running, evicted by phase 1
            req.append(one); queue.append(q); node.append((q * 50 + i) % n_nodes)
    for q in (0, 1):
        for i in range(100):                                        # 100 new jobs
            req.append(one); queue.append(q); node.append(-1)
    m = len(req)
    queued = [np.array([j for j in range(100, m) if queue[j] == q], dtype=np.int32) for q in (0, 1)]
    return cfg, node_total, np.array(req), np.array(queue, np.int32), np.array(node, np.int32), queued


def _load(lib, wl):
    cfg, node_total, req, queue, node, queued = wl
    s = Scheduler(lib, cfg)
    s.nodes_upsert(node_total)
    m = len(req)
    s.jobs_set(req, queue=queue, pc=np.zeros(m, np.int32), submit_time=np.arange(m), node=node, scheduled_at_priority=np.zeros(m, np.int32), run_timestamp=np.arange(m))
    return s, queued


def _round(s, queued):
    s.round_prepare([100.0, 1.0], queued)                       # setupTimeoutTest passes queue.PriorityFactor as the WEIGHT (AddQueueSchedulingContext(name, weight, rawWeight, ...), queue_scheduler_test.go:1202-1206)
    return s.schedule_round()


def _counts(res, queue):
    new = {0: 0, 1: 0}
    for j in res.scheduled:
        new[int(queue[j])] += 1
    return new


def _libs(request):
    return [request.getfixturevalue("oracle_lib"), request.getfixturevalue("hostsim_lib")]


CASES = {
    # name: (global soft timeout, queue soft timeout, clock step, expected new jobs per queue)
    "global soft timeout exceeded": (1, 0, 10_000_000_000, {0: 1, 1: 0}),   # SteppingClock(10 s): the first new gang already exceeds 1 ns
    "queue soft timeout exceeded": (0, 1, 0, {0: 1, 1: 1}),                 # RealClock in the Go test too: any gang attempt takes longer than 1 ns
}


def _check_soft(lib, name):
    g, q, step, want = CASES[name]
    wl = _timeout_workload(g, q, step)
    s, queued = _load(lib, wl)
    res = _round(s, queued)
    assert res.num_evicted_phase1 == 100 and not res.preempted, "the 50 + 50 evicted jobs are all rescheduled"   # expectedEvictedJobsScheduled
    assert _counts(res, wl[3]) == want                                                                         # expectedNewJobsScheduled
    assert res.termination_reason == (18 if g else 16)   # global budget: terminal reason; queue budget: queue-terminal, the loop runs dry
    reasons = set(res.job_unschedulable_reason[100:].tolist())
    assert (18 if g else 19) in reasons
    s.close()
    return res


@pytest.mark.parametrize("name", list(CASES))
def test_soft_timeouts_oracle_and_cpu_build(oracle_lib, hostsim_lib, name):
    a = _check_soft(oracle_lib, name)
    b = _check_soft(hostsim_lib, name)
    scenario.assert_same_round(a, b)


def _check_hard(lib):
    """"global hard timeout: returns error": a context that is already done -> the round returns an error and no result; the handle then
    behaves like a fresh one"""
    wl = _timeout_workload()
    s, queued = _load(lib, wl)
    want = _round(s, queued)
    s.cancel()                                   # ctx, cancel := WithCancel(ctx); cancel()
    s.round_prepare([100.0, 1.0], queued)
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    assert e.value.code == ERR_TIMEOUT
    with pytest.raises(SchedError):              # nothing of the failed round is usable: the handle wants a round_prepare
        s.schedule_round()
    got = _round(s, queued)                      # ... after which the round is the one a fresh handle computes
    scenario.assert_same_round(want, got)
    s.set_deadline(1e-9)                         # maxSchedulingDuration already expired when the round starts
    s.round_prepare([100.0, 1.0], queued)
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    assert e.value.code == ERR_TIMEOUT
    s.set_deadline(0)
    scenario.assert_same_round(want, _round(s, queued))
    s.close()


def test_hard_timeout_oracle(oracle_lib):
    _check_hard(oracle_lib)


def test_hard_timeout_cpu_build(hostsim_lib):
    _check_hard(hostsim_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_soft_timeouts_gpu(hip_lib, oracle_lib, name):
    a = _check_soft(oracle_lib, name)
    b = _check_soft(hip_lib, name)
    scenario.assert_same_round(a, b)


@pytest.mark.gpu
def test_hard_timeout_gpu(hip_lib):
    wl = _timeout_workload()
    s, queued = _load(hip_lib, wl)
    want = _round(s, queued)
    s.cancel()
    s.round_prepare([100.0, 1.0], queued)
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    assert e.value.code == ERR_TIMEOUT
    scenario.assert_same_round(want, _round(s, queued))
    s.close()


@pytest.mark.gpu
def test_deadline_cancels_a_long_round_in_flight(hip_lib, oracle_lib):
    """maxSchedulingDuration on a round that takes seconds (BASELINE configs[4]'s shape, generic preemption path): the persistent kernel
    sees the host-mapped cancel word, every helper workgroup leaves, the call returns ASCHED_ERR_TIMEOUT long before the round would have
    finished, and the next round on the handle is identical to the oracle's"""
    import time
    wl = W.config3(n_nodes=20_000, n_jobs=200_000, n_queues=32, seed=W.SEED, occupied=0.95)
    wl.global_burst, wl.queue_burst = 40_000, 4_000
    s = W.load(hip_lib, wl)
    W.prepare(s, wl)
    s.set_deadline(0.25)
    t0 = time.perf_counter()
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    dt = time.perf_counter() - t0
    assert e.value.code == ERR_TIMEOUT and dt < 2.0, dt
    s.set_deadline(0)
    W.prepare(s, wl)
    got = s.schedule_round()
    o = W.load(oracle_lib, wl)
    W.prepare(o, wl)
    scenario.assert_same_round(o.schedule_round(), got)


@pytest.mark.gpu
def test_deadline_cancels_a_fast_path_round_in_flight(hip_lib, oracle_lib):
    """the same on a round that runs almost entirely on the fast path (BASELINE configs[2]'s shape): there the node engine wave polls the
    cancel word, answers "no node", and the control wave leaves the fast loop through its ordinary roll-back"""
    import time
    wl = W.config3(n_nodes=50_000, n_jobs=500_000, n_queues=64, seed=W.SEED)
    wl.global_burst, wl.queue_burst = 100_000, 10_000
    s = W.load(hip_lib, wl)
    W.prepare(s, wl)
    t0 = time.perf_counter(); full = s.schedule_round(); t_full = time.perf_counter() - t0
    W.prepare(s, wl)
    s.set_deadline(t_full / 4)
    t0 = time.perf_counter()
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    dt = time.perf_counter() - t0
    assert e.value.code == ERR_TIMEOUT and dt < 0.75 * t_full, (dt, t_full)
    s.set_deadline(0)
    W.prepare(s, wl)
    scenario.assert_same_round(full, s.schedule_round())


@pytest.mark.gpu
def test_deadline_ends_a_round_whose_node_engine_is_stuck(hip_lib, monkeypatch):
    """Round 6 ("bounded waits", armada_sched.hip): ASCHED_DEBUG_HANG=<n> makes the cold-set wave of a bulk-merged stream run stop answering at its n-th command (the node engine then waits for it) — what a protocol
    defect between the waves of the control workgroup looks like from outside (profiles/r05y_bulk_skip_hang.txt was one).  Every spin on the other side (the control wave's
    ring waits, the bind wave, the cold-set wave, the engine's own waits) counts its turns and looks at the caller's cancel word: with a deadline set the call returns
    ASCHED_ERR_TIMEOUT instead of hanging, and after a fresh round_prepare the handle computes the round a healthy launch computes."""
    import time
    wl = W.config3(n_nodes=50_000, n_jobs=500_000, n_queues=64, seed=W.SEED)
    wl.global_burst, wl.queue_burst = 100_000, 10_000
    s = W.load(hip_lib, wl)
    W.prepare(s, wl)
    full = s.schedule_round()
    assert s.round_stats()["stream_jobs"] > 50_000
    monkeypatch.setenv("ASCHED_DEBUG_HANG", "500")
    W.prepare(s, wl)
    s.set_deadline(0.3)
    t0 = time.perf_counter()
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    dt = time.perf_counter() - t0
    assert e.value.code == ERR_TIMEOUT and dt < 5.0, (e.value.code, dt)
    monkeypatch.delenv("ASCHED_DEBUG_HANG")
    s.set_deadline(0)
    W.prepare(s, wl)
    scenario.assert_same_round(full, s.schedule_round())
    s.close()
