"""TestQueuedGangIterator* (queue_scheduler_test.go:1244-1431) as rounds (pass 1 of asched_schedule_round = QueueScheduler.Schedule over evicted + queued jobs).

The Go tests drive QueuedGangIterator by hand: Peek / Clear over a list of job contexts, some marked IsEvicted, with a lookback limit, OnlyYieldEvicted and
ResumeNonEvicted.  Over the ABI the iterator is the round's own (queue_scheduler.go:303-447): what it yields is observable as "which jobs were attempted"
— every job here fits, so attempted == scheduled (new jobs) / rescheduled, i.e. not preempted (evicted jobs).  (asched_schedule_queues, the
entry queue_scheduler_test.go's tables drive, takes queued jobs only; evicted streams exist in whole rounds.)  A queue's stream is MultiJobsIterator(evicted, queued) as in production
(preempting_queue_scheduler.go: evicted jobs first), so rows whose hand-made list puts a new job IN FRONT of an evicted one are restated with the evicted
jobs in front; what each row pins does not depend on that order:

  "iterates all jobs"                               every job of the stream is yielded once, evicted or not
  "max lookback"                                    after maxLookback new jobs the queue yields no further NEW job (:435-445)
  "max lookback - still returns all evicted jobs"   evicted jobs are not counted (jobsSeen, :393-396) and are all yielded
  "only yield evicted" / _TopItemEvicted            evicted-only mode (here: entered by the terminal "global scheduling rate limit exceeded" of the FIRST new
                                                    gang attempted, queue_scheduler.go:205-213): evicted jobs still come, of every queue
  _TopItemNonEvicted                                the new gang another queue had ALREADY peeked when the mode was entered is dropped, not attempted (:338-350, 521-544)
  _ResumeNonEvicted / _ResumeNonEvicted_NoEvicted   the scheduler's only caller of ResumeNonEvicted is the fair-share preemption budget (queue_scheduler.go:114-140):
                                                    tests/test_z_preemption_rate_limit.py holds TestQueueScheduler_PreemptionRateLimit's 7 rows; here the two
                                                    iterator rows — new jobs paused while evicted ones drain, then yielded; with no evicted job at all the
                                                    pause is empty and the new job still comes.
Run on the oracle, the CPU build of the device code and (-m gpu) the HIP library.
"""
import os
import sys

import numpy as np
import pytest

import scenario

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

GLOBAL_RATE_LIMIT = 3   # ASCHED_REASON_GLOBAL_RATE_LIMIT (terminal)


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


def run(lib, streams, weights=None, lookback=0, global_tokens=None, preemption_tokens=None):
    """streams: {queue: [evicted?, ...]} in stream order (evicted jobs first).  Every job is testfixtures.Test1Cpu4GiJob at PriorityClass0 on one 32-cpu node.
    The evicted ones (createJctx(true)) RUN on the node when the round starts: with protectedFractionOfFairShare 0 (TestSchedulingConfig) phase 1 evicts every
    one of them (pqs.go:120-134), and pass 1 is the QueueScheduler run over (evicted + queued) the Go tests build by hand — the construction of
    tests/test_z_preemption_rate_limit.py.  An evicted job the iterator yields goes back to its node; one it does not yield ends up in `preempted`.
    -> (result, {queue: [job index, ...]})"""
    cfg = F.TestSchedulingConfig()
    cfg["max_queue_lookback"] = lookback
    names = sorted(streams)
    jobs, idx = [], {}
    for q in names:
        assert streams[q] == sorted(streams[q], reverse=True), "evicted jobs lead a queue's stream (MultiJobsIterator(evicted, queued))"
        idx[q] = []
        for _ in streams[q]:
            idx[q].append(len(jobs)); jobs.append(F.Test1Cpu4GiJob(q, F.PriorityClass0))
    prio = cfg["priority_classes"][F.PriorityClass0]["priority"]
    running = {j: (0, prio, j + 1) for q in names for j, ev in zip(idx[q], streams[q]) if ev}
    c = scenario.Case(lib, cfg, [F.Test32CpuNode(F.TestPriorities)])
    c.set_jobs(jobs, {q: i for i, q in enumerate(names)}, running)
    s = c.sched
    queued = [[j for j, ev in zip(idx[q], streams[q]) if not ev] for q in names]
    kw = {}
    if global_tokens is not None:
        kw.update(global_tokens=float(global_tokens), global_burst=100, global_rate_inf=False)
    if preemption_tokens is not None:
        kw.update(fairshare_preemption_tokens=float(preemption_tokens))
    s.round_prepare(list(weights or [1.0] * len(names)), queued, name_rank=list(range(len(names))), demand=np.zeros((len(names), scenario.R), dtype=np.int64), **kw)
    res = s.schedule_round()
    assert res.num_evicted_phase1 == len(running)
    assert all(j not in running for j in res.scheduled), "ScheduledJobs holds new jobs only"
    c.no_oversubscription()
    return res, idx


def test_iterates_all_jobs(lib):
    res, idx = run(lib, {"A": [True, False, False]})                  # createJctx(false), (false), (true): all three are yielded
    assert set(res.scheduled) == set(idx["A"][1:]) and not res.preempted


def test_max_lookback(lib):
    res, idx = run(lib, {"A": [False] * 4}, lookback=2)               # expectReturnedIndexes {0, 1}
    assert set(res.scheduled) == set(idx["A"][:2])
    assert all(res.job_unschedulable_reason[j] == 0 for j in idx["A"][2:]), "jobs behind the lookback limit are never looked at"


def test_max_lookback_still_returns_all_evicted_jobs(lib):
    res, idx = run(lib, {"A": [True, True, False, False]}, lookback=1)   # expectReturnedIndexes {0, 2, 3}: one new job, both evicted ones
    a = idx["A"]
    assert set(res.scheduled) == {a[2]} and not res.preempted
    assert res.job_unschedulable_reason[a[3]] == 0


def test_only_yield_evicted(lib):
    """evicted-only mode for every queue: queue A (the heavier weight: its gang is cheapest, tried first) has only a new job, and no token is left for it — the
    terminal reason switches the round to evicted jobs only; queue B's evicted jobs are still rescheduled (_TopItemEvicted), its new job — which B had not even
    reached — is not"""
    res, idx = run(lib, {"A": [False], "B": [True, True, False]}, weights=[10.0, 1.0], global_tokens=0.0)
    assert not res.scheduled and not res.preempted
    assert res.termination_reason == GLOBAL_RATE_LIMIT and res.job_unschedulable_reason[idx["A"][0]] == GLOBAL_RATE_LIMIT
    assert res.job_unschedulable_reason[idx["B"][2]] == 0


def test_only_yield_evicted_drops_a_new_gang_that_was_already_peeked(lib):
    """_TopItemNonEvicted: queues A, B and C all have a NEW job at the head — the candidate iterator holds one peeked gang per queue.  A (cheapest) is tried and
    meets the global rate limit: the gangs of B and C, though already peeked, are dropped — neither is attempted, neither gets a reason"""
    res, idx = run(lib, {"A": [False], "B": [False], "C": [False]}, weights=[10.0, 2.0, 1.0], global_tokens=0.0)
    assert not res.scheduled and res.termination_reason == GLOBAL_RATE_LIMIT
    assert [int(res.job_unschedulable_reason[idx[q][0]]) for q in "ABC"] == [GLOBAL_RATE_LIMIT, 0, 0]


def test_resume_non_evicted(lib):
    """the fair-share preemption budget is spent from the start (0 tokens: AtFairsharePreemptionRateLimit, queue_scheduler.go:116): the new job is paused, the
    evicted one is yielded, the stream runs dry, ResumeNonEvicted brings the new job back — both end up scheduled"""
    res, idx = run(lib, {"A": [True, False]}, preemption_tokens=0.0)
    assert set(res.scheduled) == {idx["A"][1]} and not res.preempted


def test_resume_non_evicted_no_evicted(lib):
    res, idx = run(lib, {"A": [False]}, preemption_tokens=0.0)         # nothing to drain: the paused new job is yielded after the (empty) evicted-only phase
    assert set(res.scheduled) == set(idx["A"])
