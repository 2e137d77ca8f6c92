"""sctx.FairsharePreemptionLimiter (context/scheduling.go:508-528; queue_scheduler.go:114-121): once the pool's fair-share preemption
budget is spent, only evicted jobs are yielded.

tests/golden/preemption_rate_limit_cases.json holds the 7 cases of TestQueueScheduler_PreemptionRateLimit (queue_scheduler_test.go:
804-946), table evaluated mechanically.  The reference test evicts every existing job by hand and runs QueueScheduler over
(evicted + new) iterators; the same situation is produced here by a whole round: with protectedFractionOfFairShare = 0 (TestSchedulingConfig)
phase 1 evicts every preemptible job of a queue that holds anything (pqs.go:120-134), pass 1 then is that QueueScheduler run, and the
later phases have nothing to do.  Queue demand is passed as zero like newQueueSchedulerSctx does (:718-738).  Expectations: the number
of new jobs scheduled and the number of existing jobs preempted.
"""
import numpy as np
import pytest

import scenario
from golden_io import ids, load

CASES = load("preemption_rate_limit")


def run_case(lib, case):
    cfg = case["SchedulingConfig"]
    c = scenario.Case(lib, cfg, [case["node"]])
    pcs = cfg["priority_classes"]
    existing = list(case.get("existingJobsQueueA") or []) + list(case.get("existingJobsQueueB") or [])
    new = list(case["newJobsQueueB"])
    jobs = existing + new
    running = {i: (0, pcs[j["pc"]]["priority"], i + 1) for i, j in enumerate(existing)}
    c.set_jobs(jobs, {"A": 0, "B": 1}, running)
    queued = [[], c.sort_queued(jobs, list(range(len(existing), len(jobs))))]
    lim = case.get("rateLimit")
    c.sched.round_prepare([1.0, 1.0], queued, name_rank=[0, 1], demand=np.zeros((2, scenario.R), dtype=np.int64),
                          fairshare_preemption_tokens=None if lim is None else float(lim["burst"]))   # a fresh limiter holds `burst` tokens
    res = c.sched.schedule_round()
    assert all(j >= len(existing) for j in res.scheduled), "ScheduledJobs holds new jobs only"
    assert all(j < len(existing) for j in res.preempted)
    assert len(res.scheduled) == case["expectedNumberNewJobsScheduled"], (len(res.scheduled), len(res.preempted))
    assert len(res.preempted) == case["expectedNumberPreemptedJobs"], (len(res.scheduled), len(res.preempted))
    c.no_oversubscription()
    return res


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_oracle(oracle_lib, case):
    run_case(oracle_lib, case)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_hostsim(hostsim_lib, oracle_lib, case):
    scenario.assert_same_round(run_case(oracle_lib, case), run_case(hostsim_lib, case))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_gpu(hip_lib, oracle_lib, case):
    scenario.assert_same_round(run_case(oracle_lib, case), run_case(hip_lib, case))
