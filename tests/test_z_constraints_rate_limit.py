"""CheckJobConstraints' token-bucket rules (constraints.go:136-157) at function level.

tests/golden/constraints_rate_limit_cases.json: the 8 cases of TestCheckJobConstraints_RateLimit (constraints_test.go:452-575) — tokens held
by the global and the queue limiter at sctx.Started, their bursts, a gang cardinality, and the reason CheckJobConstraints returns.  Driven
through asched_gang_schedule (GangScheduler.Schedule -> CheckJobConstraints, gang_scheduler.go:100-148) on a pool where the gang would fit.
"""
import numpy as np
import pytest

import scenario
from golden_io import ids, load

CASES = load("constraints_rate_limit")
CFG = load("nodedb_conditional_away")[0]["SchedulingConfig"]
REASON = {  # constraints.go:34-49 -> ASCHED_REASON_*
    "": 0, "global scheduling rate limit exceeded": 3, "queue scheduling rate limit exceeded": 4,
    "gang would exceed global scheduling rate limit": 6, "gang would exceed queue scheduling rate limit": 7,
    "gang cardinality too large: exceeds global max burst size": 8, "gang cardinality too large: exceeds queue max burst size": 9,
}
GI = 2**30


def run_case(lib, case):
    n = int(case["cardinality"])
    nodes = [{"index": 1, "total": {"cpu": 1000 * 1000, "memory": 1000 * GI}, "taints": [], "labels": {}, "used": {}, "unschedulable": False}]
    gang = {"id": "g", "cardinality": n, "uniformity": ""} if n > 1 else None
    jobs = [{"created": i + 1, "queue": "queue-1", "pc": "priority-1", "priority": 1000, "gang": gang, "tolerations": [], "selector": {}, "affinity": None,
             "req": {"cpu": 1000, "memory": GI}} for i in range(n)]
    c = scenario.Case(lib, CFG, nodes)
    c.set_jobs(jobs, {"queue-1": 0}, {})
    npc = len(c.pc_names)
    c.sched.round_prepare([1.0], [[]], demand=np.zeros((1, scenario.R), dtype=np.int64), allocated_by_pc=np.zeros((1, npc, scenario.R), dtype=np.int64),
                          global_tokens=float(case["globalTokens"]), global_burst=int(case["globalBurst"]), global_rate_inf=False,
                          queue_tokens=[float(case["queueTokens"])], queue_burst=[int(case["queueBurst"])], queue_rate_inf=[False])
    ok, reason, pods = c.sched.gang_schedule(list(range(n)))
    exp = REASON[case.get("expectedReason") or ""]
    assert ok == (exp == 0) and reason == exp, (ok, reason, exp)
    assert all((p.node >= 0) == ok for p in pods)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_oracle(oracle_lib, case):
    run_case(oracle_lib, case)


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_hostsim(hostsim_lib, case):
    run_case(hostsim_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_gpu(hip_lib, case):
    run_case(hip_lib, case)
