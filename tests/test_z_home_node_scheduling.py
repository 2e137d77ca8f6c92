"""nodedb_test.go:800-928 TestHomeNodeScheduling, restated through the C ABI (oracle, CPU build of the device code, HIP library).

One 32-cpu node (priorities 29000 / 30000), optionally tainted largeJobsOnly=true:NoSchedule; one 16-cpu job of `armada-preemptible` (or of
`armada-preemptible-away`); home scheduling optionally disabled; optionally a pool-level DEFAULT TOLERATION (SchedulingOptions.DefaultTolerations,
nodedb.go:383-393: appended to the job's AdditionalTolerations for every node selection, :561-562).  Default tolerations apply to every job of the
pool in every selection, so the caller folds them into each requirement class when it interns the classes (INTEGRATION.md "What the shim interns
into the tables"): the job dict below carries the default toleration exactly like that.
"""
import os
import sys

import pytest

import scenario

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

METHOD_NO_PREEMPTION, METHOD_AWAY = 2, 5

CASES = {   # name: (disableHomeScheduling, defaultToleration key or None, addNodeTaint, addJobToleration, priority class or None, expectSuccess, expectScheduledAway)
    "should schedule home jobs": (False, None, False, False, None, True, False),
    "should schedule home jobs - when default toleration matches node taint": (False, "largeJobsOnly", True, False, None, True, False),
    "should schedule home jobs - when default toleration does not match node taint": (False, "non-matching", True, False, None, False, False),
    "should schedule home jobs - when job toleration matches node taint": (False, None, True, True, None, True, False),
    "should not schedule home jobs - when job toleration does not match node taint": (False, None, True, False, None, False, False),
    "should not schedule home job - when home scheduling disabled": (True, None, False, False, None, False, False),
    "should schedule away job - when home scheduling disabled - away scheduling available": (True, None, False, False, "armada-preemptible-away", True, True),
}


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


@pytest.mark.parametrize("name", sorted(CASES))
def test_home_node_scheduling(lib, name):
    disable_home, default_tol, taint, job_tol, pc, expect_ok, expect_away = CASES[name]
    cfg = F.TestSchedulingConfig()
    cfg["disable_home"] = disable_home
    node = F.Test32CpuNode([29000, 30000])
    if taint:
        node = F.AddTaints([node], [{"Key": "largeJobsOnly", "Value": "true", "Effect": "NoSchedule"}])[0]
    pc = pc or F.PriorityClass6Preemptible
    job = (F.Test16Cpu128GiJobWithLargeJobToleration if job_tol else F.Test16Cpu128GiJob)(F.make_env()["testfixtures.TestQueue"], pc)
    if default_tol:   # v1.Toleration{Key: k, Operator: Exists, Effect: NoSchedule} for every job of the pool
        job["tolerations"] = list(job["tolerations"]) + [{"key": default_tol, "op": "Exists", "value": "", "effect": "NoSchedule"}]
    c = scenario.Case(lib, cfg, [node])
    c.set_jobs([job], {job["queue"]: 0}, {})
    s = c.sched
    s.txn_begin()
    ok, pods, _ = s.schedule_many([0])
    assert ok == expect_ok, name
    if ok:
        s.txn_commit()
        assert pods[0].node == 0
        if expect_away:
            assert pods[0].method == METHOD_AWAY and pods[0].scheduled_at_priority == 29000
        else:
            assert pods[0].method == METHOD_NO_PREEMPTION and pods[0].scheduled_at_priority == 30000
    else:
        s.txn_abort()
        assert pods[0].node < 0
