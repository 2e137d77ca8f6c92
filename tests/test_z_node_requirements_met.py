"""JobRequirementsMet (nodematching.go:146-160): static requirements + resources at one priority, at function level.

tests/golden/node_requirements_met_cases.json: all 18 cases of TestNodeSchedulingRequirementsMet (nodematching_test.go:19-410), table
evaluated mechanically — taints vs tolerations, node selector, required node affinity (In on present / absent labels), total and
per-priority resources.  One node per case; asched_fit_select_batch(job, priority) is selectNodeForPodAtPriority on that NodeDb, i.e.
JobRequirementsMet for its only node (the HIP library answers it with k_fit_batch over the host-built static masks).
"""
import numpy as np
import pytest

import scenario
from golden_io import ids, load

CASES = load("node_requirements_met")


def run_case(lib, case):
    n = case["node"]
    node = {"index": 1, "total": n["total"], "taints": n["taints"], "labels": n["labels"], "used": {}, "unschedulable": False}
    c = scenario.Case(lib, case["SchedulingConfig"], [node])
    s = c.sched
    if n.get("alloc_by_priority") is not None:   # makeTestNodeResources: explicit AllocatableByPriority, absent levels read as empty
        total = np.array([scenario.vec(n["total"])], dtype=np.int64)
        abp = np.zeros((1, s.P, scenario.R), dtype=np.int64)
        for l, prio in enumerate(s.priorities):
            abp[0, l] = scenario.vec(n["alloc_by_priority"].get(str(prio), {}))
        s.nodes_upsert(total, total, index=[1], alloc_by_prio=abp, taints=[[]], labels=[[]])
    c.set_jobs([case["job"]], {"A": 0}, {})
    got = int(s.fit_select_batch([0], case["priority"])[0])
    assert (got == 0) == case["expectSuccess"], (got, case["expectSuccess"])


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
