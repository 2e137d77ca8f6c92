"""NumNodes / TotalKubernetesResources / NodeTypesMatchingJob (nodedb.go:345-351, 1118-1133).

TestTotalResources (nodedb_test.go:59-94) restated: an empty NodeDb reports an all-zero (not empty) total; nodes add their resources.
(nodes_upsert replaces the node set, so the test's two insert steps are two upserts of the cumulative lists.)  NodeTypesMatchingJob is
checked against the node types the fixtures create: indexed taints / labels only (node_type.go:68-131)."""
import os
import sys

import numpy as np
import pytest

import scenario

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


def test_total_resources(lib):
    cfg = F.TestSchedulingConfig()
    c = scenario.Case(lib, cfg, [])
    s = c.sched
    assert (s.total_resources() == 0).all() and s.lib.num_nodes(s.h) == 0            # :63-64
    nodes = F.N32CpuNodes(2, F.TestPriorities)
    expected = np.sum([scenario.vec(n["total"]) for n in nodes], axis=0)
    c.nodes = nodes; c.upsert_nodes(set())
    assert (s.total_resources() == expected).all() and s.lib.num_nodes(s.h) == 2     # :79
    nodes = nodes + F.N8GpuNodes(3, F.TestPriorities)
    expected = np.sum([scenario.vec(n["total"]) for n in nodes], axis=0)
    c.nodes = nodes; c.upsert_nodes(set())
    assert (s.total_resources() == expected).all() and s.lib.num_nodes(s.h) == 5     # :93


def test_node_types_matching_job(lib):
    cfg = F.TestSchedulingConfig()
    nodes = F.N32CpuNodes(3, F.TestPriorities) + F.NTainted32CpuNodes(2, F.TestPriorities) + F.N8GpuNodes(4, F.TestPriorities)
    # node types: plain (3 nodes), largeJobsOnly-tainted + labelled (2), gpu-labelled (4): taints and labels on the indexed lists only
    jobs = [F.Test1Cpu4GiJob("A", F.PriorityClass0),                                 # no tolerations: every type without an indexed taint
            F.Test32Cpu256GiJobWithLargeJobToleration("A", F.PriorityClass0),        # tolerates largeJobsOnly: all three types
            F.WithNodeSelectorJobs({"gpu": "true"}, [F.Test1Cpu4GiJob("A", F.PriorityClass0)])[0],            # indexed label: only the gpu type
            F.WithNodeSelectorJobs({"largeJobsOnly": "true"}, [F.Test1Cpu4GiJob("A", F.PriorityClass0)])[0]]  # label matches, taint not tolerated
    c = scenario.Case(lib, cfg, nodes)
    c.set_jobs(jobs, {"A": 0}, {})
    got = [c.sched.node_types_matching_job(j) for j in range(4)]
    assert got == [(2, 2), (3, 0), (1, 5), (0, 9)], got
