"""NodeTypeJobRequirementsMet (nodematching.go:127-145) — all 13 cases of TestNodeTypeSchedulingRequirementsMet (nodematching_test.go:411-562), transcribed by hand.

A node TYPE matches a job on the INDEXED taints and labels alone: an untolerated taint that is not indexed does not exclude the type, a selector on a label that is
not indexed is left to the node-level check, an indexed label the node does not carry is an "unset" label and excludes the type.  One node per case in a NodeDb
whose indexedTaints / indexedNodeLabels are the case's (`IndexedTaints: nil` = every taint indexed, `IndexedLabels: nil` = none); asched_node_types_matching_job
is NodeTypesMatchingJob (nodedb.go:1118-1133) over the one populated node type.  Oracle, CPU build of the product's host code, HIP library (-m gpu).
"""
import copy
import os
import sys

import pytest

import scenario

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gofixtures as F  # noqa: E402

FOO_TAINT = [["foo", "foo", "NoSchedule"]]
FOO_TOL = [{"key": "foo", "op": "Equal", "value": "foo", "effect": ""}]
BAR_SEL = {"bar": "bar"}
ALL = None   # IndexedTaints: nil

# name, taints, labels, indexed taints (None = all), indexed labels (None = none), tolerations, selector, expect
CASES = [
    ("nil taints and labels", [], {}, ALL, None, FOO_TOL, BAR_SEL, True),                                             # :420-428
    ("no taints or labels", [], {}, ALL, None, FOO_TOL, BAR_SEL, True),                                               # :429-437
    ("tolerated taints", FOO_TAINT, {}, ALL, None, FOO_TOL, {}, True),                                                # :438-445
    ("untolerated taints", FOO_TAINT, {}, ALL, None, [], {}, False),                                                  # :446-451
    ("untolerated non-indexed taint", FOO_TAINT, {}, [], None, [], {}, True),                                         # :452-458
    ("matched node selector", [], {"bar": "bar"}, ALL, ["bar"], [], BAR_SEL, True),                                   # :459-467
    ("unset indexed label", [], {}, ALL, ["bar"], FOO_TOL, BAR_SEL, False),                                           # :468-477
    ("different label value", [], {"bar": "baz"}, ALL, ["bar"], [], BAR_SEL, False),                                  # :478-486
    ("missing label", [], {}, ALL, None, [], BAR_SEL, True),                                                          # :487-494
    ("tolerated taints and matched node selector", FOO_TAINT, {"bar": "bar"}, ALL, ["bar"], FOO_TOL, BAR_SEL, True),  # :495-504
    ("untolerated taints and matched node selector", FOO_TAINT, {"bar": "bar"}, ALL, ["bar"], [], BAR_SEL, False),    # :505-513
    ("tolerated taints and different label value", FOO_TAINT, {"bar": "baz"}, ALL, ["bar"], FOO_TOL, BAR_SEL, False), # :514-523
    ("tolerated taints and missing label", FOO_TAINT, {}, ALL, ["bar"], FOO_TOL, BAR_SEL, False),                     # :524-533
]


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue({"oracle": "oracle_lib", "hostsim": "hostsim_lib", "hip": "hip_lib"}[request.param])


@pytest.mark.parametrize("name,taints,labels,itaints,ilabels,tolerations,selector,expect", CASES, ids=[c[0] for c in CASES])
def test_node_type_job_requirements_met(lib, name, taints, labels, itaints, ilabels, tolerations, selector, expect):
    cfg = F.TestSchedulingConfig()
    cfg["indexed_taints"] = sorted({t[0] for t in taints}) if itaints is ALL else list(itaints)
    cfg["indexed_node_labels"] = list(ilabels or [])
    node = F.Test32CpuNode(F.TestPriorities)
    node["taints"], node["labels"] = copy.deepcopy(taints), dict(labels)
    job = F.Test1Cpu4GiJob("A", F.PriorityClass0)
    job["tolerations"], job["selector"] = copy.deepcopy(tolerations), dict(selector)
    c = scenario.Case(lib, cfg, [node])
    c.set_jobs([job], {"A": 0}, {})
    matching, excluded = c.sched.node_types_matching_job(0)
    assert (matching, excluded) == ((1, 0) if expect else (0, 1)), name
