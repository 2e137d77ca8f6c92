"""Pins the CPU oracle against the reference's own test expectations.

Every case is a transcription (tests/golden/extract_reference_goldens.py) of a table entry in
internal/scheduler/**/*_test.go; the `source` field of each case gives file:line.
"""
import numpy as np
import pytest

import scenario
from golden_io import ids, load

PQS = load("pqs")
QS = load("queue_scheduler")
GANG = load("gang_scheduler")


MARKET = load("market_pqs")


@pytest.mark.parametrize("case", MARKET, ids=ids(MARKET))
def test_market_pqs_goldens(oracle_lib, case):
    """TestMarketDrivenPreemptingQueueScheduler (market_driven_preempting_queue_scheduler_test.go): whole market-driven rounds incl. spot price and billing"""
    assert scenario.run_pqs_case(oracle_lib, case) == "ok"


@pytest.mark.parametrize("case", PQS, ids=ids(PQS))
def test_preempting_queue_scheduler(oracle_lib, case):
    r = scenario.run_pqs_case(oracle_lib, case)
    if r != "ok":
        pytest.skip(r)


@pytest.mark.parametrize("case", QS, ids=ids(QS))
def test_queue_scheduler(oracle_lib, case):
    r = scenario.run_qs_case(oracle_lib, case)
    if r != "ok":
        pytest.skip(r)


@pytest.mark.parametrize("case", GANG, ids=ids(GANG))
def test_gang_scheduler(oracle_lib, case):
    r = scenario.run_gang_case(oracle_lib, case)
    if r != "ok":
        pytest.skip(r)


NTI = load("node_type_iterator")
NTSI = load("node_types_iterator")


@pytest.mark.parametrize("case", NTI + NTSI, ids=["one:" + n for n in ids(NTI)] + ["merged:" + n for n in ids(NTSI)])
def test_node_iteration_order(oracle_lib, case):
    got = scenario.run_node_iteration_case(oracle_lib, case)
    assert got == case["expected"], case["source"]


FAIR = load("fairness")


@pytest.mark.parametrize("case", FAIR, ids=ids(FAIR))
def test_drf_cost_exact(oracle_lib, case):
    """fairness_test.go:62-177 — assert.Equal on float64, i.e. bit-exact."""
    got = scenario.run_fairness_case(oracle_lib, case)
    if got is None:
        pytest.skip("pool-override config")
    assert got == case["expectedCost"], (got, case["expectedCost"], case["source"])


SHARES = load("fair_shares")


@pytest.mark.parametrize("case", SHARES, ids=ids(SHARES))
def test_fair_shares_exact(oracle_lib, case):
    """context/scheduling_test.go:89-247 — bit-exact float64 fair shares."""
    got = scenario.run_fair_share_case(oracle_lib, case)
    for q in case["queueCtxs"]:
        assert got[q][0] == case["expectedFairShares"][q], (q, got[q], case["source"])
        assert got[q][1] == case["expectedDemandCappedAdjustedFairShares"][q], (q, got[q], case["source"])
        assert got[q][2] == case["expectedUncappedAdjustedFairShares"][q], (q, got[q], case["source"])

NODEDB = load("nodedb_schedule_individually") + load("nodedb_schedule_many") + load("nodedb_away_node_scheduling")


@pytest.mark.parametrize("case", NODEDB, ids=[c["source"].split("/")[-1] + ":" + c["name"] for c in NODEDB])
def test_nodedb_schedule_many_with_txn(oracle_lib, case):
    """nodedb_test.go TestScheduleIndividually / TestScheduleMany through the NodeDb-level entry points (txn_begin, schedule_many, commit / abort)"""
    r = scenario.run_nodedb_schedule_case(oracle_lib, case)
    if r != "ok":
        pytest.skip(r)
