"""The device control code (armada_amd/csrc/round_ctl.h, round_run.h) compiled for the CPU (tests/hostsim) must agree
with the oracle: this debugs the *logic* of the kernels in a GPU-less container.  The parallel primitives themselves
(scan / compaction / atomics) are only exercised by the `-m gpu` tests."""
import os
import subprocess

import numpy as np
import pytest

import scenario
from armada_amd import workloads as W
from armada_amd.binding import Library, SchedError
from golden_io import ids, load

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(fn, lib, case):
    try:
        r = fn(lib, case)
    except SchedError as e:
        if e.code == -2:
            pytest.skip(str(e))
        raise
    assert r == "ok", f"case not run: {r}"   # a driver that declines a reference case is a failure, not a skip


PQS, QS, GANG = load("pqs"), load("queue_scheduler"), load("gang_scheduler")


@pytest.mark.parametrize("case", PQS, ids=ids(PQS))
def test_pqs_goldens(hostsim_lib, case):
    _run(scenario.run_pqs_case, hostsim_lib, case)


MARKET = load("market_pqs")


@pytest.mark.parametrize("case", MARKET, ids=ids(MARKET))
def test_market_pqs_goldens(hostsim_lib, case):
    """TestMarketDrivenPreemptingQueueScheduler: whole market-driven rounds (evict-everything node evictor, price-ordered iterators, spot price, second-price billing)"""
    _run(scenario.run_pqs_case, hostsim_lib, case)


@pytest.mark.parametrize("case", QS, ids=ids(QS))
def test_qs_goldens(hostsim_lib, case):
    _run(scenario.run_qs_case, hostsim_lib, case)


NTI, NTSI = load("node_type_iterator"), load("node_types_iterator")


@pytest.mark.parametrize("case", NTI + NTSI, ids=["one:" + n for n in ids(NTI)] + ["merged:" + n for n in ids(NTSI)])
def test_node_iteration_order(hostsim_lib, case):
    """nodeiteration_test.go:86-307, 389-635: the order NodeTypeIterator / NodeTypesIterator yield nodes in, through the device's own literal iterator
    restatement (the code rounds use off the index grid), materialised by asched_iterate_nodes"""
    got = scenario.run_node_iteration_case(hostsim_lib, case)
    assert got == case["expected"], case["source"]


def test_reference_tables_reach_the_fast_path(hostsim_lib):
    """The reference's own PQS / QueueScheduler tables must pin the code that produces the bench numbers, not only the generic path: their nodes are uploaded
    without an explicit AllocatableByPriority (scenario.Case.upsert_nodes), so the rounds go through the level-0 fast structure, the stream runs and the
    preempting fast iteration.  asched_round_stats, summed over every round of both tables."""
    scenario.GOLDEN_STATS.clear()
    for fn, cases in ((scenario.run_pqs_case, PQS), (scenario.run_qs_case, QS)):
        for case in cases:
            assert fn(hostsim_lib, case) == "ok", case.get("name")
    st = dict(scenario.GOLDEN_STATS)
    assert st["rounds_with_fast_iterations"] * 2 > st["rounds"], st
    assert st["fast_iterations"] > 2 * st["generic_iterations"] and st["stream_runs"] > 20 and st["stream_jobs"] > 500, st
    assert st["preempt_fast_iterations"] > 100 and st["l0_overflows"] == 0, st


@pytest.mark.parametrize("case", GANG, ids=ids(GANG))
def test_gang_goldens(hostsim_lib, case):
    _run(scenario.run_gang_case, hostsim_lib, case)


@pytest.mark.parametrize("seed", range(12))
def test_random_rounds_match_oracle(hostsim_lib, oracle_lib, seed):
    wl = W.small_random(n_nodes=8 + seed * 9, n_jobs=300 + seed * 50, n_queues=2 + seed % 6, seed=seed, occupied=[0.3, 0.6, 0.9, 1.0][seed % 4],
                        gangs=seed % 5, burst=None if seed % 3 else (150 + seed * 5, 60 + seed))
    res = []
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl)
        W.prepare(s, wl)
        res.append(s.schedule_round())
    scenario.assert_same_round(res[0], res[1])

@pytest.mark.parametrize("seed", range(8))
def test_random_rounds_with_away_node_types_match_oracle(hostsim_lib, oracle_lib, seed):
    """away node types (nodedb.go:613-627, 677-722): tainted nodes reachable only through a priority class's AwayNodeTypes"""
    wl = W.small_random(n_nodes=16 + seed * 7, n_jobs=300 + seed * 40, n_queues=2 + seed % 4, seed=100 + seed, occupied=[0.5, 0.9, 1.0][seed % 3],
                        gangs=seed % 3, away=True)
    res = []
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl)
        W.prepare(s, wl)
        res.append(s.schedule_round())
    scenario.assert_same_round(res[0], res[1])
    assert any(m == 5 for m in res[0].scheduled_method.values()) or seed % 3 == 0  # ASCHED_METHOD_AWAY exercised

@pytest.mark.parametrize("seed", range(10))
def test_random_rounds_off_the_index_grid_match_oracle(hostsim_lib, oracle_lib, seed):
    """requests / allocatable that are not multiples of the index resolution, several node types, id order != index order: the
    reference's skip-scan and heap merge (nodeiteration.go:74-185, 318-382) are restated literally on the device for these"""
    wl = W.small_random(n_nodes=12 + seed * 8, n_jobs=250 + seed * 40, n_queues=2 + seed % 4, seed=200 + seed, occupied=[0.4, 0.8, 0.95][seed % 3],
                        gangs=seed % 3, ragged=True, away=seed % 2 == 1)
    res = []
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl)
        W.prepare(s, wl)
        res.append(s.schedule_round())
    scenario.assert_same_round(res[0], res[1])
    assert len(res[0].scheduled) > 0

def _subset(wl, keep):
    import copy
    wl = copy.deepcopy(wl)
    idx = np.nonzero(keep)[0]
    remap = -np.ones(wl.num_jobs, dtype=np.int64); remap[idx] = np.arange(len(idx))
    for f in ("job_req", "job_queue", "job_pc", "job_submit", "job_node", "job_run_prio", "job_run_ts", "job_gang", "job_gang_card"):
        setattr(wl, f, getattr(wl, f)[idx])
    wl.queued = [np.array([remap[j] for j in q if remap[j] >= 0], dtype=np.int32) for q in wl.queued]
    return wl


def _degenerate_cases():
    import copy
    base = W.small_random(n_nodes=12, n_jobs=80, n_queues=3, seed=5, occupied=0.5, gangs=1)
    cases = {"no-queued": _subset(base, base.job_node >= 0), "no-running": _subset(base, base.job_node < 0),
             "no-jobs": _subset(base, np.zeros(base.num_jobs, dtype=bool))}
    wl = copy.deepcopy(base); wl.queued[1] = np.zeros(0, np.int32); cases["empty-queue"] = wl
    cases["one-node-one-job"] = W.small_random(n_nodes=1, n_jobs=1, n_queues=1, seed=9, occupied=0.0, gangs=0)
    wl = copy.deepcopy(base); wl.job_req[wl.job_node < 0, 1] = 10 ** 9; cases["nothing-fits"] = wl
    wl = copy.deepcopy(base); wl.global_burst, wl.queue_burst, wl.rate_inf = 0, 0, False; cases["zero-burst"] = wl
    cases["70-queues"] = W.small_random(n_nodes=30, n_jobs=700, n_queues=70, seed=13, occupied=0.7, gangs=2)  # > 64: queue loop off the fast path
    return cases


@pytest.mark.parametrize("name", ["no-queued", "no-running", "no-jobs", "empty-queue", "one-node-one-job", "nothing-fits", "zero-burst", "70-queues"])
def test_degenerate_rounds_match_oracle(hostsim_lib, oracle_lib, name):
    """empty and ragged inputs: no queued / running / any jobs, an empty queue, 1x1, unschedulable requests, exhausted limiters, Q > 64"""
    wl = _degenerate_cases()[name]
    res = []
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl)
        W.prepare(s, wl)
        res.append(s.schedule_round())
    scenario.assert_same_round(res[0], res[1])

NODEDB = load("nodedb_schedule_individually") + load("nodedb_schedule_many") + load("nodedb_away_node_scheduling")


@pytest.mark.parametrize("case", NODEDB, ids=[c["source"].split("/")[-1] + ":" + c["name"] for c in NODEDB])
def test_nodedb_schedule_many_with_txn(hostsim_lib, case):
    """nodedb_test.go TestScheduleIndividually / TestScheduleMany through the NodeDb-level entry points (txn_begin, schedule_many, commit / abort)"""
    r = scenario.run_nodedb_schedule_case(hostsim_lib, case)
    assert r == "ok", f"case not run: {r}"   # a driver that declines a reference case is a failure, not a skip

def test_config1_simulator_shape_round_matches_oracle(hostsim_lib, oracle_lib):
    """BASELINE configs[0]: 100 nodes, 1 queue, 1k single-pod jobs on an empty cluster (the cmd/simulator basic shape)"""
    wl = W.config1()
    res = []
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl)
        W.prepare(s, wl)
        res.append(s.schedule_round())
    scenario.assert_same_round(res[0], res[1])
    assert len(res[0].scheduled) > 800 and not res[0].preempted

