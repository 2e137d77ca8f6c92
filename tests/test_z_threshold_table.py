"""Fair-share threshold table (armada_amd/csrc/round_ft.h): T[shape][node] = the evicted-table Index at which the node's own entries first cover the shape's
request (fairNodeBest), a two-level maximum over it, lazy validation of a query's winner.  It replaces the wide pass of selectNodeForJobWithFairPreemption
(nodedb.go:935-1043) for home attempts of queued jobs.  Not in the default device build (measured: HISTORY.md §9, profiles/r03g_*); the CPU build of the device
code carries it, so its logic is checked against the oracle here (and by every crowded round of the suite and the soaks)."""
import pytest

import scenario
from armada_amd import workloads as W


def _round(lib, wl, fp=None):
    s = W.load(lib, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp)
    r, st = s.schedule_round(), s.round_stats()
    s.close()
    return r, st


@pytest.mark.parametrize("seed", range(6))
def test_crowded_rounds_through_the_threshold_table(hostsim_lib, oracle_lib, monkeypatch, seed):
    wl = W.config3(seed=40 + seed, n_nodes=300 + 150 * seed, n_jobs=4000 + 1500 * seed, n_queues=4 + seed, occupied=[0.9, 0.95, 1.0][seed % 3], gangs=[0, 6][seed % 2])
    wl.global_burst, wl.queue_burst = wl.num_jobs // 4, wl.num_jobs // 12
    want, _ = _round(oracle_lib, wl)
    got, st = _round(hostsim_lib, wl)
    scenario.assert_same_round(want, got)
    assert st["ft_queries"] > 50 and st["ft_node_updates"] >= st["ft_retries"], st
    assert any(m == 3 for m in got.scheduled_method.values())      # fair-share preemption happened, through the table
    monkeypatch.setenv("ASCHED_FT", "0")                             # the same round with the wide pass: the table changes nothing but the way there
    off, st0 = _round(hostsim_lib, wl)
    scenario.assert_same_round(want, off)
    assert st0["ft_queries"] == 0


def test_table_follows_transaction_aborts_and_rate_limits(hostsim_lib, oracle_lib):
    """gangs that fail half-way (undo log: evicted-table entries come back, binds are taken back) and a fair-share preemption rate limit that runs dry"""
    wl = W.small_random(n_nodes=90, n_jobs=2500, n_queues=5, seed=77, occupied=0.97, gangs=10)
    want, _ = _round(oracle_lib, wl, fp=25.0)
    got, st = _round(hostsim_lib, wl, fp=25.0)
    scenario.assert_same_round(want, got)
    assert st["ft_queries"] > 0
