"""Pools of more than 64 queues (armada_amd/csrc/round_wide.h): the stream run whose k-way merge is a bulk rank over every queue's precomputed key sequence.
CostBasedCandidateGangIterator has no queue limit (queue_scheduler.go:449-699); round 3 ran such pools on the generic iteration alone (slower than one host core).
Whole rounds against the oracle at 65 ... 1 024 queues — evicted jobs returning, gangs, rate limits, lookback limits, crowded clusters where jobs need preemption —
on the CPU build of the device code and (-m gpu) on the HIP library; the wide runs must actually serve the bulk of the iterations where nothing needs preemption."""
import numpy as np
import pytest

import scenario
from armada_amd import workloads as W


def _wl(seed, nq, nn=None, nj=None, **kw):
    rng = np.random.default_rng(seed)
    nn = nn or int(rng.integers(80, 900)); nj = nj or int(rng.integers(2000, 12000))
    wl = W.config3(seed=seed, n_nodes=nn, n_jobs=nj, n_queues=nq, gangs=kw.pop("gangs", int(rng.choice([0, 0, 5, 30]))), occupied=kw.pop("occupied", float(rng.choice([0.2, 0.5, 0.8]))))
    wl.global_burst = kw.pop("global_burst", int(rng.choice([nj, nj // 3, 700])))
    wl.queue_burst = kw.pop("queue_burst", int(rng.choice([nj, max(10, 4 * nj // nq), 24])))
    wl.rate_inf = kw.pop("rate_inf", bool(rng.random() < 0.2))
    if "lookback" in kw:
        wl.config.max_queue_lookback = kw.pop("lookback")
    return wl


def _round(lib, wl):
    s = W.load(lib, wl); W.prepare(s, wl)
    r = s.schedule_round(); st = s.round_stats()
    s.close()
    return r, st


CASES = [(65, 1), (70, 2), (130, 3), (256, 4), (257, 5), (400, 6), (1024, 7)]


@pytest.mark.parametrize("nq,seed", CASES)
def test_wide_rounds_match_oracle_cpu_build(hostsim_lib, oracle_lib, nq, seed):
    wl = _wl(seed, nq)
    a, st = _round(hostsim_lib, wl)
    b, _ = _round(oracle_lib, wl)
    scenario.assert_same_round(b, a)
    assert st["stream_runs"] > 0 and st["fast_iterations"] > 0, "the wide runs did not run"


@pytest.mark.parametrize("kw", [dict(lookback=20), dict(lookback=300), dict(queue_burst=8, rate_inf=False), dict(global_burst=300, rate_inf=False), dict(gangs=60), dict(occupied=0.95)],
                         ids=["lookback20", "lookback300", "queue-tokens", "global-tokens", "gangs", "crowded"])
def test_wide_rounds_where_runs_must_stop(hostsim_lib, oracle_lib, kw):
    for seed in (11, 12, 13):
        wl = _wl(seed, 100, **dict(kw))
        a, _ = _round(hostsim_lib, wl)
        b, _ = _round(oracle_lib, wl)
        scenario.assert_same_round(b, a)


def test_wide_runs_serve_the_round_when_nothing_needs_preemption(hostsim_lib, oracle_lib):
    """bench.py's 256-queue shape at a tenth of its size: a roomy cluster, every evicted job returns, every new job fits — all but a handful of iterations in wide runs"""
    wl = W.config3(seed=W.SEED, n_nodes=2_000, n_jobs=20_000, n_queues=256)
    wl.global_burst, wl.queue_burst = 4_000, 400
    a, st = _round(hostsim_lib, wl)
    b, _ = _round(oracle_lib, wl)
    scenario.assert_same_round(b, a)
    assert st["fast_iterations"] >= a.num_loop_iterations - 8 and st["generic_iterations"] <= 8, (st["fast_iterations"], st["generic_iterations"], a.num_loop_iterations)
    assert st["stream_runs"] <= 40


def test_wide_off_is_the_generic_path(hostsim_lib, oracle_lib, monkeypatch):
    monkeypatch.setenv("ASCHED_WIDE", "0")
    wl = _wl(3, 130)
    a, st = _round(hostsim_lib, wl)
    b, _ = _round(oracle_lib, wl)
    scenario.assert_same_round(b, a)
    assert st["stream_runs"] == 0 and st["fast_iterations"] == 0


def test_a_handle_goes_back_to_few_queues(hostsim_lib, oracle_lib):
    """the Q-sized stream buffers of a wide round must not leak into a later round of the same handle with <= 64 queues (and the other way round)"""
    wl_wide, wl_few = _wl(31, 100, nn=300, nj=4000), _wl(32, 12, nn=300, nj=4000)
    for order in ((wl_wide, wl_few, wl_wide), (wl_few, wl_wide, wl_few)):
        for wl in order:
            a, _ = _round(hostsim_lib, wl)
            b, _ = _round(oracle_lib, wl)
            scenario.assert_same_round(b, a)


@pytest.mark.gpu
@pytest.mark.parametrize("nq,seed", CASES)
def test_wide_rounds_match_oracle_on_the_device(hip_lib, oracle_lib, nq, seed):
    wl = _wl(seed, nq)
    a, st = _round(hip_lib, wl)
    b, _ = _round(oracle_lib, wl)
    scenario.assert_same_round(b, a)
    assert st["stream_runs"] > 0 and st["fast_iterations"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(lookback=20), dict(queue_burst=8, rate_inf=False), dict(global_burst=300, rate_inf=False), dict(gangs=60), dict(occupied=0.95)],
                         ids=["lookback20", "queue-tokens", "global-tokens", "gangs", "crowded"])
def test_wide_rounds_where_runs_must_stop_on_the_device(hip_lib, oracle_lib, kw):
    for seed in (11, 12):
        wl = _wl(seed, 100, **dict(kw))
        a, _ = _round(hip_lib, wl)
        b, _ = _round(oracle_lib, wl)
        scenario.assert_same_round(b, a)


@pytest.mark.gpu
def test_wide_round_at_scale_on_the_device(hip_lib, oracle_lib):
    """20 000 nodes x 200 000 queued jobs x 256 queues (bench.py's sub-record): 31 helper workgroups share the rank pass"""
    wl = W.config3(seed=W.SEED, n_nodes=20_000, n_jobs=200_000, n_queues=256)
    wl.global_burst, wl.queue_burst = 40_000, 4_000
    a, st = _round(hip_lib, wl)
    b, _ = _round(oracle_lib, wl)
    scenario.assert_same_round(b, a)
    assert st["fast_iterations"] >= 0.9 * a.num_loop_iterations


def _reuse_rounds(lib, wls):
    """ONE handle through several rounds, nodes and jobs uploaded again before each (the lifecycle of integration/gpu_round.go: one handle per pool, UploadJobs every cycle)"""
    s = W.load(lib, wls[0])
    out = []
    for i, wl in enumerate(wls):
        if i:
            s.nodes_upsert(wl.node_total, wl.node_allocatable, taints=wl.node_taints, labels=wl.node_labels, id_rank=wl.node_id_rank)
            W.set_jobs(s, wl)
        W.prepare(s, wl)
        out.append(s.schedule_round())
    s.close()
    return out


REUSE = [(100, 8), (100, 8, 100, 8), (8, 100, 8), (300, 64, 65, 3)]


def _reuse_wls(qs):
    return [_wl(40 + i, nq, nn=300, nj=4000) for i, nq in enumerate(qs)]


@pytest.mark.parametrize("qs", REUSE)
def test_a_reused_handle_drops_from_many_to_few_queues(hostsim_lib, oracle_lib, qs):
    """round-4 advisor, high: a wide round parked the 64-queue stream buffers in the handle; the next jobs_set freed and reallocated them, and the following round_prepare put
    the FREED pointers back over the fresh ones — a round with <= 64 queues then wrote through dangling pointers (segfault in the CPU build)."""
    wls = _reuse_wls(qs)
    for b, a in zip(_reuse_rounds(oracle_lib, wls), _reuse_rounds(hostsim_lib, wls)):
        scenario.assert_same_round(b, a)


@pytest.mark.gpu
@pytest.mark.parametrize("qs", REUSE)
def test_a_reused_handle_drops_from_many_to_few_queues_on_the_device(hip_lib, oracle_lib, qs):
    wls = _reuse_wls(qs)
    for b, a in zip(_reuse_rounds(oracle_lib, wls), _reuse_rounds(hip_lib, wls)):
        scenario.assert_same_round(b, a)
