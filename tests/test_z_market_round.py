"""Market-driven rounds (asched_set_market): the device restatement (round_mkt.h, run by the auxiliary kernel; the CPU build of the same code here) against
the oracle on seeded crowded rounds with tied and distinct bid prices, gangs, away node types, rate limits — every scheduled / preempted job and node, the queue
accounting, the spot price, the billable resources and the second-price overrides.  The reference's own table (TestMarketDrivenPreemptingQueueScheduler, 18 cases)
runs in test_oracle_golden.py / test_hostsim.py / test_gpu_parity.py."""
import numpy as np
import pytest

import scenario
from armada_amd import workloads as W
from armada_amd.binding import SchedError


def market_case(seed):
    rng = np.random.default_rng(seed)
    wl = W.small_random(n_nodes=int(rng.integers(3, 60)), n_jobs=int(rng.integers(40, 1500)), n_queues=int(rng.integers(1, 9)), seed=seed,
                        occupied=float(rng.choice([0.0, 0.5, 0.9, 1.0])), gangs=int(rng.integers(0, 6)),
                        burst=None if rng.random() < 0.6 else (int(rng.integers(10, 800)), int(rng.integers(5, 300))), away=bool(rng.random() < 0.3))
    levels = rng.choice([1, 2, 3, 6])
    bids = rng.integers(0, levels + 1, size=wl.num_jobs).astype(np.float64)          # few price levels: many ties, the round-robin clause of Less decides
    if rng.random() < 0.3:
        bids += rng.random(wl.num_jobs).round(2)                                       # ... or nearly all distinct
    if wl.job_gang is not None:                                                       # a gang bids one price (the reference prices the gang by its first member)
        for g in set(int(x) for x in wl.job_gang if x >= 0):
            m = np.nonzero(wl.job_gang == g)[0]
            bids[m] = bids[m[0]]
    nonpre = np.array([not wl.config.pc_preemptible[p] for p in wl.job_pc])
    bids[(wl.job_node >= 0) & nonpre] = 1_000_000.0                                    # pricing.NonPreemptibleRunningPrice (jobdb/job.go:459-463)
    cutoff = float(rng.choice([0.0, 0.05, 0.3, 0.9]))
    return wl, bids, cutoff


def market_round(lib, wl, bids, cutoff, queues_only=False):
    s = W.load(lib, wl)
    W.set_jobs(s, wl, bid_price=bids)
    pcp = np.asarray(wl.config.pc_priority)
    queued = [sorted(q, key=lambda j: (-int(pcp[wl.job_pc[j]]), -float(bids[j]), int(wl.job_submit[j]) if wl.job_submit is not None else int(j), int(j))) for q in wl.queued]   # jobdb.PriceOrder
    nq = wl.num_queues
    s.round_prepare(wl.queue_weight, queued, global_tokens=float(wl.global_burst), global_burst=wl.global_burst, global_rate_inf=wl.rate_inf,
                    queue_tokens=[float(wl.queue_burst)] * nq, queue_burst=[wl.queue_burst] * nq, queue_rate_inf=[wl.rate_inf] * nq)
    s.set_market(True, cutoff)
    res = s.schedule_queues() if queues_only else s.schedule_round()
    mr = s.market_result()
    s.close()
    return res, mr


def compare(lib, oracle_lib, seeds):
    spot = sched = over = 0
    for seed in seeds:
        wl, bids, cutoff = market_case(seed)
        a, ma = market_round(oracle_lib, wl, bids, cutoff)
        b, mb = market_round(lib, wl, bids, cutoff)
        scenario.assert_same_round(a, b)
        assert ma["spot_price"] == mb["spot_price"], (seed, ma["spot_price"], mb["spot_price"])
        assert (ma["billable"] == mb["billable"]).all(), seed
        assert ma["price_override"] == mb["price_override"], (seed, ma["price_override"], mb["price_override"])
        spot += ma["spot_price"] is not None; sched += len(a.scheduled); over += sum(v is not None for v in ma["price_override"])
    return spot, sched, over


def test_market_rounds_hostsim_equal_oracle(hostsim_lib, oracle_lib):
    spot, sched, over = compare(hostsim_lib, oracle_lib, range(7000, 7060))
    assert spot >= 20 and sched > 2000 and over >= 20   # the rounds do set spot prices and overrides


def queues_only(lib, oracle_lib, seeds):
    n = 0
    for seed in seeds:
        wl, bids, cutoff = market_case(seed)
        a, ma = market_round(oracle_lib, wl, bids, cutoff, queues_only=True)
        b, mb = market_round(lib, wl, bids, cutoff, queues_only=True)
        scenario.assert_same_round(a, b)
        assert ma["spot_price"] == mb["spot_price"] and (ma["billable"] == mb["billable"]).all() and ma["price_override"] == mb["price_override"], seed
        n += len(a.scheduled)
    return n


def test_market_queue_scheduler_alone(hostsim_lib, oracle_lib):
    """asched_schedule_queues on a market-driven pool: QueueScheduler.Schedule with the market iterator and the spot price, no eviction phases"""
    assert queues_only(hostsim_lib, oracle_lib, range(7100, 7130)) > 500


def test_market_mode_changes_the_round(oracle_lib):
    """negative control: the same inputs without market mode give other rounds (so the comparison above is about the market code)"""
    differ = 0
    for seed in range(7000, 7020):
        wl, bids, cutoff = market_case(seed)
        a, _ = market_round(oracle_lib, wl, bids, cutoff)
        s = W.load(oracle_lib, wl); W.set_jobs(s, wl, bid_price=bids); W.prepare(s, wl); b = s.schedule_round(); s.close()
        differ += (a.scheduled != b.scheduled) or (a.preempted != b.preempted)
    assert differ >= 10


def test_market_round_needs_bids_or_mode_before_jobs(hostsim_lib):
    wl = W.small_random(n_nodes=5, n_jobs=60, n_queues=2, seed=1)
    s = W.load(hostsim_lib, wl); W.prepare(s, wl)
    s.set_market(True, 0.5)
    with pytest.raises(SchedError):
        s.schedule_round()
    s.close()


def test_market_with_fairshare_preemption_limiter_is_refused(hostsim_lib, oracle_lib):
    """the reference rejects the combination at config validation (market_iterator.go:178-183)"""
    wl, bids, cutoff = market_case(7001)
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl); W.set_jobs(s, wl, bid_price=bids); W.prepare(s, wl, fairshare_preemption_tokens=3.0)
        s.set_market(True, cutoff)
        with pytest.raises(SchedError):
            s.schedule_round()
        s.close()


@pytest.mark.gpu
def test_market_rounds_gpu_equal_oracle(hip_lib, oracle_lib):
    spot, sched, over = compare(hip_lib, oracle_lib, range(7000, 7030))
    assert spot >= 10 and sched > 1000
    assert queues_only(hip_lib, oracle_lib, range(7100, 7115)) > 200
