"""Market-driven ordering's iterators: MarketBasedCandidateGangIterator (market_iterator.go:32-295), jobdb.MarketSchedulingOrderCompare
(jobdb/comparison.go:113-170) and MarketDrivenMultiJobsIterator (jobiteration.go:232-321).

The candidate order is produced by container/heap over MarketIteratorPQ.Less (:228-273), which reads the queue and price of the PREVIOUS result (round
robin between queues that bid the same price) and therefore is not a strict weak order: the heap's up / down moves are restated literally — by the oracle
(oracle_market_iterate) and by the device (armada_amd/csrc/round_market.h, run by the auxiliary kernel).  Every case below runs on the oracle, on the CPU
build of the device code and (-m gpu) on the HIP library.
Cases: market_iterator_test.go:17-34 (home before away), :36-50 (ordering), :52-122 (round robin, 4 tables), comparison_test.go:76-178 (11 cases,
transcribed mechanically), jobiteration_test.go:150-232 (3 tests).  The market ROUND (evictor, spot price, pricer) is not built: HISTORY.md §9.
"""
import pytest

from armada_amd.binding import Scheduler, SchedError
from armada_amd import workloads as W


@pytest.fixture(params=["oracle", "hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def mlib(request):
    return request.getfixturevalue(request.param + "_lib")


def handle(lib):
    wl = W.small_random(n_nodes=4, n_jobs=50, n_queues=2, seed=1)
    return Scheduler(lib, wl.config)


ROUND_ROBIN = {   # market_iterator_test.go:61-98; instantiation order = age: the first declared queue holds the oldest jobs
    "two queues": ([("A", [600, 600, 600]), ("B", [600, 600, 400])], ["A", "B", "A", "B", "A", "B"]),
    "two queues - reverse": ([("B", [600, 600, 400]), ("A", [600, 600, 600])], ["B", "A", "B", "A", "A", "B"]),
    "three queues": ([("A", [600, 600, 400]), ("B", [600, 600, 600]), ("C", [600, 600, 300])], ["A", "B", "C", "A", "B", "C", "B", "A", "C"]),
    "three queues - mixed": ([("C", [600, 600, 300]), ("A", [600, 600, 400]), ("B", [600, 600, 600])], ["C", "A", "B", "C", "A", "B", "B", "A", "C"]),
}


@pytest.mark.parametrize("name", sorted(ROUND_ROBIN))
def test_round_robin(mlib, name):
    inp, expected = ROUND_ROBIN[name]
    names = sorted(q for q, _ in inp)
    t = 0
    queues = []
    for q, prices in inp:          # createJobIterator(sctx, queue, prices...) in declaration order: submit times grow with it
        js = []
        for p in prices:
            t += 1
            js.append(dict(price=p, queued=True, submit_time=t))
        queues.append(js)
    s = handle(mlib)
    got = s.market_iterate(queues, [names.index(q) for q, _ in inp])
    assert [inp[i][0] for i in got] == expected
    s.close()


def test_ordering(mlib):
    """:36-50 TestMarketIteratorPQ_Ordering: price desc, running before queued, runtime desc, submit time asc, queue name.  The reference sorts hand-built
    items with sort.Sort; draining the iterator over one-job queues with the same item values visits them in the same order (the round-robin clause never
    decides here: F and G bid 1 when the previous result bid 2)"""
    items = {"A": dict(price=3, queued=True, runtime=10, submit_time=10), "B": dict(price=3, queued=False, runtime=10, submit_time=10),
             "C": dict(price=2, queued=True, runtime=10, submit_time=10), "D": dict(price=2, queued=True, runtime=8, submit_time=10),
             "E": dict(price=2, queued=True, runtime=8, submit_time=5), "F": dict(price=1, queued=True, runtime=8, submit_time=10),
             "G": dict(price=1, queued=True, runtime=8, submit_time=10)}
    order = ["G", "F", "E", "D", "C", "B", "A"]   # pq.items as the test builds it
    s = handle(mlib)
    got = s.market_iterate([[items[q]] for q in order], [sorted(items).index(q) for q in order])
    assert [order[i] for i in got] == ["B", "A", "C", "E", "D", "F", "G"]
    s.close()


def test_home_before_away(mlib):
    """:17-34 TestMarketIteratorPQ_HomeBeforeAway: with preemptCrossPoolJobsFirst a home item orders before an away item despite the lower price"""
    s = handle(mlib)
    queues = [[dict(price=1)], [dict(price=1000, away=True)]]     # "q", "q-away"
    assert s.market_iterate(queues, [0, 1], preempt_cross_pool_jobs_first=True) == [0, 1]
    assert s.market_iterate(queues, [0, 1], preempt_cross_pool_jobs_first=False) == [1, 0]
    s.close()


def test_bigger_random_tables_match_the_oracle(mlib, oracle_lib):
    """seeded tables with many price ties (the round-robin clause decides constantly): the device's heap moves against the oracle's"""
    import numpy as np
    for seed in range(12):
        rng = np.random.default_rng(seed)
        nq = int(rng.integers(1, 40))
        queues = [[dict(price=float(rng.integers(1, 4)), queued=bool(rng.random() < 0.7), runtime=int(rng.integers(0, 3)), submit_time=int(rng.integers(0, 5)),
                        away=bool(rng.random() < 0.2)) for _ in range(int(rng.integers(0, 12)))] for _ in range(nq)]
        ranks = rng.permutation(nq).tolist()
        a, b = handle(oracle_lib), handle(mlib)
        for pref in (False, True):
            assert a.market_iterate(queues, ranks, preempt_cross_pool_jobs_first=pref) == b.market_iterate(queues, ranks, preempt_cross_pool_jobs_first=pref), seed
        a.close(); b.close()


# ---- jobdb.MarketSchedulingOrderCompare (comparison.go:113-170) on the reference's own table (comparison_test.go:76-178), transcribed mechanically
from golden_io import ids, load  # noqa: E402

CMP = load("market_job_priority_comparer")


@pytest.mark.parametrize("case", CMP, ids=ids(CMP))
def test_market_job_priority_comparer(mlib, case):
    pool = case["currentPool"]
    names = sorted({case["a"]["id"], case["b"]["id"]})

    def job(j):
        return dict(bid_price=j["bidPrices"].get(pool, 0.0), active_run_timestamp=j["activeRunTimestamp"], submit_time=j["submittedTime"],
                    pc_priority=j["pcPriority"], active=j["active"], id_rank=names.index(j["id"]))
    s = handle(mlib)
    assert s.market_compare(job(case["a"]), job(case["b"])) == case["expected"], case["name"]
    assert s.market_compare(job(case["b"]), job(case["a"])) == -case["expected"]
    s.close()


# ---- MarketDrivenMultiJobsIterator (jobiteration.go:232-321), jobiteration_test.go:150-232 by hand.  createJctxWithPrice: a queued job of one priority class
# whose bid is its price band's number (testfixtures.SetPricing: A = 1 ... H = 8), ids (ULIDs) and submit times grow with creation order.
BAND = {c: i + 1 for i, c in enumerate("ABCDEFGH")}


def _jobs(spec, counter):
    out = []
    for band, evicted in spec:
        counter[0] += 1
        out.append(dict(bid_price=BAND[band], submit_time=counter[0], id_rank=counter[0], evicted=evicted))
    return out


def test_market_multi_jobs_iterator(mlib):
    c = [0]
    new = _jobs([("H", False), ("C", False)], c)
    ev = _jobs([("F", True), ("D", True)], c)
    s = handle(mlib)
    assert s.market_multi_iterate(new, ev) == [0, 2 + 0, 2 + 1, 1]                    # :150-171 new[0], evicted[0], evicted[1], new[1]
    s.close()


def test_market_multi_jobs_iterator_only_yield_evicted(mlib):
    c = [0]
    new = _jobs([("H", False), ("H", False)], c)
    ev = _jobs([("F", True), ("D", True)], c)
    s = handle(mlib)
    assert s.market_multi_iterate(new, ev, only_evicted_after=0) == [2 + 0, 2 + 1]    # :173-195 OnlyYieldEvicted before the first Next
    s.close()


def test_market_multi_jobs_iterator_only_yield_evicted_mid_iteration(mlib):
    c = [0]
    l1 = _jobs([("H", False), ("E", False), ("B", True)], c)
    l2 = _jobs([("F", False), ("F", False), ("D", True)], c)
    s = handle(mlib)
    assert s.market_multi_iterate(l1, l2, only_evicted_after=2) == [0, 3 + 0, 3 + 2, 2]   # :197-232 jctxs1[0], jctxs2[0], jctxs2[2], jctxs1[2]
    s.close()
