"""Jobs that need preemption served inside the fast loop (round_fast.h fastPreemptIter).

The queue side of such an iteration is the fast path's (constraints, accounting in LDS, next head), the node side is the generic cascade called as is
(SelectNodeForJobWithTxn, nodedb.go:538-838: gate, fair-share preemption, urgency preemption; gang_scheduler.go:229-262 for the transaction).  Reference
behaviour = the ordinary QueueScheduler loop (queue_scheduler.go:94-304), including the fair-share preemption rate limit that is checked before every
Peek (:114-142) and the unfeasible-key registration of a job that found no node (gang_scheduler.go:63-98): every round is compared with the oracle job by
job, with the counter saying that the iterations really stayed in the fast loop.  HS_NO_PREEMPT_FAST=1 (CPU build) switches the path off.
"""
import numpy as np
import pytest

from armada_amd import workloads as W
import bench


def crowded(seed):
    rng = np.random.default_rng(seed)
    wl = W.small_random(n_nodes=int(rng.integers(4, 200)), n_jobs=int(rng.integers(50, 4000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                        occupied=float(rng.choice([0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 8)),
                        burst=None if rng.random() < 0.5 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                        away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.2))
    fp = None if rng.random() < 0.3 else float(rng.choice([0, 1, 3, 10, 40]))
    return wl, fp


def both(lib, oracle, wl, fp=None):
    out = []
    for l in (lib, oracle):
        s = W.load(l, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp)
        r = s.schedule_round()
        out.append((r, s.round_stats()))
        s.close()
    assert bench.round_diff(out[0][0], out[1][0]) == []
    return out[0]


@pytest.mark.parametrize("seed", range(700000, 700040))
def test_crowded_rounds_equal_oracle(hostsim_lib, oracle_lib, seed):
    wl, fp = crowded(seed)
    both(hostsim_lib, oracle_lib, wl, fp)


def test_preemptions_stay_in_the_fast_loop(hostsim_lib, oracle_lib, monkeypatch):
    wl = W.config3(seed=31, n_nodes=800, n_jobs=8000, n_queues=12, occupied=0.95)
    wl.global_burst, wl.queue_burst = 3000, 600
    r, st = both(hostsim_lib, oracle_lib, wl)
    assert len(r.preempted) > 200
    assert st["preempt_fast_iterations"] > 200 and st["generic_iterations"] < st["preempt_fast_iterations"] // 10
    monkeypatch.setenv("HS_NO_PREEMPT_FAST", "1")
    r2, st2 = both(hostsim_lib, oracle_lib, wl)
    assert bench.round_diff(r, r2) == [] and st2["preempt_fast_iterations"] == 0
    assert st2["generic_iterations"] > st["preempt_fast_iterations"] // 2


@pytest.mark.parametrize("tokens", [0.0, 1.0, 5.0, 50.0])
def test_fairshare_preemption_rate_limit_inside_the_fast_loop(hostsim_lib, oracle_lib, tokens):
    """sctx.FairsharePreemptionLimiter (context/scheduling.go:508-528): the limit is looked at before every Peek (queue_scheduler.go:114-121), also
    when the preemption that spent the last token happened in the fast loop"""
    wl = W.config3(seed=32, n_nodes=300, n_jobs=3000, n_queues=8, occupied=0.95)
    wl.global_burst, wl.queue_burst = 2000, 400
    both(hostsim_lib, oracle_lib, wl, tokens)


def test_job_without_a_node_registers_its_key_inside_the_fast_loop(hostsim_lib, oracle_lib):
    """a full cluster of non-preemptible work: the first job of every shape finds no node anywhere, its scheduling key becomes unfeasible and
    the queues' later jobs of that shape are skipped at peek time (queue_scheduler.go:398-413) — including jobs already laid out in stream runs"""
    for seed in (41, 42, 43):
        wl = W.small_random(n_nodes=40, n_jobs=1500, n_queues=6, seed=seed, occupied=0.0, gangs=0)
        r, st = both(hostsim_lib, oracle_lib, wl)
        assert st["preempt_fast_iterations"] > 0


@pytest.mark.gpu
def test_crowded_rounds_gpu(hip_lib, oracle_lib):
    for seed in range(700000, 700012):
        wl, fp = crowded(seed)
        both(hip_lib, oracle_lib, wl, fp)


@pytest.mark.gpu
def test_preemptions_stay_in_the_fast_loop_gpu(hip_lib, oracle_lib):
    wl = W.config3(seed=31, n_nodes=2000, n_jobs=20000, n_queues=16, occupied=0.95)
    wl.global_burst, wl.queue_burst = 6000, 800
    r, st = both(hip_lib, oracle_lib, wl)
    assert len(r.preempted) > 500 and st["preempt_fast_iterations"] > 500
    for tokens in (0.0, 7.0):
        both(hip_lib, oracle_lib, wl, tokens)


def _long_skip_workload(variant):
    """BenchmarkPreemptingQueueScheduler's steady state (preempting_queue_scheduler_test.go:2561-2799): full nodes, queues of thousands of identical jobs that no longer fit — the
    first one registers its scheduling key as unfeasible, the rest are skipped at peek time (queue_scheduler.go:398-413).  Stretches of >= SKIP_BULK_MIN jobs are skipped as two
    bulk passes (round_wide.h skipUnfeasibleBulk); the variants put what ENDS a stretch in the middle of one: a job of another shape, a gang, the lookback limit."""
    wl = W.reference_benchmark(6, 3, 9000)
    first = 6 * 32                                  # the first round's jobs fill the 6 nodes (32 x 1 cpu each)
    node = wl.job_node.copy(); prio = wl.job_run_prio.copy()
    filled = 0
    for q in wl.queued:
        for j in q[:first // 3]:
            node[int(j)] = filled // 32; prio[int(j)] = 0; filled += 1
    wl.job_node, wl.job_run_prio = node.astype(np.int32), prio.astype(np.int32)
    wl.queued = [np.array([int(j) for j in q if node[int(j)] < 0], dtype=np.int32) for q in wl.queued]
    if variant == "other_shape":                    # a 2-cpu job 5 000 jobs into queue 1: its key is not registered — the stretch stops there, it is attempted (and fails), the rest goes on
        wl.job_req[int(wl.queued[1][5000]), W.CPU] = 2000
    elif variant == "lookback":
        wl.config.max_queue_lookback = 3000         # every queue stops looking after 3 000 jobs: the stretch is cut by the limit
    return wl


@pytest.mark.parametrize("variant", ["plain", "other_shape", "lookback"])
def test_long_stretches_of_known_unfeasible_jobs(hostsim_lib, oracle_lib, variant):
    wl = _long_skip_workload(variant)
    r, st = both(hostsim_lib, oracle_lib, wl)
    reasons = np.asarray(r.job_unschedulable_reason)
    skipped = int((reasons == 17).sum())            # ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY
    assert len(r.scheduled) == 0 and skipped > (8000 if variant == "lookback" else 26000), (len(r.scheduled), skipped)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "other_shape", "lookback"])
def test_long_stretches_of_known_unfeasible_jobs_gpu(hip_lib, oracle_lib, variant):
    wl = _long_skip_workload(variant)
    r, st = both(hip_lib, oracle_lib, wl)
    assert len(r.scheduled) == 0
