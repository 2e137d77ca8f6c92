"""Stream runs and gangs through the ring (armada_amd/csrc/round_fast.h): the part of the round where the control wave only merges
precomputed per-queue cost streams while the node engine places the jobs.  The reference behaviour is the ordinary QueueScheduler loop
(queue_scheduler.go:94-304, Less :738-798, gang_scheduler.go:46-148): every round here is compared with the oracle job by job, and the counters say that the
rounds really went through the stream / ring code.  HS_STREAM_EAGER=1 (CPU build only) starts a stream run wherever one can start, which multiplies the number
of run starts, ends and discarded entries per round; the production back-off is tested as is.
"""
import numpy as np
import pytest

from armada_amd import workloads as W
import bench


def workload(seed, gangs=0, occupied=0.5, n_nodes=1200, n_jobs=16000, n_queues=24, burst=None, lookback=0):
    wl = W.config3(seed=seed, n_nodes=n_nodes, n_jobs=n_jobs, n_queues=n_queues, gangs=gangs, occupied=occupied)
    wl.global_burst, wl.queue_burst = burst or (n_jobs // 3, max(10, n_jobs // n_queues))
    if lookback:
        wl.config.max_queue_lookback = lookback
    return wl


def both(lib, oracle, wl):
    out = []
    for l in (lib, oracle):
        s = W.load(l, wl); W.prepare(s, wl)
        r = s.schedule_round()
        out.append((r, s.round_stats()))
        s.close()
    assert bench.round_diff(out[0][0], out[1][0]) == []
    return out[0]


@pytest.mark.parametrize("seed", range(4))
def test_stream_rounds_equal_oracle(hostsim_lib, oracle_lib, seed):
    r, st = both(hostsim_lib, oracle_lib, workload(900 + seed, occupied=[0.2, 0.5, 0.8, 0.5][seed], lookback=[0, 0, 0, 400][seed]))
    assert st["stream_runs"] > 0 and st["stream_jobs"] > 0
    if seed == 0:
        assert st["stream_jobs"] > len(r.scheduled) // 2      # an uncrowded cluster: most of the round runs inside stream runs
    assert st["stream_emitted"] >= st["stream_jobs"]


@pytest.mark.parametrize("seed", range(6))
def test_stream_runs_started_everywhere_equal_oracle(hostsim_lib, oracle_lib, seed, monkeypatch):
    monkeypatch.setenv("HS_STREAM_EAGER", "1")
    rng = np.random.default_rng(seed)
    wl = workload(950 + seed, gangs=int(rng.choice([0, 5, 40])), occupied=float(rng.choice([0.3, 0.8, 0.93])), n_queues=int(rng.integers(3, 30)),
                  burst=(int(rng.choice([16000, 5000, 500])), int(rng.choice([16000, 700, 64]))))
    r, st = both(hostsim_lib, oracle_lib, wl)
    assert st["stream_runs"] >= 5   # (round 6: streams cover a queue's whole burst and every head gets one from the first preparation on — fewer, longer runs)


def test_streams_off_is_the_same_round(hostsim_lib, oracle_lib, monkeypatch):
    wl = workload(990)
    a, sa = both(hostsim_lib, oracle_lib, wl)
    monkeypatch.setenv("HS_NO_STREAM", "1")
    b, sb = both(hostsim_lib, oracle_lib, wl)
    assert sb["stream_runs"] == 0 and sa["stream_runs"] > 0 and bench.round_diff(a, b) == []


@pytest.mark.parametrize("seed", range(3))
def test_gangs_go_through_the_ring(hostsim_lib, oracle_lib, seed, monkeypatch):
    wl = workload(970 + seed, gangs=150, occupied=0.4)
    r, st = both(hostsim_lib, oracle_lib, wl)
    monkeypatch.setenv("HS_NO_GANG_RING", "1")
    r2, st2 = both(hostsim_lib, oracle_lib, wl)
    assert st["generic_iterations"] < st2["generic_iterations"] // 2          # most gangs left the generic path (those that need preemption or do not fit stay) ...
    assert bench.round_diff(r, r2) == []                                       # ... with the same result


def test_gang_that_does_not_fit_is_taken_back(hostsim_lib, oracle_lib):
    """a crowded cluster: gangs whose first members find nodes and a later member does not — nothing of the attempt may remain (held binds are dropped,
    the nodes the placed members went to are re-read), and the generic code then fails the gang exactly as the reference does"""
    wl = workload(985, gangs=120, occupied=0.93, n_nodes=300, n_jobs=6000, n_queues=6)
    r, st = both(hostsim_lib, oracle_lib, wl)
    unsched_gang = [j for j in range(wl.num_jobs) if wl.job_gang[j] >= 0 and wl.job_node[j] < 0 and j not in set(r.scheduled)]
    assert unsched_gang, "the workload should leave some gangs unscheduled"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_stream_rounds_gpu(hip_lib, oracle_lib, seed):
    r, st = both(hip_lib, oracle_lib, workload(900 + seed, gangs=[0, 30, 0][seed], occupied=[0.3, 0.5, 0.9][seed], n_nodes=4000, n_jobs=60000, n_queues=40))
    assert st["stream_runs"] > 0


@pytest.mark.gpu
def test_gangs_through_the_ring_gpu(hip_lib, oracle_lib):
    r, st = both(hip_lib, oracle_lib, workload(971, gangs=400, occupied=0.4, n_nodes=4000, n_jobs=60000, n_queues=32))
    assert st["generic_iterations"] < 400 * 2      # far fewer than one per gang member (the gangs of this workload have 2-64 members)


@pytest.mark.gpu
def test_crowded_gangs_gpu(hip_lib, oracle_lib):
    both(hip_lib, oracle_lib, workload(985, gangs=120, occupied=0.93, n_nodes=300, n_jobs=6000, n_queues=6))


def test_cancel_inside_a_stream_run(hostsim_lib, oracle_lib):
    """hard timeout (queue_scheduler.go:105-112) while the round is inside a stream run: the node engine sees the cancel word (every 256 jobs), the run ends with
    the entries it had bound, the fast loop is left at once and the round returns ASCHED_ERR_TIMEOUT; after a fresh round_prepare the handle behaves like a new one"""
    from armada_amd.binding import ERR_TIMEOUT, SchedError
    wl = workload(991, occupied=0.3)
    s = W.load(hostsim_lib, wl); W.prepare(s, wl)
    s.set_deadline(1e-9)                       # expired when the round starts (the serial build cannot be interrupted from outside)
    with pytest.raises(SchedError) as e:
        s.schedule_round()
    assert e.value.code == ERR_TIMEOUT
    s.set_deadline(0)
    W.prepare(s, wl)
    r = s.schedule_round()
    o = W.load(oracle_lib, wl); W.prepare(o, wl)
    assert bench.round_diff(r, o.schedule_round()) == []
    s.close(); o.close()


def _soak_round_workload(seed):
    """the workload tests/soak.py `rounds` builds for a seed"""
    rng = np.random.default_rng(seed)
    return W.small_random(n_nodes=int(rng.integers(4, 300)), n_jobs=int(rng.integers(50, 6000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                          occupied=float(rng.choice([0.0, 0.3, 0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 12)),
                          burst=None if rng.random() < 0.4 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                          away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.2))


@pytest.mark.parametrize("seed", [100345, 101116, 102465, 102593])
def test_gang_assembled_behind_folded_evicted_streams(hostsim_lib, oracle_lib, seed):
    """QueuedGangIterator.Peek in the fast loop (fastPeekGang): a queue's evicted jobs come from its evicted stream while the other queues' evicted lists
    are folded (skip mode); the gang behind them has a higher priority class than entries already served, so it orders BEFORE them (Less,
    queue_scheduler.go:738-798 compares the priority first).  The folded state must be rebuilt around the last entry served, not around the gang's key:
    the gang fails on the full cluster and the generic gang scheduler's preemption attempt reads the nodes of every evicted job that has come back."""
    both(hostsim_lib, oracle_lib, _soak_round_workload(seed))


@pytest.mark.parametrize("gangs", [40, 400])
def test_gang_peek_on_and_off_is_the_same_round(hostsim_lib, oracle_lib, gangs, monkeypatch):
    wl = workload(977, gangs=gangs, occupied=0.3, n_nodes=600, n_jobs=9000, n_queues=12)
    r1, st1 = both(hostsim_lib, oracle_lib, wl)
    monkeypatch.setenv("HS_NO_GANG_PEEK", "1")
    r2, st2 = both(hostsim_lib, oracle_lib, wl)
    assert bench.round_diff(r1, r2) == []
    assert st1["generic_iterations"] <= st2["generic_iterations"]


@pytest.mark.gpu
def test_gang_behind_folded_evicted_streams_gpu(hip_lib, oracle_lib):
    for seed in (100345, 102465):
        both(hip_lib, oracle_lib, _soak_round_workload(seed))


def _soak_stream_workload(seed):
    """the workload tests/soak.py `streams` builds for a seed"""
    rng = np.random.default_rng(seed)
    nn, nj, nq = int(rng.integers(100, 1500)), int(rng.integers(2000, 20000)), int(rng.integers(2, 40))
    wl = W.config3(seed=seed, n_nodes=nn, n_jobs=nj, n_queues=nq, gangs=int(rng.choice([0, 0, 5, 50])), occupied=float(rng.choice([0.2, 0.5, 0.8, 0.93])))
    wl.global_burst = int(rng.choice([nj, nj // 3, 500])); wl.queue_burst = int(rng.choice([nj, max(10, nj // nq), 64])); wl.rate_inf = bool(rng.random() < 0.2)
    if rng.random() < 0.3:
        wl.config.max_queue_lookback = int(rng.choice([50, 500, 3000]))
    return wl


@pytest.mark.parametrize("seed", [100372, 100488])
def test_deferred_replay_runs_after_a_gate_that_passed(hostsim_lib, oracle_lib, seed, monkeypatch):
    """The eviction-order replay (addEvictedJobsToNodeDb, pqs.go:589-639) is deferred to its first use, the fair-share attempt of a job whose feasibility
    gate passed (nodedb.go:724-789).  Running it earlier — before a gate that then FAILS, as a first version of the one-pass gate + evaluation did —
    changes the round: found by the stream soak with a run started wherever one can start."""
    monkeypatch.setenv("HS_STREAM_EAGER", "1")
    both(hostsim_lib, oracle_lib, _soak_stream_workload(seed))


def _long_unfeasible_tail_workload():
    """tests/soak.py `rounds` seed 100036: ONE queue of 4 489 jobs on 97 empty nodes.  The nodes fill after ~345 jobs, the failing scheduling keys are registered as
    unfeasible, and a stream run then ends in front of more than SKIP_BULK_MIN (2 048) jobs that Peek skips with a record each (queue_scheduler.go:398-413)."""
    return _soak_round_workload(100036)


def test_long_unfeasible_tail_behind_a_stream_run(hostsim_lib, oracle_lib):
    r, st = both(hostsim_lib, oracle_lib, _long_unfeasible_tail_workload())
    assert st["stream_runs"] > 0 and len(r.scheduled) > 300


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_long_unfeasible_tail_behind_a_stream_run_gpu(hip_lib, oracle_lib):
    """On the device the skip of that many jobs is two bulk passes that need every wave of the control workgroup — the engine wave must not be live (round 5: coldS said
    it never was, and this round hung the kernel; profiles/r05y_bulk_skip_hang.txt)."""
    r, st = both(hip_lib, oracle_lib, _long_unfeasible_tail_workload())
    assert st["stream_runs"] > 0 and len(r.scheduled) > 300


@pytest.mark.parametrize("lag", [1, 2, 3])
def test_stream_rounds_with_a_lagging_engine(hostsim_lib, oracle_lib, lag, monkeypatch):
    """HS_RING_LAG (tests/hostsim/fast_serial.h): the serial node engine falls behind the merge by a pseudo-random number of ring entries, as the engine wave does on the
    device, instead of serving every entry the moment it is staged — the end of a run, the nested events around gangs, a member without a node and the accounting then
    meet entries that were emitted but not yet placed.  Same rounds as the oracle's, whatever the lag."""
    monkeypatch.setenv("HS_RING_LAG", str(lag))
    emitted = jobs = 0
    for wl in (workload(900, occupied=0.2), workload(903, occupied=0.5, lookback=400),
               workload(977, gangs=400, occupied=0.3, n_nodes=600, n_jobs=9000, n_queues=12),
               _soak_round_workload(100345), _soak_round_workload(102465), _long_unfeasible_tail_workload()):
        r, st = both(hostsim_lib, oracle_lib, wl)
        emitted += st["stream_emitted"]; jobs += st["stream_jobs"]
    assert jobs > 0 and emitted > jobs      # entries were emitted ahead of the engine and taken back
