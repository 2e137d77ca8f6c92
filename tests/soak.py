#!/usr/bin/env python3
"""Differential soaks, CPU only: the oracle against the CPU build of the device code (tests/hostsim) on many more seeds than the suite runs.

    python tests/soak.py rounds 3000        # workloads.small_random rounds, random sizes / occupancy / gangs / bursts / away / ragged
    python tests/soak.py streams 1000        # medium rounds (<= 1500 nodes, <= 20000 jobs) dominated by stream runs and gangs through the ring; HS_STREAM_EAGER=1: a run wherever one can start
    python tests/soak.py preempt 1500       # small crowded rounds (60-100 % occupied), most with a fair-share preemption rate limit: the jobs that need preemption stay in the fast loop (fastPreemptIter)
    python tests/soak.py offgrid 3000       # ONE node type with requests and / or allocatable off the index grid (fast structure on next to literal iteration; narrow order-key layout), crowded, some with the default indexedResources
    python tests/soak.py optimiser 2000     # rounds with the experimental fairness optimiser on (asched_set_optimiser), incl. gangs only it can place
    python tests/soak.py market 2000        # market-driven rounds (asched_set_market): tied and distinct bids, gangs, away types, rate limits
    python tests/soak.py away 1500          # crowded rounds with a third of the running jobs cross-pool away jobs and "<queue>-away" contexts (tests/test_z_cross_pool_away.py)
    python tests/soak.py wide 500           # more than 64 queues: wide runs (round_wide.h) — 65 ... 400 queues, evicted jobs returning, gangs, bursts, lookback limits
    python tests/soak.py features 400       # tests/test_z_feature_mix.py rounds (affinity, conditional away, extra column, limits ...)
    python tests/soak.py ops 3000           # tests/test_z_nodedb_op_sequences.py NodeDb-level operation sequences
    python tests/soak.py submitcheck 400    # batched SubmitChecker vs the literal sequential restatement (and 3-entry cache)
    python tests/soak.py fit 600            # fit_select_batch at every priority on occupied NodeDbs
    python tests/soak.py sharded 1000       # ONE pool's round on 2-3 replicas with its wide passes split and all-reduced (asched_shard_round; SOAK_SHARD=direct on a GPU box: asched_shard_peers)
    python tests/soak.py excluded 1000      # NumExcludedNodesByReason (asched_excluded_nodes) of every failed selection of a round, incl. literal rows / away types / off-grid requests

HS_RING_LAG=<seed> (CPU build): the serial node engine lags behind the merge by a pseudo-random number of ring entries, as on the device (tests/hostsim/fast_serial.h).
SOAK_LIB=hip (on a GPU box): the same seeds through the product library instead of the CPU build.
Prints one line per divergence and a summary; exit code 1 if anything diverged.  (Round 1: all clean after the submit-check fix.)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tests/: it drives the oracle, which only test code may do)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from armada_amd import workloads as W  # noqa: E402
from armada_amd.binding import Library, SchedError  # noqa: E402


def libs():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if os.environ.get("SOAK_LIB") == "hip":   # on a GPU box: the product library itself against the oracle (torch's HIP runtime first: tests/conftest.py)
        import torch
        torch.cuda.init()
        import armada_amd
        return Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_"), armada_amd.load_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
    return Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_"), Library(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"), "asched_")


def sharded_rounds(lib, wl, fp, world):
    """`world` replicas of one pool as threads of this process; the all-reduce of a sharded pass is a barrier + a fold over the replicas' words (host memory in both builds:
    the CPU build calls synchronously, the HIP library's proxy passes ASCHED_ALLREDUCE_HOST_WORDS); SOAK_SHARD=direct: the GPU-to-GPU exchange instead"""
    import ctypes, threading
    direct = os.environ.get("SOAK_SHARD") == "direct"
    hs = [W.load(lib, wl) for _ in range(world)]
    bar = threading.Barrier(world, timeout=120)
    slots = [None] * world

    def make(rank):
        def allreduce(ptr, count, op):
            arr = np.ctypeslib.as_array((ctypes.c_int64 * count).from_address(ptr))
            slots[rank] = arr.copy()
            bar.wait()
            stack = np.stack(slots)
            red = stack.sum(axis=0) if (op & 15) == 0 else stack.min(axis=0) if (op & 15) == 1 else stack.max(axis=0)
            bar.wait()
            arr[:] = red
            return 0
        return allreduce
    if direct:
        areas = [h.shard_area()[0] for h in hs]
        for r, h in enumerate(hs):
            h.shard_peers(areas, r)
    else:
        for r, h in enumerate(hs):
            h.comm_init_external(make(r), r, world); h.shard_round(True)
    got, err = [None] * world, [None] * world

    def run(i):
        try:
            hs[i].set_deadline(100.0)
            W.prepare(hs[i], wl, fairshare_preemption_tokens=fp)
            got[i] = hs[i].schedule_round()
        except Exception as e:   # noqa: BLE001
            err[i] = e; print('replica', i, 'failed:', repr(e)[:300], flush=True); bar.abort()
    th = [threading.Thread(target=run, args=(i,)) for i in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=300)
    alive = any(t.is_alive() for t in th)
    for h in hs:
        if not alive: h.close()
    assert not alive, "a sharded round hung"
    for e in err:
        if e is not None: raise e
    return got


def main():
    kind, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200
    orc, hs = libs()
    import scenario
    bad, refused, t0 = 0, 0, time.time()
    first = int(os.environ.get("SOAK_FIRST", "0"))   # SOAK_FIRST=k: start at seed 100000 + k; SOAK_PROGRESS=1: print every seed before it runs (a hang names its seed)
    for seed in range(100_000 + first, 100_000 + n):
        if os.environ.get("SOAK_PROGRESS"): print("seed", seed, flush=True)
        try:
            if kind == "rounds":
                rng = np.random.default_rng(seed)
                wl = W.small_random(n_nodes=int(rng.integers(4, 300)), n_jobs=int(rng.integers(50, 6000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                                    occupied=float(rng.choice([0.0, 0.3, 0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 12)),
                                    burst=None if rng.random() < 0.4 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                                    away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.2))
                res = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl); res.append(s.schedule_round())
                scenario.assert_same_round(res[0], res[1])
            elif kind == "away":
                import test_z_cross_pool_away as A
                wl = A.with_away(seed, prefer_home=seed % 2 == 0, frac=[0.1, 0.3, 0.6][seed % 3])
                res = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl); res.append(s.schedule_round()); s.close()
                scenario.assert_same_round(res[0], res[1])
            elif kind == "preempt":
                rng = np.random.default_rng(seed)
                wl = W.small_random(n_nodes=int(rng.integers(4, 200)), n_jobs=int(rng.integers(50, 4000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                                    occupied=float(rng.choice([0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 8)),
                                    burst=None if rng.random() < 0.5 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                                    away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.2))
                fp = None if rng.random() < 0.3 else float(rng.choice([0, 1, 3, 10, 40]))
                res = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp); res.append(s.schedule_round()); s.close()
                scenario.assert_same_round(res[0], res[1])
            elif kind == "offgrid":
                rng = np.random.default_rng(seed)
                if seed % 5 == 4:
                    wl = W.default_indexed(n_nodes=int(rng.integers(4, 120)), n_jobs=int(rng.integers(50, 3000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                                           occupied=float(rng.choice([0.3, 0.7, 0.95, 1.0])), aligned=bool(rng.random() < 0.7))
                    if rng.random() < 0.5:
                        wl.global_burst, wl.queue_burst = wl.num_jobs, wl.num_jobs
                else:
                    wl = W.small_random(n_nodes=int(rng.integers(4, 200)), n_jobs=int(rng.integers(50, 4000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                                        occupied=float(rng.choice([0.3, 0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 8)),
                                        burst=None if rng.random() < 0.5 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                                        away=bool(rng.random() < 0.2), offgrid=int(rng.integers(1, 4)))
                fp = None if rng.random() < 0.6 else float(rng.choice([0, 1, 3, 10, 40]))
                res = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp); res.append(s.schedule_round()); s.close()
                scenario.assert_same_round(res[0], res[1])
            elif kind == "optimiser":   # rounds with the experimental fairness optimiser on (tests/test_z_optimiser_round.py): random and gang-placing shapes alternate
                import test_z_optimiser_round as T
                u, e_, r_ = T.compare(hs, orc, [seed], case=T._gang_case if seed % 2 else None)
                assert r_ == 0, "refused"
            elif kind == "market":   # market-driven rounds (tests/test_z_market_round.py): spot price, billing and overrides compared as well
                import test_z_market_round as T
                T.compare(hs, orc, [seed])
            elif kind == "streams":   # medium rounds that spend most of their time in stream runs / the gang ring (HS_STREAM_EAGER=1 python tests/soak.py streams N: a run wherever one can start)
                rng = np.random.default_rng(seed)
                nn, nj, nq = int(rng.integers(100, 1500)), int(rng.integers(2000, 20000)), int(rng.integers(2, 40))
                wl = W.config3(seed=seed, n_nodes=nn, n_jobs=nj, n_queues=nq, gangs=int(rng.choice([0, 0, 5, 50])), occupied=float(rng.choice([0.2, 0.5, 0.8, 0.93])))
                wl.global_burst = int(rng.choice([nj, nj // 3, 500])); wl.queue_burst = int(rng.choice([nj, max(10, nj // nq), 64])); wl.rate_inf = bool(rng.random() < 0.2)
                if rng.random() < 0.3:
                    wl.config.max_queue_lookback = int(rng.choice([50, 500, 3000]))
                res = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl); res.append(s.schedule_round()); s.close()
                scenario.assert_same_round(res[0], res[1])
            elif kind == "wide":   # more than 64 queues (round_wide.h): medium rounds, 65 ... 400 queues, evicted jobs returning, gangs, bursts, lookback limits
                rng = np.random.default_rng(seed)
                nn, nj, nq = int(rng.integers(60, 1200)), int(rng.integers(1500, 16000)), int(rng.choice([65, 70, 100, 130, 200, 257, 400]))
                wl = W.config3(seed=seed, n_nodes=nn, n_jobs=nj, n_queues=nq, gangs=int(rng.choice([0, 0, 0, 5, 40])), occupied=float(rng.choice([0.2, 0.5, 0.8, 0.93])))
                wl.global_burst = int(rng.choice([nj, nj // 3, 500])); wl.queue_burst = int(rng.choice([nj, max(10, 4 * nj // nq), 16])); wl.rate_inf = bool(rng.random() < 0.2)
                if rng.random() < 0.3:
                    wl.config.max_queue_lookback = int(rng.choice([20, 200, 3000]))
                res = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl); res.append(s.schedule_round())
                    if lib is hs and os.environ.get("SOAK_STATS"):
                        st = s.round_stats(); print(seed, nq, {k: st[k] for k in ("fast_iterations", "generic_iterations", "stream_runs", "stream_jobs")}, len(res[-1].scheduled), len(res[-1].preempted))
                    s.close()
                scenario.assert_same_round(res[0], res[1])
            elif kind == "features":
                import test_z_feature_mix as T
                scenario.assert_same_round(T.run(orc, seed), T.run(hs, seed))
            elif kind == "ops":
                import test_z_nodedb_op_sequences as T
                assert T.run_ops(orc, seed) == T.run_ops(hs, seed), "operation traces differ"
            elif kind == "submitcheck":
                import submitcheck_harness as H
                import test_zzz_submitcheck as T
                c = T.random_case(seed)
                H.same_results(H.literal_check(orc, c), H.batched_check(hs, c))
                H.same_results(H.literal_check(orc, c, cache_size=3), H.batched_check(hs, c, cache_size=3))
            elif kind == "excluded":   # NumExcludedNodesByReason of every job whose node selection failed (tests/test_z_excluded_nodes.py): random rounds incl. literal rows, away types, off-grid requests
                import test_z_excluded_nodes as X
                rng = np.random.default_rng(seed)
                wl = W.small_random(n_nodes=int(rng.integers(4, 200)), n_jobs=int(rng.integers(50, 1500)), n_queues=int(rng.integers(1, 10)), seed=seed,
                                    occupied=float(rng.choice([0.0, 0.3, 0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 8)),
                                    burst=None if rng.random() < 0.7 else (int(rng.integers(10, 500)), int(rng.integers(5, 100))),
                                    away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.3), offgrid=int(rng.choice([0, 0, 1, 3])))
                if seed % 4 == 3:   # (round 6) urgency preemption off, fair-share preemption on or off: selections that pass the gate and still end without a node
                    wl.config.disable_urgency_scheduling = True
                    wl.config.disable_fairshare_scheduling = bool(seed % 8 == 7)
                X.check_against_oracle(hs, orc, wl)
            elif kind == "fit":
                rng = np.random.default_rng(seed)
                wl = W.small_random(n_nodes=int(rng.integers(4, 200)), n_jobs=int(rng.integers(50, 2000)), n_queues=int(rng.integers(1, 8)), seed=seed,
                                    occupied=float(rng.choice([0.0, 0.5, 0.9])), gangs=0, away=bool(rng.random() < 0.3))
                out = []
                for lib in (orc, hs):
                    s = W.load(lib, wl); W.prepare(s, wl)
                    queued = [int(j) for q in wl.queued for j in q][:400]
                    out.append({p: s.fit_select_batch(queued, p).tolist() for p in s.priorities})
                assert out[0] == out[1], "fit_select_batch differs"
            elif kind == "sharded":   # ONE pool's round on 2-3 replicas (threads of this process), wide passes split + all-reduced (asched_shard_round; SOAK_SHARD=direct on a GPU box: asched_shard_peers)
                if os.environ.get("SOAK_LIB") != "hip":
                    raise SystemExit("soak.py sharded runs the replicas as threads: the product library only (SOAK_LIB=hip on a GPU box); CPU build: tests/soak_sharded_worker.py under torch.distributed.run")
                rng = np.random.default_rng(seed)
                wl = W.small_random(n_nodes=int(rng.integers(4, 300)), n_jobs=int(rng.integers(50, 4000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                                    occupied=float(rng.choice([0.3, 0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 8)),
                                    burst=None if rng.random() < 0.5 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                                    away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.2), offgrid=int(rng.choice([0, 0, 0, 3])))
                fp = None if rng.random() < 0.4 else float(rng.choice([0, 1, 3, 10, 40]))
                if os.environ.get("SOAK_SHARD_MEDIUM"):   # medium crowded pools: stream runs, gangs through the ring and the fast preempting iteration next to the split passes
                    wl = W.config3(n_nodes=int(rng.integers(300, 2500)), n_jobs=int(rng.integers(3000, 25000)), n_queues=int(rng.integers(2, 33)), seed=seed,
                                   occupied=float(rng.choice([0.5, 0.9, 0.95, 0.98])), gangs=int(rng.choice([0, 0, 20, 100])))
                    wl.global_burst, wl.queue_burst = max(1, wl.num_jobs // int(rng.integers(4, 12))), max(1, wl.num_jobs // int(rng.integers(20, 80)))
                    fp = None
                s = W.load(orc, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp); want = s.schedule_round(); s.close()
                for got in sharded_rounds(hs, wl, fp, 2 + seed % 2):
                    scenario.assert_same_round(want, got)
            else:
                raise SystemExit(__doc__)
        except SchedError as e:
            if e.code != -2:   # ASCHED_ERR_UNSUPPORTED: a documented refusal (e.g. fit_select_batch on literal-iteration rows)
                bad += 1; print("seed", seed, e)
            else:
                refused += 1
        except AssertionError as e:
            bad += 1; print("seed", seed, str(e)[:300])
    print(f"{kind}: {n} seeds, {bad} divergences, {refused} refused (ASCHED_ERR_UNSUPPORTED), {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
