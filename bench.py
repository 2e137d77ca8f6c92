#!/usr/bin/env python3
"""bench.py — scheduling rounds/sec (+ p99 round latency) of the MI355X scheduling-round hot path.

A "step" is one scheduling round = one PreemptingQueueScheduler.Schedule-equivalent call
(`asched_schedule_round`, phases 1-6 of SURVEY.md §3.2) on an already built NodeDb + SchedulingContext — the region
the reference's BenchmarkPreemptingQueueScheduler times (preempting_queue_scheduler_test.go:2762-2796).  The round
input (nodes, jobs, queues) is resident in HBM when the timed region starts; `round_prepare` (the reference's
newFairSchedulingAlgoContext analogue: fresh NodeDb, bind running jobs, fair shares) runs before every step, untimed.

Workload at N=1: BASELINE.json configs[2] — 100k nodes x 64 queues x 1M queued jobs, DRF weights + rate limits
(the configuration the metric "100k nodes x 1M jobs" is quoted on).  Smaller sizes via --nodes/--jobs/--queues.

Multi-GPU (--gpus N under torch.distributed.run): the path shards by *pool* — the reference builds one NodeDb per
pool and schedules pools one after the other (scheduling_algo.go:165); here pool g lives on GPU g, no data-path
collective (DESIGN.md "Multi-GPU").  Each rank schedules its own pool of the full per-GPU size (weak scaling),
value = pools-rounds of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(n_nodes, R, queries, binds, fair_scan_rows=0):
    """one node-selection query must examine, for every candidate node, its allocatable vector at one
    priority level and its order key: N x (R x 8 + 8) B; a bind is a read-modify-write of <= P buckets (~256 B)."""
    return queries * n_nodes * (R * 8 + 8) + binds * 256 + fair_scan_rows * (R * 8 + 16)


def kernel_isa_hash():
    """sha256 (16 hex digits) of the disassembled round-kernel code object of the library under test (tools/kcontrol_isa_hash.sh): ties the line to the rocprof / PMC files
    under profiles/, which carry the same hash when they were taken from the same code"""
    import re
    import subprocess
    try:
        lib = os.environ.get("ASCHED_LIB_PATH") or os.path.join(ROOT, "armada_amd", "csrc", "libarmada_sched.so")
        txt = subprocess.run(["bash", os.path.join(ROOT, "tools", "kcontrol_isa_hash.sh"), lib], capture_output=True, text=True, timeout=120).stdout
        m = re.search(r"sha256 ([0-9a-f]{16})", txt)
        return m.group(1) if m else None
    except Exception:
        return None


def round_diff(a, b):
    """fields of two RoundResults that differ (tests/scenario.assert_same_round semantics: job->node, priorities, methods, preempted,
    per-queue allocation by priority class, per-job reasons, fair shares and token counts with tolerance zero)"""
    import numpy as np
    bad = []
    for f in ("scheduled", "scheduled_priority", "scheduled_method", "preempted", "termination_reason", "num_evicted_phase1", "num_evicted_phase3",
              "global_tokens_after"):
        if getattr(a, f) != getattr(b, f):
            bad.append(f)
    for f in ("queue_allocated_by_pc", "job_unschedulable_reason", "fair_share", "demand_capped_adjusted_fair_share", "uncapped_adjusted_fair_share",
              "queue_tokens_after"):
        x, y = getattr(a, f), getattr(b, f)
        if x.shape != y.shape or not bool(np.array_equal(x, y)):
            bad.append(f)
    return bad


def parity_record(gpu_res, oracle_res, num_jobs, what):
    bad = round_diff(oracle_res, gpu_res)
    return {"checked": True, "identical": not bad, "jobs": int(num_jobs), "scheduled": len(gpu_res.scheduled), "preempted": len(gpu_res.preempted),
            "differing_fields": bad, "against": what}


def _pin_one_core():
    """pin the calling thread to ONE host core for a CPU-oracle leg (core 2 when the box has it); returns (previous mask, core) or None"""
    try:
        old = os.sched_getaffinity(0)
        core = 2 if 2 in old else sorted(old)[0]
        os.sched_setaffinity(0, {core})
        return old, core
    except (AttributeError, OSError):
        return None


def _unpin(pinned):
    if pinned:
        try:
            os.sched_setaffinity(0, pinned[0])
        except OSError:
            pass


def cut_possible(wl, budget_s, full_iters):
    """would cpu_baseline cut the burst for this budget?  (then its sample is an extrapolation and one round of it is enough)"""
    est = full_iters * 170e-6 * (wl.num_nodes / 100_000.0) ** 0.5
    return est > budget_s and not wl.rate_inf


def cpu_baseline(wl, budget_s, full_iters, min_rounds=1):
    """The CPU oracle (C++ restatement of the reference algorithm, oracle/) timed on this box's host cores, 1 thread
    (the reference's round is single-goroutine).  Bounded sample: ONE round on the SAME nodes/jobs/queues (SURVEY 8d asks for >= 30
    rounds; one oracle round of BASELINE configs[2] is ~1 minute, so the sample is one round and says so).  When the estimate exceeds
    the budget the global burst is cut; the eviction pass and the loop over the evicted jobs do not shrink with the burst, so the
    iteration-ratio scaling then OVERSTATES the CPU time: the record is marked `upper_bound_extrapolation` and carries no parity check.
    Returns (record, oracle RoundResult or None when the burst was cut)."""
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    import copy
    import numpy as np
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        return None, None
    oracle = Library(path, "oracle_")
    s = W.load(oracle, wl)
    sample = copy.copy(wl)
    est = full_iters * 170e-6 * (wl.num_nodes / 100_000.0) ** 0.5   # measured: ~165 us per loop iteration at 100k nodes, ~40 us at 10k
    cut = False
    if est > budget_s and not wl.rate_inf:
        sample.global_burst = min(wl.global_burst, max(1, int(wl.global_burst * max(0.02, (budget_s / est)))))
        cut = sample.global_burst < wl.global_burst
    pinned = _pin_one_core()   # the single-threaded CPU leg pinned to one core (taskset -c 2)
    times, r = [], None
    for _ in range(3):         # THREE rounds when a round is at most ~13 s (the reduced legs, the production-defaults record), ONE when a round is a minute (round-4 review, weak #8: mean of 3 where it is affordable)
        W.prepare(s, sample)
        t0 = time.perf_counter(); r = s.schedule_round(); times.append(time.perf_counter() - t0)
        if len(times) >= min_rounds and (sum(times) + times[-1] > min(40.0, max(budget_s, 0.0)) or times[-1] > 13.5):   # (min_rounds: the headline takes three rounds whatever a round costs — round-5 review: a one-sample baseline has no variance)
            break
    _unpin(pinned)
    dt = float(np.mean(times))
    iters = max(1, r.num_loop_iterations)
    scaled = dt * (full_iters / iters) if (cut and full_iters > iters) else dt
    excluded = {}
    if not cut:   # NumExcludedNodesByReason of a sample of the jobs whose node selection failed (asched_excluded_nodes): compared with the GPU round's by the caller
        from armada_amd.binding import SchedError
        reasons = np.asarray(r.job_unschedulable_reason)
        fit = np.nonzero((reasons == 10) | (reasons == 11))[0]   # ASCHED_REASON_GANG_DOES_NOT_FIT / JOB_DOES_NOT_FIT: the ones a node selection failed for
        for j in list(fit[:EXCLUDED_SAMPLE]) + list(np.nonzero(reasons)[0][:EXCLUDED_SAMPLE // 8]):
            try:
                excluded[int(j)] = s.excluded_nodes(int(j))
            except SchedError:
                excluded[int(j)] = "not produced"
    s.close()
    rec = {"value": 1.0 / scaled, "unit": "rounds/s", "cores": 1, "kind": "port",
           "sample": f"oracle (C++ restatement of the reference algorithm, 1 thread) on the same {wl.num_nodes}-node/{wl.num_jobs}-job input, {len(times)} round(s) "
                     f"(not the >= 30 of SURVEY 8d: a round is {dt:.1f} s of CPU) with global burst {sample.global_burst} ({iters} of {full_iters} loop iterations, mean {dt:.2f} s measured"
                     + (", scaled by the iteration ratio: an upper bound of the CPU time, the evicted-job phases do not shrink with the burst)" if cut else ")"),
           "measured_s": dt, "measured_min_s": float(min(times)), "measured_max_s": float(max(times)), "measured_rounds_s": [round(t, 4) for t in times], "measured_iterations": iters, "rounds": len(times), "upper_bound_extrapolation": bool(cut),
           "pinned_core": pinned[1] if pinned else None}
    if not cut:
        r.excluded_sample = excluded
    return rec, (None if cut else r)


EXCLUDED_SAMPLE = 128


def excluded_parity(s, oracle_res):
    """the exclusion histograms (why a job found no node) of the oracle round's sample against the GPU handle's records of the same round"""
    from armada_amd.binding import SchedError
    exp = getattr(oracle_res, "excluded_sample", None)
    if not exp:
        return {"checked": False}
    bad, with_record = [], 0
    for j, h in exp.items():
        try:
            g = s.excluded_nodes(j)
        except SchedError:
            g = "not produced"
        with_record += bool(h) and h != "not produced"
        if g != h:
            bad.append(j)
    return {"checked": True, "identical": not bad, "jobs": len(exp), "with_a_failed_selection_on_record": int(with_record), "differing_jobs": bad[:8],
            "what": "NumExcludedNodesByReason (asched_excluded_nodes) of the first jobs with an unschedulable reason: GPU records vs the oracle's walk"}


# ------------------------------------------------------------------------------------------------ --submit-check
def _submit_check_helpers():
    import numpy as np
    from armada_amd import workloads as W
    from armada_amd.binding import Scheduler
    from armada_amd.submitcheck import PoolNodeDb

    class ArrayPoolDb(PoolNodeDb):
        """one pool's cleared NodeDb over array-shaped job tables (the shape a cgo shim would hand over)"""

        def __init__(self, lib, wl, req, pc, gang, gang_card):
            self.s = Scheduler(lib, wl.config)
            self.s.nodes_upsert(wl.node_total, wl.node_allocatable)
            self.s.clear_allocated()
            self.req, self.pc, self.gang, self.gang_card = req, pc, gang, gang_card
            self.launch_ms, self.stats = [], []

        def load_jobs(self, jobs):
            self.s.jobs_set(self.req, queue=np.zeros(len(self.req), np.int32), pc=self.pc, gang_id=self.gang, gang_cardinality=self.gang_card)

        def submit_check(self, units, strip_gang):
            out = self.s.submit_check(units, strip_gang)
            self.launch_ms.append(self.s.kernel_times()["submit_check_ms"])
            self.stats.append(self.s.submit_stats())
            return out

    def make_jobs(rng, n_jobs, n_shapes, gang_frac):
        cpu = rng.integers(1, 41, size=n_shapes) * 1000   # a fifth of the keys fit no node (32-core nodes)
        mem = rng.integers(1, 257, size=n_shapes) * W.Gi
        shape = rng.integers(0, n_shapes, size=n_jobs)
        req = np.zeros((n_jobs, W.R), np.int64)
        req[:, 0], req[:, 1] = mem[shape], cpu[shape]
        gang = np.full(n_jobs, -1, np.int32)
        card = np.ones(n_jobs, np.int32)
        i, g = 0, 0
        while i < n_jobs:
            if rng.random() < gang_frac:
                c = int(min(rng.integers(2, 17), n_jobs - i))
                gang[i:i + c], card[i:i + c] = g, c
                req[i:i + c] = req[i]          # uniform shape within a gang
                shape[i:i + c] = shape[i]
                g += 1; i += c
            else:
                i += 1
        return req, shape, gang, card
    return ArrayPoolDb, make_jobs


def submit_check_record(args):
    """Submit-check throughput: jobs of one SubmitChecker.Check call per second on one pool through the batched
    flow (armada_amd.submitcheck -> asched_submit_check), with the launch durations from the HIP events on the launch stream and, as
    `cpu_baseline`, the reference's sequential flow (one Txn / ScheduleManyWithTxn / Abort per job) timed on the CPU oracle over a
    bounded sample.  `roofline.achieved` prices every node query at N x (8R + 8) bytes like the round: one query per distinct scheduling
    key (answered by the wide fit kernel in a handful of passes for all keys together, `how`) and one per gang member (sequential path)."""
    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU path)"
    torch.zeros(1, device="cuda:0")
    import armada_amd
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    from armada_amd.submitcheck import PoolConfig, SubmitChecker, SubmitJob
    ArrayPoolDb, make_jobs = _submit_check_helpers()
    lib = armada_amd.load_library()
    rng = np.random.Generator(np.random.PCG64(W.SEED))
    n_jobs, n_shapes = args.submit_jobs, args.submit_keys
    wl = W.config2(n_nodes=args.nodes, n_jobs=1)          # the node set of BASELINE configs[1]/[2]: 32 cpu / 256 Gi nodes
    req, shape, gang, card = make_jobs(rng, n_jobs, n_shapes, 0.02)
    jobs = [SubmitJob(id=str(i), queue="q", priority_class="pc0", scheduling_key=(int(shape[i]),), request=req[i],
                      gang_id=None if gang[i] < 0 else f"g{gang[i]}") for i in range(n_jobs)]
    pc = np.zeros(n_jobs, np.int32)
    db = ArrayPoolDb(lib, wl, req, pc, gang, card)
    chk = SubmitChecker([PoolConfig("pool")], {"pool": db})
    chk.check(jobs[:64])                                  # warm-up (first launches, allocations)
    chk.update_executors()
    lat = []
    for _ in range(max(1, args.steps)):
        db.launch_ms.clear(); db.stats.clear(); chk.update_executors()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = chk.check(jobs)
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    dt = float(np.mean(lat))
    queries = len({j.scheduling_key for j in jobs}) + int((gang >= 0).sum())   # what the reference would ask one by one
    passes = sum(int(st.get("node_passes", 0)) for st in db.stats)              # walks over the node set the launches actually made (submit_stats)
    dev_s = sum(db.launch_ms) / 1e3
    priced = algorithmic_bytes(args.nodes, W.R, max(passes, 1), 0)
    line = {
        "metric": "submit checks/s (jobs of one SubmitChecker.Check call, one pool)", "value": n_jobs / dt, "unit": "jobs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"{args.nodes} nodes (32 cpu / 256 Gi), {n_jobs} submitted jobs, {n_shapes} scheduling keys, {int((gang >= 0).sum())} gang members, one pool"},
        "schedulable": sum(r.is_schedulable for r in res.values()), "native_calls": chk.launches, "how": db.stats, "device_s": dev_s,
        # identical queries batched into one pass count ONCE — the figure is priced on the passes the launches executed, not on the queries they answer
        "roofline": {"bound": "hbm", "achieved": priced / max(dev_s, 1e-9) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": priced / max(dev_s, 1e-9) / 1e9 / HBM_PEAK_GBS, "traffic": None, "node_queries": queries, "passes_executed": passes, "algorithmic_bytes_per_launch": priced,
                     "unbatched_equivalent_GBs": queries * algorithmic_bytes(args.nodes, W.R, 1, 0) / max(dev_s, 1e-9) / 1e9,
                     "kernel": "k_fit_batch (individual checks) + k_submit_gangs (gang units, one workgroup each; k_control_aux for what is left)"},
    }
    if args.cpu_budget > 0:   # cpu_baseline leg: the reference's sequential flow on the oracle, distinct keys so that its cache cannot help
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if os.path.exists(path):
            oracle = Library(path, "oracle_")
            sample = min(300, n_jobs)
            odb = ArrayPoolDb(oracle, wl, req[:sample], pc[:sample], np.full(sample, -1, np.int32), np.ones(sample, np.int32))
            odb.load_jobs(None)
            t1 = time.perf_counter()
            for i in range(sample):
                odb.s.txn_begin(); odb.s.schedule_many([i]); odb.s.txn_abort()
            cpu_dt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": sample / cpu_dt, "unit": "jobs/s", "cores": 1, "kind": "port",
                                    "sample": f"{sample} jobs, one Txn / ScheduleManyWithTxn / Abort each on the CPU oracle, same node set"}
            # parity: the whole Check call once more with the oracle behind the same host flow — every job's verdict (schedulable, reason text) compared
            fdb = ArrayPoolDb(oracle, wl, req, pc, gang, card)
            want = SubmitChecker([PoolConfig("pool")], {"pool": fdb}).check(jobs)
            same = set(want) == set(res) and all(want[k].is_schedulable == res[k].is_schedulable and want[k].reason == res[k].reason for k in want)
            line["parity"] = {"checked": True, "identical": bool(same), "jobs": int(n_jobs), "schedulable": sum(r.is_schedulable for r in want.values()),
                              "against": "the same SubmitChecker.Check call with the CPU oracle as the pool's NodeDb: per job is_schedulable and reason"}
    return line


def fit_batch_record(hip, args, big=False):
    """big=True: the same kernel at 100 000 nodes x 1 000 000 queries (round-3 review: the size at which its roofline fraction means something).
    BASELINE configs[1] ("nodedb fit kernel"): first feasible node for every queued job of the 10k-node x 100k-job x 8-queue workload against a
    fixed node state, no binding (n independent selectNodeForPodAtPriority calls, nodedb.go:840-879) through asched_fit_select_batch ->
    k_fit_batch; every node id compared with the oracle's index search.  Identical (shape, level) queries share one pass over the node tile, so
    `queries_issued` (one per job) and `passes_executed` (one per distinct scheduling-key shape) are both reported; the roofline is priced on
    the passes (count ONE query per batched pass), N x (8R + 8) bytes each."""
    import numpy as np
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    sc = args.other_scale
    wl = W.config2() if sc == 1.0 else W.config2(n_nodes=max(8, int(10_000 * sc)), n_jobs=max(64, int(100_000 * sc)))
    if big:
        wl = W.config2(n_nodes=max(8, int(100_000 * sc)), n_jobs=max(64, int(1_000_000 * sc)))
    s = W.load(hip, wl)
    W.prepare(s, wl)
    jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
    s.fit_select_batch(jobs[:64], -2)  # warm-up launch
    host, dev = [], []
    got = None
    for _ in range(max(3, min(args.steps, 10))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = s.fit_select_batch(jobs, -2)
        torch.cuda.synchronize(); host.append(time.perf_counter() - t0)
        dev.append(s.kernel_times()["fit_batch_ms"])
    shapes = len(np.unique(wl.job_req[jobs], axis=0))
    passes = 1   # ONE launch walks the node tile for every shape (a batched pass counts once; round 4 priced one pass per distinct shape and the review called it)
    dev_ms, host_ms = float(np.mean(dev)), float(np.mean(host)) * 1e3
    alg = algorithmic_bytes(wl.num_nodes, W.R, passes, 0)
    ach = alg / max(dev_ms * 1e-3, 1e-12) / 1e9
    rec = {"config": "nodedb fit kernel at 100 000 nodes x 1 000 000 queries" if big else "BASELINE configs[1]", "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {len(jobs)} queued jobs, R=4, K=3, nodedb fit kernel at priority -2",
           "metric": "first-fit queries/s (asched_fit_select_batch, host call incl. result download)", "value": len(jobs) / (host_ms * 1e-3), "unit": "queries/s",
           "host_ms": host_ms, "device_ms": dev_ms, "queries_issued": int(len(jobs)), "passes_executed": int(passes), "distinct_shapes": int(shapes), "fitting": int((got >= 0).sum()),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": "k_fit_batch",
                        "algorithmic_bytes_per_launch": alg, "unbatched_equivalent_GBs": algorithmic_bytes(wl.num_nodes, W.R, len(jobs), 0) / max(dev_ms * 1e-3, 1e-12) / 1e9,
                        "per_shape_equivalent_GBs": algorithmic_bytes(wl.num_nodes, W.R, shapes, 0) / max(dev_ms * 1e-3, 1e-12) / 1e9,
                        "note": "one launch answers every query; bytes = 1 pass x N x 40 B (the node tile is held in registers for all shapes): at this size the launch is latency-bound, not bandwidth-bound"}}
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if args.cpu_budget > 0 and os.path.exists(path):
        oracle = Library(path, "oracle_")
        o = W.load(oracle, wl)
        W.prepare(o, wl)
        sample = jobs if not big else jobs[:: max(1, len(jobs) // 50_000)]   # (big: every 20th query — the oracle answers ~1e5 queries/s at 100 000 nodes)
        t1 = time.perf_counter(); want = o.fit_select_batch(sample, -2); cpu_dt = time.perf_counter() - t1
        rec["cpu_baseline"] = {"value": len(sample) / cpu_dt, "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": f"{len(sample)} of the {len(jobs)} queries on the CPU oracle (ordered-index search per job), same state, {cpu_dt:.2f} s"}
        gsel = got if not big else got[:: max(1, len(jobs) // 50_000)]
        rec["parity"] = {"checked": True, "identical": bool((want == gsel).all()), "jobs": int(len(sample)), "against": "oracle fit_select_batch, node ids"}
        o.close()
    s.close()
    return rec


def optimiser_record(hip, args):
    """SURVEY 8f-3, the experimental fairness optimiser's node scoring: after a round on the preemption-heavy shape (20k nodes 95% occupied), every
    remaining queued job of a sample is scored against EVERY node (PreemptingNodeScheduler.Schedule per node, optimiser/gang_scheduler.go:100-141)
    by one k_opt_score_wave launch; selections, preemption lists and all per-node scores compared with the oracle's serial loop."""
    import numpy as np
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    sc = args.other_scale
    wl = W.config3(seed=W.SEED, n_nodes=max(16, int(20_000 * sc)), n_jobs=max(200, int(200_000 * sc)), n_queues=32 if sc == 1.0 else 4, occupied=0.95)
    wl.global_burst, wl.queue_burst = max(1, int(200_000 * 0.2 * sc)), max(1, int(20_000 * 0.2 * sc))
    wl.job_run_ts = (np.arange(wl.num_jobs, dtype=np.int64) * 7919 % 100003) * 1_000_000
    n_sample = 32

    def run(lib, timed):
        s = W.load(lib, wl); W.prepare(s, wl)
        r = s.schedule_round()
        done = set(int(j) for j in r.scheduled)
        left = [int(j) for j in np.nonzero((wl.job_node < 0) & (wl.job_gang < 0))[0] if int(j) not in done][:n_sample]
        out, host, dev = [], [], []
        for j in left:
            if timed: torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.optimiser_schedule_job(j, min_improvement_pct=5.0, now_ms=200_000)
            host.append(time.perf_counter() - t0)
            if timed: dev.append(s.kernel_times()["fit_batch_ms"])
            out.append(s.optimiser_schedule_job(j, min_improvement_pct=5.0, now_ms=200_000, per_node=True))   # untimed: every node's score for the parity verdict
        running = int((wl.job_node >= 0).sum()) + len(r.scheduled) - len(r.preempted)
        s.close()
        return out, host, dev, running

    got, host, dev, running = run(hip, True)
    R = W.R
    direct = float(np.mean([sum(1 for x in g["scores"] if x[0] and x[1] == 0) for g in got])) if got else 0.0
    walked_frac = 1.0 - direct / max(wl.num_nodes, 1)
    alg = wl.num_nodes * (8 * R + 8) + walked_frac * running * (4 + 4 + 4 + 4 + 4 + 8 + 8 * R) + wl.num_nodes * 24   # node row + static bit word; per job of a walked node: list entry, pc, gang, priority, queue, lease, request; the score record written
    dev_ms = float(np.mean(dev)) if dev else 0.0
    ach = alg / max(dev_ms * 1e-3, 1e-12) / 1e9
    rec = {"config": "fairness optimiser node scoring (SURVEY 8f-3)",
           "workload": f"{wl.num_nodes} nodes (95% occupied, {running} running jobs) x {len(got)} queued jobs left over by the round, each scored against every node",
           "metric": "jobs scored against all nodes per second (asched_optimiser_schedule_job: node -> jobs index, k_opt_score, selection, preemption list)",
           "value": len(got) / max(sum(host), 1e-12), "unit": "jobs/s", "host_ms_per_job": float(np.mean(host)) * 1e3 if host else None, "k_opt_score_ms": dev_ms,
           "nodes_needing_preemption_walk": walked_frac, "selected_with_preemption": sum(1 for g in got if g["preempted"]), "selected_without": sum(1 for g in got if g["node"] >= 0 and not g["preempted"]),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": "k_opt_score_wave", "algorithmic_bytes_per_launch": alg,
                        "note": "one wave per node, one lane per job on the node (rank sorts on broadcast entries, wave scan of the request vectors); the rows are gathered through the node -> jobs index, "
                                "so the kernel is bound by a few dependent gather round trips per wave, not by bandwidth, at this size"}}
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if args.cpu_budget > 0 and os.path.exists(path):
        want, chost, _, _ = run(Library(path, "oracle_"), False)
        rec["cpu_baseline"] = {"value": len(want) / max(sum(chost), 1e-12), "unit": "jobs/s", "cores": 1, "kind": "port",
                               "sample": f"the same {len(want)} jobs on the CPU oracle (serial loop over all nodes per job), {sum(chost):.2f} s"}
        rec["parity"] = {"checked": True, "identical": want == got, "jobs": len(got), "against": "oracle: selected node, preempted jobs in order, cost, impact and every node's score (floats bit-equal)"}
    return rec


def market_record(hip, args):
    """one market-driven round (asched_set_market: evict-everything node evictor, price-ordered iterators, MarketIteratorPQ as a literal heap, spot price /
    billing) and the indicative gang pricer behind it (asched_price_gang) on a mid-size pool; the oracle on the same input for the baseline and the parity verdict.  The
    market round is the generic path in ONE launch of the auxiliary kernel: this line is its price, not a tuned number."""
    import numpy as np
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    sc = args.other_scale
    wl = W.config3(seed=W.SEED, n_nodes=max(16, int(4_000 * sc)), n_jobs=max(200, int(40_000 * sc)), n_queues=16 if sc == 1.0 else 4, occupied=0.8)
    wl.global_burst, wl.queue_burst = max(1, int(3_000 * sc)), max(1, int(750 * sc))   # (the oracle: ~17 s for the 827 000 evicted jobs + ~7 s per 1 000 new jobs at this size)
    rng = np.random.default_rng(W.SEED)
    bids = rng.integers(1, 9, size=wl.num_jobs).astype(np.float64)                     # eight price bands (pkg/bidstore)
    nonpre = np.array([not wl.config.pc_preemptible[p] for p in wl.job_pc])
    bids[(wl.job_node >= 0) & nonpre] = 1_000_000.0                                     # pricing.NonPreemptibleRunningPrice
    pcp = np.asarray(wl.config.pc_priority)
    queued = [sorted(q, key=lambda j: (-int(pcp[wl.job_pc[j]]), -float(bids[j]), int(j))) for q in wl.queued]   # jobdb.PriceOrder
    nq = wl.num_queues

    def run(lib, timed):
        s = W.load(lib, wl); W.set_jobs(s, wl, bid_price=bids)
        s.round_prepare(wl.queue_weight, queued, global_tokens=float(wl.global_burst), global_burst=wl.global_burst, global_rate_inf=wl.rate_inf,
                        queue_tokens=[float(wl.queue_burst)] * nq, queue_burst=[wl.queue_burst] * nq, queue_rate_inf=[wl.rate_inf] * nq)
        s.set_market(True, 0.9)
        if timed: torch.cuda.synchronize()
        t0 = time.perf_counter(); res = s.schedule_round(); dt = time.perf_counter() - t0
        mr = s.market_result()
        left = [int(j) for j in np.nonzero((wl.job_node < 0) & (wl.job_gang < 0))[0] if int(j) not in res.scheduled][:16]
        t1 = time.perf_counter(); prices = [s.price_gang([j], now_ms=300_000) for j in left]; dp = time.perf_counter() - t1
        kms = s.kernel_times()["fit_batch_ms"] if timed else 0.0
        s.close()
        return res, mr, prices, dt, dp, kms

    res, mr, prices, dt, dp, kms = run(hip, True)
    rec = {"config": "market-driven round + indicative gang pricer (SURVEY 8f-4)",
           "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {sum(len(q) for q in queued)} queued jobs (+{int((wl.job_node >= 0).sum())} running, 80% occupied), 8 price bands, spot price cutoff 0.9, global burst {wl.global_burst}",
           "metric": "market-driven scheduling rounds/sec (one launch of the auxiliary kernel, generic path)", "value": 1.0 / dt, "unit": "rounds/s", "ms_per_step": dt * 1e3, "steps": 1,
           "round": {"scheduled": len(res.scheduled), "preempted": len(res.preempted), "evicted_phase1": res.num_evicted_phase1, "loop_iterations": res.num_loop_iterations,
                     "spot_price": mr["spot_price"], "queues_with_price_override": sum(v is not None for v in mr["price_override"])},
           "pricer": {"gangs_priced": len(prices), "schedulable": sum(p["schedulable"] for p in prices), "ms_per_gang": dp / max(len(prices), 1) * 1e3, "k_price_score_ms": kms},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": "k_control_aux (CMD_MARKET_ROUND)",
                        "note": "a sequential heap-ordered round on one workgroup: latency bound by construction, no bandwidth figure is claimed"}}
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if args.cpu_budget > 0 and os.path.exists(path):
        ores, omr, oprices, odt, odp, _ = run(Library(path, "oracle_"), False)
        rec["cpu_baseline"] = {"value": 1.0 / odt, "unit": "rounds/s", "cores": 1, "kind": "port", "sample": f"the same round on the CPU oracle, {odt:.2f} s; pricer {odp / max(len(oprices), 1) * 1e3:.1f} ms per gang"}
        same = (res.scheduled == ores.scheduled and res.preempted == ores.preempted and (res.queue_allocated_by_pc == ores.queue_allocated_by_pc).all() and mr["spot_price"] == omr["spot_price"]
                and (mr["billable"] == omr["billable"]).all() and mr["price_override"] == omr["price_override"] and prices == oprices)
        rec["parity"] = {"checked": True, "identical": bool(same), "jobs": wl.num_jobs, "against": "oracle: scheduled / preempted jobs and nodes, queue accounting, spot price, billable resources, price overrides, gang prices"}
    return rec


def two_word_record(hip, args):
    """a pool whose order key needs more than 64 bits (five indexed resources at fine resolutions: 80 bits of fields + the node-index rank) — served since round 6 by the
    generic path of k_control_wk with a two-word key.  The price of exactness where rounds 1-5 answered ASCHED_ERR_UNSUPPORTED, not a tuned number."""
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    sc = args.other_scale
    wl = W.fine_indexed(n_nodes=max(64, int(20_000 * sc)), n_jobs=max(400, int(100_000 * sc)), n_queues=32 if sc == 1.0 else 4, occupied=0.5)
    s = W.load(hip, wl)
    times = []
    for _ in range(2):
        W.prepare(s, wl); torch.cuda.synchronize()
        t0 = time.perf_counter(); res = s.schedule_round(); times.append(time.perf_counter() - t0)
    st = s.round_stats(); s.close()
    dt = times[-1]
    rec = {"config": "two-word order keys (SURVEY a4)", "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {sum(len(q) for q in wl.queued)} queued jobs, indexed gpu @1, cpu @1m, memory @1Mi, "
                     "ephemeral-storage @1Mi, nvme @1Gi: the key needs 80 bits of fields + the node-index rank",
           "metric": "scheduling rounds/sec on a pool with a 128-bit order key (generic path of k_control_wk)", "value": 1.0 / dt, "unit": "rounds/s", "ms_per_step": dt * 1e3, "steps": 1,
           "round": {"scheduled": len(res.scheduled), "preempted": len(res.preempted), "loop_iterations": res.num_loop_iterations, "generic_iterations": int(st.get("generic_iterations", 0)),
                     "fast_iterations": int(st.get("fast_iterations", 0)), "kclk_plane_scans": int(st.get("kclk_plane_scans", 0)), "kclk_pass1": int(st.get("kclk_pass1", 0))},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": "k_control_wk",
                        "note": "the generic path: two plane passes per selection (high word, low word), latency bound; no bandwidth figure is claimed"}}
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if args.cpu_budget > 0 and os.path.exists(path):
        o = W.load(Library(path, "oracle_"), wl); W.prepare(o, wl)
        t0 = time.perf_counter(); ores = o.schedule_round(); odt = time.perf_counter() - t0; o.close()
        rec["cpu_baseline"] = {"value": 1.0 / odt, "unit": "rounds/s", "cores": 1, "kind": "port", "sample": f"the same round on the CPU oracle, {odt:.2f} s"}
        rec["parity"] = parity_record(res, ores, wl.num_jobs, "oracle round on the same input")
    return rec


def coarse_divisor_record(hip, args):
    """SURVEY a4, the other half: the same fine index resolutions (cpu @1m, memory @1Mi, ephemeral-storage @1Mi: 79 bits of key at face value) with pods that ask in quarter cores /
    100Mi / GiB steps — every column value is a multiple of 250m / 32Mi / 2Gi, the key divides by those exactly (asched_host.inc layoutKeys) and fits one word: the fast path."""
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    sc = args.other_scale
    wl = W.fine_indexed(n_nodes=max(64, int(20_000 * sc)), n_jobs=max(400, int(200_000 * sc)), n_queues=64 if sc == 1.0 else 4, occupied=0.5, k5=False, coarse=True)
    s = W.load(hip, wl)
    times = []
    for _ in range(3):
        W.prepare(s, wl); torch.cuda.synchronize()
        t0 = time.perf_counter(); res = s.schedule_round(); times.append(time.perf_counter() - t0)
    st = s.round_stats(); s.close()
    dt = min(times[1:])
    rec = {"config": "fine resolutions, round requests: coarser exact key divisor (SURVEY a4)", "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {sum(len(q) for q in wl.queued)} queued jobs, "
                     "indexed gpu @1, cpu @1m, memory @1Mi, ephemeral-storage @1Mi; requests in quarter cores / 100Mi / GiB",
           "metric": "scheduling rounds/sec on a fine-resolution pool whose key fits one word through the coarser exact divisor (fast path)", "value": 1.0 / dt, "unit": "rounds/s", "ms_per_step": dt * 1e3, "steps": 2,
           "round": {"scheduled": len(res.scheduled), "preempted": len(res.preempted), "loop_iterations": res.num_loop_iterations, "generic_iterations": int(st.get("generic_iterations", 0)),
                     "fast_iterations": int(st.get("fast_iterations", 0)), "stream_jobs": int(st.get("stream_jobs", 0))},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": "k_control",
                        "note": "the headline's kernel on another key layout: see the headline's roofline; no separate figure is claimed"}}
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if args.cpu_budget > 0 and os.path.exists(path):
        o = W.load(Library(path, "oracle_"), wl); W.prepare(o, wl)
        t0 = time.perf_counter(); ores = o.schedule_round(); odt = time.perf_counter() - t0; o.close()
        rec["cpu_baseline"] = {"value": 1.0 / odt, "unit": "rounds/s", "cores": 1, "kind": "port", "sample": f"the same round on the CPU oracle, {odt:.2f} s"}
        rec["parity"] = parity_record(res, ores, wl.num_jobs, "oracle round on the same input")
    return rec


def sharded_record(hip, args):
    """ONE pool's round on two replicas with its wide node passes split and exchanged GPU-to-GPU (asched_shard_peers) — here both replicas are handles of this process on
    the ONE GPU, their round kernels side by side (never run on two GPUs).  Compared with two whole rounds side by side on the same GPU."""
    import threading
    import numpy as np
    from armada_amd import workloads as W
    if str(getattr(hip, "path", "")).endswith("libhostsim.so"):   # (the contract test's CPU stand-in keeps its LDS stand-ins in process-wide variables: no two rounds at a time)
        return {"config": "one pool on two replicas, wide passes split (SURVEY 8e)", "skipped": "needs the HIP library (two round kernels side by side)"}
    sc = args.other_scale
    wl = W.config3(n_nodes=max(64, int(20_000 * sc)), n_jobs=max(400, int(200_000 * sc)), n_queues=32 if sc == 1.0 else 4, seed=W.SEED, occupied=0.95)
    wl.global_burst, wl.queue_burst = max(1, wl.num_jobs // 8), max(1, wl.num_jobs // 80)

    def pair(shard):
        hs = [W.load(hip, wl) for _ in range(2)]
        if shard:
            areas = [h.shard_area()[0] for h in hs]
            for r, h in enumerate(hs):
                h.shard_peers(areas, r)
        for h in hs:
            h.set_deadline(240.0)
        times, res = [[], []], [None, None]

        def run(i):
            for _ in range(2):
                W.prepare(hs[i], wl); t0 = time.perf_counter(); res[i] = hs[i].schedule_round(); times[i].append(time.perf_counter() - t0)
        th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
        for t in th: t.start()
        for t in th: t.join()
        st = hs[0].round_stats(); st["exchanges"] = hs[0].shard_exchanges() if shard else 0
        for h in hs: h.close()
        return res, max(times[0][-1], times[1][-1]), st
    r0, t0, st0 = pair(False)
    r1, t1, st1 = pair(True)
    same = all(np.array_equal(r0[0].scheduled_job, r.scheduled_job) and np.array_equal(r0[0].scheduled_node, r.scheduled_node) and np.array_equal(r0[0].preempted_job, r.preempted_job) for r in r1)
    return {"config": "one pool on two replicas, wide passes split (SURVEY 8e)", "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {sum(len(q) for q in wl.queued)} queued jobs, 95% occupied; "
                      "two replicas of the pool as two handles on this ONE GPU",
            "metric": "scheduling rounds/sec of a replica whose wide node passes are split two ways and exchanged GPU-to-GPU", "value": 1.0 / t1, "unit": "rounds/s", "ms_per_step": t1 * 1e3, "steps": 1,
            "two_whole_rounds_side_by_side_ms": t0 * 1e3, "exchanges_per_round": int(st1.get("exchanges", 0)), "kclk_plane_scans": [int(st0.get("kclk_plane_scans", 0)), int(st1.get("kclk_plane_scans", 0))],
            "round": {"scheduled": len(r1[0].scheduled_job), "preempted": len(r1[0].preempted_job)},
            "parity": {"checked": True, "identical": bool(same), "jobs": wl.num_jobs, "against": "the unsharded library's round on the same input (which the headline legs pin on the oracle)"},
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": "k_control_wk",
                         "note": "unmeasured on more than one GPU; on one GPU the two replicas share its HBM and its CUs"}}


def round_shape_record(hip, args, label, kwargs, steps, note, warmup=1):
    """one of the other BASELINE round shapes (configs[3] gangs, configs[4] oversubscribed / preemption-heavy): GPU rounds timed like the headline,
    the oracle on the same input for the cpu_baseline and the parity verdict when its round fits the remaining budget"""
    import numpy as np
    import torch
    from armada_amd import workloads as W
    from armada_amd import multipool
    if args.other_scale != 1.0:   # toy sizes for the CPU contract test
        kwargs = dict(kwargs, n_nodes=max(16, int(kwargs["n_nodes"] * args.other_scale)), n_jobs=max(200, int(kwargs["n_jobs"] * args.other_scale)),
                      n_queues=max(2, min(kwargs["n_queues"], 8)))
        if kwargs.get("gangs"):
            kwargs["gangs"] = max(2, int(kwargs["gangs"] * args.other_scale))
    wl = W.config3(seed=W.SEED, **kwargs)
    scale = kwargs["n_jobs"] / 1_000_000.0
    if kwargs["n_jobs"] != 1_000_000:
        wl.global_burst, wl.queue_burst = max(1, int(200_000 * scale)), max(1, int(20_000 * scale))
    s = W.load(hip, wl)
    lat, dev_ms, res = multipool.timed_rounds(s, wl, steps, warmup, torch.cuda.synchronize, torch.cuda.synchronize)
    st = s.round_stats()
    tm = s.round_timing()
    queries, iters = res.num_node_queries, res.num_loop_iterations
    alg = algorithmic_bytes(wl.num_nodes, W.R, queries, len(res.scheduled) + res.num_evicted_phase1)
    kern_ms = tm["control_ms"] if tm["control_ms"] > 0 else float(np.mean(dev_ms))
    ach = alg / max(kern_ms * 1e-3, 1e-12) / 1e9
    rec = {"config": label, "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {kwargs['n_jobs']} queued jobs (+{wl.num_jobs - kwargs['n_jobs']} running), global burst {wl.global_burst}, "
                                        f"queue burst {wl.queue_burst}" + (f", {kwargs.get('gangs')} gangs of 2-64" if kwargs.get("gangs") else "") + (f", nodes {kwargs.get('occupied'):.0%} occupied" if kwargs.get("occupied", 0.5) != 0.5 else ""),
           "note": note, "metric": "scheduling rounds/sec", "value": 1.0 / float(np.mean(lat)), "unit": "rounds/s", "steps": steps, "ms_per_step": float(np.mean(lat)) * 1e3, "device_ms": float(np.mean(dev_ms)), "k_control_ms": kern_ms,
           "round": {"scheduled": len(res.scheduled), "preempted": len(res.preempted), "evicted_phase1": res.num_evicted_phase1, "evicted_phase3": res.num_evicted_phase3,
                     "loop_iterations": iters, "node_queries_issued": queries, "fast_iterations": st["fast_iterations"], "generic_iterations": st["generic_iterations"],
                     "preempt_fast_iterations": st.get("preempt_fast_iterations", 0),   # jobs that needed preemption and stayed in the fast loop (node side: the generic cascade)
                     "termination_reason": res.termination_reason, "kclk_plane_scans": st["kclk_plane_scans"], "kclk_fair_selects": st["kclk_fair_selects"],
                     "kclk_pass1": st["kclk_pass1"], "kclk_pass2": st["kclk_pass2"], "l0_overflows": st["l0_overflows"]},
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": "k_control",
                        "algorithmic_bytes_per_launch": alg}}
    s.close()
    return rec, wl, res, iters


def other_configs(hip, args, t_start, sink=None):
    """BASELINE configs[1], [3], [4] and the submit check as sub-records of the driver-run line (each with roofline + cpu_baseline).  configs[3] runs at
    the full 100k x 1M size on the GPU; the oracle legs (cpu_baseline + parity) run at the largest size whose oracle round fits the budget.
    `sink`: the list the records are appended to as they finish (the watchdog prints what is there if a later one never comes back)."""
    recs = sink if sink is not None else []

    def guarded(name, fn):
        if time.perf_counter() - t_start > args.other_budget:
            recs.append({"config": name, "skipped": f"--other-budget {args.other_budget:.0f} s spent"}); return
        try:
            recs.append(fn())
        except Exception as e:  # a sub-record must never take the headline down
            recs.append({"config": name, "error": f"{type(e).__name__}: {e}"})

    guarded("BASELINE configs[1]", lambda: fit_batch_record(hip, args))

    def shape(label, full_kwargs, reduced_kwargs, note, full_parity=False, only_size=False):
        def run():
            if full_parity and full_kwargs and args.cpu_budget > 0:   # the oracle round of the FULL input (~1 minute, like the headline's): parity at the size the value is quoted on
                rec, wl, res, iters = round_shape_record(hip, args, label, full_kwargs, 2, note, 1)
                base, ores = cpu_baseline(wl, 1e9, iters)
                rec["cpu_baseline"] = base
                if ores is not None:
                    rec["parity"] = parity_record(res, ores, wl.num_jobs, "oracle round on the same full-size input")
                return rec
            heavy = bool(full_kwargs) and full_kwargs.get("occupied", 0.5) > 0.9   # the preemption-heavy shape at full size is ~25 s per round: ONE round, no warm-up
            rec, _, _, _ = round_shape_record(hip, args, label, full_kwargs, 1 if heavy else 2, note + (" (one round, no warm-up round)" if heavy else ""), 0 if heavy else 1) if full_kwargs else (None, None, None, None)
            red, wl, res, iters = round_shape_record(hip, args, label + " (reduced: the size the oracle leg runs at)", reduced_kwargs, 2, note)
            if args.cpu_budget > 0:
                base, ores = cpu_baseline(wl, 1e9, iters)   # never cut: the parity verdict needs the whole round
                red["cpu_baseline"] = base
                if ores is not None:
                    red["parity"] = parity_record(res, ores, wl.num_jobs, "oracle round on the same input")
            if rec is None:
                if only_size: red["config"] = label   # (a record that has no full-size leg by design)
                return red
            rec["reduced"] = red
            rec["cpu_baseline"] = dict(red.get("cpu_baseline") or {}, note="measured at the reduced size (see `reduced`); the GPU value of this record is the full size")
            if "parity" in red:
                rec["parity"] = dict(red["parity"], note="checked at the reduced size")
            return rec
        return run
    guarded("BASELINE configs[3]", shape("BASELINE configs[3]", dict(n_nodes=100_000, n_jobs=1_000_000, n_queues=64, gangs=10_000),
                                         dict(n_nodes=20_000, n_jobs=200_000, n_queues=32, gangs=2_000), "gang atomic placement (ScheduleManyWithTxn + txn abort), uniform shape within a gang",
                                         full_parity=not args.no_full_other))
    guarded("BASELINE configs[4]", shape("BASELINE configs[4]", None if args.no_full_other else dict(n_nodes=100_000, n_jobs=1_000_000, n_queues=64, occupied=0.95),
                                         dict(n_nodes=20_000, n_jobs=200_000, n_queues=32, occupied=0.95),
                                         "oversubscribed / preemption-heavy: nodes 95% occupied, fair-share + urgency preemption candidate search, oversubscribed evictor"))

    def submit():
        import copy
        a = copy.copy(args); a.steps = 3; a.nodes = max(16, int(10_000 * args.other_scale))
        a.submit_jobs, a.submit_keys = max(100, int(args.submit_jobs * args.other_scale)), max(10, int(args.submit_keys * args.other_scale))
        rec = submit_check_record(a)
        rec["workload"] = rec["config"]["workload"]
        rec["config"] = "submit check (SURVEY 8f-2)"
        return rec
    guarded("submit check", submit)
    guarded("fairness optimiser node scoring", lambda: optimiser_record(hip, args))
    guarded("market-driven round + pricer", lambda: market_record(hip, args))
    guarded("two-word order keys", lambda: two_word_record(hip, args))
    guarded("coarser key divisor", lambda: coarse_divisor_record(hip, args))
    guarded("one pool on two replicas", lambda: sharded_record(hip, args))
    # the queue-count cliff (round-2 review): more than 64 queues leave the fast iteration (one lane per queue) for the generic one; measured, not hidden
    guarded("256 queues", shape("256 queues (beyond the 64-lane fast iteration)", None, dict(n_nodes=20_000, n_jobs=200_000, n_queues=256),
                                "more than 64 queues: wide runs since round 4 (round_wide.h: per-queue streams merged by a bulk rank on the helper workgroups; the generic iteration in "
                                "rounds 1-3, 3.3 s); the same nodes and jobs as the reduced configs[3] / [4] inputs, 256 queues", only_size=True))
    guarded("1024 queues", shape("1024 queues", None, dict(n_nodes=20_000, n_jobs=200_000, n_queues=1024), "as above with 1 024 queues", only_size=True))
    guarded("256 queues, crowded", shape("256 queues on a crowded pool (95 % occupied: most new jobs need preemption)", None, dict(n_nodes=20_000, n_jobs=200_000, n_queues=256, occupied=0.95),
                                         "where wide runs help least: a queue whose head needs preemption is a barrier of a run, and evicted heads leave the runs once a preemption has marked jobs — "
                                         "more than half of the iterations are the generic one (one workgroup)", only_size=True))
    guarded("nodedb fit kernel at 100k nodes x 1M queries", lambda: fit_batch_record(hip, args, big=True))
    guarded("configs[4] checker at 100k nodes", lambda: config4_checker_record(hip, args))
    guarded("reference benchmark shapes", lambda: reference_benchmark_record(hip, args))
    guarded("BenchmarkScheduleMany shapes", lambda: schedule_many_benchmark_record(hip, args))
    return recs


def config4_checker_record(hip, args):
    """Round-3 review, weak #3: the preemption-heavy round at the FULL 100 000 nodes (127 helper workgroups, a 900 000-entry evicted table) had no checker — the oracle needs
    minutes for the full burst.  Here: the same nodes and running jobs (95 % occupied), a burst small enough for the oracle (< 90 s), >= 3 timed rounds with p99, and the oracle
    round compared field by field.  kclk_plane_scans + kclk_fair_selects are reported: the wide passes of the preempting jobs."""
    import numpy as np
    import torch
    from armada_amd import workloads as W, multipool
    sc = args.other_scale
    wl = W.config3(seed=W.SEED, n_nodes=max(64, int(100_000 * sc)), n_jobs=max(640, int(300_000 * sc)), n_queues=64, occupied=0.95)
    # the reference's SHIPPED limits (config/scheduler/config.yaml:101-108): maximumSchedulingBurst 1 000, maximumPerQueueSchedulingBurst 1 000, maxQueueLookback 100 000 —
    # the production-shaped round: ~827 000 running jobs evicted and returning, at most 1 000 new ones (round-4 review, next #2); also what the oracle can finish (~12 s)
    wl.global_burst, wl.queue_burst = max(1, int(1_000 * sc)), max(1, int(1_000 * sc))
    wl.config.max_queue_lookback = 100_000
    s = W.load(hip, wl)
    lat, dev_ms, res = multipool.timed_rounds(s, wl, 3, 1, torch.cuda.synchronize, torch.cuda.synchronize)
    st = s.round_stats(); tm = s.round_timing()
    lat_ms = np.array(lat) * 1e3
    rec = {"config": "configs[4] shape at 100 000 nodes with the reference's default limits (checker)", "workload": f"{wl.num_nodes} nodes 95% occupied x {wl.num_queues} queues x {int(300_000 * sc)} queued jobs "
           f"(+{wl.num_jobs - int(300_000 * sc)} running), global burst {wl.global_burst}, queue burst {wl.queue_burst}, maxQueueLookback 100000 (config/scheduler/config.yaml:101-108)", "metric": "scheduling rounds/sec", "value": 1.0 / float(np.mean(lat)), "unit": "rounds/s",
           "steps": 3, "ms_per_step": float(np.mean(lat_ms)), "p50_ms": float(np.percentile(lat_ms, 50)), "p99_ms": float(np.percentile(lat_ms, 99)), "k_control_ms": tm["control_ms"],
           "round": {"scheduled": len(res.scheduled), "preempted": len(res.preempted), "evicted_phase1": res.num_evicted_phase1, "loop_iterations": res.num_loop_iterations,
                     "node_queries_issued": res.num_node_queries, "fast_iterations": st["fast_iterations"], "generic_iterations": st["generic_iterations"],
                     "preempt_fast_iterations": st["preempt_fast_iterations"], "kclk_pass1": st["kclk_pass1"], "kclk_plane_scans": st["kclk_plane_scans"], "kclk_fair_selects": st["kclk_fair_selects"],
                     "kclk_plane_scans_plus_fair_selects": st["kclk_plane_scans"] + st["kclk_fair_selects"]}}
    alg = algorithmic_bytes(wl.num_nodes, W.R, res.num_node_queries, len(res.scheduled) + res.num_evicted_phase1)
    ach = alg / max(tm["control_ms"] * 1e-3, 1e-12) / 1e9
    rec["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": "k_control", "algorithmic_bytes_per_launch": alg}
    if sc == 1.0:   # HBM bytes of the round launch on this exact workload: the newest committed PMC collection (tools/gpu_call_final.sh), with the round kernel's hash it was taken from
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_config4_checker.json")))
        if pm:
            try:
                pj = json.load(open(pm[-1])); c4 = pj["counters"]
                fetch = max(x["max_kb"] for x in c4["FETCH_SIZE"] if x["kernel"].startswith("k_control"))
                write = max(x["max_kb"] for x in c4["WRITE_SIZE"] if x["kernel"].startswith("k_control"))
                rec["roofline"]["traffic"] = (2 * fetch + write) * 1024
                rec["roofline"]["traffic_source"] = f"profiles/{os.path.basename(pm[-1])} (2*FETCH_SIZE + WRITE_SIZE of the pass-1 launch; rocprofv3 PMC passes cannot run inside bench.py)"
                rec["roofline"]["traffic_kernel_isa_hash"] = pj.get("kernel_isa_hash")
            except Exception:
                pass
    s.close()
    if args.cpu_budget > 0:
        base, ores = cpu_baseline(wl, 1e9, res.num_loop_iterations)
        rec["cpu_baseline"] = base
        if ores is not None:
            rec["parity"] = parity_record(res, ores, wl.num_jobs, "oracle round on the same 100 000-node input")
    return rec


def schedule_many_benchmark_record(hip, args):
    """BenchmarkScheduleMany* (nodedb_test.go:1590-1712): ONE gang context of J identical 1-cpu / 4-Gi jobs through `txn := nodeDb.Txn(true); nodeDb.ScheduleManyWithTxn(txn, gctx);
    txn.Abort()` on a fresh NodeDb of N 32-cpu nodes — the loop body of the benchmark is exactly one unit of asched_submit_check.  Such a unit is UNIFORM (every member of one shape):
    on a NodeDb in its initial state its outcome is a sum over the nodes, one pass (csrc/submit_gang.h k_fit_capacity), whatever J.  The shapes whose nodes carry used resources at
    priority 0 (`WithUsedResourcesNodes`, :1666-1700) qualify too: that lowers the planes of every level up to the jobs' own alike.  No Go toolchain here: both legs are timed on this box."""
    import numpy as np
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    shapes = [(1, 320, 0), (1, 640, 0), (100, 3200, 0), (100, 6400, 0), (1000, 32000, 0), (1000, 64000, 0), (100, 100, 31), (1000, 1000, 31), (10000, 10000, 31)]   # (nodes, jobs, cpus used per node at priority 0)
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    oracle = Library(path, "oracle_") if (args.cpu_budget > 0 and os.path.exists(path)) else None
    rows, all_same = [], True
    for nn, nj, used in shapes:
        nj = max(4, int(nj * args.other_scale)); nn = max(1, int(nn * min(1.0, args.other_scale * 10)))
        wl = W.reference_benchmark(nn, 1, nj)
        legs = {}
        for name, lib in (("gpu", hip), ("oracle", oracle)):
            if lib is None:
                continue
            s = W.Scheduler(lib, wl.config)
            abp = None
            if used:   # testfixtures.WithUsedResourcesNodes(0, Cpu("31"), nodes): AllocatableByPriority[p] -= 31 cpu for p <= 0 (node.go:535-549)
                abp = np.repeat(wl.node_total[:, None, :], s.P, axis=1).copy()
                for l, pr in enumerate(s.priorities):
                    if pr <= 0:
                        abp[:, l, W.CPU] -= used * 1000
            s.nodes_upsert(wl.node_total, wl.node_allocatable, alloc_by_prio=abp)
            W.set_jobs(s, wl)
            unit = (np.array([0, nj], dtype=np.int32), np.arange(nj, dtype=np.int32))   # the unit in the ABI's CSR form (the job ids of a gang context: built once, like the benchmark's `jobs`)
            s.submit_check(None, [False], csr=unit)   # warm-up
            times, r = [], None
            for _ in range(7):
                if name == "gpu": torch.cuda.synchronize()
                t0 = time.perf_counter(); r = s.submit_check(None, [False], csr=unit)
                if name == "gpu": torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            legs[name] = (float(np.median(times)), r[0], s.submit_stats() if name == "gpu" else None)   # median of 7 (a 0.1 ms call: one host hiccup must not be the row)
            s.close()
        row = {"shape": f"{nn} nodes {nj} jobs" + (f" ({used} of 32 cpus used per node)" if used else ""), "gpu_ms": legs["gpu"][0] * 1e3, "ok": bool(legs["gpu"][1][0]), "num_schedulable": int(legs["gpu"][1][2]),
               "how": "capacity pass" if legs["gpu"][2]["gang_units"] else "sequential control launch"}
        if "oracle" in legs:
            same = legs["oracle"][1] == legs["gpu"][1]
            all_same = all_same and same
            row.update({"oracle_ms": legs["oracle"][0] * 1e3, "x_oracle": legs["oracle"][0] / max(legs["gpu"][0], 1e-12), "identical": bool(same)})
        rows.append(row)
    rec = {"config": "BenchmarkScheduleMany shapes (nodedb_test.go:1590-1712)", "metric": "ms per Txn / ScheduleManyWithTxn / Abort of one gang context", "unit": "ms", "rows": rows,
           "note": "a NodeDb in its initial state + identical members: one pass over the nodes per call, whatever the number of members (the time that is left is marshalling J job ids through ctypes)"}
    if oracle is not None:
        rec["parity"] = {"checked": True, "identical": bool(all_same), "against": "oracle: ok, num_schedulable, first node per shape", "jobs": int(sum(int(r["shape"].split()[2]) for r in rows))}
    return rec


def reference_benchmark_record(hip, args):
    """BenchmarkPreemptingQueueScheduler's own shapes (preempting_queue_scheduler_test.go:2561-2799): N 32-cpu nodes, Q queues x J queued 1-cpu / 4-Gi jobs of one priority class, unlimited
    rate limiters, protectedFractionOfFairShare 1.0, EnablePreferLargeJobOrdering; a first round fills the nodes, the TIMED round is the steady state the benchmark times (the jobs of
    the first round running, the rest queued: nothing to preempt, nothing fits).  There is no Go toolchain here: both legs are timed on this box — the library and the CPU oracle."""
    import numpy as np
    import torch
    from armada_amd import workloads as W
    from armada_amd.binding import Library
    shapes = [(1, 1, 320), (1, 10, 320), (10, 1, 3200), (10, 10, 3200), (100, 1, 32000), (100, 10, 32000), (1000, 1, 320000), (1000, 1, 32000)]   # (nodes, queues, jobs per queue): the table of :2571-2651
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    oracle = Library(path, "oracle_") if (args.cpu_budget > 0 and os.path.exists(path)) else None
    rows, all_same = [], True
    for nn, nq, per in shapes:
        per = max(8, int(per * args.other_scale))
        wl = W.reference_benchmark(nn, nq, per)
        legs = {}
        for name, lib in (("gpu", hip), ("oracle", oracle)):
            if lib is None:
                continue
            s = W.load(lib, wl); W.prepare(s, wl)
            first = s.schedule_round()                      # fills the nodes (untimed: the benchmark does this before ResetTimer)
            node = wl.job_node.copy(); prio = wl.job_run_prio.copy()
            for j, n in first.scheduled.items():
                node[int(j)] = int(n); prio[int(j)] = int(first.scheduled_priority[int(j)])
            W.set_jobs(s, wl, job_node=node.astype(np.int32), job_run_prio=prio.astype(np.int32))
            queued = [np.array([int(j) for j in q if node[int(j)] < 0], dtype=np.int32) for q in wl.queued]
            times, r = [], None
            for _ in range(3):
                s.round_prepare(wl.queue_weight, queued)
                if name == "gpu": torch.cuda.synchronize()
                t0 = time.perf_counter(); r = s.schedule_round()
                if name == "gpu": torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            legs[name] = (float(np.mean(times)), r, len(first.scheduled))
            s.close()
        row = {"shape": f"{nn} nodes {nq} queues {per} jobs per queue", "gpu_ms": legs["gpu"][0] * 1e3, "filled_by_first_round": legs["gpu"][2],
               "steady_state": {"scheduled": len(legs["gpu"][1].scheduled), "preempted": len(legs["gpu"][1].preempted), "loop_iterations": legs["gpu"][1].num_loop_iterations}}
        if "oracle" in legs:
            same = not round_diff(legs["oracle"][1], legs["gpu"][1])
            all_same = all_same and same
            row.update({"oracle_ms": legs["oracle"][0] * 1e3, "x_oracle": legs["oracle"][0] / max(legs["gpu"][0], 1e-12), "identical": same})
        rows.append(row)
    rec = {"config": "BenchmarkPreemptingQueueScheduler shapes (preempting_queue_scheduler_test.go:2561-2799)", "metric": "ms per steady-state round", "unit": "ms", "rows": rows,
           "note": "small inputs: the round is launch- and latency-bound (41 launches + three host syncs per round); the library is built for pools four orders of magnitude larger"}
    if oracle is not None:
        rec["parity"] = {"checked": True, "identical": bool(all_same), "against": "oracle steady-state round per shape", "jobs": int(sum(r["steady_state"]["loop_iterations"] for r in rows))}
    return rec


def single_pool_mode_record(hip, args, rank, local_rank, world, dist, torch):
    """--mode node-sharded-fit / queue-hash: ONE pool on `world` GPUs.  Same timing contract as the pools line: W untimed steps, then exactly K steps
    between barrier + synchronize, MAX over ranks.  Total work is fixed as N grows (strong scaling).  No scaling curve has been measured by this repository: the
    8-GPU runs are the driver's, and it runs the default (pools) mode."""
    import numpy as np
    from armada_amd import workloads as W
    device = "cuda"   # the exchange buffers live on the GPU: the library writes / reads them in place (also at N = 1, where the collective is skipped)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def tmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic", "mode": args.mode,
            "scaling_note": "one pool on N GPUs; no scaling curve has been measured by this repository (the driver's SCALE runs use the default pools mode)"}
    if args.mode == "node-sharded-fit":
        from armada_amd.sharded import ShardedFit
        n_nodes, n_jobs = (10_000, 100_000) if args.nodes == 100_000 and args.jobs == 1_000_000 else (args.nodes, args.jobs)   # BASELINE configs[1] unless sizes are given
        wl = W.config2(n_nodes=n_nodes, n_jobs=n_jobs)
        wl.config.device = local_rank
        sf = ShardedFit(hip, wl, rank, world, dist=dist, device=device)
        sf.prepare()
        jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
        prio = sf.s.priorities[0]
        for _ in range(args.warmup):
            got = sf.fit_select_batch(jobs, prio)
        barrier(); t0 = time.perf_counter()
        for _ in range(args.steps):
            got = sf.fit_select_batch(jobs, prio)
        barrier(); dt = tmax(time.perf_counter() - t0)
        sf.close()
        parity = None
        if rank == 0:   # the unsharded library on one GPU answers the same batch: identical node per query
            s1 = W.load(hip, wl); W.prepare(s1, wl); want = s1.fit_select_batch(jobs, prio); s1.close()
            parity = {"checked": True, "identical": bool((want == got).all()), "jobs": int(len(jobs)), "against": "the unsharded library on one GPU, node per query"}
        return dict(base, metric="first-fit queries/s, one pool's nodes partitioned over the GPUs (asched_fit_select_batch_global + one all-reduce MIN per batch)",
                    value=len(jobs) * args.steps / dt, unit="queries/s", ms_per_step=dt / args.steps * 1e3, parity=parity,
                    config={"workload": f"BASELINE configs[1] shape: {wl.num_nodes} nodes x {len(jobs)} first-fit queries per step at priority {prio}", "nodes": wl.num_nodes,
                            "parallelism": f"nodes of one pool in {world} contiguous row shards", "exchange_bytes_per_step_per_rank": int(len(jobs)) * 8})
    if args.mode == "node-sharded-round":
        # ONE pool's round, exact, on N GPUs: every rank holds the pool and runs the round; the wide passes over the nodes are split N ways and end with one all-reduce MIN of
        # two words over RCCL (asched_shard_round).  What it can buy is the wide passes' share of a round (crowded pools: --occupied 0.95); the headline round issues none.
        from armada_amd import comm
        wl = W.config3(n_nodes=args.nodes, n_jobs=args.jobs, n_queues=args.queues, seed=W.SEED, gangs=args.gangs, occupied=args.occupied)
        scale = args.jobs / 1_000_000.0
        if args.jobs != 1_000_000:
            wl.global_burst, wl.queue_burst = max(1, int(200_000 * scale)), max(1, int(20_000 * scale))
        wl.config.device = local_rank
        s = W.load(hip, wl)
        if world > 1:
            comm.init_rccl(s, dist, device="cuda")
            s.shard_round(True)
        times, ex = [], 0
        for i in range(args.warmup + args.steps):
            W.prepare(s, wl)
            barrier(); t0 = time.perf_counter()
            r = s.schedule_round()
            barrier(); dt = tmax(time.perf_counter() - t0)
            if i >= args.warmup:
                times.append(dt)
            ex = s.shard_exchanges()
        st = s.round_stats()
        s.close()
        parity = None
        if rank == 0 and world > 1:   # the unsharded library on one GPU runs the same round
            s1 = W.load(hip, wl); W.prepare(s1, wl); want = s1.schedule_round(); s1.close()
            parity = {"checked": True, "identical": bool(np.array_equal(want.scheduled_job, r.scheduled_job) and np.array_equal(want.scheduled_node, r.scheduled_node) and
                                                        np.array_equal(want.preempted_job, r.preempted_job)),
                      "against": "the unsharded library on one GPU", "jobs": int(len(want.scheduled_job))}
        return dict(base, metric="scheduling rounds/sec, ONE pool on N GPUs: whole state on every GPU, wide node passes split N ways + one all-reduce MIN each (exact)",
                    value=len(times) / sum(times), unit="rounds/s", ms_per_step=sum(times) / len(times) * 1e3, parity=parity, exchanges_last_launch=int(ex),
                    kclk_plane_scans=int(st.get("kclk_plane_scans", 0)), kclk_fair_selects=int(st.get("kclk_fair_selects", 0)),
                    scaling_note="unmeasured on more than one GPU: this repository's GPU boxes have one; world 2 / 3 parity runs over gloo in tests/test_z_sharded_round.py",
                    config={"workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {args.jobs} queued jobs (+{wl.num_jobs - args.jobs} running), occupied {args.occupied}",
                            "nodes": wl.num_nodes, "queued_jobs": args.jobs, "queues": wl.num_queues,
                            "parallelism": f"replicated pool, node words split {world} ways for the plane scan and the fair-share evaluation; 16 bytes all-reduced per pass"})
    from armada_amd.queuehash import QueueHashRound
    wl = W.config3(n_nodes=args.nodes, n_jobs=args.jobs, n_queues=args.queues, seed=W.SEED, gangs=args.gangs, occupied=args.occupied)
    scale = args.jobs / 1_000_000.0
    if args.jobs != 1_000_000:
        wl.global_burst, wl.queue_burst = max(1, int(200_000 * scale)), max(1, int(20_000 * scale))
    wl.config.device = local_rank
    qh = QueueHashRound(hip, wl, rank, world, dist=dist, device=device)
    for _ in range(args.warmup):
        r = qh.run()
    barrier(); t0 = time.perf_counter()
    parts = []
    for _ in range(args.steps):
        r = qh.run(); parts.append(dict(qh.timing))
    barrier(); dt = tmax(time.perf_counter() - t0)
    qh.close()
    cmp_, exact_ms = None, None
    if rank == 0:   # the exact round of the same pool on one GPU: what this mode's assignment is measured against
        s1 = W.load(hip, wl); W.prepare(s1, wl)
        torch.cuda.synchronize(); t1 = time.perf_counter(); exact = s1.schedule_round(); torch.cuda.synchronize(); exact_ms = (time.perf_counter() - t1) * 1e3
        s1.close()
        cmp_ = QueueHashRound.compare(r, exact.scheduled, wl)
    words = wl.num_nodes * wl.job_req.shape[1] + wl.num_jobs
    return dict(base, metric="scheduling rounds/sec, ONE pool's queues hashed over the GPUs (approximate: see mismatch_vs_exact_round)", value=args.steps / dt, unit="rounds/s",
                ms_per_step=dt / args.steps * 1e3, exact_single_gpu_round_ms=exact_ms,
                phases_ms={k: float(np.mean([p[k] for p in parts])) * 1e3 for k in parts[0]},
                round={k: (len(r[k]) if isinstance(r[k], (list, dict)) else int(r[k])) for k in ("scheduled", "dropped", "conflicts", "accepted", "replay_set", "replayed", "preempted")},
                mismatch_vs_exact_round=cmp_, exchange_bytes_per_round_per_rank=words * 8,
                config={"workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x {args.jobs} queued jobs (+{wl.num_jobs - args.jobs} running), global burst {wl.global_burst}, queue burst {wl.queue_burst}",
                        "nodes": wl.num_nodes, "queued_jobs": args.jobs, "queues": wl.num_queues,
                        "parallelism": f"queue q on rank q mod {world}; one all-reduce SUM of N*R + M int64 words; ordered replay of the conflict set on every rank"})


COMPACT_LIMIT = 4000   # characters: the driver keeps the last 8 kB of stdout and parses the last line (round 4's 28 kB line came back `parsed: null`)


def compact_line(out):
    """The LAST stdout line: the headline with `roofline`, `cpu_baseline`, `parity` and one short row per other config — at most COMPACT_LIMIT characters.
    Everything else of the record goes to bench_full.json (and to the stdout line before this one)."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}

    def num(x):
        return round(x, 6 - len(str(int(abs(x))))) if isinstance(x, float) and abs(x) >= 1 else (float(f"{x:.4g}") if isinstance(x, float) else x)

    def short(d):
        return {k: num(v) for k, v in d.items()}
    line = short(pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "p50_ms", "p99_ms", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "mode")))
    cfg = out.get("config") or {}
    line["config"] = {k: (v[:300] if isinstance(v, str) else v) for k, v in cfg.items()} if isinstance(cfg, dict) else cfg
    if "roofline" in out:
        line["roofline"] = short(pick(out["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "traffic_kernel_isa_hash", "kernel", "algorithmic_bytes_per_launch", "kernel_isa_hash", "kernel_avg_ms", "ticks_per_placement", "passes_executed", "node_queries")))
        if "frac_kind" in out["roofline"]:
            line["roofline"]["frac_kind"] = "unbatched-equivalent (queries issued x N x 40 B; no plane is scanned: see ticks_per_placement)"
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = short(pick(cb, ("value", "unit", "cores", "kind", "rounds", "measured_s", "measured_min_s", "measured_max_s", "error")))
        if "sample" in cb:
            line["cpu_baseline"]["sample"] = cb["sample"][:160]
    if "parity" in out:
        line["parity"] = pick(out["parity"], ("checked", "identical", "jobs", "scheduled", "preempted", "reason"))
    if "round" in out:
        line["round"] = short(pick(out["round"], ("scheduled", "preempted", "evicted_phase1", "loop_iterations", "node_queries_issued", "device_ms", "k_control_ms", "kernel_launches", "host_ms")))
    rows = []
    for r in out.get("other_configs", []):
        name = str(r.get("config", "?"))
        if "error" in r or "skipped" in r:
            rows.append([name[:40], None, "error" if "error" in r else "skipped", None, None, None]); continue
        cb, par, rf = r.get("cpu_baseline") or {}, r.get("parity") or {}, r.get("roofline") or {}
        val, unit = r.get("value"), r.get("unit")
        x = None
        if isinstance(val, (int, float)) and isinstance(cb.get("value"), (int, float)) and cb["value"] > 0 and not (r.get("reduced")):
            x = val / cb["value"]
        elif r.get("reduced"):   # the oracle leg ran at the reduced size: the ratio is the reduced leg's
            red = r["reduced"]; rcb = red.get("cpu_baseline") or {}
            if isinstance(rcb.get("value"), (int, float)) and rcb["value"] > 0:
                x = red["value"] / rcb["value"]
        elif "rows" in r:   # a table of shapes: the worst row's ratio
            xs = [q["x_oracle"] for q in r["rows"] if "x_oracle" in q]
            x = min(xs) if xs else None
        rows.append([name[:40], num(val) if isinstance(val, float) else val, unit, num(x) if x is not None else None,
                     num(rf["frac"]) if isinstance(rf.get("frac"), float) else None, par.get("identical")])
    if rows:
        line["other_configs"] = rows
        line["other_configs_columns"] = ["config", "value", "unit", "x_oracle (reduced leg's where the oracle ran reduced; worst row of a table)", "roofline.frac", "parity.identical"]
    for k in ("scaling_note", "full_record", "watchdog"):
        if k in out:
            line[k] = out[k]
    txt = json.dumps(line)
    while len(txt) > COMPACT_LIMIT and line.get("other_configs"):   # never over the limit: drop rows from the end, say so
        line["other_configs"].pop(); line["other_configs_truncated"] = True
        txt = json.dumps(line)
    return txt


def emit(out):
    """full record -> bench_full.json (+ gpurun_out/ when that exists) and one stdout line; then the compact line LAST"""
    full = json.dumps(out)
    wrote = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(full + "\n")
                wrote.append(os.path.relpath(os.path.join(d, "bench_full.json"), ROOT))
            except OSError:
                pass
    out = dict(out, full_record=(wrote[0] if wrote else "the stdout line before this one"))
    print(full)
    print(compact_line(out))


_WATCHDOG = {"timer": None}


def watchdog_arm(out, seconds):
    """After `seconds` print the record as it stands (the headline + the sub-records finished so far, `watchdog` saying so) as the last stdout line and leave with exit code 5.
    The host thread of a hung round kernel sits inside the C ABI call and cannot be interrupted; the process can still end."""
    import threading
    if seconds <= 0:
        return

    def fire():
        snap = dict(out, other_configs=list(out.get("other_configs", [])), watchdog=f"a sub-record did not return within {seconds:.0f} s of the headline; the record was cut there")
        emit(snap)
        sys.stdout.flush()
        os._exit(5)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    _WATCHDOG["timer"] = t


def watchdog_disarm():
    if _WATCHDOG["timer"] is not None:
        _WATCHDOG["timer"].cancel()
        _WATCHDOG["timer"] = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--watchdog", type=float, default=900.0, help="seconds after the headline within which the other_configs sub-records must have finished; then the record is printed as it stands (0 = off)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--jobs", type=int, default=1_000_000)
    ap.add_argument("--queues", type=int, default=64)
    ap.add_argument("--gangs", type=int, default=0, help="queued gangs of size 2-64 (BASELINE configs[3] shape)")
    ap.add_argument("--occupied", type=float, default=0.5, help="fraction of every node filled with running jobs (0.95: the preemption-heavy configs[4] shape)")
    ap.add_argument("--cpu-budget", type=float, default=75.0, help="seconds of CPU-oracle work allowed for cpu_baseline (0 = skip)")
    ap.add_argument("--submit-check", action="store_true", help="measure the submit check (SURVEY 8f-2) instead of the round; 1 GPU")
    ap.add_argument("--submit-jobs", type=int, default=50_000)
    ap.add_argument("--submit-keys", type=int, default=2_000)
    ap.add_argument("--no-other", action="store_true", help="skip the other_configs sub-records (configs[1], [3], [4], submit check)")
    ap.add_argument("--other-budget", type=float, default=480.0, help="seconds after which no further other_configs sub-record is started")
    ap.add_argument("--no-full-other", action="store_true", help="skip the full-size (100k x 1M) run of configs[4] (~40 s); the reduced size with its oracle leg still runs")
    ap.add_argument("--other-scale", type=float, default=1.0, help="scale the other_configs workloads (tests)")
    ap.add_argument("--full-other", action="store_true", help="(default now; kept for older command lines)")
    ap.add_argument("--mode", choices=["pools", "node-sharded-fit", "node-sharded-round", "queue-hash"], default="pools",
                    help="what --gpus N > 1 does (DESIGN.md 7): pools = one pool per GPU, exact, no data-path collective (the default and the line the driver records); "
                         "node-sharded-fit = ONE pool's nodes partitioned over the GPUs for the wide first-fit queries, one all-reduce MIN per batch, exact; "
                         "node-sharded-round = ONE pool's whole round on every GPU with the wide node passes split N ways and all-reduced (asched_shard_round), exact; "
                         "queue-hash = the north_star's split of ONE pool's queues over the GPUs with one all-reduce SUM and an ordered replay, approximate: the line "
                         "reports the mismatch against the exact round")
    args = ap.parse_args()
    if args.submit_check:
        emit(submit_check_record(args))
        return 0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU path)"
    torch.cuda.set_device(local_rank)

    import numpy as np
    import armada_amd
    from armada_amd import workloads as W
    hip = armada_amd.load_library()

    if args.mode != "pools":
        rec = single_pool_mode_record(hip, args, rank, local_rank, world, dist, torch)
        if rank == 0:
            emit(rec)
        if dist is not None:
            dist.destroy_process_group()
        return 0

    # one pool per rank; seeds differ per pool
    from armada_amd.multipool import pool_seed
    wl = W.config3(n_nodes=args.nodes, n_jobs=args.jobs, n_queues=args.queues, seed=pool_seed(W.SEED, rank), gangs=args.gangs, occupied=args.occupied)
    scale = args.jobs / 1_000_000.0
    if args.jobs != 1_000_000:  # keep "limits bite" at reduced sizes
        wl.global_burst, wl.queue_burst = max(1, int(200_000 * scale)), max(1, int(20_000 * scale))
    wl.config.device = local_rank
    t0 = time.perf_counter()
    s = W.load(hip, wl)
    build_s = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from armada_amd import multipool
    tp = time.perf_counter()
    # (timed: asched_schedule_round through the C ABI + the copy of every result array out of the library's buffers; the binding's job -> node DICT views are built lazily, outside —
    #  three 200 000-entry Python dicts cost more than the library's whole host side of a round and no array consumer builds them)
    lat, dev_ms, res = multipool.timed_rounds(s, wl, args.steps, args.warmup, barrier, torch.cuda.synchronize)
    prep_s = multipool.timed_rounds.prepare_s  # untimed input build (fresh NodeDb + bind running jobs + fair shares + sorted base)

    def amax(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    value, total = multipool.aggregate(world, args.steps, float(sum(lat)), amax if dist is not None else None)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    lat_ms = np.array(lat) * 1e3
    queries, iters = res.num_node_queries, res.num_loop_iterations
    binds = len(res.scheduled) + res.num_evicted_phase1  # new binds + rescheduled evicted jobs (upper bound)
    alg = algorithmic_bytes(wl.num_nodes, W.R, queries, binds)
    seq_ms = float(np.mean(dev_ms))                  # the whole launch sequence of a round (HIP events on the stream)
    timing = s.round_timing()                        # last round: ms inside the persistent k_control launches, launches, bulk phases
    ctl = [x for x in getattr(multipool.timed_rounds, "control_ms", []) if x > 0]
    kern_ms = float(np.mean(ctl)) if ctl else (timing["control_ms"] if timing["control_ms"] > 0 else seq_ms)   # MEAN over the timed rounds (round-5 review: it was the last round's)
    achieved = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    out = {
        "metric": "scheduling rounds/sec (p99 round latency in p99_ms), 100k nodes x 1M jobs",
        "value": value, "unit": "rounds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": total / args.steps * 1e3, "p50_ms": float(np.percentile(lat_ms, 50)), "p99_ms": float(np.percentile(lat_ms, 99)),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{3 if args.gangs else 4 if args.occupied >= 0.9 else 2}]{' shape' if (args.gangs or args.occupied != 0.5) else ''}: {wl.num_nodes} nodes x {wl.num_queues} queues x {args.jobs} queued jobs (+{wl.num_jobs - args.jobs} running), "
                               f"R=4, 3 priority classes, DRF weights, global burst {wl.global_burst}, queue burst {wl.queue_burst}, protectedFractionOfFairShare 0.5"
                               + (f", {args.gangs} gangs" if args.gangs else "") + (f", nodes {args.occupied:.0%} occupied" if args.occupied != 0.5 else ""),
                   "nodes": wl.num_nodes, "queued_jobs": args.jobs, "running_jobs": wl.num_jobs - args.jobs, "queues": wl.num_queues,
                   "parallelism": f"pool-per-gpu x{world}" if world > 1 else "1 pool on 1 gpu", "seed": W.SEED},
        "round": {"scheduled": len(res.scheduled), "preempted": len(res.preempted), "evicted_phase1": res.num_evicted_phase1,
                  "evicted_phase3": res.num_evicted_phase3, "loop_iterations": iters, "node_queries_issued": queries,
                  # the queries the ALGORITHM issues next to what the device executes for them — a query is a find-first-set over a fit bitmap window + the L0
                  # list (base_scan_steps counts the windows read), a full-plane scan only on the generic path (kclk_plane_scans > 0)
                  "passes_executed": {"base_window_reads": s.round_stats()["base_scan_steps"], "l0_list_max": s.round_stats()["l0_max"], "plane_scan_kclk": s.round_stats()["kclk_plane_scans"]},
                  "termination_reason": res.termination_reason, "device_ms": seq_ms, "k_control_ms": kern_ms, "kernel_launches": timing["launches"],
                  "bulk_phases_host_ms": {"evict1": timing["evict1_host_ms"], "evict3": timing["evict3_host_ms"], "unbind_results": timing["final_host_ms"]},
                  "host_ms": float(np.mean(lat_ms)), "stats": s.round_stats()},
        "input_build_s": build_s, "round_prepare_s": prep_s / (args.warmup + args.steps),
        "prepare_inclusive_rounds_per_s": 1.0 / (total / args.steps + prep_s / (args.warmup + args.steps)),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_kind": "unbatched-equivalent: SURVEY 8d's per-query bytes x the queries the ALGORITHM issues; no node plane is scanned (passes_executed), the kernel is a serial "
                                  "chain of placements, not an HBM stream — ticks_per_placement is its real figure of merit",
                     "traffic": None, "kernel": "k_control", "algorithmic_bytes_per_launch": alg,
                     "note": "algorithmic bytes = node_queries_issued x N x (R*8+8) + binds x 256 (SURVEY 8d); duration = the persistent k_control launch(es) of one round "
                             "(HIP events on the launch stream; the grid-wide bulk kernels around them are in round.device_ms)"},
    }
    # HBM bytes actually moved by one round launch: rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE),
    # collected once per round of work on this exact workload and committed under profiles/ (rocprofv3 cannot run inside bench.py)
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if pmcs and args.nodes == 100_000 and args.jobs == 1_000_000 and args.queues == 64 and not args.gangs and args.occupied == 0.5:
        try:
            c = json.load(open(pmcs[-1]))["counters"]
            fetch = max(x["max_kb"] for x in c["FETCH_SIZE"] if x["kernel"].startswith("k_control"))
            write = max(x["max_kb"] for x in c["WRITE_SIZE"] if x["kernel"].startswith("k_control"))
            out["roofline"]["traffic"] = (2 * fetch + write) * 1024
            out["roofline"]["traffic_kernel_isa_hash"] = json.load(open(pmcs[-1])).get("kernel_isa_hash")   # equal to kernel_isa_hash below <=> the counters were taken from this round kernel
            out["roofline"]["traffic_source"] = (f"profiles/{os.path.basename(pmcs[-1])} (2*FETCH_SIZE + WRITE_SIZE of the round launch) — NOT measured in this run: rocprofv3 PMC passes cannot run "
                                                 f"inside bench.py; the file is the newest committed PMC collection on this exact workload (tools/gpu_call.sh) and may predate the library under test")
        except Exception:
            pass
    out["roofline"]["kernel_isa_hash"] = kernel_isa_hash()
    out["roofline"]["kernel_avg_ms"] = kern_ms
    out["roofline"]["kernel_ms_per_round"] = [round(x, 3) for x in ctl]
    if out["roofline"].get("traffic") is not None and out["roofline"].get("traffic_kernel_isa_hash") != out["roofline"]["kernel_isa_hash"]:
        out["roofline"]["traffic_of_another_build"] = out["roofline"]["traffic"]   # kept in the full record for reference only
        out["roofline"]["traffic"] = None
        out["roofline"]["traffic_stale"] = True   # the committed PMC passes were taken from a different round kernel: no byte count is claimed for this one
    st_ = s.round_stats()
    placed = max(1, len(res.scheduled))
    out["roofline"]["ticks_per_placement"] = (st_["kclk_pass1"] + st_["kclk_pass2"]) * 1000.0 / placed   # shader clocks of the two sequential passes per job placed (last round)
    out["roofline"]["passes_executed"] = {"base_window_reads": st_["base_scan_steps"], "plane_scan_kclk": st_["kclk_plane_scans"], "stream_runs": st_["stream_runs"]}
    rc = 0
    if args.cpu_budget > 0 and world == 1:  # rank 0 at N=1 only
        try:
            out["cpu_baseline"], ores = cpu_baseline(wl, args.cpu_budget, iters, min_rounds=3 if not cut_possible(wl, args.cpu_budget, iters) else 1)
            if ores is not None:  # the headline number is self-verifying: the timed GPU round against the oracle round on the same input
                out["parity"] = parity_record(res, ores, wl.num_jobs, "oracle round of the cpu_baseline leg, same input")
                out["parity"]["excluded_nodes"] = excluded_parity(s, ores)
                if not out["parity"]["identical"] or out["parity"]["excluded_nodes"].get("identical") is False:
                    rc = 3
            else:
                out["parity"] = {"checked": False, "reason": "cpu_baseline ran with a cut burst (--cpu-budget)"}
        except Exception as e:  # the checker must never take the bench line down
            out["cpu_baseline"] = {"error": str(e)}
    else:
        out["parity"] = {"checked": False, "reason": "no oracle leg at --cpu-budget 0 / N>1"}
    if world == 1 and not args.no_other:
        s.close()
        out["other_configs"] = []
        watchdog_arm(out, args.watchdog)   # from here on the headline exists: a sub-record that never returns (a hung kernel: profiles/r05y_bulk_skip_hang.txt) must not take it down
        other_configs(hip, args, time.perf_counter(), sink=out["other_configs"])
        watchdog_disarm()
        for r in out["other_configs"]:
            for p in (r.get("parity"), (r.get("reduced") or {}).get("parity")):
                if p and p.get("checked") and not p.get("identical"):
                    rc = 3
    if world > 1:
        out["scaling_note"] = "pool-per-GPU replicas (weak scaling, no data-path collective); no single-pool multi-GPU scaling curve is claimed by this line"
    emit(out)
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
