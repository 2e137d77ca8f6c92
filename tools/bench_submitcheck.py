#!/usr/bin/env python3
"""Submit-check throughput: jobs checked per second by the batched flow
(armada_amd.submitcheck.SubmitChecker -> asched_submit_check) on one pool, next to the reference's sequential flow
(one Txn / ScheduleManyWithTxn / Abort per job) timed on the CPU oracle over a bounded sample.

    python tools/bench_submitcheck.py                      # HIP library on cuda:0 (needs the MI355X)
    python tools/bench_submitcheck.py --lib hostsim        # CPU build of the device code: checks the tool itself, not a measurement

Prints one JSON line.  `roofline.achieved` prices every node query at N x (8R + 8) bytes like bench.py: one query per distinct
scheduling key (answered by the wide fit kernel, a handful of passes over the planes for all keys together — `how` says how many) and
one per gang member (sequential path, full-plane scans).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from armada_amd import workloads as W                                   # noqa: E402
from armada_amd.binding import Library, Scheduler                       # noqa: E402
from armada_amd.submitcheck import PoolConfig, PoolNodeDb, SubmitChecker, SubmitJob  # noqa: E402


class ArrayPoolDb(PoolNodeDb):
    """one pool's cleared NodeDb over array-shaped job tables (the shape a cgo shim would hand over)"""

    def __init__(self, lib, wl, req, pc, gang, gang_card):
        self.s = Scheduler(lib, wl.config)
        self.s.nodes_upsert(wl.node_total, wl.node_allocatable)
        self.s.clear_allocated()
        self.req, self.pc, self.gang, self.gang_card = req, pc, gang, gang_card
        self.launch_ms = []
        self.stats = []

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="hip", choices=["hip", "hostsim", "oracle"])
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--jobs", type=int, default=50_000)
    ap.add_argument("--shapes", type=int, default=2_000)
    ap.add_argument("--gang-frac", type=float, default=0.02)
    ap.add_argument("--cpu-sample", type=int, default=300, help="jobs of the sequential oracle leg")
    a = ap.parse_args()
    if a.lib == "hip":
        import torch  # noqa: F401  (brings the HIP runtime the library binds to)
        import armada_amd
        lib = armada_amd.load_library()
    elif a.lib == "hostsim":
        lib = Library(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"), "asched_")
    else:
        lib = Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    oracle = Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    rng = np.random.Generator(np.random.PCG64(W.SEED))
    wl = W.config2(n_nodes=a.nodes, n_jobs=1)          # node set of BASELINE configs[1]/[2]: 32 cpu / 256 Gi nodes
    req, shape, gang, card = make_jobs(rng, a.jobs, a.shapes, a.gang_frac)
    jobs = [SubmitJob(id=str(i), queue="q", priority_class="pc0", scheduling_key=(int(shape[i]),), request=req[i],
                      gang_id=None if gang[i] < 0 else f"g{gang[i]}") for i in range(a.jobs)]
    pc = np.zeros(a.jobs, np.int32)
    db = ArrayPoolDb(lib, wl, req, pc, gang, card)
    chk = SubmitChecker([PoolConfig("pool")], {"pool": db})
    chk.check(jobs[:64])                                # warm-up (first launch, allocations)
    db.launch_ms.clear(); db.stats.clear()
    t0 = time.perf_counter()
    res = chk.check(jobs)
    dt = time.perf_counter() - t0
    n_units = len({j.scheduling_key for j in jobs}) + len({j.gang_id for j in jobs if j.gang_id})
    queries = len({j.scheduling_key for j in jobs}) + int((gang >= 0).sum())
    dev_s = sum(db.launch_ms) / 1e3 or dt   # the CPU builds report no device time: fall back to wall time
    # sequential reference flow on the oracle, bounded sample: one transaction per job (the cache is defeated on purpose: distinct keys)
    sample = min(a.cpu_sample, a.jobs)
    odb = ArrayPoolDb(oracle, wl, req[:sample], pc[:sample], np.full(sample, -1, np.int32), np.ones(sample, np.int32))
    odb.load_jobs(None)
    t1 = time.perf_counter()
    for i in range(sample):
        odb.s.txn_begin(); odb.s.schedule_many([i]); odb.s.txn_abort()
    cpu_dt = time.perf_counter() - t1
    bytes_per_query = a.nodes * (8 * W.R + 8)
    line = {
        "metric": "submit checks/s (jobs of one Check call, one pool)", "value": a.jobs / dt, "unit": "jobs/s", "lib": a.lib,
        "config": {"workload": f"{a.nodes} nodes, {a.jobs} submitted jobs, {a.shapes} scheduling keys, {int((gang >= 0).sum())} gang members"},
        "schedulable": sum(r.is_schedulable for r in res.values()), "units": n_units, "launches": chk.launches, "how": db.stats,
        "wall_s": dt, "device_s": dev_s, "dtype": "int64", "data": "synthetic",
        "roofline": {"bound": "hbm", "achieved": queries * bytes_per_query / max(dev_s, 1e-9) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": queries * bytes_per_query / max(dev_s, 1e-9) / 8e12, "traffic": None, "node_queries": queries},
        "cpu_baseline": {"value": sample / cpu_dt, "unit": "jobs/s", "cores": 1, "kind": "port",
                         "sample": f"{sample} jobs, one Txn / ScheduleManyWithTxn / Abort each on the CPU oracle, same node set"},
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
