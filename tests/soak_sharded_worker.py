#!/usr/bin/env python3
"""Differential soak of ONE pool's round on several replicas (asched_shard_round), one PROCESS per replica over gloo — the CPU build of the device code keeps its LDS stand-ins
in process-wide variables, so its replicas cannot be threads (tests/soak.py sharded runs the threads variant through the product library on a GPU box).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port 29771 tests/soak_sharded_worker.py 400 [first]

Every rank builds the seed's workload, runs the sharded round and the oracle's, and compares; rank 0 prints the summary.  ASCHED_KEY_WORDS=2 in the environment: under a two-word key."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import scenario  # noqa: E402
from armada_amd import comm, workloads as W  # noqa: E402
from armada_amd.binding import Library  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n, first = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = Library(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"), "asched_")
orc = Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
dist.init_process_group("gloo")
bad, exchanges, t0 = 0, 0, time.time()
for seed in range(200_000 + first, 200_000 + n):
    rng = np.random.default_rng(seed)
    wl = W.small_random(n_nodes=int(rng.integers(4, 300)), n_jobs=int(rng.integers(50, 4000)), n_queues=int(rng.integers(1, 12)), seed=seed,
                        occupied=float(rng.choice([0.3, 0.6, 0.9, 1.0])), gangs=int(rng.integers(0, 8)),
                        burst=None if rng.random() < 0.5 else (int(rng.integers(10, 2000)), int(rng.integers(5, 500))),
                        away=bool(rng.random() < 0.3), ragged=bool(rng.random() < 0.2), offgrid=int(rng.choice([0, 0, 0, 3])))
    fp = None if rng.random() < 0.4 else float(rng.choice([0, 1, 3, 10, 40]))
    s = W.load(lib, wl)
    comm.init_external(s, dist)
    s.shard_round(True)
    W.prepare(s, wl, fairshare_preemption_tokens=fp)
    got = s.schedule_round()
    exchanges += s.shard_exchanges()
    s.close()
    o = W.load(orc, wl); W.prepare(o, wl, fairshare_preemption_tokens=fp); want = o.schedule_round(); o.close()
    try:
        scenario.assert_same_round(want, got)
    except AssertionError as e:
        bad += 1; print("seed", seed, "rank", rank, str(e)[:300], flush=True)
res = [None] * world
dist.all_gather_object(res, bad)
if rank == 0:
    print(f"sharded x{world}: {n - first} seeds, {sum(res)} divergences (over all ranks), {exchanges} exchanges on rank 0, {time.time() - t0:.0f} s")
dist.destroy_process_group()
sys.exit(1 if bad else 0)
