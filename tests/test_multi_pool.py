"""N>1 path on CPU: two processes (gloo), one pool each, no data-path collective.  The backend here is the CPU oracle
(test infrastructure) — on the GPU box bench.py drives the HIP library through the same armada_amd.multipool code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from armada_amd import multipool, workloads as W
from armada_amd.binding import Library
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
oracle = Library(os.path.join(%(root)r, "oracle", "liboracle.so"), "oracle_")
wl = W.small_random(n_nodes=40, n_jobs=500, n_queues=4, seed=multipool.pool_seed(11, rank), occupied=0.6, gangs=2)
s = W.load(oracle, wl)
lat, dev_ms, res = multipool.timed_rounds(s, wl, steps=2, warmup=1, barrier=dist.barrier)
def amax(x):
    t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); return float(t.item())
value, total = multipool.aggregate(world, 2, sum(lat), amax)
gathered = [None] * world
dist.all_gather_object(gathered, {"rank": rank, "scheduled": sorted(res.scheduled.items()), "preempted": sorted(res.preempted.items()), "t": sum(lat)})
if rank == 0:
    print("RESULT " + json.dumps({"value": value, "total": total, "ranks": gathered}))
dist.destroy_process_group()
'''


def test_two_pools_two_processes(tmp_path, oracle_lib):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29613", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert len(r["ranks"]) == 2
    # whole-job value = pool-rounds of all ranks / slowest rank
    assert abs(r["total"] - max(x["t"] for x in r["ranks"])) < 1e-9
    assert abs(r["value"] - 2 * 2 / r["total"]) < 1e-9
    # every rank scheduled its OWN pool: identical to a single-process run of the same pool
    from armada_amd import multipool, workloads as W
    for x in r["ranks"]:
        wl = W.small_random(n_nodes=40, n_jobs=500, n_queues=4, seed=multipool.pool_seed(11, x["rank"]), occupied=0.6, gangs=2)
        s = W.load(oracle_lib, wl)
        W.prepare(s, wl)
        ref = s.schedule_round()
        assert sorted(ref.scheduled.items()) == [tuple(t) for t in x["scheduled"]]
        assert sorted(ref.preempted.items()) == [tuple(t) for t in x["preempted"]]
    assert r["ranks"][0]["scheduled"] != r["ranks"][1]["scheduled"]  # different pools
