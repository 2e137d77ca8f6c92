"""ONE pool's round with its wide passes split over the ranks (asched_shard_round), the ranks sharing device 0, gloo between them: what an exchange costs through the host proxy.
   python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29761 tools/probe_sharded_round.py [nodes jobs occupied]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
import armada_amd
from armada_amd import workloads as W, comm
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
nodes, jobs, occ = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (20_000, 200_000, 0.95)
torch.cuda.set_device(0)
lib = armada_amd.load_library()
if world > 1: dist.init_process_group("gloo")
wl = W.config3(n_nodes=nodes, n_jobs=jobs, n_queues=64, seed=W.SEED, occupied=occ)
wl.global_burst, wl.queue_burst = max(1, jobs // 5), max(1, jobs // 50)
def run(shard):
    s = W.load(lib, wl)
    if shard: comm.init_external(s, dist, device_memory=True); s.shard_round(True)
    s.set_deadline(300.0)
    out = []
    for i in range(3):
        W.prepare(s, wl)
        if world > 1: dist.barrier()
        t0 = time.perf_counter(); r = s.schedule_round(); dt = time.perf_counter() - t0
        out.append((dt, s.shard_exchanges() if shard else 0))
    st = s.round_stats(); s.close()
    return r, out, st
r0, t0, st0 = run(False)
if world > 1:
    r1, t1, st1 = run(True)
    same = np.array_equal(r0.scheduled_job, r1.scheduled_job) and np.array_equal(r0.scheduled_node, r1.scheduled_node) and np.array_equal(r0.preempted_job, r1.preempted_job)
    print(f"rank {rank}: unsharded (both ranks' rounds side by side on one GPU) {[round(x[0] * 1e3, 1) for x in t0]} ms; sharded x{world} {[round(x[0] * 1e3, 1) for x in t1]} ms, "
          f"{t1[-1][1]} exchanges -> {(t1[-1][0] - t0[-1][0]) / max(t1[-1][1], 1) * 1e6:.1f} us per exchange over the unsharded round; identical {same}; "
          f"scheduled {len(r1.scheduled_job)} preempted {len(r1.preempted_job)} kclk plane scans {st1.get('kclk_plane_scans')} fair selects {st1.get('kclk_fair_selects')}", flush=True)
    dist.destroy_process_group()
else:
    print(f"unsharded {[round(x[0] * 1e3, 1) for x in t0]} ms scheduled {len(r0.scheduled_job)} preempted {len(r0.preempted_job)}")
