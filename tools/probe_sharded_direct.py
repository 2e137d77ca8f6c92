"""two replicas of ONE pool as two handles on device 0, wide passes split two ways, GPU-to-GPU exchange (asched_shard_peers): round time against the unsharded round
   python tools/probe_sharded_direct.py [nodes jobs occupied]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.set_device(0)
import armada_amd
from armada_amd import workloads as W
lib = armada_amd.load_library()
nodes, jobs, occ = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (20_000, 200_000, 0.95)
wl = W.config3(n_nodes=nodes, n_jobs=jobs, n_queues=64, seed=W.SEED, occupied=occ)
wl.global_burst, wl.queue_burst = max(1, jobs // 5), max(1, jobs // 50)
def pair(shard):
    hs = [W.load(lib, wl) for _ in range(2)]
    if shard:
        areas = [h.shard_area()[0] for h in hs]
        for r, h in enumerate(hs): h.shard_peers(areas, r)
    for h in hs: h.set_deadline(300.0)
    times, res = [[], []], [None, None]
    def run(i):
        for _ in range(3):
            W.prepare(hs[i], wl); t0 = time.perf_counter(); res[i] = hs[i].schedule_round(); times[i].append(time.perf_counter() - t0)
    th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
    [t.start() for t in th]; [t.join() for t in th]
    st = hs[0].round_stats()
    [h.close() for h in hs]
    return res, times, st
r0, t0, st0 = pair(False)
r1, t1, st1 = pair(True)
same = all(np.array_equal(r0[0].scheduled_job, r.scheduled_job) and np.array_equal(r0[0].scheduled_node, r.scheduled_node) and np.array_equal(r0[0].preempted_job, r.preempted_job) for r in r1)
passes = st1.get("plane_scans", 0) or 0
print(f"{nodes} nodes x {jobs} jobs occupied {occ}: two whole rounds side by side {[round(x * 1e3, 1) for x in t0[0]]} ms; two replicas, passes split, GPU-to-GPU exchange {[round(x * 1e3, 1) for x in t1[0]]} ms; "
      f"identical {same}; scheduled {len(r1[0].scheduled_job)} preempted {len(r1[0].preempted_job)}; kclk plane scans {st0.get('kclk_plane_scans')} -> {st1.get('kclk_plane_scans')}, fair selects {st0.get('kclk_fair_selects')} -> {st1.get('kclk_fair_selects')}")
