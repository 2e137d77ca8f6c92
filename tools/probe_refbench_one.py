"""one BenchmarkPreemptingQueueScheduler shape's steady-state round, 5 times (for rocprofv3 --kernel-trace --stats):  python tools/probe_refbench_one.py [nodes queues jobs_per_queue]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch; torch.cuda.init()
import armada_amd
from armada_amd import workloads as W
nn, nq, per = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 1, 320000)
wl = W.reference_benchmark(nn, nq, per)
s = W.load(armada_amd.load_library(), wl); W.prepare(s, wl)
first = s.schedule_round()
node = wl.job_node.copy(); prio = wl.job_run_prio.copy()
node[first.scheduled_job] = first.scheduled_node; prio[first.scheduled_job] = first.scheduled_priority_arr
W.set_jobs(s, wl, job_node=node.astype(np.int32), job_run_prio=prio.astype(np.int32))
queued = [np.array([int(j) for j in q if node[int(j)] < 0], dtype=np.int32) for q in wl.queued]
for _ in range(5):
    s.round_prepare(wl.queue_weight, queued)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = s.schedule_round(); torch.cuda.synchronize()
    print("round ms", (time.perf_counter() - t0) * 1e3, r.num_loop_iterations, s.round_timing())
