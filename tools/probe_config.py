import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import armada_amd
from armada_amd import workloads as W
hip = armada_amd.load_library()
which = sys.argv[1]; n=int(sys.argv[2]); m=int(sys.argv[3]); q=int(sys.argv[4])
if which == 'gangs':
    wl = W.config3(n_nodes=n, n_jobs=m, n_queues=q, gangs=int(sys.argv[5]))
else:
    wl = W.config3(n_nodes=n, n_jobs=m, n_queues=q, occupied=float(sys.argv[5]))
scale = m/1e6
wl.global_burst, wl.queue_burst = max(1,int(200000*scale)), max(1,int(20000*scale))
s = W.load(hip, wl)
for i in range(2):
    W.prepare(s, wl); t=time.time(); r = s.schedule_round(); dt=time.time()-t
print(which, 'round_s', dt, 'sched', len(r.scheduled), 'pre', len(r.preempted), 'iters', r.num_loop_iterations, 'queries', r.num_node_queries, 'ev1', r.num_evicted_phase1, 'ev3', r.num_evicted_phase3, s.round_stats())
