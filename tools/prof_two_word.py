"""the two-word-key pool of bench.py's sub-record alone (for rocprofv3: k_control_wk, k_bulk_wk, k_fit_batch_wk): python tools/prof_two_word.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
import numpy as np
import armada_amd
from armada_amd import workloads as W
hip = armada_amd.load_library()
wl = W.fine_indexed(n_nodes=20_000, n_jobs=100_000, n_queues=32, occupied=0.5)
s = W.load(hip, wl)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    W.prepare(s, wl); torch.cuda.synchronize(); t0 = time.perf_counter(); r = s.schedule_round(); dt = time.perf_counter() - t0
    print(f"round {i}: {dt * 1e3:.1f} ms scheduled {len(r.scheduled_job)}", s.kernel_times())
jobs = np.concatenate(wl.queued)[:50_000]
t0 = time.perf_counter(); out = s.fit_select_batch(jobs, s.priorities[0]); dt = time.perf_counter() - t0
print(f"fit_select_batch of {len(jobs)} queries on the two-word key: {dt * 1e3:.2f} ms, {int((out >= 0).sum())} with a node", s.kernel_times())
s.close()
