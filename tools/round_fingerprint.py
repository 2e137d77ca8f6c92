"""one preemption-heavy round (BASELINE configs[4] shape at 20k x 200k, tools/prof_config4.py's input) on the library named by ASCHED_LIB_PATH: prints the
counts and a fingerprint of the whole result, so that builds of the same sources can be compared without an oracle round each"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import armada_amd
from armada_amd import workloads as W
wl = W.config3(seed=W.SEED, n_nodes=20_000, n_jobs=200_000, n_queues=32, occupied=0.95)
wl.global_burst, wl.queue_burst = 40_000, 4_000
s = W.load(armada_amd.load_library(), wl)
W.prepare(s, wl)
t = time.perf_counter(); r = s.schedule_round(); dt = time.perf_counter() - t
h = hashlib.sha256()
for d in (r.scheduled, r.scheduled_priority, r.scheduled_method, r.preempted):
    h.update(np.array(sorted(d.items()), dtype=np.int64).tobytes())
h.update(np.ascontiguousarray(r.queue_allocated_by_pc).tobytes())
print("round", round(dt * 1e3, 1), "ms scheduled", len(r.scheduled), "preempted", len(r.preempted), "ev1", r.num_evicted_phase1, "ev3", r.num_evicted_phase3, "fingerprint", h.hexdigest()[:16])
