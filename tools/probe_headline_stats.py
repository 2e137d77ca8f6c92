import sys, os, time
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
import armada_amd
from armada_amd import workloads as W
wl = W.config3(seed=W.SEED, n_nodes=100_000, n_jobs=1_000_000, n_queues=64)
lib = armada_amd.load_library()
s = W.load(lib, wl); W.prepare(s, wl)
t = time.perf_counter(); r = s.schedule_round(); dt = time.perf_counter() - t
st = s.round_stats()
print("round ms", round(dt*1e3,1), "iters", r.num_loop_iterations, "ev1", r.num_evicted_phase1, {k: v for k, v in st.items() if not k.startswith('kclk')})
