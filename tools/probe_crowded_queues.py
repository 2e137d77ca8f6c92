"""probe: a CROWDED pool (95 % occupied: most new jobs need preemption) with more than 64 queues — the shape where wide runs (round_wide.h) help least: queues whose head needs
preemption are barriers of a run.  GPU round vs the oracle round on the same input.   python tools/probe_crowded_queues.py [queues]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import armada_amd, bench
from armada_amd import workloads as W
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = W.config3(seed=W.SEED, n_nodes=20_000, n_jobs=200_000, n_queues=nq, occupied=0.95)
wl.global_burst, wl.queue_burst = 40_000, max(1, 40_000 * 8 // nq)
s = W.load(armada_amd.load_library(), wl)
for i in range(2):
    W.prepare(s, wl)
    torch.cuda.synchronize(); t0 = time.perf_counter(); res = s.schedule_round(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = s.round_stats()
base, ores = bench.cpu_baseline(wl, 1e9, res.num_loop_iterations)
print(json.dumps({"queues": nq, "gpu_ms": dt * 1e3, "oracle_ms": base["measured_s"] * 1e3, "ratio": base["measured_s"] / dt, "scheduled": len(res.scheduled), "preempted": len(res.preempted),
                  "iterations": res.num_loop_iterations, "fast_iterations": st["fast_iterations"], "generic_iterations": st["generic_iterations"], "stream_runs": st["stream_runs"], "stream_jobs": st["stream_jobs"],
                  "identical": not bench.round_diff(ores, res)}))
