"""one-off: BASELINE configs[4] (oversubscribed / preemption-heavy) at the FULL size — 100 000 nodes x 64 queues x 1 000 000 queued jobs (+ ~918 000 running) — GPU round against the
oracle round on the same input, field by field (bench.round_diff) plus the exclusion histograms of a sample of the failed jobs.  The driver-run bench line checks this shape at the
reduced size and at 100 000 nodes with an oracle-sized burst; the oracle round of the full input takes minutes, so it is run here once per round of work (profiles/r04z_config4_full_parity.txt)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import armada_amd, bench
from armada_amd import workloads as W
from armada_amd.binding import Library
wl = W.config3(seed=W.SEED, n_nodes=100_000, n_jobs=1_000_000, n_queues=64, occupied=0.95)
hip = armada_amd.load_library()
s = W.load(hip, wl); W.prepare(s, wl)
torch.cuda.synchronize(); t0 = time.perf_counter(); res = s.schedule_round(); torch.cuda.synchronize(); gpu_s = time.perf_counter() - t0
print("gpu round", round(gpu_s, 2), "s", len(res.scheduled), "scheduled", len(res.preempted), "preempted", res.num_loop_iterations, "iterations", flush=True)
base, ores = bench.cpu_baseline(wl, 1e9, res.num_loop_iterations)
par = bench.parity_record(res, ores, wl.num_jobs, "oracle round on the same full-size input")
par["excluded_nodes"] = bench.excluded_parity(s, ores)
print(json.dumps({"gpu_s": gpu_s, "oracle_s": base["measured_s"], "oracle_rounds": base["rounds"], "parity": par, "workload": f"{wl.num_nodes} nodes x {wl.num_queues} queues x 1000000 queued (+{wl.num_jobs - 1_000_000} running), 95 % occupied"}))
