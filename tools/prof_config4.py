"""profiling helper: one preemption-heavy round (BASELINE configs[4] shape at 20k x 200k) on the library named by ASCHED_LIB_PATH; with ASCHED_PRINT_SEG=1 and the
profiling build (tools/build_prof.sh) the segment clocks of the generic iteration are printed by round_stats()"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()   # torch's bundled HIP runtime before the library's (tests/conftest.py)
import armada_amd
from armada_amd import workloads as W
kw = dict(n_nodes=20_000, n_jobs=200_000, n_queues=32, occupied=0.95)
if len(sys.argv) > 1 and sys.argv[1] == "gangs": kw = dict(n_nodes=20_000, n_jobs=200_000, n_queues=32, gangs=2_000)
if len(sys.argv) > 1 and sys.argv[1].startswith("q") and sys.argv[1][1:].isdigit(): kw = dict(n_nodes=20_000, n_jobs=200_000, n_queues=int(sys.argv[1][1:]))   # bench.py's "256 queues" sub-record shape (q256), or any queue count
full = len(sys.argv) > 1 and sys.argv[1] in ("full", "gangsfull", "headline")
if full: kw = dict(n_nodes=100_000, n_jobs=1_000_000, n_queues=64, occupied=0.95)
if len(sys.argv) > 1 and sys.argv[1] == "gangsfull": kw = dict(n_nodes=100_000, n_jobs=1_000_000, n_queues=64, gangs=10_000)   # BASELINE configs[3]
if len(sys.argv) > 1 and sys.argv[1] == "headline": kw = dict(n_nodes=100_000, n_jobs=1_000_000, n_queues=64)                  # BASELINE configs[2]
checker = len(sys.argv) > 1 and sys.argv[1] == "checker"   # bench.py's configs[4] checker: 100 000 nodes 95 % occupied, the reference's default burst of 1 000 (the production-shaped round: evicted jobs returning, few new ones)
if checker: kw = dict(n_nodes=100_000, n_jobs=300_000, n_queues=64, occupied=0.95)
wl = W.config3(seed=W.SEED, **kw)
if checker: wl.global_burst, wl.queue_burst = 1_000, 1_000; wl.config.max_queue_lookback = 100_000   # (bench.py config4_checker_record: the reference's shipped limits)
elif not full: wl.global_burst, wl.queue_burst = 40_000, 4_000
lib = armada_amd.load_library()
s = W.load(lib, wl)
for i in range(1 if full else 2):
    W.prepare(s, wl)
    t = time.perf_counter(); r = s.schedule_round(); dt = time.perf_counter() - t
    st = s.round_stats()
    print("round", i, round(dt * 1e3, 1), "ms", {k: st[k] for k in ("fast_iterations", "generic_iterations", "kclk_pass1", "kclk_pass2", "kclk_plane_scans", "kclk_fair_selects", "ft_queries", "ft_retries", "ft_node_updates", "stream_runs", "stream_jobs", "window_refills", "preempt_fast_iterations", "kclk_replay", "fast_replay_steps")}, len(r.scheduled), len(r.preempted), flush=True)

import ctypes
if hasattr(lib.lib, "asched_debug_help_trace"):   # tools/build_variant.sh trace -DHELP_TRACE: the timeline of the fused wide pass (wall clock, 10 ns units)
    buf = (ctypes.c_uint64 * 64)(); lib.lib.asched_debug_help_trace(buf)
    names = {0: ["own share done", "wait done"], 1: ["seen", "args+acquire", "scan done", "fair done", "slot written"]}
    for cls, label in ((0, "control workgroup"), (1, "helper 1"), (2, "helper H/2"), (3, "helper H")):
        n = buf[56 + cls]
        if n: print(label, n, "passes:", ", ".join(f"{nm} {buf[cls * 8 + i] / n / 100:.2f} us" for i, nm in enumerate(names[0 if cls == 0 else 1])))
