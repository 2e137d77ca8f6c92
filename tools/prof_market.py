"""the market-driven round of bench.py's sub-record alone (ASCHED_LIB_PATH + ASCHED_PRINT_SEG=1 for the profiling build's generic-iteration segments):
   python tools/prof_market.py [repeats]"""
import json, os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
import armada_amd, bench

from armada_amd.binding import Scheduler
_round = Scheduler.schedule_round
def _round_and_stats(self, *a, **k):
    r = _round(self, *a, **k)
    if os.environ.get("ASCHED_PRINT_SEG"): print("round_stats", self.round_stats(), "control ms", self.kernel_times(), file=sys.stderr)   # (the library prints the segment clocks from round_stats)
    return r
Scheduler.schedule_round = _round_and_stats

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hip = armada_amd.load_library()
args = types.SimpleNamespace(other_scale=1.0, cpu_budget=float(os.environ.get("CPU_BUDGET", "0")))
for i in range(reps):
    rec = bench.market_record(hip, args)
    print(json.dumps({k: rec[k] for k in ("ms_per_step", "round", "pricer", "cpu_baseline", "parity") if k in rec}), flush=True)
