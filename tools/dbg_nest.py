"""round 5 debugging aid: the soak `rounds` workloads of the given seeds through the library at ASCHED_LIB_PATH and through the oracle — field diff + the counters of both.
   python tools/dbg_nest.py 100345 102465"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
if torch.cuda.is_available():
    torch.cuda.init()
import armada_amd, bench
from armada_amd import workloads as W
from armada_amd.binding import Library
from test_z_stream_runs import _soak_round_workload
from conftest import _oracle_path
lib = armada_amd.load_library(); orc = Library(_oracle_path(), "oracle_")
for seed in [int(x) for x in sys.argv[1:]]:
    wl = _soak_round_workload(seed); out = []
    for l in (lib, orc):
        s = W.load(l, wl); W.prepare(s, wl); r = s.schedule_round(); out.append((r, s.round_stats())); s.close()
    keys = ("fast_iterations", "generic_iterations", "stream_runs", "stream_jobs", "stream_emitted", "stream_prepared", "preempt_fast_iterations", "window_refills")
    print(seed, "diff", bench.round_diff(out[0][0], out[1][0]), "scheduled", len(out[0][0].scheduled), len(out[1][0].scheduled), {k: out[0][1].get(k) for k in keys})
