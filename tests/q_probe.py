"""GPU-vs-oracle check of one seeded round at several queue counts (debugging aid for the Q > 64 path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import armada_amd, scenario
from armada_amd import workloads as W
from armada_amd.binding import Library
hip = armada_amd.load_library(); orc = Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
for nq in [int(x) for x in sys.argv[1:]] or [64, 65, 70, 130]:
    wl = W.small_random(n_nodes=30, n_jobs=700, n_queues=nq, seed=13, occupied=0.7, gangs=2)
    res = []
    for lib in (orc, hip):
        s = W.load(lib, wl); W.prepare(s, wl); res.append(s.schedule_round())
    try:
        scenario.assert_same_round(res[0], res[1]); print(nq, "same", len(res[0].scheduled), len(res[0].preempted), flush=True)
    except AssertionError as e:
        print(nq, "MISMATCH", str(e)[:200], "iters", res[0].num_loop_iterations, res[1].num_loop_iterations, flush=True)
