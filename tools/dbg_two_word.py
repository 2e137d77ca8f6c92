"""two-word order keys on the device, step by step (a crash names the step): python tools/dbg_two_word.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import scenario
import armada_amd
from armada_amd import workloads as W
from armada_amd.binding import Library
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orc = Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
hip = armada_amd.load_library()
def run(tag, wl, fp=None):
    print("==", tag, flush=True)
    out = []
    for lib in (orc, hip):
        t0 = time.time()
        s = W.load(lib, wl); print("  loaded", flush=True)
        W.prepare(s, wl, fairshare_preemption_tokens=fp); print("  prepared", flush=True)
        r = s.schedule_round(); print("  round", round(time.time() - t0, 2), "s", len(r.scheduled), len(r.preempted), flush=True)
        out.append(r); s.close()
    try:
        scenario.assert_same_round(out[0], out[1]); print("  identical", flush=True)
    except AssertionError as e:
        print("  DIFF", str(e)[:200], flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "forced"):
    os.environ["ASCHED_KEY_WORDS"] = "2"
    for seed in range(4):
        run(f"forced small {seed}", W.small_random(n_nodes=10 + seed * 7, n_jobs=300 + seed * 60, n_queues=2 + seed, seed=7000 + seed, occupied=[0.4, 0.8, 0.95, 1.0][seed % 4], gangs=seed % 4))
    del os.environ["ASCHED_KEY_WORDS"]
if which in ("all", "fine"):
    for n, m in ((600, 6000), (2000, 20000), (20000, 100000)):
        run(f"fine {n}x{m}", W.fine_indexed(n_nodes=n, n_jobs=m, n_queues=8 if n < 20000 else 32, occupied=0.5))
    run("fine crowded 20000x40000", W.fine_indexed(n_nodes=20000, n_jobs=40000, n_queues=32, occupied=0.97))
