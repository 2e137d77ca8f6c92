"""debug helper: one crowded stream round (tests/test_z_stream_runs.py::test_stream_rounds_gpu[2]) on the library named by ASCHED_LIB_PATH / the default; dumps job -> node"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
which = sys.argv[1] if len(sys.argv) > 1 else "hip"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/hc_sched.json"
from armada_amd import workloads as W
from armada_amd.binding import Library
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if which == "hip":
    import torch; torch.cuda.init()
    import armada_amd
    lib = armada_amd.load_library()
elif which == "oracle":
    lib = Library(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
else:
    lib = Library(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"), "asched_")
import test_z_stream_runs as T
seed = int(os.environ.get("DBG_SEED", "2"))
wl = T.workload(900 + seed, gangs=[0, 30, 0][seed], occupied=[0.3, 0.5, 0.9][seed], n_nodes=4000, n_jobs=60000, n_queues=40)
s = W.load(lib, wl); W.prepare(s, wl)
r = s.schedule_round()
st = s.round_stats()
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump({"scheduled": {str(k): int(v) for k, v in r.scheduled.items()}, "preempted": sorted(int(x) for x in r.preempted), "stats": {k: int(v) for k, v in st.items()}}, open(out, "w"))
print(which, len(r.scheduled), "scheduled", len(r.preempted), "preempted", {k: v for k, v in st.items() if not k.startswith("kclk")})
