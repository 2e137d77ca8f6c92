"""the BenchmarkPreemptingQueueScheduler-shaped sub-record of bench.py alone (GPU box)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import armada_amd, bench
args = argparse.Namespace(cpu_budget=10.0, other_scale=1.0)
rec = bench.reference_benchmark_record(armada_amd.load_library(), args)
for r in rec["rows"]:
    print(r["shape"], "gpu %.2f ms" % r["gpu_ms"], "oracle %.2f ms" % r.get("oracle_ms", 0), r.get("identical"), r["steady_state"])
