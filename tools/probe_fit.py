"""bench.py's nodedb fit-kernel sub-records alone (GPU box): BASELINE configs[1] and the same kernel at 100 000 nodes x 1 000 000 queries"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import armada_amd, bench
args = argparse.Namespace(cpu_budget=10.0, other_scale=1.0, steps=10)
for big in (False, True):
    r = bench.fit_batch_record(armada_amd.load_library(), args, big=big)
    print(json.dumps({k: r[k] for k in ("config", "value", "host_ms", "device_ms", "queries_issued", "passes_executed", "parity")}), r["roofline"]["frac"], r["roofline"]["achieved"])
