"""One pool, nodes partitioned across ranks (armada_amd/sharded.py): the wide fit queries answered per shard and folded with ONE all-reduce MIN
of a 64-bit order key per query.  CPU: two / three gloo processes over the CPU build of the device code, result == the unsharded oracle.
`-m gpu`: the same on the HIP library (one process per visible GPU; with one GPU the ranks share device 0 and the collective runs on gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from armada_amd import workloads as W
from armada_amd.binding import Library
from armada_amd.sharded import ShardedFit
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
use_gpu = %(gpu)r
if use_gpu:
    import armada_amd
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(rank %% ndev)
    lib = armada_amd.load_library()
    backend = "nccl" if ndev >= world else "gloo"
else:
    lib = Library(os.path.join(%(root)r, "tests", "hostsim", "libhostsim.so"), "asched_")
    backend = "gloo"
dist.init_process_group(backend)
out = {}
for name, wl in (("config2", W.config2(n_nodes=3001, n_jobs=20000)), ("config3", W.config3(n_nodes=2500, n_jobs=9000, n_queues=8, seed=5, occupied=0.9))):
    if use_gpu:
        wl.config.device = rank %% torch.cuda.device_count()
    sf = ShardedFit(lib, wl, rank, world, dist=dist, device=("cuda" if backend == "nccl" else None))
    sf.prepare()
    jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
    for prio in (sf.s.priorities[0], sf.s.priorities[-1]):
        got = sf.fit_select_batch(jobs, prio)
        out[name + ":" + str(prio)] = got.tolist()
    sf.close()
if rank == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _run(tmp_path, world, gpu, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "gpu": gpu})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _reference(oracle_lib):
    from armada_amd import workloads as W
    want = {}
    for name, wl in (("config2", W.config2(n_nodes=3001, n_jobs=20000)), ("config3", W.config3(n_nodes=2500, n_jobs=9000, n_queues=8, seed=5, occupied=0.9))):
        s = W.load(oracle_lib, wl)
        W.prepare(s, wl)
        jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
        for prio in (s.priorities[0], s.priorities[-1]):
            want[name + ":" + str(prio)] = s.fit_select_batch(jobs, prio).tolist()
        s.close()
    return want


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fit_equals_unsharded_oracle(tmp_path, oracle_lib, hostsim_lib, world):
    got = _run(tmp_path, world, False, 29620 + world)
    want = _reference(oracle_lib)
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], k
    assert any(n >= 0 for n in want["config3:-2"]) and any(n < 0 for n in want["config3:-2"])   # both outcomes occur on the 90% occupied input


@pytest.mark.gpu
def test_sharded_fit_on_the_hip_library(tmp_path, oracle_lib):
    got = _run(tmp_path, 2, True, 29631)
    want = _reference(oracle_lib)
    for k in want:
        assert got[k] == want[k], k
