"""The handle's own communicator (include/armada_sched.h "The communicator", round 4): the collectives of the two single-pool multi-GPU modes run INSIDE
the library on the handle's stream — RCCL (asched_comm_init) in the product, any transport through asched_comm_init_external.
CPU: world size 2 / 3 over the CPU build of the device code with gloo as the external transport: asched_fit_select_batch_sharded == the unsharded oracle query by
query; asched_round_exchange == the caller-driven round_delta / all_reduce / round_delta_resolve sequence it replaces.
`-m gpu`: a real RCCL communicator of world size 1 on the MI355X (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on the handle's stream) and the external
transport with device buffers; the multi-rank RCCL path needs more than one GPU and is the driver's to run."""
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
from armada_amd.queuehash import QueueHashRound
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
lib = Library(os.path.join(%(root)r, "tests", "hostsim", "libhostsim.so"), "asched_")
dist.init_process_group("gloo")
out = {}
for name, wl in (("config2", W.config2(n_nodes=3001, n_jobs=20000)), ("config3", W.config3(n_nodes=2500, n_jobs=9000, n_queues=8, seed=5, occupied=0.9))):
    sf = ShardedFit(lib, wl, rank, world, dist=dist)
    sf.comm_init("external")
    assert sf.s.comm_rank() == (rank, world)
    sf.prepare()
    jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
    for prio in (sf.s.priorities[0], sf.s.priorities[-1]):
        out[name + ":" + str(prio)] = sf.fit_select_batch(jobs, prio).tolist()
    sf.s.comm_destroy()
    assert sf.s.comm_rank() == (0, 1)
    sf.close()
# queue-hash round: the in-library exchange gives what the caller-driven sequence gives
wl = W.config3(n_nodes=600, n_jobs=8000, n_queues=12, seed=11, occupied=0.85)
wl.global_burst, wl.queue_burst = 3000, 600
res = []
for inlib in (False, True):
    qh = QueueHashRound(lib, wl, rank, world, dist=dist)
    if inlib:
        qh.comm_init("external")
    r = qh.run(); qh.close()
    res.append(dict(scheduled=sorted(r["scheduled"].items()), dropped=r["dropped"], conflicts=r["conflicts"], accepted=r["accepted"], replay_set=r["replay_set"], preempted=r["preempted"]))
out["queuehash_same"] = res[0] == res[1]
out["queuehash_accepted"] = res[1]["accepted"]
if rank == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _run(tmp_path, world, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
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
def test_in_library_collectives_over_an_external_transport(tmp_path, oracle_lib, hostsim_lib, world):
    got = _run(tmp_path, world, 29640 + world)
    want = _reference(oracle_lib)
    for k in want:
        assert got[k] == want[k], k
    assert got["queuehash_same"] and got["queuehash_accepted"] > 0


def test_comm_entry_points_refuse_bad_arguments(hostsim_lib):
    from armada_amd import workloads as W
    from armada_amd.binding import SchedError
    wl = W.config2(n_nodes=50, n_jobs=200)
    s = W.load(hostsim_lib, wl)
    assert s.comm_rank() == (0, 1)
    with pytest.raises(SchedError):
        s.comm_init(bytes(128), 2, 2)        # rank out of range
    with pytest.raises(SchedError):
        s.comm_init(bytes(128), 0, 1)        # the CPU build has no RCCL: refused, never faked
    with pytest.raises(SchedError):
        s.fit_select_batch_sharded(np.array([10 ** 6], dtype=np.int32), s.priorities[0], [4] * len(wl.config.indexed_col), 8)   # job out of range
    s.close()


@pytest.mark.gpu
def test_rccl_communicator_inside_the_library(hip_lib, oracle_lib):
    """world size 1 on the one GPU of the box: the real ncclGetUniqueId / ncclCommInitRank / ncclAllReduce path on the handle's stream"""
    from armada_amd import workloads as W
    wl = W.config3(n_nodes=2500, n_jobs=9000, n_queues=8, seed=5, occupied=0.9)
    s = W.load(hip_lib, wl); W.prepare(s, wl)
    o = W.load(oracle_lib, wl); W.prepare(o, wl)
    uid = s.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    s.comm_init(uid, 0, 1)
    assert s.comm_rank() == (0, 1)
    jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
    cap = wl.node_total if wl.node_allocatable is None else wl.node_allocatable
    width = [max(1, int(int(cap[:, c].max(initial=0)) // int(r) + 1).bit_length()) for c, r in zip(wl.config.indexed_col, wl.config.indexed_resolution)]
    row_bits = max(1, int(wl.num_nodes - 1).bit_length())
    for prio in (s.priorities[0], s.priorities[-1]):
        got = s.fit_select_batch_sharded(jobs, prio, width, row_bits)
        assert got.tolist() == o.fit_select_batch(jobs, prio).tolist()
    # the queue-hash exchange through the same communicator: at world size 1 it reproduces the plain sequence
    r = s.schedule_round()
    assert len(r.scheduled) > 0
    import torch
    buf = torch.zeros(max(s.round_delta_words(), 1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    s.round_delta(buf.data_ptr())
    a = s.round_delta_resolve(buf.data_ptr())
    b = s.round_exchange()
    assert a[0] == b[0] and all((x == y).all() for x, y in zip(a[1:], b[1:]))
    s.comm_destroy()
    s.close(); o.close()


@pytest.mark.gpu
def test_external_transport_sees_device_memory(hip_lib, oracle_lib):
    from armada_amd import workloads as W
    wl = W.config2(n_nodes=3001, n_jobs=20000)
    s = W.load(hip_lib, wl); W.prepare(s, wl)
    o = W.load(oracle_lib, wl); W.prepare(o, wl)
    seen = []
    import torch

    def allreduce(ptr, count, op):
        t = torch.empty(count, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        from armada_amd.comm import hip_memcpy
        assert hip_memcpy(t.data_ptr(), ptr, count * 8, 3) == 0   # device -> device: the words ARE device memory, complete, stream idle
        seen.append((count, op, int((t != 2 ** 63 - 1).sum())))
        return 0
    s.comm_init_external(allreduce, 0, 1)
    jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)[:5000]
    cap = wl.node_total if wl.node_allocatable is None else wl.node_allocatable
    width = [max(1, int(int(cap[:, c].max(initial=0)) // int(r) + 1).bit_length()) for c, r in zip(wl.config.indexed_col, wl.config.indexed_resolution)]
    got = s.fit_select_batch_sharded(jobs, s.priorities[0], width, max(1, int(wl.num_nodes - 1).bit_length()))
    assert got.tolist() == o.fit_select_batch(jobs, s.priorities[0]).tolist()
    assert seen and seen[0][0] == len(jobs) and seen[0][1] == 1 and seen[0][2] == int((got >= 0).sum())
    s.close(); o.close()


FAIL_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from armada_amd import workloads as W
from armada_amd.binding import Library, SchedError
from armada_amd.sharded import ShardedFit
from armada_amd.queuehash import QueueHashRound
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
lib = Library(os.path.join(%(root)r, "tests", "hostsim", "libhostsim.so"), "asched_")
dist.init_process_group("gloo")
codes = {}
wl = W.config3(n_nodes=600, n_jobs=4000, n_queues=6, seed=3, occupied=0.8)
sf = ShardedFit(lib, wl, rank, world, dist=dist)
sf.comm_init("external")
sf.prepare()
jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
good = sf.fit_select_batch(jobs, sf.s.priorities[0])            # a healthy collective first
# (1) a layout whose fields are too narrow on ONE rank only: that rank's words do not fit, the other rank's do
bits = list(sf.width) if rank != world - 1 else [1] * len(sf.width)
try:
    sf.s.fit_select_batch_sharded(jobs, sf.s.priorities[0], bits, sf.row_bits, rank_offset=sf.lo)
    codes["sharded"] = 0
except SchedError as e:
    codes["sharded"] = e.code
codes["still_works"] = bool((sf.fit_select_batch(jobs, sf.s.priorities[0]) == good).all())   # the communicator is still usable: the same healthy collective again
# (2) the queue-hash exchange with no round result on ONE rank
if rank != world - 1:
    sf.s.schedule_round()
try:
    sf.s.round_exchange()
    codes["exchange"] = 0
except SchedError as e:
    codes["exchange"] = e.code
again = sf.fit_select_batch(jobs, sf.s.priorities[0])   # (node state differs between the ranks now: only that the collective completes and every rank holds the same words)
codes["after_exchange"] = int(np.asarray(again, dtype=np.int64).sum())
sf.close()
allc = [None] * world
dist.all_gather_object(allc, codes)
if rank == 0:
    print("RESULT " + json.dumps(allc))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_local_failure_fails_every_rank_instead_of_hanging(tmp_path, hostsim_lib, world):
    """round-4 advisor, medium: a precondition that fails on one rank only used to return before the all-reduce while the peers were already inside it (a hang under RCCL).
    Now one status word is all-reduced first: the failing rank returns its own error, every other rank ASCHED_ERR_PEER, and the communicator stays usable."""
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29660 + world), str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    codes = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    from armada_amd.binding import ERR_INVALID, ERR_PEER
    for r, c in enumerate(codes):
        want = ERR_INVALID if r == world - 1 else ERR_PEER
        assert c["sharded"] == want and c["exchange"] == want and c["still_works"] and c["after_exchange"] == codes[0]["after_exchange"], (r, c)
