"""ONE pool's round on several GPUs, exact (SURVEY 8e "Nodes (exact)"; include/armada_sched.h asched_shard_round).

Every rank holds the whole pool and runs the whole round; the round's wide passes over the nodes — the first-fit plane scan (selectNodeForPodAtPriority,
nodedb.go:840-928) and the per-node evaluation of fair-share preemption (selectNodeForJobWithFairPreemption, :935-1043) — look at the rank's 1/world of the node words and
end with one all-reduce MIN of two words on the handle's communicator.  MIN over the shares is the unsharded answer, so every rank must produce the ORACLE's round.

CPU: two / three gloo processes over the CPU build of the device code (the all-reduce is a synchronous call there) — BASELINE configs[4]'s shape (crowded pool, both
preemption kinds), a gang-heavy round, and the same under a forced two-word order key.  `-m gpu`: two processes share device 0, each with its own persistent round
kernel (k_control_wk) posting its words to its host thread, which reduces them over gloo — the device-side protocol and the host proxy; RCCL itself needs two GPUs and is
not exercised on this one-GPU box (unmeasured on more than one GPU).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import scenario
from armada_amd import workloads as W, comm
from armada_amd.binding import Library
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
use_gpu = %(gpu)r
if use_gpu:
    import armada_amd
    torch.cuda.set_device(0)
    lib = armada_amd.load_library()
else:
    lib = Library(os.path.join(%(root)r, "tests", "hostsim", "libhostsim.so"), "asched_")
orc = Library(os.path.join(%(root)r, "oracle", "liboracle.so"), "oracle_")
dist.init_process_group("gloo")
scale = %(scale)r
cases = [("preempt", W.config3(n_nodes=int(1500 * scale), n_jobs=int(12000 * scale), n_queues=8, seed=11, occupied=0.95), None),
         ("gangs", W.config3(n_nodes=int(900 * scale), n_jobs=int(6000 * scale), n_queues=6, seed=12, occupied=0.6, gangs=int(40 * min(scale, 1))), None),
         ("small", W.small_random(n_nodes=70, n_jobs=700, n_queues=5, seed=13, occupied=0.95, gangs=3, away=True, ragged=True), 5.0)]
out = {}
for name, wl, fp in cases:
    for two in (0, 1):
        if two: os.environ["ASCHED_KEY_WORDS"] = "2"
        else: os.environ.pop("ASCHED_KEY_WORDS", None)
        if two and name != "small": continue   # (the two-word key on the small mixed case: twice the exchanges per selection)
        s = W.load(lib, wl)
        comm.init_external(s, dist, device_memory=use_gpu)
        s.shard_round(True)
        if use_gpu: s.set_deadline(120.0)
        W.prepare(s, wl, fairshare_preemption_tokens=fp)
        r = s.schedule_round()
        st = s.round_stats()
        ex = s.shard_exchanges()
        s.close()
        o = W.load(orc, wl); W.prepare(o, wl, fairshare_preemption_tokens=fp)
        ro = o.schedule_round(); o.close()
        try:
            scenario.assert_same_round(ro, r); same = True
        except AssertionError as e:
            same = False; print("DIFF", rank, name, two, str(e)[:300], file=sys.stderr)
        res = [same, len(r.scheduled), len(r.preempted), int(ex), int(st.get("fast_iterations", 0))]
        allres = [None] * world
        dist.all_gather_object(allres, res)
        out[name + (":two" if two else "")] = allres
os.environ.pop("ASCHED_KEY_WORDS", None)
if rank == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _run(tmp_path, world, gpu, port, scale=1):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "gpu": gpu, "scale": scale})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("ASCHED_KEY_WORDS", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _check(res, world):
    for name, per_rank in res.items():
        assert len(per_rank) == world
        for same, nsched, npre, exchanges, _fast in per_rank:
            assert same, (name, per_rank)
            assert exchanges > 0, (name, per_rank)          # the passes really went through the communicator
        assert len({tuple(x[1:3]) for x in per_rank}) == 1, (name, per_rank)
    assert res["preempt"][0][2] > 100, res["preempt"]     # both preemption kinds at work (fair-share evaluation = the MAX half of the exchange)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_round_equals_the_oracle_on_every_rank(tmp_path, hostsim_lib, oracle_lib, world):
    _check(_run(tmp_path, world, False, 29731 + world, scale=0.4), world)


@pytest.mark.gpu
def test_sharded_round_equals_the_oracle_on_every_rank_gpu(tmp_path, hip_lib, oracle_lib):
    """two persistent round kernels on ONE GPU, each waiting for the other's words: the device protocol + the host proxy (gloo between the two processes)"""
    _check(_run(tmp_path, 2, True, 29741, scale=2), 2)


def _direct_pair(hip_lib, wl, fp=None, rounds=2):
    """two replicas of one pool as two handles of this process on device 0, their round kernels exchanging GPU-to-GPU (asched_shard_peers); one thread per handle"""
    import threading
    from armada_amd import workloads as W
    hs = [W.load(hip_lib, wl) for _ in range(2)]
    areas = [h.shard_area()[0] for h in hs]
    for r, h in enumerate(hs):
        h.shard_peers(areas, r)
        h.set_deadline(120.0)
    got, err = [None, None], [None, None]

    def run(i):
        try:
            for _ in range(rounds):
                W.prepare(hs[i], wl, fairshare_preemption_tokens=fp)
                got[i] = hs[i].schedule_round()
        except Exception as e:   # noqa: BLE001
            err[i] = e
    th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=400)
    assert not any(t.is_alive() for t in th), "a sharded round hung"
    assert err == [None, None], err
    ex = [h.shard_exchanges() for h in hs]
    assert ex[0] == ex[1] and ex[0] > 0, ex   # the round kernels counted the same exchanges: the passes went GPU-to-GPU
    for h in hs:
        h.close()
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["preempt", "gangs", "small", "small-two-word"])
def test_sharded_round_gpu_to_gpu_exchange(hip_lib, oracle_lib, monkeypatch, case):
    """the exchange without the host: each round kernel stores its words into the other replica's area and watches its own (here both areas are on the one GPU)"""
    import scenario
    from armada_amd import workloads as W
    if case == "small-two-word":
        monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl, fp = {"preempt": (W.config3(n_nodes=3000, n_jobs=24000, n_queues=8, seed=11, occupied=0.95), None),
              "gangs": (W.config3(n_nodes=1800, n_jobs=12000, n_queues=6, seed=12, occupied=0.6, gangs=40), None),
              "small": (W.small_random(n_nodes=70, n_jobs=700, n_queues=5, seed=13, occupied=0.95, gangs=3, away=True, ragged=True), 5.0),
              "small-two-word": (W.small_random(n_nodes=70, n_jobs=700, n_queues=5, seed=13, occupied=0.95, gangs=3, away=True, ragged=True), 5.0)}[case]
    o = W.load(oracle_lib, wl); W.prepare(o, wl, fairshare_preemption_tokens=fp); want = o.schedule_round(); o.close()
    got = _direct_pair(hip_lib, wl, fp)
    for g in got:
        scenario.assert_same_round(want, g)
    if case == "preempt":
        assert len(want.preempted) > 100


IPC_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import scenario, armada_amd
from armada_amd import workloads as W
from armada_amd.binding import Library
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
lib = armada_amd.load_library()
orc = Library(os.path.join(%(root)r, "oracle", "liboracle.so"), "oracle_")
dist.init_process_group("gloo")
wl = W.config3(n_nodes=3000, n_jobs=24000, n_queues=8, seed=11, occupied=0.95)
s = W.load(lib, wl)
ptr, ipc = s.shard_area()
handles = [None] * world
dist.all_gather_object(handles, ipc)          # (also the barrier the freshly zeroed areas need)
assert any(handles[rank]), "the runtime could not export the exchange area"
areas = [ptr if r == rank else s.shard_open(handles[r]) for r in range(world)]
s.shard_peers(areas, rank)
s.set_deadline(120.0)
dist.barrier()
for _ in range(2):
    W.prepare(s, wl); r = s.schedule_round()
o = W.load(orc, wl); W.prepare(o, wl); ro = o.schedule_round(); o.close()
try:
    scenario.assert_same_round(ro, r); same = True
except AssertionError as e:
    same = False; print("DIFF", rank, str(e)[:300], file=sys.stderr)
res = [None] * world
dist.all_gather_object(res, [same, len(r.scheduled), len(r.preempted)])
dist.barrier()
s.close()
if rank == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_sharded_round_gpu_to_gpu_exchange_across_processes(tmp_path, hip_lib, oracle_lib):
    """the replicas as two PROCESSES (the deployment shape: one process per GPU): each maps the other's exchange area through its hipIpc handle (asched_shard_open)"""
    script = tmp_path / "ipc_worker.py"
    script.write_text(IPC_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ASCHED_KEY_WORDS", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29751", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert all(x[0] for x in res) and res[0][2] > 100 and res[0] == res[1], res


def test_shard_round_preconditions(hostsim_lib):
    """no communicator -> INVALID; wall-clock time budgets -> UNSUPPORTED (the replicas' control flow must be a function of their inputs alone)"""
    from armada_amd import workloads as W
    from armada_amd.binding import SchedError
    wl = W.small_random(n_nodes=20, n_jobs=200, n_queues=3, seed=5)
    s = W.load(hostsim_lib, wl)
    with pytest.raises(SchedError) as e:
        s.shard_round(True)
    assert e.value.code == -1 and "communicator" in str(e.value)
    s.close()
    wl.config.max_new_job_scheduling_duration_ns = 10 ** 9
    s = W.load(hostsim_lib, wl)
    s.comm_init_external(lambda ptr, count, op: 0, 0, 2)
    with pytest.raises(SchedError) as e:
        s.shard_round(True)
    assert e.value.code == -2 and "clock" in str(e.value)
    s.close()
