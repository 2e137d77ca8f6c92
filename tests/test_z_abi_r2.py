"""Round-2 additions to the C ABI, each against the oracle on the CPU build of the device code and (`-m gpu`) on the HIP library:
IndexedNodeLabelValues (nodedb.go:340), GetNode / GetNodes read-back (:358-415), single-node Upsert (:1154-1175), floating resources in
whole rounds (gang_scheduler.go:143), argument validation of the NodeDb-level entries, two handles in one process."""
import copy

import numpy as np
import pytest

import scenario
from armada_amd import workloads as W
from armada_amd.binding import ERR_INVALID, SchedError


def _ragged(seed=301):
    return W.small_random(n_nodes=40, n_jobs=500, n_queues=4, seed=seed, occupied=0.8, gangs=3, ragged=True)


def _check_label_values(lib, oracle):
    wl = _ragged()
    a, b = W.load(oracle, wl), W.load(lib, wl)
    assert a.indexed_node_label_values(3) == b.indexed_node_label_values(3) == sorted({v for ls in wl.node_labels for k, v in ls if k == 3})
    assert a.indexed_node_label_values(77) is None and b.indexed_node_label_values(77) is None   # not an indexed label: ok == false


def _check_node_readback(lib, oracle):
    wl = W.small_random(n_nodes=50, n_jobs=700, n_queues=5, seed=41, occupied=0.9, gangs=2)
    hs = []
    for l in (oracle, lib):
        s = W.load(l, wl); W.prepare(s, wl); s.schedule_round(); hs.append(s)
    a, b = hs
    assert np.array_equal(a.get_nodes_alloc(), b.get_nodes_alloc())
    pick = [0, 7, 49, 7]
    assert np.array_equal(a.get_nodes_alloc(pick), b.get_nodes_alloc(pick))
    assert np.array_equal(b.get_nodes_alloc(pick)[1], b.get_alloc(7))
    seen = 0
    for n in range(wl.num_nodes):
        ja, jb = a.get_node_jobs(n), b.get_node_jobs(n)
        assert ja == jb, n
        seen += len(jb)
    assert seen > 100
    with pytest.raises(SchedError):
        b.get_node_jobs(wl.num_nodes)


def _check_node_upsert(lib, oracle):
    """Upsert of one node: the caller changed AllocatableByPriority on its copy (here: half of node 3 given away at every priority, node 5
    drained); the next selections see it"""
    wl = W.small_random(n_nodes=12, n_jobs=60, n_queues=2, seed=77, occupied=0.0, gangs=0)
    out = []
    for l in (oracle, lib):
        s = W.load(l, wl)
        al = s.get_alloc(3)
        al[:, W.CPU] //= 2000; al[:, W.CPU] *= 1000
        s.node_upsert(3, al)
        s.node_upsert(5, np.zeros_like(al))
        assert np.array_equal(s.get_alloc(3), al) and not s.get_alloc(5).any()
        queued = np.nonzero(wl.job_node < 0)[0]
        picks = []
        s.txn_begin()
        ok, pods, _ = s.schedule_many([int(j) for j in queued[:40]])
        picks = [p.node for p in pods]
        s.txn_abort()
        out.append((ok, picks, s.get_nodes_alloc()))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])
    assert 5 not in out[0][1]


def _floating_workload(seed, limit_scale):
    """ephemeral-storage (column 2: not indexed, not a DRF resource) declared a FLOATING resource of the pool: it stops constraining nodes and
    is limited pool-wide instead (sctx.Allocated vs the pool's total)"""
    wl = W.small_random(n_nodes=30, n_jobs=500, n_queues=4, seed=seed, occupied=0.5, gangs=3)
    wl = copy.deepcopy(wl)
    running = wl.job_node >= 0
    limit = int(wl.job_req[running, W.EPH].sum() * limit_scale)
    wl.config.floating_resource_limit = [-1, -1, limit, -1]
    wl.config.floating_counts_in_total = bool(seed % 2)
    return wl


def _check_floating_round(lib, oracle, seed, limit_scale):
    wl = _floating_workload(seed, limit_scale)
    res = []
    for l in (oracle, lib):
        s = W.load(l, wl); W.prepare(s, wl); res.append(s.schedule_round())
        assert not s.get_nodes_alloc()[:, :, W.EPH].any()      # a node holds nothing of a floating resource
    scenario.assert_same_round(res[0], res[1])
    return res[0]


def _check_validation(lib):
    wl = W.small_random(n_nodes=8, n_jobs=40, n_queues=2, seed=3, occupied=0.5, gangs=0)
    s = W.load(lib, wl)
    m, n = wl.num_jobs, wl.num_nodes
    for bad in ([m], [-1], []):
        with pytest.raises(SchedError) as e:
            s.schedule_many(bad)
        assert e.value.code == ERR_INVALID
    with pytest.raises(SchedError):
        s.schedule_many([0], pinned_nodes=[n])
    with pytest.raises(SchedError):
        s.select_node(0, pinned_node=n + 5)
    W.prepare(s, wl)
    with pytest.raises(SchedError):
        s.gang_schedule([m + 3])
    with pytest.raises(SchedError):
        s.add_evicted(0, m, 0)
    with pytest.raises(SchedError):
        s.node_upsert(n, np.zeros((s.P, s.R), np.int64))


# ---------------------------------------------------------------------------------------------------- CPU build of the device code
def test_label_values_cpu_build(hostsim_lib, oracle_lib):
    _check_label_values(hostsim_lib, oracle_lib)


def test_node_readback_cpu_build(hostsim_lib, oracle_lib):
    _check_node_readback(hostsim_lib, oracle_lib)


def test_node_upsert_cpu_build(hostsim_lib, oracle_lib):
    _check_node_upsert(hostsim_lib, oracle_lib)


@pytest.mark.parametrize("seed,scale", [(501, 1.3), (502, 1.05), (503, 0.0), (504, 4.0), (505, 1.5), (506, 1.15)])
def test_floating_rounds_cpu_build(hostsim_lib, oracle_lib, seed, scale):
    r = _check_floating_round(hostsim_lib, oracle_lib, seed, scale)
    if scale == 0.0:    # limits all zero: "floating resources not configured for pool" for every job that asks for some
        assert 20 in set(r.job_unschedulable_reason.tolist())
    if scale in (1.05, 1.15):
        assert 21 in set(r.job_unschedulable_reason.tolist())   # "not enough floating resource"


def test_floating_negative_control(hostsim_lib, oracle_lib):
    """without the declaration the same input schedules differently: the floating column really is taken off the nodes and limited pool-wide"""
    wl = _floating_workload(502, 1.05)
    plain = copy.deepcopy(wl); plain.config.floating_resource_limit = None
    out = []
    for w in (wl, plain):
        s = W.load(oracle_lib, w); W.prepare(s, w); out.append(s.schedule_round())
    assert out[0].scheduled != out[1].scheduled


def test_validation_cpu_build(hostsim_lib):
    _check_validation(hostsim_lib)


def test_validation_oracle_ranges(hostsim_lib):
    """unit sizes beyond the job table used to overrun the scratch gang slot: now ASCHED_ERR_INVALID"""
    wl = W.small_random(n_nodes=4, n_jobs=10, n_queues=1, seed=4, occupied=0.0, gangs=0)
    s = W.load(hostsim_lib, wl)
    with pytest.raises(SchedError) as e:
        s.schedule_many(list(range(wl.num_jobs)) * 3)
    assert e.value.code == ERR_INVALID


# ---------------------------------------------------------------------------------------------------- HIP library
@pytest.mark.gpu
def test_label_values_gpu(hip_lib, oracle_lib):
    _check_label_values(hip_lib, oracle_lib)


@pytest.mark.gpu
def test_node_readback_gpu(hip_lib, oracle_lib):
    _check_node_readback(hip_lib, oracle_lib)


@pytest.mark.gpu
def test_node_upsert_gpu(hip_lib, oracle_lib):
    _check_node_upsert(hip_lib, oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,scale", [(501, 1.3), (502, 1.05), (503, 0.0), (504, 4.0)])
def test_floating_rounds_gpu(hip_lib, oracle_lib, seed, scale):
    _check_floating_round(hip_lib, oracle_lib, seed, scale)


@pytest.mark.gpu
def test_validation_gpu(hip_lib):
    _check_validation(hip_lib)


@pytest.mark.gpu
def test_handles_on_two_devices_in_one_process(hip_lib, oracle_lib):
    """two pools on two GPUs in ONE process (the Go scheduler's situation): every handle owns its stream, events, mailbox and cancel word and
    re-selects its device on every call.  With one visible GPU both handles sit on device 0 (still: separate streams / mailboxes)."""
    import torch
    ndev = torch.cuda.device_count()
    wls = [W.small_random(n_nodes=60, n_jobs=900, n_queues=5, seed=61, occupied=0.9, gangs=2),
           W.small_random(n_nodes=90, n_jobs=700, n_queues=4, seed=62, occupied=0.6, gangs=1, away=True)]
    want = []
    for wl in wls:
        s = W.load(oracle_lib, wl); W.prepare(s, wl); want.append(s.schedule_round())
    hs = []
    for i, wl in enumerate(wls):
        wl.config.device = i % max(ndev, 1)
        hs.append(W.load(hip_lib, wl))
    for rep in range(2):
        for i in (1, 0):
            W.prepare(hs[i], wls[i])
        torch.cuda.set_device(0)    # whatever the caller's current device is, the handles find their own
        got = {i: hs[i].schedule_round() for i in (1, 0)}
        for i in (0, 1):
            scenario.assert_same_round(want[i], got[i])


@pytest.mark.gpu
def test_two_handles_run_rounds_concurrently(hip_lib, oracle_lib):
    """one thread per handle, rounds in flight at the same time (the shared stream / mailbox made this hang): both equal the oracle"""
    import threading
    wls = [W.config3(n_nodes=3000, n_jobs=30000, n_queues=8, seed=71), W.config3(n_nodes=2000, n_jobs=20000, n_queues=6, seed=72, occupied=0.9)]
    for wl in wls:
        wl.global_burst, wl.queue_burst = 6000, 1500
    want = []
    for wl in wls:
        s = W.load(oracle_lib, wl); W.prepare(s, wl); want.append(s.schedule_round())
    hs = [W.load(hip_lib, wl) for wl in wls]
    got = [None, None]

    def run(i):
        for _ in range(3):
            W.prepare(hs[i], wls[i])
            got[i] = hs[i].schedule_round()
    th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a round hung"
    for i in (0, 1):
        scenario.assert_same_round(want[i], got[i])


def test_null_handle_is_refused(hostsim_lib):
    """every int32-returning entry point answers ASCHED_ERR_INVALID to a NULL handle instead of dereferencing it (round-2 advice)"""
    import ctypes as C
    lib = C.CDLL(hostsim_lib.path)
    for name in ("asched_schedule_round", "asched_schedule_queues", "asched_round_prepare", "asched_jobs_set", "asched_nodes_upsert", "asched_set_market", "asched_round_stats"):
        fn = getattr(lib, name)
        fn.restype = C.c_int32
        fn.argtypes = [C.c_void_p, C.c_void_p] + ([C.c_void_p] if name == "asched_jobs_set" else [])
        assert fn(None, None, *([None] if name == "asched_jobs_set" else [])) != 0, name
