"""The words of the multi-GPU exchanges (include/armada_sched.h "one pool on several GPUs"; armada_amd/csrc/mgpu.h): produced and consumed by the library —
on the device in the HIP build (armada_sched_mgpu.hip), by the same per-element functions in serial loops in the CPU build — and restated plainly in the oracle.
The multi-process behaviour is tests/test_sharded_fit.py and tests/test_queuehash.py; here the words themselves are compared between the implementations, and the
HIP library is driven through BOTH kinds of buffer it accepts: host memory (staged) and a tensor on the handle's GPU (written / read in place)."""
import numpy as np
import pytest

from armada_amd import workloads as W
from armada_amd.binding import SchedError

NO_NODE = np.int64(2 ** 63 - 1)


def _layout(wl):
    cfg = wl.config
    cap = wl.node_total if wl.node_allocatable is None else wl.node_allocatable
    width = [max(1, int(int(cap[:, c].max(initial=0)) // int(r) + 1).bit_length()) for c, r in zip(cfg.indexed_col, cfg.indexed_resolution)]
    return width, max(1, int(wl.num_nodes - 1).bit_length())


def _fit_words(lib, wl, jobs, prio_index, rank_offset=0, use_cuda=False):
    s = W.load(lib, wl); W.prepare(s, wl)
    prio = s.priorities[prio_index]
    width, bits = _layout(wl)
    if use_cuda:
        import torch
        t = torch.empty(len(jobs), dtype=torch.int64, device="cuda")
        s.fit_select_batch_global(jobs, prio, width, bits + 4, t.data_ptr(), rank_offset=rank_offset)
        words = t.cpu().numpy()
    else:
        words = np.empty(len(jobs), dtype=np.int64)
        s.fit_select_batch_global(jobs, prio, width, bits + 4, words.ctypes.data, rank_offset=rank_offset)
    plain = s.fit_select_batch(jobs, prio)
    s.close()
    return words, plain, bits + 4


def _check_fit(lib, oracle_lib, use_cuda=False):
    for wl in (W.config2(n_nodes=700, n_jobs=3000), W.config3(n_nodes=500, n_jobs=3000, n_queues=6, seed=3, occupied=0.9)):
        jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
        for pi in (0, -1):
            for off in (0, 4321):
                got, plain, rb = _fit_words(lib, wl, jobs, pi, off, use_cuda)
                want, wplain, _ = _fit_words(oracle_lib, wl, jobs, pi, off)
                assert (got == want).all()
                node = np.where(got == NO_NODE, -1, (got & ((1 << rb) - 1)) - off).astype(np.int32)
                assert (node == plain).all() and (plain == wplain).all()


def test_fit_words_hostsim_equal_oracle(hostsim_lib, oracle_lib):
    _check_fit(hostsim_lib, oracle_lib)


def test_layout_too_narrow_is_refused(hostsim_lib):
    wl = W.config2(n_nodes=300, n_jobs=500)
    s = W.load(hostsim_lib, wl); W.prepare(s, wl)
    jobs = np.nonzero(wl.job_node < 0)[0].astype(np.int32)
    width, bits = _layout(wl)
    out = np.empty(len(jobs), dtype=np.int64)
    with pytest.raises(SchedError):   # rank field too small for the offset
        s.fit_select_batch_global(jobs, s.priorities[0], width, bits, out.ctypes.data, rank_offset=1 << 20)
    with pytest.raises(SchedError):   # a resource field too small
        s.fit_select_batch_global(jobs, s.priorities[0], [1] * len(width), bits, out.ctypes.data)
    with pytest.raises(SchedError):   # more than 62 bits
        s.fit_select_batch_global(jobs, s.priorities[0], [30] * len(width), 30, out.ctypes.data)
    s.close()


def _delta(lib, wl, factor=1, use_cuda=False):
    """one round, its delta buffer, and the resolution of `factor` identical ranks' sum (factor 2: every new placement collides with its twin where the node is full)"""
    s = W.load(lib, wl); W.prepare(s, wl)
    res = s.schedule_round()
    n = s.round_delta_words()
    if use_cuda:
        import torch
        t = torch.zeros(n, dtype=torch.int64, device="cuda")
        s.round_delta(t.data_ptr())
        buf = t.cpu().numpy()
        nr, r = wl.num_nodes * wl.job_req.shape[1], None
        red = t.clone(); red[:nr] *= factor
        out = s.round_delta_resolve(red.data_ptr())
    else:
        buf = np.zeros(n, dtype=np.int64)
        s.round_delta(buf.ctypes.data)
        red = buf.copy(); red[:wl.num_nodes * wl.job_req.shape[1]] *= factor
        out = s.round_delta_resolve(red.ctypes.data)
    s.close()
    return res, buf, out


def _check_delta(lib, oracle_lib, use_cuda=False):
    for seed in (1, 2, 3):
        wl = W.small_random(n_nodes=60, n_jobs=1500, n_queues=5, seed=seed, occupied=[0.3, 0.9, 1.0][seed % 3], gangs=4)
        for factor in (1, 2):
            res, buf, (sm, node, prio, rp) = _delta(lib, wl, factor, use_cuda)
            ores, obuf, (osm, onode, oprio, orp) = _delta(oracle_lib, wl, factor)
            assert (buf == obuf).all() and sm == osm and (node == onode).all() and (prio == oprio).all() and (rp == orp).all()
            new = [j for j in res.scheduled if wl.job_node[j] < 0]
            R = wl.job_req.shape[1]
            com = np.zeros((wl.num_nodes, R), dtype=np.int64)
            for j in new:
                com[res.scheduled[j]] += wl.job_req[j]
            assert (buf[:wl.num_nodes * R].reshape(-1, R) == com).all()
            words = buf[wl.num_nodes * R:]
            assert sorted(np.nonzero(words & 0x0fffffff)[0].tolist()) == sorted(new)
            assert sorted(np.nonzero(words >> 32)[0].tolist()) == sorted(int(j) for j in res.preempted)
            if factor == 1:   # one rank: nothing conflicts, the accepted state is the round's own result
                assert sm["conflict_nodes"] == 0 and sm["replay"] == 0 and sm["accepted"] == len(new) and sm["preempted"] == len(res.preempted)
                for j in new:
                    assert node[j] == res.scheduled[j] and prio[j] == res.scheduled_priority[j]
            else:             # accepted + replayed = every new job; a gang is replayed as a whole
                assert sm["accepted"] + sm["replay"] == len(new)
                if wl.job_gang is not None:
                    for g in set(int(x) for x in wl.job_gang[new] if x >= 0):
                        members = [j for j in new if wl.job_gang[j] == g]
                        assert len(set(int(rp[j]) for j in members)) == 1


def test_round_delta_hostsim_equal_oracle(hostsim_lib, oracle_lib):
    _check_delta(hostsim_lib, oracle_lib)


def test_round_delta_needs_a_round(hostsim_lib):
    wl = W.small_random(n_nodes=20, n_jobs=200, n_queues=3, seed=9)
    s = W.load(hostsim_lib, wl); W.prepare(s, wl)
    buf = np.zeros(s.round_delta_words(), dtype=np.int64)
    with pytest.raises(SchedError):
        s.round_delta(buf.ctypes.data)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("use_cuda", [False, True])
def test_fit_words_gpu(hip_lib, oracle_lib, use_cuda):
    _check_fit(hip_lib, oracle_lib, use_cuda)


@pytest.mark.gpu
@pytest.mark.parametrize("use_cuda", [False, True])
def test_round_delta_gpu(hip_lib, oracle_lib, use_cuda):
    _check_delta(hip_lib, oracle_lib, use_cuda)
