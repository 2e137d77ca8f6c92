"""Order keys wider than one 64-bit word (nodedb/encoding.go:22-89).

The reference's index key is one big-endian 8-byte word per indexed resource plus the node index — as wide as the configuration makes it.  The packed key of the default
kernels is one 64-bit integer; rounds 1-5 refused a pool whose fields + node-index rank need more (ASCHED_ERR_UNSUPPORTED: "packed order key needs more than 64 bits").
Round 6: such a pool gets a TWO-WORD key (one 128-bit integer, same field order; asched_host.inc layoutKeys, round_ctl.h packKey2) and is served by the generic path of a
round kernel of its own (armada_amd/csrc/armada_sched_wk.hip: k_control_wk, k_bulk_wk, k_fit_batch_wk); node selection is two plane passes (high word, then low word
among its carriers), the literal iterators carry a two-word bound.

Here (CPU build of the device code vs the oracle; the same through the HIP library with -m gpu):
  * the layouts the reviews named — K = 5 indexed resources at fine resolutions, a million nodes — are accepted and SCHEDULE, rounds identical to the oracle;
  * ASCHED_KEY_WORDS=2 forces the two-word layout (fields moved across the word boundary) on workloads whose key would fit one word: the seeded differential rounds
    (gangs, away node types, several node types with unaligned allocatable, requests off the index grid = literal iteration, crowded pools = both preemption kinds),
    NodeDb-level calls, the submit check, the batched first fit, node iteration order;
  * what is still refused says so: more than 128 bits, the multi-GPU exchange word.
The whole CPU suite was also run once under ASCHED_KEY_WORDS=2 (profiles/r06z2_two_word_keys.txt): everything but the tests that assert "ran on the fast path" passes.
"""
import numpy as np
import pytest

import scenario
from armada_amd import workloads as W
from armada_amd.binding import Config, SchedError, Scheduler

Gi, Mi = W.Gi, W.Mi
ERR_UNSUPPORTED = -2


def _k5_cfg():
    # nvidia.com/gpu @1, cpu @1m, memory @1Mi, ephemeral-storage @1Mi, local nvme @1Gi
    return Config(num_resources=5, indexed_col=[W.GPU, W.CPU, W.MEM, W.EPH, 4], indexed_resolution=[1, 1, Mi, Mi, Gi],
                  pc_priority=[0, 1, 3], pc_preemptible=[1, 1, 0], drf_multiplier=[1.0] * 5)


def _round(lib, wl, fp=None):
    s = W.load(lib, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp)
    out = (s.schedule_round(), s.round_stats())
    s.close()
    return out


def _same(libs, wl, fp=None):
    out = [_round(lib, wl, fp) for lib in libs]
    scenario.assert_same_round(out[0][0], out[1][0])
    return out[1]


def _accepts(lib, n):
    total = np.tile(np.array([1024 * Gi, 64_000, 4096 * Gi, 8, 32768 * Gi], dtype=np.int64), (n, 1))
    s = Scheduler(lib, _k5_cfg())
    s.nodes_upsert(total, total)
    assert s.num_nodes == n
    s.close()


def test_k5_fine_resolution_million_nodes_is_accepted(hostsim_lib):
    """4 + 16 + 21 + 23 + 16 bits of fields + 20 bits of node-index rank = 100 bits"""
    _accepts(hostsim_lib, 1_000_000)


@pytest.mark.parametrize("k5", [True, False])
@pytest.mark.parametrize("occupied", [0.5, 0.97])
def test_fine_indexed_rounds_match_the_oracle(hostsim_lib, oracle_lib, occupied, k5):
    wl = W.fine_indexed(n_nodes=600, n_jobs=6000, n_queues=8, occupied=occupied, k5=k5)
    r, st = _same((oracle_lib, hostsim_lib), wl)
    assert len(r.scheduled) > 500 and st["fast_iterations"] == 0
    if occupied > 0.9:
        assert len(r.preempted) > 300


def test_more_than_128_bits_is_refused_with_a_reason(hostsim_lib):
    cfg = Config(num_resources=6, indexed_col=[0, 1, 2, 3, 4, 5], indexed_resolution=[1] * 6, pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0] * 6)
    total = np.tile(np.array([1 << 40] * 6, dtype=np.int64), (4, 1))
    s = Scheduler(hostsim_lib, cfg)
    with pytest.raises(SchedError) as e:
        s.nodes_upsert(total, total)
    assert e.value.code == ERR_UNSUPPORTED and "128 bits" in str(e.value)
    s.close()


def _small(seed, **kw):
    return W.small_random(n_nodes=10 + seed % 9 * 7, n_jobs=300 + seed % 7 * 60, n_queues=2 + seed % 5, seed=7000 + seed, occupied=[0.4, 0.8, 0.95, 1.0][seed % 4],
                          gangs=seed % 4, **kw)


@pytest.mark.parametrize("seed", range(24))
def test_forced_two_word_rounds_match_the_oracle(hostsim_lib, oracle_lib, monkeypatch, seed):
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl = _small(seed, away=seed % 5 == 0, ragged=seed % 3 == 0, offgrid=[0, 0, 1, 3][seed % 4])
    r, st = _same((oracle_lib, hostsim_lib), wl, fp=None if seed % 2 else 5.0)
    assert st["fast_iterations"] == 0


def test_the_high_word_decides(hostsim_lib, oracle_lib, monkeypatch):
    """negative control of the forced layout: with the high word ignored (HOSTSIM_IGNORE_HIGH_WORD: the CPU build compares low words only) the rounds are NOT the oracle's —
    the forced layout really puts ordering fields into the high word"""
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    monkeypatch.setenv("HOSTSIM_IGNORE_HIGH_WORD", "1")
    diverged = 0
    for seed in range(6):
        wl = _small(seed)
        a, b = _round(oracle_lib, wl)[0], _round(hostsim_lib, wl)[0]
        try:
            scenario.assert_same_round(a, b)
        except AssertionError:
            diverged += 1
    assert diverged >= 4, diverged


@pytest.mark.parametrize("seed", range(6))
def test_forced_two_word_fit_batch_and_node_calls(hostsim_lib, oracle_lib, monkeypatch, seed):
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl = _small(seed + 40, ragged=False, offgrid=0)
    out = []
    for lib in (oracle_lib, hostsim_lib):
        s = W.load(lib, wl)
        queued = np.concatenate(wl.queued)[:200]
        fits = {p: s.fit_select_batch(queued, p).tolist() for p in s.priorities}
        sel = [s.select_node(int(j)) for j in queued[:40]]
        out.append((fits, sel))
        s.close()
    assert out[0] == out[1]


def test_multi_gpu_exchange_word_is_refused(hostsim_lib, monkeypatch):
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl = _small(2)
    s = W.load(hostsim_lib, wl)
    buf = np.zeros(8, dtype=np.int64)
    with pytest.raises(SchedError) as e:
        s.fit_select_batch_global(np.concatenate(wl.queued)[:8], -2, [8, 8, 8], 8, buf.ctypes.data)
    assert e.value.code == ERR_UNSUPPORTED
    s.close()


# ---------------------------------------------------------------- a coarser divisor before a second word (asched_host.inc layoutKeys)
@pytest.mark.parametrize("occupied", [0.5, 0.97])
def test_coarse_requests_bring_fine_resolutions_back_to_one_word(hostsim_lib, oracle_lib, occupied):
    """cpu @1m, memory @1Mi, ephemeral @1Mi need 79 bits at face value; the jobs ask in quarter cores / 100Mi / GiB, so every column value is a multiple of 250m / 32Mi / 2Gi and the
    key may divide by those (same order, exactly): 51 bits, one word, the FAST path — where the same pool with odd requests runs the generic path on a two-word key"""
    wl = W.fine_indexed(n_nodes=600, n_jobs=6000, n_queues=8, occupied=occupied, k5=False, coarse=True)
    r, st = _same((oracle_lib, hostsim_lib), wl)
    assert st["fast_iterations"] > 900 and st["generic_iterations"] <= 8, st
    odd = W.fine_indexed(n_nodes=600, n_jobs=6000, n_queues=8, occupied=occupied, k5=False, coarse=False)
    r2, st2 = _same((oracle_lib, hostsim_lib), odd)
    assert st2["fast_iterations"] == 0


@pytest.mark.parametrize("seed", range(16))
def test_forced_coarse_divisor_rounds_match_the_oracle(hostsim_lib, oracle_lib, monkeypatch, seed):
    """ASCHED_KEY_GCD=1: the coarser divisor wherever it applies (here memory in 1-4 GiB units against a 128Mi resolution): the whole fast path on scaled key fields"""
    monkeypatch.setenv("ASCHED_KEY_GCD", "1")
    wl = _small(seed, away=seed % 5 == 0)
    r, st = _same((oracle_lib, hostsim_lib), wl, fp=None if seed % 2 else 5.0)
    assert st["fast_iterations"] > 0


@pytest.mark.parametrize("shift", [(25, 19), (28, 20)])
@pytest.mark.parametrize("seed", range(4))
def test_scaled_pool_keeps_the_fast_path(hostsim_lib, oracle_lib, monkeypatch, capfd, seed, shift):
    """the small mixed pool with every cpu / memory quantity (nodes and requests alike) multiplied by 2^a / 2^b: at face value the key outgrows one word with its guard bits
    (a, b = 25, 19) or one word altogether (28, 20); every column value is a multiple of the scaled unit, the divisor follows it, the fields are as wide as the unscaled pool's —
    one word, guard bits, the fast path, and the round is the oracle's"""
    monkeypatch.setenv("ASCHED_PRINT_LAYOUT", "1")
    wl = _small(seed)
    a, b = shift
    for arr in (wl.node_total, wl.job_req):
        arr[:, W.CPU] <<= a; arr[:, W.MEM] <<= b
    if wl.node_allocatable is not None:
        wl.node_allocatable[:, W.CPU] <<= a; wl.node_allocatable[:, W.MEM] <<= b
    r, st = _same((oracle_lib, hostsim_lib), wl, fp=None if seed % 2 else 5.0)
    assert st["fast_iterations"] > 0, st
    assert "coarser key divisors" in capfd.readouterr().err


def test_coarse_divisor_refusals(hostsim_lib, monkeypatch):
    """what reasons in the index resolution itself is refused on such a handle: a node value off the divisor, the cross-shard key word"""
    monkeypatch.setenv("ASCHED_KEY_GCD", "1")
    wl = _small(1)
    s = W.load(hostsim_lib, wl)
    p, r = len(s.priorities), wl.node_total.shape[1]
    planes = np.tile(wl.node_total[0], (p, 1)).astype(np.int64)
    planes[:, W.MEM] += 3 * Mi   # off every plausible divisor
    with pytest.raises(SchedError) as e:
        s.node_upsert(0, planes)
    assert e.value.code == ERR_UNSUPPORTED and "divisor" in str(e.value)
    buf = np.zeros(8, dtype=np.int64)
    with pytest.raises(SchedError) as e:
        s.fit_select_batch_global(np.concatenate(wl.queued)[:8], -2, [8, 8, 8], 8, buf.ctypes.data)
    assert e.value.code == ERR_UNSUPPORTED
    s.close()


# ---------------------------------------------------------------- the HIP library (k_control_wk)
@pytest.mark.gpu
def test_coarse_requests_bring_fine_resolutions_back_to_one_word_gpu(hip_lib, oracle_lib):
    wl = W.fine_indexed(n_nodes=20_000, n_jobs=200_000, n_queues=64, occupied=0.5, k5=False, coarse=True)
    r, st = _same((oracle_lib, hip_lib), wl)
    assert len(r.scheduled) == wl.global_burst and st["fast_iterations"] >= wl.global_burst and st["generic_iterations"] <= 8 and st["stream_jobs"] > wl.global_burst // 2, st


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_forced_coarse_divisor_rounds_gpu(hip_lib, oracle_lib, monkeypatch, seed):
    monkeypatch.setenv("ASCHED_KEY_GCD", "1")
    wl = _small(seed, away=seed % 5 == 0)
    _same((oracle_lib, hip_lib), wl, fp=None if seed % 2 else 5.0)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [(25, 19), (28, 20)])
@pytest.mark.parametrize("seed", range(4))
def test_scaled_pool_keeps_the_fast_path_gpu(hip_lib, oracle_lib, seed, shift):
    wl = _small(seed)
    a, b = shift
    for arr in (wl.node_total, wl.job_req):
        arr[:, W.CPU] <<= a; arr[:, W.MEM] <<= b
    r, st = _same((oracle_lib, hip_lib), wl, fp=None if seed % 2 else 5.0)
    assert st["fast_iterations"] > 0, st


@pytest.mark.gpu
def test_forced_coarse_divisor_crowded_round_gpu(hip_lib, oracle_lib, monkeypatch):
    monkeypatch.setenv("ASCHED_KEY_GCD", "1")
    wl = W.config3(n_nodes=5_000, n_jobs=50_000, n_queues=32, occupied=0.95)
    r, st = _same((oracle_lib, hip_lib), wl)
    assert len(r.preempted) > 1000 and st["fast_iterations"] > 1000, (len(r.preempted), st)


@pytest.mark.gpu
def test_k5_fine_resolution_million_nodes_is_accepted_gpu(hip_lib):
    _accepts(hip_lib, 1_000_000)


@pytest.mark.gpu
@pytest.mark.parametrize("occupied", [0.5, 0.97])
def test_fine_indexed_round_gpu(hip_lib, oracle_lib, occupied):
    """the differential round the review asked for: 20 000 nodes on a K = 5 fine-resolution layout (95 bits), identical to the oracle"""
    wl = W.fine_indexed(n_nodes=20_000, n_jobs=100_000 if occupied < 0.9 else 40_000, n_queues=32, occupied=occupied)
    r, st = _same((oracle_lib, hip_lib), wl)
    assert len(r.scheduled) > 5000 and st["fast_iterations"] == 0
    if occupied > 0.9:
        assert len(r.preempted) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_forced_two_word_rounds_gpu(hip_lib, oracle_lib, monkeypatch, seed):
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl = _small(seed, away=seed % 5 == 0, ragged=seed % 3 == 0, offgrid=[0, 0, 1, 3][seed % 4])
    _same((oracle_lib, hip_lib), wl, fp=None if seed % 2 else 5.0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_forced_two_word_fit_batch_and_node_calls_gpu(hip_lib, oracle_lib, monkeypatch, seed):
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl = _small(seed + 40, ragged=False, offgrid=0)
    out = []
    for lib in (oracle_lib, hip_lib):
        s = W.load(lib, wl)
        queued = np.concatenate(wl.queued)[:200]
        fits = {p: s.fit_select_batch(queued, p).tolist() for p in s.priorities}
        sel = [s.select_node(int(j)) for j in queued[:40]]
        out.append((fits, sel))
        s.close()
    assert out[0] == out[1]


@pytest.mark.gpu
def test_forced_two_word_medium_round_gpu(hip_lib, oracle_lib, monkeypatch):
    """a crowded 2 000-node pool under the forced layout: helper workgroups take part in both plane passes"""
    monkeypatch.setenv("ASCHED_KEY_WORDS", "2")
    wl = W.default_indexed(n_nodes=2_000, n_jobs=20_000, n_queues=16, occupied=0.97)
    r, st = _same((oracle_lib, hip_lib), wl)
    assert len(r.preempted) > 300 and st["fast_iterations"] == 0
