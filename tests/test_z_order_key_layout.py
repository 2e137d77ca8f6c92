"""Order-key layout (nodedb/encoding.go:22-54 as one packed integer; armada_amd/csrc/asched_host.inc layoutKeys).

Round 2 widened every key field by (P + 2) for the oversubscription urgency preemption can cause and therefore refused the reference's own default
`indexedResources` (config/scheduler/config.yaml:121-129) on ordinary nodes.  The narrow layout gives every negative quotient one shared field value
(with non-negative requests such a node fits nothing at that level): these tests pin that (a) the default configuration is accepted at 100 000 nodes,
(b) rounds on it equal the oracle and run on the fast path, also when allocatable is off the index grid (1 TiB is not a multiple of 100Mi) and when the
cluster is oversubscribed, and (c) the regression this work uncovered: a job whose request is off the index grid must never be bound by key arithmetic —
an evicted one returning to its node used to be (wrong keys from then on) whenever the level-0 fast structure was active next to literal iteration rows.
"""
import numpy as np
import pytest

import scenario
from armada_amd import workloads as W
from armada_amd.binding import Config, Scheduler

Gi, Mi = W.Gi, W.Mi


def _default_cfg():
    return Config(num_resources=4, indexed_col=[W.GPU, W.CPU, W.MEM, W.EPH], indexed_resolution=[1, 100, 100 * Mi, Gi],
                  pc_priority=[0, 1, 3], pc_preemptible=[1, 1, 0], drf_multiplier=[1.0, 1.0, 1.0, 1.0])


@pytest.mark.parametrize("n,node", [(100_000, [1024 * Gi, 64_000, 4096 * Gi, 8]), (10_000, [2048 * Gi, 128_000, 8192 * Gi, 0])])
def test_default_indexed_resources_are_accepted(hostsim_lib, oracle_lib, n, node):
    """the two configurations the round-2 review ran against the host code: both were ASCHED_ERR_UNSUPPORTED ("packed order key needs more than 64 bits")"""
    total = np.tile(np.array(node, dtype=np.int64), (n, 1))
    for lib in (oracle_lib, hostsim_lib):
        s = Scheduler(lib, _default_cfg())
        s.nodes_upsert(total, total)
        assert s.num_nodes == n
        s.close()


def _same(libs, wl, fp=None):
    out = []
    for lib in libs:
        s = W.load(lib, wl); W.prepare(s, wl, fairshare_preemption_tokens=fp)
        out.append((s.schedule_round(), s.round_stats()))
        s.close()
    scenario.assert_same_round(out[0][0], out[1][0])
    return out[1]


@pytest.mark.parametrize("occupied", [0.5, 0.97])
def test_default_indexed_round_matches_oracle_on_the_fast_path(hostsim_lib, oracle_lib, occupied):
    wl = W.default_indexed(n_nodes=1500, n_jobs=15_000, n_queues=16, occupied=occupied)
    r, st = _same((oracle_lib, hostsim_lib), wl)
    assert len(r.scheduled) == wl.global_burst
    assert st["generic_iterations"] <= 8 and st["fast_iterations"] >= wl.global_burst and st["l0_overflows"] == 0, st
    if occupied > 0.9:
        assert len(r.preempted) > 500 and st["preempt_fast_iterations"] > 500, (len(r.preempted), st)


def test_default_indexed_round_with_gib_requests_takes_the_literal_path(hostsim_lib, oracle_lib):
    """memory asked for in GiB is off a 100Mi grid: those jobs are placed by the literal iterator restatement (exact, slow) — the round still equals the oracle"""
    wl = W.default_indexed(n_nodes=300, n_jobs=3000, n_queues=6, occupied=0.8, aligned=False)
    r, st = _same((oracle_lib, hostsim_lib), wl)
    assert len(r.scheduled) > 100


@pytest.mark.parametrize("seed", range(20, 36))
@pytest.mark.parametrize("offgrid", [1, 2, 3])
def test_offgrid_jobs_next_to_the_fast_structure(hostsim_lib, oracle_lib, seed, offgrid):
    """one node type: the fast structure is on; a third of the jobs (running ones too) ask for amounts off the index grid (bit 0), allocatable is off it (bit 1).
    Before round 3: 16 of 30 such seeds diverged (evicted off-grid jobs were rebound by subtracting floor(request / resolution) from the node's key fields)."""
    wl = W.small_random(n_nodes=12 + seed % 10 * 8, n_jobs=250 + seed % 10 * 40, n_queues=2 + seed % 4, seed=200 + seed, occupied=[0.4, 0.8, 0.95][seed % 3],
                        gangs=seed % 3, offgrid=offgrid)
    _same((oracle_lib, hostsim_lib), wl, fp=None if seed % 2 else 5.0)


@pytest.mark.parametrize("layout", ["narrow", "wide"])
@pytest.mark.parametrize("seed", range(8))
def test_oversubscribed_rounds_under_both_layouts(hostsim_lib, oracle_lib, monkeypatch, seed, layout):
    """crowded rounds (urgency preemption overdraws the lower levels: negative key fields) with the layout forced either way (ASCHED_KEY_LAYOUT)"""
    monkeypatch.setenv("ASCHED_KEY_LAYOUT", layout)
    wl = W.small_random(n_nodes=10 + seed * 7, n_jobs=400 + seed * 60, n_queues=2 + seed % 5, seed=900 + seed, occupied=[0.9, 1.0][seed % 2], gangs=seed % 4,
                        offgrid=[0, 2][seed % 2])
    _same((oracle_lib, hostsim_lib), wl)


@pytest.mark.gpu
def test_default_indexed_round_gpu(hip_lib, oracle_lib):
    """the reference's default indexedResources on 20 000 x {64 cpu, 1 TiB, 4 TiB ephemeral, 8 gpu} nodes, 200 000 queued jobs: identical to the oracle, on the fast path"""
    wl = W.default_indexed(n_nodes=20_000, n_jobs=200_000, n_queues=64, occupied=0.5)
    r, st = _same((oracle_lib, hip_lib), wl)
    assert len(r.scheduled) == wl.global_burst
    assert st["generic_iterations"] <= 8 and st["fast_iterations"] >= wl.global_burst and st["l0_overflows"] == 0 and st["stream_jobs"] > wl.global_burst // 2, st


@pytest.mark.gpu
def test_default_indexed_crowded_round_gpu(hip_lib, oracle_lib):
    wl = W.default_indexed(n_nodes=5_000, n_jobs=50_000, n_queues=32, occupied=0.97)
    r, st = _same((oracle_lib, hip_lib), wl)
    assert len(r.preempted) > 1000 and st["preempt_fast_iterations"] > 1000 and st["generic_iterations"] <= 16, (len(r.preempted), st)


@pytest.mark.gpu
@pytest.mark.parametrize("n,node", [(100_000, [1024 * Gi, 64_000, 4096 * Gi, 8]), (10_000, [2048 * Gi, 128_000, 8192 * Gi, 0])])
def test_default_indexed_resources_are_accepted_gpu(hip_lib, n, node):
    total = np.tile(np.array(node, dtype=np.int64), (n, 1))
    s = Scheduler(hip_lib, _default_cfg())
    s.nodes_upsert(total, total)
    assert s.num_nodes == n
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(20, 30))
def test_offgrid_jobs_next_to_the_fast_structure_gpu(hip_lib, oracle_lib, seed):
    wl = W.small_random(n_nodes=12 + seed % 10 * 8, n_jobs=250 + seed % 10 * 40, n_queues=2 + seed % 4, seed=200 + seed, occupied=[0.4, 0.8, 0.95][seed % 3],
                        gangs=seed % 3, offgrid=1 + seed % 3)
    _same((oracle_lib, hip_lib), wl)
