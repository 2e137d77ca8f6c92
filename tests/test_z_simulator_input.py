"""BASELINE configs[0] on the reference simulator's OWN input: testdata/clusters/cpu_1_1_100.yaml + the basicWorkload.yaml job template with
number: 1000 + testdata/configs/basicSchedulingConfig.yaml, read from the JSON fixture tests/golden/make_simulator_fixture.py makes of them.
With maximumResourceFractionToSchedule 0.025 on cpu and memory the round stops after the first job that pushes the scheduled resources over
2.5% of 3200 cpus: "already scheduled EXCEEDS the cap" is checked before each new gang (constraints.go:113-119, gang_scheduler.go:102-106), so
81 one-cpu jobs land per 10 s cycle and the 1000 jobs need 13 cycles.  No reference test pins node ids (they are random in the
Go simulator): parity on job->node is oracle vs library, cycle by cycle."""
import os

import numpy as np
import pytest

from armada_amd import simulator_input as S

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simulator_basic_input.json")


def test_fixture_matches_the_reference_files():
    """when the reference checkout is present (build container) the fixture is exactly what its three files say"""
    ref = "/root/reference/internal/scheduler/simulator/testdata"
    if not os.path.isdir(ref):
        pytest.skip("no reference checkout here (GPU box)")
    a = S.from_fixture(FIXTURE)
    import copy, yaml
    docs = [yaml.safe_load(open(os.path.join(ref, p))) for p in ("clusters/cpu_1_1_100.yaml", "workloads/basicWorkload.yaml", "configs/basicSchedulingConfig.yaml")]
    docs[1]["queues"][0]["jobTemplates"][0]["number"] = 1000
    b = S.from_documents(*docs)
    assert np.array_equal(a.workload.node_total, b.workload.node_total) and np.array_equal(a.workload.job_req, b.workload.job_req)
    assert a.workload.config == b.workload.config and a.node_ids == b.node_ids


def test_input_is_parsed_like_the_factory():
    sim = S.from_fixture(FIXTURE)
    wl = sim.workload
    assert sim.resource_names == ["memory", "cpu", "ephemeral-storage", "nvidia.com/gpu"]
    assert wl.num_nodes == 100 and wl.num_jobs == 1000 and wl.num_queues == 1
    assert wl.node_total[0].tolist() == [1024 * 2 ** 30, 32000, 0, 0]            # memory in bytes, cpu in milli-cpus
    assert wl.job_req[0].tolist() == [10 * 2 ** 30, 1000, 0, 0]
    c = wl.config
    assert c.indexed_col == [1, 0, 3] and c.indexed_resolution == [1000, 2 ** 20, 1]   # cpu @ 1, memory @ 1Mi, gpu @ 1
    assert c.pc_priority == [30000, 30000] and c.pc_preemptible == [0, 1]              # armada-default (non-preemptible), armada-preemptible
    assert c.max_fraction_to_schedule == [0.025, 0.025, float("inf"), float("inf")]
    assert sim.node_ids[:3] == ["cpu-01-0-0", "cpu-01-0-1", "cpu-01-0-2"] and wl.node_id_rank[10] < wl.node_id_rank[2]   # "…-10" sorts before "…-2"
    assert S.parse_quantity("1024Gi") == 1024 * 2 ** 30 and S.parse_quantity("1m") == 1e-3 and S.parse_duration_s("5m") == 300


def _cycles(lib):
    return S.run_cycles(lib, S.from_fixture(FIXTURE))


def _check(cyc):
    placed = [len(c["scheduled"]) for c in cyc]
    assert placed[:13] == [81] * 12 + [28] and sum(placed) == 1000
    assert all(c["termination_reason"] == 1 for c in cyc[:12])        # "maximum resources scheduled"
    assert not any(c["preempted"] for c in cyc)                        # armada-default is non-preemptible
    # best fit: a cycle's jobs fill the node that has the least free cpu that still fits
    first = cyc[0]["scheduled"]
    assert len(set(first.values())) == 3                               # 32 + 32 + 17 one-cpu jobs


def test_cycles_on_the_oracle(oracle_lib):
    _check(_cycles(oracle_lib))


def test_cycles_cpu_build_equals_oracle(hostsim_lib, oracle_lib):
    a, b = _cycles(oracle_lib), _cycles(hostsim_lib)
    assert [(c["scheduled"], c["preempted"], c["termination_reason"]) for c in a] == [(c["scheduled"], c["preempted"], c["termination_reason"]) for c in b]


@pytest.mark.gpu
def test_cycles_gpu_equals_oracle(hip_lib, oracle_lib):
    a, b = _cycles(oracle_lib), _cycles(hip_lib)
    _check(b)
    assert [(c["scheduled"], c["preempted"], c["termination_reason"]) for c in a] == [(c["scheduled"], c["preempted"], c["termination_reason"]) for c in b]
