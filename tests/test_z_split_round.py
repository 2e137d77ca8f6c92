"""The round as a launch sequence (grid-wide kernels for the data-parallel phases + the persistent kernel for the two sequential passes,
asched_host.inc runRoundSplit — the default) against the same round inside ONE persistent launch (ASCHED_SINGLE_LAUNCH=1, round_run.h runRound)
and against the oracle.  The switch is read once per process, so the single-launch side runs in a subprocess."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, sys
sys.path.insert(0, %(root)r)
from armada_amd import workloads as W
from armada_amd.binding import Library
if %(gpu)r:
    import torch, armada_amd
    lib = armada_amd.load_library()
else:
    lib = Library(%(root)r + "/tests/hostsim/libhostsim.so", "asched_")
out = []
for seed in range(6):
    wl = W.small_random(n_nodes=30 + 11 * seed, n_jobs=700 + 90 * seed, n_queues=3 + seed %% 4, seed=900 + seed, occupied=[0.5, 0.9, 1.0][seed %% 3], gangs=seed %% 4,
                        burst=None if seed %% 2 else (300, 120), away=seed == 3)
    s = W.load(lib, wl); W.prepare(s, wl); r = s.schedule_round()
    t = s.round_timing()
    out.append({"scheduled": sorted(r.scheduled.items()), "preempted": sorted(r.preempted.items()), "prio": sorted(r.scheduled_priority.items()), "n1": r.num_evicted_phase1,
                "n3": r.num_evicted_phase3, "reason": r.termination_reason, "alloc": r.queue_allocated_by_pc.tolist(), "reasons": r.job_unschedulable_reason.tolist(),
                "tokens": r.queue_tokens_after.tolist(), "launches": t["launches"]})
print("RESULT " + json.dumps(out))
'''


def _run(gpu, single):
    env = dict(os.environ)
    env.pop("ASCHED_SINGLE_LAUNCH", None)
    if single:
        env["ASCHED_SINGLE_LAUNCH"] = "1"
    out = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT, "gpu": gpu}], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])


def _strip(rs):
    return [{k: v for k, v in r.items() if k != "launches"} for r in rs]


def test_split_equals_single_launch_cpu_build(hostsim_lib):
    a, b = _run(False, False), _run(False, True)
    assert _strip(a) == _strip(b)
    assert all(r["launches"] > 10 for r in a) and all(r["launches"] <= 1 for r in b)    # the sequence really ran / really did not
    assert any(r["n3"] > 0 for r in a) and any(r["preempted"] for r in a)                 # phase 3 + second pass exercised


@pytest.mark.gpu
def test_split_equals_single_launch_gpu(hip_lib):
    a, b = _run(True, False), _run(True, True)
    assert _strip(a) == _strip(b)
    assert all(r["launches"] > 10 for r in a)
