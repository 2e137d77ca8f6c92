"""Guard against a silent fall-back: on the bench-shaped workload (BASELINE configs[2], reduced size, bench.py's burst scaling) the round
must run on the fast path — the host-side exactness conditions (asched_host.inc buildFast / rebuildMasks) decide that, and a change
there would not alter any result, only make the GPU round an order of magnitude slower.  Checked on the CPU build of the device code,
which counts iterations exactly like the kernel (asched_round_stats)."""
from armada_amd import workloads as W


def test_bench_shaped_round_stays_on_the_fast_path(hostsim_lib, oracle_lib):
    jobs = 50_000
    wl = W.config3(n_nodes=5_000, n_jobs=jobs, n_queues=64)
    wl.global_burst, wl.queue_burst = int(200_000 * jobs / 1e6), int(20_000 * jobs / 1e6)   # bench.py main()
    s = W.load(hostsim_lib, wl)
    W.prepare(s, wl)
    r = s.schedule_round()
    st = s.round_stats()
    assert len(r.scheduled) == wl.global_burst
    assert st["fast_iterations"] >= wl.global_burst and st["generic_iterations"] <= 8, st
    assert st["l0_overflows"] == 0, st
    o = W.load(oracle_lib, wl)
    W.prepare(o, wl)
    assert o.schedule_round().scheduled == r.scheduled
