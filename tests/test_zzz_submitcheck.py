"""Submit check (internal/scheduler/submitcheck.go): ClearAllocated + batched ScheduleManyWithTxn-and-abort.

Layers, all compared on the same inputs:
  expectations of submitcheck_test.go (tests/golden/submitcheck_cases.json, 29 cases)
    == literal one-transaction-per-attempt restatement of SubmitChecker.Check on the CPU oracle      (pins the restatement)
    == batched product flow (armada_amd.submitcheck -> asched_submit_check) on the oracle / on the CPU build of the device code
    == (-m gpu) batched product flow on the HIP library.
Seeded differential cases (several pools, away pools, submission groups, gangs, selectors, tolerations, per-queue limits) compare the
batched flow on the implementation under test with the literal flow on the oracle.
"""
import copy

import numpy as np
import pytest

import submitcheck_harness as H
from golden_io import ids, load

CASES = load("submitcheck")


def _expect(case, got):
    exp = case["expectedResult"]
    assert set(got) == set(exp), f"jobs checked {sorted(got)} != expected {sorted(exp)}"
    for jid, e in exp.items():
        assert got[jid].is_schedulable == bool(e.get("isSchedulable", False)), f"{jid}: {got[jid]} expected {e}"
        assert sorted(got[jid].pools) == sorted(e.get("pools") or []), f"{jid}: pools {got[jid].pools} expected {e.get('pools')}"


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_literal_on_oracle(oracle_lib, case):
    _expect(case, H.literal_check(oracle_lib, case))


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_batched_on_oracle(oracle_lib, case):
    _expect(case, H.batched_check(oracle_lib, case))


@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_batched_on_hostsim(hostsim_lib, case):
    _expect(case, H.batched_check(hostsim_lib, case))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ids(CASES))
def test_goldens_batched_on_gpu(hip_lib, case):
    _expect(case, H.batched_check(hip_lib, case))


# ------------------------------------------------------------------------------------------------ seeded differential cases
GI = 2**30


def random_case(seed: int, n_jobs=60, scale=1):
    rng = np.random.default_rng(1000 + seed)
    base = copy.deepcopy(CASES[0])
    cfg = base["SchedulingConfig"]
    if seed % 3 == 1:
        cfg["disable_gang_away"] = True
    pools = [{"name": "a"}, {"name": "b", "submission_group": "g"}, {"name": "c", "submission_group": "g"},
             {"name": "gpu"}, {"name": "a-away", "away_pools": ["gpu"]}, {"name": "nocpu", "disallowed_resources": ["nvidia.com/gpu"]},
             {"name": "empty"}]
    nodes, idx = [], 1
    for p, kind, cnt in (("a", "cpu", 3), ("b", "cpu", 2), ("c", "big", 1), ("gpu", "gpu", 2), ("nocpu", "gpu", 1)):
        for _ in range(cnt * scale):
            if kind == "cpu":
                n = {"total": {"cpu": int(rng.integers(2, 9)) * 1000, "memory": int(rng.integers(8, 65)) * GI}, "taints": [], "labels": {}}
            elif kind == "big":
                n = {"total": {"cpu": 32000, "memory": 256 * GI}, "taints": [["largeJobsOnly", "true", "NoSchedule"]], "labels": {"largeJobsOnly": "true"}}
            else:
                n = {"total": {"cpu": 30000, "memory": 512 * GI, "nvidia.com/gpu": 8000}, "taints": [["gpu", "true", "NoSchedule"]], "labels": {"gpu": "true"}}
            if rng.random() < 0.2:
                n["labels"]["zone"] = "z1"
            n.update(index=idx, used={}, unschedulable=bool(rng.random() < 0.15), pool=p)
            idx += 1
            nodes.append(n)
    pcs = ["priority-0", "priority-1", "armada-preemptible-away", "priority-2-non-preemptible"]
    jobs = []

    def job(queue, created):
        kind = rng.integers(0, 6)
        req = {"cpu": int(rng.integers(1, 7)) * 1000, "memory": int(rng.integers(1, 40)) * GI}
        tol, sel = [], {}
        if kind == 1:
            req = {"cpu": int(rng.integers(8, 33)) * 1000, "memory": int(rng.integers(16, 257)) * GI}
            tol = [{"key": "largeJobsOnly", "op": "Equal", "value": "true", "effect": ""}]
        elif kind == 2:
            req["nvidia.com/gpu"] = int(rng.integers(1, 10)) * 1000
            tol = [{"key": "gpu", "op": "Equal", "value": "true", "effect": ""}]
        elif kind == 3:
            sel = {"zone": "z1"} if rng.random() < 0.7 else {"foo": "bar"}
        return {"created": created, "queue": queue, "pc": str(rng.choice(pcs)), "priority": 1000, "gang": None, "req": req,
                "tolerations": tol, "selector": sel, "affinity": None}
    created = 1
    while len(jobs) < n_jobs:
        q = f"q{int(rng.integers(0, 3))}"
        if rng.random() < 0.25:
            card = int(rng.integers(2, 7))
            proto = job(q, created)
            gid = f"gang-{created}"
            members = []
            for _ in range(card):
                mj = copy.deepcopy(proto)
                mj["created"] = created
                created += 1
                mj["gang"] = {"id": gid, "cardinality": card, "uniformity": ""}
                members.append(mj)
            if rng.random() < 0.3:   # interleave a gang with other jobs, like "One job fits, one gang doesn't, out of order"
                jobs.append(members.pop(0))
                jobs.append(job(q, created)); created += 1
            jobs.extend(members)
        else:
            jobs.append(job(q, created)); created += 1
    for i, j in enumerate(jobs):
        j["id"] = f"job-{i:05d}"
    queues = [{"Name": "q0"}, {"Name": "q1", "ResourceLimitsByPriorityClassName": {"priority-1": {"MaximumResourceFraction": {"cpu": 0.2}}}}]
    case = {"name": f"seed{seed}", "SchedulingConfig": cfg, "Pools": pools, "Queues": queues, "executors": [nodes], "jobs": jobs}
    if seed % 4 == 3:
        case["submitCheckConfig"] = {"MaxDurationPerQueue": 25, "MaxDuration": 70}
        case["clockStep"] = 1
    return case


@pytest.mark.parametrize("seed", range(10))
def test_random_batched_on_hostsim_equals_literal_on_oracle(oracle_lib, hostsim_lib, seed):
    case = random_case(seed)
    lit = H.literal_check(oracle_lib, case)
    H.same_results(lit, H.batched_check(hostsim_lib, case))
    H.same_results(lit, H.batched_check(oracle_lib, case))
    assert any(r.is_schedulable for r in lit.values()) and any(not r.is_schedulable for r in lit.values())


def test_cache_hit_carries_the_limits_of_the_job_that_filled_it(oracle_lib, hostsim_lib):
    """the reference's cache is keyed by scheduling key only (submitcheck.go:279-288): a job of queue q0 that hits the entry filled by a job
    of queue q1 inherits q1's verdict although q1's per-queue limit does not apply to q0 — the batched flow reproduces that, in both orders
    (seed 198 of the seeded cases found it)"""
    base = copy.deepcopy(CASES[0])
    node = {"index": 1, "total": {"cpu": 8000, "memory": 64 * GI}, "taints": [], "labels": {}, "used": {}, "unschedulable": False, "pool": "cpu"}
    mk = lambda i, q: {"id": f"job-{i}", "created": i, "queue": q, "pc": "priority-1", "priority": 1000, "gang": None, "tolerations": [], "selector": {},  # noqa: E731
                       "affinity": None, "req": {"cpu": 4000, "memory": 4 * GI}}
    queues = [{"Name": "q0"}, {"Name": "q1", "ResourceLimitsByPriorityClassName": {"priority-1": {"MaximumResourceFraction": {"cpu": 0.2}}}}]
    for order, expect in ((("q1", "q0"), [False, False]), (("q0", "q1"), [True, True])):
        case = {"name": "x", "SchedulingConfig": base["SchedulingConfig"], "Pools": [{"name": "cpu"}], "Queues": queues, "executors": [[node]],
                "jobs": [mk(1, order[0]), mk(2, order[1])]}
        for got in (H.literal_check(oracle_lib, case), H.batched_check(oracle_lib, case), H.batched_check(hostsim_lib, case)):
            assert [got["job-1"].is_schedulable, got["job-2"].is_schedulable] == expect, (order, got)


def test_seed_198(oracle_lib, hostsim_lib):
    case = random_case(198)
    H.same_results(H.literal_check(oracle_lib, case), H.batched_check(hostsim_lib, case))


@pytest.mark.parametrize("seed", [2, 5, 198, 77])
def test_small_lru_cache_is_emulated_exactly(oracle_lib, hostsim_lib, seed):
    """the reference's cache holds 10 000 keys; with room for 3 it evicts and re-computes all the time — and since a re-computation uses the
    limits of the job at hand, the answers may change with the cache size.  The batched flow replays the same Get / Add sequence."""
    case = random_case(seed)
    H.same_results(H.literal_check(oracle_lib, case, cache_size=3), H.batched_check(hostsim_lib, case, cache_size=3))


def test_literal_on_hostsim_equals_literal_on_oracle(oracle_lib, hostsim_lib):
    """the NodeDb-level entry points (txn_begin / schedule_many / txn_abort) on a cleared NodeDb, device code vs oracle"""
    case = random_case(5)
    H.same_results(H.literal_check(oracle_lib, case), H.literal_check(hostsim_lib, case))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_random_batched_on_gpu_equals_literal_on_oracle(oracle_lib, hip_lib, seed):
    case = random_case(seed)
    H.same_results(H.literal_check(oracle_lib, case), H.batched_check(hip_lib, case))


@pytest.mark.gpu
def test_larger_batch_on_gpu(oracle_lib, hip_lib):
    case = random_case(12, n_jobs=1500, scale=20)
    H.same_results(H.batched_check(oracle_lib, case), H.batched_check(hip_lib, case))


# ------------------------------------------------------------------------------------------------ ClearAllocated (nodedb.go:1178-1199)
def _clear_allocated(lib, oracle):
    case = copy.deepcopy(CASES[0])
    nodes = [{"index": i + 1, "total": {"cpu": 8000, "memory": 64 * GI}, "taints": [], "labels": {}, "used": {}, "unschedulable": False} for i in range(5)]
    jobs = [{"created": i, "queue": "q", "pc": ["priority-0", "priority-2-non-preemptible"][i % 2], "priority": 1000, "gang": None,
             "req": {"cpu": 1000 * (1 + i % 3), "memory": (2 + i) * GI}, "tolerations": [], "selector": {}, "affinity": None} for i in range(10)]
    out = []
    for l in (lib, oracle):
        c = H.Case(l, case["SchedulingConfig"], nodes)
        c.set_jobs(jobs, {"q": 0}, {})
        s = c.sched
        for i in range(9):                       # fill some buckets
            s.bind(i, i % 5, s.priorities[2 + i % 3])
        s.evict(0, 0)
        before = [s.get_alloc(n).copy() for n in range(5)]
        s.clear_allocated()
        after = [s.get_alloc(n).copy() for n in range(5)]
        for n in range(5):                       # every priority bucket back at allocatableResources
            assert (after[n] == np.array(H.vec(nodes[n]["total"]))[None, :]).all()
        assert any((b != a).any() for b, a in zip(before, after))
        # the nodes' job bookkeeping survives the clear (DeepCopyNilKeys clones the maps, node.go:344-372): unbinding a job that was
        # bound before the clear hands its request back on top of the cleared values
        s.unbind(1, 1)
        ok, pods, _ = (s.txn_begin(), s.schedule_many([9]))[1]
        s.txn_commit()
        out.append(([s.get_alloc(n).tolist() for n in range(5)], ok, pods[0].node))
    assert out[0] == out[1]


def test_clear_allocated_hostsim(hostsim_lib, oracle_lib):
    _clear_allocated(hostsim_lib, oracle_lib)


@pytest.mark.gpu
def test_clear_allocated_gpu(hip_lib, oracle_lib):
    _clear_allocated(hip_lib, oracle_lib)


def test_submit_check_argument_errors(hostsim_lib, oracle_lib):
    from armada_amd.binding import SchedError
    case = copy.deepcopy(CASES[0])
    for lib in (oracle_lib, hostsim_lib):
        pools, dbs = H.build_state(lib, case)
        db = dbs["cpu"]
        db.set_job_dicts(case["jobs"])
        assert db.case.sched.submit_check([], []) == []
        with pytest.raises(SchedError):
            db.case.sched.submit_check([[5]], [False])      # job index out of range
        with pytest.raises(SchedError):
            db.case.sched.submit_check([[]], [False])       # empty unit


# ------------------------------------------------------------------------------------------------ wide path (fit kernel) vs sequential path
def _wide_equals_sequential(lib, seed, monkeypatch):
    case = random_case(seed)
    pools, dbs = H.build_state(lib, case)
    n = len(case["jobs"])
    units, strip = [[i] for i in range(n)], [True] * n
    used_wide = 0
    for name, db in dbs.items():
        if not db.case:
            continue
        db.set_job_dicts(case["jobs"])
        s = db.case.sched
        monkeypatch.delenv("ASCHED_SUBMIT_WIDE", raising=False)
        a, st = s.submit_check(units, strip), s.submit_stats()
        monkeypatch.setenv("ASCHED_SUBMIT_WIDE", "0")
        b, st0 = s.submit_check(units, strip), s.submit_stats()
        monkeypatch.delenv("ASCHED_SUBMIT_WIDE", raising=False)
        assert a == b, f"pool {name}: the fit-kernel answers differ from one transaction per job"
        assert st0["wide_units"] == 0 and st0["sequential_units"] == n
        assert st["wide_units"] == n and st["sequential_units"] == 0 and 1 <= st["wide_passes"] <= 3, st   # home + up to two away entries
        used_wide += st["wide_units"]
    assert used_wide > 0


@pytest.mark.parametrize("seed", range(6))
def test_wide_path_equals_sequential_hostsim(hostsim_lib, seed, monkeypatch):
    _wide_equals_sequential(hostsim_lib, seed, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_wide_path_equals_sequential_gpu(hip_lib, seed, monkeypatch):
    _wide_equals_sequential(hip_lib, seed, monkeypatch)


def _pristine_tracking(lib, oracle):
    """ClearAllocated is persistent (a later jobs_set starts from allocatable, not from the upsert's alloc_by_prio table), and a NodeDb
    that holds a job is not pristine: its individual checks take the sequential path and see the occupancy."""
    cfg = copy.deepcopy(CASES[0]["SchedulingConfig"])
    nodes = [{"index": 1, "total": {"cpu": 4000, "memory": 64 * GI}, "taints": [], "labels": {}, "used": {"0": {"cpu": 3000}}, "unschedulable": False}]
    jobs = [{"created": i + 1, "queue": "q", "pc": "priority-0", "priority": 1000, "gang": None, "tolerations": [], "selector": {}, "affinity": None,
             "req": {"cpu": c, "memory": GI}} for i, c in enumerate((2000, 4000, 1000, 2000))]
    out = []
    for l in (lib, oracle):
        c = H.Case(l, cfg, nodes)
        s = c.sched
        c.set_jobs(jobs, {"q": 0}, {})
        before = s.submit_check([[0], [1], [2]], [True] * 3)          # 3 of 4 cpus used at priority 0 and below: only the 1-cpu job fits
        st_before = s.submit_stats()
        s.clear_allocated()
        c.set_jobs(jobs, {"q": 0}, {})                                # jobs_set after the clear: the clear must survive it
        assert (s.get_alloc(0) == np.array(H.vec(nodes[0]["total"]))[None, :]).all()
        cleared = s.submit_check([[0], [1], [2]], [True] * 3)
        st_cleared = s.submit_stats()
        pod, _ = s.select_node(1)                                     # a selection that finds a node does not bind it, but may have preempted: not pristine any more ...
        assert pod.node == 0
        s.clear_allocated(); c.set_jobs(jobs, {"q": 0}, {})
        big = dict(jobs[1], req={"cpu": 64000, "memory": GI}); c.set_jobs(jobs + [big], {"q": 0}, {})
        pod, _ = s.select_node(4)                                     # ... a selection that finds NONE touched nothing (nil, nil, nil): the batch paths stay on
        assert pod.node < 0 and s.excluded_nodes(4)
        again = s.submit_check([[0], [1], [2]], [True] * 3)
        if l is lib:
            assert s.submit_stats()["wide_units"] == 3 and again == cleared
        c.set_jobs(jobs, {"q": 0}, {})
        s.bind(3, 0, 0)                                               # 2 cpus taken: the 4-cpu job no longer fits, and the NodeDb is not pristine
        occupied = s.submit_check([[0], [1], [2]], [True] * 3)
        st_occ = s.submit_stats()
        out.append(([x[0] for x in before], [x[0] for x in cleared], [x[0] for x in occupied]))
        if l is lib:
            # (before: resources used at priority 0 lower the planes of every level up to the jobs' own priority alike — still the initial state, and the level-0 plane answers; since
            #  round 4 such a NodeDb takes the wide path too.  A NodeDb that holds a bound job does not.)
            assert st_before["wide_units"] == 3 and st_cleared["wide_units"] == 3 and st_occ["wide_units"] == 0, (st_before, st_cleared, st_occ)
    assert out[0] == out[1] == ([False, False, True], [True, True, True], [True, False, True]), out


def test_pristine_tracking_hostsim(hostsim_lib, oracle_lib):
    _pristine_tracking(hostsim_lib, oracle_lib)


@pytest.mark.gpu
def test_pristine_tracking_gpu(hip_lib, oracle_lib):
    _pristine_tracking(hip_lib, oracle_lib)


def _empty_nodedb(lib):
    """a pool no executor has reported nodes for: every check answers "not schedulable", on both paths, without touching a kernel"""
    from armada_amd.binding import Config, Scheduler
    s = Scheduler(lib, Config(num_resources=2, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0, 1.0]))
    s.nodes_upsert(np.zeros((0, 2), dtype=np.int64))
    s.clear_allocated()
    s.jobs_set(np.array([[1, 1], [2, 2], [2, 2]], dtype=np.int64), gang_id=[-1, 0, 0], gang_cardinality=[1, 2, 2])
    got = s.submit_check([[0], [1, 2]], [True, False])
    assert [g[0] for g in got] == [False, False] and got[0][3] == -1
    assert list(s.fit_select_batch([0, 1], -2)) == [-1, -1]


def test_units_given_as_numpy_arrays_are_units_not_csr(hostsim_lib):
    """round-4 advisor: a tuple of two numpy index arrays used to be re-read as (unit_off, unit_jobs); the CSR form is an explicit keyword now"""
    from armada_amd.binding import Config, Scheduler
    s = Scheduler(hostsim_lib, Config(num_resources=2, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0, 1.0]))
    s.nodes_upsert(np.array([[4, 4], [4, 4]], dtype=np.int64))
    s.clear_allocated()
    s.jobs_set(np.array([[1, 1], [3, 3], [3, 3]], dtype=np.int64), gang_id=[-1, 0, 0], gang_cardinality=[1, 2, 2])
    want = s.submit_check([[0], [1, 2]], [True, False])
    assert s.submit_check((np.array([0]), np.array([1, 2])), [True, False]) == want
    assert s.submit_check(None, [True, False], csr=(np.array([0, 1, 3]), np.array([0, 1, 2]))) == want
    with pytest.raises(ValueError):
        s.submit_check([[0]], [True], csr=(np.array([0, 1]), np.array([0])))
    s.close()


def test_empty_nodedb_oracle_and_hostsim(oracle_lib, hostsim_lib):
    _empty_nodedb(oracle_lib)
    _empty_nodedb(hostsim_lib)


@pytest.mark.gpu
def test_empty_nodedb_gpu(hip_lib):
    _empty_nodedb(hip_lib)


# ---- gang units, one workgroup per unit on a pristine NodeDb (csrc/submit_gang.h) -------------------------------------------------------------------------------
def _gang_units_equal_sequential(lib, oracle, seed, monkeypatch):
    """ScheduleManyWithTxn on units of several members (submitcheck.go:345-349): uniform units (every member of one shape: capacity sums over the nodes, any size — the
    reference's BenchmarkScheduleMany* shape) and the gang path (members one after the other inside ONE workgroup, binds in the workgroup's scratch, units side by side) against the sequential control launch (ASCHED_SUBMIT_GANGS=0) and against the oracle — gangs that fit, gangs whose k-th member finds no
    node (num_schedulable = k), members that share nodes, members bigger than any node, selectors / tolerations / off-grid requests (those units keep the sequential path)."""
    from armada_amd import workloads as W
    rng = np.random.default_rng(1000 + seed)
    wl = W.small_random(n_nodes=int(rng.integers(6, 120)), n_jobs=int(rng.integers(200, 900)), n_queues=3, seed=500 + seed, occupied=0.0, gangs=0,
                        ragged=bool(seed % 4 == 3), away=bool(seed % 5 == 4))
    m = wl.num_jobs
    if seed % 3 == 0:
        wl.job_req[:, W.CPU] *= 4   # big members: gangs run out of nodes half way
    units, strip = [], []
    perm = rng.permutation(m)
    i = 0
    while i < m:
        n = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 300])) if seed % 2 else int(rng.choice([1, 2, 3, 5, 8, 17, 40]))   # (300: beyond the gang kernel's scratch — uniform units are a sum over the nodes, others go sequential)
        u = [int(x) for x in perm[i:i + n]]
        i += n
        if len(u) > 1 and rng.random() < 0.7:   # a gang shares one priority class (and here one shape, as a gang's members usually do)
            wl.job_pc[u] = wl.job_pc[u[0]]
            if rng.random() < 0.6:
                wl.job_req[u] = wl.job_req[u[0]]
                if wl.job_req_class is not None:
                    wl.job_req_class[u] = wl.job_req_class[u[0]]
        units.append(u); strip.append(len(u) == 1)
    out, stats = [], []
    for l in (lib, oracle):
        s = W.load(l, wl)
        s.clear_allocated()
        monkeypatch.delenv("ASCHED_SUBMIT_GANGS", raising=False)
        out.append(s.submit_check(units, strip)); stats.append(s.submit_stats())
        if l is lib:
            monkeypatch.setenv("ASCHED_SUBMIT_GANGS", "0")
            out.append(s.submit_check(units, strip)); stats.append(s.submit_stats())
            monkeypatch.delenv("ASCHED_SUBMIT_GANGS", raising=False)
        s.close()
    assert out[0] == out[1], [(u, a, b) for u, a, b in zip(units, out[0], out[1]) if a != b][:3]
    assert out[0] == out[2], [(u, a, b) for u, a, b in zip(units, out[0], out[2]) if a != b][:3]
    assert stats[1]["gang_units"] == 0 and stats[1]["sequential_units"] >= stats[0]["sequential_units"] + stats[0]["gang_units"]
    multi = [o for u, o in zip(units, out[0]) if len(u) > 1]
    return stats[0]["gang_units"], sum(1 for o in multi if o[0]), sum(1 for o in multi if not o[0] and o[2] > 0)


def test_gang_units_equal_sequential_hostsim(hostsim_lib, oracle_lib, monkeypatch):
    g = ok = partial = 0
    for seed in range(12):
        a, b, c = _gang_units_equal_sequential(hostsim_lib, oracle_lib, seed, monkeypatch)
        g += a; ok += b; partial += c
    assert g > 100 and ok > 100 and partial > 10, (g, ok, partial)   # the path is taken, gangs fit, and gangs fail half way (num_schedulable of the failed unit is compared too)


@pytest.mark.gpu
def test_gang_units_equal_sequential_gpu(hip_lib, oracle_lib, monkeypatch):
    g = ok = partial = 0
    for seed in range(12):
        a, b, c = _gang_units_equal_sequential(hip_lib, oracle_lib, seed, monkeypatch)
        g += a; ok += b; partial += c
    assert g > 100 and ok > 100 and partial > 10, (g, ok, partial)


# ---- the rejection message of an individual check: pctx.String() with NumExcludedNodesByReason (submitcheck.go:372-381, scheduling/context/pod.go:62-83) ------------------
def test_quantity_and_pod_context_strings():
    """resource.Quantity.String() of NewScaledQuantity (int64Amount.AsCanonicalBytes: trailing zeros into the exponent, exponent down to a multiple of three) and the
    text/tabwriter layout of PodSchedulingContext.String() (minwidth 1, tabwidth 1, padding 1, ' '), worked out by hand from the two sources"""
    from armada_amd.submitcheck import pod_context_string, quantity_string
    assert [quantity_string(*x) for x in [(1000, -3), (1500, -3), (32000, -3), (5, -3), (4294967296, 0), (100, 0), (1000000, 0), (0, 0), (12000000, -3)]] == \
        ["1", "1500m", "32", "5m", "4294967296", "100", "1M", "0", "12k"]
    assert pod_context_string(3, [("taint foo=bar:NoSchedule not tolerated", 2), ("insufficient resources available", 1)]) == (
        "Node:                       none\n"
        "Number of nodes in cluster: 3\n"
        "Excluded nodes:\n"
        " 1: insufficient resources available\n"
        " 2: taint foo=bar:NoSchedule not tolerated\n")
    assert pod_context_string(12, [("insufficient resources available", 12)]).splitlines()[-1] == " 12: insufficient resources available"
    assert pod_context_string(1, []) == "Node:                       none\nNumber of nodes in cluster: 1\nExcluded nodes:             none\n"


def _explained(lib, case):
    return H.batched_check(lib, case, explain=True)


@pytest.mark.parametrize("seed", range(6))
def test_rejection_messages_carry_the_excluded_nodes_hostsim(oracle_lib, hostsim_lib, seed):
    """with explain=True a failing individual check reports why, per pool, like the reference's: the product's text (device records + host-derived static reasons) against the
    oracle's (its iterator walk), and the counts of every message add up to the pool's nodes"""
    case = random_case(seed)
    a, b = _explained(oracle_lib, case), _explained(hostsim_lib, case)
    H.same_results(a, b)
    seen = 0
    for k in a:
        assert a[k].reason == b[k].reason, (k, a[k].reason, b[k].reason)
        for block in a[k].reason.split("---\n"):
            if "Excluded nodes:" in block and "Excluded nodes:             none" not in block:
                n = int(block.split("Number of nodes in cluster:")[1].split()[0])
                counts = [int(line.split(":")[0]) for line in block.split("Excluded nodes:\n")[1].splitlines() if line.startswith(" ")]
                assert sum(counts) == n, block
                seen += 1
    assert seen > 0
    plain = H.batched_check(hostsim_lib, case)                       # the default stays the fixed sentence (and the fast batch: no extra selections)
    assert all(("Excluded nodes" not in r.reason) for r in plain.values())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_rejection_messages_carry_the_excluded_nodes_gpu(oracle_lib, hip_lib, seed):
    case = random_case(seed)
    a, b = _explained(oracle_lib, case), _explained(hip_lib, case)
    for k in a:
        assert a[k].reason == b[k].reason, (k, a[k].reason, b[k].reason)
