"""Seeded synthetic workloads for the BASELINE.json configurations.

Resource columns (R=4), in the order of the simulator's basicSchedulingConfig.yaml supportedResourceTypes:
    0 memory [bytes]   1 cpu [milli]   2 ephemeral-storage [bytes]   3 nvidia.com/gpu [count]
Indexed resources (K=3): cpu @ 1 core, memory @ 128Mi, gpu @ 1.  All requests are multiples of the index
resolutions and node ids are zero-padded (id order == index order), as §8d prescribes.

Everything is data: numpy arrays in the layout of include/armada_sched.h.  `load()` pushes a workload into
any backend that implements the C ABI (the HIP library, or the oracle in tests/bench).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .binding import Config, Scheduler

Gi = 1024 ** 3
Mi = 1024 ** 2
SEED = 20260921
R = 4
MEM, CPU, EPH, GPU = 0, 1, 2, 3


@dataclass
class Workload:
    name: str
    config: Config
    node_total: np.ndarray            # [N][R]
    job_req: np.ndarray               # [M][R]
    job_queue: np.ndarray
    job_pc: np.ndarray
    job_submit: np.ndarray
    job_node: np.ndarray              # -1 queued
    job_run_prio: np.ndarray
    job_run_ts: np.ndarray
    job_gang: np.ndarray
    job_gang_card: np.ndarray
    queue_weight: np.ndarray
    queued: List[np.ndarray]          # per queue, job ids in SchedulingOrderCompare order
    global_burst: int = 2 ** 62
    queue_burst: int = 2 ** 62
    rate_inf: bool = True
    meta: dict = field(default_factory=dict)
    node_taints: Optional[list] = None   # per node: list of (key, value, effect)
    node_labels: Optional[list] = None   # per node: list of (key, value)
    node_allocatable: Optional[np.ndarray] = None
    node_id_rank: Optional[np.ndarray] = None
    job_req_class: Optional[np.ndarray] = None
    class_tolerations: Optional[list] = None   # per class: list of (key, op, value, effect)
    class_selectors: Optional[list] = None     # per class: list of (key, value)
    job_away: Optional[np.ndarray] = None      # [M] cross-pool away jobs (asched_jobs.away); None = none

    @property
    def num_nodes(self):
        return self.node_total.shape[0]

    @property
    def num_jobs(self):
        return self.job_req.shape[0]

    @property
    def num_queues(self):
        return len(self.queue_weight)


def _config(pcs, *, protected=0.0, prefer_large=True, max_lookback=0) -> Config:
    return Config(
        num_resources=R,
        indexed_col=[CPU, MEM, GPU], indexed_resolution=[1000, 128 * Mi, 1],
        pc_priority=[p for p, _ in pcs], pc_preemptible=[int(x) for _, x in pcs],
        drf_multiplier=[1.0, 1.0, 0.0, 1.0],   # dominantResourceFairnessResourcesToConsider: cpu, memory, gpu
        prefer_large_job_ordering=prefer_large, protected_fraction_of_fair_share=protected, max_queue_lookback=max_lookback,
    )


def _sort_queued(queue, pc_prio, submit, ids):
    """per-queue SchedulingOrderCompare order for queued jobs (jobdb/comparison.go:49-107): PC priority desc, submit asc, id"""
    order = np.lexsort((ids, submit[ids], -pc_prio[ids], queue[ids]))
    ids = ids[order]
    q = queue[ids]
    nq = int(queue.max()) + 1 if len(queue) else 0
    bounds = np.searchsorted(q, np.arange(nq + 1))
    return [ids[bounds[i]:bounds[i + 1]].astype(np.int32) for i in range(nq)]


def _fill_nodes(rng, node_total, target_frac, shapes, queue_p, pc_p, pc_prio, only=None):
    """running jobs: fill each node's cpu to ~target_frac with random shapes that fit"""
    n = node_total.shape[0]
    reqs, nodes = [], []
    free = node_total.astype(np.int64).copy()
    target = (node_total[:, CPU] * target_frac).astype(np.int64)
    used = np.zeros(n, dtype=np.int64)
    active = np.arange(n) if only is None else only
    for _ in range(64):
        if len(active) == 0:
            break
        s = shapes[rng.integers(0, len(shapes), size=len(active))]
        fits = (s <= free[active]).all(axis=1) & (used[active] + s[:, CPU] <= target[active])
        idx = active[fits]
        free[idx] -= s[fits]
        used[idx] += s[fits][:, CPU]
        reqs.append(s[fits]); nodes.append(idx)
        active = active[(used[active] < target[active])]
    req = np.concatenate(reqs) if reqs else np.zeros((0, R), dtype=np.int64)
    node = np.concatenate(nodes) if nodes else np.zeros(0, dtype=np.int64)
    m = len(req)
    queue = rng.choice(len(queue_p), size=m, p=queue_p)
    pc = rng.choice(len(pc_p), size=m, p=pc_p)
    return req, node.astype(np.int32), queue.astype(np.int32), pc.astype(np.int32), pc_prio[pc].astype(np.int32)


def _shapes16():
    return np.array([[m * Gi, c * 1000, 0, 0] for c in (1, 2, 4, 8) for m in (4, 8, 16, 32)], dtype=np.int64)


def _shapes64(gpu_frac_rows=True):
    rows = [[m * Gi, c * 1000, e * Gi, 0] for c in (1, 2, 4, 8) for m in (4, 8, 16, 32) for e in (10, 50, 100, 200)]
    return np.array(rows, dtype=np.int64)


def config1(n_nodes=100, n_jobs=1_000, seed=SEED) -> Workload:
    """100 nodes, 1 queue, 1k single-pod jobs on an empty cluster — the shape of the reference's cmd/simulator basic input
    (BASELINE config 1: the CPU-runnable case); 32-core nodes, jobs of 1-8 cores, everything that fits is scheduled in one round."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pcs = [(0, True)]
    node_total = np.tile(np.array([256 * Gi, 32000, 1024 * Gi, 0], dtype=np.int64), (n_nodes, 1))
    shapes = _shapes16()
    q_req = shapes[rng.integers(0, len(shapes), size=n_jobs)]
    none = np.zeros(0, np.int32)
    return _assemble("config1", _config(pcs), node_total, np.zeros((0, R), np.int64), none, none, none, none,
                     q_req, np.zeros(n_jobs, np.int32), np.zeros(n_jobs, np.int32), np.array([0]), np.ones(1), {})


def config2(n_nodes=10_000, n_jobs=100_000, n_queues=8, seed=SEED) -> Workload:
    """10k nodes, 8 queues, 100k jobs, 4 resource dims — the nodedb fit kernel (BASELINE config 2)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pcs = [(0, True)]
    node_total = np.tile(np.array([256 * Gi, 32000, 1024 * Gi, 0], dtype=np.int64), (n_nodes, 1))
    shapes = _shapes16()
    pre = rng.choice(n_nodes, size=n_nodes // 10, replace=False)
    # 10% of the nodes pre-filled with k in [0,31] cpus (one running job each, 4Gi per cpu)
    k = rng.integers(0, 32, size=len(pre))
    pre, k = pre[k > 0], k[k > 0]
    run_req = np.stack([k * 4 * Gi, k * 1000, np.zeros_like(k), np.zeros_like(k)], axis=1).astype(np.int64)
    run_queue = rng.integers(0, n_queues, size=len(pre)).astype(np.int32)
    q_req = shapes[rng.integers(0, len(shapes), size=n_jobs)]
    q_queue = rng.integers(0, n_queues, size=n_jobs).astype(np.int32)
    return _assemble("config2", _config(pcs), node_total, run_req, pre.astype(np.int32), run_queue, np.zeros(len(pre), np.int32),
                     np.zeros(len(pre), np.int32), q_req, q_queue, np.zeros(n_jobs, np.int32), np.array([0]), np.ones(n_queues), {})


def config3(n_nodes=100_000, n_jobs=1_000_000, n_queues=64, seed=SEED, gangs=0, occupied=0.5) -> Workload:
    """100k nodes, 64 queues, 1M jobs with DRF weights + rate limits (BASELINE config 3; gangs>0 gives config 4's shape)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pcs = [(0, True), (1, True), (3, False)]
    pc_prio = np.array([p for p, _ in pcs])
    node_total = np.tile(np.array([256 * Gi, 32000, 1024 * Gi, 0], dtype=np.int64), (n_nodes, 1))
    gpu_nodes = rng.choice(n_nodes, size=n_nodes // 10, replace=False)
    node_total[gpu_nodes, GPU] = 8
    pf = rng.integers(1, 11, size=n_queues).astype(np.float64)
    weight = 1.0 / pf
    shapes = _shapes64()
    pc_p = np.array([0.6, 0.3, 0.1])
    run_req, run_node, run_queue, run_pc, run_prio = _fill_nodes(rng, node_total, occupied, shapes, weight / weight.sum(), pc_p, pc_prio)
    zipf = 1.0 / np.arange(1, n_queues + 1) ** 1.1
    zipf /= zipf.sum()
    q_queue = rng.choice(n_queues, size=n_jobs, p=zipf).astype(np.int32)
    q_req = shapes[rng.integers(0, len(shapes), size=n_jobs)].copy()
    is_gpu = rng.random(n_jobs) < 0.05
    q_req[is_gpu, GPU] = 1
    q_pc = rng.choice(3, size=n_jobs, p=pc_p).astype(np.int32)
    wl = _assemble("config3" if not gangs else "config4", _config(pcs, protected=0.5), node_total, run_req, run_node, run_queue, run_pc, run_prio,
                   q_req, q_queue, q_pc, pc_prio, weight, {"priority_factor": pf.tolist()}, gangs=gangs, rng=rng)
    wl.global_burst, wl.queue_burst, wl.rate_inf = 200_000, 20_000, False
    return wl


def reference_benchmark(n_nodes: int, n_queues: int, jobs_per_queue: int) -> Workload:
    """BenchmarkPreemptingQueueScheduler's input (preempting_queue_scheduler_test.go:2561-2700): testfixtures.N32CpuNodes (32 cpu / 256 Gi), N1Cpu4GiJobs of PriorityClass0 in
    `n_queues` queues of weight 1, TestSchedulingConfig (unlimited rate limiters, protectedFractionOfFairShare 1.0, prefer-large ordering on), every job queued."""
    pcs = [(0, True)]
    node_total = np.tile(np.array([256 * Gi, 32000, 1024 * Gi, 0], dtype=np.int64), (n_nodes, 1))
    n_jobs = n_queues * jobs_per_queue
    q_req = np.tile(np.array([4 * Gi, 1000, 0, 0], dtype=np.int64), (n_jobs, 1))
    q_queue = np.repeat(np.arange(n_queues, dtype=np.int32), jobs_per_queue)
    none = np.zeros(0, np.int32)
    wl = _assemble("reference_benchmark", _config(pcs, protected=1.0), node_total, np.zeros((0, R), np.int64), none, none, none, none,
                   q_req, q_queue, np.zeros(n_jobs, np.int32), np.array([0]), np.ones(n_queues), {})
    return wl


def default_indexed(n_nodes=20_000, n_jobs=200_000, n_queues=64, seed=SEED, occupied=0.5, aligned=True) -> Workload:
    """config3's shape on the reference's DEFAULT indexedResources (config/scheduler/config.yaml:121-129: nvidia.com/gpu @1, cpu @100m, memory @100Mi,
    ephemeral-storage @1Gi — four indexed columns, in that order) and current hardware: 64 cpu / 1 TiB / 4 TiB ephemeral / 8 gpu per node.  The order key needs
    4 + 10 + 14 + 13 bits of fields + the node-index rank: 58 bits at 100 000 nodes (the round-2 layout widened every field for oversubscription and refused
    this).  1 TiB is not a multiple of 100Mi: allocatable is off the index grid.  aligned=True: requests are multiples of the resolutions (memory in units of
    100Mi), so node selection stays on the packed-key path; False: memory in GiB like a real job — literal iteration (exact, slow)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pcs = [(0, True), (1, True), (3, False)]
    pc_prio = np.array([p for p, _ in pcs])
    node_total = np.tile(np.array([1024 * Gi, 64000, 4096 * Gi, 8], dtype=np.int64), (n_nodes, 1))
    pf = rng.integers(1, 11, size=n_queues).astype(np.float64)
    weight = 1.0 / pf
    mem_unit = 100 * Mi if aligned else Gi
    mems = (40, 80, 160, 320) if aligned else (4, 8, 16, 32)
    shapes = np.array([[m * mem_unit, c * 1000, e * Gi, 0] for c in (1, 2, 4, 8) for m in mems for e in (10, 50, 100, 200)], dtype=np.int64)
    pc_p = np.array([0.6, 0.3, 0.1])
    run_req, run_node, run_queue, run_pc, run_prio = _fill_nodes(rng, node_total, occupied, shapes, weight / weight.sum(), pc_p, pc_prio)
    zipf = 1.0 / np.arange(1, n_queues + 1) ** 1.1
    zipf /= zipf.sum()
    q_queue = rng.choice(n_queues, size=n_jobs, p=zipf).astype(np.int32)
    q_req = shapes[rng.integers(0, len(shapes), size=n_jobs)].copy()
    q_req[rng.random(n_jobs) < 0.05, GPU] = 1
    q_req[rng.random(n_jobs) < 0.2, CPU] += 500          # cpu @100m: half cores are on the grid
    q_pc = rng.choice(3, size=n_jobs, p=pc_p).astype(np.int32)
    cfg = _config(pcs, protected=0.5)
    cfg.indexed_col = [GPU, CPU, MEM, EPH]
    cfg.indexed_resolution = [1, 100, 100 * Mi, Gi]
    cfg.drf_multiplier = [1.0, 1.0, 1.0, 1.0]
    wl = _assemble("default_indexed", cfg, node_total, run_req, run_node, run_queue, run_pc, run_prio, q_req, q_queue, q_pc, pc_prio, weight, {"priority_factor": pf.tolist()}, rng=rng)
    wl.global_burst, wl.queue_burst, wl.rate_inf = max(1, n_jobs // 5), max(1, n_jobs // 50), False
    return wl


def fine_indexed(n_nodes=20_000, n_jobs=200_000, n_queues=64, seed=SEED, occupied=0.5, k5=True, coarse=False) -> Workload:
    """default_indexed's pool on FINE index resolutions and (k5) a fifth indexed resource: nvidia.com/gpu @1, cpu @1m, memory @1Mi, ephemeral-storage @1Mi,
    a local-nvme column @1Gi on 64 cpu / 1 TiB / 4 TiB / 8 gpu / 32 TiB nodes.  The fields of the order key (nodedb/encoding.go:37-54) need 4 + 16 + 21 + 23 + 16 = 80 bits
    and the node-index rank 15-20 more: beyond one 64-bit word at any node count — the layout rounds 1-5 refused (ASCHED_ERR_UNSUPPORTED) and round 6 serves with a
    two-word key (armada_amd/csrc/armada_sched_wk.hip).  Requests are multiples of the resolutions, one node type: no literal iteration.
    coarse=True: the same resolutions, but the jobs ask in quarter cores, 100Mi and GiB steps like real pods — every value a column can take is a multiple of 250m / 32Mi /
    2Gi, the key may divide by those instead (asched_host.inc layoutKeys) and, without the nvme column, fits one word again: the fast path."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pcs = [(0, True), (1, True), (3, False)]
    pc_prio = np.array([p for p, _ in pcs])
    cols = 5 if k5 else 4
    node = [1024 * Gi, 64000, 4096 * Gi, 8] + ([32768 * Gi] if k5 else [])
    node_total = np.tile(np.array(node, dtype=np.int64), (n_nodes, 1))
    pf = rng.integers(1, 11, size=n_queues).astype(np.float64)
    weight = 1.0 / pf
    odd = 0 if coarse else 1
    rows = [[m * 100 * Mi + odd * 3 * Mi, c * 1000 + 250, e * Gi + odd * 7 * Mi, 0] + ([d * Gi] if k5 else []) for c in (1, 2, 4, 8) for m in (40, 80, 160, 320) for e, d in ((10, 100), (50, 0), (100, 2000), (200, 500))]
    shapes = np.array(rows, dtype=np.int64)
    pc_p = np.array([0.6, 0.3, 0.1])
    run_req, run_node, run_queue, run_pc, run_prio = _fill_nodes(rng, node_total, occupied, shapes, weight / weight.sum(), pc_p, pc_prio)
    if len(run_req) == 0:
        run_req = np.zeros((0, cols), dtype=np.int64)
    zipf = 1.0 / np.arange(1, n_queues + 1) ** 1.1
    zipf /= zipf.sum()
    q_queue = rng.choice(n_queues, size=n_jobs, p=zipf).astype(np.int32)
    q_req = shapes[rng.integers(0, len(shapes), size=n_jobs)].copy()
    q_req[rng.random(n_jobs) < 0.05, GPU] = 1
    q_req[rng.random(n_jobs) < 0.2, CPU] += 250 if coarse else 125
    q_pc = rng.choice(3, size=n_jobs, p=pc_p).astype(np.int32)
    cfg = _config(pcs, protected=0.5)
    cfg.num_resources = cols
    cfg.indexed_col = [GPU, CPU, MEM, EPH] + ([4] if k5 else [])
    cfg.indexed_resolution = [1, 1, Mi, Mi] + ([Gi] if k5 else [])
    cfg.drf_multiplier = [1.0] * cols
    wl = _assemble("fine_indexed", cfg, node_total, run_req, run_node, run_queue, run_pc, run_prio, q_req, q_queue, q_pc, pc_prio, weight, {"priority_factor": pf.tolist()}, rng=rng)
    wl.global_burst, wl.queue_burst, wl.rate_inf = max(1, n_jobs // 5), max(1, n_jobs // 50), False
    return wl


def small_random(n_nodes=64, n_jobs=600, n_queues=5, seed=1, occupied=0.6, gangs=4, burst=None, away=False, ragged=False, offgrid=0) -> Workload:
    """a small adversarially mixed workload for HIP-vs-oracle differential tests (preemption, gangs, limits; away=True: a quarter
    of the nodes carry a well-known node type's taint and two priority classes may run there at a reduced priority)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    pcs = [(0, True), (1, True), (3, False)]
    pc_prio = np.array([p for p, _ in pcs])
    node_total = np.tile(np.array([64 * Gi, 16000, 512 * Gi, 0], dtype=np.int64), (n_nodes, 1))
    node_total[rng.choice(n_nodes, size=max(1, n_nodes // 8), replace=False), GPU] = 4
    weight = 1.0 / rng.integers(1, 5, size=n_queues).astype(np.float64)
    shapes = _shapes64()
    pc_p = np.array([0.5, 0.3, 0.2])
    run_req, run_node, run_queue, run_pc, run_prio = _fill_nodes(rng, node_total, occupied, shapes, weight / weight.sum(), pc_p, pc_prio)
    q_queue = rng.integers(0, n_queues, size=n_jobs).astype(np.int32)
    q_req = shapes[rng.integers(0, len(shapes), size=n_jobs)].copy()
    q_req[rng.random(n_jobs) < 0.1, GPU] = 1
    q_pc = rng.choice(3, size=n_jobs, p=pc_p).astype(np.int32)
    wl = _assemble(f"small{seed}", _config(pcs, protected=float(rng.choice([0.0, 0.5, 1.0]))), node_total, run_req, run_node, run_queue, run_pc, run_prio,
                   q_req, q_queue, q_pc, pc_prio, weight, {}, gangs=gangs, rng=rng)
    if burst:
        wl.global_burst, wl.queue_burst, wl.rate_inf = burst[0], burst[1], False
    if ragged:
        # nothing resolution-aligned: several node types (an indexed label, a taint), allocatable below total by odd amounts, node ids
        # in a different order than node indexes, requests off the index grid, classes with tolerations / selectors.  Exercises
        # the literal NodeTypeIterator / NodeTypesIterator restatement (nodeiteration.go) instead of the packed-key argmin.
        wl.config.indexed_label_keys = [3]
        lab = rng.integers(0, 4, size=n_nodes)
        wl.node_labels = [[(3, int(v))] if v < 3 else [] for v in lab]
        taint = rng.random(n_nodes) < 0.2
        wl.node_taints = [[(9, 1, 1)] if t else [] for t in taint]
        alloc = wl.node_total.copy()
        alloc[:, CPU] -= rng.integers(0, 24, size=n_nodes) * 137
        alloc[:, MEM] -= rng.integers(0, 1000, size=n_nodes) * 1000003
        wl.node_allocatable = np.maximum(alloc, 0)
        wl.node_id_rank = rng.permutation(n_nodes).astype(np.int32)
        m = wl.num_jobs
        odd = rng.random(m) < 0.5
        wl.job_req[odd, CPU] += rng.integers(1, 8, size=int(odd.sum())) * 125
        oddm = rng.random(m) < 0.3
        wl.job_req[oddm, MEM] += rng.integers(1, 100, size=int(oddm.sum())) * 1000
        for g in np.unique(wl.job_gang[wl.job_gang >= 0]):      # gang members share one shape
            mem = np.nonzero(wl.job_gang == g)[0]
            wl.job_req[mem] = wl.job_req[mem[0]]
        wl.class_tolerations = [[], [(9, 1, 0, 0)], [], [(9, 0, 1, 1)]]           # class 1: Exists on key 9; class 3: key 9 == 1, NoSchedule
        wl.class_selectors = [[], [], [(3, 1)], [(3, 2)]]
        cls = rng.choice(4, size=m, p=[0.5, 0.25, 0.15, 0.1]).astype(np.int32)
        for g in np.unique(wl.job_gang[wl.job_gang >= 0]):
            mem = np.nonzero(wl.job_gang == g)[0]
            cls[mem] = cls[mem[0]]
        wl.job_req_class = cls
    if offgrid:
        # ONE node type (unlike `ragged`), so the level-0 fast structure stays on: bit 0 = a third of the jobs — running ones too — ask for cpu / memory off the
        # index grid (literal iteration for those shapes next to fast iterations for the others; an evicted one returning to its node must not be rebound by
        # key arithmetic); bit 1 = allocatable below total by odd amounts (the fast structure's key fields are floor(allocatable / resolution)); on a crowded
        # cluster that also leaves a few columns negative
        m = wl.num_jobs
        if offgrid & 1:
            odd = rng.random(m) < 0.33
            wl.job_req[odd, CPU] += rng.integers(1, 8, size=int(odd.sum())) * 125
            oddm = rng.random(m) < 0.2
            wl.job_req[oddm, MEM] += rng.integers(1, 100, size=int(oddm.sum())) * 1000
            for g in np.unique(wl.job_gang[wl.job_gang >= 0]):      # gang members share one shape
                mem = np.nonzero(wl.job_gang == g)[0]
                wl.job_req[mem] = wl.job_req[mem[0]]
        if offgrid & 2:
            alloc = wl.node_total.copy()
            alloc[:, CPU] -= rng.integers(0, 24, size=n_nodes) * 137
            alloc[:, MEM] -= rng.integers(0, 1000, size=n_nodes) * 1000003
            wl.node_allocatable = np.maximum(alloc, 0)
    if away:
        tainted = rng.random(n_nodes) < 0.25
        wl.node_taints = [[(7, 1, 1)] if t else [] for t in tainted]      # (key, value, NoSchedule)
        wl.config.wkt_taints = [[(7, 1, 1)], [(7, -1, 1)]]                 # type 0: exact value, type 1: wildcard value
        wl.config.pc_away = [[], [(0, 0)], [(2, 1), (1, 0)]]                # pc1 away at priority 0; pc2 away at 2 then at 1
    return wl


def _assemble(name, cfg, node_total, run_req, run_node, run_queue, run_pc, run_prio, q_req, q_queue, q_pc, pc_prio, weight, meta,
              gangs=0, rng=None) -> Workload:
    nr, nq = len(run_req), len(q_req)
    m = nr + nq
    req = np.concatenate([run_req, q_req]).astype(np.int64)
    queue = np.concatenate([run_queue, q_queue]).astype(np.int32)
    pc = np.concatenate([run_pc, q_pc]).astype(np.int32)
    node = np.concatenate([run_node, np.full(nq, -1, np.int32)]).astype(np.int32)
    run_prio_all = np.concatenate([run_prio, np.zeros(nq, np.int32)]).astype(np.int32)
    submit = np.arange(m, dtype=np.int64)
    run_ts = np.concatenate([np.arange(nr, dtype=np.int64), np.zeros(nq, np.int64)])
    gang = np.full(m, -1, np.int32)
    card = np.ones(m, np.int32)
    if gangs and rng is not None:
        # queued gangs: consecutive same-queue jobs get one gang id, uniform shape + priority class within the gang
        qids = np.arange(nr, m)
        by_queue = [qids[queue[qids] == q] for q in range(len(weight))]
        gid = 0
        for _ in range(gangs):
            q = int(rng.integers(0, len(weight)))
            c = int(rng.integers(2, 65))
            pool = by_queue[q]
            pool = pool[gang[pool] < 0]
            if len(pool) < c:
                continue
            start = int(rng.integers(0, len(pool) - c + 1))
            members = pool[start:start + c]
            gang[members] = gid; card[members] = c
            req[members] = req[members[0]]; pc[members] = pc[members[0]]
            # members must be adjacent in the queue order: give them the same submit time ordering block
            submit[members] = submit[members[0]] + np.arange(c) * 0  # keep relative order by id
            gid += 1
    pcp = np.asarray(pc_prio)[pc]
    queued = _sort_queued(queue, pcp, submit, np.arange(nr, m))
    while len(queued) < len(weight):
        queued.append(np.zeros(0, np.int32))
    return Workload(name=name, config=cfg, node_total=node_total, job_req=req, job_queue=queue, job_pc=pc, job_submit=submit, job_node=node,
                    job_run_prio=run_prio_all, job_run_ts=run_ts, job_gang=gang, job_gang_card=card, queue_weight=np.asarray(weight, dtype=np.float64),
                    queued=queued, meta=meta)


def load(lib, wl: Workload) -> Scheduler:
    """create a handle, upload nodes and jobs (untimed input build)"""
    s = Scheduler(lib, wl.config)
    s.nodes_upsert(wl.node_total, wl.node_allocatable, taints=wl.node_taints, labels=wl.node_labels, id_rank=wl.node_id_rank)
    set_jobs(s, wl)
    return s


def set_jobs(s: Scheduler, wl: Workload, job_node=None, job_run_prio=None, bid_price=None):
    """jobs_set with the workload's job table (optionally with other run placements: the queue-hash round re-uploads the accepted state)"""
    s.jobs_set(wl.job_req, queue=wl.job_queue, pc=wl.job_pc, submit_time=wl.job_submit, node=wl.job_node if job_node is None else job_node,
               scheduled_at_priority=wl.job_run_prio if job_run_prio is None else job_run_prio, run_timestamp=wl.job_run_ts, gang_id=wl.job_gang,
               gang_cardinality=wl.job_gang_card, req_class=wl.job_req_class, class_tolerations=wl.class_tolerations, class_selectors=wl.class_selectors, away=wl.job_away, bid_price=bid_price)


def prepare(s: Scheduler, wl: Workload, fairshare_preemption_tokens=None):
    """round_prepare with the workload's queues and rate limiters (limiters start full, as a fresh rate.Limiter)"""
    q = wl.num_queues
    s.round_prepare(wl.queue_weight, wl.queued,
                    global_tokens=float(wl.global_burst), global_burst=wl.global_burst, global_rate_inf=wl.rate_inf,
                    queue_tokens=[float(wl.queue_burst)] * q, queue_burst=[wl.queue_burst] * q, queue_rate_inf=[wl.rate_inf] * q,
                    fairshare_preemption_tokens=fairshare_preemption_tokens)
