"""Replays the reference's table-driven test cases (tests/golden/*.json, transcribed by
tests/golden/extract_reference_goldens.py) through the C ABI.

The same driver runs against the CPU oracle (`-m "not gpu"`) and against the HIP library (`-m gpu`),
mirroring the Go test drivers it cites:
  run_pqs_case   <- is/scheduling/preempting_queue_scheduler_test.go:2216-2550
  run_qs_case    <- is/scheduling/queue_scheduler_test.go:479-600
  run_gang_case  <- is/scheduling/gang_scheduler_test.go:560-729
"""
from __future__ import annotations

import math
import os
from typing import Dict, List

import numpy as np

from armada_amd.binding import AFFINITY_OPS, AWAY_COND_OPS, Config, Library, Scheduler

RES = ["memory", "cpu", "nvidia.com/gpu", "test-floating-resource"]  # TestResourceListFactory column order
R = len(RES)
RES_UNIT = {"memory": 1, "cpu": 1000, "nvidia.com/gpu": 1000, "test-floating-resource": 1000}  # factory units per whole unit (gofixtures.SCALE)
EFFECTS = {"": 0, "NoSchedule": 1, "PreferNoSchedule": 2, "NoExecute": 3}
UNSCHEDULABLE_TAINT = ["node.kubernetes.io/unschedulable", "", "NoSchedule"]


def inf(v):
    if v == "inf":
        return math.inf
    if v == "-inf":
        return -math.inf
    return v


class Interner:
    def __init__(self):
        self.d: Dict[str, int] = {}

    def __call__(self, s: str) -> int:
        if s not in self.d:
            self.d[s] = len(self.d)
        return self.d[s]


def vec(d: dict) -> List[int]:
    return [int(d.get(r, 0)) for r in RES]


class Case:
    """Holds the interning tables and the Scheduler for one golden case."""

    def __init__(self, lib: Library, cfg: dict, nodes: List[dict], pool_limit_key: str = "pool"):
        self.lib = lib
        self.cfgj = cfg
        self.S = Interner()
        self.pc_names = sorted(cfg["priority_classes"])
        self.pc_index = {n: i for i, n in enumerate(self.pc_names)}
        pcs = cfg["priority_classes"]
        wkt_names = sorted(cfg["well_known_node_types"])
        wkt_index = {n: i for i, n in enumerate(wkt_names)}
        pc_away, pc_away_nt = [], []
        for n in self.pc_names:
            # an away entry naming a well-known node type the config does not define makes the reference return an error if it is ever
            # tried (nodedb.go:653-657); the tables that redefine WellKnownNodeTypes never reach such an entry: leave it out
            ents = [e for e in pcs[n].get("away", []) if e[1] == "" or e[1] in wkt_index]
            pc_away.append([(e[0], wkt_index.get(e[1], -1)) for e in ents])
            # AwayNodeType.NodeTypes: [[name, [[resource, operator, whole-unit value], ...]], ...] as the entry's optional third element
            pc_away_nt.append([[(wkt_index[name], [(RES.index(r) if r in RES else -1, AWAY_COND_OPS.get(op, 99), int(v)) for r, op, v in conds])
                                for name, conds in (e[2] if len(e) > 2 else [])] for e in ents])
        wkt_taints = [[(self.S(k), -1 if v == "*" else self.S(v), EFFECTS[e]) for k, v, e in cfg["well_known_node_types"][n]] for n in wkt_names]
        frac = dict(cfg.get("maximum_resource_fraction_to_schedule") or {})
        bypool = cfg.get("maximum_resource_fraction_to_schedule_by_pool") or {}
        if pool_limit_key in bypool:
            frac = dict(bypool[pool_limit_key])
        self.config = Config(
            num_resources=R,
            indexed_col=[RES.index(n) for n, _ in cfg["indexed_resources"]],
            indexed_resolution=[r for _, r in cfg["indexed_resources"]],
            pc_priority=[pcs[n]["priority"] for n in self.pc_names],
            pc_preemptible=[int(pcs[n]["preemptible"]) for n in self.pc_names],
            drf_multiplier=[1.0 if r in cfg["drf_resources"] else 0.0 for r in RES],
            pc_away=pc_away, pc_away_node_types=pc_away_nt, resource_unit=[RES_UNIT[r] for r in RES],
            wkt_taints=wkt_taints,
            indexed_taint_keys=[self.S(k) for k in cfg["indexed_taints"]],
            indexed_label_keys=[self.S(k) for k in cfg["indexed_node_labels"]],
            prefer_large_job_ordering=cfg["prefer_large_job_ordering"],
            disable_home_scheduling=cfg["disable_home"], disable_away_scheduling=cfg["disable_away"],
            disable_gang_away_scheduling=cfg["disable_gang_away"], disable_fairshare_scheduling=cfg["disable_fairshare"],
            disable_urgency_scheduling=cfg["disable_urgency"],
            preempt_cross_pool_jobs_first=bool(cfg.get("preempt_cross_pool_jobs_first", False)),   # PoolConfig.ShouldPreemptCrossPoolJobsFirst (hand-written cross-pool tests)
            protected_fraction_of_fair_share=cfg["protected_fraction_of_fair_share"],
            max_queue_lookback=cfg["max_queue_lookback"],
            max_fraction_to_schedule=[float(inf(frac.get(r, "inf"))) for r in RES],
            disallowed_resource=[int(r in (cfg.get("disallowed_resources") or [])) for r in RES],
            # FloatingResources: the pool's total per floating column (the scheduling context of the Go drivers is for pool "pool"), -1 = ordinary
            floating_resource_limit=([int((cfg["floating_resources"].get(r) or {}).get(pool_limit_key, 0)) if r in cfg["floating_resources"] else -1 for r in RES]
                                     if cfg.get("floating_resources") else None),
        )
        self.sched = Scheduler(lib, self.config)
        self.nodes = nodes
        self.upsert_nodes(set())

    def upsert_nodes(self, cordoned, excluded=()):
        """cordoned: nodes that carry the unschedulable taint; excluded: nodes that are not in this round's NodeDb at all (the market table's nodes on cordoned clusters)"""
        s = self.sched
        P = s.P
        self.global_of_local = [i for i in range(len(self.nodes)) if i not in set(excluded)]
        self.local_of_global = {g: l for l, g in enumerate(self.global_of_local)}
        cordoned = {self.local_of_global[g] for g in cordoned if g in self.local_of_global}
        nodes = self.nodes_now = [self.nodes[g] for g in self.global_of_local]
        total = np.array([vec(n["total"]) for n in nodes], dtype=np.int64).reshape(len(nodes), R)
        abp = np.repeat(total[:, None, :], P, axis=1)
        for i, n in enumerate(nodes):
            for p_str, used in (n.get("used") or {}).items():
                p = int(p_str)
                for l, prio in enumerate(s.priorities):
                    if prio <= p:  # MarkAllocated(allocatableByPriority, p, rl): internaltypes/node.go:535-549
                        abp[i, l, :] -= np.array(vec(used), dtype=np.int64)
        taints, labels = [], []
        for i, n in enumerate(nodes):
            ts = [list(t) for t in n["taints"]]
            if i in cordoned:
                ts.append(UNSCHEDULABLE_TAINT)
            taints.append([(self.S(k), self.S(v), EFFECTS[e]) for k, v, e in ts])
            labels.append([(self.S(k), self.S(v)) for k, v in sorted(n["labels"].items())])
        # nodes without hand-written usage are uploaded like production nodes — no explicit AllocatableByPriority — so that the golden tables run through
        # the level-0 fast structure, the stream runs and the ring, not only through the generic path (an explicit table turns the fast structure off)
        explicit = any(n.get("used") for n in nodes) or os.environ.get("ASCHED_GOLDEN_EXPLICIT_ALLOC") == "1"
        s.nodes_upsert(total, total, index=[n["index"] for n in nodes], alloc_by_prio=abp if explicit else None, taints=taints, labels=labels)

    # ---- jobs
    def set_jobs(self, jobs: List[dict], queue_index: Dict[str, int], running: Dict[int, tuple]):
        """jobs: list of job dicts (golden format); running: local idx -> (node, prio, run_ts)"""
        classes: Dict[tuple, int] = {}
        cls_tol, cls_sel, cls_aff = [], [], []
        req_class, gang_id, gang_card, gang_uni = [], [], [], []
        gangs: Dict[tuple, int] = {}
        for j in jobs:
            tol = tuple((-1 if t["key"] == "" else self.S(t["key"]), 1 if t["op"] == "Exists" else 0, self.S(t["value"]), EFFECTS[t["effect"]]) for t in j["tolerations"])
            sel = tuple(sorted((self.S(k), self.S(v)) for k, v in j["selector"].items()))
            aff = None
            if j.get("affinity") is not None:  # required node affinity: list of terms, a term = list of [key, operator, values]
                aff = tuple(tuple((self.S(k), AFFINITY_OPS[op], tuple(self.S(v) for v in vals)) for k, op, vals in term) for term in j["affinity"])
            key = (tol, sel, aff)
            if key not in classes:
                classes[key] = len(classes)
                cls_tol.append(list(tol))
                cls_sel.append(list(sel))
                cls_aff.append(None if aff is None else [[(k, op, list(vals)) for k, op, vals in term] for term in aff])
            req_class.append(classes[key])
            g = j.get("gang")
            if g:
                gk = (j["queue"], g["id"])
                if gk not in gangs:
                    gangs[gk] = len(gangs)
                gang_id.append(gangs[gk])
                gang_card.append(g["cardinality"])
                gang_uni.append(self.S(g["uniformity"]) if g["uniformity"] else -1)
            else:
                gang_id.append(-1); gang_card.append(1); gang_uni.append(-1)
        if not classes:
            cls_tol, cls_sel, cls_aff = [[]], [[]], [None]
        m = len(jobs)
        node = [running[i][0] if i in running else -1 for i in range(m)]
        sap = [running[i][1] if i in running else 0 for i in range(m)]
        rts = [running[i][2] if i in running else 0 for i in range(m)]
        # the interned strings that strconv.ParseInt(s, 10, 64) accepts: what the Gt / Lt affinity operators compare
        ints = {}
        for text, vid in self.S.d.items():
            try:
                if text.strip() == text and text.lstrip("+-").isdigit() and -2 ** 63 <= int(text) < 2 ** 63:
                    ints[vid] = int(text)
            except ValueError:
                pass
        self.sched.set_label_value_ints(ints)
        self.sched.jobs_set(
            np.array([vec(j["req"]) for j in jobs], dtype=np.int64).reshape(m, R),
            queue=[queue_index.get(j["queue"], -1) for j in jobs],
            pc=[self.pc_index[j["pc"]] for j in jobs],
            queue_priority=[j["priority"] for j in jobs],
            submit_time=[j["created"] for j in jobs],
            req_class=req_class, gang_id=gang_id, gang_cardinality=gang_card, gang_uniformity_label=gang_uni,
            node=node, scheduled_at_priority=sap, run_timestamp=rts,
            class_tolerations=cls_tol, class_selectors=cls_sel, class_affinities=cls_aff,
            away=[1 if j.get("away") else 0 for j in jobs] if any(j.get("away") for j in jobs) else None,
            bid_price=self.bid_prices(jobs, running) if (self.cfgj.get("market") or {}).get("Enabled") else None)

    def sort_queued(self, jobs: List[dict], idxs: List[int]) -> List[int]:
        """SchedulingOrderCompare for queued (non-active) jobs: jobdb/comparison.go:49-107; on a market-driven pool jobdb.PriceOrder = MarketSchedulingOrderCompare (:113-170)"""
        pcs = self.cfgj["priority_classes"]
        if (self.cfgj.get("market") or {}).get("Enabled"):
            return sorted(idxs, key=lambda i: (-pcs[jobs[i]["pc"]]["priority"], -float(jobs[i].get("price_band", 0)), jobs[i]["created"], i))
        return sorted(idxs, key=lambda i: (-pcs[jobs[i]["pc"]]["priority"], jobs[i]["priority"], jobs[i]["created"], i))

    def bid_prices(self, jobs: List[dict], running: Dict[int, tuple]):
        """job.GetBidPrice(pool) (jobdb/job.go:459-481) with testfixtures.SetPricing (:573-585): QueuedBid = RunningBid = the price band's number, a running
        non-preemptible job bids pricing.NonPreemptibleRunningPrice"""
        pcs = self.cfgj["priority_classes"]
        return [1_000_000.0 if (i in running and not pcs[j["pc"]]["preemptible"]) else float(j.get("price_band", 0)) for i, j in enumerate(jobs)]

    def pc_limits(self, queues: List[str], queue_cfgs=None, pool="pool"):
        """calculatePerQueueLimits (constraints.go:218-256): PC defaults, overridden per queue and per (queue, pool)."""
        pcs = self.cfgj["priority_classes"]
        out = np.full((len(queues), len(self.pc_names), R), math.inf)
        any_limit = False
        for pi, n in enumerate(self.pc_names):
            for r, f in (pcs[n].get("max_fraction_per_queue") or {}).items():
                out[:, pi, RES.index(r)] = float(inf(f)); any_limit = True
        for qi, qc in enumerate(queue_cfgs or []):
            for pc, lim in (qc.get("ResourceLimitsByPriorityClassName") or {}).items():
                pi = self.pc_index[pc]
                for r, f in (lim.get("MaximumResourceFraction") or {}).items():
                    out[qi, pi, RES.index(r)] = float(inf(f)); any_limit = True
                bypool = (lim.get("MaximumResourceFractionByPool") or {}).get(pool)
                if bypool:
                    for r, f in (bypool.get("MaximumResourceFraction") or {}).items():
                        out[qi, pi, RES.index(r)] = float(inf(f)); any_limit = True
        return out if any_limit else None

    def no_oversubscription(self):
        s = self.sched
        for n in range(len(self.nodes_now)):
            a = s.get_alloc(n)
            for l, p in enumerate(s.priorities):
                if p >= 0:
                    assert (a[l] >= 0).all(), f"node {n} oversubscribed at priority {p}: {a[l]}"


def uses_unsupported(case: dict, jobs: List[dict]) -> str:
    pcs = case["SchedulingConfig"]["priority_classes"]
    for j in jobs:
        if any(op not in AFFINITY_OPS for term in (j.get("affinity") or []) for _, op, _ in term):
            return "unknown node affinity operator"
        if "test-floating-resource" in j["req"] and "test-floating-resource" not in (case["SchedulingConfig"].get("floating_resources") or {}):
            return "floating resource requested without FloatingResources in the config"
    return ""


class Tokens:
    """golang.org/x/time/rate limiter state as seen at sctx.Started of each round (new limiter starts full)."""

    def __init__(self, rate, burst):
        self.rate, self.burst = float(inf(rate)), int(burst)
        self.tokens = float(self.burst)

    @property
    def rate_inf(self):
        return math.isinf(self.rate)

    def advance(self, seconds):
        if not self.rate_inf:
            self.tokens = min(float(self.burst), self.tokens + self.rate * seconds)


# how the golden rounds ran (asched_round_stats; the oracle reports zeros): summed over every schedule call of the drivers below, read by
# the tests that assert the reference tables reach the fast structure / the stream runs / the preempting fast iteration
GOLDEN_STATS: Dict[str, int] = {}


def note_stats(s):
    st = s.round_stats()
    GOLDEN_STATS["rounds"] = GOLDEN_STATS.get("rounds", 0) + 1
    GOLDEN_STATS["rounds_with_fast_iterations"] = GOLDEN_STATS.get("rounds_with_fast_iterations", 0) + (1 if st["fast_iterations"] > 0 else 0)
    for k in ("fast_iterations", "generic_iterations", "stream_runs", "stream_jobs", "preempt_fast_iterations", "fast_replay_steps", "l0_overflows"):
        GOLDEN_STATS[k] = GOLDEN_STATS.get(k, 0) + int(st[k])


def run_pqs_case(lib: Library, case: dict):
    cfg = case["SchedulingConfig"]
    nodes = case["Nodes"]
    all_jobs = [j for r in case["Rounds"] for js in (r.get("JobsByQueue") or {}).values() for j in js]
    all_jobs += [j for js in (case.get("InitialRunningJobs") or {}).values() for j in js]
    why = uses_unsupported(case, all_jobs)
    if why:
        return "skip: " + why
    c = Case(lib, cfg, nodes)
    s = c.sched
    queues = sorted(case["PriorityFactorByQueue"])
    qidx = {q: i for i, q in enumerate(queues)}
    Q = len(queues)
    npc = len(c.pc_names)
    weight = [1.0 / case["PriorityFactorByQueue"][q] for q in queues]
    glim = Tokens(cfg["maximum_scheduling_rate"], cfg["maximum_scheduling_burst"])
    qlim = [Tokens(cfg["maximum_per_queue_scheduling_rate"], cfg["maximum_per_queue_scheduling_burst"]) for _ in queues]
    pcs = cfg["priority_classes"]

    live: List[dict] = []  # jobs in the jobDb (running ones carry 'run')
    run_counter = 0
    for node_idx in sorted(case.get("InitialRunningJobs") or {}, key=int):
        for j in case["InitialRunningJobs"][node_idx]:
            run_counter += 1
            j = dict(j); j["run"] = (int(node_idx), pcs[j["pc"]]["priority"], run_counter); j["round"] = -1; j["idx"] = -1
            live.append(j)
    allocated = np.zeros((Q, npc, R), dtype=np.int64)  # allocatedByQueueAndPriorityClass (test accounting)
    demand = np.zeros((Q, R), dtype=np.int64)
    node_by_job: Dict[int, int] = {}
    cordoned = set()
    for ri, rnd in enumerate(case["Rounds"]):
        assert not rnd.get("IndicesToUnbind"), "IndicesToUnbind not supported by the driver"
        for idx in rnd.get("NodeIndicesToCordon") or []:
            cordoned.add(int(idx))
        # market_driven_preempting_queue_scheduler_test.go:697-705: nodes on cordoned clusters are not added to this round's NodeDb (and the jobs running there are not bound)
        excluded = set(int(i) for i in rnd.get("IndiciesOfNodesOnCordonedCluster") or [])
        c.upsert_nodes(cordoned, excluded)
        queued_now = []
        for q, js in (rnd.get("JobsByQueue") or {}).items():
            for k, j in enumerate(js):
                j = dict(j); j["round"] = ri; j["idx"] = k
                queued_now.append(j)
                demand[qidx[q]] += np.array(vec(j["req"]), dtype=np.int64)
        jobs = live + queued_now
        running = {i: (c.local_of_global[j["run"][0]], j["run"][1], j["run"][2]) for i, j in enumerate(jobs) if "run" in j and j["run"][0] in c.local_of_global}
        c.set_jobs(jobs, qidx, running)
        queued = [c.sort_queued(jobs, [i for i, j in enumerate(jobs) if "run" not in j and j["queue"] == q]) for q in queues]
        s.round_prepare(weight, queued, name_rank=list(range(Q)), allocated_by_pc=allocated, demand=demand,
                        pc_resource_limit_fraction=c.pc_limits(queues),
                        global_tokens=glim.tokens, global_burst=glim.burst, global_rate_inf=glim.rate_inf,
                        queue_tokens=[t.tokens for t in qlim], queue_burst=[t.burst for t in qlim], queue_rate_inf=[t.rate_inf for t in qlim])
        # the experimental fairness optimiser runs in the rounds the table marks (pqs_test.go:236, 2330-2346: optimiserEnabled of NewPreemptingQueueScheduler)
        oc = cfg.get("optimiser")
        if oc and rnd.get("OptimiserEnabled"):
            def rl(v):   # armadaresource.ComputeResources -> factory vector (FromJobResourceListIgnoreUnknown)
                return None if v is None else vec(v)
            s.set_optimiser(True, min_improvement_pct=oc.get("MinimumFairnessImprovementPercentage", 0.0), max_jobs_per_round=oc["MaximumJobsPerRound"],
                            max_job_size_to_preempt=rl(oc.get("MaximumJobSizeToPreempt")), min_job_size_to_schedule=rl(oc.get("MinimumJobSizeToSchedule")),
                            max_resource_fraction_to_schedule=[float(inf(oc.get("MaximumResourceFractionToSchedule", {}).get(r, "inf"))) for r in RES], now_ms=0)
        else:
            s.set_optimiser(False)
        market = cfg.get("market") or {}
        s.set_market(bool(market.get("Enabled")), float(market.get("SpotPriceCutoff", 0.0)))
        res = s.schedule_round()
        res.scheduled = {j: c.global_of_local[n] for j, n in res.scheduled.items()}
        res.preempted = {j: c.global_of_local[n] for j, n in res.preempted.items()}
        note_stats(s)
        glim.tokens = res.global_tokens_after
        for t, v in zip(qlim, res.queue_tokens_after):
            t.tokens = float(v)
        glim.advance(1.0)
        for t in qlim:
            t.advance(1.0)
        # resource accounting (pqs_test.go:2387-2411)
        for job in res.preempted:
            j = jobs[job]
            allocated[qidx[j["queue"]], c.pc_index[j["pc"]]] -= np.array(vec(j["req"]), dtype=np.int64)
        for job in res.scheduled:
            j = jobs[job]
            allocated[qidx[j["queue"]], c.pc_index[j["pc"]]] += np.array(vec(j["req"]), dtype=np.int64)
        assert (allocated == res.queue_allocated_by_pc).all(), f"round {ri}: queue accounting differs"
        # node mapping checks (:2413-2447)
        for job, node in res.preempted.items():
            assert "run" in jobs[job] and jobs[job]["run"][0] == node, f"round {ri}: job preempted from unexpected node"
        for job, node in res.scheduled.items():
            assert node >= 0
            key = id(jobs[job].get("_ident", None)) if False else (jobs[job]["round"], jobs[job]["queue"], jobs[job]["idx"])
            if key in node_by_job:
                assert node_by_job[key] == node, f"round {ri}: job moved between nodes"
            node_by_job[key] = node
        # expected scheduled (:2449-2464)
        exp_s = rnd.get("ExpectedScheduledIndices") or {}
        got_s: Dict[str, List[int]] = {}
        for job in res.scheduled:
            got_s.setdefault(jobs[job]["queue"], []).append(jobs[job]["idx"])
        for q in set(exp_s) | set(got_s):
            assert sorted(exp_s.get(q) or []) == sorted(got_s.get(q, [])), \
                f"round {ri}: scheduling from queue {q}: expected {sorted(exp_s.get(q) or [])} got {sorted(got_s.get(q, []))}"
        # expected preempted (:2466-2487)
        exp_p = rnd.get("ExpectedPreemptedIndices") or {}
        got_p: Dict[str, Dict[int, List[int]]] = {}
        for job in res.preempted:
            got_p.setdefault(jobs[job]["queue"], {}).setdefault(jobs[job]["round"], []).append(jobs[job]["idx"])
        for q in set(exp_p) | set(got_p):
            e = {int(k): sorted(v) for k, v in (exp_p.get(q) or {}).items()}
            g = {k: sorted(v) for k, v in got_p.get(q, {}).items()}
            assert e == g, f"round {ri}: preempting from queue {q}: expected {e} got {g}"
        # expected market data (market_driven_preempting_queue_scheduler_test.go:855-887)
        emd = rnd.get("ExpectedMarketData")
        if emd:
            mr = s.market_result()
            assert (mr["spot_price"] or 0.0) == float(emd.get("ExpectedSpotPrice") or 0.0), f"round {ri}: spot price {mr['spot_price']} expected {emd.get('ExpectedSpotPrice')}"
            for q, exp in (emd.get("ExpectedBillableResource") or {}).items():
                assert mr["billable"][qidx[q]].tolist() == vec(exp), f"round {ri}: billable resource of {q}: {mr['billable'][qidx[q]].tolist()} expected {vec(exp)}"
            for q, exp in (emd.get("ExpectedBillablePriceOverride") or {}).items():
                assert mr["price_override"][qidx[q]] == (None if exp is None else float(exp)), f"round {ri}: billable price override of {q}: {mr['price_override'][qidx[q]]} expected {exp}"
            one = emd.get("ExpectedExactlyOneOverrideWithValue")
            if one is not None:
                have = [v for v in mr["price_override"] if v is not None]
                assert have == [float(one)], f"round {ri}: expected exactly one billable price override of {one}, got {mr['price_override']}"
        c.no_oversubscription()
        # jobDb update (:2499-2547): queued jobs deleted, preempted failed, scheduled get runs in submit order
        new_live = [j for i, j in enumerate(jobs) if "run" in j and i not in res.preempted]
        for job in sorted(res.scheduled, key=lambda i: jobs[i]["created"]):
            run_counter += 1
            j = jobs[job]
            j["run"] = (res.scheduled[job], res.scheduled_priority[job], run_counter)
            new_live.append(j)
        live = new_live
    return "ok"


def run_qs_case(lib: Library, case: dict):
    cfg = case["SchedulingConfig"]
    jobs = case["Jobs"]
    why = uses_unsupported(case, jobs)
    if why:
        return "skip: " + why
    c = Case(lib, cfg, case["Nodes"])
    s = c.sched
    qs = case["Queues"]
    queues = [q["Name"] for q in qs]
    rank = {n: i for i, n in enumerate(sorted(queues))}
    qidx = {q: i for i, q in enumerate(queues)}
    Q, npc = len(queues), len(c.pc_names)
    c.set_jobs(jobs, qidx, {})
    allocated = np.zeros((Q, npc, R), dtype=np.int64)
    for q, bypc in (case.get("InitialAllocatedByQueueAndPriorityClass") or {}).items():
        for pc, rlist in bypc.items():
            allocated[qidx[q], c.pc_index[pc]] = vec(rlist)
    demand = np.zeros((Q, R), dtype=np.int64)
    for j in jobs:
        demand[qidx[j["queue"]]] += np.array(vec(j["req"]), dtype=np.int64)
    queued = [c.sort_queued(jobs, [i for i, j in enumerate(jobs) if j["queue"] == q]) for q in queues]
    glim = Tokens(cfg["maximum_scheduling_rate"], cfg["maximum_scheduling_burst"])
    qlim = Tokens(cfg["maximum_per_queue_scheduling_rate"], cfg["maximum_per_queue_scheduling_burst"])
    s.round_prepare([1.0 / float(q["PriorityFactor"]) for q in qs], queued, name_rank=[rank[n] for n in queues],
                    allocated_by_pc=allocated, demand=demand, pc_resource_limit_fraction=c.pc_limits(queues, qs),
                    global_tokens=glim.tokens, global_burst=glim.burst, global_rate_inf=glim.rate_inf,
                    queue_tokens=[qlim.tokens] * Q, queue_burst=[qlim.burst] * Q, queue_rate_inf=[qlim.rate_inf] * Q)
    res = s.schedule_queues()
    note_stats(s)
    exp = sorted(case.get("ExpectedScheduledIndices") or [])
    assert sorted(res.scheduled) == exp, f"expected scheduled {exp} got {sorted(res.scheduled)}"
    for i in case.get("ExpectedNeverAttemptedIndices") or []:
        assert i not in res.scheduled and res.job_unschedulable_reason[i] == 0, f"job {i} was attempted"
    check_excluded_nodes(s, jobs, res.job_unschedulable_reason, len(case["Nodes"]))
    return "ok"


def check_excluded_nodes(s, jobs, reasons, num_nodes):
    """queue_scheduler_test.go:676-690: for every job that could not be scheduled and has a PodSchedulingContext, the counts of NumExcludedNodesByReason
    add up to the number of nodes (gang members are exempt there too).  asched_excluded_nodes: [] = no failed node selection on record (never attempted, skipped
    by key, failed a constraint first); ASCHED_ERR_UNSUPPORTED = documented as not produced."""
    from armada_amd.binding import SchedError
    for i, j in enumerate(jobs):
        if reasons[i] == 0 or j.get("gang"):
            continue
        try:
            h = s.excluded_nodes(i)
        except SchedError:
            continue
        if h:
            assert sum(x[-1] for x in h) == num_nodes, f"job {i}: NumExcludedNodesByReason {h} does not add up to {num_nodes} nodes"


def run_gang_case(lib: Library, case: dict):
    cfg = case["SchedulingConfig"]
    gangs = case["Gangs"]
    jobs = [j for g in gangs for j in g]
    why = uses_unsupported(case, jobs)
    if why:
        return "skip: " + why
    if case.get("TotalResources") and any(case["TotalResources"].values()):
        return "skip: explicit TotalResources"
    c = Case(lib, cfg, case["Nodes"])
    s = c.sched
    queues = sorted({j["queue"] for j in jobs})
    if case.get("AddAwayQueueContexts"):   # gang_scheduler_test.go:610-617: an "<queue>-away" context beside every queue (context.CalculateAwayQueueName)
        queues = sorted(set(queues) | {q + "-away" for q in queues})
    qidx = {q: i for i, q in enumerate(queues)}
    Q = len(queues)
    # a cross-pool away job is accounted against its queue's away context (context/scheduling.go:412-413)
    jobs = [dict(j, queue=j["queue"] + "-away") if j.get("away") else j for j in jobs]
    c.set_jobs(jobs, qidx, {})
    glim = Tokens(cfg["maximum_scheduling_rate"], cfg["maximum_scheduling_burst"])
    qlim = Tokens(cfg["maximum_per_queue_scheduling_rate"], cfg["maximum_per_queue_scheduling_burst"])
    s.round_prepare([1.0] * Q, [[] for _ in queues], demand=np.zeros((Q, R), dtype=np.int64),
                    allocated_by_pc=np.zeros((Q, len(c.pc_names), R), dtype=np.int64), pc_resource_limit_fraction=c.pc_limits(queues),
                    global_tokens=glim.tokens, global_burst=glim.burst, global_rate_inf=glim.rate_inf,
                    queue_tokens=[qlim.tokens] * Q, queue_burst=[qlim.burst] * Q, queue_rate_inf=[qlim.rate_inf] * Q)
    base = 0
    actual = []
    scheduled_gangs = 0
    for gi, g in enumerate(gangs):
        ids = list(range(base, base + len(g)))
        base += len(g)
        ok, reason, pods = s.gang_schedule(ids)
        fit = sum(1 for p in pods if p.node >= 0)
        assert fit == case["ExpectedRuntimeGangCardinality"][gi], f"gang {gi}: runtime cardinality {fit}"
        cnt = s.round_counters()
        if ok:
            assert reason == 0
            actual.append(gi)
            scheduled_gangs += 1
            uni = g[0]["gang"]["uniformity"] if g[0].get("gang") else ""
            if uni:
                vals = {case["Nodes"][p.node]["labels"].get(uni) for p in pods if p.node >= 0}
                assert len(vals) == 1 and None not in vals, f"gang {gi}: uniformity not met {vals}"
                expv = (case.get("ExpectedNodeUniformity") or {}).get(str(gi))
                if expv is not None:
                    assert vals == {expv}, f"gang {gi}: uniformity value {vals} expected {expv}"
        else:
            assert reason != 0
            assert all(p.node < 0 for p in pods)
        assert cnt["num_scheduled_gangs"] == scheduled_gangs
        assert cnt["num_scheduled_jobs"] == case["ExpectedCumulativeScheduledJobs"][gi], f"gang {gi}: cumulative {cnt}"
        assert cnt["num_evicted_jobs"] == 0
    assert actual == (case.get("ExpectedScheduledIndices") or []), f"scheduled gangs {actual}"
    exp_unf = sorted(case.get("ExpectedUnfeasibleSchedulingKeyIndices") or [])
    starts = np.cumsum([0] + [len(g) for g in gangs])
    for gi in exp_unf:
        assert s.job_key_unfeasible(int(starts[gi])), f"gang {gi}: key not marked unfeasible"
    assert s.round_counters()["num_unfeasible_keys"] == len(set(exp_unf)) or len(exp_unf) == s.round_counters()["num_unfeasible_keys"]
    return "ok"


def run_nodedb_schedule_case(lib: Library, case: dict):
    """nodedb_test.go TestScheduleIndividually (:502-667) / TestScheduleMany (:668-743): each job (or group) goes through
    ScheduleManyWithTxn inside a write transaction that is committed on success and aborted on failure; the expectation is the
    success flag per job / group, plus (on success) a node for every member and its request accounted on that node."""
    groups = case["Jobs"]
    if groups and isinstance(groups[0], dict):
        groups = [[j] for j in groups]
    jobs = [j for g in groups for j in g]
    why = uses_unsupported(case, jobs)
    if why:
        return "skip: " + why
    c = Case(lib, case["SchedulingConfig"], case["Nodes"])
    s = c.sched
    queues = sorted({j["queue"] for j in jobs})
    c.set_jobs(jobs, {q: i for i, q in enumerate(queues)}, {})
    base = 0
    for gi, g in enumerate(groups):
        ids = list(range(base, base + len(g)))
        base += len(g)
        before = {n: s.get_alloc(n).copy() for n in range(len(case["Nodes"]))}
        s.txn_begin()
        ok, pods, _ = s.schedule_many(ids)
        assert ok == case["ExpectSuccess"][gi], f"group {gi}: success {ok}, expected {case['ExpectSuccess'][gi]}"
        if not ok:
            if len(g) == 1:   # (the same property at NodeDb level: a single job that found no node accounts for every node)
                h = s.excluded_nodes(ids[0])
                assert h and sum(x[-1] for x in h) == len(case["Nodes"]), f"group {gi}: NumExcludedNodesByReason {h}"
            s.txn_abort()
            for n in before:  # the aborted transaction leaves the NodeDb exactly as it was
                assert (s.get_alloc(n) == before[n]).all(), f"group {gi}: abort did not restore node {n}"
            continue
        s.txn_commit()
        assert all(p.node >= 0 for p in pods), f"group {gi}: a scheduled member has no node"
        exp = case.get("ExpectAway")
        if exp:  # TestAwayNodeScheduling :1399-1404: the node, ScheduledAsAwayJob, the away type's priority
            assert pods[0].node == exp["node"] and pods[0].method == 5 and pods[0].scheduled_at_priority == exp["priority"], f"away result {pods[0]}"
    return "ok"


def run_node_iteration_case(lib: Library, case: dict) -> List[int]:
    """nodedb/nodeiteration_test.go:308-372 and :637-695: node ids are "0".."n-1" (string order breaks ties
    across node types); TestNodeTypesIterator also forces node.index = i."""
    import gofixtures_min as gf  # noqa: F401  (config only)
    cfg = gf.test_scheduling_config()
    nodes = case["nodes"]
    merged = "nodeTypeIds" in case
    c = Case.__new__(Case)
    c.lib, c.cfgj, c.S = lib, cfg, Interner()
    c.pc_names = sorted(cfg["priority_classes"])
    c.pc_index = {n: i for i, n in enumerate(c.pc_names)}
    pcs = cfg["priority_classes"]
    c.config = Config(num_resources=R, indexed_col=[RES.index(n) for n, _ in cfg["indexed_resources"]],
                      indexed_resolution=[r for _, r in cfg["indexed_resources"]],
                      pc_priority=[pcs[n]["priority"] for n in c.pc_names], pc_preemptible=[int(pcs[n]["preemptible"]) for n in c.pc_names],
                      drf_multiplier=[1.0, 1.0, 1.0, 0.0])
    s = c.sched = Scheduler(lib, c.config)
    n = len(nodes)
    total = np.array([vec(x["total"]) for x in nodes], dtype=np.int64).reshape(n, R)
    abp = np.repeat(total[:, None, :], s.P, axis=1)
    for i, x in enumerate(nodes):
        for p_str, used in (x.get("used") or {}).items():
            for l, prio in enumerate(s.priorities):
                if prio <= int(p_str):
                    abp[i, l, :] -= np.array(vec(used), dtype=np.int64)
    order = sorted(range(n), key=lambda i: str(i))
    rank = [0] * n
    for r_, i in enumerate(order):
        rank[i] = r_
    index = list(range(n)) if merged else [x["index"] for x in nodes]
    s.nodes_upsert(total, total, index=[i + 1 for i in index] if merged else index, id_rank=rank, alloc_by_prio=abp,
                   node_type_override=[x.get("node_type", 0) for x in nodes])
    req = case.get("resourceRequests") or {}
    ireq = [int(req.get(name, 0)) for name, _ in cfg["indexed_resources"]]
    types = case["nodeTypeIds"] if merged else [case["nodeTypeId"]]
    return s.iterate_nodes(int(case.get("priority", 0)), ireq, types)


def run_fairness_case(lib: Library, case: dict):
    cfg = case["config"]
    names = ["foo", "bar", "baz"]
    mult = {n: 0.0 for n in names}
    experimental = cfg.get("ExperimentalDominantResourceFairnessResourcesToConsider") or []
    for pool in cfg.get("Pools") or []:   # NewDominantResourceFairness (fairness.go:50-56): the pool's own list replaces the experimental one
        if pool.get("Name") == "pool" and pool.get("DominantResourceFairnessResourcesToConsider"):
            experimental = pool["DominantResourceFairnessResourcesToConsider"]
    for n in cfg.get("DominantResourceFairnessResourcesToConsider") or []:
        mult[n] = 1.0
    for r in experimental:
        m = r.get("Multiplier", 0)
        mult[r["Name"]] = float(m) if m > 0 else 1.0  # defaultMultiplier fairness.go:91-97
    s = Scheduler(lib, Config(num_resources=3, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1],
                              drf_multiplier=[mult[n] for n in names]))
    cost = s.drf_cost([case["allocation"][n] for n in names], [case["totalResources"][n] for n in names])
    return cost / float(case["weight"])  # WeightedCostFromAllocation fairness.go:99-101


def run_fair_share_case(lib: Library, case: dict):
    s = Scheduler(lib, Config(num_resources=1, indexed_col=[0], indexed_resolution=[1], pc_priority=[0], pc_preemptible=[1], drf_multiplier=[1.0]))
    queues = sorted(case["queueCtxs"])
    total = [case["availableResources"].get("cpu", 0)]
    cds = []
    for q in queues:
        d = [case["queueCtxs"][q]["Demand"].get("cpu", 0)]
        cds.append(1.0 if total[0] == 0 else s.drf_cost(d, total))  # scheduling.go:267-270
    f, dc, uc = s.fair_shares(list(range(len(queues))), [float(case["queueCtxs"][q]["Weight"]) for q in queues], cds)
    return {q: (f[i], dc[i], uc[i]) for i, q in enumerate(queues)}


def assert_same_round(a, b):
    """bit-exact equality of two RoundResults (job->node, priorities, methods, preemptions, queue accounting, reasons)"""
    assert a.scheduled == b.scheduled, "job->node assignments differ"
    assert a.scheduled_priority == b.scheduled_priority
    assert a.scheduled_method == b.scheduled_method
    assert a.preempted == b.preempted, "preempted sets differ"
    assert a.termination_reason == b.termination_reason
    assert a.num_evicted_phase1 == b.num_evicted_phase1 and a.num_evicted_phase3 == b.num_evicted_phase3
    assert (a.queue_allocated_by_pc == b.queue_allocated_by_pc).all()
    assert (a.job_unschedulable_reason == b.job_unschedulable_reason).all()
    # float64 DRF / fair shares: the stated tolerance is zero (same operation order, -ffp-contract=off)
    assert (a.fair_share == b.fair_share).all() and (a.demand_capped_adjusted_fair_share == b.demand_capped_adjusted_fair_share).all()
    assert (a.uncapped_adjusted_fair_share == b.uncapped_adjusted_fair_share).all()
    assert a.global_tokens_after == b.global_tokens_after and (a.queue_tokens_after == b.queue_tokens_after).all()
