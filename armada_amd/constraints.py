"""Resolution of the reference's resource-limit configuration into the arrays the C ABI takes.

The round itself only sees numbers: `asched_config.max_resource_fraction_to_schedule[R]` (per-round limit) and
`asched_round_input.pc_resource_limit_fraction[Q][npc][R]` (per queue and priority class; +inf = no limit), which the library multiplies with the
pool's total resources (multiplyResource, internaltypes/resource_list.go:312-331).  WHERE those fractions come from is configuration logic that lives on
the host side of the boundary — in the Go scheduler it is constraints.NewSchedulingConstraints; the cgo shim (integration/gpu_round.go) restates it in Go.
This module is the same logic for Python hosts, and the place where it is tested (tests/test_z_constraints_levels.py transcribes TestConstraints and
TestCapResources, constraints_test.go:32-294):

  per_round_fractions     calculatePerRoundLimits   constraints.go:197-212   MaximumResourceFractionToSchedule, replaced WHOLE by the pool's map if one exists
  per_queue_fractions     calculatePerQueueLimits   constraints.go:214-256   four levels, each overriding the previous one resource by resource:
                                                                             priority class < priority class by pool < queue < queue by pool
  resource_limit          ResourceList.Multiply     resource_list.go:251-262, 312-331
  cap_resources           CapResources              constraints.go:184-195   (ResourceList.Cap, resource_list.go:206-219)

Configuration objects are plain dicts with the reference's field names (the JSON / YAML spelling of configuration.SchedulingConfig, types.PriorityClass and
api.Queue), so a host can pass what it parsed.
"""
import math
from typing import Dict, List, Mapping, Optional, Sequence

INT64_MAX = 2**63 - 1
INT64_MIN = -(2**63)


def merge_maps(*maps: Optional[Mapping[str, float]]) -> Dict[str, float]:
    """util.MergeMaps (internal/common/util): later maps override earlier ones key by key; None counts as empty."""
    out: Dict[str, float] = {}
    for m in maps:
        if m:
            out.update(m)
    return out


def _fraction_list(fractions: Mapping[str, float], resources: Sequence[str], default: float) -> List[float]:
    """ResourceListFactory.MakeResourceFractionList(m, defaultValue): one slot per factory resource, names the factory does not know are ignored."""
    return [float(fractions.get(r, default)) for r in resources]


def per_round_fractions(config: Mapping, pool: str, resources: Sequence[str]) -> List[float]:
    """calculatePerRoundLimits (constraints.go:197-212): the pool-specific map REPLACES the general one (no merge — the reference says so itself)."""
    frac = config.get("MaximumResourceFractionToSchedule") or {}
    by_pool = config.get("MaximumResourceFractionToScheduleByPool") or {}
    if pool in by_pool:
        frac = by_pool[pool] or {}
    return _fraction_list(frac, resources, math.inf)


def per_queue_fractions(priority_classes: Mapping[str, Mapping], queues: Sequence[Mapping], pool: str, resources: Sequence[str],
                        pc_names: Sequence[str]) -> List[List[List[float]]]:
    """calculatePerQueueLimits (constraints.go:214-256) -> [queue][priority class][resource], +inf where nothing limits.

    priority_classes: name -> {"MaximumResourceFractionPerQueue": {...}, "MaximumResourceFractionPerQueueByPool": {pool: {...}}}
    queues: [{"Name": ..., "ResourceLimitsByPriorityClassName": {pc: {"MaximumResourceFraction": {...}, "MaximumResourceFractionByPool": {pool: {"MaximumResourceFraction": {...}}}}}}]
    A priority class the configuration does not define gets no limits (the reference's map has no entry: GetQueueResourceLimit returns the empty list)."""
    out = []
    for queue in queues:
        per_pc = []
        by_pc = queue.get("ResourceLimitsByPriorityClassName") or {}
        for name in pc_names:
            pc = priority_classes.get(name)
            if pc is None:
                per_pc.append([math.inf] * len(resources))
                continue
            fractions = merge_maps(pc.get("MaximumResourceFractionPerQueue"), (pc.get("MaximumResourceFractionPerQueueByPool") or {}).get(pool))
            qc = by_pc.get(name)
            if qc is not None:
                fractions = merge_maps(fractions, qc.get("MaximumResourceFraction"))
                qp = (qc.get("MaximumResourceFractionByPool") or {}).get(pool)
                if qp is not None:
                    fractions = merge_maps(fractions, qp.get("MaximumResourceFraction"))
            per_pc.append(_fraction_list(fractions, resources, math.inf))
        out.append(per_pc)
    return out


def multiply_resource(res: int, multiplier: float) -> int:
    """multiplyResource (resource_list.go:312-331): exact for 1.0, saturating for +-inf, truncation toward zero otherwise."""
    if multiplier == 1.0:
        return int(res)
    if math.isinf(multiplier):
        return INT64_MAX if (multiplier < 0) == (res < 0) else INT64_MIN
    return int(float(res) * multiplier)


def resource_limit(total: Sequence[int], fractions: Sequence[float]) -> List[int]:
    """ResourceList.Multiply (resource_list.go:251-262): what the library computes from a row of fractions; an all-zero total means "no limits at all"
    (calculatePerQueueLimits returns early on an empty total)."""
    return [multiply_resource(int(t), float(f)) for t, f in zip(total, fractions)]


def cap_resources(limit_by_pc: Optional[Mapping[str, Sequence[int]]], resources_by_pc: Mapping[str, Sequence[int]]) -> Dict[str, List[int]]:
    """CapResources (constraints.go:184-195): per priority class min(resources, limit); a queue without limits keeps its resources; a priority class without
    a limit row is capped by the EMPTY list, which Cap treats as "no cap" (resource_list.go:211-213)."""
    if limit_by_pc is None:
        return {pc: [int(v) for v in r] for pc, r in resources_by_pc.items()}
    out = {}
    for pc, r in resources_by_pc.items():
        lim = limit_by_pc.get(pc)
        out[pc] = [int(v) for v in r] if lim is None else [min(int(v), int(c)) for v, c in zip(r, lim)]
    return out


# ---- report figures of the scheduling context that hosts compute from the round's fair shares (context/scheduling.go); the float work is the library's
def theoretical_share(sched, name_rank: Sequence[int], weight: Sequence[float], constrained_demand_share: Sequence[float], priority: float, total: Sequence[int]) -> float:
    """SchedulingContext.CalculateTheoreticalShare (context/scheduling.go:240-256; scheduling_algo.go:984 fills ExperimentalIndicativeShares with it): the
    demand-capped adjusted fair share a NEW queue of weight 1/priority with infinite demand would get — updateFairShares (asched_fair_shares) over the round's
    queues plus that one.  Its name is a fresh ULID: digits first, it sorts in front of the queues' names (the order only fixes the float summation order)."""
    q = len(weight)
    all_max = sched.drf_cost([INT64_MAX] * len(total), list(total))   # UnweightedCostFromAllocation(MakeAllMax())
    ranks = [0] + [int(r) + 1 for r in name_rank]
    _, dc, _ = sched.fair_shares(ranks, [1.0 / float(priority)] + [float(w) for w in weight], [all_max] + [float(c) for c in constrained_demand_share])
    return float(dc[0]) if q + 1 == len(dc) else float("nan")


def fairness_error(sched, allocated: Sequence[Sequence[int]], total: Sequence[int], demand_capped_adjusted_fair_share: Sequence[float]) -> float:
    """SchedulingContext.FairnessError (context/scheduling.go:632-642; the cycle metric of metrics/cycle_metrics.go:609): the sum over the queues of how far
    the actual share (UnweightedCostFromAllocation of the queue's allocation) lies BELOW its demand-capped adjusted fair share."""
    err = 0.0
    for a, share in zip(allocated, demand_capped_adjusted_fair_share):
        delta = float(share) - sched.drf_cost(list(a), list(total))
        if delta > 0:
            err += delta
    return err
