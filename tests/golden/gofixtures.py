"""Python re-statement of the *data* produced by the reference's test fixtures
(internal/scheduler/testfixtures/testfixtures.go), used by extract_reference_goldens.py to evaluate the
reference's table-driven test cases into JSON.  Only values are reproduced (requests, taints, labels,
configs, creation order); nothing here is scheduling logic.

Quantities are converted to the TestResourceListFactory units (testfixtures.go:1222-1228):
memory in bytes, cpu and nvidia.com/gpu in milli-units.
"""
from __future__ import annotations

import copy
import itertools
import math
import re

from goparse import Struct, Unsupported

_counter = itertools.count(1)      # jobTimestamp (testfixtures.go:691-693): creation order == submit order
_node_counter = itertools.count(1)  # node factory index (node_factory.go:205-207)
_gang_counter = itertools.count(1)

SUFFIX = {"Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40, "k": 10**3, "M": 10**6, "G": 10**9, "T": 10**12}


def quantity(s) -> float:
    """k8s resource.Quantity -> float value in base units"""
    if isinstance(s, (int, float)):
        return float(s)
    m = re.fullmatch(r"([0-9.]+)(m|Ki|Mi|Gi|Ti|k|M|G|T)?", s)
    if not m:
        raise Unsupported(f"quantity {s!r}")
    v = float(m.group(1))
    suf = m.group(2)
    if suf == "m":
        return v / 1000.0
    if suf:
        return v * SUFFIX[suf]
    return v


SCALE = {"memory": 1, "cpu": 1000, "nvidia.com/gpu": 1000, "test-floating-resource": 1000}


def rl(d, round_up=False):
    out = {}
    for k, v in d.items():
        if k not in SCALE:
            raise Unsupported(f"resource {k}")
        x = quantity(v) * SCALE[k]
        out[k] = int(math.ceil(x - 1e-9)) if round_up else int(math.floor(x + 1e-9))
    return out


PriorityClass0 = "priority-0"
PriorityClass1 = "priority-1"
PriorityClass2 = "priority-2"
PriorityClass2NonPreemptible = "priority-2-non-preemptible"
PriorityClass3 = "priority-3"
PriorityClass4PreemptibleAway = "armada-preemptible-away"
PriorityClass5PreemptibleAwayLowPriority = "armada-preemptible-away-lower"
PriorityClass6Preemptible = "armada-preemptible"
PriorityClass7PreemptibleAwayConditional = "armada-preemptible-away-conditional"

TEST_PRIORITY_CLASSES = {  # testfixtures.go:78-105
    PriorityClass0: {"priority": 0, "preemptible": True},
    PriorityClass1: {"priority": 1, "preemptible": True},
    PriorityClass2: {"priority": 2, "preemptible": True},
    PriorityClass2NonPreemptible: {"priority": 2, "preemptible": False},
    PriorityClass3: {"priority": 3, "preemptible": False},
    PriorityClass4PreemptibleAway: {"priority": 30000, "preemptible": True, "away": [[29000, "gpu"], [29000, "large"]]},
    PriorityClass5PreemptibleAwayLowPriority: {"priority": 30000, "preemptible": True, "away": [[28000, "gpu"], [28000, "large"]]},
    PriorityClass6Preemptible: {"priority": 30000, "preemptible": True},
    # an away entry is [priority, WellKnownNodeTypeName, NodeTypes]; NodeTypes = [[name, [[resource, operator, c.Value.Value()], ...]], ...]
    PriorityClass7PreemptibleAwayConditional: {"priority": 30000, "preemptible": True,
                                               "away": [[29000, "large", [["gpu", [["nvidia.com/gpu", "==", 0]]]]]]},
}
TestPriorities = [0, 1, 2, 3, 28000, 29000, 30000]


class GoConfig(dict):
    """configuration.SchedulingConfig as the fixtures model it, plus the Go field assignments the test tables make in func literals"""

    def __setitem__(self, k, v):
        if k == "DefaultPriorityClassName":
            dict.__setitem__(self, "default_priority_class", v)
        elif k == "WellKnownNodeTypes":
            dict.__setitem__(self, "well_known_node_types",
                             {t["Name"]: [[x["Key"], x.get("Value", ""), x.get("Effect", "")] for x in t.get("Taints", [])] for t in v})
        elif k == "PriorityClasses":  # map[string]types.PriorityClass{name: {Priority, Preemptible, AwayNodeTypes: [{Priority, WellKnownNodeTypeName}]}}
            out = {}
            for name, pc in v.items():
                extra = set(pc) - {"Priority", "Preemptible", "AwayNodeTypes"}
                if extra:
                    raise Unsupported(f"PriorityClass fields {sorted(extra)}")
                e = {"priority": int(pc.get("Priority", 0)), "preemptible": bool(pc.get("Preemptible", False))}
                away = []
                for a in pc.get("AwayNodeTypes") or []:
                    if set(a) - {"Priority", "WellKnownNodeTypeName", "NodeTypes"}:
                        raise Unsupported(f"AwayNodeType fields {sorted(a)}")
                    ent = [int(a["Priority"]), a.get("WellKnownNodeTypeName", "")]
                    if a.get("NodeTypes"):
                        ent.append([[t["Name"], [[c["Resource"], c["Operator"], int(math.ceil(quantity(c["Value"]) - 1e-9))] for c in t.get("Conditions") or []]]
                                    for t in a["NodeTypes"]])
                    away.append(ent)
                if away:
                    e["away"] = away
                out[name] = e
            dict.__setitem__(self, "priority_classes", out)
        elif k == "ProtectedFractionOfFairShare":
            dict.__setitem__(self, "protected_fraction_of_fair_share", float(v))
        elif k == "MaximumPerQueueSchedulingBurst":
            dict.__setitem__(self, "maximum_per_queue_scheduling_burst", int(v))
        elif k == "MaximumPerQueueSchedulingRate":
            dict.__setitem__(self, "maximum_per_queue_scheduling_rate", float(v))
        elif k == "MaximumSchedulingBurst":
            dict.__setitem__(self, "maximum_scheduling_burst", int(v))
        elif k == "MaximumSchedulingRate":
            dict.__setitem__(self, "maximum_scheduling_rate", float(v))
        elif k == "FloatingResources":   # []configuration.FloatingResourceConfig{{Name, Pools: [{Name, Quantity}]}} -> {resource: {pool: factory units}}
            dict.__setitem__(self, "floating_resources",
                             {f["Name"]: {p["Name"]: int(round(quantity(p["Quantity"]) * SCALE[f["Name"]])) for p in f.get("Pools", [])} for f in v})
        elif k[:1].isupper():
            raise Unsupported(f"assignment to SchedulingConfig.{k}")
        else:
            dict.__setitem__(self, k, v)


def _norm_toleration(t):
    if isinstance(t, dict) and "Key" in t:  # v1.Toleration{Key, Operator, Value, Effect}
        return {"key": t.get("Key", ""), "op": t.get("Operator") or "Equal", "value": t.get("Value", ""), "effect": t.get("Effect", "")}
    return t


class PodReqs(dict):
    """PodRequirements as the fixtures model them; the Go field names the tables read / assign map onto the fixture keys"""
    ALIAS = {"Tolerations": "tolerations", "NodeSelector": "selector"}

    def __contains__(self, k):
        return dict.__contains__(self, self.ALIAS.get(k, k))

    def __getitem__(self, k):
        return dict.__getitem__(self, self.ALIAS.get(k, k))

    def __setitem__(self, k, v):
        k2 = self.ALIAS.get(k, k)
        if k2 != k and k2 == "tolerations":
            v = [_norm_toleration(t) for t in v]
        elif k2 == k and k[:1].isupper():
            raise Unsupported(f"assignment to PodRequirements.{k}")
        dict.__setitem__(self, k2, v)


TestFloatingResourceConfig = [{"Name": "test-floating-resource", "Pools": [{"Name": "pool", "Quantity": "10"}]}]   # testfixtures.go:64-74


def TestSchedulingConfig():  # testfixtures.go:225-249
    return GoConfig({
        "priority_classes": copy.deepcopy(TEST_PRIORITY_CLASSES),
        "maximum_scheduling_rate": math.inf, "maximum_scheduling_burst": 2**62,
        "maximum_per_queue_scheduling_rate": math.inf, "maximum_per_queue_scheduling_burst": 2**62,
        "indexed_resources": [["cpu", 1000], ["memory", 128 * 2**20], ["nvidia.com/gpu", 1000]],
        "indexed_node_labels": ["largeJobsOnly", "gpu", "cluster", "pool", "nodetype"],
        "indexed_taints": ["largeJobsOnly", "gpu"],
        "well_known_node_types": {"gpu": [["gpu", "true", "NoSchedule"]], "large": [["largeJobsOnly", "true", "NoSchedule"]]},
        "prefer_large_job_ordering": True,
        "drf_resources": ["cpu", "memory", "nvidia.com/gpu"],
        "protected_fraction_of_fair_share": 0.0,
        "max_queue_lookback": 0,
        "maximum_resource_fraction_to_schedule": {},
        "disable_home": False, "disable_away": False, "disable_gang_away": False, "disable_fairshare": False, "disable_urgency": False,
        "disallowed_resources": [],
    })


def _cfg(fn):
    def w(*a):
        *args, config = a
        config = copy.deepcopy(config)
        fn(config, *args)
        return config
    return w


@_cfg
def WithProtectedFractionOfFairShareConfig(c, v): c["protected_fraction_of_fair_share"] = float(v)
@_cfg
def WithRoundLimitsConfig(c, limits): c["maximum_resource_fraction_to_schedule"] = dict(limits)
@_cfg
def WithGlobalSchedulingRateLimiterConfig(c, rate, burst): c["maximum_scheduling_rate"] = float(rate); c["maximum_scheduling_burst"] = int(burst)
@_cfg
def WithPerQueueSchedulingLimiterConfig(c, rate, burst): c["maximum_per_queue_scheduling_rate"] = float(rate); c["maximum_per_queue_scheduling_burst"] = int(burst)
@_cfg
def WithMaxLookbackPerQueueConfig(c, n): c["max_queue_lookback"] = int(n)
@_cfg
def WithMaxQueueLookbackConfig(c, n): c["max_queue_lookback"] = int(n)
@_cfg
def WithIndexedTaintsConfig(c, t): c["indexed_taints"] = c["indexed_taints"] + list(t)
@_cfg
def WithIndexedNodeLabelsConfig(c, l): c["indexed_node_labels"] = c["indexed_node_labels"] + list(l)


def WithIndexedResourcesConfig(res, config):
    config = copy.deepcopy(config)
    out = []
    for r in res:
        name = r["Name"]
        out.append([name, int(round(quantity(r["Resolution"]) * SCALE[name]))])
    config["indexed_resources"] = out
    return config


def WithPerPriorityLimitsConfig(limits, config):
    config = copy.deepcopy(config)
    for pc, lim in limits.items():
        base = config["priority_classes"][pc]
        config["priority_classes"][pc] = {"priority": base["priority"], "preemptible": base["preemptible"], "max_fraction_per_queue": dict(lim)}
    return config


def WithPreemptionDisabled(df, du, config):
    config = copy.deepcopy(config); config["disable_fairshare"] = bool(df); config["disable_urgency"] = bool(du); return config


def MustParse(s):
    return s


def ResourceType(**kw):
    return kw


# ------------------------------------------------------------------ pod requirements / jobs
def _podreqs(requests, tolerations=None):
    return PodReqs({"req": rl(requests, round_up=True), "tolerations": tolerations or [], "selector": {}, "affinity": None})


def Test1Cpu4GiPodReqs(): return _podreqs({"cpu": "1", "memory": "4Gi"})
def Test1Cpu16GiPodReqs(): return _podreqs({"cpu": "1", "memory": "16Gi"})
def Test16Cpu128GiPodReqs(): return _podreqs({"cpu": "16", "memory": "128Gi"})
def Test32Cpu256GiPodReqs(): return _podreqs({"cpu": "32", "memory": "256Gi"})
def Test64Cpu512GiPodReqs(): return _podreqs({"cpu": "64", "memory": "512Gi"})
_LARGE_TOL = [{"key": "largeJobsOnly", "op": "Equal", "value": "true", "effect": ""}]
def Test32Cpu256GiWithLargeJobTolerationPodReqs(): return _podreqs({"cpu": "32", "memory": "256Gi"}, copy.deepcopy(_LARGE_TOL))
def Test16Cpu128GiPodReqsWithLargeJobToleration(): return _podreqs({"cpu": "16", "memory": "128Gi"}, copy.deepcopy(_LARGE_TOL))
def Test1GpuPodReqs(): return _podreqs({"cpu": "8", "memory": "128Gi", "nvidia.com/gpu": "1"}, [{"key": "gpu", "op": "Equal", "value": "true", "effect": ""}])


def TestJob(queue, job_id, pc, req):  # testfixtures.go:687-716
    created = next(_counter)
    return {"created": created, "queue": queue, "pc": pc, "priority": 1000, "gang": None, **copy.deepcopy(req)}


def ULID():
    return "ulid"


def _n(fn):
    return lambda queue, pc, n: [fn(queue, pc) for _ in range(int(n))]


def Test1Cpu4GiJob(q, pc): return TestJob(q, None, pc, Test1Cpu4GiPodReqs())
def Test1Cpu16GiJob(q, pc): return TestJob(q, None, pc, Test1Cpu16GiPodReqs())
def Test16Cpu128GiJob(q, pc): return TestJob(q, None, pc, Test16Cpu128GiPodReqs())
def Test32Cpu256GiJob(q, pc): return TestJob(q, None, pc, Test32Cpu256GiPodReqs())
def Test64Cpu512GiJob(q, pc): return TestJob(q, None, pc, Test64Cpu512GiPodReqs())
def Test32Cpu256GiJobWithLargeJobToleration(q, pc): return TestJob(q, None, pc, Test32Cpu256GiWithLargeJobTolerationPodReqs())
def Test16Cpu128GiJobWithLargeJobToleration(q, pc): return TestJob(q, None, pc, Test16Cpu128GiPodReqsWithLargeJobToleration())
def Test1GpuJob(q, pc): return TestJob(q, None, pc, Test1GpuPodReqs())


N1Cpu4GiJobs = _n(Test1Cpu4GiJob)
N1Cpu16GiJobs = _n(Test1Cpu16GiJob)
N16Cpu128GiJobs = _n(Test16Cpu128GiJob)
N32Cpu256GiJobs = _n(Test32Cpu256GiJob)
N64Cpu512GiJobs = _n(Test64Cpu512GiJob)
N32Cpu256GiJobsWithLargeJobToleration = _n(Test32Cpu256GiJobWithLargeJobToleration)
N1GpuJobs = _n(Test1GpuJob)


def WithGangJobDetails(jobs, gang_id, card, uniformity):  # testfixtures.go:559-571
    for j in jobs:
        j["gang"] = {"id": gang_id, "cardinality": int(card), "uniformity": uniformity or ""}
    return jobs


def WithNodeUniformityGangAnnotationsJobs(jobs, label):
    return WithGangJobDetails(jobs, f"gang-{next(_gang_counter)}", len(jobs), label)


def WithGangAnnotationsJobs(jobs):
    return WithNodeUniformityGangAnnotationsJobs(jobs, "")


def WithPriorityJobs(priority, jobs):
    for j in jobs:
        j["priority"] = int(priority)
    return jobs


def WithNodeSelectorJobs(selector, jobs):
    for j in jobs:
        j["selector"] = dict(selector)
    return jobs


ALLOW_FLOATING_REQUESTS = False   # the submit check handles floating resources on the host (submitcheck.go:326-333): its extraction turns this on


def WithRequestsJobs(rlist, jobs):  # testfixtures.go:496-511; the job's vector comes from FromJobResourceListIgnoreUnknown: unknown names are dropped
    if isinstance(rlist, dict) and "Resources" in rlist:   # schedulerobjects.ResourceList{Resources: ...}
        rlist = rlist["Resources"]
    if "test-floating-resource" in rlist and not ALLOW_FLOATING_REQUESTS:
        raise Unsupported("floating resources are not modelled")
    known = {k: v for k, v in rlist.items() if k in SCALE}
    out = []
    for j in jobs:                      # job.WithJobSchedulingInfo returns a new job: the caller's job keeps its requests
        j = copy.copy(j)
        j["req"] = dict(j["req"])
        j["req"].update(rl(known, round_up=True))
        out.append(j)
    return out


def WithNodeAffinityJobs(terms, jobs):  # testfixtures.go:476-494: appends to RequiredDuringSchedulingIgnoredDuringExecution.NodeSelectorTerms
    for j in jobs:
        cur = list(j.get("affinity") or [])
        for t in terms:
            if t.get("MatchFields"):
                raise Unsupported("node affinity MatchFields")
            cur.append([[e["Key"], e["Operator"], list(e.get("Values") or [])] for e in t.get("MatchExpressions") or []])
        j["affinity"] = cur   # list of terms; a term = list of [key, operator, values]
    return jobs


def WithAnnotationsJobs(annotations, jobs):
    return jobs


# ------------------------------------------------------------------ nodes
def TestNode(priorities, resources):  # testfixtures.go:985-1003
    idx = next(_node_counter)
    return {"index": idx, "total": rl(resources), "taints": [], "labels": {}, "used": {}, "unschedulable": False}


def Test32CpuNode(p): return TestNode(p, {"cpu": "32", "memory": "256Gi"})
def Test16CpuNode(p): return TestNode(p, {"cpu": "16", "memory": "128Gi"})


def TestTainted32CpuNode(p):
    n = Test32CpuNode(p)
    n["taints"].append(["largeJobsOnly", "true", "NoSchedule"])
    n["labels"]["largeJobsOnly"] = "true"
    return n


def Test8GpuNode(p):
    n = TestNode(p, {"cpu": "64", "memory": "1024Gi", "nvidia.com/gpu": "8"})
    n["labels"]["gpu"] = "true"
    return n


def N32CpuNodes(n, p): return [Test32CpuNode(p) for _ in range(int(n))]
def NTainted32CpuNodes(n, p): return [TestTainted32CpuNode(p) for _ in range(int(n))]
def N8GpuNodes(n, p): return [Test8GpuNode(p) for _ in range(int(n))]


def AddLabels(nodes, labels):
    for n in nodes:
        n["labels"].update(labels)
    return nodes


def WithGpuTaint(node): return AddTaints([node], [{"Key": "gpu", "Value": "true", "Effect": "NoSchedule"}])[0]                 # testfixtures.go:1059-1068
def WithLargeJobsOnlyTaint(node): return AddTaints([node], [{"Key": "largeJobsOnly", "Value": "true", "Effect": "NoSchedule"}])[0]  # :1070-1079
def TestPodReqs(requests): return _podreqs(requests)                                                                              # :810-816


def AddTaints(nodes, taints):
    for n in nodes:
        for t in taints:
            n["taints"].append([t["Key"], t.get("Value", ""), t.get("Effect", "")])
    return nodes


def WithUsedResourcesNodes(p, used, nodes):  # MarkAllocated(node.AllocatableByPriority, p, rl): every level <= p
    for n in nodes:
        cur = n["used"].setdefault(str(int(p)), {})
        for k, v in used.items():
            cur[k] = cur.get(k, 0) + v
    return nodes


def Cpu(c): return CpuMemGpu(c, "0", "0")
def CpuMem(c, m): return CpuMemGpu(c, m, "0")
def CpuMemGpu(c, m, g): return rl({"cpu": c, "memory": m, "nvidia.com/gpu": g})


def IntRange(a, b): return list(range(int(a), int(b) + 1))
def Repeat(v, n): return [copy.deepcopy(v) for _ in range(int(n))]


def Concatenate(*lists):
    out = []
    for l in lists:
        out.extend(l)
    return out


def SingleQueuePriorityOne(name): return [Struct("api.Queue", {"Name": name, "PriorityFactor": 1.0})]


def createOptimiserConfig(maximum_jobs_per_round, maximum_fraction_to_schedule, minimum_job_size_to_schedule):   # optimising_queue_scheduler_test.go:334-346 (same package as the PQS test)
    return {"Enabled": True, "MaximumJobsPerRound": int(maximum_jobs_per_round), "MaximumResourceFractionToSchedule": dict(maximum_fraction_to_schedule or {}),
            "MinimumJobSizeToSchedule": minimum_job_size_to_schedule, "MaximumJobSizeToPreempt": None, "MinimumFairnessImprovementPercentage": 0.0}


def WithOptimiserConfig(pool, optimiser_config, config):   # testfixtures.go:328-336: PoolConfig.ExperimentalOptimiser of the named pool
    config = copy.deepcopy(config)
    if pool == "testPool":   # (TestSchedulingConfig declares that pool; the PQS driver's scheduling context is for it)
        config["optimiser"] = optimiser_config
    return config


def WithMarketBasedSchedulingEnabled(config):   # testfixtures.go:297-306: PoolConfig.ExperimentalMarketScheduling of every pool
    config = copy.deepcopy(config)
    config["market"] = {"Enabled": True, "SpotPriceCutoff": 0.9}
    return config


def N1Cpu4GiJobsWithPriceBandAndPriorityClass(queue, band, pc, n):   # testfixtures.go:601-610 + SetPricing :573-585: QueuedBid = the band's number, RunningBid the same
    return [dict(Test1Cpu4GiJob(queue, pc), price_band=int(band)) for _ in range(int(n))]     # (pricing.NonPreemptibleRunningPrice for a non-preemptible class: the driver resolves it)


def N1Cpu4GiJobsWithPriceBand(queue, band, n): return N1Cpu4GiJobsWithPriceBandAndPriorityClass(queue, band, "priority-0", n)   # :597-599


def N1GpuJobsWithPriceBandAndPriorityClass(queue, band, pc, n):   # :612-621
    return [dict(Test1GpuJob(queue, pc), price_band=int(band)) for _ in range(int(n))]


def WithRoundLimitsPoolConfig(limits, config):
    config = copy.deepcopy(config)
    config["maximum_resource_fraction_to_schedule_by_pool"] = {k: dict(v) for k, v in limits.items()}
    return config


def go_append(base, *elems):
    return list(base or []) + list(elems)


def make_env():
    g = globals()
    env = {}
    for name in list(g):
        if name[:1].isupper() and name not in ("SUFFIX", "SCALE", "Struct", "Unsupported"):
            env["testfixtures." + name] = g[name]
    env["testfixtures.TestNodeFactory.AddLabels"] = AddLabels
    env["testfixtures.TestNodeFactory.AddTaints"] = AddTaints
    env["testfixtures.TestPool"] = "testPool"
    env["testfixtures.TestQueue"] = "testQueue"
    env["createOptimiserConfig"] = createOptimiserConfig
    env["append"] = go_append
    c = TestSchedulingConfig()
    c["prefer_large_job_ordering"] = False
    env["schedulingConfigWithPreferLargeJobDisabled"] = c
    env["armadaconfiguration.GangIdAnnotation"] = "armadaproject.io/gangId"
    env["armadaconfiguration.GangCardinalityAnnotation"] = "armadaproject.io/gangCardinality"
    env["armadaslices.Concatenate"] = Concatenate
    env["util.ULID"] = ULID
    env["resource.MustParse"] = MustParse
    env["pointer.MustParseResource"] = MustParse
    for op in ("In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt"):
        env["v1.NodeSelectorOp" + op] = op
    env["v1.TaintEffectNoSchedule"] = "NoSchedule"
    env["v1.TaintEffectNoExecute"] = "NoExecute"
    env["v1.TaintEffectPreferNoSchedule"] = "PreferNoSchedule"
    for i, b in enumerate("ABCDEFGH"):
        env["bidstore.PriceBand_PRICE_BAND_" + b] = i + 1   # pkg/bidstore/bids.pb.go:72-81
    env["testfixtures.TestResourceListFactory.MakeAllZero"] = lambda: rl({})   # an all-zero list of the factory's resources
    env["floatPtr"] = float                                  # market_driven_preempting_queue_scheduler_test.go:964-966
    env["makeIntArray"] = lambda n: list(range(int(n)))      # :956-962
    env["math.MaxInt"] = 2**62
    env["math.Inf"] = lambda s: math.inf if s >= 0 else -math.inf
    return env
