#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own table-driven tests.

Runs ONLY in the build container (it reads /root/reference, which does not exist on the GPU box);
the JSON it writes is committed, and both the CPU-oracle tests and the `-m gpu` parity tests consume
the JSON.  Every emitted case carries the reference file:line it was transcribed from.

    python tests/golden/extract_reference_goldens.py

Cases that use Go constructs outside goparse's subset (closures, node affinity, floating resources,
market pricing, cross-pool "away" jobs) are skipped and listed in tests/golden/SKIPPED.txt.
"""
import json
import math
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gofixtures  # noqa: E402
from goparse import Evaluator, Parser, Struct, Unsupported, find_matching, split_table_cases, tokenize  # noqa: E402

REF = "/root/reference/internal/scheduler"


def to_json(v):
    if isinstance(v, Struct):
        d = {k: to_json(x) for k, x in v.items()}
        d["__type__"] = v.type
        return d
    if isinstance(v, dict):
        return {str(k): to_json(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [to_json(x) for x in v]
    if isinstance(v, float) and math.isinf(v):
        return "inf" if v > 0 else "-inf"
    return v


def extract_table(path, func_name, env, skipped, table_var="tests"):
    src = open(path).read()
    fpos = src.index(f"func {func_name}(")
    m = re.compile(table_var + r"\s*:=\s*map\[string\]struct\s*\{").search(src, fpos)
    struct_open = m.end() - 1
    struct_close = find_matching(src, struct_open)
    body_open = src.index("{", struct_close + 1)
    body_close = find_matching(src, body_open)

    def line_of(pos):
        return src.count("\n", 0, pos) + 1

    cases = split_table_cases(src, body_open + 1, body_close, line_of)
    out = []
    rel = os.path.relpath(path, "/root/reference")
    ev = Evaluator(env)
    for name, lit, line in cases:
        try:
            p = Parser(tokenize(lit, line))
            ast = p.parse_expr()
            # evaluate with an anonymous struct type
            val = ev.comp(("named", "__case__"), ast[2])
            out.append({"name": name, "source": f"{rel}:{line}", **to_json(val)})
        except Unsupported as e:
            skipped.append(f"{rel}:{line} {func_name}/{name}: {e}")
        except RecursionError:
            skipped.append(f"{rel}:{line} {func_name}/{name}: recursion")
    return out


def extract_submitcheck(base_env, skipped):
    """submitcheck_test.go:28-456 TestSubmitChecker_CheckJobDbJobs.  The table refers to jobs and nodes built by the statements and
    helper functions around it (:32-58 jobs and pools, :516-544 Executor / GpuNode, :553-573 cordon / SmallNode); those are restated
    here with the same fixtures, the table itself is evaluated mechanically.  Job ids are creation-order labels (the reference's are
    random ULIDs and only key the expectation map)."""
    import copy
    import itertools
    G = gofixtures
    env = dict(base_env)
    ids = itertools.count(1)

    def with_id(j):
        jid = f"job-{next(ids):04d}"
        j["id"] = jid
        j["Id"] = (lambda s: (lambda: s))(jid)  # job.Id()
        return j

    env["smallJob1"] = with_id(G.Test1Cpu4GiJob("queue", G.PriorityClass1))                                   # :32
    env["smallJob2"] = with_id(G.Test1Cpu4GiJob("queue", G.PriorityClass1))                                   # :33
    env["smallGpuJob"] = with_id(G.Test1GpuJob("queue", G.PriorityClass4PreemptibleAway))                     # :34
    env["smallAwayJob"] = with_id(G.Test1Cpu4GiJob("queue", G.PriorityClass4PreemptibleAway))                 # :35
    env["largeJob1"] = with_id(G.Test32Cpu256GiJobWithLargeJobToleration("queue", G.PriorityClass1))          # :36
    env["smallGangJob"] = [with_id(j) for j in G.WithGangAnnotationsJobs(G.N1Cpu4GiJobs("queue", G.PriorityClass1, 2))]  # :39-40
    env["largeGangJob"] = [with_id(j) for j in G.WithGangAnnotationsJobs(G.N1Cpu4GiJobs("queue", G.PriorityClass1, 4))]  # :43-44
    env["batchJobs"] = [with_id(j) for j in G.N1Cpu4GiJobs("queue", G.PriorityClass1, 20)]                    # :46

    def small_node(pool):  # :563-573
        n = G.TestNode(None, {"cpu": "2", "memory": "64Gi"})
        n["pool"] = pool
        return n

    def gpu_node(pool):    # :527-544
        n = G.TestNode(None, {"cpu": "30", "memory": "512Gi", "nvidia.com/gpu": "8"})
        n["taints"] = [["gpu", "true", "NoSchedule"]]
        n["pool"] = pool
        return n

    def cordon(n):         # :553-561 (node.Unschedulable + the taint kubectl adds)
        n["unschedulable"] = True
        n["taints"].append(["node.kubernetes.io/unschedulable", "", "NoSchedule"])
        return n

    env.update({"SmallNode": small_node, "GpuNode": gpu_node, "cordon": cordon, "Executor": lambda *nodes: list(nodes),
                "defaultTimeout": 900, "time.Second": 1, "pointer.MustParseResource": G.MustParse,
                "testfixtures.WithNodeSelectorJob": lambda sel, j: G.WithNodeSelectorJobs(sel, [copy.copy(j)])[0]})
    G.ALLOW_FLOATING_REQUESTS = True     # floating requests stay in the job's vector; the harness splits them off (they never reach a NodeDb)
    try:
        cases = extract_table(f"{REF}/submitcheck_test.go", "TestSubmitChecker_CheckJobDbJobs", env, skipped)
    finally:
        G.ALLOW_FLOATING_REQUESTS = False
    pools = [  # schedulingConfig.Pools, :49-58
        {"name": "cpu"}, {"name": "cpu2"}, {"name": "cpu-disallowed-resources", "disallowed_resources": ["cpu"]}, {"name": "gpu"},
        {"name": "cpu-away", "away_pools": ["gpu"]}, {"name": "cpu-grouped-1", "submission_group": "group-1"},
        {"name": "cpu-grouped-2", "submission_group": "group-1"},
    ]
    out = []
    for c in cases:
        c = to_json(c)
        for j in c["jobs"]:
            j.pop("Id", None)
        c["SchedulingConfig"] = to_json(G.TestSchedulingConfig())
        c["Pools"] = pools
        c["FloatingResources"] = {"test-floating-resource": {"cpu": 10 * G.SCALE["test-floating-resource"]}}   # submitcheck_test.go:410-421: 10 in pool "cpu"
        c["Queues"] = [c.pop("queue")] if c.get("queue") else [{"Name": "queue"}]   # :400-405
        out.append(c)
    return out


def extract_pq_ordering(skipped):
    """queue_scheduler_test.go:995-1164: six stand-alone tests that sort.Sort a QueueCandidateGangIteratorPQ and compare the item order.
    They are functions, not tables: each `x := &QueueCandidateGangIteratorItem{...}`, the `pq := &QueueCandidateGangIteratorPQ{...}`
    literal and the `expectedOrder` slice are evaluated mechanically from the function body.  (TestQueueCandidateGangIteratorPQ_HomeBeforeAway
    :698-716 needs cross-pool "away" items, which one pool never has: listed in SKIPPED.txt.)"""
    path = f"{REF}/scheduling/queue_scheduler_test.go"
    src = open(path).read()
    rel = os.path.relpath(path, "/root/reference")
    out = []

    def body_of(fn):
        pos = src.index(f"func {fn}(")
        o = src.index("{", pos)
        return o, find_matching(src, o)

    def ev_at(pos, env):  # evaluate the expression starting at src[pos]
        line = src.count("\n", 0, pos) + 1
        close = find_matching(src, src.index("{", pos))
        p = Parser(tokenize(src[pos:close + 1], line))
        return Evaluator(env).ev(p.parse_expr()), line

    def items_of(o, c):
        env = {}
        for m in re.finditer(r"(\w+)\s*:=\s*(&QueueCandidateGangIteratorItem\{)", src[o:c]):
            v, _ = ev_at(o + m.start(2), {})
            v = dict(v); v["_var"] = m.group(1)
            env[m.group(1)] = v
        return env

    def emit(name, line, pq, expected, env):
        item = lambda v: {"queue": v.get("queue", ""), "proposedQueueCost": float(v.get("proposedQueueCost", 0)), "currentQueueCost": float(v.get("currentQueueCost", 0)),
                          "queueBudget": float(v.get("queueBudget", 0)), "itemSize": float(v.get("itemSize", 0)),
                          "priorityClassPriority": int(v.get("priorityClassPriority", 0)), "schedulingPriority": int(v.get("schedulingPriority", 0)), "var": v["_var"]}
        out.append({"name": name, "source": f"{rel}:{line}", "prioritiseLargerJobs": bool(pq.get("prioritiseLargerJobs", False)),
                    "compareSchedulingPriority": bool(pq.get("compareSchedulingPriority", False)),
                    "items": [item(v) for v in pq["items"]], "expectedOrder": [v["_var"] for v in expected]})

    for fn in ("TestQueueCandidateGangIteratorPQ_Ordering_BelowFairShare_EvenCurrentCost", "TestQueueCandidateGangIteratorPQ_Ordering_BelowFairShare_UnevenCurrentCost",
               "TestQueueCandidateGangIteratorPQ_Ordering_AboveFairShare", "TestQueueCandidateGangIteratorPQ_Ordering_MixedFairShare",
               "TestQueueCandidateGangIteratorPQ_Fallback"):
        o, c = body_of(fn)
        env = items_of(o, c)
        m = re.search(r"pq\s*:=\s*(&QueueCandidateGangIteratorPQ\{)", src[o:c])
        pq, line = ev_at(o + m.start(1), env)
        m = re.search(r"expectedOrder\s*:=\s*(\[\]\*QueueCandidateGangIteratorItem\{)", src[o:c])
        exp, _ = ev_at(o + m.start(1), env)
        emit(fn, src.count("\n", 0, o) + 1, pq, exp, env)
    fn = "TestQueueCandidateGangIteratorPQ_Ordering_SchedulingPriority"   # a table over two shared items; the pq literal sits in the loop body (:1150-1154)
    o, c = body_of(fn)
    env = items_of(o, c)
    for r in extract_table(path, fn, env, skipped):
        emit(f"{fn}/{r['name']}", int(r["source"].rsplit(":", 1)[1]),
             {"prioritiseLargerJobs": True, "compareSchedulingPriority": bool(r["shouldCompareSchedulingPriority"]), "items": [env["queueA"], env["queueB"]]},
             r["expectedOrder"], env)
    skipped.append(f"{rel}:698 TestQueueCandidateGangIteratorPQ_HomeBeforeAway: two hand-built items, no table: transcribed by hand in tests/test_zzz_pq_ordering.py (asched_pq_order, ASCHED_PQ_HOME_FIRST)")
    return out


def extract_comparer(skipped, func="TestJobPriorityComparer"):
    path = f"{REF}/jobdb/comparison_test.go"
    src = open(path).read()
    rel = os.path.relpath(path, "/root/reference")
    pos = src.index(f"func {func}(")
    BIDS = {"bidPricesA": {"a": 1.0, "b": 2.0}, "bidPricesB": {"a": 2.0, "b": 1.0}}   # comparison_test.go:77-78 (QueuedBid == RunningBid)
    m = re.compile(r"tests\s*:=\s*map\[string\]struct\s*\{").search(src, pos)
    body_open = src.index("{", find_matching(src, m.end() - 1) + 1)
    body_close = find_matching(src, body_open)
    line_of = lambda p: src.count("\n", 0, p) + 1  # noqa: E731

    def job_of(text):
        """&Job{...} optionally wrapped as (&Job{...}).WithNewRun(...) / .WithUpdatedRun(&JobRun{created: N})"""
        lit = text[text.index("&Job{") + 1:]
        lit = lit[:find_matching(lit, lit.index("{")) + 1]
        k = lit.find("jobDb:")                  # `jobDb: NewJobDb(...)` only gives WithNewRun something to read the default priority class from
        if k >= 0:
            o = lit.index("(", k)
            lit = lit[:k] + lit[find_matching(lit, o, "(", ")") + 1:].lstrip(" ,")
        v = Evaluator({"types.PriorityClass": None, "bidPricesA": "bidPricesA", "bidPricesB": "bidPricesB"}).ev(Parser(tokenize(lit, 1)).parse_expr())
        j = {"id": v.get("id", ""), "priority": int(v.get("priority", 0)), "pcPriority": int((v.get("priorityClass") or {}).get("Priority", 0)),
             "submittedTime": int(v.get("submittedTime", 0)), "activeRunTimestamp": int(v.get("activeRunTimestamp", 0)), "active": False}
        if func != "TestJobPriorityComparer":
            j["bidPrices"] = BIDS.get(v.get("bidPricesPool"), {})   # pool -> bid; no map = no bid (GetBidPrice returns 0)
            j["queued"] = bool(v.get("queued", False))
        if ".WithNewRun(" in text:
            j["active"] = True
        mm = re.search(r"\.WithUpdatedRun\(\s*&JobRun\{created:\s*(\d+)\}", text)
        if mm:
            j["active"], j["activeRunTimestamp"] = True, int(mm.group(1))
        return j

    out = []
    for name, lit, line in split_table_cases(src, body_open + 1, body_close, line_of):
        try:
            ma, mb, me = re.search(r"\ba:\s", lit), re.search(r"\n\s*b:\s", lit), re.search(r"\n\s*expected:\s*(-?\d+)", lit)
            a, b = job_of(lit[ma.end():mb.start()]), job_of(lit[mb.end():me.start()])
            if a["id"] == b["id"] and func == "TestJobPriorityComparer":
                continue   # "Jobs with equal id are considered equal": identity, not order
            case = {"name": name, "source": f"{rel}:{line}", "a": a, "b": b, "expected": int(me.group(1))}
            mp = re.search(r'\n\s*currentPool:\s*"([^"]*)"', lit)
            if mp:
                case["currentPool"] = mp.group(1)
            out.append(case)
        except (Unsupported, AttributeError, ValueError) as e:
            skipped.append(f"{rel}:{line} {func}/{name}: {e}")
    return out


def main():
    env = gofixtures.make_env()
    skipped = []
    out = {}
    out["pqs"] = extract_table(f"{REF}/scheduling/preempting_queue_scheduler_test.go", "TestPreemptingQueueScheduler", env, skipped)
    out["queue_scheduler"] = extract_table(f"{REF}/scheduling/queue_scheduler_test.go", "TestQueueScheduler", env, skipped)
    # gang_scheduler_test.go:943-951 addFloatingResourceRequest: the floating request stays in the job's vector (AllResourceRequirements); the
    # library keeps it away from the nodes.  createAwayJob (:821-823): Test1Cpu4GiJob(TestQueue, PriorityClass2).WithNewRun(..., pool "away", priority 1) —
    # a job whose latest run belongs to another pool (context.IsHomeJob false): the job dict carries "away": true
    genv = dict(env)
    genv["createAwayJob"] = lambda: dict(gofixtures.Test1Cpu4GiJob(env["testfixtures.TestQueue"], env["testfixtures.PriorityClass2"]), away=True)
    genv["addFloatingResourceRequest"] = lambda req, jobs: gofixtures.WithRequestsJobs({"test-floating-resource": gofixtures.MustParse(req)}, jobs)
    gofixtures.ALLOW_FLOATING_REQUESTS = True
    try:
        out["gang_scheduler"] = extract_table(f"{REF}/scheduling/gang_scheduler_test.go", "TestGangScheduler", genv, skipped)
    finally:
        gofixtures.ALLOW_FLOATING_REQUESTS = False

    # float goldens: fairness_test.go:62-177 and context/scheduling_test.go:89-247
    fenv = dict(env)
    fenv["rlFactory"] = None
    fenv["fooBarBaz"] = lambda _f, a, b, c: {"foo": int(round(gofixtures.quantity(a) * 1000)), "bar": int(round(gofixtures.quantity(b) * 1000)),
                                             "baz": int(round(gofixtures.quantity(c) * 1000))}
    fenv["poolName"] = "pool"
    out["fairness"] = extract_table(f"{REF}/scheduling/fairness/fairness_test.go", "TestDominantResourceFairness", fenv, skipped)
    senv = dict(env)
    for nm, v in (("zeroCpu", 0), ("oneCpu", 1), ("fortyCpu", 40), ("oneHundredCpu", 100), ("oneThousandCpu", 1000)):
        senv[nm] = {"cpu": v * 1000}
    out["fair_shares"] = extract_table(f"{REF}/scheduling/context/scheduling_test.go", "TestCalculateFairShares", senv, skipped)

    # node iteration orderings: nodedb/nodeiteration_test.go:75-374 (one type) and :376-695 (merged types)
    nenv = dict(env)
    for nm, tid in (("nodeTypeA", 1), ("nodeTypeB", 2), ("nodeTypeC", 3), ("nodeTypeD", 4)):
        nenv[nm] = tid
        nenv[nm + ".GetId"] = (lambda t: (lambda: t))(tid)

    def with_node_type(tid, nodes):
        for n in nodes:
            n["node_type"] = tid
        return nodes
    nenv["testfixtures.WithNodeTypeNodes"] = with_node_type
    nenv["testfixtures.TestResourceListFactory.MakeAllZero"] = lambda: {}
    out["node_type_iterator"] = extract_table(f"{REF}/nodedb/nodeiteration_test.go", "TestNodeTypeIterator", nenv, skipped)
    out["node_types_iterator"] = extract_table(f"{REF}/nodedb/nodeiteration_test.go", "TestNodeTypesIterator", nenv, skipped)

    # NodeDb-level tables: nodedb/nodedb_test.go:502-667 (one job at a time) and :668-743 (one group at a time), both through
    # ScheduleManyWithTxn inside a transaction that is committed on success and aborted on failure; newNodeDbWithNodes (:1747) uses
    # the same priority classes / indexed resources / taints / labels / well-known node types as TestSchedulingConfig
    denv = dict(env)
    denv["gangSuccess"] = gofixtures.WithGangAnnotationsJobs(gofixtures.N1Cpu4GiJobs("A", gofixtures.PriorityClass0, 32))
    denv["gangFailure"] = gofixtures.WithGangAnnotationsJobs(gofixtures.N1Cpu4GiJobs("A", gofixtures.PriorityClass0, 33))
    for key, fn in (("nodedb_schedule_individually", "TestScheduleIndividually"), ("nodedb_schedule_many", "TestScheduleMany")):
        cases = extract_table(f"{REF}/nodedb/nodedb_test.go", fn, denv, skipped)
        for c in cases:
            c["SchedulingConfig"] = to_json(gofixtures.TestSchedulingConfig())
        out[key] = cases

    # nodedb_test.go:1293-1432 TestAwayNodeScheduling: the table holds parameters, the test body builds the NodeDb, node and job from
    # them; that body is restated here (same fixtures) so that the case carries config / nodes / jobs like the other NodeDb-level cases
    import copy
    away = []
    for r in extract_table(f"{REF}/nodedb/nodedb_test.go", "TestAwayNodeScheduling", env, skipped):
        cfg = gofixtures.TestSchedulingConfig()
        t = r["wellKnownNodeTypeTaint"]
        cfg["well_known_node_types"] = {"gpu": [[t["Key"], t["Value"], t["Effect"]]], "large": [["large", "true", "NoSchedule"]]}  # :1361-1364
        cfg["disable_away"] = bool(r.get("disableAwayScheduling")); cfg["disable_gang_away"] = bool(r.get("disableGangAwayScheduling"))  # :1368-1371
        node = gofixtures.AddTaints([gofixtures.Test32CpuNode([29000, 30000])], [r["nodeTaint"]])[0]      # :1374-1378
        job = gofixtures.TestJob("testQueue", None, "armada-preemptible-away", gofixtures.Test1Cpu4GiPodReqs())  # :1380-1385
        if r.get("shouldSubmitGang"):
            job = gofixtures.WithGangAnnotationsJobs([copy.deepcopy(job), copy.deepcopy(job)])[0]         # :1386-1388
        away.append({"name": r["name"], "source": r["source"], "SchedulingConfig": to_json(cfg), "Nodes": to_json([node]), "Jobs": to_json([job]),
                     "ExpectSuccess": [bool(r["expectSuccess"])], "ExpectAway": {"node": 0, "priority": 29000}})       # :1399-1427
    out["nodedb_away_node_scheduling"] = away

    out["submitcheck"] = extract_submitcheck(env, skipped)
    out["pq_ordering"] = extract_pq_ordering(skipped)

    # queue_scheduler_test.go:804-946 TestQueueScheduler_PreemptionRateLimit: one node, existing jobs of queues A / B (all evicted and
    # registered as fair-share preemption candidates by the test body), new jobs of queue B, sctx.FairsharePreemptionLimiter
    renv = dict(env)
    renv["rate.Limit"] = float
    renv["rate.NewLimiter"] = lambda r, b: {"rate": float(r), "burst": int(b)}
    out["preemption_rate_limit"] = [dict(to_json(r), SchedulingConfig=to_json(gofixtures.TestSchedulingConfig()))
                                    for r in extract_table(f"{REF}/scheduling/queue_scheduler_test.go", "TestQueueScheduler_PreemptionRateLimit", renv, skipped)]

    # constraints/constraints_test.go:452-575 TestCheckJobConstraints_RateLimit: token-bucket states (tokens at sctx.Started, burst) and
    # a gang cardinality -> the unschedulable reason of CheckJobConstraints
    kenv = dict(env)
    for nm, txt in (("GlobalRateLimitExceededUnschedulableReason", "global scheduling rate limit exceeded"),
                    ("QueueRateLimitExceededUnschedulableReason", "queue scheduling rate limit exceeded"),
                    ("GlobalRateLimitExceededByGangUnschedulableReason", "gang would exceed global scheduling rate limit"),
                    ("QueueRateLimitExceededByGangUnschedulableReason", "gang would exceed queue scheduling rate limit"),
                    ("GangExceedsGlobalBurstSizeUnschedulableReason", "gang cardinality too large: exceeds global max burst size"),
                    ("GangExceedsQueueBurstSizeUnschedulableReason", "gang cardinality too large: exceeds queue max burst size")):
        kenv[nm] = txt   # constraints.go:34-49
    out["constraints_rate_limit"] = extract_table(f"{REF}/scheduling/constraints/constraints_test.go", "TestCheckJobConstraints_RateLimit", kenv, skipped)

    # nodedb/nodematching_test.go:19-410 TestNodeSchedulingRequirementsMet: JobRequirementsMet(node, priority, jctx) = static requirements
    # (taints / tolerations, node selector, required node affinity, total resources) + resources at one priority, on hand-built nodes
    # (makeTestNodeTaintsLabels :713-737, makeTestNodeResources :739-763)
    jenv = dict(env)
    jenv["makeTestNodeTaintsLabels"] = lambda taints, labels: {"taints": [[x["Key"], x.get("Value", ""), x.get("Effect", "")] for x in (taints or [])],
                                                                 "labels": dict(labels or {}), "total": {}, "alloc_by_priority": None}
    jenv["makeTestNodeResources"] = lambda t, abp, total: {"taints": [], "labels": {}, "total": total, "alloc_by_priority": {str(k): v for k, v in abp.items()}}
    jenv["t"] = None
    jenv["rlFactory.FromJobResourceListIgnoreUnknown"] = lambda m: gofixtures.rl(m)
    met = []
    for r in extract_table(f"{REF}/nodedb/nodematching_test.go", "TestNodeSchedulingRequirementsMet", jenv, skipped):
        req = r.get("req") or {}
        job = gofixtures.TestJob("A", None, gofixtures.PriorityClass0, gofixtures._podreqs((req.get("ResourceRequirements") or {}).get("Requests") or {}))
        job["tolerations"] = [gofixtures._norm_toleration(x) for x in req.get("Tolerations") or []]
        job["selector"] = dict(req.get("NodeSelector") or {})
        sel = ((req.get("Affinity") or {}).get("NodeAffinity") or {}).get("RequiredDuringSchedulingIgnoredDuringExecution")
        if sel is not None:
            job = gofixtures.WithNodeAffinityJobs(sel.get("NodeSelectorTerms") or [], [job])[0]
            job["affinity"] = job.get("affinity") or []
        met.append({"name": r["name"], "source": r["source"], "node": to_json(r["node"]), "job": to_json(job), "priority": int(r.get("priority") or 0),
                    "expectSuccess": bool(r["expectSuccess"]), "SchedulingConfig": to_json(gofixtures.TestSchedulingConfig())})
    out["node_requirements_met"] = met

    # preempting_queue_scheduler_test.go:3196-3397 TestPreemptingQueueScheduler_RespectNodePodLimits: the table holds the parameters; the
    # test body (one 10-cpu node with a pod capacity, incumbents running on it, priority-3 challengers queued, a whole round) is restated
    # by tests/test_z_pqs_scenarios.py
    out["pqs_pod_limits"] = extract_table(f"{REF}/scheduling/preempting_queue_scheduler_test.go", "TestPreemptingQueueScheduler_RespectNodePodLimits", env, skipped)
    for c in out["pqs_pod_limits"]:
        c["SchedulingConfig"] = to_json(gofixtures.TestSchedulingConfig())

    # jobdb/comparison_test.go:13-74 TestJobPriorityComparer: pairs of jobs and the sign of SchedulingOrderCompare.  The table builds jobs
    # with method chains ((&Job{...}).WithNewRun(...) / .WithUpdatedRun(&JobRun{created: t})): an active run whose timestamp is `created`
    # (job.go: activeRunTimestamp = run.created); the two cases about equal ids are about jobDb identity, not order, and are dropped.
    out["job_priority_comparer"] = extract_comparer(skipped)
    # comparison_test.go:76-178 TestMarketJobPriorityComparer: the same kind of table for MarketSchedulingOrderCompare (bid prices per pool, currentPool)
    out["market_job_priority_comparer"] = extract_comparer(skipped, "TestMarketJobPriorityComparer")

    # nodedb_test.go:1236-1291 TestConditionalAwayNodeScheduling: one node, one job of armada-preemptible-away-conditional, through
    # SelectNodeForJobWithTxn; the job built before the table (:1237-1241) is restated with the same fixtures
    cenv = dict(env)
    cenv["gpuJobWithoutGpuToleration"] = gofixtures.TestJob("A", None, gofixtures.PriorityClass7PreemptibleAwayConditional,
                                                            gofixtures.TestPodReqs({"cpu": "8", "memory": "128Gi", "nvidia.com/gpu": "1"}))
    cond = []
    for r in extract_table(f"{REF}/nodedb/nodedb_test.go", "TestConditionalAwayNodeScheduling", cenv, skipped):
        cond.append({"name": r["name"], "source": r["source"], "SchedulingConfig": to_json(gofixtures.TestSchedulingConfig()),
                     "Nodes": to_json([r["node"]]), "Jobs": to_json([r["job"]]), "ExpectScheduled": bool(r["expectScheduled"])})
    out["nodedb_conditional_away"] = cond

    # nodedb_test.go:1150-1234 TestMatchesConditions: matchesCondition is internal to the NodeDb; each case is kept as data (conditions,
    # job resources, expected verdict) and the tests drive it end to end through an away entry whose only taints come from a
    # conditional NodeTypes entry (tests/test_z_away_conditions.py)
    menv = dict(env)
    for nm, v in (("cpu0", 0), ("cpu2", 2), ("cpu4", 4)):
        menv[nm] = v
    menv["types.AwayNodeTypeConditionOpGreaterThan"] = ">"
    menv["types.AwayNodeTypeConditionOpLessThan"] = "<"
    menv["types.AwayNodeTypeConditionOpEqual"] = "=="
    out["matches_conditions"] = extract_table(f"{REF}/nodedb/nodedb_test.go", "TestMatchesConditions", menv, skipped)

    # market-driven rounds: the same driver as TestPreemptingQueueScheduler over jobs with price bands, plus the expected spot price / billing
    # (extracted last: the fixtures number every job they create, and the other tables keep their numbering)
    out["market_pqs"] = extract_table(f"{REF}/scheduling/market_driven_preempting_queue_scheduler_test.go", "TestMarketDrivenPreemptingQueueScheduler", env, skipped)

    for k, v in out.items():
        path = os.path.join(HERE, f"{k}_cases.json")
        with open(path, "w") as f:
            json.dump(v, f, indent=1, sort_keys=True)
        print(f"{k}: {len(v)} cases -> {os.path.relpath(path)}")
    with open(os.path.join(HERE, "SKIPPED.txt"), "w") as f:
        f.write("\n".join(skipped) + "\n")
    print(f"skipped {len(skipped)} cases (see tests/golden/SKIPPED.txt)")


if __name__ == "__main__":
    main()
