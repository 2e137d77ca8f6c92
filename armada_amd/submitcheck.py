"""Host-side mirror of the reference's SubmitChecker (internal/scheduler/submitcheck.go) over the C ABI.

The reference answers "could this newly submitted job / gang ever be scheduled?" by running ScheduleManyWithTxn on a
NodeDb that was built without jobs and cleared (submitcheck.go:180-188), one job at a time, inside a transaction it
aborts (:345-349).  Because every attempt is aborted, each one meets the *same* NodeDb state: what the NodeDb answers
depends on the scheduling key alone, and that is what this mirror exploits — the NodeDb work of a whole `Check` call becomes
ONE `asched_submit_check` call per pool for the individual checks (one unit per distinct scheduling key) and one more for the
gangs, instead of one transaction round trip per job.  Everything else of `Check` is bookkeeping over those answers and is
replayed on the host in the reference's own order: the pool loop of getSchedulingResult (:309-394: submission groups, away
pools, per-queue limits, floating resources), the LRU cache of individual results — keyed by scheduling key only, so a hit
hands a job the result computed with the queue limits of whichever job filled the entry (:279-288) — and the
first-failing-member rule for gangs (:293-297).

Equality with the reference's sequential flow is what tests/test_zzz_submitcheck.py checks: against the expectations of
submitcheck_test.go (tests/golden/submitcheck_cases.json) and against a literal one-transaction-at-a-time restatement of
`Check` (tests/submitcheck_harness.py: literal_check) on seeded inputs.

Nothing here computes a placement: `PoolNodeDb.submit_check` must be backed by the native library
(armada_amd.binding.Scheduler.submit_check); there is no Python fallback.
"""
import collections
from dataclasses import dataclass, field
from typing import Callable, Dict, Hashable, List, Optional, Sequence, Tuple


@dataclass
class SubmitJob:
    """The jobdb.Job accessors the submit check reads."""
    id: str
    queue: str
    priority_class: str
    scheduling_key: Hashable                 # job.SchedulingKey() (internaltypes/podutils.go:52-72)
    request: Sequence[int]                   # AllResourceRequirements in ResourceListFactory column order
    gang_id: Optional[str] = None            # job.GetGangInfo().Id() when job.IsInGang()
    floating_request: Dict[str, int] = field(default_factory=dict)  # floating resources only (gctx.RequestsFloatingResources)


@dataclass
class PoolConfig:
    """configuration.PoolConfig fields the submit check reads (configuration.go:430-468)."""
    name: str
    away_pools: Sequence[str] = ()
    submission_group: str = ""               # ExperimentalSubmissionGroup

    def get_submission_group(self) -> str:   # configuration.go:463-468
        return self.submission_group or self.name


@dataclass
class SchedulingResult:                       # submitcheck.go:28-32
    is_schedulable: bool
    pools: List[str] = field(default_factory=list)
    reason: str = ""


class PoolNodeDb:
    """What the checker needs from one pool's cleared NodeDb (schedulerState.nodeDbByPool + constraintsByPool,
    submitcheck.go:47-51).  Implemented by the caller on top of binding.Scheduler."""

    def load_jobs(self, jobs: Sequence[SubmitJob]) -> None:
        """asched_jobs_set: job i of every later call is jobs[i]"""
        raise NotImplementedError

    def submit_check(self, units: Sequence[Sequence[int]], strip_gang: Sequence[bool]) -> List[Tuple[bool, bool, int, int]]:
        """asched_submit_check: [(ok, scheduled_away, num_schedulable, first_node)] per unit"""
        raise NotImplementedError

    def queue_resource_limit(self, queue: str, priority_class: str) -> Optional[Sequence[int]]:
        """constraints.GetQueueResourceLimit (constraints.go:180-185); None = empty ResourceList"""
        return None

    def floating_available(self) -> Dict[str, int]:
        """floatingResourceTypes.GetTotalAvailableForPool"""
        return {}

    def explain(self, job_index: int) -> Optional[str]:
        """pctx.String() (scheduling/context/pod.go:62-83) of job `job_index` of the loaded table after a failed individual check in this pool — node, number of nodes,
        NumExcludedNodesByReason — or None when the backend does not produce it (asched_select_node + asched_excluded_nodes + pod_context_string below).  Only asked for
        with SubmitChecker(explain=True): one selection + one histogram per failing scheduling key and pool."""
        return None


def quantity_string(raw: int, scale: int) -> str:
    """resource.Quantity.String() of NewScaledQuantity(raw, scale) (DecimalSI; k8s.io/apimachinery/pkg/api/resource: int64Amount.AsCanonicalBytes + the decimal suffixes):
    trailing zeros go into the exponent, the exponent is brought down to a multiple of three, the suffix names it"""
    if raw == 0:
        return "0"
    amount, exp = int(raw), int(scale)
    while amount % 10 == 0:
        amount //= 10
        exp += 1
    r = exp % 3            # Python's % is non-negative: 1 == Go's {1, -2}, 2 == Go's {2, -1}
    if r == 1:
        amount *= 10; exp -= 1
    elif r == 2:
        amount *= 100; exp -= 2
    suffix = {-9: "n", -6: "u", -3: "m", 0: "", 3: "k", 6: "M", 9: "G", 12: "T", 15: "P", 18: "E"}.get(exp)
    return f"{amount}{suffix}" if suffix is not None else f"{amount}e{exp}"


def excluded_reason_string(entry, names) -> str:
    """the reference's reason strings (nodedb/nodematching.go:14-125, nodedb.go:27) for one asched_excluded_nodes entry; `names`: .string(id) for interned label / taint strings,
    .effect(code), .resource(col) -> (name, scale), .affinity() -> the job's NodeSelector as Go prints it"""
    kind, a, b, c, required, available, _ = entry
    if kind == "implicit":
        return "insufficient resources available"
    if kind == "untolerated_taint":
        return f"taint {names.string(a)}={names.string(b)}:{names.effect(c)} not tolerated"
    if kind == "missing_label":
        return f"node does not match pod NodeSelector: label {names.string(a)} not set"
    if kind == "unmatched_label":
        return f"node does not match pod NodeSelector: required label {names.string(a)} = {names.string(b)}, but node has {names.string(c)}"
    if kind == "unmatched_affinity":
        return f"node does not match pod NodeAffinity {names.affinity()}"
    if kind == "insufficient_resources":
        name, scale = names.resource(a)
        return f"pod requires {quantity_string(required, scale)} {name}, but only {quantity_string(available, scale)} is available"
    if kind == "disallowed_resource":
        return "job requests disallowed resource and therefore cannot be scheduled"
    return kind


def pod_context_string(num_nodes: int, excluded: Sequence[Tuple[str, int]]) -> str:
    """PodSchedulingContext.String() for a pod without a node (pod.go:62-83), laid out like text/tabwriter.NewWriter(&sb, 1, 1, 1, ' ', 0): a cell is text terminated by a tab,
    a column of consecutive lines is as wide as its widest cell + 1, text behind a line's last tab is not part of a column, a line without a tab ends the column block.
    `excluded`: (reason string, count); the reference ranges over a Go map — here sorted by reason."""
    head = [("Node:", "none"), ("Number of nodes in cluster:", str(num_nodes))]
    if not excluded:
        head.append(("Excluded nodes:", "none"))
    w0 = max(len(k) for k, _ in head) + 1
    out = [k.ljust(w0) + v for k, v in head]
    if excluded:
        out.append("Excluded nodes:")
        rows = sorted(excluded)
        w1 = max(len(f"{n}:") for _, n in rows) + 1
        out += [" " + f"{n}:".ljust(w1) + reason for reason, n in rows]   # "\t%d:\t%s": an empty first cell (width 0 + padding 1), then the count column
    return "\n".join(out) + "\n"


class _Deadline:                               # submitcheck.go:34-45
    def __init__(self, limit: float, now: float):
        self.t = None if limit <= 0 else now + limit

    def exceeded(self, now: float) -> bool:
        return self.t is not None and now > self.t


class SubmitChecker:
    """SubmitChecker.Check (submitcheck.go:225-269) with the NodeDb work batched per pool."""

    def __init__(self, pools: Sequence[PoolConfig], node_db_by_pool: Dict[str, PoolNodeDb], *, max_duration: float = 0.0,
                 max_duration_per_queue: float = 0.0, now: Optional[Callable[[], float]] = None, explain: bool = False):
        self.explain = explain   # a failing individual check carries pctx.String() with NumExcludedNodesByReason, like the reference's (:372-381), instead of a fixed sentence
        self.pools = list(pools)
        self.node_db_by_pool = node_db_by_pool
        self.max_duration, self.max_duration_per_queue = max_duration, max_duration_per_queue
        self.now = now or (lambda: 0.0)
        self.pools_by_submission_group: Dict[str, List[str]] = {}   # NewSubmitChecker :75-81
        for p in self.pools:
            self.pools_by_submission_group.setdefault(p.get_submission_group(), []).append(p.name)
        self.launches = 0   # native calls issued by the last check() (the reference issues one transaction per job per pool)
        self.cache_size = 10000
        self._cache: "collections.OrderedDict[Hashable, SchedulingResult]" = collections.OrderedDict()   # jobSchedulingResultsCache (:50, :137)

    # ---- which jobs a Check call reaches before its deadlines (the clock is read exactly where the reference reads it)
    def _select(self, jobs: Sequence[SubmitJob]):
        if self.max_duration <= 0 and self.max_duration_per_queue <= 0:
            # no deadline can fire: the clock reads below are unobservable (a zero deadline never exceeds, submitcheck.go:34-45) — the same grouping without them
            gids = [j.gang_id for j in jobs]
            by_queue: Dict[str, List[int]] = {}
            for i, j in enumerate(jobs):
                by_queue.setdefault(j.queue, []).append(i)
            events: List[List[int]] = []
            for idxs in by_queue.values():
                by_gang: Dict[str, List[int]] = {}
                for i in idxs:
                    g = gids[i]
                    if g is not None:
                        by_gang.setdefault(g, []).append(i)
                processed = set()
                for i in idxs:
                    g = gids[i]
                    if g is None:
                        events.append([i])
                    elif g not in processed:
                        events.append(by_gang[g])
                        processed.add(g)
            return events
        start = self.now()
        global_deadline = _Deadline(self.max_duration, start)
        by_queue: Dict[str, List[int]] = {}
        for i, j in enumerate(jobs):                     # armadaslices.GroupByFunc; queues visited in first-appearance order
            by_queue.setdefault(j.queue, []).append(i)   # (the reference ranges over a Go map: its own order is unspecified)
        events: List[List[int]] = []          # in the reference's processing order: [job] or the members of a gang at its first member
        for _, idxs in by_queue.items():
            if global_deadline.exceeded(self.now()):
                break
            queue_deadline = _Deadline(self.max_duration_per_queue, self.now())
            by_gang: Dict[str, List[int]] = {}
            for i in idxs:
                if jobs[i].gang_id is not None:
                    by_gang.setdefault(jobs[i].gang_id, []).append(i)
            processed = set()
            for i in idxs:
                if queue_deadline.exceeded(self.now()) or global_deadline.exceeded(self.now()):
                    break
                if jobs[i].gang_id is None:
                    events.append([i])
                elif jobs[i].gang_id not in processed:
                    events.append(by_gang[jobs[i].gang_id])
                    processed.add(jobs[i].gang_id)
        return events

    # ---- getSchedulingResult's pool loop (:309-394) over per-pool answers that are already known
    def _pool_loop(self, rep: SubmitJob, members: Sequence[SubmitJob], raw: Dict[str, Tuple[bool, bool, int, int]], rep_index: int = -1) -> SchedulingResult:
        successful: Dict[str, bool] = {}
        reason: List[str] = []
        total = [sum(m.request[r] for m in members) for r in range(len(rep.request))]        # gctx.TotalResourceRequests
        floating: Dict[str, int] = {}
        for m in members:
            for k, v in m.floating_request.items():
                floating[k] = floating.get(k, 0) + v
        for pool in self.pools:
            if successful.get(pool.name):
                continue
            if any(successful.get(a) for a in pool.away_pools):
                continue
            reason.append(f"pool {pool.name}:\n")
            db = self.node_db_by_pool[pool.name]
            if any(v > 0 for v in floating.values()):      # :326-333, floatingresources WithinLimits :60-72
                avail = db.floating_available()
                if not any(v != 0 for v in avail.values()) or any(v > avail.get(k, 0) for k, v in floating.items()):
                    reason.append("job/gang requests floating resources it cannot get in this pool\n---\n")
                    continue
            limit = db.queue_resource_limit(rep.queue, rep.priority_class)                    # :335-343
            if limit is not None and any(t > l for t, l in zip(total, limit)):
                reason.append("job/gang requests resources which exceed the total limit for its queue/priority class\n---\n")
                continue
            ok, away, nsched, _ = raw[pool.name]
            if ok:
                if not away or len(pool.away_pools) > 0:                                      # :357-363
                    for p in self.pools_by_submission_group[pool.get_submission_group()]:
                        successful[p] = True
                continue
            if len(members) == 1:
                txt = db.explain(rep_index) if (self.explain and rep_index >= 0) else None
                reason.append((txt + "\n" if txt is not None else "job does not fit on any node\n") + "---\n")   # pctx.String() + "\n" + "---" + "\n" (:377-381; String() ends its last line itself: a blank line follows)
            else:
                reason.append(f": {nsched} out of {len(members)} pods schedulable\n")
        if successful:
            return SchedulingResult(True, list(successful.keys()))
        return SchedulingResult(False, [], "".join(reason))

    def update_executors(self) -> None:
        """SubmitChecker.updateExecutors swaps in a fresh state, cache included (submitcheck.go:137, 213-217); call it after the pools'
        NodeDbs have been rebuilt"""
        self._cache.clear()

    def _cache_get(self, key):            # lru.Cache.Get: a hit becomes the most recently used entry
        if key in self._cache:
            self._cache.move_to_end(key)
            return self._cache[key]
        return None

    def _cache_add(self, key, value):     # lru.Cache.Add (size 10 000, :137)
        self._cache[key] = value
        self._cache.move_to_end(key)
        while len(self._cache) > self.cache_size:
            self._cache.popitem(last=False)

    def check(self, jobs: Sequence[SubmitJob]) -> Dict[str, SchedulingResult]:
        """Results by job id for the jobs reached before the deadlines (jobs not reached are absent, like the reference).

        Three steps: (1) the NodeDb answers for every scheduling key that occurs, all pools, one native call per pool — they depend on
        the key alone; (2) the reference's control flow replayed job by job over those answers: the LRU cache keyed by scheduling key
        (a hit returns the result computed for whichever job — of whichever queue — filled it, :279-288), the queue / priority-class
        limit of the job that fills it (:335-343), the first failing member deciding a gang (:293-297); (3) the gangs whose members
        all passed, one more native call per pool."""
        self.launches = 0
        events = self._select(jobs)
        if not events:
            return {}
        keys: Dict[Hashable, int] = {}
        for ev in events:
            for i in ev:
                keys.setdefault(jobs[i].scheduling_key, i)
        reps = list(keys.values())
        unit_of_key = {k: u for u, k in enumerate(keys)}
        for db in self.node_db_by_pool.values():
            db.load_jobs(jobs)
        raw_ind: Dict[str, list] = {}
        for pool in self.pools:
            raw_ind[pool.name] = self.node_db_by_pool[pool.name].submit_check([[i] for i in reps], [True] * len(reps))
            self.launches += 1

        def individual(i: int) -> SchedulingResult:          # getIndividualSchedulingResult :272-290
            key = jobs[i].scheduling_key
            hit = self._cache_get(key)
            if hit is not None:
                return hit
            u = unit_of_key[key]
            r = self._pool_loop(jobs[i], [jobs[i]], {p.name: raw_ind[p.name][u] for p in self.pools}, rep_index=reps[u])
            self._cache_add(key, r)
            return r

        results: Dict[str, SchedulingResult] = {}
        todo: List[List[int]] = []
        if len(self._cache) + len(keys) <= self.cache_size:
            # No entry can be evicted during this call (the cache holds at most cache_size - len(keys) other keys), so every access of a key returns what its FIRST access
            # returned and the LRU order only matters afterwards: the replay is a dict lookup per job, and the order is restored from the access sequence at the end.
            local: Dict[Hashable, SchedulingResult] = {}
            touched: List[Hashable] = []
            slow = individual

            def individual(i: int) -> SchedulingResult:   # noqa: F811 (the same function, memoised per call)
                key = jobs[i].scheduling_key
                r = local.get(key)
                if r is None:
                    r = local[key] = slow(i)
                touched.append(key)
                return r
        for ev in events:
            if jobs[ev[0]].gang_id is None:
                results[jobs[ev[0]].id] = individual(ev[0])
                continue
            bad = None                                        # getGangSchedulingResult :292-302
            for i in ev:
                r = individual(i)
                if not r.is_schedulable:
                    bad = r
                    break
            if bad is not None:
                for i in ev:
                    results[jobs[i].id] = bad
            else:
                todo.append(ev)
        if todo:
            raw_gang: Dict[str, list] = {}
            for pool in self.pools:
                raw_gang[pool.name] = self.node_db_by_pool[pool.name].submit_check(todo, [False] * len(todo))
                self.launches += 1
            for u, g in enumerate(todo):
                r = self._pool_loop(jobs[g[0]], [jobs[i] for i in g], {p.name: raw_gang[p.name][u] for p in self.pools})
                for i in g:
                    results[jobs[i].id] = r
        if len(self._cache) <= self.cache_size and "touched" in locals():   # the LRU order the per-access move_to_end would have left: keys by their LAST access
            seen = set()
            order = [k for k in reversed(touched) if not (k in seen or seen.add(k))]
            for k in reversed(order):
                self._cache.move_to_end(k)
        return results
