"""ctypes binding of the C ABI declared in include/armada_sched.h.

The binding is generic over (shared library path, symbol prefix): the product loads
``armada_amd/csrc/libarmada_sched.so`` with prefix ``asched_`` (the HIP implementation); the test
suite loads the CPU oracle with prefix ``oracle_`` through the same class so that both backends are
driven by identical code.  Nothing in this package references the oracle.

Struct layouts mirror include/armada_sched.h field for field.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

MAX_RESOURCES = 8
MAX_INDEXED = 6
MAX_PRIORITIES = 16
EVICTED_PRIORITY = -2
CROSS_POOL_PRIORITY = -1

OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_INTERNAL, ERR_TIMEOUT, ERR_PEER = -1, -2, -3, -4, -5, -6

EFFECT_NONE, EFFECT_NO_SCHEDULE, EFFECT_PREFER_NO_SCHEDULE, EFFECT_NO_EXECUTE = 0, 1, 2, 3
TOL_EQUAL, TOL_EXISTS = 0, 1

METHOD_NONE, METHOD_RESCHEDULED, METHOD_NO_PREEMPTION, METHOD_FAIRSHARE, METHOD_URGENCY, METHOD_AWAY = range(6)

REASONS = {
    0: "",
    1: "maximum resources scheduled",
    2: "maximum total resources for this queue exceeded",
    3: "global scheduling rate limit exceeded",
    4: "queue scheduling rate limit exceeded",
    5: "queue cordoned",
    6: "gang would exceed global scheduling rate limit",
    7: "gang would exceed queue scheduling rate limit",
    8: "gang cardinality too large: exceeds global max burst size",
    9: "gang cardinality too large: exceeds queue max burst size",
    10: "unable to schedule gang since minimum cardinality not met",
    11: "job does not fit on any node",
    12: "resource limit exceeded",
    13: "uniformity label is not indexed",
    14: "no nodes with uniformity label",
    15: "at least one job in the gang does not fit on any node",
    16: "no remaining candidate jobs",
    17: "skipped: scheduling key known to be unfeasible",
    18: "global new job scheduling duration exceeded",
    19: "queue new job scheduling duration exceeded",
    20: "floating resources not configured for pool",
    21: "not enough floating resource in pool",
}

_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)


class CConfig(C.Structure):
    _fields_ = [
        ("num_resources", C.c_int32), ("num_indexed", C.c_int32),
        ("indexed_col", _i32p), ("indexed_resolution", _i64p),
        ("num_priority_classes", C.c_int32),
        ("pc_priority", _i32p), ("pc_preemptible", _u8p),
        ("pc_away_off", _i32p), ("away_priority", _i32p), ("away_well_known", _i32p),
        ("num_well_known_types", C.c_int32),
        ("wkt_taint_off", _i32p), ("wkt_taint_key", _i32p), ("wkt_taint_value", _i32p), ("wkt_taint_effect", _i32p),
        ("drf_multiplier", _f64p),
        ("num_indexed_taints", C.c_int32), ("indexed_taint_keys", _i32p),
        ("num_indexed_labels", C.c_int32), ("indexed_label_keys", _i32p),
        ("prefer_large_job_ordering", C.c_uint8), ("protect_uncapped_adjusted_fair_share", C.c_uint8),
        ("disable_home_scheduling", C.c_uint8), ("disable_away_scheduling", C.c_uint8),
        ("disable_gang_away_scheduling", C.c_uint8), ("disable_fairshare_scheduling", C.c_uint8),
        ("disable_urgency_scheduling", C.c_uint8), ("preempt_cross_pool_jobs_first", C.c_uint8),
        ("protected_fraction_of_fair_share", C.c_double),
        ("max_queue_lookback", C.c_uint32), ("pad2_", C.c_uint32),
        ("max_fraction_to_schedule", _f64p), ("disallowed_resource", _u8p),
        ("device", C.c_int32), ("pad3_", C.c_int32),
        ("away_nt_off", _i32p), ("away_nt_well_known", _i32p), ("away_nt_cond_off", _i32p),
        ("away_cond_resource", _i32p), ("away_cond_op", _i32p), ("away_cond_value", _i64p), ("resource_unit", _i64p),
        ("floating_resource_limit", _i64p), ("floating_counts_in_total", C.c_uint8), ("pad4_", C.c_uint8 * 7),
        ("max_new_job_scheduling_duration_ns", C.c_int64), ("max_new_job_scheduling_duration_per_queue_ns", C.c_int64), ("clock_step_ns", C.c_int64),
    ]


class CNodes(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("index", _u64p), ("id_rank", _i32p), ("total", _i64p), ("allocatable", _i64p),
        ("alloc_by_prio", _i64p), ("unschedulable", _u8p), ("over_allocated", _u8p),
        ("taint_off", _i32p), ("taint_key", _i32p), ("taint_value", _i32p), ("taint_effect", _i32p),
        ("label_off", _i32p), ("label_key", _i32p), ("label_value", _i32p),
        ("node_type_override", _i64p),
    ]


class CReqClasses(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("tol_off", _i32p), ("tol_key", _i32p), ("tol_op", _i32p), ("tol_value", _i32p),
        ("tol_effect", _i32p), ("sel_off", _i32p), ("sel_key", _i32p), ("sel_value", _i32p),
        ("has_affinity", _u8p), ("aff_term_off", _i32p), ("aff_expr_off", _i32p), ("aff_expr_key", _i32p), ("aff_expr_op", _i32p),
        ("aff_value_off", _i32p), ("aff_values", _i32p),
    ]


AFFINITY_OPS = {"In": 0, "NotIn": 1, "Exists": 2, "DoesNotExist": 3, "Gt": 4, "Lt": 5}


class CJobs(C.Structure):
    _fields_ = [
        ("m", C.c_int32), ("queue", _i32p), ("pc", _i32p), ("queue_priority", _u32p), ("submit_time", _i64p),
        ("req", _i64p), ("req_class", _i32p), ("gang_id", _i32p), ("gang_cardinality", _i32p),
        ("gang_uniformity_label", _i32p), ("node", _i32p), ("scheduled_at_priority", _i32p), ("run_timestamp", _i64p),
        ("away", C.POINTER(C.c_uint8)), ("bid_price", _f64p),
    ]


class CQueues(C.Structure):
    _fields_ = [
        ("q", C.c_int32), ("name_rank", _i32p), ("weight", _f64p), ("allocated_by_pc", _i64p), ("demand", _i64p),
        ("short_job_penalty", _i64p), ("cordoned", _u8p), ("pc_resource_limit_fraction", _f64p),
        ("global_tokens", C.c_double), ("global_burst", C.c_int64), ("global_rate_inf", C.c_uint8), ("pad_", C.c_uint8 * 7),
        ("queue_tokens", _f64p), ("queue_burst", _i64p), ("queue_rate_inf", _u8p),
        ("has_fairshare_preemption_limiter", C.c_uint8), ("pad2_", C.c_uint8 * 7),
        ("fairshare_preemption_tokens", C.c_double),
        ("queued_off", _i32p), ("queued_jobs", _i32p),
    ]


class CPodResult(C.Structure):
    _fields_ = [("node", C.c_int32), ("scheduled_at_priority", C.c_int32), ("preempted_at_priority", C.c_int32), ("method", C.c_int32)]


class COptResult(C.Structure):
    _fields_ = [("node", C.c_int32), ("num_preempted", C.c_int32), ("scheduling_cost", C.c_double), ("maximum_queue_impact", C.c_double)]


class COptimiserConfig(C.Structure):
    _fields_ = [("enabled", C.c_uint8), ("pad_", C.c_uint8 * 7), ("min_fairness_improvement_pct", C.c_double), ("max_jobs_per_round", C.c_int32), ("pad2_", C.c_int32),
                ("max_job_size_to_preempt", C.POINTER(C.c_int64)), ("min_job_size_to_schedule", C.POINTER(C.c_int64)),
                ("max_resource_fraction_to_schedule", C.POINTER(C.c_double)), ("now_ms", C.c_int64)]


class CGangPrice(C.Structure):
    _fields_ = [("evaluated", C.c_int32), ("schedulable", C.c_int32), ("price", C.c_double), ("reason", C.c_int32), ("pad_", C.c_int32)]


class CPriceNodeScore(C.Structure):
    _fields_ = [("scheduled", C.c_int32), ("num_preempted", C.c_int32), ("price", C.c_double)]


class CMarketConfig(C.Structure):
    _fields_ = [("enabled", C.c_uint8), ("pad_", C.c_uint8 * 7), ("spot_price_cutoff", C.c_double)]


class CMarketResult(C.Structure):
    _fields_ = [("has_spot_price", C.c_int32), ("pad_", C.c_int32), ("spot_price", C.c_double), ("queue_billable_resource", _i64p),
                ("queue_billable_price_override", _f64p), ("queue_has_price_override", _u8p)]


class CGlobalKeyLayout(C.Structure):
    _fields_ = [("n_fields", C.c_int32), ("field_bits", C.c_int32 * 6), ("rank_bits", C.c_int32), ("global_rank", C.POINTER(C.c_int32)), ("rank_offset", C.c_int64)]


class CDeltaSummary(C.Structure):
    _fields_ = [("conflict_nodes", C.c_int32), ("accepted", C.c_int32), ("replay", C.c_int32), ("preempted", C.c_int32)]


class COptNodeScore(C.Structure):
    _fields_ = [("scheduled", C.c_int32), ("num_preempted", C.c_int32), ("scheduling_cost", C.c_double), ("maximum_queue_impact", C.c_double)]


class CPqItem(C.Structure):
    _fields_ = [("proposed_cost", C.c_double), ("current_cost", C.c_double), ("budget", C.c_double), ("item_size", C.c_double),
                ("pc_priority", C.c_int32), ("scheduling_priority", C.c_int32), ("name_rank", C.c_int32), ("away", C.c_int32)]


class CMarketJob(C.Structure):
    _fields_ = [("price", C.c_double), ("runtime", C.c_int64), ("submit_time", C.c_int64), ("queued", C.c_int32), ("away", C.c_int32)]


class CMarketCmpJob(C.Structure):
    _fields_ = [("bid_price", C.c_double), ("active_run_timestamp", C.c_int64), ("submit_time", C.c_int64),
                ("pc_priority", C.c_int32), ("active", C.c_int32), ("id_rank", C.c_int32), ("pad_", C.c_int32)]


class CSubmitResult(C.Structure):
    _fields_ = [("ok", C.c_int32), ("scheduled_away", C.c_int32), ("num_schedulable", C.c_int32), ("first_node", C.c_int32)]


SUBMIT_STRIP_GANG = 1


class CShardArea(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ipc", C.c_char * 64)]


class CRoundResult(C.Structure):
    _fields_ = [
        ("num_scheduled", C.c_int32), ("num_preempted", C.c_int32), ("termination_reason", C.c_int32),
        ("num_evicted_phase1", C.c_int32), ("num_evicted_phase3", C.c_int32), ("num_node_queries", C.c_int32),
        ("num_loop_iterations", C.c_int32), ("pad_", C.c_int32),
        ("scheduled_job", _i32p), ("scheduled_node", _i32p), ("scheduled_priority", _i32p), ("scheduled_method", _i32p),
        ("preempted_job", _i32p), ("preempted_node", _i32p),
        ("queue_allocated_by_pc", _i64p), ("queue_fair_share", _f64p),
        ("queue_demand_capped_adjusted_fair_share", _f64p), ("queue_uncapped_adjusted_fair_share", _f64p),
        ("job_unschedulable_reason", _i32p), ("global_tokens_after", C.c_double), ("queue_tokens_after", _f64p),
    ]


class CExcludedReason(C.Structure):  # asched_excluded_reason
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("required", C.c_int64), ("available", C.c_int64),
                ("count", C.c_int32), ("pad_", C.c_int32)]


EXCL_KINDS = ("implicit", "untolerated_taint", "missing_label", "unmatched_label", "unmatched_affinity", "insufficient_resources", "disallowed_resource")

ALL_SYMBOLS = [
    "create", "destroy", "last_error", "priorities", "nodes_upsert", "jobs_set", "txn_begin", "txn_commit",
    "txn_abort", "select_node", "schedule_many", "bind", "evict", "unbind", "add_evicted", "reset_evicted",
    "get_alloc", "get_scheduled_at_priority", "iterate_nodes", "fit_select_batch", "drf_cost", "fair_shares",
    "round_prepare", "schedule_round", "schedule_queues", "gang_schedule", "round_counters", "job_key_unfeasible", "kernel_times", "round_stats",
    "clear_allocated", "submit_check", "pq_order", "submit_stats", "num_nodes", "total_resources", "node_types_matching_job", "scheduling_order",
    "optimiser_schedule_job", "set_optimiser", "set_label_value_ints", "round_timing", "set_deadline", "cancel", "cancel_clear", "indexed_node_label_values", "get_node_jobs", "get_nodes_alloc", "node_upsert",
    "market_iterate", "market_compare", "market_multi_iterate",
    "fit_select_batch_global", "round_delta_words", "round_delta", "round_delta_resolve",
    "set_market", "market_result", "price_gang", "price_job_on_nodes",
    "comm_unique_id", "comm_init", "comm_init_external", "comm_destroy", "comm_rank", "fit_select_batch_sharded", "round_exchange",
    "shard_round", "shard_exchanges", "shard_area", "shard_open", "shard_peers",
    "excluded_nodes", "set_excluded_nodes",
]
# entry points the CPU oracle does not implement (it is the single-process checker): the communicator and the collectives that run on it
OPTIONAL_SYMBOLS = {"comm_unique_id", "comm_init", "comm_init_external", "comm_destroy", "comm_rank", "fit_select_batch_sharded", "round_exchange", "shard_round", "shard_exchanges", "shard_area", "shard_open", "shard_peers"}
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class SchedError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"asched error {code}: {msg}")
        self.code = code


def _arr(a, dtype):
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _ptr(a, ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


def _csr(lists: Sequence[Sequence[Sequence[int]]], width: int):
    """lists[i] = list of width-tuples -> (off, [col arrays])"""
    off = np.zeros(len(lists) + 1, dtype=np.int32)
    cols: List[List[int]] = [[] for _ in range(width)]
    for i, l in enumerate(lists):
        off[i + 1] = off[i] + len(l)
        for t in l:
            for k in range(width):
                cols[k].append(t[k])
    return off, [np.asarray(c, dtype=np.int32) if len(c) else np.zeros(1, dtype=np.int32) for c in cols]


@dataclass
class PodResult:
    node: int
    scheduled_at_priority: int
    preempted_at_priority: int
    method: int


class RoundResult:
    """scheduling.SchedulingResult as asched_schedule_round hands it over.  The result ARRAYS (ascending job ids with their nodes / priorities / methods, the preempted
    jobs with the nodes they leave, per-job reasons, per-queue accounting) are copied out of the library's buffers when the call returns; the job -> value dict views
    below are a convenience of this binding and are built on first use (three 200 000-entry Python dicts are tens of milliseconds — more than the library's own
    host time per round — and a caller that consumes arrays, like the cgo shim, never builds them)."""

    def __init__(self, *, scheduled_job, scheduled_node, scheduled_priority_arr, scheduled_method_arr, preempted_job, preempted_node, termination_reason, num_evicted_phase1,
                 num_evicted_phase3, num_node_queries, num_loop_iterations, queue_allocated_by_pc, fair_share, demand_capped_adjusted_fair_share,
                 uncapped_adjusted_fair_share, job_unschedulable_reason, global_tokens_after, queue_tokens_after):
        self.scheduled_job, self.scheduled_node = scheduled_job, scheduled_node                     # [num_scheduled] ascending job ids, jctx.PodSchedulingContext.NodeId
        self.scheduled_priority_arr, self.scheduled_method_arr = scheduled_priority_arr, scheduled_method_arr
        self.preempted_job, self.preempted_node = preempted_job, preempted_node                     # [num_preempted] ascending job ids, jctx.AssignedNode
        self.termination_reason = termination_reason
        self.num_evicted_phase1, self.num_evicted_phase3 = num_evicted_phase1, num_evicted_phase3
        self.num_node_queries, self.num_loop_iterations = num_node_queries, num_loop_iterations
        self.queue_allocated_by_pc = queue_allocated_by_pc
        self.fair_share, self.demand_capped_adjusted_fair_share, self.uncapped_adjusted_fair_share = fair_share, demand_capped_adjusted_fair_share, uncapped_adjusted_fair_share
        self.job_unschedulable_reason = job_unschedulable_reason
        self.global_tokens_after, self.queue_tokens_after = global_tokens_after, queue_tokens_after
        self._views: Dict[str, Dict[int, int]] = {}

    def _view(self, name, keys, vals) -> Dict[int, int]:
        v = self._views.get(name)
        if v is None:
            v = self._views[name] = dict(zip(keys.tolist(), vals.tolist()))
        return v

    @property
    def scheduled(self) -> Dict[int, int]:            # job -> node
        return self._view("scheduled", self.scheduled_job, self.scheduled_node)

    @property
    def scheduled_priority(self) -> Dict[int, int]:   # job -> nodeDb.GetScheduledAtPriority
        return self._view("scheduled_priority", self.scheduled_job, self.scheduled_priority_arr)

    @property
    def scheduled_method(self) -> Dict[int, int]:
        return self._view("scheduled_method", self.scheduled_job, self.scheduled_method_arr)

    @property
    def preempted(self) -> Dict[int, int]:            # job -> node it was preempted from
        return self._view("preempted", self.preempted_job, self.preempted_node)

    # (a caller may replace a view, e.g. the golden runner re-labels node ids: tests/scenario.py)
    @scheduled.setter
    def scheduled(self, v): self._views["scheduled"] = v

    @scheduled_priority.setter
    def scheduled_priority(self, v): self._views["scheduled_priority"] = v

    @scheduled_method.setter
    def scheduled_method(self, v): self._views["scheduled_method"] = v

    @preempted.setter
    def preempted(self, v): self._views["preempted"] = v


class Library:
    """A loaded implementation of the C ABI (one per .so)."""

    def __init__(self, path: str, prefix: str = "asched_"):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing: the native library must be built first "
                f"(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path)
        f = self._fn
        f("create", C.c_void_p, [C.POINTER(CConfig)])
        f("destroy", None, [C.c_void_p])
        f("last_error", C.c_char_p, [C.c_void_p])
        f("priorities", C.c_int32, [C.c_void_p, _i32p])
        f("nodes_upsert", C.c_int32, [C.c_void_p, C.POINTER(CNodes)])
        f("jobs_set", C.c_int32, [C.c_void_p, C.POINTER(CJobs), C.POINTER(CReqClasses)])
        f("pq_order", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(CPqItem), C.c_int32, C.c_int32, _i32p, _i32p])
        f("market_multi_iterate", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(CMarketCmpJob), C.POINTER(C.c_uint8), C.c_int32, C.POINTER(CMarketCmpJob), C.POINTER(C.c_uint8), C.c_int32, _i32p, _i32p])
        f("market_compare", C.c_int32, [C.c_void_p, C.POINTER(CMarketCmpJob), C.POINTER(CMarketCmpJob), _i32p])
        f("market_iterate", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i32p, C.POINTER(CMarketJob), C.c_int32, _i32p])
        f("submit_stats", C.c_int32, [C.c_void_p, _i32p])
        f("num_nodes", C.c_int32, [C.c_void_p])
        f("scheduling_order", C.c_int32, [C.c_void_p, C.c_int32, _i32p, C.c_int32])
        f("set_label_value_ints", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i64p])
        f("set_optimiser", C.c_int32, [C.c_void_p, C.POINTER(COptimiserConfig)])
        f("optimiser_schedule_job", C.c_int32, [C.c_void_p, C.c_int32, C.c_double, _i64p, C.c_int64, C.POINTER(COptResult), _i32p, C.c_int32, C.POINTER(COptNodeScore)])
        f("round_timing", C.c_int32, [C.c_void_p, C.POINTER(C.c_double)])
        f("set_deadline", C.c_int32, [C.c_void_p, C.c_double])
        f("cancel", C.c_int32, [C.c_void_p])
        f("cancel_clear", C.c_int32, [C.c_void_p])
        f("indexed_node_label_values", C.c_int32, [C.c_void_p, C.c_int32, _i32p, C.c_int32])
        f("get_node_jobs", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _u8p, _i32p, C.c_int32])
        f("get_nodes_alloc", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i64p])
        f("node_upsert", C.c_int32, [C.c_void_p, C.c_int32, _i64p])
        f("total_resources", C.c_int32, [C.c_void_p, _i64p])
        f("node_types_matching_job", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i32p])
        f("submit_check", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i32p, _i32p, C.POINTER(CSubmitResult)])
        for n in ("txn_begin", "txn_commit", "txn_abort", "reset_evicted", "clear_allocated"):
            f(n, C.c_int32, [C.c_void_p])
        f("select_node", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(CPodResult), _i32p, C.c_int32, _i32p])
        f("schedule_many", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i32p, C.POINTER(CPodResult), _i32p, _i32p, C.c_int32, _i32p])
        f("bind", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32])
        f("evict", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32])
        f("unbind", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32])
        f("add_evicted", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32])
        f("get_alloc", C.c_int32, [C.c_void_p, C.c_int32, _i64p])
        f("get_scheduled_at_priority", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i32p])
        f("iterate_nodes", C.c_int32, [C.c_void_p, _i64p, C.c_int32, C.c_int32, _i64p, _i32p, C.c_int32, _i32p])
        f("fit_select_batch", C.c_int32, [C.c_void_p, C.c_int32, _i32p, C.c_int32, _i32p])
        f("fit_select_batch_global", C.c_int32, [C.c_void_p, C.c_int32, _i32p, C.c_int32, C.POINTER(CGlobalKeyLayout), C.c_void_p])
        f("price_gang", C.c_int32, [C.c_void_p, C.c_int32, _i32p, C.c_int64, C.POINTER(CGangPrice)])
        f("price_job_on_nodes", C.c_int32, [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(CPriceNodeScore), C.c_int32, _i32p, C.c_int32])
        f("set_market", C.c_int32, [C.c_void_p, C.POINTER(CMarketConfig)])
        f("market_result", C.c_int32, [C.c_void_p, C.POINTER(CMarketResult)])
        f("round_delta_words", C.c_int32, [C.c_void_p, _i64p])
        f("round_delta", C.c_int32, [C.c_void_p, C.c_void_p])
        f("round_delta_resolve", C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(CDeltaSummary), _i32p, _i32p, _u8p])
        f("comm_unique_id", C.c_int32, [C.c_void_p])
        f("comm_init", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32])
        f("comm_init_external", C.c_int32, [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32])
        f("comm_destroy", C.c_int32, [C.c_void_p])
        f("comm_rank", C.c_int32, [C.c_void_p, _i32p, _i32p])
        f("shard_round", C.c_int32, [C.c_void_p, C.c_int32])
        f("shard_exchanges", C.c_int64, [C.c_void_p])
        f("shard_area", C.c_int32, [C.c_void_p, C.POINTER(CShardArea)])
        f("shard_open", C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)])
        f("shard_peers", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32])
        f("fit_select_batch_sharded", C.c_int32, [C.c_void_p, C.c_int32, _i32p, C.c_int32, C.POINTER(CGlobalKeyLayout), _i32p])
        f("round_exchange", C.c_int32, [C.c_void_p, C.POINTER(CDeltaSummary), _i32p, _i32p, _u8p])
        f("drf_cost", C.c_double, [C.c_void_p, _i64p, _i64p])
        f("fair_shares", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _f64p, _f64p, _f64p, _f64p, _f64p])
        f("round_prepare", C.c_int32, [C.c_void_p, C.POINTER(CQueues)])
        f("schedule_round", C.c_int32, [C.c_void_p, C.POINTER(CRoundResult)])
        f("schedule_queues", C.c_int32, [C.c_void_p, C.POINTER(CRoundResult)])
        f("gang_schedule", C.c_int32, [C.c_void_p, C.c_int32, _i32p, _i32p, _i32p, C.POINTER(CPodResult)])
        f("round_counters", C.c_int32, [C.c_void_p, _i32p])
        f("job_key_unfeasible", C.c_int32, [C.c_void_p, C.c_int32, _i32p])
        f("kernel_times", C.c_int32, [C.c_void_p, _f64p])
        f("round_stats", C.c_int32, [C.c_void_p, _i32p])
        f("excluded_nodes", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(CExcludedReason), C.c_int32])
        f("set_excluded_nodes", C.c_int32, [C.c_void_p, C.c_int32])

    def _fn(self, name, restype, argtypes):
        if name in OPTIONAL_SYMBOLS and not hasattr(self.lib, self.prefix + name):
            return
        if os.environ.get("ASCHED_AB_OLD_LIB") and not hasattr(self.lib, self.prefix + name):
            return   # tools/ab_call.sh only: an older build of the library (A/B inside one GPU call) lacks the entry points added since
        fn = getattr(self.lib, self.prefix + name)
        fn.restype, fn.argtypes = restype, argtypes
        setattr(self, name, fn)

    def exported(self) -> List[str]:
        return [s for s in ALL_SYMBOLS if hasattr(self.lib, self.prefix + s)]


@dataclass
class Config:
    """Hot-path subset of configuration.SchedulingConfig (see asched_config)."""
    num_resources: int
    indexed_col: Sequence[int]
    indexed_resolution: Sequence[int]
    pc_priority: Sequence[int]
    pc_preemptible: Sequence[int]
    drf_multiplier: Sequence[float]
    pc_away: Optional[Sequence[Sequence[Sequence[int]]]] = None   # per pc: list of (priority, well-known-type)
    wkt_taints: Sequence[Sequence[Sequence[int]]] = ()             # per well-known type: list of (key, value, effect)
    indexed_taint_keys: Optional[Sequence[int]] = ()                # None = index all
    indexed_label_keys: Sequence[int] = ()
    prefer_large_job_ordering: bool = True
    protect_uncapped_adjusted_fair_share: bool = False
    disable_home_scheduling: bool = False
    disable_away_scheduling: bool = False
    disable_gang_away_scheduling: bool = False
    disable_fairshare_scheduling: bool = False
    disable_urgency_scheduling: bool = False
    preempt_cross_pool_jobs_first: bool = False
    protected_fraction_of_fair_share: float = 0.0
    max_queue_lookback: int = 0
    max_fraction_to_schedule: Optional[Sequence[float]] = None
    disallowed_resource: Optional[Sequence[int]] = None
    device: int = -1
    # AwayNodeType.NodeTypes: per pc, per away entry (same order as pc_away): list of (well-known type, [(resource col, op, value)])
    pc_away_node_types: Optional[Sequence[Sequence[Sequence]]] = None
    resource_unit: Optional[Sequence[int]] = None                  # factory units per whole unit (cpu 1000, memory 1)
    floating_resource_limit: Optional[Sequence[int]] = None        # [R]: pool total of a floating column, -1 = ordinary resource
    floating_counts_in_total: bool = False
    max_new_job_scheduling_duration_ns: int = 0
    max_new_job_scheduling_duration_per_queue_ns: int = 0
    clock_step_ns: int = 0


AWAY_COND_OPS = {">": 0, "<": 1, "==": 2}


class Scheduler:
    """One handle == one reference NodeDb + SchedulingContext pair (one pool, one round at a time)."""

    def __init__(self, lib: Library, cfg: Config):
        self.lib, self.cfg = lib, cfg
        self._keep: List[np.ndarray] = []
        c = CConfig()
        k = self._k
        c.num_resources = cfg.num_resources
        c.num_indexed = len(cfg.indexed_col)
        c.indexed_col = _ptr(k(cfg.indexed_col, np.int32), C.c_int32)
        c.indexed_resolution = _ptr(k(cfg.indexed_resolution, np.int64), C.c_int64)
        c.num_priority_classes = len(cfg.pc_priority)
        c.pc_priority = _ptr(k(cfg.pc_priority, np.int32), C.c_int32)
        c.pc_preemptible = _ptr(k(cfg.pc_preemptible, np.uint8), C.c_uint8)
        if cfg.pc_away is not None:
            off, (ap, aw) = _csr(cfg.pc_away, 2)
            c.pc_away_off = _ptr(k(off, np.int32), C.c_int32)
            c.away_priority = _ptr(k(ap, np.int32), C.c_int32)
            c.away_well_known = _ptr(k(aw, np.int32), C.c_int32)
        c.num_well_known_types = len(cfg.wkt_taints)
        off, (tk, tv, te) = _csr(cfg.wkt_taints, 3)
        c.wkt_taint_off = _ptr(k(off, np.int32), C.c_int32)
        c.wkt_taint_key = _ptr(k(tk, np.int32), C.c_int32)
        c.wkt_taint_value = _ptr(k(tv, np.int32), C.c_int32)
        c.wkt_taint_effect = _ptr(k(te, np.int32), C.c_int32)
        c.drf_multiplier = _ptr(k(cfg.drf_multiplier, np.float64), C.c_double)
        if cfg.indexed_taint_keys is None:
            c.num_indexed_taints = -1
        else:
            c.num_indexed_taints = len(cfg.indexed_taint_keys)
            c.indexed_taint_keys = _ptr(k(list(cfg.indexed_taint_keys) or [0], np.int32), C.c_int32)
        c.num_indexed_labels = len(cfg.indexed_label_keys)
        c.indexed_label_keys = _ptr(k(list(cfg.indexed_label_keys) or [0], np.int32), C.c_int32)
        c.prefer_large_job_ordering = int(cfg.prefer_large_job_ordering)
        c.protect_uncapped_adjusted_fair_share = int(cfg.protect_uncapped_adjusted_fair_share)
        c.disable_home_scheduling = int(cfg.disable_home_scheduling)
        c.disable_away_scheduling = int(cfg.disable_away_scheduling)
        c.disable_gang_away_scheduling = int(cfg.disable_gang_away_scheduling)
        c.disable_fairshare_scheduling = int(cfg.disable_fairshare_scheduling)
        c.disable_urgency_scheduling = int(cfg.disable_urgency_scheduling)
        c.preempt_cross_pool_jobs_first = int(cfg.preempt_cross_pool_jobs_first)
        c.protected_fraction_of_fair_share = float(cfg.protected_fraction_of_fair_share)
        c.max_queue_lookback = int(cfg.max_queue_lookback)
        if cfg.max_fraction_to_schedule is not None:
            c.max_fraction_to_schedule = _ptr(k(cfg.max_fraction_to_schedule, np.float64), C.c_double)
        if cfg.disallowed_resource is not None:
            c.disallowed_resource = _ptr(k(cfg.disallowed_resource, np.uint8), C.c_uint8)
        c.device = int(cfg.device)
        if cfg.pc_away_node_types is not None and cfg.pc_away is not None:
            nt_off, nt_wkt, cond_off, c_res, c_op, c_val = [0], [], [0], [], [], []
            for pi, entries in enumerate(cfg.pc_away):
                per_pc = cfg.pc_away_node_types[pi] if pi < len(cfg.pc_away_node_types) else []
                for ei in range(len(entries)):
                    for wkt, conds in (per_pc[ei] if ei < len(per_pc) else []):
                        nt_wkt.append(wkt)
                        for res, op, val in conds:
                            c_res.append(res); c_op.append(op); c_val.append(val)
                        cond_off.append(len(c_res))
                    nt_off.append(len(nt_wkt))
            c.away_nt_off = _ptr(k(nt_off, np.int32), C.c_int32)
            c.away_nt_well_known = _ptr(k(nt_wkt or [0], np.int32), C.c_int32)
            c.away_nt_cond_off = _ptr(k(cond_off, np.int32), C.c_int32)
            c.away_cond_resource = _ptr(k(c_res or [0], np.int32), C.c_int32)
            c.away_cond_op = _ptr(k(c_op or [0], np.int32), C.c_int32)
            c.away_cond_value = _ptr(k(c_val or [0], np.int64), C.c_int64)
        if cfg.resource_unit is not None:
            c.resource_unit = _ptr(k(cfg.resource_unit, np.int64), C.c_int64)
        if cfg.floating_resource_limit is not None:
            c.floating_resource_limit = _ptr(k(cfg.floating_resource_limit, np.int64), C.c_int64)
        c.floating_counts_in_total = int(cfg.floating_counts_in_total)
        c.max_new_job_scheduling_duration_ns = int(cfg.max_new_job_scheduling_duration_ns)
        c.max_new_job_scheduling_duration_per_queue_ns = int(cfg.max_new_job_scheduling_duration_per_queue_ns)
        c.clock_step_ns = int(cfg.clock_step_ns)
        self.h = lib.create(C.byref(c))
        if not self.h:
            raise SchedError(ERR_INVALID, "create failed (invalid config, or no gfx950 device for the HIP backend)")
        pr = (C.c_int32 * MAX_PRIORITIES)()
        self.P = lib.priorities(self.h, pr)
        self.priorities = [pr[i] for i in range(self.P)]
        self.R = cfg.num_resources
        self.K = len(cfg.indexed_col)
        self.num_nodes = 0
        self.num_jobs = 0
        self.num_queues = 0
        self._keep = []

    def _k(self, a, dtype):
        arr = _arr(a, dtype)
        self._keep.append(arr)
        return arr

    def close(self):
        if getattr(self, "h", None):
            self.lib.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise SchedError(rc, (self.lib.last_error(self.h) or b"").decode())

    # ---------------------------------------------------------------- NodeDb level
    def nodes_upsert(self, total, allocatable=None, *, index=None, id_rank=None, alloc_by_prio=None, unschedulable=None,
                     over_allocated=None, taints=None, labels=None, node_type_override=None):
        total = _arr(total, np.int64).reshape(-1, self.R)
        n = total.shape[0]
        allocatable = total if allocatable is None else _arr(allocatable, np.int64).reshape(n, self.R)
        s = CNodes()
        s.n = n
        keep = [total, allocatable]
        idx = _arr(np.arange(1, n + 1) if index is None else index, np.uint64)
        rank = _arr(np.arange(n) if id_rank is None else id_rank, np.int32)
        keep += [idx, rank]
        s.index, s.id_rank = _ptr(idx, C.c_uint64), _ptr(rank, C.c_int32)
        s.total, s.allocatable = _ptr(total, C.c_int64), _ptr(allocatable, C.c_int64)
        if alloc_by_prio is not None:
            abp = _arr(alloc_by_prio, np.int64).reshape(n, self.P, self.R)
            keep.append(abp)
            s.alloc_by_prio = _ptr(abp, C.c_int64)
        for name, val in (("unschedulable", unschedulable), ("over_allocated", over_allocated)):
            if val is not None:
                a = _arr(val, np.uint8)
                keep.append(a)
                setattr(s, name, _ptr(a, C.c_uint8))
        if taints is not None:
            off, (tk, tv, te) = _csr(taints, 3)
            keep += [off, tk, tv, te]
            s.taint_off, s.taint_key, s.taint_value, s.taint_effect = (_ptr(x, C.c_int32) for x in (off, tk, tv, te))
        if labels is not None:
            off, (lk, lv) = _csr(labels, 2)
            keep += [off, lk, lv]
            s.label_off, s.label_key, s.label_value = (_ptr(x, C.c_int32) for x in (off, lk, lv))
        if node_type_override is not None:
            o = _arr(node_type_override, np.int64)
            keep.append(o)
            s.node_type_override = _ptr(o, C.c_int64)
        self._check(self.lib.nodes_upsert(self.h, C.byref(s)))
        self.num_nodes = n

    def jobs_set(self, req, *, queue=None, pc=None, queue_priority=None, submit_time=None, req_class=None, gang_id=None,
                 gang_cardinality=None, gang_uniformity_label=None, node=None, scheduled_at_priority=None,
                 run_timestamp=None, class_tolerations=None, class_selectors=None, class_affinities=None, away=None, bid_price=None):
        """class_affinities: per class None (no required node affinity) or a list of terms, a term = list of (key, op, [values])"""
        req = _arr(req, np.int64).reshape(-1, self.R)
        m = req.shape[0]
        s = CJobs()
        s.m = m
        keep = [req]
        s.req = _ptr(req, C.c_int64)

        def put(name, val, dtype, ctype, default=None):
            if val is None:
                if default is None:
                    return
                val = np.full(m, default)
            a = _arr(val, dtype)
            assert a.shape[0] == m, name
            keep.append(a)
            setattr(s, name, _ptr(a, ctype))

        put("bid_price", bid_price, np.float64, C.c_double)
        put("queue", queue, np.int32, C.c_int32, 0)
        put("pc", pc, np.int32, C.c_int32, 0)
        put("queue_priority", queue_priority, np.uint32, C.c_uint32)
        put("submit_time", np.arange(m) if submit_time is None else submit_time, np.int64, C.c_int64)
        put("req_class", req_class, np.int32, C.c_int32)
        put("gang_id", gang_id, np.int32, C.c_int32)
        put("gang_cardinality", gang_cardinality, np.int32, C.c_int32)
        put("gang_uniformity_label", gang_uniformity_label, np.int32, C.c_int32)
        put("node", node, np.int32, C.c_int32)
        put("scheduled_at_priority", scheduled_at_priority, np.int32, C.c_int32)
        put("run_timestamp", run_timestamp, np.int64, C.c_int64)
        put("away", away, np.uint8, C.c_uint8)   # cross-pool away jobs (context.IsHomeJob false)
        cls = CReqClasses()
        tols = class_tolerations if class_tolerations is not None else [[]]
        sels = class_selectors if class_selectors is not None else [[] for _ in tols]
        assert len(tols) == len(sels)
        cls.n = len(tols)
        off, (tk, to, tv, te) = _csr(tols, 4)
        keep += [off, tk, to, tv, te]
        cls.tol_off, cls.tol_key, cls.tol_op, cls.tol_value, cls.tol_effect = (_ptr(x, C.c_int32) for x in (off, tk, to, tv, te))
        off2, (sk, sv) = _csr(sels, 2)
        keep += [off2, sk, sv]
        cls.sel_off, cls.sel_key, cls.sel_value = (_ptr(x, C.c_int32) for x in (off2, sk, sv))
        if class_affinities is not None and any(a is not None for a in class_affinities):
            assert len(class_affinities) == len(tols)
            has, term_off, expr_off, ekey, eop, val_off, vals = [], [0], [0], [], [], [0], []
            for a in class_affinities:
                has.append(0 if a is None else 1)
                for term in (a or []):
                    for key, op, values in term:
                        ekey.append(key); eop.append(op); vals.extend(values); val_off.append(len(vals))
                    expr_off.append(len(ekey))
                term_off.append(len(expr_off) - 1)
            arrs = [_arr(has, np.uint8)] + [_arr(x or [0], np.int32) for x in (term_off, expr_off, ekey, eop, val_off, vals)]
            keep += arrs
            cls.has_affinity = _ptr(arrs[0], C.c_uint8)
            (cls.aff_term_off, cls.aff_expr_off, cls.aff_expr_key, cls.aff_expr_op, cls.aff_value_off, cls.aff_values) = (_ptr(x, C.c_int32) for x in arrs[1:])
        self._check(self.lib.jobs_set(self.h, C.byref(s), C.byref(cls)))
        self.num_jobs = m

    def txn_begin(self): self._check(self.lib.txn_begin(self.h))
    def txn_commit(self): self._check(self.lib.txn_commit(self.h))
    def txn_abort(self): self._check(self.lib.txn_abort(self.h))
    def reset_evicted(self): self._check(self.lib.reset_evicted(self.h))

    def clear_allocated(self): self._check(self.lib.clear_allocated(self.h))

    def pq_order(self, items: Sequence[dict], prioritise_larger_jobs: bool, compare_scheduling_priority: bool, preempt_cross_pool_jobs_first: bool = False):
        """sort.Sort over QueueCandidateGangIteratorPQ.Less; items: dicts with proposed_cost, current_cost, budget, item_size, pc_priority,
        scheduling_priority, name_rank -> (order, packed_key_agrees)"""
        n = len(items)
        arr = (CPqItem * max(n, 1))()
        for i, it in enumerate(items):
            for k in ("proposed_cost", "current_cost", "budget", "item_size"):
                setattr(arr[i], k, float(it.get(k, 0.0)))
            for k in ("pc_priority", "scheduling_priority", "name_rank", "away"):
                setattr(arr[i], k, int(it.get(k, 0)))
        order = (C.c_int32 * max(n, 1))()
        agrees = C.c_int32(0)
        self._check(self.lib.pq_order(self.h, n, arr, int(prioritise_larger_jobs), int(bool(compare_scheduling_priority)) | (2 if preempt_cross_pool_jobs_first else 0), order, C.byref(agrees)))
        return [order[i] for i in range(n)], bool(agrees.value)

    def market_compare(self, a: dict, b: dict) -> int:
        """jobdb.MarketSchedulingOrderCompare (oracle-only test hook); a, b: dict(bid_price, active_run_timestamp, submit_time, pc_priority, active, id_rank)"""
        def mk(j):
            c = CMarketCmpJob()
            c.bid_price = float(j.get("bid_price", 0.0)); c.active_run_timestamp = int(j.get("active_run_timestamp", 0)); c.submit_time = int(j.get("submit_time", 0))
            c.pc_priority = int(j.get("pc_priority", 0)); c.active = int(bool(j.get("active", False))); c.id_rank = int(j.get("id_rank", 0))
            return c
        ca, cb = mk(a), mk(b)
        out = C.c_int32(0)
        self._check(self.lib.market_compare(self.h, C.byref(ca), C.byref(cb), C.byref(out)))
        return int(out.value)

    def market_multi_iterate(self, list1: Sequence[dict], list2: Sequence[dict], only_evicted_after: int = -1) -> List[int]:
        """MarketDrivenMultiJobsIterator over two job lists (oracle-only test hook); a job = market_compare's dict + evicted=bool; list 2's jobs come back as len(list1) + index"""
        def pack(js):
            arr = (CMarketCmpJob * max(len(js), 1))()
            ev = (C.c_uint8 * max(len(js), 1))()
            for i, j in enumerate(js):
                arr[i].bid_price = float(j.get("bid_price", 0.0)); arr[i].active_run_timestamp = int(j.get("active_run_timestamp", 0)); arr[i].submit_time = int(j.get("submit_time", 0))
                arr[i].pc_priority = int(j.get("pc_priority", 0)); arr[i].active = int(bool(j.get("active", False))); arr[i].id_rank = int(j.get("id_rank", 0))
                ev[i] = int(bool(j.get("evicted", False)))
            return arr, ev
        a1, e1 = pack(list1)
        a2, e2 = pack(list2)
        out = np.zeros(max(len(list1) + len(list2), 1), dtype=np.int32)
        n = C.c_int32(0)
        self._check(self.lib.market_multi_iterate(self.h, len(list1), a1, e1, len(list2), a2, e2, int(only_evicted_after), _ptr(out, C.c_int32), C.byref(n)))
        return [int(x) for x in out[:n.value]]

    def market_iterate(self, queues: Sequence[Sequence[dict]], name_rank: Sequence[int], preempt_cross_pool_jobs_first: bool = False) -> List[int]:
        """MarketBasedCandidateGangIterator's Peek / Clear order (oracle-only test hook); queues[q] = that queue's jobs in iterator order, a job =
        dict(price, queued=True, runtime=0, submit_time=0, away=False) -> the queue index of every yielded job"""
        nq = len(queues)
        off = np.zeros(nq + 1, dtype=np.int32)
        for q, js in enumerate(queues):
            off[q + 1] = off[q] + len(js)
        total = int(off[nq])
        arr = (CMarketJob * max(total, 1))()
        i = 0
        for js in queues:
            for j in js:
                arr[i].price = float(j["price"]); arr[i].queued = int(j.get("queued", True)); arr[i].away = int(j.get("away", False))
                arr[i].runtime = int(j.get("runtime", 0)); arr[i].submit_time = int(j.get("submit_time", 0))
                i += 1
        nr = _arr(name_rank, np.int32)
        out = np.zeros(max(total, 1), dtype=np.int32)
        self._check(self.lib.market_iterate(self.h, nq, _ptr(nr, C.c_int32), _ptr(off, C.c_int32), arr, int(preempt_cross_pool_jobs_first), _ptr(out, C.c_int32)))
        return [int(x) for x in out[:total]]

    def submit_check(self, units: Optional[Sequence[Sequence[int]]] = None, strip_gang: Optional[Sequence[bool]] = None, *, csr=None):
        """One batch of submit-check units (submitcheck.go:342-371); returns [(ok, scheduled_away, num_schedulable, first_node)].
        `units`: a list of job-id lists; or `csr=(unit_off, unit_jobs)`, the ABI's own form (explicit keyword: never guessed from the shape of `units`)."""
        if csr is not None:   # no per-member Python work (a unit of 64 000 members)
            if units is not None:
                raise ValueError("submit_check: give `units` or `csr`, not both")
            off, jobs = np.ascontiguousarray(csr[0], dtype=np.int32), np.ascontiguousarray(csr[1], dtype=np.int32)
            nu = len(off) - 1
            if nu <= 0:
                return []
        else:
            nu = len(units)
            if nu == 0:
                return []
            off = np.zeros(nu + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(u) for u in units])
            jobs = _arr([j for u in units for j in u], np.int32)
        flags = _arr([SUBMIT_STRIP_GANG if (strip_gang is not None and strip_gang[i]) else 0 for i in range(nu)], np.int32)
        out = (CSubmitResult * nu)()
        self._check(self.lib.submit_check(self.h, nu, _ptr(off, C.c_int32), _ptr(jobs, C.c_int32), _ptr(flags, C.c_int32), out))
        return [(bool(o.ok), bool(o.scheduled_away), int(o.num_schedulable), int(o.first_node)) for o in out]

    def scheduling_order(self, queue: int) -> List[int]:
        """the queue's jobs in jobdb.SchedulingOrderCompare order"""
        cap = max(self.num_jobs, 1)
        out = (C.c_int32 * cap)()
        n = self.lib.scheduling_order(self.h, queue, out, cap)
        if n < 0:
            self._check(n)
        return [out[i] for i in range(min(n, cap))]

    def total_resources(self) -> np.ndarray:
        out = np.zeros(self.R, dtype=np.int64)
        self._check(self.lib.total_resources(self.h, _ptr(out, C.c_int64)))
        return out

    def node_types_matching_job(self, job: int):
        """NodeTypesMatchingJob -> (number of matching node types, number of nodes of the other types)"""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.node_types_matching_job(self.h, job, C.byref(a), C.byref(b)))
        return a.value, b.value

    def submit_stats(self):
        out = (C.c_int32 * 6)()
        self._check(self.lib.submit_stats(self.h, out))
        return dict(wide_units=out[0], wide_passes=out[1], sequential_units=out[2], gang_units=out[3], node_passes=out[4])

    def select_node(self, job: int, pinned_node: int = -1):
        out = CPodResult()
        pre = (C.c_int32 * 4096)()
        npre = C.c_int32(0)
        self._check(self.lib.select_node(self.h, job, pinned_node, C.byref(out), pre, 4096, C.byref(npre)))
        return PodResult(out.node, out.scheduled_at_priority, out.preempted_at_priority, out.method), [pre[i] for i in range(npre.value)]

    def schedule_many(self, jobs: Sequence[int], pinned_nodes: Optional[Sequence[int]] = None):
        n = len(jobs)
        ja = _arr(jobs, np.int32)
        pa = _arr(pinned_nodes, np.int32) if pinned_nodes is not None else None
        out = (CPodResult * max(n, 1))()
        ok = C.c_int32(0)
        pre = (C.c_int32 * 65536)()
        npre = C.c_int32(0)
        self._check(self.lib.schedule_many(self.h, n, _ptr(ja, C.c_int32), _ptr(pa, C.c_int32), out, C.byref(ok), pre, 65536, C.byref(npre)))
        res = [PodResult(o.node, o.scheduled_at_priority, o.preempted_at_priority, o.method) for o in out[:n]]
        return bool(ok.value), res, [pre[i] for i in range(npre.value)]

    def bind(self, job, node, priority): self._check(self.lib.bind(self.h, job, node, priority))
    def evict(self, job, node): self._check(self.lib.evict(self.h, job, node))
    def unbind(self, job, node): self._check(self.lib.unbind(self.h, job, node))
    def add_evicted(self, index, job, node): self._check(self.lib.add_evicted(self.h, index, job, node))

    def get_alloc(self, node: int) -> np.ndarray:
        out = np.zeros((self.P, self.R), dtype=np.int64)
        self._check(self.lib.get_alloc(self.h, node, _ptr(out, C.c_int64)))
        return out

    def set_optimiser(self, enabled: bool = True, min_improvement_pct: float = 0.0, max_jobs_per_round: int = 0, max_job_size_to_preempt=None,
                      min_job_size_to_schedule=None, max_resource_fraction_to_schedule=None, now_ms: int = 0):
        """the experimental fairness optimiser as part of the following schedule_round calls (configuration.OptimiserConfig); enabled=False: off"""
        if not enabled:
            self._check(self.lib.set_optimiser(self.h, None))
            return
        c = COptimiserConfig()
        keep = []

        def arr(v, dt, ct):
            if v is None:
                return None
            a = _arr(v, dt); keep.append(a)
            return _ptr(a, ct)
        c.enabled = 1; c.min_fairness_improvement_pct = float(min_improvement_pct); c.max_jobs_per_round = int(max_jobs_per_round); c.now_ms = int(now_ms)
        c.max_job_size_to_preempt = arr(max_job_size_to_preempt, np.int64, C.c_int64)
        c.min_job_size_to_schedule = arr(min_job_size_to_schedule, np.int64, C.c_int64)
        c.max_resource_fraction_to_schedule = arr(max_resource_fraction_to_schedule, np.float64, C.c_double)
        self._check(self.lib.set_optimiser(self.h, C.byref(c)))

    def optimiser_schedule_job(self, job: int, min_improvement_pct: float = 0.0, max_job_size_to_preempt=None, now_ms: int = 0, per_node: bool = False):
        """scheduleOnNodes of the fairness optimiser for one job -> dict(node, cost, impact, preempted[, scores: [N] (scheduled, npre, cost, impact)])"""
        out = COptResult()
        cap = 256
        pre = (C.c_int32 * cap)()
        ms = None if max_job_size_to_preempt is None else _arr(max_job_size_to_preempt, np.int64)
        scores = (COptNodeScore * max(self.num_nodes, 1))() if per_node else None
        self._check(self.lib.optimiser_schedule_job(self.h, job, float(min_improvement_pct), _ptr(ms, C.c_int64), int(now_ms), C.byref(out), pre, cap, scores))
        if out.num_preempted > cap:   # a node that needs more victims than the usual buffer holds (nothing was applied: the call is repeated with room for all of them)
            cap = out.num_preempted
            pre = (C.c_int32 * cap)()
            self._check(self.lib.optimiser_schedule_job(self.h, job, float(min_improvement_pct), _ptr(ms, C.c_int64), int(now_ms), C.byref(out), pre, cap, scores))
        r = dict(node=out.node, cost=out.scheduling_cost, impact=out.maximum_queue_impact, preempted=[pre[i] for i in range(min(out.num_preempted, cap))])
        if per_node:
            r["scores"] = [(bool(s.scheduled), s.num_preempted, s.scheduling_cost, s.maximum_queue_impact) for s in scores[: self.num_nodes]]
        return r

    def set_label_value_ints(self, ints: Dict[int, int]):
        """interned label value id -> its integer value, for the values strconv.ParseInt accepts (node-affinity Gt / Lt)"""
        ids, vals = _arr(list(ints.keys()) or [0], np.int32), _arr(list(ints.values()) or [0], np.int64)
        self._check(self.lib.set_label_value_ints(self.h, len(ints), _ptr(ids, C.c_int32), _ptr(vals, C.c_int64)))

    def set_deadline(self, seconds: float): self._check(self.lib.set_deadline(self.h, float(seconds)))
    def cancel(self): self._check(self.lib.cancel(self.h))

    def cancel_clear(self): self._check(self.lib.cancel_clear(self.h))

    def indexed_node_label_values(self, label_key: int):
        """IndexedNodeLabelValues -> sorted interned values, or None when the label is not indexed"""
        cap = max(self.num_nodes, 1)
        out = (C.c_int32 * cap)()
        n = self.lib.indexed_node_label_values(self.h, label_key, out, cap)
        if n < -1:
            self._check(n)
        return None if n < 0 else [out[i] for i in range(min(n, cap))]

    def get_node_jobs(self, node: int):
        """GetNode: [(job, evicted, scheduled-at priority)] of the jobs holding resources on the node"""
        cap = max(self.num_jobs, 1)
        jobs, ev, pr = (C.c_int32 * cap)(), (C.c_uint8 * cap)(), (C.c_int32 * cap)()
        n = self.lib.get_node_jobs(self.h, node, jobs, ev, pr, cap)
        if n < 0:
            self._check(n)
        return [(jobs[i], bool(ev[i]), pr[i]) for i in range(n)]

    def get_nodes_alloc(self, nodes: Optional[Sequence[int]] = None) -> np.ndarray:
        na = None if nodes is None else _arr(nodes, np.int32)
        n = self.num_nodes if na is None else len(na)
        out = np.zeros((n, self.P, self.R), dtype=np.int64)
        self._check(self.lib.get_nodes_alloc(self.h, n, _ptr(na, C.c_int32), _ptr(out, C.c_int64)))
        return out

    def node_upsert(self, node: int, alloc_by_prio):
        a = np.ascontiguousarray(_arr(alloc_by_prio, np.int64).reshape(self.P, self.R))
        self._check(self.lib.node_upsert(self.h, node, _ptr(a, C.c_int64)))

    def get_scheduled_at_priority(self, job: int):
        o, ok = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.get_scheduled_at_priority(self.h, job, C.byref(o), C.byref(ok)))
        return (o.value, bool(ok.value))

    def iterate_nodes(self, priority: int, indexed_req: Sequence[int], type_ids: Optional[Sequence[int]] = None) -> List[int]:
        cap = max(self.num_nodes, 1)
        out = np.zeros(cap, dtype=np.int32)
        n = C.c_int32(0)
        req = _arr(indexed_req, np.int64)
        t = _arr(type_ids, np.int64) if type_ids is not None else None
        self._check(self.lib.iterate_nodes(self.h, _ptr(t, C.c_int64), -1 if t is None else len(t), priority,
                                           _ptr(req, C.c_int64), _ptr(out, C.c_int32), cap, C.byref(n)))
        return out[: n.value].tolist()

    def fit_select_batch(self, jobs: Sequence[int], priority: int = EVICTED_PRIORITY) -> np.ndarray:
        ja = _arr(jobs, np.int32)
        out = np.full(len(ja), -1, dtype=np.int32)
        self._check(self.lib.fit_select_batch(self.h, len(ja), _ptr(ja, C.c_int32), priority, _ptr(out, C.c_int32)))
        return out

    # ---- one pool on several GPUs (include/armada_sched.h): buffers are raw addresses — a torch tensor's data_ptr() on the handle's GPU (the collective then
    # reduces that tensor in place) or a numpy array's ctypes.data
    def fit_select_batch_global(self, jobs: Sequence[int], priority: int, field_bits: Sequence[int], rank_bits: int, out_ptr: int, *, rank_offset: int = 0, global_rank=None):
        ja = _arr(jobs, np.int32)
        lay = CGlobalKeyLayout()
        lay.n_fields = len(field_bits)
        for i, b in enumerate(field_bits):
            lay.field_bits[i] = int(b)
        lay.rank_bits, lay.rank_offset = int(rank_bits), int(rank_offset)
        gr = None if global_rank is None else _arr(global_rank, np.int32)
        lay.global_rank = _ptr(gr, C.c_int32) if gr is not None else None
        self._check(self.lib.fit_select_batch_global(self.h, len(ja), _ptr(ja, C.c_int32), priority, C.byref(lay), C.c_void_p(int(out_ptr))))

    def price_gang(self, jobs: Sequence[int], now_ms: int = 0):
        """GangPricer.Price -> dict(evaluated, schedulable, price, reason)"""
        ja = _arr(jobs, np.int32)
        o = CGangPrice()
        self._check(self.lib.price_gang(self.h, len(ja), _ptr(ja, C.c_int32), int(now_ms), C.byref(o)))
        return dict(evaluated=bool(o.evaluated), schedulable=bool(o.schedulable), price=float(o.price), reason=int(o.reason))

    def price_job_on_nodes(self, job: int, now_ms: int = 0, detail_node: int = -1):
        """MinPriceNodeScheduler.Schedule against every node -> ([(scheduled, num_preempted, price)] per node, victims of detail_node in order)"""
        sc = (CPriceNodeScore * max(self.num_nodes, 1))()
        cap = 256
        pre = (C.c_int32 * cap)()
        self._check(self.lib.price_job_on_nodes(self.h, int(job), int(now_ms), sc, int(detail_node), pre, cap))
        if detail_node >= 0 and sc[detail_node].num_preempted > cap:
            cap = sc[detail_node].num_preempted
            pre = (C.c_int32 * cap)()
            self._check(self.lib.price_job_on_nodes(self.h, int(job), int(now_ms), sc, int(detail_node), pre, cap))
        scores = [(bool(x.scheduled), int(x.num_preempted), float(x.price)) for x in sc[: self.num_nodes]]
        victims = [int(pre[i]) for i in range(scores[detail_node][1])] if detail_node >= 0 else []
        return scores, victims

    def set_market(self, enabled: bool = True, spot_price_cutoff: float = 0.0):
        c = CMarketConfig(); c.enabled = 1 if enabled else 0; c.spot_price_cutoff = float(spot_price_cutoff)
        self._check(self.lib.set_market(self.h, C.byref(c)))

    def market_result(self):
        """-> dict(spot_price (None: unset), billable [Q][R] int64, price_override [Q] (None: unset))"""
        r = CMarketResult()
        self._check(self.lib.market_result(self.h, C.byref(r)))
        q = self.num_queues
        bill = np.ctypeslib.as_array(r.queue_billable_resource, shape=(q * self.R,)).astype(np.int64).reshape(q, self.R) if q else np.zeros((0, self.R), dtype=np.int64)
        ov = [float(r.queue_billable_price_override[i]) if r.queue_has_price_override[i] else None for i in range(q)]
        return dict(spot_price=float(r.spot_price) if r.has_spot_price else None, billable=bill, price_override=ov)

    def round_delta_words(self) -> int:
        n = C.c_int64(0)
        self._check(self.lib.round_delta_words(self.h, C.byref(n)))
        return int(n.value)

    def round_delta(self, buf_ptr: int):
        self._check(self.lib.round_delta(self.h, C.c_void_p(int(buf_ptr))))

    def round_delta_resolve(self, reduced_ptr: int):
        """-> (summary dict, job_node int32[M], job_priority int32[M], job_replay uint8[M])"""
        m = self.num_jobs
        node, prio, rp = np.empty(max(m, 1), dtype=np.int32), np.empty(max(m, 1), dtype=np.int32), np.empty(max(m, 1), dtype=np.uint8)
        sm = CDeltaSummary()
        self._check(self.lib.round_delta_resolve(self.h, C.c_void_p(int(reduced_ptr)), C.byref(sm), _ptr(node, C.c_int32), _ptr(prio, C.c_int32), _ptr(rp, C.c_uint8)))
        return dict(conflict_nodes=sm.conflict_nodes, accepted=sm.accepted, replay=sm.replay, preempted=sm.preempted), node[:m], prio[:m], rp[:m]

    # ---- the handle's communicator (include/armada_sched.h "The communicator"): the collectives run INSIDE the library, on the handle's stream
    def comm_unique_id(self) -> bytes:
        """ncclGetUniqueId (rank 0 creates it; hand the 128 bytes to every rank)"""
        buf = C.create_string_buffer(128)
        rc = self.lib.comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise SchedError(rc, "comm_unique_id: RCCL is not available in this library")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank on the handle's GPU (blocks until every rank has called it)"""
        assert len(unique_id) == 128
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self.lib.comm_init(self.h, C.cast(buf, C.c_void_p), int(rank), int(world)))

    def comm_init_external(self, allreduce, rank: int, world: int):
        """any other transport: allreduce(ptr, count, op) -> 0 reduces `count` int64 words at address `ptr` in place over all ranks (op 0 SUM, 1 MIN, 2 MAX)"""
        def tramp(_ctx, ptr, count, op):
            try:
                return int(allreduce(int(ptr), int(count), int(op)) or 0)
            except Exception:   # an exception must not cross the C boundary
                import traceback; traceback.print_exc()
                return 1
        self._allreduce_cb = ALLREDUCE_FN(tramp)   # keep the trampoline alive as long as the handle
        self._check(self.lib.comm_init_external(self.h, self._allreduce_cb, None, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self.lib.comm_destroy(self.h))
        self._allreduce_cb = None

    def comm_rank(self):
        r, w = C.c_int32(0), C.c_int32(1)
        self._check(self.lib.comm_rank(self.h, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def shard_round(self, on: bool = True):
        """ONE pool's round on the communicator's ranks, exact: every rank holds the whole pool, the wide passes over the nodes are split and all-reduced (include/armada_sched.h)"""
        self._check(self.lib.shard_round(self.h, 1 if on else 0))

    def shard_exchanges(self) -> int:
        return int(self.lib.shard_exchanges(self.h))

    def shard_area(self):
        """this handle's exchange area of the GPU-to-GPU variant: (device pointer, 64 IPC-handle bytes for replicas in other processes)"""
        a = CShardArea()
        self._check(self.lib.shard_area(self.h, C.byref(a)))
        return int(a.ptr), C.string_at(C.addressof(a) + CShardArea.ipc.offset, 64)

    def shard_open(self, ipc: bytes) -> int:
        out = C.c_void_p(0)
        self._check(self.lib.shard_open(self.h, C.create_string_buffer(ipc, 64).raw, C.byref(out)))
        return int(out.value)

    def shard_peers(self, areas: Optional[Sequence[int]], rank: int = 0):
        """areas[r] = replica r's exchange area as this process addresses it; None: back to whole passes"""
        if areas is None:
            self._check(self.lib.shard_peers(self.h, None, 0, 0)); return
        arr = (C.c_void_p * len(areas))(*[C.c_void_p(a) for a in areas])
        self._check(self.lib.shard_peers(self.h, arr, len(areas), int(rank)))

    def fit_select_batch_sharded(self, jobs: Sequence[int], priority: int, field_bits: Sequence[int], rank_bits: int, *, rank_offset: int = 0, global_rank=None) -> np.ndarray:
        """exact node-partitioned first fit over the communicator's ranks: global rank of the chosen node per job, -1 none (collective: every rank calls it)"""
        ja = _arr(jobs, np.int32)
        lay = CGlobalKeyLayout()
        lay.n_fields = len(field_bits)
        for i, b in enumerate(field_bits):
            lay.field_bits[i] = int(b)
        lay.rank_bits, lay.rank_offset = int(rank_bits), int(rank_offset)
        gr = None if global_rank is None else _arr(global_rank, np.int32)
        lay.global_rank = _ptr(gr, C.c_int32) if gr is not None else None
        out = np.full(max(len(ja), 1), -1, dtype=np.int32)
        self._check(self.lib.fit_select_batch_sharded(self.h, len(ja), _ptr(ja, C.c_int32), priority, C.byref(lay), _ptr(out, C.c_int32)))
        return out[:len(ja)]

    def round_exchange(self):
        """queue-hash round (approximate): round_delta + ONE all-reduce SUM + round_delta_resolve on the handle's stream -> like round_delta_resolve"""
        m = self.num_jobs
        node, prio, rp = np.empty(max(m, 1), dtype=np.int32), np.empty(max(m, 1), dtype=np.int32), np.empty(max(m, 1), dtype=np.uint8)
        sm = CDeltaSummary()
        self._check(self.lib.round_exchange(self.h, C.byref(sm), _ptr(node, C.c_int32), _ptr(prio, C.c_int32), _ptr(rp, C.c_uint8)))
        return dict(conflict_nodes=sm.conflict_nodes, accepted=sm.accepted, replay=sm.replay, preempted=sm.preempted), node[:m], prio[:m], rp[:m]

    def drf_cost(self, alloc, total) -> float:
        a, t = _arr(alloc, np.int64), _arr(total, np.int64)
        return float(self.lib.drf_cost(self.h, _ptr(a, C.c_int64), _ptr(t, C.c_int64)))

    def fair_shares(self, name_rank, weight, cds):
        q = len(weight)
        nr, w, c = _arr(name_rank, np.int32), _arr(weight, np.float64), _arr(cds, np.float64)
        f, dc, uc = (np.zeros(q) for _ in range(3))
        self._check(self.lib.fair_shares(self.h, q, _ptr(nr, C.c_int32), _ptr(w, C.c_double), _ptr(c, C.c_double),
                                         _ptr(f, C.c_double), _ptr(dc, C.c_double), _ptr(uc, C.c_double)))
        return f, dc, uc

    # ---------------------------------------------------------------- round level
    def round_prepare(self, weight, queued: Sequence[Sequence[int]], *, name_rank=None, allocated_by_pc=None, demand=None,
                      short_job_penalty=None, cordoned=None, pc_resource_limit_fraction=None,
                      global_tokens=float("inf"), global_burst=2**62, global_rate_inf=True,
                      queue_tokens=None, queue_burst=None, queue_rate_inf=None,
                      fairshare_preemption_tokens=None):
        q = len(weight)
        npc = len(self.cfg.pc_priority)
        s = CQueues()
        s.q = q
        keep = []

        def put(name, val, dtype, ctype, shape=None):
            if val is None:
                return
            a = _arr(val, dtype)
            if shape is not None:
                a = np.ascontiguousarray(a.reshape(shape))
            keep.append(a)
            setattr(s, name, _ptr(a, ctype))

        put("name_rank", np.arange(q) if name_rank is None else name_rank, np.int32, C.c_int32)
        put("weight", weight, np.float64, C.c_double)
        put("allocated_by_pc", allocated_by_pc, np.int64, C.c_int64, (q, npc, self.R))
        put("demand", demand, np.int64, C.c_int64, (q, self.R))
        put("short_job_penalty", short_job_penalty, np.int64, C.c_int64, (q, self.R))
        put("cordoned", cordoned, np.uint8, C.c_uint8)
        put("pc_resource_limit_fraction", pc_resource_limit_fraction, np.float64, C.c_double, (q, npc, self.R))
        # token buckets at sctx.Started; an infinite-rate limiter always reports a full bucket
        s.global_burst = int(global_burst)
        s.global_rate_inf = int(global_rate_inf)
        s.global_tokens = float(global_burst) if (global_rate_inf and global_tokens == float("inf")) else float(global_tokens)
        qb = np.full(q, 2**62, dtype=np.int64) if queue_burst is None else _arr(queue_burst, np.int64)
        put("queue_burst", qb, np.int64, C.c_int64)
        put("queue_rate_inf", np.ones(q) if queue_rate_inf is None else queue_rate_inf, np.uint8, C.c_uint8)
        put("queue_tokens", qb.astype(np.float64) if queue_tokens is None else queue_tokens, np.float64, C.c_double)
        if fairshare_preemption_tokens is not None:
            s.has_fairshare_preemption_limiter = 1
            s.fairshare_preemption_tokens = float(fairshare_preemption_tokens)
        off = np.zeros(q + 1, dtype=np.int32)
        parts = [np.asarray(l, dtype=np.int32).reshape(-1) for l in queued]
        for i, l in enumerate(parts):
            off[i + 1] = off[i] + len(l)
        qa = np.ascontiguousarray(np.concatenate(parts)) if (parts and off[q] > 0) else np.zeros(1, dtype=np.int32)
        keep += [off, qa]
        s.queued_off, s.queued_jobs = _ptr(off, C.c_int32), _ptr(qa, C.c_int32)
        self._check(self.lib.round_prepare(self.h, C.byref(s)))
        self.num_queues = q
        self._round_keep = keep

    def gang_schedule(self, jobs: Sequence[int]):
        """GangScheduler.Schedule for one gang -> (ok, reason, [PodResult])."""
        n = len(jobs)
        ja = _arr(jobs, np.int32)
        out = (CPodResult * max(n, 1))()
        ok, reason = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.gang_schedule(self.h, n, _ptr(ja, C.c_int32), C.byref(ok), C.byref(reason), out))
        return bool(ok.value), reason.value, [PodResult(o.node, o.scheduled_at_priority, o.preempted_at_priority, o.method) for o in out[:n]]

    def round_counters(self):
        out = (C.c_int32 * 4)()
        self._check(self.lib.round_counters(self.h, out))
        return dict(num_scheduled_jobs=out[0], num_scheduled_gangs=out[1], num_evicted_jobs=out[2], num_unfeasible_keys=out[3])

    def kernel_times(self):
        """device-side durations (HIP events on the launch stream) of the kernels behind the last round / fit batch"""
        out = (C.c_double * 4)()
        self._check(self.lib.kernel_times(self.h, out))
        return dict(round_ms=out[0], fit_batch_ms=out[1], round_launches=int(out[2]), submit_check_ms=out[3])

    def round_timing(self):
        out = (C.c_double * 8)()
        self._check(self.lib.round_timing(self.h, out))
        return dict(total_ms=out[0], control_ms=out[1], launches=int(out[2]), evict1_host_ms=out[3], evict3_host_ms=out[4], final_host_ms=out[5])

    def round_stats(self):
        out = (C.c_int32 * 24)()
        self._check(self.lib.round_stats(self.h, out))
        names = ["fast_iterations", "generic_iterations", "base_scan_steps", "window_refills", "l0_max", "fast_replay_steps", "l0_overflows", "fast_active",
                 "kclk_evict", "kclk_replay", "kclk_pass1", "kclk_oversub_evict", "kclk_pass2", "kclk_unbind_results",
                 "kclk_plane_scans", "kclk_fair_selects", "stream_runs", "stream_jobs", "stream_prepared", "stream_emitted", "preempt_fast_iterations",
                 "ft_queries", "ft_retries", "ft_node_updates"]
        return {k: out[i] for i, k in enumerate(names)}

    def excluded_nodes(self, job: int):
        """PodSchedulingContext.NumExcludedNodesByReason of the job's last node selection, which ended without a node (include/armada_sched.h
        asched_excluded_reason): [(kind, a, b, c, required, available, count)] ascending, kind as in EXCL_KINDS.  [] = nothing on record."""
        cap = 64
        while True:
            buf = (CExcludedReason * cap)()
            n = self.lib.excluded_nodes(self.h, job, buf, cap)
            if n < 0:
                self._check(n)
            if n <= cap:
                return [(EXCL_KINDS[buf[i].kind], buf[i].a, buf[i].b, buf[i].c, buf[i].required, buf[i].available, buf[i].count) for i in range(n)]
            cap = n

    def set_excluded_nodes(self, max_failed_selections: int):
        self._check(self.lib.set_excluded_nodes(self.h, max_failed_selections))

    def job_key_unfeasible(self, job: int) -> bool:
        o = C.c_int32(0)
        self._check(self.lib.job_key_unfeasible(self.h, job, C.byref(o)))
        return bool(o.value)

    def schedule_queues(self) -> RoundResult:
        return self.schedule_round(queues_only=True)

    def schedule_round(self, queues_only: bool = False) -> RoundResult:
        r = CRoundResult()
        self._check((self.lib.schedule_queues if queues_only else self.lib.schedule_round)(self.h, C.byref(r)))
        ns, npre = r.num_scheduled, r.num_preempted
        npc = len(self.cfg.pc_priority)

        def arr(p, n, dtype):
            if n == 0 or not p:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)

        sj, sn = arr(r.scheduled_job, ns, np.int32), arr(r.scheduled_node, ns, np.int32)
        sp, sm = arr(r.scheduled_priority, ns, np.int32), arr(r.scheduled_method, ns, np.int32)
        pj, pn = arr(r.preempted_job, npre, np.int32), arr(r.preempted_node, npre, np.int32)
        q = self.num_queues
        return RoundResult(
            scheduled_job=sj, scheduled_node=sn, scheduled_priority_arr=sp, scheduled_method_arr=sm, preempted_job=pj, preempted_node=pn,
            termination_reason=r.termination_reason,
            num_evicted_phase1=r.num_evicted_phase1, num_evicted_phase3=r.num_evicted_phase3,
            num_node_queries=r.num_node_queries, num_loop_iterations=r.num_loop_iterations,
            queue_allocated_by_pc=arr(r.queue_allocated_by_pc, q * npc * self.R, np.int64).reshape(q, npc, self.R),
            fair_share=arr(r.queue_fair_share, q, np.float64),
            demand_capped_adjusted_fair_share=arr(r.queue_demand_capped_adjusted_fair_share, q, np.float64),
            uncapped_adjusted_fair_share=arr(r.queue_uncapped_adjusted_fair_share, q, np.float64),
            job_unschedulable_reason=arr(r.job_unschedulable_reason, self.num_jobs, np.int32),
            global_tokens_after=r.global_tokens_after,
            queue_tokens_after=arr(r.queue_tokens_after, q, np.float64),
        )
