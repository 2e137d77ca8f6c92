/*
 * armada_sched.h — C ABI of the MI355X-native Armada scheduling-round hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI seam on this path; the
 * seam is the Go interface `SchedulingAlgo` (internal/scheduler/scheduling/scheduling_algo.go:43-47)
 * plus the concrete `*nodedb.NodeDb` (internal/scheduler/nodedb/nodedb.go:97-191).  Every entry
 * point below names the reference method it replaces; a cgo shim binds them 1:1.
 *
 * Two libraries export this ABI:
 *   - armada_amd/csrc/libarmada_sched.so   (prefix asched_)  HIP/gfx950 implementation — the product.
 *   - oracle/liboracle.so                  (prefix oracle_)  CPU restatement — TEST INFRASTRUCTURE ONLY.
 * Both are generated from this one header through ASCHED_FN(), so signatures cannot drift.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types.  All input buffers are borrowed for the call.
 *   - return value: 0 = ok, <0 = error (asched_last_error() gives text).  "no feasible node" is a
 *     normal result (node = -1), exactly like the reference's (nil, nil, nil) (nodedb.go:629).
 *   - one thread per handle; handles are independent (NodeDb is not goroutine-safe either).
 *   - strings never cross the boundary: queue names, node ids, label/taint keys and values are
 *     interned by the caller into dense int32 ids; where the reference orders by string
 *     (queue name: queue_scheduler.go:797, scheduling.go:284; node id: nodeiteration.go:184) the
 *     caller passes the *rank* of the string in lexicographic order.
 *   - resource vectors are int64[R] in ResourceListFactory column order
 *     (internaltypes/resource_list_factory.go:41-52), row-major [n][R] at the boundary.
 */
#ifndef ARMADA_SCHED_H
#define ARMADA_SCHED_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ASCHED_PREFIX
#define ASCHED_PREFIX asched_
#endif
#define ASCHED_CAT2(a, b) a##b
#define ASCHED_CAT(a, b) ASCHED_CAT2(a, b)
#define ASCHED_FN(name) ASCHED_CAT(ASCHED_PREFIX, name)

#define ASCHED_MAX_RESOURCES 8
#define ASCHED_MAX_INDEXED 6
#define ASCHED_MAX_PRIORITIES 16

/* NodeDb-internal priorities (internaltypes/node.go:17-28). */
#define ASCHED_EVICTED_PRIORITY (-2)
#define ASCHED_CROSS_POOL_PRIORITY (-1)
#define ASCHED_MIN_PRIORITY ASCHED_EVICTED_PRIORITY

/* error codes */
#define ASCHED_OK 0
#define ASCHED_ERR_INVALID (-1)      /* bad argument / inconsistent input */
#define ASCHED_ERR_UNSUPPORTED (-2)  /* feature of the reference not implemented by this backend */
#define ASCHED_ERR_DEVICE (-3)       /* HIP runtime failure or no gfx950 device */
#define ASCHED_ERR_INTERNAL (-4)     /* reference would return an error here (e.g. iteration loop) */
#define ASCHED_ERR_PEER (-6)         /* a collective entry point: another rank of the communicator failed in front of the exchange (its own call returns the cause); nothing was exchanged */
#define ASCHED_ERR_TIMEOUT (-5)      /* the round's context was done: hard timeout or cancel (queue_scheduler.go:105-112 returns ctx.Err()); no result */

/* taint effects / toleration operators (k8s core/v1), interned */
#define ASCHED_EFFECT_NONE 0
#define ASCHED_EFFECT_NO_SCHEDULE 1
#define ASCHED_EFFECT_PREFER_NO_SCHEDULE 2
#define ASCHED_EFFECT_NO_EXECUTE 3
#define ASCHED_TOLERATION_OP_EQUAL 0 /* also the empty operator */
#define ASCHED_TOLERATION_OP_EXISTS 1

/* context.SchedulingType (scheduling/context/pod.go:13-21) */
#define ASCHED_METHOD_NONE 0
#define ASCHED_METHOD_RESCHEDULED 1
#define ASCHED_METHOD_NO_PREEMPTION 2
#define ASCHED_METHOD_FAIRSHARE 3
#define ASCHED_METHOD_URGENCY 4
#define ASCHED_METHOD_AWAY 5
#define ASCHED_METHOD_OPTIMISER 6 /* ScheduledWithFairnessOptimiser (the experimental fairness optimiser, optimiser/gang_scheduler.go:240) */

/* unschedulable / termination reasons (scheduling/constraints/constraints.go:25-58) */
#define ASCHED_REASON_NONE 0
#define ASCHED_REASON_MAX_RESOURCES_SCHEDULED 1      /* terminal */
#define ASCHED_REASON_MAX_RESOURCES_PER_QUEUE 2
#define ASCHED_REASON_GLOBAL_RATE_LIMIT 3            /* terminal */
#define ASCHED_REASON_QUEUE_RATE_LIMIT 4             /* queue-terminal */
#define ASCHED_REASON_QUEUE_CORDONED 5               /* queue-terminal */
#define ASCHED_REASON_GLOBAL_RATE_LIMIT_BY_GANG 6
#define ASCHED_REASON_QUEUE_RATE_LIMIT_BY_GANG 7
#define ASCHED_REASON_GANG_EXCEEDS_GLOBAL_BURST 8
#define ASCHED_REASON_GANG_EXCEEDS_QUEUE_BURST 9
#define ASCHED_REASON_GANG_DOES_NOT_FIT 10
#define ASCHED_REASON_JOB_DOES_NOT_FIT 11
#define ASCHED_REASON_RESOURCE_LIMIT_EXCEEDED 12
#define ASCHED_REASON_UNIFORMITY_LABEL_NOT_INDEXED 13
#define ASCHED_REASON_NO_NODES_WITH_UNIFORMITY_LABEL 14
#define ASCHED_REASON_GANG_UNIFORMITY_NO_FIT 15      /* "at least one job in the gang does not fit on any node" */
#define ASCHED_REASON_NO_REMAINING_CANDIDATES 16      /* sctx.TerminationReason default, queue_scheduler.go:289-291 */
#define ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY 17       /* copied from the first failed job of the key, queue_scheduler.go:398-413 */
#define ASCHED_REASON_GLOBAL_NEW_JOB_DURATION 18      /* terminal: "global new job scheduling duration exceeded" (constraints.go:39,159-163) */
#define ASCHED_REASON_QUEUE_NEW_JOB_DURATION 19       /* queue-terminal: "queue new job scheduling duration exceeded" (constraints.go:40,165-169) */
#define ASCHED_REASON_FLOATING_NOT_CONFIGURED 20      /* "floating resources not configured for pool" (floating_resource_types.go:62-64) */
#define ASCHED_REASON_FLOATING_EXCEEDED 21            /* "not enough floating resource ... in pool" (floating_resource_types.go:66-69) */

typedef struct asched asched_t;

/* ---- configuration: the subset of configuration.SchedulingConfig that changes hot-path results
 *      (internal/scheduler/configuration/configuration.go:169-350; SURVEY.md §5 "Config / flags") ---- */
typedef struct asched_config {
  int32_t num_resources;              /* R = len(supportedResourceTypes)+floating */
  int32_t num_indexed;                /* K = len(indexedResources), config order (nodedb.go:204) */
  const int32_t* indexed_col;         /* [K] column of each indexed resource */
  const int64_t* indexed_resolution;  /* [K] nodedb.go:274-295, in factory units */

  int32_t num_priority_classes;       /* types.PriorityClass (internal/common/types/scheduling.go:56-76) */
  const int32_t* pc_priority;         /* [npc] */
  const uint8_t* pc_preemptible;      /* [npc] */
  const int32_t* pc_away_off;         /* [npc+1] CSR into away_* (may be NULL: no away node types) */
  const int32_t* away_priority;       /* AwayNodeType.Priority */
  const int32_t* away_well_known;     /* AwayNodeType.WellKnownNodeTypeName as index into wkt_*; -1 = "" (only NodeTypes entries) */
  int32_t num_well_known_types;       /* configuration.WellKnownNodeType */
  const int32_t* wkt_taint_off;       /* [nwkt+1] */
  const int32_t* wkt_taint_key;
  const int32_t* wkt_taint_value;     /* -1 = wildcard "*" (configuration.WildCardWellKnownNodeTypeValue) */
  const int32_t* wkt_taint_effect;

  const double* drf_multiplier;       /* [R] fairness.go:69-89 (1 for considered resources, 0 otherwise) */

  int32_t num_indexed_taints;         /* -1: index all taints (node_type.go:77-86) */
  const int32_t* indexed_taint_keys;
  int32_t num_indexed_labels;         /* node_type.go:101-109; 0: index none */
  const int32_t* indexed_label_keys;

  uint8_t prefer_large_job_ordering;  /* EnablePreferLargeJobOrdering */
  uint8_t protect_uncapped_adjusted_fair_share;
  uint8_t disable_home_scheduling, disable_away_scheduling, disable_gang_away_scheduling;
  uint8_t disable_fairshare_scheduling, disable_urgency_scheduling;
  uint8_t preempt_cross_pool_jobs_first;  /* PreemptCrossPoolJobsFirst: gangs of home jobs order before gangs of cross-pool away jobs, in the scheduling loop and
                                             in the eviction-order replay (queue_scheduler.go:744-746; pqs.go:603-606).  Away jobs: asched_jobs.away */
  double protected_fraction_of_fair_share;
  uint32_t max_queue_lookback;        /* 0 = unlimited (queue_scheduler.go:434-444) */
  uint32_t pad2_;
  const double* max_fraction_to_schedule; /* [R] MaximumResourceFractionToSchedule, +Inf = uncapped; NULL = all +Inf */
  const uint8_t* disallowed_resource;     /* [R] pool ExperimentalUnscheduledResources; NULL = none */
  int32_t device;                         /* HIP device ordinal this handle (one pool) lives on; <0 = the calling thread's current device */
  int32_t pad3_;
  /* AwayNodeType.NodeTypes (types.AwayTypeEntry, internal/common/types/scheduling.go:29-51): further well-known node types whose
     taints an away attempt tolerates when the entry's conditions hold for the job's requests (getEffectiveAwayNodeTaints,
     nodedb.go:650-675).  All optional: away_nt_off == NULL means no entry has NodeTypes. */
  const int32_t* away_nt_off;             /* [number of away entries + 1] CSR over the away entries (the order of away_priority) */
  const int32_t* away_nt_well_known;      /* AwayTypeEntry.Name as index into wkt_* */
  const int32_t* away_nt_cond_off;        /* [number of NodeTypes entries + 1] CSR: AwayTypeEntry.Conditions */
  const int32_t* away_cond_resource;      /* AwayNodeTypeCondition.Resource as column; -1 = not a resource of the factory (reads 0) */
  const int32_t* away_cond_op;            /* ASCHED_AWAY_COND_* ; any other value = an operator matchesCondition does not know */
  const int64_t* away_cond_value;         /* c.Value.Value(): whole units, rounded up */
  const int64_t* resource_unit;           /* [R] factory units per whole unit (cpu: 1000, memory: 1) — jobVal.Value() rounds up to whole
                                             units (nodedb.go:634-635); NULL = 1 for every resource */
  /* Floating resources (configuration.FloatingResourceConfig, floatingresources/floating_resource_types.go): pool-level quantities that are
     not on nodes.  [R]: the pool's total for a floating column (>= 0), -1 for an ordinary (Kubernetes) resource; NULL = none.  Job requests
     on a floating column count in every scheduling-context sum (AllResourceRequirements) and never constrain a node
     (KubernetesResourceRequirements, nodematching.go:184,195); GangScheduler.Schedule checks sctx.Allocated against the limit
     (gang_scheduler.go:143, context/scheduling.go:574-597).  A floating column cannot be indexed. */
  const int64_t* floating_resource_limit;
  uint8_t floating_counts_in_total;       /* 1: sctx.TotalResources = NodeDb total + floating totals (scheduling_algo.go:890); 0: NodeDb total only */
  uint8_t pad4_[7];
  /* Soft time budgets (constraints.go:159-169; configuration.go:190-209 MaxNewJobSchedulingDuration[PerQueue]); 0 = off.  Time spent on
     gangs of new jobs is accumulated per round and per queue (queue_scheduler.go:222-228) from the device's wall clock; clock_step_ns > 0
     replaces the clock by one that advances clock_step_ns per reading (testfixtures.SteppingClock, queue_scheduler_test.go:1546-1550). */
  int64_t max_new_job_scheduling_duration_ns;
  int64_t max_new_job_scheduling_duration_per_queue_ns;
  int64_t clock_step_ns;
} asched_config;

/* types.AwayNodeTypeConditionOperator (internal/common/types/scheduling.go:12-16) */
#define ASCHED_AWAY_COND_GT 0
#define ASCHED_AWAY_COND_LT 1
#define ASCHED_AWAY_COND_EQ 2

/* ---- nodes (internaltypes/node.go:32-69).  Creation order = node.index order. ---- */
typedef struct asched_nodes {
  int32_t n;
  const uint64_t* index;          /* [n] node.index (unique; node_factory.go:205-207) */
  const int32_t* id_rank;         /* [n] rank of node.id string; tie-break across node types (nodeiteration.go:184) */
  const int64_t* total;           /* [n][R] totalResources */
  const int64_t* allocatable;     /* [n][R] allocatableResources = initial AllocatableByPriority[p] for every p (node.go:79-85) */
  const int64_t* alloc_by_prio;   /* optional [n][P][R] explicit AllocatableByPriority (tests: WithUsedResourcesNodes); NULL = allocatable */
  const uint8_t* unschedulable;   /* [n] (adds the unschedulable taint, node.go:127-129) */
  const uint8_t* over_allocated;  /* [n] scheduling_algo.go:1081 */
  const int32_t* taint_off;       /* [n+1] CSR; may be NULL (no taints) */
  const int32_t* taint_key; const int32_t* taint_value; const int32_t* taint_effect;
  const int32_t* label_off;       /* [n+1] CSR; may be NULL */
  const int32_t* label_key; const int32_t* label_value;
  const int64_t* node_type_override; /* optional [n]: explicit node-type id (tests: WithNodeTypeNodes); <0 = derive from taints/labels */
} asched_nodes;

/* ---- static requirement classes: (tolerations, node selector) sets shared by many jobs.
 *      class 0 must exist; a job's scheduling key (internaltypes/podutils.go:52-72) is the
 *      equivalence class of (req class, requests, priority class) and is derived internally. ---- */
typedef struct asched_req_classes {
  int32_t n;
  const int32_t* tol_off;  /* [n+1] */
  const int32_t* tol_key;  /* -1 = empty key */
  const int32_t* tol_op;   /* ASCHED_TOLERATION_OP_* */
  const int32_t* tol_value;
  const int32_t* tol_effect; /* 0 = empty (matches all effects) */
  const int32_t* sel_off;  /* [n+1] PodRequirements.NodeSelector */
  const int32_t* sel_key; const int32_t* sel_value;
  /* PodRequirements.GetAffinityNodeSelector(): Affinity.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution, checked per node by
     NodeAffinityRequirementsMet (nodematching.go:242-255, not at node-type level :127-139).  Optional: has_affinity == NULL = no class has
     one.  Terms are ORed, the expressions of a term ANDed, a term without expressions matches no node (k8s.io/component-helpers
     v0.32.11 scheduling/corev1/nodeaffinity).  Gt / Lt compare integers: asched_set_label_value_ints tells which interned values parse.
     MatchFields is not representable: ASCHED_ERR_UNSUPPORTED. */
  const uint8_t* has_affinity;   /* [n] 0 = nil NodeSelector: every node matches */
  const int32_t* aff_term_off;   /* [n+1] CSR: NodeSelectorTerms of each class */
  const int32_t* aff_expr_off;   /* [number of terms + 1] CSR: MatchExpressions of each term */
  const int32_t* aff_expr_key;   /* interned label key */
  const int32_t* aff_expr_op;    /* ASCHED_AFFINITY_OP_* */
  const int32_t* aff_value_off;  /* [number of expressions + 1] CSR into aff_values */
  const int32_t* aff_values;     /* interned label values (In / NotIn) */
} asched_req_classes;

/* v1.NodeSelectorOperator */
#define ASCHED_AFFINITY_OP_IN 0
#define ASCHED_AFFINITY_OP_NOT_IN 1
#define ASCHED_AFFINITY_OP_EXISTS 2
#define ASCHED_AFFINITY_OP_DOES_NOT_EXIST 3
#define ASCHED_AFFINITY_OP_GT 4   /* label value and the single requirement value parsed as integers (apimachinery labels.Requirement.Matches): */
#define ASCHED_AFFINITY_OP_LT 5   /* needs asched_set_label_value_ints for both; anything that does not parse matches nothing */

/* ---- jobs: the jobdb view the round needs (jobdb/job.go accessors used on the path).
 *      Job ids are their index in this table; ROWS MUST BE IN ASCENDING JOB-ID (string) ORDER: the row is the
 *      final tie-break of SchedulingOrderCompare (jobdb/comparison.go:99-105), of MarketSchedulingOrderCompare
 *      (:160-168) and of the pricer's victim order (pricer/node_scheduler.go priceOrder.Less) — the library never
 *      sees the strings, so it cannot check this; integration/gpu_round.go sorts before it uploads. ---- */
typedef struct asched_jobs {
  int32_t m;
  const int32_t* queue;            /* [m] queue index */
  const int32_t* pc;               /* [m] priority class index */
  const uint32_t* queue_priority;  /* [m] per-queue job priority (comparison.go:74-79) */
  const int64_t* submit_time;      /* [m] comparison.go:92-97 */
  const int64_t* req;              /* [m][R] AllResourceRequirements; floating columns (asched_config.floating_resource_limit) never reach a node */
  const int32_t* req_class;        /* [m] index into asched_req_classes */
  const int32_t* gang_id;          /* [m] -1 = not in a gang; ids are unique per (queue, gang) */
  const int32_t* gang_cardinality; /* [m] */
  const int32_t* gang_uniformity_label; /* [m] interned label key, -1 = none */
  const int32_t* node;             /* [m] node (position in asched_nodes) of the active run, -1 = queued */
  const int32_t* scheduled_at_priority; /* [m] run.ScheduledAtPriority for running jobs */
  const int64_t* run_timestamp;    /* [m] activeRunTimestamp (comparison.go:83-89) */
  const uint8_t* away;             /* [m] 1 = cross-pool away job: a running job whose latest run belongs to ANOTHER pool (context.IsHomeJob false, context/util.go:9-16);
                                      NULL = none.  Such a job skips this pool's floating-resource limits (context/scheduling.go:585-594) and, when
                                      preempt_cross_pool_jobs_first is set, is bound at CrossPoolPriority (-1) whatever its run says (bindJobToNodeInPlace,
                                      nodedb.go:1055-1068: the NodeDb is told its pool exactly then, scheduling_algo.go:759-764) and orders after home jobs; it belongs to the "<queue>-away" queue context: `queue` holds that context's index
                                      (CalculateAwayQueueName; jobiteration.go:88-94, context/scheduling.go:225-226, 412-413, 646-647).  The node evictor
                                      never evicts such a job for balancing (pqs.go:102-104: job.LatestRun().Pool() != sctx.Pool); urgency preemption and the
                                      oversubscribed evictor may take it */
  const double* bid_price;         /* [m] job.GetBidPrice(pool) as the job stands at the start of the round (jobdb/job.go:459-481: QueuedBid of a queued job, RunningBid of a
                                      running one, pricing.NonPreemptibleRunningPrice = 1e6 for a running non-preemptible job); NULL = 0 everywhere.  Read by market-driven
                                      rounds only (asched_set_market) */
} asched_jobs;

/* ---- per-queue round inputs (context.AddQueueSchedulingContext, scheduling/context/scheduling.go:114-166) ---- */
typedef struct asched_queues {
  int32_t q;
  const int32_t* name_rank;        /* [q] rank of the queue name */
  const double* weight;            /* [q] 1/priorityFactor (scheduling_algo.go:811-823) */
  const int64_t* allocated_by_pc;  /* [q][npc][R] initialAllocatedByPriorityClass; NULL = derive from running jobs */
  const int64_t* demand;           /* [q][R] constrainedDemand (and demand); NULL = derive (queued + running requests) */
  const int64_t* short_job_penalty;/* [q][R] NULL = zero */
  const uint8_t* cordoned;         /* [q] NULL = none */
  const double* pc_resource_limit_fraction; /* [q][npc][R] per-queue per-PC MaximumResourceFraction (+Inf = none); NULL = none */
  /* token buckets evaluated at the fixed instant sctx.Started (constraints.go:136-157):
     tokens = limiter.TokensAt(Started); rate_inf = limiter has rate +Inf (ReserveN is a no-op) */
  double global_tokens; int64_t global_burst; uint8_t global_rate_inf; uint8_t pad_[7];
  const double* queue_tokens;      /* [q] */
  const int64_t* queue_burst;      /* [q] */
  const uint8_t* queue_rate_inf;   /* [q] */
  uint8_t has_fairshare_preemption_limiter; uint8_t pad2_[7];
  double fairshare_preemption_tokens;
  /* per-queue queued jobs in SchedulingOrderCompare order (jobdb.QueuedJobs iterator, jobiteration.go:138-176) */
  const int32_t* queued_off;       /* [q+1] */
  const int32_t* queued_jobs;      /* job indices */
} asched_queues;

/* ---- per-job outcome of a node selection (context.PodSchedulingContext, scheduling/context/pod.go:37-58) ---- */
typedef struct asched_pod_result {
  int32_t node;                 /* -1 = none */
  int32_t scheduled_at_priority;
  int32_t preempted_at_priority;
  int32_t method;               /* ASCHED_METHOD_* */
} asched_pod_result;

/* ---- PodSchedulingContext.NumExcludedNodesByReason (scheduling/context/pod.go:51) of a job whose last node selection ended WITHOUT a node: how many
 *      nodes each reason excluded.  The reference keys the histogram by the reason's string (nodematching.go:14-125); strings never cross this
 *      boundary, so an entry carries what the string is made of and the caller formats it:
 *        IMPLICIT                "insufficient resources available": nodes the iterator never yielded (nodedb.go:566-580)
 *        UNTOLERATED_TAINT       a = key, b = value, c = effect          "taint %s=%s:%s not tolerated"
 *        MISSING_LABEL           a = label key                            "node does not match pod NodeSelector: label %s not set"
 *        UNMATCHED_LABEL         a = key, b = pod value, c = node value   "... required label %s = %s, but node has %s"
 *        UNMATCHED_AFFINITY      (the job's own node affinity)            "node does not match pod NodeAffinity %s"
 *        INSUFFICIENT_RESOURCES  a = resource column, required, available "pod requires %s %s, but only %s is available" (static: the node's total; dynamic:
 *                                                                         allocatable at the priority of the attempt, nodematching.go:161-197,257-267)
 *        DISALLOWED_RESOURCE     count = NumNodes (nodedb.go:596-601)
 *      Semantics (nodedb.go:538-630,724-789,881-928): the histogram of the LAST attempt — NodeTypesMatchingJob's per-type exclusions (:1118-1133), one reason per
 *      node the iterator yielded at the job's priority (every one of them failed), the rest as IMPLICIT; the counts add up to NumNodes (the reference's own
 *      check, queue_scheduler_test.go:676-690).  A pinned (evicted) job: its node's dynamic reason, the rest IMPLICIT.
 *      Where the reference itself is order-dependent the smallest id is reported: a node selector is a Go map (nodematching.go:216-243: the first failing label
 *      in map order), node-type taints are sorted by key STRING (node_type.go:87-97) — intern taint keys in lexicographic order for the same first taint.
 *      Not produced: for jobs that got a node (the nodes rejected before the match depend on the iterator's order: nothing is on record, 0 entries).
 *      An attempt that passed the feasibility gate and still ended without a node (nodedb.go:747-789: urgency preemption disabled) reports what the reference's map
 *      holds then: the gate's walk up to the node it stopped at, plus — fair-share preemption on — the static reason of every node the failed walk over the evicted
 *      table found room on (:996-1006; such a node may be counted twice, as in the reference). ---- */
typedef struct asched_excluded_reason {
  int32_t kind;       /* ASCHED_EXCL_* */
  int32_t a, b, c;
  int64_t required, available;
  int32_t count;
  int32_t pad_;
} asched_excluded_reason;
#define ASCHED_EXCL_IMPLICIT 0
#define ASCHED_EXCL_UNTOLERATED_TAINT 1
#define ASCHED_EXCL_MISSING_LABEL 2
#define ASCHED_EXCL_UNMATCHED_LABEL 3
#define ASCHED_EXCL_UNMATCHED_AFFINITY 4
#define ASCHED_EXCL_INSUFFICIENT_RESOURCES 5
#define ASCHED_EXCL_DISALLOWED_RESOURCE 6

/* ---- one unit of a submit check: SubmitChecker.getSchedulingResult's per-pool core (internal/scheduler/submitcheck.go:345-349):
 *      txn := nodeDb.Txn(true); ok, _, err := nodeDb.ScheduleManyWithTxn(txn, gctx); txn.Abort() ---- */
typedef struct asched_submit_result {
  int32_t ok;               /* ScheduleManyWithTxn's bool */
  int32_t scheduled_away;   /* gctx.JobSchedulingContexts[0].PodSchedulingContext.ScheduledAway (submitcheck.go:358) */
  int32_t num_schedulable;  /* members whose PodSchedulingContext.IsSuccessful() (submitcheck.go:366-371) */
  int32_t first_node;       /* node of member 0, -1 = none (diagnostic; the reference only prints it) */
} asched_submit_result;
#define ASCHED_SUBMIT_STRIP_GANG 1 /* unit flag: members checked as job.WithGangInfo(BasicJobGangInfo()) (submitcheck.go:274) */

/* ---- result of a round (scheduling.SchedulingResult, scheduling/result.go:96-107) ---- */
typedef struct asched_round_result {
  int32_t num_scheduled;        /* len(ScheduledJobs) */
  int32_t num_preempted;        /* len(PreemptedJobs) */
  int32_t termination_reason;   /* sctx.TerminationReason of the first schedule() pass */
  int32_t num_evicted_phase1;
  int32_t num_evicted_phase3;
  int32_t num_node_queries;     /* selectNodeForPodAtPriority-equivalents issued (roofline accounting) */
  int32_t num_loop_iterations;  /* QueueScheduler loop iterations over both passes */
  int32_t pad_;
  const int32_t* scheduled_job;       /* [num_scheduled] job ids, ascending */
  const int32_t* scheduled_node;      /* [num_scheduled] jctx.PodSchedulingContext.NodeId */
  const int32_t* scheduled_priority;  /* [num_scheduled] nodeDb.GetScheduledAtPriority */
  const int32_t* scheduled_method;
  const int32_t* preempted_job;       /* [num_preempted] ascending */
  const int32_t* preempted_node;      /* jctx.AssignedNode */
  const int64_t* queue_allocated_by_pc; /* [q][npc][R] qctx.AllocatedByPriorityClass after the round */
  const double* queue_fair_share;       /* [q] */
  const double* queue_demand_capped_adjusted_fair_share; /* [q] */
  const double* queue_uncapped_adjusted_fair_share;      /* [q] */
  const int32_t* job_unschedulable_reason; /* [m] ASCHED_REASON_* for jobs attempted and failed (0 otherwise) */
  double global_tokens_after;
  const double* queue_tokens_after;   /* [q] */
} asched_round_result;

/* ------------------------------------------------------------------ lifecycle */
asched_t* ASCHED_FN(create)(const asched_config* cfg);            /* nodedb.NewNodeDb (nodedb.go:193) + ConfigureScheduling (:386) */
void ASCHED_FN(destroy)(asched_t*);
const char* ASCHED_FN(last_error)(asched_t*);
/* nodedb priorities: [-2,-1]+sorted-unique(PC ∪ away priorities) (nodedb.go:201-202). Returns P. */
int32_t ASCHED_FN(priorities)(asched_t*, int32_t* out /*[ASCHED_MAX_PRIORITIES]*/);

/* ------------------------------------------------------------------ NodeDb level */
/* CreateAndInsert / UpsertMany (nodedb.go:57-75,1135-1175). Replaces all nodes. */
int32_t ASCHED_FN(nodes_upsert)(asched_t*, const asched_nodes* nodes);
/* Which interned label values are integers (strconv.ParseInt(value, 10, 64) succeeds) and their value — what the Gt / Lt node-affinity
   operators compare (k8s.io/apimachinery labels.Requirement.Matches: both sides must parse, exactly one requirement value).  Replaces the
   table; call before jobs_set / nodes_upsert evaluate affinities (the static masks are rebuilt by whichever of the two comes last). */
int32_t ASCHED_FN(set_label_value_ints)(asched_t*, int32_t n, const int32_t* value_ids, const int64_t* ints);
/* registers the job + requirement tables referenced by job id in the calls below */
int32_t ASCHED_FN(jobs_set)(asched_t*, const asched_jobs* jobs, const asched_req_classes* classes);

int32_t ASCHED_FN(txn_begin)(asched_t*);    /* nodeDb.Txn(true) (nodedb.go:353) */
int32_t ASCHED_FN(txn_commit)(asched_t*);
int32_t ASCHED_FN(txn_abort)(asched_t*);

/* SelectNodeForJobWithTxn (nodedb.go:538-630). pinned_node>=0 == jctx.AssignedNode set (evicted job). */
int32_t ASCHED_FN(select_node)(asched_t*, int32_t job, int32_t pinned_node, asched_pod_result* out,
                               int32_t* preempted /*cap*/, int32_t preempted_cap, int32_t* num_preempted);
/* ScheduleManyWithTxn (nodedb.go:417-462): select+bind each member inside the caller's txn. ok=0 => caller aborts. */
int32_t ASCHED_FN(schedule_many)(asched_t*, int32_t n, const int32_t* jobs, const int32_t* pinned_nodes /*NULL*/,
                                 asched_pod_result* out /*[n]*/, int32_t* ok,
                                 int32_t* preempted, int32_t preempted_cap, int32_t* num_preempted);
/* NumExcludedNodesByReason of `job`'s last node selection (select_node, schedule_many, gang_schedule, submit_check or a round), see asched_excluded_reason.
   Entries ascending by (kind, a, b, c, required, available).  Returns the number of entries (the first `cap` are written; 0 = the job has no failed
   selection on record: never attempted, skipped by scheduling key, or failed a constraint before any node was looked at), ASCHED_ERR_UNSUPPORTED as documented
   above or when the record was dropped (more than the configured number of failed selections since the job table / round was set up, or more dynamic reasons
   than the arena holds).  asched_set_excluded_nodes: how many failed selections are kept per round / job table (default 1024; 0 = none are recorded, and the one
   wide pass per failed selection is not run). */
int32_t ASCHED_FN(excluded_nodes)(asched_t*, int32_t job, asched_excluded_reason* out, int32_t cap);
int32_t ASCHED_FN(set_excluded_nodes)(asched_t*, int32_t max_failed_selections);
/* BindJobToNode+Upsert (nodedb.go:1046-1068), EvictJobsFromNode (:1079), UnbindJobFromNode (:1108) */
int32_t ASCHED_FN(bind)(asched_t*, int32_t job, int32_t node, int32_t priority);
int32_t ASCHED_FN(evict)(asched_t*, int32_t job, int32_t node);
int32_t ASCHED_FN(unbind)(asched_t*, int32_t job, int32_t node);
/* AddEvictedJobSchedulingContextWithTxn (nodedb.go:1209) / Reset (:299).  index in [0, m): the table holds at most one entry per
   job of the job table, and the round numbers its entries 0, 1, 2 ... in eviction order (pqs.go:589-639). */
int32_t ASCHED_FN(add_evicted)(asched_t*, int32_t index, int32_t job, int32_t node);
int32_t ASCHED_FN(reset_evicted)(asched_t*);
/* NumNodes (nodedb.go:345), TotalKubernetesResources (:349: the sum of the nodes' allocatable resources, addNodeToStats :44-55) and
   NodeTypesMatchingJob (:1118-1133): how many node types the job's static requirements admit (NodeTypeJobRequirementsMet) and how many
   nodes the other types hold (the reference keys that count by a reason string; the reasons are not modelled). */
int32_t ASCHED_FN(num_nodes)(asched_t*);
int32_t ASCHED_FN(total_resources)(asched_t*, int64_t* out /*[R]*/);
int32_t ASCHED_FN(node_types_matching_job)(asched_t*, int32_t job, int32_t* num_matching_types, int32_t* num_excluded_nodes);
/* The jobs of one queue in jobdb.SchedulingOrderCompare order (jobdb/comparison.go:49-107): jobs with an active run first, then
   priority-class priority descending, queue priority ascending, (both active: run timestamp ascending,) submit time ascending, id.
   This is the order the round takes evicted jobs of a queue in (pqs.go:589-639); returns the number of jobs of the queue. */
int32_t ASCHED_FN(scheduling_order)(asched_t*, int32_t queue, int32_t* out_jobs, int32_t cap);
/* node.AllocatableByPriority, [P][R] */
int32_t ASCHED_FN(get_alloc)(asched_t*, int32_t node, int64_t* out);
int32_t ASCHED_FN(get_scheduled_at_priority)(asched_t*, int32_t job, int32_t* out, int32_t* ok); /* nodedb.go:315 */
/* ClearAllocated (nodedb.go:1178-1199): AllocatableByPriority[p] = allocatableResources for every p on every node; the
   nodes' job bookkeeping is left as it is (DeepCopyNilKeys clones the maps, internaltypes/node.go:344-372). */
int32_t ASCHED_FN(clear_allocated)(asched_t*);
/* (The HIP backend orders nodes by a packed key sized for allocatable: unbinding, after a clear, a job that was bound before it lifts a
   bucket above allocatable and is refused with ASCHED_ERR_UNSUPPORTED when the key field overflows; the submit checker's NodeDbs hold
   no jobs, submitcheck.go:180, so the reference never does that.) */
/* A batch of submit-check units against the CURRENT NodeDb state (the submit checker's NodeDb is built without jobs and
   cleared, submitcheck.go:180-188).  Unit u = jobs unit_jobs[unit_off[u] .. unit_off[u+1]) as one gang context; each unit
   runs inside its own transaction which is aborted, so units never see each other's binds (submitcheck.go:345-349).
   unit_flags[u] & ASCHED_SUBMIT_STRIP_GANG: the individual check of getIndividualSchedulingResult (submitcheck.go:272-290).
   One device launch serves the whole batch. */
int32_t ASCHED_FN(submit_check)(asched_t*, int32_t n_units, const int32_t* unit_off /*[n_units+1]*/, const int32_t* unit_jobs,
                                const int32_t* unit_flags /*[n_units] or NULL*/, asched_submit_result* out /*[n_units]*/);
/* Measurement hook (no reference counterpart): how the last submit_check ran.  out = {units answered by the wide fit kernel (individual
   checks on a pristine NodeDb), fit-kernel passes, units through the sequential control launch (a NodeDb holding jobs, away types, literal rows), gang units answered
   one workgroup per unit on a pristine NodeDb (csrc/submit_gang.h), walks over the node set the wide / capacity / gang-unit launches made (one per launch of the fit or
   capacity kernel, one per member of a gang unit: what a roofline prices), 0 (reserved)}. */
int32_t ASCHED_FN(submit_stats)(asched_t*, int32_t* out /*[6]*/);
/* NodeTypesIterator order (nodeiteration.go:74-149) for req at a priority over node types `types` (node_type_override ids; ntypes<0: all types).
   Test hook for the golden orderings of nodeiteration_test.go; the HIP backend materialises its literal iterator restatement (the one rounds use
   off the index grid) in the auxiliary kernel. */
int32_t ASCHED_FN(iterate_nodes)(asched_t*, const int64_t* type_ids, int32_t ntypes, int32_t priority,
                                 const int64_t* indexed_req /*[K]*/, int32_t* out_nodes, int32_t cap, int32_t* n_out);
/* First feasible node per job at `priority` against the CURRENT state, no binding: n independent
   selectNodeForPodAtPriority calls (nodedb.go:840-879) — BASELINE config 2 ("nodedb fit kernel"). */
int32_t ASCHED_FN(fit_select_batch)(asched_t*, int32_t n, const int32_t* jobs, int32_t priority, int32_t* out_node);

/* ------------------------------------------------------------------ one pool on several GPUs (no reference counterpart: SURVEY 8e)
 * The words the collectives reduce are produced and consumed ON THE DEVICE.  A pointer marked "device-accessible" may be memory of the
 * handle's GPU — e.g. a torch tensor's data_ptr(): the caller runs ncclAllReduce / torch.distributed.all_reduce on that tensor in place
 * (RCCL over xGMI) — or ordinary host memory (the CPU tests, gloo). */
#define ASCHED_NO_NODE_WORD INT64_MAX
/* Node-partitioned exact first fit: rows [lo, hi) of the pool's nodes live in this handle.  fit_select_batch_global answers
   fit_select_batch and packs, per query, the winning node's ORDER KEY in a layout every shard shares — (floor(allocatable / resolution) on
   the indexed resources ..., rank of the node's index among ALL nodes of the pool): RoundedNodeIndexKeyFromResourceList (encoding.go:37-54)
   as one integer — so that all-reduce MIN over the shards IS the reference's "first node in index order that fits"
   (nodeiteration.go:318-382 across the shards).  No node in this shard: ASCHED_NO_NODE_WORD.  The library's own packed key is sized per
   handle; the field widths here come from the caller (the GLOBAL maxima), n_fields must equal the number of indexed resources and
   sum(field_bits) + rank_bits <= 62. */
typedef struct asched_global_key_layout {
  int32_t n_fields;
  int32_t field_bits[6];
  int32_t rank_bits;
  const int32_t* global_rank;   /* [N] rank of each local node's index among all nodes of the pool; NULL: rank_offset + the local rank */
  int64_t rank_offset;
} asched_global_key_layout;
int32_t ASCHED_FN(fit_select_batch_global)(asched_t*, int32_t n, const int32_t* jobs, int32_t priority, const asched_global_key_layout* layout,
                                           int64_t* out_words /* device-accessible [n] */);
/* The north_star's queue-hash round: queues are split over the ranks, every rank runs schedule_round on a full replica with the queued
   jobs of ITS queues, then ONE all-reduce SUM of the buffer round_delta fills:
     buf[node * R + r]   resources this rank's NEWLY scheduled jobs (not running before the round) committed on the node
     buf[N * R + job]    (node + 1) | (priority level << 28) if this rank newly scheduled the job (at most one rank does: it owns the job's
                         queue), plus 1 << 32 if this rank preempted the job
   round_delta_resolve reads the reduced buffer: a node whose summed commitments exceed what is free there once every rank's preemptions are
   applied is a CONFLICT; new jobs on conflict nodes (and every member of a gang that has one) are not accepted — the caller replays them in
   global order through the ordinary queue scheduler (schedule_queues) on the accepted state.  Outputs, per job: the node it runs on after
   the accepted placements and all preemptions (-1: none), the priority it is bound at there, replay flag. */
typedef struct asched_delta_summary { int32_t conflict_nodes, accepted, replay, preempted; } asched_delta_summary;
int32_t ASCHED_FN(round_delta_words)(asched_t*, int64_t* n_words);
int32_t ASCHED_FN(round_delta)(asched_t*, int64_t* buf /* device-accessible [N*R + M] */);
int32_t ASCHED_FN(round_delta_resolve)(asched_t*, const int64_t* reduced /* device-accessible */, asched_delta_summary* summary,
                                       int32_t* job_node /*[M]*/, int32_t* job_priority /*[M]*/, uint8_t* job_replay /*[M]*/);

/* ---- The communicator: RCCL over xGMI inside the library (round 4).  One communicator per handle; every collective below is enqueued on the
   HANDLE'S OWN STREAM, behind the kernels that produce the words and in front of the kernels that consume them, so there is nothing for the
   caller to order (the pointer-taking entry points above leave the collective, and the stream ordering around it, to the caller).
   No reference counterpart: FairSchedulingAlgo schedules one pool on one goroutine (scheduling_algo.go:165); this is BASELINE north_star's
   "single RCCL all-reduce over xGMI".
     asched_comm_unique_id     ncclGetUniqueId: one rank creates it, the caller's control plane (the Go scheduler's leader election / gRPC, an MPI
                               broadcast, a file) hands the 128 bytes to every rank
     asched_comm_init          ncclCommInitRank on the handle's GPU; blocks until all `world` ranks have called it
     asched_comm_init_external any other transport (MPI, gloo in the CPU tests of this repository): the library calls `fn` with its stream idle;
                               `buf` is memory of the handle's GPU in the HIP library (host memory in the CPU build of the tests), count int64 words
     asched_comm_destroy       also done by asched_destroy
   op: 0 SUM, 1 MIN, 2 MAX (int64). */
typedef struct asched_unique_id { char bytes[128]; } asched_unique_id;
typedef int32_t (*asched_allreduce_fn)(void* ctx, void* buf, int64_t count, int32_t op);
int32_t ASCHED_FN(comm_unique_id)(asched_unique_id* out);
int32_t ASCHED_FN(comm_init)(asched_t*, const asched_unique_id* id, int32_t rank, int32_t world);
int32_t ASCHED_FN(comm_init_external)(asched_t*, asched_allreduce_fn fn, void* ctx, int32_t rank, int32_t world);
int32_t ASCHED_FN(comm_destroy)(asched_t*);
int32_t ASCHED_FN(comm_rank)(asched_t*, int32_t* rank, int32_t* world);   /* world = 1 without a communicator */
/* EXACT, node-partitioned: fit_select_batch over a pool whose node rows are split across the ranks of the communicator — k_fit_batch over this
   rank's rows, the order-key words packed on the device (fit_select_batch_global), ONE all-reduce MIN on the handle's stream, the winners unpacked:
   out_rank[i] = rank of the chosen node's index among ALL nodes of the pool (layout->rank_offset + local rank, or layout->global_rank), -1 none.
   Every rank must call it with the same jobs / priority / layout widths.  MIN over the shards IS the reference's first node in index order
   that fits (nodedb.go:840-879, nodeiteration.go:318-382). */
int32_t ASCHED_FN(fit_select_batch_sharded)(asched_t*, int32_t n, const int32_t* jobs, int32_t priority, const asched_global_key_layout* layout,
                                            int32_t* out_rank /*[n]*/);
/* EXACT, ONE pool's round on several GPUs (SURVEY 8e "Nodes (exact)"): every rank of the communicator holds the WHOLE pool (same nodes, jobs, queues: the state of a
   round is tens of megabytes) and runs the whole round, but after shard_round(h, 1) the round's wide passes over the nodes — the first-fit plane scan and the per-node
   evaluation of fair-share preemption (selectNodeForPodAtPriority nodedb.go:840-928, selectNodeForJobWithFairPreemption :935-1043) — look at this rank's 1/world of the
   node words only, and every pass ends with ONE all-reduce MIN of two 64-bit words (the minimum order key; the complement of the maximum evicted-table index) on the
   handle's communicator: RCCL on a side stream while the round kernel waits, or the external transport, which is then called with ASCHED_ALLREDUCE_HOST_WORDS or-ed into
   `op` (buf is host memory: reduce it in place, do not synchronise the device — the round kernel is running).  The minimum over the shares IS the unsharded pass's answer,
   so every rank computes the reference's round, bit for bit the same one.  Every rank must run the same calls on the same inputs.  shard_exchanges: all-reduces of the
   handle's last round (of its last control launch outside a round); with the GPU-to-GPU exchange below: as the round kernel counted them.  Not measured on more than one GPU. */
#define ASCHED_ALLREDUCE_HOST_WORDS 16
int32_t ASCHED_FN(shard_round)(asched_t*, int32_t on);
int64_t ASCHED_FN(shard_exchanges)(asched_t*);
/* The same exchange GPU-to-GPU, without the host (and without a communicator): every replica owns an exchange area in its HBM; the control wave of its round kernel stores
   its two words into its slot of EVERY replica's area (over xGMI for a remote one) and watches its own area fill.  shard_area: this handle's area — its device pointer and
   the hipIpcMemHandle_t bytes (all zero where the runtime cannot export it) for replicas in other processes, who map it with shard_open.  shard_peers(areas[world], world,
   rank): areas[r] = replica r's area as THIS process addresses it (areas[rank] = the own one); switches the handle's rounds to sharded passes with the direct exchange;
   world <= 64.  shard_peers(NULL, ..) switches back to whole passes.  Order: every replica calls shard_area (allocates + zeroes), the caller distributes the pointers /
   handles (that is also the barrier the zeroing needs), every replica calls shard_peers, then the replicas run the same rounds.  A replica that leaves a round early
   (deadline) leaves the counters out of step: start over with fresh handles.  Exercised with two replicas on ONE GPU (two round kernels side by side); never over xGMI. */
typedef struct asched_shard_area_t { void* ptr; char ipc[64]; } asched_shard_area_t;
int32_t ASCHED_FN(shard_area)(asched_t*, asched_shard_area_t* out);
int32_t ASCHED_FN(shard_open)(asched_t*, const char* ipc /*[64]*/, void** out);
int32_t ASCHED_FN(shard_peers)(asched_t*, void* const* areas /*[world]*/, int32_t world, int32_t rank);
/* APPROXIMATE (labelled so everywhere), queue-hash round: round_delta + ONE all-reduce SUM + round_delta_resolve as one stream-ordered sequence
   on the handle's stream; outputs as round_delta_resolve. */
int32_t ASCHED_FN(round_exchange)(asched_t*, asched_delta_summary* summary, int32_t* job_node /*[M]*/, int32_t* job_priority /*[M]*/, uint8_t* job_replay /*[M]*/);

/* ------------------------------------------------------------------ float helpers (goldens) */
/* DominantResourceFairness.UnweightedCostFromAllocation (fairness/fairness.go:103-105) */
double ASCHED_FN(drf_cost)(asched_t*, const int64_t* alloc /*[R]*/, const int64_t* total /*[R]*/);
/* SchedulingContext.updateFairShares (context/scheduling.go:262-342). cds = constrainedDemandShare per queue. */
int32_t ASCHED_FN(fair_shares)(asched_t*, int32_t q, const int32_t* name_rank, const double* weight, const double* cds,
                               double* fair_share, double* demand_capped, double* uncapped);

/* QueueCandidateGangIteratorPQ ordering (queue_scheduler.go:738-798): out_order = the item indices as sort.Sort leaves them
   (queue_scheduler_test.go:995-1164 drive Less exactly this way).  n <= 64.  Cross-pool "away" items do not exist within one pool.
   packed_agrees (may be NULL): 1 when, for every pair of items, the lexicographic key the fast path orders queues by (DESIGN 3.1
   item 4) gives the same verdict as Less; the CPU oracle reports 1. */
typedef struct asched_pq_item {
  double proposed_cost, current_cost, budget, item_size;   /* proposedQueueCost, currentQueueCost, queueBudget, itemSize */
  int32_t pc_priority, scheduling_priority, name_rank, away; /* priorityClassPriority, schedulingPriority, rank of the queue name, item.away (a cross-pool job's queue context) */
} asched_pq_item;
/* compare_scheduling_priority: bit 0 = compareSchedulingPriority, bit 1 (ASCHED_PQ_HOME_FIRST) = preemptCrossPoolJobsFirst: home items before away items
   (queue_scheduler.go:744-746) */
#define ASCHED_PQ_HOME_FIRST 2
int32_t ASCHED_FN(pq_order)(asched_t*, int32_t n, const asched_pq_item* items, int32_t prioritise_larger_jobs, int32_t compare_scheduling_priority,
                            int32_t* out_order /*[n]*/, int32_t* packed_agrees);

/* MarketBasedCandidateGangIterator (market_iterator.go:32-295): the order in which the market-driven candidate iterator yields the queues' jobs.  Queue q
   holds jobs[off[q] .. off[q+1]) in its iterator's order; out_queue[i] = the queue of the i-th Peek (Clear after each).  The priority queue is
   container/heap over MarketIteratorPQ.Less (:228-273), which is NOT a strict weak order — its round-robin clause reads the queue and price of the
   previous result — so the heap's up / down moves are restated literally (HIP backend: armada_amd/csrc/round_market.h, run by the auxiliary kernel).
   Test hook (market_iterator_test.go:17-122).  The market-driven ROUND — evict-everything node evictor, spot price / second-price billing, indicative
   pricer — is not built yet. */
typedef struct asched_market_job {
  double price;                    /* job.GetBidPrice(pool) */
  int64_t runtime, submit_time;    /* item.runtime as updatePQItem computes it (now - LatestRun().Created() for a job that is not queued, else 0); SubmitTime */
  int32_t queued, away;            /* job.Queued(); !IsHomeJob(pool) */
} asched_market_job;
int32_t ASCHED_FN(market_iterate)(asched_t*, int32_t nq, const int32_t* name_rank /*[nq]*/, const int32_t* off /*[nq+1]*/, const asched_market_job* jobs,
                                  int32_t preempt_cross_pool_jobs_first, int32_t* out_queue /*[off[nq]]*/);

/* jobdb.MarketSchedulingOrderCompare (jobdb/comparison.go:113-170): the order of a queue's jobs under market-driven scheduling — priority class priority
   (higher first), bid price for the pool (higher first), a job with an active run before one without, older run first, earlier submit time, job id.
   *out_sign = -1 / 0 / +1.  Test hook like market_iterate (comparison_test.go:76-178). */
typedef struct asched_market_cmp_job {
  double bid_price;                              /* job.GetBidPrice(currentPool) */
  int64_t active_run_timestamp, submit_time;
  int32_t pc_priority, active, id_rank, pad_;   /* priorityClass.Priority; activeRun != nil && !InTerminalState(); rank of the job id (equal rank = same id) */
} asched_market_cmp_job;
int32_t ASCHED_FN(market_compare)(asched_t*, const asched_market_cmp_job* a, const asched_market_cmp_job* b, int32_t* out_sign);
/* MarketDrivenMultiJobsIterator (jobiteration.go:232-321) over two InMemoryJobIterators (:22-65): the merge of a queue's two job lists by
   MarketSchedulingOrderCompare, with OnlyYieldEvicted called before the (only_evicted_after + 1)-th Next (negative: never).  out = the yielded jobs, list 1
   as its index, list 2 as n1 + index.  Test hook (jobiteration_test.go:150-232). */
int32_t ASCHED_FN(market_multi_iterate)(asched_t*, int32_t n1, const asched_market_cmp_job* list1, const uint8_t* evicted1, int32_t n2,
                                        const asched_market_cmp_job* list2, const uint8_t* evicted2, int32_t only_evicted_after, int32_t* out, int32_t* n_out);

/* ------------------------------------------------------------------ round level */
/* Builds round state: ConstructNodeDb/populateNodeDb (bind every running job, scheduling_algo.go:738-781,
   1019-1098) + constructSchedulingContext + UpdateFairShares (:783-867).  Untimed "input build". */
int32_t ASCHED_FN(round_prepare)(asched_t*, const asched_queues* queues);
/* PreemptingQueueScheduler.Schedule (preempting_queue_scheduler.go:86-289) — what the metric times.
   Result buffers are owned by the handle until the next round_prepare/destroy. */
int32_t ASCHED_FN(schedule_round)(asched_t*, asched_round_result* out);

/* QueueScheduler.Schedule (queue_scheduler.go:94-304) over the queued jobs only — no eviction phases; the entry the
   reference's queue_scheduler_test.go drives.  `preempted_*` lists the fair-share victims (sctx.PreemptedJobIds).
   On a market-driven pool (set_market) the queue is picked by the market iterator and the spot price is set, as NewQueueScheduler
   does when marketDriven is true (queue_scheduler.go:73-74, 176-203). */
int32_t ASCHED_FN(schedule_queues)(asched_t*, asched_round_result* out);
/* GangScheduler.Schedule (gang_scheduler.go:100-148) for one gang of queued jobs against the current round state. */
int32_t ASCHED_FN(gang_schedule)(asched_t*, int32_t n, const int32_t* jobs, int32_t* ok, int32_t* reason, asched_pod_result* out /*[n]*/);
/* sctx counters: out = {NumScheduledJobs, NumScheduledGangs, NumEvictedJobs, len(UnfeasibleSchedulingKeys)} (context/scheduling.go:55-69) */
int32_t ASCHED_FN(round_counters)(asched_t*, int32_t* out /*[4]*/);
/* Measurement hook (no reference counterpart): device time of the kernels behind the last call, taken with HIP events
   on the stream the kernels were launched on.  out[0] = ms of the last schedule_round/schedule_queues device work,
   out[1] = ms of the last fit_select_batch (or optimiser k_opt_score) kernel, out[2] = kernel launches behind out[0], out[3] = ms of the last submit_check launch.
   The CPU oracle reports zeros. */
int32_t ASCHED_FN(kernel_times)(asched_t*, double* out /*[4]*/);
/* Measurement hook (no reference counterpart): where the device time of the last schedule_round went.  out = {whole launch sequence ms (HIP
   events on the stream), ms inside the persistent k_control launches (the two sequential passes), kernel launches, host ms of the grid-wide
   evict-1 / evict-3 / final phases including their count read-backs, 0, 0}.  The CPU oracle reports zeros. */
int32_t ASCHED_FN(round_timing)(asched_t*, double* out /*[8]*/);
/* Measurement hook (no reference counterpart): how the last round ran on the device.  out = {fast iterations, generic
   iterations, base scan steps, window refills, max live dirty nodes (L0), fast replay steps, L0 overflows,
   fast structure active at the end, [8..15] kilo-ticks per phase, [16..19] stream runs / entries bound in them / stream entries prepared / emitted,
   [20] iterations whose job needed preemption and stayed in the fast loop, [21..23] fair-share threshold table: queries answered without a wide
   pass, validation retries, node re-evaluations}.  The CPU oracle reports zeros. */
int32_t ASCHED_FN(round_stats)(asched_t*, int32_t* out /*[24]*/);
/* Hard timeout of a round (maxSchedulingDuration, config/scheduler/config.yaml:83; scheduling_algo.go:130-134 wraps the context in
   WithTimeout, queue_scheduler.go:105-112 checks ctx.Done() every loop iteration and returns ctx.Err()).  asched_set_deadline: every
   following schedule_round / schedule_queues is cancelled `seconds` after it starts (0 = none).  asched_cancel = the context's cancel
   function: callable from ANY thread while a round is in flight; between rounds it hits the next one ("context already done").  A
   cancelled round returns ASCHED_ERR_TIMEOUT, delivers no result (scheduling_algo.go:262-270 applies nothing on error) and leaves the
   handle unprepared: round_prepare builds the next round from scratch. */
int32_t ASCHED_FN(set_deadline)(asched_t*, double seconds);
int32_t ASCHED_FN(cancel)(asched_t*);
/* Withdraws a cancel request that no round has consumed yet — for a caller whose watcher thread fired asched_cancel a moment after the round it was
   meant for had returned (the request would otherwise hit the NEXT round on this handle).  Same thread as the round calls; never during a round. */
int32_t ASCHED_FN(cancel_clear)(asched_t*);
/* IndexedNodeLabelValues (nodedb.go:340-343): values of an indexed node label present on the nodes (ascending interned id).
   Returns their number, or -1 when the label is not indexed (ok == false). */
int32_t ASCHED_FN(indexed_node_label_values)(asched_t*, int32_t label_key, int32_t* out_values, int32_t cap);
/* GetNode / GetNodeWithTxn (nodedb.go:358-394), the mutable part of *Node: jobs holding resources on the node (AllocatedByJobId), their
   evicted flag (EvictedJobRunIds) and scheduled-at priority; any out pointer may be NULL.  Returns the number of jobs on the node. */
int32_t ASCHED_FN(get_node_jobs)(asched_t*, int32_t node, int32_t* out_jobs, uint8_t* out_evicted, int32_t* out_priority, int32_t cap);
/* GetNodes / GetNodesWithTxn (nodedb.go:396-415): AllocatableByPriority of n nodes ([n][P][R]) in one download; nodes == NULL: all n == NumNodes nodes */
int32_t ASCHED_FN(get_nodes_alloc)(asched_t*, int32_t n, const int32_t* nodes, int64_t* out);
/* Upsert / UpsertWithTxn of one node (nodedb.go:1154-1175): replaces the node's AllocatableByPriority ([P][R]) and rebuilds its order key
   at every priority.  Job bookkeeping travels through bind / evict / unbind. */
int32_t ASCHED_FN(node_upsert)(asched_t*, int32_t node, const int64_t* alloc_by_prio);

/* ------------------------------------------------------------------ experimental fairness optimiser (scheduling/optimiser)
 * One job against EVERY node: PreemptingNodeScheduler.Schedule (optimiser/node_scheduler.go:42-132) per node — static requirements, fit at
 * the evicted priority, else the node's preemptible non-gang jobs scheduled at a priority <= the job's, ordered per queue (scheduled-at
 * priority, cost, age, id; preemption_info.go:23-55) and globally (priority preemptions first, then the queue whose cost after the
 * preemption stays highest; :57-91), preempted one at a time until the job fits; schedulingCost = DRF cost of the victims that bring their
 * queue to or below its fair share (:203-232), maximumQueueImpact = largest |cost change| / current cost over the queues (:101-113) — and
 * the candidate selection of FairnessOptimisingGangScheduler.scheduleOnNodes for that job (gang_scheduler.go:100-141): nodes in id order,
 * the first node that needs no preemption wins outright, otherwise nodes whose fairness improvement (cost of the job / scheduling cost,
 * in percent minus 100) exceeds the threshold, ordered by (schedulingCost, maximumQueueImpact) (scheduling_result.go:49-68).  The reference
 * breaks remaining ties by a random ULID; here the node with the smaller id rank wins.  Queue costs / fair shares / weights are the round
 * state of the handle (round_prepare, or whatever rounds ran since).  Nothing is applied: the caller unbinds the victims and binds the job
 * (markJobsScheduledAndPreempted, :173-246) through asched_unbind / asched_bind.
 * max_job_size_to_preempt: [R] or NULL, 0 = no limit on that resource (node_scheduler.go:248-268).  now_ms: the clock the job ages are taken
 * from (age = now - asched_jobs.run_timestamp / 1e6; jobs scheduled in this round have age 0).
 * per_node (optional, [NumNodes]): every node's result, for callers that schedule gangs member by member. */
typedef struct asched_opt_result {
  int32_t node;                  /* -1 = no candidate */
  int32_t num_preempted;         /* len(jobIdsToPreempt) of the chosen node */
  double scheduling_cost;
  double maximum_queue_impact;
} asched_opt_result;
typedef struct asched_opt_node_score { int32_t scheduled; int32_t num_preempted; double scheduling_cost; double maximum_queue_impact; } asched_opt_node_score;
int32_t ASCHED_FN(optimiser_schedule_job)(asched_t*, int32_t job, double min_fairness_improvement_pct, const int64_t* max_job_size_to_preempt,
                                          int64_t now_ms, asched_opt_result* out, int32_t* preempted /*cap*/, int32_t preempted_cap,
                                          asched_opt_node_score* per_node);

/* The experimental fairness optimiser as part of the round (preempting_queue_scheduler.go:224-253, 666-710): after the second pass
   OptimisingQueueScheduler.Schedule (optimising_queue_scheduler.go:58-180) walks the queued gangs of the queues that are below their fair share in cost
   order, and FairnessOptimisingGangScheduler.Schedule (optimiser/gang_scheduler.go:45-254) places each one by scoring EVERY node (asched_optimiser_schedule_job's
   kernel), preempting the cheapest set of running jobs; its scheduled / preempted jobs are merged into the round's result (method ASCHED_METHOD_OPTIMISER).
   configuration.OptimiserConfig (configuration.go:517-537); NULL or enabled == 0: off (the default).  Applies to the following schedule_round calls.
   Node-uniformity groups are tried in ascending interned label value and remaining ties go to the earlier node id (the reference draws random ULIDs). */
typedef struct asched_optimiser_config {
  uint8_t enabled; uint8_t pad_[7];
  double min_fairness_improvement_pct;             /* MinimumFairnessImprovementPercentage */
  int32_t max_jobs_per_round; int32_t pad2_;       /* MaximumJobsPerRound */
  const int64_t* max_job_size_to_preempt;          /* [R] or NULL; 0 = no limit on that resource (node_scheduler.go:248-268) */
  const int64_t* min_job_size_to_schedule;         /* [R] or NULL */
  const double* max_resource_fraction_to_schedule; /* [R] or NULL = +Inf everywhere (MaximumResourceFractionToSchedule) */
  int64_t now_ms;                                  /* the clock job ages are taken from (asched_optimiser_schedule_job) */
} asched_optimiser_config;
int32_t ASCHED_FN(set_optimiser)(asched_t*, const asched_optimiser_config* cfg);

/* Market-driven scheduling of a pool (configuration.MarketSchedulingConfig; preempting_queue_scheduler.go:61-62): the following schedule_round calls
     - let the node evictor take EVERY job of the pool (pqs.go:117-119; cross-pool away jobs excepted, :102-104) — the prices decide who comes back;
     - order each queue's evicted jobs with jobdb.MarketSchedulingOrderCompare (pqs.go:292-295, jobdb/comparison.go:113-170) and merge them with the queued jobs by the
       same comparer (MarketDrivenMultiJobsIterator, jobiteration.go:232-321; the caller passes asched_queues.queued_jobs in that order: jobdb.PriceOrder);
     - pick the next queue with MarketBasedCandidateGangIterator (market_iterator.go: bid price, running before queued, round robin between queues at the same price ...),
       a literal container/heap because its Less reads the previous result; the queues are pushed in queue-index order (the reference ranges over a Go map);
     - set the spot price to the lowest bid of the gang that takes the DRF cost of what this pass has scheduled beyond spot_price_cutoff, mark what is in the queue
       contexts at that moment billable, and bill the price-setting queue the highest competing bid (queue_scheduler.go:177-203);
     - do not run the fairness optimiser (pqs.go:224).
   A fair-share preemption rate limiter together with market-driven scheduling is ASCHED_ERR_INVALID (the reference rejects it at config validation).  The runtime the
   reference compares between two running jobs (time.Now() - run.Created) orders like -run_timestamp: no clock is read.  NULL or enabled == 0: off (the default). */
typedef struct asched_market_config { uint8_t enabled; uint8_t pad_[7]; double spot_price_cutoff; } asched_market_config;
int32_t ASCHED_FN(set_market)(asched_t*, const asched_market_config* cfg);
/* What the last market-driven round left in the scheduling context: sctx.SpotPrice (has_spot_price 0: nil), per queue qctx.GetBillableResource() (floored at zero) and
   qctx.BillablePriceOverride (context/queue.go:39-44, 108-127).  Buffers are owned by the handle until the next round. */
typedef struct asched_market_outcome {
  int32_t has_spot_price; int32_t pad_;
  double spot_price;
  const int64_t* queue_billable_resource;      /* [q][R] */
  const double* queue_billable_price_override; /* [q] */
  const uint8_t* queue_has_price_override;     /* [q] */
} asched_market_outcome;
int32_t ASCHED_FN(market_result)(asched_t*, asched_market_outcome* out);

/* The indicative gang pricer of a market-driven pool (scheduling/pricer/gang_pricer.go:48-161, node_scheduler.go:41-146; called after the round by
   MarketDrivenIndicativePricer.Price, preempting_queue_scheduler.go:248-252, 641-664): the lowest price at which a gang could be scheduled on the NodeDb as it stands.
   Per member, in order, MinPriceNodeScheduler.Schedule scores EVERY node of the uniformity group (one wide pass, k_price_score): static requirements, fit at the
   evicted priority (price 0), else ALL jobs on the node except the gang's own members — whatever their priority class or preemptibility — ordered by (bid price,
   age, id) and preempted one at a time until the member fits: price = the last victim's bid.  The node with the lowest price takes the member (the first price-0
   node in id order wins outright; the reference breaks other ties by a random ULID, here the earlier node id), its victims are unbound and the member is bound
   (on a transaction that is aborted at the end: no side effects), and the gang's price is the highest member price; over the uniformity groups (label values in
   ascending interned order) the cheapest schedulable group wins.  `jobs` are rows of the job set (the synthetic jobs of a configuration.GangDefinition are uploaded
   like any other job, queued nowhere); bids are asched_jobs.bid_price; ages are now_ms minus the lease time.
   reason: 0, ASCHED_REASON_JOB_DOES_NOT_FIT / ASCHED_REASON_GANG_DOES_NOT_FIT, or one of the two below. */
#define ASCHED_PRICE_REASON_LABEL_NOT_INDEXED 101   /* "uniformity label is not indexed" (gang_pricer.go:18) */
#define ASCHED_PRICE_REASON_NO_NODES_WITH_LABEL 102 /* "no nodes with uniformity label" (:19) */
typedef struct asched_gang_price { int32_t evaluated, schedulable; double price; int32_t reason, pad_; } asched_gang_price;
int32_t ASCHED_FN(price_gang)(asched_t*, int32_t n, const int32_t* jobs, int64_t now_ms, asched_gang_price* out);
/* MinPriceNodeScheduler.Schedule (node_scheduler.go:41-107) for one job against every node (test hook for node_scheduler_test.go and the differential tests):
   per node scheduled / price / number of victims; for `detail_node` >= 0 also its victims in preemption order. */
typedef struct asched_price_node_score { int32_t scheduled, num_preempted; double price; } asched_price_node_score;
int32_t ASCHED_FN(price_job_on_nodes)(asched_t*, int32_t job, int64_t now_ms, asched_price_node_score* per_node /*[N]*/, int32_t detail_node,
                                      int32_t* preempted /*[cap]*/, int32_t cap);

/* 1 if the job's scheduling key is registered in sctx.UnfeasibleSchedulingKeys (gang_scheduler.go:80-95) */
int32_t ASCHED_FN(job_key_unfeasible)(asched_t*, int32_t job, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ARMADA_SCHED_H */
