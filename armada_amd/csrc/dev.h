// dev.h — device-side data layout of the MI355X scheduling-round implementation.
//
// Everything the round touches lives in HBM as flat SoA arrays (DESIGN.md "Data layout"):
//   alloc[P][R][Npad]  int64   AllocatableByPriority (internaltypes/node.go:64), one plane per (level, resource)
//   keys [P][Npad]     uint64  packed order key per level: rounded indexed columns | node-index rank
//                              (nodedb/encoding.go:37-54 restated as a single integer compare)
//   masks[..][W]       uint64  per-node bit masks (static requirement classes, job shapes, label values)
// plus the job table, per-queue accounting and the round's iterator state.
#pragma once
#include <stdint.h>

#define MAXR 8
#define MAXK 6
#define MAXP 16
#define MAXPC 32
#define LIT_TMAX 64     // node types one requirement class may match on the literal iteration path
#define MAXAWAY 32     // away node types over all priority classes
#define MAXE 2        // non-indexed resource columns the level-0 fast structure carries per entry
#define SMAX 256      // FIT shapes with a cached base candidate (LDS); scheduling-key shapes are unlimited (their unfeasible table lives in HBM)
#define L0CAP 1024    // live dirty nodes held in LDS
#define QCAPF 64      // queues the fast iteration handles (one lane per queue)
#define WIN 4         // job records prefetched per queue and refill
#define NO_PRIORITY INT32_MIN
#define NONPREEMPTIBLE_CUTOFF INT32_MAX

// per-job flag bits (qctx.Successful/Rescheduled/Unsuccessful/EvictedJobsById, context/queue.go:64-88)
#define F_SUCCESSFUL 1
#define F_RESCHEDULED 2
#define F_UNSUCCESSFUL 4
#define F_EVICTED 8

struct DevCfg {
  int R, K, P, npc;
  int32_t prios[MAXP];
  int32_t indexedCol[MAXK];
  int64_t indexedRes[MAXK];
  int64_t keyLo[MAXK];   // bias: field = alloc/res - keyLo
  int32_t keyShift[MAXK];
  int32_t keyWidth[MAXK];
  int32_t idxBits;
  int32_t keyGuard;      // 1: every key field has a spare zero bit above it (FastCfg.guardMask)
  int32_t keyClamp;      // 1: narrow field layout — keyLo = -1 and every quotient below it shares field 0 (asched_host.inc layoutKeys); 0: every reachable quotient has its own value
  int32_t pcPriority[MAXPC];
  uint8_t pcPreemptible[MAXPC];
  double drfMult[MAXR];
  uint8_t preferLarge, protectUncapped, disableHome, disableAway, disableGangAway, disableFair, disableUrgency, hasAway;
  uint8_t preferHome;    // preemptCrossPoolJobsFirst (queue_scheduler.go:744-746)
  uint8_t shardRank, shardWorld;   // shardWorld > 1 (asched_shard_round): this handle is one of shardWorld replicas of ONE pool on as many GPUs — every replica holds the whole state
                                   // and runs the whole round, but the round's wide passes over the nodes (plane scan, fair-share evaluation) look at this rank's share of the node
                                   // words only and their results are all-reduced (MIN / MAX) across the replicas.  (In pad bytes; served by armada_sched_wk.hip.)
  uint8_t padHome_[5];
  double protectedFraction;
  uint32_t maxLookback;
  uint8_t disallowed[MAXR];
  int N, Npad, W, M, Q, S, G, C;
  int64_t totalResources[MAXR];
  int64_t maxToSchedule[MAXR];
  int evLevel;  // level index of EvictedPriority (always 0)
  // away node types (nodedb.go:613-627): CSR per priority class; an entry whose well-known node type has no taints is skipped (:703-706)
  int32_t pcAwayOff[MAXPC + 1];
  int32_t awayPrio[MAXAWAY];
  uint8_t awayUsable[MAXAWAY];
  // floating resources (floatingresources/floating_resource_types.go:60-72): pool-level quantities that are not on nodes.  A floating column
  // never constrains a node (the node planes carry FLOATING_NODE_CAPACITY there); the gang scheduler checks sctx.Allocated against the limit
  uint8_t isFloating[MAXR]; uint8_t anyFloating, floatingConfigured; uint8_t pad_[6];
  int64_t floatingLimit[MAXR];
  // soft time budgets (constraints.go:159-169): 0 = off; clockStepNs > 0 = a stepping clock (testfixtures.SteppingClock), else the device's wall clock
  int64_t maxNewJobNs, maxNewJobPerQueueNs, clockStepNs;
  int32_t wallClockKHz;
  int32_t keyWords;      // 2: the order key is TWO words (asched_host.inc layoutKeys: more than 64 bits of fields).  `keys` then holds [2][P][Npad]: the high words of every level,
                         // then the low words (node-index rank in the low idxBits of the LOW word); keyShift[c] >= 64 names a field of the high word.  (In the slot of a pad word.)
};
// Node-sharded wide passes (shardWorld above): compiled into the CPU build and k_control_wk only.
#if defined(ASCHED_HOSTSIM) || defined(ASCHED_WK_TU)
#define ASCHED_SHARDED_PASSES 1
#define SHARD_ON(cfg) ((cfg).shardWorld > 1)
#define SHARD_LO(cfg) (SHARD_ON(cfg) ? (int)((long long)(((cfg).N + 63) >> 6) * (cfg).shardRank / (cfg).shardWorld) * 64 : 0)
#define SHARD_HI(cfg) (SHARD_ON(cfg) ? ((cfg).shardRank + 1 == (cfg).shardWorld ? (cfg).N : (int)((long long)(((cfg).N + 63) >> 6) * ((cfg).shardRank + 1) / (cfg).shardWorld) * 64) : (cfg).N)
#else
#define SHARD_ON(cfg) false
#define SHARD_LO(cfg) 0
#define SHARD_HI(cfg) ((cfg).N)
#endif
// the exchange words of a sharded pass live behind the cancel word in the handle's host-mapped block (armada_sched.hip PlatCtx.cancelHost, 256 bytes): 64-bit words
// [8] request generation, [9..10] this rank's two words, [11] answer generation, [12..13] the reduced words.  The kernel posts, the host thread that waits for the launch
// runs the all-reduce on the handle's communicator (RCCL over xGMI on a side stream, or the caller's transport) and answers.
#define XCHG_WORD0 8
// Two-word order keys are served by the generic path of a round kernel of their own (armada_sched_wk.hip: k_control_wk, k_bulk_wk): in every other device code object the
// test below is a compile-time `false`, so the one-word kernels carry none of it (their ISA is what it was); the CPU build of the tests decides per handle.
#if defined(ASCHED_HOSTSIM)
#define WIDE_KEYS(cfg) ((cfg).keyWords == 2)
#define ASCHED_TWO_WORD_KEYS 1
#elif defined(ASCHED_WK_TU)
#define WIDE_KEYS(cfg) ((cfg).keyWords == 2)   // (k_control_wk also serves one-word handles whose wide passes are sharded across GPUs: SHARD_ON)
#define ASCHED_TWO_WORD_KEYS 1
#else
#define WIDE_KEYS(cfg) false
#endif
#define FLOATING_NODE_CAPACITY ((int64_t)1 << 60)

// One job as the fast path reads it: a single 128-byte burst (16 lanes x 8 B) instead of 12 dependent array reads.
struct JobRec {
  int64_t req[MAXR];   // AllResourceRequirements (zero beyond R)
  uint64_t keyDelta;   // packed (req_c / resolution_c) per indexed column: what a bind subtracts from a node's order key
  uint64_t fieldMin;   // packed (req_c / resolution_c - keyLo_c): smallest key fields of a node the job fits on
  int32_t pc, shape, gang, node0, runPrio, cls, pcPrio;   // shape: the job's FIT shape (FastCfg.F), not its scheduling-key shape (Dev.jShape)
  uint8_t never, preemptible;
  uint8_t nlPc, nlRun; // number of priority levels a bind at pcPrio / at runPrio subtracts from (levels with priority <= cutoff, nodedb.go:1321-1334)
  int64_t ex0, ex1;    // requests on the (<= MAXE) non-indexed columns
};

// Queue-order inputs of one evicted job in its queue's eviction-order stream, computed for the whole stream in one bulk
// pass (round_run.h B_EVKEYS): the DRF costs updatePQItem would compute when the job becomes the queue's head.
struct EvKey { double proposed, current, size; int32_t pcPrio; int32_t job; };

// Queued-job streams (round_fast.h "stream run").  While every queued job fits without preemption the queue side of an iteration depends on
// nothing but the queue's own allocation prefix: the same costs, for the next QS_CMAX queued jobs of every queue, are computed ahead by a
// chunked prefix pass (round_run.h B_QSSUM -> B_QSSTITCH -> B_QSKEYS) into EvKey records; the control wave then merely merges the
// precomputed streams through its lane heap and stages job records for the node engine.
#ifndef QS_CMAX
#define QS_CMAX 32768         // stream entries per queue and preparation (round 6: one preparation covers a queue's rate-limit burst, so that a bulk-merged run — round_merge.h — seldom ends because a stream ran out)
#endif
#define QS_CHUNK 64           // entries one bulk item covers
#define QS_CPQ (QS_CMAX / QS_CHUNK)
// a queue's stream as of the last hand-over to the generic code (fastQFlush): taken up again at the next fastQLoad if the generic code left the queue alone
struct QsSave { int32_t valid, sPos, sLen, itQi, gctx, numUnfeasible; double tokens; int64_t alloc[MAXR]; };
struct QsIn { int32_t base, len, skipUnf, pad; int64_t a0[MAXR]; double weight; };   // per queue: position of element 0 in queuedJobs, wanted length, allocation + penalty before element 0

// What a node must offer for a job of one scheduling-key shape (the per-job JobRec fields that depend on the shape only).  With at most 64 shapes the fast
// structure keeps, per node entry, a bit per shape "a job of this shape fits here now" instead of re-deriving it from key fields, extras and class bits
// at every query (FastCfg.maskMode): a first-fit query then tests one bit per entry.
struct ShapeReq { uint64_t fieldMin; int64_t ex0, ex1; int32_t cls, never; };

// Level-0 ("fit without preemption", priority -2) fast structure, DESIGN.md "Sorted base + LDS delta".
struct FastCfg {
  int structOk, iterOk;       // host-verified exactness conditions (asched_host.inc: fastConditions); iterOk: 1 = fast iterations (Q <= QCAPF, round_fast.h), 2 = wide runs (Q > QCAPF, round_wide.h)
  int relocAll;               // ASCHED_RELOC_ALL=1: stage the per-queue arrays in LDS for any Q (default: only Q <= 64, see armada_sched.hip relocateIn)
  int cascadeFuse;            // the gate + urgency sweep of one job may run as ONE multi-level plane pass (round_ctl.h selectAtPriority): planes are monotone in the level (no explicit alloc_by_prio, non-negative requests) and a level tag fits above the packed key
  int F;                      // fit shapes: distinct (key fields, extras, requirement class) among the scheduling-key shapes — what node selection at priority -2 depends on;
                              // JobRec.shape, the candidate cache, the shape table and the fit masks are indexed by fit shape (scheduling keys that differ only in the
                              // priority class share one base cursor)
  int maskMode;               // <= 128 fit shapes (1: <= 64, 2: <= 128) and fit bitmaps on: the L0 list holds per-node fit masks over the fit shapes; was: baseCls / l0Cls / CandRec.cls hold per-shape fit masks (current capacity and requirement class folded in), not class bits
  int debugHang;              // tests only (ASCHED_DEBUG_HANG=<n>): the cold-set wave stops answering at its n-th command of a ring session, as a protocol defect would make it — every wait must still end
  int mgMin;                  // round_merge.h: a run of at least this many entries is merged by the bulk passes (MG_MIN_ENTRIES_DEFAULT; ASCHED_MERGE_MIN=<n> for soaks: small rounds through the bulk merge and the split node engine)
  int engineHc;               // the ring session of a bulk-merged stream run uses the split level-0 structure (engine_hc.h): hot set / cold set / clean front; ASCHED_ENGINE_HC=1 turns it on
  int engine;                 // queued-job iterations run on two waves (round_fast.h "two-wave iteration"); ASCHED_ENGINE=0 turns it off
  int E; int extraCol[MAXE];  // non-indexed columns
  uint64_t fieldMask[MAXK];   // in-place mask of each packed key field
  uint64_t minFieldMin;       // per-field minimum of fieldMin over all shapes (liveness of a dirty node)
  uint64_t guardMask;         // one spare (always zero) bit above every key field when the layout has room: "every field of a >= the same field of b" is
                              // then ONE subtraction — ((a | guards) - b) keeps a field's guard bit iff that field did not borrow (0 = no room: field loop)
  int64_t minExtra[MAXE];
};

// scalars of the scheduling context (context/scheduling.go:27-77) + kernel bookkeeping
struct RoundScalars {
  int64_t allocated[MAXR], scheduled[MAXR], evicted[MAXR];
  int32_t numScheduledJobs, numScheduledGangs, numEvictedJobs, terminationReason;
  double globalTokens; int64_t globalBurst; int32_t globalRateInf;
  int32_t hasFpLimiter; double fpTokens;
  int32_t loopIterations, numNodeQueries, numScans;
  int32_t error;            // first ASCHED_ERR_* raised on the device
  int32_t errorDetail;
  int32_t numEvictedList;   // length of the evicted list produced by an evictor kernel
  int32_t numUnfeasible;
  int32_t evictedTableSize; // Index counter of addEvictedJobsToNodeDb
  int32_t undoCount;
  int32_t txnActive;   // persists across control-kernel launches (NodeDb-level API)
  int32_t fairStamp;
  // ---- fast path (round_fast.h)
  int32_t fastActive;        // the sorted base + removed flags + L0 list describe the current level-0 state
  int32_t apiDirty;          // NodeDb-level commands changed per-job state since round_prepare: fast iterations off
  int32_t numPreemptedMarks; // |sctx.PreemptedJobIds|
  int32_t lvl0NonNeg;        // no node has a negative level-0 (priority -2) allocatable column
  int32_t l0SaveCount;       // L0 entries saved to HBM between launches
  int32_t fastOverflow;      // L0 overflowed: structure dropped for the rest of the round
  int32_t replayPending;     // the eviction-order replay (evicted-table Index assignment) has been deferred: nothing has read it yet
  int32_t statFastIters, statGenericIters, statScanSteps, statRefills, statL0Max, statFastReplay;
  int32_t statStreamRuns, statStreamJobs, statStreamPrepared, statStreamEmitted, statHybrid;   // stream runs (round_fast.h): runs, entries bound, stream entries prepared, entries emitted by the merge
  int32_t awayRowPlus1;      // an away attempt is in progress: static mask row (+1) that replaces the job's home shape row
  int32_t fairIndexValid;    // the per-node index of the evicted table (fairOff/fairEnt) describes the current table
  int64_t totalNewJobNs;     // sctx.TotalNewJobSchedulingTime (context/scheduling.go:212-240)
  int32_t optMode;           // the fairness optimiser's candidate iteration is running: a job popped from a queue keeps the failure reason of an earlier attempt until its new
                             // jctx is added to the scheduling context (qctx.addJobSchedulingContext drops the old one there, context/queue.go:235-237)
  int32_t ftWanted;          // this launch may build / use the threshold table (a scheduling pass of a round: set by the pass, cleared at kernel start)
  int32_t ftValid;           // the fair-share threshold table (round_ft.h) describes the current planes + evicted table (an upper bound per entry); cleared with fairIndexValid and at every launch
  int32_t statFt[3];         // threshold table: queries, validation retries, node updates
  int64_t statSeg[40];       // [24..39]: (profiling builds) segments of the generic iteration
  int64_t gsT;                     // (profiling builds) shader-clock ticks per segment of a fast iteration
  int64_t statClk[8];        // shader-clock ticks per phase of the round (device builds): evict, replay, pass 1, oversub evict, pass 2, unbind+results
#ifdef ASCHED_RS_PAD
  char rsPad_[ASCHED_RS_PAD];   // layout experiments only (tools/lds_layout_sweep.sh): this struct is copied into the round kernel's LDS
#endif
};

// the per-queue iterator / heap arrays a QueueScheduler-style loop owns; a second set lets the eviction-order replay run
// in the middle of a scheduling pass without disturbing it (lazy replay, round_run.h ensureReplay)
struct QueueLoopArrays {
  int32_t *itEi, *itQi, *itStage, *itJobsSeen, *itNext, *itStashed; uint8_t *itJobOnlyEv, *itGangOnlyEv, *onlyEvByQueue;
  double *pqProposed, *pqCurrent, *pqBudget, *pqSize; int32_t *pqPcPrio, *pqSchedPrio, *pqGctx; uint8_t* pqInHeap;
};

// ---- NumExcludedNodesByReason (asched_excluded_nodes): what the device keeps of a node selection that ended without a node.  The reasons that do not depend on the
// round's state (taints, selectors, affinity, total resources) are worked out by the host when somebody asks; the device records what only it knows — which nodes the
// iterator yielded at the job's priority (a bit per node) and, for the yielded nodes that pass the static checks, the first resource that did not fit and what was there.
struct ExclRec { int32_t job, level, row, uni, flags, gate, fair, pad; };   // flags bit 0: more dynamic reasons than the arena holds; bit 1: inconsistent (a yielded node that fits)
// gate > 0: the attempt FOUND node gate - 1 at the job's priority and still ended without a node (nodedb.go:747-789 with urgency preemption disabled): the walk that stays
// on record is the gate's, which stopped at that node — `bits` holds the yielded nodes that order BEFORE it; fair != 0: fair-share preemption ran in between, and `fbits`
// holds the nodes its walk over the evicted table found room on (the evicted jobs it may take, added in table order, cover the request): the ones among them that fail
// the static requirements were counted once more (:996-1006) — the host knows which
struct ExclDyn { int32_t slot, node, res, pad; int64_t avail; };
struct ExclDev {
  int32_t* jobSlot;      // [M] == Dev::excl: >= 0 the job's record of a failed attempt over the iterators; < 0 an EXCL_S_* code (the outcomes that need no record)
  ExclRec* rec;          // [cap]
  uint64_t* bits;        // [cap][W] nodes the iterator yielded
  uint64_t* fbits;       // [cap][W] gate-passed records: nodes the failed fair-preemption walk found room on
  ExclDyn* dyn;          // [dynCap] arena, tagged with the record
  int64_t* pinAvail;     // [M] what the pinned node had of the resource that did not fit (EXCL_S_PINNED0 - resource)
  int32_t cap, dynCap, W, cur;   // cur: the record the pass in flight fills
  int32_t count, dynCount;
};
#define EXCL_S_NONE (-1)          // nothing on record
#define EXCL_S_DROPPED (-2)       // more failed attempts than `cap` records
#define EXCL_S_DISALLOWED (-3)    // a disallowed resource was requested: every node, one reason (nodedb.go:596-601)
#define EXCL_S_NO_ATTEMPT (-4)    // no attempt was made (home scheduling disabled, no usable away type): every node implicit
#define EXCL_S_UNSUPPORTED (-5)   // not produced (more node types than the literal walk holds; an inconsistent record)
#define EXCL_S_PINNED0 (-8)       // - resource column: a pinned (evicted) job whose node no longer covers that column (nodedb.go:583-594, 897-920)
#define EXCL_HDR 256
#define EXCL(d) ((ExclDev*)((char*)(d).excl - EXCL_HDR))

// NodeTypeIterator state (nodeiteration.go:211-251): current lower bound (raw quantities), its packed form, the node it yielded last
#ifdef ASCHED_TWO_WORD_KEYS
struct LitIt { int64_t lb[MAXK]; uint64_t bound; int32_t head; int32_t type; uint64_t boundLo; };   // (two-word keys: `bound` is the high word; the host allocates LITIT_BYTES per iterator)
#else
struct LitIt { int64_t lb[MAXK]; uint64_t bound; int32_t head; int32_t type; };
#endif
#define LITIT_BYTES (sizeof(int64_t) * MAXK + 8 + 4 + 4 + 8)

// ---- wide runs (round_wide.h): stream runs for pools of more than QCAPF queues.  Per queue a stream of at most WIDE_L entries — its remaining cheap evicted jobs, then
// its next single queued jobs — with precomputed queue-order keys; the k-way merge of QueueCandidateGangIteratorPQ over them is a BULK RANK (every entry counts, by binary
// search in every other queue's monotone key sequence, the entries that order before it) instead of a lane per queue.
#define SKIP_BULK_MIN 2048    // Peek's skip of known-unfeasible keys: from this many jobs on as two bulk passes on every workgroup (round_wide.h) instead of 64 per step on the control wave
#define SG_TMAX 256           // submit check, gang units one workgroup each (submit_gang.h): members per unit (nodes the unit's scratch can hold)
#define FIT_OSTR 16          // k_fit_batch's result words are 128 bytes apart (one cache line per shape: the words of neighbouring shapes shared lines, and every wave's look at its word queued up behind the others in ONE L2 channel)
#define WIDE_L 1024
struct WideSeg { int32_t evStart, evCnt, qBase, qLen, flags, total, qWant, pad; };   // flags: 1 stream, 2 barrier (a head the wide run cannot serve: its key stops the merge), 4 open (the queue goes on behind its last entry under a key not known here), 8 element 0 of the queued part is the peeked head
struct WideKey { uint64_t a, x, y; };               // running maximum of the packed queue-order keys up to an entry (an entry is never served before its predecessor)
struct WideEnt { int32_t job, qk; };                // qk = queue | 1 << 30 for an evicted job returning to its node
struct WideParams { int32_t evOk, queuedOk, skipUnf, preferLarge, cap, numEvictedList, replayPending, executed; uint32_t maxLookback; int32_t noNew, withQueued, pad; };
struct WideDev {
  WideSeg* seg;        // [Q]
  WideKey* key;        // [Q * WIDE_L] COMPACT: queue q's entries at off[2q] .. + off[2q + 1] (the rank pass walks every queue's keys for every entry: 256 arrays 24 KB apart cost a TLB / cache
                       // line walk per step — measured 2.4x slower than a 3 KB stride; compact, a run's keys are a few hundred KB in one piece)
  int32_t* off;        // [Q][2] start and number of the queue's entries in the compact arrays
  int32_t* own;        // [Q * WIDE_L] queue of a compact entry
  WideKey* cmax;       // [Q][WIDE_L / 8] maximum of each chunk of a queue's keys (the running maximum is stitched across the chunks)
  int64_t* part;       // [Q * WIDE_L / 8][MAXR + 2] chunk sums -> carries of the queued part's requests, first barrier of the chunk
  int32_t* rank;       // [Q * WIDE_L] position of every (compact) entry in the merged order
  WideEnt* merged;     // [Q * WIDE_L + Q]
  int32_t* cnt;        // [2Q] entries executed per queue: evicted, queued
  int32_t* cap;        // [Q] entries to prepare for the queue in the next run: follows what the queue consumes (queues advance at very different rates under DRF)
  int64_t* tot;        // [3 * MAXR]: requests of the executed queued entries, of the executed evicted entries; [2 * MAXR + 0 / 1] their counts
  uint32_t* stop;      // [2] first position the merged order is NOT valid at (atomic min); number of stream entries
  WideParams* par;
};

// ---- bulk-merged stream runs (round_merge.h): the k-way heap merge of the <= QCAPF queues' precomputed streams computed as a BULK RANK on every workgroup (the wide runs' idea,
// round_wide.h W_RANK, on the stream representation of round_fast.h); the control wave then only stages the merged order for the node engine.
struct MgQ {   // one queue of the heap as the merge sees it, written by the control wave before the passes (HBM: the helper workgroups read it)
  int32_t start, len, kind, base;      // stream elements [start, len) are still to come; kind bit 0: evicted stream (d.evKey[base + e]), else queued (d.qsKey[q][e]); bit 1: folded (skip mode)
  int32_t flags, off, total, nameRank; // flags 1: stream, 2: barrier (a head the run cannot serve: ONE entry under its heap key), 4: open (the queue goes on behind its last element under a key not known here), 8: skip mode (keys must not decrease)
  double budget; int64_t pad_;
  WideKey eff, head;                   // skip mode: the running maximum the queue's keys start from; the heap key of a barrier head
};
struct MgEnt { int32_t job, qk, e, ci; };   // merged position -> job, queue | 1 << 30 for an evicted job, stream position, compact index (its key: MgDev.key[ci])
struct MgDev {
  MgQ* q;            // [QCAPF]
  WideKey* key;      // [cap] running-maximum packed keys, compact: queue q's entries at q.off .. + q.total
  int32_t* own;      // [cap] queue of a compact entry | 1 << 30 when its own key orders before the running maximum in front of it
  int32_t* rank;     // [cap] position in the merged order
  MgEnt* merged;     // [cap]
  WideKey* cmax;     // [QCAPF * MG_CPQ] maximum of each chunk of a queue's keys
  uint32_t* stop;    // [4] first merged position that is NOT valid (atomic min); total entries; W_MG_CUT: entries the run can serve at most; 1 = the cut is on
  WideKey* cut;      // [1] W_MG_CUT: K* — entries above it are left out of this run's merged order
  uint32_t* cutPick; // [1] W_MG_CUT: (samples at or below the key) << 16 | sample, atomic min over the samples that cover the need
  int32_t cap, pad;
};
#define MG_CHUNK 64
#define MG_CPQ (QS_CMAX / MG_CHUNK)

struct Dev {
  DevCfg cfg;
  // ---- nodes
  int64_t* alloc;        // [P][R][Npad]
  uint64_t* keys;        // [P][Npad]
  int64_t* totalRes;     // [R][Npad]
  int64_t* allocatable;  // [R][Npad]
  int64_t* alloc0;       // [P][R][Npad] explicit initial state (alloc_by_prio) or NULL
  uint8_t* nodeFlags;    // [N] bit0: unschedulable && overAllocated
  int32_t* idxRank;      // [N] rank of node.index
  int32_t* nodeByRank;   // [N]
  // ---- masks
  uint64_t* shapeMask;   // [S][W] static class ∧ node-type match ∧ total >= req, per scheduling-key shape
  uint64_t* labelMask;   // [L][W] nodes carrying (uniformity label == value)
  // ---- jobs (immutable per jobs_set)
  int32_t *jQueue, *jPc, *jShape, *jGang, *jGangCard, *jGangUni, *jNode0, *jRunPrio, *jRankActive, *jRankInactive;
  uint8_t* jAway;   // [M] cross-pool away jobs (asched_jobs.away); null = none
  int64_t* jReq;         // [M][R] row-major (control path reads one job = one 8*R byte burst)
  uint8_t* jAligned;     // [M] request is a multiple of the index resolution on every indexed column
  int32_t *gangOff, *gangJobs;  // CSR of (queue,gang) -> member jobs (jobRepo.GetGangJobsByGangId)
  int64_t* shapeReq;     // [S][R]
  // literal node iteration (round_ctl.h selectAtLevelLiteral) for mask rows whose merged iteration order is not the packed-key order
  uint8_t* rowLiteral;   // [S + away rows]
  int32_t* rowTypeOff;   // [S + away rows + 1] CSR: populated node types matching the row's requirement class
  int32_t* rowTypes;
  uint64_t* typeMask;    // [T][W] nodes of each node type
  int32_t* nodeIdRank;   // [N] lexicographic rank of the node id (nodeTypesIteratorPQ.less tie-break)
  struct LitIt* lit;     // [LIT_TMAX] per-type iterator state of the query in progress
  int32_t* awayRowOff;   // [S+1] rows S + awayRowOff[s] + k of shapeMask: shape s with the tolerations of its class's k-th away node type added
  // ---- job dynamic state
  int32_t* schedAtPrio;  // [M] nodeDb.scheduledAtPriorityByJobId
  int32_t* jobNode;      // [M] node the job currently owns resources on (AllocatedByJobId), -1
  int32_t* jobCutoff;    // [M] cutoffByJobId
  uint8_t* jobEvictedOnNode;  // [M] EvictedJobRunIds
  uint8_t* jobFlags;     // [M]
  // jctx (one live JobSchedulingContext per job)
  uint8_t* jcEvicted; int32_t* jcAssigned; int32_t* jcReason; uint8_t* jcHasPctx;
  int32_t *pcNode, *pcSap, *pcPap, *pcMethod;
  int32_t* jcGangCard; uint8_t* jcPreempted;  // sctx.PreemptedJobIds
  int32_t* jcUniValue;   // additional node selector (uniformity label value mask id), -1
  int32_t* jcStagedBy;   // staged preemption (applied on txn commit)
  // result bookkeeping of PreemptingQueueScheduler.Schedule
  uint8_t* inPreempted; uint8_t* inScheduled; uint8_t* inSchedAndEvicted; int32_t* preemptedNode;
  // ---- queues
  double *qWeight, *qFair, *qDc, *qUc, *qTokens; int32_t* qNameRank; int64_t* qBurst; uint8_t *qRateInf, *qCordoned;
  int64_t *qAlloc, *qAllocByPc, *qSchedByPc, *qEvictedByPc, *qPenalty, *qPcLimit, *qDemand; int32_t hasPcLimit;
  int64_t* qDemandByPc;   // [Q][npc][R] scratch of the round-input builder on the device (round_run.h B_AGG_*): demand per queue and priority class before the per-class cap
  int32_t *queuedOff, *queuedJobs;
  // ---- evicted jobs
  int32_t* evList;       // [M] output of evictor kernels, then sorted by (queue, scheduling order)
  uint32_t* evSortKey;   // [M]
  int32_t* evOff;        // [Q+1] per-queue segment of the sorted list
  int32_t* evTabJob;     // [M] evicted table: Index -> job (EvictedJobsTable, nodedb.go:1257-1281)
  uint8_t* evTabAlive;   // [M]
  int32_t* evIndexOfJob; // [M] job -> Index or -1
  // ---- iterators (one set per pass)
  int32_t *itEi, *itQi, *itStage, *itJobsSeen, *itNext, *itStashed; uint8_t *itJobOnlyEv, *itGangOnlyEv, *onlyEvByQueue;
  int32_t* gangSeen; int32_t* gangArr; int64_t* gangTotal; uint8_t* gangAllEvicted;
  double *pqProposed, *pqCurrent, *pqBudget, *pqSize; int32_t *pqPcPrio, *pqSchedPrio, *pqGctx; uint8_t* pqInHeap;
  int64_t* replayAlloc;  // [Q][R] MinimalQueueRepository allocation (pqs.go:552-585)
  uint8_t* unfeasible; int32_t* unfeasibleReason;  // [S]
  // ---- fair preemption scratch
  int64_t* accAvail;     // [N][R]
  int32_t* accStamp;     // [N]
  uint8_t* accStaticFailed;  // [N]
  int32_t* excl;         // [M] failed node selections on record (asched_excluded_nodes; round_wide.h "excluded nodes"): -1 = none, -2 = dropped, else the job's record; NULL =
                         // recording is off.  The store's header (ExclDev) sits EXCL_HDR bytes in front of it — one pointer here, and the hot path (a job that gets a node
                         // forgets its record) is one store with no load in front of it.  (In the slot of a 4-byte field + its padding: sizeof(Dev), which the round
                         // kernel copies into its LDS, is unchanged.)
  // per-node index of the evicted table for fair-share preemption (round_run.h ensureFairIndex): CSR node -> table Indexes, descending
  int32_t* fairOff;      // [Npad+2]
  int32_t* fairEnt;      // [M] evicted-table Index
  int32_t* fairEntJob;   // [M] its job
  int32_t* fairPart;     // [FAIR_CHUNKS+1] chunk sums of the offset scan
  // fair-share threshold table (round_ft.h): T[s][n] = fairNodeBest of scheduling-key shape s on node n, its maxima per 64 / 4096 nodes; NULL = not in use
  int32_t* ftT;          // [ftS][Npad]
  int32_t* ftB1;         // [ftS][ftNB1]
  int32_t* ftB2;         // [ftS][64]
  int32_t* ftPrio;       // [ftS] priority a job of the shape asks with (its priority class's)
  int32_t ftS, ftNB1;
  uint8_t *optSched, *optPre;   // [M] how often the fairness optimiser scheduled / preempted a job in this round: its result lists are merged into the round's at the END
                         // of its phase (pqs.go:232-249), where a job it scheduled, preempted, scheduled again and preempted again comes out preempted
  int32_t* optGhost;     // [M] -1, or the node on which a job the optimiser has bound elsewhere STILL holds its evicted resources: scheduled earlier in the round, evicted by the
                         // oversubscribed evictor, not rescheduled — the reference keeps it in that node's AllocatedByJobId until the unbinding at the end of the round
  double* optQDelta;     // [Q] or NULL: QueueContext.CurrentCost changes of the gang members placed so far in the optimiser's current gang (optimiser/gang_scheduler.go:182-188)
  // ---- txn undo log
  int32_t* undo;         // [cap][4]
  int32_t undoCap;
  int32_t accEpoch_unused;   // fair-share index: queries since the last build (round_run.h ensureFairIndex).  (Moved into undoCap's padding in round 4: its old slot + padding hold `excl`.)
  // ---- misc
  RoundScalars* rs;
  uint64_t* scanResult;  // [8] HBM scratch words of bulk passes that need one ([0]: the first job of a skip stretch that is NOT skipped, round_wide.h W_SKIP_FIND)
  int32_t* nodeOver;     // [N] bitmask of oversubscribed levels (phase 3)
  uint8_t* evFlag;       // [M] job selected by the current evictor
  uint8_t* qEvictable;   // [Q] queue is above protectedFractionOfFairShare (pqs.go:124-134)
  int32_t *ordAll, *ordAllOff;  // jobs pre-sorted by (queue, SchedulingOrderCompare): active segment then queued segment per queue
  int32_t* uniOff;       // [slots+1] label-value mask ids per uniformity label slot
  int32_t* preList;      // [M] staged preemptions
  int32_t* cmdIO;        // [64 + ...] command arguments / results of the control kernel
  int32_t *resJob, *resNode, *resPrio, *resMethod, *resPreJob, *resPreNode;  // compacted results
  // ---- level-0 fast structure (HBM side; L0 + candidate cache live in LDS)
  FastCfg f;
  uint64_t* baseKey;     // [Npad] level-0 keys in ascending order as of the last build
  int32_t* baseNode;     // [Npad]
  int64_t* baseExtra;    // [MAXE][Npad]
  uint64_t* baseCls;     // [Npad] static requirement classes the node satisfies (bit per class)
  uint8_t* baseRemoved;  // [Npad] entry is stale: the node changed since the build (it is in L0 or dead)
  int32_t* posOf;        // [N] node -> base position
  int32_t* l0Slot;       // [N] node -> L0 slot or -1
  uint64_t* nodeCls;     // [N]
  ShapeReq* shapeTab;    // [F] per fit shape
  uint64_t* fitBits;     // [F][fitW] bit p of row f: base entry p is clean (unchanged since the sort) and a job of fit shape f fits on it — a base rescan is a find-first-set
  int32_t fitW, fitPad_;  // words per row = ceil(N / 64)
  JobRec* jrec;          // [M]
  int32_t* evIdxByPos;   // [M] evicted-table Index of evList[p]
  EvKey* evKey;          // [M] per evicted-list position (queues with evCheap)
  uint8_t* evCheap;      // [Q+1] the queue's evicted stream has precomputed keys (no gang members)
  int64_t* evPart;       // [evChunks][2*MAXR+4] partial request sums of the chunked prefix pass (B_EVSUM)
  uint8_t* evMono;       // [Q+1] the queue's evicted stream has non-decreasing queue-order keys (heap merge == sort by key)
  uint64_t* evEdge;      // [evChunks][8] first / last packed key of each chunk (monotonicity across chunk borders)
  int32_t evChunks;
  EvKey* qsKey;          // [QCAPF][QS_CMAX] precomputed costs of the queued-job streams
  QsIn* qsIn;            // [QCAPF]
  union {
    QsSave* qsSave;      // [QCAPF]  (pools of at most QCAPF queues: stream runs of the lane heap, round_fast.h)
    struct WideDev* wide;  // pools of MORE than QCAPF queues (f.iterOk == 2): the arrays of the wide runs (round_wide.h) — the same slot, so that sizeof(Dev), which the
                           // round kernel copies into its LDS, does not change with the feature (placement-sensitive)
  };
  int64_t* qsPart;       // [QCAPF * QS_CPQ][MAXR + 2] chunk sums -> carries, first barrier of the chunk
  int32_t* qsLen;        // [QCAPF][2] usable stream length, 1 = the queue's list ends with the stream
  int32_t* l0Save;       // [L0CAP]
  int32_t* candPosSave;  // [SMAX]
  const struct FastK* fk; // the fast loop's constants (round_fast.h), filled by the host at round_prepare
  QueueLoopArrays alt;    // second set for the lazy replay
  int64_t* qAllocSnap;    // [Q][R] queue allocations right after an evictor ran: what addEvictedJobsToNodeDb starts from
  int64_t* jLeaseMs;      // [M] lease time of the active run in ms (run_timestamp / 1e6): job ages of the fairness optimiser
  int64_t* qNewJobNs;     // [Q] qctx.TotalNewJobSchedulingTime
  MgDev* mg;              // bulk-merged stream runs (round_merge.h); NULL = off (ASCHED_MERGE=0, more than QCAPF queues)
  volatile int32_t* cancel;    // host-mapped word: != 0 = the caller's context is done (hard timeout / cancel, queue_scheduler.go:105-112); NULL = never
  volatile int32_t* progress;  // optional host-visible heartbeat (ASCHED_PROGRESS=1): [0] loop iterations, [1] phase, [2] current wide op, [3] wide ops issued
#ifdef ASCHED_DEV_PAD
  char devPad_[ASCHED_DEV_PAD];   // layout experiments only (tools/lds_layout_sweep.sh): this struct is copied into the round kernel's LDS
#endif
};

// Market-driven rounds (asched_set_market; round_mkt.h).  Deliberately NOT part of Dev / RoundScalars: those two are copied into the round kernel's LDS, and the round
// kernel's code object must not change with a feature it does not run (placement-sensitive; tools/kcontrol_isa_hash.sh).  The auxiliary kernel receives this
// struct as a kernel argument of its own.
struct MktScalars {   // MarketIteratorPQ's previous result, sctx.SpotPrice, QueueScheduler's scheduledResource
  int32_t market, heapN, prevRank, hasSpotPrice;
  double prevCost, spotCutoff, spotPrice;
  int64_t schedRes[MAXR];
};
struct MktDev {
  MktScalars* s;
  // bid per job, the pool-wide rank of every job under jobdb.MarketSchedulingOrderCompare and the per-queue job order sorted by it
  double* jBid; int64_t* jSubmit; int64_t* jRunTs; int32_t* jRank; int32_t* ord;
  // MarketIteratorPQ (items = queues; heap[] is pq.items), MarketDrivenMultiJobsIterator's two held values per queue
  int32_t* heap; double* pqPrice; int64_t* pqRuntime; int64_t* pqSubmit; uint8_t* pqQueued;
  int32_t* itV1; int32_t* itV2;
  // billing
  int64_t* qBillable; double* qOverride; uint8_t* qHasOverride; uint8_t* jobBillable;
};
