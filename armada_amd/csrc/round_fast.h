// round_fast.h — the fast path of the scheduling round (DESIGN.md "Fast path").
//
// The generic control code (round_ctl.h) restates the reference statement by statement with every piece of state behind
// generic pointers in HBM: ~100 dependent memory round trips per QueueScheduler iteration and one O(N) scan of the
// allocatable planes per node selection.  This file removes both for the overwhelmingly common iteration — a single
// (non-gang) job that is either a queued job fitting without preemption or a phase-1-evicted job returning to its node —
// and hands anything else to the generic code, which then sees exactly the state it would have produced itself.
//
//  1. Level-0 fast structure (first fit at priority -2, nodedb.go:737 → 840-879).  Nodes are kept in HBM sorted by
//     their level-0 order key as of round_prepare ("base", == the reference's memdb index, nodedb.go:1164-1175).
//     A node whose allocatable changes is flagged removed in the base and, while it can still host some shape, lives in
//     an LDS list ("L0").  First fit = min(first clean feasible base entry from a per-shape cursor, min over L0).  Both
//     candidates are exact: clean entries are unchanged since the sort, L0 holds current values, and keys are unique.
//  2. One LDS record per queue (QRec: iterator + cost state, array-of-structs so that one burst of wide LDS reads
//     fetches it), the head job record and a small prefetch window per queue; a job is one 128-byte JobRec burst.  While
//     fast iterations run these are the authoritative copy of that state; they are written back to the generic arrays
//     before any generic code runs (fastEnterGeneric) and re-read afterwards (fastEnsureLive).
//  3. DRF cost (fairness.go:99-105) evaluated across lanes: the 3 x R float64 divisions of one updatePQItem
//     (queue_scheduler.go:636-686) issue together; same IEEE operations, same results.
//  4. Queue selection by a packed total-order key equivalent to QueueCandidateGangIteratorPQ.Less
//     (queue_scheduler.go:738-798) for finite costs.
//  5. Binds and accounting as no-return HBM atomics / plain stores with explicit global address space (node.go:416-442
//     arithmetic): nothing waits on them; LDS traffic uses ds_* instructions, so the two never serialise on each other.
//  6. The loop (fastRun) is a separate, non-inlined function whose constants (FastK) and scheduling-context scalars
//     (FastS) live in registers: an iteration is ~4 dependent LDS round trips instead of ~150.
//
// Lane-parallel primitives have a device (armada_sched.hip) and a host (tests/hostsim) implementation, same contract.
#pragma once
#include "round_ctl.h"

// Per-queue scalars of CostBasedCandidateGangIterator + QueuedGangIterator (authoritative in fast mode).  The fast loop
// reads the fields it needs of one queue (independent LDS reads, one round trip) and stores back only what it changes.
struct alignas(16) QHot {
  double weight, tokens, budget, proposed, current, size;
  int64_t burst;
  int32_t itEi, itQi, itStage, itJobsSeen, itNext, gctx, evEnd, qEnd, pcPrio, schedPrio;
  int32_t rateInf, cordoned, itJobOnlyEv, itGangOnlyEv;
  int32_t headFast, headKind, headIdx;      // head job: record cached in head*, 0 evicted / 1 queued, evicted-table Index
  int32_t winKind, winStart, winCount;      // prefetch window: stream (0 evicted list, 1 queued list) and position range
  int32_t evCheap, evApplied, evDone, ewStart, ewCount, headPos;  // evicted stream with precomputed keys: served up to evDone, commits applied up to evApplied; key window; stream position of an evicted head
  int32_t effValid, skipStart;  // skip mode: the queue's evicted stream [skipStart, evEnd) was folded out of the loop; its head key is a running maximum
  int32_t sPos, sLen;           // stream run: the queue's head is element sPos of its precomputed stream of sLen entries (0: no stream)
};
static_assert(sizeof(QHot) == 176, "QHot: the stream fields use the tail padding");
struct alignas(16) JobTail {  // second half of a JobRec
  uint64_t keyDelta, fieldMin;
  int32_t pc, shape, gang, node0, runPrio, cls, pcPrio;
  uint8_t never, preemptible, nlPc, nlRun;
  int64_t ex0, ex1;
};
static_assert(sizeof(JobTail) == 64 && sizeof(JobRec) == 128 && __builtin_offsetof(JobRec, keyDelta) == 64, "JobRec = request vector + JobTail");
struct alignas(16) CandRec { int32_t pos, node; uint64_t key, cls; int64_t ex0, ex1; int64_t pad; };  // node: >=0, -1 exhausted, -2 scan from pos

// ---- two-wave iteration.  A queued-job iteration has a queue side (constraints, accounting, the queue's next head, DRF costs,
// heap) and a node side (first fit at priority -2, bind, per-job results, L0 upkeep) that meet only in "did the job fit".  The
// control wave posts the job to a second wave (the node engine) and carries on with the queue side assuming it fits; the verdict
// is collected before the next iteration starts.  On a miss the queue side of that one iteration is taken back (IterBackup) and
// the generic code runs it from exactly the state it would have found.
enum { ENG_JOB = 1, ENG_QUIT = 2, ENG_STREAM = 3, ENG_STREAM_HC = 4 };   // ENG_STREAM_HC: a ring session on the split level-0 structure (engine_hc.h)
struct alignas(16) EngineBox {
  JobTail tail; int64_t req[MAXR];          // the job: its record as fastIter read it
  int32_t job, prio, cutoff, nl, cmd, status;
  int32_t seq, ack;                          // command n is ready when seq == n; served when ack == n
  int32_t statScan, statL0Max;               // engine counters, handed over at ENG_QUIT
  int64_t busyClk; int32_t jobs, cancel;     // shader-clock ticks the engine spent serving jobs, and how many; cancel: the engine saw the caller's cancel word
  int32_t ringPub, ringAck, ringEnd, ringFail;   // stream run: entries staged in the ring / placed by the engine; no more entries will come; 1 = entry ringAck found no node, 2 = placed but L0 overflowed
  int32_t bindHold;   // 1: the bind wave waits for the verdict on the whole ring (a gang: all members or none), 2: go, 3: discard
  int32_t abandon, idleProg, idleSince, idleLast;   // a bounded wait gave up (the caller's cancel word, or the tick budget): every other wait of the launch gives up too; streamIdle's stretch of waiting without progress (armada_sched.hip)
  int32_t hcGen, cleanFrom;   // engine_hc.h: generation of the cold wave's sessions; no clean base entry lies before this position (kept for the launch)
  int32_t ringClosed, bindGen, bindDone, bindFin, bindQuit, live;   // live: the node engine runs (engineStart .. engineStop) — what coldS tells the out-of-line helpers
  //   // the engine has left the ring; stream generation / entries whose bind + result fields the bind wave has issued / generation it has finished
};
// stream run: the job-record windows are idle and serve as the ring between the control wave and the node engine
#define RING_N (QCAPF * WIN)
#define RREC(i) (((JobRec*)FL.winRec)[(i) & (RING_N - 1)])
#define RJOB(i) (((int32_t*)FL.winJob)[(i) & (RING_N - 1)])
#define RQ(i) (((int32_t*)FL.winIdx)[(i) & (RING_N - 1)])
#define RQ_EV 0x100   // ring entry of an evicted stream: nothing for the engine to do
#ifndef MG_MIN_ENTRIES_DEFAULT
#define MG_MIN_ENTRIES_DEFAULT 2048   // (round_merge.h MG_MIN_ENTRIES)
#endif
static_assert((RING_N & (RING_N - 1)) == 0, "ring size");
struct alignas(16) IterBackup {  // the job's record and request stay in the mailbox (the engine only reads them)
  QHot hot;
  uint64_t kX, kY, effX, effY; double globalTokens;
  uint32_t kA, effA; int32_t pc, inHeap;
};

struct FastLds {
  // L0: live dirty nodes (current level-0 key / non-indexed columns / class bits)
  int l0Count;
  uint64_t l0Key[L0CAP]; int32_t l0Node[L0CAP]; int64_t l0Ex0[L0CAP], l0Ex1[L0CAP]; uint64_t l0Cls[L0CAP], l0Cls2[L0CAP];   // class bits, or (mask mode) fit masks over fit shapes 0-63 / 64-127
  CandRec cand[SMAX];   // per-shape base cursor + validated candidate
  QHot hot[QCAPF];
  int64_t qAlloc[QCAPF][MAXR], qPenalty[QCAPF][MAXR], qReplay[QCAPF][MAXR];  // resource vectors: one lane per resource
  int64_t headReq[QCAPF][MAXR]; JobTail headTail[QCAPF];
  int32_t winJob[QCAPF][WIN]; int32_t winIdx[QCAPF][WIN];
  JobRec winRec[QCAPF][WIN];
  EvKey evWin[QCAPF][WIN];
  // queue order: packed keys + heap membership + name rank, one lane per queue
  uint32_t kA[QCAPF]; uint64_t kX[QCAPF]; uint64_t kY[QCAPF]; int32_t inHeap[QCAPF]; int32_t nameRank[QCAPF];
  uint32_t tmpA[64], tmpN[64]; uint64_t tmpX[64], tmpY[64]; int32_t tmpQ[64];  // scatter space of pqBuild
  uint32_t effA[QCAPF]; uint64_t effX[QCAPF], effY[QCAPF];  // skip mode: running maximum of the queue's keys
  EngineBox eng; IterBackup bk;
  uint8_t sKind[QCAPF];   // stream of queue q: 0 queued jobs (d.qsKey), 1 its evicted jobs (d.evKey)
  // evicted-stream repair (evRepair): the requests of up to 64 consecutive evicted jobs of one queue (0 for a job that carries a preempted mark), and per queue the
  // stream position below which the queue's precomputed costs are valid for its CURRENT allocation prefix
  int64_t evStage[64][MAXR]; int32_t evValidEnd[QCAPF];
};
#define EV_EXCL(d) ((d).evCheap + 4104)   // [M] per evicted-list position: the repair that produced its costs found the job marked preempted and left it out of the prefix

#ifdef ASCHED_HOSTSIM
static FastLds g_fl;
#define FLANE 0
#define GA(T, p) (p)   // a pointer known to address HBM (explicit global address space on the device)
#define GP(T) T*
#define KREF const FastK&   // the loop constants: scalar (constant address space) loads on the device
#define HD static inline
#define RS (*d.rs)
#define FOR_LANES(i, n) for (int i = 0; i < (n); i++)
#define LDS_ADD64(ref, v) ((ref) += (v))
#define DEV_NOINLINE static __attribute__((noinline))
#else
__shared__ FastLds g_fl;
__shared__ RoundScalars g_rs;  // d.rs points here for the whole launch (relocateIn, armada_sched.hip)
#define FLANE ((int)(threadIdx.x & 63))
#define HD __host__ __device__ static inline
#if defined(__HIP_DEVICE_COMPILE__)
#define GA(T, p) ((__attribute__((address_space(1))) T*)(p))
#define GP(T) __attribute__((address_space(1))) T*
#define KREF const FastK&
#else  // host pass of hipcc: same layout, plain pointers (the host fills FastK for upload)
#define GA(T, p) (p)
#define GP(T) T*
#define KREF const FastK&
#endif
#define RS g_rs
#define FOR_LANES(i, n) for (int i = FLANE; i < (n); i += 64)
#define LDS_ADD64(ref, v) ((void)__hip_atomic_fetch_add(&(ref), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))  // ds_add_u64, no return: nothing waits
#define DEV_NOINLINE __device__ static __attribute__((noinline))
#endif
#define FL g_fl
#ifdef ASCHED_HOSTSIM
#define HS_POISON(x) memset((void*)&(x), 0xA5, sizeof(x))   // the CPU build: a local the device keeps in registers starts as garbage there, not as whatever the host stack holds
#else
#define HS_POISON(x) do {} while (0)
#endif

// A value every lane of the control wave holds identically, moved to scalar registers: branches on it become scalar
// branches (no exec-mask bookkeeping) and arithmetic on it runs on the scalar unit.
#if defined(ASCHED_HOSTSIM) || !defined(__HIP_DEVICE_COMPILE__)
#define UNI32(x) (x)
#define UNI64(x) (x)
#define UNID(x) (x)
#define LANES_ANY(v) ((v) != 0)   // (serial build: FOR_LANES bodies share their locals)
#define FAST_GLOBAL_FENCE() do {} while (0)
#define LANE0_PUBLISHED() do {} while (0)
#else
#define LANES_ANY(v) (__ballot((v) != 0) != 0)
#define FAST_GLOBAL_FENCE() __threadfence()
// After a store to an LDS word that only SOME lanes execute (if (lane == 0) word = v;) and before the wave reads that word again: a wavefront-scope fence.
// It emits no instruction (LDS executes one wave's accesses in issue order) but it is what makes the read legal for the compiler: in the single-thread view
// of a lane that did not store, nothing wrote the word, so without the fence LLVM may reuse a value loaded BEFORE the store (GVN / load PRE / LICM) and the
// lanes disagree.  profiles/r04a_lds_handback_rootcause.txt has the ISA of the failing idiom; tests/lds_idiom pins it.  (Reads through UNI32 / readfirstlane
// take lane 0's value, which is the stored one either way: that is why the sites below were right before the fence was spelled out.)
#define LANE0_PUBLISHED() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
__device__ static inline int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ static inline unsigned long long uni64(unsigned long long v) {
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <class T> __device__ static inline T uniT32(T v) { return (T)uni32((int)v); }
template <class T> __device__ static inline T uniT64(T v) { return (T)uni64((unsigned long long)v); }
#define UNI32(x) uniT32(x)
#define UNI64(x) uniT64(x)
#define UNID(x) (__builtin_bit_cast(double, uni64(__builtin_bit_cast(unsigned long long, (double)(x)))))
#endif

#ifdef ASCHED_FASTPROF
#define ESEG(i) do { long long _n = CLK(); ES.eseg[i] += _n - ES.segT; ES.segT = _n; } while (0)   // node engine segments (its own FastS)
#define FSEG(i) do { long long _n = CLK(); S.eseg[i] += _n - S.segT; S.segT = _n; } while (0)
#define SEG_BEGIN() S.segT = CLK()
#define SEG(i) do { long long _n = CLK(); if (FLANE == 0 && _n - S.segT < (1ll << 32)) RS.statSeg[i] += _n - S.segT; S.segT = _n; } while (0)  // cold helpers start their own clock at 0: skip those
#else
#define SEG_BEGIN() do {} while (0)
#define SEG(i) do {} while (0)
#define ESEG(i) do {} while (0)
#define FSEG(i) do {} while (0)
#endif
#if defined(ASCHED_HOSTSIM)
void hsRingIdle();   // tests/hostsim/fast_serial.h: the serial engine makes progress while the control code waits (HS_RING_LAG)
int hsBindLag();
#define STREAM_IDLE() hsRingIdle()
#elif !defined(__HIP_DEVICE_COMPILE__)
#define STREAM_IDLE() do {} while (0)
#else
DEV void streamIdle();   // armada_sched.hip: a short sleep + the bounded-wait bookkeeping
#define STREAM_IDLE() streamIdle()
#endif
DEV void uniQHot(QHot& f) {
  f.weight = UNID(f.weight); f.tokens = UNID(f.tokens); f.budget = UNID(f.budget); f.proposed = UNID(f.proposed); f.current = UNID(f.current); f.size = UNID(f.size);
  f.burst = UNI64(f.burst);
  f.itEi = UNI32(f.itEi); f.itQi = UNI32(f.itQi); f.itStage = UNI32(f.itStage); f.itJobsSeen = UNI32(f.itJobsSeen); f.itNext = UNI32(f.itNext); f.gctx = UNI32(f.gctx);
  f.evEnd = UNI32(f.evEnd); f.qEnd = UNI32(f.qEnd); f.pcPrio = UNI32(f.pcPrio); f.schedPrio = UNI32(f.schedPrio);
  f.rateInf = UNI32(f.rateInf); f.cordoned = UNI32(f.cordoned); f.itJobOnlyEv = UNI32(f.itJobOnlyEv); f.itGangOnlyEv = UNI32(f.itGangOnlyEv);
  f.headFast = UNI32(f.headFast); f.headKind = UNI32(f.headKind); f.headIdx = UNI32(f.headIdx);
  f.winKind = UNI32(f.winKind); f.winStart = UNI32(f.winStart); f.winCount = UNI32(f.winCount);
  f.evCheap = UNI32(f.evCheap); f.evApplied = UNI32(f.evApplied); f.evDone = UNI32(f.evDone); f.ewStart = UNI32(f.ewStart); f.ewCount = UNI32(f.ewCount); f.headPos = UNI32(f.headPos);
  f.effValid = UNI32(f.effValid); f.skipStart = UNI32(f.skipStart); f.sPos = UNI32(f.sPos); f.sLen = UNI32(f.sLen);
}
DEV void uniJobTail(JobTail& r) {
  r.keyDelta = UNI64(r.keyDelta); r.fieldMin = UNI64(r.fieldMin);
  r.pc = UNI32(r.pc); r.shape = UNI32(r.shape); r.gang = UNI32(r.gang); r.node0 = UNI32(r.node0); r.runPrio = UNI32(r.runPrio); r.cls = UNI32(r.cls); r.pcPrio = UNI32(r.pcPrio);
  int fl = r.never | (r.preemptible << 8) | (r.nlPc << 16) | (r.nlRun << 24); fl = UNI32(fl);
  r.never = (uint8_t)(fl & 255); r.preemptible = (uint8_t)((fl >> 8) & 255); r.nlPc = (uint8_t)((fl >> 16) & 255); r.nlRun = (uint8_t)((fl >> 24) & 255);
  r.ex0 = UNI64(r.ex0); r.ex1 = UNI64(r.ex1);
}
DEV void uniCand(CandRec& c) { c.pos = UNI32(c.pos); c.node = UNI32(c.node); c.key = UNI64(c.key); c.cls = UNI64(c.cls); c.ex0 = UNI64(c.ex0); c.ex1 = UNI64(c.ex1); }
struct FitHandle { int src; int slot; };  // src 0: base candidate of the shape, 1: L0 slot
// what the fast loop needs of Ctl + PassCfg, by value
struct FastCtx { int withQueued; uint32_t maxLookback; int skipKnown, compareSchedPrio, preferLarge, replay, evStatic, engine, stream; };

// loop constants: configuration and array bases, read once per fastRun (registers for the whole run)
struct FastK {
  int R, K, P, E, ex0col, ex1col, N, npc, S, disableHome, hasPcLimit, anyDisallowed, anyRoundLimit;
  size_t Npad;
  uint64_t fieldMask[MAXK]; uint64_t minFieldMin; uint64_t guardMask; int64_t minEx0, minEx1;
  GP(uint64_t) baseKey; GP(int32_t) baseNode; GP(int64_t) baseExtra; GP(uint64_t) baseCls; GP(uint8_t) baseRemoved; GP(int32_t) l0Slot;
  GP(int64_t) alloc; GP(uint64_t) keys; GP(unsigned long long) jrec; GP(int32_t) evList; GP(int32_t) queuedJobs; GP(int32_t) evIdxByPos; GP(unsigned long long) evKey;
  GP(uint8_t) jcEvicted; GP(int32_t) jcAssigned; GP(int32_t) jcReason; GP(uint8_t) jcHasPctx; GP(int32_t) jcGangCard; GP(int32_t) jcUniValue; GP(int32_t) jcStagedBy;
  GP(int32_t) pcNode; GP(int32_t) pcSap; GP(int32_t) pcPap; GP(int32_t) pcMethod; GP(int32_t) jobNode; GP(int32_t) jobCutoff; GP(uint8_t) jobEvictedOnNode;
  GP(int32_t) schedAtPrio; GP(uint8_t) inSchedAndEvicted; GP(uint8_t) inPreempted; GP(uint8_t) inScheduled; GP(uint8_t) jobFlags;
  GP(uint8_t) evTabAlive; GP(int32_t) evTabJob; GP(int32_t) evIndexOfJob; GP(uint8_t) unfeasible;
  GP(int64_t) qAllocByPc; GP(int64_t) qSchedByPc; GP(int64_t) qEvictedByPc;
  GP(uint8_t) jcPreempted; GP(uint8_t) nodeFlags;
  GP(unsigned long long) qsKey;
  GP(unsigned long long) fitBits; int fitW;
  int maskMode;
  int32_t prios[MAXP];
};
// scheduling-context scalars the loop reads and writes (context/scheduling.go:27-77), written back to RS at the end of a run
struct FastS {
  double globalTokens; int64_t globalBurst; int32_t globalRateInf;
  int32_t numScheduledJobs, numScheduledGangs, numEvictedJobs, numNodeQueries, loopIterations, evictedTableSize;
  int32_t numUnfeasible, numPreemptedMarks, fastActive, lvl0NonNeg, replayPending;
  int32_t statFastIters, statScanSteps, statRefills, statL0Max, statFastReplay;
  long long segT;
  int laneL, laneX;  // device only: this lane's (level offset, resource) in a bind: lane = laneL * R + laneX
  long long engWaitClk;  // ticks this wave spent waiting for the engine's verdict
  int engSeq;            // commands posted to the engine in this session
  int engLive, engPend;  // node engine started for this run; queue whose speculative iteration awaits the engine's verdict (-1 none)
  int inlineStreak;      // queued jobs in a row that this wave placed itself (the engine starts after ENG_START_AFTER of them)
#ifdef ASCHED_FASTPROF
  long long eseg[8];
#endif
};

// The loop's constants are built once per round_prepare on the host (every value is a config field or a device pointer the
// host allocated) and live in HBM; the device reads them through the constant address space: uniform scalar loads into
// SGPRs, batched by the compiler, no LDS traffic and no vector registers.
HD void fastKInit(const Dev& d, FastK& k) {
  const DevCfg& c = d.cfg;
  k.R = c.R; k.K = c.K; k.P = c.P; k.E = d.f.E; k.ex0col = d.f.extraCol[0]; k.ex1col = d.f.extraCol[1]; k.N = c.N; k.npc = c.npc; k.S = d.f.F;   // (S: fit shapes)
  k.disableHome = c.disableHome; k.hasPcLimit = d.hasPcLimit; k.Npad = (size_t)c.Npad;
  k.anyDisallowed = 0; k.anyRoundLimit = 0;
  for (int i = 0; i < MAXR; i++) if (i < c.R && c.maxToSchedule[i] != INT64_MAX) k.anyRoundLimit = 1;  // MaximumResourceFractionToSchedule unset: +Inf x total saturates (resource_list.go:312-331)
  for (int i = 0; i < MAXR; i++) if (i < c.R && c.disallowed[i]) k.anyDisallowed = 1;
  for (int i = 0; i < MAXK; i++) k.fieldMask[i] = i < c.K ? d.f.fieldMask[i] : 0;
  k.minFieldMin = d.f.minFieldMin; k.guardMask = d.f.guardMask; k.minEx0 = d.f.minExtra[0]; k.minEx1 = d.f.minExtra[1];
  k.baseKey = GA(uint64_t, d.baseKey); k.baseNode = GA(int32_t, d.baseNode); k.baseExtra = GA(int64_t, d.baseExtra); k.baseCls = GA(uint64_t, d.baseCls);
  k.baseRemoved = GA(uint8_t, d.baseRemoved); k.l0Slot = GA(int32_t, d.l0Slot);
  k.alloc = GA(int64_t, d.alloc); k.keys = GA(uint64_t, d.keys); k.jrec = GA(unsigned long long, (unsigned long long*)d.jrec);
  k.evList = GA(int32_t, d.evList); k.queuedJobs = GA(int32_t, d.queuedJobs); k.evIdxByPos = GA(int32_t, d.evIdxByPos); k.evKey = GA(unsigned long long, (unsigned long long*)d.evKey);
  k.jcEvicted = GA(uint8_t, d.jcEvicted); k.jcAssigned = GA(int32_t, d.jcAssigned); k.jcReason = GA(int32_t, d.jcReason); k.jcHasPctx = GA(uint8_t, d.jcHasPctx);
  k.jcGangCard = GA(int32_t, d.jcGangCard); k.jcUniValue = GA(int32_t, d.jcUniValue); k.jcStagedBy = GA(int32_t, d.jcStagedBy);
  k.pcNode = GA(int32_t, d.pcNode); k.pcSap = GA(int32_t, d.pcSap); k.pcPap = GA(int32_t, d.pcPap); k.pcMethod = GA(int32_t, d.pcMethod);
  k.jobNode = GA(int32_t, d.jobNode); k.jobCutoff = GA(int32_t, d.jobCutoff); k.jobEvictedOnNode = GA(uint8_t, d.jobEvictedOnNode);
  k.schedAtPrio = GA(int32_t, d.schedAtPrio); k.inSchedAndEvicted = GA(uint8_t, d.inSchedAndEvicted); k.inPreempted = GA(uint8_t, d.inPreempted);
  k.inScheduled = GA(uint8_t, d.inScheduled); k.jobFlags = GA(uint8_t, d.jobFlags);
  k.evTabAlive = GA(uint8_t, d.evTabAlive); k.evTabJob = GA(int32_t, d.evTabJob); k.evIndexOfJob = GA(int32_t, d.evIndexOfJob); k.unfeasible = GA(uint8_t, d.unfeasible);
  k.qAllocByPc = GA(int64_t, d.qAllocByPc); k.qSchedByPc = GA(int64_t, d.qSchedByPc); k.qEvictedByPc = GA(int64_t, d.qEvictedByPc);
  k.jcPreempted = GA(uint8_t, d.jcPreempted); k.nodeFlags = GA(uint8_t, d.nodeFlags);
  k.qsKey = GA(unsigned long long, (unsigned long long*)d.qsKey);
  k.maskMode = d.f.maskMode;
  k.fitBits = GA(unsigned long long, (unsigned long long*)d.fitBits); k.fitW = d.fitW;
  for (int i = 0; i < MAXP; i++) k.prios[i] = i < c.P ? c.prios[i] : INT32_MAX;
}
// a register copy of the constants: one burst of scalar loads (constant address space) per call, then no memory traffic
#if defined(__HIP_DEVICE_COMPILE__)
// (NOT scalar loads, whatever the cast says: the struct copy goes through the copy constructor's generic reference and comes out as flat loads into vector registers.
//  Measured the other way round — a word-by-word copy through the constant address space, the constants in SGPRs — the kernel spills 130 more SGPRs and every round
//  is 4-10 % SLOWER: the vector file has room for them, the scalar file has not; profiles/r06z_uniform_arguments.txt)
__device__ static inline FastK fastKRef(const Dev& d) { return *(const __attribute__((address_space(4))) FastK*)d.fk; }
#else
HD FastK fastKRef(const Dev& d) { return *d.fk; }
#endif
#define KAL(k, l, r, n) ((k).alloc[((size_t)(l) * (k).R + (r)) * (k).Npad + (n)])
#define KKEY(k, l, n) ((k).keys[(size_t)(l) * (k).Npad + (n)])

// ------------------------------------------------------------------------------------------------ small helpers
DEV uint64_t dbits(double x) { return __builtin_bit_cast(uint64_t, x); }

DEV bool fieldsGE(KREF k, uint64_t key, uint64_t fmin) {  // every packed field of key >= the same field of fmin
  // with a guard bit above every field (host: keyGuard): set the guards, subtract; a field that is smaller borrows from ITS guard and from nothing
  // else (fmin has zero guard and index bits, so the index bits of key never borrow)
  if (k.guardMask) return (((key | k.guardMask) - fmin) & k.guardMask) == k.guardMask;
  bool ok = true;
  for (int i = 0; i < MAXK; i++) { uint64_t m = k.fieldMask[i]; ok = ok && (key & m) >= (fmin & m); }  // unused fields have mask 0
  return ok;
}
// mask mode (FastCfg.maskMode): the shape table sits in the unused upper half of the candidate array (F <= 128 = SMAX / 2)
#define SHT(s) (((ShapeReq*)&FL.cand[SMAX / 2])[s])
static_assert(sizeof(ShapeReq) == 32 && 128 * sizeof(ShapeReq) <= (SMAX / 2) * sizeof(CandRec) && SMAX / 2 >= 128, "shape table fits behind the candidates");
// one word of the fit bitmap: base entries [64 w, 64 w + 64) against fit shape f, at base build time (every entry is clean)
HD uint64_t fitBitsWord(const Dev& d, int f, int w) {
  const ShapeReq q = d.shapeTab[f];
  uint64_t m = 0;
  if (q.never) return 0;
  for (int b = 0; b < 64; b++) {
    int p = w * 64 + b;
    if (p >= d.cfg.N) break;
    uint64_t key = d.baseKey[p];
    int64_t ex0 = d.f.E > 0 ? d.baseExtra[p] : 0, ex1 = d.f.E > 1 ? d.baseExtra[(size_t)d.cfg.Npad + p] : 0;
    bool ok = ((d.nodeCls[d.baseNode[p]] >> q.cls) & 1) && q.ex0 <= ex0 && q.ex1 <= ex1;
    if (ok) {
      if (d.f.guardMask) ok = (((key | d.f.guardMask) - q.fieldMin) & d.f.guardMask) == d.f.guardMask;
      else for (int i = 0; i < MAXK; i++) { uint64_t fm = d.f.fieldMask[i]; ok = ok && (key & fm) >= (q.fieldMin & fm); }
    }
    if (ok) m |= 1ull << b;
  }
  return m;
}
DEV bool entryFits(KREF k, const JobTail& r, uint64_t key, int64_t ex0, int64_t ex1, uint64_t cls) {
  bool ok = ((cls >> r.cls) & 1) != 0;      // StaticJobRequirementsMet via the requirement class (nodematching.go:161-183)
  ok = ok && fieldsGE(k, key, r.fieldMin);  // indexed columns: alloc/res >= req/res (both resolution-aligned)
  ok = ok && r.ex0 <= ex0 && r.ex1 <= ex1;  // non-indexed columns (nodematching.go:194-197); unused extras are 0 vs 0
  return ok;
}
DEV bool entryLive(KREF k, uint64_t key, int64_t ex0, int64_t ex1) {  // could still host the smallest request of some shape
  return fieldsGE(k, key, k.minFieldMin) && k.minEx0 <= ex0 && k.minEx1 <= ex1;
}
DEV bool fastOn(Dev& d, const Ctl& c) { return c.fastEnabled && d.f.iterOk == 1; }   // (iterOk == 2: more than QCAPF queues, round_wide.h)
DEV void fastHeadInvalidate(int q) { if (q < QCAPF) FL.hot[q].headFast = 0; }
DEV void fastPassReset() { for (int q = 0; q < QCAPF; q++) { FL.hot[q].headFast = 0; FL.hot[q].winCount = 0; FL.hot[q].winKind = -1; FL.hot[q].ewCount = 0; FL.hot[q].ewStart = 0; } }

// Less (queue_scheduler.go:738-798) as a lexicographic key (kA, kX, kY, name rank); exact for finite, non-negative costs
struct KeyOut { int valid; uint32_t A; uint64_t X, Y; };
DEV KeyOut packItemKeys(int preferLarge, int q, int32_t prio, double proposed, double current, double size, double budget) {
  PackedKey pk = packKey3(preferLarge, prio, proposed, current, size, budget);
  KeyOut o; o.valid = 1; o.A = pk.A; o.X = pk.X; o.Y = pk.Y;
  FL.kA[q] = o.A; FL.kX[q] = o.X; FL.kY[q] = o.Y;
  return o;
}
DEV void fastItemKeys(Dev& d, const Ctl& c, int q) {  // from the generic arrays (generic updatePQItem)
  if (d.f.iterOk != 1 || q >= QCAPF) return;
  packItemKeys(c.preferLarge, q, c.compareSchedPrio ? d.pqSchedPrio[q] : d.pqPcPrio[q], d.pqProposed[q], d.pqCurrent[q], d.pqSize[q], d.pqBudget[q]);
}

// ------------------------------------------------------------------------------------------------ fast <-> generic state hand-over
DEV void fastQLoad(Dev& d) {
  int R = d.cfg.R;
  FOR_LANES(q, d.cfg.Q) {
    QHot& f = FL.hot[q];
    f.weight = d.qWeight[q]; f.tokens = d.qTokens[q]; f.budget = d.pqBudget[q];
    f.proposed = d.pqProposed[q]; f.current = d.pqCurrent[q]; f.size = d.pqSize[q];
    f.burst = d.qBurst[q];
    for (int x = 0; x < MAXR; x++) { bool in = x < R; FL.qAlloc[q][x] = in ? QV(d.qAlloc, q)[x] : 0; FL.qPenalty[q][x] = in ? QV(d.qPenalty, q)[x] : 0; FL.qReplay[q][x] = in ? QV(d.replayAlloc, q)[x] : 0; }
    f.itEi = d.itEi[q]; f.itQi = d.itQi[q]; f.itStage = d.itStage[q]; f.itJobsSeen = d.itJobsSeen[q];
    f.itNext = d.itNext[q]; f.gctx = d.pqGctx[q]; f.evEnd = d.evOff[q + 1]; f.qEnd = d.queuedOff[q + 1];
    f.pcPrio = d.pqPcPrio[q]; f.schedPrio = d.pqSchedPrio[q];
    f.rateInf = d.qRateInf[q]; f.cordoned = d.qCordoned[q]; f.itJobOnlyEv = d.itJobOnlyEv[q]; f.itGangOnlyEv = d.itGangOnlyEv[q];
    { int g = f.gctx; bool headEv = g >= 0 && d.jcEvicted[g];  // an evicted head was yielded from evList[itEi-1] and is not served yet
      f.evCheap = d.evCheap ? d.evCheap[q] : 0; f.evDone = f.evApplied = headEv ? f.itEi - 1 : f.itEi; f.headPos = headEv ? f.itEi - 1 : -1;
      FL.evValidEnd[q] = f.evCheap == 2 ? 0 : INT32_MAX;   // (2: the stream was repaired in this pass — how far is not kept across a generic excursion: the next cheap head repairs again)
      f.effValid = 0; f.skipStart = 0; f.sPos = 0; f.sLen = 0;
      if (d.qsSave && d.qsSave[q].valid) {   // the queue's stream from before the generic excursion still describes it: same head, same cursor, same allocation, same tokens
        const QsSave& sv = d.qsSave[q];
        bool same = sv.itQi == f.itQi && sv.gctx == f.gctx && sv.tokens == f.tokens && sv.numUnfeasible == d.rs->numUnfeasible && d.pqInHeap[q];
        for (int x = 0; x < MAXR; x++) same = same && sv.alloc[x] == FL.qAlloc[q][x];
        if (same) { f.sPos = sv.sPos; f.sLen = sv.sLen; FL.sKind[q] = 0; }
      } }
    FL.inHeap[q] = d.pqInHeap[q]; FL.nameRank[q] = d.qNameRank[q];
  }
}
DEV void fastQFlush(Dev& d) {
  int R = d.cfg.R;
  FOR_LANES(q, d.cfg.Q) {
    const QHot& f = FL.hot[q];
    d.qTokens[q] = f.tokens;
    d.pqProposed[q] = f.proposed; d.pqCurrent[q] = f.current; d.pqSize[q] = f.size;
    for (int x = 0; x < R; x++) { QV(d.qAlloc, q)[x] = FL.qAlloc[q][x]; QV(d.replayAlloc, q)[x] = FL.qReplay[q][x]; }
    d.itEi[q] = f.itEi; d.itQi[q] = f.itQi; d.itStage[q] = f.itStage; d.itJobsSeen[q] = f.itJobsSeen;
    d.itNext[q] = f.itNext; d.pqGctx[q] = f.gctx; d.pqPcPrio[q] = f.pcPrio; d.pqSchedPrio[q] = f.schedPrio;
    d.pqInHeap[q] = (uint8_t)FL.inHeap[q];
    d.itJobOnlyEv[q] = (uint8_t)f.itJobOnlyEv; d.itGangOnlyEv[q] = (uint8_t)f.itGangOnlyEv;
    if (d.qsSave) {
      QsSave sv; sv.valid = (f.sLen > f.sPos && !FL.sKind[q] && !f.effValid) ? 1 : 0; sv.sPos = f.sPos; sv.sLen = f.sLen; sv.itQi = f.itQi; sv.gctx = f.gctx; sv.numUnfeasible = d.rs->numUnfeasible; sv.tokens = f.tokens;
      for (int x = 0; x < MAXR; x++) sv.alloc[x] = FL.qAlloc[q][x];
      d.qsSave[q] = sv;
    }
  }
}
DEV void fastEnsureLive(Dev& d, Ctl& c) { if (!c.fqLive) { fastQLoad(d); c.fqLive = 1; } }

// ------------------------------------------------------------------------------------------------ lane-parallel primitives
// The queue heap of CostBasedCandidateGangIterator as the fast loop sees it.  Device: the queues sorted by key across the
// lanes of the control wave (registers); serving the head re-inserts it with one lane shift.  Host: argmin over the keys.
struct EvDyn { int preempted; int fits; };
#ifdef ASCHED_HOSTSIM
#include "fast_serial.h"          // tests/hostsim/: serial stand-ins of the primitives below for the CPU build of the control code (test infrastructure)
#else  // device versions: armada_sched.hip
struct PQState { uint32_t A, N; unsigned long long X, Y; int q; int count; };  // lane i: the i-th queue in heap order
DEV void pqBuild(PQState& s, int Q);
DEV int pqHead(PQState& s, int Q);
DEV void pqPopPush(PQState& s, const KeyOut& ko, int q);
DEV void drf3(Dev& d, int q, int k, bool replay, double w, double* proposed, double* current, double* size);
DEV void fastFence(Ctl& c);
DEV void baseScan(KREF k, FastS& S, const JobTail& r);
DEV void baseTileRemoved(KREF k, FastS& S, int pos);
DEV uint64_t l0Search(KREF k, const JobTail& r, int* slot);
DEV void winRefill(KREF k, int q, int kind, int pos, int cnt);
DEV void loadHeadRec(KREF k, int q, int job);
DEV void headFromWindow(int q, int w);
DEV void bindUpdate(KREF k, FastS& S, int n, int lo, int nl, int q, uint64_t keyDelta);
DEV void keySatSub(KREF k, int n, int lo, int nl, uint64_t keyDelta);   // keys of levels [lo, nl) of node n: every field minus keyDelta's, saturating at 0
DEV void accountVectors(Dev& d, KREF k, int q, int pc, bool ev, bool replay);
DEV void evWinRefill(KREF k, int q, int pos, int cnt);
__device__ static void applyEvictedRange(Dev& d, int q, int p0, int p1, int sign = 1);  // not inlined, reads the constants itself: nothing of the hot loop has to live in memory for it
DEV bool roundLimitExceeded(Dev& d, KREF k);
DEV void bindUpdateEng(KREF k, FastS& S, int n, int nl, uint64_t keyDelta, const int64_t* req);
DEV void accountVectorsBk(Dev& d, KREF k, int q, int pc, int sign);
DEV void engineRestore(int q);
DEV void enginePost(Dev& d, KREF k, FastS& S, int job, int q, int pc, int32_t prio, int32_t cutoff, int nl);  // backup of queue q + the job, one LDS pass
DEV int engineWait(const FastS& S);
DEV void pqHeadKey(PQState& s, int t, PackedKey* key, uint32_t* nameRank);
DEV void engineStart(Dev& d, FastS& S);
DEV void engineStop(Dev& d, FastS& S);
DEV bool headRequestsDisallowed(Dev& d, KREF k, int q);
DEV bool pinnedNodeFits(KREF k, int q, int n, int level);
DEV void exclPinnedFast(Dev& d, KREF k, int q, int job, int n, int level);   // the lane-parallel form of round_wide.h exclPinned (lane r: resource column r)
DEV EvDyn evDynLoad(KREF k, int q, int job, int n, int level, bool wantMark, bool wantPin);
DEV EvDyn evCleanLoad(KREF k, int job, int n, bool wantMark, bool wantClean);   // a returning evicted job's preempted mark and "no priority -2 column of ITS node is negative" (fits = clean), the loads in flight together
DEV unsigned long long evPendingMask(int Q);   // queues with deferred evicted-job commits (evApplied < evDone)   // a returning evicted job's preempted mark and its pinned-node check, their loads in flight together
// per-queue state of a stream run, held in the lanes of the control wave (lane q: queue q): position at the start of the run, list position of element 0,
// merge cursor, length, kind (bit 0: evicted stream, bit 1: the queue's evicted list was folded (running-maximum key)), first element of the key window in
// LDS, the queue's budget, the running-maximum key.  Reading queue t's values is a v_readlane (scalar result), no LDS round trip.
struct StreamLanes { int start, base, pos, len, kind, ws; double budget; uint32_t effA; unsigned long long effX, effY; };
#define SL_SET(sl, f, q, v) do { if ((int)(threadIdx.x & 63) == (q)) (sl).f = (v); } while (0)
#define SL_GET(sl, f, q) __builtin_amdgcn_readlane((int)(sl).f, (q))
__device__ static inline unsigned long long slGet64(unsigned long long v, int q) { unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, q), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), q); return ((unsigned long long)hi << 32) | lo; }
#define SL_GET64(sl, f, q) slGet64((sl).f, (q))
#define SL_GETD(sl, f, q) __builtin_bit_cast(double, slGet64(__builtin_bit_cast(unsigned long long, (sl).f), (q)))
DEV void qsWinRefill(KREF k, int q, int pos, int cnt);
DEV void streamBegin(int* engSeq, int hold = 0, int hc = 0);   // hc: the ring session runs on the split level-0 structure (engine_hc.h; device only)
DEV void streamRelease(Dev& d, KREF k, int go);   // gang: the held binds of every placed member are issued (go) or dropped
DEV unsigned long long streamStageIssue(KREF k, int base, int cnt);
DEV void streamStageCommit(Dev& d, KREF k, int base, int cnt, unsigned long long v);
DEV void streamEnd(int engSeq);
DEV void streamAccount(Dev& d, KREF k, int i0, int i1);   // ring entries [i0, i1): sctx / qctx sums, FL.tmpQ[queue] counts them
DEV void baseMarkRemoved(KREF k, FastS& S, int pos);   // base entry pos is stale from now on: flag + its bit in every fit shape's bitmap
DEV void capMask2(KREF k, uint64_t clsBits, uint64_t key, int64_t ex0, int64_t ex1, uint64_t* m0, uint64_t* m1);   // mask mode: which fit shapes fit (lane l evaluates shapes l and l + 64)
DEV int streamAcked(int* fail);
DEV int streamBound();   // entries whose ring slot is free again (the bind wave has read them)
DEV void wgBulk(Dev& d, int kind, int n);
#endif
DEV void candInvalidate(int S, int n) { FOR_LANES(s, S) if (FL.cand[s].node == n) FL.cand[s].node = -2; }
DEV void candResetAll(Dev& d, const int32_t* pos) { FOR_LANES(s, d.f.F < SMAX ? d.f.F : SMAX) { FL.cand[s].node = -2; FL.cand[s].pos = pos ? pos[s] : 0; FL.cand[s].key = 0; } }  // key 0: no lower bound known, the first query scans
DEV void candSaveAll(Dev& d, int32_t* pos) { FOR_LANES(s, d.f.F < SMAX ? d.f.F : SMAX) pos[s] = FL.cand[s].pos; }

// Peek's skip of known-unfeasible scheduling keys (queue_scheduler.go:398-413) for the queued jobs at queuedJobs[pos .. pos + max): how many of them, from the first on, are
// single jobs whose scheduling key is registered as unfeasible — each gets the record the generic loop gives it (JobSchedulingContextFromJob, the key's reason replaced by
// "skipped", sctx.AddJobSchedulingContext of a failed job: flags only).  The jobs are independent of each other: 64 per step, one lane each.  The caller advances the queue's
// cursor and jobs-seen count and has bounded `max` by the lookback limit.
DEV int skipUnfeasibleRun(Dev& d, int pos, int max) {
  int done = 0;
  while (done < max) {
    int nb = max - done < 64 ? max - done : 64, n = 0;
#if defined(ASCHED_HOSTSIM) || !defined(__HIP_DEVICE_COMPILE__)
    while (n < nb) { int job = d.queuedJobs[pos + done + n]; if (!(d.jGang[job] < 0 && d.unfeasible[d.jShape[job]] && !(d.jobFlags[job] & F_SUCCESSFUL))) break; n++; }
#else
    { int x = FLANE; bool ok = false;
      if (x < nb) { int job = d.queuedJobs[pos + done + x]; ok = d.jGang[job] < 0 && d.unfeasible[d.jShape[job]] && !(d.jobFlags[job] & F_SUCCESSFUL); }
      unsigned long long bad = ~__ballot(ok);
      n = bad ? __ffsll((long long)bad) - 1 : 64;
      if (n > nb) n = nb; }
#endif
    FOR_LANES(x, 64) if (x < n) {
      int job = d.queuedJobs[pos + done + x];
      d.jcEvicted[job] = 0; d.jcAssigned[job] = -1; d.jcGangCard[job] = 1; d.jcUniValue[job] = -1; d.jcStagedBy[job] = -1;   // JobSchedulingContextFromJob (context/job.go:149-158)
      d.jcHasPctx[job] = 1; d.pcNode[job] = -1; d.pcMethod[job] = ASCHED_METHOD_NONE;
      d.jobFlags[job] = (uint8_t)(d.jobFlags[job] | F_UNSUCCESSFUL);
      d.jcReason[job] = ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY;
    }
    done += n;
    if (n < nb) break;
  }
  return done;
}

// Before generic code runs: the LDS queue records back into the generic arrays, and the fast path's no-return atomics
// made visible to plain loads
DEV void fastFlushEvicted(Dev& d) {  // apply every queue's deferred evicted-job commits
  const FastK k = fastKRef(d);
  for (int q = 0; q < d.cfg.Q; q++) {
    int p0 = UNI32(FL.hot[q].evApplied), p1 = UNI32(FL.hot[q].evDone);
    if (p0 < p1) { applyEvictedRange(d, q, p0, p1); FL.hot[q].evApplied = p1; RS.numEvictedJobs -= p1 - p0; }
  }
}
// the same in front of a cascade that stays in the fast loop (fastPreemptIter): the queues with something pending are found lane-parallel (usually none or a few)
DEV void fastFlushEvictedPending(Dev& d) {
  unsigned long long m = evPendingMask(d.cfg.Q < QCAPF ? d.cfg.Q : QCAPF);
  while (m) {
    int q = __builtin_ctzll(m); m &= m - 1;
    int p0 = UNI32(FL.hot[q].evApplied), p1 = UNI32(FL.hot[q].evDone);
    applyEvictedRange(d, q, p0, p1);
    if (FLANE == 0) { FL.hot[q].evApplied = p1; RS.numEvictedJobs -= p1 - p0; }
    LANE0_PUBLISHED();
  }
}
// queue q's precomputed evicted-stream costs (B_EVKEYS) assume that every earlier evicted job of the queue came back: once one was skipped as preempted or found
// no room on its node the rest of the stream is costed from the queue's actual allocation (fastAdvance's window path)
DEV void evCheapOff(Dev& d, int q, QHot& f) {
  f.evCheap = 0;
  if (FLANE == 0) { FL.hot[q].evCheap = 0; if (d.evCheap) d.evCheap[q] = 0; }
  LANE0_PUBLISHED();
}
// May a queue's evicted stream be repaired in place?  Not while the deferred replay of the evicted jobs is still to come (it merges the very same costs, as the
// evictor left them) and not inside the replay itself.
DEV bool evRepairOk(Dev& d, const FastCtx& fc) { return fc.evStatic && !fc.replay && !RS.replayPending; }
// Queue q's precomputed costs from stream position p on no longer describe it (the job before p was skipped as preempted or found no room: the prefix the bulk pass
// counted on is not the queue's allocation): the next cheap head of the queue recomputes them (fastAdvance -> evRepair).
DEV void evInvalidateFrom(Dev& d, int q, QHot& f, int p) {
  f.evCheap = 2; f.ewCount = 0;
  if (FLANE == 0) { FL.evValidEnd[q] = p; FL.hot[q].evCheap = 2; FL.hot[q].ewCount = 0; d.evCheap[q] = 2; }
  LANE0_PUBLISHED();
}
// The costs of queue q's evicted jobs [p0, p1), p1 - p0 <= 64, recomputed in place (d.evKey) from the queue's allocation as it is NOW — every deferred commit of the queue
// applied by the caller — one lane per job: the float64 operations of B_EVKEYS (round_run.h) on the same prefix sums, with one difference that is the point: a job that
// carries a preempted mark stays out of the prefix of the jobs behind it (it will be skipped, queue_scheduler.go:150-156; its own key is the one the iterator would
// compute for it).  Such positions are flagged (EV_EXCL): reaching them later does not invalidate the stream again.
DEV void evRepair(Dev& d, KREF k, int q, int p0, int p1) {
  const int cnt = p1 - p0, R = k.R;
  FOR_LANES(i, cnt) {
    int job = k.evList[p0 + i];
    bool marked = k.jcPreempted[job] != 0;
    const int64_t* rq = JREQ(d, job);
    for (int r = 0; r < R; r++) FL.evStage[i][r] = marked ? 0 : rq[r];
    EV_EXCL(d)[p0 + i] = marked ? 1 : 0;
  }
  LANE0_PUBLISHED();
  const double w = UNID(FL.hot[q].weight);
  FOR_LANES(i, cnt) {
    int job = k.evList[p0 + i];
    int64_t a[MAXR], with[MAXR];
    for (int r = 0; r < R; r++) a[r] = FL.qAlloc[q][r] + FL.qPenalty[q][r];
    for (int j = 0; j < i; j++) for (int r = 0; r < R; r++) a[r] += FL.evStage[j][r];
    const int64_t* req = JREQ(d, job);
    for (int r = 0; r < R; r++) with[r] = a[r] + req[r];
    EvKey e;
    e.proposed = drf(d, with) / w; e.current = drf(d, a) / w; e.size = drf(d, req) * w;
    e.pcPrio = d.cfg.pcPriority[d.jPc[job]]; e.job = job;
    d.evKey[p0 + i] = e;
  }
  FAST_GLOBAL_FENCE();   // the window refills read these from HBM
  if (FLANE == 0) { FL.evValidEnd[q] = p1; FL.hot[q].ewCount = 0; FL.hot[q].evCheap = 2; d.evCheap[q] = 2; }
  LANE0_PUBLISHED();
}
DEV void fastEnterGeneric(Dev& d, Ctl& c) {
  if (c.fqLive) { fastFlushEvicted(d); fastQFlush(d); c.fqLive = 0; }
  fastFence(c);
}

// ------------------------------------------------------------------------------------------------ L0 maintenance
DEV bool l0Insert(KREF k, int n, uint64_t key, int64_t ex0, int64_t ex1, uint64_t cls, uint64_t cls2 = 0) {
  int i = FL.l0Count;
  if (i >= L0CAP) return false;
  FL.l0Key[i] = key; FL.l0Node[i] = n; FL.l0Cls[i] = cls; FL.l0Cls2[i] = cls2; FL.l0Ex0[i] = ex0; FL.l0Ex1[i] = ex1;
  FL.l0Count = i + 1;
  if (FLANE == 0) k.l0Slot[n] = i;
  return true;
}
DEV void l0Remove(KREF k, int slot) {
  int last = FL.l0Count - 1;
  int n = FL.l0Node[slot];
  if (FLANE == 0) k.l0Slot[n] = -1;
  if (slot != last) {
    FL.l0Key[slot] = FL.l0Key[last]; FL.l0Node[slot] = FL.l0Node[last]; FL.l0Cls[slot] = FL.l0Cls[last]; FL.l0Cls2[slot] = FL.l0Cls2[last];
    FL.l0Ex0[slot] = FL.l0Ex0[last]; FL.l0Ex1[slot] = FL.l0Ex1[last];
    if (FLANE == 0) k.l0Slot[FL.l0Node[slot]] = slot;
  }
  FL.l0Count = last;
}
DEV void fastDrop(Dev& d) { RS.fastActive = 0; RS.fastOverflow++; FL.l0Count = 0; }  // L0 overflow: the generic full scan takes over for the rest of the round

// node n's level-0 allocatable was changed by the generic code: bring base flags / L0 / candidates in line
DEV void fastTouch(Dev& d, int n) {
  if (!d.f.structOk || !RS.fastActive) return;
  const FastK k = fastKRef(d);
  uint64_t key = fastKeyOf(d, n);
  int64_t ex0 = k.E > 0 ? KAL(k, 0, k.ex0col, n) : 0, ex1 = k.E > 1 ? KAL(k, 0, k.ex1col, n) : 0;
  int pos = GA(int32_t, d.posOf)[n], slot = k.l0Slot[n];
  { FastS TS; HS_POISON(TS); baseMarkRemoved(k, TS, pos); }
  candInvalidate(k.S, n);
  uint64_t cls = GA(uint64_t, d.nodeCls)[n], cls2 = 0;
  bool live;
  if (k.maskMode) { capMask2(k, cls, key, ex0, ex1, &cls, &cls2); live = (cls | cls2) != 0; }
  else live = entryLive(k, key, ex0, ex1);
  if (slot >= 0) {
    if (live) { FL.l0Key[slot] = key; FL.l0Ex0[slot] = ex0; FL.l0Ex1[slot] = ex1; if (k.maskMode) { FL.l0Cls[slot] = cls; FL.l0Cls2[slot] = cls2; } }
    else l0Remove(k, slot);
  } else if (live && !l0Insert(k, n, key, ex0, ex1, cls, cls2)) fastDrop(d);
}

// first fit at priority -2 for a job record; -1 none; handle says where the winner came from
DEV int fastFirstFit(KREF k, FastS& S, const JobTail& r, FitHandle* h, CandRec* cOut) {
  if (r.never) return -1;
  int s = r.shape;
  int slot;
  uint64_t lk = l0Search(k, r, &slot);
  FSEG(7);
  CandRec c = FL.cand[s];
  uniCand(c);
  // A stale candidate (node -2: the node it named changed) still carries the key it was found with, and every clean entry the cursor can
  // still reach orders AFTER that key (the base is sorted, keys are unique, the entry itself is flagged removed).  When the best dirty node
  // already orders before it — the usual case: the node just bound moved down in the order and still has room — the dirty node wins without
  // looking at the base at all; the rescan (a dependent read of a base tile) waits until some job needs it.
  if (c.node == -2 && !(lk < c.key)) { baseScan(k, S, r); c = FL.cand[s]; uniCand(c); }
  *cOut = c;
  uint64_t bk = c.node >= 0 ? c.key : (c.node == -2 ? c.key : ~0ull);
  if (lk < bk) { h->src = 1; h->slot = slot; return UNI32(FL.l0Node[slot]); }
  if (c.node < 0) return -1;
  h->src = 0; h->slot = -1;
  return c.node;
}
DEV int fastSelectLevel0(Dev& d, int job) {
  if (!d.f.structOk || !RS.fastActive) return -2;
  const FastK k = fastKRef(d);
  FastS S; HS_POISON(S); S.statScanSteps = 0; 
  JobRec jr = d.jrec[job];
  JobTail r; memcpy(&r, &jr.keyDelta, sizeof r);
  FitHandle h; CandRec c;
  int n = fastFirstFit(k, S, r, &h, &c);
  RS.statScanSteps += S.statScanSteps;
  return n;
}
// the job of record r was bound to node n found through handle h: level-0 bookkeeping of the fast structure.
// Returns false when L0 overflowed (the caller drops the structure).
DEV bool fastAfterBind(KREF k, FastS& S, const JobTail& r, int n, const FitHandle& h, const CandRec& c) {
  if (h.src == 0) {
    uint64_t key = c.key - r.keyDelta;
    int64_t ex0 = c.ex0 - r.ex0, ex1 = c.ex1 - r.ex1;
    baseMarkRemoved(k, S, c.pos);
    baseTileRemoved(k, S, c.pos);
    candInvalidate(k.S, n);
    uint64_t cls = c.cls, cls2 = 0;   // (the base entry's class bits)
    bool live;
    if (k.maskMode) { capMask2(k, cls, key, ex0, ex1, &cls, &cls2); live = (cls | cls2) != 0; }   // which fit shapes the node can still host
    else live = entryLive(k, key, ex0, ex1);
    if (live) {
      if (!l0Insert(k, n, key, ex0, ex1, cls, cls2)) return false;
      if (FL.l0Count > S.statL0Max) S.statL0Max = FL.l0Count;
    }
  } else {
    int slot = h.slot;
    uint64_t key = FL.l0Key[slot] - r.keyDelta;
    int64_t ex0 = FL.l0Ex0[slot] - r.ex0, ex1 = FL.l0Ex1[slot] - r.ex1;
    uint64_t cls = 0, cls2 = 0;
    bool live;
    if (k.maskMode) {   // a bind only takes capacity away: the shapes that still fit are among those that did
      capMask2(k, ~0ull, key, ex0, ex1, &cls, &cls2);
      cls &= UNI64(FL.l0Cls[slot]); cls2 &= UNI64(FL.l0Cls2[slot]); live = (cls | cls2) != 0;
    } else live = entryLive(k, key, ex0, ex1);
    if (live) { FL.l0Key[slot] = key; FL.l0Ex0[slot] = ex0; FL.l0Ex1[slot] = ex1; if (k.maskMode) { FL.l0Cls[slot] = cls; FL.l0Cls2[slot] = cls2; } }
    else l0Remove(k, slot);
  }
  return true;
}

// scheduleMany's fast member (round_ctl.h): a queued, unpinned job of a gang attempt.  Runs on the control wave in generic mode (no node
// engine session is live there), on the authoritative HBM state: first fit at priority -2 through base cursor + L0, bind as lane-parallel
// no-return atomics, the job's result fields, the U_ADD undo record (a gang that does not fit is aborted, gang_scheduler.go:234-243; the
// undo path runs fastTouch on the node, which repairs L0 / base flags), L0 upkeep.
DEV bool fastGangMember(Dev& d, Ctl& c, int job) {
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_NO_GANG_FAST")) return false;
#endif
  if (!d.f.structOk || !RS.fastActive || d.cfg.disableHome || !d.jrec) return false;
  if (d.jcPreempted[job] || d.jcAssigned[job] >= 0 || d.jcUniValue[job] >= 0 || RS.awayRowPlus1) return false;
  if (d.schedAtPrio[job] != NO_PRIORITY || d.jobNode[job] >= 0) return false;   // already mapped to a priority / holding resources: the generic code knows the rules
  const FastK k = fastKRef(d);
  JobRec jr = d.jrec[job];
  JobTail r; memcpy(&r, &jr.keyDelta, sizeof r);
  uniJobTail(r);
  if (r.never) return false;
  if (k.anyDisallowed) for (int x = 0; x < k.R; x++) if (d.cfg.disallowed[x] && jr.req[x] > 0) return false;
  FastS S; HS_POISON(S); S.statScanSteps = 0;  S.statL0Max = RS.statL0Max;
  S.laneL = FLANE / (k.R > 0 ? k.R : 1); S.laneX = FLANE % (k.R > 0 ? k.R : 1);
  FitHandle h; h.src = 0; h.slot = -1;
  CandRec cand; cand.pos = 0; cand.node = -1; cand.key = 0; cand.cls = 0; cand.ex0 = cand.ex1 = 0; cand.pad = 0;
  int n = fastFirstFit(k, S, r, &h, &cand);
  RS.statScanSteps += S.statScanSteps;
  if (n < 0) return false;   // feasibility gate, preemption, away node types: the generic cascade (it counts its own queries)
  RS.numNodeQueries++;
  int32_t prio = r.pcPrio, cutoff = r.preemptible ? prio : NONPREEMPTIBLE_CUTOFF;
  int32_t oldCutoff = d.jobCutoff[job];
  FOR_LANES(x, MAXR) FL.eng.req[x] = x < k.R ? jr.req[x] : 0;   // the engine mailbox is free outside an engine session: the bind reads the request from it
  bindUpdateEng(k, S, n, r.nlPc, r.keyDelta, FL.eng.req);
  if (FLANE == 0) {
    k.jcHasPctx[job] = 1; k.pcNode[job] = n; k.pcSap[job] = prio; k.pcPap[job] = ASCHED_EVICTED_PRIORITY; k.pcMethod[job] = ASCHED_METHOD_NO_PREEMPTION;
    k.jobNode[job] = n; k.jobCutoff[job] = cutoff; k.schedAtPrio[job] = prio;
  }
  if (c.txn.active) undoPush(d, U_ADD, job, n, oldCutoff);
  if (!fastAfterBind(k, S, r, n, h, cand)) fastDrop(d);
  if (S.statL0Max > RS.statL0Max) RS.statL0Max = S.statL0Max;
  c.l1Dirty = 1;
  return true;
}
// ------------------------------------------------------------------------------------------------ launch persistence
DEV void fastLoad(Dev& d) {  // kernel start: rebuild the LDS side from HBM
  FL.l0Count = 0; FL.eng.cleanFrom = 0;
  fastPassReset();
  if (!d.f.structOk || !RS.fastActive) return;
  const FastK k = fastKRef(d);
  candResetAll(d, d.candPosSave);
  if (k.maskMode) FOR_LANES(s, k.S) SHT(s) = d.shapeTab[s];
  int cnt = RS.l0SaveCount;
  for (int i = 0; i < cnt; i++) {
    int n = d.l0Save[i];
    uint64_t key = KKEY(k, 0, n), cls = d.nodeCls[n], cls2 = 0; int64_t ex0 = k.E > 0 ? KAL(k, 0, k.ex0col, n) : 0, ex1 = k.E > 1 ? KAL(k, 0, k.ex1col, n) : 0;
    if (k.maskMode) capMask2(k, cls, key, ex0, ex1, &cls, &cls2);
    l0Insert(k, n, key, ex0, ex1, cls, cls2);
  }
}
DEV void fastSave(Dev& d) {  // kernel end
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_L0_STATS") && d.f.structOk && RS.fastActive && d.shapeTab) {
    const FastK k = fastKRef(d);
    int dead = 0;
    for (int i = 0; i < FL.l0Count; i++) {
      bool any = false;
      for (int s = 0; s < d.f.F && !any; s++) { const ShapeReq q = d.shapeTab[s]; any = !q.never && fieldsGE(k, FL.l0Key[i], q.fieldMin) && q.ex0 <= FL.l0Ex0[i] && q.ex1 <= FL.l0Ex1[i]; }
      if (!any) dead++;
    }
    fprintf(stderr, "L0 at kernel end: %d entries, %d fit no shape (max %d)\n", FL.l0Count, dead, RS.statL0Max);
  }
#endif
  if (!d.f.structOk || !RS.fastActive) { RS.l0SaveCount = 0; return; }
  candSaveAll(d, d.candPosSave);
  for (int i = 0; i < FL.l0Count; i++) d.l0Save[i] = FL.l0Node[i];
  RS.l0SaveCount = FL.l0Count;
}

// ------------------------------------------------------------------------------------------------ queue iterator, fast
// QueuedGangIterator.Peek (queue_scheduler.go:376-432) for a whole gang whose members lie next to each other at the queue's cursor — the common
// layout: a gang is submitted as one batch.  One lane per member: the jctx reset of jobItNext, the member list, the gang's total request (LDS adds),
// then updatePQItem (:636-686) for the gang.  Anything else (members apart, a cardinality that does not match the stored members, mixed priority
// classes, an unfeasible key among the members, the lookback limit falling inside the gang, skip mode) is left to the generic iterator, untouched.
DEV bool fastPeekGang(Dev& d, KREF k, FastS& S, const FastCtx& fc, int q, QHot& f, int pos, KeyOut* ko) {
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_NO_GANG_PEEK")) return false;
#endif
  if (!fc.stream || fc.replay || f.effValid) return false;
  int job0 = UNI32(k.queuedJobs[pos]);
  int g = UNI32(d.jGang[job0]);
  if (g < 0) return false;
  int card = UNI32(d.jGangCard[job0]), off = UNI32(d.gangOff[g]);
  if (card < 2 || card > 64 || card != UNI32(d.gangOff[g + 1]) - off || pos + card > f.qEnd || UNI32(d.gangSeen[g]) != 0) return false;
  if (fc.maxLookback != 0 && (uint32_t)(f.itJobsSeen + card - 1) >= fc.maxLookback) return false;
  int pc0 = UNI32(d.jPc[job0]);
  bool skip = fc.skipKnown && S.numUnfeasible > 0;
  int bad = 0;
  FOR_LANES(m, card) {
    int j = k.queuedJobs[pos + m];
    if (d.jGang[j] != g || d.jPc[j] != pc0 || d.jNode0[j] >= 0 || d.jGangCard[j] != card || (skip && k.unfeasible[d.jShape[j]])) bad = 1;
  }
  if (LANES_ANY(bad)) return false;
  FOR_LANES(x, MAXR) FL.tmpX[x] = 0;
  FOR_LANES(m, card) {
    int j = k.queuedJobs[pos + m];
    resetJctxForQueued(d, j);
    d.gangArr[off + m] = j;
    const int64_t* rq = JREQ(d, j);
    for (int r = 0; r < k.R; r++) if (rq[r]) LDS_ADD64(FL.tmpX[r], (uint64_t)rq[r]);
  }
  int64_t tot[MAXR], alloc[MAXR], with[MAXR];
  for (int r = 0; r < k.R; r++) {
    tot[r] = (int64_t)UNI64(FL.tmpX[r]);
    alloc[r] = UNI64(FL.qAlloc[q][r]) + UNI64(FL.qPenalty[q][r]); with[r] = alloc[r] + tot[r];
  }
  if (FLANE == 0) { d.gangSeen[g] = card; d.gangAllEvicted[g] = 0; for (int r = 0; r < k.R; r++) d.gangTotal[(size_t)g * k.R + r] = tot[r]; }
  double w = f.weight;
  double pr = UNID(drf(d, with) / w), cu = UNID(drf(d, alloc) / w), sz = UNID(drf(d, tot) * w);
  int32_t p = UNI32(d.cfg.pcPriority[pc0]);
  f.itQi = pos + card; f.itJobsSeen += card;
  f.itNext = -(g + 2); f.gctx = -(g + 2); f.headFast = 0; f.headKind = 1; f.headIdx = -1; f.headPos = -1;
  f.proposed = pr; f.current = cu; f.size = sz; f.pcPrio = p; f.schedPrio = p;
  *ko = packItemKeys(fc.preferLarge, q, p, pr, cu, sz, f.budget);
  return true;
}

// costItClear(top) (queue_scheduler.go:595-606) + QueuedGangIterator.Peek (:376-432) + updatePQItem (:636-686) for the
// next single job of queue q, from the prefetch window; gang members and rare iterator states go to the generic code.
// `f` is the caller's register copy of FL.hot[q]; changed fields are stored back here.  Returns false when the generic
// updateAndPush must continue for queue q (state left exactly where costItClear leaves it).
DEV bool fastAdvance(Dev& d, KREF k, FastS& S, const FastCtx& fc, int q, QHot& f, KeyOut* ko) {
  FL.inHeap[q] = 0; f.itNext = -1; f.headFast = 0;
  ko->valid = 0;
  bool ok = true, haveHead = false;
  for (;;) {
    bool generic = fc.maxLookback != 0 && !f.itGangOnlyEv && (uint32_t)f.itJobsSeen >= fc.maxLookback;  // queue_scheduler.go:434-444
    int kind = -1, pos = 0, end = 0;  // what jobItNext (jobiteration.go:179-228) yields next, not yet consumed
    if (!generic) {
      if (f.itStage == 0) {
        if (f.itEi < f.evEnd) { kind = 0; pos = f.itEi; end = f.evEnd; }
        else f.itStage = 1;
      }
      if (kind < 0 && !(f.itJobOnlyEv || !fc.withQueued) && f.itQi < f.qEnd) { kind = 1; pos = f.itQi; end = f.qEnd; }
      if (kind < 0) { f.gctx = -1; f.proposed = f.current = f.size = 0; break; }
    }
    if (!generic && kind == 1 && f.evApplied < f.evDone) {  // first queued job after cheap evicted ones: their commits feed the queue's allocation
      applyEvictedRange(d, q, f.evApplied, f.evDone);
      S.numEvictedJobs -= f.evDone - f.evApplied; f.evApplied = f.evDone;
    }
    // costs precomputed for the whole evicted stream (B_EVKEYS): no job record, no DRF evaluation.  They assume that every earlier evicted job of
    // the queue came back — true while evicted jobs always return; after a preemption-based bind an evicted job may fail or be skipped as
    // preempted, and the costs are evaluated from the queue's actual allocation again (window path below)
    // (round 5) After a preemption-based bind the costs stay valid PER QUEUE until one of the queue's own evicted jobs fails to come back: fastIter validates such a head
    // job by job (its preempted mark, its node's priority -2 columns) and switches the queue off (evCheapOff) at the first one that is skipped or does not fit.
    if (!generic && kind == 0 && f.evCheap && (fc.replay || fc.evStatic)) {  // (the replay places every evicted job by definition)
      if (pos >= UNI32(FL.evValidEnd[q])) {   // the queue's stream was invalidated at or before this position: recompute the next stretch from the queue's allocation as it is
        if (f.evApplied < f.evDone) { applyEvictedRange(d, q, f.evApplied, f.evDone); S.numEvictedJobs -= f.evDone - f.evApplied; f.evApplied = f.evDone; }
        evRepair(d, k, q, pos, end - pos > 64 ? pos + 64 : end);
        f.evCheap = 2; f.ewCount = 0;
      }
      if (!(pos >= f.ewStart && pos < f.ewStart + f.ewCount)) {
        int cnt = end - pos; if (cnt > WIN) cnt = WIN;
        evWinRefill(k, q, pos, cnt);
        f.ewStart = pos; f.ewCount = cnt;
        S.statRefills++;
      }
      SEG(2);
      EvKey e = FL.evWin[q][pos - f.ewStart];
      e.proposed = UNID(e.proposed); e.current = UNID(e.current); e.size = UNID(e.size); e.pcPrio = UNI32(e.pcPrio); e.job = UNI32(e.job);
      f.itEi = pos + 1;
      f.itNext = e.job; f.gctx = e.job; f.headKind = 2; f.headIdx = -1; f.headPos = pos; f.headFast = 1;
      f.proposed = e.proposed; f.current = e.current; f.size = e.size; f.pcPrio = e.pcPrio; f.schedPrio = e.pcPrio;
      *ko = packItemKeys(fc.preferLarge, q, e.pcPrio, e.proposed, e.current, e.size, f.budget);
      haveHead = true;
      SEG(3);
      break;
    }
    int w = 0;
    if (!generic) {
      if (!(f.winKind == kind && pos >= f.winStart && pos < f.winStart + f.winCount)) {
        int cnt = end - pos; if (cnt > WIN) cnt = WIN;
        winRefill(k, q, kind, pos, cnt);
        f.winKind = kind; f.winStart = pos; f.winCount = cnt;
        S.statRefills++;
      }
      w = pos - f.winStart;
      if (UNI32(FL.winRec[q][w].gang) >= 0) {
        if (kind == 1 && fastPeekGang(d, k, S, fc, q, f, pos, ko)) { haveHead = true; break; }
        generic = true;
      }
    }
    SEG(10);
    if (generic) { ok = false; break; }  // the generic iterator continues from the same state
    int job = UNI32(FL.winJob[q][w]);
    if (kind == 0) f.itEi = pos + 1;
    else {
      f.itQi = pos + 1; f.itJobsSeen++;
      // JobSchedulingContextFromJob (context/job.go:149-158): a queued non-gang job still has exactly the jctx that
      // round_prepare's reset gave it (fast iterations are off once a NodeDb-level call touched per-job state): nothing to store
    }
    if (fc.skipKnown && S.numUnfeasible > 0 && kind == 1 && k.unfeasible[UNI32(d.jShape[job])]) {  // queue_scheduler.go:398-413 (the scheduling key's shape: the record carries the fit shape)
      if (FLANE == 0) {
        k.jcHasPctx[job] = 1; k.pcNode[job] = -1; k.pcMethod[job] = ASCHED_METHOD_NONE;
        k.jobFlags[job] = (uint8_t)(k.jobFlags[job] | F_UNSUCCESSFUL);  // sctx.AddJobSchedulingContext of a failed job
        k.jcReason[job] = ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY;
      }
      {   // the jobs behind it that Peek skips as well, one lane each (round_ctl.h gangItPeek has the same step)
        int max = end - (pos + 1);
        if (fc.maxLookback != 0 && !f.itGangOnlyEv) { int64_t lim = (int64_t)fc.maxLookback - f.itJobsSeen; if (lim < max) max = lim < 0 ? 0 : (int)lim; }
        if (max >= 4) { int n = (max >= SKIP_BULK_MIN && !S.engLive) ? skipUnfeasibleBulk(d, pos + 1, max) : skipUnfeasibleRun(d, pos + 1, max); f.itQi += n; f.itJobsSeen += n; }
      }
      continue;
    }
    f.itNext = job; f.gctx = job;
    f.headKind = kind; f.headIdx = UNI32(FL.winIdx[q][w]); f.headPos = kind == 0 ? pos : -1; f.headFast = 1;
    headFromWindow(q, w);
    SEG(4);
    double pr, cu, sz;
    drf3(d, q, w, fc.replay != 0, f.weight, &pr, &cu, &sz);
    pr = UNID(pr); cu = UNID(cu); sz = UNID(sz);
    SEG(5);
    int32_t p = UNI32(FL.winRec[q][w].pcPrio);
    // schedulingPriority (queue_scheduler.go:660-672): pctx.ScheduledAtPriority | run.ScheduledAtPriority | PC priority.  An evicted job's jctx is
    // fresh (no pctx, eviction.go:246-253); it has a run unless it was scheduled in this very round (phase-3 eviction of a new job)
    int32_t sp = (kind == 0 && UNI32(FL.winRec[q][w].node0) >= 0) ? UNI32(FL.winRec[q][w].runPrio) : p;
    f.proposed = pr; f.current = cu; f.size = sz; f.pcPrio = p; f.schedPrio = sp;
    *ko = packItemKeys(fc.preferLarge, q, fc.compareSchedPrio ? sp : p, pr, cu, sz, f.budget);
    if (f.effValid) {  // skip mode: a head cannot be served before anything that precedes it in its queue (heap merge == order by running maximum)
      PackedKey own, eff; own.A = ko->A; own.X = ko->X; own.Y = ko->Y; eff.A = UNI32(FL.effA[q]); eff.X = UNI64(FL.effX[q]); eff.Y = UNI64(FL.effY[q]);
      if (packedLess(own, 0, eff, 0)) { ko->A = eff.A; ko->X = eff.X; ko->Y = eff.Y; FL.kA[q] = eff.A; FL.kX[q] = eff.X; FL.kY[q] = eff.Y; }
      else { FL.effA[q] = own.A; FL.effX[q] = own.X; FL.effY[q] = own.Y; }
    }
    haveHead = true;
    break;
  }
  // store back what this iteration may have changed
  QHot& o = FL.hot[q];
  o.tokens = f.tokens; o.proposed = f.proposed; o.current = f.current; o.size = f.size;
  o.itEi = f.itEi; o.itQi = f.itQi; o.itStage = f.itStage; o.itJobsSeen = f.itJobsSeen; o.itNext = f.itNext; o.gctx = f.gctx;
  o.pcPrio = f.pcPrio; o.schedPrio = f.schedPrio; o.headFast = f.headFast; o.headKind = f.headKind; o.headIdx = f.headIdx;
  o.winKind = f.winKind; o.winStart = f.winStart; o.winCount = f.winCount;
  o.evApplied = f.evApplied; o.evDone = f.evDone; o.ewStart = f.ewStart; o.ewCount = f.ewCount; o.headPos = f.headPos;
  o.effValid = f.effValid; o.skipStart = f.skipStart; o.itJobOnlyEv = f.itJobOnlyEv; o.itGangOnlyEv = f.itGangOnlyEv;
  if (haveHead) FL.inHeap[q] = 1;
  SEG(6);
  return ok;
}

// Node side of one queued-job iteration (the second wave on the device): first fit at priority -2 for the job in the mailbox,
// BindJobToNode (nodedb.go:1046-1068), the job's result fields, level-0 bookkeeping.  0 = does not fit (nothing touched),
// 1 = bound, 2 = bound and the L0 list overflowed (the caller drops the structure).
// BindJobToNode (nodedb.go:1046-1068) + the job's result fields: HBM only — no-return atomics on the node's planes and keys, plain stores per job
DEV void bindJob(KREF k, FastS& ES, int n, int nl, uint64_t keyDelta, const int64_t* req, int job, int32_t prio, int32_t cutoff) {
  bindUpdateEng(k, ES, n, nl, keyDelta, req);
  ESEG(3);
  if (FLANE == 0) {  // jcReason, jobEvictedOnNode, inSchedAndEvicted are still 0 for a queued job
    k.jcHasPctx[job] = 1; k.pcNode[job] = n; k.pcSap[job] = prio;
    k.jobNode[job] = n; k.jobCutoff[job] = cutoff; k.schedAtPrio[job] = prio;
    k.pcPap[job] = ASCHED_EVICTED_PRIORITY; k.pcMethod[job] = ASCHED_METHOD_NO_PREEMPTION; k.jobFlags[job] = F_SUCCESSFUL; k.inScheduled[job] = 1;
  }
}
DEV int engineServeAt(Dev& d, KREF k, FastS& ES, const JobTail& tailSrc, const int64_t* reqSrc, int job, int32_t prio, int32_t cutoff, int nl, int ringIdx = -1) {
  // hard timeout / cancel (queue_scheduler.go:105-112): the node engine has the slack of the two waves, so IT reads the host-mapped word (every
  // 256 jobs; a read crosses PCIe) and answers "no node" without touching anything: the control wave takes the iteration back, leaves the fast
  // loop, finds the flag and raises ASCHED_ERR_TIMEOUT.  The control wave's loop carries no extra instruction for this.
  if ((++ES.engSeq & 255) == 0 && cancelRequested(d)) { if (FLANE == 0) FL.eng.cancel = 1; LANE0_PUBLISHED(); return 0; }
  JobTail r = tailSrc;
  uniJobTail(r);
  FitHandle h; h.src = 0; h.slot = -1;
  CandRec cand; cand.pos = 0; cand.node = -1; cand.key = 0; cand.cls = 0; cand.ex0 = cand.ex1 = 0; cand.pad = 0;
  ESEG(1);
  int n = fastFirstFit(k, ES, r, &h, &cand);
  ESEG(2);
  if (n < 0) return 0;
#if !defined(ASCHED_HOSTSIM)
  if (ringIdx >= 0) { if (FLANE == 0) RREC(ringIdx).node0 = n; LANE0_PUBLISHED(); }   // stream run: the bind wave issues the HBM side (bindJob) from the ring entry
  else
#else
  if (ringIdx >= 0 && (FL.eng.bindHold || hsBindLag())) RREC(ringIdx).node0 = n;     // (serial build: binds at once unless they are held for a gang's verdict — or HS_RING_LAG makes the bind side lag, fast_serial.h)
  else
#endif
  bindJob(k, ES, n, nl, r.keyDelta, reqSrc, job, prio, cutoff);
  ESEG(4);
  bool okAb = fastAfterBind(k, ES, r, n, h, cand);
  ESEG(5);
  return okAb ? 1 : 2;
}
DEV int engineServe(Dev& d, KREF k, FastS& ES) {   // the job in the mailbox
  return engineServeAt(d, k, ES, FL.eng.tail, FL.eng.req, UNI32(FL.eng.job), UNI32(FL.eng.prio), UNI32(FL.eng.cutoff), UNI32(FL.eng.nl));
}
DEV int engineServeRing(Dev& d, KREF k, FastS& ES, int i) {   // ring entry i of a stream run, read in place
  const JobRec& r = RREC(i);
  int32_t p = UNI32(r.pcPrio);
  return engineServeAt(d, k, ES, *(const JobTail*)&r.keyDelta, r.req, UNI32(RJOB(i)), p, UNI32((int)r.preemptible) ? p : NONPREEMPTIBLE_CUTOFF, UNI32((int)r.nlPc), i);
}
#ifdef ASCHED_HOSTSIM
#include "fast_serial_engine.h"   // tests/hostsim/: the node engine of the serial build
#endif

// head of queue q was peeked by the generic code: fetch its record (one burst) and classify it
DEV void fastLoadHead(KREF k, int q, int job, QHot& f) {
  loadHeadRec(k, q, job);
  int ev = UNI32((int)k.jcEvicted[job]);
  f.headKind = ev ? 0 : 1;
  f.headIdx = ev ? UNI32(k.evIndexOfJob[job]) : -1;
  f.headPos = ev ? f.itEi - 1 : -1;
  f.headFast = 1;
}


DEV void qlPutWin(int q, const QHot& f) { QHot& o = FL.hot[q]; o.winKind = f.winKind; o.winStart = f.winStart; o.winCount = f.winCount; }

// One QueueScheduler iteration (queue_scheduler.go:94-304 body) for the head of queue `top` when it is a single job that
// (a) is queued and fits at priority -2 or (b) is a phase-1-evicted job returning to its node.  Returns 0 WITHOUT side
// effects when the iteration needs the generic code (any constraint failing, preemption, gangs, ...); 1 = done;
// 2 = done, but the queue's next head must be produced by the generic updateAndPush; 8 = nothing done: a queued job that passed every constraint
// and has no node at priority -2 (fastPreemptIter takes it from there).  Fast mode only.
#ifndef ENG_START_AFTER
#define ENG_START_AFTER 3
#endif
DEV int fastIter(Dev& d, KREF k, FastS& S, const FastCtx& fc, int top, KeyOut* ko) {
  int q = top;
  if (k.hasPcLimit) return 0;  // per-queue per-priority-class caps: generic
  QHot f = FL.hot[q];
  uniQHot(f);
  if (f.sLen) { if (FLANE == 0) { FL.hot[q].sLen = 0; FL.hot[q].sPos = 0; } LANE0_PUBLISHED(); f.sLen = 0; f.sPos = 0; }   // the queue is served outside a stream run: its stream no longer describes it
  int job = f.gctx;
  if (f.headFast && f.headKind == 2) {  // evicted job with precomputed costs
    SEG(1);
    if (!fc.evStatic) return 0;  // generic (it re-reads everything from HBM; pending commits are flushed on the way)
    bool cheapOk = S.lvl0NonNeg && S.numPreemptedMarks == 0;
    if (!cheapOk) {
      // Jobs have been preempted in this round (marks), or some node's priority -2 column is negative (an urgency preemption overdrew it).  "Evicted jobs always
      // return" still holds for THIS job when it carries no preempted mark (it is still on its node) and no priority -2 column of ITS node is negative:
      // alloc[level] - alloc[-2] >= req is bucket arithmetic of one node.  Then the precomputed costs are the right ones and the commit can wait
      // (applyEvictedRange; fastPreemptIter and fastEnterGeneric apply what is pending before any cascade reads the planes).  Otherwise the head becomes an ordinary
      // evicted head and the dynamic code below decides (skip, "does not fit", or a rebind with the node's row read).
      loadHeadRec(k, q, job);
      LANE0_PUBLISHED();
      JobTail hr = FL.headTail[q];
      uniJobTail(hr);
      const EvDyn cl = evCleanLoad(k, job, hr.node0, S.numPreemptedMarks != 0, !S.lvl0NonNeg);
      cheapOk = !cl.preempted && cl.fits && !(hr.never & 2);
#ifdef ASCHED_HOSTSIM
      { static long okN = 0, badN = 0; static const bool st = getenv("HS_EV_STATS") != nullptr; if (st) { (cheapOk ? okN : badN)++; if (((okN + badN) % 100) == 0) fprintf(stderr, "validated evicted heads: %ld cheap, %ld to the dynamic path\n", okN, badN); } }
#endif
      if (!cheapOk) {
        f.headKind = 0; f.headIdx = UNI32(k.evIdxByPos[f.headPos]); f.headFast = 1;
        if (FLANE == 0) { FL.hot[q].headKind = 0; FL.hot[q].headIdx = f.headIdx; FL.hot[q].headFast = 1; }
        LANE0_PUBLISHED();
      }
    }
    if (cheapOk) {
      f.evDone = f.headPos + 1;  // served; its commit is deferred (applyEvictedRange)
      return fastAdvance(d, k, S, fc, q, f, ko) ? 1 : 2;
    }
  }
  if (!f.headFast) { fastLoadHead(k, q, job, f); FL.hot[q].headFast = 1; FL.hot[q].headKind = f.headKind; FL.hot[q].headIdx = f.headIdx; FL.hot[q].headPos = f.headPos; }
  JobTail r = FL.headTail[q];
  uniJobTail(r);
  bool ev = f.headKind == 0;
  SEG(1);
  int pcx = r.pc;
  int32_t prio;
  int n;
  FitHandle h; h.src = 0; h.slot = -1;
  CandRec cand; cand.pos = 0; cand.node = -1; cand.key = 0; cand.cls = 0; cand.ex0 = cand.ex1 = 0; cand.pad = 0;
  bool evInRound = true, wasPre = true; int evNl = 0; int32_t evCutoff = 0; (void)evCutoff;
  bool useEngine = false;
  if (!ev) {
    if (!S.fastActive) return 0;
    if (k.anyRoundLimit && roundLimitExceeded(d, k)) return 0;  // CheckRoundConstraints (constraints.go:113-119)
    if (f.cordoned || S.globalTokens < 1 || S.globalBurst < 1 || f.burst < 1) return 0;  // CheckJobConstraints (:121-157)
    if (f.tokens < 1) {
      // QueueRateLimitExceeded: a queue-terminal reason (constraints.go:25-58).  The reference adds the gang to the context, fails the
      // constraint check, takes it out again and records it as failed (gang_scheduler.go:63-98: net effect = the job's reason and
      // "unsuccessful" flag), pops the queue's item, which peeks the next job, and then restricts the queue to evicted jobs: the
      // peeked job is stashed and the queue leaves the heap (queue_scheduler.go:213-220, 338-350, 546-566).
      if (S.numUnfeasible > 0 || f.itStage == 0) return 0;
      bool lookback = fc.maxLookback != 0 && !f.itGangOnlyEv && (uint32_t)f.itJobsSeen >= fc.maxLookback;
      int nextJob = -1;
      if (!lookback && !f.itJobOnlyEv && f.itQi < f.qEnd) {
        int pos = f.itQi;
        if (!(f.winKind == 1 && pos >= f.winStart && pos < f.winStart + f.winCount)) {
          int cnt = f.qEnd - pos; if (cnt > WIN) cnt = WIN;
          winRefill(k, q, 1, pos, cnt);
          f.winKind = 1; f.winStart = pos; f.winCount = cnt;
          S.statRefills++;
        }
        int w = pos - f.winStart;
        if (UNI32(FL.winRec[q][w].gang) >= 0) { qlPutWin(q, f); return 0; }  // the generic iterator assembles gangs (only the window moved: harmless)
        nextJob = UNI32(FL.winJob[q][w]);
        f.itQi = pos + 1; f.itJobsSeen++;
      }
      if (FLANE == 0) { k.jcReason[job] = ASCHED_REASON_QUEUE_RATE_LIMIT; k.jobFlags[job] = F_UNSUCCESSFUL; }
      d.itStashed[q] = nextJob; d.onlyEvByQueue[q] = 1;
      f.itGangOnlyEv = 1; f.itJobOnlyEv = 1;
      f.itNext = -1; f.gctx = -1; f.headFast = 0; f.proposed = f.current = f.size = 0;
      FL.inHeap[q] = 0;
      QHot& o = FL.hot[q];
      o.itQi = f.itQi; o.itJobsSeen = f.itJobsSeen; o.itNext = -1; o.gctx = -1; o.headFast = 0; o.proposed = o.current = o.size = 0;
      o.itGangOnlyEv = 1; o.itJobOnlyEv = 1; o.winKind = f.winKind; o.winStart = f.winStart; o.winCount = f.winCount;
      ko->valid = 0;
      return 1;
    }
    if (k.anyDisallowed && headRequestsDisallowed(d, k, q)) return 0;
    if (k.disableHome) return 0;
    prio = r.pcPrio;
    SEG(8);
    // the node engine is worth its start / stop (a workgroup-wide hand-shake and a release fence) when queued jobs keep fitting: where most of them
    // need preemption (an oversubscribed cluster) this wave places the odd fitting one itself and the engine stays down
    useEngine = fc.engine && (S.engLive || S.inlineStreak >= ENG_START_AFTER);
    if (useEngine) {
      // two-wave iteration: the node engine takes first fit + bind; this wave goes on with the queue side assuming the job fits
      if (!S.engLive) { engineStart(d, S); S.engLive = 1; }
      SEG(9);
      enginePost(d, k, S, job, q, pcx, prio, r.preemptible ? prio : NONPREEMPTIBLE_CUTOFF, r.nlPc);  // also saves queue q's state: FL.hot[q] is still as of the start of this iteration
      n = -1;
      // (round 5) fastAdvance may record jobs it skips as "known-unfeasible key" (queue_scheduler.go:398-413).  Rounds 2-4 waited for the engine's verdict before the queue
      // side whenever such keys existed — which made every two-wave iteration of a round with ONE failed gang synchronous (configs[3]: 16 k instead of 9 k ticks per job,
      // profiles/r05j_gangs_segments.txt).  The records are taken back with the rest of the iteration now (fastRollback: the skipped jobs lie between the backup's cursor and
      // the current one and held their round_prepare defaults).  The optimiser's candidate iteration keeps the wait (jcReason doubles as its report there).
#ifdef ASCHED_SYNC_UNFEASIBLE
      if (fc.skipKnown && S.numUnfeasible > 0) {
#else
      if (fc.skipKnown && S.numUnfeasible > 0 && UNI32(RS.optMode)) {
#endif
        int v = engineWait(S);
        if (v == 0) return 8;
        if (v == 2) { S.fastActive = 0; fastDrop(d); }
      } else S.engPend = q;
    } else {
      n = fastFirstFit(k, S, r, &h, &cand);
      if (n < 0) return 8;  // every constraint holds, no node at priority -2: the cascade (gate, fair-share, urgency) decides — fastPreemptIter or the generic loop
      S.inlineStreak++;
    }
    S.numNodeQueries++;
  } else {
    // An evicted job returns to its node (nodedb.go:583-594, dynamic check only :897-906).  Phase-1 evictions while no priority -2 column
    // is negative: alloc[level] >= alloc[-2] + req >= req on every column (bucket arithmetic, DESIGN.md "Evicted jobs always return"), no
    // node read.  Otherwise — a preemption-based bind overdrew some priority -2 column, or the job was evicted by the oversubscribed evictor
    // (pass 2: node / priority / bookkeeping are this round's, not the original run's) — the node's current allocatable is read.
    prio = r.runPrio; n = r.node0;
#ifdef ASCHED_HOSTSIM
    { static long slowN = 0, cheapQ = 0; static const bool st = getenv("HS_EV_STATS") != nullptr; if (st) { slowN++; if (f.evCheap) cheapQ++; if ((slowN % 500) == 0) fprintf(stderr, "dynamic evicted heads: %ld (queue still cheap: %ld) marks %d lvl0NonNeg %d\n", slowN, cheapQ, S.numPreemptedMarks, S.lvl0NonNeg); } }
#endif
    if (r.never & 2) return 0;   // a request off the index grid: the bind below takes keyDelta off the node's keys, which is exact for multiples of the resolution only
#ifdef ASCHED_HOSTSIM
    if (getenv("HS_NO_DYN_EV") && (!fc.evStatic || !S.lvl0NonNeg || S.numPreemptedMarks != 0)) return 0;
    if (getenv("HS_NO_DYN_EV2") && !fc.evStatic) return 0;
    if (getenv("HS_NO_DYN_EV3") && fc.evStatic && (!S.lvl0NonNeg || S.numPreemptedMarks != 0)) return 0;
#endif
    if (!fc.evStatic) {
      n = UNI32(k.jcAssigned[job]); prio = UNI32(k.schedAtPrio[job]);
      evInRound = (UNI32((int)k.jobFlags[job]) & F_EVICTED) != 0; wasPre = UNI32((int)k.inPreempted[job]) != 0;
      if (n < 0 || prio == NO_PRIORITY) return 0;
    }
    // After a preemption-based bind a returning evicted job needs its preempted mark and its node's current row: three HBM reads that used to be three round trips of the control
    // wave, one after the other (mark, node flags, planes) — 340 000 such iterations in a configs[4] round (profiles/r04h_wide_pass_timeline.txt).  One round trip now.
    int dynLevel = -1;
    const bool dynPin = fc.evStatic && !S.lvl0NonNeg;
    if (dynPin) for (int l = 0; l < MAXP; l++) if (l < k.P && k.prios[l] == prio) dynLevel = l;
    const EvDyn dyn = evDynLoad(k, q, job, n, dynLevel, S.numPreemptedMarks != 0, dynPin && dynLevel >= 0);
    if (S.numPreemptedMarks != 0 && dyn.preempted != 0) {
      // a job preempted earlier in this round is skipped: Clear() + continue, no scheduling attempt, not a counted iteration
      // (queue_scheduler.go:150-156); a queued job never carries the mark
      if (f.evApplied < f.evDone) { applyEvictedRange(d, q, f.evApplied, f.evDone); S.numEvictedJobs -= f.evDone - f.evApplied; }
      f.evApplied = f.evDone = f.headPos + 1;
      if (f.evCheap && !(f.evCheap == 2 && f.headPos < UNI32(FL.evValidEnd[q]) && UNI32((int)EV_EXCL(d)[f.headPos]))) {   // the queue's later precomputed costs counted this job in (unless the repair that made them knew the mark)
        if (evRepairOk(d, fc)) evInvalidateFrom(d, q, f, f.headPos + 1); else evCheapOff(d, q, f);
      }
      return (fastAdvance(d, k, S, fc, q, f, ko) ? 1 : 2) | 4;
    }
    if (!fc.evStatic || !S.lvl0NonNeg) {
      int level = -1;
      for (int l = 0; l < MAXP; l++) if (l < k.P && k.prios[l] == prio) level = l;
      if (level < 0) return 0;
      if (!(dynPin ? dyn.fits != 0 : pinnedNodeFits(k, q, n, level))) {
        // The job does not fit on its node any more (urgency preemption took the space): SelectNodeForJobWithTxn returns no node,
        // the gang fails with "job does not fit on any node" (gang_scheduler.go:229-262, 63-98).  Net effect of AddGangSchedulingContext /
        // EvictGang / re-add-as-failed on the scheduling context (scheduling.go:391-449, 551-572; queue.go:231-265, 351-386): the job's
        // "evicted in this round" mark goes (its requests leave EvictedResourcesByPriorityClass), it is recorded unsuccessful; every
        // other sum is back where it was.  No unfeasible-key registration: an evicted job's key is not valid (context/job.go:104-109).
        if (f.evApplied < f.evDone) { applyEvictedRange(d, q, f.evApplied, f.evDone); S.numEvictedJobs -= f.evDone - f.evApplied; }
        f.evApplied = f.evDone = f.headPos + 1;
        if (f.evCheap) { if (evRepairOk(d, fc)) evInvalidateFrom(d, q, f, f.headPos + 1); else evCheapOff(d, q, f); }   // the queue's later precomputed costs counted this job in
        if (d.excl) exclPinnedFast(d, k, q, job, n, level);   // (asched_excluded_nodes: the dynamic reason on its node; inline — a call here costs the whole loop registers)
        if (evInRound) FOR_LANES(x, k.R) {
          int64_t v = FL.headReq[q][x];
          if (v) {
#ifdef ASCHED_HOSTSIM
            k.qEvictedByPc[((size_t)q * k.npc + pcx) * k.R + x] -= v;
#else
            __hip_atomic_fetch_add(&k.qEvictedByPc[((size_t)q * k.npc + pcx) * k.R + x], -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
          }
        }
        if (FLANE == 0) {
          k.jcHasPctx[job] = 1; k.pcNode[job] = -1; k.pcSap[job] = prio; k.pcPap[job] = ASCHED_MIN_PRIORITY; k.pcMethod[job] = ASCHED_METHOD_NONE;
          k.jcReason[job] = ASCHED_REASON_JOB_DOES_NOT_FIT; k.jobFlags[job] = F_UNSUCCESSFUL;
        }
        return fastAdvance(d, k, S, fc, q, f, ko) ? 1 : 2;
      }
      evCutoff = r.preemptible ? prio : NONPREEMPTIBLE_CUTOFF;
      evNl = 0;
      for (int l = 0; l < MAXP; l++) if (l < k.P && k.prios[l] <= evCutoff) evNl = l + 1;   // levels with priority <= cutoff (sorted ascending)
    } else evNl = r.nlRun;
    if (f.evApplied < f.evDone) { applyEvictedRange(d, q, f.evApplied, f.evDone); S.numEvictedJobs -= f.evDone - f.evApplied; }
    f.evApplied = f.evDone = f.headPos + 1;  // committed right here
  }
  SEG(2);
  // ---- commit: sctx.AddGangSchedulingContext (scheduling.go:391-434).  A job that was scheduled or rescheduled earlier in this round and then
  // evicted by the oversubscribed evictor does not carry the "evicted in this round" mark (sctx.EvictJob, :551-572): it is accounted like a new job
  bool evAcct = ev && evInRound;
  accountVectors(d, k, q, pcx, evAcct, false);
  if (evAcct) S.numEvictedJobs--; else { S.numScheduledJobs++; S.numScheduledGangs++; }
  // ---- SelectNodeForJobWithTxn result + BindJobToNode (nodedb.go:538-630, 1046-1068)
  int32_t cutoff = r.preemptible ? prio : NONPREEMPTIBLE_CUTOFF;
  int nl = ev ? evNl : r.nlPc;
  bool nodeSideHere = ev || !useEngine;  // a queued job's bind, result fields and L0 upkeep are the node engine's in two-wave mode
  // evicted job: level -2 gets -req (bind) and +req (un-evict): unchanged (node.go:416-442).  The packed key subtraction cannot borrow when the job fits at
  // priority -2 (it then fits at every level) or returns while no priority -2 column is negative; an evicted job returning onto an overdrawn node was only
  // checked at ITS level — the levels below may go negative: per-field saturating subtraction there (field 0 = "negative", asched_host.inc layoutKeys)
  bool evDynamic = ev && (!fc.evStatic || !S.lvl0NonNeg);
  if (nodeSideHere) bindUpdate(k, S, n, ev ? 1 : 0, nl, q, evDynamic ? 0 : r.keyDelta);
  if (nodeSideHere && evDynamic) keySatSub(k, n, 1, nl, r.keyDelta);
  if (nodeSideHere && FLANE == 0) {
    k.jcHasPctx[job] = 1; k.pcNode[job] = n; k.pcSap[job] = prio;
    k.jobNode[job] = n; k.jobCutoff[job] = cutoff; k.schedAtPrio[job] = prio;
    if (ev) {
      k.jcReason[job] = 0; k.jobEvictedOnNode[job] = 0; k.inSchedAndEvicted[job] = 0;
      k.pcPap[job] = prio; k.pcMethod[job] = ASCHED_METHOD_RESCHEDULED; k.jobFlags[job] = evInRound ? F_RESCHEDULED : F_SUCCESSFUL;
      if (wasPre) k.inPreempted[job] = 0; else k.inScheduled[job] = 1;   // pqs.go:160-166 / :213-220
      if (!S.replayPending) { k.evTabAlive[f.headIdx] = 0; k.evIndexOfJob[job] = -1; }  // nodedb.go:441-446 (a deferred replay marks it dead itself)
    } else {  // jcReason, jobEvictedOnNode, inSchedAndEvicted are still 0 for a queued job
      k.pcPap[job] = ASCHED_EVICTED_PRIORITY; k.pcMethod[job] = ASCHED_METHOD_NO_PREEMPTION; k.jobFlags[job] = F_SUCCESSFUL; k.inScheduled[job] = 1;
    }
  }
  if (!ev) {
    if (nodeSideHere && !fastAfterBind(k, S, r, n, h, cand)) { S.fastActive = 0; fastDrop(d); }
    if (!S.globalRateInf && 1 <= S.globalBurst) S.globalTokens -= 1.0;  // gang_scheduler.go:118-123, rate.Limiter.ReserveN
    if (!f.rateInf && 1 <= f.burst) f.tokens -= 1.0;
  }
  SEG(3);
  return fastAdvance(d, k, S, fc, q, f, ko) ? 1 : 2;
}

// the jobs fastAdvance skipped behind a speculative iteration's job as "known-unfeasible key" (queue_scheduler.go:398-413): list positions [p0, p1) — the backup's cursor to the
// current one; nothing had looked at them before (the cursor only moves forward), so their records go back to round_prepare's (B_RESET_JOBS); the generic code peeks — and
// skips — them again if it gets that far.  (Out of line: rare, and nothing of the main loop's registers has to live across it.)
DEV_NOINLINE void fastUndoSkipRecords(Dev& d, int p0, int p1) {
  const FastK k = fastKRef(d);
  for (int b = p0; b < p1; b += 64) {
    FOR_LANES(x, 64) if (b + x < p1) {
      int job = k.queuedJobs[b + x];
      if (d.jcReason[job] == ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY) {
        d.jcReason[job] = 0; d.jcHasPctx[job] = 0; d.pcMethod[job] = ASCHED_METHOD_NONE;
        d.jobFlags[job] = (uint8_t)(d.jobFlags[job] & ~F_UNSUCCESSFUL);
      }
    }
  }
}
// The node engine found no node for the job of queue q's last (speculative) iteration: take the queue side of that iteration
// back — accounting, counters, tokens, the queue's head / iterator / key — so that the generic code meets the state fastIter
// would have left by returning 0.  The heap lanes are rebuilt by the caller.
DEV void fastRollback(Dev& d, KREF k, FastS& S, int q) {
#ifdef ASCHED_HOSTSIM
  if (getenv("HOSTSIM_TRACE_ROLLBACK")) fprintf(stderr, "rollback q%d\n", q);
#endif
  int curApplied = UNI32(FL.hot[q].evApplied), bkApplied = UNI32(FL.bk.hot.evApplied);
  if (curApplied > bkApplied) { applyEvictedRange(d, q, bkApplied, curApplied, -1); S.numEvictedJobs += curApplied - bkApplied; }
  accountVectorsBk(d, k, q, UNI32(FL.bk.pc), -1);
  S.numScheduledJobs--; S.numScheduledGangs--; S.numNodeQueries--;
  S.globalTokens = UNID(FL.bk.globalTokens);
  { int g = UNI32(FL.hot[q].gctx); if (g < -1 && FLANE == 0) d.gangSeen[-g - 2] = 0; }   // the iteration had assembled the gang behind its job (fastPeekGang): not seen yet
  { int p0 = UNI32(FL.bk.hot.itQi), p1 = UNI32(FL.hot[q].itQi); if (S.numUnfeasible > 0 && p1 > p0) fastUndoSkipRecords(d, p0, p1); }
  engineRestore(q);
  FL.hot[q].winKind = -1; FL.hot[q].ewCount = 0;  // the windows may have moved on: refill on demand
  S.loopIterations--; S.statFastIters--;
}

// one step of addEvictedJobsToNodeDb (preempting_queue_scheduler.go:589-639) for a single evicted job.  Fast mode only.
DEV int fastReplayStep(Dev& d, KREF k, FastS& S, const FastCtx& fc, int top, int* counter, KeyOut* ko) {
  int q = top;
  QHot f = FL.hot[q];
  uniQHot(f);
  int job = f.gctx, i = *counter;
  bool cheap = f.headFast && f.headKind == 2;
  if (!f.headFast) fastLoadHead(k, q, job, f);
  if (FLANE == 0) { k.evTabJob[i] = job; k.evTabAlive[i] = 1; k.evIndexOfJob[job] = i; }
  if (i + 1 > S.evictedTableSize) S.evictedTableSize = i + 1;
  *counter = i + 1;
  SEG(1);
  if (f.headPos >= 0) f.evApplied = f.evDone = f.headPos + 1;  // nothing is deferred in the replay: it only assigns evicted-table indices
  if (!cheap) accountVectors(d, k, q, 0, true, true);
  S.statFastReplay++;
  SEG(4);
  return fastAdvance(d, k, S, fc, q, f, ko) ? 1 : 2;
}

// ---- skip mode.  In a pass whose evicted jobs all return to their nodes (fastIter's evicted branch: no node read, nothing a
// queued job's placement depends on), the heap merge of the per-queue streams [evicted..., queued...] equals the order by each
// entry's running-maximum key, so the evicted entries need not be stepped through one by one: their commits are applied in
// bulk up front, every queue enters the heap with its first queued job under max(own key, last evicted key), and the loop
// visits queued jobs only.  The first iteration that needs the generic code ends the mode: evicted entries ordering after the
// current head are taken back (applyEvictedRange sign -1) and become ordinary heads again, which reproduces the exact state.
struct SkipDelta { int evicted, iters, refills; };  // what a cold helper changed of the scheduling-context scalars the loop keeps in registers
DEV void coldS(Dev& d, FastS& S) {
#ifdef ASCHED_HOSTSIM
  HS_POISON(S);                        // the CPU build poisons what is not set below: a cold helper that reads such a field fails the differential tests here instead of reading a
                                       // register's leftovers on the device (round 4: engLive — the bulk skip never ran on the GPU while the CPU build happened to read 0)
#endif
   S.laneL = 0; S.laneX = 0; S.numEvictedJobs = 0; S.loopIterations = 0; S.statRefills = 0; S.statScanSteps = 0; S.statL0Max = 0;
  S.numUnfeasible = RS.numUnfeasible; S.numPreemptedMarks = RS.numPreemptedMarks; S.fastActive = RS.fastActive; S.lvl0NonNeg = RS.lvl0NonNeg; S.replayPending = RS.replayPending;
  S.globalTokens = 0; S.globalBurst = 0; S.globalRateInf = 1; S.numScheduledJobs = S.numScheduledGangs = S.numNodeQueries = S.evictedTableSize = 0; S.statFastIters = S.statFastReplay = 0; S.segT = 0;
  // fastAdvance asks before it skips a long stretch of known-unfeasible keys with the bulk passes: they need every wave of the workgroup at the mailbox, and a live
  // engine wave never gets there.  Until round 5 this said 0 — "every caller of a cold helper has the engine stopped" — which is not true of the end of a stream run, of a
  // gang through the ring or of the run's events: a queue with >= SKIP_BULK_MIN (2 048) such jobs behind its stream hung the round kernel on the device (tests/soak.py
  // rounds, seed 100036: one queue, 4 489 jobs, 97 empty nodes; the CPU build runs the passes serially and cannot hang).  profiles/r05y_bulk_skip_hang.txt
  S.engLive = UNI32(FL.eng.live); S.engPend = -1;
}
DEV_NOINLINE SkipDelta fastEnterSkip(Dev& d, FastCtx fc, int Q) {
  const FastK k = fastKRef(d);
  FastS S; coldS(d, S);
  for (int q = 0; q < Q; q++) {
    QHot f = FL.hot[q];
    uniQHot(f);
    if (!f.evCheap || !UNI32((int)d.evMono[q]) || f.effValid) continue;
    int p0 = f.evDone, p1 = f.evEnd;
    if (p0 >= p1 || p0 != f.evApplied) continue;
    if (f.gctx < 0 || f.headPos != p0) continue;  // expected: the head is the first evicted job, peeked by the generic passInit
    applyEvictedRange(d, q, p0, p1);
    S.numEvictedJobs -= p1 - p0;
    EvKey e; memcpy(&e, (const char*)k.evKey + (size_t)(p1 - 1) * sizeof(EvKey), sizeof(EvKey));
    PackedKey last = packKey3(fc.preferLarge, UNI32(e.pcPrio), UNID(e.proposed), UNID(e.current), UNID(e.size), f.budget);
    FL.effA[q] = last.A; FL.effX[q] = last.X; FL.effY[q] = last.Y;
    f.effValid = 1; f.skipStart = p0; f.itEi = p1; f.evApplied = f.evDone = p1;
    KeyOut ko;
    if (!fastAdvance(d, k, S, fc, q, f, &ko)) {  // first queued job is a gang member / needs the generic iterator: leave this queue alone
      applyEvictedRange(d, q, p0, p1, -1);
      S.numEvictedJobs += p1 - p0;
      QHot g = FL.hot[q]; uniQHot(g);
      g.effValid = 0; g.itEi = p0; g.itStage = 0; g.evApplied = g.evDone = p0;
      fastAdvance(d, k, S, fc, q, g, &ko);  // the evicted head again (cheap path: cannot fail)
    }
  }
  SkipDelta r; r.evicted = S.numEvictedJobs; r.iters = S.loopIterations; r.refills = S.statRefills;
  return r;
}
// `top` / (tk, tn): the entry the exact state is rebuilt around — the current head about to go to the generic code, or the entry
// just served (every evicted entry ordering before it has been served, none after it); top < 0: everything has been served.
DEV_NOINLINE SkipDelta fastExitSkip(Dev& d, FastCtx fc, int Q, int top, PackedKey tk, uint32_t tn) {
  const FastK k = fastKRef(d);
  FastS S; coldS(d, S);
  if (top < 0) { tn = ~0u; tk.A = ~0u; tk.X = ~0ull; tk.Y = ~0ull; }
  for (int q = 0; q < Q; q++) {
    QHot f = FL.hot[q];
    uniQHot(f);
    if (!f.effValid) continue;
    f.effValid = 0;
    int b0 = f.skipStart, b1 = f.evEnd;
    uint32_t qn = (uint32_t)UNI32(FL.nameRank[q]);
    int lo = b0, hi = b1;  // first evicted entry that does NOT order before the current head (keys are non-decreasing)
    if (q == top) lo = b1;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      EvKey e; memcpy(&e, (const char*)k.evKey + (size_t)mid * sizeof(EvKey), sizeof(EvKey));
      PackedKey pk = packKey3(fc.preferLarge, UNI32(e.pcPrio), UNID(e.proposed), UNID(e.current), UNID(e.size), f.budget);
      if (packedLess(pk, qn, tk, tn)) lo = mid + 1; else hi = mid;
    }
    S.loopIterations += lo - b0;  // the reference spent one loop iteration on each of them
    if (lo == b1) { FL.hot[q].effValid = 0; continue; }
    applyEvictedRange(d, q, lo, b1, -1);
    S.numEvictedJobs += b1 - lo;
    if (f.gctx >= 0) { f.itQi -= 1; f.itJobsSeen -= 1; }  // the queued head goes back into its stream
    if (f.sLen) { f.sLen = 0; f.sPos = 0; if (FLANE == 0) { FL.hot[q].sLen = 0; FL.hot[q].sPos = 0; } LANE0_PUBLISHED(); }   // its precomputed stream started at that head
    f.itStage = 0; f.itEi = lo; f.evApplied = f.evDone = lo;
    KeyOut ko;
    fastAdvance(d, k, S, fc, q, f, &ko);  // head = evicted entry `lo` (cheap path)
  }
  SkipDelta r; r.evicted = S.numEvictedJobs; r.iters = S.loopIterations; r.refills = S.statRefills;
  return r;
}

// ---- drain.  After a terminal reason (queue_scheduler.go:205-212) every queue yields evicted jobs only and nothing can bring
// the queued jobs back (resuming needs an empty termination reason, :125-142).  If all that is left are gang-free evicted
// streams of a pass in which evicted jobs always return, the remaining iterations are order-independent: apply them in bulk.
DEV_NOINLINE SkipDelta fastDrain(Dev& d, int Q) {
  FastS S; coldS(d, S);
  SkipDelta r; r.evicted = r.iters = r.refills = 0;
  for (int q = 0; q < Q; q++) {  // eligibility first: all or nothing
    QHot f = FL.hot[q];
    uniQHot(f);
    bool headEv = f.gctx >= 0 && f.headPos >= 0;
    if (f.gctx != -1 && !headEv) return r;                       // a queued job or a gang at the head
    int p0 = headEv ? f.headPos : f.itEi;
    if (p0 < f.evEnd && !f.evCheap) return r;                      // gang members among the remaining evicted jobs
    if (!UNI32(FL.inHeap[q]) && f.gctx != -1) return r;
  }
  for (int q = 0; q < Q; q++) {
    QHot f = FL.hot[q];
    uniQHot(f);
    bool headEv = f.gctx >= 0 && f.headPos >= 0;
    int p0 = headEv ? f.headPos : f.itEi, p1 = f.evEnd;
    if (f.evApplied < f.evDone) { applyEvictedRange(d, q, f.evApplied, f.evDone); S.numEvictedJobs -= f.evDone - f.evApplied; }
    if (p0 < p1) {
      applyEvictedRange(d, q, p0, p1);
      S.numEvictedJobs -= p1 - p0; S.loopIterations += p1 - p0;
    }
    QHot& o = FL.hot[q];
    o.itEi = p1; o.itStage = 1; o.evApplied = o.evDone = p1; o.gctx = -1; o.itNext = -1; o.headFast = 0; o.headPos = -1;
    o.proposed = o.current = o.size = 0;
    FL.inHeap[q] = 0;
  }
  r.evicted = S.numEvictedJobs; r.iters = S.loopIterations;
  return r;
}


// ---- stream run.  While queued jobs fit without preemption, the queue side of an iteration — which queue is served next, with which job — depends on
// nothing the node side produces: the costs that order the queues are functions of each queue's own allocation prefix.  They are computed ahead for the
// next jobs of every queue (fastStreamPrepare: chunked prefix pass over the queued lists -> EvKey streams in HBM), and the control wave then only
// MERGES: it pops the head of its lane heap, emits (job, queue) into a ring, takes the queue's next precomputed costs from a small LDS window, packs the
// key and re-inserts the queue.  Job records are gathered four entries at a time while the merge goes on and staged in the ring (the idle prefetch
// windows); the node engine (wave 1) walks the ring: first fit at priority -2, bind, result fields, L0 upkeep — it is now the only sequential chain
// and never waits for the control wave.  Queue-side bookkeeping follows the engine's progress (accounting of an entry once it is bound); the queues'
// iterator state is materialised when the run ends: each queue's head becomes the first of its elements that was not bound, produced by the
// ordinary fastAdvance from the queue's then-current allocation.  The run ends, before any side effect that would have to be taken back, when
//   * the head of the heap is not a stream element (an evicted job, a gang, a queue that was not eligible) — its key sits in the heap like any other,
//     so nothing that orders after it is emitted;
//   * a queue's stream is used up (the next element is a gang member / known-unfeasible key / beyond the prepared length / beyond its rate-limit tokens);
//   * the global rate limiter has no token left for another job;
//   * the engine finds no node for an entry (the generic cascade decides: that entry and everything emitted after it are simply forgotten — no
//     accounting had been done for them — and the entry is its queue's head again).
// Exactness: the merge is the heap of QueueCandidateGangIteratorPQ on the very keys fastAdvance would compute (same float64 operations on the same
// prefix sums, round_run.h B_QSKEYS), the engine executes the entries in emission order, and integer accounting is order independent.
struct StreamIn { double globalTokens; int64_t globalBurst; int32_t globalRateInf, engSeq; int32_t skip, haveLast; uint32_t lastA, lastN; uint64_t lastX, lastY; int32_t resume; int32_t bulkV; };   // skip / last*: skip mode is on, key of the entry served last; bulkV > 0: the run's merged order has been computed by the bulk passes (round_merge.h), that many entries of it are valid
struct StreamOut { int executed, executedEv, pend, dropped, engSeq, emitted, refills, evicted, maxConsumed, failed, lastQ; uint32_t lastA, lastN; uint64_t lastX, lastY;
                   int gangJobs, gangs;    // gangs placed INSIDE the run (round 5) and their members: the caller accounts them like fastGangRun's (ReserveN, one scheduled gang each)
                   int event, evT; };      // event != 0: the run is PARKED at one of its two rare events (streamNestSettle / streamNestGang on queue evT) — the caller runs it and calls again with in.resume
// streams persist between runs: head of queue q == element sPos of its stream, elements [sPos, sLen) are still to come.  A queue's stream is dropped
// when anything but a stream run serves the queue (fastIter) or the generic code runs (fastQLoad).  prepare: 0 = the top queue has no stream,
// 1 = ready, 2 = some queue needs the bulk passes and allowBulk was 0 (the node engine must be stopped first: they use every wave of the workgroup)
DEV_NOINLINE int fastStreamPrepare(Dev& d, FastCtx fc, int Q, int allowed, int allowBulk, int top, int capHint);   // round_run.h
DEV_NOINLINE int fastStreamPrepareOne(Dev& d, FastCtx fc, int q, int allowed, int capHint);                          // round_run.h
DEV EvKey streamKey(KREF k, StreamLanes& sl, int q, int pos, int sLen, int kind, int base) {
  int ws = SL_GET(sl, ws, q);
  if (!(pos >= ws && pos < ws + WIN)) {
    int cnt = sLen - pos; if (cnt > WIN) cnt = WIN;
    if (kind & 1) evWinRefill(k, q, base + pos, cnt); else qsWinRefill(k, q, pos, cnt);
    SL_SET(sl, ws, q, pos);
    ws = pos;
  }
  EvKey e = FL.evWin[q][pos - ws];
  e.proposed = UNID(e.proposed); e.current = UNID(e.current); e.size = UNID(e.size); e.pcPrio = UNI32(e.pcPrio); e.job = UNI32(e.job);
  return e;
}
struct GangOut { int handled, cnt, pend, dropped, engSeq, refills, evicted, koValid; uint32_t koA; uint64_t koX, koY; };
DEV_NOINLINE GangOut fastGangRun(Dev& d, FastCtx fc, StreamIn in, int t);
struct RunState {   // the state of a run, handed by value between its three out-of-line parts (fastStreamRun: set-up and settling; streamMerge: the merge loop; streamNest: the two rare events)
  PQState pq; StreamLanes sl;
  PackedKey lastK; uint32_t lastN; int haveLast, lastQ;
  int emitted, emittedQ, acc, stageBase, stageCnt, issuedTo, fail, allowed, engSeq, sessLive, emittedPrev, doneQmid, maxMid, gangJobs, gangs, refills, evicted, pend, dropped, go;
  int ev, evT, evSLen;   // why the merge loop returned: 0 the run ends, 1 queue evT's queued stream (evSLen elements) is used up in front of a gang member, 2 the head of the heap, queue evT, is an assembled gang
  unsigned long long stageV;
};
// what the rare events of a run (streamNestSettle / streamNestGang) read and change: wave-uniform scalars only.  (A first version handed the whole RunState by value
// through a noinline call and the device then emitted 52 000 evicted entries that HEAD folds — put down to lane-private values not surviving the call boundary at the
// time, profiles/r05l; it was the miscompiled key test described at streamMerge.  The split stands on its own merits: the events are rare, large, and out of line.)
struct NestIO {
  int emitted, emittedQ, acc, fail, allowed, engSeq, sessLive, emittedPrev, doneQmid, maxMid, gangJobs, gangs, refills, evicted, pend, dropped, go;
  int replace, relane, koValid; uint32_t koA; uint64_t koX, koY;   // replace: queue t's heap entry becomes (koValid, koA, koX, koY); relane: the queue was settled — its stream lanes (and its count of done entries) start again from its record
};
// A run parks at an event and RETURNS to the round's main loop, which calls the event's function and then the run again (in.resume).  The events' functions (and what they
// call: fastGangRun, fastAdvance, fastStreamPrepareOne) clobber every VGPR and ~80 AGPRs; called from inside the run, everything the run holds had to live above that, the
// run's own footprint grew from 56 to 206 AGPRs, and the main loop — whose register allocation must keep clear of what its callees clobber — spilled more in EVERY round,
// stream run or not (configs[4] +15 %, profiles/r05z_nest_register_footprint.txt).  Parked state: lane-private words per lane + the uniform scalars, in d.qsPart (the
// carries of the preparation passes: idle during a run).
enum { PK_haveLast, PK_lastQ, PK_emitted, PK_emittedQ, PK_acc, PK_stageBase, PK_stageCnt, PK_issuedTo, PK_fail, PK_allowed, PK_engSeq, PK_sessLive, PK_emittedPrev, PK_doneQmid,
       PK_maxMid, PK_gangJobs, PK_gangs, PK_refills, PK_evicted, PK_pend, PK_dropped, PK_go, PK_ev, PK_evT, PK_evSLen, PK_lastA, PK_lastN, PK_lastX, PK_lastY,
       PK_replace, PK_relane, PK_koValid, PK_koA, PK_koX, PK_koY, PK_COUNT };
#define PK_LANE_WORDS 12
#define PK_BASE(d) ((d).qsPart)
#define PK_INTS(X) X(haveLast) X(lastQ) X(emitted) X(emittedQ) X(acc) X(stageBase) X(stageCnt) X(issuedTo) X(fail) X(allowed) X(engSeq) X(sessLive) X(emittedPrev) X(doneQmid) \
                   X(maxMid) X(gangJobs) X(gangs) X(refills) X(evicted) X(pend) X(dropped) X(go) X(ev) X(evT) X(evSLen)
#define NEST_INTS(X) X(emitted) X(emittedQ) X(acc) X(fail) X(allowed) X(engSeq) X(sessLive) X(emittedPrev) X(doneQmid) X(maxMid) X(gangJobs) X(gangs) X(refills) X(evicted) X(pend) X(dropped) X(go)
// The uniform scalars are word i of one 64-word vector (lane i owns word i: one coalesced store / load, fields read back with v_readlane); the lane-private words are stored
// word-major (word w of lane l at [w * 64 + l]).  Plain loads and stores, ordered for the compiler by workgroup-scope fences: the wave reads back its own stores through its
// own CU's L1.  (The first version used volatile accesses — system-scope on gfx950, each batch a round trip to memory: configs[3], 20 000 events per round, lost 9 %.)
#ifdef ASCHED_HOSTSIM
static thread_local RunState hsParkedRun;   // (the serial build's lane state is arrays inside the struct)
static thread_local unsigned long long hsParkedWords[64];
#define PK_WORDS_LOAD(d) 0ull
#define PK_I(w, i) ((int)(unsigned)hsParkedWords[i])
#define PK_L(w, i) (hsParkedWords[i])
#define PK_PUT(w, i, v) (hsParkedWords[i] = (unsigned long long)(v))
#define PK_WORDS_STORE(d, w, mask) do { (void)(w); } while (0)
#else
#define PK_WORDS_LOAD(d) pkWordsLoad(d)
#define PK_I(w, i) ((int)__builtin_amdgcn_readlane((int)(unsigned)(w), (i)))
#define PK_L(w, i) slGet64((w), (i))
#define PK_PUT(w, i, v) ((w) = (int)(threadIdx.x & 63) == (i) ? (unsigned long long)(v) : (w))
#define PK_WORDS_STORE(d, w, mask) pkWordsStore(d, w, mask)
__device__ static inline unsigned long long pkWordsLoad(Dev& d) {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ((const unsigned long long*)PK_BASE(d))[64 * PK_LANE_WORDS + (threadIdx.x & 63)];
}
__device__ static inline void pkWordsStore(Dev& d, unsigned long long w, unsigned long long mask) {   // lanes in `mask` store their word
  if ((mask >> (threadIdx.x & 63)) & 1) ((unsigned long long*)PK_BASE(d))[64 * PK_LANE_WORDS + (threadIdx.x & 63)] = w;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
#endif
#define PK_BIT(f) (1ull << PK_##f)
DEV void runPark(Dev& d, const RunState& m) {
  unsigned long long w = 0, mask = 0;
#define X(f) PK_PUT(w, PK_##f, (unsigned)m.f); mask |= PK_BIT(f);
  PK_INTS(X)
#undef X
  PK_PUT(w, PK_lastA, m.lastK.A); PK_PUT(w, PK_lastX, m.lastK.X); PK_PUT(w, PK_lastY, m.lastK.Y); PK_PUT(w, PK_lastN, m.lastN);
  mask |= PK_BIT(lastA) | PK_BIT(lastX) | PK_BIT(lastY) | PK_BIT(lastN);
#ifdef ASCHED_HOSTSIM
  hsParkedRun = m;
#else
  unsigned long long* sv = (unsigned long long*)PK_BASE(d) + (threadIdx.x & 63);
  sv[0 * 64] = m.pq.X; sv[1 * 64] = m.pq.Y; sv[2 * 64] = ((unsigned long long)m.pq.A << 32) | m.pq.N; sv[3 * 64] = ((unsigned long long)(unsigned)m.pq.q << 32) | (unsigned)m.pq.count;
  sv[4 * 64] = ((unsigned long long)(unsigned)m.sl.start << 32) | (unsigned)m.sl.base; sv[5 * 64] = ((unsigned long long)(unsigned)m.sl.pos << 32) | (unsigned)m.sl.len;
  sv[6 * 64] = ((unsigned long long)(unsigned)m.sl.kind << 32) | (unsigned)m.sl.ws; sv[7 * 64] = __builtin_bit_cast(unsigned long long, m.sl.budget);
  sv[8 * 64] = m.sl.effX; sv[9 * 64] = m.sl.effY; sv[10 * 64] = m.sl.effA; sv[11 * 64] = m.stageV;
#endif
  PK_WORDS_STORE(d, w, mask);
}
DEV void runUnpark(Dev& d, RunState& m) {
  const unsigned long long w = PK_WORDS_LOAD(d);
#ifdef ASCHED_HOSTSIM
  m = hsParkedRun;
#else
  const unsigned long long* sv = (const unsigned long long*)PK_BASE(d) + (threadIdx.x & 63);
  unsigned long long w2 = sv[2 * 64], w3 = sv[3 * 64], w4 = sv[4 * 64], w5 = sv[5 * 64], w6 = sv[6 * 64];
  m.pq.X = sv[0 * 64]; m.pq.Y = sv[1 * 64]; m.pq.A = (uint32_t)(w2 >> 32); m.pq.N = (uint32_t)w2; m.pq.q = (int)(w3 >> 32); m.pq.count = (int)(uint32_t)w3;
  m.sl.start = (int)(w4 >> 32); m.sl.base = (int)(uint32_t)w4; m.sl.pos = (int)(w5 >> 32); m.sl.len = (int)(uint32_t)w5;
  m.sl.kind = (int)(w6 >> 32); m.sl.ws = (int)(uint32_t)w6; m.sl.budget = __builtin_bit_cast(double, sv[7 * 64]);
  m.sl.effX = sv[8 * 64]; m.sl.effY = sv[9 * 64]; m.sl.effA = (uint32_t)sv[10 * 64]; m.stageV = sv[11 * 64];
#endif
#define X(f) m.f = PK_I(w, PK_##f);
  PK_INTS(X)
#undef X
  m.lastK.A = (uint32_t)PK_I(w, PK_lastA); m.lastK.X = PK_L(w, PK_lastX); m.lastK.Y = PK_L(w, PK_lastY); m.lastN = (uint32_t)PK_I(w, PK_lastN);
}
// what the event asked for (read by the resumed run)
struct NestAsk { int replace, relane, koValid; uint32_t koA; uint64_t koX, koY; };
DEV NestAsk nestAsked(Dev& d) {
  const unsigned long long w = PK_WORDS_LOAD(d);
  NestAsk a; a.replace = PK_I(w, PK_replace); a.relane = PK_I(w, PK_relane); a.koValid = PK_I(w, PK_koValid); a.koA = (uint32_t)PK_I(w, PK_koA); a.koX = PK_L(w, PK_koX); a.koY = PK_L(w, PK_koY);
  return a;
}
DEV NestIO nestLoad(Dev& d) {
  const unsigned long long w = PK_WORDS_LOAD(d);
  NestIO st;
#define X(f) st.f = PK_I(w, PK_##f);
  NEST_INTS(X)
#undef X
  st.replace = st.relane = st.koValid = 0; st.koA = 0; st.koX = st.koY = 0;
  return st;
}
DEV void nestStore(Dev& d, const NestIO& st) {
  unsigned long long w = 0, mask = 0;
#define X(f) PK_PUT(w, PK_##f, (unsigned)st.f); mask |= PK_BIT(f);
  NEST_INTS(X)
#undef X
  PK_PUT(w, PK_replace, (unsigned)st.replace); PK_PUT(w, PK_relane, (unsigned)st.relane); PK_PUT(w, PK_koValid, (unsigned)st.koValid); PK_PUT(w, PK_koA, st.koA); PK_PUT(w, PK_koX, st.koX); PK_PUT(w, PK_koY, st.koY);
  mask |= PK_BIT(replace) | PK_BIT(relane) | PK_BIT(koValid) | PK_BIT(koA) | PK_BIT(koX) | PK_BIT(koY);
  PK_WORDS_STORE(d, w, mask);
}
DEV_NOINLINE void streamNestSettle(Dev& d, FastCtx fc, int t);
DEV NestIO streamNestSettleBody(Dev& d, FastCtx fc, NestIO st, int t);
DEV NestIO streamNestGangBody(Dev& d, FastCtx fc, StreamIn in, NestIO st, int t);
DEV_NOINLINE void streamNestGang(Dev& d, FastCtx fc, StreamIn in, int t);
DEV RunState streamMerge(Dev& d, FastCtx fc, int Q, int skip, int nest, RunState m);
DEV RunState streamStaged(Dev& d, FastCtx fc, int Q, int skip, int V, RunState m);   // the same for a run whose merged order exists already (round_merge.h)
DEV_NOINLINE int mgPrepare(Dev& d, FastCtx fc, int Q, int skip, int need);                    // round_merge.h
// ---- (round 5) a run does not end where a queue's stream ends at a gang.  A queue's stream is cut at its next gang member (B_QSSUM), and until round 4 the run ended
// the moment any queue used its stream up — on gang-heavy pools (BASELINE configs[3]) most single jobs therefore took the per-job iteration, whose queue side costs the
// control wave 8.5 k ticks against ~4 k for a merged entry, while the node engine idled 58 % (profiles/r05j_gangs_segments.txt).  Now, when the element behind a used-up
// QUEUED stream is a gang that fastPeekGang can assemble, the run (a) drains — every emitted entry placed, accounted and its ring slot read by the bind wave —, (b) settles
// that ONE queue exactly as the end of a run would (cursor, tokens, fastAdvance -> the gang is the head, its key enters the heap), and goes on merging; when the gang reaches
// the top it (c) drains again, closes the ring session, places the gang through a session of its own (fastGangRun: binds held until every member is placed), gives the queue
// its next stretch of single jobs as a stream (fastStreamPrepareOne) and opens a new session.  Anything else — a single job behind the stream, a gang the fast path does not
// take, a member without a node, skip mode — ends the run as before.  Exactness: every step is one the loop outside would have taken in the same order on the same state
// (a drained run is a finished run); soaks: tests/soak.py streams / rounds with gangs.
DEV_NOINLINE StreamOut fastStreamRun(Dev& d, FastCtx fc, int Q, StreamIn in) {
  const FastK k = fastKRef(d);
  FastS S; coldS(d, S);
  StreamOut out; memset(&out, 0, sizeof out); out.pend = -1; out.engSeq = in.engSeq;
  int doneQmid = 0, maxMid = 0, sessLive = 1, emittedPrev = 0;
#ifdef ASCHED_HOSTSIM
  static const bool nestOff = getenv("HS_NO_GANG_NEST") != nullptr;
#elif defined(ASCHED_NO_NEST)
  const bool nestOff = true;
#else
  const bool nestOff = false;
#endif
  const bool nest = !nestOff && !fc.replay;
  const int skip = UNI32(in.skip);
  PQState pq; memset(&pq, 0, sizeof pq);
  StreamLanes sl; memset(&sl, 0, sizeof sl);
  PackedKey lastK; lastK.A = ~0u; lastK.X = lastK.Y = ~0ull; uint32_t lastN = ~0u; int lastQ = -1;
  int engSeq = in.engSeq;
  // The merge loop lives in a function of its own (streamMerge), the run's state going in and out by value.  (The first version of the nested runs had the event code inline in
  // the loop and the headline lost 14-20 %, profiles/r05l_nested_gang_runs.txt — put down to register allocation then; in hindsight more likely the miscompiled key test
  // described at streamMerge, which makes skip mode rebuild its state around a stale key.)
  RunState m; memset(&m, 0, sizeof m);
  bool go = true;
  if (!UNI32(in.resume)) {
    pqBuild(pq, Q);
    int allowed = INT32_MAX;
    if (!in.globalRateInf) allowed = in.globalTokens >= 2147483000.0 ? INT32_MAX : (in.globalTokens < 1 ? 0 : (int)in.globalTokens);
    if (in.globalBurst < 1 || in.globalTokens < 1) allowed = 0;
    lastK.A = ~0u; lastK.X = lastK.Y = ~0ull; lastN = ~0u;
    int haveLast = 0;
    if (skip && UNI32(in.haveLast)) { lastK.A = UNI32(in.lastA); lastK.X = UNI64(in.lastX); lastK.Y = UNI64(in.lastY); lastN = UNI32(in.lastN); haveLast = 1; }
    memset(&sl, 0, sizeof sl);
    FOR_LANES(q, QCAPF) {   // (the head of a queue is element sPos of its stream = the list entry before the queue's cursor)
      const QHot& f = FL.hot[q];
      int kind = FL.sKind[q] ? 1 : 0;
      SL_SET(sl, start, q, f.sPos); SL_SET(sl, pos, q, f.sPos); SL_SET(sl, len, q, q < Q ? f.sLen : 0);
      SL_SET(sl, base, q, (kind ? f.itEi : f.itQi) - 1 - f.sPos);
      SL_SET(sl, kind, q, kind | (f.effValid ? 2 : 0));
      SL_SET(sl, ws, q, -2 * WIN);
      SL_SET(sl, budget, q, f.budget);
      SL_SET(sl, effA, q, FL.effA[q]); SL_SET(sl, effX, q, FL.effX[q]); SL_SET(sl, effY, q, FL.effY[q]);
      FL.tmpQ[q] = 0; FL.tmpA[q] = (uint32_t)f.sPos;   // (tmpA: a staged run's merge cursor per queue, streamStaged)
    }
    engSeq = in.engSeq;
    const int hc = UNI32(in.bulkV) > 0 && d.f.engineHc;
    if (hc) {   // the session's mailbox lives in the queues' key windows (FL.evWin, engine_hc.h): no queue's window survives it
      FOR_LANES(q, QCAPF) { FL.hot[q].ewCount = 0; FL.hot[q].ewStart = 0; }
      LANE0_PUBLISHED();
    }
    streamBegin(&engSeq, 0, hc);
    m.pq = pq; m.sl = sl; m.lastK = lastK; m.lastN = lastN; m.haveLast = haveLast; m.lastQ = -1;
    m.stageBase = -1; m.allowed = allowed; m.engSeq = engSeq; m.sessLive = 1; m.pend = -1;
  } else {
    // the event the run was parked at has run (streamNestSettle / streamNestGang, called by the main loop): the heap / lane updates it asks for, then on with the merge
    runUnpark(d, m);
    const NestAsk ask = nestAsked(d);
    const int t = m.evT, replace = ask.replace, relane = ask.relane;
    if (m.emitted == 0) m.issuedTo = 0;   // (a new ring session)
    if (replace) {
      KeyOut nk; nk.valid = ask.koValid; nk.A = ask.koA; nk.X = ask.koX; nk.Y = ask.koY;
      pqPopPush(m.pq, nk, t);
    }
    if (relane) {
      const QHot& f = FL.hot[t];   // queue t's lane state of the run from its (just materialised) record
      int kd = UNI32((int)FL.sKind[t]) ? 1 : 0, sp = UNI32(f.sPos), sn = UNI32(f.sLen);
      SL_SET(m.sl, start, t, sp); SL_SET(m.sl, pos, t, sp); SL_SET(m.sl, len, t, sn);
      SL_SET(m.sl, base, t, (kd ? UNI32(f.itEi) : UNI32(f.itQi)) - 1 - sp);
      SL_SET(m.sl, kind, t, kd | (UNI32(f.effValid) ? 2 : 0));
      SL_SET(m.sl, ws, t, -2 * WIN);
      SL_SET(m.sl, budget, t, UNID(f.budget));
      if (FLANE == 0) FL.tmpQ[t] = 0;
      LANE0_PUBLISHED();
    }
    go = m.go != 0;
  }
  if (go) {
    if (!UNI32(in.resume) && UNI32(in.bulkV) > 0) m = streamStaged(d, fc, Q, skip, UNI32(in.bulkV), m);
    else m = streamMerge(d, fc, Q, skip, nest ? 1 : 0, m);
    if (m.ev != 0) {
      // an event: everything emitted so far is staged here (the gathered records in flight are lane-private), the run parks and the main loop runs the event
      if (m.stageBase >= 0) { streamStageCommit(d, k, m.stageBase, m.stageCnt, m.stageV); m.stageBase = -1; }
      if (m.emitted > m.issuedTo) { unsigned long long v = streamStageIssue(k, m.issuedTo, m.emitted - m.issuedTo); streamStageCommit(d, k, m.issuedTo, m.emitted - m.issuedTo, v); m.issuedTo = m.emitted; }   // (never twice: the serial build serves an entry when it is staged)
      runPark(d, m);
      out.event = m.ev; out.evT = m.evT;
      return out;
    }
  }
  pq = m.pq; sl = m.sl; lastK = m.lastK; lastN = m.lastN; lastQ = m.lastQ;
  int emitted = m.emitted, acc = m.acc, stageBase = m.stageBase, stageCnt = m.stageCnt, issuedTo = m.issuedTo, fail = m.fail;
  unsigned long long stageV = m.stageV;
  engSeq = m.engSeq; sessLive = m.sessLive; emittedPrev = m.emittedPrev; doneQmid = m.doneQmid; maxMid = m.maxMid;
  out.gangJobs = m.gangJobs; out.gangs = m.gangs;
  out.pend = m.pend; out.dropped = m.dropped;
  S.statRefills += m.refills; S.numEvictedJobs += m.evicted;
  // drain: what is still in flight, then the tail group
  if (sessLive) {
    if (!fail) {
      if (stageBase >= 0) streamStageCommit(d, k, stageBase, stageCnt, stageV);
      if (emitted > issuedTo) { stageV = streamStageIssue(k, issuedTo, emitted - issuedTo); streamStageCommit(d, k, issuedTo, emitted - issuedTo, stageV); issuedTo = emitted; }
    }
    streamEnd(engSeq);
    { int a = streamAcked(&fail); if (a > acc) { streamAccount(d, k, acc, a); acc = a; } }
  }
  if (fail == 2) out.dropped = 1;
  out.failed = fail ? 1 : 0; out.lastQ = lastQ;
  // ---- the queues' iterator state as of the acc entries done: each queue's head becomes the first of its elements that was not done
  FOR_LANES(q, QCAPF) { FL.hot[q].winKind = -1; FL.hot[q].winCount = 0; }   // the windows served as the ring
  int doneQ = 0, doneEv = 0;
  for (int q = 0; q < Q; q++) {
    QHot f = FL.hot[q];
    uniQHot(f);
    if (f.sLen == 0) continue;
    int start = SL_GET(sl, start, q), cq = UNI32(FL.tmpQ[q]), moved = SL_GET(sl, pos, q) - start, kind = SL_GET(sl, kind, q) & 1;
    int pos = start + cq;                          // the new head
    bool more = pos < f.sLen;
#ifdef ASCHED_HOSTSIM
    if (getenv("HS_NO_STREAM_KEEP")) more = false;
#endif
    if (FLANE == 0) { FL.hot[q].sPos = more ? pos : 0; FL.hot[q].sLen = more ? f.sLen : 0; FL.hot[q].ewCount = 0; FL.hot[q].ewStart = 0; }
    LANE0_PUBLISHED();
    f.sPos = more ? pos : 0; f.sLen = more ? f.sLen : 0; f.ewCount = 0; f.ewStart = 0; f.winKind = -1; f.winCount = 0;
    if (moved == 0) continue;                     // never reached the top of the heap: nothing of the queue changed
    if (kind) doneEv += cq; else { doneQ += cq; if (cq > out.maxConsumed) out.maxConsumed = cq; }
    if (f.effValid) {                             // the running maximum restarts from the queue's last folded evicted entry (fastEnterSkip); any clamp between it and
      EvKey e; memcpy(&e, (const char*)k.evKey + (size_t)(f.evEnd - 1) * sizeof(EvKey), sizeof(EvKey));   // the true running maximum gives the same order
      PackedKey last = packKey3(fc.preferLarge, UNI32(e.pcPrio), UNID(e.proposed), UNID(e.current), UNID(e.size), f.budget);
      FL.effA[q] = last.A; FL.effX[q] = last.X; FL.effY[q] = last.Y;
    }
    if (cq == 0) {                                // emitted, not done: the head it had is the head again, under the key it was peeked with
      KeyOut ko = packItemKeys(fc.preferLarge, q, fc.compareSchedPrio ? f.schedPrio : f.pcPrio, f.proposed, f.current, f.size, f.budget);
      if (f.effValid) {
        PackedKey own, eff; own.A = ko.A; own.X = ko.X; own.Y = ko.Y; eff.A = UNI32(FL.effA[q]); eff.X = UNI64(FL.effX[q]); eff.Y = UNI64(FL.effY[q]);
        if (packedLess(own, 0, eff, 0)) { FL.kA[q] = eff.A; FL.kX[q] = eff.X; FL.kY[q] = eff.Y; }
        else { FL.effA[q] = own.A; FL.effX[q] = own.X; FL.effY[q] = own.Y; }
      }
      FL.inHeap[q] = 1;
      continue;
    }
    if (kind) {   // cq evicted jobs came back to their nodes: commits deferred exactly as the cheap evicted head's (fastIter: evDone = headPos + 1)
      f.evDone = f.itEi - 1 + cq; f.itEi = f.itEi - 1 + cq;
    } else {      // the next peek yields element pos (the old head was already peeked and counted)
      f.itQi = f.itQi - 1 + cq; f.itJobsSeen = f.itJobsSeen - 1 + cq;
      if (!f.rateInf && 1 <= f.burst) f.tokens -= (double)cq;
    }
    KeyOut ko;
    if (!fastAdvance(d, k, S, fc, q, f, &ko)) out.pend = q;
  }
  doneQ += doneQmid; if (maxMid > out.maxConsumed) out.maxConsumed = maxMid;
  out.executed = doneQ; out.executedEv = doneEv; out.engSeq = engSeq; out.emitted = emittedPrev + emitted; out.refills = S.statRefills; out.evicted = S.numEvictedJobs;
  out.lastA = lastK.A; out.lastX = lastK.X; out.lastY = lastK.Y; out.lastN = lastN;
  return out;
}

// the merge loop of a run (fastStreamRun): pop the head of the lane heap, emit (job, queue) into the ring, take the queue's next precomputed costs, re-insert
// NOTE (round 5, profiles/r05y_lastkey_miscompile.txt): the two "served in non-decreasing key order" tests below branch on a condition made wave-uniform by hand (UNI32).  Written
// as a plain `if (haveLast && packedLess(...)) break;` — haveLast arrives in a VGPR, so the branch is compiled as a divergent one — hipcc 7.2 emitted, in some builds of this
// function, a masked update that refreshes lastK.X / lastK.Y after the test but leaves lastK.A, lastN and haveLast at their old values on the "not less" path (the (A, N) pair
// is reset to the old pair after the compare and never set again): fastExitSkip then rebuilt the state around a key with a stale priority field and the round diverged (test
// test_gang_behind_folded_evicted_streams_gpu, seeds 100345 / 102465).  Whether a build had it depended on unrelated code around the loop; the CPU build never had it.
DEV RunState streamMerge(Dev& d, FastCtx fc, int Q, int skip, int nest, RunState m) {
  const FastK k = fastKRef(d);
  FastS S; coldS(d, S);
  PQState pq = m.pq; StreamLanes sl = m.sl;
  PackedKey lastK = m.lastK; uint32_t lastN = m.lastN; int haveLast = m.haveLast, lastQ = m.lastQ;
  int emitted = m.emitted, emittedQ = m.emittedQ, acc = m.acc, stageBase = m.stageBase, stageCnt = m.stageCnt, issuedTo = m.issuedTo, fail = m.fail;
  const int allowed = m.allowed;
  unsigned long long stageV = m.stageV;
  int ev = 0, evT = -1, evSLen = 0;
  SEG_BEGIN();
  for (;;) {
    int a = streamAcked(&fail);
    if (a > acc) { streamAccount(d, k, acc, a); acc = a; }
    if (fail) break;
    if (emitted - acc >= RING_N - 8 || emitted - streamBound() >= RING_N - 8) { STREAM_IDLE(); SEG(15); continue; }   // the ring is full: the engine is the pace
    SEG(11);
    int t = pqHead(pq, Q);
    if (t < 0) break;
    int sPos = SL_GET(sl, pos, t), sLen = SL_GET(sl, len, t);
    if (sPos >= sLen) {                           // the head of the heap is not a stream element
      if (!(nest && UNI32(FL.hot[t].gctx) < -1 && UNI32(FL.hot[t].sLen) == 0)) break;
      // an assembled gang (fastPeekGang): through a ring session of its own, inside the run (streamNest)
      if (skip) {   // skip mode: entries are served in non-decreasing key order only (as for a stream element below; fastRun makes the same test in front of a gang)
        PackedKey curK; uint32_t curN;
        pqHeadKey(pq, t, &curK, &curN);
        if (UNI32((int)(haveLast && packedLess(curK, curN, lastK, lastN)))) break;   // (a scalar branch: see the note above the loop)
        lastK = curK; lastN = curN; haveLast = 1;
      }
      lastQ = t;
      ev = 2; evT = t;
      break;
    }
    int kind = SL_GET(sl, kind, t);
    if (!(kind & 1) && emittedQ >= allowed) break;   // no global token left for another new job
    int base = SL_GET(sl, base, t);
    EvKey e = streamKey(k, sl, t, sPos, sLen, kind, base);
    SEG(12);
    if (skip) {                                   // skip mode (fastRun): the folded evicted streams are merged around keys served in non-decreasing order only
      PackedKey curK; uint32_t curN;
      pqHeadKey(pq, t, &curK, &curN);
      if (UNI32((int)(haveLast && packedLess(curK, curN, lastK, lastN)))) break;   // (a scalar branch: see the note above the loop)
      lastK = curK; lastN = curN; haveLast = 1;   // the key this entry is served under (fastExitSkip rebuilds the state around the last one)
    }
    lastQ = t;
    if (FLANE == 0) { RJOB(emitted) = e.job; RQ(emitted) = t | ((kind & 1) ? RQ_EV : 0); }
    LANE0_PUBLISHED();
    emitted++; if (!(kind & 1)) emittedQ++;
    if (emitted - issuedTo == 4) {                // records: gather the last four entries; the four before them have arrived by now
      if (stageBase >= 0) streamStageCommit(d, k, stageBase, stageCnt, stageV);
      stageBase = issuedTo; stageCnt = 4; issuedTo += 4;
      stageV = streamStageIssue(k, stageBase, 4);
    }
    sPos++;
    SL_SET(sl, pos, t, sPos);
    SEG(13);
    KeyOut ko; ko.valid = 0; ko.A = 0; ko.X = ko.Y = 0;
    if (sPos < sLen) {
      EvKey n = streamKey(k, sl, t, sPos, sLen, kind, base);
      PackedKey own = packKey3(fc.preferLarge, n.pcPrio, n.proposed, n.current, n.size, SL_GETD(sl, budget, t));
      if (kind & 2) {   // skip mode: as fastAdvance — a head is not served before what precedes it in its queue
        PackedKey eff; eff.A = (uint32_t)SL_GET(sl, effA, t); eff.X = SL_GET64(sl, effX, t); eff.Y = SL_GET64(sl, effY, t);
        if (packedLess(own, 0, eff, 0)) own = eff;
        else { SL_SET(sl, effA, t, own.A); SL_SET(sl, effX, t, own.X); SL_SET(sl, effY, t, own.Y); }
      }
      ko.valid = 1; ko.A = own.A; ko.X = own.X; ko.Y = own.Y;
      SEG(14);
      pqPopPush(pq, ko, t);
      SEG(10);
    } else {
      bool listEnds = !(kind & 1) && UNI32(d.qsLen[2 * t + 1]);
      if (listEnds) { pqPopPush(pq, ko, t); continue; }     // the queue's list ends where its stream ends: it leaves the heap
      // the queue goes on beyond its stream (its queued jobs after the evicted ones / more of its list): the next key is not known here ...
      bool gangNext = false;
      if (nest && !(kind & 1) && !UNI32(FL.hot[t].effValid)) {   // ... unless the element behind a queued stream is a gang member (where B_QSSUM cuts): settle the queue, peek the gang (streamNest)
        int nx = base + sLen;                                     // list position of the element behind the stream
        if (nx < UNI32(FL.hot[t].qEnd)) gangNext = UNI32(d.jGang[UNI32(k.queuedJobs[nx])]) >= 0;
      }
      if (!gangNext) { pqPopPush(pq, ko, t); break; }
      ev = 1; evT = t; evSLen = sLen;
      break;
    }
  }
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_STREAM_TRACE")) { int t = pqHead(pq, Q); fprintf(stderr, "merge loop returns %d: emitted %d (new %d) acc %d fail %d allowed %d top %d", ev, emitted, emittedQ, acc, fail, allowed, t); if (t >= 0) fprintf(stderr, " sPos %d sLen %d skind %d kind %d gctx %d stage %d inHeap %d", SL_GET(sl, pos, t), SL_GET(sl, len, t), SL_GET(sl, kind, t), FL.hot[t].headKind, FL.hot[t].gctx, FL.hot[t].itStage, FL.inHeap[t]); fprintf(stderr, "\n"); }
#endif
  m.pq = pq; m.sl = sl; m.lastK = lastK; m.lastN = lastN; m.haveLast = haveLast; m.lastQ = lastQ;
  m.emitted = emitted; m.emittedQ = emittedQ; m.acc = acc; m.stageBase = stageBase; m.stageCnt = stageCnt; m.issuedTo = issuedTo; m.fail = fail; m.stageV = stageV;
  m.ev = ev; m.evT = evT; m.evSLen = evSLen;
  return m;
}
// A run whose merged order was computed ahead (round_merge.h mgPrepare: d.mg->merged[0 .. V)): this wave only STAGES — four entries per step: ring space, the (job, queue)
// words, the job records' gather — and accounts what the engine has placed; the state it hands back is what streamMerge would have left after emitting the same entries.
DEV RunState streamStaged(Dev& d, FastCtx fc, int Q, int skip, int V, RunState m) {
  (void)fc; (void)Q;
  const FastK k = fastKRef(d);
  StreamLanes sl = m.sl;
  int emitted = 0, emittedQ = 0, acc = m.acc, stageBase = -1, stageCnt = 0, issuedTo = 0, fail = 0, lastQ = -1, lastCi = -1;
  const int allowed = m.allowed;
  unsigned long long stageV = 0;
  const MgEnt* mer = d.mg->merged;
  bool over = false;
  for (int base = 0; base < V && !fail && !over; base += 64) {
    const int nb = V - base < 64 ? V - base : 64;
    FOR_LANES(x, 64) if (x < nb) { const MgEnt en = mer[base + x]; FL.tmpX[x] = ((uint64_t)(uint32_t)en.qk << 32) | (uint32_t)en.job; FL.tmpY[x] = ((uint64_t)(uint32_t)en.ci << 32) | (uint32_t)en.e; }
    LANE0_PUBLISHED();
    for (int g = 0; g < nb && !over; g += 4) {
      const int n4 = nb - g < 4 ? nb - g : 4;
      for (;;) {   // the ring is the engine's pace
        int a = streamAcked(&fail);
        if (a > acc) { streamAccount(d, k, acc, a); acc = a; }
        if (fail || (emitted + 4 - a <= RING_N - 8 && emitted + 4 - streamBound() <= RING_N - 8)) break;
        STREAM_IDLE();
      }
      if (fail) break;
      int put = 0;
      for (int x = 0; x < n4; x++) {
        const uint64_t v = UNI64(FL.tmpX[g + x]), w = UNI64(FL.tmpY[g + x]);
        const int job = (int)(uint32_t)v, qk = (int)(uint32_t)(v >> 32), e = (int)(uint32_t)w;
        const int ev = (qk >> 30) & 1, q = qk & 0xffffff;
        if (!ev && emittedQ >= allowed) { over = true; break; }   // no global token left for another new job (constraints.go:129-141)
        if (FLANE == 0) { RJOB(emitted + put) = job; RQ(emitted + put) = q | (ev ? RQ_EV : 0); FL.tmpA[q] = (uint32_t)(e + 1); }
        LANE0_PUBLISHED();
        put++; if (!ev) emittedQ++;
        lastQ = q; lastCi = (int)(uint32_t)(w >> 32);
      }
      if (put > 0) {
        if (stageBase >= 0) streamStageCommit(d, k, stageBase, stageCnt, stageV);
        stageBase = emitted; stageCnt = put; emitted += put; issuedTo = emitted;
        stageV = streamStageIssue(k, stageBase, put);
      }
    }
  }
  // every queue's merge cursor: behind its last emitted element
  FOR_LANES(q, QCAPF) SL_SET(sl, pos, q, (int)FL.tmpA[q]);
  m.sl = sl;
  if (lastCi >= 0) {   // the key the last entry was served under (its running maximum: in skip mode every emitted entry's own key IS that — the passes stop in front of a decrease)
    const WideKey lk = d.mg->key[lastCi];
    m.lastK.A = (uint32_t)UNI64(lk.a); m.lastK.X = UNI64(lk.x); m.lastK.Y = UNI64(lk.y); m.lastN = (uint32_t)UNI32(FL.nameRank[lastQ]); m.haveLast = 1;
  }
  (void)skip;
  m.lastQ = lastQ;
  m.emitted = emitted; m.emittedQ = emittedQ; m.acc = acc; m.stageBase = stageBase; m.stageCnt = stageCnt; m.issuedTo = issuedTo; m.fail = fail; m.stageV = stageV;
  m.ev = 0; m.evT = -1; m.evSLen = 0;
  return m;
}
// the two rare events of fastStreamRun (see there), on wave-uniform state; the caller has staged everything emitted.  st.go = 1: the run goes on; 0: it ends (st.pend /
// st.dropped / st.fail say why).  Two functions, each as small as it can be: what they (and their callees) clobber is what the run's caller must keep clear of.
DEV bool streamDrain(Dev& d, KREF k, NestIO& st) {   // everything emitted placed, accounted, and its ring slot read by the bind wave (the job-record windows ARE the ring: fastAdvance refills them)
  for (;;) {
    int a = streamAcked(&st.fail);
    if (a > st.acc) { streamAccount(d, k, st.acc, a); st.acc = a; }
    if (st.fail) return false;
    if (st.acc >= st.emitted && streamBound() >= st.emitted) return true;
    STREAM_IDLE();
  }
}
// queue t's queued stream is used up and the element behind it is a gang member
DEV NestIO streamNestSettleBody(Dev& d, FastCtx fc, NestIO st, int t) {
  const FastK k = fastKRef(d);
  FastS S; coldS(d, S);
  st.go = 0; st.replace = 1; st.relane = 0; st.koValid = 0; st.koA = 0; st.koX = st.koY = 0;
  if (!streamDrain(d, k, st)) return st;   // (an entry found no node: the queue leaves the heap, as at the end of its stream before round 5)
  // every element of queue t's stream is done: its iterator state as the end of a run leaves it, then the next head through the ordinary fastAdvance
  QHot f = FL.hot[t];
  uniQHot(f);
  int cq = UNI32(FL.tmpQ[t]);
  if (FLANE == 0) { FL.hot[t].sPos = 0; FL.hot[t].sLen = 0; FL.hot[t].ewCount = 0; FL.hot[t].ewStart = 0; FL.hot[t].winKind = -1; FL.hot[t].winCount = 0; }
  LANE0_PUBLISHED();
  f.sPos = 0; f.sLen = 0; f.ewCount = 0; f.ewStart = 0; f.winKind = -1; f.winCount = 0;
  st.doneQmid += cq; if (cq > st.maxMid) st.maxMid = cq;
  f.itQi = f.itQi - 1 + cq; f.itJobsSeen = f.itJobsSeen - 1 + cq;
  if (!f.rateInf && 1 <= f.burst) f.tokens -= (double)cq;
  KeyOut nk;
  bool more = fastAdvance(d, k, S, fc, t, f, &nk);
  if (FLANE == 0) { FL.hot[t].winKind = -1; FL.hot[t].winCount = 0; }   // (its window lies in the ring)
  LANE0_PUBLISHED();
  st.koValid = nk.valid; st.koA = nk.A; st.koX = nk.X; st.koY = nk.Y; st.relane = 1;
  st.refills += S.statRefills; st.evicted += S.numEvictedJobs;
  if (!more) { st.pend = t; return st; }
  st.go = 1;
  return st;
}
// the head of the heap, queue t, is an assembled gang
DEV_NOINLINE void streamNestSettle(Dev& d, FastCtx fc, int t) { NestIO st = nestLoad(d); st = streamNestSettleBody(d, fc, st, t); nestStore(d, st); }
DEV NestIO streamNestGangBody(Dev& d, FastCtx fc, StreamIn in, NestIO st, int t) {
  const FastK k = fastKRef(d);
  st.go = 0; st.replace = 0; st.relane = 0; st.koValid = 0; st.koA = 0; st.koX = st.koY = 0;
  if (!streamDrain(d, k, st)) return st;
  streamEnd(st.engSeq); st.sessLive = 0;
  FOR_LANES(q, QCAPF) FL.tmpN[q] = (uint32_t)FL.tmpQ[q];   // (fastGangRun and fastStreamPrepareOne use tmpQ as scratch: the run's per-queue counts wait in the heap's scatter space)
  LANE0_PUBLISHED();
  StreamIn gi = in;
  gi.globalTokens = in.globalTokens - (double)st.emittedQ - (double)st.gangJobs; gi.engSeq = st.engSeq; gi.skip = 0; gi.haveLast = 0;
  GangOut go = fastGangRun(d, fc, gi, t);
  st.engSeq = go.engSeq;
  if (go.dropped) st.dropped = 1;
  bool goOn = false;
  if (go.handled) {
    st.gangJobs += go.cnt; st.gangs++;
    if (st.allowed != INT32_MAX) st.allowed -= go.cnt;
    st.refills += go.refills; st.evicted += go.evicted;
    st.replace = 1; st.relane = 1; st.koValid = go.koValid; st.koA = go.koA; st.koX = go.koX; st.koY = go.koY;
    if (go.pend >= 0 || st.dropped) st.pend = go.pend;
    else {
      if (go.koValid && UNI32(FL.hot[t].gctx) >= 0) {   // the single jobs behind the gang as the queue's next stream, prepared right here
        int want = st.allowed == INT32_MAX ? INT32_MAX : (st.allowed - st.emittedQ > 0 ? st.allowed - st.emittedQ : 0);
        (void)fastStreamPrepareOne(d, fc, t, want, QS_CMAX);
      }
      goOn = true;
    }
  }   // (not handled: nothing of the gang was done — the loop outside meets it as the head)
  FOR_LANES(q, QCAPF) FL.tmpQ[q] = (int32_t)FL.tmpN[q];
  LANE0_PUBLISHED();
  if (!goOn) return st;
  FOR_LANES(q, QCAPF) { FL.hot[q].winKind = -1; FL.hot[q].winCount = 0; }   // the windows serve as the ring again
  streamBegin(&st.engSeq); st.sessLive = 1;
  st.emittedPrev += st.emitted; st.emitted = 0; st.acc = 0;
  st.go = 1;
  return st;
}

DEV_NOINLINE void streamNestGang(Dev& d, FastCtx fc, StreamIn in, int t) { NestIO st = nestLoad(d); st = streamNestGangBody(d, fc, in, st, t); nestStore(d, st); }

// ---- a gang through the ring.  GangScheduler.Schedule for the common gang (gang_scheduler.go:46-148, 229-262): every member a queued job, no
// uniformity label, every member fits without preemption.  The members go through the ring like a stream run's entries — the node engine places them one
// after the other (first fit at priority -2, L0 upkeep) — but the bind wave HOLDS the HBM side until the engine has placed them all: ScheduleManyWithTxn
// is all or nothing.  All placed: the binds and result fields are issued, the members are accounted like cnt new jobs and ONE scheduled gang, the rate
// limiters give up cnt tokens (ReserveN), the queue's next head is produced.  A member without a node: nothing has reached HBM but "this base entry is
// stale" flags; the nodes the placed members went to are re-read into the level-0 structure (fastTouch) and the generic code runs the gang from the state
// it would have found (it decides about preemption, the failure reason, the unfeasible-key registration).
#ifndef ASCHED_STREAM_BACKOFF_MAX
#define ASCHED_STREAM_BACKOFF_MAX (1 << 18)   // fast iterations between two attempts to start a stream run after short runs (doubling from _MIN): A/B-measured, profiles/r03r_*
#endif
#ifndef ASCHED_STREAM_BACKOFF_MIN
#define ASCHED_STREAM_BACKOFF_MIN 8
#endif
DEV_NOINLINE GangOut fastGangRun(Dev& d, FastCtx fc, StreamIn in, int t) {
  const FastK k = fastKRef(d);
  FastS S; coldS(d, S);
  GangOut out; memset(&out, 0, sizeof out); out.pend = -1; out.engSeq = in.engSeq;
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_NO_GANG_RING")) return out;
#endif
  QHot f = FL.hot[t];
  uniQHot(f);
  int ref = f.gctx;
  if (ref >= -1 || k.hasPcLimit || k.anyRoundLimit || k.anyDisallowed || k.disableHome || !fc.stream) return out;
  int g = -ref - 2;
  int cnt = UNI32(d.gangSeen[g]), off = UNI32(d.gangOff[g]);
  if (cnt < 2 || cnt > RING_N - 16) return out;
  SEG_BEGIN();   // (profiling builds: statSeg[26..31] = member checks, staging, waiting for the engine's verdict, release of the held binds, accounting, the queue's next head)
  // CheckJobConstraints for the gang (constraints.go:121-157): anything that fails is the generic code's to report
  if (f.cordoned || in.globalTokens < (double)cnt || in.globalBurst < cnt || f.tokens < (double)cnt || f.burst < cnt) return out;
  if (UNI32((int)d.gangAllEvicted[g])) return out;
  // members: queued jobs untouched in this round (fastGangMember's conditions), no uniformity label
  int bad = 0;
  FOR_LANES(m, cnt) {
    int j = d.gangArr[off + m];
    if (d.jcEvicted[j] || d.jcPreempted[j] || d.jcAssigned[j] >= 0 || d.jcUniValue[j] >= 0 || d.schedAtPrio[j] != NO_PRIORITY || d.jobNode[j] >= 0 || d.jGangUni[j] != -1) bad = 1;
    RJOB(m) = j; RQ(m) = t;
  }
  FOR_LANES(x, 64) FL.tmpQ[x] = bad; 
  { int any = 0; for (int x = 0; x < 64; x++) any |= UNI32(FL.tmpQ[x]); if (any || RS.awayRowPlus1) return out; }
  FOR_LANES(q, QCAPF) FL.tmpQ[q] = 0;
  // the ring overwrites the prefetch windows of the first queues: they refill on demand
  FOR_LANES(q, QCAPF) if (q * WIN < cnt + 4) { FL.hot[q].winKind = -1; FL.hot[q].winCount = 0; }
  if (f.sLen) { if (FLANE == 0) { FL.hot[t].sLen = 0; FL.hot[t].sPos = 0; } LANE0_PUBLISHED(); f.sLen = 0; f.sPos = 0; }
  int engSeq = in.engSeq;
  SEG(26);
  streamBegin(&engSeq, 1);
  for (int b = 0; b < cnt; b += 4) {
    int n4 = cnt - b < 4 ? cnt - b : 4;
    unsigned long long v = streamStageIssue(k, b, n4);
    streamStageCommit(d, k, b, n4, v);
  }
  streamEnd(engSeq);
  SEG(27);
  int fail = 0;
  int placed = streamAcked(&fail);
  SEG(28);
  out.engSeq = engSeq;
  if (fail == 2) out.dropped = 1;
  if (placed < cnt || fail) {
    streamRelease(d, k, 0);
    for (int i = 0; i < placed; i++) fastTouch(d, UNI32(RREC(i).node0));   // authoritative planes untouched: the node's level-0 entry as it was
    return out;
  }
  streamRelease(d, k, 1);
  SEG(29);
  streamAccount(d, k, 0, cnt);
  SEG(30);
  if (!f.rateInf && cnt <= f.burst) f.tokens -= (double)cnt;   // rate.Limiter.ReserveN(cnt) (gang_scheduler.go:118-123)
  KeyOut ko;
  if (!fastAdvance(d, k, S, fc, t, f, &ko)) out.pend = t;
  SEG(31);
  out.koValid = ko.valid; out.koA = ko.A; out.koX = ko.X; out.koY = ko.Y;
  out.handled = 1; out.cnt = cnt; out.refills = S.statRefills; out.evicted = S.numEvictedJobs;
  return out;
}

// ---- a queued job that needs preemption, without leaving the fast loop.  The queue side of the iteration is fastIter's (constraints already
// checked, accounting in LDS, next head through fastAdvance); the node side is the generic one, called as is: tryScheduleGang for a one-job gang
// (gang_scheduler.go:229-262) = SelectNodeForJobWithTxn's cascade (nodedb.go:538-838: gate, fair-share preemption, urgency preemption, away node
// types), the victims' removal, BindJobToNode, the evicted-table upkeep, inside a transaction; it keeps the level-0 structure in line itself
// (fastTouch).  What the generic loop would add around it — AddGangSchedulingContext / EvictGang / re-add on failure (scheduling.go:391-449), the
// unfeasible-key registration (gang_scheduler.go:63-98), the rate limiters (:118-123), the PQS bookkeeping (pqs.go:160-166) — is applied to the fast
// state.  Saves the hand-over of every queue's state out of and back into LDS, the generic Less over all queues and the generic peek per job.
// The caller has stopped the node engine (the cascade's wide passes need every wave of the workgroup; it reads planes the binds must have reached)
// and written its scalars back to RS.  Returns 0 = not handled (nothing touched), 1 = done, 2 = done + the generic code must produce the queue's
// next head, 3 = done + leave the fast loop (the fair-share preemption rate limit ran dry: queue_scheduler.go:125-142 is the generic loop's).
DEV __attribute__((always_inline)) int fastPreemptIter(Dev& d, Ctl& c, FastCtx fc, int t, KeyOut* koOut) {   // (inlined: as a call it saves and restores 109 registers per job)
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_NO_PREEMPT_FAST")) return 0;
#endif
  const FastK k = fastKRef(d);
  if (RS.replayPending || !RS.fastActive || c.txn.active || RS.error) return 0;   // (the deferred replay runs fast loops of its own: the generic iteration triggers it once)
  fastFlushEvictedPending(d);   // the cascade reads planes above priority -2 and the evicted table: returned evicted jobs whose commits were deferred go in first
  QHot f = FL.hot[t];
  uniQHot(f);
  int job = f.gctx;
  if (job < 0 || !f.headFast || f.headKind != 1 || f.effValid) return 0;
  JobTail r = FL.headTail[t];
  uniJobTail(r);
  int pcx = r.pc;
  FastS S; coldS(d, S);
  XSEG_BEGIN();
  c.l1Dirty = 1;   // the fast iterations since the last fence bound through no-return atomics
  fastFence(c);
  XSEG(35);
  int reason = 0;
#ifdef ASCHED_TRYGANG_COLD
  bool ok = tryGangCold(d, c, job, &reason);
#else
  bool ok = tryGang(d, c, job, &reason);
#endif
  XSEG(36);
  if (RS.error) return 1;
  if (FLANE == 0) { RS.loopIterations++; RS.statFastIters++; RS.statHybrid++; }
  LANE0_PUBLISHED();
  uint8_t fl = k.jobFlags[job];
  if (ok) {
    accountVectors(d, k, t, pcx, false, false);
    if (FLANE == 0) {
      RS.numScheduledJobs++; RS.numScheduledGangs++;
      if (!RS.globalRateInf && 1 <= RS.globalBurst) RS.globalTokens -= 1.0;
      k.jobFlags[job] = (uint8_t)((fl & ~F_UNSUCCESSFUL) | F_SUCCESSFUL); k.inScheduled[job] = 1; k.inSchedAndEvicted[job] = 0;
    }
    LANE0_PUBLISHED();
    if (!f.rateInf && 1 <= f.burst) f.tokens -= 1.0;
  } else {
    failJob(d, job, reason);
    if (FLANE == 0) k.jobFlags[job] = (uint8_t)((fl & ~F_SUCCESSFUL) | F_UNSUCCESSFUL);
    if (!c.skipKeyCheck && isPropertyOfGang(reason) && keyValid(d, job)) {
      int sh = d.jShape[job];
      if (!d.unfeasible[sh]) {
        d.unfeasible[sh] = 1; d.unfeasibleReason[sh] = reason; RS.numUnfeasible++;
        FOR_LANES(q, QCAPF) { FL.hot[q].sLen = 0; FL.hot[q].sPos = 0; }   // the queues' precomputed streams were laid out before this key was known to be unfeasible (queue_scheduler.go:398-413 skips its jobs at peek time)
        f.sLen = 0; f.sPos = 0;
      }
    }
  }
  S.numUnfeasible = UNI32(RS.numUnfeasible); S.numPreemptedMarks = UNI32(RS.numPreemptedMarks); S.lvl0NonNeg = UNI32(RS.lvl0NonNeg); S.fastActive = UNI32(RS.fastActive);
  // A queue's evicted jobs as a stream (FL.sKind 1) rest on "evicted jobs always return": once this cascade has preempted — marks, or a priority -2 column overdrawn —
  // the streams that were prepared before it must not outlive it (a new preparation asks evOk itself).  Never met while the replay was deferred: the first preempting
  // job went through the generic loop, which drops every stream; with the evicted table built up front (replay_rank.h) the first one comes through here.
  if (S.numPreemptedMarks != 0 || !S.lvl0NonNeg) { FOR_LANES(q, QCAPF) if (FL.sKind[q] && FL.hot[q].sLen) { FL.hot[q].sLen = 0; FL.hot[q].sPos = 0; } LANE0_PUBLISHED(); }
  XSEG(37);
  KeyOut ko;
  bool more = fastAdvance(d, k, S, fc, t, f, &ko);
  *koOut = ko;
  XSEG(38);
  if (FLANE == 0) { RS.numEvictedJobs += S.numEvictedJobs; RS.statRefills += S.statRefills; }
  LANE0_PUBLISHED();
  if (!more) return 2;
  if (RS.hasFpLimiter && RS.fpTokens < 1 && !c.fpLimitHit) return 3;
  return 1;
}

// Is the run about to start long enough for the bulk merge (round_merge.h)?  The sum of the streams' remaining lengths, lane-parallel — asked BEFORE the node engine
// is stopped for the passes (mgPrepare asks again, exactly).
DEV bool mgWorth(Dev& d, int Q) {
#ifdef ASCHED_HOSTSIM
  static const int minEntries = getenv("HS_MG_MIN") ? atoi(getenv("HS_MG_MIN")) : MG_MIN_ENTRIES_DEFAULT;
  if (getenv("HS_NO_MERGE")) return false;
#else
  const int minEntries = d.f.mgMin;
#endif
  const FastK k = fastKRef(d);
  // one lane per queue, the verdict through ballots (a serial walk over the queues' LDS records costs ~13 k ticks per call, and gang-heavy rounds ask twenty thousand times:
  // BASELINE configs[3] 692 -> 843 ms, profiles/r06r).  Worth it: one long stream, or many of some length — mgPrepare counts exactly.
  int big = 0, some = 0, bad = 0;
  FOR_LANES(q, QCAPF) {
    int t = 0;
    if (q < Q && FL.inHeap[q] && FL.hot[q].sLen > FL.hot[q].sPos) t = FL.hot[q].sLen - FL.hot[q].sPos;
#if defined(ASCHED_HOSTSIM) || !defined(__HIP_DEVICE_COMPILE__)
    if (t >= minEntries) big = 1;
    if (t >= minEntries / 16) some++;
#else
    big = t >= minEntries; some = t >= minEntries / 16;
#endif
  }
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  big = __ballot(big != 0) != 0; some = __popcll(__ballot(some != 0));
#endif
  if (!(big || some >= 16)) return false;
  // (mgPrepare's rules, asked here BEFORE the node engine is stopped for nothing: a run that would nest a gang stays with the control wave's merge; these look at HBM,
  // so only once the lengths say the run is worth it)
  FOR_LANES(q, QCAPF) {
    int b2 = 0;
    if (q < Q && FL.inHeap[q]) {
      const QHot& f = FL.hot[q];
      if (f.sLen > f.sPos) {
        const bool kd = FL.sKind[q] != 0;
        if (!kd && !d.qsLen[2 * q + 1] && !f.effValid) { const int nx = f.itQi - 1 - f.sPos + f.sLen; if (nx < f.qEnd && d.jGang[k.queuedJobs[nx]] >= 0) b2 = 1; }
        if (!kd && f.sLen > QS_CMAX) b2 = 1;
      } else if (f.gctx < -1) b2 = 1;
    }
#if defined(ASCHED_HOSTSIM) || !defined(__HIP_DEVICE_COMPILE__)
    if (b2) bad = 1;
#else
    bad = b2;
#endif
  }
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  bad = __ballot(bad != 0) != 0;
#endif
  return !bad;
}


// Run fast iterations of the QueueScheduler loop (mode 0) or of the eviction-order replay (mode 1) until one needs the
// generic code.  Returns the queue whose next head the generic updateAndPush must produce, or -1.  Leaves fast mode live.
DEV_NOINLINE int fastRun(Dev& d, Ctl& c, const PassCfg& pc, int mode, int* counter) {
  FastCtx fc;
  fc.withQueued = pc.withQueued; fc.maxLookback = pc.maxLookback; fc.skipKnown = pc.skipKnown; fc.compareSchedPrio = c.compareSchedPrio;
  fc.preferLarge = c.preferLarge; fc.replay = mode; fc.evStatic = c.fastEvStatic; fc.engine = !mode && d.f.engine;
  fc.stream = 0;
  fastEnsureLive(d, c);
  c.l1Dirty = 1;
  const FastK k = fastKRef(d);
  FastS S; HS_POISON(S);
   S.engLive = 0; S.engPend = -1; S.engWaitClk = 0; S.engSeq = 0; S.inlineStreak = 0;
  if (FLANE == 0) FL.eng.live = 0;
  LANE0_PUBLISHED();
  S.laneL = FLANE / (d.cfg.R > 0 ? d.cfg.R : 1); S.laneX = FLANE % (d.cfg.R > 0 ? d.cfg.R : 1);
  // the scheduling-context scalars this loop keeps in registers
#define FAST_SCALARS_IN() \
  S.globalTokens = UNID(RS.globalTokens); S.globalBurst = UNI64(RS.globalBurst); S.globalRateInf = UNI32(RS.globalRateInf); \
  S.numScheduledJobs = UNI32(RS.numScheduledJobs); S.numScheduledGangs = UNI32(RS.numScheduledGangs); S.numEvictedJobs = UNI32(RS.numEvictedJobs); \
  S.numNodeQueries = UNI32(RS.numNodeQueries); S.loopIterations = UNI32(RS.loopIterations); S.evictedTableSize = UNI32(RS.evictedTableSize); \
  S.numUnfeasible = UNI32(RS.numUnfeasible); S.numPreemptedMarks = UNI32(RS.numPreemptedMarks); S.fastActive = UNI32(RS.fastActive); S.lvl0NonNeg = UNI32(RS.lvl0NonNeg); S.replayPending = UNI32(RS.replayPending); \
  S.statFastIters = UNI32(RS.statFastIters); S.statScanSteps = UNI32(RS.statScanSteps); S.statRefills = UNI32(RS.statRefills); S.statL0Max = UNI32(RS.statL0Max); S.statFastReplay = UNI32(RS.statFastReplay);
#define FAST_SCALARS_OUT() \
  RS.globalTokens = S.globalTokens; \
  RS.numScheduledJobs = S.numScheduledJobs; RS.numScheduledGangs = S.numScheduledGangs; RS.numEvictedJobs = S.numEvictedJobs; \
  RS.numNodeQueries = S.numNodeQueries; RS.loopIterations = S.loopIterations; \
  if (RS.evictedTableSize != S.evictedTableSize) { RS.fairIndexValid = 0; RS.ftValid = 0; }  /* the replay added table entries */ \
  RS.evictedTableSize = S.evictedTableSize; \
  RS.statFastIters = S.statFastIters; RS.statScanSteps = S.statScanSteps; RS.statRefills = S.statRefills; RS.statL0Max = S.statL0Max; RS.statFastReplay = S.statFastReplay;
  FAST_SCALARS_IN()
  int Q = UNI32(d.cfg.Q);
  fc.withQueued = UNI32(fc.withQueued); fc.maxLookback = UNI32(fc.maxLookback); fc.skipKnown = UNI32(fc.skipKnown); fc.compareSchedPrio = UNI32(fc.compareSchedPrio);
  fc.preferLarge = UNI32(fc.preferLarge); fc.evStatic = UNI32(fc.evStatic); fc.engine = UNI32(fc.engine);
  mode = UNI32(mode);
  int cnt = counter ? UNI32(*counter) : 0, pend = -1, lastTop = -1;
  fc.stream = fc.engine && d.qsKey != nullptr && !k.hasPcLimit && !k.anyRoundLimit && !k.disableHome && fc.withQueued;
  if (fc.stream && (d.fitPad_ & 1)) fc.stream |= 2;   // bit 1: after a gang its queue's next stretch of single jobs is prepared as a stream at once (round_run.h fastStreamPrepareOne; host flag)
  fc.stream = UNI32(fc.stream);
  int streamNextAt = UNI32(c.streamNextAt), streamBackoff = UNI32(c.streamBackoff), streamCap = UNI32(c.streamCap), cheapTry = 0;
  PackedKey refK; refK.A = ~0u; refK.X = refK.Y = ~0ull; uint32_t refN = ~0u; int haveRef = 0;
  if (!mode && c.onlyEvicted && RS.terminationReason != 0 && S.lvl0NonNeg && S.numPreemptedMarks == 0 && fc.evStatic && fc.withQueued) { SkipDelta dl = fastDrain(d, Q); S.numEvictedJobs += dl.evicted; S.loopIterations += dl.iters; }
  if (c.skipEnter && !mode && S.lvl0NonNeg && S.numPreemptedMarks == 0 && fc.evStatic) { SkipDelta dl = fastEnterSkip(d, fc, Q); S.numEvictedJobs += dl.evicted; S.loopIterations += dl.iters; S.statRefills += dl.refills; c.skipActive = 1; }
  c.skipEnter = 0;
  PQState pq;
  pqBuild(pq, Q);
  SEG_BEGIN();
  // collect the node engine's verdict on the speculative iteration in flight; on a miss take it back and stop (the generic code redoes it)
#define ENGINE_SETTLE(onMiss) \
  if (S.engPend >= 0) { \
    long long w0_ = CLK(); \
    int sq = S.engPend, sv = engineWait(S); \
    S.engWaitClk += CLK() - w0_; \
    S.engPend = -1; \
    if (sv == 2) { S.fastActive = 0; fastDrop(d); } \
    if (sv == 0) { \
      fastRollback(d, k, S, sq); \
      pqBuild(pq, Q); \
      lastTop = sq; pqHeadKey(pq, sq, &refK, &refN); \
      pend = -1; \
      onMiss; \
    } \
  }
  unsigned pollCount = 0;
  for (;;) {
    int missed = -1;
    ENGINE_SETTLE(missed = sq)
    // hard timeout / cancel (queue_scheduler.go:105-112): a live node engine looks at the host-mapped word itself (engineServeAt); without one this loop does
    if (!S.engLive && (++pollCount & 63) == 0 && cancelRequested(d)) { c.cancelSeen = 1; break; }
    int t = pqHead(pq, Q);
    SEG(0);
#ifdef ASCHED_HOSTSIM
    if (!c.skipActive) {  // in skip mode heads carry running-maximum keys the generic Less knows nothing about
      fastQFlush(d);
      if (t != pqTop(d, c)) {
        int g = pqTop(d, c);
        fprintf(stderr, "hostsim: packed queue key disagrees with Less (fast %d, generic %d) mode %d\n", t, g, mode);
        for (int q : {t, g}) if (q >= 0) { const QHot& h = FL.hot[q];
          fprintf(stderr, " q%d A %08x X %016llx Y %016llx prop %.17g cur %.17g size %.17g budget %.17g pcPrio %d eff %d kind %d gctx %d itEi %d evEnd %d itQi %d stage %d inHeap %d/%d\n", q, FL.kA[q],
            (unsigned long long)FL.kX[q], (unsigned long long)FL.kY[q], h.proposed, h.current, h.size, h.budget, h.pcPrio, h.effValid, h.headKind, h.gctx, h.itEi, h.evEnd, h.itQi, h.itStage, FL.inHeap[q], d.pqInHeap[q]); }
        abort();
      }
    }
#endif
    if (t < 0) { lastTop = t; break; }
    if (c.skipActive) {
      // the folded evicted streams (skip mode) are merged around the keys served so far, which must not decrease: a queue's next element may order
      // before an entry already served (a higher priority class behind a lower one; a gang behind the queue's own evicted jobs) — rebuild the exact
      // state around the last entry served and go on without the mode
      PackedKey curK; uint32_t curN;
      pqHeadKey(pq, t, &curK, &curN);
      if (UNI32((int)(haveRef && packedLess(curK, curN, refK, refN)))) {   // (a scalar branch: see the note at streamMerge)
        SkipDelta dl = fastExitSkip(d, fc, Q, lastTop, refK, refN);
        S.numEvictedJobs += dl.evicted; S.loopIterations += dl.iters; S.statRefills += dl.refills;
        c.skipActive = 0;
        pqBuild(pq, Q);
        continue;
      }
      refK = curK; refN = curN; haveRef = 1;
    }
    lastTop = t;
    int st = -1;
    if (missed >= 0) st = t == missed ? 8 : 0;   // the node engine found no node for this head (taken back above): as if fastIter had said so
    if (st < 0) {
    if (UNI32(FL.hot[t].gctx) < 0) {  // a gang: through the ring when every member is an untouched queued job (fastGangRun), else generic
      if (mode || !fc.stream || !S.fastActive || UNI32(FL.hot[t].gctx) == -1) break;
      if (!S.engLive) { engineStart(d, S); S.engLive = 1; }
      StreamIn in; in.globalTokens = S.globalTokens; in.globalBurst = S.globalBurst; in.globalRateInf = S.globalRateInf; in.engSeq = S.engSeq; in.resume = 0;
      in.skip = 0; in.haveLast = 0; in.lastA = in.lastN = 0; in.lastX = in.lastY = 0; in.bulkV = 0;
      GangOut go = fastGangRun(d, fc, in, t);
      S.engSeq = go.engSeq;
      if (go.dropped) { S.fastActive = 0; fastDrop(d); }
      
      if (!go.handled) break;
      S.numScheduledJobs += go.cnt; S.numScheduledGangs += 1; S.numNodeQueries += go.cnt; S.loopIterations++; S.statFastIters++;
      if (!S.globalRateInf && go.cnt <= S.globalBurst) S.globalTokens -= (double)go.cnt;
      S.statRefills += go.refills; S.numEvictedJobs += go.evicted;
      if (go.pend >= 0) { pend = go.pend; break; }
      { KeyOut gk; gk.valid = go.koValid; gk.A = go.koA; gk.X = go.koX; gk.Y = go.koY; pqPopPush(pq, gk, t); }   // the gang's queue is the head: re-insert it under its next key
      if ((fc.stream & 2) && go.koValid) {   // the single jobs behind the gang as a stream, prepared right here (the queue's old stream ended at the gang)
        int want = INT32_MAX;
        if (!S.globalRateInf) want = S.globalTokens >= 2147483000.0 ? INT32_MAX : (S.globalTokens < 1 ? 0 : (int)S.globalTokens);
        if (fastStreamPrepareOne(d, fc, t, want, streamCap) > 0) cheapTry = 1;   // a run may start again at once — with the streams there are, no bulk preparation
      }
      continue;
    }
    if (fc.stream && S.fastActive && (S.statFastIters >= streamNextAt || cheapTry)) {
      // stream run: the queues' next costs come precomputed (bulk passes where a queue has none left), this wave merges + stages, the node engine binds
      int want = INT32_MAX;
      if (!S.globalRateInf) want = S.globalTokens >= 2147483000.0 ? INT32_MAX : (S.globalTokens < 1 ? 0 : (int)S.globalTokens);
      // (fc.stream & 2, fastStreamPrepareOne) the node engine is never stopped for the bulk passes: queues that would need them go without a stream, and when the
      // queue at the top is one of them the control wave prepares its stream alone.  The bulk passes still run where no engine is live (the start of a pass).
      const int cheap = (fc.stream & 2) && (S.engLive || S.statFastIters < streamNextAt);
      cheapTry = 0;
      int code = fastStreamPrepare(d, fc, Q, want, cheap ? -1 : !S.engLive, t, streamCap);
      if (code == 2) { engineStop(d, S); S.engLive = 0; code = fastStreamPrepare(d, fc, Q, want, 1, t, streamCap); }   // the bulk passes need every wave of the workgroup at the mailbox
      if (code == 0 && (fc.stream & 2) && fastStreamPrepareOne(d, fc, t, want, streamCap) > 0) code = 1;
      int E = 0, so_max = 0;
      if (code == 1) {
        // (round 6) the run's merged order on every workgroup (round_merge.h) when the streams are long enough to pay for it: the node engine stops for the passes
        int bulkV = 0;
        if (d.mg && mgWorth(d, Q)) {
          if (S.engLive) { engineStop(d, S); S.engLive = 0; if (UNI32(FL.eng.cancel)) { c.cancelSeen = 1; break; } }
          bulkV = mgPrepare(d, fc, Q, c.skipActive, want);
        }
        if (!S.engLive) { engineStart(d, S); S.engLive = 1; }
        StreamIn in; in.globalTokens = S.globalTokens; in.globalBurst = S.globalBurst; in.globalRateInf = S.globalRateInf; in.engSeq = S.engSeq; in.bulkV = bulkV;
        in.skip = c.skipActive; in.haveLast = haveRef; in.lastA = refK.A; in.lastX = refK.X; in.lastY = refK.Y; in.lastN = refN;   // (the current top's key: it is the first entry of the run)
        in.resume = 0;
        StreamOut so = fastStreamRun(d, fc, Q, in);
        while (so.event) {   // the run is parked at one of its rare events (see runPark): run it from here, then the run again
          if (so.event == 1) streamNestSettle(d, fc, so.evT); else streamNestGang(d, fc, in, so.evT);
          in.resume = 1;
          so = fastStreamRun(d, fc, Q, in);
        }
        S.engSeq = so.engSeq;
        E = so.executed; so_max = so.maxConsumed;
        S.numScheduledJobs += E; S.numScheduledGangs += E; S.numNodeQueries += E;
        if (!S.globalRateInf && 1 <= S.globalBurst) S.globalTokens -= (double)E;
        if (so.gangs) {   // gangs placed inside the run: accounted like fastGangRun's (cnt new jobs, ONE scheduled gang, ReserveN(cnt) on the global limiter, one loop iteration)
          S.numScheduledJobs += so.gangJobs; S.numScheduledGangs += so.gangs; S.numNodeQueries += so.gangJobs;
          if (!S.globalRateInf) S.globalTokens -= (double)so.gangJobs;
          S.loopIterations += so.gangs; S.statFastIters += so.gangs;
        }
        E += so.executedEv;
        S.loopIterations += E; S.statFastIters += E;
        S.statRefills += so.refills; S.numEvictedJobs += so.evicted;
        if (FLANE == 0) { RS.statStreamRuns++; RS.statStreamJobs += E; RS.statStreamEmitted += so.emitted; }
        LANE0_PUBLISHED();
        if (so.dropped) { S.fastActive = 0; fastDrop(d); }
        
        pqBuild(pq, Q);
        if (so.pend >= 0) pend = so.pend;
        if (c.skipActive && so.emitted > 0) {
          if (!so.failed) { lastTop = so.lastQ; refK.A = so.lastA; refK.X = so.lastX; refK.Y = so.lastY; refN = so.lastN; haveRef = 1; }   // every emitted entry was served
          else {   // served: the entries before the one that found no node, all of them ordering before it; it is the head of the heap again (nothing emitted after it was done)
            int t2 = pqHead(pq, Q);
            lastTop = t2; if (t2 >= 0) pqHeadKey(pq, t2, &refK, &refN);
          }
        }
      }
      (void)so_max;   // (entries prepared per queue stay at QS_CMAX: the sum pass stops at a queue's first gang member, so gang-heavy queues prepare little anyway)
      if (fc.stream & 2) {   // attempts are cheap: try again at once after a run that got somewhere, after a few iterations otherwise
        streamBackoff = (code == 1 && E >= 1) ? 0 : (streamBackoff ? (streamBackoff < 64 ? streamBackoff * 2 : streamBackoff) : 4);
        streamNextAt = S.statFastIters + streamBackoff;
      } else
      if (!cheap) {   // (the back-off belongs to the attempts that may run the bulk passes)
      if (E >= 64) streamBackoff = 0;   // an attempt costs little (streams persist, the bulk passes run on the helper workgroups): back off gently, but for good when runs stay short
      else streamBackoff = streamBackoff ? (streamBackoff < ASCHED_STREAM_BACKOFF_MAX ? streamBackoff * 2 : streamBackoff) : ASCHED_STREAM_BACKOFF_MIN;
      streamNextAt = S.statFastIters + streamBackoff;
#ifdef ASCHED_HOSTSIM
      if (getenv("HS_STREAM_EAGER")) { streamBackoff = 0; streamNextAt = S.statFastIters + (E == 0 ? 1 : 0); }   // tests: a stream run wherever one can start
#endif
      }
      if (pend >= 0) break;
      if (S.engLive && UNI32(FL.eng.cancel)) break;   // the engine saw the caller's cancel word: leave the loop (queueSchedule raises the timeout)
      if (code == 1) continue;
    }
    }
    KeyOut ko; ko.valid = 0; ko.A = 0; ko.X = ko.Y = 0;
    if (st < 0) st = mode ? fastReplayStep(d, k, S, fc, t, &cnt, &ko) : fastIter(d, k, S, fc, t, &ko);
    if (st == 8) {   // the job needs preemption: the node side through the generic cascade, the queue side stays here (fastPreemptIter)
      if (c.skipActive) {   // it reads the nodes of evicted jobs: the folded evicted streams must be in their exact state first
        SkipDelta dl = fastExitSkip(d, fc, Q, lastTop, refK, refN);
        S.numEvictedJobs += dl.evicted; S.loopIterations += dl.iters; S.statRefills += dl.refills;
        c.skipActive = 0;
        pqBuild(pq, Q);
        continue;
      }
      if (S.engLive) { engineStop(d, S); S.engLive = 0; if (UNI32(FL.eng.cancel)) { c.cancelSeen = 1; break; } }
      S.inlineStreak = 0;
      FAST_SCALARS_OUT()
      int hc = fastPreemptIter(d, c, fc, t, &ko);
      FAST_SCALARS_IN()
      if (hc == 0) break;
      pqPopPush(pq, ko, t);   // (only the served queue's entry changes: re-inserting it is one lane shift, sorting the heap again is 64 rounds of lane exchanges)
      XSEG(39);
      if (hc == 2) { pend = t; break; }
      if (hc == 3 || RS.error) break;
      continue;
    }
    if (st == 0) break;
    pqPopPush(pq, ko, t);
    SEG(7);
    if (!mode) { if (!(st & 4)) S.loopIterations++; S.statFastIters++; }
    if ((st & 3) == 2) { pend = t; break; }  // refK / refN: the entry just served
  }
  ENGINE_SETTLE((void)0)
#undef ENGINE_SETTLE
  if (S.engLive) { engineStop(d, S); S.engLive = 0; if (UNI32(FL.eng.cancel)) c.cancelSeen = 1; }
#ifndef ASCHED_FASTPROF
  if (FLANE == 0 && S.engWaitClk) RS.statSeg[0] += S.engWaitClk;
  LANE0_PUBLISHED();
#endif
  if (c.skipActive) {  // generic code comes next: rebuild the exact interleaved state
    SkipDelta dl = fastExitSkip(d, fc, Q, lastTop, refK, refN);
    S.numEvictedJobs += dl.evicted; S.loopIterations += dl.iters; S.statRefills += dl.refills;
    c.skipActive = 0;
  }
  c.streamNextAt = streamNextAt; c.streamBackoff = streamBackoff; c.streamCap = streamCap;
  if (counter) *counter = cnt;
  FAST_SCALARS_OUT()
#undef FAST_SCALARS_IN
#undef FAST_SCALARS_OUT
  return pend;
}
