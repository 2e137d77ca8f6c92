// round_ctl.h — the sequential part of the scheduling round, written once as "wave-uniform" code.
//
// On the GPU this code is executed by wave 0 of a single persistent workgroup (all 64 lanes run the
// same control flow on the same data — stores of identical values by 64 lanes are benign), and it calls
// workgroup-wide primitives for the data-parallel parts:
//     wgFirstFit()  : LDS-staged argmin scan over the (level, resource) planes of `alloc` and `keys`
//     pqTop()       : lane-per-queue DRF argmin with the reference's Less (queue_scheduler.go:738-798)
//     wgForEach()   : block-stride loops for the bulk phases (evictors, unbind, result compaction)
// The same file compiles for the host with -DASCHED_HOSTSIM (tests/hostsim): there the primitives
// degrade to serial loops, which lets the control logic be debugged in a GPU-less container.  The
// host build is test infrastructure only and is never loaded by the product.
//
// Each function cites the reference code it restates (is/ = internal/scheduler/).
#pragma once
#include "dev.h"
#include "../../include/armada_sched.h"

#ifdef ASCHED_HOSTSIM
#define CLK() 0ll
#include <cmath>
#include <cstring>
#define DEV static inline
#define DEV_COLD static
#define WG_THREADS 1
#else
#define DEV __device__ static inline
// the big generic-path functions: out of line.  Inlined into every call site they made the round kernel's code several times larger (the
// whole cascade sits in selectNodeForJob twice — home and away), which costs compile time and instruction-cache room next to the hot loop
#define DEV_COLD __device__ static __attribute__((noinline))
#define WG_THREADS 1024
#define CLK() ((long long)__builtin_readcyclecounter())
#endif

// ------------------------------------------------------------------------------------------------
// workgroup primitives (implemented differently per build)
// min key >= lowBound among masked nodes (that fit req at level, unless noFit).  levelHi > level: multi-level mode — per node the LOWEST level in
// [level, levelHi] at which it fits, result = min over nodes of (that level << 60 | key at that level): the urgency sweep and the feasibility gate
// of one job in one pass over the planes (selectAtPriority)
#ifdef ASCHED_TWO_WORD_KEYS
#define SCAN_NO_LOW_WORD(a) ((a).lowBoundLo = 0)
#else
#define SCAN_NO_LOW_WORD(a) ((void)0)
#endif
#ifdef ASCHED_TWO_WORD_KEYS
// (two-word keys: lowBound / lowBoundLo = the bound's two words; pad = 1: the second pass of a selection — lowBound is the high word to match, lowBoundLo the least low word)
struct ScanArgs { int64_t req[MAXR]; const uint64_t* maskA; const uint64_t* maskB; int level; int noFit; uint64_t lowBound; int32_t levelHi, pad; uint64_t lowBoundLo; };
#else
struct ScanArgs { int64_t req[MAXR]; const uint64_t* maskA; const uint64_t* maskB; int level; int noFit; uint64_t lowBound; int32_t levelHi, pad; };
#endif
#define SCAN_LEVEL_SHIFT 60

struct FairArgs { int64_t req[MAXR]; const uint64_t* maskA; const uint64_t* maskB; int32_t prio; int32_t pad; };
#define FAIR_CHUNKS 256
#define FAIR_BAD_ENTRY 0x7ffffff0   // an alive table entry without a scheduled-at priority was met (nodedb.go:950-953)

DEV int wgFirstFit(Dev& d, const ScanArgs& a);                       // -> node or -1
DEV uint64_t wgFirstFitKey(Dev& d, const ScanArgs& a);               // -> the raw minimum (~0 = none): packed key, in multi-level mode tagged with the level
DEV int wgScanFair(Dev& d, const ScanArgs& a, const FairArgs& f, uint64_t* bestKey);   // both in ONE wide pass (one hand-shake with the helper workgroups): *bestKey = wgFirstFitKey(a), returns wgFairSelect(f)
DEV int wgFairSelect(Dev& d, const FairArgs& a);                     // -> evicted-table Index or -1 (max over nodes of fairNodeBest)
DEV int atomicFetchAddI32(int32_t* p, int32_t v);
#if !defined(ASCHED_HOSTSIM)
DEV int waveMax32(int v);
#endif
template <class F> DEV void wgForEach(Dev& d, int n, F f);           // f(i) for i in [0,n), then workgroup barrier
DEV int wgCompact(Dev& d, const int32_t* src, int n, const uint8_t* flagByValue, int32_t* dst, const int32_t* segOff, int nseg, int32_t* outSegOff);
DEV void atomicAddI64(int64_t* p, int64_t v);
DEV void atomicAddI32(int32_t* p, int32_t v);
DEV void atomicOrI32(int32_t* p, int32_t v);
DEV void atomicMinU32(uint32_t* p, uint32_t v);

// fast path (round_fast.h): level-0 sorted base + LDS delta, LDS-resident heads, key-based queue argmin
struct Ctl; struct PassCfg;
DEV void fastTouch(Dev& d, int n);                 // node n's allocatable changed through the generic code
DEV int fastSelectLevel0(Dev& d, int job);         // first fit at priority -2 through the fast structure (-2 = structure not usable)
#ifdef ASCHED_HOSTSIM
static int fastRun(Dev& d, Ctl& c, const PassCfg& pc, int mode, int* counter);  // fast iterations until one needs the generic code
#else
__device__ static int fastRun(Dev& d, Ctl& c, const PassCfg& pc, int mode, int* counter);
#endif
DEV void fastItemKeys(Dev& d, const Ctl& c, int q);
DEV void fastHeadInvalidate(int q);
DEV void fastPassReset();
DEV bool fastOn(Dev& d, const Ctl& c);
DEV void fastEnterGeneric(Dev& d, Ctl& c);
DEV bool fastGangMember(Dev& d, Ctl& c, int job);  // select (fit at priority -2) + bind of one unpinned queued gang member through the fast structure; false = not done
DEV void fastFence(Ctl& c);
DEV_COLD void ensureReplaySlow(Dev& d, Ctl& c);
// pools of more than QCAPF queues (round_wide.h): stream runs whose k-way merge is a bulk rank over all queues' precomputed key sequences
DEV int skipUnfeasibleBulk(Dev& d, int pos, int max);   // round_wide.h: the same on every workgroup of the launch (two passes), for long stretches; the caller's other waves must be parked (no live node engine)
DEV int skipUnfeasibleRun(Dev& d, int pos, int max);   // round_fast.h: Peek's skip of known-unfeasible scheduling keys (queue_scheduler.go:398-413) for a stretch of queued jobs, 64 at a time
DEV_COLD int wideRun(Dev& d, Ctl& c, const PassCfg& pc);
DEV bool wideHeadOk(Dev& d, const Ctl& c, const PassCfg& pc, int t);
DEV_COLD void exclRecordWide(Dev& d, int job, int level, int gate = -1);   // round_wide.h "excluded nodes": what asched_excluded_nodes needs of an attempt that found no node at the job's priority
DEV_COLD void exclPinned(Dev& d, int job, int node, int level);
DEV void exclForget(Dev& d, int job) { if (d.excl) d.excl[job] = -1; }   // the job got a node: its PodSchedulingContext is a new one
DEV void ensureReplay(Dev& d, Ctl& c) { if (d.rs->replayPending) ensureReplaySlow(d, c); }   // (the test stays with the caller: a call costs a register save / restore)
//             // run the deferred eviction-order replay before anything reads the evicted table         // leave fast mode: LDS queue state back into the generic arrays, HBM updates visible

// ------------------------------------------------------------------------------------------------
#define AL(d, l, r, n) ((d).alloc[((size_t)(l) * (d).cfg.R + (r)) * (d).cfg.Npad + (n)])
#define KEY(d, l, n) ((d).keys[(size_t)(l) * (d).cfg.Npad + (n)])
#define KEYLO(d, l, n) ((d).keys[((size_t)(d).cfg.P + (l)) * (d).cfg.Npad + (n)])   // two-word keys (dev.h keyWords): KEY is the high word, this the low one
#define JREQ(d, j) ((d).jReq + (size_t)(j) * (d).cfg.R)

#if defined(ASCHED_FASTPROF) && defined(__HIP_DEVICE_COMPILE__)
#define XSEG_BEGIN() (d.rs->gsT = CLK())
#define XSEG(i) do { long long n_ = CLK(); if ((threadIdx.x & 63) == 0) { d.rs->statSeg[i] += n_ - d.rs->gsT; d.rs->gsT = n_; } } while (0)
#else
#define XSEG_BEGIN() do {} while (0)
#define XSEG(i) do {} while (0)
#endif
DEV void raise(Dev& d, int code, int detail) {
  if (d.rs->error == 0) { d.rs->error = code; d.rs->errorDetail = detail; }
}

// The caller's context is done (armadacontext with maxSchedulingDuration, scheduling_algo.go:130-134; checked once per loop iteration,
// queue_scheduler.go:105-112): the host flips a word in host-mapped memory (deadline reached, or asched_cancel from another thread).
// A read crosses PCIe (~2 us): call sites poll it every few hundred iterations.
DEV bool cancelRequested(const Dev& d) {
#ifdef ASCHED_HOSTSIM
  return d.cancel && *d.cancel != 0;
#else
  return d.cancel && __hip_atomic_load((const int32_t*)d.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
#endif
}
// sch.clock.Now() in nanoseconds for the soft time budgets (queue_scheduler.go:157, 222-228): the device's constant-rate wall clock
DEV int64_t clockNowNs(const Dev& d) {
#ifdef ASCHED_HOSTSIM
  (void)d;
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
#else
  int64_t khz = d.cfg.wallClockKHz > 0 ? d.cfg.wallClockKHz : 100000;
  return (int64_t)wall_clock64() * 1000000ll / khz;
#endif
}

DEV int levelOf(const DevCfg& c, int32_t prio) {
  for (int i = 0; i < c.P; i++) if (c.prios[i] == prio) return i;
  return -1;
}

// packed order key of node n at level l: (rounded indexed columns ..., rank of node.index)
// == RoundedNodeIndexKeyFromResourceList (is/nodedb/encoding.go:37-54) as one integer compare.
DEV uint64_t packKey(Dev& d, int l, int n) {
  const DevCfg& c = d.cfg;
  uint64_t k = (uint64_t)d.idxRank[n];
  for (int i = 0; i < c.K; i++) {
    int64_t q = AL(d, l, c.indexedCol[i], n) / c.indexedRes[i];  // truncation toward zero, like Go (encoding.go:56-58)
    // quotients below the layout's lowest value share field 0 (asched_host.inc layoutKeys: with non-negative requests such a node fits nothing at this
    // level, so its place among the others like it is never observed); above allocatable is out of range (an unbind after ClearAllocated)
    int64_t f = q - c.keyLo[i];
    if (f < 0 && c.keyClamp) f = 0;
    if (f < 0 || (c.keyWidth[i] < 63 && f >= ((int64_t)1 << c.keyWidth[i]))) { raise(d, ASCHED_ERR_UNSUPPORTED, 100 + i); f = 0; }
    k |= (uint64_t)f << c.keyShift[i];
  }
  return k;
}
// The level-0 key as the fast structure takes it in.  Its queries test key fields only — "field >= what the shape needs" is "allocatable >= request" for
// requests that are non-negative multiples of the resolution and allocatable >= 0 — but a column in (-resolution, 0) truncates to quotient 0 like the
// reference's key (encoding.go:56-58: Go division) although nothing fits there: such a field is forced to 0, below what any shape needs.  (Allocatable that
// is not a multiple of the resolution is otherwise harmless: floor((a - k*res) / res) = floor(a / res) - k while the result is >= 0.)
DEV uint64_t fastKeyOf(const Dev& d, int n) {
  uint64_t key = KEY(d, 0, n);
  for (int i = 0; i < d.cfg.K; i++) if (AL(d, 0, d.cfg.indexedCol[i], n) < 0) key &= ~d.f.fieldMask[i];
  return key;
}
// the same key as two words (dev.h keyWords == 2): ONE 128-bit integer — field i at bit keyShift[i] (< 128; a field may lie across the word boundary), the node-index
// rank in the low idxBits — stored as its high and its low 64 bits
typedef unsigned __int128 Key2;
#define KEY2_NONE (~(Key2)0)
DEV Key2 key2Of(uint64_t hi, uint64_t lo) { return ((Key2)hi << 64) | lo; }
DEV Key2 packKey2(Dev& d, int l, int n) {
  const DevCfg& c = d.cfg;
  Key2 k = (Key2)(uint64_t)d.idxRank[n];
  for (int i = 0; i < c.K; i++) {
    int64_t q = AL(d, l, c.indexedCol[i], n) / c.indexedRes[i];
    int64_t f = q - c.keyLo[i];
    if (f < 0 && c.keyClamp) f = 0;
    if (f < 0 || (c.keyWidth[i] < 63 && f >= ((int64_t)1 << c.keyWidth[i]))) { raise(d, ASCHED_ERR_UNSUPPORTED, 100 + i); f = 0; }
    k |= (Key2)(uint64_t)f << c.keyShift[i];
  }
  return k;
}
DEV void storeKey(Dev& d, int l, int n) {
  if (WIDE_KEYS(d.cfg)) { Key2 k = packKey2(d, l, n); KEY(d, l, n) = (uint64_t)(k >> 64); KEYLO(d, l, n) = (uint64_t)k; }
  else KEY(d, l, n) = packKey(d, l, n);
}
// node a orders before node b at this level (both words of a two-word key)
DEV bool keyBefore(Dev& d, int l, int a, int b) {
  uint64_t ka = KEY(d, l, a), kb = KEY(d, l, b);
  if (WIDE_KEYS(d.cfg) && ka == kb) return KEYLO(d, l, a) < KEYLO(d, l, b);
  return ka < kb;
}
#ifdef ASCHED_TWO_WORD_KEYS
DEV void updateKeys(Dev& d, int n) { for (int l = 0; l < d.cfg.P; l++) storeKey(d, l, n); }
#else   // (the one-word kernels: the statements as they were — their instruction text is pinned, tools/kcontrol_isa_hash.sh)
DEV void updateKeys(Dev& d, int n) { for (int l = 0; l < d.cfg.P; l++) KEY(d, l, n) = packKey(d, l, n); }
#endif
// The control code runs on one full wave whose lanes all execute the same statements.  Where a statement is a loop over (level, resource) elements of one
// node, the lanes take one element each instead: the loop's dependent HBM round trips (~0.5 us each) become one.  CTL_WAVE(): this really is a full wave.
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
#define CTL_WAVE() (__builtin_popcountll(__builtin_amdgcn_read_exec()) == 64)
#define CTL_LANE() ((int)(threadIdx.x & 63))
#else
#define CTL_WAVE() false
#define CTL_LANE() 0
#endif
DEV_COLD void ftUpdateNode(Dev& d, int n);   // round_ft.h: the fair-share threshold table follows every change the generic code makes to a node
DEV_COLD int ftQuery(Dev& d, int s);
DEV_COLD void ftAfterAbort(Dev& d, int undoCount);
// The fair-share threshold table (round_ft.h) is NOT part of the default device build.  Measured on the MI355X (profiles/r03g_*): BASELINE configs[4] at full
// size 17.2 -> 15.1 s per round, the same shape at 20k nodes 2.55 -> 2.70 s (half of the preempting jobs need urgency preemption, which still takes a wide
// pass), and its mere presence in k_control costs the headline 2-3 % (code placement: profiles/r03f_*).  -DASCHED_WITH_FT builds it in; the CPU build of the
// tests always has it (its logic is soaked against the oracle like everything else).
#if !defined(ASCHED_WITH_FT) && !defined(ASCHED_HOSTSIM)
#define ASCHED_NO_FT 1
#endif
#ifdef ASCHED_NO_FT
DEV void ftTouch(Dev&, int) {}
#else
DEV void ftTouch(Dev& d, int n) { if (d.ftT && d.rs->ftValid) ftUpdateNode(d, n); }
#endif
DEV void updateKeysCtl(Dev& d, int n) {  // control-flow call sites (not the bulk rebuild)
#ifdef ASCHED_TWO_WORD_KEYS
  if (CTL_WAVE()) { int l = CTL_LANE(); if (l < d.cfg.P) storeKey(d, l, n); }   // one level per lane
#else
  if (CTL_WAVE()) { int l = CTL_LANE(); if (l < d.cfg.P) KEY(d, l, n) = packKey(d, l, n); }   // one level per lane
#endif
  else updateKeys(d, n);
  fastTouch(d, n);
  ftTouch(d, n);
}

DEV bool fitsAlloc(Dev& d, const int64_t* req, int level, int n) {  // DynamicJobRequirementsMet (is/nodedb/nodematching.go:194-197)
  for (int r = 0; r < d.cfg.R; r++) if (req[r] > AL(d, level, r, n)) return false;
  return true;
}

// ---- undo log (memdb write txn: gang attempts are aborted on failure, gang_scheduler.go:234-243)
enum { U_ADD = 1, U_REMOVE = 2, U_EVTAB_DEL = 3, U_EVTAB_INS = 4 };
struct Txn { int active; };
DEV void undoPush(Dev& d, int op, int a, int b, int c) {
  int i = d.rs->undoCount;
  if (i >= d.undoCap) { raise(d, ASCHED_ERR_INTERNAL, 200); return; }
  d.undo[i * 4 + 0] = op; d.undo[i * 4 + 1] = a; d.undo[i * 4 + 2] = b; d.undo[i * 4 + 3] = c;
  d.rs->undoCount = i + 1;
}

// markAllocatable (is/internaltypes/node.go:539-549): alloc[p] += sign*req for every level p <= cutoff
DEV void markAllocatable(Dev& d, int n, int32_t cutoff, const int64_t* req, int sign) {
  const DevCfg& c = d.cfg;
  if (CTL_WAVE()) {   // one (level, resource) element per lane
    for (int i = CTL_LANE(); i < c.P * c.R; i += 64) {
      int l = i / c.R, r = i % c.R;
      if (c.prios[l] <= cutoff && req[r] != 0) AL(d, l, r, n) += sign * req[r];
    }
    return;
  }
  for (int l = 0; l < c.P; l++)
    if (c.prios[l] <= cutoff)
      for (int r = 0; r < c.R; r++) AL(d, l, r, n) += sign * req[r];
}
// bindJobToNodeInPlace (nodedb.go:1055-1068): a cross-pool ("away") job is accounted at CrossPoolPriority whatever priority it is bound with — when the
// NodeDb knows its pool, which it does exactly when cross-pool preemption ordering is on for the pool (scheduling_algo.go:759-764)
DEV int32_t bindPriority(const Dev& d, int job, int32_t prio) { return (d.cfg.preferHome && d.jAway && d.jAway[job]) ? ASCHED_CROSS_POOL_PRIORITY : prio; }
DEV int32_t cutoffFor(Dev& d, int job, int32_t prio) {  // priorityCutoffFor (is/nodedb/nodedb.go:1329-1334)
  return d.cfg.pcPreemptible[d.jPc[job]] ? prio : NONPREEMPTIBLE_CUTOFF;
}

// Node.AddJob (node.go:416-442); aux bits recorded for undo: bit0 = wasEvicted, old cutoff in c
DEV int addJob(Dev& d, int n, int job, int32_t cutoff, bool log) {
  bool onNode = d.jobNode[job] == n;
  bool wasEvicted = onNode && d.jobEvictedOnNode[job];
  if (!wasEvicted && d.jobNode[job] >= 0) { raise(d, ASCHED_ERR_INTERNAL, 300); return -1; }  // "job already has resources allocated"
  int32_t oldCutoff = d.jobCutoff[job];
  d.jobEvictedOnNode[job] = 0;
  d.jobNode[job] = n;
  const int64_t* req = JREQ(d, job);
  markAllocatable(d, n, cutoff, req, -1);
  if (wasEvicted) markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, +1);
  d.jobCutoff[job] = cutoff;
  if (log) undoPush(d, U_ADD | (wasEvicted ? 256 : 0), job, n, oldCutoff);
  return 0;
}
// Node.RemoveJob (node.go:480-506); unknown job => no-op
DEV void removeJob(Dev& d, int n, int job, bool log) {
  if (d.jobNode[job] != n) return;
  const int64_t* req = JREQ(d, job);
  bool wasEvicted = d.jobEvictedOnNode[job];
  int32_t cutoff = d.jobCutoff[job];
  if (wasEvicted) markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, +1);
  else markAllocatable(d, n, cutoff, req, +1);
  d.jobEvictedOnNode[job] = 0;
  d.jobNode[job] = -1;
  if (log) undoPush(d, U_REMOVE | (wasEvicted ? 256 : 0), job, n, cutoff);
}
// Node.EvictJob (node.go:449-474)
DEV int evictJobOnNode(Dev& d, int n, int job) {
  if (d.jobNode[job] != n) { raise(d, ASCHED_ERR_INTERNAL, 301); return -1; }
  if (d.jobEvictedOnNode[job]) { raise(d, ASCHED_ERR_INTERNAL, 302); return -1; }
  d.jobEvictedOnNode[job] = 1;
  const int64_t* req = JREQ(d, job);
  markAllocatable(d, n, d.jobCutoff[job], req, +1);
  markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, -1);
  return 0;
}

DEV void evTabDelete(Dev& d, int idx, bool log) {
  if (!d.evTabAlive[idx]) return;
  d.evTabAlive[idx] = 0;
  d.evIndexOfJob[d.evTabJob[idx]] = -1;
  if (log) undoPush(d, U_EVTAB_DEL, idx, 0, 0);
}
DEV void evTabInsert(Dev& d, int idx, int job) {
  d.rs->fairIndexValid = 0; d.rs->ftValid = 0;
  d.evTabJob[idx] = job; d.evTabAlive[idx] = 1; d.evIndexOfJob[job] = idx;
  if (idx + 1 > d.rs->evictedTableSize) d.rs->evictedTableSize = idx + 1;
}

DEV void txnBegin(Dev& d, Txn& t) { t.active = 1; d.rs->undoCount = 0; }
DEV void txnCommit(Dev& d, Txn& t) { t.active = 0; d.rs->undoCount = 0; }
DEV void txnAbort(Dev& d, Txn& t) {
  if (!t.active) return;
  t.active = 0;
  for (int i = d.rs->undoCount - 1; i >= 0; i--) {
    int op = d.undo[i * 4], a = d.undo[i * 4 + 1], b = d.undo[i * 4 + 2], c = d.undo[i * 4 + 3];
    int kind = op & 255; bool wasEvicted = op & 256;
    if (kind == U_ADD) {
      int job = a, n = b;
      const int64_t* req = JREQ(d, job);
      markAllocatable(d, n, d.jobCutoff[job], req, +1);
      if (wasEvicted) { markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, -1); d.jobEvictedOnNode[job] = 1; d.jobCutoff[job] = c; }
      else { d.jobNode[job] = -1; d.jobCutoff[job] = c; }
      updateKeysCtl(d, n);
    } else if (kind == U_REMOVE) {
      int job = a, n = b;
      const int64_t* req = JREQ(d, job);
      if (wasEvicted) markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, -1);
      else markAllocatable(d, n, c, req, -1);
      d.jobNode[job] = n; d.jobCutoff[job] = c; d.jobEvictedOnNode[job] = wasEvicted ? 1 : 0;
      updateKeysCtl(d, n);
    } else if (kind == U_EVTAB_DEL) {
      d.evTabAlive[a] = 1; d.evIndexOfJob[d.evTabJob[a]] = a;
      d.rs->fairIndexValid = 0; d.rs->ftValid = 0;   // an entry comes back: the per-node index may have been built without it (ensureFairIndex)
    }
  }
#ifndef ASCHED_NO_FT
  if (d.ftT && d.rs->ftValid && d.rs->undoCount > 0) ftAfterAbort(d, d.rs->undoCount);   // (round_ft.h; out of line: txnAbort is inlined into the fast loop's preempting iteration)
#endif
  d.rs->undoCount = 0;
}

// ------------------------------------------------------------------------------------------------
// gang context references: >=0 single job; <=-2 gang (dense id g = -(ref)-2); -1 nil
DEV int gcCount(Dev& d, int ref) { return ref >= 0 ? 1 : d.gangSeen[-ref - 2]; }
DEV int gcJob(Dev& d, int ref, int k) { return ref >= 0 ? ref : d.gangArr[d.gangOff[-ref - 2] + k]; }
DEV const int64_t* gcTotal(Dev& d, int ref) { return ref >= 0 ? JREQ(d, ref) : d.gangTotal + (size_t)(-ref - 2) * d.cfg.R; }
DEV bool gcAllEvicted(Dev& d, int ref) { return ref >= 0 ? d.jcEvicted[ref] : d.gangAllEvicted[-ref - 2]; }
DEV bool gcIsGang(Dev& d, int ref) { return d.jGang[gcJob(d, ref, 0)] >= 0; }
DEV int gcQueue(Dev& d, int ref) { return d.jQueue[gcJob(d, ref, 0)]; }

DEV double drf(Dev& d, const int64_t* a) {  // fairness.go:103-105 (float64, this operation order; -ffp-contract=off)
  const DevCfg& c = d.cfg;
  double m = -INFINITY;
  for (int i = 0; i < c.R; i++) {
    double f = 0.0;
    if (c.totalResources[i] != 0) f = (double)a[i] / (double)c.totalResources[i];
    double x = f * c.drfMult[i];
    if (x > m) m = x;
  }
  return m > 0 ? m : 0.0;
}

// ------------------------------------------------------------------------------------------------
// scheduling-context accounting
#define QV(arr, q) ((arr) + (size_t)(q) * d.cfg.R)
#define QPV(arr, q, pc) ((arr) + ((size_t)(q) * d.cfg.npc + (pc)) * d.cfg.R)
DEV void vadd(Dev& d, int64_t* a, const int64_t* b, int sign) { for (int r = 0; r < d.cfg.R; r++) a[r] += sign * b[r]; }

// qctx.addJobSchedulingContext + sctx.AddJobSchedulingContext (context/queue.go:231-265, scheduling.go:410-434)
// The market-driven round exists in the auxiliary kernel and the CPU build only (round_mkt.h): the round kernel's own translation unit compiles none of it.
#if (defined(ASCHED_AUX_TU) || defined(ASCHED_WK_TU) || defined(ASCHED_HOSTSIM)) && !defined(ASCHED_MARKET_ROUND)
#define ASCHED_MARKET_ROUND 1
#endif
#include "round_mkt.h"
DEV bool sctxAddJob(Dev& d, int job) {
  int q = d.jQueue[job], pc = d.jPc[job];
  const int64_t* req = JREQ(d, job);
  if (d.jobFlags[job] & F_SUCCESSFUL) { raise(d, ASCHED_ERR_INTERNAL, 401); return false; }   // "failed adding job to queue: job already marked successful" (context/queue.go:232-234)
  uint8_t f = d.jobFlags[job] & ~F_UNSUCCESSFUL;
  bool evictedInRound = f & F_EVICTED;
  RoundScalars& s = *d.rs;
  if (d.jcReason[job] == 0) {
    vadd(d, QPV(d.qAllocByPc, q, pc), req, +1);
    vadd(d, QV(d.qAlloc, q), req, +1);
    if (evictedInRound) {
      f &= ~F_EVICTED; f |= F_RESCHEDULED;
      vadd(d, QPV(d.qEvictedByPc, q, pc), req, -1);
      vadd(d, s.evicted, req, -1); s.numEvictedJobs--;
    } else {
      f |= F_SUCCESSFUL;
      vadd(d, QPV(d.qSchedByPc, q, pc), req, +1);
      vadd(d, s.scheduled, req, +1); s.numScheduledJobs++;
    }
    vadd(d, s.allocated, req, +1);
  } else {
    f |= F_UNSUCCESSFUL;
  }
  d.jobFlags[job] = f;
  return evictedInRound;
}
DEV void sctxAddGang(Dev& d, int ref) {  // scheduling.go:391-406
  bool allEvicted = true, allOk = true;
  int n = gcCount(d, ref);
  for (int k = 0; k < n; k++) { int j = gcJob(d, ref, k); bool ev = sctxAddJob(d, j); allEvicted = allEvicted && ev; allOk = allOk && d.jcReason[j] == 0; }
  if (allOk && !allEvicted) d.rs->numScheduledGangs++;
}
// qctx.evictJob + sctx.EvictJob (queue.go:351-386, scheduling.go:551-572)
DEV bool sctxEvictJob(Dev& d, int job) {
  int q = d.jQueue[job], pc = d.jPc[job];
  const int64_t* req = JREQ(d, job);
  uint8_t f = d.jobFlags[job];
  RoundScalars& s = *d.rs;
  if (f & (F_UNSUCCESSFUL | F_EVICTED)) { raise(d, ASCHED_ERR_INTERNAL, 400); return false; }
  bool sched = f & F_SUCCESSFUL, resched = f & F_RESCHEDULED;
  if (sched || resched) {
    if (sched) { vadd(d, QPV(d.qSchedByPc, q, pc), req, -1); f &= ~F_SUCCESSFUL; }
    if (resched) f &= ~F_RESCHEDULED;
    // context/queue.go:368-376: a billable jctx that leaves the queue context takes its AllResourceRequirements out of the bill
    MK(if (mkOn(d) && MKD.jobBillable[job]) { vadd(d, QV(MKD.qBillable, q), req, -1); MKD.jobBillable[job] = 0; })
  } else {
    vadd(d, QPV(d.qEvictedByPc, q, pc), req, +1);
    f |= F_EVICTED;
  }
  vadd(d, QPV(d.qAllocByPc, q, pc), req, -1);
  vadd(d, QV(d.qAlloc, q), req, -1);
  if (sched) { vadd(d, s.scheduled, req, -1); s.numScheduledJobs--; }
  else { vadd(d, s.evicted, req, +1); s.numEvictedJobs++; }
  vadd(d, s.allocated, req, -1);
  d.jobFlags[job] = f;
  return sched;
}
DEV void sctxEvictGang(Dev& d, int ref) {  // scheduling.go:436-449
  bool all = true;
  int n = gcCount(d, ref);
  for (int k = 0; k < n; k++) { bool s = sctxEvictJob(d, gcJob(d, ref, k)); all = all && s; }
  if (all) d.rs->numScheduledGangs--;
}

// ------------------------------------------------------------------------------------------------
// constraints (is/scheduling/constraints/constraints.go:113-178)
DEV bool vexceeds(Dev& d, const int64_t* a, const int64_t* b) { for (int r = 0; r < d.cfg.R; r++) if (a[r] > b[r]) return true; return false; }
DEV int checkRound(Dev& d) { return vexceeds(d, d.rs->scheduled, d.cfg.maxToSchedule) ? ASCHED_REASON_MAX_RESOURCES_SCHEDULED : 0; }
DEV int checkJob(Dev& d, int ref) {
  int q = gcQueue(d, ref), card = gcCount(d, ref);
  if (d.qCordoned[q]) return ASCHED_REASON_QUEUE_CORDONED;
  double tokens = d.rs->globalTokens;
  if (tokens < 1) return ASCHED_REASON_GLOBAL_RATE_LIMIT;
  if (d.rs->globalBurst < card) return ASCHED_REASON_GANG_EXCEEDS_GLOBAL_BURST;
  if (tokens < (double)card) return ASCHED_REASON_GLOBAL_RATE_LIMIT_BY_GANG;
  tokens = d.qTokens[q];
  if (tokens < 1) return ASCHED_REASON_QUEUE_RATE_LIMIT;
  if (d.qBurst[q] < card) return ASCHED_REASON_GANG_EXCEEDS_QUEUE_BURST;
  if (tokens < (double)card) return ASCHED_REASON_QUEUE_RATE_LIMIT_BY_GANG;
  // soft time budgets (constraints.go:159-169)
  if (d.cfg.maxNewJobNs > 0 && d.rs->totalNewJobNs > d.cfg.maxNewJobNs) return ASCHED_REASON_GLOBAL_NEW_JOB_DURATION;
  if (d.cfg.maxNewJobPerQueueNs > 0 && d.qNewJobNs && d.qNewJobNs[q] > d.cfg.maxNewJobPerQueueNs) return ASCHED_REASON_QUEUE_NEW_JOB_DURATION;
  int pc = d.jPc[gcJob(d, ref, 0)];
  if (d.hasPcLimit && vexceeds(d, QPV(d.qAllocByPc, q, pc), QPV(d.qPcLimit, q, pc))) return ASCHED_REASON_RESOURCE_LIMIT_EXCEEDED;
  return 0;
}
// sctx.IsWithinFloatingResourceLimits (context/scheduling.go:574-597) + FloatingResourceTypes.WithinLimits (floating_resource_types.go:60-72)
DEV int checkFloating(Dev& d, int ref) {
  const DevCfg& c = d.cfg;
  if (!c.anyFloating) return 0;
  bool requests = false;  // gctx.RequestsFloatingResources (context/gang.go:27-32)
  int cnt = gcCount(d, ref);
  for (int k = 0; k < cnt && !requests; k++) { const int64_t* rq = JREQ(d, gcJob(d, ref, k)); for (int r = 0; r < c.R; r++) if (c.isFloating[r] && rq[r] != 0) requests = true; }
  if (!requests) return 0;
  if (d.jAway) for (int k = 0; k < cnt; k++) if (d.jAway[gcJob(d, ref, k)]) return 0;   // a gang from another pool passed this check there (context/scheduling.go:583-594)
  if (!c.floatingConfigured) return ASCHED_REASON_FLOATING_NOT_CONFIGURED;   // available.AllZero()
  for (int r = 0; r < c.R; r++) if (c.isFloating[r] && d.rs->allocated[r] > c.floatingLimit[r]) return ASCHED_REASON_FLOATING_EXCEEDED;
  return 0;
}
DEV bool isTerminal(int r) { return r == ASCHED_REASON_MAX_RESOURCES_SCHEDULED || r == ASCHED_REASON_GLOBAL_RATE_LIMIT || r == ASCHED_REASON_GLOBAL_NEW_JOB_DURATION; }
DEV bool isQueueTerminal(int r) { return r == ASCHED_REASON_QUEUE_RATE_LIMIT || r == ASCHED_REASON_QUEUE_CORDONED || r == ASCHED_REASON_QUEUE_NEW_JOB_DURATION; }
DEV bool isPropertyOfGang(int r) { return r == ASCHED_REASON_GANG_EXCEEDS_GLOBAL_BURST || r == ASCHED_REASON_JOB_DOES_NOT_FIT || r == ASCHED_REASON_GANG_DOES_NOT_FIT; }
DEV void reserveN(double* tokens, int64_t burst, int rateInf, int n) { if (rateInf) return; if (n > burst) return; *tokens -= (double)n; }

// ------------------------------------------------------------------------------------------------
// node selection
struct Ctl {
  Txn txn;
  int skipKeyCheck;
  int compareSchedPrio, preferLarge, useReplayAlloc, onlyEvicted;
  int* preList;      // staged preemptions of the current gang attempt (job ids), in global memory
  int preCount;
  int fairStamp;
  int fastEnabled;     // this launch may run fast iterations (host conditions hold, no NodeDb-level API calls since prepare)
  int fastEvStatic;    // evicted jobs of the current pass are phase-1 evictions: node / priority are the job's static run
  int l1Dirty;         // fire-and-forget atomics outstanding: plain loads of alloc/keys need an L1 invalidate first
  int fqLive;          // the LDS copy of the per-queue state (round_fast.h FastQueues) is the authoritative one
  int skipEnter;       // the next fast run may fold the gang-free evicted streams out of the loop (round_fast.h "skip mode")
  int skipActive;
  int cancelSeen;      // a fast run saw the cancel word
  int fpLimitHit;      // queueSchedule has seen the fair-share preemption rate limit in this pass (it is acted on once: queue_scheduler.go:114-121)
  int streamNextAt, streamBackoff;   // stream runs (round_fast.h): not before this many fast iterations; doubled after a run too short to pay for its preparation
  int streamCap;                     // stream entries prepared per queue: follows what the last run consumed (a short run must not be followed by a long preparation)
};
// Less (queue_scheduler.go:738-798) as a lexicographic key (A, X, Y, then the queue-name rank); exact for finite, non-negative costs
struct PackedKey { uint32_t A; uint64_t X, Y; };
DEV PackedKey packKey3(int preferLarge, int32_t prio, double proposed, double current, double size, double budget) {
  PackedKey o;
  o.A = ~((uint32_t)prio ^ 0x80000000u);  // higher priority first
  if (preferLarge) {
    if (proposed <= budget) { o.X = __builtin_bit_cast(uint64_t, current); o.Y = ~__builtin_bit_cast(uint64_t, size); }  // under budget: lower current cost, then larger item
    else { o.X = __builtin_bit_cast(uint64_t, proposed) | (1ull << 63); o.Y = 0; }                                     // over budget: after every under-budget item, lower proposed cost
  } else { o.X = __builtin_bit_cast(uint64_t, proposed); o.Y = 0; }
  return o;
}
DEV bool packedLess(const PackedKey& a, uint32_t an, const PackedKey& b, uint32_t bn) {
  return a.A != b.A ? a.A < b.A : a.X != b.X ? a.X < b.X : a.Y != b.Y ? a.Y < b.Y : an < bn;
}

// static mask of the job's scheduling-key shape; during an away attempt the row with the away node type's tolerations added
DEV const uint64_t* shapeMaskOf(Dev& d, int job) { int row = d.rs->awayRowPlus1 ? d.rs->awayRowPlus1 - 1 : d.jShape[job]; return d.shapeMask + (size_t)row * d.cfg.W; }
DEV const uint64_t* uniMask(Dev& d, int job) { int v = d.jcUniValue[job]; return v >= 0 ? d.labelMask + (size_t)v * d.cfg.W : (const uint64_t*)0; }

// selectNodeForPodAtPriority + selectNodeForPodWithItAtPriority (nodedb.go:840-928): first node, in index order, passing
// static + dynamic checks.  Under the alignment conditions checked at upload (DESIGN.md "Exactness conditions") the merged
// iterator order of nodeiteration.go equals the order of the packed key, so this is one argmin scan.
// ---- literal node iteration.  When a request is not a multiple of the index resolution the skip-scan of NodeTypeIterator
// (nodeiteration.go:318-382) can jump over fitting nodes, and when a class matches several node types whose allocatable is not
// resolution-aligned the heap merge of NodeTypesIterator (:74-185, raw quantities then node id) is not the packed-key order.
// For such mask rows the iterators are restated step by step; the only data-parallel piece is "next node of this type at or
// after a key" (one plane pass through wgFirstFit with noFit).
DEV int64_t litCeilDiv(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && a > 0) ? q + 1 : q; }  // b > 0
// packed form of memdb LowerBound(NodeIndexKey(type, b)) within one node type: first packed key whose raw quantity tuple is >= b
DEV uint64_t litBound(const DevCfg& c, const int64_t* b) {
  // (a key field occupies keyWidth + keyGuard bits: the guard bit above it is zero in every key, so a bound that carries into one is still
  //  "after every key with the smaller prefix and before every key with the next one")
  uint64_t acc = 0; int bits = 0; bool stop = false;
  for (int i = 0; i < c.K; i++) {
    int w = c.keyWidth[i], ws = w + c.keyGuard;
    if (stop) { acc <<= ws; bits += ws; continue; }
    int64_t res = c.indexedRes[i];
    bool aligned = b[i] % res == 0;
    int64_t f = (aligned ? b[i] / res : litCeilDiv(b[i], res)) - c.keyLo[i];
    if (f < 0) { f = 0; stop = true; }                       // every node is above b at this column: later columns do not matter
    else if (w < 63 && f >= ((int64_t)1 << w)) {             // no node reaches b here with this prefix: the prefix has to be larger
      if (bits == 0) return ~0ull;
      acc += 1;
      if (bits < 64 && acc >= (1ull << bits)) return ~0ull;
      f = 0; stop = true;
    } else if (!aligned) stop = true;                        // a key column (multiple of the resolution) is never equal to an unaligned b
    acc = (acc << ws) | (uint64_t)f; bits += ws;
  }
  return acc << c.idxBits;
}
#ifdef ASCHED_TWO_WORD_KEYS
// litBound for a two-word key (packKey2's layout: the fields from bit keyShift[K - 1] up, no guard bits)
DEV Key2 litBound2(const DevCfg& c, const int64_t* b) {
  Key2 acc = 0; int bits = 0; bool stop = false;
  for (int i = 0; i < c.K; i++) {
    int w = c.keyWidth[i];
    if (stop) { acc <<= w; bits += w; continue; }
    int64_t res = c.indexedRes[i];
    bool aligned = b[i] % res == 0;
    int64_t f = (aligned ? b[i] / res : litCeilDiv(b[i], res)) - c.keyLo[i];
    if (f < 0) { f = 0; stop = true; }
    else if (w < 63 && f >= ((int64_t)1 << w)) {
      if (bits == 0) return KEY2_NONE;
      acc += 1;
      if (acc >= ((Key2)1 << bits)) return KEY2_NONE;
      f = 0; stop = true;
    } else if (!aligned) stop = true;
    acc = (acc << w) | (Key2)(uint64_t)f; bits += w;
  }
  return acc << c.keyShift[c.K - 1];
}
DEV void litSetBound(const DevCfg& c, LitIt& it) {
  if (WIDE_KEYS(c)) { Key2 b = litBound2(c, it.lb); it.bound = (uint64_t)(b >> 64); it.boundLo = (uint64_t)b; }
  else it.bound = litBound(c, it.lb);
}
#define LIT_SET_BOUND(c, it) litSetBound(c, it)
#else
#define LIT_SET_BOUND(c, it) ((it).bound = litBound(c, (it).lb))
#endif
DEV bool litLbLess(const DevCfg& c, const int64_t* a, const int64_t* b) {  // bytes.Compare(it.key, it.newKey) == -1 (:364)
  for (int i = 0; i < c.K; i++) { if (a[i] < b[i]) return true; if (a[i] > b[i]) return false; }
  return false;
}
// NodeTypeIterator.NextNode (:318-382)
#ifdef ASCHED_TWO_WORD_KEYS
DEV void litAdvance2(Dev& d, int level, LitIt& it, const int64_t* ireq) {   // litAdvance below, on a two-word key
  const DevCfg& c = d.cfg;
  for (;;) {
    if (it.bound == ~0ull && it.boundLo == ~0ull) { it.head = -1; return; }
    ScanArgs a;
    for (int r = 0; r < MAXR; r++) a.req[r] = 0;
    a.maskA = d.typeMask + (size_t)it.type * c.W; a.maskB = nullptr; a.level = level; a.noFit = 1; a.lowBound = it.bound; a.lowBoundLo = it.boundLo; a.levelHi = 0; a.pad = 0;
    int n = wgFirstFit(d, a);
    if (n < 0) { it.head = -1; return; }
    Key2 key = key2Of(KEY(d, level, n), KEYLO(d, level, n));
    int64_t nlb[MAXK];
    bool yielded = false, sought = false;
    for (int i = 0; i < c.K; i++) {
      int64_t nodeQ = AL(d, level, c.indexedCol[i], n);
      nlb[i] = (nodeQ / c.indexedRes[i]) * c.indexedRes[i];
      if (nodeQ < ireq[i]) {
        for (int j = i; j < c.K; j++) nlb[j] = ireq[j];
        if (litLbLess(c, it.lb, nlb)) { for (int j = 0; j < c.K; j++) it.lb[j] = nlb[j]; litSetBound(c, it); sought = true; }
        break;
      } else if (i == c.K - 1) yielded = true;
    }
    if (!sought) { Key2 nx = key == KEY2_NONE ? KEY2_NONE : key + 1; it.bound = (uint64_t)(nx >> 64); it.boundLo = (uint64_t)nx; }
    if (yielded) { it.head = n; return; }
  }
}
#endif
DEV void litAdvance(Dev& d, int level, LitIt& it, const int64_t* ireq) {
  const DevCfg& c = d.cfg;
#ifdef ASCHED_TWO_WORD_KEYS
  if (WIDE_KEYS(c)) { litAdvance2(d, level, it, ireq); return; }
#endif
  for (;;) {
    if (it.bound == ~0ull) { it.head = -1; return; }
    ScanArgs a;
    for (int r = 0; r < MAXR; r++) a.req[r] = 0;
    a.maskA = d.typeMask + (size_t)it.type * c.W; a.maskB = nullptr; a.level = level; a.noFit = 1; a.lowBound = it.bound; a.levelHi = 0; a.pad = 0;
    int n = wgFirstFit(d, a);
    if (n < 0) { it.head = -1; return; }
    uint64_t key = KEY(d, level, n);
    int64_t nlb[MAXK];
    bool yielded = false, sought = false;
    for (int i = 0; i < c.K; i++) {
      int64_t nodeQ = AL(d, level, c.indexedCol[i], n);
      nlb[i] = (nodeQ / c.indexedRes[i]) * c.indexedRes[i];  // roundQuantityToResolution (encoding.go:56-58)
      if (nodeQ < ireq[i]) {
        for (int j = i; j < c.K; j++) nlb[j] = ireq[j];
        if (litLbLess(c, it.lb, nlb)) { for (int j = 0; j < c.K; j++) it.lb[j] = nlb[j]; it.bound = litBound(c, nlb); sought = true; }
        break;  // else: "new lower-bound is not greater than current bound" (:371-376): keep scanning linearly
      } else if (i == c.K - 1) yielded = true;
    }
    if (!sought) it.bound = key == ~0ull ? ~0ull : key + 1;  // memdbIterator.Next(): strictly after this node
    if (yielded) { it.head = n; return; }
  }
}
DEV bool litNodeLess(Dev& d, int level, int a, int b) {  // nodeTypesIteratorPQ.less (:170-185)
  for (int i = 0; i < d.cfg.K; i++) {
    int64_t qa = AL(d, level, d.cfg.indexedCol[i], a), qb = AL(d, level, d.cfg.indexedCol[i], b);
    if (qa < qb) return true;
    if (qa > qb) return false;
  }
  return d.nodeIdRank[a] < d.nodeIdRank[b];
}
// selectNodeForPodAtPriority + selectNodeForPodWithItAtPriority (nodedb.go:840-928) over the literal iterators
DEV_COLD int selectAtLevelLiteral(Dev& d, int job, int32_t prio, int level, int row) {
  const DevCfg& c = d.cfg;
  const int64_t* req = JREQ(d, job);
  int64_t ireq[MAXK];
  for (int i = 0; i < c.K; i++) ireq[i] = req[c.indexedCol[i]];
  int t0 = d.rowTypeOff[row], nT = d.rowTypeOff[row + 1] - t0;
  if (nT > LIT_TMAX) { raise(d, ASCHED_ERR_UNSUPPORTED, 510); return -1; }
  for (int k = 0; k < nT; k++) {  // NewNodeTypesIterator (:84-123)
    LitIt& it = d.lit[k];
    it.type = d.rowTypes[t0 + k];
    for (int i = 0; i < MAXK; i++) it.lb[i] = i < c.K ? ireq[i] : 0;
    LIT_SET_BOUND(c, it);
    litAdvance(d, level, it, ireq);
  }
  const uint64_t* mA = shapeMaskOf(d, job);
  const uint64_t* mB = uniMask(d, job);
  for (;;) {
    int best = -1;
    for (int k = 0; k < nT; k++) if (d.lit[k].head >= 0 && (best < 0 || litNodeLess(d, level, d.lit[k].head, d.lit[best].head))) best = k;
    if (best < 0) return -1;
    int n = d.lit[best].head;
    litAdvance(d, level, d.lit[best], ireq);  // NextNode (:134-149) advances the popped iterator before returning the node
    bool st = (mA[n >> 6] >> (n & 63)) & 1;
    if (st && mB) st = (mB[n >> 6] >> (n & 63)) & 1;
    if (st && fitsAlloc(d, req, level, n)) { d.pcNode[job] = n; d.pcPap[job] = prio; return n; }
  }
}

DEV int selectAtLevel(Dev& d, int job, int32_t prio) {
  d.rs->numNodeQueries++;
  int level = levelOf(d.cfg, prio);
  if (level < 0) { raise(d, ASCHED_ERR_INTERNAL, 500); return -1; }
  int row = d.rs->awayRowPlus1 ? d.rs->awayRowPlus1 - 1 : d.jShape[job];
  if (d.rowLiteral && d.rowLiteral[row]) return selectAtLevelLiteral(d, job, prio, level, row);
  if (level == 0 && d.jcUniValue[job] < 0 && !d.rs->awayRowPlus1) {
    int fn = fastSelectLevel0(d, job);
    if (fn != -2) { if (fn >= 0) { d.pcNode[job] = fn; d.pcPap[job] = prio; } return fn; }
  }
  ScanArgs a;
  const int64_t* req = JREQ(d, job);
  for (int r = 0; r < MAXR; r++) a.req[r] = r < d.cfg.R ? req[r] : 0;
  a.maskA = shapeMaskOf(d, job);
  a.maskB = uniMask(d, job);
  a.level = level; a.noFit = 0; a.lowBound = 0; a.levelHi = 0; a.pad = 0; SCAN_NO_LOW_WORD(a);
  long long t0 = CLK();
  if (d.progress) { d.progress[2] = 1; d.progress[3]++; }
  int n = wgFirstFit(d, a);
  if (d.progress) d.progress[2] = 0;
  d.rs->statClk[6] += CLK() - t0;
  if (n >= 0) { d.pcNode[job] = n; d.pcPap[job] = prio; }
  return n;
}

// selectNodeForJobWithFairPreemption (nodedb.go:935-1043).  The reference walks the evicted jobs by descending Index, adds
// each one's request to a per-node running total (starting from the node's allocatable at the evicted priority) and
// returns the node of the first entry at which the total covers the request and the static requirements hold; a node whose
// static check fails once is skipped from then on (:1000-1004).  Entries of different nodes never interact, so that walk
// is the maximum, over statically matching nodes, of "the Index at which this node's own entries first cover the request"
// — fairNodeBest() for one node, evaluated for all nodes at once over a per-node index of the table (ensureFairIndex).
DEV int fairNodeBest(const Dev& d, const FairArgs& a, int n, int floorIdx) {
  const DevCfg& cf = d.cfg;
  int k0 = d.fairOff[n], k1 = d.fairOff[n + 1];
  if (k0 == k1) return -1;
  uint64_t w = a.maskA[n >> 6];
  if (a.maskB) w &= a.maskB[n >> 6];
  if (!((w >> (n & 63)) & 1)) return -1;
  int64_t av[MAXR];
  for (int r = 0; r < MAXR; r++) av[r] = r < cf.R ? AL(d, cf.evLevel, r, n) : 0;
  for (int k = k0; k < k1; k++) {
    int idx = d.fairEnt[k];
    if (idx <= floorIdx) return -1;  // descending: cannot beat what the caller already has
    if (!d.evTabAlive[idx]) continue;
    int ej = d.fairEntJob[k];
    int32_t ep = d.schedAtPrio[ej];
    if (ep == NO_PRIORITY) return FAIR_BAD_ENTRY;
    if (ep > a.prio) continue;
    const int64_t* er = JREQ(d, ej);
    bool fits = true;
    for (int r = 0; r < MAXR; r++) if (r < cf.R) { av[r] += er[r]; if (a.req[r] > av[r]) fits = false; }
    if (fits) return idx;
  }
  return -1;
}
DEV_COLD void ensureFairIndex(Dev& d);
DEV_COLD int fairApply(Dev& d, Ctl& c, int job, int idx, int32_t jobPrio);
DEV_COLD int selectWithFairPreemption(Dev& d, Ctl& c, int job) {
  ensureReplay(d, c);
  long long t0 = CLK();
  if (d.progress) { d.progress[2] = 2; d.progress[3]++; }
  ensureFairIndex(d);
  if (d.progress) d.progress[2] = 3;
  const DevCfg& cf = d.cfg;
  FairArgs a;
  const int64_t* req = JREQ(d, job);
  for (int r = 0; r < MAXR; r++) a.req[r] = r < cf.R ? req[r] : 0;
  a.maskA = shapeMaskOf(d, job);
  a.maskB = uniMask(d, job);
  a.prio = d.pcSap[job]; a.pad = 0;
  int idx = wgFairSelect(d, a);
  if (d.progress) d.progress[2] = 0;
  d.rs->statClk[7] += CLK() - t0;
  XSEG(12);
  return fairApply(d, c, job, idx, a.prio);
}
// the node of evicted-table entry idx wins: its considered entries are preempted (nodedb.go:1012-1023)
DEV_COLD int fairApply(Dev& d, Ctl& c, int job, int idx, int32_t jobPrio) {
  struct { int32_t prio; } a; a.prio = jobPrio;
  if (idx < 0) return -1;
  if (idx >= d.rs->evictedTableSize) { raise(d, ASCHED_ERR_INTERNAL, 600); return -1; }
  int n = d.jcAssigned[d.evTabJob[idx]];
  if (n < 0) { raise(d, ASCHED_ERR_INTERNAL, 601); return -1; }
  XSEG(13);
  // preempt every considered evicted job of node n (Index >= idx, alive, priority <= ours), in scan order (:1012-1023)
  int32_t maxPriority = ASCHED_MIN_PRIORITY;
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  if (CTL_WAVE()) {   // the node's entries one per lane (index, alive, job, priority: one round of loads), then the victims in scan order
    int k0 = d.fairOff[n], k1 = d.fairOff[n + 1], lane = CTL_LANE();
    for (int base = k0; base < k1; base += 64) {
      int k = base + lane;
      bool in = k < k1;
      int i2 = in ? d.fairEnt[k] : -1, e2 = -1;
      int32_t p = 0;
      bool stop = in && i2 < idx, take = false;
      if (in && !stop && d.evTabAlive[i2]) { e2 = d.fairEntJob[k]; p = d.schedAtPrio[e2]; take = p <= a.prio; }
      unsigned long long stopMask = __ballot(stop), takeMask = __ballot(take);
      if (stopMask) takeMask &= (stopMask & (0ull - stopMask)) - 1;   // (descending index order: nothing after the first entry below idx)
      while (takeMask) {
        int l = __builtin_ctzll(takeMask);
        takeMask &= takeMask - 1;
        int ti = __builtin_amdgcn_readlane(i2, l), te = __builtin_amdgcn_readlane(e2, l);
        int32_t tp = __builtin_amdgcn_readlane(p, l);
        evTabDelete(d, ti, true);
        if (tp > maxPriority) maxPriority = tp;
        c.preList[c.preCount++] = te;
      }
      if (stopMask) break;
    }
  } else
#endif
  for (int k = d.fairOff[n]; k < d.fairOff[n + 1]; k++) {
    int i2 = d.fairEnt[k];
    if (i2 < idx) break;
    if (!d.evTabAlive[i2]) continue;
    int e2 = d.fairEntJob[k];
    int32_t p = d.schedAtPrio[e2];
    if (p > a.prio) continue;
    evTabDelete(d, i2, true);
    if (p > maxPriority) maxPriority = p;
    c.preList[c.preCount++] = e2;
  }
  d.pcNode[job] = n; d.pcPap[job] = maxPriority;
  return n;
}

// selectNodeForJobWithTxnAtPriority (nodedb.go:724-789)
//
// The feasibility gate (fit at the job's priority) and the urgency sweep (:805-838: fit at -1, 0, ... up to the job's priority, first level with a
// node wins) look at the same planes.  A bind subtracts a job's requests from every level up to its cutoff, so with non-negative requests and
// no hand-written AllocatableByPriority a node's allocatable is non-decreasing in the level: a node that fits at level l fits at every level
// above.  Hence "first level with any fitting node, then the minimum key at that level" == the minimum over nodes of (lowest level the node
// fits at, its key at that level), and the gate is "is there any such node".  ONE multi-level pass (ScanArgs.levelHi) answers both; the
// fair-share attempt in between changes nothing when it fails.  Query counts are kept as the level-by-level loop would have issued them.
DEV_COLD int selectAtPriority(Dev& d, Ctl& c, int job) {
  XSEG(31);
  int n = selectAtLevel(d, job, ASCHED_EVICTED_PRIORITY);
  XSEG(32);
  if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_NO_PREEMPTION; return n; }
  int row = d.rs->awayRowPlus1 ? d.rs->awayRowPlus1 - 1 : d.jShape[job];
  int32_t sap = d.pcSap[job];
  int lp = levelOf(d.cfg, sap);
  if (d.f.cascadeFuse && lp >= 1 && !(d.rowLiteral && d.rowLiteral[row])) {
    ScanArgs a;
    const int64_t* req = JREQ(d, job);
    for (int r = 0; r < MAXR; r++) a.req[r] = r < d.cfg.R ? req[r] : 0;
    a.maskA = shapeMaskOf(d, job); a.maskB = uniMask(d, job);
    a.level = 1; a.levelHi = lp; a.noFit = 0; a.lowBound = 0; a.pad = 0; SCAN_NO_LOW_WORD(a);
    if (lp == 1) { a.level = 1; a.levelHi = 0; }   // one level: the plain pass
    if (d.cfg.disableUrgency) { a.level = lp; a.levelHi = 0; }   // no sweep to fuse with: the gate alone, at the job's level — its node is what a failed attempt's record needs (exclRecordWide)
    long long t0 = CLK();
    uint64_t best;
    if (!d.cfg.disableFair && !d.rs->replayPending) {
      // the gate and the per-node evaluation of fair-share preemption read the same nodes and neither depends on the other's answer: ONE wide pass,
      // one hand-shake with the helper workgroups.  The per-node index of the evicted table is due before it (a pure function of the table); while
      // the replay of the evicted jobs is still deferred the two questions are asked one after the other as before — the replay runs loops of its
      // own and belongs where the reference runs it: after a gate that passed.
      ensureFairIndex(d);
      // The threshold table (round_ft.h) answers the fair-share question for a home attempt of a queued job without a pass over the nodes.  A node found
      // that way also passes the gate — its considered entries are evicted jobs, which no level above -2 counts: alloc[level] >= alloc[-2] + their requests
      // >= the request — so the gate (counted as issued) needs no scan.  No node: the gate and the urgency sweep take the multi-level pass as before.
#ifndef ASCHED_NO_FT
      if (d.ftT && d.rs->ftValid && !d.rs->awayRowPlus1 && !a.maskB && d.jShape[job] < d.ftS && d.ftPrio[d.jShape[job]] == sap) {
        long long t1 = CLK();
        int idx = ftQuery(d, d.jShape[job]);
        d.rs->statClk[7] += CLK() - t1;
        XSEG(33);
        if (idx >= 0) {
          d.rs->statClk[6] += CLK() - t0;
          d.rs->numNodeQueries++;                      // the gate
          d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
          n = fairApply(d, c, job, idx, sap);
          XSEG(34);
          if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_FAIRSHARE; return n; }
          return -1;                                   // (fairApply raised an error)
        }
        if (idx == -1) {   // no node can be freed by fair-share preemption: gate + urgency sweep in one multi-level scan
          best = wgFirstFitKey(d, a);
          d.rs->statClk[6] += CLK() - t0;
          XSEG(23);
          d.rs->numNodeQueries++;                      // the gate
          if (best == ~0ull) return -1;
          d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
          if (d.cfg.disableUrgency) return -2 - d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];
          int l = lp == 1 ? 1 : (int)(best >> SCAN_LEVEL_SHIFT);
          d.rs->numNodeQueries += l;                     // levels 1 .. l of the sweep
          n = d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];
          d.pcNode[job] = n; d.pcPap[job] = d.cfg.prios[l]; d.pcMethod[job] = ASCHED_METHOD_URGENCY;
          return n;
        }
        // idx == -2: the table was dropped (inconsistent maxima): the wide pass below answers
      }
#endif
      FairArgs fa;
      for (int r = 0; r < MAXR; r++) fa.req[r] = a.req[r];
      fa.maskA = a.maskA; fa.maskB = a.maskB; fa.prio = sap; fa.pad = 0;
      long long t1 = CLK();
      int idx = wgScanFair(d, a, fa, &best);
      d.rs->statClk[7] += CLK() - t1;   // (the pass itself; [6]: with the index check and the argument block)
      d.rs->statClk[6] += CLK() - t0;
      d.rs->numNodeQueries++;                        // the gate
      XSEG(33);
      if (best == ~0ull) return -1;
      d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
      n = fairApply(d, c, job, idx, fa.prio);
      XSEG(34);
      if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_FAIRSHARE; return n; }
    } else {
      best = wgFirstFitKey(d, a);
      d.rs->statClk[6] += CLK() - t0;
      d.rs->numNodeQueries++;                        // the gate
      if (best == ~0ull) return -1;
      d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
      if (!d.cfg.disableFair) {
        n = selectWithFairPreemption(d, c, job);
        if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_FAIRSHARE; return n; }
      }
    }
    d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
    if (d.cfg.disableUrgency) return -2 - d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];   // (<= -2: the gate had found node -2 - n — selectNodeForJob's record of the failure needs it)
    int l = lp == 1 ? 1 : (int)(best >> SCAN_LEVEL_SHIFT);
    d.rs->numNodeQueries += l;                       // levels 1 .. l of the sweep
    n = d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];
    d.pcNode[job] = n; d.pcPap[job] = d.cfg.prios[l]; d.pcMethod[job] = ASCHED_METHOD_URGENCY;
    return n;
  }
  n = selectAtLevel(d, job, d.pcSap[job]);  // feasibility gate
  if (n < 0) return -1;
  const int gateNode = n;
  d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
  if (!d.cfg.disableFair) {
    n = selectWithFairPreemption(d, c, job);
    if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_FAIRSHARE; return n; }
  }
  d.pcNode[job] = -1; d.pcPap[job] = ASCHED_MIN_PRIORITY;
  if (!d.cfg.disableUrgency) {  // selectNodeForJobWithUrgencyPreemption :805-838
    for (int l = 0; l < d.cfg.P; l++) {
      int32_t pr = d.cfg.prios[l];
      if (pr == ASCHED_EVICTED_PRIORITY) continue;
      if (pr > d.pcSap[job]) break;
      n = selectAtLevel(d, job, pr);
      if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_URGENCY; return n; }
    }
  }
  return -2 - gateNode;
}

// SelectNodeForJobWithTxn (nodedb.go:538-630)
DEV int selectNodeForJob(Dev& d, Ctl& c, int job) {
  if (d.jcPreempted[job]) return -1;  // :541-544 (GetPreemptingJob != nil; PreemptionDetails are set by applyPreemptions)
  int32_t prio = d.schedAtPrio[job];
  if (prio == NO_PRIORITY) prio = d.cfg.pcPriority[d.jPc[job]];
  d.jcHasPctx[job] = 1;
  d.pcNode[job] = -1; d.pcSap[job] = prio; d.pcPap[job] = ASCHED_MIN_PRIORITY; d.pcMethod[job] = ASCHED_METHOD_NONE;
  int pinned = d.jcAssigned[job];
  if (pinned >= 0) {  // :583-594 — evicted jobs may only return to their node; dynamic check only (:897-906)
    int level = levelOf(d.cfg, prio);
    if (level < 0) { raise(d, ASCHED_ERR_INTERNAL, 501); return -1; }
    bool ok = (d.nodeFlags[pinned] & 1) || fitsAlloc(d, JREQ(d, job), level, pinned);
    d.pcMethod[job] = ASCHED_METHOD_RESCHEDULED;
    if (ok) { d.pcNode[job] = pinned; d.pcPap[job] = prio; exclForget(d, job); return pinned; }
    if (d.excl) exclPinned(d, job, pinned, level);
    return -1;
  }
  const int64_t* req = JREQ(d, job);
  for (int r = 0; r < d.cfg.R; r++) if (d.cfg.disallowed[r] && req[r] > 0) { if (d.excl) d.excl[job] = EXCL_S_DISALLOWED; return -1; }  // :596-601
  bool recorded = false;
  if (!d.cfg.disableHome) {
    int n = selectAtPriority(d, c, job);
    if (n >= 0) { exclForget(d, job); return n; }
    if (d.excl && !d.rs->error) { exclRecordWide(d, job, levelOf(d.cfg, d.pcSap[job]), n <= -2 ? -2 - n : -1); recorded = true; }
  }
  if (d.cfg.hasAway) {
    bool awayDisabled = d.cfg.disableAway || (d.jGang[job] >= 0 && d.cfg.disableGangAway);
    if (!awayDisabled) {
      int pc = d.jPc[job], s = d.jShape[job];
      for (int k = d.cfg.pcAwayOff[pc]; k < d.cfg.pcAwayOff[pc + 1]; k++) {  // selectNodeForJobWithTxnAndAwayNodeType :677-722
        if (!d.cfg.awayUsable[k]) continue;  // no extra taints to tolerate (:703-706)
        d.rs->awayRowPlus1 = d.cfg.S + d.awayRowOff[s] + (k - d.cfg.pcAwayOff[pc]) + 1;
        d.pcSap[job] = d.cfg.awayPrio[k];      // :719 (stays at the last away priority when every attempt fails)
        int n = selectAtPriority(d, c, job);
        if (n < 0 && d.excl && !d.rs->error) { exclRecordWide(d, job, levelOf(d.cfg, d.pcSap[job]), n <= -2 ? -2 - n : -1); recorded = true; }   // (this attempt's row: the last one stays, nodedb.go:736,748)
        d.rs->awayRowPlus1 = 0;
        if (d.rs->error) return -1;
        if (n >= 0) { d.pcMethod[job] = ASCHED_METHOD_AWAY; exclForget(d, job); return n; }
      }
    }
  }
  if (d.excl && !recorded) d.excl[job] = EXCL_S_NO_ATTEMPT;   // (no attempt at all — home scheduling disabled, no usable away type — leaves every node implicit, nodedb.go:566-580)
  return -1;
}

// preemptSiblingGangJobs (nodedb.go:468-525)
DEV void preemptSiblings(Dev& d, Ctl& c, int firstPre, int lastPre) {
  if (firstPre < lastPre) ensureReplay(d, c);
  for (int i = firstPre; i < lastPre; i++) {
    int job = c.preList[i];
    int g = d.jGang[job];
    if (g < 0) continue;
    for (int k = d.gangOff[g]; k < d.gangOff[g + 1]; k++) {
      int s = d.gangJobs[k];
      int idx = d.evIndexOfJob[s];
      if (idx < 0) continue;
      int n = d.jcAssigned[s];
      if (n < 0) { raise(d, ASCHED_ERR_INTERNAL, 700); return; }
      removeJob(d, n, s, true);
      updateKeysCtl(d, n);
      evTabDelete(d, idx, true);
      c.preList[c.preCount++] = s;
    }
  }
}

// ScheduleManyWithTxn (nodedb.go:417-462)
DEV_COLD bool scheduleMany(Dev& d, Ctl& c, int ref) {
  int cnt = gcCount(d, ref);
  for (int k = 0; k < cnt; k++) {
    int job = gcJob(d, ref, k);
    d.jcReason[job] = 0;
    // the common member: a queued job that fits without preemption.  SelectNodeForJobWithTxn's first step and BindJobToNode through the
    // level-0 fast structure (lane-parallel atomics + the undo record an abort needs); anything else — no fit at priority -2, pinned,
    // uniformity selector, away attempt — falls through to the generic cascade on the state this leaves
    if (cnt > 1 && fastGangMember(d, c, job)) continue;
    fastFence(c);
    int pre0 = c.preCount;
    XSEG(30);
    int n = selectNodeForJob(d, c, job);
    XSEG(25);
    if (d.rs->error) return false;
    if (n < 0) return false;   // (fenced above: the abort's undo reads what the fast members wrote)
    int pre1 = c.preCount;
    for (int i = pre0; i < pre1; i++) removeJob(d, n, c.preList[i], true);  // victims leave the returned node copy (nodedb.go:1012-1023)
    XSEG(26);
    int32_t prio = bindPriority(d, job, d.pcSap[job]);
    if (addJob(d, n, job, cutoffFor(d, job, prio), true)) return false;     // BindJobToNode :1046-1068
    XSEG(27);
    d.schedAtPrio[job] = prio;                                              // not rolled back on abort (plain Go map)
    if (d.pcMethod[job] != ASCHED_METHOD_NO_PREEMPTION && d.pcMethod[job] != ASCHED_METHOD_RESCHEDULED) d.rs->lvl0NonNeg = 0;  // preemption may overdraw priority -2
    updateKeysCtl(d, n);
    int eidx = d.evIndexOfJob[job];
    if (eidx >= 0) evTabDelete(d, eidx, true);
    preemptSiblings(d, c, pre0, pre1);
    for (int i = pre0; i < c.preCount; i++) d.jcStagedBy[c.preList[i]] = job;
    XSEG(28);
  }
  fastFence(c);  // the fast members' binds are no-return atomics: visible to whatever reads the planes next (commit bookkeeping, abort's undo)
  return true;
}

// applyPreemptions (gang_scheduler.go:268-273) + MarkJobPreempted (scheduling.go:508-516)
DEV void applyPreemptions(Dev& d, Ctl& c) {
  for (int i = 0; i < c.preCount; i++) {
    int p = c.preList[i];
    d.jcStagedBy[p] = -1;
    if (!d.jcPreempted[p]) { d.jcPreempted[p] = 1; d.rs->numPreemptedMarks++; if (d.rs->hasFpLimiter) d.rs->fpTokens -= 1.0; }
  }
  c.preCount = 0;
}
DEV void unstagePreemptions(Dev& d, Ctl& c) { for (int i = 0; i < c.preCount; i++) d.jcStagedBy[c.preList[i]] = -1; c.preCount = 0; }

// tryScheduleGang (gang_scheduler.go:229-262)
DEV bool tryGang(Dev& d, Ctl& c, int ref, int* reason) {
  c.preCount = 0;
  txnBegin(d, c.txn);
  bool ok = scheduleMany(d, c, ref);
  *reason = 0;
  if (!ok) *reason = gcCount(d, ref) > 1 ? ASCHED_REASON_GANG_DOES_NOT_FIT : ASCHED_REASON_JOB_DOES_NOT_FIT;
  if (ok && !d.rs->error) { txnCommit(d, c.txn); applyPreemptions(d, c); }
  else { txnAbort(d, c.txn); unstagePreemptions(d, c); }
  return ok;
}
DEV_COLD bool tryGangCold(Dev& d, Ctl& c, int ref, int* reason) { return tryGang(d, c, ref, reason); }   // (out of line for the fast loop's preempting iteration: A/B builds)
DEV void fitOf(Dev& d, int ref, int* num, double* mean) {  // gctx.Fit (context/gang.go:94-110)
  int n = 0; int32_t tot = 0;
  int cnt = gcCount(d, ref);
  for (int k = 0; k < cnt; k++) { int j = gcJob(d, ref, k); if (!(d.jcHasPctx[j] && d.pcNode[j] >= 0)) continue; n++; tot += d.pcPap[j]; }
  *num = n; *mean = n == 0 ? (double)tot : (double)tot / (double)n;
}
// host-built table of uniformity label values: uniOff[label slot], uniVals[] = labelMask ids in ascending value order
struct UniTable { const int32_t* slotOfLabel; const int32_t* off; int nLabels; };
// trySchedule (gang_scheduler.go:150-227).  jGangUni[job] holds the label *slot* (-1 none, -2 label not indexed).
DEV_COLD bool trySchedule(Dev& d, Ctl& c, int ref, int* reason, const int32_t* uniOff) {
  int j0 = gcJob(d, ref, 0);
  int slot = d.jGang[j0] >= 0 ? d.jGangUni[j0] : -1;
  if (slot == -1) return tryGang(d, c, ref, reason);
  if (slot == -2) { *reason = ASCHED_REASON_UNIFORMITY_LABEL_NOT_INDEXED; return false; }
  int v0 = uniOff[slot], v1 = uniOff[slot + 1];
  if (v1 == v0) { *reason = ASCHED_REASON_NO_NODES_WITH_UNIFORMITY_LABEL; return false; }
  int cnt = gcCount(d, ref);
  bool haveBest = false; int bestValue = -1, bestNum = 0; double bestMean = 0;
  for (int v = v0; v < v1; v++) {  // ascending interned value (the reference iterates a Go map: order unspecified)
    for (int k = 0; k < cnt; k++) d.jcUniValue[gcJob(d, ref, k)] = v;
    c.preCount = 0;
    txnBegin(d, c.txn);
    bool ok = scheduleMany(d, c, ref);
    *reason = 0;
    if (!ok) *reason = cnt > 1 ? ASCHED_REASON_GANG_DOES_NOT_FIT : ASCHED_REASON_JOB_DOES_NOT_FIT;
    if (d.rs->error) { txnAbort(d, c.txn); return false; }
    if (ok) {
      int num; double mean; fitOf(d, ref, &num, &mean);
      if (num == cnt && mean == (double)ASCHED_MIN_PRIORITY) { txnCommit(d, c.txn); applyPreemptions(d, c); *reason = 0; return true; }
      bool better = !haveBest || (bestNum < num || (bestNum == num && bestMean > mean));
      if (better) {
        if (v == v1 - 1) { txnCommit(d, c.txn); applyPreemptions(d, c); *reason = 0; return true; }
        haveBest = true; bestValue = v; bestNum = num; bestMean = mean;
      }
    }
    txnAbort(d, c.txn); unstagePreemptions(d, c);
  }
  if (!haveBest) { *reason = ASCHED_REASON_GANG_UNIFORMITY_NO_FIT; return false; }
  for (int k = 0; k < cnt; k++) d.jcUniValue[gcJob(d, ref, k)] = bestValue;
  return tryGang(d, c, ref, reason);
}

DEV bool keyValid(Dev& d, int job) { return d.jcAssigned[job] < 0 && d.jcUniValue[job] < 0; }  // jctx.SchedulingKey (context/job.go:104-109)

DEV void failJob(Dev& d, int job, int reason) {  // jctx.Fail (context/job.go:115-121)
  d.jcReason[job] = reason;
  if (d.jcHasPctx[job]) { d.pcNode[job] = -1; d.pcMethod[job] = ASCHED_METHOD_NONE; }
}

// GangScheduler.Schedule incl. deferred bookkeeping (gang_scheduler.go:46-148)
DEV_COLD bool gangSchedule(Dev& d, Ctl& c, int ref, int* reason, const int32_t* uniOff) {
  *reason = 0;
  bool allEv = gcAllEvicted(d, ref);
  if (!allEv) {
    int r = checkRound(d);  // returns BEFORE the deferred bookkeeping is registered (:102-106)
    if (r) { *reason = r; return false; }
  }
  XSEG_BEGIN();
  sctxAddGang(d, ref);
  bool ok = false;
  int r = 0;
  if (!allEv) r = checkJob(d, ref);
  if (!r) r = checkFloating(d, ref);  // gang_scheduler.go:143: for evicted gangs too
  XSEG(24);
  if (r) *reason = r;
  else ok = trySchedule(d, c, ref, reason, uniOff);
  XSEG(29);
  if (d.rs->error) return false;
  int cnt = gcCount(d, ref), q = gcQueue(d, ref);
  if (ok && !allEv) {  // :118-123
    reserveN(&d.rs->globalTokens, d.rs->globalBurst, d.rs->globalRateInf, cnt);
    reserveN(&d.qTokens[q], d.qBurst[q], d.qRateInf[q], cnt);
  }
  if (ok) {  // updateGangSchedulingContextOnSuccess :46-61
    for (int k = 0; k < cnt; k++) { int j = gcJob(d, ref, k); if (d.jcReason[j] != 0) sctxEvictJob(d, j); }
    return true;
  }
  sctxEvictGang(d, ref);  // updateGangSchedulingContextOnFailure :63-98
  for (int k = 0; k < cnt; k++) failJob(d, gcJob(d, ref, k), *reason);
  sctxAddGang(d, ref);
  if (!c.skipKeyCheck && cnt == 1 && isPropertyOfGang(*reason)) {
    int j = gcJob(d, ref, 0);
    if (keyValid(d, j)) {
      int s = d.jShape[j];
      if (!d.unfeasible[s]) { d.unfeasible[s] = 1; d.unfeasibleReason[s] = *reason; d.rs->numUnfeasible++; }
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// job / gang iterators (is/scheduling/jobiteration.go, queue_scheduler.go:306-444)
DEV void resetJctxForQueued(Dev& d, int job) {  // JobSchedulingContextFromJob (context/job.go:149-158)
  d.jcEvicted[job] = 0; d.jcAssigned[job] = -1; d.jcHasPctx[job] = 0;
  if (!d.rs->optMode) d.jcReason[job] = 0;   // (optimiser: jcReason is also the REPORT of an earlier failed attempt, kept until the new context is added: dev.h optMode)
  d.jcGangCard[job] = d.jGang[job] >= 0 ? d.jGangCard[job] : 1; d.jcUniValue[job] = -1; d.jcStagedBy[job] = -1;
}
DEV int jobItNext(Dev& d, int q, bool withQueued) {  // MultiJobsIterator(evicted, queued) :179-228
#ifdef ASCHED_MARKET_ROUND
  if (mkOn(d) && withQueued) {   // MarketDrivenMultiJobsIterator.Next (jobiteration.go:250-294; pqs.go:730-731: only when there is a jobRepo)
    if (MKD.itV1[q] < 0 && d.itEi[q] < d.evOff[q + 1]) MKD.itV1[q] = d.evList[d.itEi[q]++];                       // InMemoryJobIterator.Next: every entry is an evicted job
    if (MKD.itV2[q] < 0 && !d.itJobOnlyEv[q] && d.itQi[q] < d.queuedOff[q + 1]) { int job = d.queuedJobs[d.itQi[q]++]; resetJctxForQueued(d, job); MKD.itV2[q] = job; }   // QueuedJobsIterator.Next :152-161
    int j1 = MKD.itV1[q], j2 = MKD.itV2[q];
    if (j1 >= 0 && j2 >= 0) {
      if (MKD.jRank[j1] < MKD.jRank[j2]) { MKD.itV1[q] = -1; return j1; }   // MarketSchedulingOrderCompare(j1, j2) < 0: the pool-wide rank under that order (asched_host.inc)
      MKD.itV2[q] = -1; return j2;
    }
    if (j1 >= 0) { MKD.itV1[q] = -1; return j1; }
    if (j2 >= 0) { MKD.itV2[q] = -1; return j2; }
    return -1;
  }
#endif
  if (d.itStage[q] == 0) {
    if (d.itEi[q] < d.evOff[q + 1]) return d.evList[d.itEi[q]++];
    d.itStage[q] = 1;
  }
  if (d.itJobOnlyEv[q] || !withQueued) return -1;
  if (d.itQi[q] < d.queuedOff[q + 1]) { int job = d.queuedJobs[d.itQi[q]++]; resetJctxForQueued(d, job); return job; }
  return -1;
}
DEV void gangItOnlyEvicted(Dev& d, int q) {  // :338-350
  if (d.itGangOnlyEv[q]) return;
  d.itGangOnlyEv[q] = 1; d.itJobOnlyEv[q] = 1;
  MK(if (mkOn(d) && MKD.itV2[q] >= 0 && !d.jcEvicted[MKD.itV2[q]]) MKD.itV2[q] = -1;)   // MarketDrivenMultiJobsIterator.OnlyYieldEvicted :296-309 (it1Value is always an evicted job)
  int nx = d.itNext[q];
  if (nx != -1 && !gcAllEvicted(d, nx)) { d.itStashed[q] = nx; d.itNext[q] = -1; }
}
DEV void gangItResume(Dev& d, int q) {  // :352-361
  if (!d.itGangOnlyEv[q]) return;
  d.itGangOnlyEv[q] = 0; d.itJobOnlyEv[q] = 0; d.itStage[q] = 0;
  d.itNext[q] = d.itStashed[q]; d.itStashed[q] = -1;
}
DEV int gangItPeek(Dev& d, Ctl& c, int q, bool withQueued, uint32_t maxLookback, bool skipKnown) {  // :376-432
  if (d.itNext[q] != -1) return d.itNext[q];
  for (;;) {
    if (maxLookback != 0 && !d.itGangOnlyEv[q] && (uint32_t)d.itJobsSeen[q] >= maxLookback) gangItOnlyEvicted(d, q);
    int job = jobItNext(d, q, withQueued);
    if (job < 0) return -1;
    if (!d.jcEvicted[job]) d.itJobsSeen[q]++;
    if (skipKnown && d.rs->numUnfeasible > 0 && keyValid(d, job)) {  // :398-413
      int s = d.jShape[job];
      if (d.unfeasible[s]) {
        d.jcReason[job] = d.unfeasibleReason[s];
        d.jcHasPctx[job] = 1; d.pcNode[job] = -1; d.pcMethod[job] = ASCHED_METHOD_NONE;
        sctxAddJob(d, job);
        d.jcReason[job] = ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY;
        // A queue of thousands of identical jobs that do not fit (the steady state BenchmarkPreemptingQueueScheduler times, preempting_queue_scheduler_test.go:2561-2799) is
        // skipped job by job here; the jobs behind this one that Peek would skip as well are independent of each other: one lane each (round 4).
        if (d.itStage[q] == 1 && !d.itJobOnlyEv[q] && withQueued && !d.rs->optMode) {
          int max = d.queuedOff[q + 1] - d.itQi[q];
          if (maxLookback != 0 && !d.itGangOnlyEv[q]) { int64_t lim = (int64_t)maxLookback - d.itJobsSeen[q]; if (lim < max) max = lim < 0 ? 0 : (int)lim; }   // (the switch to evicted-only happens at the top of this loop)
          if (max >= 4) { int n = max >= SKIP_BULK_MIN ? skipUnfeasibleBulk(d, d.itQi[q], max) : skipUnfeasibleRun(d, d.itQi[q], max); d.itQi[q] += n; d.itJobsSeen[q] += n; }
        }
        continue;
      }
    }
    int g = d.jGang[job];
    if (g >= 0) {
      int k = d.gangSeen[g];
      d.gangArr[d.gangOff[g] + k] = job;
      d.gangSeen[g] = k + 1;
      if (k + 1 == d.jcGangCard[job]) {
        int64_t* tot = d.gangTotal + (size_t)g * d.cfg.R;
        bool allEv = true;
        for (int r = 0; r < d.cfg.R; r++) tot[r] = 0;
        for (int i = 0; i <= k; i++) { int m = d.gangArr[d.gangOff[g] + i]; vadd(d, tot, JREQ(d, m), +1); allEv = allEv && d.jcEvicted[m]; }
        d.gangAllEvicted[g] = allEv;
        d.itNext[q] = -(g + 2);
        return d.itNext[q];
      }
    } else {
      d.itNext[q] = job;
      return job;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CostBasedCandidateGangIterator (queue_scheduler.go:446-699)
struct PassCfg { bool withQueued; uint32_t maxLookback; bool skipKnown; };

DEV void updateItem(Dev& d, Ctl& c, int q, const PassCfg& pc) {  // updatePQItem :636-686
  fastHeadInvalidate(q);
  d.pqGctx[q] = -1; d.pqProposed[q] = d.pqCurrent[q] = d.pqSize[q] = 0;
  int ref = gangItPeek(d, c, q, pc.withQueued, pc.maxLookback, pc.skipKnown);
  if (ref == -1) return;
  d.pqGctx[q] = ref;
  MK(if (mkOn(d)) { mkItemOf(d, q, ref, gcJob(d, ref, 0)); return; })
  const DevCfg& cf = d.cfg;
  int64_t alloc[MAXR], withGang[MAXR];
  const int64_t* base = c.useReplayAlloc ? QV(d.replayAlloc, q) : QV(d.qAlloc, q);
  const int64_t* tot = gcTotal(d, ref);
  for (int r = 0; r < cf.R; r++) { alloc[r] = base[r] + QV(d.qPenalty, q)[r]; withGang[r] = alloc[r] + tot[r]; }
  double w = d.qWeight[q];
  d.pqProposed[q] = drf(d, withGang) / w;
  d.pqCurrent[q] = drf(d, alloc) / w;
  d.pqSize[q] = drf(d, tot) * w;
  int32_t pcp = INT32_MAX, sp = INT32_MAX;
  int cnt = gcCount(d, ref);
  for (int k = 0; k < cnt; k++) {
    int j = gcJob(d, ref, k);
    int32_t p = cf.pcPriority[d.jPc[j]];
    int32_t s = p;
    if (d.jcHasPctx[j]) s = d.pcSap[j];
    else if (d.jNode0[j] >= 0) s = d.jRunPrio[j];
    if (s < sp) sp = s;
    if (p < pcp) pcp = p;
  }
  d.pqPcPrio[q] = pcp; d.pqSchedPrio[q] = sp;
  fastItemKeys(d, c, q);
}
DEV void updateAndPush(Dev& d, Ctl& c, int q, const PassCfg& pc) {
  updateItem(d, c, q, pc);
  MK(if (mkOn(d)) { XSEG(17); if (d.pqGctx[q] != -1) mkPush(d, q); else d.pqInHeap[q] = 0; XSEG(18); return; })   // updateAndPushPQItem market_iterator.go:103-111
  d.pqInHeap[q] = d.pqGctx[q] != -1;
}

// QueueCandidateGangIteratorPQ.Less (queue_scheduler.go:738-798)
DEV bool pqAway(Dev& d, int q) { int ref = d.pqGctx[q]; return d.jAway && ref != -1 && gcCount(d, ref) > 0 && d.jAway[gcJob(d, ref, 0)] != 0; }   // item.away (:649)
DEV bool pqLess(Dev& d, const Ctl& c, int a, int b) {
  if (d.cfg.preferHome && d.jAway) { bool aa = pqAway(d, a), ab = pqAway(d, b); if (aa != ab) return !aa; }   // :744-746
  if (c.compareSchedPrio) { if (d.pqSchedPrio[a] != d.pqSchedPrio[b]) return d.pqSchedPrio[a] > d.pqSchedPrio[b]; }
  else { if (d.pqPcPrio[a] != d.pqPcPrio[b]) return d.pqPcPrio[a] > d.pqPcPrio[b]; }
  double pa = d.pqProposed[a], pb = d.pqProposed[b], ba = d.pqBudget[a], bb = d.pqBudget[b];
  if (c.preferLarge) {
    if (pa <= ba && pb <= bb) {
      double ca = d.pqCurrent[a], cb = d.pqCurrent[b];
      if (ca == cb && d.pqSize[a] != d.pqSize[b]) return d.pqSize[a] > d.pqSize[b];
      if (ca != cb) return ca < cb;
    } else if (pa > ba && pb > bb) {
      if (pa != pb) return pa < pb;
    } else if (pa <= ba) return true;
    else if (pb <= bb) return false;
  } else {
    if (pa != pb) return pa < pb;
  }
  return d.qNameRank[a] < d.qNameRank[b];
}
DEV int pqTop(Dev& d, const Ctl& c);  // argmin over items with pqInHeap (Less is a strict total order => heap top)
#ifdef ASCHED_MARKET_ROUND
DEV int pqTopAny(Dev& d, const Ctl& c) { return mkOn(d) ? mkTop(d) : pqTop(d, c); }   // MarketBasedCandidateGangIterator.Peek: pq.items[0]
#else
#define pqTopAny pqTop
#endif

DEV void costItOnlyEvicted(Dev& d, Ctl& c, const PassCfg& pc) {  // :521-544
#ifdef ASCHED_MARKET_ROUND
  if (mkOn(d)) {   // MarketBasedCandidateGangIterator.OnlyYieldEvicted (market_iterator.go:153-176): the items in array order, then heap.Init
    if (!c.onlyEvicted) {
      int n = MKS.heapN, m = 0;
      for (int i = 0; i < n; i++) {
        int q = MKD.heap[i];
        gangItOnlyEvicted(d, q);
        updateItem(d, c, q, pc);
        if (d.pqGctx[q] != -1) MKD.heap[m++] = q; else d.pqInHeap[q] = 0;
      }
      MKS.heapN = m;
      mkInit(d);
    }
    c.onlyEvicted = 1;
    return;
  }
#endif
  if (!c.onlyEvicted) {
    for (int q = 0; q < d.cfg.Q; q++) {
      if (!d.pqInHeap[q]) continue;
      gangItOnlyEvicted(d, q);
      updateItem(d, c, q, pc);
      if (d.pqGctx[q] == -1) d.pqInHeap[q] = 0;
    }
  }
  c.onlyEvicted = 1;
}
DEV void costItOnlyEvictedForQueue(Dev& d, Ctl& c, int q, const PassCfg& pc) {  // :546-566
#ifdef ASCHED_MARKET_ROUND
  if (mkOn(d)) {   // OnlyYieldEvictedForQueue (market_iterator.go:185-201): heap.Remove / heap.Fix at the item's index
    if (!c.onlyEvicted && !d.onlyEvByQueue[q]) {
      for (int i = 0; i < MKS.heapN; i++) if (MKD.heap[i] == q) {
        gangItOnlyEvicted(d, q);
        updateItem(d, c, q, pc);
        if (d.pqGctx[q] == -1) mkRemove(d, i); else mkFix(d, i);
        break;
      }
    }
    d.onlyEvByQueue[q] = 1;
    return;
  }
#endif
  if (!c.onlyEvicted && !d.onlyEvByQueue[q] && d.pqInHeap[q]) {
    gangItOnlyEvicted(d, q);
    updateItem(d, c, q, pc);
    if (d.pqGctx[q] == -1) d.pqInHeap[q] = 0;
  }
  d.onlyEvByQueue[q] = 1;
}
DEV void costItResume(Dev& d, Ctl& c, const PassCfg& pc) {  // :572-591
  if (!c.onlyEvicted) return;
  c.onlyEvicted = 0;
  for (int q = 0; q < d.cfg.Q; q++) {
    d.pqInHeap[q] = 0;
    if (!d.onlyEvByQueue[q]) gangItResume(d, q);
    d.pqBudget[q] = d.qDc[q] / d.qWeight[q];
    updateAndPush(d, c, q, pc);
  }
}
DEV void costItClear(Dev& d, Ctl& c, int top, const PassCfg& pc) {  // :595-606
  if (top < 0) return;
#ifdef ASCHED_MARKET_ROUND
  if (mkOn(d)) {   // MarketBasedCandidateGangIterator.Clear (market_iterator.go:74-89): Pop, item.it.Clear(), remember the result, update and push
    XSEG_BEGIN();
    int q = mkPop(d);
    XSEG(16);
    d.itNext[q] = -1;
    MKS.prevRank = d.qNameRank[q]; MKS.prevCost = MKD.pqPrice[q];
    updateAndPush(d, c, q, pc);
    return;
  }
#endif
  d.pqInHeap[top] = 0;
  d.itNext[top] = -1;
  updateAndPush(d, c, top, pc);
}

DEV void passInit(Dev& d, Ctl& c, const PassCfg& pc) {
  int Q = d.cfg.Q;
  fastEnterGeneric(d, c);
  fastPassReset();
  for (int q = 0; q < Q; q++) {
    d.itEi[q] = d.evOff[q]; d.itQi[q] = d.queuedOff[q]; d.itStage[q] = 0; d.itJobsSeen[q] = 0; d.itNext[q] = -1; d.itStashed[q] = -1;
    d.itJobOnlyEv[q] = 0; d.itGangOnlyEv[q] = 0; d.onlyEvByQueue[q] = 0; d.pqInHeap[q] = 0;
    d.pqBudget[q] = d.qDc[q] / d.qWeight[q];  // pushQueue :509-519
    if (d.f.iterOk == 1 && d.qsSave && q < QCAPF) d.qsSave[q].valid = 0;   // a new pass: no stream carries over (iterOk == 2: the slot holds the wide runs' arrays)
    MK(if (mkOn(d)) { MKD.itV1[q] = -1; MKD.itV2[q] = -1; })
  }
  MK(if (mkOn(d)) { MKS.heapN = 0; MKS.prevCost = 0.0; MKS.prevRank = -1; })   // a fresh MarketIteratorPQ (NewMarketCandidateGangIterator :38-60; previousResultQueue "" orders before every name)
  c.onlyEvicted = 0;
  for (int q = 0; q < Q; q++) updateAndPush(d, c, q, pc);
}

// QueueScheduler.Schedule (queue_scheduler.go:94-304)
#if defined(ASCHED_FASTPROF) && defined(__HIP_DEVICE_COMPILE__)
#define GSEG_BEGIN() long long gsT_ = CLK()
#define GSEG(i) do { long long n_ = CLK(); if ((threadIdx.x & 63) == 0) d.rs->statSeg[i] += n_ - gsT_; gsT_ = n_; } while (0)
#else
#define GSEG_BEGIN() do {} while (0)
#define GSEG(i) do {} while (0)
#endif
#ifdef ASCHED_MARKET_ROUND
// queue_scheduler.go:176-203 after a gang has been scheduled on a market-driven pool: the spot price is the lowest bid of the gang that takes the DRF cost of what this
// pass has scheduled beyond the cutoff; what the queue contexts hold at that moment is billable; the price-setting queue pays the highest competing bid
DEV_COLD void mkAfterScheduled(Dev& d, int ref, int gangQueue) {
  const DevCfg& cf = d.cfg;
  const int64_t* tot = gcTotal(d, ref);
  for (int r = 0; r < cf.R; r++) MKS.schedRes[r] += tot[r];
  if (MKS.hasSpotPrice) return;
  if (!(drf(d, MKS.schedRes) > MKS.spotCutoff)) return;
  int cnt = gcCount(d, ref);
  double price = MKD.jBid ? MKD.jBid[gcJob(d, ref, 0)] : 0.0;
  for (int k = 0; k < cnt; k++) { double b = MKD.jBid ? MKD.jBid[gcJob(d, ref, k)] : 0.0; if (b < price) price = b; }
  MKS.hasSpotPrice = 1; MKS.spotPrice = price;
  for (int i = 0; i < cf.Q * cf.R; i++) MKD.qBillable[i] = 0;
  for (int j = 0; j < cf.M; j++) {                                      // qctx.SetBillableResource (context/queue.go:108-119)
    if (!(d.jobFlags[j] & (F_SUCCESSFUL | F_RESCHEDULED))) continue;
    int q = d.jQueue[j];
    if (q < 0 || q >= cf.Q) continue;
    MKD.jobBillable[j] = 1;
    for (int r = 0; r < cf.R; r++) if (!d.cfg.isFloating[r]) QV(MKD.qBillable, q)[r] += JREQ(d, j)[r];   // jctx.KubernetesResourceRequirements
  }
  MKD.qOverride[gangQueue] = mkSecondPrice(d, gangQueue); MKD.qHasOverride[gangQueue] = 1;
}
#endif
DEV void queueSchedule(Dev& d, Ctl& c, const PassCfg& pc, const int32_t* uniOff) {
  bool limitHit = false, resumed = false;
  MK(if (mkOn(d)) { for (int r = 0; r < MAXR; r++) MKS.schedRes[r] = 0; if (d.rs->hasFpLimiter) { raise(d, ASCHED_ERR_INVALID, 950); return; } })
  c.fpLimitHit = 0;
  const bool softClock = d.cfg.maxNewJobNs > 0 || d.cfg.maxNewJobPerQueueNs > 0;
  unsigned pollCount = 0;
  int fastSkip = 0, fastStreak = 0;
  for (;;) {
    if (d.rs->error) return;
    if ((pollCount++ & 63) == 0 && cancelRequested(d)) { raise(d, ASCHED_ERR_TIMEOUT, 900); return; }  // hard timeout: abort with an error (queue_scheduler.go:105-112)
    if (!limitHit && d.rs->hasFpLimiter && d.rs->fpTokens < 1) { fastEnterGeneric(d, c); limitHit = true; c.fpLimitHit = 1; costItOnlyEvicted(d, c, pc); }
    // A fast run that ends without a single fast iteration (the head is a gang, a job that needs preemption, ...) costs a hand-over in and
    // out of the fast loop for nothing: after such a run the next ones are skipped, doubling up to 32 generic iterations, until one makes
    // progress again.  Skipping is always exact — the generic code handles every iteration.
    GSEG_BEGIN();
    // Entering and leaving the fast loop moves every queue's state between HBM and LDS.  When the generic state is the live one and the queue at the top
    // holds a single queued job that does not fit at priority -2 (asked of the level-0 structure, which the generic cascade would ask first anyway), the
    // fast loop could only hand the iteration straight back: go generic directly.
    bool headNeedsGeneric = false;
    if (fastOn(d, c) && !c.fqLive && fastSkip == 0 && d.rs->fastActive) {
      int t0 = pqTop(d, c);
      int r0 = t0 >= 0 ? d.pqGctx[t0] : -1;
      // (once the deferred replay of the evicted jobs has run, the fast loop serves such a head itself: fastPreemptIter)
      if (d.rs->replayPending && r0 >= 0 && !d.jcEvicted[r0] && d.jcAssigned[r0] < 0 && fastSelectLevel0(d, r0) == -1) headNeedsGeneric = true;
    }
    if (c.fastEnabled && d.f.iterOk == 2 && d.rs->fastActive) {
      // more than QCAPF queues: a wide run (round_wide.h) serves every head it can in one go; the generic iteration below takes the first one it cannot.  After a
      // run that got nowhere the next attempts wait (doubling, up to 64 generic iterations): an attempt costs the bulk preparation.
      if (fastSkip > 0) fastSkip--;
      else {
        int t0 = pqTop(d, c);
        int E = (t0 >= 0 && wideHeadOk(d, c, pc, t0)) ? wideRun(d, c, pc) : -1;
        if (c.cancelSeen) { raise(d, ASCHED_ERR_TIMEOUT, 901); return; }
        if (d.rs->error) return;
        if (E >= 16) fastStreak = 0;
        else if (E >= 0) { if (fastStreak < 6) fastStreak++; fastSkip = (1 << fastStreak) - 1; }
        if (E > 0) continue;
      }
    }
    else if (headNeedsGeneric) {}
    else if (fastOn(d, c) && fastSkip > 0) fastSkip--;
    else if (fastOn(d, c)) {
      int before = d.rs->statFastIters + d.rs->loopIterations;
      int pend = fastRun(d, c, pc, 0, (int*)0);
      GSEG(11);
      fastEnterGeneric(d, c);
      GSEG(12);
      if (c.cancelSeen) { raise(d, ASCHED_ERR_TIMEOUT, 901); return; }
      if (d.rs->statFastIters + d.rs->loopIterations == before && pend < 0) { if (fastStreak < 5) fastStreak++; fastSkip = (1 << fastStreak) - 1; }
      else fastStreak = 0;
      if (pend >= 0) { updateAndPush(d, c, pend, pc); continue; }
      if (!limitHit && d.rs->hasFpLimiter && d.rs->fpTokens < 1) continue;   // the fast loop's last iteration preempted (fastPreemptIter): the rate limit check comes before the next Peek (:114-121)
    }
    int top = pqTopAny(d, c);
    int ref = top >= 0 ? d.pqGctx[top] : -1;
    d.rs->statGenericIters++;
    if (d.progress) { d.progress[0] = d.rs->loopIterations; d.progress[4] = d.rs->statGenericIters; }
    if (ref == -1) {
      if (limitHit && !resumed && d.rs->terminationReason == 0) { resumed = true; costItResume(d, c, pc); continue; }
      break;
    }
    int cnt = gcCount(d, ref);
    if (cnt == 0) { costItClear(d, c, top, pc); continue; }
    bool hasPre = false;
    for (int k = 0; k < cnt; k++) if (d.jcPreempted[gcJob(d, ref, k)]) hasPre = true;
    // (a queue's precomputed evicted-stream costs, round_fast.h evCheapOff: they count every earlier evicted job of the queue as returned — one that is skipped as preempted
    //  or, below, finds no room on its node ends that for the rest of the queue's stream)
    if (hasPre) { if (d.evCheap && top < d.cfg.Q) d.evCheap[top] = 0; costItClear(d, c, top, pc); continue; }
    int reason;
    bool gangAllEv = gcAllEvicted(d, ref); int gangQueue = gcQueue(d, ref);
    int64_t tStart = (softClock && d.cfg.clockStepNs <= 0) ? clockNowNs(d) : 0;   // start := sch.clock.Now() (:157)
    GSEG(13);
    bool ok = gangSchedule(d, c, ref, &reason, uniOff);
    GSEG(14);
    if (d.rs->error) return;
    if (!ok && gangAllEv && d.evCheap && gangQueue >= 0 && gangQueue < d.cfg.Q) d.evCheap[gangQueue] = 0;
    costItClear(d, c, top, pc);
    GSEG(15);
    if (ok) {
      // scheduled jobs are recorded per job (pcNode >= 0); the PQS bookkeeping below reads them
      for (int k = 0; k < cnt; k++) {
        int j = gcJob(d, ref, k);
        if (d.jcHasPctx[j] && d.pcNode[j] >= 0) {  // pqs.go:160-166 / :213-220
          if (d.inPreempted[j]) d.inPreempted[j] = 0; else d.inScheduled[j] = 1;
          d.inSchedAndEvicted[j] = 0;
        }
      }
      MK(if (mkOn(d)) mkAfterScheduled(d, ref, gangQueue);)
    } else if (isTerminal(reason)) {
      d.rs->terminationReason = reason;
      costItOnlyEvicted(d, c, pc);
    } else if (isQueueTerminal(reason)) {
      costItOnlyEvictedForQueue(d, c, gangQueue, pc);
    }
    if (softClock && !gangAllEv) {  // RecordNewJobSchedulingDuration (:222-228, context/scheduling.go:212-240); a stepping clock advances once per Now()
      int64_t dur = d.cfg.clockStepNs > 0 ? d.cfg.clockStepNs : clockNowNs(d) - tStart;
      if (dur < 0) dur = 0;
      d.rs->totalNewJobNs += dur;
      if (d.qNewJobNs) d.qNewJobNs[gangQueue] += dur;
    }
    d.rs->loopIterations++;
  }
  if (d.rs->terminationReason == 0) d.rs->terminationReason = ASCHED_REASON_NO_REMAINING_CANDIDATES;
}

// addEvictedJobsToNodeDb (preempting_queue_scheduler.go:589-639): replay the DRF order over the evicted gangs
DEV_COLD void replayEvicted(Dev& d, Ctl& c) {
  int Q = d.cfg.Q;
  for (int q = 0; q < Q; q++) for (int r = 0; r < d.cfg.R; r++) QV(d.replayAlloc, q)[r] = QV(d.qAllocSnap, q)[r];  // allocations as the evictor left them
  int savedCmp = c.compareSchedPrio;
  c.compareSchedPrio = 0; c.useReplayAlloc = 1;
  PassCfg pc{false, 0, false};
  passInit(d, c, pc);
  int i = 0;
  for (;;) {
    if (fastOn(d, c)) {
      int pend = fastRun(d, c, pc, 1, &i);
      fastEnterGeneric(d, c);
      if (pend >= 0) { updateAndPush(d, c, pend, pc); continue; }
    }
    int top = pqTopAny(d, c);
    int ref = top >= 0 ? d.pqGctx[top] : -1;
    if (ref == -1) break;
    int cnt = gcCount(d, ref);
    for (int k = 0; k < cnt; k++) { evTabInsert(d, i, gcJob(d, ref, k)); i++; }
    vadd(d, QV(d.replayAlloc, gcQueue(d, ref)), gcTotal(d, ref), +1);
    costItClear(d, c, top, pc);
  }
  c.useReplayAlloc = 0; c.compareSchedPrio = savedCmp;
}

#include "round_ft.h"
