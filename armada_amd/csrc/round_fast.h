// round_fast.h — the fast path of the scheduling round (DESIGN.md "Fast path").
//
// The generic control code (round_ctl.h) restates the reference statement by statement with every piece of state in
// HBM: ~100 dependent memory round trips per QueueScheduler iteration and one O(N) scan of the allocatable planes per
// node selection.  This file removes both for the overwhelmingly common iteration — a single (non-gang) job that is
// either a queued job fitting without preemption or a phase-1-evicted job returning to its node — and hands anything
// else to the generic code, which sees exactly the state it would have produced itself.
//
//  1. Level-0 fast structure (first fit at priority -2, nodedb.go:737 → 840-879).  Nodes are kept in HBM sorted by
//     their level-0 order key as of round_prepare ("base", == the reference's memdb index, nodedb.go:1164-1175).
//     A node whose allocatable changes is flagged removed in the base and, while it can still host some shape, lives in
//     an LDS list ("L0").  First fit = min(first clean feasible base entry from a per-shape cursor, min over L0).  Both
//     candidates are exact: clean entries are unchanged since the sort, L0 holds current values, and keys are unique.
//  2. Per-queue head job records and small prefetch windows in LDS; a job is one 128-byte JobRec burst.
//  3. DRF cost (fairness.go:99-105) evaluated across lanes: the 3 x R float64 divisions of one updatePQItem
//     (queue_scheduler.go:636-686) issue together; same IEEE operations, same results.
//  4. Queue selection by a packed total-order key equivalent to QueueCandidateGangIteratorPQ.Less
//     (queue_scheduler.go:738-798) for finite costs.
//  5. Binds as no-return HBM atomics (node.go:416-442 arithmetic), nothing waits on them.
//
// Lane-parallel primitives have a device and a host (tests/hostsim) implementation with the same contract.
#pragma once
#include "round_ctl.h"

struct FastLds {
  // L0: live dirty nodes (current level-0 key / non-indexed columns / class bits)
  int l0Count;
  uint64_t l0Key[L0CAP]; int32_t l0Node[L0CAP]; int64_t l0Extra[MAXE][L0CAP]; uint64_t l0Cls[L0CAP];
  // per-shape base cursor + validated candidate (candNode: >=0 node, -1 exhausted, -2 needs a scan from candPos)
  int32_t candPos[SMAX]; int32_t candNode[SMAX]; uint64_t candKey[SMAX]; int64_t candExtra[MAXE][SMAX]; uint64_t candCls[SMAX];
  // queue heads + prefetch windows
  uint8_t headFast[QCAPF]; uint8_t headKind[QCAPF]; int32_t headIdx[QCAPF]; JobRec headRec[QCAPF];
  int32_t winKind[QCAPF]; int32_t winStart[QCAPF]; int32_t winCount[QCAPF]; int32_t winJob[QCAPF][WIN]; int32_t winIdx[QCAPF][WIN];
  JobRec winRec[QCAPF][WIN];
  // packed queue-order keys
  uint32_t kA[QCAPF]; uint64_t kX[QCAPF]; uint64_t kY[QCAPF];
};

#ifdef ASCHED_HOSTSIM
static FastLds g_fl;
#define FLANE 0
#else
__shared__ FastLds g_fl;
#define FLANE ((int)(threadIdx.x & 63))
#endif
#define FL g_fl

struct FitHandle { int src; int slot; };  // src 0: base candidate of the shape, 1: L0 slot

// ------------------------------------------------------------------------------------------------ small helpers
DEV uint64_t dbits(double x) { return __builtin_bit_cast(uint64_t, x); }

DEV bool fieldsGE(const Dev& d, uint64_t key, uint64_t fmin) {  // every packed field of key >= the same field of fmin
  for (int i = 0; i < d.cfg.K; i++) { uint64_t m = d.f.fieldMask[i]; if ((key & m) < (fmin & m)) return false; }
  return true;
}
DEV bool entryFits(const Dev& d, const JobRec& r, uint64_t key, int64_t ex0, int64_t ex1, uint64_t cls) {
  if (!((cls >> r.cls) & 1)) return false;          // StaticJobRequirementsMet via the requirement class (nodematching.go:161-183)
  if (!fieldsGE(d, key, r.fieldMin)) return false;  // indexed columns: alloc/res >= req/res (both resolution-aligned)
  if (d.f.E > 0 && r.req[d.f.extraCol[0]] > ex0) return false;  // non-indexed columns (nodematching.go:194-197)
  if (d.f.E > 1 && r.req[d.f.extraCol[1]] > ex1) return false;
  return true;
}
DEV bool entryLive(const Dev& d, uint64_t key, int64_t ex0, int64_t ex1) {  // could still host the smallest request of some shape
  if (!fieldsGE(d, key, d.f.minFieldMin)) return false;
  if (d.f.E > 0 && d.f.minExtra[0] > ex0) return false;
  if (d.f.E > 1 && d.f.minExtra[1] > ex1) return false;
  return true;
}
DEV bool fastOn(Dev& d, const Ctl& c) { return c.fastEnabled && d.f.iterOk; }
DEV void fastHeadInvalidate(int q) { if (q < QCAPF) FL.headFast[q] = 0; }
DEV void fastPassReset() { for (int q = 0; q < QCAPF; q++) { FL.headFast[q] = 0; FL.winCount[q] = 0; FL.winKind[q] = -1; } }

// Less (queue_scheduler.go:738-798) as a lexicographic key (kA, kX, kY, name rank); exact for finite, non-negative costs
DEV void fastItemKeys(Dev& d, const Ctl& c, int q) {
  if (!d.f.iterOk || q >= QCAPF) return;
  int32_t prio = c.compareSchedPrio ? d.pqSchedPrio[q] : d.pqPcPrio[q];
  FL.kA[q] = ~((uint32_t)prio ^ 0x80000000u);  // higher priority first
  double pa = d.pqProposed[q];
  if (c.preferLarge) {
    if (pa <= d.pqBudget[q]) { FL.kX[q] = dbits(d.pqCurrent[q]); FL.kY[q] = ~dbits(d.pqSize[q]); }  // under budget: lower current cost, then larger item
    else { FL.kX[q] = dbits(pa) | (1ull << 63); FL.kY[q] = 0; }                                       // over budget: after every under-budget item, lower proposed cost
  } else { FL.kX[q] = dbits(pa); FL.kY[q] = 0; }
}

// ------------------------------------------------------------------------------------------------ lane-parallel primitives
#ifdef ASCHED_HOSTSIM
DEV int pqTopFast(Dev& d) {
  int best = -1;
  for (int q = 0; q < d.cfg.Q; q++) {
    if (!d.pqInHeap[q]) continue;
    if (best < 0) { best = q; continue; }
    bool less;
    if (FL.kA[q] != FL.kA[best]) less = FL.kA[q] < FL.kA[best];
    else if (FL.kX[q] != FL.kX[best]) less = FL.kX[q] < FL.kX[best];
    else if (FL.kY[q] != FL.kY[best]) less = FL.kY[q] < FL.kY[best];
    else less = d.qNameRank[q] < d.qNameRank[best];
    if (less) best = q;
  }
  return best;
}
// DRF costs of one updatePQItem: proposed = drf(alloc+req)/w, current = drf(alloc)/w, size = drf(req)*w
DEV void drf3(Dev& d, const int64_t* base, const int64_t* pen, const int64_t* req, double w, double* proposed, double* current, double* size) {
  int64_t alloc[MAXR], with[MAXR];
  for (int r = 0; r < d.cfg.R; r++) { alloc[r] = base[r] + pen[r]; with[r] = alloc[r] + req[r]; }
  *proposed = drf(d, with) / w; *current = drf(d, alloc) / w; *size = drf(d, req) * w;
}
DEV void fastEnterGeneric(Dev&, Ctl&) {}
// advance the base cursor of shape r.shape to the next clean entry the job fits on
DEV void baseScan(Dev& d, const JobRec& r) {
  int s = r.shape, N = d.cfg.N;
  for (int p = FL.candPos[s]; p < N; p++) {
    if (d.baseRemoved[p]) continue;
    int64_t ex[MAXE] = {0, 0};
    for (int e = 0; e < d.f.E; e++) ex[e] = d.baseExtra[(size_t)e * d.cfg.Npad + p];
    if (!entryFits(d, r, d.baseKey[p], ex[0], ex[1], d.baseCls[p])) continue;
    FL.candPos[s] = p; FL.candNode[s] = d.baseNode[p]; FL.candKey[s] = d.baseKey[p]; FL.candCls[s] = d.baseCls[p];
    for (int e = 0; e < MAXE; e++) FL.candExtra[e][s] = ex[e];
    d.rs->statScanSteps++;
    return;
  }
  FL.candPos[s] = N; FL.candNode[s] = -1;
}
DEV uint64_t l0Search(Dev& d, const JobRec& r, int* slot) {
  uint64_t best = ~0ull; *slot = -1;
  for (int i = 0; i < FL.l0Count; i++) {
    if (FL.l0Key[i] < best && entryFits(d, r, FL.l0Key[i], FL.l0Extra[0][i], FL.l0Extra[1][i], FL.l0Cls[i])) { best = FL.l0Key[i]; *slot = i; }
  }
  return best;
}
DEV void candInvalidate(Dev& d, int n) { for (int s = 0; s < d.cfg.S; s++) if (FL.candNode[s] == n) FL.candNode[s] = -2; }
DEV void candResetAll(Dev& d, const int32_t* pos) { for (int s = 0; s < d.cfg.S && s < SMAX; s++) { FL.candNode[s] = -2; FL.candPos[s] = pos ? pos[s] : 0; } }
DEV void candSaveAll(Dev& d, int32_t* pos) { for (int s = 0; s < d.cfg.S && s < SMAX; s++) pos[s] = FL.candPos[s]; }
// load jobs [pos, pos+cnt) of a queue stream (kind 0: evicted list, 1: queued list) into the queue's window
DEV void winRefill(Dev& d, int q, int kind, int pos, int cnt) {
  const int32_t* stream = kind == 0 ? d.evList : d.queuedJobs;
  for (int k = 0; k < cnt; k++) {
    int job = stream[pos + k];
    FL.winJob[q][k] = job; FL.winIdx[q][k] = kind == 0 ? d.evIdxByPos[pos + k] : -1; FL.winRec[q][k] = d.jrec[job];
  }
  FL.winKind[q] = kind; FL.winStart[q] = pos; FL.winCount[q] = cnt;
  d.rs->statRefills++;
}
// alloc[l][r][n] -= req[r], keys[l][n] -= keyDelta for levels l in [lo, nl)  (markAllocatable, node.go:539-549)
DEV void bindUpdate(Dev& d, int n, int lo, int nl, const JobRec& r) {
  for (int l = lo; l < nl; l++) { for (int x = 0; x < d.cfg.R; x++) AL(d, l, x, n) -= r.req[x]; KEY(d, l, n) -= r.keyDelta; }
}
DEV void loadJobRec(Dev& d, int job, JobRec* out) { *out = d.jrec[job]; }
#else  // device versions: armada_sched.hip
DEV int pqTopFast(Dev& d);
DEV void drf3(Dev& d, const int64_t* base, const int64_t* pen, const int64_t* req, double w, double* proposed, double* current, double* size);
DEV void fastEnterGeneric(Dev& d, Ctl& c);
DEV void baseScan(Dev& d, const JobRec& r);
DEV uint64_t l0Search(Dev& d, const JobRec& r, int* slot);
DEV void candInvalidate(Dev& d, int n);
DEV void candResetAll(Dev& d, const int32_t* pos);
DEV void candSaveAll(Dev& d, int32_t* pos);
DEV void winRefill(Dev& d, int q, int kind, int pos, int cnt);
DEV void bindUpdate(Dev& d, int n, int lo, int nl, const JobRec& r);
DEV void loadJobRec(Dev& d, int job, JobRec* out);
#endif

// ------------------------------------------------------------------------------------------------ L0 maintenance
DEV void fastDrop(Dev& d) { d.rs->fastActive = 0; d.rs->fastOverflow++; FL.l0Count = 0; }
DEV void l0Insert(Dev& d, int n, uint64_t key, int64_t ex0, int64_t ex1, uint64_t cls) {
  int i = FL.l0Count;
  if (i >= L0CAP) { fastDrop(d); return; }
  FL.l0Key[i] = key; FL.l0Node[i] = n; FL.l0Cls[i] = cls;
  FL.l0Extra[0][i] = ex0; FL.l0Extra[1][i] = ex1;
  FL.l0Count = i + 1;
  if (i + 1 > d.rs->statL0Max) d.rs->statL0Max = i + 1;
  if (FLANE == 0) d.l0Slot[n] = i;
}
DEV void l0Remove(Dev& d, int slot) {
  int last = FL.l0Count - 1;
  int n = FL.l0Node[slot];
  if (FLANE == 0) d.l0Slot[n] = -1;
  if (slot != last) {
    FL.l0Key[slot] = FL.l0Key[last]; FL.l0Node[slot] = FL.l0Node[last]; FL.l0Cls[slot] = FL.l0Cls[last];
    FL.l0Extra[0][slot] = FL.l0Extra[0][last]; FL.l0Extra[1][slot] = FL.l0Extra[1][last];
    if (FLANE == 0) d.l0Slot[FL.l0Node[slot]] = slot;
  }
  FL.l0Count = last;
}

// node n's level-0 allocatable was changed by the generic code: bring base flags / L0 / candidates in line
DEV void fastTouch(Dev& d, int n) {
  if (!d.f.structOk || !d.rs->fastActive) return;
  uint64_t key = KEY(d, 0, n);
  int64_t ex0 = d.f.E > 0 ? AL(d, 0, d.f.extraCol[0], n) : 0, ex1 = d.f.E > 1 ? AL(d, 0, d.f.extraCol[1], n) : 0;
  int pos = d.posOf[n], slot = d.l0Slot[n];
  if (FLANE == 0) d.baseRemoved[pos] = 1;
  candInvalidate(d, n);
  bool live = entryLive(d, key, ex0, ex1);
  if (slot >= 0) {
    if (live) { FL.l0Key[slot] = key; FL.l0Extra[0][slot] = ex0; FL.l0Extra[1][slot] = ex1; }
    else l0Remove(d, slot);
  } else if (live) l0Insert(d, n, key, ex0, ex1, d.nodeCls[n]);
}

// first fit at priority -2 for a job record; -1 none; handle says where the winner came from
DEV int fastFirstFit(Dev& d, const JobRec& r, FitHandle* h) {
  if (r.never) return -1;
  int s = r.shape;
  if (FL.candNode[s] == -2) baseScan(d, r);
  uint64_t bk = FL.candNode[s] >= 0 ? FL.candKey[s] : ~0ull;
  int slot;
  uint64_t lk = l0Search(d, r, &slot);
  if (lk < bk) { h->src = 1; h->slot = slot; return FL.l0Node[slot]; }
  if (bk == ~0ull) return -1;
  h->src = 0; h->slot = -1;
  return FL.candNode[s];
}
DEV int fastSelectLevel0(Dev& d, int job) {
  if (!d.f.structOk || !d.rs->fastActive) return -2;
  JobRec r;
  loadJobRec(d, job, &r);
  FitHandle h;
  return fastFirstFit(d, r, &h);
}
// the job of record r was bound to node n found through handle h: level-0 bookkeeping of the fast structure
DEV void fastAfterBind(Dev& d, const JobRec& r, int n, const FitHandle& h) {
  int64_t q0 = d.f.E > 0 ? r.req[d.f.extraCol[0]] : 0, q1 = d.f.E > 1 ? r.req[d.f.extraCol[1]] : 0;
  if (h.src == 0) {
    int s = r.shape;
    uint64_t key = FL.candKey[s] - r.keyDelta, cls = FL.candCls[s];
    int64_t ex0 = FL.candExtra[0][s] - q0, ex1 = FL.candExtra[1][s] - q1;
    if (FLANE == 0) d.baseRemoved[FL.candPos[s]] = 1;
    candInvalidate(d, n);
    if (entryLive(d, key, ex0, ex1)) l0Insert(d, n, key, ex0, ex1, cls);
  } else {
    int slot = h.slot;
    uint64_t key = FL.l0Key[slot] - r.keyDelta;
    int64_t ex0 = FL.l0Extra[0][slot] - q0, ex1 = FL.l0Extra[1][slot] - q1;
    if (entryLive(d, key, ex0, ex1)) { FL.l0Key[slot] = key; FL.l0Extra[0][slot] = ex0; FL.l0Extra[1][slot] = ex1; }
    else l0Remove(d, slot);
  }
}

// ------------------------------------------------------------------------------------------------ launch persistence
DEV void fastLoad(Dev& d) {  // kernel start: rebuild the LDS side from HBM
  FL.l0Count = 0;
  for (int q = 0; q < QCAPF; q++) { FL.headFast[q] = 0; FL.winCount[q] = 0; FL.winKind[q] = -1; }
  if (!d.f.structOk || !d.rs->fastActive) return;
  candResetAll(d, d.candPosSave);
  int cnt = d.rs->l0SaveCount;
  for (int i = 0; i < cnt; i++) {
    int n = d.l0Save[i];
    l0Insert(d, n, KEY(d, 0, n), d.f.E > 0 ? AL(d, 0, d.f.extraCol[0], n) : 0, d.f.E > 1 ? AL(d, 0, d.f.extraCol[1], n) : 0, d.nodeCls[n]);
  }
}
DEV void fastSave(Dev& d) {  // kernel end
  if (!d.f.structOk || !d.rs->fastActive) { d.rs->l0SaveCount = 0; return; }
  candSaveAll(d, d.candPosSave);
  for (int i = 0; i < FL.l0Count; i++) d.l0Save[i] = FL.l0Node[i];
  d.rs->l0SaveCount = FL.l0Count;
}

// ------------------------------------------------------------------------------------------------ queue iterator, fast
DEV int pqTopAny(Dev& d, const Ctl& c) {
  if (!fastOn(d, c)) return pqTop(d, c);
  int t = pqTopFast(d);
#ifdef ASCHED_HOSTSIM
  if (t != pqTop(d, c)) { fprintf(stderr, "hostsim: packed queue key disagrees with Less (fast %d, generic %d)\n", t, pqTop(d, c)); abort(); }
#endif
  return t;
}

// costItClear(top) (queue_scheduler.go:595-606) + QueuedGangIterator.Peek (:376-432) + updatePQItem (:636-686) for the
// next single job of queue q, from the prefetch window; gang members and rare iterator states go to the generic code.
DEV void fastAdvance(Dev& d, Ctl& c, int q, const PassCfg& pc) {
  d.pqInHeap[q] = 0; d.itNext[q] = -1;
  if (q >= QCAPF) { updateAndPush(d, c, q, pc); return; }
  FL.headFast[q] = 0;
  for (;;) {
    if (pc.maxLookback != 0 && !d.itGangOnlyEv[q] && (uint32_t)d.itJobsSeen[q] >= pc.maxLookback) gangItOnlyEvicted(d, q);
    int kind = -1, pos = 0, end = 0;  // what jobItNext (jobiteration.go:179-228) yields next, not yet consumed
    if (d.itStage[q] == 0) {
      if (d.itEi[q] < d.evOff[q + 1]) { kind = 0; pos = d.itEi[q]; end = d.evOff[q + 1]; }
      else d.itStage[q] = 1;
    }
    if (kind < 0 && !(d.itJobOnlyEv[q] || !pc.withQueued) && d.itQi[q] < d.queuedOff[q + 1]) { kind = 1; pos = d.itQi[q]; end = d.queuedOff[q + 1]; }
    if (kind < 0) { d.pqGctx[q] = -1; d.pqProposed[q] = d.pqCurrent[q] = d.pqSize[q] = 0; return; }
    if (kind == 0 && !c.fastEvStatic) { updateAndPush(d, c, q, pc); return; }
    if (!(FL.winKind[q] == kind && pos >= FL.winStart[q] && pos < FL.winStart[q] + FL.winCount[q])) {
      int cnt = end - pos; if (cnt > WIN) cnt = WIN;
      winRefill(d, q, kind, pos, cnt);
    }
    int k = pos - FL.winStart[q];
    const JobRec& r = FL.winRec[q][k];
    int job = FL.winJob[q][k];
    if (r.gang >= 0) { updateAndPush(d, c, q, pc); return; }  // the generic iterator continues from the same state
    if (kind == 0) d.itEi[q] = pos + 1;
    else { d.itQi[q] = pos + 1; if (FLANE == 0) resetJctxForQueued(d, job); d.itJobsSeen[q]++; }
    if (pc.skipKnown && d.rs->numUnfeasible > 0 && kind == 1 && d.unfeasible[r.shape]) {  // queue_scheduler.go:398-413
      fastEnterGeneric(d, c);
      d.jcReason[job] = d.unfeasibleReason[r.shape];
      d.jcHasPctx[job] = 1; d.pcNode[job] = -1; d.pcMethod[job] = ASCHED_METHOD_NONE;
      sctxAddJob(d, job);
      d.jcReason[job] = ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY;
      continue;
    }
    d.itNext[q] = job; d.pqGctx[q] = job;
    FL.headRec[q] = r; FL.headKind[q] = (uint8_t)kind; FL.headIdx[q] = FL.winIdx[q][k]; FL.headFast[q] = 1;
    const int64_t* base = c.useReplayAlloc ? QV(d.replayAlloc, q) : QV(d.qAlloc, q);
    double pr, cu, sz;
    drf3(d, base, QV(d.qPenalty, q), FL.headRec[q].req, d.qWeight[q], &pr, &cu, &sz);
    d.pqProposed[q] = pr; d.pqCurrent[q] = cu; d.pqSize[q] = sz;
    int32_t p = d.cfg.pcPriority[r.pc];
    d.pqPcPrio[q] = p; d.pqSchedPrio[q] = kind == 0 ? r.runPrio : p;  // evicted: run.ScheduledAtPriority (queue_scheduler.go:660-672)
    fastItemKeys(d, c, q);
    d.pqInHeap[q] = 1;
    return;
  }
}

DEV int levelsUpTo(const DevCfg& c, int32_t cutoff) { int nl = 0; while (nl < c.P && c.prios[nl] <= cutoff) nl++; return nl; }  // prios ascend

// One QueueScheduler iteration (queue_scheduler.go:94-304 body) for the head of queue `top` when it is a single job that
// (a) is queued and fits at priority -2 or (b) is a phase-1-evicted job returning to its node.  Returns false WITHOUT side
// effects when the iteration needs the generic code (any constraint failing, preemption, gangs, ...).
DEV bool fastIter(Dev& d, Ctl& c, const PassCfg& pc, int top) {
  int q = top;
  if (q >= QCAPF || !FL.headFast[q]) return false;
  const JobRec& r = FL.headRec[q];
  int job = d.pqGctx[q], kind = FL.headKind[q], R = d.cfg.R;
  RoundScalars& s = *d.rs;
  if (s.numPreemptedMarks != 0) return false;  // queue_scheduler.go:150-156 needs the per-job flag: generic
  bool ev = kind == 0;
  int pcx = r.pc;
  int32_t prio;
  int n;
  FitHandle h; h.src = 0; h.slot = -1;
  if (!ev) {
    if (!s.fastActive) return false;
    if (vexceeds(d, s.scheduled, d.cfg.maxToSchedule)) return false;  // CheckRoundConstraints (constraints.go:113-119)
    if (d.qCordoned[q] || s.globalTokens < 1 || s.globalBurst < 1 || d.qTokens[q] < 1 || d.qBurst[q] < 1) return false;  // CheckJobConstraints (:121-157)
    if (d.hasPcLimit) { const int64_t* a = QPV(d.qAllocByPc, q, pcx); const int64_t* lim = QPV(d.qPcLimit, q, pcx); for (int x = 0; x < R; x++) if (a[x] + r.req[x] > lim[x]) return false; }
    for (int x = 0; x < R; x++) if (d.cfg.disallowed[x] && r.req[x] > 0) return false;  // nodedb.go:596-601
    if (d.cfg.disableHome) return false;
    prio = d.cfg.pcPriority[pcx];
    n = fastFirstFit(d, r, &h);
    if (n < 0) return false;  // the generic cascade (gate, fair-share, urgency) decides
    s.numNodeQueries++;
  } else {
    if (!c.fastEvStatic) return false;
    prio = r.runPrio; n = r.node0;
    if (!s.lvl0NonNeg) {  // nodedb.go:897-906
      fastEnterGeneric(d, c);
      int level = levelOf(d.cfg, prio);
      if (level < 0) return false;
      if (!((d.nodeFlags[n] & 1) || fitsAlloc(d, r.req, level, n))) return false;
    }
    // else: alloc[level] >= alloc[-2] + req >= req on every column (bucket arithmetic, DESIGN.md "Evicted jobs always return")
  }
  // ---- commit: sctx.AddGangSchedulingContext (scheduling.go:391-434)
  int64_t* qa = QV(d.qAlloc, q); int64_t* qap = QPV(d.qAllocByPc, q, pcx);
  for (int x = 0; x < R; x++) { qa[x] += r.req[x]; qap[x] += r.req[x]; s.allocated[x] += r.req[x]; }
  if (ev) {
    int64_t* qe = QPV(d.qEvictedByPc, q, pcx);
    for (int x = 0; x < R; x++) { qe[x] -= r.req[x]; s.evicted[x] -= r.req[x]; }
    s.numEvictedJobs--;
  } else {
    int64_t* qs = QPV(d.qSchedByPc, q, pcx);
    for (int x = 0; x < R; x++) { qs[x] += r.req[x]; s.scheduled[x] += r.req[x]; }
    s.numScheduledJobs++; s.numScheduledGangs++;
  }
  // ---- SelectNodeForJobWithTxn result + BindJobToNode (nodedb.go:538-630, 1046-1068)
  int32_t cutoff = d.cfg.pcPreemptible[pcx] ? prio : NONPREEMPTIBLE_CUTOFF;
  int nl = levelsUpTo(d.cfg, cutoff);
  bindUpdate(d, n, ev ? 1 : 0, nl, r);  // evicted job: level -2 gets -req (bind) and +req (un-evict): unchanged (node.go:416-442)
  c.l1Dirty = 1;
  if (FLANE == 0) {
    d.jcReason[job] = 0; d.jcHasPctx[job] = 1; d.pcNode[job] = n; d.pcSap[job] = prio;
    d.jobNode[job] = n; d.jobCutoff[job] = cutoff; d.jobEvictedOnNode[job] = 0; d.schedAtPrio[job] = prio; d.inSchedAndEvicted[job] = 0;
    if (ev) {
      d.pcPap[job] = prio; d.pcMethod[job] = ASCHED_METHOD_RESCHEDULED; d.jobFlags[job] = F_RESCHEDULED; d.inPreempted[job] = 0;
      int idx = FL.headIdx[q];
      d.evTabAlive[idx] = 0; d.evIndexOfJob[job] = -1;  // nodedb.go:441-446
    } else {
      d.pcPap[job] = ASCHED_EVICTED_PRIORITY; d.pcMethod[job] = ASCHED_METHOD_NO_PREEMPTION; d.jobFlags[job] = F_SUCCESSFUL; d.inScheduled[job] = 1;
    }
  }
  if (!ev) {
    fastAfterBind(d, r, n, h);
    reserveN(&s.globalTokens, s.globalBurst, s.globalRateInf, 1);  // gang_scheduler.go:118-123
    reserveN(&d.qTokens[q], d.qBurst[q], d.qRateInf[q], 1);
  }
  fastAdvance(d, c, q, pc);
  return true;
}

// one step of addEvictedJobsToNodeDb (preempting_queue_scheduler.go:589-639) for a single evicted job
DEV bool fastReplayStep(Dev& d, Ctl& c, const PassCfg& pc, int top, int* counter) {
  int q = top;
  if (q >= QCAPF || !FL.headFast[q]) return false;
  const JobRec& r = FL.headRec[q];
  int job = d.pqGctx[q], i = *counter;
  if (FLANE == 0) { d.evTabJob[i] = job; d.evTabAlive[i] = 1; d.evIndexOfJob[job] = i; }
  if (i + 1 > d.rs->evictedTableSize) d.rs->evictedTableSize = i + 1;
  *counter = i + 1;
  int64_t* ra = QV(d.replayAlloc, q);
  for (int x = 0; x < d.cfg.R; x++) ra[x] += r.req[x];
  d.rs->statFastReplay++;
  fastAdvance(d, c, q, pc);
  return true;
}
