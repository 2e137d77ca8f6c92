// round_run.h — the bulk (data-parallel) phases of PreemptingQueueScheduler.Schedule and the command
// dispatcher of the control kernel.  Bulk phases are per-element functions run by every thread of the
// workgroup through wgBulk(); cross-job sums use int64 atomics (integer => order independent => exact).
#pragma once
#include "round_ctl.h"
#include "round_fast.h"
#include "round_market.h"

enum BulkKind {
  B_RESET_GANGSEEN = 1, B_FILTER1, B_NODE_OVER, B_FILTER3, B_GANG_CLOSURE, B_EVICT_APPLY1, B_EVICT_APPLY3, B_KEYS_ALL,
  B_UNBIND, B_RESET_EVTAB, B_CLEAR_UNFEASIBLE, B_INIT_ALLOC, B_POPULATE, B_RESET_JOBS, B_GATHER_SCHED, B_GATHER_PRE,
  B_EVIDX, B_LVL0, B_EVKEYS, B_EVKEYS_OFF, B_EVKEYS_ON, B_EVSUM, B_SNAP, B_EVALIVE,
  B_FAIR_ZERO, B_FAIR_COUNT, B_FAIR_PSUM, B_FAIR_POFF, B_FAIR_SCATTER, B_FAIR_SORT,
  B_QSSUM, B_QSSTITCH, B_QSKEYS,
  B_AGG_RUN, B_AGG_QUEUED,
};

DEV void wgBulk(Dev& d, int kind, int n);  // every element i in [0,n) through bulkElem(), then a workgroup barrier
DEV void wgFtBuild(Dev& d, int phase, int n);  // one pass of the fair-share threshold table's build (round_ft.h ftBuildAny), helper workgroups included
DEV void wgBulkWide(Dev& d, int kind, int n);  // the same with the helper workgroups taking their share (bodies that touch HBM only)
#include "round_wide.h"
#include "round_merge.h"
DEV int wgCompactFlagged(Dev& d, const int32_t* order, const int32_t* segOff, int nseg, int n, const uint8_t* flag, int32_t* dst, int32_t* outSegOff);
DEV int wgCompactIota(Dev& d, int n, const uint8_t* flag, int32_t* dst);

DEV void atomicMarkAllocatable(Dev& d, int n, int32_t cutoff, const int64_t* req, int sign) {
  const DevCfg& c = d.cfg;
  for (int l = 0; l < c.P; l++)
    if (c.prios[l] <= cutoff)
      for (int r = 0; r < c.R; r++) if (req[r]) atomicAddI64(&AL(d, l, r, n), sign * req[r]);
}
DEV void atomicVadd(Dev& d, int64_t* a, const int64_t* b, int sign) { for (int r = 0; r < d.cfg.R; r++) if (b[r]) atomicAddI64(&a[r], sign * b[r]); }

// Evictor.Evict body for one job (is/scheduling/eviction.go:245-260) + sctx.EvictJob (context/scheduling.go:551-572),
// order-independent form.  phase3: PQS bookkeeping of pqs.go:182-195.
DEV void evictApply(Dev& d, int j, bool phase3) {
  if (!d.evFlag[j]) return;
  int n = d.jobNode[j];
  if (d.schedAtPrio[j] == NO_PRIORITY) { raise(d, ASCHED_ERR_INTERNAL, 800); return; }  // EvictJobsFromNode nodedb.go:1085-1088
  const int64_t* req = JREQ(d, j);
  d.jobEvictedOnNode[j] = 1;  // Node.EvictJob node.go:449-474
  atomicMarkAllocatable(d, n, d.jobCutoff[j], req, +1);
  atomicMarkAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, -1);
  // fresh jctx pinned to the node (eviction.go:246-253)
  d.jcEvicted[j] = 1; d.jcAssigned[j] = n; d.jcReason[j] = 0; d.jcHasPctx[j] = 0; d.jcUniValue[j] = -1; d.jcStagedBy[j] = -1;
  int g = d.jGang[j];
  d.jcGangCard[j] = g >= 0 ? d.gangOff[g + 1] - d.gangOff[g] : 1;  // setEvictedGangCardinality pqs.go:462-483
  // sctx.EvictJob
  int q = d.jQueue[j], pc = d.jPc[j];
  uint8_t f = d.jobFlags[j];
  bool sched = f & F_SUCCESSFUL, resched = f & F_RESCHEDULED;
  if (sched || resched) {
    if (sched) { atomicVadd(d, QPV(d.qSchedByPc, q, pc), req, -1); f &= ~F_SUCCESSFUL; }
    if (resched) f &= ~F_RESCHEDULED;
    MK(if (mkOn(d) && MKD.jobBillable[j]) { atomicVadd(d, QV(MKD.qBillable, q), req, -1); MKD.jobBillable[j] = 0; })   // context/queue.go:368-376
  } else {
    atomicVadd(d, QPV(d.qEvictedByPc, q, pc), req, +1);
    f |= F_EVICTED;
  }
  atomicVadd(d, QPV(d.qAllocByPc, q, pc), req, -1);
  atomicVadd(d, QV(d.qAlloc, q), req, -1);
  if (sched) { atomicVadd(d, d.rs->scheduled, req, -1); atomicAddI32(&d.rs->numScheduledJobs, -1); }
  else { atomicVadd(d, d.rs->evicted, req, +1); atomicAddI32(&d.rs->numEvictedJobs, 1); }
  atomicVadd(d, d.rs->allocated, req, -1);
  d.jobFlags[j] = f;
  if (!phase3) { d.inPreempted[j] = 1; d.preemptedNode[j] = n; }
  else if (d.inScheduled[j]) { d.inScheduled[j] = 0; d.inSchedAndEvicted[j] = 1; d.preemptedNode[j] = n; }
  else { d.inPreempted[j] = 1; d.preemptedNode[j] = n; }
}

DEV void bulkElem(Dev& d, int kind, int i) {
  const DevCfg& c = d.cfg;
  switch (kind) {
    case B_RESET_GANGSEEN: d.gangSeen[i] = 0; break;
    case B_FILTER1: {  // NewNodeEvictor job filter (pqs.go:101-136)
      int n = d.jobNode[i], q = d.jQueue[i];
      // a cross-pool away job (job.LatestRun().Pool() != sctx.Pool, :102-104) is never evicted for balancing: urgency preemption and the oversubscribed evictor take it
#ifdef ASCHED_MARKET_ROUND
      if (mkOn(d)) { d.evFlag[i] = n >= 0 && !d.jobEvictedOnNode[i] && q >= 0 && q < c.Q && !(d.jAway && d.jAway[i]); break; }   // pqs.go:117-119: a market-driven pool evicts every job
#endif
      bool ok = n >= 0 && !d.jobEvictedOnNode[i] && q >= 0 && q < c.Q && !(d.jAway && d.jAway[i]) && c.pcPreemptible[d.jPc[i]];
      d.evFlag[i] = ok && d.qEvictable[q];
    } break;
    case B_NODE_OVER: {  // NewOversubscribedEvictor node filter (eviction.go:145-157)
      int m = 0;
      for (int l = 0; l < c.P; l++) {
        if (c.prios[l] == ASCHED_EVICTED_PRIORITY) continue;
        for (int r = 0; r < c.R; r++) if (AL(d, l, r, i) < 0) { m |= 1 << l; break; }
      }
      d.nodeOver[i] = m;
    } break;
    case B_FILTER3: {  // NewOversubscribedEvictor job filter (eviction.go:158-178)
      int n = d.jobNode[i], q = d.jQueue[i];
      bool f = n >= 0 && !d.jobEvictedOnNode[i] && q >= 0 && q < c.Q && c.pcPreemptible[d.jPc[i]] && d.schedAtPrio[i] != NO_PRIORITY;
      if (f) { int l = levelOf(c, d.schedAtPrio[i]); f = l >= 0 && ((d.nodeOver[n] >> l) & 1); }
      d.evFlag[i] = f;
    } break;
    case B_GANG_CLOSURE: {  // evictGangs (pqs.go:357-424): a partially evicted gang is evicted entirely
      bool any = false;
      for (int k = d.gangOff[i]; k < d.gangOff[i + 1]; k++) any = any || d.evFlag[d.gangJobs[k]];
      if (any) for (int k = d.gangOff[i]; k < d.gangOff[i + 1]; k++) { int j = d.gangJobs[k]; if (d.jobNode[j] >= 0 && !d.jobEvictedOnNode[j]) d.evFlag[j] = 1; }
    } break;
    case B_EVICT_APPLY1: evictApply(d, i, false); break;
    case B_EVICT_APPLY3: evictApply(d, i, true); break;
    case B_KEYS_ALL: updateKeys(d, i); break;
    case B_UNBIND: {  // unbindJobs (pqs.go:775-798): RemoveJob of preempted ∪ scheduled-and-evicted on their node
      if (!(d.inPreempted[i] || d.inSchedAndEvicted[i])) break;
      int n = d.preemptedNode[i];
      if (d.jobNode[i] != n) break;  // RemoveJob of an unknown job is a no-op
      const int64_t* req = JREQ(d, i);
      if (d.jobEvictedOnNode[i]) atomicMarkAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, +1);
      else atomicMarkAllocatable(d, n, d.jobCutoff[i], req, +1);
      d.jobEvictedOnNode[i] = 0; d.jobNode[i] = -1;
    } break;
    case B_RESET_EVTAB: if (d.evTabAlive[i]) d.evIndexOfJob[d.evTabJob[i]] = -1; d.evTabAlive[i] = 0; break;
    case B_GATHER_SCHED: { int j = d.resJob[i]; d.resNode[i] = d.pcNode[j]; d.resPrio[i] = d.schedAtPrio[j]; d.resMethod[i] = d.pcMethod[j]; } break;
    case B_GATHER_PRE: { int j = d.resPreJob[i]; d.resPreNode[i] = d.preemptedNode[j]; } break;
    case B_CLEAR_UNFEASIBLE: d.unfeasible[i] = 0; break;
    case B_EVIDX: if (d.evIdxByPos) d.evIdxByPos[i] = d.evIndexOfJob[d.evList[i]]; break;  // evicted-table Index per evicted-list position (fast path)
    case B_SNAP: d.qAllocSnap[i] = d.qAlloc[i]; break;
    case B_EVALIVE: {  // a table entry of the deferred replay is alive iff its job is still evicted on its node (not rescheduled, not preempted since)
      int j = d.evTabJob[i];
      bool alive = d.jobEvictedOnNode[j] != 0;
      d.evTabAlive[i] = alive;
      if (!alive) d.evIndexOfJob[j] = -1;
    } break;
    case B_EVKEYS_OFF: d.evCheap[i] = 0; d.evMono[i] = 0; break;
    case B_EVKEYS_ON: d.evCheap[i] = 1; d.evMono[i] = 1; break;
    // Queue-order costs of every evicted job, in eviction-list order (pqs.go:589-639 replays exactly these; pass 1 re-reads
    // them).  The list is ordered by queue; thread i owns positions [i*C, (i+1)*C): pass A sums the requests of its range per
    // queue segment and flags gang members, one thread stitches the carries, pass B evaluates the DRF costs.
    case B_EVSUM: {
      int n = d.rs->numEvictedList, T = d.evChunks, C = (n + T - 1) / T;
      int p0 = i * C, p1 = p0 + C < n ? p0 + C : n;
      int64_t* part = d.evPart + (size_t)i * (2 * MAXR + 4);
      int64_t head[MAXR], tail[MAXR];
      for (int r = 0; r < MAXR; r++) head[r] = tail[r] = 0;
      int q = 0, qFirst = -1, qLast = -1; bool crossed = false;
      if (p0 < p1) { while (d.evOff[q + 1] <= p0) q++; qFirst = q; }
      for (int p = p0; p < p1; p++) {
        while (d.evOff[q + 1] <= p) { q++; crossed = true; for (int r = 0; r < c.R; r++) tail[r] = 0; }
        int job = d.evList[p];
        if (d.jGang[job] >= 0 || (d.jAligned && !d.jAligned[job])) d.evCheap[q] = 0;   // (a request off the index grid: its rebind recomputes the node's keys — generic)
        const int64_t* req = JREQ(d, job);
        for (int r = 0; r < c.R; r++) { tail[r] += req[r]; if (!crossed) head[r] += req[r]; }
        qLast = q;
      }
      for (int r = 0; r < MAXR; r++) { part[r] = head[r]; part[MAXR + r] = tail[r]; }
      part[2 * MAXR] = qFirst; part[2 * MAXR + 1] = qLast; part[2 * MAXR + 2] = crossed;
    } break;
    case B_EVKEYS: {
      int n = d.rs->numEvictedList, T = d.evChunks, C = (n + T - 1) / T;
      int p0 = i * C, p1 = p0 + C < n ? p0 + C : n;
      if (p0 >= p1) break;
      const int64_t* carry = d.evPart + (size_t)i * (2 * MAXR + 4) + 0;  // rewritten by the stitch step: carry-in of this range
      int q = 0;
      while (d.evOff[q + 1] <= p0) q++;
      int64_t a[MAXR], with[MAXR];
      for (int r = 0; r < c.R; r++) a[r] = QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r] + carry[r];
      PackedKey prev; bool havePrev = false;
      uint64_t* edge = d.evEdge + (size_t)i * 8;
      for (int p = p0; p < p1; p++) {
        while (d.evOff[q + 1] <= p) { q++; havePrev = false; for (int r = 0; r < c.R; r++) a[r] = QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r]; }
        int job = d.evList[p];
        const int64_t* req = JREQ(d, job);
        for (int r = 0; r < c.R; r++) with[r] = a[r] + req[r];
        double w = d.qWeight[q];
        EvKey e;
        e.proposed = drf(d, with) / w; e.current = drf(d, a) / w; e.size = drf(d, req) * w;
        e.pcPrio = c.pcPriority[d.jPc[job]]; e.job = job;
        d.evKey[p] = e;
        PackedKey pk = packKey3(c.preferLarge, e.pcPrio, e.proposed, e.current, e.size, d.qDc[q] / w);
        if (havePrev && packedLess(pk, 0, prev, 0)) d.evMono[q] = 0;  // a later evicted job orders before an earlier one: heap merge != key sort
        if (p == p0) { edge[0] = pk.A; edge[1] = pk.X; edge[2] = pk.Y; edge[3] = (uint64_t)q; }
        prev = pk; havePrev = true;
        for (int r = 0; r < c.R; r++) a[r] = with[r];
      }
      edge[4] = prev.A; edge[5] = prev.X; edge[6] = prev.Y; edge[7] = (uint64_t)q;
    } break;
    // Queued-job streams (dev.h QS_*, round_fast.h "stream run"): element e of queue q's stream is queuedJobs[base + e].  Item i of B_QSSUM /
    // B_QSKEYS owns elements [c * QS_CHUNK, (c + 1) * QS_CHUNK) of queue q = i / QS_CPQ, c = i % QS_CPQ: sums of the requests and the first element the
    // stream cannot contain — a gang member (the generic iterator assembles gangs), a job whose scheduling key is known to be unfeasible (it is skipped
    // with a record, queue_scheduler.go:398-413), a job requesting a disallowed resource — then one item per queue turns the sums into carries and
    // cuts the stream at the first such element, then the costs (the same float64 operations as updatePQItem, queue_scheduler.go:636-686).
    case B_QSSUM: {
      int q = i / QS_CPQ, ch = i % QS_CPQ;
      const QsIn& in = d.qsIn[q];
      int64_t* part = d.qsPart + (size_t)i * (MAXR + 2);
      int e0 = ch * QS_CHUNK, e1 = e0 + QS_CHUNK < in.len ? e0 + QS_CHUNK : in.len;
      int64_t sum[MAXR]; for (int r = 0; r < MAXR; r++) sum[r] = 0;
      int barrier = INT32_MAX;
      for (int e = e0; e < e1; e++) {
        int job = d.queuedJobs[in.base + e];
        const int64_t* req = JREQ(d, job);
        bool stop = d.jGang[job] >= 0 || (e > 0 && in.skipUnf && d.unfeasible[d.jShape[job]]);
        for (int r = 0; r < c.R; r++) if (c.disallowed[r] && req[r] > 0) stop = true;
        if (stop) { barrier = e; break; }
        for (int r = 0; r < c.R; r++) sum[r] += req[r];
      }
      for (int r = 0; r < MAXR; r++) part[r] = sum[r];
      part[MAXR] = barrier;
    } break;
    case B_QSSTITCH: {
      const QsIn& in = d.qsIn[i];
      int64_t run[MAXR]; for (int r = 0; r < MAXR; r++) run[r] = 0;
      int len = in.len; bool cut = false;
      for (int ch = 0; ch < QS_CPQ && ch * QS_CHUNK < in.len; ch++) {
        int64_t* part = d.qsPart + ((size_t)i * QS_CPQ + ch) * (MAXR + 2);
        int barrier = (int)part[MAXR];
        for (int r = 0; r < MAXR; r++) { int64_t s = part[r]; part[r] = run[r]; run[r] += s; }
        if (barrier != INT32_MAX) { len = barrier; cut = true; break; }
      }
      if (in.pad) break;   // a kept stream: its length and end flag stand
      d.qsLen[2 * i] = len;
      d.qsLen[2 * i + 1] = (!cut && in.len > 0 && in.base + in.len == d.queuedOff[i + 1]) ? 1 : 0;   // the queue's list ends where the stream ends
    } break;
    case B_QSKEYS: {
      int q = i / QS_CPQ, ch = i % QS_CPQ;
      const QsIn& in = d.qsIn[q];
      if (in.len == 0) break;   // no new stream for this queue (a kept one stays as it is)
      int len = d.qsLen[2 * q];
      int e0 = ch * QS_CHUNK, e1 = e0 + QS_CHUNK < len ? e0 + QS_CHUNK : len;
      if (e0 >= e1) break;
      const int64_t* carry = d.qsPart + (size_t)i * (MAXR + 2);
      int64_t a[MAXR], with[MAXR];
      for (int r = 0; r < c.R; r++) a[r] = in.a0[r] + carry[r];
      double w = in.weight;
      EvKey* out = d.qsKey + (size_t)q * QS_CMAX;
      for (int e = e0; e < e1; e++) {
        int job = d.queuedJobs[in.base + e];
        const int64_t* req = JREQ(d, job);
        for (int r = 0; r < c.R; r++) with[r] = a[r] + req[r];
        EvKey k;
        k.proposed = drf(d, with) / w; k.current = drf(d, a) / w; k.size = drf(d, req) * w;
        k.pcPrio = c.pcPriority[d.jPc[job]]; k.job = job;
        out[e] = k;
        for (int r = 0; r < c.R; r++) a[r] = with[r];
      }
    } break;
    // The round-input builder's per-queue aggregates on the device (scheduling_algo.go:591-698, 805; asched_round_prepare when the caller passes none):
    // allocation per (queue, priority class) = requests of the running jobs; demand per (queue, class) = requests of the running jobs + of the queued jobs of
    // every queue that is not cordoned.  Integer atomics: order independent.  SM_AGG_FIN caps each class by the queue's limit and sums over the classes.
    case B_AGG_RUN: {
      if (d.jNode0[i] < 0) break;
      int q = d.jQueue[i];
      if (q < 0 || q >= c.Q) break;
      const int64_t* req = JREQ(d, i);
      size_t o = ((size_t)q * c.npc + d.jPc[i]) * c.R;
      for (int r = 0; r < c.R; r++) if (req[r]) { atomicAddI64(&d.qAllocByPc[o + r], req[r]); atomicAddI64(&d.qDemandByPc[o + r], req[r]); }
    } break;
    case B_AGG_QUEUED: {
      int lo = 0, hi = c.Q;   // the queue whose list holds position i
      while (lo < hi) { int mid = (lo + hi) >> 1; if (d.queuedOff[mid + 1] <= i) lo = mid + 1; else hi = mid; }
      int q = lo;
      if (q >= c.Q || d.qCordoned[q]) break;
      int job = d.queuedJobs[i];
      const int64_t* req = JREQ(d, job);
      size_t o = ((size_t)q * c.npc + d.jPc[job]) * c.R;
      for (int r = 0; r < c.R; r++) if (req[r]) atomicAddI64(&d.qDemandByPc[o + r], req[r]);
    } break;
    // per-node index of the evicted table (ensureFairIndex): count -> offsets -> scatter -> per-node sort by descending Index
    case B_FAIR_ZERO: d.accStamp[i] = 0; break;
    case B_FAIR_COUNT: { int n = d.evTabAlive[i] ? d.jcAssigned[d.evTabJob[i]] : -1; if (n >= 0) atomicAddI32(&d.accStamp[n], 1); } break;   // (dead entries are left out: ensureFairIndex)
    case B_FAIR_PSUM: {
      int C = (c.N + FAIR_CHUNKS - 1) / FAIR_CHUNKS, n0 = i * C, n1 = n0 + C < c.N ? n0 + C : c.N;
      int sum = 0;
      for (int n = n0; n < n1; n++) sum += d.accStamp[n];
      d.fairPart[i] = sum;
    } break;
    case B_FAIR_POFF: {
      int C = (c.N + FAIR_CHUNKS - 1) / FAIR_CHUNKS, n0 = i * C, n1 = n0 + C < c.N ? n0 + C : c.N;
      int run = d.fairPart[i];
      for (int n = n0; n < n1; n++) { int cnt = d.accStamp[n]; d.fairOff[n] = run; d.accStamp[n] = run; run += cnt; }
    } break;
    case B_FAIR_SCATTER: { int n = d.evTabAlive[i] ? d.jcAssigned[d.evTabJob[i]] : -1; if (n >= 0) d.fairEnt[atomicFetchAddI32(&d.accStamp[n], 1)] = i; } break;
    case B_FAIR_SORT: {
      int k0 = d.fairOff[i], k1 = d.fairOff[i + 1];
      for (int a = k0 + 1; a < k1; a++) {
        int v = d.fairEnt[a], b = a - 1;
        while (b >= k0 && d.fairEnt[b] < v) { d.fairEnt[b + 1] = d.fairEnt[b]; b--; }
        d.fairEnt[b + 1] = v;
      }
      for (int k = k0; k < k1; k++) d.fairEntJob[k] = d.evTabJob[d.fairEnt[k]];
    } break;
    case B_LVL0: { bool neg = false; for (int r = 0; r < c.R; r++) neg = neg || AL(d, 0, r, i) < 0; if (neg) d.rs->lvl0NonNeg = 0; } break;
    case B_INIT_ALLOC: {  // fresh NodeDb (scheduling_algo.go:517): AllocatableByPriority[p] = allocatable (node.go:79-85)
      for (int l = 0; l < c.P; l++) for (int r = 0; r < c.R; r++)
        AL(d, l, r, i) = d.alloc0 ? d.alloc0[((size_t)l * c.R + r) * c.Npad + i] : d.allocatable[(size_t)r * c.Npad + i];
    } break;
    case B_RESET_JOBS: {
      d.schedAtPrio[i] = NO_PRIORITY; d.jobNode[i] = -1; d.jobCutoff[i] = 0; d.jobEvictedOnNode[i] = 0; d.jobFlags[i] = 0;
      d.jcEvicted[i] = 0; d.jcAssigned[i] = -1; d.jcReason[i] = 0; d.jcHasPctx[i] = 0; d.pcNode[i] = -1; d.pcSap[i] = 0; d.pcPap[i] = ASCHED_MIN_PRIORITY;
      d.pcMethod[i] = 0; d.jcGangCard[i] = d.jGang[i] >= 0 ? d.jGangCard[i] : 1; d.jcPreempted[i] = 0; d.jcUniValue[i] = -1; d.jcStagedBy[i] = -1;
      d.inPreempted[i] = d.inScheduled[i] = d.inSchedAndEvicted[i] = 0; d.preemptedNode[i] = -1; d.evFlag[i] = 0;
      d.evTabAlive[i] = 0; d.evIndexOfJob[i] = -1;
      if (d.excl) { d.excl[i] = -1; if (i == 0) { EXCL(d)->count = 0; EXCL(d)->dynCount = 0; } }   // (asched_excluded_nodes: a round starts with nothing on record)
    } break;
    case B_POPULATE: {  // populateNodeDb: bind every running job (nodedb.go:57-75, scheduling_algo.go:1019-1098)
      int n = d.jNode0[i];
      if (n < 0) break;
      int32_t prio = bindPriority(d, i, d.jRunPrio[i]);
      int32_t cutoff = cutoffFor(d, i, prio);
      atomicMarkAllocatable(d, n, cutoff, JREQ(d, i), -1);
      d.jobNode[i] = n; d.jobCutoff[i] = cutoff; d.schedAtPrio[i] = prio;
    } break;
  }
}

// SchedulingContext.updateFairShares (context/scheduling.go:262-342): float64, queues in name order, this exact operation order.
// Scratch: pqProposed = constrainedDemandShare, pqCurrent = spareShare, pqInHeap = achievedDemand, itNext = name order.
DEV_COLD void updateFairShares(Dev& d, const double* givenCds) {
  const DevCfg& cf = d.cfg;
  int Q = cf.Q;
  double weightSum = 0;
  bool totalZero = true;
  for (int r = 0; r < cf.R; r++) if (cf.totalResources[r] != 0) totalZero = false;
  for (int q = 0; q < Q; q++) weightSum += d.qWeight[q];  // sctx.WeightSum, accumulated in AddQueueSchedulingContext order
  for (int q = 0; q < Q; q++) d.itNext[d.qNameRank[q]] = q;  // name ranks are a permutation of 0..Q-1
  for (int q = 0; q < Q; q++) {
    d.pqProposed[q] = givenCds ? givenCds[q] : (totalZero ? 1.0 : drf(d, QV(d.qDemand, q)));
    d.qFair[q] = d.qWeight[q] / weightSum; d.qDc[q] = 0; d.qUc[q] = 0; d.pqCurrent[q] = 0; d.pqInHeap[q] = 0;
  }
  double unallocated = 1.0;
  for (int it = 0; it < 10 && unallocated > 0.01; it++) {
    double totalWeight = 0.0;
    for (int k = 0; k < Q; k++) { int q = d.itNext[k]; if (!d.pqInHeap[q]) totalWeight += d.qWeight[q]; }
    for (int k = 0; k < Q; k++) {
      int q = d.itNext[k];
      double tw = totalWeight;
      if (d.pqInHeap[q]) tw += d.qWeight[q];
      d.qUc[q] += (d.qWeight[q] / tw) * (unallocated - d.pqCurrent[q]);
    }
    if (totalWeight <= 0.0) break;
    for (int k = 0; k < Q; k++) { int q = d.itNext[k]; if (!d.pqInHeap[q]) d.qDc[q] += (d.qWeight[q] / totalWeight) * unallocated; }
    unallocated = 0.0;
    for (int k = 0; k < Q; k++) {
      int q = d.itNext[k];
      double s = d.qDc[q] - d.pqProposed[q];
      if (s > 0) { d.qDc[q] = d.pqProposed[q]; d.pqInHeap[q] = 1; d.pqCurrent[q] = s; unallocated += s; }
      else d.pqCurrent[q] = 0;
    }
  }
  for (int q = 0; q < Q; q++) { d.itNext[q] = -1; d.pqInHeap[q] = 0; }
}

// PreemptingQueueScheduler.evict (pqs.go:291-353) for an evictor whose job filter has been evaluated into evFlag
DEV_COLD int pqsEvict(Dev& d, Ctl& c, bool phase3) {
  long long t0 = CLK();
  wgBulk(d, B_GANG_CLOSURE, d.cfg.G);
  wgBulk(d, phase3 ? B_EVICT_APPLY3 : B_EVICT_APPLY1, d.cfg.M);
  wgBulk(d, B_KEYS_ALL, d.cfg.N);
  // InMemoryJobRepository.EnqueueMany (jobiteration.go:85-108): per-queue lists in SchedulingOrderCompare order ==
  // order-preserving compaction of the pre-sorted job order
  // (a market-driven pool orders them with MarketSchedulingOrderCompare, pqs.go:292-295: the same compaction of the job order sorted by that comparer)
  const int32_t* order = d.ordAll;
  MK(if (mkOn(d)) order = MKD.ord;)
  int n = wgCompactFlagged(d, order, d.ordAllOff, d.cfg.Q, d.ordAllOff[d.cfg.Q], d.evFlag, d.evList, d.evOff);
  d.rs->numEvictedList = n;
  wgBulk(d, B_RESET_EVTAB, d.rs->evictedTableSize);  // nodeDb.Reset() (nodedb.go:299-313)
  d.rs->evictedTableSize = 0; d.rs->fairIndexValid = 0; d.rs->ftValid = 0;
  wgBulk(d, B_RESET_GANGSEEN, d.cfg.G);
  d.rs->replayPending = 0;
  wgBulk(d, B_SNAP, d.cfg.Q * d.cfg.R);
  bool lazy = false;
  if (d.evCheap) {
    if (!phase3 && fastOn(d, c) && n > 0) {
      wgBulk(d, B_EVKEYS_ON, d.cfg.Q);
      wgBulk(d, B_EVSUM, d.evChunks);
      // stitch: carry-in of range i = requests of the same queue summed over the ranges before it (sequential over evChunks entries)
      int64_t run[MAXR]; int runQ = -1;
      for (int r = 0; r < MAXR; r++) run[r] = 0;
      for (int i = 0; i < d.evChunks; i++) {
        int64_t* part = d.evPart + (size_t)i * (2 * MAXR + 4);
        int qFirst = (int)part[2 * MAXR], qLast = (int)part[2 * MAXR + 1]; bool crossed = part[2 * MAXR + 2] != 0;
        int64_t carry[MAXR];
        for (int r = 0; r < MAXR; r++) carry[r] = (qFirst >= 0 && qFirst == runQ) ? run[r] : 0;
        if (qFirst >= 0) {
          if (!crossed) { for (int r = 0; r < MAXR; r++) run[r] = carry[r] + part[r]; runQ = qFirst; }   // whole range in one queue
          else { for (int r = 0; r < MAXR; r++) run[r] = part[MAXR + r]; runQ = qLast; }
        }
        for (int r = 0; r < MAXR; r++) part[r] = carry[r];
      }
      wgBulk(d, B_EVKEYS, d.evChunks);
      {  // monotonicity across chunk borders
        int C = (n + d.evChunks - 1) / d.evChunks, nc = (n + C - 1) / C;
        for (int i = 1; i < nc; i++) {
          const uint64_t* a = d.evEdge + (size_t)(i - 1) * 8; const uint64_t* b = d.evEdge + (size_t)i * 8;
          if (a[7] != b[3]) continue;
          PackedKey last, first; last.A = (uint32_t)a[4]; last.X = a[5]; last.Y = a[6]; first.A = (uint32_t)b[0]; first.X = b[1]; first.Y = b[2];
          if (packedLess(first, 0, last, 0)) d.evMono[(int)b[3]] = 0;
        }
      }
      lazy = true;  // every queue's evicted stream is gang-free: the replay is a pure merge of precomputed costs and can wait
      for (int q = 0; q < d.cfg.Q; q++) if (!d.evCheap[q]) lazy = false;
    } else wgBulk(d, B_EVKEYS_OFF, d.cfg.Q);
  }
  long long t1 = CLK();
  if (lazy) d.rs->replayPending = 1;  // the evicted-table Index is only read by fair-share preemption (nodedb.go:935-1043): assign it on first use
  else { replayEvicted(d, c); wgBulk(d, B_EVIDX, n); }  // addEvictedJobsToNodeDb
  long long t2 = CLK();
  d.rs->statClk[phase3 ? 3 : 0] += t1 - t0; d.rs->statClk[1] += t2 - t1;
  return n;
}

// ---- the serial slivers of the split round (host-driven sequence): each runs as one thread of a tiny kernel between grid-wide passes
enum SmallKind { SM_QEVICTABLE = 1, SM_EVICT_POST, SM_STITCH, SM_MONO, SM_LVL0_BEGIN, SM_FINAL, SM_PREPARE_FIN, SM_AGG_FIN };
DEV void roundSmall(Dev& d, int what, int arg) {
  const DevCfg& cf = d.cfg;
  switch (what) {
    case SM_QEVICTABLE:  // balance-evictor filter inputs: the start-of-round queue allocations (pqs.go:124-134)
      for (int q = 0; q < cf.Q; q++) {
        double actual = drf(d, QV(d.qAlloc, q));
        double fair = d.qDc[q] > d.qFair[q] ? d.qDc[q] : d.qFair[q];  // math.Max
        if (cf.protectUncapped) fair = d.qUc[q];
        double frac = actual / fair;
        d.qEvictable[q] = !(frac <= cf.protectedFraction);
      }
      break;
    case SM_EVICT_POST:  // after the compaction of an evictor's job list: nodeDb.Reset() bookkeeping (nodedb.go:299-313)
      d.rs->numEvictedList = arg;
      d.rs->evictedTableSize = 0; d.rs->fairIndexValid = 0; d.rs->ftValid = 0; d.rs->replayPending = 0;
      break;
    case SM_STITCH: {  // carry-in of range i = requests of the same queue summed over the ranges before it (B_EVSUM -> B_EVKEYS)
      int64_t run[MAXR]; int runQ = -1;
      for (int r = 0; r < MAXR; r++) run[r] = 0;
      for (int i = 0; i < d.evChunks; i++) {
        int64_t* part = d.evPart + (size_t)i * (2 * MAXR + 4);
        int qFirst = (int)part[2 * MAXR], qLast = (int)part[2 * MAXR + 1]; bool crossed = part[2 * MAXR + 2] != 0;
        int64_t carry[MAXR];
        for (int r = 0; r < MAXR; r++) carry[r] = (qFirst >= 0 && qFirst == runQ) ? run[r] : 0;
        if (qFirst >= 0) {
          if (!crossed) { for (int r = 0; r < MAXR; r++) run[r] = carry[r] + part[r]; runQ = qFirst; }
          else { for (int r = 0; r < MAXR; r++) run[r] = part[MAXR + r]; runQ = qLast; }
        }
        for (int r = 0; r < MAXR; r++) part[r] = carry[r];
      }
    } break;
    case SM_MONO: {  // monotonicity across chunk borders; every stream gang-free -> the replay can wait (lazy)
      int n = arg;
      int C = (n + d.evChunks - 1) / d.evChunks, nc = C > 0 ? (n + C - 1) / C : 0;
      for (int i = 1; i < nc; i++) {
        const uint64_t* a = d.evEdge + (size_t)(i - 1) * 8; const uint64_t* b = d.evEdge + (size_t)i * 8;
        if (a[7] != b[3]) continue;
        PackedKey last, first; last.A = (uint32_t)a[4]; last.X = a[5]; last.Y = a[6]; first.A = (uint32_t)b[0]; first.X = b[1]; first.Y = b[2];
        if (packedLess(first, 0, last, 0)) d.evMono[(int)b[3]] = 0;
      }
      bool lazy = true;
      for (int q = 0; q < cf.Q; q++) if (!d.evCheap[q]) lazy = false;
#ifdef ASCHED_HOSTSIM
      if (getenv("HS_NO_LAZY")) lazy = false;   // tests: the walk at the start of the pass even when it could wait
#endif
      if (lazy) d.rs->replayPending = 1;
    } break;
    case SM_LVL0_BEGIN: d.rs->lvl0NonNeg = 1; break;
    case SM_AGG_FIN: {   // after B_AGG_*: queue allocation, capped demand, the scheduling context's allocated total
      const DevCfg& cf = d.cfg;
      for (int r = 0; r < cf.R; r++) d.rs->allocated[r] = 0;
      for (int q = 0; q < cf.Q; q++) for (int r = 0; r < cf.R; r++) {
        int64_t a = 0, dm = 0;
        for (int p = 0; p < cf.npc; p++) {
          size_t i = ((size_t)q * cf.npc + p) * cf.R + r;
          a += d.qAllocByPc[i];
          int64_t v = d.qDemandByPc[i];
          dm += (d.hasPcLimit && d.qPcLimit[i] < v) ? d.qPcLimit[i] : v;   // constraints.go:187-197
        }
        QV(d.qAlloc, q)[r] = a; QV(d.qDemand, q)[r] = dm; d.rs->allocated[r] += a;
      }
    } break;
    case SM_PREPARE_FIN:  // tail of CMD_PREPARE: fresh evicted table, fair shares (context/scheduling.go:262-342)
      d.rs->fastActive = 0; d.rs->evictedTableSize = 0; d.rs->fairIndexValid = 0; d.rs->ftValid = 0; d.rs->numUnfeasible = 0;
      updateFairShares(d, (const double*)0);
      break;
    case SM_FINAL:
      d.rs->fastActive = 0;  // unbinding changed priority -2 allocatable behind the fast structure: next round_prepare rebuilds it
      d.rs->terminationReason = d.cmdIO[4];
      break;
  }
}

DEV void schedulePass(Dev& d, Ctl& c, bool withQueued, bool skipKey, bool cmpPrio) {  // pqs.schedule (pqs.go:712-772)
  wgBulk(d, B_CLEAR_UNFEASIBLE, d.cfg.S);
  d.rs->numUnfeasible = 0;
  wgBulk(d, B_RESET_GANGSEEN, d.cfg.G);
  c.skipKeyCheck = skipKey; c.compareSchedPrio = cmpPrio; c.useReplayAlloc = 0;
  PassCfg pc{withQueued, d.cfg.maxLookback, true};
  passInit(d, c, pc);
  queueSchedule(d, c, pc, d.uniOff);
}

// Per-node index of the evicted table (CSR node -> table Indexes, descending) for fair-share preemption.  It holds the entries that were ALIVE when it was
// built: late in a pass most of the table is dead (rescheduled or preempted jobs), and the per-node evaluation of the wide pass walks a node's slice entry by
// entry — two dependent gathers per dead entry, nine entries per node at 100 000 nodes x 900 000 evicted jobs.  Rebuilt when the table has grown, when an entry
// has come back (a transaction abort, a taken-back commit of the fast path: both clear fairIndexValid) and every FAIR_REBUILD_EVERY queries, which
// drops the entries that have died since (entries that die after a build stay in it and are skipped by their alive flag, as before).
#ifndef FAIR_REBUILD_EVERY
#ifdef ASCHED_HOSTSIM
#define FAIR_REBUILD_EVERY 5      // (the CPU build of the tests rebuilds all the time: every seeded round exercises an index built in the middle of a pass)
#else
#define FAIR_REBUILD_EVERY 8192   // (measured on configs[4], profiles/r03z: what pays is leaving out the entries that are dead when the index is first built in a pass — 128 / 512 / 2 048 / 8 192 queries between
                                  //  rebuilds give the same time inside the passes, and a rebuild costs 4.5 ms at 900 000 table entries: 17.3 / 14.7 / 14.1 / 13.7 s per round)
#endif
#endif
// The threshold table from the planes and the evicted table as they are now: three grid-wide passes shared with the helper workgroups.  They have an op of their
// own (wgFtBuild) instead of three more kinds in bulkElem: a call inside that switch cost the stream preparation 3-5 % of the headline round (measured, profiles/r03f).
#ifndef ASCHED_NO_FT
DEV void ftBuild(Dev& d) {
#ifdef ASCHED_HOSTSIM
  { static long builds = 0; static const bool st = getenv("HS_FT_STATS") != nullptr; if (st && (++builds % 100) == 1) fprintf(stderr, "ftBuild %ld (queries %d retries %d node updates %d)\n", builds, d.rs->statFt[0], d.rs->statFt[1], d.rs->statFt[2]); }
#endif
  wgFtBuild(d, 0, d.cfg.N * ((d.ftS + FT_CHUNK - 1) / FT_CHUNK));
  wgFtBuild(d, 1, d.ftS * d.ftNB1);
  wgFtBuild(d, 2, d.ftS * 64);
  d.rs->ftValid = 1;
#ifdef ASCHED_FT_COUNT_BUILDS
  d.rs->statFt[1] += 1000;   // (experiment builds: table builds show in ft_retries as thousands)
#endif
}
#endif
DEV_COLD void ensureFairIndex(Dev& d) {
  const bool periodic = d.rs->fairIndexValid && ++d.accEpoch_unused >= FAIR_REBUILD_EVERY;   // (queries since the last build: a counter of this launch, in the descriptor's LDS copy)
#ifndef ASCHED_NO_FT
  if (d.rs->fairIndexValid && !periodic) { if (d.ftT && !d.rs->ftValid && d.rs->ftWanted) ftBuild(d); return; }
#endif
  if (d.rs->fairIndexValid && !periodic) return;
  d.accEpoch_unused = 0;
  int E = d.rs->evictedTableSize, N = d.cfg.N;
  wgBulk(d, B_FAIR_ZERO, N);
  wgBulk(d, B_FAIR_COUNT, E);
  wgBulk(d, B_FAIR_PSUM, FAIR_CHUNKS);
  int run = 0;
  for (int i = 0; i < FAIR_CHUNKS; i++) { int v = d.fairPart[i]; d.fairPart[i] = run; run += v; }
  d.fairOff[N] = run;
  wgBulk(d, B_FAIR_POFF, FAIR_CHUNKS);
  wgBulk(d, B_FAIR_SCATTER, E);
  wgBulk(d, B_FAIR_SORT, N);
  d.rs->fairIndexValid = 1;
#ifndef ASCHED_NO_FT
  if (d.ftT && d.rs->ftWanted) ftBuild(d);
#endif
}

DEV void swapLoopArrays(Dev& d) {
  QueueLoopArrays t;
  t.itEi = d.itEi; t.itQi = d.itQi; t.itStage = d.itStage; t.itJobsSeen = d.itJobsSeen; t.itNext = d.itNext; t.itStashed = d.itStashed;
  t.itJobOnlyEv = d.itJobOnlyEv; t.itGangOnlyEv = d.itGangOnlyEv; t.onlyEvByQueue = d.onlyEvByQueue;
  t.pqProposed = d.pqProposed; t.pqCurrent = d.pqCurrent; t.pqBudget = d.pqBudget; t.pqSize = d.pqSize; t.pqPcPrio = d.pqPcPrio; t.pqSchedPrio = d.pqSchedPrio; t.pqGctx = d.pqGctx; t.pqInHeap = d.pqInHeap;
  d.itEi = d.alt.itEi; d.itQi = d.alt.itQi; d.itStage = d.alt.itStage; d.itJobsSeen = d.alt.itJobsSeen; d.itNext = d.alt.itNext; d.itStashed = d.alt.itStashed;
  d.itJobOnlyEv = d.alt.itJobOnlyEv; d.itGangOnlyEv = d.alt.itGangOnlyEv; d.onlyEvByQueue = d.alt.onlyEvByQueue;
  d.pqProposed = d.alt.pqProposed; d.pqCurrent = d.alt.pqCurrent; d.pqBudget = d.alt.pqBudget; d.pqSize = d.alt.pqSize; d.pqPcPrio = d.alt.pqPcPrio; d.pqSchedPrio = d.alt.pqSchedPrio; d.pqGctx = d.alt.pqGctx; d.pqInHeap = d.alt.pqInHeap;
  d.alt = t;
}
// The deferred addEvictedJobsToNodeDb (pqs.go:589-639): its result — the evicted-table Index of every evicted job — is a pure
// function of the state the evictor left (qAllocSnap, the eviction lists), so it can be computed at first use.  It runs on its
// own set of iterator / heap arrays; entries of jobs that have been rescheduled or preempted in the meantime come out dead.
DEV_COLD void ensureReplaySlow(Dev& d, Ctl& c) {
  if (!d.rs->replayPending) return;
  d.rs->replayPending = 0;
  const long long tReplay0 = CLK();
  fastEnterGeneric(d, c);
  int sOnly = c.onlyEvicted, sCmp = c.compareSchedPrio, sUse = c.useReplayAlloc, sSkip = c.skipKeyCheck, sEv = c.fastEvStatic;
  swapLoopArrays(d);
  replayEvicted(d, c);
  fastEnterGeneric(d, c);
  wgBulk(d, B_EVIDX, d.rs->numEvictedList);
  wgBulk(d, B_EVALIVE, d.rs->evictedTableSize);
  d.rs->fairIndexValid = 0;   // the alive flags were rewritten: entries may have come back
  swapLoopArrays(d);
  c.onlyEvicted = sOnly; c.compareSchedPrio = sCmp; c.useReplayAlloc = sUse; c.skipKeyCheck = sSkip; c.fastEvStatic = sEv;
  fastPassReset();
  for (int q = 0; q < d.cfg.Q; q++) if (d.pqGctx[q] != -1) fastItemKeys(d, c, q);  // the pass's own heads again
  d.rs->statClk[1] += CLK() - tReplay0;   // (round_stats kclk_replay: the deferred replay sits inside the pass that needed it)
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_RANK_CHECK")) {   // the walk's table against replay_rank.h's closed form
    extern int hsRankCheck(Dev& d);
    hsRankCheck(d);
  }
#endif
}

// PreemptingQueueScheduler.Schedule (pqs.go:86-289)
DEV_COLD void runRound(Dev& d, Ctl& c) {
  const DevCfg& cf = d.cfg;
  for (int q = 0; q < cf.Q; q++) {  // balance-evictor filter inputs are the start-of-round queue allocations (pqs.go:124-134)
    double actual = drf(d, QV(d.qAlloc, q));
    double fair = d.qDc[q] > d.qFair[q] ? d.qDc[q] : d.qFair[q];  // math.Max
    if (cf.protectUncapped) fair = d.qUc[q];
    double frac = actual / fair;
    d.qEvictable[q] = !(frac <= cf.protectedFraction);
  }
  long long ta = CLK();
  if (d.progress) d.progress[1] = 1;
  wgBulk(d, B_FILTER1, cf.M);
  d.rs->statClk[0] += CLK() - ta;
  c.fastEvStatic = 1;
  int n1 = pqsEvict(d, c, false);
  d.rs->lvl0NonNeg = 1;
  wgBulk(d, B_LVL0, cf.N);
  long long tb = CLK();
  if (d.progress) d.progress[1] = 2;
  c.skipEnter = fastOn(d, c) && d.evMono != nullptr && d.rs->lvl0NonNeg && n1 > 0;
  schedulePass(d, c, true, false, false);
  c.skipEnter = 0;
  long long tc = CLK();
  d.rs->statClk[2] += tc - tb;
  c.fastEvStatic = 0;
  if (d.rs->error) return;
  int firstTermination = d.rs->terminationReason;
  if (d.progress) d.progress[1] = 3;
  wgBulk(d, B_NODE_OVER, cf.N);
  wgBulk(d, B_FILTER3, cf.M);
  int n3 = pqsEvict(d, c, true);
  long long td = CLK();
  if (d.progress) d.progress[1] = 4;
  { int fe = c.fastEnabled; if (d.jAway) c.fastEnabled = 0;   // (cross-pool away jobs may be among the jobs the oversubscribed evictor took: the generic code knows their rules)
    if (n3 > 0) schedulePass(d, c, false, true, true);
    c.fastEnabled = fe; }
  if (d.rs->error) return;
  long long te = CLK();
  d.rs->statClk[4] += te - td;
  if (d.progress) d.progress[1] = 5;
  wgBulk(d, B_UNBIND, cf.M);
  wgBulk(d, B_KEYS_ALL, cf.N);
  d.rs->fastActive = 0;  // unbinding changed priority -2 allocatable behind the fast structure: next round_prepare rebuilds it
  d.rs->terminationReason = firstTermination;
  d.cmdIO[0] = n1; d.cmdIO[1] = n3;
  d.cmdIO[2] = wgCompactIota(d, cf.M, d.inScheduled, d.resJob);
  d.cmdIO[3] = wgCompactIota(d, cf.M, d.inPreempted, d.resPreJob);
  wgBulk(d, B_GATHER_SCHED, d.cmdIO[2]);
  wgBulk(d, B_GATHER_PRE, d.cmdIO[3]);
  d.rs->statClk[5] += CLK() - te;
}

// ------------------------------------------------------------------------------------------------
// control-kernel commands (the NodeDb-level entry points of the C ABI run through the same device code as the round)

// Stream preparation (round_fast.h "stream run").  Per queue: a stream with elements left is kept; otherwise the queue may get
//   * an evicted stream: its head is a phase-1-evicted job of a gang-free eviction list in a pass where evicted jobs always return — the costs are
//     the ones B_EVKEYS computed for the whole list, nothing to prepare;
//   * a queued stream: its head is a single queued job peeked from the queue's list and the queue may still schedule new jobs — how far the stream may
//     reach (the list, QS_CMAX, the queue's rate-limit tokens, the lookback limit, the global tokens), then the three bulk passes over the queues
//     that need one (they read only what is written to d.qsIn, and run on every wave of the workgroup: the node engine must not be live).
DEV_NOINLINE int fastStreamPrepare(Dev& d, FastCtx fc, int Q, int allowed, int allowBulk, int top, int capHint) {
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_NO_STREAM")) return 0;
#endif
  const FastK k = fastKRef(d);
  int skipUnf = fc.skipKnown && RS.numUnfeasible > 0;
  int cap = allowed < QS_CMAX ? allowed : QS_CMAX;
  if (cap > capHint) cap = capHint;
  if (d.cfg.G > 0 && cap > 4096) cap = 4096;   // a pool with gangs: a stream is cut at the queue's next gang member, and the chunks behind the cut are scanned for nothing (BASELINE configs[3]: 686 -> 838 ms with 32 768)
  bool evOk = fc.evStatic && RS.lvl0NonNeg && RS.numPreemptedMarks == 0;
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_NO_EV_STREAM")) evOk = false;
#endif
  // (round 6) a head the generic code peeked and no fast iteration has looked at yet has no cached record (headFast 0), and such a queue used to get its first stream only
  // after a fast iteration had served it once: a pass started with 64 short runs, each ending at the next such queue.  Their records are fetched here.
  unsigned long long needHead = 0;   // (found one lane per queue: a serial look at 64 LDS records costs ~20 k ticks per call, and crowded rounds prepare streams thousands of times)
  if (allowBulk > 0) {
#if defined(ASCHED_HOSTSIM) || !defined(__HIP_DEVICE_COMPILE__)
    for (int q = 0; q < Q && q < 64; q++) if (FL.inHeap[q] && !FL.hot[q].headFast && !(FL.hot[q].sLen > FL.hot[q].sPos) && FL.hot[q].gctx >= 0) needHead |= 1ull << q;
#else
    { const int q = FLANE; needHead = __ballot(q < Q && FL.inHeap[q] && !FL.hot[q].headFast && !(FL.hot[q].sLen > FL.hot[q].sPos) && FL.hot[q].gctx >= 0); }
#endif
  }
  while (needHead) {
    const int q = __builtin_ctzll(needHead); needHead &= needHead - 1;
    const int job = UNI32(FL.hot[q].gctx);
    QHot f = FL.hot[q];
    uniQHot(f);
    fastLoadHead(k, q, job, f);
    if (FLANE == 0) { FL.hot[q].headFast = 1; FL.hot[q].headKind = f.headKind; FL.hot[q].headIdx = f.headIdx; FL.hot[q].headPos = f.headPos; }
    LANE0_PUBLISHED();
  }
  FOR_LANES(q, QCAPF) FL.tmpQ[q] = 0;
  FOR_LANES(q, Q) {
    QHot& f = FL.hot[q];
    QsIn in; in.base = 0; in.len = 0; in.skipUnf = skipUnf; in.pad = 1; in.weight = f.weight;   // pad 1: the stitch pass leaves the queue's d.qsLen alone
    for (int r = 0; r < MAXR; r++) in.a0[r] = FL.qAlloc[q][r] + FL.qPenalty[q][r];
    int status = 0;
    if (f.sLen > f.sPos) status = 1;
    else {
      f.sLen = 0; f.sPos = 0;
      bool head = FL.inHeap[q] && f.gctx >= 0;
      if (head && evOk && f.itStage == 0 && f.evCheap && !f.effValid && f.headPos >= 0 && f.headPos == f.itEi - 1 && f.headPos < f.evEnd && f.evDone == f.headPos &&
          f.evApplied <= f.evDone && k.evList[f.headPos] == f.gctx) {
        FL.sKind[q] = 1; f.sPos = 0; f.sLen = f.evEnd - f.headPos; f.ewCount = 0; f.ewStart = 0;
        status = 1;
      } else if (head && cap >= 1 && f.headFast && f.headKind == 1 && f.itStage == 1 && !f.itJobOnlyEv && !f.cordoned && f.burst >= 1 && f.tokens >= 1 &&
                 f.evApplied == f.evDone && f.itQi >= 1 && f.itQi <= f.qEnd && k.queuedJobs[f.itQi - 1] == f.gctx) {   // (a stashed job is not at the list position before the cursor)
        int len = f.qEnd - (f.itQi - 1);
        if (len > cap) len = cap;
        if (!f.rateInf && f.tokens < (double)len) len = (int)f.tokens;
        if (fc.maxLookback != 0 && !f.itGangOnlyEv) {   // element e >= 1 is peeked when itJobsSeen = seen + e - 1 < maxLookback (queue_scheduler.go:434-444)
          int64_t lim = (int64_t)fc.maxLookback - f.itJobsSeen + 1;
          if (lim < 1) lim = 1;
          if (len > lim) len = (int)lim;
        }
        in.base = f.itQi - 1; in.len = len; in.pad = 0;
        status = 2;
      }
    }
    FL.tmpQ[q] = status;
    d.qsIn[q] = in;
  }
  int bulk = 0;
  for (int q = 0; q < Q; q++) if (UNI32(FL.tmpQ[q]) == 2) bulk++;
  if (bulk && allowBulk < 0) bulk = 0;   // a cheap attempt: the queues that would need the bulk passes stay without a stream
  if (bulk) {
    if (!allowBulk) return 2;
    wgBulkWide(d, B_QSSUM, Q * QS_CPQ);
    wgBulkWide(d, B_QSSTITCH, Q);
    wgBulkWide(d, B_QSKEYS, Q * QS_CPQ);
    int total = 0;
    FOR_LANES(q, Q) if (FL.tmpQ[q] == 2) { int len = d.qsLen[2 * q]; FL.hot[q].sLen = len; FL.hot[q].sPos = 0; FL.sKind[q] = 0; FL.hot[q].ewCount = 0; FL.hot[q].ewStart = 0; }
    for (int q = 0; q < Q; q++) if (UNI32(FL.tmpQ[q]) == 2) total += d.qsLen[2 * q];
    if (FLANE == 0) RS.statStreamPrepared += total;
    LANE0_PUBLISHED();
  }
  return UNI32(FL.hot[top].sLen) > UNI32(FL.hot[top].sPos) ? 1 : 0;
}

// A queue's stream ends at its next gang member, and preparing streams again is a bulk job (three passes over all queues, the node engine stopped): on gang-heavy
// rounds most single jobs therefore went through the per-job iteration (profiles/r03r_*).  After a gang has been placed its queue's next stretch of single jobs —
// up to the next gang member, at most QS_ONE_MAX entries — is prepared by the CONTROL WAVE ALONE, one lane per entry, while the engine stays live: the same
// barrier rules and the same float64 operations per entry as B_QSSUM / B_QSSTITCH / B_QSKEYS (entry e's allocation = the queue's allocation + the requests of entries
// 0 .. e-1, which every lane sums for itself: integer adds, so the order is free).  Returns the prepared length (0: the queue's head is not a streamable single job).
#define QS_ONE_MAX 1024
DEV_NOINLINE int fastStreamPrepareOne(Dev& d, FastCtx fc, int q, int allowed, int capHint) {
  const FastK k = fastKRef(d);
  const DevCfg& c = d.cfg;
  QHot f = FL.hot[q];
  uniQHot(f);
  if (f.sLen > f.sPos) return 0;
  bool head = UNI32(FL.inHeap[q]) && f.gctx >= 0;
  int cap = allowed < QS_ONE_MAX ? allowed : QS_ONE_MAX;
  if (cap > capHint) cap = capHint;
  if (!(head && cap >= 1 && f.headFast && f.headKind == 1 && f.itStage == 1 && !f.itJobOnlyEv && !f.cordoned && f.burst >= 1 && f.tokens >= 1 && f.evApplied == f.evDone &&
        f.itQi >= 1 && f.itQi <= f.qEnd && UNI32(k.queuedJobs[f.itQi - 1]) == f.gctx)) return 0;
  int want = f.qEnd - (f.itQi - 1);
  if (want > cap) want = cap;
  if (!f.rateInf && f.tokens < (double)want) want = (int)f.tokens;
  if (fc.maxLookback != 0 && !f.itGangOnlyEv) {   // (fastStreamPrepare: element e >= 1 is peeked when itJobsSeen = seen + e - 1 < maxLookback)
    int64_t lim = (int64_t)fc.maxLookback - f.itJobsSeen + 1;
    if (lim < 1) lim = 1;
    if (want > lim) want = (int)lim;
  }
  if (want < 1) return 0;
  const int base = f.itQi - 1;
  const int skipUnf = fc.skipKnown && RS.numUnfeasible > 0;
  int64_t carry[MAXR];   // the queue's allocation (+ penalty) before the batch in hand
  for (int r = 0; r < MAXR; r++) carry[r] = r < c.R ? (int64_t)(UNI64(FL.qAlloc[q][r]) + UNI64(FL.qPenalty[q][r])) : 0;
  const double w = f.weight;
  EvKey* out = d.qsKey + (size_t)q * QS_CMAX;
  int len = 0; bool cut = false;
  for (int b = 0; b * 64 < want && !cut; b++) {
    // the first entry of this batch the stream cannot contain (B_QSSUM's barrier)
    int nb = want - b * 64 < 64 ? want - b * 64 : 64;
    FOR_LANES(x, 64) {
      int e = b * 64 + x, stop = 0;
      if (x < nb) {
        int job = k.queuedJobs[base + e];
        const int64_t* req = JREQ(d, job);
        stop = d.jGang[job] >= 0 || (e > 0 && skipUnf && k.unfeasible[d.jShape[job]]);
        for (int r = 0; r < c.R; r++) if (c.disallowed[r] && req[r] > 0) stop = 1;
      }
      FL.tmpQ[x] = stop;
    }
    for (int x = 0; x < nb; x++) if (UNI32(FL.tmpQ[x])) { nb = x; cut = true; break; }
    // entry e's allocation = carry + the requests of the batch's entries before it (integer adds: every lane sums for itself), its costs with B_QSKEYS' operations
    FOR_LANES(x, MAXR) FL.tmpX[x] = 0;
    FOR_LANES(x, 64) {
      if (x < nb) {
        int e = b * 64 + x;
        int64_t a[MAXR], with[MAXR];
        for (int r = 0; r < MAXR; r++) a[r] = carry[r];
        for (int m = b * 64; m < e; m++) { const int64_t* rq = JREQ(d, k.queuedJobs[base + m]); for (int r = 0; r < c.R; r++) a[r] += rq[r]; }
        int job = k.queuedJobs[base + e];
        const int64_t* req = JREQ(d, job);
        for (int r = 0; r < c.R; r++) { with[r] = a[r] + req[r]; if (req[r]) LDS_ADD64(FL.tmpX[r], (uint64_t)req[r]); }
        EvKey key;
        key.proposed = drf(d, with) / w; key.current = drf(d, a) / w; key.size = drf(d, req) * w;   // updatePQItem (queue_scheduler.go:636-686)
        key.pcPrio = c.pcPriority[d.jPc[job]]; key.job = job;
        out[e] = key;
      }
    }
    for (int r = 0; r < c.R; r++) carry[r] += (int64_t)UNI64(FL.tmpX[r]);
    len += nb;
  }
  FOR_LANES(x, QCAPF) FL.tmpQ[x] = 0;
  FOR_LANES(x, MAXR) FL.tmpX[x] = 0;
  if (len < 1) return 0;
  FAST_GLOBAL_FENCE();   // the keys are read back through global loads by this wave (streamKey)
  if (FLANE == 0) {
    FL.hot[q].sLen = len; FL.hot[q].sPos = 0; FL.hot[q].ewCount = 0; FL.hot[q].ewStart = 0; FL.sKind[q] = 0;
    d.qsLen[2 * q] = len;
    d.qsLen[2 * q + 1] = (!cut && base + want == d.queuedOff[q + 1]) ? 1 : 0;   // the queue's list ends where the stream ends (B_QSSTITCH)
    RS.statStreamPrepared += len;
  }
  LANE0_PUBLISHED();
  FAST_GLOBAL_FENCE();
  return len;
}

enum Cmd {
  CMD_PREPARE = 1, CMD_ROUND, CMD_QUEUES_ONLY, CMD_GANG_SCHEDULE, CMD_SELECT, CMD_SCHEDULE_MANY, CMD_BIND, CMD_EVICT, CMD_UNBIND,
  CMD_ADD_EVICTED, CMD_RESET_EVICTED, CMD_TXN_BEGIN, CMD_TXN_COMMIT, CMD_TXN_ABORT, CMD_FIT_BATCH, CMD_UPSERT_RESET, CMD_RESET_JOBS,
  CMD_PASS1, CMD_PASS2,   // the two sequential passes of a round whose data-parallel phases run as grid-wide kernels between them (asched_host.inc runRoundSplit)
  CMD_SUBMIT_CHECK,  // first command of the auxiliary kernel (k_control_aux, armada_sched_aux.hip)
  CMD_PQ_ORDER,
  CMD_NODE_UPSERT,
  CMD_ITERATE_NODES,
  CMD_MARKET,
  CMD_OPT_BEGIN, CMD_OPT_NEXT, CMD_OPT_APPLY, CMD_OPT_FAIL, CMD_OPT_END, CMD_OPT_TENT, CMD_OPT_TENT_UNDO,   // the fairness optimiser inside the round, gang by gang (asched_host.inc runOptimiserPhase)
  CMD_MARKET_QUEUES,  // QueueScheduler.Schedule of a market-driven pool (asched_schedule_queues after asched_set_market)
  CMD_MARKET_ROUND,   // a whole market-driven round (asched_set_market; round_mkt.h): PreemptingQueueScheduler.Schedule in one launch of the auxiliary kernel
};
#define CMD_AUX_FIRST CMD_SUBMIT_CHECK
// cmdIO layout: [0..15] results, [16..] arguments
#define ARG(k) d.cmdIO[16 + (k)]

DEV void setupPinned(Dev& d, int job, int pinned) {
  resetJctxForQueued(d, job);
  if (pinned >= 0) {
    d.jcEvicted[job] = 1; d.jcAssigned[job] = pinned;
    int g = d.jGang[job];
    d.jcGangCard[job] = g >= 0 ? d.gangOff[g + 1] - d.gangOff[g] : 1;
  }
}

DEV void runCommand(Dev& d, Ctl& c, int cmd) {
  const DevCfg& cf = d.cfg;
  switch (cmd) {
    case CMD_UPSERT_RESET:  // nodes_upsert: node state only
      d.rs->fastActive = 0;
      wgBulk(d, B_INIT_ALLOC, cf.N);
      wgBulk(d, B_KEYS_ALL, cf.N);
      break;
    case CMD_RESET_JOBS:
      wgBulk(d, B_RESET_JOBS, cf.M);
      d.rs->evictedTableSize = 0; d.rs->fairIndexValid = 0; d.rs->ftValid = 0; d.rs->txnActive = 0; d.rs->undoCount = 0;
      break;
    case CMD_PREPARE:
      d.rs->fastActive = 0;
      wgBulk(d, B_INIT_ALLOC, cf.N);
      wgBulk(d, B_RESET_JOBS, cf.M);
      wgBulk(d, B_POPULATE, cf.M);
      wgBulk(d, B_KEYS_ALL, cf.N);
      d.rs->evictedTableSize = 0; d.rs->fairIndexValid = 0; d.rs->ftValid = 0;
      wgBulk(d, B_CLEAR_UNFEASIBLE, cf.S);
      updateFairShares(d, (const double*)0);
      break;
    case CMD_ROUND: runRound(d, c); break;
    // The round as the host drives it by default: evictor filters, eviction, key rebuild, compaction, evicted-stream costs, unbind and the result
    // lists run as ordinary grid-wide kernels over all CUs (armada_sched.hip k_bulk / k_evict_apply / k_cmp_*); only the two inherently
    // sequential passes stay in this persistent kernel.  ARG(0) = number of jobs the evictor before this pass evicted.
    case CMD_PASS1: {  // pqs.go:149-166: schedule(evicted ++ queued); before it the tail of evict(): addEvictedJobsToNodeDb unless deferred
      long long t0 = CLK();
      int n1 = ARG(0);
      c.fastEvStatic = 1;
      if (!d.rs->replayPending && !(n1 > 0 && d.rs->evictedTableSize == n1)) { replayEvicted(d, c); wgBulk(d, B_EVIDX, n1); }   // (evictedTableSize == n1: the table was built by rank in front of this launch, replay_rank.h)
      long long t1 = CLK();
      c.skipEnter = fastOn(d, c) && d.evMono != nullptr && d.rs->lvl0NonNeg && n1 > 0;
      schedulePass(d, c, true, false, false);
      c.skipEnter = 0; c.fastEvStatic = 0;
      d.cmdIO[4] = d.rs->terminationReason;   // the round reports the first pass's reason (pqs.go:149-166)
      d.rs->statClk[1] += t1 - t0; d.rs->statClk[2] += CLK() - t1;
    } break;
    case CMD_PASS2: {  // pqs.go:199-221: schedule(evicted only; skipKeyCheck; compareSchedulingPriority)
      long long t0 = CLK();
      int n3 = ARG(0);
      if (!(n3 > 0 && d.rs->evictedTableSize == n3)) { replayEvicted(d, c); wgBulk(d, B_EVIDX, n3); }   // (== n3: built by rank in front of this launch, replay_rank.h)
      schedulePass(d, c, false, true, true);
      d.rs->statClk[4] += CLK() - t0;
    } break;
    case CMD_QUEUES_ONLY: {
      // no evicted jobs: empty per-queue evicted segments
      for (int q = 0; q <= cf.Q; q++) d.evOff[q] = 0;
      if (d.evCheap) wgBulk(d, B_EVKEYS_OFF, cf.Q);
      c.fastEvStatic = 1;
      d.rs->lvl0NonNeg = 1;
      wgBulk(d, B_LVL0, cf.N);
      schedulePass(d, c, true, false, false);
      d.cmdIO[2] = wgCompactIota(d, cf.M, d.inScheduled, d.resJob);
      d.cmdIO[3] = wgCompactIota(d, cf.M, d.jcPreempted, d.resPreJob);
      wgBulk(d, B_GATHER_SCHED, d.cmdIO[2]);
      wgBulk(d, B_GATHER_PRE, d.cmdIO[3]);
    } break;
    case CMD_GANG_SCHEDULE: {
      d.rs->apiDirty = 1;
      int n = ARG(0);
      int ref;
      if (n == 1 && d.jGang[ARG(1)] < 0) { ref = ARG(1); resetJctxForQueued(d, ref); }
      else {
        int g = d.jGang[ARG(1)];
        int64_t* tot = d.gangTotal + (size_t)g * cf.R;
        for (int r = 0; r < cf.R; r++) tot[r] = 0;
        for (int k = 0; k < n; k++) { int j = ARG(1 + k); resetJctxForQueued(d, j); d.gangArr[d.gangOff[g] + k] = j; vadd(d, tot, JREQ(d, j), +1); }
        d.gangSeen[g] = n; d.gangAllEvicted[g] = 0;
        ref = -(g + 2);
      }
      c.skipKeyCheck = 0;
      int reason = 0;
      bool ok = gangSchedule(d, c, ref, &reason, d.uniOff);
      d.cmdIO[0] = ok; d.cmdIO[1] = reason;
    } break;
    case CMD_SELECT: {
      d.rs->apiDirty = 1;
      int job = ARG(0);
      setupPinned(d, job, ARG(1));
      c.preCount = 0;
      selectNodeForJob(d, c, job);
      d.cmdIO[0] = c.preCount;
      c.preCount = 0;
    } break;
    case CMD_SCHEDULE_MANY: {
      d.rs->apiDirty = 1;
      int n = ARG(0);
      // members are passed explicitly: use the scratch gang slot G (one past the real gangs)
      int g = cf.G;
      int64_t* tot = d.gangTotal + (size_t)g * cf.R;
      for (int r = 0; r < cf.R; r++) tot[r] = 0;
      for (int k = 0; k < n; k++) {
        int j = ARG(1 + 2 * k), pin = ARG(2 + 2 * k);
        if (!(pin >= 0 && d.evIndexOfJob[j] >= 0)) setupPinned(d, j, pin);  // reuse the evicted jctx when the job is in the evicted table
        d.gangArr[d.gangOff[g] + k] = j;
      }
      d.gangSeen[g] = n;
      c.preCount = 0;
      if (!c.txn.active) d.rs->undoCount = 0;
      bool ok = scheduleMany(d, c, -(g + 2));
      if (!c.txn.active) d.rs->undoCount = 0;
      d.cmdIO[0] = ok; d.cmdIO[1] = c.preCount;
      for (int i = 0; i < c.preCount; i++) d.jcStagedBy[c.preList[i]] = -1;
      c.preCount = 0;
    } break;
    case CMD_BIND: {
      d.rs->apiDirty = 1;
      int job = ARG(0), n = ARG(1), prio = bindPriority(d, ARG(0), ARG(2));
      if (addJob(d, n, job, cutoffFor(d, job, prio), c.txn.active) == 0) {
        d.schedAtPrio[job] = prio;
        updateKeysCtl(d, n);
        int e = d.evIndexOfJob[job];
        if (e >= 0) evTabDelete(d, e, c.txn.active);
      }
    } break;
    case CMD_EVICT: {
      d.rs->apiDirty = 1;
      int job = ARG(0), n = ARG(1);
      if (d.schedAtPrio[job] == NO_PRIORITY) { raise(d, ASCHED_ERR_INTERNAL, 801); break; }
      if (evictJobOnNode(d, n, job) == 0) updateKeysCtl(d, n);
    } break;
    case CMD_UNBIND: {
      d.rs->apiDirty = 1; removeJob(d, ARG(1), ARG(0), c.txn.active); updateKeysCtl(d, ARG(1)); } break;
    case CMD_ADD_EVICTED: {
      d.rs->apiDirty = 1;
      int idx = ARG(0), job = ARG(1), n = ARG(2);
      if (d.evIndexOfJob[job] >= 0) { raise(d, ASCHED_ERR_INTERNAL, 802); break; }
      setupPinned(d, job, n);
      evTabInsert(d, idx, job);
    } break;
    case CMD_RESET_EVICTED:
      d.rs->apiDirty = 1;
      wgBulk(d, B_RESET_EVTAB, cf.M);
      d.rs->evictedTableSize = 0; d.rs->fairIndexValid = 0; d.rs->ftValid = 0;
      break;
    case CMD_TXN_BEGIN: txnBegin(d, c.txn); break;
    case CMD_TXN_COMMIT: txnCommit(d, c.txn); break;
    case CMD_TXN_ABORT: txnAbort(d, c.txn); break;
  }
}

// Commands of the submit check.  They run in their own kernel, in their own code object (k_control_aux,
// armada_sched_aux.hip), so that the round kernel's code — everything above is inlined into it — stays exactly what was measured.
struct SubmitArgs { int32_t nu, pad; int32_t* off; int32_t* jobs; int32_t* flags; int32_t* out; };  // at cmdIO + 16, written by the host
struct PqOrderArgs { int32_t n, preferLarge, compareSchedPrio, homeFirst; int32_t* out; const int32_t* away; };  // at cmdIO + 16; out = [n] order, then 1 flag
// ---- the experimental fairness optimiser inside the round (preempting_queue_scheduler.go:224-253, 666-710).  OptimisingQueueScheduler.Schedule
// (optimising_queue_scheduler.go:58-180) is driven gang by gang from the host: the candidate iteration, the checks and the bookkeeping run here (the round's own
// generic code), the node scoring of every member is the wide kernel k_opt_score launched by the host in between.
struct OptLoopArgs { int32_t first, hasMinSize, pad0, pad1; int64_t current[MAXR], maxToSchedule[MAXR], minSize[MAXR]; };
// createCandidateGangIterator (:200-226): queued jobs only, of the queues below their demand-capped adjusted fair share; ClearUnfeasibleSchedulingKeys (pqs.go:692)
DEV void optBegin(Dev& d, Ctl& c) {
  const DevCfg& cf = d.cfg;
  wgBulk(d, B_CLEAR_UNFEASIBLE, cf.S);
  d.rs->numUnfeasible = 0;
  wgBulk(d, B_RESET_GANGSEEN, cf.G);
  c.skipKeyCheck = 0; c.compareSchedPrio = 0; c.useReplayAlloc = 0; c.onlyEvicted = 0;
  d.rs->optMode = 1;
  { int l0 = CTL_WAVE() ? CTL_LANE() : 0, st = CTL_WAVE() ? 64 : 1; for (int j = l0; j < cf.M; j += st) { d.optSched[j] = 0; d.optPre[j] = 0; d.optGhost[j] = -1; } }
  PassCfg pc{true, cf.maxLookback, true};
  for (int q = 0; q < cf.Q; q++) {
    d.itEi[q] = d.evOff[q + 1]; d.itQi[q] = d.queuedOff[q]; d.itStage[q] = 1; d.itJobsSeen[q] = 0; d.itNext[q] = -1; d.itStashed[q] = -1;
    d.itJobOnlyEv[q] = 0; d.itGangOnlyEv[q] = 0; d.onlyEvByQueue[q] = 0; d.pqInHeap[q] = 0; d.pqGctx[q] = -1;
    d.pqBudget[q] = d.qDc[q] / d.qWeight[q];
  }
  for (int q = 0; q < cf.Q; q++) {
    int64_t a[MAXR];
    for (int r = 0; r < cf.R; r++) a[r] = QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r];
    if (drf(d, a) >= d.qDc[q]) continue;           // at or above its fair share: no iterator
    updateAndPush(d, c, q, pc);
  }
}
// the loop of Schedule up to the point where a gang goes to the gang scheduler: cmdIO[0] = 1 + the gang (cmdIO[1] members in preList, [2] queue, [3] uniformity slot), or 0 = done
DEV void optNext(Dev& d, Ctl& c, const OptLoopArgs& a) {
  const DevCfg& cf = d.cfg;
  PassCfg pc{true, cf.maxLookback, true};
  bool first = a.first != 0;
  for (;;) {
    d.cmdIO[0] = 0;
    if (d.rs->error) return;
    if (!first) costItClear(d, c, pqTop(d, c), pc);   // (the cached keys have not changed since the Peek: the same top)
    first = false;
    int top = pqTop(d, c);
    int ref = top >= 0 ? d.pqGctx[top] : -1;
    if (ref == -1) return;
    int cnt = gcCount(d, ref);
    if (cnt == 0) continue;
    if (gcAllEvicted(d, ref)) continue;
    int q = gcQueue(d, ref);
    if (d.pqProposed[top] > d.qDc[q] / d.qWeight[q]) continue;                          // :99-101
    bool skip = false;
    for (int k = 0; k < cnt && !skip; k++) {
      int j = gcJob(d, ref, k);
      if (d.jobFlags[j] & F_SUCCESSFUL) skip = true;                                      // scheduled earlier in this round (:104-108)
      if (a.hasMinSize) for (int r = 0; r < cf.R; r++) if (a.minSize[r] > JREQ(d, j)[r]) skip = true;   // minimumJobSizeToSchedule.Exceeds(job) (:109-112)
    }
    if (skip) continue;
    const int64_t* tot = gcTotal(d, ref);
    for (int r = 0; r < cf.R; r++) if (a.current[r] + tot[r] > a.maxToSchedule[r]) skip = true;   // :115-117
    if (skip) continue;
    // checkIfWillBreachSchedulingLimits (:228-256): a failing constraint returns BEFORE EvictGang — the gang stays in the scheduling context (the reference's TODO)
    int r = checkRound(d);
    if (!r) {
      for (int k = 0; k < cnt; k++) d.jcReason[gcJob(d, ref, k)] = 0;   // the fresh job contexts (no unschedulable reason) replace the earlier attempt's (dev.h optMode)
      sctxAddGang(d, ref); r = checkJob(d, ref); if (!r) r = checkFloating(d, ref); if (!r) sctxEvictGang(d, ref);
    }
    if (r) {
      if (isTerminal(r)) return;
      if (isQueueTerminal(r)) costItOnlyEvictedForQueue(d, c, q, pc);
      continue;
    }
    if (cancelRequested(d)) return;                                                       // hasContextExpired
    int j0 = gcJob(d, ref, 0);
    for (int k = 0; k < cnt; k++) d.preList[k] = gcJob(d, ref, k);
    d.cmdIO[1] = cnt; d.cmdIO[2] = q; d.cmdIO[3] = d.jGang[j0] >= 0 ? d.jGangUni[j0] : -1;
    d.cmdIO[0] = 1;
    return;
  }
}
// sctx.PreemptJob + qctx.preemptJob (context/scheduling.go:530-549, context/queue.go:322-349)
DEV void sctxPreemptJob(Dev& d, int job) {
  int q = d.jQueue[job], pc = d.jPc[job];
  const int64_t* req = JREQ(d, job);
  uint8_t f = d.jobFlags[job];
  bool sched = f & F_SUCCESSFUL;
  if (sched) { vadd(d, QPV(d.qSchedByPc, q, pc), req, -1); f &= ~F_SUCCESSFUL; }
  f &= ~F_RESCHEDULED;
  vadd(d, QPV(d.qAllocByPc, q, pc), req, -1);
  vadd(d, QV(d.qAlloc, q), req, -1);
  if (sched) { vadd(d, d.rs->scheduled, req, -1); d.rs->numScheduledJobs--; }
  vadd(d, d.rs->allocated, req, -1);
  d.jobFlags[job] = f;
}
// markJobsScheduledAndPreempted (optimiser/gang_scheduler.go:192-254) + the rate limiters (optimising_queue_scheduler.go:150-154) + the PQS result sets (pqs.go:232-249).
// ARG: members, queue, then per member: job, node, number of victims, the victims
DEV void optApply(Dev& d, Ctl& c) {
  const DevCfg& cf = d.cfg;
  (void)c;
  int cnt = ARG(0), q = ARG(1), at = 2;
  for (int m = 0; m < cnt; m++) {
    int job = ARG(at), n = ARG(at + 1), npre = ARG(at + 2);
    // A job scheduled earlier in this round, evicted by the oversubscribed evictor and not rescheduled still holds (evicted) resources on its node until the
    // unbinding at the end of the round; the reference would now bind it on a second node as well.  This backend keeps one node per job: refused, not approximated
    // In the reference the job is then in TWO nodes' AllocatedByJobId.  Here the old node keeps the job's evicted resources in its planes (nothing is touched) and
    // remembers it as a ghost (dev.h optGhost): the scoring kernels list it there, a later victim selection may take it, optEnd gives the resources back.
    if (d.jobNode[job] >= 0 && d.jobNode[job] != n) {
      if (!d.jobEvictedOnNode[job] || d.optGhost[job] >= 0) { raise(d, ASCHED_ERR_INTERNAL, 300); return; }   // "job already has resources allocated" (node.go:416-442)
      d.optGhost[job] = d.jobNode[job];
      d.jobNode[job] = -1; d.jobEvictedOnNode[job] = 0;
    }
    for (int i = 0; i < npre; i++) {                                                  // UnbindJobsFromNode
      int v = ARG(at + 3 + i);
      if (d.optGhost[v] == n) { markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, JREQ(d, v), +1); d.optGhost[v] = -1; }   // RemoveJob of the evicted copy on this node (node.go:480-506)
      else removeJob(d, n, v, false);
    }
    int32_t prio = bindPriority(d, job, cf.pcPriority[d.jPc[job]]);
    if (addJob(d, n, job, cutoffFor(d, job, prio), false)) return;                  // BindJobToNode
    d.schedAtPrio[job] = prio;
    updateKeysCtl(d, n);
    for (int i = 0; i < npre; i++) {
      int v = ARG(at + 3 + i);
      sctxPreemptJob(d, v);
      if (d.optPre[v] < 255) d.optPre[v]++;                                            // (merged into the round's sets by optEnd)
      d.preemptedNode[v] = n;
    }
    d.jcHasPctx[job] = 1; d.pcNode[job] = n; d.pcSap[job] = cf.pcPriority[d.jPc[job]]; d.pcPap[job] = ASCHED_MIN_PRIORITY; d.pcMethod[job] = ASCHED_METHOD_OPTIMISER;
    d.jcReason[job] = 0;
    sctxAddJob(d, job);
    if (d.optSched[job] < 255) d.optSched[job]++;
    at += 3 + npre;
  }
  reserveN(&d.rs->globalTokens, d.rs->globalBurst, d.rs->globalRateInf, cnt);
  reserveN(&d.qTokens[q], d.qBurst[q], d.qRateInf[q], cnt);
}
// updateState (optimiser/gang_scheduler.go:148-190) for one placed member, tentatively: the victims leave the node, the member is bound — inside the device
// transaction the host has opened (the reference works on node copies).  What the undo log cannot express — a ghost made or consumed (dev.h optGhost) — is reported
// to the host, which replays it backwards after the abort (optTentUndo).  ARG: job, node, number of victims, the victims.  cmdIO[0] = the member's old node if it
// became a ghost there (else -1), cmdIO[1] = number of ghost victims, cmdIO[2..] those victims
DEV void optTent(Dev& d, Ctl& c) {
  int job = ARG(0), n = ARG(1), npre = ARG(2), ng = 0;
  d.cmdIO[0] = -1;
  if (d.jobNode[job] >= 0 && d.jobNode[job] != n) {
    if (!d.jobEvictedOnNode[job] || d.optGhost[job] >= 0) { raise(d, ASCHED_ERR_INTERNAL, 300); return; }
    d.cmdIO[0] = d.jobNode[job];
    d.optGhost[job] = d.jobNode[job]; d.jobNode[job] = -1; d.jobEvictedOnNode[job] = 0;
  }
  for (int i = 0; i < npre; i++) {
    int v = ARG(3 + i);
    if (d.optGhost[v] == n) { markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, JREQ(d, v), +1); d.optGhost[v] = -1; d.cmdIO[2 + ng++] = v; }
    else removeJob(d, n, v, c.txn.active);
  }
  d.cmdIO[1] = ng;
  int32_t prio = bindPriority(d, job, d.cfg.pcPriority[d.jPc[job]]);
  if (addJob(d, n, job, cutoffFor(d, job, prio), c.txn.active)) return;
  d.schedAtPrio[job] = prio;                         // (NodeDb.scheduledAtPriorityByJobId: one map for every node copy, not rolled back)
  updateKeysCtl(d, n);
}
// ARG: job, its old node (-1: none), node, number of ghost victims, the victims — after the transaction's abort
DEV void optTentUndo(Dev& d) {
  int job = ARG(0), old = ARG(1), n = ARG(2), ng = ARG(3);
  for (int i = 0; i < ng; i++) { int v = ARG(4 + i); d.optGhost[v] = n; markAllocatable(d, n, ASCHED_EVICTED_PRIORITY, JREQ(d, v), -1); }
  if (ng) updateKeysCtl(d, n);
  if (old >= 0) { d.optGhost[job] = -1; d.jobNode[job] = old; d.jobEvictedOnNode[job] = 1; }
}
// The optimiser's ScheduledJobs and PreemptedJobs lists merged into the round's sets, scheduled jobs first (pqs.go:232-249): a scheduled job leaves the preempted set or
// enters the scheduled one; each preemption takes its job out of the scheduled set if it is there ("scheduled and preempted in the same round") and puts it into the
// preempted set otherwise — so the second preemption of the same job does.
DEV void optEnd(Dev& d) {
  int l0 = CTL_WAVE() ? CTL_LANE() : 0, st = CTL_WAVE() ? 64 : 1;
  for (int j = l0; j < d.cfg.M; j += st) {
    if (d.optSched[j]) { if (d.inPreempted[j]) d.inPreempted[j] = 0; else d.inScheduled[j] = 1; }
    int k = d.optPre[j];
    if (k) {
      if (d.inScheduled[j]) { d.inScheduled[j] = 0; if (k >= 2) d.inPreempted[j] = 1; }
      else d.inPreempted[j] = 1;
    }
    int g = d.optGhost[j];   // still on its old node as an evicted job: the unbinding of the round's end (pqs.go:775-798, the job is in scheduledAndEvicted) takes it off there
    if (g >= 0) { atomicMarkAllocatable(d, g, ASCHED_EVICTED_PRIORITY, JREQ(d, j), +1); d.optGhost[j] = -1; }
  }
  d.rs->optMode = 0;
}
// updateUnfeasibleSchedulingKeys (optimising_queue_scheduler.go:258-277).  ARG: job (a single-job gang), reason.  The registered jctx carries no unschedulable reason
// (FairnessOptimisingGangScheduler.Schedule returns it without calling jctx.Fail): a later job skipped for this key is added as if scheduled (:398-413) — restated as it is
DEV void optFail(Dev& d, Ctl& c) {
  int j = ARG(0), reason = ARG(1);
  if (!c.skipKeyCheck && isPropertyOfGang(reason) && keyValid(d, j)) {
    int s = d.jShape[j];
    if (!d.unfeasible[s]) { d.unfeasible[s] = 1; d.unfeasibleReason[s] = 0; d.rs->numUnfeasible++; }
  }
}

DEV void runAuxCommand(Dev& d, Ctl& c, int cmd) {
  const DevCfg& cf = d.cfg;
  switch (cmd) {
#ifdef ASCHED_MARKET_ROUND
    case CMD_MARKET_ROUND: runRound(d, c); break;   // MKS.market is set: every market-specific step is behind mkOn(d) (round_mkt.h)
    case CMD_MARKET_QUEUES: {   // QueueScheduler.Schedule on a market-driven pool (queue_scheduler.go:73-74, 176-203): CMD_QUEUES_ONLY's steps, here because the market code lives in this kernel
      for (int q = 0; q <= cf.Q; q++) d.evOff[q] = 0;
      if (d.evCheap) wgBulk(d, B_EVKEYS_OFF, cf.Q);
      c.fastEvStatic = 1;
      d.rs->lvl0NonNeg = 1;
      wgBulk(d, B_LVL0, cf.N);
      schedulePass(d, c, true, false, false);
      d.cmdIO[2] = wgCompactIota(d, cf.M, d.inScheduled, d.resJob);
      d.cmdIO[3] = wgCompactIota(d, cf.M, d.jcPreempted, d.resPreJob);
      wgBulk(d, B_GATHER_SCHED, d.cmdIO[2]);
      wgBulk(d, B_GATHER_PRE, d.cmdIO[3]);
    } break;
#endif
    case CMD_PQ_ORDER: {
      // sort.Sort over QueueCandidateGangIteratorPQ.Less (queue_scheduler.go:738-798) — the float goldens of queue_scheduler_test.go:995-1164.
      // The items sit in the per-queue arrays the round uses (the host points d.pq* / d.qNameRank at a scratch copy for this launch),
      // so this is the round's own pqLess; the packed key of the fast path (packKey3 / packedLess) is checked pair by pair against it.
      PqOrderArgs a = *(const PqOrderArgs*)(d.cmdIO + 16);
      c.preferLarge = a.preferLarge; c.compareSchedPrio = a.compareSchedPrio;
      int ord[64];
      for (int i = 0; i < a.n; i++) ord[i] = i;
      for (int i = 1; i < a.n; i++) {   // insertion sort: Less is a strict total order, the result is the unique sorted order
        int x = ord[i], k = i - 1;
        while (k >= 0 && ((a.homeFirst && a.away[x] != a.away[ord[k]]) ? !a.away[x] : pqLess(d, c, x, ord[k]))) { ord[k + 1] = ord[k]; k--; }   // :744-746: home before away
        ord[k + 1] = x;
      }
      for (int i = 0; i < a.n; i++) a.out[i] = ord[i];
      int agrees = 1;
      for (int x = 0; x < a.n; x++) for (int y = 0; y < a.n; y++) {
        if (x == y || (a.homeFirst && a.away[x] != a.away[y])) continue;   // (the packed key orders within a group)
        PackedKey kx = packKey3(a.preferLarge, a.compareSchedPrio ? d.pqSchedPrio[x] : d.pqPcPrio[x], d.pqProposed[x], d.pqCurrent[x], d.pqSize[x], d.pqBudget[x]);
        PackedKey ky = packKey3(a.preferLarge, a.compareSchedPrio ? d.pqSchedPrio[y] : d.pqPcPrio[y], d.pqProposed[y], d.pqCurrent[y], d.pqSize[y], d.pqBudget[y]);
        if (packedLess(kx, (uint32_t)d.qNameRank[x], ky, (uint32_t)d.qNameRank[y]) != pqLess(d, c, x, y)) agrees = 0;
      }
      a.out[a.n] = agrees;
    } break;
    case CMD_OPT_BEGIN: optBegin(d, c); break;
    case CMD_OPT_NEXT: { OptLoopArgs a = *(const OptLoopArgs*)(d.cmdIO + 16); optNext(d, c, a); } break;
    case CMD_OPT_APPLY: optApply(d, c); break;
    case CMD_OPT_FAIL: optFail(d, c); break;
    case CMD_OPT_END: optEnd(d); break;
    case CMD_OPT_TENT: optTent(d, c); break;
    case CMD_OPT_TENT_UNDO: optTentUndo(d); break;
    case CMD_MARKET: {   // market-driven ordering's iterators (round_market.h)
      MarketArgs a = *(const MarketArgs*)(d.cmdIO + 16);
      if (a.op == 0) marketIterate(a);
      else if (a.op == 1) a.out[0] = marketCompare(a.l1[0], a.l2[0]);
      else marketMultiIterate(a);
    } break;
    case CMD_ITERATE_NODES: {
      // NodeTypesIterator (nodeiteration.go:74-149) over the given node types at one priority, materialised: the test hook behind the ordering goldens of
      // nodeiteration_test.go.  The iterators are the literal restatement the round uses off the index grid (round_ctl.h litAdvance / litNodeLess): one
      // NodeTypeIterator per type, the heap as an argmin under nodeTypesIteratorPQ.less, the popped iterator advanced before its node is returned (:134-149).
      // ARG: level, number of types, output capacity, -, K x (request lo, hi), then one mask row index per type (rows of d.typeMask the host built for the call)
      int level = ARG(0), nT = ARG(1), cap = ARG(2);
      if (nT > LIT_TMAX) { raise(d, ASCHED_ERR_UNSUPPORTED, 511); break; }
      int64_t ireq[MAXK];
      for (int i = 0; i < MAXK; i++) ireq[i] = i < cf.K ? (int64_t)(((uint64_t)(uint32_t)ARG(4 + 2 * i + 1) << 32) | (uint32_t)ARG(4 + 2 * i)) : 0;
      for (int k = 0; k < nT; k++) {
        LitIt& it = d.lit[k];
        it.type = ARG(4 + 2 * MAXK + k);
        for (int i = 0; i < MAXK; i++) it.lb[i] = ireq[i];
        LIT_SET_BOUND(cf, it);
        litAdvance(d, level, it, ireq);
      }
      int n = 0;
      for (;;) {
        if (d.rs->error) break;
        int best = -1;
        for (int k = 0; k < nT; k++) if (d.lit[k].head >= 0 && (best < 0 || litNodeLess(d, level, d.lit[k].head, d.lit[best].head))) best = k;
        if (best < 0) break;
        int node = d.lit[best].head;
        litAdvance(d, level, d.lit[best], ireq);
        if (n < cap) d.nodeOver[n] = node;
        n++;
        if (n > cf.N) { raise(d, ASCHED_ERR_INTERNAL, 512); break; }   // "iteration loop detected" (nodeiteration.go:330-336)
      }
      d.cmdIO[0] = n;
    } break;
    case CMD_NODE_UPSERT: {  // UpsertWithTxn of one node (nodedb.go:1164-1175): new AllocatableByPriority, every order key rebuilt
      d.rs->apiDirty = 1;
      int n = ARG(0);
      for (int l = 0; l < cf.P; l++) for (int r = 0; r < cf.R; r++) {
        int k = l * cf.R + r;
        AL(d, l, r, n) = (int64_t)(((uint64_t)(uint32_t)ARG(2 + 2 * k) << 32) | (uint32_t)ARG(1 + 2 * k));
      }
      updateKeysCtl(d, n);
    } break;
    case CMD_SUBMIT_CHECK: {
      // SubmitChecker.getSchedulingResult, per-pool core (submitcheck.go:342-371), for a whole batch of units in one launch:
      // copyGangContext (fresh jctxs), nodeDb.Txn, ScheduleManyWithTxn, txn.Abort — every unit meets the same NodeDb state
      d.rs->apiDirty = 1;
      SubmitArgs a = *(const SubmitArgs*)(d.cmdIO + 16);
      int g = cf.G;
      if (c.txn.active) { raise(d, ASCHED_ERR_INVALID, 810); break; }
      for (int u = 0; u < a.nu && !d.rs->error; u++) {
        int o0 = a.off[u], n = a.off[u + 1] - o0;
        // individual check: the job is re-made with job.WithGangInfo(BasicJobGangInfo()) (:274) — for the duration of the unit the
        // job's gang id reads "none" wherever the device code asks IsInGang (away scheduling :613, gang cardinality)
        bool strip = (a.flags[u] & ASCHED_SUBMIT_STRIP_GANG) != 0;
        int j0 = a.jobs[o0];
        int savedGang = d.jGang[j0];
        if (strip) { if (n != 1) { raise(d, ASCHED_ERR_INVALID, 811); break; } d.jGang[j0] = -1; }
        for (int k = 0; k < n; k++) {
          int j = a.jobs[o0 + k];
          setupPinned(d, j, -1);
          d.gangArr[d.gangOff[g] + k] = j;
        }
        d.gangSeen[g] = n;
        c.preCount = 0;
        txnBegin(d, c.txn);
        bool ok = scheduleMany(d, c, -(g + 2));
        txnAbort(d, c.txn);
        unstagePreemptions(d, c);
        if (strip) d.jGang[j0] = savedGang;
        int ns = 0;
        for (int k = 0; k < n; k++) { int j = a.jobs[o0 + k]; if (d.jcHasPctx[j] && d.pcNode[j] >= 0) ns++; }  // pctx.IsSuccessful
        a.out[4 * u] = ok; a.out[4 * u + 1] = d.jcHasPctx[j0] && d.pcMethod[j0] == ASCHED_METHOD_AWAY;
        a.out[4 * u + 2] = ns; a.out[4 * u + 3] = d.jcHasPctx[j0] ? d.pcNode[j0] : -1;
      }
    } break;
  }
}

// body of the control kernel's wave 0 (and of the hostsim debug build): one command with the fast path's LDS side
// restored from / saved to HBM around it
DEV void controlMainAux(Dev& d, int cmd) {
  Ctl c;
  c.txn.active = d.rs->txnActive; c.fairStamp = d.rs->fairStamp; c.preList = d.preList; c.preCount = 0;
  c.skipKeyCheck = 0; c.compareSchedPrio = 0; c.preferLarge = d.cfg.preferLarge; c.useReplayAlloc = 0; c.onlyEvicted = 0;
  c.fastEnabled = 0; c.fastEvStatic = 0; c.l1Dirty = 0; c.fqLive = 0; c.skipEnter = 0; c.skipActive = 0; c.cancelSeen = 0; c.fpLimitHit = 0; c.streamNextAt = 0; c.streamBackoff = 0; c.streamCap = QS_CMAX;
  d.rs->ftValid = 0; d.rs->ftWanted = 0;
  fastLoad(d);
  runAuxCommand(d, c, cmd);
  fastEnterGeneric(d, c);
  fastSave(d);
  d.rs->txnActive = c.txn.active; d.rs->fairStamp = c.fairStamp;
}
DEV void controlMain(Dev& d, int cmd) {
  Ctl c;
  c.txn.active = d.rs->txnActive; c.fairStamp = d.rs->fairStamp; c.preList = d.preList; c.preCount = 0;
  c.skipKeyCheck = 0; c.compareSchedPrio = 0; c.preferLarge = d.cfg.preferLarge; c.useReplayAlloc = 0; c.onlyEvicted = 0;
  c.fastEnabled = d.f.iterOk && !d.rs->apiDirty && (cmd == CMD_ROUND || cmd == CMD_QUEUES_ONLY || cmd == CMD_PASS1 || cmd == CMD_PASS2);
  if (d.jAway && (cmd == CMD_PASS2 || cmd == CMD_QUEUES_ONLY)) c.fastEnabled = 0;   // cross-pool away jobs can sit in the queues of these passes (the oversubscribed evictor takes them; a caller's queue-only pass): generic
  c.fastEvStatic = 0; c.l1Dirty = 0; c.fqLive = 0; c.skipEnter = 0; c.skipActive = 0; c.cancelSeen = 0; c.fpLimitHit = 0; c.streamNextAt = 0; c.streamBackoff = 0; c.streamCap = QS_CMAX;
  // the fair-share threshold table (round_ft.h) lives for one launch of a scheduling pass: the grid-wide phases between launches rewrite planes wholesale
  d.rs->ftValid = 0;
  d.rs->ftWanted = d.ftT != nullptr && !d.rs->apiDirty && (cmd == CMD_ROUND || cmd == CMD_QUEUES_ONLY || cmd == CMD_PASS1 || cmd == CMD_PASS2);
  fastLoad(d);
  runCommand(d, c, cmd);
  fastEnterGeneric(d, c);
  fastSave(d);
  d.rs->txnActive = c.txn.active; d.rs->fairStamp = c.fairStamp;
}
