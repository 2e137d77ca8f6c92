// round_wide.h — wide runs: the stream run (round_fast.h) for pools of MORE than QCAPF queues.
//
// Why.  The fast iteration keeps one queue per lane of the control wave and 1.1 KB of per-queue state in LDS: beyond 64 queues a round used to run on the generic
// code alone, one iteration per job at ~86 k clocks with every queue's state behind HBM pointers — slower than one host core (256 queues 0.59x).
// CostBasedCandidateGangIterator itself has no queue limit (queue_scheduler.go:449-699), and production pools hold hundreds of queues.
//
// What.  While the heads of the queues are single jobs that need no preemption — phase-1-evicted jobs returning to their nodes (nodedb.go:897-906) and queued
// jobs that fit at priority -2 — the order in which QueueCandidateGangIteratorPQ serves them is a function of every queue's OWN allocation prefix (the Less
// inputs of queue_scheduler.go:636-686, 738-798), not of anything the node side produces.  So per queue the next <= WIDE_L entries are laid out ahead — the rest
// of its gang-free eviction list with the costs B_EVKEYS computed (round_run.h), then its next single queued jobs with the costs of the chunked prefix passes
// B_QSSUM / B_QSSTITCH / B_QSKEYS (the same float64 operations as updatePQItem on the same sums) — and the k-way heap merge over those sequences is computed as a
// BULK RANK on the helper workgroups: an entry's position = the number of entries that order before it = its index in its own queue + for every other queue a
// binary search in that queue's sequence.  A heap merge serves a queue's entries in list order whatever their keys, so the sequences are compared by their
// running-maximum keys (the argument of round_fast.h "skip mode": an entry is never served before its predecessor), ties broken like Less by the queue's name
// rank.  The control wave then only STAGES the merged order into the ring; the node engine and the bind wave place the queued jobs exactly as in a stream run
// (first fit at priority -2 through the sorted base + L0; evicted entries need no node work: DESIGN.md 3.1 item 2); when the run ends the entries that were
// executed are committed per queue (one thread per queue: accounting, the evicted jobs' return, iterator cursors, tokens) and the generic updatePQItem
// produces every touched queue's next head from the state the reference would have at that point.
//
// Where a run must stop — BEFORE any side effect that would have to be taken back:
//   * a queue in the heap whose head the run cannot serve (a gang, a job that is pinned / preempted-marked, an evicted job of a list with gangs ...): its
//     current key is an entry of the merge ("barrier"); nothing that orders after it is executed;
//   * a queue that goes on behind its last prepared entry under a key that is not known here (the cap, a gang member, a known-unfeasible scheduling key, a
//     disallowed request, the queue's rate-limit tokens, the lookback limit): the run ends with that entry ("open");
//   * the global rate limiter has no token for another new job; the engine finds no node for an entry (that entry and everything after it are forgotten: no
//     accounting had been done for them; the generic cascade decides about the job); the caller's cancel word.
// Everything else (per-class resource caps, round resource limits, floating resources, soft time budgets, cross-pool away jobs, market mode, the pass over
// evicted jobs only ordered by scheduled-at priority) keeps the pool on the generic path: asched_host.inc decides (FastCfg.iterOk == 2).
//
// Exactness rests on the same three facts as the stream run: the keys are the ones updatePQItem would compute (same operations, same order); the packed key
// orders like Less for finite non-negative costs; integer accounting is order independent.  tests/test_z_wide_runs.py and `tests/soak.py wide` compare whole
// rounds with the oracle for 65 ... 1 024 queues (CPU build and -m gpu).
#pragma once

enum { W_SEG = 0, W_EVSUM, W_QSUM, W_STITCH, W_PACK, W_FIX, W_RANK, W_SCATTER, W_COMMIT_E, W_COMMIT_Q, W_EXCL, W_SKIP_FIND, W_SKIP_MARK, W_EXCLF };   // (W_EXCL: not a pass of a wide run — the pass behind asched_excluded_nodes shares the op)
#define W_PER 16                        // queues one item of the rank pass walks for its entry: an entry's walk over the other queues is cut into Q / W_PER items (the entries of a run
                                        // alone fill a third of the lanes once)
#define WQ_CHUNK 8                      // entries one item of the chunked passes covers (streams here are tens of entries long, not thousands: round_run.h uses 64)
#define WQ_CPQ (WIDE_L / WQ_CHUNK)
#define W_RG 8                          // binary searches of the rank pass that run side by side (independent loads in flight together)

DEV bool wideKeyLess(const WideKey& a, const WideKey& b) { return a.a != b.a ? a.a < b.a : a.x != b.x ? a.x < b.x : a.y < b.y; }
DEV bool wideKeyEq(const WideKey& a, const WideKey& b) { return a.a == b.a && a.x == b.x && a.y == b.y; }
DEV WideKey widePack(int preferLarge, const EvKey& e, double budget) {
  PackedKey p = packKey3(preferLarge, e.pcPrio, e.proposed, e.current, e.size, budget);
  WideKey k; k.a = p.A; k.x = p.X; k.y = p.Y;
  return k;
}

// ---- the bulk bodies (one element per call: every workgroup of the launch takes a share; they read and write HBM only and never the scheduling-context
// scalars, which live in the control workgroup's LDS for the launch — what they need of those comes through WideParams)
// ---- excluded nodes (asched_excluded_nodes; NumExcludedNodesByReason, nodedb.go:566-580,881-928).  One pass over the nodes for a selection that found no node at the job's
// priority: element i = node i.  The iterator of the attempt yields exactly the nodes whose INDEXED columns cover the request at that level (nodeiteration.go:318-382, for
// requests on the index grid; rows on the literal path walk the iterators instead, exclLiteral below) — a bit per node; the host intersects it with the node types the job
// matches.  A yielded node that passes the static checks (the row's mask) fails the dynamic one, since the attempt found nothing: its reason is the first column, in factory
// order, that the allocatable at that level does not cover (resource_list.go:167-191) — an arena entry.  Every other reason is a function of the inputs alone: the host's.
DEV void exclPutBits(uint64_t* words, int n, bool bit) {
#ifdef ASCHED_HOSTSIM
  if ((n & 63) == 0) words[n >> 6] = 0;            // (the serial build visits the nodes in ascending order)
  if (bit) words[n >> 6] |= 1ull << (n & 63);
#else
  unsigned long long m = __ballot(bit);            // (the 64 nodes of a word are the 64 lanes of one wave: the pass runs over W * 64 elements)
  if ((threadIdx.x & 63) == 0) words[n >> 6] = m;
#endif
}
DEV void exclDynPut(Dev& d, ExclDev& x, int slot, int n, int level, const int64_t* req) {
  int res = -1; int64_t avail = 0;
  for (int r = 0; r < d.cfg.R; r++) if (res < 0 && req[r] > AL(d, level, r, n)) { res = r; avail = AL(d, level, r, n); }
  int at = atomicFetchAddI32(&x.dynCount, 1);
  if (res < 0 || at >= x.dynCap) { atomicOrI32(&x.rec[slot].flags, res < 0 ? 2 : 1); return; }   // (bit 1: a yielded, statically matching node that fits — the attempt would have taken it)
  ExclDyn e; e.slot = slot; e.node = n; e.res = res; e.pad = 0; e.avail = avail;
  x.dyn[at] = e;
}
DEV void exclBulk(Dev& d, int n) {
  const DevCfg& c = d.cfg;
  ExclDev& x = *EXCL(d);
  const int slot = x.cur;
  const ExclRec& r = x.rec[slot];
  const int64_t* req = JREQ(d, r.job);
  bool fit = n < c.N;
  if (fit) for (int k = 0; k < c.K; k++) fit = fit && AL(d, r.level, c.indexedCol[k], n) >= req[c.indexedCol[k]];
  if (fit && r.gate > 0) fit = WIDE_KEYS(c) ? keyBefore(d, r.level, n, r.gate - 1) : d.keys[(size_t)r.level * c.Npad + n] < d.keys[(size_t)r.level * c.Npad + (r.gate - 1)];   // the gate's walk stopped at its node: what orders before it was yielded
  exclPutBits(x.bits + (size_t)slot * x.W, n, fit);
  if (!fit) return;
  bool st = (d.shapeMask[(size_t)r.row * c.W + (n >> 6)] >> (n & 63)) & 1;
  if (st && r.uni >= 0) st = (d.labelMask[(size_t)r.uni * c.W + (n >> 6)] >> (n & 63)) & 1;
  if (st) exclDynPut(d, x, slot, n, r.level, req);
}

// gate-passed records: the failed fair-preemption walk (nodedb.go:935-1043 without a node at its end: every entry of the evicted table is visited).  Per node, its entries in
// walk order (the per-node index: descending table Index): the ones the job may take (scheduled at or below its priority, :963) are added to what priority -2 leaves
// on the node; the bit says the request was covered at some point — where the reference then looks at the static requirements (:996-1006).
DEV void exclFairBulk(Dev& d, int n) {
  const DevCfg& c = d.cfg;
  ExclDev& x = *EXCL(d);
  const int slot = x.cur;
  const ExclRec& r = x.rec[slot];
  const int64_t* req = JREQ(d, r.job);
  bool bit = false;
  if (n < c.N) {
    const int32_t sap = c.prios[r.level];
    int64_t av[MAXR];
    for (int q = 0; q < c.R; q++) av[q] = AL(d, 0, q, n);
    for (int k = d.fairOff[n]; k < d.fairOff[n + 1] && !bit; k++) {
      if (!d.evTabAlive[d.fairEnt[k]]) continue;
      const int ej = d.fairEntJob[k];
      if (d.schedAtPrio[ej] > sap) continue;
      const int64_t* er = JREQ(d, ej);
      bool met = true;
      for (int q = 0; q < c.R; q++) { av[q] += er[q]; met = met && av[q] >= req[q]; }
      bit = met;
    }
  }
  exclPutBits(x.fbits + (size_t)slot * x.W, n, bit);
}

// ---- Peek's skip of known-unfeasible scheduling keys (queue_scheduler.go:398-413) for a LONG stretch of queued jobs: the steady state the reference's own benchmark times
// (BenchmarkPreemptingQueueScheduler: full nodes, a queue of 320 000 identical jobs that no longer fit — each one gets a failed job context and is skipped).  The jobs are
// independent, the stretch ends at the first job that is not skippable: pass 1 finds that job (atomic minimum), pass 2 writes the records of the jobs in front of it.
// The stretch's start travels in the high bits of the pass's kind (the two words of a bulk command are all the helpers read).
DEV bool skipOk(Dev& d, int job) { return d.jGang[job] < 0 && d.unfeasible[d.jShape[job]] && !(d.jobFlags[job] & F_SUCCESSFUL); }
DEV void skipBulk(Dev& d, int kind, int pos, int i) {
  int job = d.queuedJobs[pos + i];
  if (kind == W_SKIP_FIND) { if (!skipOk(d, job)) atomicMinU32((uint32_t*)d.scanResult, (uint32_t)i); return; }
  d.jcEvicted[job] = 0; d.jcAssigned[job] = -1; d.jcGangCard[job] = 1; d.jcUniValue[job] = -1; d.jcStagedBy[job] = -1;   // JobSchedulingContextFromJob (context/job.go:149-158)
  d.jcHasPctx[job] = 1; d.pcNode[job] = -1; d.pcMethod[job] = ASCHED_METHOD_NONE;
  d.jobFlags[job] = (uint8_t)(d.jobFlags[job] | F_UNSUCCESSFUL);
  d.jcReason[job] = ASCHED_REASON_SKIPPED_UNFEASIBLE_KEY;
}

DEV_COLD void mergeBulkAny(Dev& d, int kind, int i);   // round_merge.h: the passes of a bulk-merged stream run share the op
DEV_COLD void wideBulkAny(Dev& d, int kind, int i) {
  if (kind >= 32 && kind < 40) { mergeBulkAny(d, kind, i); return; }   // (W_MG_*)
  if (kind == W_EXCL) { exclBulk(d, i); return; }
  if (kind == W_EXCLF) { exclFairBulk(d, i); return; }
  if ((kind & 255) >= W_SKIP_FIND) { skipBulk(d, kind & 255, kind >> 8, i); return; }
  const DevCfg& c = d.cfg;
  WideDev& w = *d.wide;
  const WideParams& P = *w.par;
  switch (kind) {
    case W_SEG: {   // queue i: what its stream consists of, the inputs of the queued part's prefix passes (QsIn)
      int q = i;
      WideSeg s; s.evStart = 0; s.evCnt = 0; s.qBase = 0; s.qLen = 0; s.flags = 0; s.total = 0; s.qWant = 0; s.pad = 0;
      QsIn in; in.base = 0; in.len = 0; in.skipUnf = P.skipUnf; in.pad = 0; in.weight = d.qWeight[q];
      for (int r = 0; r < MAXR; r++) in.a0[r] = r < c.R ? QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r] : 0;
      w.cnt[2 * q] = 0; w.cnt[2 * q + 1] = 0;
      int cap = w.cap[q] > 0 ? w.cap[q] : P.cap;
      if (cap > WIDE_L) cap = WIDE_L;
      if (d.pqInHeap[q]) {
        int ref = d.pqGctx[q];
        bool stream = false, queuedPart = false, headQueued = false;
        int qBase = 0;
        const bool queuedOk = P.queuedOk && !d.onlyEvByQueue[q] && !d.itJobOnlyEv[q] && !d.itGangOnlyEv[q];
        if (ref >= 0 && d.itNext[q] == ref && !d.jcPreempted[ref] && d.jGang[ref] < 0) {
          if (d.jcEvicted[ref]) {   // an evicted job on its way back (fastStreamPrepare's evicted stream: the costs exist, B_EVKEYS)
            int p = d.itEi[q] - 1, end = d.evOff[q + 1];
            if (P.evOk && d.evCheap[q] && d.itStage[q] == 0 && p >= d.evOff[q] && p < end && d.evList[p] == ref) {
              stream = true;
              s.evStart = p; s.evCnt = end - p;
              if (s.evCnt > cap) { s.evCnt = cap; s.flags |= 4; }   // more evicted jobs behind the cap
              else if (queuedOk) { queuedPart = true; qBase = d.itQi[q]; }
              // (not queuedOk: the queue yields nothing behind its evicted jobs — it leaves the heap, the merge goes on without it)
            }
          } else {                  // a queued job: element 0 of the queued part is the peeked head (fastStreamPrepare's queued stream)
            int b = d.itQi[q] - 1;
            if (d.itStage[q] == 1 && queuedOk && b >= d.queuedOff[q] && b < d.queuedOff[q + 1] && d.queuedJobs[b] == ref && d.jcAssigned[ref] < 0) {
              stream = true; queuedPart = true; headQueued = true; qBase = b; s.flags |= 8;
            }
          }
        }
        if (stream && queuedPart) {
          int avail = d.queuedOff[q + 1] - qBase;
          int len = avail, room = cap - s.evCnt;
          bool open = false;
          // CheckJobConstraints (constraints.go:121-157) on the queue side: a queue that may not schedule new jobs answers with a queue-terminal reason — the generic code's
          if (avail > 0 && (P.noNew || d.qCordoned[q] || d.qBurst[q] < 1 || d.qTokens[q] < 1)) { len = 0; open = true; }   // (noNew: the global limiter is empty — the first new job to reach the top gets the terminal reason)
          if (len > room) { len = room; open = true; }
          if (len > 0 && !d.qRateInf[q] && d.qTokens[q] < (double)len) { len = (int)d.qTokens[q]; open = true; }
          if (len > 0 && P.maxLookback != 0) {   // element e is peeked while itJobsSeen < maxLookback (queue_scheduler.go:434-444); the head was counted when it was peeked
            int64_t lim = (int64_t)P.maxLookback - d.itJobsSeen[q] + (headQueued ? 1 : 0);
            if (lim < (headQueued ? 1 : 0)) lim = headQueued ? 1 : 0;
            if (len > lim) { len = (int)lim; open = true; }   // (behind the limit the queue yields evicted jobs only: the generic iterator makes that switch)
          }
          // element 0 of a queued part that has not been peeked yet goes through Peek's unfeasible-key skip like every later one (B_QSSUM exempts element 0: there it is the head)
          if (len > 0 && !headQueued && P.skipUnf && d.unfeasible[d.jShape[d.queuedJobs[qBase]]]) { len = 0; open = true; }
          if (open) s.flags |= 4;
          s.qBase = qBase; s.qWant = len;
          in.base = qBase; in.len = len;
        }
        if (stream) s.flags |= 1;
        else { s.flags = 2 | 4; s.total = 1; }
      }
      d.qsIn[q] = in;
      w.seg[q] = s;
    } break;
    case W_EVSUM: {   // eviction-list position i: the queued part of its queue starts from the allocation the queue has once these evicted jobs are back
      int p = i;
      if (p >= P.numEvictedList) break;
      int job = d.evList[p];
      int q = d.jQueue[job];
      if (q < 0 || q >= c.Q) break;
      const WideSeg& s = w.seg[q];
      if (!(s.flags & 1) || s.qWant <= 0 || p < s.evStart || p >= s.evStart + s.evCnt) break;
      const int64_t* req = JREQ(d, job);
      for (int r = 0; r < c.R; r++) if (req[r]) atomicAddI64(&d.qsIn[q].a0[r], req[r]);
    } break;
    // ---- the queued part's costs: B_QSSUM / B_QSSTITCH / B_QSKEYS of round_run.h with chunks of WQ_CHUNK entries (a chunk is one thread's serial walk: dependent loads per job)
    case W_QSUM: {    // item i = (queue, chunk): sums of the requests, the first element the stream cannot contain (B_QSSUM's barrier rules)
      int q = i / WQ_CPQ, ch = i % WQ_CPQ;
      const QsIn& in = d.qsIn[q];
      int e0 = ch * WQ_CHUNK, e1 = e0 + WQ_CHUNK < in.len ? e0 + WQ_CHUNK : in.len;
      if (e0 >= in.len) break;
      int64_t* part = w.part + (size_t)i * (MAXR + 2);
      int64_t sum[MAXR]; for (int r = 0; r < MAXR; r++) sum[r] = 0;
      int barrier = INT32_MAX;
      const bool headPeeked = (w.seg[q].flags & 8) != 0;
      for (int e = e0; e < e1; e++) {
        int job = d.queuedJobs[in.base + e];
        const int64_t* req = JREQ(d, job);
        bool stop = d.jGang[job] >= 0 || ((e > 0 || !headPeeked) && in.skipUnf && d.unfeasible[d.jShape[job]]);   // (element 0 is exempt only where it is the peeked head: Peek's skip ran for it)
        for (int r = 0; r < c.R; r++) if (c.disallowed[r] && req[r] > 0) stop = true;
        if (stop) { barrier = e; break; }
        for (int r = 0; r < c.R; r++) sum[r] += req[r];
      }
      for (int r = 0; r < MAXR; r++) part[r] = sum[r];
      part[MAXR] = barrier;
    } break;
    case W_STITCH: {  // queue i: chunk sums -> carries, the stream's final length, whether the queue goes on behind it
      int q = i;
      WideSeg s = w.seg[q];
      if (!(s.flags & 3)) { w.off[2 * q] = 0; w.off[2 * q + 1] = 0; break; }
      if (s.flags & 1) {
        const QsIn& in = d.qsIn[q];
        int len = in.len; bool cut = false;
        int64_t run[MAXR]; for (int r = 0; r < MAXR; r++) run[r] = 0;
        for (int ch = 0; ch < WQ_CPQ && ch * WQ_CHUNK < in.len; ch++) {
          int64_t* part = w.part + ((size_t)q * WQ_CPQ + ch) * (MAXR + 2);
          int barrier = (int)part[MAXR];
          for (int r = 0; r < MAXR; r++) { int64_t v = part[r]; part[r] = run[r]; run[r] += v; }
          if (barrier != INT32_MAX) { len = barrier; cut = true; break; }
        }
        if (s.qWant > 0) {
          if (cut) s.flags |= 4;                                                        // a gang member / unfeasible key / disallowed request: the generic iterator's
          else if (in.base + in.len != d.queuedOff[q + 1]) s.flags |= 4;               // the queue's list goes on behind the prepared entries
        }
        s.qLen = s.qWant > 0 ? len : 0; s.total = s.evCnt + s.qLen;
        if (s.total == 0) { s.flags = 2 | 4; }   // nothing of this queue can be laid out (no token, cordoned, a barrier at its head): its head stops the merge like any other the run cannot serve
      }
      if (s.flags & 2) s.total = 1;   // a head the run cannot serve: its key as the generic Less sees it is ONE entry of the merge (W_PACK writes it)
      w.seg[q] = s;
      // where the queue's entries live in the compact arrays: any order will do (positions in the merged order do not depend on it), so a fetch-add hands out the space
      w.off[2 * q] = s.total > 0 ? atomicFetchAddI32((int32_t*)&w.stop[1], s.total) : 0;
      w.off[2 * q + 1] = s.total;
    } break;
    case W_PACK: {    // item i = (queue, chunk of the whole stream): packed keys of its entries, running maximum within the chunk, the chunk's maximum
      int q = i / WQ_CPQ, ch = i % WQ_CPQ;
      const WideSeg s = w.seg[q];
      if ((s.flags & 2) && ch == 0) {
        EvKey e; e.proposed = d.pqProposed[q]; e.current = d.pqCurrent[q]; e.size = d.pqSize[q]; e.pcPrio = d.pqPcPrio[q]; e.job = -1;
        w.key[w.off[2 * q]] = widePack(P.preferLarge, e, d.pqBudget[q]); w.own[w.off[2 * q]] = q; w.rank[w.off[2 * q]] = 0;
      }
      if (!(s.flags & 1)) break;
      int e0 = ch * WQ_CHUNK, e1 = e0 + WQ_CHUNK < s.total ? e0 + WQ_CHUNK : s.total;
      if (e0 >= e1) break;
      const QsIn& in = d.qsIn[q];
      const double budget = d.pqBudget[q], wgt = in.weight;
      const int base = w.off[2 * q];
      WideKey* out = w.key + base;
      WideKey eff; eff.a = 0; eff.x = 0; eff.y = 0;
      int64_t a[MAXR], with[MAXR]; bool haveA = false;
      for (int e = e0; e < e1; e++) {
        WideKey pk;
        if (e < s.evCnt) pk = widePack(P.preferLarge, d.evKey[s.evStart + e], budget);   // B_EVKEYS' costs of the whole eviction list
        else {
          int e2 = e - s.evCnt;   // position in the queued part: allocation before it = a0 + carry of its chunk + the requests of the chunk's entries before it
          if (!haveA || e2 % WQ_CHUNK == 0) {
            const int64_t* carry = w.part + ((size_t)q * WQ_CPQ + e2 / WQ_CHUNK) * (MAXR + 2);
            for (int r = 0; r < c.R; r++) a[r] = in.a0[r] + carry[r];
            for (int m = (e2 / WQ_CHUNK) * WQ_CHUNK; m < e2; m++) { const int64_t* rq = JREQ(d, d.queuedJobs[in.base + m]); for (int r = 0; r < c.R; r++) a[r] += rq[r]; }
            haveA = true;
          }
          int job = d.queuedJobs[in.base + e2];
          const int64_t* req = JREQ(d, job);
          for (int r = 0; r < c.R; r++) with[r] = a[r] + req[r];
          EvKey k;   // updatePQItem (queue_scheduler.go:636-686): the same float64 operations in the same order as B_QSKEYS
          k.proposed = drf(d, with) / wgt; k.current = drf(d, a) / wgt; k.size = drf(d, req) * wgt;
          k.pcPrio = c.pcPriority[d.jPc[job]]; k.job = job;
          pk = widePack(P.preferLarge, k, budget);
          for (int r = 0; r < c.R; r++) a[r] = with[r];
        }
        if (wideKeyLess(eff, pk)) eff = pk;
        out[e] = eff;
        w.own[base + e] = q; w.rank[base + e] = e;   // (the entries of its own queue that order before it; W_RANK adds the other queues')
      }
      w.cmax[(size_t)q * WQ_CPQ + ch] = eff;
    } break;
    case W_FIX: {     // queue i: running maximum across the chunks (almost always a no-op: DRF costs grow along a queue)
      int q = i;
      const WideSeg s = w.seg[q];
      if (!(s.flags & 1)) break;
      WideKey* out = w.key + w.off[2 * q];
      WideKey run; run.a = 0; run.x = 0; run.y = 0;
      int nch = (s.total + WQ_CHUNK - 1) / WQ_CHUNK;
      for (int ch = 0; ch < nch; ch++) {
        int e0 = ch * WQ_CHUNK, e1 = e0 + WQ_CHUNK < s.total ? e0 + WQ_CHUNK : s.total;
        if (ch > 0 && wideKeyLess(out[e0], run)) { for (int e = e0; e < e1; e++) { if (wideKeyLess(out[e], run)) out[e] = run; else break; } }
        const WideKey m = w.cmax[(size_t)q * WQ_CPQ + ch];
        if (wideKeyLess(run, m)) run = m;
      }
    } break;
    case W_RANK: {    // entry i: its position in the merged order = its index in its own queue + for every other queue the number of that queue's entries that order before it
      // (a binary search: all sequences are sorted by their running-maximum keys).  The mapping of items to threads is the point: the 64 lanes of a wave take 64
      // CONSECUTIVE entries of ONE queue, so every search step of the wave reads the same other queue's (small) key array — a broadcast or a handful of cache lines, not
      // 64 lines of 64 different arrays (measured on the MI355X: a wave-wide gather over 64 arrays costs ~2 k clocks per step: 57 % of a 256-queue round) — and
      // consecutive waves take DIFFERENT queues, so the entries that exist (index < total) are spread over all workgroups.
      const int total = (int)w.stop[1];
      const int ent = i % total, slice = i / total;   // (the lanes of a wave: consecutive compact entries — almost always of one queue — and ONE slice of the other queues)
      const int q = w.own[ent];
      const WideKey key = w.key[ent];
      const int myName = d.qNameRank[q];
      const int per = W_PER;
      const int g0 = slice * per, g1 = g0 + per < c.Q ? g0 + per : c.Q;
      int rank = 0;
      // W_RG searches side by side: their loads are independent, so a step of the group is ONE memory round trip (a lone wave per SIMD hides nothing: every
      // dependent load is a full trip to L2 / HBM — measured ~2 k clocks per step with one search at a time)
      for (int g = g0; g < g1; g += W_RG) {
        int lo[W_RG], hi[W_RG]; bool nb[W_RG]; const WideKey* kk[W_RG];
#pragma unroll
        for (int j = 0; j < W_RG; j++) {
          int q2 = g + j;
          bool valid = q2 < g1 && q2 != q;
          int q2c = valid ? q2 : q;
          const int o2 = w.off[2 * q2c];
          lo[j] = 0; hi[j] = valid ? w.off[2 * q2c + 1] : 0;
          nb[j] = valid && d.qNameRank[q2c] < myName;   // equal keys: Less ends with the queue name (queue_scheduler.go:796-797)
          kk[j] = w.key + o2;
        }
        for (;;) {
          WideKey m[W_RG]; bool live[W_RG]; bool any = false;
#pragma unroll
          for (int j = 0; j < W_RG; j++) { live[j] = lo[j] < hi[j]; any = any || live[j]; if (live[j]) m[j] = kk[j][(lo[j] + hi[j]) >> 1]; else { m[j].a = 0; m[j].x = 0; m[j].y = 0; } }
          if (!any) break;
#pragma unroll
          for (int j = 0; j < W_RG; j++) if (live[j]) {
            int mid = (lo[j] + hi[j]) >> 1;
            bool before = wideKeyLess(m[j], key) || (nb[j] && wideKeyEq(m[j], key));
            if (before) lo[j] = mid + 1; else hi[j] = mid;
          }
        }
#pragma unroll
        for (int j = 0; j < W_RG; j++) rank += lo[j];
      }
      if (rank) atomicAddI32(&w.rank[ent], rank);
    } break;
    case W_SCATTER: { // compact entry i into its position of the merged order; where the order stops being valid
      const int q = w.own[i], e = i - w.off[2 * q];
      const WideSeg s = w.seg[q];
      const int rank = w.rank[i];
      if (s.flags & 2) { atomicMinU32(&w.stop[0], (uint32_t)rank); break; }
      WideEnt en;
      if (e < s.evCnt) { en.job = d.evList[s.evStart + e]; en.qk = q | (1 << 30); }
      else { en.job = d.queuedJobs[s.qBase + (e - s.evCnt)]; en.qk = q; }
      w.merged[rank] = en;
      if (e == s.total - 1 && (s.flags & 4)) atomicMinU32(&w.stop[0], (uint32_t)(rank + 1));
    } break;
    case W_COMMIT_E: {   // merged position i < P.executed has been executed: the job's side and the per-(queue, class) sums
      if (i >= P.executed) break;
      const WideEnt en = w.merged[i];
      const int q = en.qk & 0xffffff, job = en.job;
      if ((en.qk >> 30) & 1) {
        // an evicted job is back on its node: the evicted branch of fastIter's commit == applyEvictedRange (node.go:416-442 arithmetic on the levels above the
        // evicted priority; level 0 is unchanged: +req un-evicts, -req binds), sctx.AddJobSchedulingContext of a rescheduled job (context/queue.go:231-265)
        const JobRec& jr = d.jrec[job];
        int n = jr.node0, pcx = jr.pc; int32_t prio = jr.runPrio;
        int32_t cutoff = jr.preemptible ? prio : NONPREEMPTIBLE_CUTOFF;
        for (int x = 0; x < c.R; x++) {
          int64_t v = jr.req[x];
          if (!v) continue;
          size_t ix = ((size_t)q * c.npc + pcx) * c.R + x;
          atomicAddI64(&d.qAllocByPc[ix], v); atomicAddI64(&d.qEvictedByPc[ix], -v);
          atomicAddI64(&QV(d.qAlloc, q)[x], v); atomicAddI64(&w.tot[MAXR + x], v);
          for (int l = 1; l < jr.nlRun; l++) atomicAddI64(&d.alloc[((size_t)l * c.R + x) * c.Npad + n], -v);
        }
        if (jr.keyDelta) for (int l = 1; l < jr.nlRun; l++) atomicAddI64((int64_t*)&d.keys[(size_t)l * c.Npad + n], -(int64_t)jr.keyDelta);
        d.jcReason[job] = 0; d.jcHasPctx[job] = 1; d.pcNode[job] = n; d.pcSap[job] = prio;
        d.jobNode[job] = n; d.jobCutoff[job] = cutoff; d.jobEvictedOnNode[job] = 0; d.schedAtPrio[job] = prio; d.inSchedAndEvicted[job] = 0;
        d.pcPap[job] = prio; d.pcMethod[job] = ASCHED_METHOD_RESCHEDULED; d.jobFlags[job] = F_RESCHEDULED; d.inPreempted[job] = 0;
        if (!P.replayPending) { int idx = d.evIndexOfJob[job]; if (idx >= 0) d.evTabAlive[idx] = 0; d.evIndexOfJob[job] = -1; }
        atomicAddI32(&w.cnt[2 * q], 1);
        atomicAddI64(&w.tot[2 * MAXR + 1], 1);
      } else {
        // a new job: the bind wave has issued BindJobToNode and the job's result fields (round_fast.h bindJob); the queue side of sctx.AddGangSchedulingContext here
        const int64_t* req = JREQ(d, job);
        int pcx = d.jPc[job];
        for (int x = 0; x < c.R; x++) {
          int64_t v = req[x];
          if (!v) continue;
          size_t ix = ((size_t)q * c.npc + pcx) * c.R + x;
          atomicAddI64(&d.qAllocByPc[ix], v); atomicAddI64(&d.qSchedByPc[ix], v);
          atomicAddI64(&QV(d.qAlloc, q)[x], v); atomicAddI64(&w.tot[x], v);
        }
        atomicAddI32(&w.cnt[2 * q + 1], 1);
        atomicAddI64(&w.tot[2 * MAXR], 1);
      }
    } break;
    case W_COMMIT_Q: {   // queue i after W_COMMIT_E: cursors and tokens as Clear + the Peeks of the entries served leave them, the next cap, and — where it is the plain case — the next head
      int q = i;
      WideSeg s = w.seg[q];
      s.pad = 0;
      if (!(s.flags & 1) || s.total == 0) { w.seg[q] = s; break; }
      const int cEv = w.cnt[2 * q], cQ = w.cnt[2 * q + 1];
      if (cEv + cQ >= s.total && (s.flags & 4)) {   // the queue used up what it had been given and goes on: IT ended the run — four times as many entries next time.  (Queues advance
        int have = w.cap[q] > 0 ? w.cap[q] : P.cap;   //  at very different rates under DRF; the caps only grow: the rank pass is cheap next to a run cut short)
        w.cap[q] = 4 * have > WIDE_L ? WIDE_L : 4 * have;
      }
      if (cEv + cQ == 0) { w.seg[q] = s; break; }
      // the iterators (queue_scheduler.go:595-606, 376-444; jobiteration.go:179-228): the next Peek yields the first entry that was not served.  The head had been
      // peeked (and, a queued one, counted) before the run.
      if (cEv) d.itEi[q] = s.evStart + cEv;
      if (cQ) {
        d.itQi[q] = s.qBase + cQ;
        d.itJobsSeen[q] += (s.flags & 8) ? cQ - 1 : cQ;
        if (!d.qRateInf[q] && 1 <= d.qBurst[q]) d.qTokens[q] -= (double)cQ;   // rate.Limiter.ReserveN per scheduled job (gang_scheduler.go:118-123)
      }
      d.itNext[q] = -1; d.pqInHeap[q] = 0;
      s.pad = 2;   // the control wave produces the next head through the generic updateAndPush ...
      // ... unless it is the plain case, done right here: Clear (:595-606) + Peek (:376-432) + updatePQItem (:636-686) for a next job that is a single job with
      // nothing special about it.  Anything else (the lookback switch, a gang member, a known-unfeasible key: records and counters of the scheduling context) is left
      // untouched for the generic code.
      do {
        if (P.maxLookback != 0 && !d.itGangOnlyEv[q] && (uint32_t)d.itJobsSeen[q] >= P.maxLookback) break;
        int job = -1; bool fromEv = false;
        if (d.itStage[q] == 0 && d.itEi[q] < d.evOff[q + 1]) { job = d.evList[d.itEi[q]]; fromEv = true; }
        else if (!(d.itJobOnlyEv[q] || !P.withQueued) && d.itQi[q] < d.queuedOff[q + 1]) job = d.queuedJobs[d.itQi[q]];
        d.pqGctx[q] = -1; d.pqProposed[q] = d.pqCurrent[q] = d.pqSize[q] = 0;
        if (job < 0) { if (d.itStage[q] == 0) d.itStage[q] = 1; s.pad = 1; break; }   // the queue has nothing more to yield: it stays out of the heap
        if (d.jGang[job] >= 0) break;
        if (!fromEv && P.skipUnf && d.unfeasible[d.jShape[job]]) break;   // (an evicted job has a node assigned: its key is not valid, keyValid)
        if (fromEv) d.itEi[q]++; else { if (d.itStage[q] == 0) d.itStage[q] = 1; d.itQi[q]++; }
        if (!fromEv) {   // JobSchedulingContextFromJob (context/job.go:149-158)
          d.jcEvicted[job] = 0; d.jcAssigned[job] = -1; d.jcHasPctx[job] = 0; d.jcReason[job] = 0; d.jcGangCard[job] = 1; d.jcUniValue[job] = -1; d.jcStagedBy[job] = -1;
        }
        if (!d.jcEvicted[job]) d.itJobsSeen[q]++;
        d.itNext[q] = job; d.pqGctx[q] = job;
        int64_t al[MAXR], with[MAXR];
        const int64_t* req = JREQ(d, job);
        for (int r = 0; r < c.R; r++) { al[r] = QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r]; with[r] = al[r] + req[r]; }
        double wq = d.qWeight[q];
        d.pqProposed[q] = drf(d, with) / wq; d.pqCurrent[q] = drf(d, al) / wq; d.pqSize[q] = drf(d, req) * wq;
        int32_t pp = c.pcPriority[d.jPc[job]], sp = pp;
        if (d.jcHasPctx[job]) sp = d.pcSap[job]; else if (d.jNode0[job] >= 0) sp = d.jRunPrio[job];
        d.pqPcPrio[q] = pp; d.pqSchedPrio[q] = sp;
        d.pqInHeap[q] = 1;
        s.pad = 1;
      } while (0);
      w.seg[q] = s;
    } break;
  }
}

#ifdef ASCHED_HOSTSIM
void hsEngineMustBeStopped(const char* what);   // tests/hostsim/hostsim.cpp: a wide op posted with the node engine live would hang the device
DEV void wgWide(Dev& d, int kind, int n) { hsEngineMustBeStopped("wgWide"); for (int i = 0; i < n; i++) wideBulkAny(d, kind, i); }
#else
DEV void wgWide(Dev& d, int kind, int n);   // armada_sched.hip: the pass on the control workgroup and the helper workgroups (OP_WIDE)
#endif

DEV_COLD int skipUnfeasibleBulk(Dev& d, int pos, int max) {
  if (pos >= (1 << 22)) return skipUnfeasibleRun(d, pos, max);   // (the start has 23 bits of the kind word)
  if (FLANE == 0) *(volatile uint32_t*)d.scanResult = (uint32_t)max;
  wgWide(d, W_SKIP_FIND | (pos << 8), max);
  int n = (int)UNI32(*(volatile uint32_t*)d.scanResult);
  if (n > max) n = max;
  if (n > 0) wgWide(d, W_SKIP_MARK | (pos << 8), n);
#ifdef ASCHED_HOSTSIM
  if (getenv("HS_TRACE_SKIP")) fprintf(stderr, "bulk skip: %d of %d jobs from position %d\n", n, max, pos);
#endif
  return n;
}

// rows on the literal iteration path: the nodes the iterators yield, one after the other (selectAtLevelLiteral's walk without the early exit)
DEV_COLD void exclLiteral(Dev& d, ExclDev& x, int slot) {
  const DevCfg& c = d.cfg;
  const ExclRec r = x.rec[slot];
  const int64_t* req = JREQ(d, r.job);
  uint64_t* bits = x.bits + (size_t)slot * x.W;
  FOR_LANES(w, x.W) bits[w] = 0;
  int64_t ireq[MAXK];
  for (int i = 0; i < c.K; i++) ireq[i] = req[c.indexedCol[i]];
  int t0 = d.rowTypeOff[r.row], nT = d.rowTypeOff[r.row + 1] - t0;
  if (nT > LIT_TMAX) { d.excl[r.job] = EXCL_S_UNSUPPORTED; return; }
  for (int k = 0; k < nT; k++) {
    LitIt& it = d.lit[k];
    it.type = d.rowTypes[t0 + k];
    for (int i = 0; i < MAXK; i++) it.lb[i] = i < c.K ? ireq[i] : 0;
    LIT_SET_BOUND(c, it);
    litAdvance(d, r.level, it, ireq);
  }
  for (;;) {
    int best = -1;
    for (int k = 0; k < nT; k++) if (d.lit[k].head >= 0 && (best < 0 || litNodeLess(d, r.level, d.lit[k].head, d.lit[best].head))) best = k;
    if (best < 0) return;
    int n = d.lit[best].head;
    if (r.gate > 0 && n == r.gate - 1) return;   // the node the gate found: its walk stopped here
    litAdvance(d, r.level, d.lit[best], ireq);
    if (FLANE == 0) {
      bits[n >> 6] |= 1ull << (n & 63);
      bool st = (d.shapeMask[(size_t)r.row * c.W + (n >> 6)] >> (n & 63)) & 1;
      if (st && r.uni >= 0) st = (d.labelMask[(size_t)r.uni * c.W + (n >> 6)] >> (n & 63)) & 1;
      if (st) exclDynPut(d, x, slot, n, r.level, req);
    }
  }
}
// the record of an attempt that found no node at the job's priority (round_ctl.h selectNodeForJob)
DEV_COLD void exclRecordWide(Dev& d, int job, int level, int gate) {   // gate >= 0: the node the attempt found at the job's priority before it ended without one (ExclRec)
  ExclDev& x = *EXCL(d);
  int slot = d.excl[job];
  if (slot == EXCL_S_DROPPED) return;
  if (slot < 0) {
    int cnt = x.count;
    if (cnt >= x.cap) { d.excl[job] = EXCL_S_DROPPED; return; }
    slot = cnt; x.count = cnt + 1; d.excl[job] = slot;
  }
  ExclRec r;
  r.job = job; r.level = level; r.row = d.rs->awayRowPlus1 ? d.rs->awayRowPlus1 - 1 : d.jShape[job]; r.uni = d.jcUniValue[job]; r.flags = 0;
  r.gate = gate >= 0 ? gate + 1 : 0; r.fair = gate >= 0 && !d.cfg.disableFair; r.pad = 0;
  x.rec[slot] = r;
  x.cur = slot;
  if (r.fair) { ensureFairIndex(d); wgWide(d, W_EXCLF, x.W * 64); }
  if (d.rowLiteral && d.rowLiteral[r.row]) { exclLiteral(d, x, slot); return; }
  wgWide(d, W_EXCL, x.W * 64);
}
// a pinned (evicted) job that no longer fits on its node: DynamicJobRequirementsMet's reason on that one node (nodedb.go:897-920) — the first column in factory order
DEV_COLD void exclPinned(Dev& d, int job, int node, int level) {
  const int64_t* req = JREQ(d, job);
  int res = -1; int64_t av = 0;
  for (int q = 0; q < d.cfg.R; q++) if (res < 0 && req[q] > AL(d, level, q, node)) { res = q; av = AL(d, level, q, node); }
  if (res < 0) { d.excl[job] = EXCL_S_UNSUPPORTED; return; }
  EXCL(d)->pinAvail[job] = av;
  d.excl[job] = EXCL_S_PINNED0 - res;
}


// Is the head of queue t something a wide run can start with?  (The bulk preparation costs a few hundred microseconds: not worth it for a head the run stops at.)
DEV bool wideHeadOk(Dev& d, const Ctl& c, const PassCfg& pc, int t) {
  int ref = d.pqGctx[t];
  if (ref < 0 || d.itNext[t] != ref || d.jcPreempted[ref] || d.jGang[ref] >= 0) return false;
  if (d.jcEvicted[ref]) {
    int p = d.itEi[t] - 1;
    return c.fastEvStatic && d.rs->lvl0NonNeg && d.rs->numPreemptedMarks == 0 && d.evCheap[t] && d.itStage[t] == 0 && p >= d.evOff[t] && p < d.evOff[t + 1] && d.evList[p] == ref;
  }
  int b = d.itQi[t] - 1;
  return d.itStage[t] == 1 && pc.withQueued && !c.onlyEvicted && !d.onlyEvByQueue[t] && !d.itJobOnlyEv[t] && !d.itGangOnlyEv[t] && !d.qCordoned[t] && d.qBurst[t] >= 1 &&
         d.qTokens[t] >= 1 && d.rs->globalTokens >= 1 && d.rs->globalBurst >= 1 && b >= d.queuedOff[t] && d.queuedJobs[b] == ref && d.jcAssigned[ref] < 0;
}

// One wide run.  Called by queueSchedule with the generic state live (every queue's iterator, heap item and allocation in the generic arrays) and the queue at
// the top of the heap known to be servable; returns the number of entries executed (0: nothing changed).  On return the generic state is the live one again.
// coarse segment clocks of a run (always on: a dozen clock reads per run of thousands of entries) in statSeg[24..35]: [24] W_SEG + W_EVSUM, [25] queued-part prefix passes,
// [26] W_KEYS, [27] W_RANK, [28] engine start, [29] staging (the engine's pace), [30] engine stop, [31] W_COMMIT, [32] the touched queues' next heads, [33] runs, [34] entries ranked
#define WSEG(i) do { long long n_ = CLK(); rs.statSeg[i] += n_ - wT_; wT_ = n_; } while (0)
DEV_COLD int wideRun(Dev& d, Ctl& c, const PassCfg& pc) {
  const FastK k = fastKRef(d);
  const int Q = d.cfg.Q, R = d.cfg.R;
  WideDev& w = *d.wide;
  RoundScalars& rs = *d.rs;
  if (!rs.fastActive || c.compareSchedPrio || c.useReplayAlloc || c.txn.active || k.hasPcLimit || k.anyRoundLimit || k.disableHome || !d.qsIn) return 0;
  int allowed = INT32_MAX;
  if (!rs.globalRateInf) allowed = rs.globalTokens >= 2147483000.0 ? INT32_MAX : (rs.globalTokens < 1 ? 0 : (int)rs.globalTokens);
  if (rs.globalBurst < 1 || rs.globalTokens < 1) allowed = 0;
  // ---- preparation: every body below reads HBM only; the helper workgroups take their share
  fastFence(c);
  {
    WideParams P; memset(&P, 0, sizeof P);
    P.evOk = c.fastEvStatic && rs.lvl0NonNeg && rs.numPreemptedMarks == 0;
    P.queuedOk = pc.withQueued && !c.onlyEvicted; P.noNew = allowed <= 0;
    P.skipUnf = pc.skipKnown && rs.numUnfeasible > 0;
    P.preferLarge = c.preferLarge; P.cap = 32;   // (a queue's first run; WideDev.cap[q] follows its consumption from then on)
    P.numEvictedList = rs.numEvictedList; P.replayPending = rs.replayPending; P.executed = 0; P.maxLookback = pc.maxLookback; P.withQueued = pc.withQueued;
    *w.par = P;
    w.stop[0] = 0xffffffffu; w.stop[1] = 0;
    FOR_LANES(x, 3 * MAXR) w.tot[x] = 0;
  }
  FAST_GLOBAL_FENCE();
  long long wT_ = CLK();
  wgWide(d, W_SEG, Q);
  if (c.fastEvStatic && rs.numEvictedList > 0) wgWide(d, W_EVSUM, rs.numEvictedList);
  WSEG(24);
  wgWide(d, W_QSUM, Q * WQ_CPQ);
  wgWide(d, W_STITCH, Q);
  WSEG(25);
  wgWide(d, W_PACK, Q * WQ_CPQ);
  wgWide(d, W_FIX, Q);
  WSEG(26);
  uint32_t total = UNI32(w.stop[1]);
  wgWide(d, W_RANK, (int)total * ((Q + W_PER - 1) / W_PER));
  wgWide(d, W_SCATTER, (int)total);
  WSEG(27);
  uint32_t stop = UNI32(w.stop[0]);
  int V = stop < total ? (int)stop : (int)total;
#ifdef ASCHED_HOSTSIM
  if (V > 0) {   // the merged order starts with the queue the generic Less serves next
    int t = pqTop(d, c);
    if ((w.merged[0].qk & 0xffffff) != t) { fprintf(stderr, "hostsim: wide merge disagrees with Less (merged %d, generic %d)\n", w.merged[0].qk & 0xffffff, t); abort(); }
  }
  if (getenv("HS_WIDE_TRACE")) {
    fprintf(stderr, "wide run: V %d stop %u total %u allowed %d", V, stop, total, allowed);
    for (int q = 0; q < Q; q++) { const WideSeg& s = w.seg[q]; for (int e = 0; e < s.total; e++) { int r = w.rank[w.off[2 * q] + e];
      if ((s.flags & 2) && r == (int)stop) fprintf(stderr, "  | barrier q%d gctx %d ev %d stage %d evCheap %d itNext %d", q, d.pqGctx[q], d.pqGctx[q] >= 0 ? d.jcEvicted[d.pqGctx[q]] : -1, d.itStage[q], d.evCheap[q], d.itNext[q]);
      if (!(s.flags & 2) && (s.flags & 4) && e == s.total - 1 && r + 1 == (int)stop) fprintf(stderr, "  | open q%d evCnt %d qLen %d qWant %d cap %d avail %d tokens %.0f", q, s.evCnt, s.qLen, s.qWant, w.cap[q], d.queuedOff[q + 1] - s.qBase, d.qTokens[q]); } }
    fprintf(stderr, "\n");
  }
#endif
  if (V <= 0) return 0;
  // ---- execution: stage the merged order into the ring; the node engine places the queued jobs, the bind wave issues the HBM side behind it
  FastS S; HS_POISON(S);
  S.engLive = 0; S.engPend = -1; S.engWaitClk = 0; S.engSeq = 0; S.inlineStreak = 0;
  S.laneL = FLANE / (R > 0 ? R : 1); S.laneX = FLANE % (R > 0 ? R : 1);
  S.statScanSteps = 0; S.statL0Max = UNI32(rs.statL0Max); S.fastActive = 1; S.numUnfeasible = UNI32(rs.numUnfeasible); S.statRefills = 0;
  S.lvl0NonNeg = UNI32(rs.lvl0NonNeg); S.numPreemptedMarks = UNI32(rs.numPreemptedMarks); S.replayPending = UNI32(rs.replayPending);
  rs.statSeg[33] += 1; rs.statSeg[34] += total;
  wT_ = CLK();
  engineStart(d, S);
  int engSeq = S.engSeq;
  streamBegin(&engSeq);
  WSEG(28);
  int emitted = 0, emittedQ = 0, stageBase = -1, stageCnt = 0, fail = 0;
  unsigned long long stageV = 0;
  for (int base = 0; base < V && !fail; base += 64) {
    int nb = V - base < 64 ? V - base : 64;
    FOR_LANES(x, 64) if (x < nb) { WideEnt en = w.merged[base + x]; FL.tmpX[x] = ((uint64_t)(uint32_t)en.qk << 32) | (uint32_t)en.job; }
    LANE0_PUBLISHED();
    bool over = false;
    for (int x = 0; x < nb; x++) {
      for (;;) {   // the ring is the engine's pace
        int a = streamAcked(&fail);
        if (fail || (emitted - a < RING_N - 8 && emitted - streamBound() < RING_N - 8)) break;
        STREAM_IDLE();
      }
      if (fail) break;
      uint64_t v = UNI64(FL.tmpX[x]);
      int job = (int)(uint32_t)v, qk = (int)(uint32_t)(v >> 32);
      int ev = (qk >> 30) & 1;
      if (!ev && emittedQ >= allowed) { over = true; break; }   // no global token left for another new job (constraints.go:129-141)
      if (FLANE == 0) { RJOB(emitted) = job; RQ(emitted) = ev ? RQ_EV : 0; }
      LANE0_PUBLISHED();
      emitted++; if (!ev) emittedQ++;
      if ((emitted & 3) == 0) {
        if (stageBase >= 0) streamStageCommit(d, k, stageBase, stageCnt, stageV);
        stageBase = emitted - 4; stageCnt = 4;
        stageV = streamStageIssue(k, stageBase, 4);
      }
    }
    if (over) break;
  }
  if (!fail) {
    if (stageBase >= 0) streamStageCommit(d, k, stageBase, stageCnt, stageV);
    int done = stageBase >= 0 ? stageBase + stageCnt : 0;
    if (emitted > done) { stageV = streamStageIssue(k, done, emitted - done); streamStageCommit(d, k, done, emitted - done, stageV); }
  }
  streamEnd(engSeq);
  int E = streamAcked(&fail);
  S.engSeq = engSeq;
  WSEG(29);
  engineStop(d, S);
  WSEG(30);
  if (UNI32(FL.eng.cancel)) c.cancelSeen = 1;
  c.l1Dirty = 1;
  fastFence(c);
  if (fail == 2) { rs.fastActive = 0; fastDrop(d); }   // placed, but the L0 list overflowed: the full plane scan takes over (counted in round_stats)
  rs.statScanSteps += S.statScanSteps;
  if (S.statL0Max > rs.statL0Max) rs.statL0Max = S.statL0Max;
  // ---- commit: per queue on all workgroups, then the scheduling context's scalars here
  int EQ = 0, EEv = 0;
  if (E > 0) {
    w.par->executed = E;
    FAST_GLOBAL_FENCE();
    wT_ = CLK();
    wgWide(d, W_COMMIT_E, E);
    wgWide(d, W_COMMIT_Q, Q);
    WSEG(31);
    EQ = (int)UNI64(w.tot[2 * MAXR]); EEv = (int)UNI64(w.tot[2 * MAXR + 1]);
    for (int x = 0; x < R; x++) {
      int64_t sq = (int64_t)UNI64(w.tot[x]), se = (int64_t)UNI64(w.tot[MAXR + x]);
      rs.allocated[x] += sq + se; rs.scheduled[x] += sq; rs.evicted[x] -= se;
    }
    rs.numScheduledJobs += EQ; rs.numScheduledGangs += EQ; rs.numNodeQueries += EQ; rs.numEvictedJobs -= EEv;
    if (!rs.globalRateInf && 1 <= rs.globalBurst) rs.globalTokens -= (double)EQ;
    rs.loopIterations += E; rs.statFastIters += E;
    rs.statStreamRuns++; rs.statStreamJobs += E; rs.statStreamEmitted += emitted; rs.statStreamPrepared += (int)total;
    // the next heads of the queues W_COMMIT_Q left to the generic iterator (a gang member, a known-unfeasible key, the lookback switch)
    for (int q0 = 0; q0 < Q; q0 += 64) {
      FOR_LANES(x, 64) FL.tmpQ[x] = q0 + x < Q ? w.seg[q0 + x].pad : 0;
      LANE0_PUBLISHED();
      for (int x = 0; x < 64 && q0 + x < Q; x++) if (UNI32(FL.tmpQ[x]) == 2) updateAndPush(d, c, q0 + x, pc);
    }
    FOR_LANES(x, 64) FL.tmpQ[x] = 0;
    WSEG(32);
  }
  return E;
}
