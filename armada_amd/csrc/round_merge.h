// round_merge.h — bulk-merged stream runs: the merge of a stream run (round_fast.h) computed ahead, on every workgroup.
//
// Why.  In a stream run the control wave merges the queues' precomputed cost streams one entry at a time (pop the lane heap, fetch the queue's next costs, pack the key,
// re-insert: ~4 k shader clocks per entry) while the node engine places the entries: on BASELINE configs[2] the two waves were each other's pace for four rounds
// (profiles/r05b_headline_engine_segments.txt).  Which queue is served next with which job depends on nothing the node side produces (round_fast.h "stream run"), so
// the whole order can be computed before the first entry is placed.
//
// What.  QueueCandidateGangIteratorPQ (queue_scheduler.go:701-798) over the heads' keys is a k-way heap merge of the queues' streams.  A heap merge serves a queue's
// elements in list order whatever their keys, so it equals the sort of all elements by (running maximum of the packed key within the queue, queue-name rank, position) —
// the argument of round_fast.h "skip mode" and of the wide runs (round_wide.h W_RANK).  The sequences being sorted, an element's position is its index in its own queue
// plus, for every other queue, the number of that queue's elements that order before it: one binary search per (element, other queue), every element independent.
// Passes (bodies below, run through wgWide on the control workgroup and the helper workgroups; they read and write HBM only):
//   W_MG_PACK     (queue, chunk of 64): packed keys (packKey3 on the precomputed costs: d.qsKey / d.evKey), running maximum within the chunk, the chunk's maximum
//   W_MG_FIX      queue: the running maximum across the chunks
//   W_MG_CUT      (round 6) a run can serve at most `need` entries (the global tokens left + the evicted entries there are): a key K* with at least that many entries at or
//                 below it is found from every 256th key of each queue (a sample at or below K* stands for 256 entries of its queue at or below K*), and only
//                 entries with key <= K* are ranked and scattered — plus each queue's first entry above it, whose rank is where the merged order ends.  On the
//                 headline round: 1 M entries, 200 k tokens -> the rank pass works on ~215 k (profiles/r06z_merge_cut.txt)
//   W_MG_RANK     (element, slice of the other queues): the binary searches
//   W_MG_SCATTER  element: merged[rank] = (job, queue, stream position); where the merged order stops being valid
// The run stops — before any side effect — at the first of: a queue in the heap whose head is not a stream element (ONE entry under its heap key: nothing that orders
// after it is served); the last element of a stream whose queue goes on under a key not known here; in skip mode, an element whose own key orders before the running
// maximum in front of it (served keys must not decrease there: round_fast.h fastRun); then the control wave's own limits while it stages (global tokens, a job without a
// node, cancel).  Stopping early is always exact: a run may end anywhere between two entries (fastStreamRun's settling produces every queue's next head from its
// then-current allocation).
//
// The CPU build replays the heap merge literally after the passes and aborts on the first difference (mgCheck).
#pragma once

enum { W_MG_PACK = 32, W_MG_FIX, W_MG_RANK, W_MG_SCATTER, W_MG_CUT };
#define MG_SAMPLE 256   // W_MG_CUT: every 256th key of a queue is a sample
#define MG_SPQ (QS_CMAX / MG_SAMPLE)
#define MG_PER 16      // other queues one item of the rank pass walks for its element
#define MG_RG 16       // binary searches that run side by side (independent loads in flight together: a step of the group is one memory round trip)
#define MG_F_STREAM 1
#define MG_F_BARRIER 2
#define MG_F_OPEN 4
#define MG_F_SKIP 8
#define MG_F_PREFER_LARGE 16
#ifndef MG_MIN_ENTRIES
#define MG_MIN_ENTRIES MG_MIN_ENTRIES_DEFAULT   // below this a run is merged by the control wave as before (the passes cost a few hand-shakes with the helper workgroups, and the node engine stops for them)
#endif

DEV EvKey mgCost(const Dev& d, const MgQ& s, int q, int pos) { return (s.kind & 1) ? d.evKey[s.base + pos] : d.qsKey[(size_t)q * QS_CMAX + pos]; }

DEV_COLD void mergeBulkAny(Dev& d, int kind, int i) {
  MgDev& mg = *d.mg;
  switch (kind) {
    case W_MG_PACK: {
      const int q = i / MG_CPQ, ch = i % MG_CPQ;
      const MgQ s = mg.q[q];
      if ((s.flags & MG_F_BARRIER) && ch == 0) { mg.key[s.off] = s.head; mg.own[s.off] = q; mg.rank[s.off] = 0; }
      if (!(s.flags & MG_F_STREAM)) break;
      const int e0 = ch * MG_CHUNK, e1 = e0 + MG_CHUNK < s.total ? e0 + MG_CHUNK : s.total;
      if (e0 >= e1) break;
      const bool folded = (s.kind & 2) != 0;
      WideKey run; run.a = 0; run.x = 0; run.y = 0;
      if (folded) run = s.eff;
      for (int e = e0; e < e1; e++) {
        const EvKey c = mgCost(d, s, q, s.start + e);
        const WideKey pk = widePack((s.flags & MG_F_PREFER_LARGE) ? 1 : 0, c, s.budget);
        const bool dec = !folded && wideKeyLess(pk, run);   // (a folded queue's keys are max(own, running maximum) by definition: never a decrease)
        if (wideKeyLess(run, pk)) run = pk;
        mg.key[s.off + e] = run;
        mg.own[s.off + e] = q | (dec ? (1 << 30) : 0);
        mg.rank[s.off + e] = e;   // the elements of its own queue that order before it; W_MG_RANK adds the other queues'
      }
      mg.cmax[(size_t)q * MG_CPQ + ch] = run;
    } break;
    case W_MG_FIX: {
      const int q = i;
      const MgQ s = mg.q[q];
      if (!(s.flags & MG_F_STREAM)) break;
      const bool folded = (s.kind & 2) != 0;
      WideKey* out = mg.key + s.off;
      WideKey run; run.a = 0; run.x = 0; run.y = 0;
      const int nch = (s.total + MG_CHUNK - 1) / MG_CHUNK;
      for (int ch = 0; ch < nch; ch++) {
        const int e0 = ch * MG_CHUNK, e1 = e0 + MG_CHUNK < s.total ? e0 + MG_CHUNK : s.total;
        if (ch > 0) for (int e = e0; e < e1; e++) {
          if (!wideKeyLess(out[e], run)) break;   // (the chunk's keys are non-decreasing: the rest is at or above the running maximum)
          out[e] = run;
          if (!folded) mg.own[s.off + e] |= 1 << 30;
        }
        const WideKey m = mg.cmax[(size_t)q * MG_CPQ + ch];
        if (wideKeyLess(run, m)) run = m;
      }
    } break;
    case W_MG_RANK: {
      // The lanes of a wave take consecutive compact entries (almost always of one queue) and ONE slice of the other queues: every search step of the wave reads the same
      // other queue's key array at neighbouring places (round_wide.h W_RANK has the measurements behind this mapping).
      const int total = (int)mg.stop[1];
      const int ent = i % total, slice = i / total;
      const int q = mg.own[ent] & 0xffffff;
      const WideKey key = mg.key[ent];
      if (mg.stop[3]) {   // the cut: entries above K* are not ranked — but for the first such entry of a queue (its rank ends the merged order)
        const WideKey ks = mg.cut[0];
        if (wideKeyLess(ks, key) && ent != mg.q[q].off && wideKeyLess(ks, mg.key[ent - 1])) break;
      }
      const int myName = mg.q[q].nameRank;
      const int Q = d.cfg.Q;
      const int g0 = slice * MG_PER, g1 = g0 + MG_PER < Q ? g0 + MG_PER : Q;
      int rank = 0;
      for (int g = g0; g < g1; g += MG_RG) {
        int lo[MG_RG], hi[MG_RG]; bool nb[MG_RG]; const WideKey* kk[MG_RG];
#pragma unroll
        for (int j = 0; j < MG_RG; j++) {
          const int q2 = g + j;
          const bool valid = q2 < g1 && q2 != q;
          const int q2c = valid ? q2 : q;
          const MgQ& o = mg.q[q2c];
          lo[j] = 0; hi[j] = valid ? o.total : 0;
          nb[j] = valid && o.nameRank < myName;   // equal keys: Less ends with the queue name (queue_scheduler.go:796-797)
          kk[j] = mg.key + o.off;
        }
        for (;;) {
          WideKey m[MG_RG]; bool live[MG_RG]; bool any = false;
#pragma unroll
          for (int j = 0; j < MG_RG; j++) { live[j] = lo[j] < hi[j]; any = any || live[j]; if (live[j]) m[j] = kk[j][(lo[j] + hi[j]) >> 1]; else { m[j].a = 0; m[j].x = 0; m[j].y = 0; } }
          if (!any) break;
#pragma unroll
          for (int j = 0; j < MG_RG; j++) if (live[j]) {
            const int mid = (lo[j] + hi[j]) >> 1;
            const bool before = wideKeyLess(m[j], key) || (nb[j] && wideKeyEq(m[j], key));
            if (before) lo[j] = mid + 1; else hi[j] = mid;
          }
        }
#pragma unroll
        for (int j = 0; j < MG_RG; j++) rank += lo[j];
      }
      if (rank) atomicAddI32(&mg.rank[ent], rank);
    } break;
    case W_MG_SCATTER: {
      const int q = mg.own[i] & 0xffffff;
      const bool dec = ((mg.own[i] >> 30) & 1) != 0;
      const MgQ s = mg.q[q];
      const int e = i - s.off;
      const int rank = mg.rank[i];
      if (s.flags & MG_F_BARRIER) { atomicMinU32(&mg.stop[0], (uint32_t)rank); break; }
      if (mg.stop[3] && wideKeyLess(mg.cut[0], mg.key[i])) {   // above the cut: not part of this run's merged order, which ends in front of the first of them
        if (e == 0 || !wideKeyLess(mg.cut[0], mg.key[i - 1])) atomicMinU32(&mg.stop[0], (uint32_t)rank);
        break;
      }
      const int pos = s.start + e;
      MgEnt en; en.job = mgCost(d, s, q, pos).job; en.qk = q | ((s.kind & 1) ? (1 << 30) : 0); en.e = pos; en.ci = i;
      mg.merged[rank] = en;
      if (e == s.total - 1 && (s.flags & MG_F_OPEN)) atomicMinU32(&mg.stop[0], (uint32_t)(rank + 1));
      if (dec && (s.flags & MG_F_SKIP)) atomicMinU32(&mg.stop[0], (uint32_t)rank);
    } break;
    case W_MG_CUT: {   // item = (queue, sample): how many entries are AT LEAST at or below this sample's key — 256 per sample at or below it, over all queues
      const int q = i / MG_SPQ, j = i % MG_SPQ;
      const MgQ s = mg.q[q];
      if (!(s.flags & MG_F_STREAM) || (j + 1) * MG_SAMPLE > s.total) break;
      const WideKey ks = mg.key[s.off + j * MG_SAMPLE + MG_SAMPLE - 1];
      const int Q = d.cfg.Q;
      uint32_t cnt = 0;
      for (int q2 = 0; q2 < Q; q2++) {
        const MgQ o = mg.q[q2];
        if (!(o.flags & MG_F_STREAM)) continue;
        int lo = 0, hi = o.total / MG_SAMPLE;   // samples of q2 at or below ks: a prefix (the keys are running maxima)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (!wideKeyLess(ks, mg.key[o.off + mid * MG_SAMPLE + MG_SAMPLE - 1])) lo = mid + 1; else hi = mid; }
        cnt += (uint32_t)lo;
      }
      if ((uint64_t)cnt * MG_SAMPLE >= (uint64_t)mg.stop[2]) atomicMinU32(&mg.cutPick[0], (cnt << 16) | (uint32_t)i);   // the smallest such key (count and key grow together; equal keys, equal counts)
    } break;
  }
}

#ifdef ASCHED_HOSTSIM
// the heap merge, literally (streamMerge's loop without the ring): the bulk passes must give the same sequence — possibly shorter (they stop at every decrease in skip
// mode, streamMerge only at one below the key served last), never different
static void mgCheck(Dev& d, const FastCtx& fc, int Q, int skip, int V) {
  static const bool off = getenv("HS_NO_MG_CHECK") != nullptr;
  MgDev& mg = *d.mg;
  if (off || (int)mg.stop[1] > 400000) return;
  struct HQ { bool in; PackedKey key; int pos; } h[QCAPF];
  for (int q = 0; q < Q; q++) { h[q].in = FL.inHeap[q] != 0; h[q].key.A = FL.kA[q]; h[q].key.X = FL.kX[q]; h[q].key.Y = FL.kY[q]; h[q].pos = mg.q[q].start; }
  PackedKey effK[QCAPF];
  for (int q = 0; q < Q; q++) { effK[q].A = (uint32_t)mg.q[q].eff.a; effK[q].X = mg.q[q].eff.x; effK[q].Y = mg.q[q].eff.y; }
  int n = 0;
  PackedKey lastK; lastK.A = 0; lastK.X = lastK.Y = 0; uint32_t lastN = 0; bool haveLast = false;
  for (;;) {
    int t = -1;
    for (int q = 0; q < Q; q++) if (h[q].in && (t < 0 || packedLess(h[q].key, (uint32_t)FL.nameRank[q], h[t].key, (uint32_t)FL.nameRank[t]))) t = q;
    if (t < 0) break;
    const MgQ& s = mg.q[t];
    if (!(s.flags & MG_F_STREAM)) break;
    if (skip) { if (haveLast && packedLess(h[t].key, (uint32_t)FL.nameRank[t], lastK, lastN)) break; lastK = h[t].key; lastN = (uint32_t)FL.nameRank[t]; haveLast = true; }
    if (n < V) {
      const MgEnt& en = mg.merged[n];
      if ((en.qk & 0xffffff) != t || en.e != h[t].pos) {
        fprintf(stderr, "hostsim: bulk merge disagrees with the heap merge at position %d of %d: merged (queue %d, element %d), heap (queue %d, element %d); skip %d\n", n, V, en.qk & 0xffffff, en.e, t, h[t].pos, skip);
        abort();
      }
    }
    n++;
    h[t].pos++;
    if (h[t].pos < s.len) {
      const EvKey c = mgCost(d, s, t, h[t].pos);
      PackedKey own = packKey3(fc.preferLarge, c.pcPrio, c.proposed, c.current, c.size, s.budget);
      if (s.kind & 2) { if (packedLess(own, 0, effK[t], 0)) own = effK[t]; else effK[t] = own; }
      h[t].key = own;
    } else if (s.flags & MG_F_OPEN) break;
    else h[t].in = false;
    if (n >= V + 4) break;
  }
  if (V > n) { fprintf(stderr, "hostsim: bulk merge is valid for %d entries, the heap merge ends after %d\n", V, n); abort(); }
}
#endif

// The merged order of the run that is about to start, for the queues' streams as fastStreamRun will see them.  Control wave, node engine STOPPED (the passes use every
// wave of the workgroup).  Returns the number of valid merged entries; 0 = this run is merged by the control wave as before (too few entries; a gang the run would nest;
// a stream longer than the key arrays).
DEV_NOINLINE int mgPrepare(Dev& d, FastCtx fc, int Q, int skip, int need) {   // need: queued jobs the run can serve at most (the global tokens), INT32_MAX = no such bound
#ifdef ASCHED_HOSTSIM
  { static const bool off = getenv("HS_NO_MERGE") != nullptr; if (off) return 0; }
  static const int minEntries = getenv("HS_MG_MIN") ? atoi(getenv("HS_MG_MIN")) : MG_MIN_ENTRIES;
#else
  const int minEntries = d.f.mgMin;
#endif
  if (!d.mg || fc.replay || Q > QCAPF) return 0;
  const FastK k = fastKRef(d);
  MgDev& mg = *d.mg;
  // per queue: what the merge sees of it (lane q); the compact offsets are a prefix sum over the queues' entry counts (through LDS)
  auto describe = [&](int q, int off) {
    const QHot& f = FL.hot[q];
    MgQ s; memset(&s, 0, sizeof s);
    s.nameRank = FL.nameRank[q]; s.budget = f.budget; s.off = off;
    int bad = 0;
    if (FL.inHeap[q]) {
      if (f.sLen > f.sPos) {
        const int kd = FL.sKind[q] ? 1 : 0;
        s.kind = kd | (f.effValid ? 2 : 0); s.start = f.sPos; s.len = f.sLen; s.total = f.sLen - f.sPos;
        s.base = (kd ? f.itEi : f.itQi) - 1 - f.sPos;
        s.flags = MG_F_STREAM | (skip ? MG_F_SKIP : 0) | (fc.preferLarge ? MG_F_PREFER_LARGE : 0);
        const bool listEnds = !kd && d.qsLen[2 * q + 1] != 0;
        if (!listEnds) {
          s.flags |= MG_F_OPEN;
          // the element behind a queued stream is a gang member: streamMerge settles the queue and goes on (streamNest) — such runs stay with it
          if (!kd && !f.effValid) { const int nx = s.base + f.sLen; if (nx < f.qEnd && d.jGang[k.queuedJobs[nx]] >= 0) bad = 1; }
        }
        if (s.kind & 2) { s.eff.a = FL.effA[q]; s.eff.x = FL.effX[q]; s.eff.y = FL.effY[q]; }
        if (!kd && s.len > QS_CMAX) bad = 1;
      } else {
        s.flags = MG_F_BARRIER | (fc.preferLarge ? MG_F_PREFER_LARGE : 0); s.total = 1;
        s.head.a = FL.kA[q]; s.head.x = FL.kX[q]; s.head.y = FL.kY[q];
        if (f.gctx < -1) bad = 1;   // an assembled gang at the head of a queue: placed inside the run (streamNest)
      }
    }
    s.pad_ = bad;
    return s;
  };
  FOR_LANES(q, QCAPF) { FL.tmpQ[q] = 0; FL.tmpN[q] = 0; }
  LANE0_PUBLISHED();
  FOR_LANES(q, Q) { const MgQ s = describe(q, 0); FL.tmpQ[q] = s.total; FL.tmpN[q] = (uint32_t)s.pad_; }
  LANE0_PUBLISHED();
  int total = 0, bad = 0, streams = 0;
  long long needAll = need;   // the cut (W_MG_CUT): evicted entries cost no token — all of them count on top of `need`
  for (int q = 0; q < Q; q++) {
    const int t = UNI32(FL.tmpQ[q]); bad |= (int)UNI32(FL.tmpN[q]); if (FLANE == 0) FL.tmpA[q] = (uint32_t)total; total += t; if (t > 1) streams++;
    if (UNI32(FL.sKind[q]) && UNI32(FL.inHeap[q]) && UNI32(FL.hot[q].sLen) > UNI32(FL.hot[q].sPos)) needAll += t;
  }
  LANE0_PUBLISHED();
  if (!(bad || total < minEntries || total > mg.cap || streams < 1)) FOR_LANES(q, Q) mg.q[q] = describe(q, (int)FL.tmpA[q]);
  FOR_LANES(q, QCAPF) { FL.tmpQ[q] = 0; FL.tmpN[q] = 0; }
  LANE0_PUBLISHED();
  if (bad || total < minEntries || total > mg.cap || streams < 1) return 0;
  const bool cut = need < INT32_MAX && needAll + 2 * MG_SAMPLE < total && total < (1 << 16) * MG_SAMPLE && Q * MG_SPQ <= (1 << 16);
  if (FLANE == 0) { mg.stop[0] = 0xffffffffu; mg.stop[1] = (uint32_t)total; mg.stop[2] = cut ? (uint32_t)needAll : 0u; mg.stop[3] = 0; mg.cutPick[0] = 0xffffffffu; }
  LANE0_PUBLISHED();
  FAST_GLOBAL_FENCE();
  const long long t0_ = CLK();
  wgWide(d, W_MG_PACK, Q * MG_CPQ);
  wgWide(d, W_MG_FIX, Q);
  if (cut) {
    wgWide(d, W_MG_CUT, Q * MG_SPQ);
    const uint32_t pick = UNI32(*(volatile uint32_t*)&mg.cutPick[0]);
    if (pick != 0xffffffffu) {   // (none: fewer complete samples than entries needed — everything is ranked)
      const int si = (int)(pick & 0xffffu), sq = si / MG_SPQ, sj = si % MG_SPQ;
      if (FLANE == 0) { mg.cut[0] = mg.key[mg.q[sq].off + sj * MG_SAMPLE + MG_SAMPLE - 1]; mg.stop[3] = 1; }
      LANE0_PUBLISHED();
      FAST_GLOBAL_FENCE();
    }
  }
  const long long t1_ = CLK();
  wgWide(d, W_MG_RANK, total * ((Q + MG_PER - 1) / MG_PER));
  const long long t2_ = CLK();
  wgWide(d, W_MG_SCATTER, total);
  if (FLANE == 0) { RS.statSeg[37] += t1_ - t0_; RS.statSeg[38] += t2_ - t1_; RS.statSeg[39] += CLK() - t2_; }   // (always on: three clock reads per run) ticks of the key passes, the rank pass, the scatter
  LANE0_PUBLISHED();
  const uint32_t stop = UNI32(*(volatile uint32_t*)&mg.stop[0]);
  const int V = stop < (uint32_t)total ? (int)stop : total;
#ifdef ASCHED_HOSTSIM
  mgCheck(d, fc, Q, skip, V);
  if (getenv("HS_MG_TRACE")) {
    fprintf(stderr, "bulk merge: %d entries of %d queues, valid %d, skip %d, cut %d (need %lld)", total, streams, V, skip, (int)mg.stop[3], needAll);
    for (int q = 0; q < Q; q++) { const MgQ& s = mg.q[q]; if (!s.total) continue;
      if ((s.flags & MG_F_BARRIER) && mg.rank[s.off] == (int)stop) fprintf(stderr, " | barrier q%d gctx %d headKind %d headFast %d stage %d itEi %d evEnd %d itQi %d qEnd %d tokens %.0f eff %d sLen %d", q, FL.hot[q].gctx, FL.hot[q].headKind, FL.hot[q].headFast, FL.hot[q].itStage, FL.hot[q].itEi, FL.hot[q].evEnd, FL.hot[q].itQi, FL.hot[q].qEnd, FL.hot[q].tokens, FL.hot[q].effValid, FL.hot[q].sLen);
      if ((s.flags & MG_F_STREAM)) { int last = mg.rank[s.off + s.total - 1]; if ((s.flags & MG_F_OPEN) && last + 1 == (int)stop) fprintf(stderr, " | open q%d total %d kind %d", q, s.total, s.kind);
        for (int e = 0; e < s.total; e++) if (((mg.own[s.off + e] >> 30) & 1) && mg.rank[s.off + e] == (int)stop) fprintf(stderr, " | decrease q%d e %d", q, e); }
    }
    fprintf(stderr, "\n");
  }
#endif
  return V;
}
