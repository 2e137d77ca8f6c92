// round_mkt.h — the market-driven ROUND on the device (asched_set_market): what a market pool does differently inside
// PreemptingQueueScheduler.Schedule, restated over the round's own queue / gang iterators.
//
//   * MarketBasedCandidateGangIterator (market_iterator.go:22-201) instead of the cost-based iterator: a literal container/heap over MarketIteratorPQ.Less
//     (:225-273).  Less reads the previous result (round robin between queues that bid the same price), so it is not a strict weak order and the array layout of
//     the heap decides: Push / Pop / Init / Fix / Remove move elements exactly as the Go standard library does.  Items are queues (at most one per queue).
//   * the spot price, the billable resource and the second-price override (queue_scheduler.go:177-203, context/queue.go:108-119).
// The iterator half of market mode that works on caller-supplied lists (the reference's iterator tests) is round_market.h.
//
// Compiled only where ASCHED_MARKET_ROUND is defined: the auxiliary kernel (CMD_MARKET_ROUND runs the whole round there, one launch, no helper workgroups) and
// the CPU build of the tests.  The round kernel's translation unit does not see any of it (its code is the measured one and is placement-sensitive: DESIGN.md 9).
#pragma once
#ifdef ASCHED_MARKET_ROUND

// the market state of the launch in progress: a kernel argument of the auxiliary kernel kept in its LDS (armada_sched.hip g_mk), a global of the CPU build
DEV MktDev* mktDev();
#define MKD (*mktDev())
#define MKS (*mktDev()->s)
DEV bool mkOn(const Dev&) { const MktDev* m = mktDev(); return m->s != nullptr && m->s->market != 0; }
DEV bool pqAway(Dev& d, int q);
// MarketIteratorPQ.Less (market_iterator.go:225-273) between the items of queues a and b
DEV bool mkLess(Dev& d, int a, int b) {
  int ra = d.qNameRank[a], rb = d.qNameRank[b];
  if (d.cfg.preferHome) { bool aa = pqAway(d, a), ab = pqAway(d, b); if (aa != ab) return !aa; }            // :228-232
  double pa = MKD.pqPrice[a], pb = MKD.pqPrice[b];
  if (pa != pb) return pa > pb;                                                                               // :235-237
  if (MKD.pqQueued[a] != MKD.pqQueued[b]) return !MKD.pqQueued[a];                                        // :241-243
  if (pa == MKS.prevCost) {                                                                               // :248-258
    int pr = MKS.prevRank;
    if (ra > pr && rb > pr) return ra < rb;
    if (ra > pr || rb == pr) return true;
    if (rb > pr || ra == pr) return false;
  }
  if (MKD.pqRuntime[a] != MKD.pqRuntime[b]) return MKD.pqRuntime[a] > MKD.pqRuntime[b];                  // :261-263
  if (MKD.pqSubmit[a] != MKD.pqSubmit[b]) return MKD.pqSubmit[a] < MKD.pqSubmit[b];                      // :266-268
  return ra < rb;                                                                                             // :271
}
// container/heap (heap.go) on MKD.heap[0 .. mkHeapN)
DEV void mkSwap(Dev& d, int i, int j) { int t = MKD.heap[i]; MKD.heap[i] = MKD.heap[j]; MKD.heap[j] = t; }
DEV void mkUp(Dev& d, int j) { for (;;) { int i = (j - 1) / 2; if (i == j || !mkLess(d, MKD.heap[j], MKD.heap[i])) break; mkSwap(d, i, j); j = i; } }
DEV bool mkDown(Dev& d, int i0, int n) {
  int i = i0;
  for (;;) {
    int j1 = 2 * i + 1;
    if (j1 >= n || j1 < 0) break;
    int j = j1, j2 = j1 + 1;
    if (j2 < n && mkLess(d, MKD.heap[j2], MKD.heap[j1])) j = j2;
    if (!mkLess(d, MKD.heap[j], MKD.heap[i])) break;
    mkSwap(d, i, j); i = j;
  }
  return i > i0;
}
DEV void mkPush(Dev& d, int q) { MKD.heap[MKS.heapN++] = q; d.pqInHeap[q] = 1; mkUp(d, MKS.heapN - 1); }
DEV int mkPop(Dev& d) { int n = MKS.heapN - 1; mkSwap(d, 0, n); mkDown(d, 0, n); MKS.heapN = n; int q = MKD.heap[n]; d.pqInHeap[q] = 0; return q; }
DEV void mkInit(Dev& d) { int n = MKS.heapN; for (int i = n / 2 - 1; i >= 0; i--) mkDown(d, i, n); }
DEV void mkRemove(Dev& d, int i) {
  int n = MKS.heapN - 1;
  if (n != i) { mkSwap(d, i, n); if (!mkDown(d, i, n)) mkUp(d, i); }
  d.pqInHeap[MKD.heap[n]] = 0; MKS.heapN = n;
}
DEV void mkFix(Dev& d, int i) { if (!mkDown(d, i, MKS.heapN)) mkUp(d, i); }
DEV int mkTop(Dev& d) { return MKS.heapN > 0 ? MKD.heap[0] : -1; }
// updatePQItem (market_iterator.go:108-135) for the gang `ref` at the head of queue q
DEV void mkItemOf(Dev& d, int q, int ref, int firstJob) {
  (void)ref;
  MKD.pqPrice[q] = MKD.jBid ? MKD.jBid[firstJob] : 0.0;                 // job.GetBidPrice(pool): resolved by the caller (asched_jobs.bid_price)
  bool queued = d.jNode0[firstJob] < 0;                                    // job.Queued(): the jobDb's view
  MKD.pqQueued[q] = queued;
  MKD.pqRuntime[q] = queued ? 0 : -MKD.jRunTs[firstJob];                 // time.Now() - run.Created(): compared between running jobs only, where it orders like -created
  MKD.pqSubmit[q] = MKD.jSubmit[firstJob];
}
// SecondPrice (:137-151): highest remaining bid of another queue, away contexts ignored ("<queue>-away" contexts are exactly the items whose gang is an away gang)
DEV double mkSecondPrice(Dev& d, int priceSettingQueue) {
  double second = 0.0;
  for (int i = 0; i < MKS.heapN; i++) { int q = MKD.heap[i]; if (q == priceSettingQueue || pqAway(d, q)) continue; if (MKD.pqPrice[q] > second) second = MKD.pqPrice[q]; }
  return second;
}
#define MK(x) x
#else
#define MK(x)
#endif
