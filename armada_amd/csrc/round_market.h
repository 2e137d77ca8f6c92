// round_market.h — market-driven ordering, the iterator half, on the device.
//
// The reference's market mode replaces the cost-based candidate iterator by MarketBasedCandidateGangIterator (market_iterator.go:32-295: a container/heap over
// MarketIteratorPQ.Less, which is NOT a strict weak order — its round-robin clause reads the queue and the price of the previous result, so the heap's
// internal layout matters and heap.Push / heap.Pop are restated move by move), orders a queue's jobs by jobdb.MarketSchedulingOrderCompare
// (jobdb/comparison.go:113-170) and merges a queue's evicted and queued jobs with MarketDrivenMultiJobsIterator (jobiteration.go:232-321).  These three are
// what follows, as wave-uniform control code run by the auxiliary kernel (CMD_MARKET), pinned by the reference's own tests through the same hooks the oracle
// is pinned by (tests/test_zzz_market_iterator.py).  The round around them — the evict-everything node evictor (pqs.go:117-119), spot price and second-price
// billing (queue_scheduler.go:177-203), the indicative pricer — is not built (HISTORY.md §9).
#pragma once

struct MarketArgs {
  int32_t op, nq, n1, n2, preferHome, onlyAfter, pad0, pad1;
  const int32_t* nameRank; const int32_t* off; const asched_market_job* jobs;      // op 0: per queue name rank, CSR of the queues' job lists
  const asched_market_cmp_job* l1; const asched_market_cmp_job* l2;               // op 1: l1[0] vs l2[0]; op 2: the two lists
  const uint8_t* ev1; const uint8_t* ev2;
  int32_t* out;      // op 0: [off[nq]] queue per Peek; op 1: [1] sign; op 2: [n1 + n2 + 1] yielded jobs, then their number
  int32_t* work;     // op 0: [2 * nq] heap, per-queue position
};

// MarketIteratorPQ.Less (market_iterator.go:228-273) for items = (queue qa at job ja) and (queue qb at job jb)
DEV bool marketLess(const MarketArgs& a, int qa, int ja, int qb, int jb, double prevCost, int prevRank) {
  const asched_market_job& x = a.jobs[ja]; const asched_market_job& y = a.jobs[jb];
  int ra = a.nameRank[qa], rb = a.nameRank[qb];
  if (a.preferHome && (x.away != 0) != (y.away != 0)) return x.away == 0;   // :231-233
  if (x.price != y.price) return x.price > y.price;                          // :236-238
  if ((x.queued != 0) != (y.queued != 0)) return x.queued == 0;              // :242-244
  if (x.price == prevCost) {                                                 // :249-259 round robin between queues bidding the same price
    if (ra > prevRank && rb > prevRank) return ra < rb;
    if (ra > prevRank || rb == prevRank) return true;
    if (rb > prevRank || ra == prevRank) return false;
  }
  if (x.runtime != y.runtime) return x.runtime > y.runtime;                  // :262-264
  if (x.submit_time != y.submit_time) return x.submit_time < y.submit_time;  // :267-269
  return ra < rb;                                                            // :272
}

// NewMarketCandidateGangIterator + (Peek, Clear)* until the queues run dry (:38-101): out[i] = the queue of the i-th Peek
DEV void marketIterate(const MarketArgs& a) {
  int32_t* heap = a.work; int32_t* pos = a.work + a.nq;
  int hn = 0, n = 0;
  double prevCost = 0.0; int prevRank = -1;   // previousResultCost, previousResultQueue ("" orders before every name)
#define MLESS(i, j) marketLess(a, heap[i], pos[heap[i]], heap[j], pos[heap[j]], prevCost, prevRank)
  for (int q = 0; q < a.nq; q++) {
    pos[q] = a.off[q];
    if (pos[q] >= a.off[q + 1]) continue;
    heap[hn++] = q;                                                                                   // heap.Push: append, up
    for (int j = hn - 1;;) { int i = (j - 1) / 2; if (i == j || !MLESS(j, i)) break; int t = heap[i]; heap[i] = heap[j]; heap[j] = t; j = i; }
  }
  while (hn > 0) {
    int top = heap[0];
    a.out[n++] = top;                                                                                 // Peek :91-101
    int last = hn - 1;                                                                                // Clear :74-89: heap.Pop = swap(0, n - 1), down(0, n - 1), drop the last
    { int t = heap[0]; heap[0] = heap[last]; heap[last] = t; }
    for (int i = 0;;) {
      int j1 = 2 * i + 1;
      if (j1 >= last || j1 < 0) break;
      int j = j1, j2 = j1 + 1;
      if (j2 < last && MLESS(j2, j1)) j = j2;
      if (!MLESS(j, i)) break;
      int t = heap[i]; heap[i] = heap[j]; heap[j] = t; i = j;
    }
    hn = last;
    prevRank = a.nameRank[top]; prevCost = a.jobs[pos[top]].price;                                    // the item's values as of this Peek
    pos[top]++;                                                                                       // item.it.Clear(), updatePQItem :108-135
    if (pos[top] < a.off[top + 1]) {
      heap[hn++] = top;
      for (int j = hn - 1;;) { int i = (j - 1) / 2; if (i == j || !MLESS(j, i)) break; int t = heap[i]; heap[i] = heap[j]; heap[j] = t; j = i; }
    }
  }
#undef MLESS
}

// jobdb.MarketSchedulingOrderCompare (jobdb/comparison.go:113-170)
DEV int marketCompare(const asched_market_cmp_job& a, const asched_market_cmp_job& b) {
  if (a.id_rank == b.id_rank) return 0;                                                             // :116-118
  if (a.pc_priority != b.pc_priority) return a.pc_priority > b.pc_priority ? -1 : 1;                 // :122-126
  if (a.bid_price != b.bid_price) return a.bid_price > b.bid_price ? -1 : 1;                         // :128-135
  bool ja = a.active != 0, jb = b.active != 0;                                                       // :141-155
  if (ja || jb) {
    if (!jb) return -1;
    if (!ja) return 1;
    if (a.active_run_timestamp != b.active_run_timestamp) return a.active_run_timestamp < b.active_run_timestamp ? -1 : 1;
  }
  if (a.submit_time != b.submit_time) return a.submit_time < b.submit_time ? -1 : 1;                 // :158-162
  return a.id_rank < b.id_rank ? -1 : 1;                                                             // :166-170
}

// InMemoryJobIterator.Next (jobiteration.go:34-51)
DEV int marketMemNext(const uint8_t* ev, int n, int* i, bool only) {
  if (*i >= n) return -1;
  int v = (*i)++;
  if (only && !ev[v]) { while (!ev[v]) { if (*i >= n) return -1; v = (*i)++; } }
  return v;
}
// MarketDrivenMultiJobsIterator (jobiteration.go:232-321) over two InMemoryJobIterators; OnlyYieldEvicted before the (onlyAfter + 1)-th Next
DEV void marketMultiIterate(const MarketArgs& a) {
  int i1 = 0, i2 = 0, v1 = -1, v2 = -1, n = 0;
  bool have1 = false, have2 = false, only = false;
  for (int step = 0;; step++) {
    if (a.onlyAfter >= 0 && step >= a.onlyAfter && !only) {   // OnlyYieldEvicted :296-309
      only = true;
      if (have1 && !a.ev1[v1]) have1 = false;
      if (have2 && !a.ev2[v2]) have2 = false;
    }
    if (!have1) { v1 = marketMemNext(a.ev1, a.n1, &i1, only); have1 = v1 >= 0; }      // Next :250-294
    if (!have2) { v2 = marketMemNext(a.ev2, a.n2, &i2, only); have2 = v2 >= 0; }
    if (have1 && have2) {
      if (marketCompare(a.l1[v1], a.l2[v2]) < 0) { a.out[n++] = v1; have1 = false; } else { a.out[n++] = a.n1 + v2; have2 = false; }
    } else if (have1) { a.out[n++] = v1; have1 = false; }
    else if (have2) { a.out[n++] = a.n1 + v2; have2 = false; }
    else break;
  }
  a.out[a.n1 + a.n2] = n;
}
