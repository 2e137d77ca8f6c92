// replay_rank.h — addEvictedJobsToNodeDb (preempting_queue_scheduler.go:589-639) without the walk.
//
// The reference replays the DRF order over the evicted jobs once per evictor to give every evicted job its evicted-table Index (the order fair-share preemption
// takes victims in, nodedb.go:935-1043).  That replay is a k-way heap merge of the queues' evicted streams on QueueCandidateGangIteratorPQ.Less — and the costs that
// order it are functions of each queue's own allocation prefix, which B_EVKEYS (round_run.h) has computed for every evicted-list position.  When every stream is
// gang-free (evCheap) and its keys never decrease along the stream (evMono), the heap merge of the streams IS the sort of all entries by (packed key, queue-name rank,
// position): an entry's Index is the number of entries that precede it, i.e. its offset within its own queue plus, per other queue, a binary search in that queue's
// key sequence.  One element per thread, every CU busy, ~1 ms for 900 000 evicted jobs — the serial walk (one heap step per job on one wave of the round kernel) cost
// 2.4 s there and was 84 % of the production-shaped round (profiles/r05e…).  Exactly the table the walk would build: same Index per job.
//
// One element per call: armada_sched_mgpu.hip runs them one per thread, tests/hostsim serially.  The kernels live outside the round kernel's code object.
#pragma once
#include "dev.h"
#ifndef MGPU_FN
#define MGPU_FN static inline
#endif

struct RrKey { uint32_t A; uint64_t X, Y; };
// == round_ctl.h packKey3 (Less, queue_scheduler.go:738-798, as a lexicographic key; exact for finite non-negative costs) — keep the two in step
MGPU_FN RrKey rrPack(int preferLarge, int32_t prio, double proposed, double current, double size, double budget) {
  RrKey o;
  o.A = ~((uint32_t)prio ^ 0x80000000u);
  if (preferLarge) {
    if (proposed <= budget) { o.X = __builtin_bit_cast(uint64_t, current); o.Y = ~__builtin_bit_cast(uint64_t, size); }
    else { o.X = __builtin_bit_cast(uint64_t, proposed) | (1ull << 63); o.Y = 0; }
  } else { o.X = __builtin_bit_cast(uint64_t, proposed); o.Y = 0; }
  return o;
}
MGPU_FN bool rrLess(const RrKey& a, const RrKey& b) { return a.A != b.A ? a.A < b.A : a.X != b.X ? a.X < b.X : a.Y < b.Y; }
MGPU_FN RrKey rrKeyAt(const Dev& d, int p, double budget) {
  const EvKey e = d.evKey[p];
  return rrPack(d.cfg.preferLarge, e.pcPrio, e.proposed, e.current, e.size, budget);   // (the replay compares priority-class priorities: compareSchedulingPriority is pass 2's, pqs.go:603-604)
}
// May the table be built by rank?  The evictor left the replay pending (every stream gang-free, costs precomputed: SM_MONO) and every non-empty stream is key-monotone.
MGPU_FN bool rrOk(const Dev& d, int n) {
  if (n <= 0 || !d.rs->replayPending || !d.evMono || !d.evKey || !d.evCheap) return false;
  for (int q = 0; q < d.cfg.Q; q++) if (d.evOff[q + 1] > d.evOff[q] && (!d.evMono[q] || !d.evCheap[q])) return false;
  return true;
}
MGPU_FN int rrRank(const Dev& d, int p) {
  const int Q = d.cfg.Q;
  int lo = 0, hi = Q;                                   // the queue whose segment holds p: the last q with evOff[q] <= p
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (d.evOff[mid] <= p) lo = mid; else hi = mid; }
  const int q = lo;
  const RrKey me = rrKeyAt(d, p, d.qDc[q] / d.qWeight[q]);   // budget: pushQueue (queue_scheduler.go:509-519) == passInit's pqBudget
  const int32_t myName = d.qNameRank[q];
  int rank = p - d.evOff[q];                            // its own queue's earlier entries (the stream is served in order; keys do not decrease along it)
  for (int o = 0; o < Q; o++) {
    if (o == q) continue;
    int s = d.evOff[o], e = d.evOff[o + 1];
    if (s >= e) continue;
    const double budget = d.qDc[o] / d.qWeight[o];
    const bool tiesFirst = d.qNameRank[o] < myName;     // equal keys: Less falls through to the queue name
    int a = s, b = e;
    while (a < b) {                                     // entries of queue o that are served before p
      int mid = (a + b) >> 1;
      const RrKey k = rrKeyAt(d, mid, budget);
      bool before = tiesFirst ? !rrLess(me, k) : rrLess(k, me);
      if (before) a = mid + 1; else b = mid;
    }
    rank += a - s;
  }
  return rank;
}
MGPU_FN void rrElem(const Dev& d, int p) {
  const int rank = rrRank(d, p);
  const int job = d.evList[p];
  d.evTabJob[rank] = job; d.evTabAlive[rank] = 1; d.evIndexOfJob[job] = rank;   // evTabInsert (round_ctl.h)
  if (d.evIdxByPos) d.evIdxByPos[p] = rank;                                      // B_EVIDX
}
MGPU_FN void rrFinish(const Dev& d, int n) {
  d.rs->evictedTableSize = n; d.rs->fairIndexValid = 0; d.rs->ftValid = 0;
  d.rs->replayPending = 0;   // the table exists: nothing is deferred (CMD_PASS1 / CMD_PASS2 see evictedTableSize == their evicted count and do not walk)
}
