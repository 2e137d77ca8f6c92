// mgpu.h — one pool on several GPUs: the words the collectives reduce, per element.
//
// There is no reference counterpart (the reference schedules a pool on one goroutine); what the words must ORDER like is the reference's:
// the node-partitioned first fit reduces RoundedNodeIndexKeyFromResourceList (nodedb/encoding.go:37-54) with MIN, the queue-hash round
// (BASELINE north_star) reduces per-node committed resources with SUM.  Every function here handles ONE element (one query, one result row,
// one node, one job): armada_sched_mgpu.hip runs them one element per thread, tests/hostsim runs them in serial loops.
#pragma once
#include "dev.h"

#ifndef MGPU_FN
#define MGPU_FN static inline
#endif
#ifndef MGPU_ADD64
#define MGPU_ADD64(p, v) (*(p) += (v))
#endif
#ifndef MGPU_ADD32
#define MGPU_ADD32(p, v) (*(p) += (v))
#endif
#ifndef MGPU_OR8
#define MGPU_OR8(p) (*(p) = 1)
#endif

#define MGPU_NO_NODE 0x7fffffffffffffffll
#define MGPU_NODE_MASK 0x0fffffffll
#define MGPU_LEVEL_SHIFT 28
#define MGPU_PRE_SHIFT 32

struct GlobalKeyLayout {
  int32_t nFields;
  int32_t bits[MAXK];
  int32_t rankBits;
  const int32_t* globalRank;   // [N] device-accessible, or NULL
  long long rankOffset;
};

// word of one query: localKey = what k_fit_batch's min-reduction left for the query's shape (~0: no node of this shard fits)
MGPU_FN long long mgpuPackQuery(const Dev& d, const GlobalKeyLayout& L, int level, unsigned long long localKey, int32_t* bad) {
  if (localKey == ~0ull) return MGPU_NO_NODE;
  const DevCfg& c = d.cfg;
  int rank = (int)(localKey & ((1ull << c.idxBits) - 1));
  int n = d.nodeByRank[rank];
  unsigned long long w = 0;
  for (int i = 0; i < c.K; i++) {
    long long q = d.alloc[((size_t)level * c.R + c.indexedCol[i]) * c.Npad + n] / c.indexedRes[i];   // the reference's own quotient, not the handle's biased / clamped field
    if (q < 0 || q >= (1ll << L.bits[i])) { *bad = 1; q = 0; }
    w = (w << L.bits[i]) | (unsigned long long)q;
  }
  long long g = L.globalRank ? (long long)L.globalRank[n] : L.rankOffset + rank;
  if (g < 0 || g >= (1ll << L.rankBits)) { *bad = 1; g = 0; }
  return (long long)((w << L.rankBits) | (unsigned long long)g);
}

// ---- queue-hash round: buf = [N*R committed | M job words]
// i-th scheduled result of this rank's round
MGPU_FN void mgpuDeltaScheduled(const Dev& d, long long* buf, int i) {
  const DevCfg& c = d.cfg;
  int j = d.resJob[i], n = d.resNode[i];
  if (d.jNode0[j] >= 0 || n < 0) return;   // an evicted job that was rescheduled: nothing new is committed
  for (int r = 0; r < c.R; r++) MGPU_ADD64(buf + (size_t)n * c.R + r, (long long)d.jReq[(size_t)j * c.R + r]);
  int level = 0;
  for (int l = 0; l < c.P; l++) if (c.prios[l] == d.resPrio[i]) level = l;
  MGPU_ADD64(buf + (size_t)c.N * c.R + j, (long long)(n + 1) | ((long long)level << MGPU_LEVEL_SHIFT));
}
MGPU_FN void mgpuDeltaPreempted(const Dev& d, long long* buf, int i) {
  MGPU_ADD64(buf + (size_t)d.cfg.N * d.cfg.R + d.resPreJob[i], 1ll << MGPU_PRE_SHIFT);
}

// resolve, step 1 (per node): what is free at the evicted priority on this replica after its own round
MGPU_FN void mgpuFreeInit(const Dev& d, long long* freeC, int n) {
  for (int r = 0; r < d.cfg.R; r++) freeC[(size_t)n * d.cfg.R + r] = d.alloc[((size_t)0 * d.cfg.R + r) * d.cfg.Npad + n];
}
// step 2 (per own scheduled result): ... plus what this rank's own new jobs took = free before anybody's new jobs, after this rank's preemptions
MGPU_FN void mgpuFreeOwnScheduled(const Dev& d, long long* freeC, int i) {
  int j = d.resJob[i], n = d.resNode[i];
  if (d.jNode0[j] >= 0 || n < 0) return;
  for (int r = 0; r < d.cfg.R; r++) MGPU_ADD64(freeC + (size_t)n * d.cfg.R + r, (long long)d.jReq[(size_t)j * d.cfg.R + r]);
}
MGPU_FN void mgpuOwnPreempted(const Dev& d, uint8_t* ownPre, int i) { MGPU_OR8(ownPre + d.resPreJob[i]); }
// step 3 (per job): ... plus what the OTHER ranks' preemptions free (every rank's preemptions are applied: a job somebody preempted is gone)
MGPU_FN void mgpuFreeForeignPreempted(const Dev& d, const long long* red, const uint8_t* ownPre, long long* freeC, int j) {
  const DevCfg& c = d.cfg;
  long long w = red[(size_t)c.N * c.R + j];
  if ((w >> MGPU_PRE_SHIFT) == 0 || ownPre[j]) return;
  int n = d.jNode0[j];
  if (n < 0) return;
  for (int r = 0; r < c.R; r++) MGPU_ADD64(freeC + (size_t)n * c.R + r, (long long)d.jReq[(size_t)j * c.R + r]);
}
// step 4 (per node)
MGPU_FN bool mgpuConflict(const Dev& d, const long long* red, const long long* freeC, uint8_t* conflict, int n) {
  bool over = false;
  for (int r = 0; r < d.cfg.R; r++) over = over || red[(size_t)n * d.cfg.R + r] > freeC[(size_t)n * d.cfg.R + r];
  conflict[n] = over;
  return over;
}
// step 5 (per job): a gang with a member on a conflict node is replayed as a whole (gang placement is atomic: gang_scheduler.go:100-148)
MGPU_FN void mgpuGangConflict(const Dev& d, const long long* red, const uint8_t* conflict, uint8_t* gangReplay, int j) {
  long long w = red[(size_t)d.cfg.N * d.cfg.R + j];
  int place = (int)(w & MGPU_NODE_MASK);
  if (place && conflict[place - 1] && d.jGang[j] >= 0) MGPU_OR8(gangReplay + d.jGang[j]);
}
// step 6 (per job): the job's node / priority on the accepted state, replay flag
// returns which summary counter the job belongs to (0: none, 1: accepted, 2: replay, 3: preempted) — the caller counts (per wave on the device: one atomic per wave, not per job)
MGPU_FN int mgpuJobOutcome(const Dev& d, const long long* red, const uint8_t* conflict, const uint8_t* gangReplay, int32_t* node, int32_t* prio, uint8_t* replay, int j) {
  const DevCfg& c = d.cfg;
  long long w = red[(size_t)c.N * c.R + j];
  int place = (int)(w & MGPU_NODE_MASK), level = (int)((w >> MGPU_LEVEL_SHIFT) & 15);
  node[j] = d.jNode0[j]; prio[j] = d.jRunPrio[j]; replay[j] = 0;
  if (place) {
    bool rp = conflict[place - 1] || (d.jGang[j] >= 0 && gangReplay[d.jGang[j]]);
    if (rp) { node[j] = -1; replay[j] = 1; return 2; }
    node[j] = place - 1; prio[j] = c.prios[level]; return 1;
  }
  if ((w >> MGPU_PRE_SHIFT) != 0) { node[j] = -1; return 3; }
  return 0;
}
