// submit_gang.h — the submit check's gang units, one WORKGROUP per unit (internal/scheduler/submitcheck.go:345-349:
//   txn := nodeDb.Txn(true); ok, _, err := nodeDb.ScheduleManyWithTxn(txn, gctx); txn.Abort()).
// Units never see each other (every transaction is aborted), so they run side by side; inside a unit the members go one after the other because each one's bind is what the
// next one's first fit sees (nodedb.go:417-462).  On a pristine NodeDb (every priority plane == allocatable, nothing evicted: asched_host.inc `allocPristine`) a member's
// SelectNodeForJobWithTxn is "first node of its mask row, in key order, that fits at priority -2" — if that finds nothing, every later step of the cascade sees the same numbers
// minus the same members' requests (all members of a gang share a priority class, hence a cutoff) and fails too (nodedb.go:724-789).  So a unit is: for each member a
// pass over the nodes — the pristine planes and keys for nodes no earlier member sits on, the planes minus what those members took (and the key recomputed from them,
// round_ctl.h packKey) for the few that one does — then a bind recorded in the workgroup's own scratch: no HBM state changes, nothing to abort.
// The host sends here only what this argument covers (asched_submit_check: home scheduling on, no usable away type, no row on the literal iteration path, one priority class
// per unit, at most SG_TMAX members); everything else keeps the sequential path.  armada_sched_mgpu.hip runs one unit per workgroup; tests/hostsim runs the same function serially.
#pragma once
#include "dev.h"

struct SgShared {
  int32_t tn[SG_TMAX];             // nodes earlier members of this unit were bound to
  int64_t td[SG_TMAX * MAXR];      // what they took there
  unsigned long long wmin[16];     // per-wave minima of the reduction
  int32_t nT, pad;
};

#ifndef SG_FN
#define SG_FN static inline
#define SG_TID 0
#define SG_NT 1
#define SG_SYNC() do {} while (0)
#define SG_WGMIN(s, v) (v)
#endif

// packKey (round_ctl.h) of node n at level 0 with `took` off its planes; *bad: a field outside the layout (the host redoes the unit on the sequential path)
SG_FN unsigned long long sgKey(const Dev& d, int n, const int64_t* took, int32_t* bad) {
  const DevCfg& c = d.cfg;
  unsigned long long k = (unsigned long long)d.idxRank[n];
  for (int i = 0; i < c.K; i++) {
    int col = c.indexedCol[i];
    int64_t q = (d.alloc[(size_t)col * c.Npad + n] - took[col]) / c.indexedRes[i];
    int64_t f = q - c.keyLo[i];
    if (f < 0 && c.keyClamp) f = 0;
    if (f < 0 || (c.keyWidth[i] < 63 && f >= ((int64_t)1 << c.keyWidth[i]))) { *bad = 1; f = 0; }
    k |= (unsigned long long)f << c.keyShift[i];
  }
  return k;
}

// one unit: jobs[0..n); bits = a zeroed bitmap over the nodes (left zeroed); out = {ok (-1: redo on the sequential path), scheduled_away, num_schedulable, first_node}
SG_FN void submitGangUnit(const Dev& d, const int32_t* jobs, int n, SgShared& s, uint32_t* bits, int32_t* out) {
  const DevCfg& c = d.cfg;
  if (SG_TID == 0) { s.nT = 0; s.pad = 0; }   // (pad: a key field fell outside the layout)
  SG_SYNC();
  int ok = 1, nSched = 0, first = -1;
  for (int m = 0; m < n && ok == 1; m++) {
    const int job = jobs[m];
    const int64_t* req = d.jReq + (size_t)job * c.R;
    const uint64_t* mask = d.shapeMask + (size_t)d.jShape[job] * c.W;
    unsigned long long best = ~0ull;
    for (int node = SG_TID; node < c.N; node += SG_NT) {   // nodes nobody of this unit sits on: pristine planes, pristine keys
      if (!((mask[node >> 6] >> (node & 63)) & 1) || ((bits[node >> 5] >> (node & 31)) & 1)) continue;
      bool fit = true;
      for (int r = 0; r < c.R; r++) fit = fit && d.alloc[(size_t)r * c.Npad + node] >= req[r];
      if (fit) { unsigned long long k = d.keys[node]; if (k < best) best = k; }
    }
    const int nT = s.nT;
    for (int t = SG_TID; t < nT; t += SG_NT) {             // nodes an earlier member was bound to
      int node = s.tn[t];
      if (!((mask[node >> 6] >> (node & 63)) & 1)) continue;
      bool fit = true;
      for (int r = 0; r < c.R; r++) fit = fit && d.alloc[(size_t)r * c.Npad + node] - s.td[t * MAXR + r] >= req[r];
      if (fit) { unsigned long long k = sgKey(d, node, &s.td[t * MAXR], &s.pad); if (k < best) best = k; }
    }
    best = SG_WGMIN(s, best);
    if (best == ~0ull) { ok = 0; break; }
    const int node = d.nodeByRank[best & ((1ull << c.idxBits) - 1)];
    if (SG_TID == 0) {   // BindJobToNode (nodedb.go:1046-1068) into the unit's scratch
      int t = 0;
      while (t < s.nT && s.tn[t] != node) t++;
      if (t == s.nT) { s.tn[t] = node; for (int r = 0; r < MAXR; r++) s.td[t * MAXR + r] = 0; s.nT = t + 1; bits[node >> 5] |= 1u << (node & 31); }
      for (int r = 0; r < c.R; r++) s.td[t * MAXR + r] += req[r];
    }
    SG_SYNC();
    if (m == 0) first = node;
    nSched++;
  }
  SG_SYNC();
  if (SG_TID == 0) {
    for (int t = 0; t < s.nT; t++) bits[s.tn[t] >> 5] = 0;   // (every set bit of such a word belongs to this unit)
    out[0] = s.pad ? -1 : ok; out[1] = 0; out[2] = nSched; out[3] = first;
  }
  SG_SYNC();
}

// ---- uniform units: every member has the same scheduling-key shape (the usual gang: identical replicas; the reference's own BenchmarkScheduleMany*, nodedb_test.go:1590-1712, is
// ONE such unit of up to 64 000 members).  On a pristine NodeDb the members fill the nodes in ascending key order: the node a member was bound to keeps the smallest key among the
// nodes that fit (a bind only lowers it; nobody else moves) until it no longer fits, and never fits again.  So the unit's outcome is a sum, not a sequence: node n takes
// cap_n = min over requested columns of floor(allocatable / request) members, num_schedulable = min(members, sum of cap_n over the row's nodes), ok = that sum covers the unit,
// first_node = the smallest key with cap_n >= 1 — per SHAPE, whatever the number of units or members: one pass over the nodes (k_fit_capacity), no per-member work at all.
#define SG_CAP_CLAMP (1ll << 31)   // a node's capacity as it enters the sum (N * 2^31 < 2^63)
SG_FN long long sgNodeCapacity(const Dev& d, int shape, int node) {
  const DevCfg& c = d.cfg;
  const uint64_t* mask = d.shapeMask + (size_t)shape * c.W;
  if (!((mask[node >> 6] >> (node & 63)) & 1)) return 0;
  const int64_t* req = d.shapeReq + (size_t)shape * c.R;
  long long cap = SG_CAP_CLAMP;
  for (int r = 0; r < c.R; r++) {
    long long a = d.alloc[(size_t)r * c.Npad + node];
    if (a < req[r]) return 0;                       // (also a negative column under a zero request: DynamicJobRequirementsMet fails, nodematching.go:194-197)
    if (req[r] > 0) { long long q = a / req[r]; if (q < cap) cap = q; }
  }
  return cap;
}
