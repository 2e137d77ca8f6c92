// round_ft.h — fair-share preemption without a wide pass per job: the per-(shape, node) threshold table.
//
// selectNodeForJobWithFairPreemption (nodedb.go:935-1043) is, per node, "the evicted-table Index at which the node's own entries first cover
// the request" (fairNodeBest, round_ctl.h) and the answer is the maximum over the nodes.  Round 2 evaluated that for ALL nodes per preempting
// job: one wide pass over 127 helper workgroups on other XCDs, ~160 k clocks each at 100 000 nodes, plus an L2 write-back / invalidate on
// either side of it that made everything the control wave touched afterwards a miss (BASELINE configs[4]: 16.8 s per round).
//
// A job's question depends on its scheduling-key shape only (request vector, static mask row, priority), and a bind / preemption changes the
// answer of ONE node.  So: T[s][n] = fairNodeBest(shape s, node n) for every shape and node, built once per launch by a grid-wide pass
// (B_FT_NODE, shared with the helper workgroups), with a two-level maximum over it (B1: per 64 nodes, B2: per 4096 nodes; upper bounds, lowered
// lazily by the queries).  A query is three dependent wave-wide loads (B2 row -> B1 block -> T block); a node that changes is re-evaluated
// for all shapes by the control wave (ftUpdateNode: the node's entries one per lane, broadcast entry by entry, one shape per lane and slot).
//
// Staleness.  The generic code calls ftUpdateNode wherever it changes a node (updateKeysCtl) or brings an evicted-table entry back (txnAbort).
// The fast paths (node engine, bind wave, evicted jobs returning) do not: everything they do to a node — a bind at priority -2, the deletion of
// an evicted-table entry — can only LOWER its thresholds, so a stored value is an upper bound of the true one and a query validates its
// winner by re-evaluating that node: if the winner's value stands it is the true maximum (every other stored value bounds its node from
// above, Indexes are unique), else the entry is corrected and the query repeats.  The rare fast-path event that raises a threshold
// (a speculative commit of evicted jobs taken back, applyEvictedRange with sign -1) drops the table; it is rebuilt at the next query.
// The table never outlives a launch (ftValid is cleared at kernel start): the grid-wide phases between the passes rewrite planes wholesale.
#pragma once

#define FT_MAXS 512          // scheduling-key shapes the table covers
#define FT_CHUNK 8           // shapes one build item evaluates per walk over a node's entries
#define FT_T(d, s, n) ((d).ftT[(size_t)(s) * (d).cfg.Npad + (n)])
#define FT_B1(d, s, b) ((d).ftB1[(size_t)(s) * (d).ftNB1 + (b)])
#define FT_B2(d, s, b) ((d).ftB2[(size_t)(s) * 64 + (b)])

// fairNodeBest for shapes [s0, s1) of node n in one walk over the node's entries (build; one thread)
DEV void ftEvalChunk(const Dev& d, int n, int s0, int s1, int32_t* out) {
  const DevCfg& cf = d.cfg;
  int k0 = d.fairOff[n], k1 = d.fairOff[n + 1];
  int64_t av[FT_CHUNK][MAXR];
  bool open[FT_CHUNK];
  int nOpen = 0;
  for (int i = 0; i < s1 - s0; i++) {
    int s = s0 + i;
    out[i] = -1;
    open[i] = k0 < k1 && ((d.shapeMask[(size_t)s * cf.W + (n >> 6)] >> (n & 63)) & 1);
    if (open[i]) { nOpen++; for (int r = 0; r < MAXR; r++) av[i][r] = r < cf.R ? AL(d, cf.evLevel, r, n) : 0; }
  }
  for (int k = k0; k < k1 && nOpen > 0; k++) {
    int idx = d.fairEnt[k];
    if (!d.evTabAlive[idx]) continue;
    int ej = d.fairEntJob[k];
    int32_t ep = d.schedAtPrio[ej];
    const int64_t* er = JREQ(d, ej);
    for (int i = 0; i < s1 - s0; i++) {
      if (!open[i]) continue;
      if (ep == NO_PRIORITY) { out[i] = FAIR_BAD_ENTRY; open[i] = false; nOpen--; continue; }
      if (ep > d.ftPrio[s0 + i]) continue;
      const int64_t* rq = d.shapeReq + (size_t)(s0 + i) * cf.R;
      bool fits = true;
      for (int r = 0; r < MAXR; r++) if (r < cf.R) { av[i][r] += er[r]; if (rq[r] > av[i][r]) fits = false; }
      if (fits) { out[i] = idx; open[i] = false; nOpen--; }
    }
  }
}

DEV void ftBuildItem(Dev& d, int item) {   // B_FT_NODE: item = node * chunks + chunk
  int chunks = (d.ftS + FT_CHUNK - 1) / FT_CHUNK;
  int n = item / chunks, s0 = (item % chunks) * FT_CHUNK, s1 = s0 + FT_CHUNK < d.ftS ? s0 + FT_CHUNK : d.ftS;
  int32_t out[FT_CHUNK];
  ftEvalChunk(d, n, s0, s1, out);
  for (int i = 0; i < s1 - s0; i++) FT_T(d, s0 + i, n) = out[i];
}
DEV void ftBuildB1(Dev& d, int item) {     // B_FT_B1: item = shape * NB1 + block
  int s = item / d.ftNB1, b = item % d.ftNB1;
  int32_t m = -1;
  for (int j = 0; j < 64; j++) { int n = b * 64 + j; if (n < d.cfg.N) { int32_t v = FT_T(d, s, n); m = v > m ? v : m; } }
  FT_B1(d, s, b) = m;
}
DEV void ftBuildB2(Dev& d, int item) {     // B_FT_B2: item = shape * 64 + superblock
  int s = item / 64, sb = item % 64;
  int32_t m = -1;
  for (int j = 0; j < 64; j++) { int b = sb * 64 + j; if (b < d.ftNB1) { int32_t v = FT_B1(d, s, b); m = v > m ? v : m; } }
  FT_B2(d, s, sb) = m;
}

DEV_COLD void ftBuildAny(Dev& d, int phase, int item) { if (phase == 0) ftBuildItem(d, item); else if (phase == 1) ftBuildB1(d, item); else ftBuildB2(d, item); }

// The maxima are upper bounds: an update only ever RAISES them (two loads, two conditional stores); a value that fell leaves them too high until a
// query walks into the block, finds the true maximum below the stored one, lowers it and starts over (ftTop) — per shape, only for shapes that are asked.
DEV void ftRaise(Dev& d, int s, int n, int32_t t) {
  int b = n >> 6, sb = b >> 6;
  int32_t b1 = FT_B1(d, s, b), b2 = FT_B2(d, s, sb);
  if (t > b1) FT_B1(d, s, b) = t;
  if (t > b2) FT_B2(d, s, sb) = t;
}

#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
// a node's evicted-table entries, one per lane (three rounds of loads instead of a dependent chain per entry)
struct FtEntries { int cnt, eIdx, eAlive; int32_t ePrio; int64_t eReq[MAXR]; int64_t av0[MAXR]; };
DEV void ftStage(const Dev& d, int n, FtEntries& E) {
  const DevCfg& cf = d.cfg;
  int k0 = d.fairOff[n], lane = CTL_LANE();
  E.cnt = d.fairOff[n + 1] - k0;
  bool in = lane < E.cnt;
  E.eIdx = in ? d.fairEnt[k0 + lane] : -1;
  int eJob = in ? d.fairEntJob[k0 + lane] : 0;
  E.eAlive = in ? (int)d.evTabAlive[E.eIdx] : 0;
  E.ePrio = in ? d.schedAtPrio[eJob] : 0;
#pragma unroll
  for (int r = 0; r < MAXR; r++) { E.eReq[r] = (in && r < cf.R) ? JREQ(d, eJob)[r] : 0; E.av0[r] = r < cf.R ? AL(d, cf.evLevel, r, n) : 0; }
}
DEV int64_t ftBcast64(int64_t v, int e) {
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, e);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), e);
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}
#define FT_BATCH 4   // shapes per lane evaluated in one walk over the entries (S / 64 slots in batches of FT_BATCH)
#endif

// fairNodeBest of ONE shape on node n from the current planes and table (the validation of a query's winner)
DEV int32_t ftEvalOne(const Dev& d, int s, int n) {
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  const DevCfg& cf = d.cfg;
  if (CTL_WAVE() && d.fairOff[n + 1] - d.fairOff[n] <= 64) {
    if (!((d.shapeMask[(size_t)s * cf.W + (n >> 6)] >> (n & 63)) & 1)) return -1;
    FtEntries E; ftStage(d, n, E);
    int32_t myPrio = d.ftPrio[s];
    int64_t rq[MAXR], av[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) { rq[r] = r < cf.R ? d.shapeReq[(size_t)s * cf.R + r] : 0; av[r] = E.av0[r]; }
    for (int e = 0; e < E.cnt; e++) {   // every lane does the same arithmetic: the result is wave-uniform
      if (!__builtin_amdgcn_readlane(E.eAlive, e)) continue;
      int32_t bPrio = __builtin_amdgcn_readlane(E.ePrio, e);
      if (bPrio == NO_PRIORITY) return FAIR_BAD_ENTRY;
      if (bPrio > myPrio) continue;
      bool fits = true;
#pragma unroll
      for (int r = 0; r < MAXR; r++) { if (r >= cf.R) break; av[r] += ftBcast64(E.eReq[r], e); if (rq[r] > av[r]) fits = false; }
      if (fits) return __builtin_amdgcn_readlane(E.eIdx, e);
    }
    return -1;
  }
#endif
  int32_t t;
  ftEvalChunk(d, n, s, s + 1, &t);
  return t;
}

// node n changed: T[s][n] for every shape (and the maxima raised where a value rose)
DEV_COLD void ftUpdateNode(Dev& d, int n) {
  d.rs->statFt[2]++;
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  const DevCfg& cf = d.cfg;
  if (CTL_WAVE() && d.fairOff[n + 1] - d.fairOff[n] <= 64) {
    // entry by entry, broadcast to every lane; each lane accumulates for its own shapes — one shape per lane and slot, FT_BATCH slots per walk
    FtEntries E; ftStage(d, n, E);
    int lane = CTL_LANE();
    uint64_t mw[FT_MAXS / 64];   // (uniform) not used: the static bit is read per lane below
    (void)mw;
    int slots = (d.ftS + 63) >> 6;
    for (int base = 0; base < slots; base += FT_BATCH) {
      bool valid[FT_BATCH], open[FT_BATCH]; int32_t myPrio[FT_BATCH], t[FT_BATCH]; int64_t rq[FT_BATCH][MAXR], av[FT_BATCH][MAXR];
#pragma unroll
      for (int k = 0; k < FT_BATCH; k++) {   // all loads of the batch are independent: one memory round trip
        int s = (base + k) * 64 + lane;
        valid[k] = base + k < slots && s < d.ftS;
        int ss = valid[k] ? s : 0;
        open[k] = valid[k] && E.cnt > 0 && ((d.shapeMask[(size_t)ss * cf.W + (n >> 6)] >> (n & 63)) & 1);
        myPrio[k] = d.ftPrio[ss]; t[k] = -1;
#pragma unroll
        for (int r = 0; r < MAXR; r++) { rq[k][r] = r < cf.R ? d.shapeReq[(size_t)ss * cf.R + r] : 0; av[k][r] = E.av0[r]; }
      }
      for (int e = 0; e < E.cnt; e++) {          // wave-uniform trip count and exits: the broadcasts are cross-lane
        bool any = false;
#pragma unroll
        for (int k = 0; k < FT_BATCH; k++) any = any || open[k];
        if (!__ballot(any)) break;                // every shape of the batch has its answer
        if (!__builtin_amdgcn_readlane(E.eAlive, e)) continue;
        int32_t bPrio = __builtin_amdgcn_readlane(E.ePrio, e);
        int bIdx = __builtin_amdgcn_readlane(E.eIdx, e);
        if (bPrio == NO_PRIORITY) {
#pragma unroll
          for (int k = 0; k < FT_BATCH; k++) if (open[k]) { t[k] = FAIR_BAD_ENTRY; open[k] = false; }
          continue;
        }
        int64_t v[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; r++) v[r] = r < cf.R ? ftBcast64(E.eReq[r], e) : 0;
#pragma unroll
        for (int k = 0; k < FT_BATCH; k++) {
          bool take = open[k] && bPrio <= myPrio[k];
          bool fits = true;
#pragma unroll
          for (int r = 0; r < MAXR; r++) { if (r >= cf.R) break; if (take) { av[k][r] += v[r]; if (rq[k][r] > av[k][r]) fits = false; } }
          if (take && fits) { t[k] = bIdx; open[k] = false; }
        }
      }
#pragma unroll
      for (int k = 0; k < FT_BATCH; k++) if (valid[k]) { int s = (base + k) * 64 + lane; FT_T(d, s, n) = t[k]; if (t[k] >= 0) ftRaise(d, s, n, t[k]); }
    }
    return;
  }
#endif
  int lane0 = CTL_WAVE() ? CTL_LANE() : 0, stride = CTL_WAVE() ? 64 : 1;
  for (int s = lane0; s < d.ftS; s += stride) {   // (a node with more than 64 entries, and the CPU build)
    int32_t t;
    ftEvalChunk(d, n, s, s + 1, &t);
    FT_T(d, s, n) = t;
    if (t >= 0) ftRaise(d, s, n, t);
  }
}

// The table's current maximum for shape s: value and node (wave-uniform).  Walks down the upper bounds; a bound found too high is lowered to the
// true maximum of what is below it and the walk starts over.
DEV int32_t ftTop(Dev& d, int s, int* node) {
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  if (CTL_WAVE()) {
    int lane = CTL_LANE();
    for (;;) {
      int32_t v2 = FT_B2(d, s, lane);                    // (rows are 64 wide; unused super-blocks hold -1)
      int32_t m2 = waveMax32(v2);
      if (m2 < 0) { *node = -1; return -1; }
      int sb = __builtin_ctzll(__ballot(v2 == m2));
      int b0 = sb * 64 + lane;
      int32_t v1 = b0 < d.ftNB1 ? FT_B1(d, s, b0) : -1;
      int32_t m1 = waveMax32(v1);
      if (m1 < m2) { if (lane == 0) FT_B2(d, s, sb) = m1; continue; }
      int b = sb * 64 + __builtin_ctzll(__ballot(v1 == m1));
      int n0 = b * 64 + lane;
      int32_t v0 = n0 < d.cfg.N ? FT_T(d, s, n0) : -1;
      int32_t m0 = waveMax32(v0);
      if (m0 < m1) { if (lane == 0) FT_B1(d, s, b) = m0; continue; }
      *node = b * 64 + __builtin_ctzll(__ballot(v0 == m0));
      return m0;
    }
  }
#endif
  for (;;) {
    int32_t m2 = -1; int sb = -1;
    for (int j = 0; j < 64; j++) { int32_t v = FT_B2(d, s, j); if (v > m2) { m2 = v; sb = j; } }
    if (m2 < 0) { *node = -1; return -1; }
    int32_t m1 = -1; int b = -1;
    for (int j = 0; j < 64; j++) { int x = sb * 64 + j; if (x < d.ftNB1) { int32_t v = FT_B1(d, s, x); if (v > m1) { m1 = v; b = x; } } }
    if (m1 < m2) { FT_B2(d, s, sb) = m1; continue; }
    int32_t m0 = -1; int nb = -1;
    for (int j = 0; j < 64; j++) { int x = b * 64 + j; if (x < d.cfg.N) { int32_t v = FT_T(d, s, x); if (v > m0) { m0 = v; nb = x; } } }
    if (m0 < m1) { FT_B1(d, s, b) = m0; continue; }
    *node = nb;
    return m0;
  }
}

// max over nodes of fairNodeBest for shape s, validated against the current state (see "Staleness" above); -1 = no node
DEV_COLD int ftQuery(Dev& d, int s) {
  d.rs->statFt[0]++;
  for (int tries = 0; tries < 1 << 22; tries++) {
    int node;
    int32_t m = ftTop(d, s, &node);
    if (m < 0) return -1;
    int32_t now = ftEvalOne(d, s, node);     // the winner from the planes and the table as they are now
    if (now == m) return m;
    if (now > m) { d.rs->ftValid = 0; return -2; }   // a stored value must bound its node from above: the table is dropped, the wide pass answers
    ftUpdateNode(d, node);                    // stale (a fast-path bind or a rescheduled evicted job lowered it): the node is stale for every shape — corrected for all of
    d.rs->statFt[1]++;                        // them at once, or each shape's queries would meet it again; the bounds above repair themselves
  }
  d.rs->ftValid = 0;
  return -2;
}

// txnAbort's hook: evicted-table entries that came back raise their node's thresholds — re-evaluate those nodes once everything is restored
DEV_COLD void ftAfterAbort(Dev& d, int undoCount) {
  for (int i = undoCount - 1; i >= 0; i--)
    if ((d.undo[i * 4] & 255) == 3 /* U_EVTAB_DEL */) { int n = d.jcAssigned[d.evTabJob[d.undo[i * 4 + 1]]]; if (n >= 0) ftUpdateNode(d, n); }
}
