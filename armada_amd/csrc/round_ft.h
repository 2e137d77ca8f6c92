// round_ft.h — fair-share preemption without a wide pass per job: the per-(shape, node) threshold table.
//
// selectNodeForJobWithFairPreemption (nodedb.go:935-1043) is, per node, "the evicted-table Index at which the node's own entries first cover
// the request" (fairNodeBest, round_ctl.h) and the answer is the maximum over the nodes.  Round 2 evaluated that for ALL nodes per preempting
// job: one wide pass over 127 helper workgroups on other XCDs, ~160 k clocks each at 100 000 nodes, plus an L2 write-back / invalidate on
// either side of it that made everything the control wave touched afterwards a miss (BASELINE configs[4]: 16.8 s per round).
//
// A job's question depends on its scheduling-key shape only (request vector, static mask row, priority), and a bind / preemption changes the
// answer of ONE node.  So: T[s][n] = fairNodeBest(shape s, node n) for every shape and node, built once per launch by a grid-wide pass
// (B_FT_NODE, shared with the helper workgroups), with a two-level maximum over it (B1: per 64 nodes, B2: per 4096 nodes).  A query is three
// dependent wave-wide loads (B2 row -> B1 block -> T block); a node that changes is re-evaluated for all shapes by the control wave
// (ftUpdateNode: the node's entries one per lane, broadcast entry by entry, one shape per lane and slot).
//
// Staleness.  The generic code calls ftUpdateNode wherever it changes a node (updateKeysCtl) or brings an evicted-table entry back (txnAbort).
// The fast paths (node engine, bind wave, evicted jobs returning) do not: everything they do to a node — a bind at priority -2, the deletion of
// an evicted-table entry — can only LOWER its thresholds, so a stored value is an upper bound of the true one and a query validates its
// winner by re-evaluating that node: if the winner's value stands it is the true maximum (every other stored value bounds its node from
// above, Indexes are unique), else the structure has been corrected and the query repeats.  The rare fast-path event that raises a threshold
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

// after T[s][n] went from `old` to `t`: the two maxima above it
DEV void ftFixMax(Dev& d, int s, int n, int32_t old, int32_t t) {
  int b = n >> 6, sb = b >> 6;
  int32_t b1 = FT_B1(d, s, b);
  if (t > b1) {
    FT_B1(d, s, b) = t;
    if (t > FT_B2(d, s, sb)) FT_B2(d, s, sb) = t;
    return;
  }
  if (old != b1 || t >= old) return;       // the node was not its block's maximum (Indexes are unique; -1 / -1 changes nothing)
  int32_t m = -1;
  for (int j = 0; j < 64; j++) { int x = b * 64 + j; if (x < d.cfg.N) { int32_t v = x == n ? t : FT_T(d, s, x); m = v > m ? v : m; } }
  FT_B1(d, s, b) = m;
  if (FT_B2(d, s, sb) != old) return;
  int32_t m2 = -1;
  for (int j = 0; j < 64; j++) { int x = sb * 64 + j; if (x < d.ftNB1) { int32_t v = x == b ? m : FT_B1(d, s, x); m2 = v > m2 ? v : m2; } }
  FT_B2(d, s, sb) = m2;
}

// node n changed: T[s][n] for every shape, and the maxima
DEV_COLD void ftUpdateNode(Dev& d, int n) {
  const DevCfg& cf = d.cfg;
  d.rs->statFt[2]++;
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  int k0 = d.fairOff[n], cnt = d.fairOff[n + 1] - k0;
  if (CTL_WAVE() && cnt <= 64) {
    // the node's entries, one per lane (three rounds of loads instead of a dependent chain per entry); then entry by entry, broadcast to every lane,
    // each lane accumulating for its own shape — one shape per lane and slot, S / 64 slots
    int lane = CTL_LANE();
    bool in = lane < cnt;
    int eIdx = in ? d.fairEnt[k0 + lane] : -1;
    int eJob = in ? d.fairEntJob[k0 + lane] : 0;
    int eAlive = in ? (int)d.evTabAlive[eIdx] : 0;
    int32_t ePrio = in ? d.schedAtPrio[eJob] : 0;
    int64_t eReq[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) eReq[r] = (in && r < cf.R) ? JREQ(d, eJob)[r] : 0;
    int64_t av0[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) av0[r] = r < cf.R ? AL(d, cf.evLevel, r, n) : 0;
    int slots = (d.ftS + 63) >> 6;
    for (int slot = 0; slot < slots; slot++) {
      int s = slot * 64 + lane;
      bool valid = s < d.ftS;
      int ss = valid ? s : 0;
      bool open = valid && cnt > 0 && ((d.shapeMask[(size_t)ss * cf.W + (n >> 6)] >> (n & 63)) & 1);
      int32_t myPrio = d.ftPrio[ss];
      int64_t rq[MAXR], av[MAXR];
#pragma unroll
      for (int r = 0; r < MAXR; r++) { rq[r] = r < cf.R ? d.shapeReq[(size_t)ss * cf.R + r] : 0; av[r] = av0[r]; }
      int32_t t = -1;
      for (int e = 0; e < cnt; e++) {          // wave-uniform trip count: the broadcasts below are cross-lane
        int bAlive = __builtin_amdgcn_readlane(eAlive, e);
        if (!bAlive) continue;
        int32_t bPrio = __builtin_amdgcn_readlane(ePrio, e);
        int bIdx = __builtin_amdgcn_readlane(eIdx, e);
        if (bPrio == NO_PRIORITY) { if (open) { t = FAIR_BAD_ENTRY; open = false; } continue; }
        bool take = open && bPrio <= myPrio;
        bool fits = true;
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
          if (r >= cf.R) break;
          unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)eReq[r], e);
          unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)eReq[r] >> 32), e);
          int64_t v = (int64_t)(((unsigned long long)hi << 32) | lo);
          if (take) { av[r] += v; if (rq[r] > av[r]) fits = false; }
        }
        if (take && fits) { t = bIdx; open = false; }
      }
      if (valid) {
        int32_t old = FT_T(d, s, n);
        if (old != t) { FT_T(d, s, n) = t; ftFixMax(d, s, n, old, t); }
      }
    }
    return;
  }
#endif
  int lane0 = CTL_WAVE() ? CTL_LANE() : 0, stride = CTL_WAVE() ? 64 : 1;
  for (int s = lane0; s < d.ftS; s += stride) {   // (a node with more than 64 entries, and the CPU build)
    int32_t t;
    ftEvalChunk(d, n, s, s + 1, &t);
    int32_t old = FT_T(d, s, n);
    if (old != t) { FT_T(d, s, n) = t; ftFixMax(d, s, n, old, t); }
  }
}

// the table's current maximum for shape s: value and node (wave-uniform)
DEV int32_t ftTop(Dev& d, int s, int* node) {
#if !defined(ASCHED_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  if (CTL_WAVE()) {
    int lane = CTL_LANE();
    int32_t v = FT_B2(d, s, lane);                       // (rows are 64 wide; unused super-blocks hold -1)
    int32_t m = waveMax32(v);
    if (m < 0) { *node = -1; return -1; }
    int sb = __builtin_ctzll(__ballot(v == m));
    int b0 = sb * 64 + lane;
    v = b0 < d.ftNB1 ? FT_B1(d, s, b0) : -1;
    int b = sb * 64 + __builtin_ctzll(__ballot(v == m));
    int n0 = b * 64 + lane;
    v = n0 < cf_N(d) ? FT_T(d, s, n0) : -1;
    unsigned long long hit = __ballot(v == m);
    if (!hit) { *node = -2; return m; }                   // the maxima are above the table (cannot happen while ftFixMax runs after every store): caller rebuilds
    *node = b * 64 + __builtin_ctzll(hit);
    return m;
  }
#endif
  int32_t m = -1; int sbBest = -1;
  for (int j = 0; j < 64; j++) { int32_t v = FT_B2(d, s, j); if (v > m) { m = v; sbBest = j; } }
  if (m < 0) { *node = -1; return -1; }
  int bBest = -1;
  for (int j = 0; j < 64; j++) { int b = sbBest * 64 + j; if (b < d.ftNB1 && FT_B1(d, s, b) == m) { bBest = b; break; } }
  if (bBest < 0) { *node = -2; return m; }
  for (int j = 0; j < 64; j++) { int n = bBest * 64 + j; if (n < d.cfg.N && FT_T(d, s, n) == m) { *node = n; return m; } }
  *node = -2;
  return m;
}

// max over nodes of fairNodeBest for shape s, validated against the current state (see "Staleness" above); -1 = no node
DEV_COLD int ftQuery(Dev& d, int s) {
  d.rs->statFt[0]++;
  for (int tries = 0; tries < 1 << 20; tries++) {
    int node;
    int32_t m = ftTop(d, s, &node);
    if (m < 0) return -1;
    if (node == -2) { d.rs->ftValid = 0; return -2; }
    int32_t before = m;
    ftUpdateNode(d, node);                 // re-evaluates the winner from the planes and the table as they are now
    int32_t after = FT_T(d, s, node);
    if (after == before) return m;
    d.rs->statFt[1]++;
  }
  d.rs->ftValid = 0;
  return -2;
}
