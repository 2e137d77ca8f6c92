// Serial stand-ins of the lane-parallel primitives of armada_amd/csrc/round_fast.h for the CPU build of the device control code (tests/hostsim).
// TEST INFRASTRUCTURE: included by round_fast.h only under ASCHED_HOSTSIM (this directory is on the include path of tests/hostsim/Makefile only);
// the product (armada_sched.hip) has the wave-level versions.  One "lane" = one loop iteration (FOR_LANES), the node engine and the bind wave run inline.
struct PQState { int unused; };
DEV int pqTopFast(int Q);
DEV void pqBuild(PQState&, int) {}
DEV int pqHead(PQState&, int Q) { return pqTopFast(Q); }
DEV void pqPopPush(PQState&, const KeyOut& ko, int q);
DEV void pqPopPush(PQState&, const KeyOut& ko, int q) {   // the serial heap is the key arrays themselves
  if (ko.valid) { FL.kA[q] = ko.A; FL.kX[q] = ko.X; FL.kY[q] = ko.Y; FL.inHeap[q] = 1; } else FL.inHeap[q] = 0;
}
DEV int pqTopFast(int Q) {
  int best = -1;
  for (int q = 0; q < Q; q++) {
    if (!FL.inHeap[q]) continue;
    if (best < 0) { best = q; continue; }
    bool less;
    if (FL.kA[q] != FL.kA[best]) less = FL.kA[q] < FL.kA[best];
    else if (FL.kX[q] != FL.kX[best]) less = FL.kX[q] < FL.kX[best];
    else if (FL.kY[q] != FL.kY[best]) less = FL.kY[q] < FL.kY[best];
    else less = FL.nameRank[q] < FL.nameRank[best];
    if (less) best = q;
  }
  return best;
}
// DRF costs of one updatePQItem for queue q and the job record in window slot k (the LDS vectors must be current):
// proposed = drf(alloc+req)/w, current = drf(alloc)/w, size = drf(req)*w
DEV void drf3(Dev& d, int q, int k, bool replay, double w, double* proposed, double* current, double* size) {
  int64_t alloc[MAXR], with[MAXR];
  const int64_t* req = FL.winRec[q][k].req;
  for (int r = 0; r < d.cfg.R; r++) { alloc[r] = (replay ? FL.qReplay[q][r] : FL.qAlloc[q][r]) + FL.qPenalty[q][r]; with[r] = alloc[r] + req[r]; }
  *proposed = drf(d, with) / w; *current = drf(d, alloc) / w; *size = drf(d, req) * w;
}
DEV void fastFence(Ctl&) {}
DEV void baseTileRemoved(KREF, FastS&, int) {}
DEV void baseMarkRemoved(KREF k, FastS&, int pos) {
  k.baseRemoved[pos] = 1;
  if (k.fitBits) for (int f = 0; f < k.S; f++) k.fitBits[(size_t)f * k.fitW + (pos >> 6)] &= ~(1ull << (pos & 63));
}
// advance the base cursor of shape r.shape to the next clean entry the job fits on.  With shape-fit masks a clean entry says which shapes fit it, so the
// walk also refreshes the candidate of every OTHER shape whose candidate is stale and whose cursor is not behind the walk's start (one bind of a base node
// makes the candidates of all shapes that pointed at it stale: they are found again by this one walk instead of one walk each); such shapes that the
// walk does not satisfy move their cursor to its end — every clean entry up to there has been tested against them.
DEV void baseScan(KREF k, FastS& S, const JobTail& r) {
  int s = r.shape;
  if (k.fitBits) {   // find-first-set in the shape's "clean and fits" bitmap from the cursor on
    S.statScanSteps++;
    int p = FL.cand[s].pos;
    if (FL.cand[s].node == -2 && FL.cand[s].key != 0) p++;   // a stale candidate: the entry at the cursor is the one that was used up
    for (int w = p >> 6; w < k.fitW; w++) {
      uint64_t word = k.fitBits[(size_t)s * k.fitW + w];
      if (w == (p >> 6)) word &= ~0ull << (p & 63);
      if (!word) continue;
      int q = w * 64 + __builtin_ctzll(word);
      CandRec& c = FL.cand[s];
      c.pos = q; c.node = k.baseNode[q]; c.key = k.baseKey[q]; c.cls = k.baseCls[q]; c.ex0 = k.E > 0 ? k.baseExtra[q] : 0; c.ex1 = k.E > 1 ? k.baseExtra[k.Npad + q] : 0;
      return;
    }
    FL.cand[s].pos = k.N; FL.cand[s].node = -1;
    return;
  }
  static long hsScans = 0, hsRemoved = 0, hsUnfit = 0, hsTiles = 0; static bool hsDump = getenv("HS_SCAN_STATS") != nullptr;
  if (hsDump) { hsScans++; if ((hsScans % 20000) == 0) fprintf(stderr, "base scans %ld: removed entries walked %ld, clean unfit walked %ld, 64-entry tiles touched %ld\n", hsScans, hsRemoved, hsUnfit, hsTiles); }
  int hsP0 = FL.cand[s].pos;
  for (int p = FL.cand[s].pos; p < k.N; p++) {
    if (hsDump && ((p - hsP0) % 64) == 0) hsTiles++;
    if (k.baseRemoved[p]) { if (hsDump) hsRemoved++; continue; }
    int64_t ex0 = k.E > 0 ? k.baseExtra[p] : 0, ex1 = k.E > 1 ? k.baseExtra[k.Npad + p] : 0;
    if (!entryFits(k, r, k.baseKey[p], ex0, ex1, k.baseCls[p])) { if (hsDump) hsUnfit++; continue; }
    CandRec& c = FL.cand[s];
    c.pos = p; c.node = k.baseNode[p]; c.key = k.baseKey[p]; c.cls = k.baseCls[p]; c.ex0 = ex0; c.ex1 = ex1;
    S.statScanSteps++;
    return;
  }
  FL.cand[s].pos = k.N; FL.cand[s].node = -1;
}
DEV uint64_t l0Search(KREF k, const JobTail& r, int* slot) {
  uint64_t best = ~0ull; *slot = -1;
  { static long calls = 0, sum = 0, hits = 0; static bool dump = getenv("HS_L0_STATS") != nullptr; if (dump) { calls++; sum += FL.l0Count; if ((calls % 50000) == 0) fprintf(stderr, "l0Search calls %ld avg entries %.1f\n", calls, (double)sum / calls); } (void)hits; }
  for (int i = 0; i < FL.l0Count; i++)
  {
    bool fits = k.maskMode ? (((r.shape < 64 ? FL.l0Cls[i] : FL.l0Cls2[i]) >> (r.shape & 63)) & 1) != 0 : entryFits(k, r, FL.l0Key[i], FL.l0Ex0[i], FL.l0Ex1[i], FL.l0Cls[i]);
    if (FL.l0Key[i] < best && fits) { best = FL.l0Key[i]; *slot = i; }
  }
  return best;
}
// load jobs [pos, pos+cnt) of a queue stream (kind 0: evicted list, 1: queued list) into the queue's window
DEV void winRefill(KREF k, int q, int kind, int pos, int cnt) {
  GP(int32_t) stream = kind == 0 ? k.evList : k.queuedJobs;
  for (int i = 0; i < cnt; i++) {
    int job = stream[pos + i];
    FL.winJob[q][i] = job; FL.winIdx[q][i] = kind == 0 ? k.evIdxByPos[pos + i] : -1;
    memcpy(&FL.winRec[q][i], (const char*)k.jrec + (size_t)job * sizeof(JobRec), sizeof(JobRec));
  }
}
// the head record of queue q := JobRec of `job` from HBM / := window slot w
DEV void loadHeadRec(KREF k, int q, int job) {
  JobRec r; memcpy(&r, (const char*)k.jrec + (size_t)job * sizeof(JobRec), sizeof(JobRec));
  memcpy(FL.headReq[q], r.req, sizeof r.req); memcpy(&FL.headTail[q], &r.keyDelta, sizeof(JobTail));
}
DEV void headFromWindow(int q, int w) { memcpy(FL.headReq[q], FL.winRec[q][w].req, sizeof(int64_t) * MAXR); memcpy(&FL.headTail[q], &FL.winRec[q][w].keyDelta, sizeof(JobTail)); }
// alloc[l][r][n] -= req[r], keys[l][n] -= keyDelta for levels l in [lo, nl)  (markAllocatable, node.go:539-549); req = head of queue q
DEV void bindUpdate(KREF k, FastS&, int n, int lo, int nl, int q, uint64_t keyDelta) {
  for (int l = lo; l < nl; l++) { for (int x = 0; x < k.R; x++) KAL(k, l, x, n) -= FL.headReq[q][x]; KKEY(k, l, n) -= keyDelta; }
}
DEV void keySatSub(KREF k, int n, int lo, int nl, uint64_t keyDelta) {   // per-field saturating subtraction (field 0 = every negative quotient)
  for (int l = lo; l < nl; l++) {
    uint64_t key = KKEY(k, l, n), out = key;
    for (int c = 0; c < k.K; c++) { uint64_t m = k.fieldMask[c], f = key & m, dq = keyDelta & m; out = (out & ~m) | (f > dq ? f - dq : 0); }
    KKEY(k, l, n) = out;
  }
}
// sctx / qctx resource vectors (context/scheduling.go:410-434, context/queue.go:231-265) for the head job of queue q:
// accumulate-only, one lane per resource
DEV void accountVectors(Dev& d, KREF k, int q, int pc, bool ev, bool replay) {
  for (int x = 0; x < k.R; x++) {
    int64_t v = FL.headReq[q][x];
    if (replay) { FL.qReplay[q][x] += v; continue; }
    FL.qAlloc[q][x] += v; RS.allocated[x] += v;
    if (ev) RS.evicted[x] -= v; else RS.scheduled[x] += v;
    size_t i = ((size_t)q * k.npc + pc) * k.R + x;
    k.qAllocByPc[i] += v;
    if (ev) k.qEvictedByPc[i] -= v; else k.qSchedByPc[i] += v;
  }
}
DEV void evWinRefill(KREF k, int q, int pos, int cnt) { for (int i = 0; i < cnt; i++) memcpy(&FL.evWin[q][i], (const char*)k.evKey + (size_t)(pos + i) * sizeof(EvKey), sizeof(EvKey)); }
// Deferred commits of evicted jobs [p0, p1) of queue q's eviction list returning to their nodes: exactly the evicted branch of
// fastIter's commit, applied to many jobs at once (all updates are integer adds or per-job stores: order independent).
// sign = -1 takes the commits back (skip mode left before these jobs' turn): the jobs are evicted again, exactly as the evictor left them.
DEV void applyEvictedRange(Dev& d, int q, int p0, int p1, int sign = 1) {
  const FastK k = fastKRef(d);
  for (int p = p0; p < p1; p++) {
    int job = k.evList[p];
    const JobRec& r = d.jrec[job];
    int n = r.node0, pcx = r.pc; int32_t prio = r.runPrio;
    int32_t cutoff = r.preemptible ? prio : NONPREEMPTIBLE_CUTOFF;
    for (int x = 0; x < k.R; x++) {
      int64_t v = sign * r.req[x];
      FL.qAlloc[q][x] += v; RS.allocated[x] += v; RS.evicted[x] -= v;
      size_t i = ((size_t)q * k.npc + pcx) * k.R + x;
      k.qAllocByPc[i] += v; k.qEvictedByPc[i] -= v;
      for (int l = 1; l < r.nlRun; l++) KAL(k, l, x, n) -= v;
    }
    for (int l = 1; l < r.nlRun; l++) { if (sign > 0) KKEY(k, l, n) -= r.keyDelta; else KKEY(k, l, n) += r.keyDelta; }
    if (sign < 0) {  // state of a job the evictor has just evicted (eviction.go:245-260, evictApply)
      k.jcHasPctx[job] = 0; k.pcNode[job] = -1; k.pcSap[job] = 0; k.pcPap[job] = ASCHED_MIN_PRIORITY; k.pcMethod[job] = ASCHED_METHOD_NONE;
      k.jobEvictedOnNode[job] = 1; k.jobFlags[job] = F_EVICTED; k.inPreempted[job] = 1;
      if (!RS.replayPending) { int idx = k.evIdxByPos[p]; k.evTabAlive[idx] = 1; k.evIndexOfJob[job] = idx; RS.ftValid = 0; RS.fairIndexValid = 0; }   // (an entry comes back: round_ft.h "Staleness", ensureFairIndex)
      continue;
    }
    k.jcReason[job] = 0; k.jcHasPctx[job] = 1; k.pcNode[job] = n; k.pcSap[job] = prio;
    k.jobNode[job] = n; k.jobCutoff[job] = cutoff; k.jobEvictedOnNode[job] = 0; k.schedAtPrio[job] = prio; k.inSchedAndEvicted[job] = 0;
    k.pcPap[job] = prio; k.pcMethod[job] = ASCHED_METHOD_RESCHEDULED; k.jobFlags[job] = F_RESCHEDULED; k.inPreempted[job] = 0;
    if (!RS.replayPending) { k.evTabAlive[k.evIdxByPos[p]] = 0; k.evIndexOfJob[job] = -1; }
  }
}
DEV bool roundLimitExceeded(Dev& d, KREF k) { for (int x = 0; x < k.R; x++) if (RS.scheduled[x] > d.cfg.maxToSchedule[x]) return true; return false; }  // constraints.go:113-119
DEV bool headRequestsDisallowed(Dev& d, KREF k, int q) { for (int x = 0; x < k.R; x++) if (d.cfg.disallowed[x] && FL.headReq[q][x] > 0) return true; return false; }  // nodedb.go:596-601
// two-wave iteration: the engine's bind takes the request from its mailbox; the rollback's accounting from the backup
DEV void bindUpdateEng(KREF k, FastS&, int n, int nl, uint64_t keyDelta, const int64_t* req) {
  for (int l = 0; l < nl; l++) { for (int x = 0; x < k.R; x++) KAL(k, l, x, n) -= req[x]; KKEY(k, l, n) -= keyDelta; }
}
DEV void accountVectorsBk(Dev& d, KREF k, int q, int pc, int sign) {
  for (int x = 0; x < k.R; x++) {
    int64_t v = sign * FL.eng.req[x];
    FL.qAlloc[q][x] += v; RS.allocated[x] += v; RS.scheduled[x] += v;
    size_t i = ((size_t)q * k.npc + pc) * k.R + x;
    k.qAllocByPc[i] += v; k.qSchedByPc[i] += v;
  }
}
DEV void engineRestore(int q) {
  const IterBackup& b = FL.bk;
  FL.hot[q] = b.hot; FL.headTail[q] = FL.eng.tail; memcpy(FL.headReq[q], FL.eng.req, sizeof FL.eng.req);
  FL.kA[q] = b.kA; FL.kX[q] = b.kX; FL.kY[q] = b.kY; FL.effA[q] = b.effA; FL.effX[q] = b.effX; FL.effY[q] = b.effY;
  FL.inHeap[q] = b.inHeap;
}
// an evicted job's pinned-node check (nodedb.go:897-906) against the node's CURRENT allocatable at the job's priority: head request of queue q
DEV bool pinnedNodeFits(KREF k, int q, int n, int level) {
  if (k.nodeFlags[n] & 1) return true;   // unschedulable && overAllocated: always allowed back (:902-903)
  for (int x = 0; x < k.R; x++) if (FL.headReq[q][x] > KAL(k, level, x, n)) return false;
  return true;
}
DEV void exclPinnedFast(Dev& d, KREF, int, int job, int n, int level) { exclPinned(d, job, n, level); }
DEV EvDyn evDynLoad(KREF k, int q, int job, int n, int level, bool wantMark, bool wantPin) {
  EvDyn r; r.preempted = wantMark ? (int)k.jcPreempted[job] : 0; r.fits = (!wantPin || pinnedNodeFits(k, q, n, level)) ? 1 : 0;
  return r;
}
DEV EvDyn evCleanLoad(KREF k, int job, int n, bool wantMark, bool wantClean) {
  EvDyn r; r.preempted = wantMark ? (int)k.jcPreempted[job] : 0; r.fits = 1;
  if (wantClean) for (int x = 0; x < k.R; x++) if (KAL(k, 0, x, n) < 0) r.fits = 0;
  return r;
}
DEV unsigned long long evPendingMask(int Q) { unsigned long long m = 0; for (int q = 0; q < Q && q < 64; q++) if (FL.hot[q].evApplied < FL.hot[q].evDone) m |= 1ull << q; return m; }
DEV void pqHeadKey(PQState&, int t, PackedKey* key, uint32_t* nameRank) { key->A = FL.kA[t]; key->X = FL.kX[t]; key->Y = FL.kY[t]; *nameRank = (uint32_t)FL.nameRank[t]; }
// ---- stream run, serial build.  The engine serves an entry when its record is staged, so that entries emitted but not yet staged when a job
// does not fit are discarded exactly as on the device (there the engine runs behind the merge by up to a ring's worth of entries).
DEV void capMask2(KREF k, uint64_t clsBits, uint64_t key, int64_t ex0, int64_t ex1, uint64_t* m0, uint64_t* m1) {
  *m0 = 0; *m1 = 0;
  for (int s = 0; s < k.S && s < 128; s++) {
    const ShapeReq q = SHT(s);
    if (!q.never && ((clsBits >> q.cls) & 1) && fieldsGE(k, key, q.fieldMin) && q.ex0 <= ex0 && q.ex1 <= ex1) { if (s < 64) *m0 |= 1ull << s; else *m1 |= 1ull << (s - 64); }
  }
}
struct StreamLanes { int start[QCAPF], base[QCAPF], pos[QCAPF], len[QCAPF], kind[QCAPF], ws[QCAPF]; double budget[QCAPF]; uint32_t effA[QCAPF]; uint64_t effX[QCAPF], effY[QCAPF]; };
#define SL_SET(sl, f, q, v) ((sl).f[q] = (v))
#define SL_GET(sl, f, q) ((sl).f[q])
#define SL_GET64(sl, f, q) ((sl).f[q])
#define SL_GETD(sl, f, q) ((sl).f[q])
DEV void qsWinRefill(KREF k, int q, int pos, int cnt) { for (int i = 0; i < cnt; i++) memcpy(&FL.evWin[q][i], (const char*)k.qsKey + ((size_t)q * QS_CMAX + pos + i) * sizeof(EvKey), sizeof(EvKey)); }
DEV int engineServe(Dev& d, KREF k, FastS& ES);
static FastS g_engS;
static void hsLagRead();
DEV void streamBegin(int* engSeq, int hold = 0, int hc = 0) { (void)hc; hsLagRead(); FL.eng.ringPub = FL.eng.ringAck = FL.eng.ringEnd = FL.eng.ringFail = 0; FL.eng.bindDone = 0; FL.eng.bindHold = hold; FL.eng.cmd = ENG_STREAM; (*engSeq)++; }
DEV void bindJob(KREF k, FastS& ES, int n, int nl, uint64_t keyDelta, const int64_t* req, int job, int32_t prio, int32_t cutoff);
DEV void streamRelease(Dev& d, KREF k, int go) {
  (void)d;
  if (go) for (int i = 0; i < FL.eng.ringAck; i++) { const JobRec& r = RREC(i); bindJob(k, g_engS, r.node0, r.nlPc, r.keyDelta, r.req, RJOB(i), r.pcPrio, r.preemptible ? r.pcPrio : NONPREEMPTIBLE_CUTOFF); }
  FL.eng.bindHold = 0;
}
static JobRec g_hsStage[4];
DEV unsigned long long streamStageIssue(KREF k, int base, int cnt) { for (int i = 0; i < cnt; i++) if (!(RQ(base + i) & RQ_EV)) memcpy(&g_hsStage[i], (const char*)k.jrec + (size_t)RJOB(base + i) * sizeof(JobRec), sizeof(JobRec)); return 0; }
DEV int engineServeRing(Dev& d, KREF k, FastS& ES, int i);
DEV void streamServeOne(Dev& d, KREF k, int i) {
  if (RQ(i) & RQ_EV) { FL.eng.ringAck = i + 1; return; }
  int st = engineServeRing(d, k, g_engS, i);
  if (st == 0) { FL.eng.ringFail = 1; return; }
  FL.eng.ringAck = i + 1;
  if (st == 2) FL.eng.ringFail = 2;
}
// HS_RING_LAG=<seed>: the serial engine does NOT serve an entry when it is staged but falls behind by a pseudo-random number of entries, as the device's engine wave does
// (there the merge runs ahead by up to a ring's worth): entries are served a few at a time whenever the control code looks at the ring or idles, and all that are left when the
// session ends.  Everything the control code does with entries that were emitted but not yet placed — accounting, the end of a run, a run's nested events, a member
// without a node — then sees the interleavings the device produces, in the CPU soaks (tests/soak.py with HS_RING_LAG set).  Unset: every entry is served at once, as before.
static Dev* g_hsRingDev = nullptr;
static uint32_t g_hsLagState = 0;
static int g_hsLagSeed = 0;   // read at every streamBegin: tests switch it per round (monkeypatch.setenv)
static int hsLagSeed() { return g_hsLagSeed; }
static void hsLagRead() { const char* e = getenv("HS_RING_LAG"); int v = e && *e ? atoi(e) + 1 : 0; if (v != g_hsLagSeed) { g_hsLagSeed = v; g_hsLagState = 0; } }
static uint32_t hsLagRand() { if (!g_hsLagState) g_hsLagState = 0x9E3779B9u * (uint32_t)hsLagSeed() + 12345u; g_hsLagState = g_hsLagState * 1664525u + 1013904223u; return g_hsLagState >> 16; }
// ... and the BIND side lags too (the device's bind wave issues the HBM side of a placed entry from its ring slot, later): in lag mode engineServeAt only records the node in
// the ring entry (as on the device) and the binds are issued from the slots in order, a few at a time; streamBound() is what the bind side has read.  A ring slot overwritten too
// early, or HBM state read before its bind, shows up as a wrong round.
int hsBindLag() { return hsLagSeed() != 0; }
DEV void hsBindSome(int count) {
  if (!g_hsRingDev || FL.eng.bindHold || !(FL.eng.bindDone < FL.eng.ringAck)) return;
  Dev& d = *g_hsRingDev;
  const FastK k = fastKRef(d);
  while (count-- > 0 && FL.eng.bindDone < FL.eng.ringAck) {
    int i = FL.eng.bindDone;
    if (!(RQ(i) & RQ_EV)) { const JobRec& r = RREC(i); bindJob(k, g_engS, r.node0, r.nlPc, r.keyDelta, r.req, RJOB(i), r.pcPrio, r.preemptible ? r.pcPrio : NONPREEMPTIBLE_CUTOFF); }
    FL.eng.bindDone = i + 1;
  }
}
DEV void hsRingServe(int count) {
  if (!g_hsRingDev || !(FL.eng.ringAck < FL.eng.ringPub) || FL.eng.ringFail) return;   // (nothing pending: the pointer may be a previous launch's)
  Dev& d = *g_hsRingDev;
  const FastK k = fastKRef(d);
  while (count-- > 0 && FL.eng.ringAck < FL.eng.ringPub && !FL.eng.ringFail) streamServeOne(d, k, FL.eng.ringAck);
}
void hsRingIdle() { if (hsLagSeed()) { hsRingServe(1 + (int)(hsLagRand() % 3)); hsBindSome(1 + (int)(hsLagRand() % 3)); } }   // STREAM_IDLE of the serial build: engine and bind side make progress while the control code waits
DEV void streamStageCommit(Dev& d, KREF k, int base, int cnt, unsigned long long) {
  for (int i = 0; i < cnt; i++) if (!(RQ(base + i) & RQ_EV)) RREC(base + i) = g_hsStage[i];
  FL.eng.ringPub = base + cnt;
  g_hsRingDev = &d;
  if (hsLagSeed()) { uint32_t r = hsLagRand(); hsRingServe((r & 7) < 3 ? 0 : (int)((r >> 3) % 6)); hsBindSome((int)((r >> 8) % 4)); return; }   // (often nothing: the backlog grows towards the ring's size)
  for (int i = base; i < base + cnt; i++) if (!FL.eng.ringFail) streamServeOne(d, k, i);
}
DEV void streamEnd(int) { FL.eng.ringEnd = 1; if (hsLagSeed()) { hsRingServe(1 << 30); hsBindSome(1 << 30); } }   // (the device waits for the engine's acknowledgement and for the bind wave's last bind)
DEV int streamAcked(int* fail) { if (hsLagSeed() && (hsLagRand() & 3) == 0) { hsRingServe(1); hsBindSome((int)(hsLagRand() & 1)); } *fail = FL.eng.ringFail; return FL.eng.ringAck; }
DEV int streamBound() { return hsLagSeed() && !FL.eng.bindHold ? FL.eng.bindDone : FL.eng.ringAck; }
DEV void streamAccount(Dev& d, KREF k, int i0, int i1) {
  for (int i = i0; i < i1; i++) {
    int rq = RQ(i), q = rq & 0xff;
    FL.tmpQ[q]++;
    if (rq & RQ_EV) continue;
    const JobRec& r = RREC(i);
    int pc = r.pc;
    for (int x = 0; x < k.R; x++) {
      int64_t v = r.req[x];
      FL.qAlloc[q][x] += v; RS.allocated[x] += v; RS.scheduled[x] += v;
      size_t j = ((size_t)q * k.npc + pc) * k.R + x;
      k.qAllocByPc[j] += v; k.qSchedByPc[j] += v;
    }
  }
}
