// engine_hc.h — the node engine of a bulk-merged stream run (device only; included by armada_sched.hip).
//
// Why.  With the merge off the control wave (round_merge.h) the node engine is the round's only serial chain: first fit at priority -2 for one job after the other
// (nodedb.go:737 -> 840-879), ~4.4 k shader clocks per job on BASELINE configs[2] (profiles/r05b_headline_engine_segments.txt: 1.9 k searching the LDS list of dirty
// nodes, 0.9 k in the rest of first fit — a base rescan costs two HBM round trips —, 1.0 k of list upkeep).  What the chain really needs per job is small: on that
// round 97 % of the dirty-node picks are nodes modified within the last 64 binds, and a clean node is needed every fourth job, in base order
// (measured on the CPU build: DESIGN.md).
//
// What.  The level-0 structure of round_fast.h (sorted base + list of dirty nodes, "L0") split three ways for the length of one ring session:
//   H   the HOT set: the <= 48 (HC_H_MAX) dirty nodes modified most recently, one per lane of the engine wave, in registers.  A query is one branch-free fit test per
//       lane (hcFits: requirement class, every key field through the guard bits, the extras) and a 64-lane minimum (two 32-bit DPP reductions); a bind rewrites one lane.
//   C   the COLD set: every other dirty node = the LDS list itself (FL.l0*), mirrored in the registers of wave 3 (16 rows x 64 lanes).  Wave 3 answers "minimum-key
//       entry a job fits on", takes the entries H evicts, gives up the ones that are picked — and keeps, per fit shape, a LOWER BOUND of the keys of its entries the
//       shape fits on (HCB.clb[128]: lowered by every hand-over, refreshed by every answer).  The engine asks only when that bound lies below both the hot and the clean
//       candidate it has at hand: 2 246 of 200 000 jobs on the headline round; every other job costs wave 3 nothing and the engine one LDS word in its read batch.
//   S   the CLEAN side per fit shape (HcShapes): fastFirstFit's base cursor, the bitmap word it stands in, the head candidate's fields and a SPARE (the next set bit,
//       fetched in the head's round trip), two shapes per lane of the engine wave, in registers.  A consumed head is replaced by the spare without a memory access, else
//       the shape scans on: ONE fetch serves every scanning shape at once (hcShapesFetch; 20 750 fetches for 51 533 clean picks on the headline round).
//       (Round 6 tried a "clean front" of the next 64 base entries first: shapes that fit few of them made 175 k of 200 k jobs fall through it — profiles/r06g.)
// First fit = min(H, C, clean candidate): the same three-way minimum as fastFirstFit — clean entries are unchanged since the sort, H and C hold current values, keys
// are unique.  When the ring closes the three are folded back into the LDS list / candidate cursors exactly as the serial engine would have left them (same set of
// dirty nodes, same removed flags and bitmaps; slot numbers differ, which nothing depends on).
//
// Protocol engine wave -> wave 3: a ring of 64-byte commands in LDS (the idle key windows FL.evWin: the queues' windows are invalidated for the session), in order:
//   HC_Q  what a job needs (key fields, extras, class)              -> reply slot [seq & 3]: best entry (slot, key, node, extras, class bits), inserts done so far
//   HC_I  an entry H evicts (payload ring)                             it stays in H ("evicting") until wave 3's counter (HCB.insDone) covers it: no query can miss it
//   HC_D  slot picked by the engine: the entry leaves C                HC_C  an evicting entry was picked while in flight: its insert is taken back
//   HC_E  the session ends: compact the LDS list, publish its length
// Every wait of this protocol reads the launch's `abandon` word (armada_sched.hip "bounded waits") and leaves through an exit its loop has anyway.
#pragma once

#define HC_CMDS 32
#define HC_INS 16      // payload slots of entries H hands over (at most HC_INS / 2 are in flight)
enum { HC_Q = 1, HC_I, HC_D, HC_C, HC_E };
struct alignas(16) HcCmd { int32_t type, a; uint64_t fieldMin; int64_t ex0, ex1; int32_t cls, seq; uint64_t pad[2]; };
struct alignas(16) HcRep { int32_t seq, slot; uint64_t key; int32_t node, insDone; int64_t ex0, ex1; uint64_t cls; uint64_t pad; };
struct alignas(16) HcIns { uint64_t key; int32_t node, pad; int64_t ex0, ex1; uint64_t cls; uint64_t pad2[2]; };
struct HcBox {
  int32_t cmdPub, cmdDone, overflow, insDone;   // insDone: hand-overs wave 3 has taken in (-1: its list overflowed)
  unsigned long long clb[128];                  // per fit shape: a LOWER bound of the smallest key in C the shape fits on (0: not known yet); written by wave 3 only
  int32_t scratch[64];
  int16_t freeStack[512];
  int32_t insSlot[HC_CMDS];
  HcCmd cmd[HC_CMDS]; HcRep rep[4]; HcIns ins[HC_INS];
};
static_assert(sizeof(HcCmd) == 64 && sizeof(HcRep) == 64 && sizeof(HcIns) == 64, "HC mailbox records are 64 bytes");
static_assert(sizeof(HcBox) <= sizeof(g_fl.evWin), "the HC mailbox lives in the idle key windows");
#define HCB (*(HcBox*)&g_fl.evWin[0][0])
#define HC_ROWS (L0CAP / 64)
#define HC_H_MAX 48     // normal entries H keeps; beyond that the oldest is handed to C

// minimum over the 64 lanes, wave-uniform.  One v_min_u32 with a DPP source per step (a lane without a source keeps its value); the DPP read of a freshly written
// register needs two wait states, which the assembler does not insert inside an asm statement.
__device__ static inline unsigned hcMin32(unsigned v) {
  asm volatile(
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum of the lanes' keys (~0 = no candidate) and the lane that holds it (keys are unique: node-index rank in the low bits); -1 = none
__device__ static inline unsigned long long hcMinKey(unsigned long long key, int* laneOut) {
  const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
  const unsigned mh = hcMin32(hi);
  const unsigned ml = hcMin32(hi == mh ? lo : 0xffffffffu);
  const unsigned long long mn = ((unsigned long long)mh << 32) | ml;
  const unsigned long long who = __ballot(key == mn);
  *laneOut = mn == ~0ull ? -1 : (int)__builtin_ctzll(who | (1ull << 63));
  return mn;
}
__device__ static inline unsigned long long hcRead64(unsigned long long v, int lane) { return slGet64(v, lane); }
__device__ static inline int hcLoadI32(const int32_t* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ static inline void hcStoreI32(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (every lane stores the same word: no exec-mask bookkeeping)
// what a job needs of a level-0 entry, as the lanes hold it (every lane the same values: plain vector operands, nothing is moved to scalar registers for it)
struct HcNeed { unsigned long long fmin; long long ex0, ex1; int cls; };
// entryFits (round_fast.h) for key layouts with guard bits (the only ones this engine runs on: asched_host.inc), branch-free: requirement class, every key field >= the
// job's (one subtraction: a field that is smaller borrows from ITS guard bit), the non-indexed columns
template <int E> __device__ static inline bool hcFits(unsigned long long G, const HcNeed& q, unsigned long long key, unsigned long long cls, long long ex0, long long ex1) {
  bool ok = ((cls >> q.cls) & 1) != 0;
  ok &= (((key | G) - q.fmin) & G) == G;
  if (E > 0) ok &= q.ex0 <= ex0;
  if (E > 1) ok &= q.ex1 <= ex1;
  return ok;
}

// ------------------------------------------------------------------------------------------------ wave 3: the cold set
struct HcRows { unsigned long long key[HC_ROWS], cls[HC_ROWS]; long long ex0[HC_ROWS], ex1[HC_ROWS]; };   // one entry per lane and row; cls 0 = no entry (a hole, or beyond the list)
template <int R> __device__ static inline void hcRowSet(HcRows& rows, int lane, unsigned long long key, long long ex0, long long ex1, unsigned long long cls) {
  if ((int)(threadIdx.x & 63) == lane) { rows.key[R] = key; rows.ex0[R] = ex0; rows.ex1[R] = ex1; rows.cls[R] = cls; }
}
__device__ static inline void hcRowSetAt(HcRows& rows, int slot, unsigned long long key, long long ex0, long long ex1, unsigned long long cls) {
  const int lane = slot & 63;
  switch (slot >> 6) {   // (a uniform branch: the rows are registers)
#define HC_CASE(R) case R: hcRowSet<R>(rows, lane, key, ex0, ex1, cls); break;
    HC_CASE(0) HC_CASE(1) HC_CASE(2) HC_CASE(3) HC_CASE(4) HC_CASE(5) HC_CASE(6) HC_CASE(7) HC_CASE(8) HC_CASE(9) HC_CASE(10) HC_CASE(11) HC_CASE(12) HC_CASE(13) HC_CASE(14) HC_CASE(15)
#undef HC_CASE
  }
}
static_assert(HC_ROWS == 16, "hcRowSetAt enumerates the rows");

template <int E> __device__ static void coldSession(Dev& d, KREF k) {
  (void)d;
  const int lane = threadIdx.x & 63;
  const unsigned long long G = UNI64(k.guardMask);
  HcRows rows;
  int hi = __builtin_amdgcn_readfirstlane(g_fl.l0Count);   // slots [0, hi) are entries or holes
  int nfree = 0, insDone = 0, overflow = 0;
  ShapeReq sq[2];   // what fit shapes (lane, lane + 64) need: an entry handed over lowers the bound of every shape it fits
#pragma unroll
  for (int x = 0; x < 2; x++) { const int t = lane + 64 * x; if (t < k.S) sq[x] = d.shapeTab[t]; else { sq[x].fieldMin = 0; sq[x].ex0 = sq[x].ex1 = 0; sq[x].cls = 0; sq[x].never = 1; } }
#pragma unroll
  for (int r = 0; r < HC_ROWS; r++) {
    const int s = r * 64 + lane;
    const bool in = s < hi;
    rows.key[r] = in ? g_fl.l0Key[s] : ~0ull; rows.ex0[r] = in ? g_fl.l0Ex0[s] : 0; rows.ex1[r] = in ? g_fl.l0Ex1[s] : 0; rows.cls[r] = in ? g_fl.l0Cls[s] : 0ull;
  }
  int done = 0;
  unsigned spins = 0;
  const int debugHang = d.f.debugHang;
#ifdef ASCHED_FASTPROF
  long long cBusy = 0, cT0 = 0; int cQ = 0, cI = 0;
#endif
  for (;;) {
    // the publication counter and the next command's words in ONE batch of LDS reads (the counter first: a command it covers is complete)
    const HcCmd& c = HCB.cmd[done & (HC_CMDS - 1)];
    const int pubV = __hip_atomic_load(&HCB.cmdPub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    LDS_ORDER();
    const int typeV = c.type, aV = c.a, seqV = c.seq;
    HcNeed q; q.fmin = c.fieldMin; q.ex0 = c.ex0; q.ex1 = c.ex1; q.cls = c.cls;   // (every lane reads the same words)
    if (__builtin_amdgcn_readfirstlane(pubV) == done) { __builtin_amdgcn_s_sleep(1); if (waitGaveUp(spins)) return; continue; }   // (a wait given up: armada_sched.hip "bounded waits")
    spins = 0;
    if (debugHang > 0 && done == debugHang) { for (;;) { __builtin_amdgcn_s_sleep(1); if (waitGaveUp(spins)) return; } }   // (tests: a cold-set wave that stops answering; only the bounded waits end this)
#ifdef ASCHED_FASTPROF
    cT0 = CLK();
#endif
    {
      const int type = __builtin_amdgcn_readfirstlane(typeV), a = __builtin_amdgcn_readfirstlane(aV);
      if (type == HC_Q) {
#ifdef ASCHED_FASTPROF
        cQ++;
#endif
        const int seq = __builtin_amdgcn_readfirstlane(seqV);
        unsigned long long best = ~0ull; int bs = -1;
        const int nrows = (hi + 63) >> 6;
#pragma unroll
        for (int rr = 0; rr < HC_ROWS; rr++) {
          if (rr < nrows) {   // (uniform)
            const bool ok = hcFits<E>(G, q, rows.key[rr], rows.cls[rr], rows.ex0[rr], rows.ex1[rr]) & (rows.key[rr] < best);
            best = ok ? rows.key[rr] : best; bs = ok ? rr * 64 + lane : bs;
          }
        }
        int bl;
        const unsigned long long mn = hcMinKey(best, &bl);
        const int slot = bl >= 0 ? __builtin_amdgcn_readlane(bs, bl) : -1;
        HcRep& o = HCB.rep[seq & 3];
        if (lane == 0) { o.slot = slot; o.key = mn; o.insDone = overflow ? -1 : insDone; HCB.clb[a & 127] = mn; }   // (the exact minimum is the best bound there is; ~0: no entry fits)
        LDS_ORDER();
        hcStoreI32(&o.seq, seq);
      } else if (type == HC_I) {
#ifdef ASCHED_FASTPROF
        cI++;
#endif
        const HcIns& in = HCB.ins[a & (HC_INS - 1)];
        const unsigned long long key = UNI64(in.key), cls = UNI64(in.cls); const long long ex0 = (long long)UNI64(in.ex0), ex1 = (long long)UNI64(in.ex1);
        const int node = __builtin_amdgcn_readfirstlane(in.node);
        int slot;
        if (nfree > 0) { nfree--; slot = __builtin_amdgcn_readfirstlane((int)HCB.freeStack[nfree]); }
        else if (hi < L0CAP) slot = hi++;
        else slot = -1;
        if (slot < 0) overflow = 1;   // the list is full: every later answer says so (insDone < 0); the engine ends the session and reports it (the generic full scan takes over, counted in round_stats)
        else {
          if (lane == 0) { g_fl.l0Key[slot] = key; g_fl.l0Node[slot] = node; g_fl.l0Ex0[slot] = ex0; g_fl.l0Ex1[slot] = ex1; g_fl.l0Cls[slot] = cls; g_fl.l0Cls2[slot] = 0; HCB.insSlot[a & (HC_CMDS - 1)] = slot; }
          LANE0_PUBLISHED();
          hcRowSetAt(rows, slot, key, ex0, ex1, cls);
#pragma unroll
          for (int x = 0; x < 2; x++) {   // every fit shape the entry fits: its bound may not stay above the entry's key
            HcNeed nq; nq.fmin = sq[x].fieldMin; nq.ex0 = sq[x].ex0; nq.ex1 = sq[x].ex1; nq.cls = sq[x].cls;
            if (!sq[x].never && hcFits<E>(G, nq, key, cls, ex0, ex1)) (void)__hip_atomic_fetch_min(&HCB.clb[lane + 64 * x], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        insDone = a + 1;
        LDS_ORDER();
        hcStoreI32(&HCB.insDone, overflow ? -1 : insDone);
      } else if (type == HC_D || type == HC_C) {
        const int slot = type == HC_D ? a : __builtin_amdgcn_readfirstlane(HCB.insSlot[a & (HC_CMDS - 1)]);
        if (slot >= 0) {
          if (lane == 0) { const int nd = g_fl.l0Node[slot]; if (nd >= 0) k.l0Slot[nd] = -1; g_fl.l0Node[slot] = -1; if (nfree < 512) HCB.freeStack[nfree] = (int16_t)slot; }
          LANE0_PUBLISHED();
          if (nfree < 512) nfree++;
          hcRowSetAt(rows, slot, ~0ull, 0, 0, 0ull);
        }
      } else if (type == HC_E) {
        // the LDS list dense again: entries move down over the holes, row by row (a row's entries are read before any of them is written; destinations never pass sources)
        int base = 0;
        const int nrows = (hi + 63) >> 6;
#pragma unroll
        for (int rr = 0; rr < HC_ROWS; rr++) {
          if (rr < nrows) {
            const int s = rr * 64 + lane;
            const int node = s < hi ? g_fl.l0Node[s] : -1;
            const bool valid = node >= 0;
            const unsigned long long b = __ballot(valid);
            const int dst = base + __popcll(b & ((1ull << lane) - 1));
            LDS_ORDER();
            if (valid) { g_fl.l0Key[dst] = rows.key[rr]; g_fl.l0Node[dst] = node; g_fl.l0Ex0[dst] = rows.ex0[rr]; g_fl.l0Ex1[dst] = rows.ex1[rr]; g_fl.l0Cls[dst] = rows.cls[rr]; g_fl.l0Cls2[dst] = 0; k.l0Slot[node] = dst; }
            LDS_ORDER();
            base += __popcll(b);
          }
        }
        if (lane == 0) g_fl.l0Count = base;
        LANE0_PUBLISHED();
#ifdef ASCHED_FASTPROF
        cBusy += CLK() - cT0;
        if (lane == 0) { g_rs.statSeg[24] += cBusy; g_rs.statSeg[25] += cQ * 1000ll; g_rs.statSeg[26] += cI * 1000ll; g_rs.statSeg[27] += (long long)hi * 1000; g_rs.statSeg[28] += 1000; }   // cold wave: busy ticks, queries, inserts, list slots at the end, sessions
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the slot map (HBM) before whoever reads it next
        LDS_ORDER();
        hcStoreI32(&HCB.cmdDone, done + 1);
        return;
      }
      LDS_ORDER();
      done++;   // (cmdDone is published with HC_E only: the engine never looks at it before)
    }
#ifdef ASCHED_FASTPROF
    cBusy += CLK() - cT0;
#endif
  }
}
// wave 3 during an engine session (OP_ENGINE): one cold session per HC stream
__device__ static void coldLoop(Dev& d) {
  const FastK k = fastKRef(d);
  int gen = 0;
  unsigned spins = 0;
  for (;;) {
    for (;;) {
      int g = hcLoadI32(&g_fl.eng.hcGen);
      if (g != gen) { gen = g; break; }
      if (hcLoadI32(&g_fl.eng.bindQuit)) return;
      __builtin_amdgcn_s_sleep(2);
      if (waitGaveUp(spins)) return;
    }
    spins = 0;
    LDS_ORDER();
    if (k.E == 0) coldSession<0>(d, k); else if (k.E == 1) coldSession<1>(d, k); else coldSession<2>(d, k);
  }
}

// ------------------------------------------------------------------------------------------------ wave 1: hot set + per-shape clean candidates
struct HcHot { unsigned long long key, cls; long long ex0, ex1; int node, state, seq; };   // state 0 empty, 1 entry, 2 entry handed to C (insert `seq` in flight)
// The clean side: fastFirstFit's per-shape base cursor (FL.cand, round_fast.h) held in registers for the session, two fit shapes per lane (lane, lane + 64), plus the
// bitmap word ("clean and fits", d.fitBits) the cursor stands in.  A consumed candidate's successor is then known without a memory access (the next set bit); its
// fields cost ONE HBM round trip, fetched on demand for every shape that lacks them at once (hcShapesFetch).  st: 0 nothing left, 1 scanning from pos (inclusive), 2 the
// head candidate at pos has its fields here.  lb: key of the candidate consumed last = a lower bound for what the base can still offer the shape (the base is sorted).
enum { SC_NONE = 0, SC_SCAN = 1, SC_HEAD = 2 };
// The SPARE (pos2 >= 0): the candidate behind the head — the next set bit of the same bitmap word, fetched in the head's round trip.  A consumed head is replaced by it
// without a memory access (half of the fetches of a headline round, profiles/r06z_spare_candidates.txt); a spare whose node another shape takes is dropped.
struct HcShapes { int pos[2], st[2], node[2], wIdx[2]; unsigned long long w[2], key[2], cls[2], lb[2]; long long ex0[2], ex1[2];
                  int pos2[2], node2[2]; unsigned long long key2[2], cls2[2]; long long ex02[2], ex12[2]; };
__device__ static inline void hcShapesLoad(KREF k, HcShapes& sc) {   // from the LDS cursors, as the serial engine left them
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int x = 0; x < 2; x++) {
    const int s = lane + 64 * x;
    sc.st[x] = SC_NONE; sc.pos[x] = 0; sc.node[x] = -1; sc.wIdx[x] = -1; sc.w[x] = 0; sc.key[x] = 0; sc.cls[x] = 0; sc.lb[x] = 0; sc.ex0[x] = 0; sc.ex1[x] = 0;
    sc.pos2[x] = -1; sc.node2[x] = -1; sc.key2[x] = 0; sc.cls2[x] = 0; sc.ex02[x] = 0; sc.ex12[x] = 0;
    if (s < k.S) {
      const CandRec c = g_fl.cand[s];
      if (c.node >= 0) { sc.st[x] = SC_HEAD; sc.pos[x] = c.pos; sc.node[x] = c.node; sc.key[x] = c.key; sc.cls[x] = c.cls; sc.ex0[x] = c.ex0; sc.ex1[x] = c.ex1; }
      else if (c.node == -2) { sc.st[x] = SC_SCAN; sc.pos[x] = c.pos + (c.key != 0 ? 1 : 0); sc.lb[x] = c.key; }   // (a stale candidate: the entry at the cursor is the one that was used up)
    }
  }
}
__device__ static inline void hcShapesStore(KREF k, const HcShapes& sc) {   // back into the LDS cursors, in the serial engine's terms
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int x = 0; x < 2; x++) {
    const int s = lane + 64 * x;
    if (s < k.S) {
      CandRec c; c.pad = 0;
      if (sc.st[x] == SC_HEAD) { c.pos = sc.pos[x]; c.node = sc.node[x]; c.key = sc.key[x]; c.cls = sc.cls[x]; c.ex0 = sc.ex0[x]; c.ex1 = sc.ex1[x]; }
      else if (sc.st[x] == SC_SCAN) { c.node = -2; c.key = sc.lb[x]; c.pos = sc.lb[x] != 0 ? sc.pos[x] - 1 : sc.pos[x]; c.cls = 0; c.ex0 = c.ex1 = 0; }   // (baseScan starts behind a stale candidate's position, at a fresh cursor's)
      else { c.node = -1; c.pos = k.N; c.key = 0; c.cls = 0; c.ex0 = c.ex1 = 0; }
      g_fl.cand[s] = c;
    }
  }
}
// base position P was used up: its bit leaves every cached word; a shape whose head it was scans on from behind it
__device__ static inline void hcShapesUsed(HcShapes& sc, int P) {
  const int wi = P >> 6; const unsigned long long bit = 1ull << (P & 63);
#pragma unroll
  for (int x = 0; x < 2; x++) {
    sc.w[x] = sc.wIdx[x] == wi ? sc.w[x] & ~bit : sc.w[x];
    const bool was = (sc.st[x] == SC_HEAD) & (sc.pos[x] == P);
    const bool promote = was & (sc.pos2[x] >= 0);   // (the spare lies behind the head: P is not its position)
    const bool scan = was & !promote;
    sc.lb[x] = was ? sc.key[x] : sc.lb[x];
    sc.key[x] = promote ? sc.key2[x] : sc.key[x]; sc.cls[x] = promote ? sc.cls2[x] : sc.cls[x]; sc.node[x] = promote ? sc.node2[x] : sc.node[x];
    sc.ex0[x] = promote ? sc.ex02[x] : sc.ex0[x]; sc.ex1[x] = promote ? sc.ex12[x] : sc.ex1[x];
    sc.pos[x] = promote ? sc.pos2[x] : (scan ? P + 1 : sc.pos[x]); sc.st[x] = scan ? SC_SCAN : sc.st[x];
    sc.pos2[x] = (promote | (sc.pos2[x] == P)) ? -1 : sc.pos2[x];
  }
}
// every scanning shape advances to its next candidate (bitmap word where the cached one does not cover the cursor: one HBM round trip; then the candidate's fields and
// its removed flag: one more) until fit shape `target` has a head or nothing left.  The removed flag guards against a bit whose clearing is still on its way to L2.
__device__ static inline void hcShapesFetch(KREF k, FastS& ES, HcShapes& sc, int target) {
  const int lane = threadIdx.x & 63;
  const int tx = target >> 6, tl = target & 63;
  for (;;) {
    unsigned long long nw[2]; bool needW[2];
#pragma unroll
    for (int x = 0; x < 2; x++) {
      const int s = lane + 64 * x, wi = sc.pos[x] >> 6;
      if (sc.st[x] == SC_SCAN && wi >= k.fitW) sc.st[x] = SC_NONE;
      needW[x] = sc.st[x] == SC_SCAN && sc.wIdx[x] != wi;
      nw[x] = 0;
      if (needW[x]) nw[x] = __hip_atomic_load(&k.fitBits[(size_t)s * k.fitW + wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ES.statScanSteps++;
    unsigned long long fKey[2], fCls[2]; long long fEx0[2], fEx1[2]; int fNode[2], fRem[2], hp[2]; bool needF[2];
    unsigned long long gKey[2], gCls[2]; long long gEx0[2], gEx1[2]; int gNode[2], gRem[2], hp2[2]; bool needG[2];   // the spare: the set bit behind the first
#pragma unroll
    for (int x = 0; x < 2; x++) {
      const int wi = sc.pos[x] >> 6;
      if (needW[x]) { sc.w[x] = nw[x]; sc.wIdx[x] = wi; }
      const unsigned long long m = sc.st[x] == SC_SCAN ? sc.w[x] & (~0ull << (sc.pos[x] & 63)) : 0ull;
      needF[x] = m != 0;
      hp[x] = wi * 64 + (int)__builtin_ctzll(m | (1ull << 63));
      if (sc.st[x] == SC_SCAN && m == 0) sc.pos[x] = (wi + 1) << 6;   // nothing left in this word: the next one in the next round
      fKey[x] = 0; fCls[x] = 0; fEx0[x] = 0; fEx1[x] = 0; fNode[x] = -1; fRem[x] = 1;
      if (needF[x]) {
        const int q = hp[x];
        fRem[x] = (int)__hip_atomic_load(&k.baseRemoved[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fKey[x] = k.baseKey[q]; fCls[x] = k.baseCls[q]; fNode[x] = k.baseNode[q];
        fEx0[x] = k.E > 0 ? k.baseExtra[q] : 0; fEx1[x] = k.E > 1 ? k.baseExtra[k.Npad + q] : 0;
      }
      const unsigned long long m2 = m & (m - 1);
      needG[x] = needF[x] & (m2 != 0);
      hp2[x] = wi * 64 + (int)__builtin_ctzll(m2 | (1ull << 63));
      gKey[x] = 0; gCls[x] = 0; gEx0[x] = 0; gEx1[x] = 0; gNode[x] = -1; gRem[x] = 1;
      if (needG[x]) {
        const int q = hp2[x];
        gRem[x] = (int)__hip_atomic_load(&k.baseRemoved[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gKey[x] = k.baseKey[q]; gCls[x] = k.baseCls[q]; gNode[x] = k.baseNode[q];
        gEx0[x] = k.E > 0 ? k.baseExtra[q] : 0; gEx1[x] = k.E > 1 ? k.baseExtra[k.Npad + q] : 0;
      }
    }
#pragma unroll
    for (int x = 0; x < 2; x++) {
      if (needF[x]) {
        if (fRem[x]) { sc.w[x] &= ~(1ull << (hp[x] & 63)); sc.pos[x] = hp[x] + 1; }
        else {
          sc.st[x] = SC_HEAD; sc.pos[x] = hp[x]; sc.key[x] = fKey[x]; sc.cls[x] = fCls[x]; sc.node[x] = fNode[x]; sc.ex0[x] = fEx0[x]; sc.ex1[x] = fEx1[x];
          const bool sp = needG[x] & (gRem[x] == 0);   // (a removed entry behind the head: no spare; the scan meets its bit later)
          sc.pos2[x] = sp ? hp2[x] : -1; sc.key2[x] = gKey[x]; sc.cls2[x] = gCls[x]; sc.node2[x] = gNode[x]; sc.ex02[x] = gEx0[x]; sc.ex12[x] = gEx1[x];
        }
      }
    }
    const int ts = tx ? __builtin_amdgcn_readlane(sc.st[1], tl) : __builtin_amdgcn_readlane(sc.st[0], tl);
    if (ts != SC_SCAN) return;
  }
}

// a ring entry as the engine's lanes hold it (every lane the same words: vector operands; nothing is moved to scalar registers unless a branch needs it)
struct HcJobV { unsigned long long keyDelta, fmin; long long ex0, ex1; int cls, never, rq, pub, shape; };
// ring entry idx and the publication counter in ONE batch of LDS reads (the counter first: LDS executes a wave's reads in order, so an entry the counter covers is complete)
__device__ static inline void hcLoadEntry(int idx, HcJobV& o) {
  o.pub = __hip_atomic_load(&g_fl.eng.ringPub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  LDS_ORDER();   // (no instruction: it keeps the compiler from moving the entry's reads in front of the counter's)
  o.rq = RQ(idx);
  const JobTail& t = *(const JobTail*)&RREC(idx).keyDelta;
  o.keyDelta = t.keyDelta; o.fmin = t.fieldMin; o.cls = t.cls; o.never = (int)t.never; o.ex0 = t.ex0; o.ex1 = t.ex1; o.shape = t.shape;
}
// a command for wave 3 (no look at the ring's space: once the answer to a job's question is in, everything posted before that question is done, and a job posts at most
// three commands — the ring holds sixteen)
__device__ static inline void hcPostQ(int& cmdPub, const HcJobV& j, int seq) {
  HcCmd& c = HCB.cmd[cmdPub & (HC_CMDS - 1)];
  if ((threadIdx.x & 63) == 0) { c.type = HC_Q; c.a = j.shape; c.seq = seq; c.fieldMin = j.fmin; c.ex0 = j.ex0; c.ex1 = j.ex1; c.cls = j.cls; }
  LDS_ORDER();
  cmdPub++;
  hcStoreI32(&HCB.cmdPub, cmdPub);
}
__device__ static inline void hcPostA(int& cmdPub, int type, int a) {
  HcCmd& c = HCB.cmd[cmdPub & (HC_CMDS - 1)];
  if ((threadIdx.x & 63) == 0) { c.type = type; c.a = a; }
  LDS_ORDER();
  cmdPub++;
  hcStoreI32(&HCB.cmdPub, cmdPub);
}

// One ring session with the split structure.  Same contract as the ENG_STREAM walk of engineLoop: entry i is ready when ringPub > i; the chosen node goes into the
// ring entry (the bind wave issues the HBM side); ringAck counts the entries placed; ringFail 1 = entry ringAck found no node, 2 = the list overflowed.
// The loop is written around two costs measured on the MI355X (profiles/r06g_hc_engine_segments.txt): ~8 shader clocks per instruction of a lone wave, ~140 per
// dependent LDS round trip.  Per job: ONE batch of LDS reads for the next entry and the cold set's answer (issued before the tests, used after them).
// (A function of its own, not inlined: inside engineLoop its loop shared the register allocation of the serial engine's paths and lived partly in AGPRs.  Its constants
// and counters are copied in and out: nothing in the loop goes through the references.)
struct HcOut { int engSeq, statScanSteps, statL0Max; long long segT, eseg[8]; };
template <int E> __device__ static __attribute__((noinline)) HcOut engineStreamHcT(Dev& d, const FastK k, const int engSeq0, const int statScan0, const int statL0Max0, const long long segT0) {
  // (every argument BY VALUE: a reference to the caller's constants or counters would put them in memory over there — the serial engine's loops read them too;
  //  measured: 20 % on gang-heavy rounds, profiles/r06v)
  // (The arguments of a function that is not inlined count as lane-DIVERGENT for the compiler, and so does everything loaded through the flat `Dev&`: with the cancel word
  //  on the way to a `break` the per-job loop has a divergent exit, its carried values live in vector registers and 65 of its 82 branches are exec-masked ones
  //  (opt -passes='print<uniformity>').  Making the inputs uniform — g_dev, v_readfirstlane on the counters and the cancel word: 33 divergent branches left — was
  //  measured SLOWER, 262 -> 291 ms on the headline round: the scalar file is what this kernel is short of.  profiles/r06z_uniform_arguments.txt)
  FastS ES; ES.engSeq = engSeq0; ES.statScanSteps = statScan0; ES.statL0Max = statL0Max0; ES.segT = segT0; ES.laneL = 0; ES.laneX = 0;
#ifdef ASCHED_FASTPROF
  for (int x = 0; x < 8; x++) ES.eseg[x] = 0;
#endif
  const int lane = threadIdx.x & 63;
  const unsigned long long G = UNI64(k.guardMask), MFM = UNI64(k.minFieldMin);
  const long long ME0 = (long long)UNI64(k.minEx0), ME1 = (long long)UNI64(k.minEx1);
  // ---- session start: the mailbox, wave 3, the shapes' cursors
  if (lane == 0) { HCB.cmdPub = 0; HCB.cmdDone = 0; HCB.overflow = 0; HCB.insDone = 0; for (int x = 0; x < 4; x++) HCB.rep[x].seq = -1; }
  HCB.clb[lane] = 0; HCB.clb[lane + 64] = 0;   // no shape's bound is known yet: its first job asks
  LDS_ORDER();
  hcStoreI32(&g_fl.eng.hcGen, __builtin_amdgcn_readfirstlane(g_fl.eng.hcGen) + 1);
  int cmdPub = 0, insSeq = 0, rrNext = 0, qSeq = 0, inFlight = 0;   // inFlight: lanes of H whose entry has been handed to C and not yet been taken in
  HcHot h; h.key = ~0ull; h.cls = 0; h.ex0 = h.ex1 = 0; h.node = -1; h.state = 0; h.seq = 0;
  HcShapes sc; hcShapesLoad(k, sc);
  int total = __builtin_amdgcn_readfirstlane(g_fl.l0Count);   // dirty nodes alive (H + C): the list's high-water mark for round_stats
  int i = 0, fail = 0;
  // (bounded waits, armada_sched.hip: a turn of a wait here reads the launch's `abandon` word — 0 or 1 — and leaves through an exit the loop has anyway; NO flag is carried
  //  around the per-job loop: a bool live across this loop is a lane mask merged with three scalar instructions at every join, +5 % on the headline round)
  HcJobV cur; bool haveCur = false;
#ifdef ASCHED_FASTPROF
  int pSrc[4] = {0, 0, 0, 0}, pFetch = 0, pDry = 0, pAsk = 0;
#endif
  for (;;) {
    if (!haveCur) {   // the ring ran dry (or the session starts): wait for entry i
#ifdef ASCHED_FASTPROF
      pDry++;
#endif
      for (;;) {
        hcLoadEntry(i, cur);
        if (__builtin_amdgcn_readfirstlane(cur.pub) > i) break;
        if (hcLoadI32(&g_fl.eng.ringEnd) | hcLoadI32(&g_fl.eng.abandon)) { hcLoadEntry(i, cur); break; }   // (a wait given up leaves through the exits the loop has, and nothing about it is carried around the loop: see `gaveUp` below)
        __builtin_amdgcn_s_sleep(1);
      }
      if (__builtin_amdgcn_readfirstlane(cur.pub) <= i) break;
      haveCur = true;
    }
    ESEG(0);   // [16] waiting for a ring entry
    const int flags = __builtin_amdgcn_readfirstlane((cur.rq & RQ_EV) | (cur.never << 16));   // (ONE scalar look at what makes an entry special)
    if (flags) {
      if (flags >> 16) { fail = 1; break; }
      i++; haveCur = false;   // an evicted job returning to its node: nothing to select or bind here
      hcStoreI32(&g_fl.eng.ringAck, i);
      continue;
    }
    if ((++ES.engSeq & 255) == 0 && cancelRequested(d)) { if (lane == 0) g_fl.eng.cancel = 1; LANE0_PUBLISHED(); fail = 1; break; }
    // ---- ONE batch of LDS reads, in flight during the tests: the next entry, wave 3's hand-over counter, then the shape's lower bound for the cold set
    HcJobV nxt; hcLoadEntry(i + 1, nxt);
    const int rIns = __hip_atomic_load(&HCB.insDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    LDS_ORDER();   // (the bound is read behind the counter: a hand-over the counter covers has lowered it already)
    const unsigned long long rLb = HCB.clb[cur.shape & 127];
    // ---- H: one test per lane, the 64-lane minimum
    HcNeed q; q.fmin = cur.fmin; q.ex0 = cur.ex0; q.ex1 = cur.ex1; q.cls = cur.cls;
    const bool fitH = hcFits<E>(G, q, h.key, h.cls, h.ex0, h.ex1);   // (an empty lane has class bits 0: it fits nothing)
    int hLane;
    unsigned long long hk = hcMinKey(fitH ? h.key : ~0ull, &hLane);
    // ---- the shape's clean candidate as it stands
    const int shape = __builtin_amdgcn_readfirstlane(cur.shape), sl = shape & 63;
    const bool hiSet = cur.shape >= 64;   // (the same in every lane: a select per word, then ONE v_readlane — no branch on the set)
    int cst = __builtin_amdgcn_readlane(hiSet ? sc.st[1] : sc.st[0], sl);
    unsigned long long cKey = hcRead64(hiSet ? sc.key[1] : sc.key[0], sl);        // (SC_HEAD: the candidate's key)
    const unsigned long long cLb = hcRead64(hiSet ? sc.lb[1] : sc.lb[0], sl);     // (SC_SCAN: a lower bound)
    ESEG(1);   // [17] H test, clean candidate
    int insDone = __builtin_amdgcn_readfirstlane(rIns);
    if (insDone < 0) { fail = 2; break; }   // the cold list overflowed
    if (inFlight > 0) {   // entries whose hand-over wave 3 has taken in leave H: the bound read behind the counter covers them.  The best lane among them: ask C (it has that very entry)
      const bool rel = (h.state == 2) & (h.seq < insDone);
      const unsigned long long rb = __ballot(rel);
      h.state = rel ? 0 : h.state; h.cls = rel ? 0ull : h.cls; h.key = rel ? ~0ull : h.key;
      inFlight -= __popcll(rb);
      const bool drop = hLane >= 0 && ((rb >> hLane) & 1);
      hLane = drop ? -1 : hLane; hk = drop ? ~0ull : hk;
    }
    // ---- does the cold set matter?  Only when its bound lies below both the hot candidate and the clean candidate at hand: then the exact answer, synchronously (rare)
    const unsigned long long ckLb = UNI64(rLb);
    const unsigned long long known = cst == SC_HEAD && cKey < hk ? cKey : hk;
    unsigned long long ck = ~0ull; int cSlot = -1;
    if (ckLb < known) {
#ifdef ASCHED_FASTPROF
      pAsk++;
#endif
      hcPostQ(cmdPub, cur, qSeq);
      HcRep& rp = HCB.rep[qSeq & 3];
      for (;;) { if ((int)(hcLoadI32(&rp.seq) == qSeq) | hcLoadI32(&g_fl.eng.abandon)) break; __builtin_amdgcn_s_sleep(1); }
      LDS_ORDER();
      qSeq++;
      cSlot = __builtin_amdgcn_readfirstlane(rp.slot); ck = UNI64(rp.key);
      insDone = __builtin_amdgcn_readfirstlane(rp.insDone) | -hcLoadI32(&g_fl.eng.abandon);   // (given up: -1, the overflow's exit)
      if (insDone < 0) { fail = 2; break; }
      const bool rel = (h.state == 2) & (h.seq < insDone);   // the answer counts these hand-overs: the entries are C's now
      const unsigned long long rb = __ballot(rel);
      h.state = rel ? 0 : h.state; h.cls = rel ? 0ull : h.cls; h.key = rel ? ~0ull : h.key;
      inFlight -= __popcll(rb);
      const bool drop = hLane >= 0 && ((rb >> hLane) & 1);
      hLane = drop ? -1 : hLane; hk = drop ? ~0ull : hk;
    }
    ESEG(2);   // [18] the cold set: bound, or question and answer
    // ---- first fit = min(H, C, clean): fastFirstFit's rule
    const unsigned long long lk = hk < ck ? hk : ck;
    if (cst == SC_SCAN && !(lk < cLb)) {   // the base may hold something better than the dirty candidate: the shape's next clean candidate (and every other scanning shape's)
#ifdef ASCHED_FASTPROF
      pFetch++;
#endif
      hcShapesFetch(k, ES, sc, shape);
      cst = __builtin_amdgcn_readlane(hiSet ? sc.st[1] : sc.st[0], sl);
      cKey = hcRead64(hiSet ? sc.key[1] : sc.key[0], sl);
      // (the cold set needs no second look: it was asked above unless its bound lies at or above the hot candidate, and then it cannot win whatever the base offers)
    }
    const unsigned long long lk2 = hk < ck ? hk : ck;
    const unsigned long long bk = cst == SC_HEAD ? cKey : (cst == SC_SCAN ? cLb : ~0ull);
    // 0 H, 1 C, 2 the shape's clean candidate — selects, not branches (every branch in this loop is an exec-masked one: profiles/r06z_uniform_arguments.txt); only the
    // cold set's node, which is rare and in LDS, is fetched under one
    const bool dirtyWins = lk2 < bk;
    if (!dirtyWins & (cst != SC_HEAD)) { fail = 1; break; }
    const int src = dirtyWins ? (hk < ck ? 0 : 1) : 2;
    // ---- the pick: its node goes to the bind wave at once; its level-0 entry after the bind (fastAfterBind) is worked out where it lives
    const int nH = __builtin_amdgcn_readlane(h.node, hLane & 63), nB = __builtin_amdgcn_readlane(hiSet ? sc.node[1] : sc.node[0], sl);
    int n = dirtyWins ? nH : nB;
    if (src == 1) n = __builtin_amdgcn_readfirstlane(g_fl.l0Node[cSlot]);
#ifdef ASCHED_FASTPROF
    pSrc[src & 3]++;
#endif
    ESEG(3);   // [19] the three-way minimum (incl. fetching clean candidates)
    RREC(i).node0 = n;   // (every lane stores the same word)
    LDS_ORDER();
    i++;
    hcStoreI32(&g_fl.eng.ringAck, i);
    // ---- what C must know: an entry it gives up, a hand-over taken back
    if (src == 1) hcPostA(cmdPub, HC_D, cSlot);
    if (inFlight > 0 && src == 0 && __builtin_amdgcn_readlane(h.state, hLane) == 2) { hcPostA(cmdPub, HC_C, __builtin_amdgcn_readlane(h.seq, hLane)); inFlight--; }
    const bool haveNext = __builtin_amdgcn_readfirstlane(nxt.pub) > i;
    ESEG(4);   // [20] verdict, commands
    // ---- the bind's effect on the node's level-0 entry: key and extras go down; an entry that can no longer host anything is dropped
    if (src == 0) {   // in place, in its lane
      const unsigned long long nk = h.key - cur.keyDelta; const long long n0 = h.ex0 - cur.ex0, n1 = h.ex1 - cur.ex1;
      const bool alive = ((((nk | G) - MFM) & G) == G) & (ME0 <= n0) & (ME1 <= n1);
      const bool me = lane == hLane;
      h.key = me ? (alive ? nk : ~0ull) : h.key; h.ex0 = me ? n0 : h.ex0; h.ex1 = me ? n1 : h.ex1; h.state = me ? (alive ? 1 : 0) : h.state; h.cls = (me & !alive) ? 0ull : h.cls;
      if (__ballot(me & !alive)) total--;
    } else {
      unsigned long long okey, ocls; long long oex0, oex1;
      if (src == 1) { okey = ck; ocls = UNI64(g_fl.l0Cls[cSlot]); oex0 = (long long)UNI64(g_fl.l0Ex0[cSlot]); oex1 = (long long)UNI64(g_fl.l0Ex1[cSlot]); }
      else {
        okey = cKey;
        ocls = hcRead64(hiSet ? sc.cls[1] : sc.cls[0], sl);
        oex0 = (long long)hcRead64((unsigned long long)(hiSet ? sc.ex0[1] : sc.ex0[0]), sl);
        oex1 = (long long)hcRead64((unsigned long long)(hiSet ? sc.ex1[1] : sc.ex1[0]), sl);
        const int usedPos = __builtin_amdgcn_readlane(hiSet ? sc.pos[1] : sc.pos[0], sl);
        baseMarkRemoved(k, ES, usedPos);   // a clean entry is used up (fastAfterBind's base branch): its flag and bitmap bits; the shapes' cursors are in registers here
        hcShapesUsed(sc, usedPos);
      }
      const unsigned long long nk = okey - UNI64(cur.keyDelta); const long long n0 = oex0 - (long long)UNI64(cur.ex0), n1 = oex1 - (long long)UNI64(cur.ex1);
      const bool alive = ((((nk | G) - MFM) & G) == G) && ME0 <= n0 && ME1 <= n1;
      if (alive) {
        // a free lane (there is always one: at most HC_H_MAX entries + the few in flight)
        const unsigned long long occ = __ballot(h.state != 0);
        if (occ == ~0ull) { fail = 2; break; }   // (cannot happen: HC_H_MAX entries + the hand-overs in flight are fewer than the lanes; reported as an overflow if it ever does)
        const int fl = (int)__builtin_ctzll(~occ);
        const bool me = lane == fl;
        h.key = me ? nk : h.key; h.cls = me ? ocls : h.cls; h.ex0 = me ? n0 : h.ex0; h.ex1 = me ? n1 : h.ex1; h.node = me ? n : h.node; h.state = me ? 1 : h.state;
        if (src != 1) { total++; if (total > ES.statL0Max) ES.statL0Max = total; }
        // H over its size: the next entry in lane order (round robin) goes to C
        const unsigned long long normal = __ballot(h.state == 1);
        if (__popcll(normal) > HC_H_MAX) {
          while (insSeq - insDone >= HC_INS / 2) {   // (wave 3 is far behind with the hand-overs: never seen; the payload slots and the command ring are bounded by this)
            __builtin_amdgcn_s_sleep(1);
            insDone = hcLoadI32(&HCB.insDone) | -hcLoadI32(&g_fl.eng.abandon);
            if (insDone < 0) break;
          }
          if (insDone < 0) { fail = 2; break; }
          const unsigned long long cand = normal & ~(1ull << fl);
          const unsigned long long hiPart = cand & (~0ull << rrNext);
          const int v = hiPart ? (int)__builtin_ctzll(hiPart) : (int)__builtin_ctzll(cand);
          rrNext = (v + 1) & 63;
          HcIns& in = HCB.ins[insSeq & (HC_INS - 1)];
          if (lane == v) { in.key = h.key; in.node = h.node; in.ex0 = h.ex0; in.ex1 = h.ex1; in.cls = h.cls; }
          h.state = lane == v ? 2 : h.state; h.seq = lane == v ? insSeq : h.seq;
          LDS_ORDER();
          hcPostA(cmdPub, HC_I, insSeq);
          insSeq++; inFlight++;
        }
      } else if (src == 1) total--;
    }
    cur = nxt; haveCur = haveNext;
    ESEG(5);   // [21] hot-set / base upkeep
  }
  bool gaveUp = waitAbandoned();   // the session ends without its hand-shakes
  if (gaveUp) fail = 1;
#ifdef ASCHED_FASTPROF
  if (lane == 0) { g_rs.statSeg[29] += pSrc[0] * 1000ll; g_rs.statSeg[30] += pSrc[1] * 1000ll; g_rs.statSeg[31] += pSrc[2] * 1000ll; g_rs.statSeg[33] += pFetch * 1000ll; g_rs.statSeg[35] += pDry * 1000ll; g_rs.statSeg[36] += pAsk * 1000ll; }   // picks from H / C / the base; clean-candidate fetches; ring-dry waits; questions put to the cold set
#endif
  // ---- session end: the shapes' cursors go back to LDS; C compacts the LDS list; H's entries are appended; a pending failure is reported after the structure is whole again
  hcShapesStore(k, sc);
  if (!gaveUp) {
    hcPostA(cmdPub, HC_E, 0);
    for (;;) { if (hcLoadI32(&HCB.cmdDone) == cmdPub) break; __builtin_amdgcn_s_sleep(1); if (waitAbandoned()) { gaveUp = true; break; } }
  }
  LDS_ORDER();
  if (!gaveUp) {
    int cnt = __builtin_amdgcn_readfirstlane(g_fl.l0Count);
    const bool mine = h.state == 1;   // (state 2: C has it — every insert is done once HC_E is)
    const unsigned long long b = __ballot(mine);
    const int dst = cnt + __popcll(b & ((1ull << lane) - 1));
    const int add = __popcll(b);
    if (cnt + add > L0CAP) { if (fail != 1) fail = 2; else fail = 3; }   // (3: no node AND the list overflowed — reported as the overflow)
    else {
      if (mine) { g_fl.l0Key[dst] = h.key; g_fl.l0Node[dst] = h.node; g_fl.l0Ex0[dst] = h.ex0; g_fl.l0Ex1[dst] = h.ex1; g_fl.l0Cls[dst] = h.cls; g_fl.l0Cls2[dst] = 0; k.l0Slot[h.node] = dst; }
      if (lane == 0) g_fl.l0Count = cnt + add;
      LANE0_PUBLISHED();
    }
    // nodes that left the list during the session keep no slot: C cleared the ones it gave up; H's dead entries never had one (they came from the base or from C)
  }
  if (fail == 3) fail = 2;
  if (fail) { LDS_ORDER(); hcStoreI32(&g_fl.eng.ringFail, fail); }
  HcOut o; o.engSeq = ES.engSeq; o.statScanSteps = ES.statScanSteps; o.statL0Max = ES.statL0Max; o.segT = ES.segT;
  for (int x = 0; x < 8; x++) o.eseg[x] = 0;
#ifdef ASCHED_FASTPROF
  for (int x = 0; x < 8; x++) o.eseg[x] = ES.eseg[x];
#endif
  return o;
}
__device__ static inline void engineStreamHc(Dev& d, KREF k, FastS& ES) {
  HcOut o;
  if (k.E == 0) o = engineStreamHcT<0>(d, k, ES.engSeq, ES.statScanSteps, ES.statL0Max, ES.segT); else if (k.E == 1) o = engineStreamHcT<1>(d, k, ES.engSeq, ES.statScanSteps, ES.statL0Max, ES.segT); else o = engineStreamHcT<2>(d, k, ES.engSeq, ES.statScanSteps, ES.statL0Max, ES.segT);
  ES.engSeq = o.engSeq; ES.statScanSteps = o.statScanSteps; ES.statL0Max = o.statL0Max;
#ifdef ASCHED_FASTPROF
  ES.segT = o.segT; for (int x = 0; x < 8; x++) ES.eseg[x] += o.eseg[x];
#endif
}
