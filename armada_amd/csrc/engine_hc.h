// engine_hc.h — the node engine of a bulk-merged stream run (device only; included by armada_sched.hip).
//
// Why.  With the merge off the control wave (round_merge.h) the node engine is the round's only serial chain: first fit at priority -2 for one job after the other
// (nodedb.go:737 -> 840-879), ~4.4 k shader clocks per job on BASELINE configs[2] (profiles/r05b_headline_engine_segments.txt: 1.9 k searching the LDS list of dirty
// nodes, 0.9 k in the rest of first fit — a base rescan costs two HBM round trips —, 1.0 k of list upkeep).  What the chain really needs per job is small: on that
// round 97 % of the dirty-node picks are nodes modified within the last 64 binds, and a clean node is needed every fourth job, in base order
// (measured on the CPU build: DESIGN.md).
//
// What.  The level-0 structure of round_fast.h (sorted base + list of dirty nodes, "L0") split three ways for the length of one ring session:
//   H   the HOT set: the <= 64 dirty nodes modified most recently, one per lane of the engine wave, in registers.  A query is one entryFits per lane, a ballot and
//       a 64-lane minimum (two 32-bit DPP reductions); a bind rewrites one lane.
//   C   the COLD set: every other dirty node = the LDS list itself (FL.l0*), mirrored in the registers of wave 3 (16 rows x 64 lanes).  Wave 3 answers "minimum-key
//       entry a job fits on" for the NEXT job while the engine wave finishes the current one, takes the entries H evicts, and gives up the ones that are picked.
//   CF  the CLEAN FRONT: the next <= 64 clean base entries in base (= key) order, one per lane of the engine wave, gathered from HBM in one go (removed flags, then
//       the entries' fields).  Base order is key order, so the first lane that fits is the first feasible clean entry; a consumed entry is struck in its lane.
//       A job no lane fits falls back on its shape's cursor behind the front (baseScan, as before).
// First fit = min(H, C, clean candidate): the same three-way minimum as fastFirstFit — clean entries are unchanged since the sort, H and C hold current values, keys
// are unique.  When the ring closes the three are folded back into the LDS list / candidate cursors exactly as the serial engine would have left them (same set of
// dirty nodes, same removed flags and bitmaps; slot numbers differ, which nothing depends on).
//
// Protocol engine wave -> wave 3: a ring of 64-byte commands in LDS (the idle key windows FL.evWin), processed strictly in order:
//   HC_Q  what a job needs (key fields, extras, class)              -> reply slot [seq & 3]: best entry (slot, key, node, extras, class bits), inserts done so far
//   HC_I  an entry H evicts (payload ring)                             it stays in H ("evicting") until a reply says the insert is done: no query can miss it
//   HC_D  slot picked by the engine: the entry leaves C                HC_C  an evicting entry was picked while in flight: its insert is taken back
//   HC_E  the session ends: compact the LDS list, publish its length
#pragma once

#define HC_CMDS 16
enum { HC_Q = 1, HC_I, HC_D, HC_C, HC_E };
struct alignas(16) HcCmd { int32_t type, a; uint64_t fieldMin; int64_t ex0, ex1; int32_t cls, seq; uint64_t pad[2]; };
struct alignas(16) HcRep { int32_t seq, slot; uint64_t key; int32_t node, insDone; int64_t ex0, ex1; uint64_t cls; uint64_t pad; };
struct alignas(16) HcIns { uint64_t key; int32_t node, pad; int64_t ex0, ex1; uint64_t cls; uint64_t pad2[2]; };
struct HcBox {
  int32_t cmdPub, cmdDone, overflow, pad0;
  int32_t scratch[64];
  int16_t freeStack[512];
  int32_t insSlot[HC_CMDS];
  HcCmd cmd[HC_CMDS]; HcRep rep[4]; HcIns ins[HC_CMDS];
};
static_assert(sizeof(HcCmd) == 64 && sizeof(HcRep) == 64 && sizeof(HcIns) == 64, "HC mailbox records are 64 bytes");
static_assert(sizeof(HcBox) <= sizeof(g_fl.evWin), "the HC mailbox lives in the idle key windows");
#define HCB (*(HcBox*)&g_fl.evWin[0][0])
#define HC_ROWS (L0CAP / 64)
#define HC_H_MAX 48     // normal entries H keeps; beyond that the oldest is handed to C
#define HC_CF_LOW 6     // clean-front entries left when it is gathered again

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
__device__ static inline void hcStoreI32(int32_t* p, int v) { if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
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
#pragma unroll
  for (int r = 0; r < HC_ROWS; r++) {
    const int s = r * 64 + lane;
    const bool in = s < hi;
    rows.key[r] = in ? g_fl.l0Key[s] : ~0ull; rows.ex0[r] = in ? g_fl.l0Ex0[s] : 0; rows.ex1[r] = in ? g_fl.l0Ex1[s] : 0; rows.cls[r] = in ? g_fl.l0Cls[s] : 0ull;
  }
  int done = 0;
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
    if (__builtin_amdgcn_readfirstlane(pubV) == done) { __builtin_amdgcn_s_sleep(1); continue; }
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
        if (lane == 0) { o.slot = slot; o.key = mn; o.insDone = overflow ? -1 : insDone; }
        LDS_ORDER();
        hcStoreI32(&o.seq, seq);
      } else if (type == HC_I) {
#ifdef ASCHED_FASTPROF
        cI++;
#endif
        const HcIns& in = HCB.ins[a & (HC_CMDS - 1)];
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
        }
        insDone = a + 1;
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
  for (;;) {
    for (;;) {
      int g = hcLoadI32(&g_fl.eng.hcGen);
      if (g != gen) { gen = g; break; }
      if (hcLoadI32(&g_fl.eng.bindQuit)) return;
      __builtin_amdgcn_s_sleep(2);
    }
    LDS_ORDER();
    if (k.E == 0) coldSession<0>(d, k); else if (k.E == 1) coldSession<1>(d, k); else coldSession<2>(d, k);
  }
}

// ------------------------------------------------------------------------------------------------ wave 1: hot set + clean front
struct HcHot { unsigned long long key, cls; long long ex0, ex1; int node, state, seq; };   // state 0 empty, 1 entry, 2 entry handed to C (insert `seq` in flight)
struct HcFront { unsigned long long key, cls; long long ex0, ex1; int node, pos; };           // node -1: no entry in this lane

// the clean front from base position `from` on: the next <= 64 clean entries in base order.  Returns G: every clean entry in [from, G) is in a lane.
__device__ static inline int hcFrontFill(KREF k, FastS& ES, HcFront& cf, int from) {   // (inlined at its ONE call site: a call would put the front and the loop constants in memory)
  const int lane = threadIdx.x & 63;
  const int N = k.N;
  int have = 0, pos0 = from, G = from;
  cf.node = -1; cf.pos = -1;
  bool seenClean = false;
  for (int w = 0; w < 32 && have < 64 && pos0 < N; w++, pos0 += 64) {
    const int p = pos0 + lane;
    const int rem = p < N ? (int)__hip_atomic_load(&k.baseRemoved[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
    ES.statScanSteps++;
    const unsigned long long b = __ballot(rem == 0);
    if (!seenClean) {   // nothing clean before this position: where the next session's front may start (this launch)
      if (b) { seenClean = true; hcStoreI32(&g_fl.eng.cleanFrom, pos0 + (__ffsll((long long)b) - 1)); } else hcStoreI32(&g_fl.eng.cleanFrom, pos0 + 64 < N ? pos0 + 64 : N);
    }
    const int dst = have + __popcll(b & ((1ull << lane) - 1));
    if (rem == 0 && dst < 64) HCB.scratch[dst] = p;
    const int cnt = __popcll(b);
    if (have + cnt >= 64) {   // the lane that received slot 63 ends the front
      const unsigned long long last = __ballot(rem == 0 && dst == 63);
      G = pos0 + (__ffsll((long long)last) - 1) + 1;
      have = 64;
      break;
    }
    have += cnt;
    G = pos0 + 64 < N ? pos0 + 64 : N;
  }
  LDS_ORDER();
  const int q = lane < have ? HCB.scratch[lane] : -1;
  LDS_ORDER();
  if (q >= 0) {
    cf.pos = q; cf.key = k.baseKey[q]; cf.cls = k.baseCls[q]; cf.node = k.baseNode[q];
    cf.ex0 = k.E > 0 ? k.baseExtra[q] : 0; cf.ex1 = k.E > 1 ? k.baseExtra[k.Npad + q] : 0;
  }
  return G;
}
// a clean entry is used up (fastAfterBind's base branch): its flag and bitmap bits, every shape's candidate that named it
__device__ static inline void hcCleanUsed(KREF k, FastS& ES, int pos, int n) {
  baseMarkRemoved(k, ES, pos);
  candInvalidate(k.S, n);
}
// no lane of the front fits: the shape's own cursor behind the front (fastFirstFit's rule).  Returns 1 when the candidate c is the pick, 0 when the dirty candidate (key lk) is, -1: no node.
__device__ static inline int hcBehindFront(KREF k, FastS& ES, const JobTail& r, unsigned long long lk, int cfG, CandRec* cOut) {
  const int lane = threadIdx.x & 63;
  const int s = r.shape;
  CandRec c = g_fl.cand[s]; uniCand(c);
  if (c.node != -1 && c.pos < cfG) {   // every clean entry in front of cfG is in a lane and none fits: the cursor moves up to the front's end
    c.node = -2;
    if (c.key != 0) c.pos = cfG - 1; else c.pos = cfG;   // baseScan starts behind a stale candidate's position, at a fresh cursor's
    if (lane == 0) { g_fl.cand[s].pos = c.pos; g_fl.cand[s].node = -2; }
    LANE0_PUBLISHED();
  }
  if (c.node == -2 && !(lk < c.key)) { baseScan(k, ES, r); c = g_fl.cand[s]; uniCand(c); }
  const unsigned long long bk = c.node >= 0 ? c.key : (c.node == -2 ? c.key : ~0ull);
  *cOut = c;
  if (lk < bk) return 0;
  return c.node >= 0 ? 1 : -1;
}

// a ring entry as the engine's lanes hold it (every lane the same words: vector operands; nothing is moved to scalar registers unless a branch needs it)
struct HcJobV { unsigned long long keyDelta, fmin; long long ex0, ex1; int cls, never, rq, pub; };
// ring entry idx and the publication counter in ONE batch of LDS reads (the counter first: LDS executes a wave's reads in order, so an entry the counter covers is complete)
__device__ static inline void hcLoadEntry(int idx, HcJobV& o) {
  o.pub = __hip_atomic_load(&g_fl.eng.ringPub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  LDS_ORDER();   // (no instruction: it keeps the compiler from moving the entry's reads in front of the counter's)
  o.rq = RQ(idx);
  const JobTail& t = *(const JobTail*)&RREC(idx).keyDelta;
  o.keyDelta = t.keyDelta; o.fmin = t.fieldMin; o.cls = t.cls; o.never = (int)t.never; o.ex0 = t.ex0; o.ex1 = t.ex1;
}
// a command for wave 3 (no look at the ring's space: once the answer to a job's question is in, everything posted before that question is done, and a job posts at most
// four commands — the ring holds sixteen)
__device__ static inline void hcPostQ(int& cmdPub, const HcJobV& j, int seq) {
  HcCmd& c = HCB.cmd[cmdPub & (HC_CMDS - 1)];
  if ((threadIdx.x & 63) == 0) { c.type = HC_Q; c.a = 0; c.seq = seq; c.fieldMin = j.fmin; c.ex0 = j.ex0; c.ex1 = j.ex1; c.cls = j.cls; }
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
// The loop is written around two costs measured on the MI355X (profiles/r06f_hc_segments.txt): ~8 shader clocks per instruction of a lone wave, ~140 per dependent LDS
// round trip.  Per job: ONE batch of LDS reads for the next entry (issued before the tests, used after the decision) and one for the cold set's answer.
template <int E> __device__ static void engineStreamHcT(Dev& d, KREF k, FastS& ES) {
  const int lane = threadIdx.x & 63;
  const unsigned long long G = UNI64(k.guardMask), MFM = UNI64(k.minFieldMin);
  const long long ME0 = (long long)UNI64(k.minEx0), ME1 = (long long)UNI64(k.minEx1);
  const int N = UNI32(k.N);
  // ---- session start: the mailbox, wave 3, the clean front
  if (lane == 0) { HCB.cmdPub = 0; HCB.cmdDone = 0; HCB.overflow = 0; for (int x = 0; x < 4; x++) HCB.rep[x].seq = -1; }
  LDS_ORDER();
  hcStoreI32(&g_fl.eng.hcGen, __builtin_amdgcn_readfirstlane(g_fl.eng.hcGen) + 1);
  int cmdPub = 0, insSeq = 0, rrNext = 0;
  HcHot h; h.key = ~0ull; h.cls = 0; h.ex0 = h.ex1 = 0; h.node = -1; h.state = 0; h.seq = 0;
  HcFront cf;
  int cfG;
  {   // no clean entry a job could fit on lies before the smallest shape cursor
    unsigned mp = 0xffffffffu;
    for (int s = lane; s < k.S; s += 64) { unsigned p = (unsigned)g_fl.cand[s].pos; mp = p < mp ? p : mp; }
    unsigned m = hcMin32(mp);
    const unsigned cfrom = (unsigned)__builtin_amdgcn_readfirstlane(g_fl.eng.cleanFrom);
    if (m == 0xffffffffu) m = (unsigned)N;
    if (cfrom > m) m = cfrom;
    cf.node = -1; cf.pos = -1; cf.key = 0; cf.cls = 0; cf.ex0 = cf.ex1 = 0;
    cfG = (int)m < N ? (int)m : N;   // (the front is gathered at the top of the loop: cfFill)
  }
#ifdef HC_NO_CF
  cfG = 0;   // (debugging: no clean front — every clean candidate through the shape cursors)
  int cfFill = -1;
#else
  int cfFill = cfG;   // >= 0: gather the front from this base position before the next job
#endif
  int total = __builtin_amdgcn_readfirstlane(g_fl.l0Count);   // dirty nodes alive (H + C): the list's high-water mark for round_stats
  int i = 0, qUpTo = 0, fail = 0;
  HcJobV cur; bool haveCur = false;
#ifdef ASCHED_FASTPROF
  int pSrc[4] = {0, 0, 0, 0}, pBehind = 0, pFill = 0, pDry = 0, pRepWait = 0;
#endif
  for (;;) {
    if (cfFill >= 0) { cfG = hcFrontFill(k, ES, cf, cfFill); cfFill = -1;
#ifdef ASCHED_FASTPROF
      pFill++;
#endif
    }
    if (!haveCur) {
#ifdef ASCHED_FASTPROF
      pDry++;
#endif   // the ring ran dry (or the session starts): wait for entry i
      for (;;) {
        hcLoadEntry(i, cur);
        if (__builtin_amdgcn_readfirstlane(cur.pub) > i) break;
        if (hcLoadI32(&g_fl.eng.ringEnd)) { hcLoadEntry(i, cur); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (__builtin_amdgcn_readfirstlane(cur.pub) <= i) break;
      haveCur = true;
    }
    ESEG(0);   // [16] waiting for a ring entry
    if (__builtin_amdgcn_readfirstlane(cur.rq) & RQ_EV) {   // an evicted job returning to its node: nothing to select or bind here
      i++; haveCur = false;
      hcStoreI32(&g_fl.eng.ringAck, i);
      continue;
    }
    if ((++ES.engSeq & 255) == 0 && cancelRequested(d)) { if (lane == 0) g_fl.eng.cancel = 1; LANE0_PUBLISHED(); fail = 1; break; }
    if (__builtin_amdgcn_readfirstlane(cur.never)) { fail = 1; break; }
    if (qUpTo <= i) { hcPostQ(cmdPub, cur, i); qUpTo = i + 1; }
#ifndef HC_NO_CF
    if (cfG < N && __ballot(cf.node >= 0) == 0) { cfFill = cfG; continue; }   // the front is used up (or its stretch of the base held nothing clean): the next stretch first
#endif
    // ---- the next entry and the cold set's answer: their LDS reads are in flight during the tests
    HcJobV nxt; hcLoadEntry(i + 1, nxt);
    HcRep& rp = HCB.rep[i & 3];
    int rSeq = __hip_atomic_load(&rp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    LDS_ORDER();
    int rSlot = rp.slot, rIns = rp.insDone; unsigned long long rKey = rp.key;
    // ---- H and the clean front: one test per lane each
    HcNeed q; q.fmin = cur.fmin; q.ex0 = cur.ex0; q.ex1 = cur.ex1; q.cls = cur.cls;
    const bool fitH = (h.state != 0) & hcFits<E>(G, q, h.key, h.cls, h.ex0, h.ex1);
    int hLane;
    unsigned long long hk = hcMinKey(fitH ? h.key : ~0ull, &hLane);
#ifdef HC_NO_CF
    const unsigned long long bCf = 0;
#else
    const unsigned long long bCf = __ballot((cf.node >= 0) & hcFits<E>(G, q, cf.key, cf.cls, cf.ex0, cf.ex1));
#endif
    const int cfLane = bCf ? (int)__builtin_ctzll(bCf) : -1;
    ESEG(1);   // [17] H and front tests
    // ---- the cold set's answer for this job
    while (__builtin_amdgcn_readfirstlane(rSeq) != i) {
#ifdef ASCHED_FASTPROF
      pRepWait++;
#endif
      __builtin_amdgcn_s_sleep(1);
      rSeq = __hip_atomic_load(&rp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      LDS_ORDER();
      rSlot = rp.slot; rIns = rp.insDone; rKey = rp.key;
    }
    ESEG(2);   // [18] waiting for the cold set's answer
    const int cSlot = __builtin_amdgcn_readfirstlane(rSlot), insDone = __builtin_amdgcn_readfirstlane(rIns);
    const unsigned long long ck = UNI64(rKey);   // (~0 when no entry fits)
    if (insDone < 0) { fail = 2; break; }   // the cold list overflowed
    {   // entries whose hand-over to C this answer already counts leave H; the best lane among them: the answer is at least as good (it saw that very entry)
      const bool rel = (h.state == 2) & (h.seq < insDone);
      const unsigned long long rb = __ballot(rel);
      if (rb) {
        if (rel) { h.state = 0; h.node = -1; h.key = ~0ull; }
        if (hLane >= 0 && ((rb >> hLane) & 1)) { hLane = -1; hk = ~0ull; }
      }
    }
    // ---- first fit = min(H, C, clean): fastFirstFit's rule with the front in place of the shape's cursor
    const unsigned long long lk = hk < ck ? hk : ck;
    int src;   // 0 H, 1 C, 2 front, 3 the shape's cursor behind the front
    CandRec c; c.node = -1; c.pos = 0; c.key = 0; c.cls = 0; c.ex0 = c.ex1 = 0; c.pad = 0;
    if (cfLane >= 0) {
      const unsigned long long fk = hcRead64(cf.key, cfLane);
      src = lk < fk ? (hk < ck ? 0 : 1) : 2;
    } else {
#ifdef ASCHED_FASTPROF
      pBehind++;
#endif
      JobTail r = *(const JobTail*)&RREC(i).keyDelta; uniJobTail(r);
      const int v = hcBehindFront(k, ES, r, lk, cfG, &c);
      if (v < 0) { fail = 1; break; }
      src = v ? 3 : (hk < ck ? 0 : 1);
    }
    // ---- the pick: its node goes to the bind wave at once; its level-0 entry after the bind (fastAfterBind) is worked out where it lives
    int n;
    if (src == 0) n = __builtin_amdgcn_readlane(h.node, hLane);
    else if (src == 1) n = __builtin_amdgcn_readfirstlane(g_fl.l0Node[cSlot]);
    else if (src == 2) n = __builtin_amdgcn_readlane(cf.node, cfLane);
    else n = c.node;
#ifdef ASCHED_FASTPROF
    pSrc[src & 3]++;
#endif
    ESEG(3);   // [19] the three-way minimum (incl. a base rescan behind the front)
    if (lane == 0) RREC(i).node0 = n;
    LANE0_PUBLISHED();
    LDS_ORDER();
    i++;
    hcStoreI32(&g_fl.eng.ringAck, i);
    // ---- what C must know before it answers for the next job, then the next job's question
    if (src == 1) hcPostA(cmdPub, HC_D, cSlot);
    if (src == 0 && __builtin_amdgcn_readlane(h.state, hLane) == 2) hcPostA(cmdPub, HC_C, __builtin_amdgcn_readlane(h.seq, hLane));
    const bool haveNext = __builtin_amdgcn_readfirstlane(nxt.pub) > i;
    if (haveNext && !(__builtin_amdgcn_readfirstlane(nxt.rq) & RQ_EV) && !__builtin_amdgcn_readfirstlane(nxt.never)) { hcPostQ(cmdPub, nxt, i); qUpTo = i + 1; }
    ESEG(4);   // [20] verdict, commands, the next job's question
    // ---- the bind's effect on the node's level-0 entry: key and extras go down; an entry that can no longer host anything is dropped
    if (src == 0) {   // in place, in its lane
      const unsigned long long nk = h.key - cur.keyDelta; const long long n0 = h.ex0 - cur.ex0, n1 = h.ex1 - cur.ex1;
      const bool alive = ((((nk | G) - MFM) & G) == G) & (ME0 <= n0) & (ME1 <= n1);
      if (lane == hLane) { if (alive) { h.key = nk; h.ex0 = n0; h.ex1 = n1; h.state = 1; } else { h.state = 0; h.node = -1; h.key = ~0ull; } }
      if (__ballot((lane == hLane) & !alive)) total--;
    } else {
      unsigned long long okey, ocls; long long oex0, oex1; int usedPos = 0;
      if (src == 1) { okey = ck; ocls = UNI64(g_fl.l0Cls[cSlot]); oex0 = (long long)UNI64(g_fl.l0Ex0[cSlot]); oex1 = (long long)UNI64(g_fl.l0Ex1[cSlot]); }
      else if (src == 2) {
        okey = hcRead64(cf.key, cfLane); ocls = hcRead64(cf.cls, cfLane); oex0 = (long long)hcRead64((unsigned long long)cf.ex0, cfLane); oex1 = (long long)hcRead64((unsigned long long)cf.ex1, cfLane);
        usedPos = __builtin_amdgcn_readlane(cf.pos, cfLane);
        if (lane == cfLane) cf.node = -1;
      } else { okey = c.key; ocls = c.cls; oex0 = c.ex0; oex1 = c.ex1; usedPos = c.pos; }
      if (src >= 2) hcCleanUsed(k, ES, usedPos, n);
      const unsigned long long nk = okey - UNI64(cur.keyDelta); const long long n0 = oex0 - (long long)UNI64(cur.ex0), n1 = oex1 - (long long)UNI64(cur.ex1);
      const bool alive = ((((nk | G) - MFM) & G) == G) && ME0 <= n0 && ME1 <= n1;
      if (alive) {
        // a free lane (there is always one: at most HC_H_MAX entries + the few in flight)
        const unsigned long long occ = __ballot(h.state != 0);
        if (occ == ~0ull) { fail = 2; break; }   // (cannot happen: HC_H_MAX entries + the inserts in flight are fewer than the lanes; reported as an overflow if it ever does)
        const int fl = (int)__builtin_ctzll(~occ);
        if (lane == fl) { h.key = nk; h.cls = ocls; h.ex0 = n0; h.ex1 = n1; h.node = n; h.state = 1; h.seq = 0; }
        if (src != 1) { total++; if (total > ES.statL0Max) ES.statL0Max = total; }
        // H over its size: the next entry in lane order (round robin) goes to C
        const unsigned long long normal = __ballot(h.state == 1);
        if (__popcll(normal) > HC_H_MAX) {
          const unsigned long long cand = normal & ~(1ull << fl);
          const unsigned long long hiPart = cand & (~0ull << rrNext);
          const int v = hiPart ? (int)__builtin_ctzll(hiPart) : (int)__builtin_ctzll(cand);
          rrNext = (v + 1) & 63;
          HcIns& in = HCB.ins[insSeq & (HC_CMDS - 1)];   // (free: an insert sixteen inserts old has been read)
          if (lane == v) { in.key = h.key; in.node = h.node; in.ex0 = h.ex0; in.ex1 = h.ex1; in.cls = h.cls; h.state = 2; h.seq = insSeq; }
          LDS_ORDER();
          hcPostA(cmdPub, HC_I, insSeq);
          insSeq++;
        }
      } else if (src == 1) total--;
    }
    ESEG(5);   // [21] hot-set / base upkeep
    // ---- the front runs low: gather it again from its first remaining entry
    if (src == 2) {
      const unsigned long long left = __ballot(cf.node >= 0);
      if (__popcll(left) <= HC_CF_LOW && cfG < N) cfFill = left ? __builtin_amdgcn_readlane(cf.pos, (int)__builtin_ctzll(left)) : cfG;
    }
    cur = nxt; haveCur = haveNext;
    ESEG(6);   // [22] gathering the front
  }
#ifdef ASCHED_FASTPROF
  if (lane == 0) { g_rs.statSeg[29] += pSrc[0] * 1000ll; g_rs.statSeg[30] += pSrc[1] * 1000ll; g_rs.statSeg[31] += pSrc[2] * 1000ll; g_rs.statSeg[32] += pSrc[3] * 1000ll; g_rs.statSeg[33] += pBehind * 1000ll; g_rs.statSeg[34] += pFill * 1000ll; g_rs.statSeg[35] += pDry * 1000ll; g_rs.statSeg[36] += pRepWait * 1000ll; }   // picks from H / C / front / behind the front; jobs no front lane fitted; front gathers; ring-dry waits; polls of the answer
#endif
  // ---- session end: C compacts the LDS list; H's entries are appended; a pending failure is reported after the structure is whole again
  hcPostA(cmdPub, HC_E, 0);
  for (;;) { if (hcLoadI32(&HCB.cmdDone) == cmdPub) break; __builtin_amdgcn_s_sleep(1); }
  LDS_ORDER();
  {
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
}
__device__ static void engineStreamHc(Dev& d, KREF k, FastS& ES) {
  if (k.E == 0) engineStreamHcT<0>(d, k, ES); else if (k.E == 1) engineStreamHcT<1>(d, k, ES); else engineStreamHcT<2>(d, k, ES);
}
