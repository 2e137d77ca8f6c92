// hostsim.cpp — TEST INFRASTRUCTURE ONLY.
//
// Compiles the device control code (armada_amd/csrc/round_ctl.h, round_run.h) and the host marshalling
// (asched_host.inc) for the CPU, replacing the workgroup primitives by serial loops.  Purpose: debug the
// *logic* of the device code against the golden vectors in a container without a GPU.  It is never loaded
// by armada_amd (the product loads only armada_amd/csrc/libarmada_sched.so, built by hipcc for gfx950) and
// it proves nothing about the kernels' parallel primitives — the `-m gpu` tests do that on the MI355X.
#define ASCHED_HOSTSIM 1
#define ASCHED_PREFIX asched_
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include <chrono>
#include <time.h>
#include "../../armada_amd/csrc/round_run.h"
#include "../../armada_amd/csrc/round_opt.h"
#include "../../armada_amd/csrc/round_price.h"
static thread_local MktDev g_mk;   // market-driven rounds (round_mkt.h): the market state of the launch in progress (a kernel argument on the device)
DEV MktDev* mktDev() { return &g_mk; }
// optional per-primitive wall-clock profile of the serial build (HOSTSIM_PROF=1): where would a wide device primitive matter?
struct HsProf { double t[40]; long n[40]; bool on; HsProf() : on(getenv("HOSTSIM_PROF") != nullptr) { for (int i = 0; i < 40; i++) { t[i] = 0; n[i] = 0; } }
  ~HsProf() { if (on) for (int i = 0; i < 40; i++) if (n[i]) fprintf(stderr, "hsprof[%d] calls=%ld total=%.3fs\n", i, n[i], t[i]); } };
static HsProf g_prof;
struct HsScope { int k; std::chrono::steady_clock::time_point t0; HsScope(int k_) : k(k_) { if (g_prof.on) t0 = std::chrono::steady_clock::now(); }
  ~HsScope() { if (g_prof.on) { g_prof.t[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); g_prof.n[k]++; } } };

// ---- serial versions of the workgroup primitives
DEV void atomicAddI64(int64_t* p, int64_t v) { *p += v; }
DEV void atomicAddI32(int32_t* p, int32_t v) { *p += v; }
DEV void atomicOrI32(int32_t* p, int32_t v) { *p |= v; }
DEV void atomicMinU32(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
DEV int atomicFetchAddI32(int32_t* p, int32_t v) { int o = *p; *p += v; return o; }
void hsEngineMustBeStopped(const char* what);
// sharded wide passes (dev.h shardWorld; asched_shard_round): this rank's share of the node words, then the all-reduce on the handle's communicator — here synchronously
// through the caller's transport (the device posts the words to the host thread that drives the launch: armada_sched.hip shardReduce / plat_run_control)
static int plat_allreduce(long long* buf, size_t count, int op);
static long g_shardExchanges = 0;
static uint64_t hsShardMin(uint64_t v) {
  long long w = (long long)(v ^ 0x8000000000000000ull);
  if (plat_allreduce(&w, 1, 1)) { fprintf(stderr, "hostsim: the all-reduce of a sharded pass failed\n"); abort(); }
  g_shardExchanges++;
  return (uint64_t)w ^ 0x8000000000000000ull;
}
DEV int wgFairSelect(Dev& d, const FairArgs& a) {
  hsEngineMustBeStopped("wgFairSelect");
  HsScope prof(39);
  int best = -1;
  static long calls = 0, maxSeg = 0, sumMax = 0;
  long callMax = 0;
  if (SHARD_ON(d.cfg)) {
    for (int n = SHARD_LO(d.cfg); n < SHARD_HI(d.cfg); n++) { int v = fairNodeBest(d, a, n, -1); if (v > best) best = v; }
    return (int)(uint32_t)~hsShardMin(~(uint64_t)(uint32_t)(best + 1)) - 1;
  }
  for (int n = 0; n < d.cfg.N; n++) { int v = fairNodeBest(d, a, n, -1); if (v > best) best = v; long L = d.fairOff[n + 1] - d.fairOff[n]; if (L > callMax) callMax = L; }
  if (g_prof.on) { calls++; sumMax += callMax; if (callMax > maxSeg) maxSeg = callMax; if ((calls & 4095) == 0) fprintf(stderr, "fair calls=%ld max segment=%ld avg max=%.1f E=%d\n", calls, maxSeg, (double)sumMax / calls, d.rs->evictedTableSize); }
  return best;
}

// On the device every wide op needs all waves of the control workgroup at the mailbox: one posted while the node engine wave sits in its serve loop never completes
// (round 5: the bulk skip of unfeasible keys at the end of a stream run, profiles/r05y_bulk_skip_hang.txt).  The serial build cannot hang — it can say so.
void hsEngineMustBeStopped(const char* what) {
  if (g_fl.eng.live == 1) { fprintf(stderr, "hostsim: %s posted while the node engine is live: the device would hang here\n", what); abort(); }
}
DEV uint64_t wgFirstFitKey(Dev& d, const ScanArgs& a) {
  hsEngineMustBeStopped("wgFirstFitKey");
  HsScope prof(38);
  const DevCfg& c = d.cfg;
  uint64_t best = ~0ull, bestHi = ~0ull;
  d.rs->numScans++;
  if (WIDE_KEYS(c) && a.levelHi > a.level) { fprintf(stderr, "hostsim: the fused multi-level pass on a two-word key\n"); abort(); }
  for (int n = SHARD_LO(c); n < SHARD_HI(c); n++) {
    if (!((a.maskA[n >> 6] >> (n & 63)) & 1)) continue;
    if (a.maskB && !((a.maskB[n >> 6] >> (n & 63)) & 1)) continue;
    if (a.levelHi > a.level) {   // multi-level mode
      for (int l = a.level; l <= a.levelHi; l++) {
        uint64_t v = ((uint64_t)l << SCAN_LEVEL_SHIFT) | KEY(d, l, n);
        if (v >= best) break;
        if (fitsAlloc(d, a.req, l, n)) { best = v; break; }
      }
      continue;
    }
    uint64_t k = KEY(d, a.level, n);
    if (WIDE_KEYS(c)) {   // two-word keys: (high word, low word); the LOW word of the minimum is returned (armada_sched.hip wgFirstFitKey does it in two passes)
      uint64_t lo = KEYLO(d, a.level, n);
      if (k < a.lowBound || (k == a.lowBound && lo < a.lowBoundLo)) continue;
      if (getenv("HOSTSIM_IGNORE_HIGH_WORD")) k = 0;   // negative control for the tests: a selection that compares low words only
      if (k > bestHi || (k == bestHi && lo >= best)) continue;
      if (!a.noFit && !fitsAlloc(d, a.req, a.level, n)) continue;
      bestHi = k; best = lo;
      continue;
    }
    if (k >= best || k < a.lowBound) continue;
    if (!a.noFit && !fitsAlloc(d, a.req, a.level, n)) continue;
    best = k;
  }
  if (SHARD_ON(c)) {
    if (WIDE_KEYS(c) && !(a.levelHi > a.level)) {   // (high word, low word): the minimum high word first, then the low words of the ranks that hold it
      uint64_t hi = hsShardMin(bestHi);
      if (hi == ~0ull) return ~0ull;   // (no rank has a candidate: every rank leaves here)
      best = hsShardMin(bestHi == hi ? best : ~0ull);
    } else best = hsShardMin(best);
  }
  return best;
}
DEV int wgFirstFit(Dev& d, const ScanArgs& a) {
  uint64_t best = wgFirstFitKey(d, a);
  if (best == ~0ull) return -1;
  return d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];
}
DEV int wgScanFair(Dev& d, const ScanArgs& a, const FairArgs& f, uint64_t* bestKey) { *bestKey = wgFirstFitKey(d, a); return wgFairSelect(d, f); }
DEV void wgBulk(Dev& d, int kind, int n) { hsEngineMustBeStopped("wgBulk"); HsScope prof(kind); for (int i = 0; i < n; i++) bulkElem(d, kind, i); }
DEV void wgBulkWide(Dev& d, int kind, int n) { wgBulk(d, kind, n); }
DEV void wgFtBuild(Dev& d, int phase, int n) { hsEngineMustBeStopped("wgFtBuild"); for (int i = 0; i < n; i++) ftBuildAny(d, phase, i); }
DEV int wgCompactFlagged(Dev& d, const int32_t* order, const int32_t* segOff, int nseg, int n, const uint8_t* flag, int32_t* dst, int32_t* outSegOff) {
  int cnt = 0, q = 0;
  for (int p = 0; p <= n; p++) {
    while (q <= nseg && segOff[q] == p) outSegOff[q++] = cnt;
    if (p < n && flag[order[p]]) dst[cnt++] = order[p];
  }
  while (q <= nseg) outSegOff[q++] = cnt;
  return cnt;
}
DEV int wgCompactIota(Dev&, int n, const uint8_t* flag, int32_t* dst) { int c = 0; for (int i = 0; i < n; i++) if (flag[i]) dst[c++] = i; return c; }
DEV int pqTop(Dev& d, const Ctl& c) {
  int best = -1;
  for (int q = 0; q < d.cfg.Q; q++) if (d.pqInHeap[q] && (best < 0 || pqLess(d, c, q, best))) best = q;
  if (getenv("HOSTSIM_TOURNAMENT")) {  // the device's lane tournament (armada_sched.hip pqTop), emulated: must pick the same queue
    int lb[64];
    for (int lane = 0; lane < 64; lane++) { lb[lane] = -1; for (int q = lane; q < d.cfg.Q; q += 64) if (d.pqInHeap[q] && (lb[lane] < 0 || pqLess(d, c, q, lb[lane]))) lb[lane] = q; }
    for (int off = 32; off; off >>= 1) {
      int nb[64];
      for (int lane = 0; lane < 64; lane++) { int o = lb[lane ^ off], b = lb[lane]; if (o >= 0 && (b < 0 || pqLess(d, c, o, b))) b = o; nb[lane] = b; }
      for (int lane = 0; lane < 64; lane++) lb[lane] = nb[lane];
    }
    if (lb[0] != best) { fprintf(stderr, "hostsim: lane tournament picks queue %d, serial argmin %d (Q=%d)\n", lb[0], best, d.cfg.Q); abort(); }
  }
  return best;
}

// ---- platform layer
static std::string g_err;
struct PlatCtx { int32_t cancelWord = 0; double deadlineS = 0; bool inRound = false; int launches = 0; asched_allreduce_fn extFn = nullptr; void* extCtx = nullptr; int commRank = 0, commWorld = 1; };   // per handle, like the device build's (stream / events / mailbox there)
static thread_local PlatCtx* t_ctx = nullptr;
static void* plat_malloc(size_t n) { return malloc(n); }
static void plat_free(void* p) { free(p); }
static void plat_memset(void* p, int v, size_t n) { memset(p, v, n); }
static void plat_h2d(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void plat_d2h(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void* plat_pinned(size_t n) { return malloc(n); }
static void plat_pinned_free(void* p) { free(p); }
static void plat_d2h_async(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void plat_sync() {}
static PlatCtx* plat_open(std::string&, int) { t_ctx = new PlatCtx(); return t_ctx; }
static void plat_close(PlatCtx* c) { if (t_ctx == c) t_ctx = nullptr; delete c; }
static void plat_enter(PlatCtx* c) { t_ctx = c; }
static bool plat_take_failure() { return false; }
static int plat_wall_clock_khz() { return 100000; }
static void plat_set_deadline(double s) { if (t_ctx) t_ctx->deadlineS = s; }
static void plat_cancel(PlatCtx* c) { if (c) __atomic_store_n(&c->cancelWord, 1, __ATOMIC_RELEASE); }
static void plat_cancel_clear(PlatCtx* c) { if (c) __atomic_store_n(&c->cancelWord, 0, __ATOMIC_RELEASE); }
static double plat_last_control_ms() { return 0; }
static int plat_last_control_launches() { return 0; }
static double plat_last_fit_ms() { return 0; }
static const char* plat_last_error() { return g_err.c_str(); }
static void plat_set_market_dev(const MktDev* m) { if (m) g_mk = *m; else memset(&g_mk, 0, sizeof g_mk); }
static int plat_run_control(Dev& dev, int cmd) {
  Dev d = dev;
  d.cancel = &t_ctx->cancelWord;
  // the serial build cannot be interrupted from its own thread: a deadline of <= 1 us stands for "already expired" (tests), asched_cancel from
  // another thread works as on the device
  bool isRound = cmd == CMD_ROUND || cmd == CMD_QUEUES_ONLY || cmd == CMD_PASS1 || cmd == CMD_PASS2;
  if (isRound && t_ctx->deadlineS > 0 && t_ctx->deadlineS <= 1e-6) t_ctx->cancelWord = 1;
  if (!getenv("HS_NO_LDS_POISON")) memset((void*)&g_fl, 0xA5, sizeof g_fl);   // on the device this is LDS: whatever the previous kernel left there.  A launch that reads a field before writing it
                                                                              // must not pass here on a zeroed (or a still valid) one.
  if (cmd >= CMD_AUX_FIRST) controlMainAux(d, cmd); else controlMain(d, cmd);
  if (isRound && !t_ctx->inRound) t_ctx->cancelWord = 0;
  if (t_ctx->inRound) t_ctx->launches++;
  return 0;
}
// the grid-wide kernels of the split round, serially
static void plat_round_begin() { t_ctx->inRound = true; t_ctx->launches = 0; g_shardExchanges = 0; }
static void plat_round_end() { t_ctx->inRound = false; t_ctx->cancelWord = 0; }
static void plat_round_times(double* out) { out[0] = 0; out[1] = 0; out[2] = t_ctx->launches; }
static int plat_bulk(Dev& dev, int kind, int n) { Dev d = dev; for (int i = 0; i < n; i++) bulkElem(d, kind, i); t_ctx->launches++; return 0; }
static int plat_small(Dev& dev, int what, int arg) { Dev d = dev; roundSmall(d, what, arg); t_ctx->launches++; return 0; }
static int plat_agg(Dev& dev, int queued, int total) { Dev d = dev; if (queued) { for (int i = 0; i < total; i++) bulkElem(d, B_AGG_QUEUED, i); } else { for (int i = 0; i < total; i++) bulkElem(d, B_AGG_RUN, d.ordAll[i]); } return 0; }
static int plat_evict_apply(Dev& dev, int phase3, int total) {
  Dev d = dev;
  for (int i = 0; i < total; i++) evictApply(d, d.ordAll[i], phase3 != 0);
  t_ctx->launches++;
  return 0;
}
static int plat_compact(Dev& dev, const int32_t* order, int n, const uint8_t* flag, int32_t* dst, uint32_t* prefix, const int32_t* segOff, int nseg, int32_t* outSegOff, int* total) {
  (void)dev;
  int cnt = 0;
  for (int p = 0; p < n; p++) { int v = order ? order[p] : p; if (prefix) prefix[p] = (uint32_t)cnt; if (flag[v]) dst[cnt++] = v; }
  if (segOff) for (int q = 0; q <= nseg; q++) outSegOff[q] = segOff[q] < n ? (int32_t)prefix[segOff[q]] : cnt;
  *total = cnt;
  t_ctx->launches++;
  return 0;
}
// sorted base of the level-0 fast structure (round_fast.h): the device build sorts with a bitonic network
static int plat_build_base(Dev& d) {
  const DevCfg& c = d.cfg;
  int N = c.N;
  std::vector<uint64_t> keys(N);
  for (int i = 0; i < N; i++) keys[i] = fastKeyOf(d, i);
  std::sort(keys.begin(), keys.end());
  uint64_t mask = (1ull << c.idxBits) - 1;
  for (int i = 0; i < N; i++) {
    int node = d.nodeByRank[keys[i] & mask];
    d.baseKey[i] = keys[i]; d.baseNode[i] = node; d.posOf[node] = i; d.baseRemoved[i] = 0; d.baseCls[i] = d.nodeCls[node]; d.l0Slot[node] = -1;
    for (int e = 0; e < d.f.E; e++) d.baseExtra[(size_t)e * c.Npad + i] = AL(d, 0, d.f.extraCol[e], node);
  }
  if (d.fitBits) for (int f = 0; f < d.f.F; f++) for (int w = 0; w < d.fitW; w++) d.fitBits[(size_t)f * d.fitW + w] = fitBitsWord(d, f, w);
  return 0;
}
static int plat_run_shape_mask(Dev& d, const uint64_t* classMask, const int32_t* shapeClass) {
  const DevCfg& c = d.cfg;
  for (int s = 0; s < c.S; s++) for (int w = 0; w < c.W; w++) {
    uint64_t m = 0;
    for (int b = 0; b < 64; b++) {
      int n = w * 64 + b;
      if (n >= c.N) break;
      if (!((classMask[(size_t)shapeClass[s] * c.W + w] >> b) & 1)) continue;
      bool ok = true;
      for (int r = 0; r < c.R; r++) if (d.shapeReq[(size_t)s * c.R + r] > d.totalRes[(size_t)r * c.Npad + n]) ok = false;  // nodematching.go:184
      if (ok) m |= 1ull << b;
    }
    d.shapeMask[(size_t)s * c.W + w] = m;
  }
  return 0;
}
static int plat_run_fit_batch(Dev& d, const std::vector<int32_t>& shapes, int level, std::vector<int32_t>& out, const int32_t* = nullptr) {
  for (size_t i = 0; i < shapes.size(); i++) {
    ScanArgs a; memset(&a, 0, sizeof a);
    for (int r = 0; r < d.cfg.R; r++) a.req[r] = d.shapeReq[(size_t)shapes[i] * d.cfg.R + r];
    a.maskA = d.shapeMask + (size_t)shapes[i] * d.cfg.W; a.maskB = nullptr; a.level = level; a.levelHi = 0;
    out[i] = wgFirstFit(d, a);
  }
  return 0;
}
// the handle's communicator: the CPU build has no RCCL — only the caller's transport (asched_comm_init_external; gloo in the world-size-2 tests)
static int plat_comm_unique_id(char*) { g_err = "the CPU build of the tests has no RCCL: asched_comm_init_external only"; return -1; }
static int plat_comm_init(const char*, int, int) { g_err = "the CPU build of the tests has no RCCL: asched_comm_init_external only"; return -1; }
static int plat_comm_init_external(asched_allreduce_fn fn, void* ctx, int rank, int world) { t_ctx->extFn = fn; t_ctx->extCtx = ctx; t_ctx->commRank = rank; t_ctx->commWorld = world; return 0; }
static void plat_comm_destroy() { if (t_ctx) { t_ctx->extFn = nullptr; t_ctx->extCtx = nullptr; t_ctx->commRank = 0; t_ctx->commWorld = 1; } }
static bool plat_comm_live() { return t_ctx && t_ctx->extFn; }
static long plat_last_shard_exchanges() { return g_shardExchanges; }
// the GPU-to-GPU exchange (asched_shard_area / shard_open / shard_peers) is stores of one round kernel into another GPU's memory: the CPU build has the communicator path only
static int plat_shard_area(void**, char*) { g_err = "the CPU build of the tests has no exchange areas: asched_shard_round over asched_comm_init_external only"; return -1; }
static int plat_shard_open(const char*, void**) { g_err = "the CPU build of the tests has no exchange areas"; return -1; }
static int plat_shard_peers(void* const* areas, int, int) { if (!areas) return 0; g_err = "the CPU build of the tests has no exchange areas"; return -1; }
static void plat_comm_info(int* rank, int* world) { *rank = t_ctx ? t_ctx->commRank : 0; *world = t_ctx ? t_ctx->commWorld : 1; }
static int plat_allreduce(long long* buf, size_t count, int op) {
  if (!t_ctx->extFn) return 0;
  if (t_ctx->extFn(t_ctx->extCtx, buf, (int64_t)count, op) != 0) { g_err = "the external all-reduce transport failed"; return -1; }
  return 0;
}
// the submit check's gang units (armada_amd/csrc/submit_gang.h): the same per-unit function, one unit after the other on one thread
#include "../../armada_amd/csrc/submit_gang.h"
#define SG_MAX_NODES 262144
static int plat_run_submit_gangs(Dev& d, const std::vector<int32_t>& off, const std::vector<int32_t>& jobs, std::vector<int32_t>& out) {
  int nu = (int)off.size() - 1;
  out.assign((size_t)std::max(nu, 0) * 4, 0);
  std::vector<uint32_t> bits((d.cfg.N + 31) / 32 + 1, 0);
  static SgShared s;
  for (int u = 0; u < nu; u++) submitGangUnit(d, jobs.data() + off[u], off[u + 1] - off[u], s, bits.data(), out.data() + 4 * (size_t)u);
  return 0;
}
static int plat_run_fit_capacity(Dev& d, const std::vector<int32_t>& shapes, std::vector<int32_t>& firstNode, std::vector<long long>& capacity, const int32_t*) {
  firstNode.assign(shapes.size(), -1); capacity.assign(shapes.size(), 0);
  for (size_t i = 0; i < shapes.size(); i++) {
    unsigned long long best = ~0ull;
    for (int n = 0; n < d.cfg.N; n++) { long long c = sgNodeCapacity(d, shapes[i], n); capacity[i] += c; if (c > 0 && d.keys[n] < best) { best = d.keys[n]; firstNode[i] = n; } }
  }
  return 0;
}
// one pool on several GPUs (armada_amd/csrc/mgpu.h): the per-element functions of the grid kernels in serial loops
#include "../../armada_amd/csrc/mgpu.h"
#include "../../armada_amd/csrc/replay_rank.h"
int hsRankCheck(Dev& d) {
  int n = d.rs->numEvictedList, bad = 0;
  bool mono = true;
  for (int q = 0; q < d.cfg.Q; q++) if (d.evOff[q + 1] > d.evOff[q] && (!d.evMono[q] || !d.evCheap[q])) mono = false;
  if (!mono) { fprintf(stderr, "rank check: not every stream monotone / cheap — skipped\n"); return 0; }
  std::vector<int> idxOf(d.cfg.M, -1);
  for (int i = 0; i < d.rs->evictedTableSize; i++) idxOf[d.evTabJob[i]] = i;   // (dead entries have lost their evIndexOfJob)
  for (int p = 0; p < n; p++) {
    int want = idxOf[d.evList[p]], got = rrRank(d, p);
    if (want != got && bad++ < 10) {
      int q = 0; while (d.evOff[q + 1] <= p) q++;
      const EvKey e = d.evKey[p];
      fprintf(stderr, "rank check: position %d (queue %d, job %d) walk %d rank %d  key prop %.17g cur %.17g size %.17g pc %d budget %.17g\n", p, q, d.evList[p], want, got, e.proposed, e.current, e.size, e.pcPrio, d.qDc[q] / d.qWeight[q]);
    }
  }
  fprintf(stderr, "rank check: %d positions, %d differ\n", n, bad);
  return bad;
}
static int plat_replay_rank(Dev& d, int n, int keepPending) {   // armada_sched_mgpu.hip k_replay_check / k_replay_rank / k_replay_fin, serially
  if (n <= 0) return 0;
  if (rrOk(d, n)) { for (int p = 0; p < n; p++) rrElem(d, p); rrFinish(d, n); }
  else if (!keepPending) d.rs->replayPending = 0;
  return 0;
}
static int plat_run_fit_batch_global(Dev& d, const std::vector<int32_t>& shapes, const std::vector<int32_t>& slot, int level, GlobalKeyLayout L, const int32_t* globalRank, long long* out, int* badOut) {
  std::vector<int32_t> node(shapes.size(), -1);
  if (d.cfg.N > 0) plat_run_fit_batch(d, shapes, level, node);
  L.globalRank = globalRank;
  int32_t bad = 0;
  for (size_t i = 0; i < slot.size(); i++) { int n = node[slot[i]]; out[i] = mgpuPackQuery(d, L, level, n < 0 ? ~0ull : (unsigned long long)d.idxRank[n], &bad); }
  *badOut = bad;
  return 0;
}
static int plat_round_delta(Dev& d, int ns, int np, long long* buf) {
  memset(buf, 0, ((size_t)d.cfg.N * d.cfg.R + d.cfg.M) * 8);
  for (int i = 0; i < ns; i++) mgpuDeltaScheduled(d, buf, i);
  for (int i = 0; i < np; i++) mgpuDeltaPreempted(d, buf, i);
  return 0;
}
static int plat_delta_resolve(Dev& d, const long long* red, int ns, int np, int32_t* counts, int32_t* node, int32_t* prio, uint8_t* replay) {
  int N = d.cfg.N, M = d.cfg.M, R = d.cfg.R;
  std::vector<long long> freeC((size_t)N * R + 1, 0); std::vector<uint8_t> ownPre(M + 1, 0), conflict(N + 1, 0), gangReplay(std::max(d.cfg.G, 1), 0);
  for (int n = 0; n < N; n++) mgpuFreeInit(d, freeC.data(), n);
  for (int i = 0; i < ns; i++) mgpuFreeOwnScheduled(d, freeC.data(), i);
  for (int i = 0; i < np; i++) mgpuOwnPreempted(d, ownPre.data(), i);
  for (int j = 0; j < M; j++) mgpuFreeForeignPreempted(d, red, ownPre.data(), freeC.data(), j);
  for (int n = 0; n < N; n++) counts[0] += mgpuConflict(d, red, freeC.data(), conflict.data(), n);
  for (int j = 0; j < M; j++) mgpuGangConflict(d, red, conflict.data(), gangReplay.data(), j);
  for (int j = 0; j < M; j++) { int k = mgpuJobOutcome(d, red, conflict.data(), gangReplay.data(), node, prio, replay, j); if (k) counts[k]++; }
  return 0;
}
// fairness optimiser: the per-node routine of round_opt.h over all nodes, serially
static const double* g_lastQCost = nullptr;
static int plat_opt_qcosts(Dev&, double* out, int Q) { for (int q = 0; q < Q; q++) out[q] = g_lastQCost[q]; return 0; }
static int plat_opt_score(Dev& dev, const OptArgs& a, std::vector<OptNodeOut>& scores, double* jobCost, int detailNode, OptNodeOut* detail, std::vector<int32_t>* pre, bool detailOnly = false, bool reuseIndex = false) { (void)detailOnly; (void)reuseIndex;
  Dev d = dev;
  int N = d.cfg.N, M = d.cfg.M, Q = d.cfg.Q;
  std::vector<int32_t> off(N + 2, 0), jobs(2 * (size_t)std::max(M, 1));
  auto ghost = [&](int j) { return d.rs->optMode ? d.optGhost[j] : -1; };   // (dev.h optGhost)
  for (int j = 0; j < M; j++) { if (d.jobNode[j] >= 0) off[d.jobNode[j] + 1]++; if (ghost(j) >= 0) off[ghost(j) + 1]++; }
  for (int n = 0; n < N; n++) off[n + 1] += off[n];
  { std::vector<int32_t> cur(off.begin(), off.end()); for (int j = 0; j < M; j++) { if (d.jobNode[j] >= 0) jobs[cur[d.jobNode[j]]++] = j; if (ghost(j) >= 0) jobs[cur[ghost(j)]++] = j; } }
  static std::vector<double> qCost;
  qCost.assign(Q + 1, 0.0);
  for (int q = 0; q < Q; q++) { int64_t v[MAXR]; for (int r = 0; r < MAXR; r++) v[r] = r < d.cfg.R ? QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r] : 0; qCost[q] = d.optQDelta ? d.optQDelta[q] : drf(d, v); }
  g_lastQCost = qCost.data();
  *jobCost = drf(d, JREQ(d, a.job));
  scores.resize(N);
  // the device keeps OPT_MAXJ entries per thread and re-scores fuller nodes with a list in HBM (armada_sched.hip plat_opt_score); here: the private list first, the big list on overflow — the same two steps
  std::vector<OptEntry> big;
  auto score = [&](int n, OptNodeOut* o, int32_t* p) {
    optScoreNode(d, a, qCost.data(), off.data(), jobs.data(), d.jLeaseMs, n, o, p && off[n + 1] - off[n] <= OPT_MAXJ ? p : nullptr);
    if (o->scheduled < 0 || (p && off[n + 1] - off[n] > OPT_MAXJ)) { big.resize((size_t)(off[n + 1] - off[n]) + 1); optScoreNodeE(d, a, qCost.data(), off.data(), jobs.data(), d.jLeaseMs, n, o, p, big.data(), off[n + 1] - off[n]); }
  };
  for (int n = 0; n < N; n++) score(n, &scores[n], nullptr);
  if (detailNode >= 0) { pre->assign((size_t)std::max(off[detailNode + 1] - off[detailNode], OPT_MAXJ), -1); score(detailNode, detail, pre->data()); }
  return 0;
}
static double plat_last_opt_ms() { return 0; }
static int plat_opt_select(Dev&, const OptArgs&, double, bool, int32_t*, int32_t*, double*, double*, std::vector<int32_t>*) { return 1; }   // device-side selection: the CPU build always takes the host loop
// indicative pricer: round_price.h's per-node routine over all nodes, serially
static int plat_price_score(Dev& dev, const PriceArgs& a, std::vector<PriceNodeOut>& scores, int detailNode, std::vector<int32_t>* pre) {
  Dev d = dev;
  int N = d.cfg.N, M = d.cfg.M;
  std::vector<int32_t> off(N + 2, 0), jobs((size_t)std::max(M, 1));
  for (int j = 0; j < M; j++) if (d.jobNode[j] >= 0) off[d.jobNode[j] + 1]++;
  for (int n = 0; n < N; n++) off[n + 1] += off[n];
  { std::vector<int32_t> cur(off.begin(), off.end()); for (int j = 0; j < M; j++) if (d.jobNode[j] >= 0) jobs[cur[d.jobNode[j]]++] = j; }
  std::vector<PriceEntry> entries((size_t)std::max(M, 1) + 1);
  scores.resize(N);
  for (int n = 0; n < N; n++) priceScoreNode(d, a, off.data(), jobs.data(), d.jLeaseMs, n, &scores[n], nullptr, entries.data() + off[n]);
  if (detailNode >= 0) { PriceNodeOut o; pre->assign((size_t)std::max(off[detailNode + 1] - off[detailNode], 1), -1); priceScoreNode(d, a, off.data(), jobs.data(), d.jLeaseMs, detailNode, &o, pre->data(), entries.data() + off[detailNode]); }
  return 0;
}
static int plat_run_drf(Dev& dev, const std::vector<int64_t>& a, const std::vector<int64_t>& t, double* out) {
  Dev d = dev;
  for (int r = 0; r < d.cfg.R; r++) d.cfg.totalResources[r] = t[r];
  *out = drf(d, a.data());
  return 0;
}
static int plat_run_fair_shares(Dev& dev, int q, const int32_t* nameRank, const double* weight, const double* cds, double* fair, double* dc, double* uc) {
  Dev d = dev;
  d.cfg.Q = q;
  std::vector<double> w(weight, weight + q), f(q), c1(q), c2(q), pp(q), pc(q);
  std::vector<int32_t> nr(nameRank, nameRank + q), nx(q);
  std::vector<uint8_t> ih(q);
  d.qWeight = w.data(); d.qNameRank = nr.data(); d.qFair = f.data(); d.qDc = c1.data(); d.qUc = c2.data();
  d.pqProposed = pp.data(); d.pqCurrent = pc.data(); d.pqInHeap = ih.data(); d.itNext = nx.data();
  updateFairShares(d, cds);
  for (int i = 0; i < q; i++) { fair[i] = f[i]; dc[i] = c1[i]; uc[i] = c2[i]; }
  return 0;
}

#include "../../armada_amd/csrc/asched_host.inc"
