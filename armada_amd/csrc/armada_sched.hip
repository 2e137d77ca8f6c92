// armada_sched.hip — MI355X (gfx950) implementation of the C ABI in include/armada_sched.h.
//
// Kernels in this file:
//   k_control     persistent "round" kernel.  Workgroup 0: wave 0 runs the sequential DRF/gang control flow (round_ctl.h,
//                 round_fast.h), its 4 waves serve the data-parallel requests through an LDS mailbox; workgroups 1..H
//                 (round launches only) are helpers that share OP_SCAN / OP_FAIR through a mailbox in fine-grained HBM:
//                   OP_SCAN    first feasible node = argmin of the packed order key over nodes whose
//                              alloc[level][r][n] >= req[r]  (coalesced SoA planes, wave shuffle + LDS reduction)
//                   OP_BULK    evictors / unbind / populate / key rebuild as block-stride loops with int64 atomics
//                   OP_COMPACT order-preserving stream compaction (wave ballot + LDS prefix) for the per-queue
//                              evicted lists and the result lists
//                   OP_FAIR    fair-share preemption: per-node first covering Index over the evicted-table index, max
//   k_fit_batch   wide kernel: first feasible node for many (shape, level) queries against a fixed node state
//                 (BASELINE config 2, "nodedb fit kernel"): node tile in registers, wave-level min, one atomicMin/wave
//   k_shape_mask  per-shape static mask = requirement-class mask ∧ (total >= request)
//   k_drf / k_fair  float64 goldens (fairness.go / context/scheduling.go) evaluated on the device
//
// There is no CPU compute path in this library: without a gfx950 device asched_create() fails.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <type_traits>
#include <string>
#include <unistd.h>
#include <vector>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the functions are bound with dlsym at asched_comm_init (no link-time dependency on librccl)

#if defined(ASCHED_AUX_TU) && defined(HELP_TRACE)
#undef HELP_TRACE   // (the trace variant instruments the round kernel only)
#endif
#define ASCHED_PREFIX asched_
#include "round_run.h"
#include "round_opt.h"
#include "round_price.h"

// ------------------------------------------------------------------------------------------------ device primitives
#define CTL_THREADS 256
enum { OP_EXIT = 0, OP_SCAN = 1, OP_BULK = 2, OP_COMPACT = 3, OP_FAIR = 4, OP_ENGINE = 5, OP_BULKW = 6, OP_SCANFAIR = 7, OP_FTBUILD = 8, OP_HELPERS_EXIT = 9, OP_WIDE = 10 };
struct BulkWArgs { int32_t kind, n; };   // a bulk pass whose bodies touch HBM only: shared with the helper workgroups

struct Mailbox {
  int op, kind, n;
  ScanArgs scan;
  FairArgs fair;
  unsigned long long partial[16];
  const int32_t* order; const uint8_t* flag; int32_t* dst; uint32_t* prefix;
  int waveCount[16];
  int total;
};
__shared__ Mailbox g_mb;
__shared__ Dev g_dev;
#if defined(ASCHED_AUX_TU) || defined(ASCHED_WK_TU)
__shared__ MktDev g_mk;   // market-driven rounds (round_mkt.h): this launch's market state, a kernel argument of k_control_aux
__device__ static inline MktDev* mktDev() { return &g_mk; }
#endif

// Helper workgroups.  A round launch carries H extra workgroups (one per CU) that spin on a mailbox in HBM and take a share of
// the two read-only full-width queries of the generic path: the first-fit plane scan (OP_SCAN) and the per-node evaluation of
// fair-share preemption (OP_FAIR).  They read only HBM state (never the control workgroup's LDS); the hand-shake is a
// generation counter (release store by the control wave, relaxed polls + acquire fence by the helpers) and a completion
// counter (release increments, acquire poll) at agent scope, so it is correct across XCDs (separate L2s).
struct HelpSlot { unsigned long long gen, mn, mx, pad; };   // one per helper workgroup: written by that workgroup alone (plain stores, the generation last with release)
#define HELP_MAX 255
struct HelpBox {
  unsigned long long cmd;      // (generation << 8) | op, published with ONE release store: a helper can never pair a new generation with an old op
  unsigned long long pad[3];
  unsigned long long args[28]; // ScanArgs / FairArgs image (OP_SCANFAIR: ScanArgs at word 0, FairArgs at word HELP_ARGS2), read by the helpers with agent-scope loads
  // Results and completion (round 4): helper h folds its workgroup's minimum / maximum into slot[h] and stores the command's generation there LAST (release); the
  // control wave polls the generations one slot per lane and folds the values across its lanes.  No read-modify-write on a shared word: round 3 counted ~180 clocks
  // of serialised atomics per helper on result / result2 / done (23 k of a pass with 127 helpers, profiles/r03z_fair_index_alive_only.txt).
  HelpSlot slot[HELP_MAX];
};
#define HELP_ARGS2 14
static_assert(sizeof(ScanArgs) <= HELP_ARGS2 * 8 && sizeof(FairArgs) <= (28 - HELP_ARGS2) * 8 && sizeof(ScanArgs) % 8 == 0 && sizeof(FairArgs) % 8 == 0, "HelpBox args image");
__shared__ HelpBox* g_box;
__shared__ int g_H;
__shared__ unsigned int g_gen;

#ifdef HELP_TRACE
// Timeline of the fused wide pass (tools/build_variant.sh trace -DHELP_TRACE; never in the product): the control wave stamps the wall clock (s_memrealtime, 100 MHz, one time base for
// the whole chip) when it issues an OP_SCANFAIR; helper workgroups 1, H/2 and H add (their stamp - the issue stamp) at five points, the control workgroup at two.
__device__ unsigned long long g_traceT0[1024];
__device__ unsigned long long g_traceSum[64];   // [cls * 8 + point]: cls 0 control (0 own share done, 1 wait done), 1..3 helpers (0 seen, 1 args + acquire, 2 scan done, 3 fair done, 4 slot written); [56 + cls] counts
#define TRACE_ADD(cls, pt, gen) atomicAdd(&g_traceSum[(cls) * 8 + (pt)], (unsigned long long)(wall_clock64() - g_traceT0[(gen) & 1023]))
#endif
// 64-bit words of an object of another type: through a may_alias type.  (Round 2 read the argument structs through a plain unsigned long long* — undefined
// under strict aliasing: int64_t is `long`, so the compiler was free to treat the freshly written struct as never written; `minsize` on the callers made it do
// so and the helper workgroups received garbage requests: profiles/r03a_minsize_rootcause.txt.  The device code is also built with -fno-strict-aliasing now.)
typedef unsigned long long __attribute__((may_alias)) ull_alias;
template <class A, class B = A> __device__ static inline void helpIssue(int op, const A* args, const B* args2 = nullptr) {  // one lane of the control wave
  HelpBox* b = g_box;
  if (args) {
    const ull_alias* src = (const ull_alias*)args;
    for (int i = 0; i < (int)(sizeof(A) / 8); i++) __hip_atomic_store(&b->args[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (args2) {
    const ull_alias* src = (const ull_alias*)args2;
    for (int i = 0; i < (int)(sizeof(B) / 8); i++) __hip_atomic_store(&b->args[HELP_ARGS2 + i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  g_gen++;
  __hip_atomic_store(&b->cmd, ((unsigned long long)g_gen << 8) | (unsigned)op, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static inline unsigned long long waveMin64Dpp(unsigned long long v);
// Completion of the command issued last: every helper's slot carries its generation.  Whole control wave, uniformly (no lane-divergent spin): lane l watches
// slots l, l + 64, ...  Returns the folded minimum; the folded maximum goes back through *mxOut — both in REGISTERS, wave-uniform (round 3's slot variants handed
// the maximum back through a new __shared__ word written by lane 0 and read by the wave with nothing in between: profiles/r04a_lds_handback_rootcause.txt).
__device__ static inline unsigned long long helpWait(unsigned long long* mxOut = nullptr) {
  HelpBox* b = g_box;
  int lane = threadIdx.x & 63;
  int H = __builtin_amdgcn_readfirstlane(g_H);
  unsigned long long gen = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)g_gen);   // (written by lane 0 in helpIssue; a workgroup barrier lies between)
  unsigned long long mn = ~0ull, mxInv = ~0ull;
  unsigned int spins = 0;
  bool gaveUp = false;
  for (int base = 0; base < H && !gaveUp; base += 64) {
    int i = base + lane;
    bool mine = i < H;
    for (;;) {
      unsigned long long g = mine ? __hip_atomic_load(&b->slot[mine ? i : 0].gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : gen;
      if (__ballot(g != gen) == 0) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 0xffff) == 0 && cancelRequested(g_dev)) {  // the caller gave up (hard timeout): do not wait for a helper that may never answer
        raise(g_dev, ASCHED_ERR_TIMEOUT, 902);
        gaveUp = true;
        break;
      }
      if ((spins & 0xffff) == 0 && g_dev.progress) { g_dev.progress[5] = __popcll(__ballot(g == gen)); g_dev.progress[6] = H; g_dev.progress[7] = (int)gen; g_dev.progress[8] = (int)(b->cmd >> 8); g_dev.progress[9] = (int)(b->cmd & 255); }
    }
    if (mine && !gaveUp) {
      unsigned long long a = __hip_atomic_load(&b->slot[i].mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long m = ~__hip_atomic_load(&b->slot[i].mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mn = a < mn ? a : mn; mxInv = m < mxInv ? m : mxInv;
    }
  }
  mn = waveMin64Dpp(mn);
  if (mxOut) *mxOut = ~waveMin64Dpp(mxInv);   // max(x) = ~min(~x)
  return mn;
}

__device__ static inline void atomicAddI64(int64_t* p, int64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
__device__ static inline void atomicAddI32(int32_t* p, int32_t v) { atomicAdd(p, v); }
__device__ static inline void atomicOrI32(int32_t* p, int32_t v) { atomicOr(p, v); }
__device__ static inline void atomicMinU32(uint32_t* p, uint32_t v) { atomicMin(p, v); }
__device__ static inline int atomicFetchAddI32(int32_t* p, int32_t v) { return atomicAdd(p, v); }
__device__ static inline int waveMax32(int v) {
  for (int off = 32; off; off >>= 1) { int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return v;
}

__device__ static inline unsigned long long waveMin64(unsigned long long v) {
  for (int off = 32; off; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// ---- cross-lane moves on the VALU (DPP) instead of through the LDS crossbar.  __shfl_* compile to ds_bpermute_b32: an LDS round trip (>100 clocks)
// per 32-bit word, which a lone wave cannot hide; a DPP move is one VALU instruction.  gfx9-family controls: row_shr:n = 0x110+n, wave_shl:1 = 0x130,
// row_half_mirror = 0x141, row_bcast:15 = 0x142, row_bcast:31 = 0x143, quad_perm = 0x00..0xff.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf> __device__ static inline unsigned long long dppMove64(unsigned long long old, unsigned long long v) {
  unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)old, (int)(unsigned)v, CTRL, ROW_MASK, BANK_MASK, false);
  unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(old >> 32), (int)(unsigned)(v >> 32), CTRL, ROW_MASK, BANK_MASK, false);
  return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf> __device__ static inline int dppMove32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false);
}
// minimum over the 64 lanes, returned wave-uniform.  min is idempotent: a lane without a valid source keeps its own value (old = v).
__device__ static inline unsigned long long waveMin64Dpp(unsigned long long v) {
  unsigned long long t;
  t = dppMove64<0x111>(v, v); v = t < v ? t : v;              // row_shr:1
  t = dppMove64<0x112>(v, v); v = t < v ? t : v;              // row_shr:2
  t = dppMove64<0x114>(v, v); v = t < v ? t : v;              // row_shr:4
  t = dppMove64<0x118>(v, v); v = t < v ? t : v;              // row_shr:8  -> lane 15 of every row of 16 holds the row's minimum
  t = dppMove64<0x142, 0xa>(v, v); v = t < v ? t : v;         // row_bcast:15 into rows 1 and 3
  t = dppMove64<0x143, 0xc>(v, v); v = t < v ? t : v;         // row_bcast:31 into rows 2 and 3 -> lane 63 holds the minimum
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// one thread per node (block-stride): reject by mask bit and key first, touch the alloc planes only for improving candidates
__device__ static unsigned long long scanPart(const Dev& d, const ScanArgs& a, int tid, int nthreads) {
  const DevCfg& c = d.cfg;
  unsigned long long best = ~0ull;
  if (a.levelHi > a.level) {  // multi-level mode: per node the lowest level in [level, levelHi] it fits at, tagged key
    for (int n = SHARD_LO(c) + tid; n < SHARD_HI(c); n += nthreads) {
      uint64_t w = a.maskA[n >> 6];
      if (a.maskB) w &= a.maskB[n >> 6];
      if (!((w >> (n & 63)) & 1)) continue;
      for (int l = a.level; l <= a.levelHi; l++) {
        unsigned long long v = ((unsigned long long)l << SCAN_LEVEL_SHIFT) | d.keys[(size_t)l * c.Npad + n];
        if (v >= best) break;   // higher levels only order later
        const int64_t* plane = d.alloc + (size_t)l * c.R * c.Npad;
        bool fits = true;
        for (int r = 0; r < c.R; r++) fits = fits && (a.req[r] <= plane[(size_t)r * c.Npad + n]);
        if (fits) { best = v; break; }
      }
    }
    return waveMin64(best);
  }
  const uint64_t* keys = d.keys + (size_t)a.level * c.Npad;
  const int64_t* plane = d.alloc + (size_t)a.level * c.R * c.Npad;
#ifdef ASCHED_TWO_WORD_KEYS
  if (WIDE_KEYS(c)) {   // two-word keys (dev.h keyWords): pass 1 (a.pad == 0) the minimum HIGH word among the fitting nodes, pass 2 (a.pad == 1, a.lowBound = that word) the minimum LOW word among those that carry it
    const uint64_t* lows = d.keys + ((size_t)c.P + a.level) * c.Npad;
    for (int n = SHARD_LO(c) + tid; n < SHARD_HI(c); n += nthreads) {
      uint64_t w = a.maskA[n >> 6];
      if (a.maskB) w &= a.maskB[n >> 6];
      if (!((w >> (n & 63)) & 1)) continue;
      unsigned long long k = keys[n];
      if (a.pad) { if (k != a.lowBound) continue; k = lows[n]; if (k < a.lowBoundLo) continue; }
      else if (k < a.lowBound || (k == a.lowBound && a.lowBoundLo && lows[n] < a.lowBoundLo)) continue;
      if (k >= best) continue;
      bool fits = true;
      if (!a.noFit) for (int r = 0; r < c.R; r++) fits = fits && (a.req[r] <= plane[(size_t)r * c.Npad + n]);
      if (fits) best = k;
    }
    return waveMin64(best);
  }
#endif
  for (int n = SHARD_LO(c) + tid; n < SHARD_HI(c); n += nthreads) {
    uint64_t w = a.maskA[n >> 6];
    if (a.maskB) w &= a.maskB[n >> 6];
    if (!((w >> (n & 63)) & 1)) continue;
    unsigned long long k = keys[n];
    if (k >= best || k < a.lowBound) continue;
    bool fits = true;
    if (!a.noFit) for (int r = 0; r < c.R; r++) fits = fits && (a.req[r] <= plane[(size_t)r * c.Npad + n]);
    if (fits) best = k;
  }
  return waveMin64(best);
}

#ifdef ASCHED_SHARDED_PASSES
// A sharded wide pass (dev.h shardWorld): the minimum of the word and the maximum of the index over the replicas' shares.  The control wave posts its two words (the
// index as its complement: one MIN all-reduce serves both) in the handle's host-mapped block and waits for the answer of the host thread that drives the launch
// (plat_run_control: the all-reduce runs on the handle's communicator).  A PCIe round trip + the collective per pass: worth it where a pass is long (100 000 nodes and
// up) — measured numbers for one GPU only.  Bounded like every wait of this kernel: the caller's cancel word ends it.
__shared__ unsigned int g_xgen;
__shared__ unsigned long long g_xpeers;   // 0: the exchange goes through the host proxy; else the device address of the peer table (asched_shard_peers): GPU-to-GPU
// The same exchange GPU-to-GPU (asched_shard_peers): every replica owns an exchange area in its HBM (fine-grained; the peers map it: peer access inside a process, hipIpc across
// processes) — word 0 the owner's generation counter (it outlives a launch), from word 8 two banks (generation parity) of one 4-word slot per rank: {generation, word 0, word 1, -}.
// Lane r of the control wave stores this rank's two words and then the generation (release, system scope) into ITS slot of rank r's area — over xGMI for a remote r —, then
// watches slot r of the own area; when every rank's generation is there the words are folded across the lanes.  No host, no PCIe: an exchange is a round of posted stores
// and one polling read of local HBM.  A rank can run at most one exchange ahead of another (it needs the other's words to finish its own), hence two banks.
__device__ static inline void shardReduceDirect(Dev& d, unsigned long long* mn, int* mxIdx) {
  unsigned long long* const* peers = (unsigned long long* const*)g_xpeers;
  const int lane = threadIdx.x & 63, W = d.cfg.shardWorld, me = d.cfg.shardRank;
  unsigned int gen = (unsigned int)__builtin_amdgcn_readfirstlane((int)g_xgen) + 1;
  const size_t bank = (size_t)(gen & 1) * 256;
  unsigned long long w0 = *mn, w1 = ~(unsigned long long)(unsigned int)(*mxIdx + 1);
  unsigned long long* own = peers[me];
  if (lane == 0) { g_xgen = gen; __hip_atomic_store(own, (unsigned long long)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  if (lane < W) {
    unsigned long long* slot = peers[lane] + 8 + (bank + me) * 4;
    __hip_atomic_store(slot + 1, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 2, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot, (unsigned long long)gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned long long* mine = own + 8 + (bank + (lane < W ? lane : 0)) * 4;
  unsigned int spins = 0;
  for (;;) {
    bool there = lane >= W || __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == gen;
    if (__ballot(!there) == 0) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 0xfff) == 0 && cancelRequested(d)) { raise(d, ASCHED_ERR_TIMEOUT, 904); return; }
  }
  unsigned long long v0 = lane < W ? __hip_atomic_load(mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
  unsigned long long v1 = lane < W ? __hip_atomic_load(mine + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
  v0 = waveMin64Dpp(v0); v1 = waveMin64Dpp(v1);
  *mn = v0; *mxIdx = (int)(unsigned int)~v1 - 1;
}
__device__ static inline void shardReduce(Dev& d, unsigned long long* mn, int* mxIdx) {
  if (__builtin_amdgcn_readfirstlane((int)(g_xpeers != 0))) { shardReduceDirect(d, mn, mxIdx); return; }
  unsigned long long* X = (unsigned long long*)d.cancel;
  if (!X) { raise(d, ASCHED_ERR_INTERNAL, 530); return; }
  int lane = threadIdx.x & 63;
  unsigned int gen = (unsigned int)__builtin_amdgcn_readfirstlane((int)g_xgen) + 1;
  unsigned long long w0 = *mn, w1 = ~(unsigned long long)(unsigned int)(*mxIdx + 1);
  if (lane == 0) {
    g_xgen = gen;
    __hip_atomic_store(&X[XCHG_WORD0 + 1], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&X[XCHG_WORD0 + 2], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&X[XCHG_WORD0], (unsigned long long)gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned int spins = 0;
  for (;;) {
    unsigned long long g = __hip_atomic_load(&X[XCHG_WORD0 + 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (g == gen) break;
    __builtin_amdgcn_s_sleep(4);
    if ((++spins & 0x3ff) == 0 && cancelRequested(d)) { raise(d, ASCHED_ERR_TIMEOUT, 903); return; }
  }
  w0 = __hip_atomic_load(&X[XCHG_WORD0 + 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  w1 = __hip_atomic_load(&X[XCHG_WORD0 + 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  *mn = w0; *mxIdx = (int)(unsigned int)~w1 - 1;
}
#define SHARD_REDUCE(d, mn, idx) do { if (SHARD_ON((d).cfg)) shardReduce(d, &(mn), &(idx)); } while (0)
#define SHARD_REDUCE_MIN(d, mn) do { if (SHARD_ON((d).cfg)) { int none_ = -1; shardReduce(d, &(mn), &none_); } } while (0)
#define SHARD_REDUCE_MAX(d, idx) do { if (SHARD_ON((d).cfg)) { unsigned long long none_ = ~0ull; shardReduce(d, &none_, &(idx)); } } while (0)
#else
#define SHARD_REDUCE(d, mn, idx) do {} while (0)
#define SHARD_REDUCE_MIN(d, mn) do {} while (0)
#define SHARD_REDUCE_MAX(d, idx) do {} while (0)
#endif
#ifdef ASCHED_TWO_WORD_KEYS
__device__ static inline uint64_t wgFirstFitKeyPass(Dev& d, const ScanArgs& a) {
#else
__device__ static inline uint64_t wgFirstFitKey(Dev& d, const ScanArgs& a) {
#endif
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_mb.op = OP_SCAN; g_mb.scan = a; if (g_H) helpIssue(OP_SCAN, &a); }
  __syncthreads();
  unsigned long long v = scanPart(d, g_mb.scan, threadIdx.x, (g_H + 1) * (int)blockDim.x);
  if (lane == 0) g_mb.partial[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long best = ~0ull;
  int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) { unsigned long long p = g_mb.partial[w]; best = p < best ? p : best; }
  if (g_H) {
    unsigned long long hb = helpWait();
    best = hb < best ? hb : best;
  }
  SHARD_REDUCE_MIN(d, best);
  d.rs->numScans++;
  return best;
}
#ifdef ASCHED_TWO_WORD_KEYS
// -> the minimum order key among the fitting nodes, ~0 = none.  Two-word keys: the LOW word of the minimum (it carries the node-index rank the callers look at), found in two
// passes — the order is (high word, low word), so the second pass only looks at the nodes that carry the first pass's high word.
__device__ static inline uint64_t wgFirstFitKey(Dev& d, const ScanArgs& a) {
  uint64_t best = wgFirstFitKeyPass(d, a);
  if (WIDE_KEYS(d.cfg)) {
    if (a.levelHi > a.level) { raise(d, ASCHED_ERR_UNSUPPORTED, 520); return ~0ull; }   // (the fused multi-level pass is one-word only: the host never selects it)
    if (best == ~0ull) return best;
    ScanArgs b = a; b.pad = 1; b.lowBound = best; b.lowBoundLo = best == a.lowBound ? a.lowBoundLo : 0;
    best = wgFirstFitKeyPass(d, b);
  }
  return best;
}
#endif
__device__ static inline int wgFirstFit(Dev& d, const ScanArgs& a) {
  unsigned long long best = wgFirstFitKey(d, a);
  if (best == ~0ull) return -1;
  return d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];
}

// one thread per node (grid-stride over the participating workgroups): highest evicted-table Index at which a node covers the request
__device__ static int fairPart(const Dev& d, const FairArgs& a, int tid, int nthreads) {
  int best = -1;
  for (int n = SHARD_LO(d.cfg) + tid; n < SHARD_HI(d.cfg); n += nthreads) { int v = fairNodeBest(d, a, n, best); best = v > best ? v : best; }
  return waveMax32(best);
}
__device__ static inline int wgFairSelect(Dev& d, const FairArgs& a) {
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_mb.op = OP_FAIR; g_mb.fair = a; if (g_H) helpIssue(OP_FAIR, &a); }
  __syncthreads();
  int v = fairPart(d, g_mb.fair, threadIdx.x, (g_H + 1) * (int)blockDim.x);
  if (lane == 0) g_mb.waveCount[threadIdx.x >> 6] = v;
  __syncthreads();
  int best = -1;
  int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) { int p = g_mb.waveCount[w]; best = p > best ? p : best; }
  if (g_H) {
    unsigned long long hmx;
    (void)helpWait(&hmx);
    int h = (int)(unsigned int)hmx - 1;
    best = h > best ? h : best;
  }
  SHARD_REDUCE_MAX(d, best);
  return best;
}

// gate + fair-share evaluation in one pass (selectAtPriority, round_ctl.h): every participating thread walks its nodes once for each question; one
// command, one completion count, two results
__device__ static inline int wgScanFair(Dev& d, const ScanArgs& a, const FairArgs& f, uint64_t* bestKey) {
  int lane = threadIdx.x & 63;
#ifdef HELP_TRACE
  if (lane == 0 && g_H) { g_traceT0[(g_gen + 1) & 1023] = wall_clock64(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
#endif
  if (lane == 0) { g_mb.op = OP_SCANFAIR; g_mb.scan = a; g_mb.fair = f; if (g_H) helpIssue(OP_SCANFAIR, &a, &f); }
  __syncthreads();
  unsigned long long v = scanPart(d, g_mb.scan, threadIdx.x, (g_H + 1) * (int)blockDim.x);
  int w = fairPart(d, g_mb.fair, threadIdx.x, (g_H + 1) * (int)blockDim.x);
  if (lane == 0) { g_mb.partial[threadIdx.x >> 6] = v; g_mb.waveCount[threadIdx.x >> 6] = w; }
  __syncthreads();
#ifdef HELP_TRACE
  if (lane == 0 && g_H) { TRACE_ADD(0, 0, g_gen); atomicAdd(&g_traceSum[56], 1ull); }
#endif
  unsigned long long best = ~0ull; int idx = -1;
  int nw = blockDim.x >> 6;
  for (int k = 0; k < nw; k++) { unsigned long long p = g_mb.partial[k]; best = p < best ? p : best; int q = g_mb.waveCount[k]; idx = q > idx ? q : idx; }
  if (g_H) {
    unsigned long long hmx;
    unsigned long long hb = helpWait(&hmx);
#ifdef HELP_TRACE
    if (lane == 0) TRACE_ADD(0, 1, g_gen);
#endif
    best = hb < best ? hb : best;
    int h = (int)(unsigned int)hmx - 1;
    idx = h > idx ? h : idx;
  }
  SHARD_REDUCE(d, best, idx);
  d.rs->numScans++;
  *bestKey = best;
  return idx;
}

// A bulk pass over n elements on the control workgroup AND the helper workgroups (grid-stride over all of them).  Only for bodies that read and write HBM and
// nothing the control workgroup keeps in LDS (the stream preparation, round_run.h B_QS*): the helpers see the kernel argument's Dev, i.e. the HBM homes.
__device__ static inline void wgBulkWide(Dev& d, int kind, int n) {
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_mb.op = OP_BULKW; g_mb.kind = kind; g_mb.n = n; if (g_H) { BulkWArgs a; a.kind = kind; a.n = n; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); helpIssue(OP_BULKW, &a); } }
  __syncthreads();
  { int nthreads = (g_H + 1) * (int)blockDim.x; int kd = g_mb.kind, nn = g_mb.n; for (int i = threadIdx.x; i < nn; i += nthreads) bulkElem(d, kd, i); }
  __threadfence();
  __syncthreads();
  if (g_H) { (void)helpWait(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
}

// one pass of a wide run's preparation / commit (round_wide.h): like wgBulkWide, with its own op and an out-of-line body — a call inside bulkElem's switch moves the
// hot loops of the headline round
__device__ static inline void wgWide(Dev& d, int kind, int n) {
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_mb.op = OP_WIDE; g_mb.kind = kind; g_mb.n = n; if (g_H) { BulkWArgs a; a.kind = kind; a.n = n; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); helpIssue(OP_WIDE, &a); } }
  __syncthreads();
  { int nthreads = (g_H + 1) * (int)blockDim.x; int kd = g_mb.kind, nn = g_mb.n; for (int i = threadIdx.x; i < nn; i += nthreads) wideBulkAny(d, kd, i); }
  __threadfence();
  __syncthreads();
  if (g_H) { (void)helpWait(); }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

#ifndef ASCHED_NO_FT
// one pass of the fair-share threshold table's build (round_ft.h): like wgBulkWide, with its own op and an out-of-line body
__device__ static inline void wgFtBuild(Dev& d, int phase, int n) {
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_mb.op = OP_FTBUILD; g_mb.kind = phase; g_mb.n = n; if (g_H) { BulkWArgs a; a.kind = phase; a.n = n; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); helpIssue(OP_FTBUILD, &a); } }
  __syncthreads();
  { int nthreads = (g_H + 1) * (int)blockDim.x; int ph = g_mb.kind, nn = g_mb.n; for (int i = threadIdx.x; i < nn; i += nthreads) ftBuildAny(d, ph, i); }
  __threadfence();
  __syncthreads();
  if (g_H) { (void)helpWait(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
}
#endif

// Most elements of the per-job passes do nothing (queued jobs have no node, few jobs are flagged for eviction): read the one
// field that decides that for BULK_U elements at once — independent loads, all in flight together — and run the body only for
// the survivors.  The bodies themselves are unchanged (round_run.h bulkElem).
#define BULK_U 8
__device__ static inline int bulkGate(Dev& d, int kind, int i) {
  switch (kind) {
    case B_EVICT_APPLY1: case B_EVICT_APPLY3: return d.evFlag[i];
    case B_UNBIND: return d.inPreempted[i] | d.inSchedAndEvicted[i];
    case B_FILTER1: case B_FILTER3: return d.jobNode[i] >= 0;
  }
  return 1;
}
__device__ static void bulkPart(Dev& d, int kind, int n) {
  int stride = blockDim.x;
  if (kind == B_EVICT_APPLY1 || kind == B_EVICT_APPLY3 || kind == B_UNBIND || kind == B_FILTER1 || kind == B_FILTER3) {
    bool filter = kind == B_FILTER1 || kind == B_FILTER3;
    for (int base = threadIdx.x; base < n; base += stride * BULK_U) {
      int gate[BULK_U];
#pragma unroll
      for (int u = 0; u < BULK_U; u++) { int i = base + u * stride; gate[u] = i < n ? bulkGate(d, kind, i) : -1; }
#pragma unroll
      for (int u = 0; u < BULK_U; u++) {
        int i = base + u * stride;
        if (gate[u] > 0) bulkElem(d, kind, i);
        else if (gate[u] == 0 && filter) d.evFlag[i] = 0;  // a job without a node is never evicted (pqs.go:101-136, eviction.go:158-178)
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += stride) bulkElem(d, kind, i);
  }
  __threadfence();  // int64 atomics land in L2: make them (and the plain stores) visible to the control wave
}
__device__ static inline void wgBulk(Dev& d, int kind, int n) {
  if (n <= 0) return;
  if ((threadIdx.x & 63) == 0) { g_mb.op = OP_BULK; g_mb.kind = kind; g_mb.n = n; }
  __syncthreads();
  bulkPart(d, g_mb.kind, g_mb.n);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// order-preserving compaction of {order[p] : flag[order[p]]} (order == NULL: identity); prefix[p] = #flagged before p
__device__ static int compactPart(Dev& d) {
  (void)d;
  int n = g_mb.n;
  const int32_t* order = g_mb.order; const uint8_t* flag = g_mb.flag; int32_t* dst = g_mb.dst; uint32_t* prefix = g_mb.prefix;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int base = 0;
  const int U = 4;  // element / flag loads of U consecutive tiles are issued together; the ordered prefix then runs tile by tile
  for (int start0 = 0; start0 < n; start0 += blockDim.x * U) {
    int vv[U]; bool ff[U];
#pragma unroll
    for (int u = 0; u < U; u++) { int p = start0 + u * (int)blockDim.x + (int)threadIdx.x; vv[u] = p < n ? (order ? order[p] : p) : 0; }
#pragma unroll
    for (int u = 0; u < U; u++) { int p = start0 + u * (int)blockDim.x + (int)threadIdx.x; ff[u] = p < n && flag[vv[u]]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      int start = start0 + u * (int)blockDim.x;
      if (start >= n) break;
      int p = start + threadIdx.x;
      int v = vv[u];
      bool f = ff[u];
      unsigned long long b = __ballot(f);
      int rank = __popcll(b & ((1ull << lane) - 1));
      if (lane == 0) g_mb.waveCount[wave] = __popcll(b);
      __syncthreads();
      int off = 0, tot = 0;
      for (int w = 0; w < nw; w++) { int cw = g_mb.waveCount[w]; if (w < wave) off += cw; tot += cw; }
      if (p < n && prefix) prefix[p] = base + off + rank;
      if (f) dst[base + off + rank] = v;
      base += tot;
      __syncthreads();
    }
  }
  __threadfence();
  return base;
}
__device__ static inline int wgCompactRun(Dev& d, const int32_t* order, int n, const uint8_t* flag, int32_t* dst, uint32_t* prefix) {
  if ((threadIdx.x & 63) == 0) { g_mb.op = OP_COMPACT; g_mb.n = n; g_mb.order = order; g_mb.flag = flag; g_mb.dst = dst; g_mb.prefix = prefix; }
  __syncthreads();
  int total = compactPart(d);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return total;
}
__device__ static inline int wgCompactFlagged(Dev& d, const int32_t* order, const int32_t* segOff, int nseg, int n, const uint8_t* flag, int32_t* dst, int32_t* outSegOff) {
  int total = wgCompactRun(d, order, n, flag, dst, d.evSortKey);
  for (int base = 0; base <= nseg; base += 64) {  // same trip count on every lane
    int q = base + (int)(threadIdx.x & 63);
    if (q <= nseg) outSegOff[q] = segOff[q] < n ? (int32_t)d.evSortKey[segOff[q]] : total;
  }
  __threadfence();
  return total;
}
__device__ static inline int wgCompactIota(Dev& d, int n, const uint8_t* flag, int32_t* dst) { return wgCompactRun(d, nullptr, n, flag, dst, nullptr); }

// lane-per-queue argmin under the reference's Less (a strict total order, so argmin == heap top).  Every lane reads its queue's Less inputs once; the
// tournament moves the VALUES between lanes (no memory access per round)
struct PqVal { int q; int32_t prio; double proposed, budget, current, size; int32_t name; int32_t away; };
__device__ static inline bool pqLessV(const Ctl& c, const PqVal& a, const PqVal& b) {   // pqLess (round_ctl.h) on values
  if (a.away != b.away) return !a.away;   // (0 everywhere unless preemptCrossPoolJobsFirst is set and cross-pool away jobs exist)
  if (a.prio != b.prio) return a.prio > b.prio;
  if (c.preferLarge) {
    if (a.proposed <= a.budget && b.proposed <= b.budget) {
      if (a.current == b.current && a.size != b.size) return a.size > b.size;
      if (a.current != b.current) return a.current < b.current;
    } else if (a.proposed > a.budget && b.proposed > b.budget) {
      if (a.proposed != b.proposed) return a.proposed < b.proposed;
    } else if (a.proposed <= a.budget) return true;
    else if (b.proposed <= b.budget) return false;
  } else {
    if (a.proposed != b.proposed) return a.proposed < b.proposed;
  }
  return a.name < b.name;
}
__device__ static inline double shflD(double v, int off) { return __shfl_xor(v, off, 64); }
__device__ static inline int pqTop(Dev& d, const Ctl& c) {
  int lane = threadIdx.x & 63;
  PqVal best; best.q = -1; best.prio = 0; best.proposed = best.budget = best.current = best.size = 0; best.name = 0; best.away = 0;
  int Q = d.cfg.Q;
  const bool homeFirst = d.cfg.preferHome && d.jAway;
  for (int base = 0; base < Q; base += 64) {  // same trip count on every lane: the wave stays converged for the shuffles below
    int q = base + lane;
    bool in = q < Q && d.pqInHeap[q < Q ? q : 0];
    if (in) {
      PqVal v; v.q = q; v.prio = c.compareSchedPrio ? d.pqSchedPrio[q] : d.pqPcPrio[q]; v.proposed = d.pqProposed[q]; v.budget = d.pqBudget[q]; v.current = d.pqCurrent[q]; v.size = d.pqSize[q]; v.name = d.qNameRank[q];
      v.away = homeFirst ? (pqAway(d, q) ? 1 : 0) : 0;
      if (best.q < 0 || pqLessV(c, v, best)) best = v;
    }
  }
  for (int off = 32; off; off >>= 1) {
    PqVal o; o.q = __shfl_xor(best.q, off, 64); o.prio = __shfl_xor(best.prio, off, 64); o.proposed = shflD(best.proposed, off); o.budget = shflD(best.budget, off);
    o.current = shflD(best.current, off); o.size = shflD(best.size, off); o.name = __shfl_xor(best.name, off, 64); o.away = __shfl_xor(best.away, off, 64);
    if (o.q >= 0 && (best.q < 0 || pqLessV(c, o, best))) best = o;
  }
  return best.q;
}


// ------------------------------------------------------------------------------------------------ fast-path primitives (round_fast.h contracts)
// All of these run on wave 0 only (the control wave); results are wave-uniform.  LDS state is reached through the
// __shared__ objects themselves (ds_* instructions), HBM through explicit global-address-space pointers (FastK): no flat
// accesses, so LDS work never waits for the outstanding HBM stores/atomics and vice versa.
__device__ static inline bool keyLess(uint32_t A, unsigned long long X, unsigned long long Y, uint32_t N, uint32_t oA, unsigned long long oX, unsigned long long oY, uint32_t oN) {
  return A != oA ? A < oA : X != oX ? X < oX : Y != oY ? Y < oY : N < oN;
}
// sort the queues in the heap by key across the lanes (rank by counting; Q <= 64, done once per fastRun)
__device__ static inline void pqBuild(PQState& s, int Q) {
  int lane = threadIdx.x & 63;
  bool in = lane < Q && g_fl.inHeap[lane];
  uint32_t A = in ? g_fl.kA[lane] : ~0u, N = lane < Q ? (uint32_t)g_fl.nameRank[lane] : ~0u;
  unsigned long long X = in ? g_fl.kX[lane] : ~0ull, Y = in ? g_fl.kY[lane] : ~0ull;
  int rank = 0;
  for (int j = 0; j < 64; j++) {
    uint32_t jA = __shfl(A, j, 64), jN = __shfl(N, j, 64); unsigned long long jX = __shfl(X, j, 64), jY = __shfl(Y, j, 64);
    int jin = __shfl((int)in, j, 64);
    bool before = jin && !in ? true : (!jin && in ? false : (keyLess(jA, jX, jY, jN, A, X, Y, N) || (jA == A && jX == X && jY == Y && jN == N && j < lane)));
    if (j != lane && before) rank++;
  }
  // scatter by rank through LDS, gather in lane order
  g_fl.tmpA[rank] = A; g_fl.tmpN[rank] = N; g_fl.tmpX[rank] = X; g_fl.tmpY[rank] = Y; g_fl.tmpQ[rank] = in ? lane : -1;
  s.A = g_fl.tmpA[lane]; s.N = g_fl.tmpN[lane]; s.X = g_fl.tmpX[lane]; s.Y = g_fl.tmpY[lane]; s.q = g_fl.tmpQ[lane];
  s.count = __popcll(__ballot(in));
}
__device__ static inline int pqHead(PQState& s, int Q) {
  int q = __builtin_amdgcn_readfirstlane(s.q);
  return (s.count > 0 && q >= 0 && q < Q) ? q : -1;
}
// the head (queue q) was served: drop it and, if it still has a candidate gang, insert it again under its new key
__device__ static inline void pqPopPush(PQState& s, const KeyOut& ko, int q) {
  int lane = threadIdx.x & 63;
  uint32_t hN = __builtin_amdgcn_readfirstlane(s.N);  // the name rank travels with the entry
  // everything after the head moves up one lane
  // wave_shl:1 — lane i takes lane i+1's value; lane 63 has no source and is overwritten below (lane >= cnt)
  uint32_t dA = (uint32_t)dppMove32<0x130>((int)s.A, (int)s.A), dN = (uint32_t)dppMove32<0x130>((int)s.N, (int)s.N);
  unsigned long long dX = dppMove64<0x130>(s.X, s.X), dY = dppMove64<0x130>(s.Y, s.Y);
  int dq = dppMove32<0x130>(s.q, s.q);
  int cnt = s.count - 1;  // entries other than the head
  if (lane >= cnt) { dA = ~0u; dN = ~0u; dX = ~0ull; dY = ~0ull; dq = -1; }
  if (!ko.valid) { s.A = dA; s.N = dN; s.X = dX; s.Y = dY; s.q = dq; s.count = cnt; return; }
  // position of the new key among the others = number of them that order before it
  bool before = lane < cnt && keyLess(dA, dX, dY, dN, ko.A, ko.X, ko.Y, hN);
  int pos = __popcll(__ballot(before));
  // lanes < pos take the shifted entry, lane pos the new key, lanes > pos keep their own (shift up and down cancel)
  if (lane < pos) { s.A = dA; s.N = dN; s.X = dX; s.Y = dY; s.q = dq; }
  else if (lane == pos) { s.A = ko.A; s.N = hN; s.X = ko.X; s.Y = ko.Y; s.q = q; }
  s.count = cnt + 1;
}

// fairness.go:99-105 three times (alloc+req, alloc, req): lane (8*which + r) evaluates one float64 ratio, the max over a
// group of 8 lanes is the dominant share; identical IEEE operations to drf() in round_ctl.h, evaluated side by side.
// Operands come straight from LDS (the queue's resource vectors and the window record), one lane-indexed read each.
__device__ static inline void drf3(Dev& d, int q, int k, bool replay, double w, double* proposed, double* current, double* size) {
  int lane = threadIdx.x & 63;
  int which = lane >> 3, r = lane & 7;
  double x = -INFINITY;
  if (which < 3 && r < d.cfg.R) {
    int64_t a = (replay ? g_fl.qReplay[q][r] : g_fl.qAlloc[q][r]) + g_fl.qPenalty[q][r];
    int64_t rq = g_fl.winRec[q][k].req[r];
    int64_t v = which == 0 ? a + rq : which == 1 ? a : rq;
    int64_t t = d.cfg.totalResources[r];
    double f = 0.0;
    if (t != 0) f = (double)v / (double)t;
    x = f * d.cfg.drfMult[r];
  }
  {  // max over each group of 8 lanes, on every lane of the group (max is idempotent): lane^1, lane^2 inside the quad, then the mirrored quad
    unsigned long long b = __builtin_bit_cast(unsigned long long, x), t; double o;
    t = dppMove64<0xB1>(b, b); o = __builtin_bit_cast(double, t); x = o > x ? o : x; b = __builtin_bit_cast(unsigned long long, x);   // quad_perm [1,0,3,2]
    t = dppMove64<0x4E>(b, b); o = __builtin_bit_cast(double, t); x = o > x ? o : x; b = __builtin_bit_cast(unsigned long long, x);   // quad_perm [2,3,0,1]
    t = dppMove64<0x141>(b, b); o = __builtin_bit_cast(double, t); x = o > x ? o : x;                                                 // row_half_mirror: lane i <-> 7-i of its half row
  }
  double m = x > 0 ? x : 0.0;
  double res = which == 2 ? m * w : m / w;
  {  // lanes 0, 8, 16 hold the three results: scalar reads
    unsigned long long rb = __builtin_bit_cast(unsigned long long, res);
    auto rl = [&](int l) { unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rb, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rb >> 32), l); return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); };
    *proposed = rl(0); *current = rl(8); *size = rl(16);
  }
}

__device__ static inline void fastFence(Ctl& c) {
  if (c.l1Dirty) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); c.l1Dirty = 0; }  // vmcnt(0) + L1 invalidate: the no-return atomics are now what plain loads see
}

__device__ static inline void baseTileRemoved(KREF, FastS&, int) {}   // (the tile walk of ASCHED_FIT_BITS=0 keeps nothing between scans)
// base entry pos is stale from now on: its flag, and its bit in every fit shape's "clean and fits" bitmap (lane f clears row f; no-return atomics at L2 —
// the scans read the bitmaps with agent-scope loads, so every wave sees them)
__device__ static inline void baseMarkRemoved(KREF k, FastS& S, int pos) {
  (void)S;
  int lane = threadIdx.x & 63;
  if (lane == 0) k.baseRemoved[pos] = 1;
  if (k.fitBits) {
    unsigned long long bit = 1ull << (pos & 63);
    for (int f = lane; f < k.S; f += 64) (void)__hip_atomic_fetch_and(&k.fitBits[(size_t)f * k.fitW + (pos >> 6)], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ static inline void baseScan(KREF k, FastS& S, const JobTail& r) {
  int lane = threadIdx.x & 63;
#ifdef ASCHED_FASTPROF
  if (lane == 0) g_rs.statSeg[3] += 1000;   // profiling: scans
#endif
  int s = r.shape;
  int p0 = UNI32(g_fl.cand[s].pos);
  int N = k.N;
  if (k.fitBits) {
    // find-first-set over the shape's "clean and fits" bitmap: 64 lanes x 64 bits = 4096 base entries per memory round trip, whatever lies between
    // the cursor and the next usable entry (entries used up by other shapes, clean entries this shape does not fit on)
    S.statScanSteps++;
    if (UNI32(g_fl.cand[s].node) == -2 && UNI64(g_fl.cand[s].key) != 0) p0++;   // a stale candidate: the entry at the cursor is the one that was used up (its bit may still be on its way to L2)
    for (;;) {
      if (p0 >= N) { if (lane == 0) { g_fl.cand[s].pos = N; g_fl.cand[s].node = -1; } LANE0_PUBLISHED(); return; }
      int w0 = p0 >> 6, w = w0 + lane;
      unsigned long long word = w < k.fitW ? __hip_atomic_load(&k.fitBits[(size_t)s * k.fitW + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#ifdef ASCHED_FASTPROF
      if (lane == 0) g_rs.statSeg[4] += 1000;   // profiling: bitmap windows read
#endif
      if (lane == 0) word &= ~0ull << (p0 & 63);
      unsigned long long b = __ballot(word != 0);
      if (!b) { p0 = (w0 + 64) << 6; continue; }
      int L = __ffsll((long long)b) - 1;
      unsigned long long wv = slGet64(word, L);
      int q = ((w0 + L) << 6) + (__ffsll((long long)wv) - 1);
      // the entry itself: one more round trip, four independent loads
      unsigned long long key = k.baseKey[q], cls = k.baseCls[q]; int node = k.baseNode[q];
      long long ex0 = k.E > 0 ? k.baseExtra[q] : 0, ex1 = k.E > 1 ? k.baseExtra[k.Npad + q] : 0;
      if (lane == 0) { CandRec c; c.pos = q; c.node = node; c.key = key; c.cls = cls; c.ex0 = ex0; c.ex1 = ex1; c.pad = 0; g_fl.cand[s] = c; }
      LANE0_PUBLISHED();
      return;
    }
  }
  for (;;) {   // ASCHED_FIT_BITS=0: walk the base 64 entries at a time (keys, removed flags, extras, class bits are stored in base order: coalesced)
    if (p0 >= N) { g_fl.cand[s].pos = N; g_fl.cand[s].node = -1; return; }
#ifdef ASCHED_FASTPROF
    if (lane == 0) g_rs.statSeg[4] += 1000;   // profiling: tile loads
#endif
    int p = p0 + lane;
    unsigned long long tKey = 0, tCls = 0; int tNode = -1, tRem = 1; long long tEx0 = 0, tEx1 = 0;
    if (p < N) {
      tKey = k.baseKey[p]; tCls = k.baseCls[p]; tNode = k.baseNode[p]; tRem = k.baseRemoved[p];
      if (k.E > 0) tEx0 = k.baseExtra[p];
      if (k.E > 1) tEx1 = k.baseExtra[k.Npad + p];
    }
    bool ok = !tRem && entryFits(k, r, tKey, tEx0, tEx1, tCls);
    unsigned long long b = __ballot(ok);
    S.statScanSteps++;
    if (b) {
      int f = __ffsll((long long)b) - 1;
      CandRec c;
      c.pos = p0 + f; c.node = __shfl(tNode, f, 64); c.key = __shfl(tKey, f, 64); c.cls = __shfl(tCls, f, 64); c.ex0 = __shfl(tEx0, f, 64); c.ex1 = __shfl(tEx1, f, 64); c.pad = 0;
      g_fl.cand[s] = c;
      return;
    }
    p0 += 64;
  }
}

// mask mode: which fit shapes fit a node with these level-0 key fields / extras / class bits — lane l evaluates shapes l and l + 64 against the table in LDS
__device__ static inline void capMask2(KREF k, uint64_t clsBits, uint64_t key, int64_t ex0, int64_t ex1, uint64_t* m0, uint64_t* m1) {
  int lane = threadIdx.x & 63;
  bool ok0 = false, ok1 = false;
  if (lane < k.S) { const ShapeReq q = SHT(lane); ok0 = !q.never && ((clsBits >> q.cls) & 1) && fieldsGE(k, key, q.fieldMin) && q.ex0 <= ex0 && q.ex1 <= ex1; }
  if (lane + 64 < k.S) { const ShapeReq q = SHT(lane + 64); ok1 = !q.never && ((clsBits >> q.cls) & 1) && fieldsGE(k, key, q.fieldMin) && q.ex0 <= ex0 && q.ex1 <= ex1; }
  *m0 = __ballot(ok0); *m1 = __ballot(ok1);
}
__device__ static inline uint64_t l0Search(KREF k, const JobTail& r, int* slot) {
  int lane = threadIdx.x & 63;
  unsigned long long best = ~0ull; int bs = -1;
  int cnt = UNI32(g_fl.l0Count);
  int rounds = (cnt + 63) >> 6;   // the same trip count on every lane: the cross-lane reduction below sees a converged wave
  if (k.maskMode) {   // one bit per entry says whether the job's shape fits: key + mask, nothing else
    int sh = r.shape & 63; bool hi = r.shape >= 64;   // (wave-uniform: the mask word that holds the job's fit shape)
    for (int r0 = 0; r0 < rounds; r0 += 8) {   // up to 512 entries per group of loads: the usual list is searched with one LDS latency
      unsigned long long key[8], m[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { int i = ((r0 + u) << 6) + lane, j = i < cnt ? i : 0; key[u] = g_fl.l0Key[j]; m[u] = hi ? g_fl.l0Cls2[j] : g_fl.l0Cls[j]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { int i = ((r0 + u) << 6) + lane; if (i < cnt && ((m[u] >> sh) & 1) && key[u] < best) { best = key[u]; bs = i; } }
    }
    unsigned long long mn = waveMin64Dpp(best);
    if (mn == ~0ull) { *slot = -1; return mn; }
    unsigned long long who = __ballot(best == mn);
    *slot = __builtin_amdgcn_readlane(bs, __ffsll((long long)who) - 1);
    return mn;
  }
  // groups of 4, 2, 1 rounds: the loads of a group are issued together (the LDS latency is paid once per group) and only as many rounds as the list has are
  // loaded and tested — late in a round the list is short (140 entries on average on configs[2], 377 over the first quarter)
  auto group = [&](int r0, auto W) {
    constexpr int G = decltype(W)::value;
    unsigned long long key[G], cls[G]; long long e0[G], e1[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
      int i = ((r0 + u) << 6) + lane, j = i < cnt ? i : 0;
      key[u] = g_fl.l0Key[j]; e0[u] = g_fl.l0Ex0[j]; e1[u] = g_fl.l0Ex1[j]; cls[u] = g_fl.l0Cls[j];
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      int i = ((r0 + u) << 6) + lane;
      if (i < cnt && key[u] < best && entryFits(k, r, key[u], e0[u], e1[u], cls[u])) { best = key[u]; bs = i; }
    }
  };
  int r0 = 0;
  for (; r0 + 4 <= rounds; r0 += 4) group(r0, std::integral_constant<int, 4>{});
  if (r0 + 2 <= rounds) { group(r0, std::integral_constant<int, 2>{}); r0 += 2; }
  if (r0 < rounds) group(r0, std::integral_constant<int, 1>{});
  unsigned long long mn = waveMin64Dpp(best);   // keys are unique (node-index rank in the low bits): the lane that holds the minimum names the slot
  if (mn == ~0ull) { *slot = -1; return mn; }
  unsigned long long who = __ballot(best == mn);
  *slot = __builtin_amdgcn_readlane(bs, __ffsll((long long)who) - 1);
  return mn;
}

// WIN(=4) records x 16 lanes x 8 bytes: one coalesced 128-byte burst per job record
__device__ static inline void winRefill(KREF k, int q, int kind, int pos, int cnt) {
  int lane = threadIdx.x & 63;
  int i = lane >> 4, part = lane & 15;
  if (i < cnt) {
    int job = kind == 0 ? k.evList[pos + i] : k.queuedJobs[pos + i];
    int idx = kind == 0 ? k.evIdxByPos[pos + i] : -1;
    unsigned long long v = k.jrec[(size_t)job * (sizeof(JobRec) / 8) + part];
    ((unsigned long long*)&g_fl.winRec[q][i])[part] = v;
    if (part == 0) { g_fl.winJob[q][i] = job; g_fl.winIdx[q][i] = idx; }
  }
}
__device__ static inline void loadHeadRec(KREF k, int q, int job) {
  int lane = threadIdx.x & 63;
  if (lane < 16) {
    unsigned long long v = k.jrec[(size_t)job * (sizeof(JobRec) / 8) + lane];
    if (lane < 8) ((unsigned long long*)g_fl.headReq[q])[lane] = v; else ((unsigned long long*)&g_fl.headTail[q])[lane - 8] = v;
  }
}
__device__ static inline void headFromWindow(int q, int w) {
  int lane = threadIdx.x & 63;
  if (lane < 16) {
    unsigned long long v = ((const unsigned long long*)&g_fl.winRec[q][w])[lane];
    if (lane < 8) ((unsigned long long*)g_fl.headReq[q])[lane] = v; else ((unsigned long long*)&g_fl.headTail[q])[lane - 8] = v;
  }
}

// markAllocatable (node.go:539-549) for levels [lo, nl) as no-return HBM atomics, one (level, resource) per lane
__device__ static inline void bindUpdate(KREF k, FastS& S, int n, int lo, int nl, int q, uint64_t keyDelta) {
  int lane = threadIdx.x & 63;
  int l = lo + S.laneL;
  if (l < nl) {  // (nl - lo) * R <= 64 lanes whenever P * R <= 64; larger configurations take the second round below
    int64_t v = g_fl.headReq[q][S.laneX];
    if (v) __hip_atomic_fetch_add(&KAL(k, l, S.laneX, n), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int R = k.R;
  for (int i = lane + 64; i < (nl - lo) * R; i += 64) {
    int l2 = lo + i / R, x = i % R;
    int64_t v = g_fl.headReq[q][x];
    if (v) __hip_atomic_fetch_add(&KAL(k, l2, x, n), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane < nl - lo && keyDelta) __hip_atomic_fetch_add(&KKEY(k, lo + lane, n), 0ull - keyDelta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// per-field saturating subtraction on packed order keys: field 0 stands for every negative quotient, so a field that would drop below it stays there
__device__ static inline uint64_t keyFieldsSatSub(KREF k, uint64_t key, uint64_t delta) {
  uint64_t out = key;
#pragma unroll
  for (int c = 0; c < MAXK; c++) {
    if (c >= k.K) break;
    uint64_t m = k.fieldMask[c], f = key & m, dq = delta & m;
    out = (out & ~m) | (f > dq ? f - dq : 0);
  }
  return out;
}
__device__ static inline void keySatSub(KREF k, int n, int lo, int nl, uint64_t keyDelta) {
  int lane = threadIdx.x & 63;
  if (lane < nl - lo && keyDelta) {   // one level per lane; the bind wave / node engine may be adding to the same word: compare-and-swap
    GP(uint64_t) p = &KKEY(k, lo + lane, n);
    uint64_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (!__hip_atomic_compare_exchange_strong(p, &old, keyFieldsSatSub(k, old, keyDelta), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {}
  }
}
// sctx / qctx resource vectors for the head job of queue q: accumulate-only, lane x handles resource x;
// LDS vectors through ds_add_u64, HBM by-priority-class vectors through global atomics, all without a return value
__device__ static inline void accountVectors(Dev& d, KREF k, int q, int pc, bool ev, bool replay) {
  (void)d;
  int lane = threadIdx.x & 63;
  if (lane < k.R) {
    int64_t v = g_fl.headReq[q][lane];
    if (v) {
      if (replay) { LDS_ADD64(g_fl.qReplay[q][lane], v); return; }
      LDS_ADD64(g_fl.qAlloc[q][lane], v); LDS_ADD64(g_rs.allocated[lane], v);
      if (ev) LDS_ADD64(g_rs.evicted[lane], -v); else LDS_ADD64(g_rs.scheduled[lane], v);
      size_t i = ((size_t)q * k.npc + pc) * k.R + lane;
      __hip_atomic_fetch_add(&k.qAllocByPc[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ev) __hip_atomic_fetch_add(&k.qEvictedByPc[i], -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(&k.qSchedByPc[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// ---- two-wave iteration (round_fast.h): LDS mailbox between the control wave (0) and the node engine (wave 1).
// LDS executes one wave's accesses in issue order, so "payload, then sequence number" needs no hardware fence — only the
// compiler must keep the order (wavefront-scope fences emit nothing).  No s_waitcnt vmcnt anywhere on this path: neither wave
// ever waits for its own outstanding HBM atomics.
#define LDS_ORDER() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
// ---- bounded waits (round 6).  Every spin of the control workgroup's protocols (mailbox words, ring counters, the engine's acknowledgements) counts its turns; every
// 16 384 turns (~1 ms) it looks at the caller's cancel word (hard timeout / asched_cancel: a read across PCIe) and at the launch's `abandon` flag, and gives up when either is
// set — or when the wait has lasted ~2^26 turns (seconds: no wait of a healthy launch is that long), which raises ASCHED_ERR_DEVICE.  Giving up sets `abandon`, so the
// waves waiting on the other side give up as well and everybody meets at the end barrier of the engine session; the round then returns its error (the handle wants a
// fresh round_prepare, as after any failed round).  What this cannot bound is a workgroup BARRIER that a wave never reaches (profiles/r05y_bulk_skip_hang.txt): those
// are ruled out by construction (eng.live; the CPU build aborts on a wide op posted with the engine live).
#define SPIN_CHECK 0x3fffu
#define SPIN_LIMIT (1u << 26)
__device__ static inline bool waitExpired(unsigned& spins) {   // (inline, and streamIdle's watch too: out of line — tried for the gang rounds, which pay ~2.5 % for these waits — the callers stop being leaf functions and the headline round loses 7 ms)
  if ((++spins & SPIN_CHECK) != 0) return false;
  bool ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.abandon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0;
  if (!ab && (spins >= SPIN_LIMIT || cancelRequested(g_dev))) {
    if (spins >= SPIN_LIMIT) raise(g_dev, ASCHED_ERR_DEVICE, 950);
    __hip_atomic_store(&g_fl.eng.abandon, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&g_fl.eng.cancel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ab = true;
  }
  return ab;
}
// The node engine's waits (engine_hc.h) sit inside the per-job loop, whose instruction schedule is the headline (its region holds lane-divergent branches, so every branch
// in it costs exec-mask code: a counter with a test per wait cost 4-20 % of the round, profiles/r06z_bounded_waits.txt).  They carry no counter: a turn reads `abandon`
// in LDS, nothing else.  Setting it is the other waves' business: whenever the engine (or the cold wave, or the bind wave behind it) is stuck the control wave ends up in
// streamIdle (the ring is full, or drains) — which looks at the cancel word and keeps the tick budget for all of them — or in one of its own counted waits.
__device__ static inline bool waitAbandoned() { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.abandon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0; }
__device__ static inline bool waitGaveUp(unsigned& spins) {   // the cold wave's polls: `abandon` every 1 024 turns
  if ((++spins & 0x3ffu) != 0) return false;
  return waitAbandoned();
}
#define IDLE_BUDGET (1u << 22)   // in units of 1 024 shader-clock ticks: ~2 s without one entry placed or bound while the control wave does nothing but wait
__device__ static inline void streamIdleWatch(unsigned long long clk) {
  const int abV = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.abandon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  // the tick budget: entries placed + entries bound stand still over a stretch of CONTINUOUS waiting (a visit more than three periods after the last one starts a new stretch)
  const unsigned now = (unsigned)(clk >> 10) | 1u;
  const int prog = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringAck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  const unsigned since = (unsigned)__builtin_amdgcn_readfirstlane(g_fl.eng.idleSince), last = (unsigned)__builtin_amdgcn_readfirstlane(g_fl.eng.idleLast);
  const bool fresh = since == 0 || prog != __builtin_amdgcn_readfirstlane(g_fl.eng.idleProg) || now - last > (3u << 11);
  bool expired = !fresh && now - since > IDLE_BUDGET;
  if (fresh) { __hip_atomic_store(&g_fl.eng.idleProg, prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_store(&g_fl.eng.idleSince, (int)now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __hip_atomic_store(&g_fl.eng.idleLast, (int)now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (expired && abV == 0) raise(g_dev, ASCHED_ERR_DEVICE, 950);
  if (abV != 0 || expired || cancelRequested(g_dev)) {
    if (abV == 0) __hip_atomic_store(&g_fl.eng.abandon, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&g_fl.eng.cancel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringFail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) __hip_atomic_store(&g_fl.eng.ringFail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__device__ static inline void streamIdle() {   // the control wave while the ring is full / drains: its waits look at ringFail every turn, so giving up = a failure posted there
  const unsigned long long clk = __builtin_readcyclecounter();   // (issued BEFORE the sleep: its ~80 clocks pass while the wave sleeps; behind the sleep they delayed every wake-up — gang rounds +3.5 %)
  __builtin_amdgcn_s_sleep(2);
  // No counter in LDS or registers (the macro has no state; a 64-lane LDS add per turn took the LDS from the node engine): the shader clock says when to look — one window
  // of 2 048 ticks in every 2^21 (~1 ms); a turn of any of these waits is shorter than the window, so every period is seen at least once.
  if ((((unsigned)clk) & 0x1fffffu) < 0x800u) streamIdleWatch(clk);
}
__device__ static inline void bindUpdateEng(KREF k, FastS& S, int n, int nl, uint64_t keyDelta, const int64_t* req) {
  int lane = threadIdx.x & 63;
  int l = S.laneL;
  if (l < nl) {
    int64_t v = req[S.laneX];
    if (v) __hip_atomic_fetch_add(&KAL(k, l, S.laneX, n), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int R = k.R;
  for (int i = lane + 64; i < nl * R; i += 64) {
    int l2 = i / R, x = i % R;
    int64_t v = req[x];
    if (v) __hip_atomic_fetch_add(&KAL(k, l2, x, n), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane < nl && keyDelta) __hip_atomic_fetch_add(&KKEY(k, lane, n), 0ull - keyDelta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static inline void accountVectorsBk(Dev& d, KREF k, int q, int pc, int sign) {
  (void)d;
  int lane = threadIdx.x & 63;
  if (lane < k.R) {
    int64_t v = sign * g_fl.eng.req[lane];
    if (v) {
      LDS_ADD64(g_fl.qAlloc[q][lane], v); LDS_ADD64(g_rs.allocated[lane], v); LDS_ADD64(g_rs.scheduled[lane], v);
      size_t i = ((size_t)q * k.npc + pc) * k.R + lane;
      __hip_atomic_fetch_add(&k.qAllocByPc[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&k.qSchedByPc[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// One LDS pass, one word per lane: FL.bk := queue q's record and keys (what a rollback restores), FL.eng := the job (record, request,
// parameters); then the sequence number.  The job's record and request are not duplicated in the backup: the engine only reads them.
__device__ static inline void enginePost(Dev& d, KREF k, FastS& S, int job, int q, int pc, int32_t prio, int32_t cutoff, int nl) {
  (void)d; (void)k;
  int lane = threadIdx.x & 63;
  constexpr int HW = sizeof(QHot) / 8, TW = sizeof(JobTail) / 8, B = HW + TW + MAXR;
  static_assert(sizeof(QHot) % 8 == 0 && B + 9 <= 64, "backup + post fit one wave");
  if (lane < HW) ((unsigned long long*)&g_fl.bk.hot)[lane] = ((const unsigned long long*)&g_fl.hot[q])[lane];
  else if (lane < HW + TW) ((unsigned long long*)&g_fl.eng.tail)[lane - HW] = ((const unsigned long long*)&g_fl.headTail[q])[lane - HW];
  else if (lane < B) g_fl.eng.req[lane - HW - TW] = g_fl.headReq[q][lane - HW - TW];
  if (lane == 0) {
    g_fl.bk.kA = g_fl.kA[q]; g_fl.bk.kX = g_fl.kX[q]; g_fl.bk.kY = g_fl.kY[q];
    g_fl.bk.effA = g_fl.effA[q]; g_fl.bk.effX = g_fl.effX[q]; g_fl.bk.effY = g_fl.effY[q];
    g_fl.bk.inHeap = g_fl.inHeap[q]; g_fl.bk.globalTokens = S.globalTokens; g_fl.bk.pc = pc;
    g_fl.eng.job = job; g_fl.eng.prio = prio; g_fl.eng.cutoff = cutoff; g_fl.eng.nl = nl; g_fl.eng.cmd = ENG_JOB;
  }
  S.engSeq++;
  LDS_ORDER();
  if (lane == 0) __hip_atomic_store(&g_fl.eng.seq, S.engSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ static inline void engineRestore(int q) {
  int lane = threadIdx.x & 63;
  constexpr int HW = sizeof(QHot) / 8, TW = sizeof(JobTail) / 8, B = HW + TW + MAXR;
  if (lane < HW) ((unsigned long long*)&g_fl.hot[q])[lane] = ((const unsigned long long*)&g_fl.bk.hot)[lane];
  else if (lane < HW + TW) ((unsigned long long*)&g_fl.headTail[q])[lane - HW] = ((const unsigned long long*)&g_fl.eng.tail)[lane - HW];
  else if (lane < B) g_fl.headReq[q][lane - HW - TW] = g_fl.eng.req[lane - HW - TW];
  else if (lane == B) { g_fl.kA[q] = g_fl.bk.kA; g_fl.kX[q] = g_fl.bk.kX; g_fl.kY[q] = g_fl.bk.kY; }
  else if (lane == B + 1) { g_fl.effA[q] = g_fl.bk.effA; g_fl.effX[q] = g_fl.bk.effX; g_fl.effY[q] = g_fl.bk.effY; g_fl.inHeap[q] = g_fl.bk.inHeap; }
  LANE0_PUBLISHED();   // lanes B, B + 1 wrote fixed words the whole wave reads next (fastRollback's caller rebuilds the heap from them)
}
__device__ static inline int engineWait(const FastS& S) {
  int want = S.engSeq;
  unsigned spins = 0;
  for (;;) {
    int a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (a == want) break;
    __builtin_amdgcn_s_sleep(1);
    if (waitExpired(spins)) return 0;   // (as if the job had found no node: the caller takes the iteration back and meets the cancel flag)
  }
  LDS_ORDER();
  return __builtin_amdgcn_readfirstlane(g_fl.eng.status);
}

// ---- stream run (round_fast.h): ring between the control wave (merge + record staging) and the node engine
__device__ static inline void qsWinRefill(KREF k, int q, int pos, int cnt) {
  int lane = threadIdx.x & 63;
  if (lane < cnt * 4) ((unsigned long long*)&g_fl.evWin[q][0])[lane] = k.qsKey[((size_t)q * QS_CMAX + pos) * 4 + lane];
}
__device__ static inline void streamBegin(int* engSeq, int hold, int hc) {
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_fl.eng.bindHold = hold; g_fl.eng.ringPub = 0; g_fl.eng.ringAck = 0; g_fl.eng.ringEnd = 0; g_fl.eng.ringFail = 0; g_fl.eng.ringClosed = 0; g_fl.eng.bindDone = 0; g_fl.eng.idleSince = 0; g_fl.eng.cmd = hc ? ENG_STREAM_HC : ENG_STREAM; }
  (*engSeq)++;
  LDS_ORDER();
  if (lane == 0) __hip_atomic_store(&g_fl.eng.bindGen, g_fl.eng.bindGen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  LDS_ORDER();
  if (lane == 0) __hip_atomic_store(&g_fl.eng.seq, *engSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// records of ring entries [base, base + cnt), cnt <= 4: 16 lanes x 8 bytes each, one coalesced 128-byte burst per record; the value is consumed by
// streamStageCommit one group later, so the HBM latency runs under the merge
__device__ static inline unsigned long long streamStageIssue(KREF k, int base, int cnt) {
  int lane = threadIdx.x & 63;
  int i = lane >> 4, part = lane & 15;
  unsigned long long v = 0;
  if (i < cnt && !(RQ(base + i) & RQ_EV)) { int job = RJOB(base + i); v = k.jrec[(size_t)job * (sizeof(JobRec) / 8) + part]; }
  return v;
}
__device__ static inline void streamStageCommit(Dev& d, KREF k, int base, int cnt, unsigned long long v) {
  (void)d; (void)k;
  int lane = threadIdx.x & 63;
  int i = lane >> 4, part = lane & 15;
  if (i < cnt && !(RQ(base + i) & RQ_EV)) ((unsigned long long*)&RREC(base + i))[part] = v;
  LDS_ORDER();
  if (lane == 0) __hip_atomic_store(&g_fl.eng.ringPub, base + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ static inline void streamEnd(int engSeq) {
  int lane = threadIdx.x & 63;
  LDS_ORDER();
  if (lane == 0) __hip_atomic_store(&g_fl.eng.ringEnd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  for (;;) {   // the engine acknowledges the ENG_STREAM command when it has left the ring
    int a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (a == engSeq) break;
    __builtin_amdgcn_s_sleep(1);
    if (waitExpired(spins)) { LDS_ORDER(); return; }
  }
  if (__builtin_amdgcn_readfirstlane(g_fl.eng.bindHold)) { LDS_ORDER(); return; }   // a gang: the verdict comes first (streamRelease)
  int gen = __builtin_amdgcn_readfirstlane(g_fl.eng.bindGen);
  for (;;) {   // ... and the bind wave has issued (and released) the binds of every entry placed
    int f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindFin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (f == gen) break;
    __builtin_amdgcn_s_sleep(1);
    if (waitExpired(spins)) break;
  }
  LDS_ORDER();
}
__device__ static inline void streamRelease(Dev& d, KREF k, int go) {
  (void)d; (void)k;
  int lane = threadIdx.x & 63;
  if (lane == 0) __hip_atomic_store(&g_fl.eng.bindHold, go ? 2 : 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  int gen = __builtin_amdgcn_readfirstlane(g_fl.eng.bindGen);
  unsigned spins = 0;
  for (;;) {
    int f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindFin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (f == gen) break;
    __builtin_amdgcn_s_sleep(1);
    if (waitExpired(spins)) break;
  }
  LDS_ORDER();
}
__device__ static inline int streamBound() { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ static inline int streamAcked(int* fail) {
  int a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringAck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  int f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringFail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  // a failure flag read together with an older ack count: entries bound before the failing one are all counted when the flag is seen again after the ack
  if (f) a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringAck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  *fail = f;
  return a;
}
// sctx / qctx accounting of ring entries [i0, i1) once the engine has bound them (accountVectors for a new job): 4 (8 when R > 4) lanes per entry, one per
// resource, so a batch of acknowledgements costs one pass; FL.tmpQ[q] counts queue q's entries.  An entry of an evicted stream only counts: its
// commit is deferred like a cheap evicted head's (applyEvictedRange)
__device__ static inline void streamAccount(Dev& d, KREF k, int i0, int i1) {
  (void)d;
  int lane = threadIdx.x & 63;
  int sh = k.R <= 4 ? 2 : 3, per = 64 >> sh;
  int e = lane >> sh, x = lane & ((1 << sh) - 1);
  for (int b = i0; b < i1; b += per) {
    int i = b + e;
    if (i < i1) {
      int rq = RQ(i), q = rq & 0xff;
      if (x == 0) (void)__hip_atomic_fetch_add(&g_fl.tmpQ[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!(rq & RQ_EV) && x < k.R) {
        const JobRec& r = RREC(i);
        int64_t v = r.req[x];
        if (v) {
          LDS_ADD64(g_fl.qAlloc[q][x], v); LDS_ADD64(g_rs.allocated[x], v); LDS_ADD64(g_rs.scheduled[x], v);
          size_t j = ((size_t)q * k.npc + r.pc) * k.R + x;
          __hip_atomic_fetch_add(&k.qAllocByPc[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&k.qSchedByPc[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}
// key and name rank of the heap's head entry (queue t): lane 0 of the heap lanes, no LDS access
__device__ static inline void pqHeadKey(PQState& s, int t, PackedKey* key, uint32_t* nameRank) {
  if (__builtin_amdgcn_readfirstlane(s.q) == t) {
    key->A = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.A); *nameRank = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.N);
    key->X = UNI64(s.X); key->Y = UNI64(s.Y);
  } else {
    key->A = UNI32(g_fl.kA[t]); key->X = UNI64(g_fl.kX[t]); key->Y = UNI64(g_fl.kY[t]); *nameRank = (uint32_t)UNI32(g_fl.nameRank[t]);
  }
}
// the engine session is one mailbox op of the control workgroup: wave 1 serves jobs until ENG_QUIT, waves 2.. wait at the end barrier
__device__ static inline void engineStart(Dev& d, FastS& S) {
  (void)d;
  S.engSeq = 0;
  if ((threadIdx.x & 63) == 0) { g_fl.eng.seq = 0; g_fl.eng.ack = 0; g_fl.eng.statScan = 0; g_fl.eng.statL0Max = S.statL0Max; g_fl.eng.busyClk = 0; g_fl.eng.jobs = 0; g_fl.eng.cancel = 0; g_fl.eng.bindQuit = 0; g_fl.eng.bindGen = 0; g_fl.eng.bindFin = 0; g_fl.eng.hcGen = 0; g_fl.eng.idleSince = 0; g_fl.eng.live = 1; g_mb.op = OP_ENGINE; }
  __syncthreads();
}
__device__ static inline void engineStop(Dev& d, FastS& S) {
  (void)d;
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_fl.eng.cmd = ENG_QUIT; __hip_atomic_store(&g_fl.eng.bindQuit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  LDS_ORDER();
  S.engSeq++;
  if (lane == 0) __hip_atomic_store(&g_fl.eng.seq, S.engSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)engineWait(S);
  if (lane == 0) g_fl.eng.live = 0;
  LANE0_PUBLISHED();
  S.statScanSteps += __builtin_amdgcn_readfirstlane(g_fl.eng.statScan);
  int m = __builtin_amdgcn_readfirstlane(g_fl.eng.statL0Max);
  if (m > S.statL0Max) S.statL0Max = m;
#ifndef ASCHED_FASTPROF
  if (lane == 0) { g_rs.statSeg[1] += g_fl.eng.busyClk; g_rs.statSeg[2] += g_fl.eng.jobs; }
  LANE0_PUBLISHED();
#endif
  __syncthreads();  // end barrier of the OP_ENGINE op
}
#include "engine_hc.h"
__device__ static void engineLoop(Dev& d) {  // wave 1
  const FastK k = fastKRef(d);
  int lane = threadIdx.x & 63;
  FastS ES;
   ES.engLive = 0; ES.engPend = -1;
  ES.laneL = lane / (k.R > 0 ? k.R : 1); ES.laneX = lane % (k.R > 0 ? k.R : 1);
  ES.statScanSteps = 0; ES.statL0Max = __builtin_amdgcn_readfirstlane(g_fl.eng.statL0Max);
  ES.fastActive = 1; ES.engSeq = 0;
  long long busy = 0; int jobs = 0;
  int seen = 0;
#ifdef ASCHED_FASTPROF
  for (int i = 0; i < 8; i++) ES.eseg[i] = 0;
  ES.segT = CLK();
#endif
  unsigned spins = 0;
  for (;;) {
    int sq;
    for (;;) {
      sq = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (sq != seen) break;
      __builtin_amdgcn_s_sleep(1);
      if (waitExpired(spins)) return;   // (to the end barrier of the engine session: the control wave, whose waits give up too, comes there through engineStop)
    }
    spins = 0;
    seen = sq;
    LDS_ORDER();
    int cmd = __builtin_amdgcn_readfirstlane(g_fl.eng.cmd);
    if (cmd == ENG_QUIT) {
#ifdef ASCHED_FASTPROF
      if (lane == 0) for (int i = 0; i < 8; i++) g_rs.statSeg[16 + i] += ES.eseg[i];   // [16] waiting for a job, [17] record, [18] first fit, [19] bind, [20] result fields, [21] L0 upkeep, [22] verdict
#endif
      if (lane == 0) { g_fl.eng.statScan = ES.statScanSteps; g_fl.eng.statL0Max = ES.statL0Max; g_fl.eng.busyClk = busy; g_fl.eng.jobs = jobs; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this wave's binds and result stores are complete before the generic code reads them
      LDS_ORDER();
      if (lane == 0) __hip_atomic_store(&g_fl.eng.ack, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
    if (cmd == ENG_STREAM_HC) {   // a bulk-merged run's ring session on the split level-0 structure (engine_hc.h); same ring contract as below
      long long b0 = (long long)__builtin_readcyclecounter();
      engineStreamHc(d, k, ES);
      busy += (long long)__builtin_readcyclecounter() - b0; jobs += __builtin_amdgcn_readfirstlane(g_fl.eng.ringAck);
      LDS_ORDER();
      if (lane == 0) __hip_atomic_store(&g_fl.eng.ringClosed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0) __hip_atomic_store(&g_fl.eng.ack, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
    if (cmd == ENG_STREAM) {
      // walk the ring: entry i is ready when ringPub > i; stop at the first job that finds no node (ringFail 1), after an L0 overflow (2), or when the
      // control wave has closed the ring and everything staged is bound
      int i = 0, pub = 0;
      for (;;) {
        while (pub <= i) {   // (the counter is read again only when the entries known to be staged are used up)
          pub = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringPub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (pub > i) break;
          int end = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringEnd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (end) { pub = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringPub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); break; }
          __builtin_amdgcn_s_sleep(1);
          if (waitExpired(spins)) return;
        }
        if (pub <= i) break;
        LDS_ORDER();
        if (__builtin_amdgcn_readfirstlane(RQ(i)) & RQ_EV) {   // an evicted job returning to its node: nothing to select or bind here
          i++;
          if (lane == 0) __hip_atomic_store(&g_fl.eng.ringAck, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          continue;
        }
        ESEG(0);
#ifdef ASCHED_FASTPROF
        long long b0 = (long long)__builtin_readcyclecounter();
#endif
        int st = engineServeRing(d, k, ES, i);
#ifdef ASCHED_FASTPROF
        busy += (long long)__builtin_readcyclecounter() - b0;   // (the clock reads sit on the engine's chain: profiling builds only)
#endif
        jobs++;
        if (st == 0) { if (lane == 0) __hip_atomic_store(&g_fl.eng.ringFail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
        i++;
        if (lane == 0) __hip_atomic_store(&g_fl.eng.ringAck, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (st == 2) { LDS_ORDER(); if (lane == 0) __hip_atomic_store(&g_fl.eng.ringFail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
        ESEG(6);
      }
      LDS_ORDER();
      if (lane == 0) __hip_atomic_store(&g_fl.eng.ringClosed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0) __hip_atomic_store(&g_fl.eng.ack, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
    ESEG(0);
    long long b0 = (long long)__builtin_readcyclecounter();
    int st = engineServe(d, k, ES);
    busy += (long long)__builtin_readcyclecounter() - b0; jobs++;
    if (lane == 0) g_fl.eng.status = st;
    LDS_ORDER();
    if (lane == 0) __hip_atomic_store(&g_fl.eng.ack, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ESEG(6);
  }
}

// Wave 2 during an engine session: the HBM side of a stream run's placements.  The node engine decides (first fit, L0 upkeep: LDS) and leaves the node in
// the ring entry; this wave follows its acknowledgement counter and issues BindJobToNode's plane / key atomics and the job's result fields.  Nothing on the
// engine's chain reads what is written here (level-0 state lives in the base flags + L0), so the two run concurrently; the release fence at the end of a
// stream orders the writes before whatever the control wave does next.
__device__ static void bindLoop(Dev& d) {
  const FastK k = fastKRef(d);
  int lane = threadIdx.x & 63;
  FastS BS;
   BS.laneL = lane / (k.R > 0 ? k.R : 1); BS.laneX = lane % (k.R > 0 ? k.R : 1);
#ifdef ASCHED_FASTPROF
  for (int i = 0; i < 8; i++) BS.eseg[i] = 0;
  BS.segT = 0;
#endif
  int gen = 0;
  unsigned spins = 0;
  for (;;) {
    for (;;) {
      int g = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindGen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (g != gen) { gen = g; break; }
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindQuit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return;
      __builtin_amdgcn_s_sleep(2);
      if (waitExpired(spins)) return;
    }
    spins = 0;
    int i = 0;
    bool discard = false;
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindHold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {   // a gang: all members or none
      int hmode;
      for (;;) {
        hmode = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.bindHold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (hmode != 1) break;
        __builtin_amdgcn_s_sleep(1);
        if (waitExpired(spins)) return;
      }
      discard = hmode == 3;
    }
    for (; !discard;) {
      int ack = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringAck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (i >= ack) {
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
          ack = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&g_fl.eng.ringAck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (i >= ack) break;
        } else { __builtin_amdgcn_s_sleep(1); if (waitExpired(spins)) return; continue; }
      }
      LDS_ORDER();
      for (; i < ack; i++) {
        if (__builtin_amdgcn_readfirstlane(RQ(i)) & RQ_EV) continue;
        const JobRec& r = RREC(i);
        int n = __builtin_amdgcn_readfirstlane(r.node0), job = __builtin_amdgcn_readfirstlane(RJOB(i));
        int32_t p = __builtin_amdgcn_readfirstlane(r.pcPrio);
        int32_t cutoff = __builtin_amdgcn_readfirstlane((int)r.preemptible) ? p : NONPREEMPTIBLE_CUTOFF;
        FastS& ES = BS;
        bindJob(k, ES, n, __builtin_amdgcn_readfirstlane((int)r.nlPc), UNI64(r.keyDelta), r.req, job, p, cutoff);
      }
      LDS_ORDER();
      if (lane == 0) __hip_atomic_store(&g_fl.eng.bindDone, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    LDS_ORDER();
    if (lane == 0) __hip_atomic_store(&g_fl.eng.bindFin, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// WIN EvKey records (32 B each) of queue q's evicted stream: 4 lanes x 8 bytes per record
__device__ static inline void evWinRefill(KREF k, int q, int pos, int cnt) {
  int lane = threadIdx.x & 63;
  if (lane < cnt * 4) ((unsigned long long*)&g_fl.evWin[q][0])[lane] = k.evKey[(size_t)pos * 4 + lane];
}
// Deferred commits of evicted jobs [p0, p1) of queue q returning to their nodes, one job per lane: the evicted branch of
// fastIter's commit (node.go:416-442 arithmetic, sctx/qctx accounting) as no-return atomics and plain stores.  Node planes are
// hit at distinct addresses; the per-queue / per-priority-class sums are accumulated per lane and reduced across the wave once,
// so that 64 lanes do not serialise on one counter.  Not inlined: a cold, register-hungry path next to the hot loop.
__device__ static inline int64_t waveSum64(int64_t v) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
#define APPLY_PCS 4
__device__ static __attribute__((noinline)) void applyEvictedRange(Dev& d, int q, int p0, int p1, int sign) {
  const FastK k = fastKRef(d);
  int lane = threadIdx.x & 63;
  int R = k.R;
  int pending = g_rs.replayPending;
  if (sign < 0 && !pending && lane == 0) g_rs.ftValid = 0;
  LANE0_PUBLISHED();   // evicted-table entries come back: thresholds may rise (round_ft.h "Staleness") — the table is rebuilt at the next query
  int64_t accQ[MAXR], accPc[APPLY_PCS][MAXR];
#pragma unroll
  for (int x = 0; x < MAXR; x++) { accQ[x] = 0;
#pragma unroll
    for (int c = 0; c < APPLY_PCS; c++) accPc[c][x] = 0; }
  for (int p = p0 + lane; p < p1; p += 64) {
    int job = k.evList[p];
    GP(unsigned long long) rec = k.jrec + (size_t)job * (sizeof(JobRec) / 8);
    unsigned long long keyDelta = rec[8];
    unsigned long long w10 = rec[10], w11 = rec[11], w12 = rec[12], w13 = rec[13];
    int pcx = (int)(unsigned)w10, n = (int)(unsigned)(w11 >> 32), prio = (int)(unsigned)w12;
    unsigned flags = (unsigned)(w13 >> 32);
    int preemptible = (flags >> 8) & 255, nlRun = (flags >> 24) & 255;
    int32_t cutoff = preemptible ? prio : NONPREEMPTIBLE_CUTOFF;
#pragma unroll
    for (int x = 0; x < MAXR; x++) {
      if (x >= R) break;
      int64_t v = sign * (int64_t)rec[x];
      if (!v) continue;
      accQ[x] += v;
      if (pcx < APPLY_PCS) {
#pragma unroll
        for (int c = 0; c < APPLY_PCS; c++) if (c == pcx) accPc[c][x] += v;
      } else {
        size_t i = ((size_t)q * k.npc + pcx) * R + x;
        __hip_atomic_fetch_add(&k.qAllocByPc[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&k.qEvictedByPc[i], -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int l = 1; l < nlRun; l++) __hip_atomic_fetch_add(&KAL(k, l, x, n), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (keyDelta) for (int l = 1; l < nlRun; l++) __hip_atomic_fetch_add(&KKEY(k, l, n), sign > 0 ? 0ull - keyDelta : keyDelta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sign < 0) {  // taken back: the job is evicted again, exactly as the evictor left it (eviction.go:245-260, evictApply)
      k.jcHasPctx[job] = 0; k.pcNode[job] = -1; k.pcSap[job] = 0; k.pcPap[job] = ASCHED_MIN_PRIORITY; k.pcMethod[job] = ASCHED_METHOD_NONE;
      k.jobEvictedOnNode[job] = 1; k.jobFlags[job] = F_EVICTED; k.inPreempted[job] = 1;
      if (!pending) { int idx = k.evIdxByPos[p]; k.evTabAlive[idx] = 1; k.evIndexOfJob[job] = idx; g_rs.fairIndexValid = 0; }   // (an entry comes back: ensureFairIndex)
      continue;
    }
    k.jcReason[job] = 0; k.jcHasPctx[job] = 1; k.pcNode[job] = n; k.pcSap[job] = prio;
    k.jobNode[job] = n; k.jobCutoff[job] = cutoff; k.jobEvictedOnNode[job] = 0; k.schedAtPrio[job] = prio; k.inSchedAndEvicted[job] = 0;
    k.pcPap[job] = prio; k.pcMethod[job] = ASCHED_METHOD_RESCHEDULED; k.jobFlags[job] = F_RESCHEDULED; k.inPreempted[job] = 0;
    if (!pending) { k.evTabAlive[k.evIdxByPos[p]] = 0; k.evIndexOfJob[job] = -1; }
  }
#pragma unroll
  for (int x = 0; x < MAXR; x++) {
    if (x >= R) break;
    int64_t v = waveSum64(accQ[x]);
    if (x == 0) LANE0_PUBLISHED();   // (the loop above: g_rs.fairIndexValid = 0 under a lane-divergent condition)
    if (lane == 0 && v) { LDS_ADD64(g_fl.qAlloc[q][x], v); LDS_ADD64(g_rs.allocated[x], v); LDS_ADD64(g_rs.evicted[x], -v); }
#pragma unroll
    for (int c = 0; c < APPLY_PCS; c++) {
      if (c >= k.npc) break;
      int64_t w = waveSum64(accPc[c][x]);
      if (lane == 0 && w) {
        size_t i = ((size_t)q * k.npc + c) * R + x;
        __hip_atomic_fetch_add(&k.qAllocByPc[i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&k.qEvictedByPc[i], -w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__device__ static inline bool roundLimitExceeded(Dev& d, KREF k) {
  int lane = threadIdx.x & 63;
  bool ex = lane < k.R && g_rs.scheduled[lane] > d.cfg.maxToSchedule[lane];
  return __ballot(ex) != 0;  // ballot results are scalar
}
__device__ static inline bool headRequestsDisallowed(Dev& d, KREF k, int q) {
  int lane = threadIdx.x & 63;
  bool bad = lane < k.R && d.cfg.disallowed[lane] && g_fl.headReq[q][lane] > 0;
  return __ballot(bad) != 0;
}

// pinned-node check of a returning evicted job against the node's current allocatable (nodedb.go:897-906): one lane per resource; the planes are
// updated by no-return atomics that execute at L2, so the reads go there too (agent-scope atomic loads, never the L1)
__device__ static inline bool pinnedNodeFits(KREF k, int q, int n, int level) {
  int lane = threadIdx.x & 63;
  if (UNI32((int)k.nodeFlags[n]) & 1) return true;
  bool bad = false;
  if (lane < k.R) {
    int64_t have = __hip_atomic_load(&KAL(k, level, lane, n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bad = g_fl.headReq[q][lane] > have;
  }
  return __ballot(bad) == 0;
}

__device__ static inline void exclPinnedFast(Dev& d, KREF k, int q, int job, int n, int level) {
  int lane = threadIdx.x & 63;
  int64_t have = 0;
  if (lane < k.R) have = __hip_atomic_load(&KAL(k, level, lane, n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool bad = lane < k.R && g_fl.headReq[q][lane] > have;
  unsigned long long m = __ballot(bad);
  if (!m) { if (lane == 0) d.excl[job] = EXCL_S_UNSUPPORTED; return; }
  int res = __builtin_ctzll(m);
  if (lane == res) { EXCL(d)->pinAvail[job] = have; d.excl[job] = EXCL_S_PINNED0 - res; }
}
__device__ static inline EvDyn evDynLoad(KREF k, int q, int job, int n, int level, bool wantMark, bool wantPin) {
  int lane = threadIdx.x & 63;
  // three independent loads, issued back to back; the first use below waits for all of them once
  int mark = wantMark ? (int)k.jcPreempted[job] : 0;
  int nf = wantPin ? (int)k.nodeFlags[n] : 0;
  int64_t have = 0;
  if (wantPin && lane < k.R) have = __hip_atomic_load(&KAL(k, level, lane, n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  EvDyn r;
  r.preempted = UNI32(mark);
  bool bad = wantPin && lane < k.R && g_fl.headReq[q][lane] > have;
  r.fits = (!wantPin || (UNI32(nf) & 1) || __ballot(bad) == 0) ? 1 : 0;
  return r;
}

__device__ static inline EvDyn evCleanLoad(KREF k, int job, int n, bool wantMark, bool wantClean) {
  int lane = threadIdx.x & 63;
  int mark = wantMark ? (int)k.jcPreempted[job] : 0;   // independent loads, issued back to back
  int64_t have = 0;
  if (wantClean && lane < k.R) have = __hip_atomic_load(&KAL(k, 0, lane, n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  EvDyn r;
  r.preempted = UNI32(mark);
  r.fits = __ballot(wantClean && lane < k.R && have < 0) == 0 ? 1 : 0;
  return r;
}
__device__ static inline unsigned long long evPendingMask(int Q) {
  int q = threadIdx.x & 63;
  return __ballot(q < Q && g_fl.hot[q].evApplied < g_fl.hot[q].evDone);
}

// ------------------------------------------------------------------------------------------------ LDS residency of the round's small state
// Every per-queue array and the scheduling-context scalars are moved into LDS for the duration of the launch by
// re-pointing the Dev descriptor (which itself lives in LDS): generic and fast code alike then pay LDS latency for them.
#define ARENA_BYTES (16 * 1024)
__shared__ unsigned long long g_arena[ARENA_BYTES / 8];
struct Reloc { void** pp; void* global; int bytes; };
#define MAX_RELOC 48
__shared__ Reloc g_reloc[MAX_RELOC];
__shared__ int g_nreloc;
__shared__ RoundScalars* g_rsGlobal;

__device__ static void relocateIn(Dev& d, int cmd) {
  // executed by every thread; the table is built by thread 0
  if (threadIdx.x == 0) {
    g_nreloc = 0;
    g_rsGlobal = d.rs;
    int Q = d.cfg.Q, R = d.cfg.R, npc = d.cfg.npc;
#if defined(ASCHED_AUX_TU) || defined(ASCHED_WK_TU)
    const bool marketCmd = cmd == CMD_MARKET_ROUND || cmd == CMD_MARKET_QUEUES;   // (the auxiliary kernel's rounds; the round kernel's code does not see this)
#else
    const bool marketCmd = false;
#endif
    bool want = (cmd == CMD_ROUND || cmd == CMD_QUEUES_ONLY || cmd == CMD_PASS1 || cmd == CMD_PASS2 || marketCmd) && Q > 0 && d.qWeight != nullptr && (Q <= QCAPF || d.f.relocAll);
    if (want) {
      int off = 0, n = 0; bool fits = true;
      auto add = [&](void** pp, int bytes) {
        if (!*pp || !fits) return;
        int b = (bytes + 7) & ~7;
        if (off + b > ARENA_BYTES || n >= MAX_RELOC) { fits = false; return; }
        g_reloc[n].pp = pp; g_reloc[n].global = *pp; g_reloc[n].bytes = bytes; n++; off += b;
      };
      int q1 = Q + 1;
      add((void**)&d.qWeight, Q * 8); add((void**)&d.qNameRank, Q * 4); add((void**)&d.qTokens, Q * 8); add((void**)&d.qBurst, Q * 8);
      add((void**)&d.qRateInf, Q); add((void**)&d.qCordoned, Q); add((void**)&d.qAlloc, Q * R * 8); add((void**)&d.qPenalty, Q * R * 8);
      // qAllocByPc / qSchedByPc / qEvictedByPc stay in HBM: the fast path accumulates into them with global atomics
      add((void**)&d.queuedOff, q1 * 4); add((void**)&d.evOff, (Q + 2) * 4);
      add((void**)&d.itEi, q1 * 4); add((void**)&d.itQi, q1 * 4); add((void**)&d.itStage, q1 * 4); add((void**)&d.itJobsSeen, q1 * 4);
      add((void**)&d.itNext, q1 * 4); add((void**)&d.itStashed, q1 * 4);
      add((void**)&d.itJobOnlyEv, q1); add((void**)&d.itGangOnlyEv, q1); add((void**)&d.onlyEvByQueue, q1); add((void**)&d.qEvictable, q1);
      add((void**)&d.pqProposed, q1 * 8); add((void**)&d.pqCurrent, q1 * 8); add((void**)&d.pqBudget, q1 * 8); add((void**)&d.pqSize, q1 * 8);
      add((void**)&d.pqPcPrio, q1 * 4); add((void**)&d.pqSchedPrio, q1 * 4); add((void**)&d.pqGctx, q1 * 4); add((void**)&d.pqInHeap, q1);
      add((void**)&d.replayAlloc, q1 * R * 8);
#if defined(ASCHED_AUX_TU) || defined(ASCHED_WK_TU)
      if (marketCmd && g_mk.s) {   // MarketIteratorPQ's items and heap, the merge iterators' held values, the round's market scalars (round_mkt.h)
        add((void**)&g_mk.s, (int)sizeof(MktScalars));
        add((void**)&g_mk.heap, q1 * 4); add((void**)&g_mk.pqPrice, q1 * 8); add((void**)&g_mk.pqRuntime, q1 * 8); add((void**)&g_mk.pqSubmit, q1 * 8); add((void**)&g_mk.pqQueued, q1);
        add((void**)&g_mk.itV1, q1 * 4); add((void**)&g_mk.itV2, q1 * 4);
        add((void**)&g_mk.qBillable, Q * R * 8); add((void**)&g_mk.qOverride, Q * 8); add((void**)&g_mk.qHasOverride, Q);
      }
#endif
      g_nreloc = fits ? n : 0;
    }
  }
  __syncthreads();
  // scalars
  {
    const int* src = (const int*)g_rsGlobal; int* dst = (int*)&g_rs;
    for (int i = threadIdx.x; i < (int)(sizeof(RoundScalars) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
  }
  int off = 0;
  for (int k = 0; k < g_nreloc; k++) {
    const unsigned char* src = (const unsigned char*)g_reloc[k].global; unsigned char* dst = (unsigned char*)g_arena + off;
    for (int i = threadIdx.x; i < g_reloc[k].bytes; i += blockDim.x) dst[i] = src[i];
    off += (g_reloc[k].bytes + 7) & ~7;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int o = 0;
    for (int k = 0; k < g_nreloc; k++) { *g_reloc[k].pp = (unsigned char*)g_arena + o; o += (g_reloc[k].bytes + 7) & ~7; }
    d.rs = &g_rs;
  }
  __syncthreads();
}
__device__ static void relocateOut() {
  __syncthreads();
  {
    int* dst = (int*)g_rsGlobal; const int* src = (const int*)&g_rs;
    for (int i = threadIdx.x; i < (int)(sizeof(RoundScalars) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
  }
  int off = 0;
  for (int k = 0; k < g_nreloc; k++) {
    unsigned char* dst = (unsigned char*)g_reloc[k].global; const unsigned char* src = (const unsigned char*)g_arena + off;
    for (int i = threadIdx.x; i < g_reloc[k].bytes; i += blockDim.x) dst[i] = src[i];
    off += (g_reloc[k].bytes + 7) & ~7;
  }
  __threadfence();
}

// ------------------------------------------------------------------------------------------------ kernels
template <class A> __device__ static inline A helpArgs(HelpBox* b, int at = 0) {
  A a;
  ull_alias* w = (ull_alias*)&a;
  for (int i = 0; i < (int)(sizeof(A) / 8); i++) w[i] = __hip_atomic_load(&b->args[at + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return a;
}
// Helper workgroup.  Wave 0 polls the command word in HBM (backing off to ~30 us between polls when the round has not asked
// for anything for a while, so an idle helper costs no measurable fabric traffic) and republishes it in LDS; the other waves
// poll that LDS word.  Each wave takes its share of the nodes, folds its result into an LDS word, and the wave that
// arrives last sends the workgroup's result and ONE completion increment to HBM.  Every loop in here is wave-uniform and
// there is no workgroup barrier inside the loop on purpose: a "thread 0 polls, the others wait at the barrier" loop gets
// rotated by the compiler so that the polling lane's tail and head merge across the back edge, and the rest of its wave
// then runs ahead through the barriers without it.
__shared__ unsigned long long g_hCmd, g_hMin, g_hMax;
__shared__ unsigned int g_hArrived;
__device__ static inline unsigned long long waveUniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ static void helperMain(const Dev& d, HelpBox* b, int H) {
  if (threadIdx.x == 0) { g_hCmd = 0; g_hMin = ~0ull; g_hMax = 0; g_hArrived = 0; }
  __syncthreads();
  unsigned long long seen = 0;
  int tid = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x, nthreads = (H + 1) * (int)blockDim.x;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  for (;;) {
    unsigned long long g;
    if (wave == 0) {
      unsigned int idle = 0;
      for (;;) {
        g = waveUniform64(__hip_atomic_load(&b->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (g != seen) break;
        idle++;
        if (idle < 2048) __builtin_amdgcn_s_sleep(2);
        else for (int k = 0; k < 8; k++) __builtin_amdgcn_s_sleep(127);
      }
      if (lane == 0) __hip_atomic_store(&g_hCmd, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      for (;;) {
        g = waveUniform64(__hip_atomic_load(&g_hCmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (g != seen) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the round state written before the command was published
    seen = g;
    unsigned int op = (unsigned int)(seen & 255);
    if (d.progress && threadIdx.x == 0 && blockIdx.x < 40) d.progress[16 + blockIdx.x] = (int)((seen >> 8) * 16 + op);
    if (op == OP_HELPERS_EXIT) return;
    if (op == OP_SCAN) {
      ScanArgs a = helpArgs<ScanArgs>(b);
      unsigned long long v = scanPart(d, a, tid, nthreads);
      if (lane == 0 && v != ~0ull) __hip_atomic_fetch_min(&g_hMin, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (op == OP_FAIR) {
      FairArgs a = helpArgs<FairArgs>(b);
      int v = fairPart(d, a, tid, nthreads);
      if (lane == 0 && v >= 0) __hip_atomic_fetch_max(&g_hMax, (unsigned long long)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (op == OP_SCANFAIR) {
#ifdef HELP_TRACE
      const int tcls = (int)blockIdx.x == 1 ? 1 : ((int)blockIdx.x == (H + 1) / 2 ? 2 : ((int)blockIdx.x == H ? 3 : 0));
      const int tgen = (int)(seen >> 8);
      if (tcls && threadIdx.x == 0) { TRACE_ADD(tcls, 0, tgen); atomicAdd(&g_traceSum[56 + tcls], 1ull); }
#endif
      ScanArgs a = helpArgs<ScanArgs>(b);
      FairArgs f = helpArgs<FairArgs>(b, HELP_ARGS2);
#ifdef HELP_TRACE
      if (tcls && threadIdx.x == 0) TRACE_ADD(tcls, 1, tgen);
#endif
      unsigned long long v = scanPart(d, a, tid, nthreads);
#ifdef HELP_TRACE
      if (tcls && threadIdx.x == 0) TRACE_ADD(tcls, 2, tgen);
#endif
      int w = fairPart(d, f, tid, nthreads);
#ifdef HELP_TRACE
      if (tcls && threadIdx.x == 0) TRACE_ADD(tcls, 3, tgen);
#endif
      if (lane == 0 && v != ~0ull) __hip_atomic_fetch_min(&g_hMin, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0 && w >= 0) __hip_atomic_fetch_max(&g_hMax, (unsigned long long)(w + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (op == OP_BULKW) {
      BulkWArgs a = helpArgs<BulkWArgs>(b);
      Dev& dm = const_cast<Dev&>(d);
      for (int i = tid; i < a.n; i += nthreads) bulkElem(dm, a.kind, i);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this workgroup's writes before its completion count
    }
    else if (op == OP_WIDE) {
      BulkWArgs a = helpArgs<BulkWArgs>(b);
      Dev& dm = const_cast<Dev&>(d);
      for (int i = tid; i < a.n; i += nthreads) wideBulkAny(dm, a.kind, i);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
#ifndef ASCHED_NO_FT
    else if (op == OP_FTBUILD) {
      BulkWArgs a = helpArgs<BulkWArgs>(b);
      Dev& dm = const_cast<Dev&>(d);
      for (int i = tid; i < a.n; i += nthreads) ftBuildAny(dm, a.kind, i);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
#endif
    if (lane == 0) {
      unsigned int before = __hip_atomic_fetch_add(&g_hArrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (before == (unsigned)(nw - 1)) {  // last wave of the workgroup: forward the folded result, reset the LDS words for the next command
        unsigned long long mn = __hip_atomic_load(&g_hMin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned long long mx = __hip_atomic_load(&g_hMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&g_hMin, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&g_hMax, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&g_hArrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        HelpSlot* sl = &b->slot[blockIdx.x - 1];
        __hip_atomic_store(&sl->mn, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sl->mx, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sl->gen, seen >> 8, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // completion: this workgroup's results (and, for the bulk ops, its writes) are visible before it
#ifdef HELP_TRACE
        if (op == OP_SCANFAIR) { const int tc2 = (int)blockIdx.x == 1 ? 1 : ((int)blockIdx.x == (H + 1) / 2 ? 2 : ((int)blockIdx.x == H ? 3 : 0)); if (tc2) TRACE_ADD(tc2, 4, (int)(seen >> 8)); }
#endif
      }
    }
  }
}

#ifndef ASCHED_AUX_TU
// armada_sched_ft.hip compiles this file once more with ASCHED_FT_TU + ASCHED_WITH_FT: the same round kernel WITH the fair-share threshold table (round_ft.h) under the name
// k_control_ft, in a code object of its own — the table's call sites cost the default round kernel 2-3 % by code placement alone (profiles/r03f_*), so it is not in k_control;
// the host launches k_control_ft for rounds whose handle carries a table (asched_host.inc ensureFt: crowded pools of >= 50 000 nodes, from the second round on).
// armada_sched_wk.hip compiles it a third time with ASCHED_WK_TU: the round kernel for handles whose order key takes TWO words (dev.h keyWords, WIDE_KEYS), as k_control_wk —
// every control command of such a handle runs there (the auxiliary ones too), and its key rebuild as k_bulk_wk.
#if defined(ASCHED_FT_TU)
#define K_CONTROL_NAME k_control_ft
#elif defined(ASCHED_WK_TU)
#define K_CONTROL_NAME k_control_wk
#else
#define K_CONTROL_NAME k_control
#endif
#ifdef ASCHED_WK_TU
__global__ __launch_bounds__(CTL_THREADS) void K_CONTROL_NAME(Dev dev, int cmd, HelpBox* box, int H, MktDev mk) {
  if (threadIdx.x == 0 && blockIdx.x != 0) g_mk = mk;   // (helper workgroups: the market state's HBM homes — none of the bodies they serve looks at it; never garbage)
  if (threadIdx.x == 0 && blockIdx.x == 0) {   // (market-driven rounds of such a handle run here too: round_mkt.h)
    g_mk = mk; g_xgen = 0; g_xpeers = 0;       // the exchange generation of sharded passes restarts with every launch through the host proxy ...
    if (dev.cfg.shardWorld > 1 && dev.cancel) {
      const unsigned long long* X = (const unsigned long long*)dev.cancel;
      unsigned long long pt = __hip_atomic_load(&X[XCHG_WORD0 + 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (pt) {                                // ... and goes on where the last launch left it GPU-to-GPU (the peers' counters do not restart either)
        g_xpeers = pt;
        g_xgen = (unsigned int)__hip_atomic_load(((unsigned long long* const*)pt)[dev.cfg.shardRank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)&X[XCHG_WORD0 + 7], (unsigned long long)g_xgen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the launch's first generation: its end turns this into a count)
      }
    }
  }
#else
__global__ __launch_bounds__(CTL_THREADS) void K_CONTROL_NAME(Dev dev, int cmd, HelpBox* box, int H) {
#endif
  if (blockIdx.x != 0) { helperMain(dev, box, H); return; }
  if (threadIdx.x == 0) { g_box = box; g_H = H; g_gen = 0; g_fl.eng.abandon = 0; g_fl.eng.idleSince = 0; g_fl.eng.idleLast = 0; g_fl.eng.idleProg = 0; }
  // the Dev descriptor (pointers + config) is staged in LDS once; every wave reads it from there
  {
    const int* src = (const int*)&dev; int* dst = (int*)&g_dev;
    for (int i = threadIdx.x; i < (int)(sizeof(Dev) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  Dev& d = g_dev;
  relocateIn(d, cmd);
  if (threadIdx.x >= 64) {  // worker waves: serve mailbox requests until OP_EXIT
    for (;;) {
      __syncthreads();
      int op = g_mb.op;
      if (op == OP_EXIT) break;
      if (op == OP_SCAN) {
        unsigned long long v = scanPart(d, g_mb.scan, threadIdx.x, (g_H + 1) * (int)blockDim.x);
        if ((threadIdx.x & 63) == 0) g_mb.partial[threadIdx.x >> 6] = v;
      } else if (op == OP_FAIR) {
        int v = fairPart(d, g_mb.fair, threadIdx.x, (g_H + 1) * (int)blockDim.x);
        if ((threadIdx.x & 63) == 0) g_mb.waveCount[threadIdx.x >> 6] = v;
      } else if (op == OP_SCANFAIR) {
        unsigned long long v = scanPart(d, g_mb.scan, threadIdx.x, (g_H + 1) * (int)blockDim.x);
        int w = fairPart(d, g_mb.fair, threadIdx.x, (g_H + 1) * (int)blockDim.x);
        if ((threadIdx.x & 63) == 0) { g_mb.partial[threadIdx.x >> 6] = v; g_mb.waveCount[threadIdx.x >> 6] = w; }
      } else if (op == OP_BULK) {
        bulkPart(d, g_mb.kind, g_mb.n);
      } else if (op == OP_BULKW) {
        int nthreads = (g_H + 1) * (int)blockDim.x; int kd = g_mb.kind, nn = g_mb.n;
        for (int i = threadIdx.x; i < nn; i += nthreads) bulkElem(d, kd, i);
        __threadfence();
      }
      else if (op == OP_WIDE) {
        int nthreads = (g_H + 1) * (int)blockDim.x; int kd = g_mb.kind, nn = g_mb.n;
        for (int i = threadIdx.x; i < nn; i += nthreads) wideBulkAny(d, kd, i);
        __threadfence();
      }
#ifndef ASCHED_NO_FT
      else if (op == OP_FTBUILD) {
        int nthreads = (g_H + 1) * (int)blockDim.x; int ph = g_mb.kind, nn = g_mb.n;
        for (int i = threadIdx.x; i < nn; i += nthreads) ftBuildAny(d, ph, i);
        __threadfence();
      }
#endif
      else if (op == OP_COMPACT) {
        compactPart(d);
      } else if (op == OP_ENGINE) {
        if ((threadIdx.x >> 6) == 1) engineLoop(d); else if ((threadIdx.x >> 6) == 2) bindLoop(d); else if ((threadIdx.x >> 6) == 3 && d.f.engineHc) coldLoop(d);
      }
      __syncthreads();
    }
    relocateOut();
    return;
  }
#ifdef ASCHED_WK_TU
  if (cmd >= CMD_AUX_FIRST) controlMainAux(d, cmd); else
#endif
  controlMain(d, cmd);
#ifdef ASCHED_WK_TU
  // GPU-to-GPU exchanges of this launch, for asched_shard_exchanges (the host counts the proxy's itself): the counter went on from the area's word
  if (threadIdx.x == 0 && g_xpeers) {
    unsigned long long* X = (unsigned long long*)d.cancel;
    unsigned int start = (unsigned int)__hip_atomic_load(&X[XCHG_WORD0 + 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&X[XCHG_WORD0 + 7], (unsigned long long)(g_xgen - start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#endif
  __threadfence();
  if ((threadIdx.x & 63) == 0) { g_mb.op = OP_EXIT; if (g_H) helpIssue(OP_HELPERS_EXIT, (const ScanArgs*)nullptr); }
  __syncthreads();
  relocateOut();
}

#ifdef ASCHED_FT_TU
extern "C" __attribute__((visibility("hidden"))) int asched_internal_ft_launch(const Dev* dev, int cmd, hipStream_t stream, void* helpBox, int H) {
  hipLaunchKernelGGL(k_control_ft, dim3(1 + H), dim3(CTL_THREADS), 0, stream, *dev, cmd, (HelpBox*)helpBox, H);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#elif defined(ASCHED_WK_TU)
__global__ __launch_bounds__(256) void k_bulk_wk(Dev d, int kind, int n) {
  // (the element bodies ask mkOn(): this code object carries the market-driven round, whose state is an LDS copy of a kernel argument of k_control_wk.  A market round is ONE
  //  launch of that kernel — the grid-wide phases never belong to one: no market state here)
  if (threadIdx.x == 0) memset(&g_mk, 0, sizeof g_mk);
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bulkElem(d, kind, i);
}
// k_fit_batch (below, main translation unit) for a two-word key, one launch per word: pass 0 leaves the minimum HIGH word among a shape's fitting nodes in out[i][0],
// pass 1 the minimum LOW word among the fitting nodes that carry it in out[i][1] (the node-index rank is in its low bits).
#define FIT_TILE_WK 256
__global__ __launch_bounds__(FIT_TILE_WK) void k_fit_batch_wk(Dev d, const int32_t* shapes, int nshapes, int level, unsigned long long* out, int pass) {
  const DevCfg& c = d.cfg;
  int n = blockIdx.x * FIT_TILE_WK + threadIdx.x;
  bool valid = n < c.N;
  unsigned long long hi = valid ? d.keys[(size_t)level * c.Npad + n] : ~0ull;
  unsigned long long lo = (valid && pass) ? d.keys[((size_t)c.P + level) * c.Npad + n] : ~0ull;
  int64_t al[MAXR];
  for (int r = 0; r < MAXR; r++) al[r] = (valid && r < c.R) ? d.alloc[((size_t)level * c.R + r) * c.Npad + n] : 0;
  int per = (nshapes + gridDim.y - 1) / gridDim.y;
  int s0 = blockIdx.y * per, s1 = min(nshapes, s0 + per);
  int word = n >> 6, bit = n & 63;
  __shared__ unsigned long long wmin[FIT_TILE_WK / 64];
  for (int i = s0; i < s1; i++) {
    int s = shapes[i];
    bool f = valid && ((d.shapeMask[(size_t)s * c.W + word] >> bit) & 1);
    const int64_t* req = d.shapeReq + (size_t)s * c.R;
    for (int r = 0; r < c.R; r++) f = f && req[r] <= al[r];
    unsigned long long key = hi;
    if (pass) { f = f && hi == out[(size_t)i * FIT_OSTR]; key = lo; }   // (word 0 is final: pass 0 completed on this stream)
    unsigned long long v = __ballot(f) ? waveMin64Dpp(f ? key : ~0ull) : ~0ull;
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wmin[0];
      for (int w = 1; w < FIT_TILE_WK / 64; w++) m = wmin[w] < m ? wmin[w] : m;
      unsigned long long* o = &out[(size_t)i * FIT_OSTR + pass];
      if (m != ~0ull && m < __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(o, m);
    }
    __syncthreads();
  }
}
extern "C" __attribute__((visibility("hidden"))) int asched_internal_wk_fit_batch(const Dev* dev, const int32_t* shapes, int ns, int level, unsigned long long* out, int tiles, int ysplit, hipStream_t stream) {
  for (int pass = 0; pass < 2; pass++) hipLaunchKernelGGL(k_fit_batch_wk, dim3(tiles, ysplit), dim3(FIT_TILE_WK), 0, stream, *dev, shapes, ns, level, out, pass);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("hidden"))) int asched_internal_wk_launch(const Dev* dev, int cmd, hipStream_t stream, void* helpBox, int H, const MktDev* mk) {
  MktDev none; memset(&none, 0, sizeof none);
  hipLaunchKernelGGL(k_control_wk, dim3(1 + H), dim3(CTL_THREADS), 0, stream, *dev, cmd, (HelpBox*)helpBox, H, mk ? *mk : none);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("hidden"))) int asched_internal_wk_bulk(const Dev* dev, int kind, int n, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(k_bulk_wk, dim3(grid), dim3(256), 0, stream, *dev, kind, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#else   // the grid-wide kernels and the host side exist once, in the main translation unit
// ---- grid-wide kernels of the split round (asched_host.inc runRoundSplit): the data-parallel phases of PreemptingQueueScheduler.Schedule over
// all CUs.  Between launches the authoritative state is in HBM (relocateOut), so the per-element bodies of round_run.h run unchanged.
__global__ __launch_bounds__(256) void k_bulk(Dev d, int kind, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bulkElem(d, kind, i);
}
__global__ void k_round_small(Dev d, int what, int arg) { if (blockIdx.x == 0 && threadIdx.x == 0) roundSmall(d, what, arg); }

// sum of v over the lanes selected by `sel` (wave-uniform mask), returned on every lane
__device__ static inline int64_t waveSumSel(int64_t v, unsigned long long sel) {
  int lane = threadIdx.x & 63;
  int64_t x = ((sel >> lane) & 1) ? v : 0;
  for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}
// Evictor.Evict + sctx.EvictJob for every flagged job (round_run.h evictApply), grid-wide.  Jobs are walked in the pre-sorted (queue, scheduling
// order) list, so the lanes of a wave mostly share a queue: the per-queue / per-priority-class / pool sums are reduced across the wave first and
// leave as ONE atomic per (wave, key, resource) instead of one per job — same integer sums, ~64x fewer same-address atomics.
__global__ __launch_bounds__(256) void k_evict_apply(Dev d, int phase3, int total) {
  const DevCfg& c = d.cfg;
  int lane = threadIdx.x & 63;
  int rounds = (total + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  for (int it = 0; it < rounds; it++) {   // wave-uniform trip count: every lane takes part in the reductions
    int i = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    int j = i < total ? d.ordAll[i] : -1;
    bool act = j >= 0 && d.evFlag[j];
    int64_t A[MAXR], S[MAXR], E1[MAXR], E2[MAXR];
    int key = -1, q = 0, pc = 0, cntSched = 0, cntEv = 0;
#pragma unroll
    for (int r = 0; r < MAXR; r++) { A[r] = S[r] = E1[r] = E2[r] = 0; }
    if (act) {
      int n = d.jobNode[j];
      if (d.schedAtPrio[j] == NO_PRIORITY) { raise(d, ASCHED_ERR_INTERNAL, 800); act = false; }   // EvictJobsFromNode nodedb.go:1085-1088
      else {
        const int64_t* req = JREQ(d, j);
        d.jobEvictedOnNode[j] = 1;  // Node.EvictJob node.go:449-474
        atomicMarkAllocatable(d, n, d.jobCutoff[j], req, +1);
        atomicMarkAllocatable(d, n, ASCHED_EVICTED_PRIORITY, req, -1);
        d.jcEvicted[j] = 1; d.jcAssigned[j] = n; d.jcReason[j] = 0; d.jcHasPctx[j] = 0; d.jcUniValue[j] = -1; d.jcStagedBy[j] = -1;   // fresh jctx pinned to the node (eviction.go:246-253)
        int g = d.jGang[j];
        d.jcGangCard[j] = g >= 0 ? d.gangOff[g + 1] - d.gangOff[g] : 1;  // setEvictedGangCardinality pqs.go:462-483
        q = d.jQueue[j]; pc = d.jPc[j]; key = q * c.npc + pc;
        uint8_t f = d.jobFlags[j];
        bool sched = f & F_SUCCESSFUL, resched = f & F_RESCHEDULED;
        if (sched || resched) { if (sched) f &= ~F_SUCCESSFUL; if (resched) f &= ~F_RESCHEDULED; } else f |= F_EVICTED;
        d.jobFlags[j] = f;
        for (int r = 0; r < MAXR; r++) if (r < c.R) { A[r] = -req[r]; S[r] = sched ? -req[r] : 0; E1[r] = (!sched && !resched) ? req[r] : 0; E2[r] = !sched ? req[r] : 0; }
        cntSched = sched ? -1 : 0; cntEv = sched ? 0 : 1;
        if (!phase3) { d.inPreempted[j] = 1; d.preemptedNode[j] = n; }
        else if (d.inScheduled[j]) { d.inScheduled[j] = 0; d.inSchedAndEvicted[j] = 1; d.preemptedNode[j] = n; }
        else { d.inPreempted[j] = 1; d.preemptedNode[j] = n; }
      }
    }
    unsigned long long todo = __ballot(act);
    if (!todo) continue;
    // pool-wide sums: every active lane
    for (int r = 0; r < c.R; r++) {
      int64_t a = waveSumSel(A[r], todo), s2 = waveSumSel(S[r], todo), e2 = waveSumSel(E2[r], todo);
      if (lane == 0) { if (a) atomicAddI64(&d.rs->allocated[r], a); if (s2) atomicAddI64(&d.rs->scheduled[r], s2); if (e2) atomicAddI64(&d.rs->evicted[r], e2); }
    }
    { int cs = (int)waveSumSel(cntSched, todo), ce = (int)waveSumSel(cntEv, todo);
      if (lane == 0) { if (cs) atomicAddI32(&d.rs->numScheduledJobs, cs); if (ce) atomicAddI32(&d.rs->numEvictedJobs, ce); } }
    // per (queue, priority class): one group per distinct key in the wave
    while (todo) {
      int first = __ffsll((long long)todo) - 1;
      int k0 = __shfl(key, first, 64);
      unsigned long long sel = __ballot(act && key == k0) & todo;
      int q0 = k0 / c.npc;
      for (int r = 0; r < c.R; r++) {
        int64_t a = waveSumSel(A[r], sel), s2 = waveSumSel(S[r], sel), e1 = waveSumSel(E1[r], sel);
        if (lane == 0) {
          size_t ix = (size_t)k0 * c.R + r;
          if (a) { atomicAddI64(&d.qAllocByPc[ix], a); atomicAddI64(&d.qAlloc[(size_t)q0 * c.R + r], a); }
          if (s2) atomicAddI64(&d.qSchedByPc[ix], s2);
          if (e1) atomicAddI64(&d.qEvictedByPc[ix], e1);
        }
      }
      todo &= ~sel;
    }
  }
}

// order-preserving compaction of {order[p] : flag[order[p]]} over the whole grid (order == NULL: identity): count per 4096-element block, scan
// of the block counts, ordered write.  prefix[p] = number of flagged elements before p (may be NULL).
#define CMP_CHUNK 4096
__global__ __launch_bounds__(256) void k_cmp_count(const int32_t* order, int n, const uint8_t* flag, int32_t* blockCount) {
  __shared__ int wsum[4];
  int base = blockIdx.x * CMP_CHUNK, cnt = 0;
  for (int o = threadIdx.x; o < CMP_CHUNK; o += 256) { int p = base + o; if (p < n && flag[order ? order[p] : p]) cnt++; }
  for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) blockCount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void k_cmp_scan(int32_t* blockCount, int nblocks, int32_t* totalOut) {   // one thread: a few hundred blocks at most
  if (blockIdx.x || threadIdx.x) return;
  int run = 0;
  for (int b = 0; b < nblocks; b++) { int v = blockCount[b]; blockCount[b] = run; run += v; }
  *totalOut = run;
}
__global__ __launch_bounds__(256) void k_cmp_write(const int32_t* order, int n, const uint8_t* flag, int32_t* dst, uint32_t* prefix, const int32_t* blockOffset) {
  __shared__ int wcnt[4];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int run = blockOffset[blockIdx.x];
  for (int t = 0; t < CMP_CHUNK / 256; t++) {
    int p = blockIdx.x * CMP_CHUNK + t * 256 + threadIdx.x;
    int v = p < n ? (order ? order[p] : p) : 0;
    bool f = p < n && flag[v];
    unsigned long long b = __ballot(f);
    if (lane == 0) wcnt[wave] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < 4; w++) { int cw = wcnt[w]; if (w < wave) off += cw; tot += cw; }
    int rank = run + off + __popcll(b & ((1ull << lane) - 1));
    if (p < n && prefix) prefix[p] = rank;
    if (f) dst[rank] = v;
    run += tot;
    __syncthreads();
  }
}
__global__ void k_seg_off(const int32_t* segOff, int nseg, int n, const uint32_t* prefix, const int32_t* total, int32_t* outSegOff) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q <= nseg) outSegOff[q] = segOff[q] < n ? (int32_t)prefix[segOff[q]] : *total;
}

// ---- fairness optimiser (round_opt.h): per-node job lists (count / scan / scatter), queue costs, then every node scored for one job at once
__global__ __launch_bounds__(256) void k_opt_count(Dev d, int32_t* cnt) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < d.cfg.M; j += gridDim.x * blockDim.x) { int n = d.jobNode[j]; if (n >= 0) atomicAdd(&cnt[n], 1); int g = d.rs->optMode ? d.optGhost[j] : -1; if (g >= 0) atomicAdd(&cnt[g], 1); }   // (ghost: dev.h optGhost)
}
__global__ __launch_bounds__(1024) void k_opt_scan(const int32_t* cnt, int32_t* off, int32_t* cursor, int N) {   // one block: chunk sums, serial scan of 1024 partials, chunk offsets
  __shared__ int part[1024];
  int C = (N + 1023) / 1024, n0 = threadIdx.x * C, n1 = n0 + C < N ? n0 + C : N;
  int sum = 0;
  for (int n = n0; n < n1; n++) sum += cnt[n];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 1024; i++) { int v = part[i]; part[i] = run; run += v; } off[N] = run; }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int n = n0; n < n1; n++) { off[n] = run; cursor[n] = run; run += cnt[n]; }
}
__global__ __launch_bounds__(256) void k_opt_scatter(Dev d, int32_t* cursor, int32_t* jobs) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < d.cfg.M; j += gridDim.x * blockDim.x) { int n = d.jobNode[j]; if (n >= 0) jobs[atomicAdd(&cursor[n], 1)] = j; int g = d.rs->optMode ? d.optGhost[j] : -1; if (g >= 0) jobs[atomicAdd(&cursor[g], 1)] = j; }
}
__global__ void k_opt_qcost(Dev d, int job, double* qCost) {   // QueueContext.CurrentCost per queue (scheduling_context.go:19-24); [Q]: the job's own DRF cost
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < d.cfg.Q) {
    int64_t a[MAXR];
    for (int r = 0; r < MAXR; r++) a[r] = r < d.cfg.R ? QV(d.qAlloc, q)[r] + QV(d.qPenalty, q)[r] : 0;
    qCost[q] = d.optQDelta ? d.optQDelta[q] : drf(d, a);   // (later members of a gang: CurrentCost as updateState left it, kept by the host)
  } else if (q == d.cfg.Q) qCost[q] = drf(d, JREQ(d, job));
}
__global__ __launch_bounds__(128) void k_opt_score(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, OptNodeOut* out) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < d.cfg.N) optScoreNode(d, a, qCost, off, jobs, d.jLeaseMs, n, &out[n], nullptr);
}
// ---- k_opt_score_wave: PreemptingNodeScheduler.Schedule (round_opt.h optScoreNodeE) with ONE WAVE per node, one lane per job on the node.
// The one-node-per-thread kernel walks its node's job list serially — a chain of dependent gathers per job and two insertion sorts in private memory (2.4 KB of
// scratch per thread) — and leaves most of the chip idle (20 000 nodes = 313 waves).  Here the gathers of a node are one round trip (lane k loads job k's row), the two
// orderings are rank sorts (lane k counts the entries that order before its own; entries are broadcast with v_readlane, so the loop is wave-uniform and as long as the
// node's job count), the fit prefix is a wave scan of the request vectors and "first prefix that fits" a ballot.  What the reference computes with SEQUENTIAL float
// arithmetic keeps its order: the running queue cost (rounded after every subtraction), the sum of the preemption costs and the per-queue cost changes are serial
// loops over broadcast values — every lane performs the same operations in the same order, so the doubles are the ones the serial routine produces.
// Nodes with more than 64 jobs report overflow (-1) like the private-list kernel and go through k_opt_score_big.
template <class T> __device__ static inline T wvRead(T v, int lane) {   // lane: wave-uniform
  static_assert(sizeof(T) % 4 == 0, "dword multiples");
  int w[sizeof(T) / 4]; T r;
  __builtin_memcpy(w, &v, sizeof(T));
  for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = __builtin_amdgcn_readlane(w[k], lane);
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
template <class T> __device__ static inline T wvPush(T v, int dstLane) {   // lane dstLane receives this lane's v (ds_permute: dstLane must be a permutation of the lanes)
  int w[sizeof(T) / 4]; T r;
  __builtin_memcpy(w, &v, sizeof(T));
  for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = __builtin_amdgcn_ds_permute(dstLane << 2, w[k]);
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
template <class T> __device__ static inline T wvPull(T v, int srcLane) {   // this lane receives lane srcLane's v
  int w[sizeof(T) / 4]; T r;
  __builtin_memcpy(w, &v, sizeof(T));
  for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = __builtin_amdgcn_ds_bpermute(srcLane << 2, w[k]);
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
struct OptLane { int32_t job, queue, sap, ordinal, prioPre, ctpZero; int64_t age; double cost, wcap; };
__device__ static inline bool optInQueueLessL(const OptLane& a, const OptLane& b) {   // round_opt.h optInQueueLess
  if (a.queue != b.queue) return a.queue < b.queue;
  if (a.sap != b.sap) return a.sap < b.sap;
  if (a.cost != b.cost) return a.cost < b.cost;
  if (a.age != b.age) return a.age < b.age;
  return a.job < b.job;
}
__device__ static inline bool optGlobalLessL(const OptLane& a, const OptLane& b) {    // round_opt.h optGlobalLess
  if (a.queue == b.queue) return a.ordinal < b.ordinal;
  if (a.prioPre != b.prioPre) return a.prioPre != 0;
  if (a.wcap > b.wcap) return true;
  if (a.wcap == b.wcap) {
    if (a.sap != b.sap) return a.sap < b.sap;
    if (a.cost != b.cost) return a.cost < b.cost;
    if (a.age != b.age) return a.age < b.age;
    return a.job < b.job;
  }
  return false;
}
// entries to their ranks: valid lanes go to lane `rank` (0 .. m-1), the others fill m .. 63 in lane order, so the move is a permutation
__device__ static inline int optDest(bool valid, int rank, unsigned long long validMask, int lane) {
  int m = __builtin_popcountll(validMask);
  int invalidBefore = __builtin_popcountll(~validMask & ((1ull << lane) - 1));
  return valid ? rank : m + invalidBefore;
}
// one wave, node n: *outp = the node's score (written by lane 0), preOut (optional) = the victims in preemption order
__device__ static void optScoreNodeWave(Dev& d, const OptArgs& a, const double* qCost, const int32_t* off, const int32_t* jobs, int n, OptNodeOut* outp, int32_t* preOut) {
  const DevCfg& c = d.cfg;
  const int lane = threadIdx.x & 63;
  OptNodeOut* out = outp - n;                                                  // (the body below writes out[n])
  OptNodeOut res; res.scheduled = 0; res.npre = 0; res.cost = 0; res.impact = 0;
  const int job = a.job;
  const uint64_t* mask = d.shapeMask + (size_t)d.jShape[job] * c.W;
  if (!((mask[n >> 6] >> (n & 63)) & 1)) { if (lane == 0) out[n] = res; return; }
  const int64_t* req = JREQ(d, job);
  int64_t avail[MAXR];
  bool fits0 = true;
  for (int r = 0; r < MAXR; r++) { avail[r] = r < c.R ? AL(d, c.evLevel, r, n) : 0; if (r < c.R && req[r] > avail[r]) fits0 = false; }
  if (fits0) { res.scheduled = 1; if (lane == 0) out[n] = res; return; }
  const int k0 = off[n], cnt = off[n + 1] - k0;
  if (cnt > 64) { res.scheduled = -1; if (lane == 0) out[n] = res; return; }   // more jobs than lanes: scored by k_opt_score_big
  const int32_t jobPrio = c.pcPriority[d.jPc[job]];
  // ---- one lane per job on the node (node.AllocatedByJobId, node_scheduler.go:137-200)
  OptLane e; e.job = 0x7fffffff; e.queue = 0; e.sap = 0; e.ordinal = 0; e.prioPre = 0; e.ctpZero = 1; e.age = 0; e.cost = 0; e.wcap = 0;
  int64_t jr[MAXR];
  for (int r = 0; r < MAXR; r++) jr[r] = 0;
  bool valid = false;
  if (lane < cnt) {
    int j = jobs[k0 + lane];
    bool ok = c.pcPreemptible[d.jPc[j]] != 0 && d.jGang[j] < 0;
    const int64_t* q = JREQ(d, j);
    if (ok && a.hasMaxSize) for (int r = 0; r < c.R; r++) if (a.maxSize[r] != 0 && q[r] > a.maxSize[r]) ok = false;
    int32_t sap = d.schedAtPrio[j];
    ok = ok && sap != NO_PRIORITY && sap <= jobPrio;
    if (ok) {
      valid = true;
      e.job = j; e.queue = d.jQueue[j]; e.sap = sap;
      e.age = d.jNode0[j] < 0 ? 0 : a.nowMs - d.jLeaseMs[j];
      e.cost = drf(d, q);
      for (int r = 0; r < c.R; r++) jr[r] = q[r];
    }
  }
  unsigned long long vm = __ballot(valid);
  const int m = __builtin_popcountll(vm);
  if (m == 0) { if (lane == 0) out[n] = res; return; }
  // ---- per queue order (optInQueueLess): rank = how many entries order before mine
  {
    int rank = 0;
    for (int i = 0; i < cnt; i++) {
      if (!((vm >> i) & 1)) continue;
      OptLane o = wvRead(e, i);
      if (valid && optInQueueLessL(o, e)) rank++;
    }
    int dst = optDest(valid, rank, vm, lane);
    e = wvPush(e, dst);
    for (int r = 0; r < c.R; r++) jr[r] = wvPush(jr[r], dst);
  }
  valid = lane < m;
  // the queue's cost, weight and capped fair share, one gather per lane (broadcast below)
  double qc = valid ? qCost[e.queue] : 0.0, qw = valid ? d.qWeight[e.queue] : 1.0, qd = valid ? d.qDc[e.queue] : 0.0;
  // ---- populateQueueImpactFields (:203-232): the running queue cost is rounded after every subtraction — in order, on broadcast values
  {
    double updated = 0; int prevQ = -1, ord = 0;
    for (int i = 0; i < m; i++) {
      int qi = wvRead(e.queue, i);
      if (qi != prevQ) { updated = wvRead(qc, i); ord = 0; prevQ = qi; }
      updated = optRound8(updated - wvRead(e.cost, i));
      double w = updated / wvRead(qw, i);
      int sapi = wvRead(e.sap, i);
      int prioPre = sapi < jobPrio, ctpZero = (sapi < jobPrio) || (updated > wvRead(qd, i));
      if (lane == i) { e.wcap = w; e.prioPre = prioPre; e.ctpZero = ctpZero; e.ordinal = ord; }
      ord++;
    }
  }
  // ---- global preemption order (optGlobalLess)
  {
    int rank = 0;
    for (int i = 0; i < m; i++) {
      OptLane o = wvRead(e, i);
      if (valid && optGlobalLessL(o, e)) rank++;
    }
    unsigned long long m2 = m >= 64 ? ~0ull : ((1ull << m) - 1);
    int dst = optDest(valid, rank, m2, lane);
    e = wvPush(e, dst); qc = wvPush(qc, dst);
    for (int r = 0; r < c.R; r++) jr[r] = wvPush(jr[r], dst);
  }
  // ---- preempt one job at a time until the job fits (:84-99): inclusive prefix sums of the victims' requests, first prefix that fits
  for (int r = 0; r < c.R; r++) {
    int64_t v = valid ? jr[r] : 0;
    for (int s = 1; s < 64; s <<= 1) { int64_t o = wvPull(v, lane >= s ? lane - s : lane); if (lane >= s) v += o; }
    jr[r] = v;
  }
  bool f = valid;
  for (int r = 0; r < c.R; r++) if (req[r] > avail[r] + jr[r]) f = false;
  unsigned long long fm = __ballot(f);
  if (fm == 0) { if (lane == 0) out[n] = res; return; }
  const int used = __builtin_ctzll(fm) + 1;
  double total = 0;
  for (int i = 0; i < used; i++) total += wvRead(e.ctpZero, i) ? 0.0 : wvRead(e.cost, i);
  // maximumQueueImpact (:101-113): per queue |sum of the preempted jobs' costs, in preemption order| / CurrentCost
  double change = 0;
  for (int i = 0; i < used; i++) { int qi = wvRead(e.queue, i); double ci = wvRead(e.cost, i); if (qi == e.queue) change -= ci; }
  double imp = lane < used ? fabs(change) / qc : 0.0;
  if (!(imp > 0.0)) imp = 0.0;   // (the serial routine keeps a value only if it compares greater than the running maximum: a NaN never does)
  for (int s = 32; s; s >>= 1) { double o = __shfl_xor(imp, s, 64); imp = o > imp ? o : imp; }
  res.scheduled = 1; res.npre = used; res.cost = total; res.impact = imp;
  if (preOut && lane < used) preOut[lane] = e.job;
  if (lane == 0) out[n] = res;
}
__global__ __launch_bounds__(256) void k_opt_score_wave(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, OptNodeOut* out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= d.cfg.N) return;                                                    // (whole waves leave together: one node per wave)
  optScoreNodeWave(d, a, qCost, off, jobs, n, &out[n], nullptr);
}
__global__ __launch_bounds__(64) void k_opt_detail_wave(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, int n, OptNodeOut* out, int32_t* pre) {
  optScoreNodeWave(d, a, qCost, off, jobs, n, out, pre);
}
// ---- the candidate selection of FairnessOptimisingGangScheduler.scheduleOnNodes (gang_scheduler.go:100-141) on the device, so that one asched_optimiser_schedule_job is one
// stream-ordered sequence (queue costs -> scores -> selection -> victims of the selected node) with a single small download instead of 20 000 scores and two round trips.
// Nodes in id order: the first that needs no preemption wins outright; otherwise the smallest (schedulingCost, maximumQueueImpact) among those whose fairness improvement
// exceeds the threshold, the earlier id on a tie (the reference draws a ULID).  `overflow` counts nodes the wave kernel could not score (more than 64 jobs): the host then
// takes the long way (k_opt_score_big + its own loop).
struct OptSel { int32_t node, npre, big, overflow; double cost, impact; };
struct OptSelKey { int32_t cat, rank, node, npre; double cost, impact; };   // cat 0: no preemption needed, 1: candidate, 2: nothing
__device__ static inline bool optSelLess(const OptSelKey& a, const OptSelKey& b) {
  if (a.cat != b.cat) return a.cat < b.cat;
  if (a.cat == 2) return false;
  if (a.cat == 1) { if (a.cost != b.cost) return a.cost < b.cost; if (a.impact != b.impact) return a.impact < b.impact; }
  return a.rank < b.rank;
}
__device__ static inline OptSelKey optSelReduceWave(OptSelKey k) {
  for (int s = 32; s; s >>= 1) {
    OptSelKey o;
    o.cat = __shfl_xor(k.cat, s, 64); o.rank = __shfl_xor(k.rank, s, 64); o.node = __shfl_xor(k.node, s, 64); o.npre = __shfl_xor(k.npre, s, 64);
    o.cost = __shfl_xor(k.cost, s, 64); o.impact = __shfl_xor(k.impact, s, 64);
    if (optSelLess(o, k)) k = o;
  }
  return k;
}
__global__ __launch_bounds__(256) void k_opt_select(Dev d, const OptNodeOut* out, const uint8_t* mask, const double* jobCostPtr, double minPct, OptSelKey* partial, int32_t* overflow) {
  __shared__ OptSelKey wk[4];
  int n = blockIdx.x * 256 + threadIdx.x;
  OptSelKey k; k.cat = 2; k.rank = 0x7fffffff; k.node = -1; k.npre = 0; k.cost = 0; k.impact = 0;
  if (n < d.cfg.N && (!mask || mask[n])) {
    OptNodeOut r = out[n];
    if (r.scheduled < 0) atomicAdd(overflow, 1);
    if (r.scheduled > 0) {
      double jobCost = *jobCostPtr;
      bool ideal = r.cost == 0 && r.npre == 0;                                   // :112-116
      double improvement = ((jobCost / r.cost) * 100) - 100;                     // :118-121 (cost 0 with victims: +Inf)
      if (ideal || improvement > minPct) { k.cat = ideal ? 0 : 1; k.rank = d.nodeIdRank ? d.nodeIdRank[n] : n; k.node = n; k.npre = r.npre; k.cost = r.cost; k.impact = r.impact; }
    }
  }
  k = optSelReduceWave(k);
  if ((threadIdx.x & 63) == 0) wk[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < 4; w++) if (optSelLess(wk[w], k)) k = wk[w]; partial[blockIdx.x] = k; }
}
__global__ __launch_bounds__(256) void k_opt_select_final(const OptSelKey* partial, int nb, const int32_t* overflow, OptSel* sel) {
  __shared__ OptSelKey wk[4];
  OptSelKey k; k.cat = 2; k.rank = 0x7fffffff; k.node = -1; k.npre = 0; k.cost = 0; k.impact = 0;
  for (int i = threadIdx.x; i < nb; i += 256) if (optSelLess(partial[i], k)) k = partial[i];
  k = optSelReduceWave(k);
  if ((threadIdx.x & 63) == 0) wk[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) if (optSelLess(wk[w], k)) k = wk[w];
    sel->node = k.cat == 2 ? -1 : k.node; sel->npre = k.npre; sel->big = 0; sel->overflow = *overflow; sel->cost = k.cost; sel->impact = k.impact;
  }
}
__global__ __launch_bounds__(64) void k_opt_detail_sel(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, OptSel* sel, OptNodeOut* scratchOut, int32_t* pre) {
  int n = sel->node;
  if (n < 0 || sel->npre == 0 || sel->overflow) return;
  if (off[n + 1] - off[n] > 64) { if (threadIdx.x == 0) sel->big = 1; return; }   // (cannot happen while overflow == 0; kept as a guard)
  optScoreNodeWave(d, a, qCost, off, jobs, n, scratchOut, pre);
}
__global__ void k_opt_detail(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, int n, OptNodeOut* out, int32_t* pre) {
  if (blockIdx.x == 0 && threadIdx.x == 0) optScoreNode(d, a, qCost, off, jobs, d.jLeaseMs, n, out, pre);
}
// nodes with more than OPT_MAXJ candidates (k_opt_score reported overflow): the same routine with the entry list in an HBM scratch sized by the node's job count
__global__ void k_opt_score_big(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, const int32_t* nodes, const long long* eOff, int nb, OptEntry* scratch, OptNodeOut* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) { int n = nodes[i]; optScoreNodeE(d, a, qCost, off, jobs, d.jLeaseMs, n, &out[n], nullptr, scratch + eOff[i], off[n + 1] - off[n]); }
}
__global__ void k_opt_detail_big(Dev d, OptArgs a, const double* qCost, const int32_t* off, const int32_t* jobs, int n, OptNodeOut* out, int32_t* pre, OptEntry* scratch) {
  if (blockIdx.x == 0 && threadIdx.x == 0) optScoreNodeE(d, a, qCost, off, jobs, d.jLeaseMs, n, out, pre, scratch, off[n + 1] - off[n]);
}

// the indicative gang pricer (round_price.h): every node priced for one gang member; the entry list shares the layout of the node -> jobs index
__global__ __launch_bounds__(128) void k_price_score(Dev d, PriceArgs a, const int32_t* off, const int32_t* jobs, PriceEntry* entries, PriceNodeOut* out) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < d.cfg.N) priceScoreNode(d, a, off, jobs, d.jLeaseMs, n, &out[n], nullptr, entries + off[n]);
}
__global__ void k_price_detail(Dev d, PriceArgs a, const int32_t* off, const int32_t* jobs, PriceEntry* entries, int n, PriceNodeOut* out, int32_t* pre) {
  if (blockIdx.x == 0 && threadIdx.x == 0) priceScoreNode(d, a, off, jobs, d.jLeaseMs, n, out, pre, entries + off[n]);
}

__global__ void k_shape_mask(Dev d, const uint64_t* classMask, const int32_t* shapeClass) {
  const DevCfg& c = d.cfg;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)c.S * c.W) return;
  int s = (int)(t / c.W), w = (int)(t % c.W);
  uint64_t cm = classMask[(size_t)shapeClass[s] * c.W + w], m = 0;
  for (int b = 0; b < 64; b++) {
    int n = w * 64 + b;
    if (n >= c.N) break;
    if (!((cm >> b) & 1)) continue;
    bool ok = true;
    for (int r = 0; r < c.R; r++) ok = ok && d.shapeReq[(size_t)s * c.R + r] <= d.totalRes[(size_t)r * c.Npad + n];  // nodematching.go:184
    if (ok) m |= 1ull << b;
  }
  d.shapeMask[t] = m;
}

// First feasible node for a batch of (shape) queries at one level against the current node state.
// grid.x tiles the nodes (one node per thread, its key and R alloc values stay in registers for the whole
// shape loop), grid.y splits the shape list.  HBM traffic per launch = N*(8 + 8R) bytes + masks.
#define FIT_TILE 256
__global__ __launch_bounds__(FIT_TILE) void k_fit_batch(Dev d, const int32_t* shapes, int nshapes, int level, unsigned long long* out) {
  const DevCfg& c = d.cfg;
  int n = blockIdx.x * FIT_TILE + threadIdx.x;
  bool valid = n < c.N;
  unsigned long long key = valid ? d.keys[(size_t)level * c.Npad + n] : ~0ull;
  int64_t al[MAXR];
  for (int r = 0; r < MAXR; r++) al[r] = (valid && r < c.R) ? d.alloc[((size_t)level * c.R + r) * c.Npad + n] : 0;
  int per = (nshapes + gridDim.y - 1) / gridDim.y;
  int s0 = blockIdx.y * per, s1 = min(nshapes, s0 + per);
  int word = n >> 6, bit = n & 63;
  __shared__ unsigned long long wmin[FIT_TILE / 64];
  for (int i = s0; i < s1; i++) {
    int s = shapes[i];
    bool f = valid && ((d.shapeMask[(size_t)s * c.W + word] >> bit) & 1);
    const int64_t* req = d.shapeReq + (size_t)s * c.R;
    for (int r = 0; r < c.R; r++) f = f && req[r] <= al[r];
    unsigned long long v = __ballot(f) ? waveMin64Dpp(f ? key : ~0ull) : ~0ull;
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = v;
    __syncthreads();
    // ONE look / write per workgroup and shape, at a word that has a cache line of its own.  (Round 3: every wave sent its minimum to out[i], 8 bytes from out[i + 1]: 100 000
    // read-modify-writes on four cache lines at 100 000 nodes x 64 shapes, serialised in one L2 channel — 0.19 ms for a 4 MB problem.)  The word only ever falls, and first fit
    // means it falls early: look first, write only what improves it.
    if (threadIdx.x == 0) {
      unsigned long long m = wmin[0];
      for (int w = 1; w < FIT_TILE / 64; w++) m = wmin[w] < m ? wmin[w] : m;
      if (m != ~0ull && m < __hip_atomic_load(&out[(size_t)i * FIT_OSTR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&out[(size_t)i * FIT_OSTR], m);
    }
    __syncthreads();
  }
}

// ---- sorted base of the level-0 fast structure: the ordered index of the fresh NodeDb (nodedb.go:1164-1175), built in round_prepare
__global__ void k_base_fill(Dev d, int nb2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb2) d.baseKey[i] = i < d.cfg.N ? fastKeyOf(d, i) : ~0ull;  // level 0 plane of keys (a negative column: field 0, fits nothing)
}
__global__ void k_bitonic_step(unsigned long long* a, int j, int k) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned l = i ^ (unsigned)j;
  if (l > i) {
    unsigned long long x = a[i], y = a[l];
    bool up = (i & (unsigned)k) == 0;
    if (up ? x > y : x < y) { a[i] = y; a[l] = x; }
  }
}
// the in-LDS part of the network: every (k, j) step with j < 2048 for one 4096-key tile, 1024 threads
__global__ __launch_bounds__(1024) void k_bitonic_tile(unsigned long long* a, int kStart, int kEnd, int jStart) {
  __shared__ unsigned long long t[4096];
  unsigned base = blockIdx.x * 4096u;
  for (int i = threadIdx.x; i < 4096; i += 1024) t[i] = a[base + i];
  __syncthreads();
  for (int k = kStart; k <= kEnd; k <<= 1) {
    for (int j = (k == kStart ? jStart : k >> 1); j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < 4096; i += 1024) {
        unsigned l = (unsigned)i ^ (unsigned)j;
        if (l > (unsigned)i) {
          unsigned long long x = t[i], y = t[l];
          bool up = ((base + i) & (unsigned)k) == 0;
          if (up ? x > y : x < y) { t[i] = y; t[l] = x; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < 4096; i += 1024) a[base + i] = t[i];
}
__global__ void k_base_finish(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const DevCfg& c = d.cfg;
  if (i >= c.N) return;
  unsigned long long key = d.baseKey[i];
  int node = d.nodeByRank[key & ((1ull << c.idxBits) - 1)];
  d.baseNode[i] = node; d.posOf[node] = i; d.baseRemoved[i] = 0; d.baseCls[i] = d.nodeCls[node]; d.l0Slot[node] = -1;
  for (int e = 0; e < d.f.E; e++) d.baseExtra[(size_t)e * c.Npad + i] = d.alloc[(size_t)d.f.extraCol[e] * c.Npad + node];  // level 0 planes
}


// The round-input builder's sums (round_run.h B_AGG_RUN / B_AGG_QUEUED) grid-wide with the wave-level pre-reduction of k_evict_apply: both walks are
// ordered by queue (the pre-sorted job order; the queued lists), so a wave holds a handful of (queue, class) keys and leaves one atomic per key and resource.
__global__ __launch_bounds__(256) void k_agg(Dev d, int queued, int total) {
  const DevCfg& c = d.cfg;
  int lane = threadIdx.x & 63;
  int rounds = (total + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  for (int it = 0; it < rounds; it++) {   // wave-uniform trip count
    int i = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    bool act = false; int key = -1; int64_t V[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) V[r] = 0;
    if (i < total) {
      int j, q;
      if (queued) {
        int lo = 0, hi = c.Q;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (d.queuedOff[mid + 1] <= i) lo = mid + 1; else hi = mid; }
        q = lo; j = d.queuedJobs[i];
        act = q < c.Q && !d.qCordoned[q];
      } else {
        j = d.ordAll[i]; q = d.jQueue[j];
        act = d.jNode0[j] >= 0 && q >= 0 && q < c.Q;
      }
      if (act) { key = q * c.npc + d.jPc[j]; const int64_t* req = JREQ(d, j); for (int r = 0; r < MAXR; r++) if (r < c.R) V[r] = req[r]; }
    }
    unsigned long long todo = __ballot(act);
    while (todo) {
      int first = __ffsll((long long)todo) - 1;
      int k0 = __shfl(key, first, 64);
      unsigned long long sel = __ballot(act && key == k0) & todo;
      for (int r = 0; r < c.R; r++) {
        int64_t v = waveSumSel(V[r], sel);
        if (lane == 0 && v) { size_t ix = (size_t)k0 * c.R + r; atomicAddI64(&d.qDemandByPc[ix], v); if (!queued) atomicAddI64(&d.qAllocByPc[ix], v); }
      }
      todo &= ~sel;
    }
  }
}
// the fit bitmaps of a fresh base (round_fast.h fitBitsWord): one thread per (fit shape, 64 entries)
__global__ void k_base_fitbits(Dev d) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)d.f.F * d.fitW;
  if (i >= total) return;
  d.fitBits[i] = fitBitsWord(d, (int)(i / d.fitW), (int)(i % d.fitW));
}
__global__ void k_drf(Dev d, const int64_t* alloc, double* out) { if (threadIdx.x == 0) *out = drf(d, alloc); }
__global__ void k_fair(Dev d, const double* cds) { if (threadIdx.x == 0) updateFairShares(d, cds); }

// ------------------------------------------------------------------------------------------------ platform layer
// Everything a handle needs from the HIP runtime lives in its PlatCtx: device ordinal, launch stream, events, the helper mailbox, the
// host-mapped cancel word.  Handles are independent — two pools on two GPUs in one process, one thread per handle (include/armada_sched.h).
// Every ABI entry starts with plat_enter(handle context): hipSetDevice for the calling thread (the current device is thread-local in HIP,
// and a goroutine may run on any OS thread) and the thread-local pointer the plat_* helpers below work on.
struct HelpBox;
struct PlatCtx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, fitEv0 = nullptr, fitEv1 = nullptr;
  HelpBox* helpBox = nullptr;
  int helpers = -1, cus = 0, wallClockKHz = 100000;
  float lastControlMs = 0.f, lastFitMs = 0.f;
  int lastControlLaunches = 0;
  int32_t* progress = nullptr;      // ASCHED_PROGRESS=1: host-visible heartbeat of the round kernel
  int32_t* cancelHost = nullptr;    // host-mapped, coherent: written by the host (deadline / asched_cancel), polled by the round kernel
  int32_t* cancelDev = nullptr;
  double deadlineS = 0;             // maxSchedulingDuration for every following round launch; 0 = none
  bool inRound = false;             // between plat_round_begin / plat_round_end: the deadline runs from the begin, the cancel word is consumed at the end
  std::chrono::steady_clock::time_point roundT0;
  hipEvent_t rEv0 = nullptr, rEv1 = nullptr;
  float roundTotalMs = 0.f, roundControlMs = 0.f; int roundLaunches = 0;
  int32_t* cmpScratch = nullptr; size_t cmpScratchInts = 0;   // block counts + total of the grid-wide compaction
  int optIndexN = -1, optIndexM = -1;   // sizes the optimiser's node -> jobs index in the scratch was built for (asched_host.inc decides when it may be reused)
  void* fitScratch = nullptr; size_t fitScratchBytes = 0;   // keys + shape list of a fit batch
  void* optSel = nullptr; size_t optSelBytes = 0;   // block partials + result of the device-side candidate selection
  void* optScratch = nullptr; size_t optScratchBytes = 0;     // node -> jobs index, queue costs and per-node scores of the fairness optimiser, kept across calls
  std::string err;
  bool failed = false;              // sticky: an allocation / copy / memset failed since the last plat_take_failure()
  // the handle's communicator (asched_comm_init: RCCL over xGMI; asched_comm_init_external: the caller's transport)
  ncclComm_t comm = nullptr; int commRank = 0, commWorld = 1;
  unsigned long long* xArea = nullptr; unsigned long long** xPeerTable = nullptr; bool xDirect = false;   // GPU-to-GPU exchange of sharded passes (asched_shard_area / asched_shard_peers)
  hipStream_t xStream = nullptr; long long* xBuf = nullptr;   // sharded wide passes (dev.h shardWorld) over RCCL: the exchanged words' all-reduce runs here, beside the persistent kernel
  long lastShardExchanges = 0;
  asched_allreduce_fn extFn = nullptr; void* extCtx = nullptr;
};
static thread_local PlatCtx* t_ctx = nullptr;
static std::string g_noCtxErr;

static bool hipOk(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  if (t_ctx) { t_ctx->err = m; t_ctx->failed = true; } else g_noCtxErr = m;
  return false;
}
static const char* plat_last_error() { return t_ctx ? t_ctx->err.c_str() : g_noCtxErr.c_str(); }
// true (once) when an upload / download / memset / allocation failed since the last call: input-build entry points return ASCHED_ERR_DEVICE
static bool plat_take_failure() { if (!t_ctx) return true; bool f = t_ctx->failed; t_ctx->failed = false; return f; }
static void plat_enter(PlatCtx* c) { t_ctx = c; if (c) (void)hipSetDevice(c->device); }
static PlatCtx* plat_open(std::string& err, int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { err = "no HIP device: libarmada_sched.so is the gfx950 implementation and has no CPU path"; return nullptr; }
  if (device >= n) { err = "device ordinal out of range"; return nullptr; }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) { err = "hipGetDevice failed"; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) { err = "hipGetDeviceProperties failed"; return nullptr; }
  if (std::string(p.gcnArchName).find("gfx950") == std::string::npos) { err = std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only"; return nullptr; }
  auto* c = new PlatCtx();
  c->device = device;
  c->cus = p.multiProcessorCount;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wallClockKHz = khz;
  bool ok = hipStreamCreate(&c->stream) == hipSuccess && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
            hipEventCreate(&c->fitEv0) == hipSuccess && hipEventCreate(&c->fitEv1) == hipSuccess && hipEventCreate(&c->rEv0) == hipSuccess && hipEventCreate(&c->rEv1) == hipSuccess;
  // the mailbox is written from both sides across XCDs: it must not live in an XCD-private L2 -> fine-grained (uncached, device-coherent) memory
  ok = ok && hipExtMallocWithFlags((void**)&c->helpBox, sizeof(HelpBox), hipDeviceMallocFinegrained) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->cancelHost, 256, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;   // [0] the cancel word; from byte 64: the exchange words of sharded passes (dev.h XCHG_WORD0)
  if (ok) { memset(c->cancelHost, 0, 256); ok = hipHostGetDevicePointer((void**)&c->cancelDev, c->cancelHost, 0) == hipSuccess; }
  if (!ok) { err = "HIP resource creation failed (stream / events / mailbox / cancel word)"; delete c; return nullptr; }
  // helper workgroups of a round launch: one per CU, an eighth of the device by default — measured flat between 15 and 63 (ASCHED_HELPERS overrides; 0 = none)
  c->helpers = c->cus >= 16 ? c->cus / 8 - 1 : 0;
  if (const char* e = getenv("ASCHED_HELPERS")) c->helpers = atoi(e);
  if (c->helpers > c->cus - 1) c->helpers = c->cus - 1;
  if (c->helpers < 0) c->helpers = 0;
  if (getenv("ASCHED_PROGRESS")) {
    if (hipHostMalloc((void**)&c->progress, 64 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) c->progress = nullptr;
    if (c->progress) for (int i = 0; i < 64; i++) c->progress[i] = 0;
  }
  t_ctx = c;
  return c;
}
// ---- RCCL, bound at run time.  dlopen by soname: when the process already holds an RCCL (torch bundles one and loads it before this library in the Python
// harness) the loader hands back THAT copy — one RCCL per process, on the HIP runtime the process already uses; a Go scheduler gets /opt/rocm/lib's.
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclGetErrorString) errorString = nullptr;
};
static RcclApi* rcclApi(std::string& err) {
  static RcclApi api; static bool tried = false; static std::string why;
  if (!tried) {
    tried = true;
    const char* names[] = {getenv("ASCHED_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { if (!n || !*n) continue; api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.lib) break; why = dlerror(); }
    if (api.lib) {
      api.getUniqueId = (decltype(api.getUniqueId))dlsym(api.lib, "ncclGetUniqueId");
      api.commInitRank = (decltype(api.commInitRank))dlsym(api.lib, "ncclCommInitRank");
      api.commDestroy = (decltype(api.commDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.allReduce = (decltype(api.allReduce))dlsym(api.lib, "ncclAllReduce");
      api.errorString = (decltype(api.errorString))dlsym(api.lib, "ncclGetErrorString");
      if (!api.getUniqueId || !api.commInitRank || !api.commDestroy || !api.allReduce) { why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce"; dlclose(api.lib); api.lib = nullptr; }
    }
  }
  if (!api.lib) { err = "RCCL is not available: " + why; return nullptr; }
  return &api;
}
static bool rcclOk(RcclApi* a, ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return true;
  std::string m = std::string(what) + ": " + (a->errorString ? a->errorString(r) : "RCCL error");
  if (t_ctx) { t_ctx->err = m; t_ctx->failed = true; } else g_noCtxErr = m;
  return false;
}
static int plat_comm_unique_id(char* out128) {
  std::string err; RcclApi* a = rcclApi(err);
  if (!a) { g_noCtxErr = err; if (t_ctx) t_ctx->err = err; return -1; }
  ncclUniqueId id;
  static_assert(sizeof(id) == 128, "asched_unique_id carries an ncclUniqueId");
  if (!rcclOk(a, a->getUniqueId(&id), "ncclGetUniqueId")) return -1;
  memcpy(out128, &id, sizeof id);
  return 0;
}
static void plat_comm_destroy_ctx(PlatCtx* c) {
  if (c->comm) { std::string err; if (RcclApi* a = rcclApi(err)) (void)a->commDestroy(c->comm); c->comm = nullptr; }
  c->extFn = nullptr; c->extCtx = nullptr; c->commRank = 0; c->commWorld = 1;
}
static int plat_comm_init(const char* id128, int rank, int world) {
  PlatCtx* c = t_ctx;
  std::string err; RcclApi* a = rcclApi(err);
  if (!a) { c->err = err; return -1; }
  plat_comm_destroy_ctx(c);
  ncclUniqueId id; memcpy(&id, id128, sizeof id);
  if (!rcclOk(a, a->commInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) { c->comm = nullptr; return -1; }
  c->commRank = rank; c->commWorld = world;
  return 0;
}
static int plat_comm_init_external(asched_allreduce_fn fn, void* ctx, int rank, int world) {
  PlatCtx* c = t_ctx;
  plat_comm_destroy_ctx(c);
  c->extFn = fn; c->extCtx = ctx; c->commRank = rank; c->commWorld = world;
  return 0;
}
static void plat_comm_destroy() { if (t_ctx) { (void)hipStreamSynchronize(t_ctx->stream); plat_comm_destroy_ctx(t_ctx); } }
static void plat_comm_info(int* rank, int* world) { *rank = t_ctx ? t_ctx->commRank : 0; *world = t_ctx ? t_ctx->commWorld : 1; }
static bool plat_comm_live() { return t_ctx && (t_ctx->comm || t_ctx->extFn); }
// in-place all-reduce of `count` int64 words in memory of this handle's GPU, on the handle's stream: behind whatever produced the words there, in front of
// whatever the caller enqueues next.  op: 0 SUM, 1 MIN, 2 MAX.
static int plat_allreduce(long long* dbuf, size_t count, int op) {
  PlatCtx* c = t_ctx;
  if (c->commWorld <= 1 && !c->comm && !c->extFn) return 0;
  if (c->comm) {
    std::string err; RcclApi* a = rcclApi(err);
    if (!a) { c->err = err; return -1; }
    ncclRedOp_t o = op == 0 ? ncclSum : op == 1 ? ncclMin : ncclMax;
    if (!rcclOk(a, a->allReduce(dbuf, dbuf, count, ncclInt64, o, c->comm, c->stream), "ncclAllReduce")) return -1;
    return 0;
  }
  if (!hipOk(hipStreamSynchronize(c->stream), "all-reduce (external transport): stream sync")) return -1;   // the transport sees finished words and an idle stream
  if (c->extFn(c->extCtx, dbuf, (int64_t)count, op) != 0) { c->err = "the external all-reduce transport failed"; return -1; }
  return 0;
}
// all-reduce MIN of a few UNSIGNED 64-bit words that live in HOST memory, while the handle's stream is busy with the persistent kernel that waits for the answer
// (shardReduce): RCCL on a side stream through a device staging buffer, or the caller's transport with ASCHED_ALLREDUCE_HOST_WORDS in `op` (the words are host memory: reduce
// them where they are, do not synchronise the device).  The collectives compare int64: the sign bit is flipped around them.
static int plat_allreduce_host_min(unsigned long long* w, int count) {
  PlatCtx* c = t_ctx;
  long long v[8];
  if (count > 8) return -1;
  for (int i = 0; i < count; i++) v[i] = (long long)(w[i] ^ 0x8000000000000000ull);
  if (c->comm) {
    std::string err; RcclApi* a = rcclApi(err);
    if (!a) { c->err = err; return -1; }
    if (!c->xStream && !hipOk(hipStreamCreateWithFlags(&c->xStream, hipStreamNonBlocking), "hipStreamCreate (exchange)")) return -1;
    if (!c->xBuf && !hipOk(hipMalloc((void**)&c->xBuf, 8 * sizeof(long long)), "hipMalloc (exchange)")) return -1;
    if (!hipOk(hipMemcpyAsync(c->xBuf, v, count * sizeof(long long), hipMemcpyHostToDevice, c->xStream), "exchange h2d")) return -1;
    if (!rcclOk(a, a->allReduce(c->xBuf, c->xBuf, count, ncclInt64, ncclMin, c->comm, c->xStream), "ncclAllReduce (exchange)")) return -1;
    if (!hipOk(hipMemcpyAsync(v, c->xBuf, count * sizeof(long long), hipMemcpyDeviceToHost, c->xStream), "exchange d2h") || !hipOk(hipStreamSynchronize(c->xStream), "exchange sync")) return -1;
  } else if (c->extFn) {
    if (c->extFn(c->extCtx, v, (int64_t)count, 1 | ASCHED_ALLREDUCE_HOST_WORDS) != 0) { c->err = "the external all-reduce transport failed"; return -1; }
  }
  for (int i = 0; i < count; i++) w[i] = (unsigned long long)v[i] ^ 0x8000000000000000ull;
  return 0;
}
static long plat_last_shard_exchanges() { return t_ctx ? t_ctx->lastShardExchanges : 0; }
#define XCHG_AREA_BYTES (64 + 2 * 256 * 32)
// this handle's exchange area (device memory, fine-grained where the runtime offers it: remote GPUs store into it) and its IPC handle for replicas in other processes
static int plat_shard_area(void** ptr, char* ipc64) {
  PlatCtx* c = t_ctx;
  if (!c->xArea) {
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, XCHG_AREA_BYTES, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); if (!hipOk(hipMalloc(&p, XCHG_AREA_BYTES), "hipMalloc (exchange area)")) return -1; }
    if (!hipOk(hipMemset(p, 0, XCHG_AREA_BYTES), "exchange area reset")) { (void)hipFree(p); return -1; }
    c->xArea = (unsigned long long*)p;
  }
  *ptr = c->xArea;
  if (ipc64) {
    hipIpcMemHandle_t h; memset(&h, 0, sizeof h);
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle");
    memset(ipc64, 0, 64);
    if (hipIpcGetMemHandle(&h, c->xArea) == hipSuccess) memcpy(ipc64, &h, sizeof h); else (void)hipGetLastError();   // (all zero: not exportable here; in-process peers still work)
  }
  return 0;
}
static int plat_shard_open(const char* ipc64, void** out) {
  hipIpcMemHandle_t h; memcpy(&h, ipc64, sizeof h);
  return hipOk(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle (exchange area)") ? 0 : -1;
}
static int plat_shard_peers(void* const* areas, int world, int rank) {
  PlatCtx* c = t_ctx;
  if (!areas) { c->xDirect = false; return 0; }
  if (!c->xArea || areas[rank] != (void*)c->xArea) { c->err = "shard_peers: areas[rank] must be this handle's own area (asched_shard_area)"; return -1; }
  for (int r = 0; r < world; r++) {   // a peer area on another GPU of this process: let this GPU store into it
    hipPointerAttribute_t at; memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, areas[r]) == hipSuccess && at.device != c->device) { hipError_t e = hipDeviceEnablePeerAccess(at.device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { hipOk(e, "hipDeviceEnablePeerAccess"); return -1; } (void)hipGetLastError(); }
    else (void)hipGetLastError();
  }
  if (!c->xPeerTable && !hipOk(hipMalloc((void**)&c->xPeerTable, 256 * sizeof(void*)), "hipMalloc (peer table)")) return -1;
  if (!hipOk(hipMemcpy(c->xPeerTable, areas, world * sizeof(void*), hipMemcpyHostToDevice), "peer table upload")) return -1;
  c->xDirect = true;
  return 0;
}
static void plat_close(PlatCtx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->xStream) (void)hipStreamDestroy(c->xStream);
  if (c->xBuf) (void)hipFree(c->xBuf);
  if (c->xArea) (void)hipFree(c->xArea);
  if (c->xPeerTable) (void)hipFree(c->xPeerTable);
  plat_comm_destroy_ctx(c);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  for (hipEvent_t e : {c->ev0, c->ev1, c->fitEv0, c->fitEv1, c->rEv0, c->rEv1}) if (e) (void)hipEventDestroy(e);
  if (c->helpBox) (void)hipFree(c->helpBox);
  if (c->cmpScratch) (void)hipFree(c->cmpScratch);
  if (c->optScratch) (void)hipFree(c->optScratch);
  if (c->optSel) (void)hipFree(c->optSel);
  if (c->fitScratch) (void)hipFree(c->fitScratch);
  if (c->cancelHost) (void)hipHostFree(c->cancelHost);
  if (c->progress) (void)hipHostFree(c->progress);
  if (t_ctx == c) t_ctx = nullptr;
  delete c;
}
static int plat_wall_clock_khz() { return t_ctx ? t_ctx->wallClockKHz : 100000; }
static void plat_set_deadline(double s) { if (t_ctx) t_ctx->deadlineS = s > 0 ? s : 0; }
static void plat_cancel(PlatCtx* c) { if (c && c->cancelHost) __atomic_store_n(c->cancelHost, 1, __ATOMIC_RELEASE); }  // any thread: a plain store to host memory
static void plat_cancel_clear(PlatCtx* c) { if (c && c->cancelHost) __atomic_store_n(c->cancelHost, 0, __ATOMIC_RELEASE); }
static void* plat_malloc(size_t n) { void* p = nullptr; if (!hipOk(hipMalloc(&p, n), "hipMalloc")) return nullptr; return p; }
static void plat_free(void* p) { if (p) (void)hipFree(p); }
static void plat_memset(void* p, int v, size_t n) { if (!p) { hipOk(hipErrorInvalidValue, "memset of a failed allocation"); return; } hipOk(hipMemsetAsync(p, v, n, t_ctx->stream), "hipMemsetAsync"); }
static void plat_h2d(void* d, const void* s, size_t n) {
  if (!d) { hipOk(hipErrorInvalidValue, "upload into a failed allocation"); return; }
  if (hipOk(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, t_ctx->stream), "hipMemcpyAsync (h2d)")) hipOk(hipStreamSynchronize(t_ctx->stream), "h2d sync");
}
// pinned host memory + asynchronous downloads on the handle's stream (the round's result arrays: one wait for all of them)
static void* plat_pinned(size_t n) { void* p = nullptr; if (!hipOk(hipHostMalloc(&p, n, hipHostMallocDefault), "hipHostMalloc")) return nullptr; return p; }
static void plat_pinned_free(void* p) { if (p) (void)hipHostFree(p); }
static void plat_d2h_async(void* d, const void* s, size_t n) {
  if (!s) { hipOk(hipErrorInvalidValue, "download from a failed allocation"); std::memset(d, 0, n); return; }
  hipOk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, t_ctx->stream), "hipMemcpyAsync (d2h)");
}
static void plat_sync() { hipOk(hipStreamSynchronize(t_ctx->stream), "stream sync"); }
static void plat_d2h(void* d, const void* s, size_t n) {
  if (!s) { hipOk(hipErrorInvalidValue, "download from a failed allocation"); std::memset(d, 0, n); return; }
  if (hipOk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, t_ctx->stream), "hipMemcpyAsync (d2h)")) hipOk(hipStreamSynchronize(t_ctx->stream), "d2h sync");
}

// device time of the last control-kernel launch (HIP events recorded on the launch stream) — bench.py's roofline input
static double plat_last_control_ms() { return t_ctx ? (double)t_ctx->lastControlMs : 0.0; }
static int plat_last_control_launches() { return t_ctx ? t_ctx->lastControlLaunches : 0; }

extern "C" int asched_internal_aux_launch(const Dev* dev, int cmd, hipStream_t stream, void* helpBox, const MktDev* mk);  // armada_sched_aux.hip
extern "C" int asched_internal_ft_launch(const Dev* dev, int cmd, hipStream_t stream, void* helpBox, int H);               // armada_sched_ft.hip
extern "C" int asched_internal_wk_launch(const Dev* dev, int cmd, hipStream_t stream, void* helpBox, int H, const MktDev* mk);               // armada_sched_wk.hip: handles with a two-word order key
extern "C" int asched_internal_wk_bulk(const Dev* dev, int kind, int n, int grid, hipStream_t stream);
extern "C" int asched_internal_wk_fit_batch(const Dev* dev, const int32_t* shapes, int ns, int level, unsigned long long* out, int tiles, int ysplit, hipStream_t stream);
// market-driven rounds: the market state the next auxiliary launch of this thread's handle runs with (asched_host.inc sets it around CMD_MARKET_ROUND)
static thread_local const MktDev* t_mkt = nullptr;
static void plat_set_market_dev(const MktDev* m) { t_mkt = m; }
static int plat_run_control(Dev& dev, int cmd) {
  PlatCtx* c = t_ctx;
  if (c->failed) return -1;  // an earlier upload failed: the kernel would read unset pointers
  static_assert(sizeof(HelpBox) == 256 + HELP_MAX * 32, "mailbox allocation");
  bool isRound = cmd == CMD_ROUND || cmd == CMD_QUEUES_ONLY || cmd == CMD_PASS1 || cmd == CMD_PASS2;
  int H = isRound ? c->helpers : 0;
  // the wide queries of the generic path (plane scan, fair-share evaluation) are one node per thread: from ~50k nodes on half of the CUs pay off (measured at 100k
  // nodes x 1M jobs 95% occupied: 29.5 -> 23.0 s per round with 127 helpers, 24.6 s with 255; flat between 15 and 63 at 20k nodes)
  if (isRound && !getenv("ASCHED_HELPERS") && dev.cfg.N >= 50000 && c->cus >= 128) H = c->cus / 2 - 1;
  // more than QCAPF queues (round_wide.h): the merge of a wide run is a bulk rank over all queues' entries — work for every workgroup the launch can bring
  if (isRound && !getenv("ASCHED_HELPERS") && dev.f.iterOk == 2 && c->cus >= 128) H = c->cus / 2 - 1;
  dev.progress = ((cmd == CMD_ROUND || cmd == CMD_PASS1 || cmd == CMD_PASS2) && c->progress) ? c->progress : nullptr;
  dev.cancel = c->cancelDev;
  if (!hipOk(hipMemsetAsync(c->helpBox, 0, sizeof(HelpBox), c->stream), "help box reset")) return -1;
  (void)hipEventRecord(c->ev0, c->stream);
  const bool shard = dev.cfg.shardWorld > 1;
  volatile unsigned long long* X = (volatile unsigned long long*)c->cancelHost;
  const bool direct = shard && c->xDirect;   // GPU-to-GPU exchange (asched_shard_peers): the kernel finds the peer table's address in the block; no proxy
  if (shard) { for (int i = 0; i < 6; i++) X[XCHG_WORD0 + i] = 0; X[XCHG_WORD0 + 6] = direct ? (unsigned long long)c->xPeerTable : 0; X[XCHG_WORD0 + 7] = 0; __atomic_thread_fence(__ATOMIC_SEQ_CST); if (!c->inRound) c->lastShardExchanges = 0; }
  if (dev.cfg.keyWords == 2 || shard) {   // a two-word order key, or wide passes sharded across GPUs: every control command on the kernel built for them (armada_sched_wk.hip)
    if (asched_internal_wk_launch(&dev, cmd, c->stream, c->helpBox, H, t_mkt)) { c->err = "k_control_wk launch failed"; return -1; }
  } else if (cmd >= CMD_AUX_FIRST) {  // submit-check commands: their kernel lives in its own code object (armada_sched_aux.hip)
    if (asched_internal_aux_launch(&dev, cmd, c->stream, c->helpBox, t_mkt)) { c->err = "k_control_aux launch failed"; return -1; }
  } else if (isRound && dev.ftT != nullptr) {   // this handle carries a fair-share threshold table: the round kernel that uses it (armada_sched_ft.hip)
    if (asched_internal_ft_launch(&dev, cmd, c->stream, c->helpBox, H)) { c->err = "k_control_ft launch failed"; return -1; }
  } else
  hipLaunchKernelGGL(k_control, dim3(1 + H), dim3(CTL_THREADS), 0, c->stream, dev, cmd, c->helpBox, H);
  (void)hipEventRecord(c->ev1, c->stream);
  if (!hipOk(hipGetLastError(), "k_control launch")) return -1;
  static const double safetyS = [] { const char* e = getenv("ASCHED_SAFETY_DEADLINE_S"); return e ? atof(e) : 0.0; }();   // test / measurement runs of new builds: no launch outlives this
  double deadlineS = c->deadlineS > 0 ? c->deadlineS : safetyS;
  if (shard && !direct) {
    // the exchange proxy of sharded passes: the kernel posts (generation, two words), this thread runs the all-reduce on the handle's communicator and answers (dev.h XCHG_WORD0)
    auto t0 = c->inRound ? c->roundT0 : std::chrono::steady_clock::now();
    unsigned long long served = 0; unsigned int idle = 0; bool failed = false;
    for (;;) {
      // (the stream is asked only now and then: a query costs microseconds of the runtime's time on the path of every exchange; the request word is a load of host memory)
      if ((idle & 63) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) break;
      unsigned long long g = __atomic_load_n(&X[XCHG_WORD0], __ATOMIC_ACQUIRE);
      if (g != served && !failed) {
        unsigned long long w[2] = {X[XCHG_WORD0 + 1], X[XCHG_WORD0 + 2]};
        if (plat_allreduce_host_min(w, 2)) { failed = true; plat_cancel(c); continue; }   // (the kernel's wait ends on the cancel word: ASCHED_ERR_TIMEOUT 903, reported as a device error below)
        X[XCHG_WORD0 + 4] = w[0]; X[XCHG_WORD0 + 5] = w[1];
        __atomic_store_n(&X[XCHG_WORD0 + 3], g, __ATOMIC_RELEASE);
        served = g; c->lastShardExchanges++; idle = 1;
        continue;
      }
      if ((++idle & 0xfff) == 0 && isRound && deadlineS > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadlineS) plat_cancel(c);
    }
    if (failed) { (void)hipStreamSynchronize(c->stream); if (!c->inRound) __atomic_store_n(c->cancelHost, 0, __ATOMIC_RELEASE); return -1; }
  } else
  if (dev.progress || (isRound && deadlineS > 0)) {
    // hard timeout (scheduling_algo.go:130-134): the kernel polls the cancel word; the host sets it when the deadline passes
    auto t0 = c->inRound ? c->roundT0 : std::chrono::steady_clock::now();
    int ticks = 0;
    volatile int32_t* progress = c->progress;
    while (hipStreamQuery(c->stream) == hipErrorNotReady) {
      usleep(dev.progress ? 100000 : 100);
      double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (isRound && deadlineS > 0 && el > deadlineS) plat_cancel(c);
      if (dev.progress && ++ticks % 10 == 0) { fprintf(stderr, "[asched progress] t=%ds iterations=%d generic=%d phase=%d op=%d ops=%d | wait: done=%d H=%d gen=%d box.gen=%d box.op=%d | helpers:", ticks / 10, progress[0], progress[4], progress[1], progress[2], progress[3], progress[5], progress[6], progress[7], progress[8], progress[9]); for (int i = 17; i < 56; i++) fprintf(stderr, " %x", progress[i]); fprintf(stderr, "\n"); }
    }
  }
  if (!hipOk(hipStreamSynchronize(c->stream), "k_control")) return -1;
  if (direct) c->lastShardExchanges += (long)X[XCHG_WORD0 + 7];   // (written by the kernel at its end: the GPU-to-GPU exchanges of this launch)
  if (isRound && !c->inRound) __atomic_store_n(c->cancelHost, 0, __ATOMIC_RELEASE);  // a cancel request is consumed by the round it hit (or the next one, if it came between rounds)
  (void)hipEventElapsedTime(&c->lastControlMs, c->ev0, c->ev1);
  c->lastControlLaunches = 1;
  if (c->inRound) { c->roundControlMs += c->lastControlMs; c->roundLaunches++; }
  return 0;
}

// ---- the split round: grid-wide kernels between the persistent passes, all on the handle's stream (no host sync except where a count is needed)
static void plat_round_begin() {
  PlatCtx* c = t_ctx;
  c->inRound = true; c->roundT0 = std::chrono::steady_clock::now(); c->roundControlMs = 0.f; c->roundLaunches = 0; c->lastShardExchanges = 0;
  (void)hipEventRecord(c->rEv0, c->stream);
}
static void plat_round_end() {
  PlatCtx* c = t_ctx;
  (void)hipEventRecord(c->rEv1, c->stream);
  (void)hipStreamSynchronize(c->stream);
  (void)hipEventElapsedTime(&c->roundTotalMs, c->rEv0, c->rEv1);
  c->inRound = false;
  __atomic_store_n(c->cancelHost, 0, __ATOMIC_RELEASE);
}
static void plat_round_times(double* out) { PlatCtx* c = t_ctx; out[0] = c->roundTotalMs; out[1] = c->roundControlMs; out[2] = c->roundLaunches; }
static int bulkGrid(int n) { int b = (n + 255) / 256; int cap = (t_ctx->cus > 0 ? t_ctx->cus : 256) * 8; return b < 1 ? 1 : (b > cap ? cap : b); }
static int plat_bulk(Dev& d, int kind, int n) {
  if (n <= 0) return 0;
  if (d.cfg.keyWords == 2) { if (asched_internal_wk_bulk(&d, kind, n, bulkGrid(n), t_ctx->stream)) { t_ctx->err = "k_bulk_wk launch failed"; return -1; } }
  else
  hipLaunchKernelGGL(k_bulk, dim3(bulkGrid(n)), dim3(256), 0, t_ctx->stream, d, kind, n);
  t_ctx->roundLaunches++;
  return hipOk(hipGetLastError(), "k_bulk launch") ? 0 : -1;
}
static int plat_small(Dev& d, int what, int arg) {
  hipLaunchKernelGGL(k_round_small, dim3(1), dim3(64), 0, t_ctx->stream, d, what, arg);
  t_ctx->roundLaunches++;
  return hipOk(hipGetLastError(), "k_round_small launch") ? 0 : -1;
}
static int plat_agg(Dev& d, int queued, int total) {
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_agg, dim3(bulkGrid(total)), dim3(256), 0, t_ctx->stream, d, queued, total);
  return hipOk(hipGetLastError(), "k_agg launch") ? 0 : -1;
}
static int plat_evict_apply(Dev& d, int phase3, int total) {
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_evict_apply, dim3(bulkGrid(total)), dim3(256), 0, t_ctx->stream, d, phase3, total);
  t_ctx->roundLaunches++;
  return hipOk(hipGetLastError(), "k_evict_apply launch") ? 0 : -1;
}
// fairness optimiser: every node scored for one job (k_opt_score), scores downloaded; detailNode >= 0: that node's preemption list as well
static float g_lastOptMs = 0.f;
static int plat_opt_score(Dev& d, const OptArgs& a, std::vector<OptNodeOut>& scores, double* jobCost, int detailNode, OptNodeOut* detail, std::vector<int32_t>* pre, bool detailOnly = false,
                          bool reuseIndex = false) {   // detailOnly: the index and scores of the previous call are still in the scratch; reuseIndex: so is the node -> jobs index (nothing was bound since)
  PlatCtx* c = t_ctx;
  int N = d.cfg.N, M = d.cfg.M, Q = d.cfg.Q;
  // one allocation, carved: [scores N+1][queue costs Q+1][cnt N+1][off N+2][cursor N+1][jobs M][pre OPT_MAXJ]
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t bOut = up(sizeof(OptNodeOut) * (size_t)(N + 1)), bQ = up(sizeof(double) * (size_t)(Q + 1)), bN = up(sizeof(int32_t) * (size_t)(N + 2)), bM = up(sizeof(int32_t) * 2 * (size_t)std::max(M, 1)), bP = up(sizeof(int32_t) * 64);
  size_t need = bOut + bQ + 3 * bN + bM + bP;
  bool ok = true;
  if (c->optScratchBytes < need) {
    if (c->optScratch) (void)hipFree(c->optScratch);
    c->optScratch = nullptr; c->optScratchBytes = 0; c->optIndexN = c->optIndexM = -1;
    ok = hipOk(hipMalloc(&c->optScratch, need), "optimiser scratch");
    if (ok) c->optScratchBytes = need;
  }
  char* base = (char*)c->optScratch;
  OptNodeOut* out = (OptNodeOut*)base; double* qCost = (double*)(base + bOut);
  int32_t* cnt = (int32_t*)(base + bOut + bQ); int32_t* off = (int32_t*)(base + bOut + bQ + bN); int32_t* cursor = (int32_t*)(base + bOut + bQ + 2 * bN);
  int32_t* jobs = (int32_t*)(base + bOut + bQ + 3 * bN); int32_t* dPre = (int32_t*)(base + bOut + bQ + 3 * bN + bM);
  // the preemption list of one node: the private entry list when its job count fits, an HBM list otherwise
  auto runDetail = [&]() -> bool {
    int32_t o2[2] = {0, 0};
    if (!hipOk(hipMemcpy(o2, off + detailNode, sizeof o2, hipMemcpyDeviceToHost), "opt detail")) return false;
    int cnt = o2[1] - o2[0];
    static const bool perThread = [] { const char* e = getenv("ASCHED_OPT_PER_THREAD"); return e && e[0] == '1'; }();
    if (cnt <= (perThread ? OPT_MAXJ : 64)) {
      pre->assign(64, -1);
      if (perThread) hipLaunchKernelGGL(k_opt_detail, dim3(1), dim3(64), 0, c->stream, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, detailNode, out + N, dPre);
      else hipLaunchKernelGGL(k_opt_detail_wave, dim3(1), dim3(64), 0, c->stream, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, detailNode, out + N, dPre);
      return hipOk(hipGetLastError(), "optimiser launch") && hipOk(hipMemcpyAsync(detail, out + N, sizeof(OptNodeOut), hipMemcpyDeviceToHost, c->stream), "opt detail") &&
             hipOk(hipMemcpyAsync(pre->data(), dPre, sizeof(int32_t) * 64, hipMemcpyDeviceToHost, c->stream), "opt detail") && hipOk(hipStreamSynchronize(c->stream), "optimiser kernels");
    }
    pre->assign((size_t)cnt, -1);
    OptEntry* es = nullptr; int32_t* dp = nullptr;
    bool k = hipOk(hipMalloc(&es, sizeof(OptEntry) * (size_t)cnt), "optimiser scratch") && hipOk(hipMalloc(&dp, sizeof(int32_t) * (size_t)cnt), "optimiser scratch");
    if (k) {
      hipLaunchKernelGGL(k_opt_detail_big, dim3(1), dim3(64), 0, c->stream, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, detailNode, out + N, dp, es);
      k = hipOk(hipGetLastError(), "optimiser launch") && hipOk(hipMemcpyAsync(detail, out + N, sizeof(OptNodeOut), hipMemcpyDeviceToHost, c->stream), "opt detail") &&
          hipOk(hipMemcpyAsync(pre->data(), dp, sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToHost, c->stream), "opt detail") && hipOk(hipStreamSynchronize(c->stream), "optimiser kernels");
    }
    (void)hipFree(es); (void)hipFree(dp);
    return k;
  };
  if (ok && detailOnly) return runDetail() ? 0 : -1;
  if (ok) {
    if (!(reuseIndex && c->optIndexN == N && c->optIndexM == M)) {
      (void)hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(N + 1), c->stream);
      hipLaunchKernelGGL(k_opt_count, dim3(bulkGrid(M)), dim3(256), 0, c->stream, d, cnt);
      hipLaunchKernelGGL(k_opt_scan, dim3(1), dim3(1024), 0, c->stream, (const int32_t*)cnt, off, cursor, N);
      hipLaunchKernelGGL(k_opt_scatter, dim3(bulkGrid(M)), dim3(256), 0, c->stream, d, cursor, jobs);
      c->optIndexN = N; c->optIndexM = M;
    }
    hipLaunchKernelGGL(k_opt_qcost, dim3((Q + 1 + 63) / 64), dim3(64), 0, c->stream, d, a.job, qCost);
    (void)hipEventRecord(c->fitEv0, c->stream);
    static const bool perThread = [] { const char* e = getenv("ASCHED_OPT_PER_THREAD"); return e && e[0] == '1'; }();   // A/B: the one-node-per-thread kernel of rounds 2-3
    if (perThread) hipLaunchKernelGGL(k_opt_score, dim3((N + 127) / 128), dim3(128), 0, c->stream, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, out);
    else hipLaunchKernelGGL(k_opt_score_wave, dim3((N + 3) / 4), dim3(256), 0, c->stream, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, out);
    (void)hipEventRecord(c->fitEv1, c->stream);
    ok = hipOk(hipGetLastError(), "optimiser launch") && hipOk(hipStreamSynchronize(c->stream), "optimiser kernels");
    (void)hipEventElapsedTime(&g_lastOptMs, c->fitEv0, c->fitEv1);
  }
  if (ok) {
    scores.resize(N);
    if (N) ok = hipOk(hipMemcpy(scores.data(), out, sizeof(OptNodeOut) * (size_t)N, hipMemcpyDeviceToHost), "opt scores");
    if (ok) ok = hipOk(hipMemcpy(jobCost, qCost + Q, sizeof(double), hipMemcpyDeviceToHost), "opt job cost");
  }
  if (ok) {   // nodes whose candidates did not fit the private list: scored again with a list in HBM (one thread per such node; they are few)
    std::vector<int32_t> big;
    for (int n = 0; n < N; n++) if (scores[n].scheduled < 0) big.push_back(n);
    if (!big.empty()) {
      std::vector<int32_t> hOff((size_t)N + 2);
      ok = hipOk(hipMemcpy(hOff.data(), off, sizeof(int32_t) * (size_t)(N + 1), hipMemcpyDeviceToHost), "opt index");
      std::vector<long long> eOff(big.size());
      long long total = 0;
      for (size_t i = 0; i < big.size(); i++) { eOff[i] = total; total += hOff[big[i] + 1] - hOff[big[i]]; }
      OptEntry* es = nullptr; int32_t* dn = nullptr; long long* de = nullptr;
      ok = ok && hipOk(hipMalloc(&es, sizeof(OptEntry) * (size_t)std::max<long long>(total, 1)), "optimiser scratch") && hipOk(hipMalloc(&dn, sizeof(int32_t) * big.size()), "optimiser scratch") &&
           hipOk(hipMalloc(&de, sizeof(long long) * big.size()), "optimiser scratch");
      if (ok) {
        (void)hipMemcpyAsync(dn, big.data(), sizeof(int32_t) * big.size(), hipMemcpyHostToDevice, c->stream);
        (void)hipMemcpyAsync(de, eOff.data(), sizeof(long long) * big.size(), hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(k_opt_score_big, dim3(((int)big.size() + 63) / 64), dim3(64), 0, c->stream, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, (const int32_t*)dn, (const long long*)de,
                           (int)big.size(), es, out);
        ok = hipOk(hipGetLastError(), "optimiser launch") && hipOk(hipStreamSynchronize(c->stream), "optimiser kernels");
        for (size_t i = 0; ok && i < big.size(); i++) ok = hipOk(hipMemcpy(&scores[big[i]], out + big[i], sizeof(OptNodeOut), hipMemcpyDeviceToHost), "opt scores");
      }
      (void)hipFree(es); (void)hipFree(dn); (void)hipFree(de);
    }
  }
  if (ok && detailNode >= 0) ok = runDetail();
  return ok ? 0 : -1;
}
static double plat_last_opt_ms() { return (double)g_lastOptMs; }
// asched_optimiser_schedule_job without per-node scores: index (when stale), queue costs, scores, selection and the selected node's victims as ONE stream-ordered sequence.
// Returns 1 when a node overflowed the wave kernel (the caller takes plat_opt_score's path), 0 on success, -1 on a device error.
static int plat_opt_select(Dev& d, const OptArgs& a, double minPct, bool reuseIndex, int32_t* node, int32_t* npre, double* cost, double* impact, std::vector<int32_t>* pre) {
  static const bool perThread = [] { const char* e = getenv("ASCHED_OPT_PER_THREAD"); return e && e[0] == '1'; }();
  if (perThread) return 1;   // A/B runs of the round-2 kernel take the host-side selection as well
  PlatCtx* c = t_ctx;
  int N = d.cfg.N, M = d.cfg.M, Q = d.cfg.Q;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t bOut = up(sizeof(OptNodeOut) * (size_t)(N + 1)), bQ = up(sizeof(double) * (size_t)(Q + 1)), bN = up(sizeof(int32_t) * (size_t)(N + 2)), bM = up(sizeof(int32_t) * 2 * (size_t)std::max(M, 1)), bP = up(sizeof(int32_t) * 64);
  size_t need = bOut + bQ + 3 * bN + bM + bP;
  if (c->optScratchBytes < need) {
    if (c->optScratch) (void)hipFree(c->optScratch);
    c->optScratch = nullptr; c->optScratchBytes = 0; c->optIndexN = c->optIndexM = -1;
    if (!hipOk(hipMalloc(&c->optScratch, need), "optimiser scratch")) return -1;
    c->optScratchBytes = need;
  }
  int nb = (N + 255) / 256;
  size_t selBytes = up(sizeof(OptSelKey) * (size_t)std::max(nb, 1)) + 256;
  if (c->optSelBytes < selBytes) {
    if (c->optSel) (void)hipFree(c->optSel);
    c->optSel = nullptr; c->optSelBytes = 0;
    if (!hipOk(hipMalloc(&c->optSel, selBytes), "optimiser selection scratch")) return -1;
    c->optSelBytes = selBytes;
  }
  char* base = (char*)c->optScratch;
  OptNodeOut* out = (OptNodeOut*)base; double* qCost = (double*)(base + bOut);
  int32_t* cnt = (int32_t*)(base + bOut + bQ); int32_t* off = (int32_t*)(base + bOut + bQ + bN); int32_t* cursor = (int32_t*)(base + bOut + bQ + 2 * bN);
  int32_t* jobs = (int32_t*)(base + bOut + bQ + 3 * bN); int32_t* dPre = (int32_t*)(base + bOut + bQ + 3 * bN + bM);
  OptSelKey* partial = (OptSelKey*)c->optSel; OptSel* dSel = (OptSel*)((char*)c->optSel + selBytes - 256); int32_t* dOver = (int32_t*)((char*)c->optSel + selBytes - 128);
  hipStream_t st = c->stream;
  if (!(reuseIndex && c->optIndexN == N && c->optIndexM == M)) {
    (void)hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(N + 1), st);
    hipLaunchKernelGGL(k_opt_count, dim3(bulkGrid(M)), dim3(256), 0, st, d, cnt);
    hipLaunchKernelGGL(k_opt_scan, dim3(1), dim3(1024), 0, st, (const int32_t*)cnt, off, cursor, N);
    hipLaunchKernelGGL(k_opt_scatter, dim3(bulkGrid(M)), dim3(256), 0, st, d, cursor, jobs);
    c->optIndexN = N; c->optIndexM = M;
  }
  (void)hipMemsetAsync(dOver, 0, sizeof(int32_t), st);
  hipLaunchKernelGGL(k_opt_qcost, dim3((Q + 1 + 63) / 64), dim3(64), 0, st, d, a.job, qCost);
  (void)hipEventRecord(c->fitEv0, st);
  hipLaunchKernelGGL(k_opt_score_wave, dim3((N + 3) / 4), dim3(256), 0, st, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, out);
  (void)hipEventRecord(c->fitEv1, st);
  hipLaunchKernelGGL(k_opt_select, dim3(std::max(nb, 1)), dim3(256), 0, st, d, (const OptNodeOut*)out, (const uint8_t*)nullptr, (const double*)(qCost + Q), minPct, partial, dOver);
  hipLaunchKernelGGL(k_opt_select_final, dim3(1), dim3(256), 0, st, (const OptSelKey*)partial, nb, (const int32_t*)dOver, dSel);
  hipLaunchKernelGGL(k_opt_detail_sel, dim3(1), dim3(64), 0, st, d, a, (const double*)qCost, (const int32_t*)off, (const int32_t*)jobs, dSel, out + N, dPre);
  OptSel hs; pre->assign(64, -1);
  bool ok = hipOk(hipGetLastError(), "optimiser launch") && hipOk(hipMemcpyAsync(&hs, dSel, sizeof hs, hipMemcpyDeviceToHost, st), "opt selection") &&
            hipOk(hipMemcpyAsync(pre->data(), dPre, sizeof(int32_t) * 64, hipMemcpyDeviceToHost, st), "opt victims") && hipOk(hipStreamSynchronize(st), "optimiser kernels");
  (void)hipEventElapsedTime(&g_lastOptMs, c->fitEv0, c->fitEv1);
  if (!ok) return -1;
  if (hs.overflow || hs.big) return 1;
  *node = hs.node; *npre = hs.node >= 0 ? hs.npre : 0; *cost = hs.node >= 0 ? hs.cost : 0; *impact = hs.node >= 0 ? hs.impact : 0;
  return 0;
}
// indicative pricer: every node priced for one job (k_price_score over the node -> jobs index of the current binding state); detailNode >= 0: that node's victims in order
static int plat_price_score(Dev& d, const PriceArgs& a, std::vector<PriceNodeOut>& scores, int detailNode, std::vector<int32_t>* pre) {
  PlatCtx* c = t_ctx;
  int N = d.cfg.N, M = d.cfg.M;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t bOut = up(sizeof(PriceNodeOut) * (size_t)(N + 1)), bN = up(sizeof(int32_t) * (size_t)(N + 2)), bM = up(sizeof(int32_t) * 2 * (size_t)std::max(M, 1)),
         bE = up(sizeof(PriceEntry) * 2 * (size_t)std::max(M, 1));
  char* base = nullptr;
  if (!hipOk(hipMalloc(&base, bOut + 3 * bN + 2 * bM + bE), "pricer scratch")) return -1;
  PriceNodeOut* out = (PriceNodeOut*)base;
  int32_t* cnt = (int32_t*)(base + bOut); int32_t* off = (int32_t*)(base + bOut + bN); int32_t* cursor = (int32_t*)(base + bOut + 2 * bN);
  int32_t* jobs = (int32_t*)(base + bOut + 3 * bN); int32_t* dPre = (int32_t*)(base + bOut + 3 * bN + bM); PriceEntry* entries = (PriceEntry*)(base + bOut + 3 * bN + 2 * bM);
  (void)hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(N + 1), c->stream);
  hipLaunchKernelGGL(k_opt_count, dim3(bulkGrid(M)), dim3(256), 0, c->stream, d, cnt);
  hipLaunchKernelGGL(k_opt_scan, dim3(1), dim3(1024), 0, c->stream, (const int32_t*)cnt, off, cursor, N);
  hipLaunchKernelGGL(k_opt_scatter, dim3(bulkGrid(M)), dim3(256), 0, c->stream, d, cursor, jobs);
  (void)hipEventRecord(c->fitEv0, c->stream);
  hipLaunchKernelGGL(k_price_score, dim3((N + 127) / 128), dim3(128), 0, c->stream, d, a, (const int32_t*)off, (const int32_t*)jobs, entries, out);
  (void)hipEventRecord(c->fitEv1, c->stream);
  if (detailNode >= 0) hipLaunchKernelGGL(k_price_detail, dim3(1), dim3(64), 0, c->stream, d, a, (const int32_t*)off, (const int32_t*)jobs, entries, detailNode, out + N, dPre);
  bool ok = hipOk(hipGetLastError(), "pricer launch") && hipOk(hipStreamSynchronize(c->stream), "pricer kernels");
  (void)hipEventElapsedTime(&g_lastOptMs, c->fitEv0, c->fitEv1);
  if (ok) {
    scores.resize(N);
    if (N) ok = hipOk(hipMemcpy(scores.data(), out, sizeof(PriceNodeOut) * (size_t)N, hipMemcpyDeviceToHost), "pricer scores");
    if (ok && detailNode >= 0) {
      int npre = scores[detailNode].npre;
      pre->assign((size_t)std::max(npre, 1), -1);
      if (npre > 0) ok = hipOk(hipMemcpy(pre->data(), dPre, sizeof(int32_t) * (size_t)npre, hipMemcpyDeviceToHost), "pricer victims");
    }
  }
  (void)hipFree(base);
  return ok ? 0 : -1;
}
// the queue costs the last plat_opt_score evaluated (QueueContext.CurrentCost per queue)
static int plat_opt_qcosts(Dev& d, double* out, int Q) {
  PlatCtx* c = t_ctx;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t bOut = up(sizeof(OptNodeOut) * (size_t)(d.cfg.N + 1));
  return hipOk(hipMemcpy(out, (char*)c->optScratch + bOut, sizeof(double) * (size_t)Q, hipMemcpyDeviceToHost), "opt queue costs") ? 0 : -1;
}

// grid-wide order-preserving compaction; *total comes back to the host (the next launches are sized by it)
static int plat_compact(Dev& d, const int32_t* order, int n, const uint8_t* flag, int32_t* dst, uint32_t* prefix, const int32_t* segOff, int nseg, int32_t* outSegOff, int* total) {
  (void)d;
  PlatCtx* c = t_ctx;
  *total = 0;
  int nb = (n + CMP_CHUNK - 1) / CMP_CHUNK;
  size_t need = (size_t)nb + 8;
  if (c->cmpScratchInts < need) {
    if (c->cmpScratch) (void)hipFree(c->cmpScratch);
    c->cmpScratch = nullptr; c->cmpScratchInts = 0;
    if (!hipOk(hipMalloc((void**)&c->cmpScratch, need * 2 * sizeof(int32_t)), "compaction scratch")) return -1;
    c->cmpScratchInts = need * 2;
  }
  int32_t* blockCount = c->cmpScratch; int32_t* dTotal = c->cmpScratch + c->cmpScratchInts - 1;
  if (nb > 0) {
    hipLaunchKernelGGL(k_cmp_count, dim3(nb), dim3(256), 0, c->stream, order, n, flag, blockCount);
    hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(64), 0, c->stream, blockCount, nb, dTotal);
    hipLaunchKernelGGL(k_cmp_write, dim3(nb), dim3(256), 0, c->stream, order, n, flag, dst, prefix, (const int32_t*)blockCount);
    c->roundLaunches += 3;
  } else (void)hipMemsetAsync(dTotal, 0, sizeof(int32_t), c->stream);
  if (segOff) { hipLaunchKernelGGL(k_seg_off, dim3((nseg + 256) / 256), dim3(256), 0, c->stream, segOff, nseg, n, (const uint32_t*)prefix, (const int32_t*)dTotal, outSegOff); c->roundLaunches++; }
  if (!hipOk(hipGetLastError(), "compaction launch")) return -1;
  int32_t t = 0;
  if (!hipOk(hipMemcpyAsync(&t, dTotal, sizeof t, hipMemcpyDeviceToHost, c->stream), "compaction total") || !hipOk(hipStreamSynchronize(c->stream), "compaction")) return -1;
  *total = t;
  return 0;
}
static int plat_build_base(Dev& d) {
  int N = d.cfg.N;
  int nb2 = 64; while (nb2 < N) nb2 <<= 1;
  hipLaunchKernelGGL(k_base_fill, dim3((nb2 + 255) / 256), dim3(256), 0, t_ctx->stream, d, nb2);
  unsigned long long* a = (unsigned long long*)d.baseKey;
  if (nb2 <= 4096) {
    // pad region beyond nb2 is never touched: the tile kernel is only used when the array is a multiple of 4096
    for (int k = 2; k <= nb2; k <<= 1) for (int j = k >> 1; j > 0; j >>= 1) hipLaunchKernelGGL(k_bitonic_step, dim3((nb2 + 255) / 256), dim3(256), 0, t_ctx->stream, a, j, k);
  } else {
    int tiles = nb2 / 4096;
    hipLaunchKernelGGL(k_bitonic_tile, dim3(tiles), dim3(1024), 0, t_ctx->stream, a, 2, 4096, 1);  // all steps with k <= 4096
    for (int k = 8192; k <= nb2; k <<= 1) {
      int j = k >> 1;
      for (; j >= 4096; j >>= 1) hipLaunchKernelGGL(k_bitonic_step, dim3((nb2 + 255) / 256), dim3(256), 0, t_ctx->stream, a, j, k);
      hipLaunchKernelGGL(k_bitonic_tile, dim3(tiles), dim3(1024), 0, t_ctx->stream, a, k, k, 2048);      // remaining steps j = 2048..1 inside tiles
    }
  }
  hipLaunchKernelGGL(k_base_finish, dim3((N + 255) / 256), dim3(256), 0, t_ctx->stream, d);
  if (d.fitBits) { size_t total = (size_t)d.f.F * d.fitW; hipLaunchKernelGGL(k_base_fitbits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, t_ctx->stream, d); }
  if (!hipOk(hipGetLastError(), "base build launch")) return -1;
  if (!hipOk(hipStreamSynchronize(t_ctx->stream), "base build")) return -1;
  return 0;
}
static int plat_run_shape_mask(Dev& d, const uint64_t* classMask, const int32_t* shapeClass) {
  size_t total = (size_t)d.cfg.S * d.cfg.W;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_shape_mask, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, t_ctx->stream, d, classMask, shapeClass);
  if (!hipOk(hipGetLastError(), "k_shape_mask launch")) return -1;
  if (!hipOk(hipStreamSynchronize(t_ctx->stream), "k_shape_mask")) return -1;
  return 0;
}

// kernel duration of the last fit batch, measured with HIP events on the launch stream
static double plat_last_fit_ms() { return t_ctx ? (double)t_ctx->lastFitMs : 0.0; }

// (the scratch of a fit batch is kept across calls and the rank -> node table is the host's own copy: the call is launch + one small download, nothing else)
static int plat_run_fit_batch(Dev& d, const std::vector<int32_t>& shapes, int level, std::vector<int32_t>& out, const int32_t* nodeByRankHost = nullptr) {
  int ns = (int)shapes.size();
  if (ns == 0) return 0;
  PlatCtx* c = t_ctx;
  size_t need = (size_t)ns * (sizeof(int32_t) + FIT_OSTR * sizeof(unsigned long long)) + 16;
  if (c->fitScratchBytes < need) {
    if (c->fitScratch) (void)hipFree(c->fitScratch);
    c->fitScratch = nullptr; c->fitScratchBytes = 0;
    if (!hipOk(hipMalloc(&c->fitScratch, need * 2), "hipMalloc")) return -1;
    c->fitScratchBytes = need * 2;
  }
  unsigned long long* dOut = (unsigned long long*)c->fitScratch; int32_t* dShapes = (int32_t*)(dOut + (size_t)ns * FIT_OSTR);
  (void)hipMemcpyAsync(dShapes, shapes.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, t_ctx->stream);
  (void)hipMemsetAsync(dOut, 0xff, (size_t)ns * FIT_OSTR * sizeof(unsigned long long), t_ctx->stream);
  int tiles = (d.cfg.N + FIT_TILE - 1) / FIT_TILE;
  int ysplit = std::max(1, std::min(ns, (2048 + tiles - 1) / tiles));  // >= ~2048 workgroups when the node count alone cannot fill 256 CUs
  hipEvent_t e0 = t_ctx->fitEv0, e1 = t_ctx->fitEv1;
  (void)hipEventRecord(e0, t_ctx->stream);
  const bool two = d.cfg.keyWords == 2;   // a two-word order key: one launch per word (armada_sched_wk.hip k_fit_batch_wk), the low word of the minimum in word 1
  if (two) { if (asched_internal_wk_fit_batch(&d, dShapes, ns, level, dOut, tiles, ysplit, t_ctx->stream)) { c->err = "k_fit_batch_wk launch failed"; return -1; } }
  else
  hipLaunchKernelGGL(k_fit_batch, dim3(tiles, ysplit), dim3(FIT_TILE), 0, t_ctx->stream, d, dShapes, ns, level, dOut);
  (void)hipEventRecord(e1, t_ctx->stream);
  std::vector<unsigned long long> wide((size_t)ns * FIT_OSTR), keys(ns);
  bool ok = hipOk(hipGetLastError(), "k_fit_batch launch") && hipOk(hipMemcpyAsync(wide.data(), dOut, wide.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, t_ctx->stream), "hipMemcpy") &&
            hipOk(hipStreamSynchronize(t_ctx->stream), "k_fit_batch");
  (void)hipEventElapsedTime(&t_ctx->lastFitMs, e0, e1);
  if (!ok) return -1;
  for (int i = 0; i < ns; i++) keys[i] = (two && wide[(size_t)i * FIT_OSTR] != ~0ull) ? wide[(size_t)i * FIT_OSTR + 1] : wide[(size_t)i * FIT_OSTR];
  std::vector<int32_t> nodeByRank;
  if (!nodeByRankHost) { nodeByRank.resize(d.cfg.N); if (d.cfg.N) (void)hipMemcpy(nodeByRank.data(), d.nodeByRank, d.cfg.N * sizeof(int32_t), hipMemcpyDeviceToHost); nodeByRankHost = nodeByRank.data(); }
  unsigned long long mask = (1ull << d.cfg.idxBits) - 1;
  for (int i = 0; i < ns; i++) out[i] = keys[i] == ~0ull ? -1 : nodeByRankHost[keys[i] & mask];
  return 0;
}
// ---- one pool on several GPUs: the kernels live in armada_sched_mgpu.hip (their own code object)
#include "mgpu.h"
extern "C" int asched_internal_mgpu_pack(const Dev* d, const GlobalKeyLayout* L, int level, const unsigned long long* keys, const int32_t* slot, int nq, long long* out, int32_t* bad, hipStream_t s);
extern "C" int asched_internal_mgpu_delta(const Dev* d, long long* buf, int ns, int np, hipStream_t s);
extern "C" int asched_internal_mgpu_resolve(const Dev* d, const long long* red, long long* freeC, uint8_t* ownPre, uint8_t* conflict, uint8_t* gangReplay,
                                            int32_t* node, int32_t* prio, uint8_t* replay, int32_t* counts, int ns, int np, hipStream_t s);
// the handle's scratch buffer of the fit / capacity / gang-unit launches (kept across calls: an allocation per call showed up as a 13 ms outlier among 0.06 ms calls)
static void* plat_fit_scratch(size_t need) {
  PlatCtx* c = t_ctx;
  if (c->fitScratchBytes < need) {
    if (c->fitScratch) (void)hipFree(c->fitScratch);
    c->fitScratch = nullptr; c->fitScratchBytes = 0;
    if (!hipOk(hipMalloc(&c->fitScratch, need * 2), "hipMalloc")) return nullptr;
    c->fitScratchBytes = need * 2;
  }
  return c->fitScratch;
}
// the submit check's gang units, one workgroup per unit (submit_gang.h; the kernel lives in armada_sched_mgpu.hip).  out: 4 words per unit; the kernel time goes to lastFitMs
extern "C" int asched_internal_submit_gangs(const Dev* d, const int32_t* off, const int32_t* jobs, int nu, int32_t* out, hipStream_t s);
#define SG_MAX_NODES 262144   // the workgroup's node bitmap lives in LDS (32 KB at this size)
static int plat_run_submit_gangs(Dev& d, const std::vector<int32_t>& off, const std::vector<int32_t>& jobs, std::vector<int32_t>& out) {
  int nu = (int)off.size() - 1;
  out.assign((size_t)std::max(nu, 0) * 4, 0);
  if (nu <= 0) return 0;
  hipStream_t st = t_ctx->stream;
  size_t nOff = (off.size() + 3) & ~(size_t)3, nJobs = (std::max<size_t>(jobs.size(), 1) + 3) & ~(size_t)3;
  int32_t* base = (int32_t*)plat_fit_scratch((nOff + nJobs + out.size()) * 4);
  bool ok = base != nullptr;
  int32_t *dOff = base, *dJobs = base + nOff, *dOut = base + nOff + nJobs;
  if (ok) {
    (void)hipMemcpyAsync(dOff, off.data(), off.size() * 4, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(dJobs, jobs.data(), jobs.size() * 4, hipMemcpyHostToDevice, st);
    (void)hipEventRecord(t_ctx->fitEv0, st);
    ok = asched_internal_submit_gangs(&d, dOff, dJobs, nu, dOut, st) == 0;
    (void)hipEventRecord(t_ctx->fitEv1, st);
    ok = ok && hipOk(hipMemcpyAsync(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost, st), "hipMemcpy") && hipOk(hipStreamSynchronize(st), "k_submit_gangs");
    (void)hipEventElapsedTime(&t_ctx->lastFitMs, t_ctx->fitEv0, t_ctx->fitEv1);
  }
  return ok ? 0 : -1;
}
// the evicted table by rank (replay_rank.h; kernels in armada_sched_mgpu.hip): three launches on the handle's stream, no read-back
extern "C" int asched_internal_replay_rank(const Dev* d, int n, int keepPending, hipStream_t s);
static int plat_replay_rank(Dev& d, int n, int keepPending) {
  if (n <= 0) return 0;
  t_ctx->roundLaunches += 3;
  return asched_internal_replay_rank(&d, n, keepPending, t_ctx->stream) == 0 && hipOk(hipGetLastError(), "k_replay_rank launch") ? 0 : -1;
}
// uniform submit-check units (submit_gang.h): per shape {first node or -1, members all nodes take together}
extern "C" int asched_internal_fit_capacity(const Dev* d, const int32_t* shapes, int ns, unsigned long long* out, hipStream_t s);
static int plat_run_fit_capacity(Dev& d, const std::vector<int32_t>& shapes, std::vector<int32_t>& firstNode, std::vector<long long>& capacity, const int32_t* nodeByRankHost) {
  int ns = (int)shapes.size();
  firstNode.assign(ns, -1); capacity.assign(ns, 0);
  if (ns == 0 || d.cfg.N == 0) return 0;
  hipStream_t st = t_ctx->stream;
  size_t words = (size_t)ns * FIT_OSTR;
  unsigned long long* dOut = (unsigned long long*)plat_fit_scratch(words * 8 + (size_t)ns * 4 + 16);
  int32_t* dShapes = (int32_t*)(dOut + words);
  bool ok = dOut != nullptr;
  std::vector<unsigned long long> init(words, 0), got(words);
  for (int i = 0; i < ns; i++) init[(size_t)i * FIT_OSTR] = ~0ull;
  if (ok) {
    (void)hipMemcpyAsync(dOut, init.data(), words * 8, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(dShapes, shapes.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st);
    (void)hipEventRecord(t_ctx->fitEv0, st);
    ok = asched_internal_fit_capacity(&d, dShapes, ns, dOut, st) == 0;
    (void)hipEventRecord(t_ctx->fitEv1, st);
    ok = ok && hipOk(hipMemcpyAsync(got.data(), dOut, words * 8, hipMemcpyDeviceToHost, st), "hipMemcpy") && hipOk(hipStreamSynchronize(st), "k_fit_capacity");
    (void)hipEventElapsedTime(&t_ctx->lastFitMs, t_ctx->fitEv0, t_ctx->fitEv1);
  }
  if (!ok) return -1;
  unsigned long long mask = (1ull << d.cfg.idxBits) - 1;
  for (int i = 0; i < ns; i++) {
    unsigned long long k = got[(size_t)i * FIT_OSTR];
    firstNode[i] = k == ~0ull ? -1 : nodeByRankHost[k & mask];
    capacity[i] = (long long)got[(size_t)i * FIT_OSTR + 1];
  }
  return 0;
}
// a caller-side buffer may be memory of this handle's GPU (a tensor the collective reduces in place: used directly) or host memory (staged)
static bool plat_is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice && a.device == t_ctx->device;
}
static int plat_run_fit_batch_global(Dev& d, const std::vector<int32_t>& shapes, const std::vector<int32_t>& slot, int level, GlobalKeyLayout L, const int32_t* globalRank, long long* out, int* badOut) {
  int ns = (int)shapes.size(), nq = (int)slot.size();
  bool direct = plat_is_device_ptr(out);
  int32_t *dShapes = nullptr, *dSlot = nullptr, *dRank = nullptr, *dBad = nullptr; unsigned long long* dKeys = nullptr; long long* dWords = direct ? out : nullptr;
  bool ok = hipOk(hipMalloc(&dShapes, ns * sizeof(int32_t)), "hipMalloc") && hipOk(hipMalloc(&dKeys, (size_t)ns * FIT_OSTR * sizeof(unsigned long long)), "hipMalloc") &&
            hipOk(hipMalloc(&dSlot, nq * sizeof(int32_t)), "hipMalloc") && hipOk(hipMalloc(&dBad, sizeof(int32_t)), "hipMalloc") &&
            (direct || hipOk(hipMalloc(&dWords, nq * sizeof(long long)), "hipMalloc")) && (!globalRank || hipOk(hipMalloc(&dRank, std::max(d.cfg.N, 1) * sizeof(int32_t)), "hipMalloc"));
  if (ok) {
    hipStream_t st = t_ctx->stream;
    (void)hipMemcpyAsync(dShapes, shapes.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(dSlot, slot.data(), nq * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (globalRank) (void)hipMemcpyAsync(dRank, globalRank, d.cfg.N * sizeof(int32_t), hipMemcpyHostToDevice, st);
    (void)hipMemsetAsync(dKeys, 0xff, (size_t)ns * FIT_OSTR * sizeof(unsigned long long), st);
    (void)hipMemsetAsync(dBad, 0, sizeof(int32_t), st);
    L.globalRank = dRank;
    int tiles = (d.cfg.N + FIT_TILE - 1) / FIT_TILE;
    int ysplit = std::max(1, std::min(ns, (2048 + tiles - 1) / tiles));
    (void)hipEventRecord(t_ctx->fitEv0, st);
    if (d.cfg.N > 0) hipLaunchKernelGGL(k_fit_batch, dim3(tiles, ysplit), dim3(FIT_TILE), 0, st, d, dShapes, ns, level, dKeys);
    ok = asched_internal_mgpu_pack(&d, &L, level, dKeys, dSlot, nq, dWords, dBad, st) == 0;
    (void)hipEventRecord(t_ctx->fitEv1, st);
    ok = ok && hipOk(hipGetLastError(), "fit_select_batch_global launch") && hipOk(hipStreamSynchronize(st), "fit_select_batch_global");
    (void)hipEventElapsedTime(&t_ctx->lastFitMs, t_ctx->fitEv0, t_ctx->fitEv1);
    int32_t bad = 0;
    if (ok) ok = hipOk(hipMemcpy(&bad, dBad, sizeof bad, hipMemcpyDeviceToHost), "hipMemcpy");
    if (ok && !direct) ok = hipOk(hipMemcpy(out, dWords, nq * sizeof(long long), hipMemcpyDeviceToHost), "hipMemcpy");
    *badOut = bad;
  }
  (void)hipFree(dShapes); (void)hipFree(dKeys); (void)hipFree(dSlot); (void)hipFree(dBad); (void)hipFree(dRank); if (!direct) (void)hipFree(dWords);
  return ok ? 0 : -1;
}
static int plat_round_delta(Dev& d, int ns, int np, long long* buf) {
  size_t words = (size_t)d.cfg.N * d.cfg.R + d.cfg.M;
  bool direct = plat_is_device_ptr(buf);
  long long* dBuf = direct ? buf : nullptr;
  if (!direct && !hipOk(hipMalloc(&dBuf, std::max<size_t>(words, 1) * 8), "hipMalloc")) return -1;
  hipStream_t st = t_ctx->stream;
  bool ok = hipOk(hipMemsetAsync(dBuf, 0, words * 8, st), "hipMemsetAsync") && asched_internal_mgpu_delta(&d, dBuf, ns, np, st) == 0 && hipOk(hipStreamSynchronize(st), "round_delta");
  if (ok && !direct) ok = hipOk(hipMemcpy(buf, dBuf, words * 8, hipMemcpyDeviceToHost), "hipMemcpy");
  if (!direct) (void)hipFree(dBuf);
  return ok ? 0 : -1;
}
static int plat_delta_resolve(Dev& d, const long long* red, int ns, int np, int32_t* counts, int32_t* node, int32_t* prio, uint8_t* replay) {
  int N = d.cfg.N, M = d.cfg.M, R = d.cfg.R, G = std::max(d.cfg.G, 1);
  size_t words = (size_t)N * R + M;
  bool direct = plat_is_device_ptr(red);
  long long *dRed = nullptr, *freeC = nullptr; uint8_t* bytes = nullptr; int32_t* ints = nullptr;
  size_t nb = (size_t)M + N + G + M, ni = 4 + 2 * (size_t)M;   // ownPre | conflict | gangReplay | replay ; counts | node | prio
  bool ok = (direct || hipOk(hipMalloc(&dRed, std::max<size_t>(words, 1) * 8), "hipMalloc")) && hipOk(hipMalloc(&freeC, std::max<size_t>((size_t)N * R, 1) * 8), "hipMalloc") &&
            hipOk(hipMalloc(&bytes, nb), "hipMalloc") && hipOk(hipMalloc(&ints, ni * 4), "hipMalloc");
  if (ok) {
    hipStream_t st = t_ctx->stream;
    if (!direct) (void)hipMemcpyAsync(dRed, red, words * 8, hipMemcpyHostToDevice, st);
    (void)hipMemsetAsync(bytes, 0, nb, st); (void)hipMemsetAsync(ints, 0, 16, st);
    uint8_t *ownPre = bytes, *conflict = bytes + M, *gangReplay = conflict + N, *rp = gangReplay + G;
    ok = asched_internal_mgpu_resolve(&d, direct ? red : dRed, freeC, ownPre, conflict, gangReplay, ints + 4, ints + 4 + M, rp, ints, ns, np, st) == 0 && hipOk(hipStreamSynchronize(st), "round_delta_resolve");
    if (ok) ok = hipOk(hipMemcpy(counts, ints, 16, hipMemcpyDeviceToHost), "hipMemcpy");
    if (ok && M) ok = hipOk(hipMemcpy(node, ints + 4, (size_t)M * 4, hipMemcpyDeviceToHost), "hipMemcpy") && hipOk(hipMemcpy(prio, ints + 4 + M, (size_t)M * 4, hipMemcpyDeviceToHost), "hipMemcpy") &&
                     hipOk(hipMemcpy(replay, rp, M, hipMemcpyDeviceToHost), "hipMemcpy");
  }
  if (!direct) (void)hipFree(dRed);
  (void)hipFree(freeC); (void)hipFree(bytes); (void)hipFree(ints);
  return ok ? 0 : -1;
}
static int plat_run_drf(Dev& dev, const std::vector<int64_t>& a, const std::vector<int64_t>& t, double* out) {
  Dev d = dev;
  for (int r = 0; r < d.cfg.R; r++) d.cfg.totalResources[r] = t[r];
  int64_t* da = nullptr; double* dout = nullptr;
  (void)hipMalloc(&da, MAXR * sizeof(int64_t)); (void)hipMalloc(&dout, sizeof(double));
  (void)hipMemcpy(da, a.data(), a.size() * sizeof(int64_t), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_drf, dim3(1), dim3(64), 0, t_ctx->stream, d, da, dout);
  (void)hipStreamSynchronize(t_ctx->stream);
  (void)hipMemcpy(out, dout, sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(da); (void)hipFree(dout);
  return 0;
}
static int plat_run_fair_shares(Dev& dev, int q, const int32_t* nameRank, const double* weight, const double* cds, double* fair, double* dc, double* uc) {
  Dev d = dev;
  d.cfg.Q = q;
  size_t nb = (size_t)std::max(q, 1);
  double *dw, *df, *ddc, *duc, *dpp, *dpc, *dcds; int32_t *dnr, *dnx; uint8_t* dih;
  (void)hipMalloc(&dw, nb * 8); (void)hipMalloc(&df, nb * 8); (void)hipMalloc(&ddc, nb * 8); (void)hipMalloc(&duc, nb * 8);
  (void)hipMalloc(&dpp, nb * 8); (void)hipMalloc(&dpc, nb * 8); (void)hipMalloc(&dcds, nb * 8);
  (void)hipMalloc(&dnr, nb * 4); (void)hipMalloc(&dnx, nb * 4); (void)hipMalloc(&dih, nb);
  (void)hipMemcpy(dw, weight, q * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dcds, cds, q * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dnr, nameRank, q * 4, hipMemcpyHostToDevice);
  d.qWeight = dw; d.qNameRank = dnr; d.qFair = df; d.qDc = ddc; d.qUc = duc; d.pqProposed = dpp; d.pqCurrent = dpc; d.pqInHeap = dih; d.itNext = dnx;
  hipLaunchKernelGGL(k_fair, dim3(1), dim3(64), 0, t_ctx->stream, d, dcds);
  bool ok = hipOk(hipStreamSynchronize(t_ctx->stream), "k_fair");
  (void)hipMemcpy(fair, df, q * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(dc, ddc, q * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(uc, duc, q * 8, hipMemcpyDeviceToHost);
  (void)hipFree(dw); (void)hipFree(df); (void)hipFree(ddc); (void)hipFree(duc); (void)hipFree(dpp); (void)hipFree(dpc); (void)hipFree(dcds);
  (void)hipFree(dnr); (void)hipFree(dnx); (void)hipFree(dih);
  return ok ? 0 : -1;
}

#include "asched_host.inc"

#endif  // main translation unit
#else  // ASCHED_AUX_TU ----------------------------------------------------------------------------------------------------
// armada_sched_aux.hip compiles this file a second time with ASCHED_AUX_TU defined: the device code above, ONE kernel
// (k_control_aux: the submit-check commands, round_run.h runAuxCommand) and no host ABI.  A separate translation unit = a
// separate code object: whatever is added to the auxiliary commands can never move a register, an LDS offset or an inlining
// decision in the round kernel, whose code is the measured one.
__global__ __launch_bounds__(CTL_THREADS) void k_control_aux(Dev dev, int cmd, HelpBox* box, MktDev mk) {
  // workgroup 0 of k_control without helper workgroups: wave 0 runs the command, the other waves serve its mailbox
  if (threadIdx.x == 0) { g_box = box; g_H = 0; g_gen = 0; g_mk = mk; g_fl.eng.abandon = 0; g_fl.eng.idleSince = 0; g_fl.eng.idleLast = 0; g_fl.eng.idleProg = 0; }
  {
    const int* src = (const int*)&dev; int* dst = (int*)&g_dev;
    for (int i = threadIdx.x; i < (int)(sizeof(Dev) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  Dev& d = g_dev;
  relocateIn(d, cmd);
  if (threadIdx.x >= 64) {
    for (;;) {
      __syncthreads();
      int op = g_mb.op;
      if (op == OP_EXIT) break;
      if (op == OP_SCAN) {
        unsigned long long v = scanPart(d, g_mb.scan, threadIdx.x, (int)blockDim.x);
        if ((threadIdx.x & 63) == 0) g_mb.partial[threadIdx.x >> 6] = v;
      } else if (op == OP_FAIR) {
        int v = fairPart(d, g_mb.fair, threadIdx.x, (int)blockDim.x);
        if ((threadIdx.x & 63) == 0) g_mb.waveCount[threadIdx.x >> 6] = v;
      } else if (op == OP_SCANFAIR) {
        unsigned long long v = scanPart(d, g_mb.scan, threadIdx.x, (int)blockDim.x);
        int w = fairPart(d, g_mb.fair, threadIdx.x, (int)blockDim.x);
        if ((threadIdx.x & 63) == 0) { g_mb.partial[threadIdx.x >> 6] = v; g_mb.waveCount[threadIdx.x >> 6] = w; }
      } else if (op == OP_BULK) {
        bulkPart(d, g_mb.kind, g_mb.n);
      } else if (op == OP_BULKW) {
        int kd = g_mb.kind, nn = g_mb.n;
        for (int i = threadIdx.x; i < nn; i += (int)blockDim.x) bulkElem(d, kd, i);
        __threadfence();
      } else if (op == OP_COMPACT) {
        compactPart(d);
      } else if (op == OP_ENGINE) {
        if ((threadIdx.x >> 6) == 1) engineLoop(d); else if ((threadIdx.x >> 6) == 2) bindLoop(d); else if ((threadIdx.x >> 6) == 3 && d.f.engineHc) coldLoop(d);
      }
      __syncthreads();
    }
    relocateOut();
    return;
  }
  controlMainAux(d, cmd);
  __threadfence();
  if ((threadIdx.x & 63) == 0) g_mb.op = OP_EXIT;
  __syncthreads();
  relocateOut();
}
extern "C" __attribute__((visibility("hidden"))) int asched_internal_aux_launch(const Dev* dev, int cmd, hipStream_t stream, void* helpBox, const MktDev* mk) {
  MktDev none; memset(&none, 0, sizeof none);
  hipLaunchKernelGGL(k_control_aux, dim3(1), dim3(CTL_THREADS), 0, stream, *dev, cmd, (HelpBox*)helpBox, mk ? *mk : none);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif  // ASCHED_AUX_TU

#ifdef HELP_TRACE
extern "C" int asched_debug_help_trace(unsigned long long* out64) { return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_traceSum), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1; }
#endif
