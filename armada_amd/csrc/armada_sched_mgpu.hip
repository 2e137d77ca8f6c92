// armada_sched_mgpu.hip — third translation unit of libarmada_sched.so: grid kernels outside the round kernel's code object — the ones that produce and consume the words of the
// multi-GPU exchanges, and since round 4 the submit check's gang units, one workgroup per unit (submit_gang.h): one element per thread over queries / result rows / nodes / jobs, all plain coalesced streaming
// (the per-element logic is mgpu.h, shared with the CPU build of the tests).  A separate code object so that nothing here moves the
// round kernel's code (k_control is placement-sensitive: DESIGN.md 9).
#include <hip/hip_runtime.h>
#include <stdint.h>
#define MGPU_FN __device__ static inline
#define MGPU_ADD64(p, v) atomicAdd((unsigned long long*)(p), (unsigned long long)(v))
#define MGPU_ADD32(p, v) atomicAdd((int*)(p), (int)(v))
#define MGPU_OR8(p) (*(volatile uint8_t*)(p) = 1)   // every writer stores the same value
#include "mgpu.h"

#define MG_THREADS 256
static inline int mgBlocks(long long n) { return (int)((n + MG_THREADS - 1) / MG_THREADS); }
#define MG_IDX() ((long long)blockIdx.x * MG_THREADS + threadIdx.x)

// one atomic per wave: ballot + popcount (a counter every thread bumps serialises the whole grid on one address)
__device__ static inline void waveCount(int32_t* counter, bool pred) {
  unsigned long long m = __ballot(pred);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(counter, (int)__builtin_popcountll(m));
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_pack(Dev d, GlobalKeyLayout L, int level, const unsigned long long* keys, const int32_t* slot, int nq, long long* out, int32_t* bad) {
  long long i = MG_IDX();
  if (i < nq) out[i] = mgpuPackQuery(d, L, level, keys[(size_t)slot[i] * FIT_OSTR], bad);   // (k_fit_batch's result words are FIT_OSTR apart)
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_delta(Dev d, long long* buf, int ns, int np) {
  long long i = MG_IDX();
  if (i < ns) mgpuDeltaScheduled(d, buf, (int)i);
  else if (i < (long long)ns + np) mgpuDeltaPreempted(d, buf, (int)(i - ns));
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_free_init(Dev d, long long* freeC) {
  long long n = MG_IDX();
  if (n < d.cfg.N) mgpuFreeInit(d, freeC, (int)n);
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_own(Dev d, long long* freeC, uint8_t* ownPre, int ns, int np) {
  long long i = MG_IDX();
  if (i < ns) mgpuFreeOwnScheduled(d, freeC, (int)i);
  else if (i < (long long)ns + np) mgpuOwnPreempted(d, ownPre, (int)(i - ns));
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_foreign(Dev d, const long long* red, const uint8_t* ownPre, long long* freeC) {
  long long j = MG_IDX();
  if (j < d.cfg.M) mgpuFreeForeignPreempted(d, red, ownPre, freeC, (int)j);
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_conflict(Dev d, const long long* red, const long long* freeC, uint8_t* conflict, int32_t* counts) {
  long long n = MG_IDX();
  bool over = n < d.cfg.N && mgpuConflict(d, red, freeC, conflict, (int)n);
  waveCount(counts + 0, over);
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_gang(Dev d, const long long* red, const uint8_t* conflict, uint8_t* gangReplay) {
  long long j = MG_IDX();
  if (j < d.cfg.M) mgpuGangConflict(d, red, conflict, gangReplay, (int)j);
}
__global__ __launch_bounds__(MG_THREADS) void k_mgpu_outcome(Dev d, const long long* red, const uint8_t* conflict, const uint8_t* gangReplay, int32_t* node, int32_t* prio, uint8_t* replay, int32_t* counts) {
  long long j = MG_IDX();
  int k = j < d.cfg.M ? mgpuJobOutcome(d, red, conflict, gangReplay, node, prio, replay, (int)j) : 0;
  waveCount(counts + 1, k == 1); waveCount(counts + 2, k == 2); waveCount(counts + 3, k == 3);
}

extern "C" int asched_internal_mgpu_pack(const Dev* d, const GlobalKeyLayout* L, int level, const unsigned long long* keys, const int32_t* slot, int nq, long long* out, int32_t* bad, hipStream_t s) {
  if (nq > 0) hipLaunchKernelGGL(k_mgpu_pack, dim3(mgBlocks(nq)), dim3(MG_THREADS), 0, s, *d, *L, level, keys, slot, nq, out, bad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int asched_internal_mgpu_delta(const Dev* d, long long* buf, int ns, int np, hipStream_t s) {
  if (ns + np > 0) hipLaunchKernelGGL(k_mgpu_delta, dim3(mgBlocks((long long)ns + np)), dim3(MG_THREADS), 0, s, *d, buf, ns, np);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// freeC [N*R], ownPre [M], conflict [N], gangReplay [max(G,1)], counts [4]: zeroed by the caller
extern "C" int asched_internal_mgpu_resolve(const Dev* d, const long long* red, long long* freeC, uint8_t* ownPre, uint8_t* conflict, uint8_t* gangReplay,
                                            int32_t* node, int32_t* prio, uint8_t* replay, int32_t* counts, int ns, int np, hipStream_t s) {
  int N = d->cfg.N, M = d->cfg.M;
  if (N > 0) hipLaunchKernelGGL(k_mgpu_free_init, dim3(mgBlocks(N)), dim3(MG_THREADS), 0, s, *d, freeC);
  if (ns + np > 0) hipLaunchKernelGGL(k_mgpu_own, dim3(mgBlocks((long long)ns + np)), dim3(MG_THREADS), 0, s, *d, freeC, ownPre, ns, np);
  if (M > 0) hipLaunchKernelGGL(k_mgpu_foreign, dim3(mgBlocks(M)), dim3(MG_THREADS), 0, s, *d, red, ownPre, freeC);
  if (N > 0) hipLaunchKernelGGL(k_mgpu_conflict, dim3(mgBlocks(N)), dim3(MG_THREADS), 0, s, *d, red, freeC, conflict, counts);
  if (M > 0) hipLaunchKernelGGL(k_mgpu_gang, dim3(mgBlocks(M)), dim3(MG_THREADS), 0, s, *d, red, conflict, gangReplay);
  if (M > 0) hipLaunchKernelGGL(k_mgpu_outcome, dim3(mgBlocks(M)), dim3(MG_THREADS), 0, s, *d, red, conflict, gangReplay, node, prio, replay, counts);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------ submit check: gang units, one workgroup each (submit_gang.h)
#define SG_FN __device__ static inline
#define SG_TID ((int)threadIdx.x)
#define SG_NT ((int)blockDim.x)
#define SG_SYNC() __syncthreads()
struct SgShared;
__device__ static inline unsigned long long sgWgMin(SgShared& s, unsigned long long v);
#define SG_WGMIN(s, v) sgWgMin(s, v)
#include "submit_gang.h"
__device__ static inline unsigned long long sgWgMin(SgShared& s, unsigned long long v) {
  for (int off = 32; off; off >>= 1) { unsigned long long o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
  __syncthreads();                                   // (the previous reduction's readers are done with wmin)
  if ((threadIdx.x & 63) == 0) s.wmin[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = s.wmin[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = s.wmin[w] < r ? s.wmin[w] : r;
  return r;
}
#define SG_THREADS 256
__global__ __launch_bounds__(SG_THREADS) void k_submit_gangs(Dev d, const int32_t* off, const int32_t* jobs, int nu, int32_t* out) {
  extern __shared__ uint32_t sgBits[];               // a bit per node: an earlier member of the unit in flight sits there
  __shared__ SgShared s;
  for (int i = threadIdx.x; i < (d.cfg.N + 31) / 32; i += SG_THREADS) sgBits[i] = 0;
  __syncthreads();
  for (int u = blockIdx.x; u < nu; u += gridDim.x) submitGangUnit(d, jobs + off[u], off[u + 1] - off[u], s, sgBits, out + 4 * (size_t)u);
}
extern "C" int asched_internal_submit_gangs(const Dev* d, const int32_t* off, const int32_t* jobs, int nu, int32_t* out, hipStream_t st) {
  if (nu <= 0) return 0;
  size_t lds = (size_t)((d->cfg.N + 31) / 32) * 4;
  hipLaunchKernelGGL(k_submit_gangs, dim3(nu < 4096 ? nu : 4096), dim3(SG_THREADS), lds, st, *d, off, jobs, nu, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the evicted table by rank (replay_rank.h): check (one thread) -> rank (one thread per evicted-list position) -> finish (one thread).  replayPending == 2 between the
// check and the finish says "by rank"; the round kernel never sees it (the three launches sit back to back on the handle's stream, in front of CMD_PASS1 / CMD_PASS2).
#include "replay_rank.h"
__global__ void k_replay_check(Dev d, int n) { if (threadIdx.x == 0 && blockIdx.x == 0 && rrOk(d, n)) d.rs->replayPending = 2; }
__global__ __launch_bounds__(MG_THREADS) void k_replay_rank(Dev d, int n) {
  if (d.rs->replayPending != 2) return;
  long long i = MG_IDX();
  if (i < n) rrElem(d, (int)i);
}
__global__ void k_replay_fin(Dev d, int n, int keepPending) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (d.rs->replayPending == 2) rrFinish(d, n);
  else if (!keepPending) d.rs->replayPending = 0;   // (pass 2 has no lazy replay: CMD_PASS2 walks when the table was not built here)
}
extern "C" int asched_internal_replay_rank(const Dev* d, int n, int keepPending, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_replay_check, dim3(1), dim3(64), 0, st, *d, n);
  hipLaunchKernelGGL(k_replay_rank, dim3(mgBlocks(n)), dim3(MG_THREADS), 0, st, *d, n);
  hipLaunchKernelGGL(k_replay_fin, dim3(1), dim3(64), 0, st, *d, n, keepPending);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// uniform units: per shape the smallest key among the nodes that take at least one member, and the number of members all nodes take together (submit_gang.h)
__global__ __launch_bounds__(256) void k_fit_capacity(Dev d, const int32_t* shapes, int ns, unsigned long long* out /*[ns][FIT_OSTR]: [0] min key, [1] capacity sum*/) {
  __shared__ unsigned long long wmin[4], wsum[4];
  int n = blockIdx.x * 256 + threadIdx.x;
  for (int i = blockIdx.y; i < ns; i += gridDim.y) {
    long long cap = n < d.cfg.N ? sgNodeCapacity(d, shapes[i], n) : 0;
    unsigned long long key = cap > 0 ? d.keys[n] : ~0ull, sum = (unsigned long long)cap;
    for (int off = 32; off; off >>= 1) { unsigned long long o = __shfl_xor(key, off, 64); key = o < key ? o : key; sum += __shfl_xor(sum, off, 64); }
    if ((threadIdx.x & 63) == 0) { wmin[threadIdx.x >> 6] = key; wsum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wmin[0], t = wsum[0];
      for (int w = 1; w < 4; w++) { m = wmin[w] < m ? wmin[w] : m; t += wsum[w]; }
      if (m != ~0ull) atomicMin(&out[(size_t)i * FIT_OSTR], m);
      if (t) atomicAdd(&out[(size_t)i * FIT_OSTR + 1], t);
    }
    __syncthreads();
  }
}
extern "C" int asched_internal_fit_capacity(const Dev* d, const int32_t* shapes, int ns, unsigned long long* out, hipStream_t st) {
  if (ns <= 0 || d->cfg.N <= 0) return 0;
  int tiles = (d->cfg.N + 255) / 256;
  int ysplit = ns < 1 ? 1 : (ns < (2048 + tiles - 1) / tiles ? ns : (2048 + tiles - 1) / tiles);
  hipLaunchKernelGGL(k_fit_capacity, dim3(tiles, ysplit < 1 ? 1 : ysplit), dim3(256), 0, st, *d, shapes, ns, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
