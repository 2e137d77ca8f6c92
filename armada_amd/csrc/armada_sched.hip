// armada_sched.hip — MI355X (gfx950) implementation of the C ABI in include/armada_sched.h.
//
// Kernels in this file:
//   k_control     persistent single-workgroup "round" kernel: wave 0 runs the sequential DRF/gang control flow
//                 (round_ctl.h), all 16 waves serve its data-parallel requests through an LDS mailbox:
//                   OP_SCAN    first feasible node = argmin of the packed order key over nodes whose
//                              alloc[level][r][n] >= req[r]  (coalesced SoA planes, wave shuffle + LDS reduction)
//                   OP_BULK    evictors / unbind / populate / key rebuild as block-stride loops with int64 atomics
//                   OP_COMPACT order-preserving stream compaction (wave ballot + LDS prefix) for the per-queue
//                              evicted lists and the result lists
//   k_fit_batch   wide kernel: first feasible node for many (shape, level) queries against a fixed node state
//                 (BASELINE config 2, "nodedb fit kernel"): node tile in registers, wave-level min, one atomicMin/wave
//   k_shape_mask  per-shape static mask = requirement-class mask ∧ (total >= request)
//   k_drf / k_fair  float64 goldens (fairness.go / context/scheduling.go) evaluated on the device
//
// There is no CPU compute path in this library: without a gfx950 device asched_create() fails.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

#define ASCHED_PREFIX asched_
#include "round_run.h"

// ------------------------------------------------------------------------------------------------ device primitives
enum { OP_EXIT = 0, OP_SCAN = 1, OP_BULK = 2, OP_COMPACT = 3 };

struct Mailbox {
  int op, kind, n;
  ScanArgs scan;
  unsigned long long partial[16];
  const int32_t* order; const uint8_t* flag; int32_t* dst; uint32_t* prefix;
  int waveCount[16];
  int total;
};
__shared__ Mailbox g_mb;
__shared__ Dev g_dev;

__device__ static inline void atomicAddI64(int64_t* p, int64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
__device__ static inline void atomicAddI32(int32_t* p, int32_t v) { atomicAdd(p, v); }
__device__ static inline void atomicOrI32(int32_t* p, int32_t v) { atomicOr(p, v); }

__device__ static inline unsigned long long waveMin64(unsigned long long v) {
  for (int off = 32; off; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// one thread per node (block-stride): reject by mask bit and key first, touch the alloc planes only for improving candidates
__device__ static unsigned long long scanPart(const Dev& d, const ScanArgs& a) {
  const DevCfg& c = d.cfg;
  const uint64_t* keys = d.keys + (size_t)a.level * c.Npad;
  const int64_t* plane = d.alloc + (size_t)a.level * c.R * c.Npad;
  unsigned long long best = ~0ull;
  for (int n = threadIdx.x; n < c.N; n += blockDim.x) {
    uint64_t w = a.maskA[n >> 6];
    if (a.maskB) w &= a.maskB[n >> 6];
    if (!((w >> (n & 63)) & 1)) continue;
    unsigned long long k = keys[n];
    if (k >= best) continue;
    bool fits = true;
    for (int r = 0; r < c.R; r++) fits = fits && (a.req[r] <= plane[(size_t)r * c.Npad + n]);
    if (fits) best = k;
  }
  return waveMin64(best);
}

__device__ static inline int wgFirstFit(Dev& d, const ScanArgs& a) {
  int lane = threadIdx.x & 63;
  if (lane == 0) { g_mb.op = OP_SCAN; g_mb.scan = a; }
  __syncthreads();
  unsigned long long v = scanPart(d, g_mb.scan);
  if (lane == 0) g_mb.partial[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long best = ~0ull;
  int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) { unsigned long long p = g_mb.partial[w]; best = p < best ? p : best; }
  d.rs->numScans++;
  if (best == ~0ull) return -1;
  return d.nodeByRank[best & ((1ull << d.cfg.idxBits) - 1)];
}

__device__ static void bulkPart(Dev& d, int kind, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) bulkElem(d, kind, i);
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
  for (int start = 0; start < n; start += blockDim.x) {
    int p = start + threadIdx.x;
    int v = p < n ? (order ? order[p] : p) : 0;
    bool f = p < n && flag[v];
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
  for (int q = threadIdx.x & 63; q <= nseg; q += 64) outSegOff[q] = segOff[q] < n ? (int32_t)d.evSortKey[segOff[q]] : total;
  __threadfence();
  return total;
}
__device__ static inline int wgCompactIota(Dev& d, int n, const uint8_t* flag, int32_t* dst) { return wgCompactRun(d, nullptr, n, flag, dst, nullptr); }

// lane-per-queue argmin under the reference's Less (a strict total order, so argmin == heap top)
__device__ static inline int pqTop(Dev& d, const Ctl& c) {
  int lane = threadIdx.x & 63;
  int best = -1;
  for (int q = lane; q < d.cfg.Q; q += 64)
    if (d.pqInHeap[q] && (best < 0 || pqLess(d, c, q, best))) best = q;
  for (int off = 32; off; off >>= 1) {
    int o = __shfl_xor(best, off, 64);
    if (o >= 0 && (best < 0 || pqLess(d, c, o, best))) best = o;
  }
  return best;
}

// ------------------------------------------------------------------------------------------------ kernels
__global__ __launch_bounds__(1024) void k_control(Dev dev, int cmd) {
  // the Dev descriptor (pointers + config) is staged in LDS once; every wave reads it from there
  {
    const int* src = (const int*)&dev; int* dst = (int*)&g_dev;
    for (int i = threadIdx.x; i < (int)(sizeof(Dev) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  Dev& d = g_dev;
  if (threadIdx.x >= 64) {  // worker waves: serve mailbox requests until OP_EXIT
    for (;;) {
      __syncthreads();
      int op = g_mb.op;
      if (op == OP_EXIT) break;
      if (op == OP_SCAN) {
        unsigned long long v = scanPart(d, g_mb.scan);
        if ((threadIdx.x & 63) == 0) g_mb.partial[threadIdx.x >> 6] = v;
      } else if (op == OP_BULK) {
        bulkPart(d, g_mb.kind, g_mb.n);
      } else if (op == OP_COMPACT) {
        compactPart(d);
      }
      __syncthreads();
    }
    return;
  }
  Ctl c;
  c.txn.active = d.rs->txnActive; c.fairStamp = d.rs->fairStamp; c.preList = d.preList; c.preCount = 0;
  c.skipKeyCheck = 0; c.compareSchedPrio = 0; c.preferLarge = d.cfg.preferLarge; c.useReplayAlloc = 0; c.onlyEvicted = 0;
  runCommand(d, c, cmd);
  d.rs->txnActive = c.txn.active; d.rs->fairStamp = c.fairStamp;
  __threadfence();
  if ((threadIdx.x & 63) == 0) g_mb.op = OP_EXIT;
  __syncthreads();
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
  for (int i = s0; i < s1; i++) {
    int s = shapes[i];
    bool f = valid && ((d.shapeMask[(size_t)s * c.W + word] >> bit) & 1);
    const int64_t* req = d.shapeReq + (size_t)s * c.R;
    for (int r = 0; r < c.R; r++) f = f && req[r] <= al[r];
    unsigned long long v = waveMin64(f ? key : ~0ull);
    if ((threadIdx.x & 63) == 0 && v != ~0ull) atomicMin(&out[i], v);
  }
}

__global__ void k_drf(Dev d, const int64_t* alloc, double* out) { if (threadIdx.x == 0) *out = drf(d, alloc); }
__global__ void k_fair(Dev d, const double* cds) { if (threadIdx.x == 0) updateFairShares(d, cds); }

// ------------------------------------------------------------------------------------------------ platform layer
static std::string g_err;
static hipStream_t g_stream = nullptr;
static bool g_inited = false;

static bool hipOk(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}
static const char* plat_last_error() { return g_err.c_str(); }
static bool plat_init(std::string& err, int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { err = "no HIP device: libarmada_sched.so is the gfx950 implementation and has no CPU path"; return false; }
  if (device >= n) { err = "device ordinal out of range"; return false; }
  if (device >= 0 && hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return false; }
  if (g_inited) return true;
  int cur = 0; (void)hipGetDevice(&cur);
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, cur) != hipSuccess) { err = "hipGetDeviceProperties failed"; return false; }
  if (std::string(p.gcnArchName).find("gfx950") == std::string::npos) { err = std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only"; return false; }
  if (hipStreamCreate(&g_stream) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
  g_inited = true;
  return true;
}
static void* plat_malloc(size_t n) { void* p = nullptr; if (!hipOk(hipMalloc(&p, n), "hipMalloc")) return nullptr; return p; }
static void plat_free(void* p) { if (p) (void)hipFree(p); }
static void plat_memset(void* p, int v, size_t n) { (void)hipMemsetAsync(p, v, n, g_stream); }
static void plat_h2d(void* d, const void* s, size_t n) { (void)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, g_stream); (void)hipStreamSynchronize(g_stream); }
static void plat_d2h(void* d, const void* s, size_t n) { (void)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, g_stream); (void)hipStreamSynchronize(g_stream); }

// device time of the last control-kernel launch (HIP events recorded on the launch stream) — bench.py's roofline input
static float g_lastControlMs = 0.f;
static int g_lastControlLaunches = 0;
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static double plat_last_control_ms() { return (double)g_lastControlMs; }
static int plat_last_control_launches() { return g_lastControlLaunches; }

static int plat_run_control(Dev& dev, int cmd) {
  if (!g_ev0) { (void)hipEventCreate(&g_ev0); (void)hipEventCreate(&g_ev1); }
  (void)hipEventRecord(g_ev0, g_stream);
  hipLaunchKernelGGL(k_control, dim3(1), dim3(1024), 0, g_stream, dev, cmd);
  (void)hipEventRecord(g_ev1, g_stream);
  if (!hipOk(hipGetLastError(), "k_control launch")) return -1;
  if (!hipOk(hipStreamSynchronize(g_stream), "k_control")) return -1;
  (void)hipEventElapsedTime(&g_lastControlMs, g_ev0, g_ev1);
  g_lastControlLaunches = 1;
  return 0;
}
static int plat_run_shape_mask(Dev& d, const uint64_t* classMask, const int32_t* shapeClass) {
  size_t total = (size_t)d.cfg.S * d.cfg.W;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_shape_mask, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, g_stream, d, classMask, shapeClass);
  if (!hipOk(hipGetLastError(), "k_shape_mask launch")) return -1;
  if (!hipOk(hipStreamSynchronize(g_stream), "k_shape_mask")) return -1;
  return 0;
}

// kernel duration of the last fit batch, measured with HIP events on the launch stream
static float g_lastFitMs = 0.f;
static double plat_last_fit_ms() { return (double)g_lastFitMs; }

static int plat_run_fit_batch(Dev& d, const std::vector<int32_t>& shapes, int level, std::vector<int32_t>& out) {
  int ns = (int)shapes.size();
  if (ns == 0) return 0;
  int32_t* dShapes = nullptr; unsigned long long* dOut = nullptr;
  if (!hipOk(hipMalloc(&dShapes, ns * sizeof(int32_t)), "hipMalloc") || !hipOk(hipMalloc(&dOut, ns * sizeof(unsigned long long)), "hipMalloc")) return -1;
  (void)hipMemcpyAsync(dShapes, shapes.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, g_stream);
  (void)hipMemsetAsync(dOut, 0xff, ns * sizeof(unsigned long long), g_stream);
  int tiles = (d.cfg.N + FIT_TILE - 1) / FIT_TILE;
  int ysplit = std::max(1, std::min(ns, (2048 + tiles - 1) / tiles));  // >= ~2048 workgroups when the node count alone cannot fill 256 CUs
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, g_stream);
  hipLaunchKernelGGL(k_fit_batch, dim3(tiles, ysplit), dim3(FIT_TILE), 0, g_stream, d, dShapes, ns, level, dOut);
  (void)hipEventRecord(e1, g_stream);
  bool ok = hipOk(hipGetLastError(), "k_fit_batch launch") && hipOk(hipStreamSynchronize(g_stream), "k_fit_batch");
  (void)hipEventElapsedTime(&g_lastFitMs, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  std::vector<unsigned long long> keys(ns);
  if (ok) ok = hipOk(hipMemcpy(keys.data(), dOut, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost), "hipMemcpy");
  (void)hipFree(dShapes); (void)hipFree(dOut);
  if (!ok) return -1;
  std::vector<int32_t> nodeByRank(d.cfg.N);
  if (d.cfg.N) (void)hipMemcpy(nodeByRank.data(), d.nodeByRank, d.cfg.N * sizeof(int32_t), hipMemcpyDeviceToHost);
  unsigned long long mask = (1ull << d.cfg.idxBits) - 1;
  for (int i = 0; i < ns; i++) out[i] = keys[i] == ~0ull ? -1 : nodeByRank[keys[i] & mask];
  return 0;
}
static int plat_run_drf(Dev& dev, const std::vector<int64_t>& a, const std::vector<int64_t>& t, double* out) {
  Dev d = dev;
  for (int r = 0; r < d.cfg.R; r++) d.cfg.totalResources[r] = t[r];
  int64_t* da = nullptr; double* dout = nullptr;
  (void)hipMalloc(&da, MAXR * sizeof(int64_t)); (void)hipMalloc(&dout, sizeof(double));
  (void)hipMemcpy(da, a.data(), a.size() * sizeof(int64_t), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_drf, dim3(1), dim3(64), 0, g_stream, d, da, dout);
  (void)hipStreamSynchronize(g_stream);
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
  hipLaunchKernelGGL(k_fair, dim3(1), dim3(64), 0, g_stream, d, dcds);
  bool ok = hipOk(hipStreamSynchronize(g_stream), "k_fair");
  (void)hipMemcpy(fair, df, q * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(dc, ddc, q * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(uc, duc, q * 8, hipMemcpyDeviceToHost);
  (void)hipFree(dw); (void)hipFree(df); (void)hipFree(ddc); (void)hipFree(duc); (void)hipFree(dpp); (void)hipFree(dpc); (void)hipFree(dcds);
  (void)hipFree(dnr); (void)hipFree(dnx); (void)hipFree(dih);
  return ok ? 0 : -1;
}

#include "asched_host.inc"
