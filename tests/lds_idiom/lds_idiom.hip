// Test infrastructure (never linked into the product): the LDS hand-back idiom that produced the wrong rounds of round 3's helper-slot variants
// (profiles/r04a_lds_handback_rootcause.txt).  One lane stores a __shared__ word, the whole wave reads it "right after":
//   variant 0  plain          — a data race between lanes in the C++ model: LLVM may reuse a value loaded before the store for the lanes that did not store
//   variant 1  fenced         — LANE0_PUBLISHED() (wavefront-scope fence) between the store and the read: what the product does at every such site
//   variant 2  readfirstlane  — the wave takes lane 0's value (UNI32): right whether or not the compiler forwards the store
// Each kernel runs `rounds` iterations; in iteration r the folded value is (base + r) and every lane accumulates what it read back.
#include <hip/hip_runtime.h>
#include <cstdint>
__shared__ int g_word;
__device__ __attribute__((noinline)) int foldMax(const int* slots, int r) {   // stands for the fold over the helpers' result slots
  int lane = threadIdx.x & 63;
  int m = slots[(r * 64 + lane) & 1023];
  for (int off = 32; off; off >>= 1) { int o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
  return m;
}
template <int V> __global__ void k_idiom(const int* slots, int rounds, long long* out) {
  int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) g_word = -1;
  __syncthreads();
  long long acc = 0;
  for (int r = 0; r < rounds; r++) {
    int m = foldMax(slots, r);
    if (lane == 0) g_word = m;
    if (V == 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    int h = g_word;                                  // read by the whole wave right after
    if (V == 2) h = __builtin_amdgcn_readfirstlane(h);
    acc += h;
  }
  out[threadIdx.x] = acc;
}
extern "C" int lds_idiom_run(int variant, const int* slots_host, int rounds, long long* out_host) {
  int* slots = nullptr; long long* out = nullptr;
  if (hipMalloc(&slots, 1024 * sizeof(int)) != hipSuccess || hipMalloc(&out, 64 * sizeof(long long)) != hipSuccess) return -1;
  hipMemcpy(slots, slots_host, 1024 * sizeof(int), hipMemcpyHostToDevice);
  if (variant == 0) hipLaunchKernelGGL(k_idiom<0>, dim3(1), dim3(64), 0, 0, slots, rounds, out);
  else if (variant == 1) hipLaunchKernelGGL(k_idiom<1>, dim3(1), dim3(64), 0, 0, slots, rounds, out);
  else hipLaunchKernelGGL(k_idiom<2>, dim3(1), dim3(64), 0, 0, slots, rounds, out);
  int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
  hipMemcpy(out_host, out, 64 * sizeof(long long), hipMemcpyDeviceToHost);
  hipFree(slots); hipFree(out);
  return rc;
}
