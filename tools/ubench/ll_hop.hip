// Micro-benchmark (round 6): one-way latency of an 8-byte {payload, step}
// hand-over word between two resident workgroups, by placement (same XCD =
// blocks 0 and 8, other XCD = blocks 0 and 1) and by the scope of the store /
// poll (agent = sc1: what fg_persist_kernel uses; workgroup = sc0: L1 bypassed,
// the XCD's own L2 serves it -- only meaningful on the same XCD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ll_hop.hip -o /tmp/ll_hop && /tmp/ll_hop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

template <int SCOPE>
__global__ __launch_bounds__(64) void k(u64* words, unsigned* xcc, int partner, int iters, u64* cyc, int* bad) {
  const int b = blockIdx.x;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) xcc[b] = x & 0xf;
  if (b != 0 && b != partner) return;
  const int lane = threadIdx.x;
  u64* mine = words + (b == 0 ? 0 : 64) + lane;      // the word this side WRITES
  u64* theirs = words + (b == 0 ? 64 : 0) + lane;    // the word this side POLLS
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  int errs = 0;
  for (int i = 1; i <= iters; ++i) {
    if (b == 0) {
      __hip_atomic_store(mine, ((u64)i << 32) | (unsigned)(i * 3 + lane), __ATOMIC_RELAXED, SCOPE);
      u64 w;
      unsigned spins = 0;
      do { w = __hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE); } while ((unsigned)(w >> 32) != (unsigned)i && ++spins < 200000u && !*(volatile int*)bad);
      if ((unsigned)(w >> 32) != (unsigned)i) { atomicAdd(bad, 1000000); break; }
      if ((unsigned)w != (unsigned)(i * 5 + lane)) ++errs;
    } else {
      u64 w;
      unsigned spins = 0;
      do { w = __hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE); } while ((unsigned)(w >> 32) != (unsigned)i && ++spins < 200000u && !*(volatile int*)bad);
      if ((unsigned)(w >> 32) != (unsigned)i) { atomicAdd(bad, 1000000); break; }
      if ((unsigned)w != (unsigned)(i * 3 + lane)) ++errs;
      __hip_atomic_store(mine, ((u64)i << 32) | (unsigned)(i * 5 + lane), __ATOMIC_RELAXED, SCOPE);
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) cyc[0] = t1 - t0;
  if (errs) atomicAdd(bad, errs);
}

int main() {
  u64 *words, *cyc; unsigned* xcc; int* bad;
  (void)hipMalloc(&words, 128 * 8); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&xcc, 64 * 4); (void)hipMalloc(&bad, 4);
  const int iters = 4000;
  for (int scope = 0; scope < 2; ++scope)
    for (int partner : {8, 1, 16, 9}) {
      (void)hipMemset(words, 0, 128 * 8); (void)hipMemset(bad, 0, 4);
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(words, 0, 128 * 8);
        if (scope == 0) hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_AGENT>), 32, 64, 0, 0, words, xcc, partner, iters, cyc, bad);
        else hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_WORKGROUP>), 32, 64, 0, 0, words, xcc, partner, iters, cyc, bad);
        (void)hipDeviceSynchronize();
      }
      u64 hc; unsigned hx[64]; int hb;
      (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
      (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("%-9s scope, blocks 0 (XCD %u) <-> %2d (XCD %u): %.3f us per one-way hop  (payload errors %d; >= 1000000: a poll never saw its word)\n",
             scope == 0 ? "agent" : "workgroup", hx[0], partner, hx[partner], (double)hc * 0.01 / iters / 2, hb);
      fflush(stdout);
    }
  return 0;
}
