// What a persistent multi-CU fast-generation launch would trade: kernel
// boundaries inside a replayed hipGraph against in-launch hand-overs between
// workgroups (payload + flag through memory).  Two numbers per variant:
//   * boundary: a graph of 200 x 4 dependent one-workgroup kernels (each does a
//     32-float read-modify-write of the previous kernel's output), time per node;
//   * hop: two resident workgroups ping-pong a 128-byte payload + flag
//     (sc1 stores, s_waitcnt vmcnt(0), relaxed agent flag; the consumer polls
//     the flag and reads the payload with sc1 loads), time per one-way hop, for
//     partners on the same XCD (workgroups 0 and 8) and on different XCDs (0, 1).
//   hipcc --offload-arch=gfx950 -O3 -o hop tools/ubench/hop_vs_boundary.hip && ./hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); exit(1); } } while (0)

__global__ void node_kernel(float* buf, int step) {
  const int t = threadIdx.x;
  if (t < 32) buf[(step + 1) * 32 + t] = buf[step * 32 + t] * 1.0001f + 1.f;
}

__global__ __launch_bounds__(64) void pingpong_kernel(float* payload, unsigned* flags, int partner_b,
                                                      int rounds, unsigned long long* out) {
  // workgroup 0 <-> workgroup partner_b; the others exit
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_b ? 1 : -1);
  if (me < 0) return;
  const int lane = threadIdx.x;
  float* mine = payload + me * 32;          // what I publish
  float* theirs = payload + (1 - me) * 32;
  unsigned* my_flag = flags + me * 32;      // (separate 128-byte lines)
  unsigned* their_flag = flags + (1 - me) * 32;
  float v = (float)lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 1; r <= rounds; ++r) {
    // 0 sends first, 1 answers
    if (me == 1) {
      while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r)
        __builtin_amdgcn_s_sleep(1);
      float x = 0.f;
      if (lane < 32) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(theirs + lane) : "memory");
      v = x + 1.f;
    }
    if (lane < 32) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(mine + lane), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(my_flag, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (me == 0) {
      while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r)
        __builtin_amdgcn_s_sleep(1);
      float x = 0.f;
      if (lane < 32) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(theirs + lane) : "memory");
      v = x + 1.f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { out[me * 2] = t1 - t0; out[me * 2 + 1] = (unsigned long long)v; }
}

int main() {
  float* buf; unsigned* flags; unsigned long long* out; float* payload;
  CK(hipMalloc(&buf, 32 * 4 * 1024));
  CK(hipMalloc(&payload, 4096));
  CK(hipMalloc(&flags, 4096));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, 32 * 4 * 1024));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  // ---- kernel boundaries inside a replayed graph
  const int NODES = 800;
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < NODES; ++i) hipLaunchKernelGGL(node_kernel, dim3(1), dim3(64), 0, s, buf, i);
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  const int REP = 10;
  for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(exec, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("graph replay, %d dependent one-workgroup kernels: %.2f us per kernel (launch + boundary)\n",
         NODES, ms * 1e3 / (REP * NODES));
  // ---- in-launch hops
  for (int partner : {8, 1}) {
    CK(hipMemset(flags, 0, 4096));
    const int rounds = 2000;
    hipLaunchKernelGGL(pingpong_kernel, dim3(16), dim3(64), 0, s, payload, flags, partner, rounds, out);
    CK(hipStreamSynchronize(s));
    unsigned long long h[4];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("ping-pong workgroups 0 <-> %d (%s XCD): %.2f us per one-way hop (payload 128 B sc1 + flag), check %llu\n",
           partner, partner == 8 ? "same" : "other", h[0] / 100.0 / (2.0 * rounds), h[1]);
  }
  return 0;
}
