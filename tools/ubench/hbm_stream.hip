// Practical HBM streaming rates of this part for the read : write mixes of the
// two persistent stack launches (forward 1 : 3 planes, backward 7 : 2), next to
// read-only, write-only and copy -- hand-written streams (16-byte accesses,
// 2048 workgroups of 256 threads striding over planes much larger than the
// 256 MB Infinity Cache), so that the "fraction of the streaming rate" in
// DESIGN.md does not rest on a library copy kernel.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_stream tools/ubench/hbm_stream.hip && ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ src,
                                                     f32x4* __restrict__ dst, long n4) {
  // R source planes and W destination planes of n4 16-byte elements each
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f32x4 v = NT ? __builtin_nontemporal_load(src + r * n4 + i) : src[r * n4 + i];
      acc += v;
    }
    if (W == 0) {
      if (acc[0] == 123456.f) dst[i] = acc;          // never true: keeps the loads
    } else {
#pragma unroll
      for (int w = 0; w < W; ++w) {
        acc[0] += 1.f;
        if (NT) __builtin_nontemporal_store(acc, dst + w * n4 + i);
        else dst[w * n4 + i] = acc;
      }
    }
  }
}

template <int R, int W, bool NT>
static void run(const char* name, const f32x4* src, f32x4* dst, long n4) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((stream_kernel<R, W, NT>), dim3(2048), dim3(256), 0, 0, src, dst, n4);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((stream_kernel<R, W, NT>), dim3(2048), dim3(256), 0, 0, src, dst, n4);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(R + W) * n4 * 16.0 * reps;
  printf("%-28s %d read : %d written planes of %.0f MB  %7.1f us  %.2f TB/s\n", name, R, W,
         n4 * 16.0 / 1e6, ms / reps * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const long n4 = 32L << 20;               // 512 MB per plane
  f32x4 *src = nullptr, *dst = nullptr;
  if (hipMalloc(&src, 7 * n4 * 16) != hipSuccess || hipMalloc(&dst, 3 * n4 * 16) != hipSuccess) {
    printf("allocation failed\n");
    return 1;
  }
  CK(hipMemset(src, 0, 7 * n4 * 16));
  CK(hipMemset(dst, 0, 3 * n4 * 16));
  run<1, 0, false>("read only", src, dst, n4);
  run<4, 0, false>("read only, 4 planes", src, dst, n4);
  run<0, 1, false>("write only", src, dst, n4);
  run<0, 1, true>("write only, nt", src, dst, n4);
  run<1, 1, false>("copy", src, dst, n4);
  run<1, 1, true>("copy, nt", src, dst, n4);
  run<1, 3, false>("forward stack mix", src, dst, n4);
  run<1, 3, true>("forward stack mix, nt", src, dst, n4);
  run<7, 2, false>("backward stack mix", src, dst, n4);
  run<7, 2, true>("backward stack mix, nt", src, dst, n4);
  // round 4: the backward's own dx rows are ONE plane rewritten in place (cache
  // resident): what still streams is 6 planes read, 1 written
  run<6, 1, false>("backward stack mix r4", src, dst, n4);
  run<6, 1, true>("backward stack mix r4, nt", src, dst, n4);
  CK(hipFree(src));
  CK(hipFree(dst));
  return 0;
}
