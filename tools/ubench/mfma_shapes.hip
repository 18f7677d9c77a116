// Micro-benchmark: f32 MFMA shapes on random data -- v_mfma_f32_32x32x2_f32
// (4 accumulators of 16 registers = a 64 x 64 tile per wave) against
// v_mfma_f32_16x16x4_f32 (16 accumulators of 4 registers = the same tile):
// cycles per FLOP are equal by the data sheet; does the chip hold the same
// clock on both (MI355X_MICROARCH.md, DVFS give-back item 7)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters,
                                         unsigned long long* cyc) {
  const int tid = threadIdx.x;
  // operands: 8 A and 8 B values per lane, random
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(blockIdx.x * 256 + tid) * 16 + i];
    b[i] = in[(blockIdx.x * 256 + tid) * 16 + 8 + i];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {      // 8 k-steps of 2: 32 MFMAs = 64x64x16
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[(u + 1) & 7], acc[3], 0, 0, 0);
      }
    }
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
  } else {
    f32x4 acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // 4 k-steps of 4: 64 MFMAs = 64x64x16
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                a[(u + m) & 7], b[(u + n) & 7], acc[m * 4 + n], 0, 0, 0);
      }
    }
    for (int q = 0; q < 16; ++q) for (int i = 0; i < 4; ++i) s += acc[q][i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
double run(const float* in, float* out, unsigned long long* cyc, int wgs, int iters, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE>), dim3(wgs), dim3(256), 0, 0, in, out, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE>), dim3(wgs), dim3(256), 0, 0, in, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(wgs);
  hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += v; c /= wgs;
  const double flops = (double)wgs * 4 /*waves*/ * iters * 2.0 * 64 * 64 * 16 * reps;
  const double tf = flops / (ms * 1e-3) / 1e12;
  const double wave_cyc_per_iter = c / iters;
  printf("%-22s wgs %5d: %7.1f TFLOP/s   %7.1f cycles per 64x64x16 per wave (in-kernel clock estimate %.2f GHz)\n",
         name, wgs, tf, wave_cyc_per_iter, c / (ms / reps * 1e-3) / 1e9);
  return tf;
}

int main() {
  const int maxwg = 256 * 8;
  float* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, (size_t)maxwg * 256 * 16 * 4); hipMalloc(&out, (size_t)maxwg * 256 * 4); hipMalloc(&cyc, maxwg * 8);
  std::vector<float> h((size_t)maxwg * 256 * 16);
  srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int wpc : {1, 2, 4}) {          // workgroups (of 4 waves) per CU = waves per SIMD
    run<32>(in, out, cyc, 256 * wpc, 20000 / wpc, "32x32x2  (4 acc)");
    run<16>(in, out, cyc, 256 * wpc, 20000 / wpc, "16x16x4  (16 acc)");
  }
  return 0;
}
