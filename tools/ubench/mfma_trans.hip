// Micro-benchmark: what do transcendental (v_exp_f32, v_rcp_f32) and packed
// (v_pk_fma_f32) vector instructions cost on gfx950, alone and beside f32
// MFMAs (same wave / SIMD partner)?  Companion of mfma_valu.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_trans.hip -o /tmp/mfma_trans && /tmp/mfma_trans
// KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_rcp_f32, 3: v_pk_fma_f32 (2 floats).
// Roles as in mfma_valu.hip: 0 idle, 1 MFMA chain, 2 vector chain (NV
// independent instructions per slot), 3 MFMA + NV vector instructions per slot
// in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV>
__device__ __forceinline__ void vec_block(f32x2 (&v)[8], float k) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    f32x2& x = v[i & 7];
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x.x) : "v"(k));
    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x.x));
    if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x.x));
    if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(f32x2{k, k}));
  }
}

template <int ROLE_A, int ROLE_B, int KIND, int NV>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? ROLE_A : ROLE_B;
  f32x16 acc = {0};
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f};
  const float a = 1.0f + threadIdx.x * 1e-9f, b = 1.0f, kk = 0.999f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { vec_block<KIND, NV>(v, kk); __builtin_amdgcn_sched_barrier(0); }
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        vec_block<KIND, NV>(v, kk);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB, int KIND, int NV>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000, grid = 256;
  static const char* kn[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_pk_fma_f32"};
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL((k<RA, RB, KIND, NV>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sa = 0, sb = 0;
  for (int b = 0; b < grid; ++b) { for (int w = 0; w < 4; ++w) sa += h[b * 8 + w]; for (int w = 4; w < 8; ++w) sb += h[b * 8 + w]; }
  const double per = 1.0 / (grid * 4.0) / (iters * 16.0);
  printf("%-40s %-13s x%2d  waves 0-3: %7.1f cyc/slot   waves 4-7: %7.1f cyc/slot\n", name, kn[KIND], NV, sa * per, sb * per);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  run<1, 0, 0, 8>("MFMA chain alone", out, cyc);
  run<2, 0, 0, 8>("vector alone (one wave per SIMD)", out, cyc);
  run<2, 0, 1, 8>("vector alone (one wave per SIMD)", out, cyc);
  run<2, 0, 2, 8>("vector alone (one wave per SIMD)", out, cyc);
  run<2, 0, 3, 8>("vector alone (one wave per SIMD)", out, cyc);
  run<2, 2, 0, 8>("vector on both waves of a SIMD", out, cyc);
  run<2, 2, 1, 8>("vector on both waves of a SIMD", out, cyc);
  run<2, 2, 2, 8>("vector on both waves of a SIMD", out, cyc);
  run<2, 2, 3, 8>("vector on both waves of a SIMD", out, cyc);
  run<3, 0, 0, 8>("MFMA + vector in one wave", out, cyc);
  run<3, 0, 1, 4>("MFMA + vector in one wave", out, cyc);
  run<3, 0, 1, 8>("MFMA + vector in one wave", out, cyc);
  run<3, 0, 2, 8>("MFMA + vector in one wave", out, cyc);
  run<3, 0, 3, 8>("MFMA + vector in one wave", out, cyc);
  run<1, 2, 0, 8>("MFMA wave + vector wave", out, cyc);
  run<1, 2, 1, 8>("MFMA wave + vector wave", out, cyc);
  run<1, 2, 2, 8>("MFMA wave + vector wave", out, cyc);
  run<1, 2, 3, 8>("MFMA wave + vector wave", out, cyc);
  run<3, 3, 1, 8>("MFMA + vector, both waves", out, cyc);
  run<3, 3, 3, 8>("MFMA + vector, both waves", out, cyc);
  return 0;
}
