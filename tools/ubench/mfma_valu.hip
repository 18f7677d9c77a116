// Micro-benchmark: do f32 MFMA (v_mfma_f32_32x32x2_f32) and VALU work overlap
// on one SIMD (a) inside one wave, (b) across the two waves of a SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
// One workgroup per CU; waves 0-3 run role A, waves 4-7 (their SIMD partners)
// role B.  Roles: 0 idle, 1 MFMA chain (N dependent MFMAs), 2 VALU chain
// (NV v_fma per MFMA slot, independent of the MFMAs), 3 both interleaved in ONE
// wave (MFMA, then NV independent v_fma, repeat); 4 LDS reads (NV ds_read_b128
// per slot, results consumed one slot later); 5 MFMA + NV ds_read_b128
// interleaved in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__device__ __forceinline__ void valu_block(float (&v)[8], float k) {
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], k, 1.0f);
}

template <int ROLE_A, int ROLE_B, int NV>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? ROLE_A : ROLE_B;
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i;
  f32x16 acc = {0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
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
      for (int u = 0; u < 16; ++u) { valu_block<NV>(v, kk); __builtin_amdgcn_sched_barrier(0); }
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        valu_block<NV>(v, kk);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  else if (role == 4 || role == 5) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    f32x4 r[NV > 8 ? 8 : NV];
    constexpr int NR = NV > 8 ? 8 : NV;
    for (int i = 0; i < NR; ++i) r[i] = lp[i * 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (role == 5) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i) t += r[i][0];       // consume last slot's reads
        asm volatile("" : "+v"(t));
        v[0] = t;
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = lp[i * 64 + ((u & 1) << 9)];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB, int NV>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000, grid = 256;
  hipLaunchKernelGGL((k<RA, RB, NV>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<RA, RB, NV>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sa = 0, sb = 0;
  for (int b = 0; b < grid; ++b) { for (int w = 0; w < 4; ++w) sa += h[b * 8 + w]; for (int w = 4; w < 8; ++w) sb += h[b * 8 + w]; }
  const double per = 1.0 / (grid * 4.0) / (iters * 16.0);
  printf("%-46s NV=%2d  waves 0-3: %7.1f cyc/slot   waves 4-7: %7.1f cyc/slot\n", name, NV, sa * per, sb * per);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  run<1, 0, 8>("MFMA chain alone (one wave per SIMD)", out, cyc);
  run<1, 1, 8>("MFMA chain on both waves of a SIMD", out, cyc);
  run<2, 0, 8>("VALU alone", out, cyc);
  run<2, 0, 16>("VALU alone", out, cyc);
  run<2, 2, 8>("VALU on both waves", out, cyc);
  run<1, 2, 4>("MFMA wave + VALU wave", out, cyc);
  run<1, 2, 8>("MFMA wave + VALU wave", out, cyc);
  run<1, 2, 16>("MFMA wave + VALU wave", out, cyc);
  run<3, 0, 4>("MFMA + VALU interleaved in one wave", out, cyc);
  run<3, 0, 8>("MFMA + VALU interleaved in one wave", out, cyc);
  run<3, 0, 16>("MFMA + VALU interleaved in one wave", out, cyc);
  run<3, 3, 8>("interleaved, both waves", out, cyc);
  run<3, 3, 16>("interleaved, both waves", out, cyc);
  run<4, 0, 2>("LDS b128 reads alone", out, cyc);
  run<4, 0, 4>("LDS b128 reads alone", out, cyc);
  run<4, 0, 8>("LDS b128 reads alone", out, cyc);
  run<1, 4, 2>("MFMA wave + LDS-read wave", out, cyc);
  run<1, 4, 4>("MFMA wave + LDS-read wave", out, cyc);
  run<1, 4, 8>("MFMA wave + LDS-read wave", out, cyc);
  run<5, 0, 2>("MFMA + LDS reads in one wave", out, cyc);
  run<5, 0, 4>("MFMA + LDS reads in one wave", out, cyc);
  run<5, 0, 8>("MFMA + LDS reads in one wave", out, cyc);
  run<5, 5, 4>("MFMA + LDS reads, both waves", out, cyc);
  return 0;
}
