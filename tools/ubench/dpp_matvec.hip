// Micro-benchmark (round 6): latency of ONE wave's dependent chain of 32 -> 64
// mat-vecs (the fast-generation chain's filter | gate product), per variant:
//   0  LDS broadcast: ds_write x, 8 broadcast ds_read_b128, 32 v_fma (4 chains)
//   1  v_permlane16_swap + 32 v_fmac_f32_dpp row_newbcast (4 chains)
//   2  the same on 8 chains (two of four per 16 inputs)
//   3  32 plain v_fmac (no broadcast at all: the floor of the FMAs alone)
//   4  32 v_readlane_b32 + 32 v_fmac with the SGPR operand
//   5  v_mov_b32_dpp row_newbcast + v_fma (what hipcc emits for update_dpp + fmaf)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_matvec.hip -o /tmp/dpp_matvec && /tmp/dpp_matvec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DPPF(A, W, K) "v_fmac_f32_dpp " A ", %[x], " W " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void mv16(float& a0, float& a1, float& a2, float& a3, float x,
                                     const f32x4& w0, const f32x4& w1, const f32x4& w2, const f32x4& w3) {
  asm volatile("s_nop 1\n\t"
      DPPF("%[a0]", "%[w00]", 0) DPPF("%[a1]", "%[w01]", 1) DPPF("%[a2]", "%[w02]", 2) DPPF("%[a3]", "%[w03]", 3)
      DPPF("%[a0]", "%[w10]", 4) DPPF("%[a1]", "%[w11]", 5) DPPF("%[a2]", "%[w12]", 6) DPPF("%[a3]", "%[w13]", 7)
      DPPF("%[a0]", "%[w20]", 8) DPPF("%[a1]", "%[w21]", 9) DPPF("%[a2]", "%[w22]", 10) DPPF("%[a3]", "%[w23]", 11)
      DPPF("%[a0]", "%[w30]", 12) DPPF("%[a1]", "%[w31]", 13) DPPF("%[a2]", "%[w32]", 14) DPPF("%[a3]", "%[w33]", 15)
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)
      : [x] "v"(x), [w00] "v"(w0[0]), [w01] "v"(w0[1]), [w02] "v"(w0[2]), [w03] "v"(w0[3]),
        [w10] "v"(w1[0]), [w11] "v"(w1[1]), [w12] "v"(w1[2]), [w13] "v"(w1[3]),
        [w20] "v"(w2[0]), [w21] "v"(w2[1]), [w22] "v"(w2[2]), [w23] "v"(w2[3]),
        [w30] "v"(w3[0]), [w31] "v"(w3[1]), [w32] "v"(w3[2]), [w33] "v"(w3[3]));
}
template <int K> __device__ __forceinline__ float bc(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xf, 0xf, true));
}

template <int V>
__global__ __launch_bounds__(64) void k(const f32x4* w, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float inv[64];
  const int lane = threadIdx.x;
  f32x4 q[8];
  for (int c = 0; c < 8; ++c) q[c] = w[c * 64 + lane];
  float x = out[lane & 31];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (V == 0) {
      if (lane < 32) inv[lane] = x;
      __builtin_amdgcn_wave_barrier();
      f32x4 xv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) xv[c] = *reinterpret_cast<const f32x4*>(inv + 4 * c);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        a0 = fmaf(xv[c][0], q[c][0], a0); a1 = fmaf(xv[c][1], q[c][1], a1);
        a2 = fmaf(xv[c][2], q[c][2], a2); a3 = fmaf(xv[c][3], q[c][3], a3);
      }
      __builtin_amdgcn_wave_barrier();
    } else if (V == 1 || V == 2) {
      const auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      const float lo = __uint_as_float(pr[0]), hi = __uint_as_float(pr[1]);
      if (V == 1) {
        mv16(a0, a1, a2, a3, lo, q[0], q[1], q[2], q[3]);
        mv16(a0, a1, a2, a3, hi, q[4], q[5], q[6], q[7]);
      } else {
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        mv16(a0, a1, a2, a3, lo, q[0], q[1], q[2], q[3]);
        mv16(b0, b1, b2, b3, hi, q[4], q[5], q[6], q[7]);
        a0 += b0; a1 += b1; a2 += b2; a3 += b3;
      }
    } else if (V == 3) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        a0 = fmaf(x, q[c][0], a0); a1 = fmaf(x, q[c][1], a1);
        a2 = fmaf(x, q[c][2], a2); a3 = fmaf(x, q[c][3], a3);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
    } else if (V == 4) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        a0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c)), q[c][0], a0);
        a1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c + 1)), q[c][1], a1);
        a2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c + 2)), q[c][2], a2);
        a3 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c + 3)), q[c][3], a3);
      }
    } else if (V == 5) {
      const auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      const float lo = __uint_as_float(pr[0]), hi = __uint_as_float(pr[1]);
#define S4(c, X, o) a0 = fmaf(bc<4*c+0>(X), q[o+c][0], a0); a1 = fmaf(bc<4*c+1>(X), q[o+c][1], a1); a2 = fmaf(bc<4*c+2>(X), q[o+c][2], a2); a3 = fmaf(bc<4*c+3>(X), q[o+c][3], a3);
      S4(0, lo, 0) S4(1, lo, 0) S4(2, lo, 0) S4(3, lo, 0)
      S4(0, hi, 4) S4(1, hi, 4) S4(2, hi, 4) S4(3, hi, 4)
    }
    else if (V == 6 || V == 7 || V == 8) {
      // plain v_fma on 8 / 16 / 1 chains: is an instruction ~5 cycles whatever the dependency distance?
      constexpr int NC = V == 6 ? 8 : (V == 7 ? 16 : 1);
      float ch[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) ch[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        ch[i % NC] = fmaf(x, q[i >> 2][i & 3], ch[i % NC]);
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) asm volatile("" : "+v"(ch[i]));
      if (NC == 1) a0 = ch[0];
      else {
#pragma unroll
        for (int i = 0; i < NC; i += 4) { a0 += ch[i]; a1 += ch[i + 1]; a2 += ch[i + 2]; a3 += ch[i + 3]; }
      }
    } else if (V == 9) {
      // 16 v_pk_fma_f32, x from a VGPR pair (no broadcast): the packed floor
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
      const f32x2 xx = {x, x};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p01) : "v"(xx), "v"(f32x2{q[c][0], q[c][1]}));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p23) : "v"(xx), "v"(f32x2{q[c][2], q[c][3]}));
      }
      a0 = p01[0]; a1 = p01[1]; a2 = p23[0]; a3 = p23[1];
    } else if (V == 10) {
      // 32 v_readlane into SGPR pairs + 16 v_pk_fma_f32 with the SGPR pair as a source
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        f32x2 s01, s23;
        s01[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c));
        s01[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c + 1));
        s23[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c + 2));
        s23[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 4 * c + 3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p01) : "s"(s01), "v"(f32x2{q[c][0], q[c][1]}));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p23) : "s"(s23), "v"(f32x2{q[c][2], q[c][3]}));
      }
      a0 = p01[0]; a1 = p01[1]; a2 = p23[0]; a3 = p23[1];
    } else if (V == 11) {
      // LDS broadcast + 16 v_pk_fma_f32
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      if (lane < 32) inv[lane] = x;
      __builtin_amdgcn_wave_barrier();
      f32x4 xv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) xv[c] = *reinterpret_cast<const f32x4*>(inv + 4 * c);
      f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p01) : "v"(f32x2{xv[c][0], xv[c][1]}), "v"(f32x2{q[c][0], q[c][1]}));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p23) : "v"(f32x2{xv[c][2], xv[c][3]}), "v"(f32x2{q[c][2], q[c][3]}));
      }
      a0 = p01[0]; a1 = p01[1]; a2 = p23[0]; a3 = p23[1];
      __builtin_amdgcn_wave_barrier();
    } else if (V == 12) {
      // 32 v_readlane + 32 s_nop-free scalar consumers: the readlanes alone (sum in SALU)
      float acc = 0.f;
      int si = 0;
#pragma unroll
      for (int c = 0; c < 32; ++c) si ^= __builtin_amdgcn_readlane(__float_as_int(x), c);
      a0 = __int_as_float(si & 0x3fffffff);
    }
    x = ((a0 + a1) + (a2 + a3)) * 0.03f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + lane] = x;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

int main() {
  f32x4* w; float* out; unsigned long long* cyc;
  (void)hipMalloc(&w, 8 * 64 * sizeof(f32x4)); (void)hipMalloc(&out, 128 * 4); (void)hipMalloc(&cyc, 16);
  float hw[8 * 64 * 4];
  for (int i = 0; i < 8 * 64 * 4; ++i) hw[i] = ((i * 7919) % 1000) * 1e-3f - 0.5f;
  (void)hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
  float ho[128]; for (int i = 0; i < 128; ++i) ho[i] = 0.1f * (i % 32);
  const int iters = 10000;
  const char* names[] = {"LDS broadcast + 32 fma", "permlane16_swap + 32 fmac_dpp (4 chains)",
                         "permlane16_swap + 32 fmac_dpp (8 chains)", "32 plain fma (no broadcast)",
                         "32 readlane + 32 fma", "permlane16_swap + 32 mov_dpp + 32 fma",
                         "32 plain fma, 8 chains", "32 plain fma, 16 chains", "32 plain fma, ONE chain",
                         "16 pk_fma (no broadcast)", "32 readlane + 16 pk_fma (SGPR pair)",
                         "LDS broadcast + 16 pk_fma", "32 readlane + s_xor"};
  for (int v = 0; v < 13; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemcpy(out, ho, sizeof(ho), hipMemcpyHostToDevice);
      switch (v) {
        case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 7: hipLaunchKernelGGL(k<7>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 8: hipLaunchKernelGGL(k<8>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 9: hipLaunchKernelGGL(k<9>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 10: hipLaunchKernelGGL(k<10>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 11: hipLaunchKernelGGL(k<11>, 1, 64, 0, 0, w, out, cyc, iters); break;
        case 12: hipLaunchKernelGGL(k<12>, 1, 64, 0, 0, w, out, cyc, iters); break;
      }
      (void)hipDeviceSynchronize();
    }
    unsigned long long hc[2]; float res[128];
    (void)hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(res, out, sizeof(res), hipMemcpyDeviceToHost);
    printf("variant %d  %-46s  %7.1f s_memtime ticks / mat-vec   %7.1f ns / mat-vec   (x[3] = %.6g)\n", v, names[v],
           (double)hc[0] / iters, (double)hc[1] * 10.0 / iters, res[64 + 3]);
  }
  return 0;
}
