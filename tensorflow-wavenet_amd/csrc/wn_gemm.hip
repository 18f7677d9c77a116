// fp32 MFMA GEMMs of the WaveNet stack (v_mfma_f32_32x32x2_f32, exact fp32).
//
//  wn_gemm_nn : C[M,N] = epi(A[M,K] * W[K,N])      rows M = B*T audio samples
//     - the 50 skip 1x1 convs as ONE contraction over the z planes
//       (sum(outputs), wavenet/model.py:303-305 + 430:  K = L*32, N = S)
//     - postprocess1 / postprocess2 (model.py:432-440), ReLU/bias fused
//     - backward-data of the same (dY * W^T with the ReLU mask fused)
//  wn_gemm_tn : dW[Mw,Nw] = sum_rows A[row,:]^T G[row,:]   (weight grads),
//     split over rows into slabs that wn_reduce_slabs sums in a fixed order
//     (deterministic; no float atomics).  A may be dense, the z planes, or a
//     one-hot generated on the fly from the mu-law codes (causal-layer grad,
//     model.py:227-234 / 518-531: the one-hot tensor is never materialised).
//
// Orientation: "time on the MFMA N axis" (see wn_common.h): activations are
// the B operand as 32x32 fragments, weights the A operand from LDS.
//
// Kernels in this file
//   gemm_nn3_kernel      default NN: LDS-DMA staging, 4 workgroups / CU
//   gemm_nn_kernel       NN fallback (K % 16 != 0): register
//                        staging, K chunks of 32, every edge predicated
//   gemm_tn3_kernel      default TN for wide outputs (dWs / dW1 / dW2)
//   gemm_tn2_kernel      its register-staged fallback (ragged row counts)
//   gemm_tn_kernel       LDS-free TN for narrow / one-hot operands
//   gemm_nn_split_kernel, gemm_tn_split_kernel, split_w_kernel
//                        opt-in split-bf16 products (wn_gemm_*_split)
//   reduce_slabs_kernel, transpose_pad_kernel, mfma_peak_kernel
#include "wn_common.h"
#include <cstdlib>
#include <cstddef>
#include <cstring>

// ---------------------------------------------------------------------------
// NN
// ---------------------------------------------------------------------------
struct GemmNN {
  const float* A;
  long lda;             // dense mode row stride (floats)
  long a_plane_stride;  // plane mode: floats between planes
  int a_planes;         // 0 = dense [M][lda]; >0 = planes [a_planes][M][32]
  const float* W;       // [K][ldw]
  int ldw;
  const float* bias;    // [N] or null
  const float* mask;    // [M][ld_mask] multiply by (mask > 0) or null
  long ld_mask;
  const float* addend;  // [M][ld_add] added after relu/mask, or null
  long ld_add;
  float* C;
  long ldc;
  long c_plane_stride;
  int c_planes;         // 0 = dense; >0 = planes [c_planes][M][32]
  float* Cpre;          // optional pre-activation copy (dense, ldc) or null
  long M;
  int N, K;
  int relu;
  int tiles_n;
  int nwg;
  int tiles_m;
  int tile_rows;        // rows a tile OWNS (<= 128; 128 = all it computes)
};

// address-space-qualified pointers of __builtin_amdgcn_global_load_lds
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define NN_TM 128
#define NN_TN 128
#define NN_KC 32
#ifndef NN3_PF
#define NN3_PF 2   // steps the NN3 loop's weight-operand reads run ahead of their MFMAs
#endif

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "contiguous chunk per XCD" remap (8 XCDs, round-robin dispatch)
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Coalesced epilogue shared by the NN kernels.  An
// accumulator fragment store touches 32 rows x 32 B per instruction; instead
// each wave passes its 64 x 64 result through a private LDS tile (two halves
// of 32 rows, row stride 68 floats: conflict-free both ways) and writes / reads
// C, Cpre, mask and addend as 256-byte row segments (lane -> row l>>4, 16-byte
// chunk l&15).  Must be called after a __syncthreads() that retires every
// read of the operand tiles (the LDS is reused).
#define EP_LD 68
// cache policy of the epilogue's stores / mask + addend loads (2 = nt hint)
// (outputs and mask / addend operands are touched once per launch: with the
// hint they do not push the reused operands out of the L2; six NN launches
// 4606 -> 4580 us, the 820 MB dZ output alone 1558 -> 1531)
#ifndef EP_ST_AUX
#define EP_ST_AUX 2
#endif
#ifndef EP_LD_AUX
#define EP_LD_AUX 2
#endif
// Round 3: every global access of the epilogue goes through a buffer resource
// whose base is the wave's corner of the tile (64-bit arithmetic once per
// 32-row half, on the scalar unit), the lane's part of the address is one
// 32-bit VGPR per operand and the row step an SGPR: the first version spent
// ~1000 vector instructions per wave and tile on 64-bit address arithmetic
// (128 v_mul_lo_u32, 270 64-bit adds / mads) -- on gfx950 vector instructions
// of a wave beside other workgroups' f32 MFMAs run at a quarter of their rate
// (DESIGN.md 3b), and the epilogue took 22 - 30 us per tile.  Rows are handled
// four at a time: LDS reads and mask / addend loads first, then the stores.
typedef unsigned ep_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ep_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void gemm_epilogue(const GemmNN& g, f32x16 (&acc)[2][2],
                                              float* lds, long m0, int n0,
                                              int wm, int wn, int wave, int lane,
                                              long m_end = -1) {
  if (m_end < 0) m_end = g.M;
  const int j = lane & 31, h = lane >> 5;
  float* tile = lds + wave * (32 * EP_LD);
  const int rr = lane >> 4, cc = (lane & 15) * 4;   // row-in-group, column
  const int nw = n0 + wn * 64;                      // the wave's first column
  const bool ncol = nw + cc < g.N;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias && ncol) bias4 = *reinterpret_cast<const f32x4*>(g.bias + nw + cc);
  // per-lane byte offsets inside the wave's 32-row half
  const int vC = g.c_planes ? (int)(((long)(cc >> 5) * g.c_plane_stride + rr * 32 + (cc & 31)) * 4)
                            : (int)((rr * g.ldc + cc) * 4);
  const int sC = g.c_planes ? 4 * 32 * 4 : (int)(4 * g.ldc * 4);   // four rows further
  const int vP = (int)((rr * g.ldc + cc) * 4);
  const int sP = (int)(4 * g.ldc * 4);
  const int vM = (int)((rr * g.ld_mask + cc) * 4), sM = (int)(4 * g.ld_mask * 4);
  // (addend in the output's plane layout when ld_add == 0 and c_planes > 0:
  // x_{l+1} planes = x_l planes + z_l Wd of the channel-block models)
  const bool addp = g.addend && g.c_planes && g.ld_add == 0;
  const int vA = addp ? vC : (int)((rr * g.ld_add + cc) * 4);
  const int sA = addp ? sC : (int)(4 * g.ld_add * 4);
#pragma unroll
  for (int fm = 0; fm < 2; ++fm) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[fn][fm][4 * q], acc[fn][fm][4 * q + 1],
                   acc[fn][fm][4 * q + 2], acc[fn][fm][4 * q + 3]};
        *reinterpret_cast<f32x4*>(tile + j * EP_LD + fn * 32 + 8 * q + 4 * h) = v;
      }
    __builtin_amdgcn_wave_barrier();
    const long mbase = m0 + wm * 64 + fm * 32;
    const long left = m_end - mbase;                 // rows of this half that exist
    const int nrows = left > 32 ? 32 : (left < 0 ? 0 : (int)left);
    const __amdgpu_buffer_rsrc_t rC = ep_rsrc(
        g.c_planes ? g.C + (long)(nw >> 5) * g.c_plane_stride + mbase * 32
                   : g.C + mbase * g.ldc + nw);
    const __amdgpu_buffer_rsrc_t rP = ep_rsrc(g.Cpre ? g.Cpre + mbase * g.ldc + nw : g.C);
    const __amdgpu_buffer_rsrc_t rM = ep_rsrc(g.mask ? g.mask + mbase * g.ld_mask + nw : g.C);
    const __amdgpu_buffer_rsrc_t rA = ep_rsrc(
        !g.addend ? g.C
        : addp ? g.addend + (long)(nw >> 5) * g.c_plane_stride + mbase * 32
               : g.addend + mbase * g.ld_add + nw);
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      f32x4 v[4], mk[4], ad[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int it = bt * 4 + q, row = it * 4 + rr;
        const bool ok = ncol && row < nrows;
        v[q] = *reinterpret_cast<const f32x4*>(tile + row * EP_LD + cc);
        if (g.mask && ok)
          mk[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rM, vM, it * sM, EP_LD_AUX));
        if (g.addend && ok)
          ad[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, vA, it * sA, EP_LD_AUX));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int it = bt * 4 + q, row = it * 4 + rr;
        if (!(ncol && row < nrows)) continue;
        f32x4 x = v[q] + bias4;
        if (g.Cpre)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ep_u32x4, x), rP, vP, it * sP, EP_ST_AUX);
        if (g.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        if (g.mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = mk[q][e] > 0.f ? x[e] : 0.f;
        }
        if (g.addend) x += ad[q];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ep_u32x4, x), rC, vC, it * sC, EP_ST_AUX);
      }
    }
  }
}

// The same for a wave tile of 32 rows x 128 columns (gemm_nn_split_kernel: wave
// w owns rows 32 w .. 32 w + 31 of the workgroup tile and all 128 columns, as
// 2 x 8 accumulator tiles of 16 x 16): two passes of 64 columns through the
// wave's LDS tile.
// (accumulator register i of lane l of tile [mt][nt]: row 16 mt + (l & 15),
// column 16 nt + 4 (l >> 4) + i)
__device__ __forceinline__ void gemm_epilogue_wide(const GemmNN& g, f32x4 (&acc)[2][8],
                                                   float* lds, long m0, int n0,
                                                   int wave, int lane) {
  const long m_end = g.M;
  float* tile = lds + wave * (32 * EP_LD);
  const int rr = lane >> 4, cc = (lane & 15) * 4;   // row-in-group, column
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
  const int nw = n0 + cb * 64;                      // the pass's first column
  const bool ncol = nw + cc < g.N;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias && ncol) bias4 = *reinterpret_cast<const f32x4*>(g.bias + nw + cc);
  // per-lane byte offsets inside the wave's 32-row half
  const int vC = g.c_planes ? (int)(((long)(cc >> 5) * g.c_plane_stride + rr * 32 + (cc & 31)) * 4)
                            : (int)((rr * g.ldc + cc) * 4);
  const int sC = g.c_planes ? 4 * 32 * 4 : (int)(4 * g.ldc * 4);   // four rows further
  const int vP = (int)((rr * g.ldc + cc) * 4);
  const int sP = (int)(4 * g.ldc * 4);
  const int vM = (int)((rr * g.ld_mask + cc) * 4), sM = (int)(4 * g.ld_mask * 4);
  // (addend in the output's plane layout when ld_add == 0 and c_planes > 0:
  // x_{l+1} planes = x_l planes + z_l Wd of the channel-block models)
  const bool addp = g.addend && g.c_planes && g.ld_add == 0;
  const int vA = addp ? vC : (int)((rr * g.ld_add + cc) * 4);
  const int sA = addp ? sC : (int)(4 * g.ld_add * 4);
  {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ntl = 0; ntl < 4; ++ntl)
        *reinterpret_cast<f32x4*>(tile + (mt * 16 + (lane & 15)) * EP_LD + ntl * 16 +
                                  4 * (lane >> 4)) = acc[mt][4 * cb + ntl];
    __builtin_amdgcn_wave_barrier();
    const long mbase = m0 + wave * 32;
    const long left = m_end - mbase;                 // rows of this half that exist
    const int nrows = left > 32 ? 32 : (left < 0 ? 0 : (int)left);
    const __amdgpu_buffer_rsrc_t rC = ep_rsrc(
        g.c_planes ? g.C + (long)(nw >> 5) * g.c_plane_stride + mbase * 32
                   : g.C + mbase * g.ldc + nw);
    const __amdgpu_buffer_rsrc_t rP = ep_rsrc(g.Cpre ? g.Cpre + mbase * g.ldc + nw : g.C);
    const __amdgpu_buffer_rsrc_t rM = ep_rsrc(g.mask ? g.mask + mbase * g.ld_mask + nw : g.C);
    const __amdgpu_buffer_rsrc_t rA = ep_rsrc(
        !g.addend ? g.C
        : addp ? g.addend + (long)(nw >> 5) * g.c_plane_stride + mbase * 32
               : g.addend + mbase * g.ld_add + nw);
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      f32x4 v[4], mk[4], ad[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int it = bt * 4 + q, row = it * 4 + rr;
        const bool ok = ncol && row < nrows;
        v[q] = *reinterpret_cast<const f32x4*>(tile + row * EP_LD + cc);
        if (g.mask && ok)
          mk[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rM, vM, it * sM, EP_LD_AUX));
        if (g.addend && ok)
          ad[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, vA, it * sA, EP_LD_AUX));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int it = bt * 4 + q, row = it * 4 + rr;
        if (!(ncol && row < nrows)) continue;
        f32x4 x = v[q] + bias4;
        if (g.Cpre)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ep_u32x4, x), rP, vP, it * sP, EP_ST_AUX);
        if (g.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        if (g.mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = mk[q][e] > 0.f ? x[e] : 0.f;
        }
        if (g.addend) x += ad[q];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ep_u32x4, x), rC, vC, it * sC, EP_ST_AUX);
      }
    }
  }
  }
}

template <int WMT>   // time-waves per workgroup: 2 -> 128-row tile, 256 threads
__global__ __launch_bounds__(WMT * 128) void gemm_nn_kernel(GemmNN g) {
  constexpr int TM = WMT * 64, NT = WMT * 128;   // rows per tile, threads
  constexpr int NA = TM * 8 / NT, NB = 1024 / NT; // float4 per thread (A, W)
  // one buffer: the epilogue's per-wave tiles (WMT*2*32*EP_LD floats) reuse it
  __shared__ __attribute__((aligned(16))) float smem[2 * TM * NN_KC + 2 * NN_KC * NN_TN];
  float (*As)[TM * NN_KC] = reinterpret_cast<float (*)[TM * NN_KC]>(smem);
  float (*Bs)[NN_KC * NN_TN] =
      reinterpret_cast<float (*)[NN_KC * NN_TN]>(smem + 2 * TM * NN_KC);
  static_assert(WMT * 2 * 32 * EP_LD <= 2 * TM * NN_KC + 2 * NN_KC * NN_TN, "epilogue LDS");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave % WMT, wn = wave / WMT;
  const int logical = xcd_remap(blockIdx.x, g.nwg);
  const int tile_n = logical % g.tiles_n;
  const long m0 = (long)(logical / g.tiles_n) * TM;
  const int n0 = tile_n * NN_TN;
  const int nk = (g.K + NN_KC - 1) / NN_KC;
  f32x4 ra[NA], rb[NB];
  const bool full_m = m0 + TM <= g.M, full_n = n0 + NN_TN <= g.N;
  auto gload = [&](int kc) {
    const bool full_k = (kc + 1) * NN_KC <= g.K;
    if (full_m && full_k) {  // interior tile: no predicates, no branches
      const float* pa = g.a_planes
                            ? g.A + (long)kc * g.a_plane_stride + m0 * 32
                            : g.A + m0 * g.lda + kc * NN_KC;
      const long ld = g.a_planes ? 32 : g.lda;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        ra[i] = *reinterpret_cast<const f32x4*>(
            pa + (long)((tid >> 3) + (NT / 8) * i) * ld + (tid & 7) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int row = (tid >> 3) + (NT / 8) * i, ch = tid & 7;
        const long m = m0 + row;
        const int k = kc * NN_KC + ch * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < g.M && k < g.K) {
          const float* p = g.a_planes
                               ? g.A + (long)kc * g.a_plane_stride + m * 32 + ch * 4
                               : g.A + m * g.lda + k;
          v = *reinterpret_cast<const f32x4*>(p);
        }
        ra[i] = v;
      }
    }
    if (full_n && full_k) {
      const float* pb = g.W + (long)kc * NN_KC * g.ldw + n0;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        rb[i] = *reinterpret_cast<const f32x4*>(
            pb + (long)((tid >> 5) + (NT / 32) * i) * g.ldw + (tid & 31) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int kk = (tid >> 5) + (NT / 32) * i, c4 = tid & 31;
        const int k = kc * NN_KC + kk, n = n0 + c4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < g.K && n < g.N)
          v = *reinterpret_cast<const f32x4*>(g.W + (long)k * g.ldw + n);
        rb[i] = v;
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (tid >> 3) + (NT / 8) * i, ch = tid & 7;
      const int sw = ch ^ ((row >> 1) & 7);  // 16-byte chunk swizzle
      *reinterpret_cast<f32x4*>(&As[buf][row * NN_KC + sw * 4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kk = (tid >> 5) + (NT / 32) * i, c4 = tid & 31;
      *reinterpret_cast<f32x4*>(&Bs[buf][kk * NN_TN + c4 * 4]) = rb[i];
    }
  };

  f32x16 acc[2][2];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = frag_zero();

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) gload(kc + 1);
    // fragments of the activation tile (B operand)
    f32x16 fa[2];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
      const int row = wm * 64 + fm * 32 + j;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sw = (2 * q + h) ^ ((j >> 1) & 7);
        const f32x4 v =
            *reinterpret_cast<const f32x4*>(&As[buf][row * NN_KC + sw * 4]);
        fa[fm][4 * q + 0] = v[0];
        fa[fm][4 * q + 1] = v[1];
        fa[fm][4 * q + 2] = v[2];
        fa[fm][4 * q + 3] = v[3];
      }
    }
    const float* bl = &Bs[buf][4 * h * NN_TN + wn * 64 + j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ko = (8 * (r >> 2) + (r & 3)) * NN_TN;
      const float w0 = bl[ko], w1 = bl[ko + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, fa[0][r], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, fa[1][r], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, fa[0][r], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, fa[1][r], acc[1][1], 0, 0, 0);
    }
    if (kc + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // coalesced epilogue through LDS (the K loop ended with a barrier)
  gemm_epilogue(g, acc, smem, m0, n0, wm, wn, wave, lane);
}

// ---------------------------------------------------------------------------
// LDS-DMA variant (the default): same 128 x 128 output tile and wave
// decomposition as gemm_nn_kernel, but
//  * operands are staged by global_load_lds_dwordx4 (no staging registers, no
//    ds_write pass) in K chunks of 16, two 16 KB stages: the DMA of chunk k+1
//    is issued right after the barrier that opens chunk k;
//  * 35 KB of LDS and < 128 registers: four workgroups per CU, so other
//    workgroups' MFMAs cover this one's first fetch, barriers and output
//    store (the register-staged kernel fits two and loses a fixed ~2.6 chunk
//    times per tile: efficiency(K) = K / (1.19 K + 82) of the MFMA peak);
//  * s_setprio(1) around a wave's MFMA block.
// Measured (tools/kbench.py nnexp, B*T = 128000): 135 / 130 / 126 / 122 / 129
// TFLOP/s on the five shapes of a training step vs 127 / 119 / 116 / 108 /
// 116 for the register-staged kernel and 130 / 136 / 133 / 124 / 132 for the
// vendor library (torch.mm) on the same device.
// One wave-instruction writes 1 KiB of LDS linearly (lane l -> base + 16 l),
// so the XOR swizzle of the activation image is applied to the per-lane
// SOURCE address: LDS slot s of row r (64-byte rows) holds the global 16-byte
// chunk s ^ ((r >> 2) & 3).  Rows / columns past the edge are clamped to valid
// addresses: they only feed outputs that are never stored.  Needs K % 16 == 0.
// ---------------------------------------------------------------------------
#define N3_KC 16
#define N3_STAGE (NN_TM * N3_KC + N3_KC * NN_TN)   // floats per stage

// fp32 accuracy from bf16 matrix instructions (opt-in, WN_GEMM_MODE=bf16x6 /
// bf16x9 / bf16x3): every fp32 operand is split EXACTLY into three bf16 pieces
// x = x1 + x2 + x3 (8 + 8 + 8 significant bits, by truncation: two AND, two
// exact SUB), and the fp32 product is rebuilt from piecewise products on
// v_mfma_f32_32x32x16_bf16 (exact 16-bit products, fp32 accumulation):
//   x6: x1y1 + x1y2 + x2y1 + x1y3 + x3y1 + x2y2   (dropped terms <= 3 * 2^-24 |xy|)
//   x9: all nine (error-free products);  x3: x1y1 + x1y2 + x2y1 (~2^-16 |xy|).
// The bf16 instruction does 8x the MACs of v_mfma_f32_32x32x2_f32 in half the
// cycles, so x6 costs 6/16 of the fp32 MFMA time plus ~5.5 VALU per element.
// NOT the default: BASELINE's configuration names fp32 arithmetic, and this
// changes the summation order and the instruction class.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Split3 {
  bf16x8 p[3];  // high, middle, low pieces of 8 consecutive-k values
};

__device__ __forceinline__ Split3 split3(const float (&v)[8]) {
  unsigned int hi[8], mi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned int u = __float_as_uint(v[e]) & 0xffff0000u;
    const float r = v[e] - __uint_as_float(u);           // exact
    const unsigned int m = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(m);              // exact, <= 8 bits
    hi[e] = u;
    mi[e] = m;
    lo[e] = __float_as_uint(r2);
  }
  Split3 s;
  u32x4 ph, pm, pl;
#pragma unroll
  for (int w = 0; w < 4; ++w) {  // pack the top halves of two floats
    ph[w] = __builtin_amdgcn_perm(hi[2 * w + 1], hi[2 * w], 0x07060302u);
    pm[w] = __builtin_amdgcn_perm(mi[2 * w + 1], mi[2 * w], 0x07060302u);
    pl[w] = __builtin_amdgcn_perm(lo[2 * w + 1], lo[2 * w], 0x07060302u);
  }
  s.p[0] = __builtin_bit_cast(bf16x8, ph);
  s.p[1] = __builtin_bit_cast(bf16x8, pm);
  s.p[2] = __builtin_bit_cast(bf16x8, pl);
  return s;
}

// acc += A (rows on lanes) * B (columns on lanes) with NPROD piece products,
// smallest terms first
template <int NPROD>
__device__ __forceinline__ void mma_split(f32x16& acc, const Split3& a, const Split3& b) {
#define WN_MM(i, k) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[i], b.p[k], acc, 0, 0, 0)
  if (NPROD >= 9) { WN_MM(2, 2); WN_MM(1, 2); WN_MM(2, 1); }
  if (NPROD >= 6) { WN_MM(1, 1); WN_MM(0, 2); WN_MM(2, 0); }
  WN_MM(0, 1);
  WN_MM(1, 0);
  WN_MM(0, 0);
#undef WN_MM
}

// One 128 x 128 output tile of C = epi(A W): tile column `tile_n`, rows from m0
// (the tile owns [m0, m_end)).
// Waves whose 64 columns lie entirely beyond N (the second half of the last
// column tile when N % 128 == 64: the dZ GEMM's N = 1600) stage and
// synchronise but issue no MFMAs: 3.8 % of that launch's matrix work.
template <bool STAMPS>
__device__ __forceinline__ void nn3_tile(const GemmNN& g, float* smem, int tile_n, long m0,
                                         long m_end, unsigned long long* dbg) {
#define NSTAMP(i) if (STAMPS && threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memtime()
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int n0 = tile_n * NN_TN;
  const int nk = g.K / N3_KC;
  const bool live = n0 + wn * 64 < g.N;     // wave-uniform

  // Wave w stages pieces p = w + 4 i (i = 0, 1) of each operand: 16 activation
  // rows / 2 weight k-rows per piece; the swizzle term (row >> 2) & 3 =
  // (l >> 4) & 3 does not depend on the piece.
  const int a_slot = lane & 3;
  const int a_chunk = a_slot ^ ((lane >> 4) & 3);
  long am0 = m0 + 16 * wave + (lane >> 2), am1 = am0 + 64;
  am0 = am0 < g.M ? am0 : g.M - 1;
  am1 = am1 < g.M ? am1 : g.M - 1;
  // staging addresses: buffer resources (bases in SGPRs), loop-invariant 32-bit
  // lane offsets, one scalar offset per chunk -- no vector address arithmetic
  // inside the K loop (it is matrix-pipe time on gfx950)
  const long lda = g.a_planes ? 32 : g.lda;
  const int va0 = (int)((am0 * lda + a_chunk * 4) * 4);
  const int va1 = (int)((am1 * lda + a_chunk * 4) * 4);
  int wcol = n0 + (lane & 31) * 4;
  wcol = wcol < g.N ? wcol : 0;
  const int vw0 = (int)(((long)(2 * wave + (lane >> 5)) * g.ldw + wcol) * 4);
  const int vw1 = vw0 + (int)(8L * g.ldw * 4);
  auto stage = [&](int kc, float* base) {
    // the chunk's first element as a 64-bit SCALAR base (plane mode: chunk kc
    // is half (kc & 1) of plane kc >> 1): operands larger than 2 GB -- many or
    // long planes, wide models -- stay on this kernel; only a lane's own
    // offset (its row inside the operand) has to fit 32 bits
    const __amdgpu_buffer_rsrc_t rsA = ep_rsrc(
        g.A + (g.a_planes ? (long)(kc >> 1) * g.a_plane_stride + (kc & 1) * 16 : (long)kc * N3_KC));
    const __amdgpu_buffer_rsrc_t rsW = ep_rsrc(g.W + (long)kc * N3_KC * g.ldw);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + wave * 256), 16, va0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + (wave + 4) * 256), 16, va1, 0, 0, 0);
    float* wb = base + NN_TM * N3_KC;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(wb + wave * 256), 16, vw0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(wb + (wave + 4) * 256), 16, vw1, 0, 0, 0);
  };

  f32x16 acc[2][2];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = frag_zero();

  stage(0, smem);
  // a wave without columns only stages and synchronises (a loop of its own:
  // an early `continue` in the loop below makes the compiler copy the 64
  // accumulator registers in every chunk)
  if (!live) {
    for (int kc = 0; kc < nk; ++kc) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (kc + 1 < nk) stage(kc + 1, smem + ((kc & 1) ^ 1) * N3_STAGE);
    }
  }
  // (two copies of the loop body with compile-time LDS stages -- every LDS
  // address a lane constant plus an immediate -- measured 2 % SLOWER over the
  // six shapes of a step than this one body with the stage picked at run time)
  for (int kc = 0; live && kc < nk; ++kc) {
    const int ST = kc & 1;
    // chunk kc landed (this wave's pieces), then everybody's; the barrier
    // also retires every wave's reads of the stage that is refilled next
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (STAMPS && kc == 0) { NSTAMP(1); }
    if (kc + 1 < nk) stage(kc + 1, smem + (ST ^ 1) * N3_STAGE);
    const float* As = smem + ST * N3_STAGE;
    const float* Bs = As + NN_TM * N3_KC;
    f32x4 fa[2][2];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
      const int row = wm * 64 + fm * 32 + j;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        fa[fm][q] = *reinterpret_cast<const f32x4*>(
            As + row * N3_KC + (((2 * q + h) ^ ((row >> 2) & 3)) << 2));
    }
    const float* bl = Bs + 4 * h * NN_TN + wn * 64 + j;
    // raised priority while this wave feeds the matrix pipe: the other
    // workgroups' waves on the SIMD do their staging / barrier work in the gaps
    __builtin_amdgcn_s_setprio(1);
    // the weight operands of step r + PF are requested before the MFMAs of
    // step r (pinned with scheduling barriers: left alone the compiler issues
    // every read right before its use and waits for it -- an LDS round trip
    // per four MFMAs that only other waves on the SIMD can cover; six shapes
    // of a step 4641 -> 4608 us, at B = 1 672 -> 655 us; same MFMA order)
    constexpr int PF = NN3_PF;
    float wv[8][2];
#pragma unroll
    for (int r = 0; r < PF; ++r) {
      const int ko = (8 * (r >> 2) + (r & 3)) * NN_TN;
      wv[r][0] = bl[ko];
      wv[r][1] = bl[ko + 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r + PF < 8) {
        const int ko = (8 * ((r + PF) >> 2) + ((r + PF) & 3)) * NN_TN;
        wv[r + PF][0] = bl[ko];
        wv[r + PF][1] = bl[ko + 32];
      }
      __builtin_amdgcn_sched_barrier(0);
      const float w0 = wv[r][0], w1 = wv[r][1];
      const float x0 = fa[0][r >> 2][r & 3], x1 = fa[1][r >> 2][r & 3];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  // every wave is done with the operand stages before they become the
  // epilogue's staging tiles
  NSTAMP(2);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  NSTAMP(3);
  gemm_epilogue(g, acc, smem, m0, n0, wm, wn, wave, lane, m_end);
  NSTAMP(4);
#undef NSTAMP
}

constexpr int NN3_LDSF = 2 * N3_STAGE > 4 * 32 * EP_LD ? 2 * N3_STAGE : 4 * 32 * EP_LD;

__global__ __launch_bounds__(256, 4) void gemm_nn3_kernel(GemmNN g) {
  // two LDS stages (3 with a counted vmcnt: measured equal or slower)
  __shared__ __attribute__((aligned(1024))) float smem[NN3_LDSF];
  unsigned long long* dbg = nullptr;
#ifdef NN3_STAMPS   // diagnostic build: g.Cpre carries a stamp buffer [nwg][8]
  dbg = reinterpret_cast<unsigned long long*>(g.Cpre) + (size_t)blockIdx.x * 8;
  g.Cpre = nullptr;
  if (threadIdx.x == 0) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[6] = __builtin_amdgcn_s_memrealtime(); }
#endif
  const int logical = xcd_remap(blockIdx.x, g.nwg);
  const int tile_n = logical % g.tiles_n;
  const long m0 = (long)(logical / g.tiles_n) * g.tile_rows;
  const long m_end = m0 + g.tile_rows < g.M ? m0 + g.tile_rows : g.M;
#ifdef NN3_STAMPS
  nn3_tile<true>(g, smem, tile_n, m0, m_end, dbg);
  if (threadIdx.x == 0) dbg[7] = __builtin_amdgcn_s_memrealtime();
#else
  nn3_tile<false>(g, smem, tile_n, m0, m_end, dbg);
#endif
}


// ---------------------------------------------------------------------------
// Split-bf16 NN GEMM (wn_gemm_nn_split): gemm_nn3_kernel's staging structure.
// The weights are split ONCE per call by split_w_kernel into MFMA-fragment
// order and DMA-staged as they are, so a weight fragment is three ds_read_b128
// and no VALU; activations are staged as fp32 and split after the LDS read (44
// VALU per fragment, one fragment per 48 MFMAs at nprod = 6).
// ---------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16, K chunks of 32: stage = 16 KB activations (rows of
// 128 B, 16-byte slots XOR-swizzled with (row >> 1) & 7) + 24 KB weight pieces in
// fragment order [k/32][n/16][piece][lane][8 bf16]; two stages, two workgroups
// per CU; a wave owns 32 rows x ALL 128 columns = 2 x 8 accumulator tiles.
// Round 5 (profiles/r05_split_nn.txt): the kernel runs at the clock the chip
// holds under bf16 MFMA load (1.7 - 2.0 GHz in the K loop, tools/nsp_stamps.py),
// and its time did not move with the occupancy (1 - 3 workgroups per CU), the
// stage count (2 - 4), the activation split's VALU (removed: -4 %) or the weight
// fragments' LDS reads (removed: 0 %); what moved it was the MFMA shape (the
// chip holds a higher clock on 16x16x32 than on 32x32x16: -6 % time) and the
// wave tile (64 x 64 -> 32 x 128: one activation split per weight fragment set,
// -2 %).
#define S16_KC 32
#define S16_STAGE (NN_TM * S16_KC + 8 * 3 * 256)   // floats per stage

__global__ void split_w_kernel(const float* __restrict__ W, int ldw, int K, int N,
                                 unsigned int* __restrict__ out, int nt16) {
  const int kc = blockIdx.y, nt = blockIdx.x, lane = threadIdx.x;
  const int n = nt * 16 + (lane & 15);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kc * S16_KC + 8 * (lane >> 4) + e;
    v[e] = (k < K && n < N) ? W[(long)k * ldw + n] : 0.f;
  }
  const Split3 s = split3(v);
  u32x4* dst = reinterpret_cast<u32x4*>(out) + ((long)(kc * nt16 + nt) * 3) * 64 + lane;
#pragma unroll
  for (int p = 0; p < 3; ++p) dst[p * 64] = __builtin_bit_cast(u32x4, s.p[p]);
}

template <int NPROD>
__device__ __forceinline__ void mma_split16(f32x4& acc, const Split3& a, const Split3& b) {
#define WN_MM(i, k) \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[i], b.p[k], acc, 0, 0, 0)
  if (NPROD >= 9) { WN_MM(2, 2); WN_MM(1, 2); WN_MM(2, 1); }
  if (NPROD >= 6) { WN_MM(1, 1); WN_MM(0, 2); WN_MM(2, 0); }
  WN_MM(0, 1);
  WN_MM(1, 0);
  WN_MM(0, 0);
#undef WN_MM
}

template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_nn_split_kernel(GemmNN g, const unsigned int* wp,
                                                                 int nt16) {
  constexpr int LDSF = 2 * S16_STAGE;
  static_assert(LDSF >= 4 * 32 * EP_LD, "epilogue LDS");
  __shared__ __attribute__((aligned(1024))) float smem[LDSF];
#ifdef NSP_STAMPS   // diagnostic build: g.Cpre carries a stamp buffer [nwg][8]
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(g.Cpre) + (size_t)blockIdx.x * 8;
  g.Cpre = nullptr;
  if (threadIdx.x == 0) { dbg[4] = __builtin_amdgcn_s_memtime(); dbg[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, g.nwg);
  const int tile_n = logical % g.tiles_n;
  const long m0 = (long)(logical / g.tiles_n) * NN_TM;
  const int n0 = tile_n * NN_TN;
  const int nk = (g.K + S16_KC - 1) / S16_KC;

  // activations: wave w stages row blocks b = w + 4 t (8 rows x 128 B = one
  // 1 KB LDS-DMA instruction); lane i -> row 8 b + (i >> 3), LDS slot i & 7,
  // which holds the row's 16-byte slot (i & 7) ^ swz, swz = (row >> 1) & 7 =
  // (4 (w & 1) + (i >> 4)) & 7 for every t
  const int gslot = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  // K % 32 == 16: the last chunk's slots 4 .. 7 lie past the row; they re-read
  // slot 0 of the chunk (their weight pieces are zero)
  const int gslot_tail = (g.K % S16_KC) && gslot >= 4 ? 0 : gslot;
  const long lda = g.a_planes ? 32 : g.lda;
  const float* arow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    long am = m0 + 8 * (wave + 4 * t) + (lane >> 3);
    am = am < g.M ? am : g.M - 1;
    arow[t] = g.A + am * lda;
  }
  // weight pieces: wave w stages blocks p = w + 4 i (i < 6) of the 24 (8
  // n-tiles x 3 pieces) of a chunk; n-tiles past the matrix are clamped
  const int nt0 = n0 >> 4;
  const unsigned int* wsrc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = wave + 4 * i;
    int nt = nt0 + p / 3;
    nt = nt < nt16 ? nt : nt16 - 1;
    wsrc[i] = wp + ((long)nt * 3 + p % 3) * 256 + lane * 4;
  }
  const long w_chunk = (long)nt16 * 3 * 256;   // uint32 per k chunk
  auto stage = [&](int kc, int st) {
    float* base = smem + st * S16_STAGE;
    const long aoff = g.a_planes ? (long)kc * g.a_plane_stride : (long)kc * S16_KC;
    const int gs = (kc == nk - 1 ? gslot_tail : gslot) * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((gptr_t)(arow[t] + aoff + gs),
                                       (lptr_t)(base + (wave + 4 * t) * 256), 16, 0, 0);
    float* wb = base + NN_TM * S16_KC;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kc * w_chunk),
                                       (lptr_t)(wb + (wave + 4 * i) * 256), 16, 0, 0);
  };

  f32x4 acc[2][8];  // [mt][nt]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  int st = 0;
#ifdef NSP_STAMPS
  if (threadIdx.x == 0) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
  for (int kc = 0; kc < nk; ++kc) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kc + 1 < nk) stage(kc + 1, st ^ 1);
    const float* As = smem + st * S16_STAGE;
    const float* Ws = As + NN_TM * S16_KC;
    Split3 xs[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wave * 32 + mt * 16 + r16;
      float v[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(
            As + row * S16_KC + (((2 * kq + u) ^ ((row >> 1) & 7)) << 2));
        v[4 * u] = t[0]; v[4 * u + 1] = t[1]; v[4 * u + 2] = t[2]; v[4 * u + 3] = t[3];
      }
      xs[mt] = split3(v);
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      Split3 ws;
      const u32x4* q = reinterpret_cast<const u32x4*>(Ws + (nt * 3) * 256) + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p) ws.p[p] = __builtin_bit_cast(bf16x8, q[p * 64]);
      mma_split16<NPROD>(acc[0][nt], ws, xs[0]);
      mma_split16<NPROD>(acc[1][nt], ws, xs[1]);
    }
    st ^= 1;
  }
#ifdef NSP_STAMPS
  if (threadIdx.x == 0) { dbg[2] = __builtin_amdgcn_s_memtime(); dbg[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  gemm_epilogue_wide(g, acc, smem, m0, n0, wave, lane);
#ifdef NSP_STAMPS
  if (threadIdx.x == 0) { dbg[6] = __builtin_amdgcn_s_memtime(); dbg[7] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// ---------------------------------------------------------------------------
// TN (weight gradients), LDS-free: operands are 4-byte loads of full lines.
// ---------------------------------------------------------------------------
struct GemmTN {
  const float* A;       // dense [rows][lda] / planes [a_planes][rows][32]
  long lda;
  long a_plane_stride;
  int a_planes;
  const int32_t* codes; // one-hot mode: A[row][m] = (codes[row-shift] == m)
  int shift;            // rows (within a clip of length T) the codes lag
  int T;
  const float* G;       // [rows][ldg]
  long ldg;
  float* slabs;         // [splits][Mw*Nw + Nw]
  long slab_stride;
  long rows;
  long rows_per_split;  // even
  int Mw, Nw;
  int tiles_m, tiles_n; // wave tiles
  int want_colsum;
};

// Software-pipelined: a group = TN_U steps (2*TN_U rows).  The operands of
// group g+1 are loaded into a second named register set while the MFMAs of
// group g issue, so the b32 loads' HBM/L2 latency hides under ~TN_U*MF*NF
// MFMAs (64 cycles each) instead of serialising with them.
// steps per pipeline group: 4, or 3 for the 160-accumulator <5,2> tile
#define TN_UF(MF, NF) (((MF) * (NF) >= 10) ? 3 : 4)

template <int MF, int NF>
struct TnRegs {
  float a[TN_UF(MF, NF)][MF];
  float b[TN_UF(MF, NF)][NF];
};

// Guarded (edge) loader: any shape, 64-bit addressing, per-element predicates.
template <int MF, int NF>
__device__ __forceinline__ void tn_load(const GemmTN& g, TnRegs<MF, NF>& R,
                                        long r0, long r_end, int h, int i,
                                        int m0, int n0, const bool* mv,
                                        const bool* nv, int& tcl) {
#pragma unroll
  for (int u = 0; u < TN_UF(MF, NF); ++u) {
    const long row = r0 + 2 * u + h;
    const bool rv = row < r_end;
    if (g.codes) {
      int code = -1;
      if (rv && tcl >= g.shift) code = g.codes[row - g.shift];
      tcl += 2;
      while (tcl >= g.T) tcl -= g.T;
#pragma unroll
      for (int a = 0; a < MF; ++a)
        R.a[u][a] = (code == m0 + a * 32 + i) ? 1.f : 0.f;
    } else {
#pragma unroll
      for (int a = 0; a < MF; ++a) {
        const int m = m0 + a * 32 + i;
        float v = 0.f;
        if (rv && mv[a])
          v = g.a_planes ? g.A[(long)(m >> 5) * g.a_plane_stride + row * 32 + (m & 31)]
                         : g.A[row * g.lda + m];
        R.a[u][a] = v;
      }
    }
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      float v = 0.f;
      if (rv && nv[b]) v = g.G[row * g.ldg + n0 + b * 32 + i];
      R.b[u][b] = v;
    }
  }
}

// Fast loader for interior groups of full tiles: no predicates, no branches,
// wave-uniform base pointers + 32-bit per-lane offsets (the common case: every
// group but the last of a split, channel counts multiples of 32).
template <int MF, int NF>
__device__ __forceinline__ void tn_load_fast(TnRegs<MF, NF>& R,
                                             const float* const* abase,
                                             const float* const* gbase,
                                             unsigned aoff, unsigned goff,
                                             unsigned astep, unsigned gstep) {
#pragma unroll
  for (int u = 0; u < TN_UF(MF, NF); ++u) {
#pragma unroll
    for (int a = 0; a < MF; ++a) R.a[u][a] = abase[a][aoff + u * astep];
#pragma unroll
    for (int b = 0; b < NF; ++b) R.b[u][b] = gbase[b][goff + u * gstep];
  }
}

template <int MF, int NF>
__device__ __forceinline__ void tn_mma(const TnRegs<MF, NF>& R,
                                       f32x16 (&acc)[MF][NF], float (&cs)[NF]) {
#pragma unroll
  for (int u = 0; u < TN_UF(MF, NF); ++u) {
#pragma unroll
    for (int b = 0; b < NF; ++b) cs[b] += R.b[u][b];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
      for (int b = 0; b < NF; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a[u][a], R.b[u][b],
                                                          acc[a][b], 0, 0, 0);
  }
}

template <int MF, int NF>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTN g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wt = blockIdx.x * 4 + wave;
  if (wt >= g.tiles_m * g.tiles_n) return;
  // consecutive waves share the m tile (same A columns -> L1/L2 reuse)
  const int tn = wt % g.tiles_n, tm = wt / g.tiles_n;
  const int m0 = tm * MF * 32, n0 = tn * NF * 32;
  const long r_begin = (long)blockIdx.y * g.rows_per_split;
  long r_end = r_begin + g.rows_per_split;
  if (r_end > g.rows) r_end = g.rows;

  f32x16 acc[MF][NF];
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = frag_zero();
  float cs[NF];
#pragma unroll
  for (int b = 0; b < NF; ++b) cs[b] = 0.f;

  bool mv[MF], nv[NF];
#pragma unroll
  for (int a = 0; a < MF; ++a) mv[a] = (m0 + a * 32 + i) < g.Mw;
#pragma unroll
  for (int b = 0; b < NF; ++b) nv[b] = (n0 + b * 32 + i) < g.Nw;

  int tcl = g.codes ? (int)((r_begin + h) % g.T) : 0;  // time inside the clip
  constexpr int GR = 2 * TN_UF(MF, NF);                // rows per group
  TnRegs<MF, NF> RA, RB;
  // fast path: dense/plane A (not one-hot), full channel tiles, offsets < 2^32
  const long a_ld = g.a_planes ? 32 : g.lda;
  const bool fast = !g.codes && (m0 + MF * 32 <= g.Mw) && (n0 + NF * 32 <= g.Nw) &&
                    (g.rows * (a_ld > g.ldg ? a_ld : g.ldg) < (1L << 30));
  const float* abase[MF];
  const float* gbase[NF];
#pragma unroll
  for (int a = 0; a < MF; ++a)
    abase[a] = g.a_planes ? g.A + (long)((m0 >> 5) + a) * g.a_plane_stride
                          : g.A + m0 + a * 32;
#pragma unroll
  for (int b = 0; b < NF; ++b) gbase[b] = g.G + n0 + b * 32;
  const unsigned astep = 2u * (unsigned)a_ld, gstep = 2u * (unsigned)g.ldg;
  long r = r_begin;
#define TN_LOAD(REG, R0)                                                       \
  do {                                                                         \
    if (fast && (R0) + GR <= r_end)                                            \
      tn_load_fast<MF, NF>(REG, abase, gbase,                               \
                           (unsigned)(((R0) + h) * a_ld) + i,                  \
                           (unsigned)(((R0) + h) * g.ldg) + i, astep, gstep);  \
    else                                                                       \
      tn_load<MF, NF>(g, REG, (R0), r_end, h, i, m0, n0, mv, nv, tcl);         \
  } while (0)
  if (r < r_end) TN_LOAD(RA, r);
  while (r < r_end) {
    if (r + GR < r_end) TN_LOAD(RB, r + GR);
    tn_mma<MF, NF>(RA, acc, cs);
    r += GR;
    if (r >= r_end) break;
    if (r + GR < r_end) TN_LOAD(RA, r + GR);
    tn_mma<MF, NF>(RB, acc, cs);
    r += GR;
  }
#undef TN_LOAD

  float* slab = g.slabs + (long)blockIdx.y * g.slab_stride;
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      const int n = n0 + b * 32 + i;
      if (n >= g.Nw) continue;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = m0 + a * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
        if (m < g.Mw) slab[(long)m * g.Nw + n] = acc[a][b][rr];
      }
    }
  if (g.want_colsum && tm == 0) {
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      float v = cs[b] + __shfl_xor(cs[b], 32);
      const int n = n0 + b * 32 + i;
      if (h == 0 && n < g.Nw) slab[(long)g.Mw * g.Nw + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// TN, LDS-staged variant for wide outputs (dWs, dW1, dW2): one workgroup
// owns a (MF*32) x (4*NF*32) output tile; the 4 waves share the A chunk
// through LDS and each owns NF*32 columns of the G chunk.  A 32-row chunk is
// 16*MF*NF MFMAs per wave (10 240 cycles at MF=5, NF=2) between barriers, the
// next chunk's global loads (coalesced 16-byte) are in flight meanwhile.
// ---------------------------------------------------------------------------
template <int MF, int NF>
__global__ __launch_bounds__(256, 2) void gemm_tn2_kernel(GemmTN g) {
  constexpr int TM = MF * 32, TNW = 4 * NF * 32;
  constexpr int A4 = TM / 4;            // float4 per A row
  constexpr int G4 = TNW / 4;           // float4 per G row
  constexpr int NA = (32 * A4 + 255) / 256, NG = (32 * G4) / 256;
  __shared__ __attribute__((aligned(16))) float As[32 * TM];
  __shared__ __attribute__((aligned(16))) float Gs[32 * TNW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int tn = blockIdx.x % g.tiles_n, tm = blockIdx.x / g.tiles_n;
  const int m0 = tm * TM, n0 = tn * TNW;
  const long r_begin = (long)blockIdx.y * g.rows_per_split;
  long r_end = r_begin + g.rows_per_split;
  if (r_end > g.rows) r_end = g.rows;

  f32x16 acc[MF][NF];
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = frag_zero();
  float cs[NF];
#pragma unroll
  for (int b = 0; b < NF; ++b) cs[b] = 0.f;

  f32x4 ra[NA], rg[NG];
  auto gload = [&](long r0) {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int idx = tid + 256 * k;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < 32 * A4) {
        const int row = idx / A4, c4 = idx - row * A4;
        const long r = r0 + row;
        if (r < r_end) {
          const int m = m0 + c4 * 4;
          const float* p = g.a_planes
                               ? g.A + (long)(m >> 5) * g.a_plane_stride + r * 32 + (m & 31)
                               : g.A + r * g.lda + m;
          v = *reinterpret_cast<const f32x4*>(p);
        }
      }
      ra[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int idx = tid + 256 * k;
      const int row = idx / G4, c4 = idx - row * G4;
      const long r = r0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < r_end)
        v = *reinterpret_cast<const f32x4*>(g.G + r * g.ldg + n0 + c4 * 4);
      rg[k] = v;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int idx = tid + 256 * k;
      if (idx < 32 * A4) *reinterpret_cast<f32x4*>(&As[idx * 4]) = ra[k];
    }
#pragma unroll
    for (int k = 0; k < NG; ++k)
      *reinterpret_cast<f32x4*>(&Gs[(tid + 256 * k) * 4]) = rg[k];
  };

  gload(r_begin);
  sstore();
  __syncthreads();
  for (long r0 = r_begin; r0 < r_end; r0 += 32) {
    const bool more = r0 + 32 < r_end;
    if (more) gload(r0 + 32);
    const float* al = As + h * TM + i;
    const float* gl = Gs + h * TNW + wave * (NF * 32) + i;
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {
      float av[MF], bv[NF];
#pragma unroll
      for (int a = 0; a < MF; ++a) av[a] = al[2 * s * TM + a * 32];
#pragma unroll
      for (int b = 0; b < NF; ++b) {
        bv[b] = gl[2 * s * TNW + b * 32];
        cs[b] += bv[b];
      }
#pragma unroll
      for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < NF; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    if (more) sstore();
    __syncthreads();
  }

  float* slab = g.slabs + (long)blockIdx.y * g.slab_stride;
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      const int n = n0 + wave * (NF * 32) + b * 32 + i;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = m0 + a * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
        slab[(long)m * g.Nw + n] = acc[a][b][rr];
      }
    }
  if (g.want_colsum && (tm == 0 || g.want_colsum == 2)) {
    // (spread tail layout: this kernel sums in tile row 0 only; the other tile
    // rows' partial rows are zero)
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      float v = cs[b] + __shfl_xor(cs[b], 32);
      const int n = n0 + wave * (NF * 32) + b * 32 + i;
      if (h == 0) slab[(long)g.Mw * g.Nw + (long)tm * g.Nw + n] = tm == 0 ? v : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------
// TN, LDS-DMA variant of gemm_tn2_kernel (the default for dWs / dW1 / dW2):
// same output tile per workgroup, but 16-row chunks staged by
// global_load_lds_dwordx4 into two stages (no staging registers, one barrier
// per chunk), 36 / 48 KB of LDS -> 4 / 3 workgroups per CU, s_setprio(1)
// around the MFMA block.  Both operands are k-major in memory ([row][m],
// [row][n]) which is what the MFMA A/B operand reads want, so the LDS image is
// the plain [row][TM] / [row][TNW] array and a DMA piece is 1 KiB of it.
// Needs whole 16-row chunks (rows and rows_per_split multiples of 16).
// ---------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct wn_true { static constexpr bool value = true; };
struct wn_false { static constexpr bool value = false; };
template <int MF, int NF>
__global__ __launch_bounds__(256, (MF * NF >= 10 ? 2 : MF * NF > 5 ? 3 : 4)) void gemm_tn3_kernel(GemmTN g) {
  constexpr int TM = MF * 32, TNW = 4 * NF * 32, KR = 16;
  constexpr int STAGE = KR * TM + KR * TNW;     // floats
  constexpr int PA = KR * TM / 256, PG = KR * TNW / 256;  // 1 KiB pieces
  __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  // XCD-aware order: the tiles of one split (same rows of A and G) are given
  // to blocks of the SAME XCD (contiguous logical ids per XCD), so each operand
  // chunk is fetched from HBM once per XCD instead of once per tile
  // (measured 4.1 GB -> see DESIGN.md per dWs launch before the remap).
  const int ntile = gridDim.x;
  const int logical = xcd_remap(blockIdx.x + ntile * blockIdx.y, ntile * gridDim.y);
  const int tile = logical % ntile, split = logical / ntile;
  const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
  const int m0 = tm * TM, n0 = tn * TNW;
  const long r_begin = (long)split * g.rows_per_split;
  long r_end = r_begin + g.rows_per_split;
  if (r_end > g.rows) r_end = g.rows;
  // (the last split's last chunk may be ragged -- B * T is whatever the reader
  // cut: its missing rows are staged as zeros)
  const int nchunks = r_end > r_begin ? (int)((r_end - r_begin + KR - 1) / KR) : 0;
  const int tail_rows = nchunks ? (int)(r_end - r_begin) - (nchunks - 1) * KR : KR;

  // Staging addresses: a buffer resource per operand (base in SGPRs), per piece
  // ONE loop-invariant 32-bit lane offset, per chunk one scalar row offset.
  // (Round 2 rebuilt every piece's 64-bit address in vector registers in every
  // chunk: 184 vector instructions per 40 MFMAs -- matrix-pipe time on gfx950.)
  constexpr int NPA = (PA + 3) / 4, NPG = (PG + 3) / 4;   // pieces per wave
  // (bases: the split's first row of the tile's first plane / first column --
  // 64-bit, scalar; what is left for the 32-bit offsets is the tile's TM / 32
  // planes and the split's rows, so operands larger than 2 GB stay here)
  const long rowA = g.a_planes ? 32 * 4 : g.lda * 4, rowG = g.ldg * 4;   // bytes per row
  const __amdgpu_buffer_rsrc_t rsA = ep_rsrc(
      g.a_planes ? g.A + (long)(m0 >> 5) * g.a_plane_stride + r_begin * 32
                 : g.A + r_begin * g.lda + m0);
  const __amdgpu_buffer_rsrc_t rsG = ep_rsrc(g.G + r_begin * g.ldg + n0);
  int voA[3], voG[4];
  static_assert(NPA <= 3 && NPG <= 4, "pieces per wave");
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    const int e = (wave + 4 * k) * 256 + lane * 4;
    const int row = e / TM, col = e - row * TM;
    voA[k] = g.a_planes ? (int)(((long)(col >> 5) * g.a_plane_stride + row * 32 + (col & 31)) * 4)
                        : (int)(((long)row * g.lda + col) * 4);
  }
#pragma unroll
  for (int k = 0; k < NPG; ++k) {
    const int e = (wave + 4 * k) * 256 + lane * 4;
    const int row = e / TNW, col = e - row * TNW;
    voG[k] = (int)(((long)row * g.ldg + col) * 4);
  }
  auto stage = [&](int c, int st) {
    float* base = smem + st * STAGE;
    const long r0 = (long)c * KR;
    const unsigned sA = (unsigned)(r0 * rowA), sG = (unsigned)(r0 * rowG);
    float* gb = base + KR * TM;
    if (tail_rows < KR && c == nchunks - 1) {
      // ragged last chunk: this wave's pieces are zeroed, then only the lanes
      // whose row exists load (a wave's DS write and its later LDS-DMA into
      // the same bytes are ordered by the wait)
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NPA; ++k)
        if (wave + 4 * k < PA)
          *reinterpret_cast<f32x4*>(base + (wave + 4 * k) * 256 + lane * 4) = zero;
#pragma unroll
      for (int k = 0; k < NPG; ++k)
        if (wave + 4 * k < PG)
          *reinterpret_cast<f32x4*>(gb + (wave + 4 * k) * 256 + lane * 4) = zero;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NPA; ++k) {
        const int p = wave + 4 * k;
        if (p < PA && (p * 256 + lane * 4) / TM < tail_rows)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + p * 256), 16, voA[k], sA, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NPG; ++k) {
        const int p = wave + 4 * k;
        if (p < PG && (p * 256 + lane * 4) / TNW < tail_rows)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lptr_t)(gb + p * 256), 16, voG[k], sG, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
      const int p = wave + 4 * k;
      if (p < PA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + p * 256), 16, voA[k], sA, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NPG; ++k) {
      const int p = wave + 4 * k;
      if (p < PG)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lptr_t)(gb + p * 256), 16, voG[k], sG, 0, 0);
    }
  };

  f32x16 acc[MF][NF];
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = frag_zero();
  float cs[NF];
  f32x2 cs2[NF];   // the pinned loop's sums: even / odd steps
#pragma unroll
  for (int b = 0; b < NF; ++b) {
    cs[b] = 0.f;
    cs2[b] = f32x2{0.f, 0.f};
  }

  if (nchunks > 0) stage(0, 0);
  // Operands of step s + 1 are requested before the MFMAs of step s (as in
  // gemm_nn3_kernel).  Column sums (the bias gradients) are wanted from the
  // first row of tiles only: the K loop exists twice, with and without their
  // adds, behind ONE workgroup-uniform branch.  (As a run-time flag inside the
  // loop the adds stayed in every workgroup's loop as add + select, and on
  // gfx950 they are matrix-pipe time; a branch per chunk sends the
  // accumulators through memory.)
  auto kloop = [&](auto cs_tag, int c_begin, int c_end) {
    constexpr bool SUM = decltype(cs_tag)::value;
    for (int c = c_begin; c < c_end; ++c) {
      const int st = c & 1;
      // chunk c landed for every wave; all reads of the other stage retired
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (c + 1 < nchunks) stage(c + 1, st ^ 1);
      const float* al = smem + st * STAGE + h * TM + i;
      const float* gl = smem + st * STAGE + KR * TM + h * TNW + wave * (NF * 32) + i;
      __builtin_amdgcn_s_setprio(1);
      if (SUM && MF * NF == 8) {
        // (<4, 2> with the sums: the pinned order below spills 34 registers;
        // the sums two steps at a time behind their MFMAs measured slower)
#pragma unroll
        for (int s2 = 0; s2 < KR / 2; ++s2) {
          float a_[MF], b_[NF];
#pragma unroll
          for (int a = 0; a < MF; ++a) a_[a] = al[2 * s2 * TM + a * 32];
#pragma unroll
          for (int b = 0; b < NF; ++b) {
            b_[b] = gl[2 * s2 * TNW + b * 32];
            cs[b] += b_[b];
          }
#pragma unroll
          for (int a = 0; a < MF; ++a)
#pragma unroll
            for (int b = 0; b < NF; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[a], b_[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        continue;
      }
      float av[2][MF], bv[2][NF], keep[KR / 2][NF];
      static_assert(KR == 16, "column-sum tree");
      auto fetch = [&](int s2, float (&a_)[MF], float (&b_)[NF]) {
#pragma unroll
        for (int a = 0; a < MF; ++a) a_[a] = al[2 * s2 * TM + a * 32];
#pragma unroll
        for (int b = 0; b < NF; ++b) b_[b] = gl[2 * s2 * TNW + b * 32];
      };
      fetch(0, av[0], bv[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < KR / 2; ++s2) {
        if (s2 + 1 < KR / 2) fetch(s2 + 1, av[(s2 + 1) & 1], bv[(s2 + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < MF; ++a)
#pragma unroll
          for (int b = 0; b < NF; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2 & 1][a], bv[s2 & 1][b],
                                                             acc[a][b], 0, 0, 0);
        if (SUM) {
#pragma unroll
          for (int b = 0; b < NF; ++b) keep[s2][b] = bv[s2 & 1][b];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (SUM) {
        // the chunk's column sums in one group behind its MFMAs (an add
        // between two MFMAs costs the wave ~12 cycles of matrix-pipe time)
#pragma unroll
        for (int b = 0; b < NF; ++b) {
          const f32x2 p0 = {keep[0][b], keep[1][b]}, p1 = {keep[2][b], keep[3][b]};
          const f32x2 p2 = {keep[4][b], keep[5][b]}, p3 = {keep[6][b], keep[7][b]};
          cs2[b] += (p0 + p1) + (p2 + p3);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
  };
  // (which row of tiles sums the columns rotates with the split and the column
  // tile: with tm == 0 the workgroups that carry the extra adds have
  // consecutive ids, share CUs and set the time of a one-round launch)
  // want_colsum == 2 ("spread", the host's default): EVERY tile row sums the
  // columns of its share of the split's chunks -- chunk c when c % tiles_m ==
  // its residue -- into its own partial row of the slab's tail (tiles_m rows;
  // the slab reduction adds them up).  With ONE owner tile row per (split,
  // column tile) the four owners of a split were the slowest workgroups of a
  // one-round launch (2258 vs 2220 us per step without sums) and, falling
  // behind the other 36, re-fetched their chunks from beyond the L2 (2.5 GB of
  // traffic per dWs launch against 1.08 GB of operands).  The K loop stays
  // free of per-chunk branches: per group of tiles_m chunks three plain
  // loops (before / the summed chunk / after).
  const bool spread = g.want_colsum == 2;
  const bool do_cs = spread ? true : (g.want_colsum && tm == (split + 3 * tn) % g.tiles_m);
  if (spread) {
    const int own = (tm + split + 3 * tn) % g.tiles_m;
    for (int c0 = 0; c0 < nchunks; c0 += g.tiles_m) {
      const int ce = min(nchunks, c0 + g.tiles_m);
      const int cs_at = min(ce, c0 + own);
      kloop(wn_false{}, c0, cs_at);
      kloop(wn_true{}, cs_at, min(ce, cs_at + 1));
      kloop(wn_false{}, min(ce, cs_at + 1), ce);
    }
  } else if (do_cs) {
    kloop(wn_true{}, 0, nchunks);
  } else {
    kloop(wn_false{}, 0, nchunks);
  }

  float* slab = g.slabs + (long)split * g.slab_stride;
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      const int n = n0 + wave * (NF * 32) + b * 32 + i;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = m0 + a * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
        // (written once, read by the slab reduction: nt hint, the three TN
        // launches 2264 -> 2242 us)
        __builtin_nontemporal_store(acc[a][b][rr], slab + (long)m * g.Nw + n);
      }
    }
  if (do_cs) {
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      cs[b] += cs2[b].x + cs2[b].y;
      float v = cs[b] + __shfl_xor(cs[b], 32);
      const int n = n0 + wave * (NF * 32) + b * 32 + i;
      if (h == 0) slab[(long)g.Mw * g.Nw + (spread ? (long)tm * g.Nw : 0) + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Split-bf16 TN GEMM (wn_gemm_tn_split, opt-in): workgroup tile 128 x 128,
// 2 x 2 waves of 64 x 64, 16-row chunks of both operands staged as fp32 by
// LDS-DMA and split into bf16 pieces after the LDS read (both operands are
// activations, nothing can be pre-split).  Lane (i, h) of an operand fragment
// holds rows 8h .. 8h+7 of column i: eight ds_read_b32 of consecutive floats
// across lanes.
// ---------------------------------------------------------------------------
template <int NPROD>
__global__ __launch_bounds__(256, 3) void gemm_tn_split_kernel(GemmTN g) {
  constexpr int KR = 16, TT = 128;
  constexpr int STAGE = 2 * KR * TT;   // floats: A chunk then G chunk
  __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int ntile = gridDim.x;
  const int logical = xcd_remap(blockIdx.x + ntile * blockIdx.y, ntile * gridDim.y);
  const int tile = logical % ntile, split = logical / ntile;
  const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
  const int m0 = tm * TT, n0 = tn * TT;
  const long r_begin = (long)split * g.rows_per_split;
  long r_end = r_begin + g.rows_per_split;
  if (r_end > g.rows) r_end = g.rows;
  const int nchunks = r_end > r_begin ? (int)((r_end - r_begin) / KR) : 0;

  // wave w stages pieces w and w + 4 (2 rows x 128 floats each) of A and of G;
  // columns past the matrix are clamped (they feed outputs never stored)
  const int prow = lane >> 5, pcol = (lane & 31) * 4;
  int am = m0 + pcol;
  am = am < g.Mw ? am : g.Mw - 4;
  int gn = n0 + pcol;
  gn = gn < g.Nw ? gn : g.Nw - 4;
  auto stage = [&](int c, int st) {
    float* base = smem + st * STAGE;
    const long r0 = r_begin + (long)c * KR;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave + 4 * q;
      const long row = r0 + 2 * p + prow;
      const float* src = g.a_planes
                             ? g.A + (long)(am >> 5) * g.a_plane_stride + row * 32 + (am & 31)
                             : g.A + row * g.lda + am;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + p * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(g.G + row * g.ldg + gn),
                                       (lptr_t)(base + KR * TT + p * 256), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];  // [fm][fn]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = frag_zero();
  float cs[2] = {0.f, 0.f};

  if (nchunks > 0) stage(0, 0);
  for (int c = 0; c < nchunks; ++c) {
    const int st = c & 1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + 1 < nchunks) stage(c + 1, st ^ 1);
    const float* al = smem + st * STAGE + 8 * h * TT + wm * 64 + i;
    const float* gl = smem + st * STAGE + KR * TT + 8 * h * TT + wn * 64 + i;
    Split3 as[2], gs[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = al[e * TT + f * 32];
      as[f] = split3(v);
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = gl[e * TT + f * 32];
        cs[f] += v[e];
      }
      gs[f] = split3(v);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) mma_split<NPROD>(acc[a][b], as[a], gs[b]);
  }

  float* slab = g.slabs + (long)split * g.slab_stride;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = n0 + wn * 64 + b * 32 + i;
      if (n >= g.Nw) continue;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = m0 + wm * 64 + a * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
        if (m < g.Mw) slab[(long)m * g.Nw + n] = acc[a][b][rr];
      }
    }
  if (g.want_colsum && tm == 0 && wm == 0) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float v = cs[b] + __shfl_xor(cs[b], 32);
      const int n = n0 + wn * 64 + b * 32 + i;
      if (h == 0 && n < g.Nw) slab[(long)g.Mw * g.Nw + n] = v;
    }
  }
}

// out[b][rep][e] = sum_s slabs[b][s][offset + e]   (fixed order over s)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs,
                                    int num_slabs, long slab_stride,
                                    long in_batch_stride, long offset, long n,
                                    float* __restrict__ dst,
                                    long out_batch_stride, int replicate,
                                    long rep_stride) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int b = blockIdx.y;
  const float* p = slabs + (long)b * in_batch_stride + offset + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 3 < num_slabs; s += 4) {
    s0 += p[(long)s * slab_stride];
    s1 += p[(long)(s + 1) * slab_stride];
    s2 += p[(long)(s + 2) * slab_stride];
    s3 += p[(long)(s + 3) * slab_stride];
  }
  for (; s < num_slabs; ++s) s0 += p[(long)s * slab_stride];
  const float v = (s0 + s1) + (s2 + s3);
  for (int r = 0; r < replicate; ++r)
    dst[(long)b * out_batch_stride + (long)r * rep_stride + e] = v;
}

// the same with 16-byte loads and eight slabs in flight per thread (every
// stride / offset / count a multiple of 4 floats): a thread's loads are the
// only memory-level parallelism these launches have (at B = 1 they are pure
// load latency, at B = 8 four-byte loads leave HBM bandwidth on the table).
// Fixed order: partial sums over s = k mod 8, then a fixed tree.
template <int PARTS>
__global__ void reduce_slabs4_kernel(const float* __restrict__ slabs,
                                     int num_slabs, long slab_stride,
                                     long in_batch_stride, long offset, long n4,
                                     float* __restrict__ dst,
                                     long out_batch_stride, int replicate,
                                     long rep_stride) {
  // blockDim.x = 64 * PARTS: 64 columns of four floats, the slabs cut into
  // PARTS contiguous ranges (fixed: ceil(num_slabs / PARTS) each) that are
  // summed side by side and combined through LDS in a fixed tree
  __shared__ f32x4 part[PARTS > 1 ? PARTS : 1][64];
  const int col = threadIdx.x & 63, pt = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + col;
  const int b = blockIdx.y;
  const int per = (num_slabs + PARTS - 1) / PARTS;
  const int s0 = pt * per, s1 = min(num_slabs, s0 + per);
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs + (long)b * in_batch_stride + offset) + e;
    const long st = slab_stride / 4;
    int s = s0;
    for (; s + 7 < s1; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(long)(s + k) * st];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (s + k < s1) acc[k] += p[(long)(s + k) * st];
  }
  f32x4 v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (PARTS > 1) {
    part[pt][col] = v;
    __syncthreads();
    if (pt != 0) return;
    if (PARTS == 4) v = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
    else v = part[0][col] + part[1][col];
  }
  if (e >= n4) return;
  for (int r = 0; r < replicate; ++r)
    *reinterpret_cast<f32x4*>(dst + (long)b * out_batch_stride + (long)r * rep_stride + 4 * e) = v;
}

// A weight-gradient GEMM's slabs in ONE launch: elements [0, n_main) of every
// slab (the matrix) to dst_main, the n_tail elements behind them (the column
// sums = bias gradient) to dst_tail, `replicate` copies rep_stride apart (the
// skip convs' bias gradients are the same row for every layer).  Same fixed
// order as reduce_slabs4<4>.
// (RMT_PARTS waves per workgroup, each that share of the slabs -- and of the
// (slab, row) pairs of a spread column-sum tail.  Measured inside the B = 8 step
// under rocprofv3, average of the three launches: 4 waves 32.0 us, 8 32.3,
// 16 37.8 -- the two tail workgroups of a dWs reduction are faster with sixteen,
// the launch as a whole is not.)
#ifndef RMT_PARTS
#define RMT_PARTS 4
#endif
__global__ __launch_bounds__(64 * RMT_PARTS) void reduce_slabs_mt_kernel(
    const float* __restrict__ slabs, int num_slabs, long slab_stride, long n_main4,
    float* __restrict__ dst_main, long n_tail4, float* __restrict__ dst_tail,
    int replicate, long rep_stride, int tail_rows) {
  __shared__ f32x4 part[RMT_PARTS][64];
  const int col = threadIdx.x & 63, pt = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + col;
  const long n4 = n_main4 + n_tail4;
  const int per = (num_slabs + RMT_PARTS - 1) / RMT_PARTS;
  const int s0 = min(num_slabs, pt * per), s1 = min(num_slabs, s0 + per);
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + e;
    const long st = slab_stride / 4;
    int s = s0;
    for (; s + 7 < s1; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(long)(s + k) * st];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (s + k < s1) acc[k] += p[(long)(s + k) * st];
    // the further partial rows of a spread column-sum tail: (slab, row) pairs
    // in batches of eight loads, like the slabs themselves
    if (e >= n_main4 && tail_rows > 1) {
      const int nv = (s1 - s0) * (tail_rows - 1);
      int v = 0;
      for (; v + 7 < nv; v += 8) {
        f32x4 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int q = s0 + (v + k) / (tail_rows - 1), t = 1 + (v + k) % (tail_rows - 1);
          w[k] = p[(long)q * st + (long)t * n_tail4];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += w[k];
      }
      for (; v < nv; ++v) {
        const int q = s0 + v / (tail_rows - 1), t = 1 + v % (tail_rows - 1);
        acc[v & 7] += p[(long)q * st + (long)t * n_tail4];
      }
    }
  }
  part[pt][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (pt != 0 || e >= n4) return;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < RMT_PARTS; q += 4)
    v += (part[q][col] + part[q + 1][col]) + (part[q + 2][col] + part[q + 3][col]);
  if (e < n_main4) {
    reinterpret_cast<f32x4*>(dst_main)[e] = v;
  } else {
    for (int r = 0; r < replicate; ++r)
      *reinterpret_cast<f32x4*>(dst_tail + (long)r * rep_stride + 4 * (e - n_main4)) = v;
  }
}

// Channel-block models (wavenet/blocked.py): the weight-gradient slabs of the
// CB x CB (input block a, output block b) pairs of one layer,
// slabs[pair][slab][(2K+1) * 1024 + 96] (wn_layer_wgrad_k's layout: Wf taps,
// Wg taps, Wd, bf | bg | bd), summed in the fixed order of reduce_slabs4<4> and
// written straight into the layer's [K][C][C] / [C][C] gradient matrices and
// (pairs with a == 0) bias vectors -- one launch per layer instead of one
// reduction plus three to six strided copies per pair.
// (16 waves per workgroup, each a sixteenth of the slabs: the grid is only
// ~21 x CB^2 workgroups, so with four waves a 64-channel layer's 43 MB were
// read at 1.8 TB/s -- 24 us per layer, 1.2 ms per step)
#define RPS_PARTS 16
__global__ __launch_bounds__(64 * RPS_PARTS) void reduce_pair_slabs_kernel(
    const float* __restrict__ slabs, int num_slabs, int CB, int K, int has_dense,
    int use_bias, float* __restrict__ g, int C, long off_bias, int tap0, int Ktot) {
  __shared__ f32x4 part[RPS_PARTS][64];
  const int WF = (2 * K + 1) * 1024, n4 = (WF + 96) / 4;
  const int col = threadIdx.x & 63, pt = threadIdx.x >> 6;
  const int e4 = blockIdx.x * 64 + col;
  const int pair = blockIdx.y, a = pair / CB, b = pair - a * CB;
  const int per = (num_slabs + RPS_PARTS - 1) / RPS_PARTS;
  const int s0 = min(num_slabs, pt * per), s1 = min(num_slabs, s0 + per);
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e4 < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(
                         slabs + (size_t)pair * num_slabs * (WF + 96)) + e4;
    const long st = (WF + 96) / 4;
    int s = s0;
    for (; s + 7 < s1; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(long)(s + k) * st];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (s + k < s1) acc[k] += p[(long)(s + k) * st];
  }
  part[pt][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (pt != 0 || e4 >= n4) return;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < RPS_PARTS; q += 4)
    v += (part[q][col] + part[q + 1][col]) + (part[q + 2][col] + part[q + 3][col]);
  const int e = 4 * e4;
  if (e < WF) {
    const int m = e >> 10, r = (e >> 5) & 31, c = e & 31;   // matrix (Wf taps, Wg taps, Wd)
    if (m == 2 * K && !has_dense) return;
    // (the slab holds taps tap0 .. tap0 + K - 1 of a filter of Ktot taps)
    const int md = m < K ? tap0 + m : m < 2 * K ? Ktot + tap0 + (m - K) : 2 * Ktot;
    *reinterpret_cast<f32x4*>(g + ((size_t)md * C + a * 32 + r) * C + b * 32 + c) = v;
  } else if (a == 0 && use_bias) {
    const int q = (e - WF) >> 5, c = (e - WF) & 31;         // bf, bg, bd
    if (q == 2 && !has_dense) return;
    *reinterpret_cast<f32x4*>(g + off_bias + (size_t)q * C + b * 32 + c) = v;
  }
}

__global__ void transpose_pad_kernel(const float* __restrict__ in, int rows,
                                     int cols, long in_ld,
                                     float* __restrict__ out, long out_ld) {
  // out[c][r] = in[r][c]; 32x32 LDS tiles
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 thr: ty 0..7
  for (int k = ty; k < 32; k += 8) {
    const int r = by + k, c = bx + tx;
    tile[k][tx] = (r < rows && c < cols) ? in[(long)r * in_ld + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = bx + k, r = by + tx;
    if (c < cols && r < rows) out[(long)c * out_ld + r] = tile[tx][k];
  }
}

// Diagnostic: fp32 MFMA loop -- calibrates the MFMA ceiling of the device the
// benchmark runs on.  mode 0: operands in registers (no memory traffic);
// mode 1: the A operand pair of every 4-MFMA group is read from LDS right
// before use (the pattern of the GEMM inner loops).
template <int MODE>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  __shared__ float lds[4096];
  for (int k = threadIdx.x; k < 4096; k += 256) lds[k] = 1.0f + k * 1e-4f;
  __syncthreads();
  f32x16 a0 = frag_zero(), a1 = frag_zero(), a2 = frag_zero(), a3 = frag_zero();
  float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f - threadIdx.x * 1e-3f;
  const float* lp = lds + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 1) {
        x = lp[(u * 128 + it) & 4095 & ~63];
        y = lp[((u * 128 + 64 + it) & 4095) & ~63];
      }
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" {

// Launches `blocks` x 4 waves, each issuing iters*32 MFMAs (4096 FLOP each).
int wn_diag_mfma_peak(float* out, int blocks, int iters, void* stream) {
  if (!out) return WN_ERR_NULL;
  if (blocks <= 0 || iters == 0) return WN_ERR_BAD_SHAPE;
  if (iters > 0)
    hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, out, iters);
  else  // negative iters: LDS-fed variant
    hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, out, -iters);
  return wn_check_launch();
}

// argument checks of one NN problem + its kernel descriptor
static int gemm_nn_describe(GemmNN& g, const float* A, long lda, int a_planes,
               long a_plane_stride, const float* W, int ldw, const float* bias,
               const float* mask, long ld_mask, const float* addend, long ld_add,
               float* C, long ldc, int c_planes, long c_plane_stride, float* Cpre,
               long M, int N, int K, int relu) {
  if (!A || !W || !C) return WN_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0) return WN_ERR_BAD_SHAPE;
  if ((N & 3) || (K & 3) || (ldw & 3)) return WN_ERR_UNSUPPORTED;
  if (a_planes) {
    if (K != a_planes * 32) return WN_ERR_BAD_SHAPE;
  } else if (lda & 3) {
    return WN_ERR_UNSUPPORTED;
  }
  if (c_planes) {
    if (N != c_planes * 32) return WN_ERR_BAD_SHAPE;
  } else if (ldc & 3) {
    return WN_ERR_UNSUPPORTED;
  }
  if ((mask && (ld_mask & 3)) || (addend && (ld_add & 3)))
    return WN_ERR_UNSUPPORTED;
  // (addend with ld_add == 0: in the output's plane layout, needs c_planes;
  // the epilogue addresses a lane's neighbour plane with a 32-bit byte offset)
  if (addend && ld_add == 0 && !c_planes) return WN_ERR_BAD_SHAPE;
  if (c_planes && c_plane_stride * 4 >= (1L << 31)) return WN_ERR_UNSUPPORTED;
  const void* ptrs[] = {A, W, bias, mask, addend, C, Cpre};
  for (const void* p : ptrs)
    if (p && !wn_aligned16(p)) return WN_ERR_MISALIGNED;
  g.A = A; g.lda = lda; g.a_plane_stride = a_plane_stride; g.a_planes = a_planes;
  g.W = W; g.ldw = ldw; g.bias = bias; g.mask = mask; g.ld_mask = ld_mask;
  g.addend = addend; g.ld_add = ld_add; g.C = C; g.ldc = ldc;
  g.c_plane_stride = c_plane_stride; g.c_planes = c_planes; g.Cpre = Cpre;
  g.M = M; g.N = N; g.K = K; g.relu = relu;
  long tiles_m = (M + NN_TM - 1) / NN_TM;
  g.tiles_n = (N + NN_TN - 1) / NN_TN;
  g.tile_rows = NN_TM;
  const long nwg = tiles_m * g.tiles_n;
  if (nwg > 0x7fffffffL) return WN_ERR_BAD_SHAPE;
  g.nwg = (int)nwg;
  g.tiles_m = (int)tiles_m;
  return WN_OK;
}

// (gemm_nn3_kernel stages through 32-bit byte offsets of buffer resources:
// a lane's row offset inside the A operand and inside a 16-row W chunk)
static bool gemm_nn3_ok(const GemmNN& g) {
  const long a_bytes = (g.a_planes ? g.M * 32 : g.M * g.lda) * 4;
  return (g.K % N3_KC) == 0 && a_bytes < (1L << 31) &&
         (long)N3_KC * g.ldw * 4 < (1L << 31);
}

static int gemm_nn_launch(const float* A, long lda, int a_planes, long a_plane_stride,
               const float* W, int ldw, const float* bias, const float* mask,
               long ld_mask, const float* addend, long ld_add, float* C,
               long ldc, int c_planes, long c_plane_stride, float* Cpre,
               long M, int N, int K, int relu, void* stream,
               void* wsplit = nullptr, int nprod = 0) {
  GemmNN g;
  const int rc = gemm_nn_describe(g, A, lda, a_planes, a_plane_stride, W, ldw, bias, mask,
                                  ld_mask, addend, ld_add, C, ldc, c_planes,
                                  c_plane_stride, Cpre, M, N, K, relu);
  if (rc != WN_OK) return rc;
  const long nwg = g.nwg;
  // default: LDS-DMA kernel, 4 workgroups / CU; K % 16 != 0 (or operands
  // beyond 32-bit byte offsets): register-staged two-stage kernel
  if (wsplit && (K % N3_KC) != 0) return WN_ERR_UNSUPPORTED;
  if (wsplit && !wn_aligned16(wsplit)) return WN_ERR_MISALIGNED;
  if (!wsplit && !gemm_nn3_ok(g))
    hipLaunchKernelGGL(gemm_nn_kernel<2>, dim3((unsigned)nwg), dim3(256), 0,
                       (hipStream_t)stream, g);
  else if (wsplit) {
    hipStream_t s = (hipStream_t)stream;
    dim3 gr((unsigned)nwg), bl(256);
    const unsigned int* wp = (const unsigned int*)wsplit;
    const int nt16 = (N + 15) / 16;
    hipLaunchKernelGGL(split_w_kernel, dim3(nt16, (K + S16_KC - 1) / S16_KC), dim3(64), 0, s,
                       W, ldw, K, N, (unsigned int*)wsplit, nt16);
    if (nprod == 3) hipLaunchKernelGGL(gemm_nn_split_kernel<3>, gr, bl, 0, s, g, wp, nt16);
    else if (nprod == 9) hipLaunchKernelGGL(gemm_nn_split_kernel<9>, gr, bl, 0, s, g, wp, nt16);
    else hipLaunchKernelGGL(gemm_nn_split_kernel<6>, gr, bl, 0, s, g, wp, nt16);
  } else
    hipLaunchKernelGGL(gemm_nn3_kernel, dim3((unsigned)nwg), dim3(256), 0,
                       (hipStream_t)stream, g);
  return wn_check_launch();
}

int wn_gemm_nn(const float* A, long lda, int a_planes, long a_plane_stride,
               const float* W, int ldw, const float* bias, const float* mask,
               long ld_mask, const float* addend, long ld_add, float* C,
               long ldc, int c_planes, long c_plane_stride, float* Cpre,
               long M, int N, int K, int relu, void* stream) {
  return gemm_nn_launch(A, lda, a_planes, a_plane_stride, W, ldw, bias, mask,
                        ld_mask, addend, ld_add, C, ldc, c_planes,
                        c_plane_stride, Cpre, M, N, K, relu, stream);
}

// Split-bf16 variant of wn_gemm_nn (opt-in): fp32 accuracy from bf16 matrix
// instructions, see gemm_nn_split_kernel.  `w_scratch` (wn_gemm_split_w_bytes
// bytes, 16-byte aligned) receives the weight pieces; nprod = 3, 6 or 9.
long wn_gemm_split_w_bytes(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  return (long)((K + 31) / 32) * ((N + 15) / 16) * 3 * 1024;
}

int wn_gemm_nn_split(const float* A, long lda, int a_planes, long a_plane_stride,
                     const float* W, int ldw, const float* bias, const float* mask,
                     long ld_mask, const float* addend, long ld_add, float* C,
                     long ldc, int c_planes, long c_plane_stride, float* Cpre,
                     long M, int N, int K, int relu, void* w_scratch, int nprod,
                     void* stream) {
  if (!w_scratch) return WN_ERR_NULL;
  if (nprod != 3 && nprod != 6 && nprod != 9) return WN_ERR_BAD_SHAPE;
  return gemm_nn_launch(A, lda, a_planes, a_plane_stride, W, ldw, bias, mask,
                        ld_mask, addend, ld_add, C, ldc, c_planes,
                        c_plane_stride, Cpre, M, N, K, relu, stream, w_scratch,
                        nprod);
}

// Number of floats one slab needs for wn_gemm_tn.


static int tn_device_cus() { return wn_device_cus(); }

// Workgroup-tile shape of the LDS-staged TN kernels for an output, or false.
static bool tn_wg_tile(int Mw, int Nw, int* mf, int* nf);

// Rows of column sums in a slab's tail: the tile rows of the workgroup-tile
// kernels (every tile row sums its share of a split's chunks, want_colsum = 2),
// 1 for the shapes of the LDS-free kernel.
int wn_gemm_tn_tail_rows(int Mw, int Nw) {
  int mf = 0, nf = 0;
  if (Mw <= 0 || Nw <= 0 || !tn_wg_tile(Mw, Nw, &mf, &nf)) return 1;
  return Mw / (mf * 32);
}

// Number of floats one slab needs for wn_gemm_tn: the matrix and
// wn_gemm_tn_tail_rows rows of column sums (want_colsum = 1 uses the first).
long wn_gemm_tn_slab_floats(int Mw, int Nw) {
  return (long)Mw * Nw + (long)wn_gemm_tn_tail_rows(Mw, Nw) * Nw;
}

static bool tn_wg_tile(int Mw, int Nw, int* mf, int* nf) {
  int m = 0, n = 0;
  if (Nw % 256 == 0) n = 2; else if (Nw % 128 == 0) n = 1;
  if (Mw % 160 == 0) m = 5; else if (Mw % 128 == 0) m = 4;
  if (m == 5 && n == 2) n = 1;  // <5,2>: 160 accumulators, 2 waves / SIMD
  *mf = m; *nf = n;
  return m && n;
}

// Recommended split count: the grid (tiles x splits) should fill the CUs'
// resident-workgroup capacity exactly once (measured: 768 / 1000 workgroups
// for dW1 / dWs run 129-130 TFLOP/s, 1.33 x that capacity 98); every extra
// split also costs a slab write + read.
int wn_gemm_tn_splits(long rows, int Mw, int Nw, int onehot) {
  if (rows <= 0 || Mw <= 0 || Nw <= 0) return 1;
  int mf, nf;
  long s;
  if (onehot == 2) {  // split-bf16 kernel: 128 x 128 tiles, 3 workgroups / CU
    const int tiles = ((Mw + 127) / 128) * ((Nw + 127) / 128);
    s = 3L * tn_device_cus() / tiles;
  } else if (!onehot && tn_wg_tile(Mw, Nw, &mf, &nf)) {
    const int tiles = (Mw / (mf * 32)) * (Nw / (4 * nf * 32));
    const int occ = mf * nf > 5 ? 3 : 4;
    s = (long)occ * tn_device_cus() / tiles;
  } else {
    const int m32 = (Mw + 31) / 32, n32 = (Nw + 31) / 32;
    int f = (n32 % 2 == 0) ? 2 : 1, e;
    if (onehot) { f = 1; e = (m32 % 2 == 0) ? 2 : 1; }
    else e = m32 % 5 == 0 ? 5 : m32 % 4 == 0 ? 4 : m32 % 2 == 0 ? 2 : 1;
    const int wtiles = ((m32 + e - 1) / e) * ((n32 + f - 1) / f);
    s = (onehot ? 256 : 1024) / ((wtiles + 3) / 4);
  }
  if (s > rows / 64) s = rows / 64;
  return (int)(s < 1 ? 1 : s);
}

// Split-bf16 variant of wn_gemm_tn (dense / plane A only; opt-in).  Needs
// rows % 16 == 0, Mw % 4 == 0, Nw % 4 == 0, 16-byte aligned operands.
int wn_gemm_tn_split(const float* A, long lda, int a_planes, long a_plane_stride,
                     const float* G, long ldg, float* slabs, int splits,
                     long rows, int Mw, int Nw, int want_colsum, int nprod,
                     void* stream) {
  if (!A || !G || !slabs) return WN_ERR_NULL;
  if (rows <= 0 || Mw <= 0 || Nw <= 0 || splits <= 0) return WN_ERR_BAD_SHAPE;
  if (nprod != 3 && nprod != 6 && nprod != 9) return WN_ERR_BAD_SHAPE;
  if (a_planes && Mw != a_planes * 32) return WN_ERR_BAD_SHAPE;
  if ((rows % 16) || (Mw & 3) || (Nw & 3) || (ldg & 3) || (!a_planes && (lda & 3)))
    return WN_ERR_UNSUPPORTED;
  if (!wn_aligned16(A) || !wn_aligned16(G)) return WN_ERR_MISALIGNED;
  GemmTN g;
  g.A = A; g.lda = lda; g.a_plane_stride = a_plane_stride; g.a_planes = a_planes;
  g.codes = nullptr; g.shift = 0; g.T = 1; g.G = G; g.ldg = ldg;
  g.slabs = slabs; g.slab_stride = wn_gemm_tn_slab_floats(Mw, Nw);
  g.rows = rows;
  long rps = (rows + splits - 1) / splits;
  g.rows_per_split = (rps + 15) / 16 * 16;
  g.Mw = Mw; g.Nw = Nw; g.want_colsum = want_colsum;
  g.tiles_m = (Mw + 127) / 128;
  g.tiles_n = (Nw + 127) / 128;
  dim3 grid(g.tiles_m * g.tiles_n, splits), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (nprod == 3) hipLaunchKernelGGL(gemm_tn_split_kernel<3>, grid, block, 0, s, g);
  else if (nprod == 9) hipLaunchKernelGGL(gemm_tn_split_kernel<9>, grid, block, 0, s, g);
  else hipLaunchKernelGGL(gemm_tn_split_kernel<6>, grid, block, 0, s, g);
  return wn_check_launch();
}

int wn_gemm_tn(const float* A, long lda, int a_planes, long a_plane_stride,
               const int32_t* codes, int shift, int T, const float* G,
               long ldg, float* slabs, int splits, long rows, int Mw, int Nw,
               int want_colsum, void* stream) {
  if (!G || !slabs) return WN_ERR_NULL;
  if (!A && !codes) return WN_ERR_NULL;
  if (rows <= 0 || Mw <= 0 || Nw <= 0 || splits <= 0) return WN_ERR_BAD_SHAPE;
  if (a_planes && Mw != a_planes * 32) return WN_ERR_BAD_SHAPE;
  if (codes && T <= 0) return WN_ERR_BAD_SHAPE;
  GemmTN g;
  g.A = A; g.lda = lda; g.a_plane_stride = a_plane_stride; g.a_planes = a_planes;
  g.codes = codes; g.shift = shift; g.T = T > 0 ? T : 1; g.G = G; g.ldg = ldg;
  g.slabs = slabs; g.slab_stride = wn_gemm_tn_slab_floats(Mw, Nw);
  g.rows = rows;
  long rps = (rows + splits - 1) / splits;
  rps = (rps + 95) / 96 * 96;  // whole 32-row chunks and 6/8-row groups
  g.rows_per_split = rps;
  g.Mw = Mw; g.Nw = Nw; g.want_colsum = want_colsum;
  hipStream_t s = (hipStream_t)stream;
  // LDS-staged workgroup-tile kernel when the output is wide and tile-aligned
  if (!codes && (lda % 4 == 0) && (ldg % 4 == 0) && wn_aligned16(A) &&
      wn_aligned16(G)) {
    int mf2 = 0, nf2 = 0;
    tn_wg_tile(Mw, Nw, &mf2, &nf2);
    if (mf2 && nf2) {
      g.tiles_m = Mw / (mf2 * 32);
      g.tiles_n = Nw / (4 * nf2 * 32);
      dim3 grid2(g.tiles_m * g.tiles_n, splits), block2(256);
      // LDS-DMA kernel (its staging addresses are 32-bit byte offsets of a
      // buffer resource: per workgroup a tile's planes -- at most five -- and a
      // split's rows; ragged row counts: the kernel zero-fills its last chunk),
      // else the register-staged kernel
      const long rps = (rows + splits - 1) / splits + 96;
      const long a_bytes = (a_planes ? 5 * a_plane_stride + rps * 32 : rps * lda) * 4;
      const bool dma = a_bytes < (1L << 31) && rps * ldg * 4 < (1L << 31);
#define LAUNCH2(mf, nf)                                                          \
  do {                                                                           \
    if (dma) hipLaunchKernelGGL((gemm_tn3_kernel<mf, nf>), grid2, block2, 0, s, g); \
    else hipLaunchKernelGGL((gemm_tn2_kernel<mf, nf>), grid2, block2, 0, s, g);  \
  } while (0)
      if (mf2 == 5 && nf2 == 2) LAUNCH2(5, 2);
      else if (mf2 == 5) LAUNCH2(5, 1);
      else if (nf2 == 2) LAUNCH2(4, 2);
      else LAUNCH2(4, 1);
#undef LAUNCH2
      return wn_check_launch();
    }
  }
  // (the LDS-free kernel knows the single-row tail only)
  if (want_colsum == 2) return WN_ERR_UNSUPPORTED;
  // wave-tile shape: least padding waste, then largest tile
  const int m32 = (Mw + 31) / 32, n32 = (Nw + 31) / 32;
  int NF = (n32 % 2 == 0) ? 2 : 1;
  int MF;
  if (codes) { NF = 1; MF = (m32 % 2 == 0) ? 2 : 1; }  // one-hot: A is free
  else if (m32 % 5 == 0) MF = 5;
  else if (m32 % 4 == 0) MF = 4;
  else if (m32 % 2 == 0) MF = 2;
  else MF = 1;
  g.tiles_m = (m32 + MF - 1) / MF;
  g.tiles_n = (n32 + NF - 1) / NF;
  const int wtiles = g.tiles_m * g.tiles_n;
  dim3 grid((wtiles + 3) / 4, splits), block(256);
#define LAUNCH(mf, nf) \
  hipLaunchKernelGGL((gemm_tn_kernel<mf, nf>), grid, block, 0, s, g)
  if (NF == 2) {
    if (MF == 5) LAUNCH(5, 2);
    else if (MF == 4) LAUNCH(4, 2);
    else if (MF == 2) LAUNCH(2, 2);
    else LAUNCH(1, 2);
  } else {
    if (MF == 5) LAUNCH(5, 1);
    else if (MF == 4) LAUNCH(4, 1);
    else if (MF == 2) LAUNCH(2, 1);
    else LAUNCH(1, 1);
  }
#undef LAUNCH
  return wn_check_launch();
}

// Few outputs, many slabs (the loss: one value out of ~1000 partials): one wave
// per output, lane l sums slabs l, l + 64, ... in order, then a fixed xor tree.
// Deterministic; replaces one thread walking 1000 dependent loads (26 us).
__global__ __launch_bounds__(64) void reduce_slabs_wave_kernel(
    const float* __restrict__ slabs, int num_slabs, long slab_stride,
    long in_batch_stride, long offset, float* __restrict__ dst,
    long out_batch_stride, int replicate, long rep_stride) {
  const long e = blockIdx.x;
  const int b = blockIdx.y;
  const float* p = slabs + (long)b * in_batch_stride + offset + e;
  float v = 0.f;
  for (int s = threadIdx.x; s < num_slabs; s += 64) v += p[(long)s * slab_stride];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  if (threadIdx.x == 0)
    for (int r = 0; r < replicate; ++r)
      dst[(long)b * out_batch_stride + (long)r * rep_stride + e] = v;
}

int wn_reduce_slabs(const float* slabs, int num_slabs, long slab_stride,
                    int batch, long in_batch_stride, long offset, long n,
                    float* dst, long out_batch_stride, int replicate,
                    long rep_stride, void* stream) {
  if (!slabs || !dst) return WN_ERR_NULL;
  if (num_slabs <= 0 || n <= 0 || batch <= 0 || replicate <= 0)
    return WN_ERR_BAD_SHAPE;
  if (n <= 4 && num_slabs >= 256) {
    hipLaunchKernelGGL(reduce_slabs_wave_kernel, dim3((unsigned)n, batch),
                       dim3(64), 0, (hipStream_t)stream, slabs, num_slabs,
                       slab_stride, in_batch_stride, offset, dst,
                       out_batch_stride, replicate, rep_stride);
    return wn_check_launch();
  }
  if (((n | offset | slab_stride | in_batch_stride | out_batch_stride | rep_stride) & 3) == 0 &&
      wn_aligned16(slabs) && wn_aligned16(dst)) {
    const long n4 = n / 4;
    dim3 grid4((unsigned)((n4 + 63) / 64), batch);
    // few output elements and many slabs: a thread's serial loop over the
    // slabs is the launch's duration -- cut the slabs over four threads
    const bool split = num_slabs >= 32 && (long)grid4.x * batch <= 4096;
    if (split)
      hipLaunchKernelGGL(reduce_slabs4_kernel<4>, grid4, dim3(256), 0, (hipStream_t)stream,
                         slabs, num_slabs, slab_stride, in_batch_stride, offset, n4,
                         dst, out_batch_stride, replicate, rep_stride);
    else
      hipLaunchKernelGGL(reduce_slabs4_kernel<1>, grid4, dim3(64), 0, (hipStream_t)stream,
                         slabs, num_slabs, slab_stride, in_batch_stride, offset, n4,
                         dst, out_batch_stride, replicate, rep_stride);
    return wn_check_launch();
  }
  dim3 grid((unsigned)((n + 255) / 256), batch), block(256);
  hipLaunchKernelGGL(reduce_slabs_kernel, grid, block, 0, (hipStream_t)stream,
                     slabs, num_slabs, slab_stride, in_batch_stride, offset, n,
                     dst, out_batch_stride, replicate, rep_stride);
  return wn_check_launch();
}

int wn_reduce_slabs_mt(const float* slabs, int num_slabs, long slab_stride,
                       long n_main, float* dst_main, long n_tail, float* dst_tail,
                       int replicate, long rep_stride, int tail_rows, void* stream) {
  if (!slabs || !dst_main || (n_tail > 0 && !dst_tail)) return WN_ERR_NULL;
  if (num_slabs <= 0 || n_main <= 0 || n_tail < 0 || replicate < 1 || tail_rows < 1 ||
      slab_stride < n_main + (long)tail_rows * n_tail)
    return WN_ERR_BAD_SHAPE;
  if (((n_main | n_tail | slab_stride | rep_stride) & 3) != 0) return WN_ERR_UNSUPPORTED;
  if (!wn_aligned16(slabs) || !wn_aligned16(dst_main) || (dst_tail && !wn_aligned16(dst_tail)))
    return WN_ERR_MISALIGNED;
  const long n4 = (n_main + n_tail) / 4;
  hipLaunchKernelGGL(reduce_slabs_mt_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(64 * RMT_PARTS), 0,
                     (hipStream_t)stream, slabs, num_slabs, slab_stride, n_main / 4, dst_main,
                     n_tail / 4, dst_tail, replicate, rep_stride, tail_rows);
  return wn_check_launch();
}

int wn_reduce_pair_slabs(const float* slabs, int num_slabs, int CB, int K,
                         int has_dense, int use_bias, float* layer_grad, int C,
                         long off_bias, int tap0, int Ktot, void* stream) {
  if (!slabs || !layer_grad) return WN_ERR_NULL;
  if (num_slabs <= 0 || CB <= 0 || K < 1 || C != CB * 32 || off_bias < 0 || (off_bias & 3) ||
      tap0 < 0 || Ktot < tap0 + K)
    return WN_ERR_BAD_SHAPE;
  if (!wn_aligned16(slabs) || !wn_aligned16(layer_grad)) return WN_ERR_MISALIGNED;
  const int n4 = ((2 * K + 1) * 1024 + 96) / 4;
  hipLaunchKernelGGL(reduce_pair_slabs_kernel, dim3((n4 + 63) / 64, CB * CB), dim3(64 * RPS_PARTS), 0,
                     (hipStream_t)stream, slabs, num_slabs, CB, K, has_dense, use_bias,
                     layer_grad, C, off_bias, tap0, Ktot);
  return wn_check_launch();
}

int wn_transpose(const float* in, int rows, int cols, long in_ld, float* out,
                 long out_ld, void* stream) {
  if (!in || !out) return WN_ERR_NULL;
  if (rows <= 0 || cols <= 0) return WN_ERR_BAD_SHAPE;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
  hipLaunchKernelGGL(transpose_pad_kernel, grid, block, 0, (hipStream_t)stream,
                     in, rows, cols, in_ld, out, out_ld);
  return wn_check_launch();
}

}  // extern "C"
