// Shared device helpers for the gfx950 (CDNA4 / MI355X) WaveNet kernels.
//
// Conventions used by every MFMA kernel in this directory
// -------------------------------------------------------
// * Activations live in HBM as "planes": [rows][32] fp32, rows = B*T, one
//   128-byte line per audio sample (reference layout [B,T,C] with C padded to
//   32; wavenet/model.py keeps residual/dilation channels innermost too).
// * A "fragment" is a 32(time) x 32(channel) tile held by one 64-lane wave in
//   16 VGPRs per lane.  Lane l = (j = l & 31, h = l >> 5) owns time row j and
//   the 16 channels  c(r,h) = 8*(r>>2) + 4*h + (r&3),  r = 0..15.
//   This is exactly the C/D layout of v_mfma_f32_32x32x2_f32 when the product
//   is computed transposed (channels on the M axis, time on the N axis):
//       OUT^T[ch, time] += W^T[ch, k] * X^T[k, time]
//   so an accumulator IS a fragment: it feeds the next MFMA as the B operand
//   with no lane movement and no LDS round trip, and it is loaded/stored with
//   four 16-byte accesses per lane (channels 8q+4h .. 8q+4h+3, q = 0..3).
// * Weights are the A operand, read from LDS in the reference's own
//   [k][n] (Cin-major, Cout contiguous) order: lane (i, h) reads
//   W[c(r,h)][n0 + i] -> 32 consecutive floats per half-wave, conflict-free.
// * fp32 in / fp32 accumulate MFMA is bit-for-bit an fmaf chain, so parity
//   with the fp32 CPU oracle is limited only by summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WN_OK 0
#define WN_ERR_BAD_SHAPE (-1)
#define WN_ERR_UNSUPPORTED (-2)
#define WN_ERR_MISALIGNED (-3)
#define WN_ERR_LAUNCH (-4)
#define WN_ERR_NULL (-5)
// (variant word of the stack launches: see include/wavenet_hip.h)

#define WN_CH 32  // padded residual/dilation channel count of every plane

// Layer parameter block (floats), contiguous per layer in the flat buffer:
//   Wf[2][32][32] Wg[2][32][32] Wd[32][32] bf[32] bg[32] bd[32]
//   (+ Wgc_filter[G][32] Wgc_gate[G][32] when globally conditioned)
#define LAYER_W_FLOATS (5 * 1024)
#define LAYER_OFF_BF 5120
#define LAYER_OFF_BG 5152
#define LAYER_OFF_BD 5184
#define LAYER_BLOCK_FLOATS (LAYER_W_FLOATS + 96)
#define LAYER_OFF_GC LAYER_BLOCK_FLOATS

static inline int wn_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? WN_OK : WN_ERR_LAUNCH;
}

// Compute units of the CURRENT device.  Queried on every call (the HIP
// runtime answers from its own device table; no mutable global here, see the
// threading contract in include/wavenet_hip.h).
static inline int wn_device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) !=
          hipSuccess || n <= 0)
    n = 256;
  return n;
}

static inline bool wn_aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

#ifdef __HIPCC__

__device__ __forceinline__ f32x16 frag_zero() {
  f32x16 f;
#pragma unroll
  for (int r = 0; r < 16; ++r) f[r] = 0.f;
  return f;
}

// Load the fragment of plane rows [row0, row0+32) ; `rowp` already points at
// this lane's row (row0 + j), `h` = lane >> 5.  valid=false gives zeros (rows
// before the clip start / past its end: the causal zero padding of
// wavenet/ops.py:50-51).
__device__ __forceinline__ f32x16 frag_load(const float* __restrict__ rowp,
                                            int h, bool valid) {
  f32x16 f;
  const f32x4* p = reinterpret_cast<const f32x4*>(rowp + 4 * h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) v = p[2 * q];
    f[4 * q + 0] = v[0];
    f[4 * q + 1] = v[1];
    f[4 * q + 2] = v[2];
    f[4 * q + 3] = v[3];
  }
  return f;
}

__device__ __forceinline__ void frag_store(float* __restrict__ rowp, int h,
                                           bool valid, const f32x16& f) {
  if (!valid) return;
  f32x4* p = reinterpret_cast<f32x4*>(rowp + 4 * h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {f[4 * q + 0], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
    p[2 * q] = v;
  }
}

// ---------------------------------------------------------------------------
// Coalesced plane I/O through a wave-private LDS tile.
// A fragment access touches 32 rows x 32 B per instruction; HBM/L2 like whole
// 128-byte lines better (measured: 3.6 -> see DESIGN.md).  So global memory is
// always accessed "row-contiguous" (lane l <-> row 8c + (l>>3), 16-byte chunk
// l&7: one instruction = 8 full rows = 1 KiB contiguous) and the 32x32 tile is
// turned into fragment layout through LDS.  Tile layout: [row][32 floats],
// chunk c of row r stored at chunk c ^ (r & 7): conflict-free for the
// row-contiguous side, 2-way for the fragment side.  A wave's DS operations
// execute in order, so no barrier is needed between the two sides.
// ---------------------------------------------------------------------------
struct RowRegs {
  f32x4 v[4];
};

// rows [lo, hi) of the 32-row tile are real, the others read as zero
__device__ __forceinline__ RowRegs rows_load(const float* __restrict__ tile0,
                                             int lane, int lo, int hi) {
  RowRegs R;
  const float* p = tile0 + (long)(lane >> 3) * 32 + (lane & 7) * 4;
  if (lo <= 0 && hi >= 32) {  // wave-uniform: whole tile real -> no predicates
#pragma unroll
    for (int c = 0; c < 4; ++c)
      R.v[c] = *reinterpret_cast<const f32x4*>(p + c * 256);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 8 * c + (lane >> 3);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r >= lo && r < hi) v = *reinterpret_cast<const f32x4*>(p + c * 256);
      R.v[c] = v;
    }
  }
  return R;
}

// address-space-qualified pointers of __builtin_amdgcn_global_load_lds
typedef const __attribute__((address_space(1))) void* wn_gptr_t;
typedef __attribute__((address_space(3))) void* wn_lptr_t;

// LDS-DMA of one [32][32] fp32 tile (global rows 128 B apart) into the
// swizzled LDS tile layout of rows_to_lds, without staging registers: one
// wave-instruction writes 1 KiB of LDS linearly (lane l -> base + 16 l = row
// l >> 3, slot l & 7), so the swizzle is applied to the per-lane SOURCE chunk:
// slot s of row r holds global chunk s ^ (r & 7).  Rows outside [lo, hi) read
// as zero: the tile is zero-filled by ds_write first and the DMA of those rows
// is masked off.  `lds_tile` must be wave-uniform.  Completion: vmcnt.
// AUX: cache policy of the loads (16 = sc1, device scope: see wn_stack.hip).
template <int AUX = 0>
__device__ __forceinline__ void tile_dma(float* lds_tile,
                                         const float* __restrict__ tile0,
                                         int lane, int lo, int hi) {
  const int rr = lane >> 3;
  const float* src = tile0 + (long)rr * 32 + (((lane & 7) ^ (rr & 7)) << 2);
  if (lo <= 0 && hi >= 32) {  // wave-uniform: whole tile real
#pragma unroll
    for (int c = 0; c < 4; ++c)
      __builtin_amdgcn_global_load_lds((wn_gptr_t)(src + c * 256),
                                       (wn_lptr_t)(lds_tile + c * 256), 16, 0, AUX);
  } else {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<f32x4*>(lds_tile + c * 256 + lane * 4) = zero;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 8 * c + rr;
      if (r >= lo && r < hi)
        __builtin_amdgcn_global_load_lds((wn_gptr_t)(src + c * 256),
                                         (wn_lptr_t)(lds_tile + c * 256), 16, 0, AUX);
    }
  }
}

__device__ __forceinline__ void rows_store(float* __restrict__ tile0, int lane,
                                           int hi, const RowRegs& R) {
  float* p = tile0 + (long)(lane >> 3) * 32 + (lane & 7) * 4;
  if (hi >= 32) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(p + c * 256) = R.v[c];
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 8 * c + (lane >> 3);
      if (r < hi) *reinterpret_cast<f32x4*>(p + c * 256) = R.v[c];
    }
  }
}

__device__ __forceinline__ void rows_to_lds(float* tile, int lane,
                                            const RowRegs& R) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = 8 * c + (lane >> 3);
    *reinterpret_cast<f32x4*>(tile + r * 32 + (((lane & 7) ^ (r & 7)) << 2)) =
        R.v[c];
  }
}

__device__ __forceinline__ RowRegs rows_from_lds(const float* tile, int lane) {
  RowRegs R;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = 8 * c + (lane >> 3);
    R.v[c] = *reinterpret_cast<const f32x4*>(
        tile + r * 32 + (((lane & 7) ^ (r & 7)) << 2));
  }
  return R;
}

__device__ __forceinline__ f32x16 frag_from_lds(const float* tile, int j,
                                                int h) {
  f32x16 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(
        tile + j * 32 + (((2 * q + h) ^ (j & 7)) << 2));
    f[4 * q + 0] = v[0];
    f[4 * q + 1] = v[1];
    f[4 * q + 2] = v[2];
    f[4 * q + 3] = v[3];
  }
  return f;
}

__device__ __forceinline__ void frag_to_lds(float* tile, int j, int h,
                                            const f32x16& f) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
    *reinterpret_cast<f32x4*>(tile + j * 32 + (((2 * q + h) ^ (j & 7)) << 2)) = v;
  }
}

// element [row][ch] of a swizzled tile (transposed, channel-on-lane reads)
__device__ __forceinline__ float tile_elem(const float* tile, int row, int ch) {
  return tile[row * 32 + (((ch >> 2) ^ (row & 7)) << 2) + (ch & 3)];
}

// Per-channel vector (bias) from LDS/global as a fragment (same value for
// every time row).
__device__ __forceinline__ f32x16 frag_bcast(const float* __restrict__ vec,
                                             int h) {
  f32x16 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[4 * q + e] = vec[8 * q + 4 * h + e];
  }
  return f;
}

// acc^T[n0+i, time] += sum_k W[k][n0+i] * frag[time, k]   (k over 32 channels)
// wl: LDS, row-major [32][ldw] with the contraction index k as the row.
// `wl_lane` must already be  wl + n0 + i + 4*h*ldw  (per-lane base), so the
// 16 reads are base + compile-time offsets.
template <int LDW>
__device__ __forceinline__ void mma32(f32x16& acc, const f32x16& frag,
                                      const float* wl_lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float a = wl_lane[(8 * (r >> 2) + (r & 3)) * LDW];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, frag[r], acc, 0, 0, 0);
  }
}

__device__ __forceinline__ float wn_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}
__device__ __forceinline__ float wn_tanh(float x) {
  // 1 - 2/(e^{2x}+1); saturates correctly at +-inf of the exp.
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
}

// Backward of a residual block without pre-activation-gradient planes
// (layer_bwd2d_kernel in wn_layer.hip, stack_bwd_kernel in wn_stack.hip):
// transposed weight image per layer, 5 x [cout][33] + padding = 21 KiB
#define B2_WIMG 5376

// (da_f, da_g) fragments from dz, z, sigmoid fragments; tanh = z / sigmoid
__device__ __forceinline__ void gate_grad(const f32x16& dz, const f32x16& zz,
                                          const f32x16& ss, f32x16& df,
                                          f32x16& dg) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    // seven vector instructions per element (add, rcp, mul, fma, mul, fma,
    // mul): on gfx950 the f32 MFMA and the vector ALU are one resource, so
    // every instruction here is matrix-pipe time (tools/ubench/mfma_valu.hip).
    // A sigmoid that underflowed to 0 must not turn 0 * inf into NaN: + 1e-30
    // (no change to any sigmoid above 1e-22; fmaxf compiles to TWO v_max, the
    // first one canonicalising its operand).
    const float sgm = ss[r];
    const float th = zz[r] * __builtin_amdgcn_rcpf(sgm + 1e-30f);
    df[r] = dz[r] * __builtin_fmaf(-zz[r], th, sgm);      // dz * sig * (1 - tanh^2)
    dg[r] = dz[r] * __builtin_fmaf(-zz[r], sgm, zz[r]);   // dz * tanh * sig * (1 - sig)
  }
}

#endif  // __HIPCC__
