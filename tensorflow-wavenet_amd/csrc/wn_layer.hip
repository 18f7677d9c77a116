// Fused residual-block kernels (forward, backward-data, backward-weights).
//
// Replaces, per dilation layer, the ~40 TensorFlow ops of
// WaveNetModel._create_dilation_layer (wavenet/model.py:236-330) and the two
// causal_conv calls inside it (wavenet/ops.py:46-62: pad, time_to_batch,
// conv1d, batch_to_time, slice -- for filter and for gate):
//
//   a_f = x[t-d]*Wf[0] + x[t]*Wf[1] + b_f (+gc_f)     model.py:269-290
//   a_g = x[t-d]*Wg[0] + x[t]*Wg[1] + b_g (+gc_g)
//   z   = tanh(a_f) * sigmoid(a_g)                     model.py:292
//   x'  = x + z*Wd + b_d                               model.py:294-300,330
// The skip 1x1 (model.py:303-305) is NOT done here: z is written once as a
// plane and all L skip convs become one [B*T, L*32] x [L*32, S] MFMA GEMM
// (wn_gemm.hip) -- see DESIGN.md.
//
// Layer parameter block (all channel dims padded to 32, reference [K,Cin,Cout]
// order inside each matrix), contiguous in the flat parameter buffer:
//   Wf[2][32][32]  Wg[2][32][32]  Wd[32][32]  bf[32] bg[32] bd[32]
#include "wn_common.h"


// Persistent workgroups: one 1024-thread workgroup (16 waves, 4 per SIMD) per
// CU stages the layer's weights into LDS ONCE, then every wave walks 32-row
// tiles (tile = clip b, rows t0..t0+31) with a grid stride.
#define LAYER_WG 1024
#define LAYER_WAVES (LAYER_WG / 64)

// SAVE: what the backward pass will need besides x and z.  0 = nothing
// (inference), 1 = tanh and sigmoid planes (un-fused / generic backward
// kernels), 2 = the sigmoid plane only (layer_bwd2d_kernel recovers tanh as
// z / sigmoid: 512 instead of 640 bytes per sample and layer).
template <bool HAS_DENSE, int SAVE>
__global__ __launch_bounds__(LAYER_WG) void layer_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ xo, float* __restrict__ z,
    float* __restrict__ th, float* __restrict__ sg,
    const float* __restrict__ wblock,      // LAYER_BLOCK_FLOATS
    const float* __restrict__ bias_fg,     // [B or 1][64] (bias + gc), or null
    int bias_clip_stride, int B, int T, int d) {
#ifdef FWD_STAMPS   // diagnostic build: th carries a stamp buffer [grid][2][16]
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(th);
#define FWSTAMP(i)                                                          \
  if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 15)) \
    dbg[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 15)) * 16 + (i)] = __builtin_amdgcn_s_memtime()
  FWSTAMP(0);
  if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 15))
    dbg[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 15)) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
#else
#define FWSTAMP(i)
#endif
  __shared__ __attribute__((aligned(1024))) float wl[LAYER_W_FLOATS + 32];
  __shared__ __attribute__((aligned(16))) float tiles[LAYER_WAVES * 2 * 1024];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  // weights: 20 pieces of 1 KiB straight into LDS by LDS-DMA (asynchronous, no
  // staging registers), issued BEFORE the first tile's loads and waited for
  // together with them: the two latencies overlap
  for (int p = wave; p < LAYER_W_FLOATS / 256; p += LAYER_WAVES)
    __builtin_amdgcn_global_load_lds((wn_gptr_t)(wblock + p * 256 + lane * 4),
                                     (wn_lptr_t)(wl + p * 256), 16, 0, 0);
  if (tid < 32) wl[LAYER_W_FLOATS + tid] = wblock[LAYER_OFF_BD + tid];
  float* ta = tiles + wave * 2048;
  float* tb = ta + 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  // few tiles (small batches): one tile per wave, spread over ALL CUs wave by
  // wave (layer_grid sizes the grid for it), so a SIMD runs one tile, not four
  const bool sparse = ntiles < (int)gridDim.x * LAYER_WAVES;
  int tile = sparse ? wave * (int)gridDim.x + (int)blockIdx.x
                    : (int)blockIdx.x * LAYER_WAVES + wave;
  // coalesced loads (8 full rows per instruction); always assigns rc / rp
  // (zeros for a wave without a tile)
  RowRegs rc, rp;
  auto load_tile = [&](int tl) {
    const bool any = tl < ntiles;
    const int b = any ? tl / tiles_per_clip : 0;
    const int t0 = any ? (tl - b * tiles_per_clip) * 32 : 0;
    const int hi = any ? min(32, T - t0) : 0;
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    rc = rows_load(x + off0, lane, 0, hi);
    rp = rows_load(x + off0 - (size_t)d * WN_CH, lane, max(0, d - t0), hi);
  };
  load_tile(tile);
  FWSTAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FWSTAMP(2);
  __syncthreads();                 // every wave's weight pieces have landed
  FWSTAMP(3);
  for (bool first = true; tile < ntiles;
       tile += gridDim.x * LAYER_WAVES, first = false) {
    // opaque per-iteration LDS offset: keeps the 80 weight reads next to their
    // MFMAs instead of being hoisted into (and spilling from) registers
    int woff = j + 4 * h * 32;  // n0 = 0, i = j
    asm volatile("" : "+v"(woff));
    const float* wlane = wl + woff;
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);          // real rows of this tile
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    if (!first) load_tile(tile);
    // global rows -> LDS -> fragments
    rows_to_lds(ta, lane, rc);
    rows_to_lds(tb, lane, rp);
    __builtin_amdgcn_wave_barrier();
    f32x16 xc = frag_from_lds(ta, j, h);
    f32x16 xp = frag_from_lds(tb, j, h);
    FWSTAMP(4);
    f32x16 af, ag;
    if (bias_fg) {
      const float* bp = bias_fg + (size_t)b * bias_clip_stride;
      af = frag_bcast(bp, h);
      ag = frag_bcast(bp + 32, h);
    } else {
      af = frag_zero();
      ag = frag_zero();
    }
    // current tap first: the order of stack_fwd_kernel (wn_stack.hip), where
    // these products run while the dilated tap is still on its way
    mma32<32>(af, xc, wlane + 1 * 1024);  // Wf[1]: current tap
    mma32<32>(ag, xc, wlane + 3 * 1024);  // Wg[1]
    mma32<32>(af, xp, wlane + 0 * 1024);  // Wf[0]: past tap
    mma32<32>(ag, xp, wlane + 2 * 1024);  // Wg[0]
    FWSTAMP(5);
    f32x16 zz;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      af[r] = wn_tanh(af[r]);
      ag[r] = wn_sigmoid(ag[r]);
      zz[r] = af[r] * ag[r];
    }
    FWSTAMP(6);
    // fragments -> LDS -> coalesced stores
    __builtin_amdgcn_wave_barrier();
    frag_to_lds(ta, j, h, zz);
    if (SAVE == 1) frag_to_lds(tb, j, h, af);
    if (SAVE == 2) frag_to_lds(tb, j, h, ag);
    __builtin_amdgcn_wave_barrier();
    rows_store(z + off0, lane, hi, rows_from_lds(ta, lane));
    if (SAVE == 1) {
      rows_store(th + off0, lane, hi, rows_from_lds(tb, lane));
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(tb, j, h, ag);
      __builtin_amdgcn_wave_barrier();
      rows_store(sg + off0, lane, hi, rows_from_lds(tb, lane));
    }
    if (SAVE == 2) rows_store(sg + off0, lane, hi, rows_from_lds(tb, lane));
    FWSTAMP(7);
    if (HAS_DENSE) {
      const f32x16 bd = frag_bcast(wl + LAYER_W_FLOATS + (woff - j - 128 * h), h);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = xc[r] + bd[r];
      mma32<32>(acc, zz, wlane + 4 * 1024);  // Wd
      FWSTAMP(8);
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, acc);
      __builtin_amdgcn_wave_barrier();
      rows_store(xo + off0, lane, hi, rows_from_lds(ta, lane));
    }
    FWSTAMP(9);
#ifdef FWD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores drained (diagnostic only)
#endif
    FWSTAMP(10);
    __builtin_amdgcn_wave_barrier();
  }
#ifdef FWD_STAMPS
  if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 15))
    dbg[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 15)) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
#undef FWSTAMP
}

// Backward-weights of one layer.  Every wave walks 32-row tiles and keeps 5
// accumulator tiles with channels on BOTH MFMA axes (the row pair of a step
// is the contraction):
//   dWf[0] += x[t-d]^T da_f   dWf[1] += x[t]^T da_f   (same for gate)
//   dWd    += z^T dxin        db_f,g += colsum(da)    dbd += colsum(dxin)
// Tiles are loaded as coalesced 16-byte fragments (time on lanes), pushed
// through a wave-private XOR-swizzled LDS tile and read back transposed
// (channel on lanes) as MFMA operands -- conflict-free both ways; a wave's DS
// operations execute in order, so no barrier is needed.  The 4 waves of a
// workgroup reduce through LDS and the workgroup writes ONE slab;
// wn_reduce_slabs sums slabs in a fixed order (deterministic, no atomics).
// Slab layout == layer gradient block layout (LAYER_BLOCK_FLOATS).
#define WG_TILE 1024  // floats per 32x32 LDS tile


template <bool HAS_DENSE>
__global__ __launch_bounds__(256) void layer_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ daf,
    const float* __restrict__ dag, const float* __restrict__ z,
    const float* __restrict__ dxin, float* __restrict__ slabs, int B, int T,
    int d, int CB, long plane_stride) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 4 * WG_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // blockIdx.y: the (input block a, output block b) pair of a channel-block
  // layer with two taps (wn_layer_wgrad_k: every plane of the pair read ONCE
  // for all five products; the generic-tap kernel makes a pass per tap)
  {
    const int pair = blockIdx.y, pa = pair / CB, pb = pair - pa * CB;
    x += (size_t)pa * plane_stride;
    daf += (size_t)pb * plane_stride;
    dag += (size_t)pb * plane_stride;
    if (HAS_DENSE) {
      z += (size_t)pa * plane_stride;
      dxin += (size_t)pb * plane_stride;
    }
  }
  float* t_xp = lds + wave * 4 * WG_TILE;
  float* t_xc = t_xp + WG_TILE;
  float* t_f = t_xc + WG_TILE;
  float* t_g = t_f + WG_TILE;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  const int nwaves = gridDim.x * 4;
  f32x16 cf0 = frag_zero(), cf1 = frag_zero(), cg0 = frag_zero(),
         cg1 = frag_zero(), cd = frag_zero();
  float sf = 0.f, sgs = 0.f, sd = 0.f;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += nwaves) {
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    {
      const RowRegs rxc = rows_load(x + off0, lane, 0, hi);
      const RowRegs rxp = rows_load(x + off0 - (size_t)d * WN_CH, lane,
                                    max(0, d - t0), hi);
      const RowRegs rf = rows_load(daf + off0, lane, 0, hi);
      const RowRegs rg = rows_load(dag + off0, lane, 0, hi);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(t_xp, lane, rxp);
      rows_to_lds(t_xc, lane, rxc);
      rows_to_lds(t_f, lane, rf);
      rows_to_lds(t_g, lane, rg);
      __builtin_amdgcn_wave_barrier();
    }
    RowRegs rz, rd;
    if (HAS_DENSE) {
      rz = rows_load(z + off0, lane, 0, hi);
      rd = rows_load(dxin + off0, lane, 0, hi);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * s + h;
      const float axp = tile_elem(t_xp, row, i), axc = tile_elem(t_xc, row, i);
      const float bf = tile_elem(t_f, row, i), bg = tile_elem(t_g, row, i);
      cf0 = __builtin_amdgcn_mfma_f32_32x32x2f32(axp, bf, cf0, 0, 0, 0);
      cf1 = __builtin_amdgcn_mfma_f32_32x32x2f32(axc, bf, cf1, 0, 0, 0);
      cg0 = __builtin_amdgcn_mfma_f32_32x32x2f32(axp, bg, cg0, 0, 0, 0);
      cg1 = __builtin_amdgcn_mfma_f32_32x32x2f32(axc, bg, cg1, 0, 0, 0);
      sf += bf;
      sgs += bg;
    }
    if (HAS_DENSE) {
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(t_xp, lane, rz);   // reuse: the reads above are already issued
      rows_to_lds(t_xc, lane, rd);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int row = 2 * s + h;
        const float az = tile_elem(t_xp, row, i), bd = tile_elem(t_xc, row, i);
        cd = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bd, cd, 0, 0, 0);
        sd += bd;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // column sums: add the two row-parity halves
  sf += __shfl_xor(sf, 32);
  sgs += __shfl_xor(sgs, 32);
  sd += __shfl_xor(sd, 32);
  __syncthreads();  // all waves done with their LDS tiles; reuse as `red`
  float* red = lds;
  // cross-wave reduction through LDS, fixed order wave 0..3
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      // C tile: row m = 8*(r>>2)+4*h+(r&3) (A-operand channel), col = i.
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 8 * (r >> 2) + 4 * h + (r & 3);
        const int e = m * 32 + i;
        if (w == 0) {
          red[0 * 1024 + e] = cf0[r];
          red[1 * 1024 + e] = cf1[r];
          red[2 * 1024 + e] = cg0[r];
          red[3 * 1024 + e] = cg1[r];
          red[4 * 1024 + e] = cd[r];
        } else {
          red[0 * 1024 + e] += cf0[r];
          red[1 * 1024 + e] += cf1[r];
          red[2 * 1024 + e] += cg0[r];
          red[3 * 1024 + e] += cg1[r];
          red[4 * 1024 + e] += cd[r];
        }
      }
      if (h == 0) {
        if (w == 0) {
          red[LAYER_W_FLOATS + i] = sf;
          red[LAYER_W_FLOATS + 32 + i] = sgs;
          red[LAYER_W_FLOATS + 64 + i] = sd;
        } else {
          red[LAYER_W_FLOATS + i] += sf;
          red[LAYER_W_FLOATS + 32 + i] += sgs;
          red[LAYER_W_FLOATS + 64 + i] += sd;
        }
      }
    }
    __syncthreads();
  }
  float* out = slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * LAYER_BLOCK_FLOATS;
  for (int e = tid; e < LAYER_BLOCK_FLOATS; e += 256) out[e] = red[e];
}

// The same for 64- and 128-channel layers (two / four channel blocks, two taps)
// with ALL input blocks and TWO output blocks per pass: per pair and workgroup
// the kernel above reads x_a twice, z_a, da_f / da_g / dxin of b -- every plane
// of a 64-channel layer twice per launch (89 us a layer at B x T = 128000,
// 4.4 TB/s), of a 128-channel layer four times (321 us).  Here a workgroup
// stages the tiles of a 32-row step ONCE per pass (LDS-DMA through buffer
// resources: a row outside the clip asks for an offset past the resource and
// lands as zeros; two buffers: the next step lands during this one's
// products) and its eight waves are the pass's CB x 2 pairs (CB = 4), or the
// four pairs x two halves of a step's rows (CB = 2: two waves per SIMD either
// way, one's barrier / wait under the other's MFMAs; the upper halves hand
// their sums to the lower ones through LDS at the end).  Every pair writes its
// own slab: no reduction across waves.  Slab layout as layer_wgrad_kernel;
// blockIdx.y = pass (output blocks 2 y, 2 y + 1).
template <int CB, bool HAS_DENSE>
__global__ __launch_bounds__(512) void layer_wgrad_cbn_kernel(
    const float* __restrict__ x, const float* __restrict__ daf,
    const float* __restrict__ dag, const float* __restrict__ z,
    const float* __restrict__ dxin, float* __restrict__ slabs, int B, int T,
    int d, long plane_stride) {
  // slots of a step: x_a[t] | x_a[t-d] | z_a (a < CB), then f_b | g_b | d_b (b < 2)
  // (CB = 2: three buffers, two steps in flight and a counted wait; CB = 4:
  // two, 144 KiB either way)
  constexpr int SLOTS = 3 * CB + 6, NH = 8 / (2 * CB), NBUF = CB == 2 ? 3 : 2;
  __shared__ __attribute__((aligned(1024))) float lds[NBUF * SLOTS * WG_TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int pair = wave % (2 * CB), half = wave / (2 * CB);   // (half: CB = 2 only)
  const int pa = pair >> 1, pb = pair & 1;
  const int b0 = 2 * blockIdx.y;                               // the pass's output blocks
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  auto slot_plane = [&](int s) -> const float* {
    if (s < 2 * CB) return x + (long)(s % CB) * plane_stride;
    if (s < 3 * CB) return z + (long)(s - 2 * CB) * plane_stride;
    const int q = s - 3 * CB, bb = b0 + (q & 1);
    return (q < 2 ? daf : q < 4 ? dag : dxin) + (long)bb * plane_stride;
  };
  // the lane's swizzled source offset inside a tile (tile_dma's layout: slot
  // l & 7 of row l >> 3 holds global chunk (l & 7) ^ (row & 7))
  const int vswz = ((lane >> 3) * 32 + (((lane & 7) ^ ((lane >> 3) & 7)) << 2)) * 4;
  // wave w stages slots w, w + 8, w + 16: 4 instructions a slot (the slots'
  // planes looked up once, not per step)
  const float* my_plane[3];
  bool my_on[3], my_past[3];
  int my_slots = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int s = wave + 8 * k;
    my_on[k] = s < SLOTS && (HAS_DENSE || !((s >= 2 * CB && s < 3 * CB) || s >= 3 * CB + 4));
    my_past[k] = s >= CB && s < 2 * CB;
    my_plane[k] = slot_plane(my_on[k] ? s : 0);
    my_slots += my_on[k] ? 1 : 0;
  }
  auto stage = [&](int buf, int tile) {
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);
    const int off0 = (b * T + t0) * (WN_CH * 4);           // bytes (< 2^31: host check)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!my_on[k]) continue;
      const int s = wave + 8 * k;
      const bool past = my_past[k];
      const int lo = past ? max(0, d - t0) : 0;
      const int base = off0 - (past ? d * (WN_CH * 4) : 0) + vswz;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)my_plane[k], 0, 0x7fffffff, 0x00020000);
      float* dst = lds + (buf * SLOTS + s) * WG_TILE;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = 8 * c + (lane >> 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs, (__attribute__((address_space(3))) void*)(dst + c * 256), 16,
            r >= lo && r < hi ? base + c * 1024 : (int)0x80000000u, 0, 0, 0);
      }
    }
  };
  // element [row 2 s + h][channel i] of slot n of buffer 0:
  // tp[s & 3][n * WG_TILE + 64 * s]  (the XOR swizzle of tile_elem, once)
  const float* tp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int off = 32 * h + ((((i >> 2) ^ (2 * k + h)) & 7) << 2) + (i & 3);
    asm volatile("" : "+v"(off));
    tp[k] = lds + off;
  }
  f32x16 cf0 = frag_zero(), cf1 = frag_zero(), cg0 = frag_zero(),
         cg1 = frag_zero(), cd = frag_zero();
  float sf = 0.f, sgs = 0.f, sd = 0.f;
  int tile = blockIdx.x, buf = 0;
  const int G = gridDim.x;
  if (tile < ntiles) stage(0, tile);
  if (NBUF == 3 && tile + G < ntiles) stage(1, tile + G);
  for (; tile < ntiles; tile += G, buf = buf == NBUF - 1 ? 0 : buf + 1) {
    // this step's tiles have landed (every wave waits for its own DMAs; with
    // three buffers the step behind it -- four instructions a slot -- stays in
    // flight), and every wave is through the step before: its buffer takes the
    // step after (next)
    if (NBUF == 3 && tile + G < ntiles && my_slots == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (NBUF == 3 && tile + G < ntiles && my_slots == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (tile + (NBUF - 1) * G < ntiles)
      stage(buf == 0 ? NBUF - 1 : buf - 1, tile + (NBUF - 1) * G);
    const int bo = buf * SLOTS * WG_TILE;
    const int oxc = bo + pa * WG_TILE, oxp = bo + (CB + pa) * WG_TILE,
              oz = bo + (2 * CB + pa) * WG_TILE;
    const int of = bo + (3 * CB + pb) * WG_TILE, og = bo + (3 * CB + 2 + pb) * WG_TILE,
              od = bo + (3 * CB + 4 + pb) * WG_TILE;
    // (the six operands of step s + 1 are requested before the MFMAs of step
    // s, order pinned)
    constexpr int NS = 16 / NH;                  // steps of two rows per wave
    const float* q0 = tp[0] + 64 * NS * half;
    float axc = q0[oxc], axp = q0[oxp], bf = q0[of], bg = q0[og];
    float az = HAS_DENSE ? q0[oz] : 0.f, bd = HAS_DENSE ? q0[od] : 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float* q = tp[(s + 1) & 3] + 64 * ((s + 1) & (NS - 1)) + 64 * NS * half;
      const float naxc = q[oxc], naxp = q[oxp], nbf = q[of], nbg = q[og];
      const float naz = HAS_DENSE ? q[oz] : 0.f, nbd = HAS_DENSE ? q[od] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      cf0 = __builtin_amdgcn_mfma_f32_32x32x2f32(axp, bf, cf0, 0, 0, 0);
      cf1 = __builtin_amdgcn_mfma_f32_32x32x2f32(axc, bf, cf1, 0, 0, 0);
      cg0 = __builtin_amdgcn_mfma_f32_32x32x2f32(axp, bg, cg0, 0, 0, 0);
      cg1 = __builtin_amdgcn_mfma_f32_32x32x2f32(axc, bg, cg1, 0, 0, 0);
      sf += bf;
      sgs += bg;
      if (HAS_DENSE) {
        cd = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bd, cd, 0, 0, 0);
        sd += bd;
      }
      __builtin_amdgcn_sched_barrier(0);
      axc = naxc; axp = naxp; bf = nbf; bg = nbg; az = naz; bd = nbd;
    }
  }
  sf += __shfl_xor(sf, 32);
  sgs += __shfl_xor(sgs, 32);
  sd += __shfl_xor(sd, 32);
  if (NH == 2) {
    // the upper-half waves hand their sums to their pair's lower-half wave
    // (lane-linear through the staging buffers, which nobody reads any more)
    __syncthreads();
    float* hand = lds + pair * (5 * 1024 + 256);
    if (half == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hand[(0 * 16 + r) * 64 + lane] = cf0[r];
        hand[(1 * 16 + r) * 64 + lane] = cf1[r];
        hand[(2 * 16 + r) * 64 + lane] = cg0[r];
        hand[(3 * 16 + r) * 64 + lane] = cg1[r];
        hand[(4 * 16 + r) * 64 + lane] = cd[r];
      }
      hand[5 * 1024 + lane] = sf;
      hand[5 * 1024 + 64 + lane] = sgs;
      hand[5 * 1024 + 128 + lane] = sd;
    }
    __syncthreads();
    if (half == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      cf0[r] += hand[(0 * 16 + r) * 64 + lane];
      cf1[r] += hand[(1 * 16 + r) * 64 + lane];
      cg0[r] += hand[(2 * 16 + r) * 64 + lane];
      cg1[r] += hand[(3 * 16 + r) * 64 + lane];
      cd[r] += hand[(4 * 16 + r) * 64 + lane];
    }
    sf += hand[5 * 1024 + lane];
    sgs += hand[5 * 1024 + 64 + lane];
    sd += hand[5 * 1024 + 128 + lane];
  }
  // the pair's slab (pair index a * CB + b as the per-pair kernel's
  // blockIdx.y); C tile row m = 8 (r >> 2) + 4 h + (r & 3) (A-operand channel),
  // column i
  float* out = slabs + ((size_t)(pa * CB + b0 + pb) * gridDim.x + blockIdx.x) * LAYER_BLOCK_FLOATS;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int e = (8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i;
    out[0 * 1024 + e] = cf0[r];
    out[1 * 1024 + e] = cf1[r];
    out[2 * 1024 + e] = cg0[r];
    out[3 * 1024 + e] = cg1[r];
    out[4 * 1024 + e] = cd[r];
  }
  if (h == 0) {
    out[LAYER_W_FLOATS + i] = sf;
    out[LAYER_W_FLOATS + 32 + i] = sgs;
    out[LAYER_W_FLOATS + 64 + i] = sd;
  }
}

// ---------------------------------------------------------------------------
#define B2_WAVES 8
// (B2_WIMG and gate_grad live in wn_common.h: wn_stack.hip shares them)

// Weight image of layer_bwd2d_kernel: the five 32 x 32 matrices of a layer
// block transposed with row stride 33 ([m][cout cc][cin rr] at m*1056 + cc*33 +
// rr; the layout the kernel's LDS reads are conflict-free on), one 21 KiB
// image per layer, rebuilt once per step for all layers.
__global__ void bwd2_pack_kernel(const float* __restrict__ layer0, long layer_stride,
                                 float* __restrict__ img) {
  const float* blk = layer0 + (long)blockIdx.x * layer_stride;
  float* out = img + (long)blockIdx.x * B2_WIMG;
  for (int i = threadIdx.x; i < 5120; i += blockDim.x) {
    const int m = i >> 10, rr = (i >> 5) & 31, cc = i & 31;  // W[m][rr][cc]
    out[m * 1056 + cc * 33 + rr] = blk[i];
  }
  for (int i = 5280 + threadIdx.x; i < B2_WIMG; i += blockDim.x) out[i] = 0.f;
}

// ---------------------------------------------------------------------------
// Every input tile goes global -> LDS by global_load_lds (tile_dma) instead
// of through staging registers (the round-1 register-staged form measured 46.6
// against 44.2 us at B*T = 128000 and is gone).  Without the 4..6 x 16 staging registers two
// waves per SIMD fit without spills (one wave's gate math / LDS traffic runs
// under the other's MFMAs), the rows-t tiles are in flight during the rows
// t+d math and the x tiles during the rows-t math.
// ---------------------------------------------------------------------------
#define WN_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <bool HAS_DXIN>
__global__ __launch_bounds__(B2_WAVES * 64) void layer_bwd2d_kernel(
    const float* __restrict__ x, const float* __restrict__ z,
    const float* __restrict__ sg, const float* __restrict__ dZ,
    const float* __restrict__ dxin, float* __restrict__ dx_out,
    const float* __restrict__ wimg, float* __restrict__ slabs,
    float* __restrict__ tile_colsum, int B, int T, int d) {
#ifdef B2_STAMPS
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(tile_colsum);
  tile_colsum = nullptr;
  int tix = 0;
#define STAMP(i)                                                          \
  if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7)) \
    dbg[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 7)) * 64 + (i)] = __builtin_amdgcn_s_memtime()
  STAMP(0);
  if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7))
    dbg[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 7)) * 64 + 40] = __builtin_amdgcn_s_memrealtime();
#else
#define STAMP(i)
#endif
  constexpr int LDT = 33, MT = 32 * LDT;
  __shared__ __attribute__((aligned(1024))) float wl[B2_WIMG];
  __shared__ __attribute__((aligned(1024))) float tiles[B2_WAVES * 4 * 1024];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the layer's transposed weights (wn_layer_bwd2_pack) go straight into LDS
  // by LDS-DMA, 21 pieces of 1 KiB, issued before the first tile's loads and
  // waited for together with them
  for (int p = wave; p < B2_WIMG / 256; p += B2_WAVES)
    __builtin_amdgcn_global_load_lds((wn_gptr_t)(wimg + p * 256 + lane * 4),
                                     (wn_lptr_t)(wl + p * 256), 16, 0, 0);
  const int j = lane & 31, h = lane >> 5;
  float* t0 = tiles + wave * 4096;
  float* t1 = t0 + 1024;
  float* t2 = t1 + 1024;
  float* t3 = t2 + 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  f32x16 cf0 = frag_zero(), cf1 = frag_zero(), cg0 = frag_zero(),
         cg1 = frag_zero(), cd = frag_zero();
  float sf = 0.f, sgs = 0.f, sd = 0.f;
  // The rows t+d of the NEXT tile are fetched into registers during the
  // weight-gradient products (the only phase with registers to spare and no
  // free LDS tile), everything else by LDS-DMA one phase ahead of its use.
  const int tstep = gridDim.x * B2_WAVES;
  RowRegs a0, a1, a2, a3;
  // (always assigns a0..a3 -- zeros when there is nothing to load -- so that
  // their live range ends at the rows_to_lds of the next iteration)
  auto load_shifted = [&](int tl) {
    const bool any = tl < ntiles;
    const int b = any ? tl / tiles_per_clip : 0;
    const int tt0 = any ? (tl - b * tiles_per_clip) * 32 : 0;
    const int hif = any ? min(min(32, T - tt0), T - d - tt0) : 0;
    const size_t offd = ((size_t)b * T + tt0 + d) * WN_CH;
    if (HAS_DXIN) a0 = rows_load(dxin + offd, lane, 0, hif);
    a1 = rows_load(dZ + offd, lane, 0, hif);
    a2 = rows_load(z + offd, lane, 0, hif);
    a3 = rows_load(sg + offd, lane, 0, hif);
  };
  // few tiles (small batches): one tile per wave, spread over all CUs
  const int tile0 = ntiles < tstep ? wave * (int)gridDim.x + (int)blockIdx.x
                                   : (int)blockIdx.x * B2_WAVES + wave;
  load_shifted(tile0);
  STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();               // every wave's weight pieces have landed
  STAMP(2);
  for (int tile = tile0; tile < ntiles; tile += tstep) {
    STAMP(3 + 10 * tix + 0);
    int woff = j + 4 * h * LDT;  // opaque: no hoisting of the weight reads
    asm volatile("" : "+v"(woff));
    const float* wlane = wl + woff;
    const int b = tile / tiles_per_clip;
    const int tt0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - tt0);
    const int hi_f = min(hi, T - d - tt0);  // rows whose t+d tap exists
    const int lo_p = max(0, d - tt0);       // rows whose t-d tap exists
    const size_t off0 = ((size_t)b * T + tt0) * WN_CH;
    f32x16 dx = frag_zero();
    f32x16 dz, di, zz, ss;
    if (hi_f > 0) {                         // rows t+d -> LDS -> fragments
      if (HAS_DXIN) rows_to_lds(t0, lane, a0);
      rows_to_lds(t1, lane, a1);
      rows_to_lds(t2, lane, a2);
      rows_to_lds(t3, lane, a3);
      __builtin_amdgcn_wave_barrier();
      dz = frag_from_lds(t1, j, h);
      if (HAS_DXIN) di = frag_from_lds(t0, j, h);
      zz = frag_from_lds(t2, j, h);
      ss = frag_from_lds(t3, j, h);
      WN_WAIT_LGKM0();                      // tiles free again
    }
    STAMP(3 + 10 * tix + 1);
    // rows t: in flight during the rows t+d math
    if (HAS_DXIN) tile_dma(t0, dxin + off0, lane, 0, hi);
    tile_dma(t1, dZ + off0, lane, 0, hi);
    tile_dma(t2, z + off0, lane, 0, hi);
    tile_dma(t3, sg + off0, lane, 0, hi);
    STAMP(3 + 10 * tix + 2);
    if (hi_f > 0) {
      if (HAS_DXIN) mma32<LDT>(dz, di, wlane + 4 * MT);  // dx_{l+1}[t+d] * Wd^T
      f32x16 df, dg;
      gate_grad(dz, zz, ss, df, dg);
      mma32<LDT>(dx, df, wlane + 0 * MT);        // da_f[t+d] * Wf[0]^T
      mma32<LDT>(dx, dg, wlane + 2 * MT);        // da_g[t+d] * Wg[0]^T
    }
    STAMP(3 + 10 * tix + 3);
    WN_WAIT_VM0();
    STAMP(3 + 10 * tix + 4);
    if (HAS_DXIN) {                              // dWd += z^T dx_{l+1}
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int row = 2 * s + h;
        const float az = tile_elem(t2, row, j), bd = tile_elem(t0, row, j);
        cd = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bd, cd, 0, 0, 0);
        sd += bd;
      }
    }
    dz = frag_from_lds(t1, j, h);
    if (HAS_DXIN) di = frag_from_lds(t0, j, h);
    zz = frag_from_lds(t2, j, h);
    ss = frag_from_lds(t3, j, h);
    WN_WAIT_LGKM0();
    // the x tiles: in flight during the rows-t math
    tile_dma(t2, x + off0, lane, 0, hi);
    tile_dma(t0, x + off0 - (size_t)d * WN_CH, lane, lo_p, hi);
    STAMP(3 + 10 * tix + 5);
    {
      if (HAS_DXIN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[r] += di[r];
        mma32<LDT>(dz, di, wlane + 4 * MT);      // dx_{l+1}[t] * Wd^T
      }
      f32x16 df, dg;
      gate_grad(dz, zz, ss, df, dg);
      mma32<LDT>(dx, df, wlane + 1 * MT);        // da_f[t] * Wf[1]^T
      mma32<LDT>(dx, dg, wlane + 3 * MT);        // da_g[t] * Wg[1]^T
      frag_to_lds(t1, j, h, dx);
      __builtin_amdgcn_wave_barrier();
      rows_store(dx_out + off0, lane, hi, rows_from_lds(t1, lane));
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(t1, j, h, df);                 // t1 / t3 now hold da[t]
      frag_to_lds(t3, j, h, dg);
    }
    STAMP(3 + 10 * tix + 6);
    WN_WAIT_VM0();
    STAMP(3 + 10 * tix + 7);
    load_shifted(tile + tstep);
    __builtin_amdgcn_wave_barrier();
    float tsf = 0.f, tsg = 0.f;          // this tile's column sums of da[t]
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {       // dW[1] += x[t]^T da, dW[0] += x[t-d]^T da
      const int row = 2 * s + h;
      const float axc = tile_elem(t2, row, j), axp = tile_elem(t0, row, j);
      const float bf = tile_elem(t1, row, j), bg = tile_elem(t3, row, j);
      cf1 = __builtin_amdgcn_mfma_f32_32x32x2f32(axc, bf, cf1, 0, 0, 0);
      cg1 = __builtin_amdgcn_mfma_f32_32x32x2f32(axc, bg, cg1, 0, 0, 0);
      cf0 = __builtin_amdgcn_mfma_f32_32x32x2f32(axp, bf, cf0, 0, 0, 0);
      cg0 = __builtin_amdgcn_mfma_f32_32x32x2f32(axp, bg, cg0, 0, 0, 0);
      tsf += bf;
      tsg += bg;
    }
    sf += tsf;
    sgs += tsg;
    STAMP(3 + 10 * tix + 8);
    if (tile_colsum) {
      const float a = tsf + __shfl_xor(tsf, 32), b2 = tsg + __shfl_xor(tsg, 32);
      if (h == 0) {
        tile_colsum[(size_t)tile * 64 + j] = a;
        tile_colsum[(size_t)tile * 64 + 32 + j] = b2;
      }
    }
    WN_WAIT_LGKM0();                     // tiles free for the next tile's DMA
    __builtin_amdgcn_wave_barrier();
    STAMP(3 + 10 * tix + 9);
#ifdef B2_STAMPS
    ++tix;
#endif
  }
  // ---- weight-gradient slab of this workgroup.  Fixed-order tree over the
  // eight waves through LDS: waves 0-3 store, waves 4-7 add into the region of
  // wave w-4 (the four regions in parallel), then every thread sums the four
  // regions ((r0+r1)+(r2+r3)) on the way out.
  sf += __shfl_xor(sf, 32);
  sgs += __shfl_xor(sgs, 32);
  sd += __shfl_xor(sd, 32);
  STAMP(30);
  __syncthreads();
  STAMP(31);
  constexpr int RS = 5248;               // floats per region (>= block, 128 B multiple)
  static_assert(4 * RS <= B2_WAVES * 4 * 1024, "reduction regions exceed the tile area");
  float* red = tiles + (wave & 3) * RS;
  for (int ph = 0; ph < 2; ++ph) {
    if ((wave >> 2) == ph) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 8 * (r >> 2) + 4 * h + (r & 3);
        const int e = m * 32 + j;
        if (ph == 0) {
          red[0 * 1024 + e] = cf0[r];
          red[1 * 1024 + e] = cf1[r];
          red[2 * 1024 + e] = cg0[r];
          red[3 * 1024 + e] = cg1[r];
          red[4 * 1024 + e] = cd[r];
        } else {
          red[0 * 1024 + e] += cf0[r];
          red[1 * 1024 + e] += cf1[r];
          red[2 * 1024 + e] += cg0[r];
          red[3 * 1024 + e] += cg1[r];
          red[4 * 1024 + e] += cd[r];
        }
      }
      if (h == 0) {
        if (ph == 0) {
          red[LAYER_W_FLOATS + j] = sf;
          red[LAYER_W_FLOATS + 32 + j] = sgs;
          red[LAYER_W_FLOATS + 64 + j] = sd;
        } else {
          red[LAYER_W_FLOATS + j] += sf;
          red[LAYER_W_FLOATS + 32 + j] += sgs;
          red[LAYER_W_FLOATS + 64 + j] += sd;
        }
      }
    }
    __syncthreads();
  }
  STAMP(32);
  float* out = slabs + (size_t)blockIdx.x * LAYER_BLOCK_FLOATS;
  for (int e = tid; e < LAYER_BLOCK_FLOATS; e += B2_WAVES * 64)
    out[e] = (tiles[e] + tiles[RS + e]) + (tiles[2 * RS + e] + tiles[3 * RS + e]);
  STAMP(33);
#ifdef B2_STAMPS
  if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7))
    dbg[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 7)) * 64 + 41] = __builtin_amdgcn_s_memrealtime();
#endif
#undef STAMP
}

// ---------------------------------------------------------------------------
// Generic tap count (filter_width K >= 2, runtime): the off-default variant
// of the three layer kernels.  Layer block = Wf[K][32][32] Wg[K][32][32]
// Wd[32][32] bf bg bd.  Tap k reads x[t - s_k], s_k = (K-1-k + (K-1)/2) * d
// (TF 'SAME' centring inside causal_conv for K > 2, ops.py:46-62), so for
// K > 2 no tap has shift 0 and the residual x[t] is a separate load.
// Same tile machinery as the K = 2 kernels; correctness-first (taps are
// processed one after another, the weight-gradient kernel re-reads da per
// tap).
// ---------------------------------------------------------------------------
#define GEN_WG 512
#define GEN_WAVES (GEN_WG / 64)

__device__ __forceinline__ int tap_shift(int K, int k, int d) {
  return (K - 1 - k + (K - 1) / 2) * d;
}

template <bool HAS_DENSE, bool SAVE_TS>
__global__ __launch_bounds__(GEN_WG) void layer_fwd_gen_kernel(
    const float* __restrict__ x, float* __restrict__ xo, float* __restrict__ z,
    float* __restrict__ th, float* __restrict__ sg,
    const float* __restrict__ wblock, const float* __restrict__ bias_fg,
    int bias_clip_stride, int B, int T, int d, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int WF = (2 * K + 1) * 1024;
  float* wl = smem;                 // weights + bd[32]
  float* tiles = smem + WF + 32;
  const int tid = threadIdx.x;
  for (int i = tid; i < WF / 4; i += GEN_WG)
    reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(wblock)[i];
  if (tid < 32) wl[WF + tid] = wblock[WF + 64 + tid];   // bd
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  float* ta = tiles + wave * 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  for (int tile = blockIdx.x * GEN_WAVES + wave; tile < ntiles;
       tile += gridDim.x * GEN_WAVES) {
    int woff = j + 4 * h * 32;
    asm volatile("" : "+v"(woff));
    const float* wlane = wl + woff;
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    f32x16 af, ag;
    if (bias_fg) {
      const float* bp = bias_fg + (size_t)b * bias_clip_stride;
      af = frag_bcast(bp, h);
      ag = frag_bcast(bp + 32, h);
    } else {
      af = frag_zero();
      ag = frag_zero();
    }
    for (int k = 0; k < K; ++k) {
      const int sh = tap_shift(K, k, d);
      const RowRegs r = rows_load(x + off0 - (size_t)sh * WN_CH, lane,
                                  max(0, sh - t0), hi);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(ta, lane, r);
      __builtin_amdgcn_wave_barrier();
      const f32x16 xk = frag_from_lds(ta, j, h);
      mma32<32>(af, xk, wlane + k * 1024);
      mma32<32>(ag, xk, wlane + (K + k) * 1024);
    }
    f32x16 zz;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      af[r] = wn_tanh(af[r]);
      ag[r] = wn_sigmoid(ag[r]);
      zz[r] = af[r] * ag[r];
    }
    __builtin_amdgcn_wave_barrier();
    frag_to_lds(ta, j, h, zz);
    __builtin_amdgcn_wave_barrier();
    rows_store(z + off0, lane, hi, rows_from_lds(ta, lane));
    if (SAVE_TS) {
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, af);
      __builtin_amdgcn_wave_barrier();
      rows_store(th + off0, lane, hi, rows_from_lds(ta, lane));
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, ag);
      __builtin_amdgcn_wave_barrier();
      rows_store(sg + off0, lane, hi, rows_from_lds(ta, lane));
    }
    if (HAS_DENSE) {
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(ta, lane, rows_load(x + off0, lane, 0, hi));
      __builtin_amdgcn_wave_barrier();
      f32x16 acc = frag_from_lds(ta, j, h);
      const f32x16 bd = frag_bcast(wl + WF + (woff - j - 128 * h), h);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += bd[r];
      mma32<32>(acc, zz, wlane + 2 * K * 1024);
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, acc);
      __builtin_amdgcn_wave_barrier();
      rows_store(xo + off0, lane, hi, rows_from_lds(ta, lane));
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool DO_B, bool DO_A, bool HAS_DXIN>
__global__ __launch_bounds__(GEN_WG) void layer_bwd_gen_kernel(
    const float* __restrict__ daf_cur, const float* __restrict__ dag_cur,
    const float* __restrict__ dxin, float* __restrict__ dx_out,
    const float* __restrict__ wblock_b, const float* __restrict__ dZ,
    const float* __restrict__ th, const float* __restrict__ sg,
    const float* __restrict__ wblock_a, float* __restrict__ daf_next,
    float* __restrict__ dag_next, int B, int T, int d, int K, long blk_stride) {
  constexpr int LDT = 33, MT = 32 * LDT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // blockIdx.y (phase A only): the dilation-channel block of a channel-block
  // layer, its planes blk_stride floats apart
  if (DO_A) {
    const size_t o = (size_t)blockIdx.y * blk_stride;
    dZ += o; th += o; sg += o; daf_next += o; dag_next += o;
  }
  float* wl = smem;                         // (2K+1) transposed matrices
  float* tiles = smem + ((2 * K + 1) * MT + 3) / 4 * 4;
  const int tid = threadIdx.x;
  if (DO_B) {
    for (int i = tid; i < 2 * K * 1024; i += GEN_WG) {
      const int m = i >> 10, rr = (i >> 5) & 31, cc = i & 31;
      wl[m * MT + cc * LDT + rr] = wblock_b[i];
    }
  }
  if (DO_A) {
    for (int i = tid; i < 1024; i += GEN_WG) {
      const int rr = i >> 5, cc = i & 31;
      wl[2 * K * MT + cc * LDT + rr] = wblock_a[2 * K * 1024 + i];
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  float* ta = tiles + wave * 2048;
  float* tb = ta + 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  for (int tile = blockIdx.x * GEN_WAVES + wave; tile < ntiles;
       tile += gridDim.x * GEN_WAVES) {
    int woff = j + 4 * h * LDT;
    asm volatile("" : "+v"(woff));
    const float* wlane = wl + woff;
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    f32x16 dx;
    if (HAS_DXIN) {
      rows_to_lds(ta, lane, rows_load(dxin + off0, lane, 0, hi));
      __builtin_amdgcn_wave_barrier();
      dx = frag_from_lds(ta, j, h);
      __builtin_amdgcn_wave_barrier();
    } else {
      dx = frag_zero();
    }
    if (DO_B) {
      for (int k = 0; k < K; ++k) {
        const int sh = tap_shift(K, k, d);
        const int hi_f = min(hi, T - sh - t0);
        const RowRegs rf = rows_load(daf_cur + off0 + (size_t)sh * WN_CH, lane, 0, hi_f);
        const RowRegs rg = rows_load(dag_cur + off0 + (size_t)sh * WN_CH, lane, 0, hi_f);
        __builtin_amdgcn_wave_barrier();
        rows_to_lds(ta, lane, rf);
        rows_to_lds(tb, lane, rg);
        __builtin_amdgcn_wave_barrier();
        const f32x16 fk = frag_from_lds(ta, j, h);
        const f32x16 gk = frag_from_lds(tb, j, h);
        mma32<LDT>(dx, fk, wlane + k * MT);
        mma32<LDT>(dx, gk, wlane + (K + k) * MT);
      }
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, dx);
      __builtin_amdgcn_wave_barrier();
      rows_store(dx_out + off0, lane, hi, rows_from_lds(ta, lane));
      __builtin_amdgcn_wave_barrier();
    }
    if (DO_A) {
      const RowRegs rz = rows_load(dZ + off0, lane, 0, hi);
      const RowRegs rt = rows_load(th + off0, lane, 0, hi);
      const RowRegs rs = rows_load(sg + off0, lane, 0, hi);
      rows_to_lds(ta, lane, rz);
      rows_to_lds(tb, lane, rt);
      __builtin_amdgcn_wave_barrier();
      f32x16 dz = frag_from_lds(ta, j, h);
      const f32x16 tt = frag_from_lds(tb, j, h);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(ta, lane, rs);
      __builtin_amdgcn_wave_barrier();
      const f32x16 ss = frag_from_lds(ta, j, h);
      if (DO_B || HAS_DXIN) mma32<LDT>(dz, dx, wlane + 2 * K * MT);
      f32x16 df, dg;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float zs = dz[r] * ss[r];
        df[r] = zs * (1.f - tt[r] * tt[r]);
        dg[r] = zs * tt[r] * (1.f - ss[r]);
      }
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, df);
      frag_to_lds(tb, j, h, dg);
      __builtin_amdgcn_wave_barrier();
      rows_store(daf_next + off0, lane, hi, rows_from_lds(ta, lane));
      rows_store(dag_next + off0, lane, hi, rows_from_lds(tb, lane));
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------
// More than 32 residual / dilation channels (the reference constructor has no
// limit, model.py:46-60): channels are cut into 32-wide BLOCKS, every block of
// an activation is its own [rows][32] plane (so the skip / dZ / dWs GEMMs see
// L * blocks planes and need no change), and a (tap k, input block i) pair is
// one "virtual tap" of the generic kernels above: output block jb of
//     a_f = sum_{k,i} x_i[t - s_k] * Wf[k][32 i .. 32 i + 31][32 jb .. 32 jb + 31]
// One launch per OUTPUT block; the weights stay in the reference's
// [K][Cin][Cout] layout (row stride ldw = padded channel count) and are read
// block-wise.  The 1x1 dense conv and dz = dZ + dx' Wd^T are plane-mode GEMMs
// (wn_gemm_nn) on the host side.  Correctness-first, like the generic-tap
// kernels (off-default configurations).
// ---------------------------------------------------------------------------
template <bool SAVE_TS>
__global__ __launch_bounds__(GEN_WG) void layer_fwd_blk_kernel(
    const float* __restrict__ x, long in_plane_stride, int in_blocks,
    float* __restrict__ z, float* __restrict__ th, float* __restrict__ sg,
    const float* __restrict__ wf, const float* __restrict__ wg, int ldw,
    const float* __restrict__ bias_f, const float* __restrict__ bias_g,
    int bias_clip_stride, int B, int T, int d, int K, int tap_rows,
    const float* __restrict__ pre_in, float* __restrict__ pre_out,
    long pre_plane_stride, int k0, int Ktot, long out_plane_stride) {
  // (K taps k0 .. k0 + K - 1 of a filter of Ktot taps: filter widths above 8
  // run in groups of taps like wide layers run in chunks of blocks)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // blockIdx.y: the 32-wide OUTPUT block (its z / tanh / sigmoid planes
  // out_plane_stride floats apart, its weight columns and bias entries 32
  // further, its partial pre-activation planes 2 * pre_plane_stride further)
  {
    const int jb = blockIdx.y;
    if (z) z += (size_t)jb * out_plane_stride;
    if (th) th += (size_t)jb * out_plane_stride;
    if (sg) sg += (size_t)jb * out_plane_stride;
    wf += jb * 32;
    wg += jb * 32;
    if (bias_f) bias_f += jb * 32;
    if (bias_g) bias_g += jb * 32;
    if (pre_in) pre_in += (size_t)jb * 2 * pre_plane_stride;
    if (pre_out) pre_out += (size_t)jb * 2 * pre_plane_stride;
  }
  const int NV = K * in_blocks;               // virtual taps (k, i)
  float* wl = smem;                           // [2 NV][32][32]: filter, gate
  float* tiles = smem + 2 * NV * 1024;
  const int tid = threadIdx.x;
  // `in_blocks` may be a CHUNK of the layer's input blocks (more than 8
  // virtual taps do not fit the LDS): tap k's rows start at k * tap_rows, the
  // caller offsets x / wf / wg to the chunk's first block, the partial
  // pre-activations travel through pre_out -> pre_in (planes af | ag) and the
  // last chunk applies the gate
  for (int i = tid; i < NV * 1024; i += GEN_WG) {
    const int v = i >> 10, r = (i >> 5) & 31, c = i & 31;
    const int k = v / in_blocks, ib = v - k * in_blocks;
    const long src = (long)(k * tap_rows + ib * 32 + r) * ldw + c;
    wl[i] = wf[src];
    wl[NV * 1024 + i] = wg[src];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  float* ta = tiles + wave * 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  for (int tile = blockIdx.x * GEN_WAVES + wave; tile < ntiles;
       tile += gridDim.x * GEN_WAVES) {
    int woff = j + 4 * h * 32;
    asm volatile("" : "+v"(woff));
    const float* wlane = wl + woff;
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    f32x16 af, ag;
    if (pre_in) {                              // sums of the chunks before this one
      rows_to_lds(ta, lane, rows_load(pre_in + off0, lane, 0, hi));
      __builtin_amdgcn_wave_barrier();
      af = frag_from_lds(ta, j, h);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(ta, lane, rows_load(pre_in + pre_plane_stride + off0, lane, 0, hi));
      __builtin_amdgcn_wave_barrier();
      ag = frag_from_lds(ta, j, h);
      __builtin_amdgcn_wave_barrier();
    } else {
      af = bias_f ? frag_bcast(bias_f + (size_t)b * bias_clip_stride, h) : frag_zero();
      ag = bias_g ? frag_bcast(bias_g + (size_t)b * bias_clip_stride, h) : frag_zero();
    }
    for (int v = 0; v < NV; ++v) {
      const int k = v / in_blocks, i = v - k * in_blocks;
      const int sh = tap_shift(Ktot, k0 + k, d);
      const RowRegs r = rows_load(x + (size_t)i * in_plane_stride + off0 - (size_t)sh * WN_CH,
                                  lane, max(0, sh - t0), hi);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(ta, lane, r);
      __builtin_amdgcn_wave_barrier();
      const f32x16 xk = frag_from_lds(ta, j, h);
      mma32<32>(af, xk, wlane + v * 1024);
      mma32<32>(ag, xk, wlane + (NV + v) * 1024);
    }
    if (pre_out) {                             // not the last chunk: raw sums
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, af);
      __builtin_amdgcn_wave_barrier();
      rows_store(pre_out + off0, lane, hi, rows_from_lds(ta, lane));
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, ag);
      __builtin_amdgcn_wave_barrier();
      rows_store(pre_out + pre_plane_stride + off0, lane, hi, rows_from_lds(ta, lane));
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    f32x16 zz;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      af[r] = wn_tanh(af[r]);
      ag[r] = wn_sigmoid(ag[r]);
      zz[r] = af[r] * ag[r];
    }
    __builtin_amdgcn_wave_barrier();
    frag_to_lds(ta, j, h, zz);
    __builtin_amdgcn_wave_barrier();
    rows_store(z + off0, lane, hi, rows_from_lds(ta, lane));
    if (SAVE_TS) {
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, af);
      __builtin_amdgcn_wave_barrier();
      rows_store(th + off0, lane, hi, rows_from_lds(ta, lane));
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, ag);
      __builtin_amdgcn_wave_barrier();
      rows_store(sg + off0, lane, hi, rows_from_lds(ta, lane));
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// data gradient of output (residual-channel) block rb:
//   dx[t] = dxin[t] + sum_{k,jb} da_f[jb][t + s_k] Wf[k][rb rows][jb cols]^T + (gate)
template <bool HAS_DXIN>
__global__ __launch_bounds__(GEN_WG) void layer_bwd_blk_kernel(
    const float* __restrict__ daf, const float* __restrict__ dag,
    long da_plane_stride, int da_blocks, const float* __restrict__ dxin,
    float* __restrict__ dx_out, const float* __restrict__ wf,
    const float* __restrict__ wg, int ldw, long tap_stride, int B, int T, int d,
    int K, int k0, int Ktot, long dx_plane_stride) {
  constexpr int LDT = 33, MT = 32 * LDT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // blockIdx.y: the residual-channel block (its dx planes dx_plane_stride
  // floats apart, its weight rows 32 further)
  {
    const int rb = blockIdx.y;
    if (HAS_DXIN) dxin += (size_t)rb * dx_plane_stride;
    dx_out += (size_t)rb * dx_plane_stride;
    wf += (size_t)rb * 32 * ldw;
    wg += (size_t)rb * 32 * ldw;
  }
  const int NV = K * da_blocks;               // virtual taps (k, jb)
  float* wl = smem;                           // [2 NV] transposed blocks
  float* tiles = smem + (2 * NV * MT + 3) / 4 * 4;
  const int tid = threadIdx.x;
  for (int i = tid; i < NV * 1024; i += GEN_WG) {
    const int v = i >> 10, rr = (i >> 5) & 31, cc = i & 31;
    const int k = v / da_blocks, jb = v - k * da_blocks;
    const long src = (long)k * tap_stride + (long)rr * ldw + jb * 32 + cc;
    wl[v * MT + cc * LDT + rr] = wf[src];
    wl[(NV + v) * MT + cc * LDT + rr] = wg[src];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  float* ta = tiles + wave * 2048;
  float* tb = ta + 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  for (int tile = blockIdx.x * GEN_WAVES + wave; tile < ntiles;
       tile += gridDim.x * GEN_WAVES) {
    int woff = j + 4 * h * LDT;
    asm volatile("" : "+v"(woff));
    const float* wlane = wl + woff;
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * 32;
    const int hi = min(32, T - t0);
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;
    f32x16 dx;
    if (HAS_DXIN) {
      rows_to_lds(ta, lane, rows_load(dxin + off0, lane, 0, hi));
      __builtin_amdgcn_wave_barrier();
      dx = frag_from_lds(ta, j, h);
      __builtin_amdgcn_wave_barrier();
    } else {
      dx = frag_zero();
    }
    for (int v = 0; v < NV; ++v) {
      const int k = v / da_blocks, jb = v - k * da_blocks;
      const int sh = tap_shift(Ktot, k0 + k, d);
      const int hi_f = min(hi, T - sh - t0);
      const size_t o = (size_t)jb * da_plane_stride + off0 + (size_t)sh * WN_CH;
      const RowRegs rf = rows_load(daf + o, lane, 0, hi_f);
      const RowRegs rg = rows_load(dag + o, lane, 0, hi_f);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(ta, lane, rf);
      rows_to_lds(tb, lane, rg);
      __builtin_amdgcn_wave_barrier();
      const f32x16 fk = frag_from_lds(ta, j, h);
      const f32x16 gk = frag_from_lds(tb, j, h);
      mma32<LDT>(dx, fk, wlane + v * MT);
      mma32<LDT>(dx, gk, wlane + (NV + v) * MT);
    }
    __builtin_amdgcn_wave_barrier();
    frag_to_lds(ta, j, h, dx);
    __builtin_amdgcn_wave_barrier();
    rows_store(dx_out + off0, lane, hi, rows_from_lds(ta, lane));
    __builtin_amdgcn_wave_barrier();
  }
}

// weight gradients, one tap per pass; slab layout == layer block layout
template <bool HAS_DENSE>
__global__ __launch_bounds__(256) void layer_wgrad_gen_kernel(
    const float* __restrict__ x, const float* __restrict__ daf,
    const float* __restrict__ dag, const float* __restrict__ z,
    const float* __restrict__ dxin, float* __restrict__ slabs, int B, int T,
    int d, int K, int k0, int Ktot, int CB, long plane_stride) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 3 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  float* t_x = lds + wave * 3 * 1024;
  float* t_f = t_x + 1024;
  float* t_g = t_f + 1024;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * B;
  const int nwaves = gridDim.x * 4;
  const int WF = (2 * K + 1) * 1024;
  // blockIdx.y: the (input block a, output block b) pair of a channel-block
  // layer (planes plane_stride floats apart; slabs [pair][slab][WF + 96])
  {
    const int pair = blockIdx.y, pa = pair / CB, pb = pair - pa * CB;
    x += (size_t)pa * plane_stride;
    daf += (size_t)pb * plane_stride;
    dag += (size_t)pb * plane_stride;
    if (HAS_DENSE) {
      z += (size_t)pa * plane_stride;
      dxin += (size_t)pb * plane_stride;
    }
  }
  float* out = slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (WF + 96);
  for (int k = 0; k < K; ++k) {
    const int sh = tap_shift(Ktot, k0 + k, d);
    f32x16 cf = frag_zero(), cg = frag_zero(), cd = frag_zero();
    float sf = 0.f, sgs = 0.f, sd = 0.f;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += nwaves) {
      const int b = tile / tiles_per_clip;
      const int t0 = (tile - b * tiles_per_clip) * 32;
      const int hi = min(32, T - t0);
      const size_t off0 = ((size_t)b * T + t0) * WN_CH;
      const RowRegs rx = rows_load(x + off0 - (size_t)sh * WN_CH, lane,
                                   max(0, sh - t0), hi);
      const RowRegs rf = rows_load(daf + off0, lane, 0, hi);
      const RowRegs rg = rows_load(dag + off0, lane, 0, hi);
      __builtin_amdgcn_wave_barrier();
      rows_to_lds(t_x, lane, rx);
      rows_to_lds(t_f, lane, rf);
      rows_to_lds(t_g, lane, rg);
      __builtin_amdgcn_wave_barrier();
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int row = 2 * s + h;
        const float ax = tile_elem(t_x, row, i);
        const float bf = tile_elem(t_f, row, i), bg = tile_elem(t_g, row, i);
        cf = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bf, cf, 0, 0, 0);
        cg = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bg, cg, 0, 0, 0);
        sf += bf;
        sgs += bg;
      }
      if (HAS_DENSE && k == 0) {
        const RowRegs rz = rows_load(z + off0, lane, 0, hi);
        const RowRegs rd = rows_load(dxin + off0, lane, 0, hi);
        __builtin_amdgcn_wave_barrier();
        rows_to_lds(t_x, lane, rz);
        rows_to_lds(t_f, lane, rd);
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
          const int row = 2 * s + h;
          const float az = tile_elem(t_x, row, i), bd = tile_elem(t_f, row, i);
          cd = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bd, cd, 0, 0, 0);
          sd += bd;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    sf += __shfl_xor(sf, 32);
    sgs += __shfl_xor(sgs, 32);
    sd += __shfl_xor(sd, 32);
    __syncthreads();
    float* red = lds;   // [0] cf, [1] cg, [2] cd, then 96 sums
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int e = (8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i;
          if (w == 0) {
            red[e] = cf[r]; red[1024 + e] = cg[r]; red[2048 + e] = cd[r];
          } else {
            red[e] += cf[r]; red[1024 + e] += cg[r]; red[2048 + e] += cd[r];
          }
        }
        if (h == 0) {
          if (w == 0) {
            red[3072 + i] = sf; red[3104 + i] = sgs; red[3136 + i] = sd;
          } else {
            red[3072 + i] += sf; red[3104 + i] += sgs; red[3136 + i] += sd;
          }
        }
      }
      __syncthreads();
    }
    for (int e = tid; e < 1024; e += 256) {
      out[k * 1024 + e] = red[e];
      out[(K + k) * 1024 + e] = red[1024 + e];
      if (k == 0) out[2 * K * 1024 + e] = red[2048 + e];
    }
    if (k == 0 && tid < 96) out[WF + tid] = red[3072 + tid];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
// one persistent workgroup per CU (fewer when there is less work)
// (`spread`: kernels with the sparse tile mapping get one workgroup per CU as
// soon as there are that many tiles, each running ceil(ntiles / cus) waves)
static int layer_grid(int B, int T, int waves = LAYER_WAVES, bool spread = false) {
  const int cus = wn_device_cus();
  const long ntiles = (long)B * ((T + 31) / 32);
  long g = (ntiles + waves - 1) / waves;
  if (spread && ntiles < (long)cus * waves) g = ntiles;
  if (g > cus) g = cus;
  return (int)(g < 1 ? 1 : g);
}

// ---------------------------------------------------------------------------
// The 1x1 convs of the channel-block models (33 - 128 channels) as a streaming
// kernel: out planes = addend planes + in planes * W (+ bias), W [C][C] in the
// reference's [Cin][Cout] order, C = 32 * CB.  Forward: x_{l+1} = x_l + z_l Wd
// (+ bd) (model.py:294-300, 330); backward: dz = dZ + dx_{l+1} Wd^T.  As
// plane-mode wn_gemm_nn launches these are K = N = C <= 128 GEMMs: four K chunks
// between a tile's first load and its epilogue, half of every 128-column tile
// masked at C = 64 -- 54 us a launch at 8 x 16000 x 64 channels against an HBM
// floor of 20, 98 of them per step.  Here a wave takes a 32-row tile: the CB
// input planes' rows in one batch of coalesced loads (8 full rows per
// instruction), fragments through the wave's LDS tile, CB x CB x 16 MFMAs with
// the weights in LDS in their own layout, the addend rows loaded meanwhile.
// ---------------------------------------------------------------------------
// GATE: the result is dz of the backward; instead of storing it the gate
// gradients da_f = dz s (1 - t^2), da_g = dz s t (1 - s) (t, s: the saved tanh /
// sigmoid planes) go out -- elementwise in the row layout the result is stored
// in anyway, the same operations in the same order as the separate pass
// (layer_bwd_gen_kernel, gate gradients only), which read dz back: bitwise the
// same planes, one launch and a plane round trip per block less.
template <int CB, bool GATE>
__global__ __launch_bounds__(256) void dense_planes_kernel(
    const float* __restrict__ in, long in_pstride, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ addend, long add_pstride,
    float* __restrict__ out, long out_pstride, long rows,
    const float* __restrict__ th, const float* __restrict__ sg, long ts_pstride,
    float* __restrict__ dag) {
  constexpr int C = 32 * CB;
  __shared__ __attribute__((aligned(16))) float wl[C * C];
  __shared__ __attribute__((aligned(16))) float tiles[4 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  for (int p = wave; p < C * C / 256; p += 4)
    __builtin_amdgcn_global_load_lds((wn_gptr_t)(W + p * 256 + lane * 4),
                                     (wn_lptr_t)(wl + p * 256), 16, 0, 0);
  float* ta = tiles + wave * 1024;
  const long ntiles = (rows + 31) / 32;
  const long tile = (long)blockIdx.x * 4 + wave;
  const bool any = tile < ntiles;
  const long r0 = any ? tile * 32 : 0;
  const int hi = any ? (int)(rows - r0 < 32 ? rows - r0 : 32) : 0;
  RowRegs rin[CB], radd[CB];
#pragma unroll
  for (int b = 0; b < CB; ++b) rin[b] = rows_load(in + b * in_pstride + r0 * 32, lane, 0, hi);
#pragma unroll
  for (int b = 0; b < CB; ++b)
    if (addend) radd[b] = rows_load(addend + b * add_pstride + r0 * 32, lane, 0, hi);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                 // the weights are in LDS
  f32x16 fr[CB];
#pragma unroll
  for (int b = 0; b < CB; ++b) {
    __builtin_amdgcn_wave_barrier();
    rows_to_lds(ta, lane, rin[b]);
    __builtin_amdgcn_wave_barrier();
    fr[b] = frag_from_lds(ta, j, h);
  }
#pragma unroll
  for (int ob = 0; ob < CB; ++ob) {
    f32x16 acc = frag_zero();
#pragma unroll
    for (int ib = 0; ib < CB; ++ib)
      mma32<C>(acc, fr[ib], wl + (ib * 32 + 4 * h) * C + ob * 32 + j);
    __builtin_amdgcn_wave_barrier();
    frag_to_lds(ta, j, h, acc);
    __builtin_amdgcn_wave_barrier();
    RowRegs ro = rows_from_lds(ta, lane);
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + ob * 32 + (lane & 7) * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ro.v[c] += b4;
      if (addend) ro.v[c] += radd[ob].v[c];
    }
    if (GATE) {
      // (`out` = the da_f planes, `dag` the da_g planes)
      const RowRegs rt = rows_load(th + ob * ts_pstride + r0 * 32, lane, 0, hi);
      const RowRegs rs = rows_load(sg + ob * ts_pstride + r0 * 32, lane, 0, hi);
      RowRegs rg;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zs = ro.v[c][e] * rs.v[c][e];
          ro.v[c][e] = zs * (1.f - rt.v[c][e] * rt.v[c][e]);
          rg.v[c][e] = zs * rt.v[c][e] * (1.f - rs.v[c][e]);
        }
      rows_store(dag + ob * out_pstride + r0 * 32, lane, hi, rg);
    }
    rows_store(out + ob * out_pstride + r0 * 32, lane, hi, ro);
  }
}

extern "C" {

int wn_layer_fwd(const float* x, float* x_out, float* z, float* th, float* sg,
                 const float* wblock, const float* bias_fg,
                 int bias_clip_stride, int B, int T, int dilation,
                 int has_dense, int save_ts, void* stream) {
  if (!x || !z || !wblock) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || dilation <= 0) return WN_ERR_BAD_SHAPE;
  if (has_dense && !x_out) return WN_ERR_NULL;
  if (save_ts < 0 || save_ts > 2) return WN_ERR_BAD_SHAPE;
  if ((save_ts == 1 && !th) || (save_ts && !sg)) return WN_ERR_NULL;
  if (!wn_aligned16(x) || !wn_aligned16(z) || !wn_aligned16(wblock) ||
      (x_out && !wn_aligned16(x_out)) || (th && !wn_aligned16(th)) ||
      (sg && !wn_aligned16(sg)))
    return WN_ERR_MISALIGNED;
  dim3 grid(layer_grid(B, T, LAYER_WAVES, true)), block(LAYER_WG);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HD, TS)                                                      \
  hipLaunchKernelGGL((layer_fwd_kernel<HD, TS>), grid, block, 0, s, x,      \
                     x_out, z, th, sg, wblock, bias_fg, bias_clip_stride, B, \
                     T, dilation)
  if (has_dense) {
    if (save_ts == 2) LAUNCH(true, 2); else if (save_ts) LAUNCH(true, 1); else LAUNCH(true, 0);
  } else {
    if (save_ts == 2) LAUNCH(false, 2); else if (save_ts) LAUNCH(false, 1); else LAUNCH(false, 0);
  }
#undef LAUNCH
  return wn_check_launch();
}

int wn_layer_wgrad_slab_floats(void) { return LAYER_BLOCK_FLOATS; }

// ---- generic filter width (K taps); block = (2K+1)*1024 + 96 floats
int wn_layer_fwd_k(const float* x, float* x_out, float* z, float* th,
                   float* sg, const float* wblock, const float* bias_fg,
                   int bias_clip_stride, int B, int T, int dilation, int K,
                   int has_dense, int save_ts, void* stream) {
  if (!x || !z || !wblock) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || dilation <= 0 || K < 2) return WN_ERR_BAD_SHAPE;
  if (K > 8) return WN_ERR_UNSUPPORTED;
  if (has_dense && !x_out) return WN_ERR_NULL;
  if (save_ts && (!th || !sg)) return WN_ERR_NULL;
  dim3 grid(layer_grid(B, T, GEN_WAVES)), block(GEN_WG);
  const size_t lds = ((size_t)(2 * K + 1) * 1024 + 32 + GEN_WAVES * 1024) * 4;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HD, TS)                                                       \
  if (hipFuncSetAttribute((const void*)layer_fwd_gen_kernel<HD, TS>,         \
                          hipFuncAttributeMaxDynamicSharedMemorySize,        \
                          (int)lds) != hipSuccess)                           \
    return WN_ERR_LAUNCH;                                                    \
  hipLaunchKernelGGL((layer_fwd_gen_kernel<HD, TS>), grid, block, lds, s, x, \
                     x_out, z, th, sg, wblock, bias_fg, bias_clip_stride, B, \
                     T, dilation, K)
  if (has_dense && save_ts) { LAUNCH(true, true); }
  else if (has_dense) { LAUNCH(true, false); }
  else if (save_ts) { LAUNCH(false, true); }
  else { LAUNCH(false, false); }
#undef LAUNCH
  return wn_check_launch();
}

int wn_layer_bwd_k(const float* daf_cur, const float* dag_cur,
                   const float* dxin, float* dx_out, const float* wblock_b,
                   const float* dZ, const float* th, const float* sg,
                   const float* wblock_a, float* daf_next, float* dag_next,
                   int B, int T, int dilation, int K, int do_b, int do_a,
                   int blocks, long blk_stride, void* stream) {
  if (B <= 0 || T <= 0 || dilation <= 0 || K < 2 || blocks < 1) return WN_ERR_BAD_SHAPE;
  if (K > 8) return WN_ERR_UNSUPPORTED;
  if (!do_a && !do_b) return WN_ERR_BAD_SHAPE;
  // (several blocks per launch: the gate-gradient phase alone)
  if (blocks > 1 && (do_b || dxin || blk_stride < (long)B * T * 32)) return WN_ERR_BAD_SHAPE;
  if (do_b && (!daf_cur || !dag_cur || !dx_out || !wblock_b)) return WN_ERR_NULL;
  if (do_a && (!dZ || !th || !sg || !wblock_a || !daf_next || !dag_next))
    return WN_ERR_NULL;
  dim3 grid(layer_grid(B, T, GEN_WAVES), blocks), block(GEN_WG);
  const size_t lds = (((size_t)(2 * K + 1) * 33 * 32 + 3) / 4 * 4 +
                      GEN_WAVES * 2048) * 4;
  hipStream_t s = (hipStream_t)stream;
  const bool hx = dxin != nullptr;
#define LAUNCH(DB, DA, HX)                                                    \
  if (hipFuncSetAttribute((const void*)layer_bwd_gen_kernel<DB, DA, HX>,      \
                          hipFuncAttributeMaxDynamicSharedMemorySize,         \
                          (int)lds) != hipSuccess)                            \
    return WN_ERR_LAUNCH;                                                     \
  hipLaunchKernelGGL((layer_bwd_gen_kernel<DB, DA, HX>), grid, block, lds, s, \
                     daf_cur, dag_cur, dxin, dx_out, wblock_b, dZ, th, sg,    \
                     wblock_a, daf_next, dag_next, B, T, dilation, K, blk_stride)
  if (do_b && do_a) { if (hx) { LAUNCH(true, true, true); } else { LAUNCH(true, true, false); } }
  else if (do_b) { if (hx) { LAUNCH(true, false, true); } else { LAUNCH(true, false, false); } }
  else { if (hx) { LAUNCH(false, true, true); } else { LAUNCH(false, true, false); } }
#undef LAUNCH
  return wn_check_launch();
}

// ---- more than 32 channels: one launch per 32-wide OUTPUT block
int wn_layer_fwd_blk(const float* x, long in_plane_stride, int in_blocks,
                     float* z, float* th, float* sg, const float* wf,
                     const float* wg, int ldw, const float* bias_f,
                     const float* bias_g, int bias_clip_stride, int B, int T,
                     int dilation, int K, int save_ts, int tap_rows,
                     const float* pre_in, float* pre_out, long pre_plane_stride,
                     int k0, int Ktot, int out_blocks, long out_plane_stride,
                     void* stream) {
  if (!x || !wf || !wg) return WN_ERR_NULL;
  if (k0 < 0 || Ktot < k0 + K || out_blocks < 1 ||
      (out_blocks > 1 && (out_plane_stride < (long)B * T * 32 || (out_plane_stride & 3))))
    return WN_ERR_BAD_SHAPE;
  if (!pre_out && !z) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || dilation <= 0 || K < 1 || in_blocks < 1 || ldw < 32 ||
      tap_rows < in_blocks * 32)
    return WN_ERR_BAD_SHAPE;
  if (K * in_blocks > 8) return WN_ERR_UNSUPPORTED;
  if (!pre_out && save_ts && (!th || !sg)) return WN_ERR_NULL;
  if ((pre_in && !wn_aligned16(pre_in)) || (pre_out && !wn_aligned16(pre_out)) ||
      (pre_plane_stride & 3) != 0)
    return WN_ERR_MISALIGNED;
  const void* ptrs[] = {x, z, th, sg};
  for (const void* p : ptrs)
    if (p && !wn_aligned16(p)) return WN_ERR_MISALIGNED;
  if ((in_plane_stride & 3) != 0) return WN_ERR_MISALIGNED;
  dim3 grid(layer_grid(B, T, GEN_WAVES), out_blocks), block(GEN_WG);
  const size_t lds = ((size_t)2 * K * in_blocks * 1024 + GEN_WAVES * 1024) * 4;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(TS)                                                            \
  if (hipFuncSetAttribute((const void*)layer_fwd_blk_kernel<TS>,              \
                          hipFuncAttributeMaxDynamicSharedMemorySize,         \
                          (int)lds) != hipSuccess)                            \
    return WN_ERR_LAUNCH;                                                     \
  hipLaunchKernelGGL((layer_fwd_blk_kernel<TS>), grid, block, lds, s, x,      \
                     in_plane_stride, in_blocks, z, th, sg, wf, wg, ldw,      \
                     bias_f, bias_g, bias_clip_stride, B, T, dilation, K,     \
                     tap_rows, pre_in, pre_out, pre_plane_stride, k0, Ktot,     \
                     out_plane_stride)
  if (save_ts) { LAUNCH(true); } else { LAUNCH(false); }
#undef LAUNCH
  return wn_check_launch();
}

int wn_layer_bwd_blk(const float* daf, const float* dag, long da_plane_stride,
                     int da_blocks, const float* dxin, float* dx_out,
                     const float* wf, const float* wg, int ldw, long tap_stride,
                     int B, int T, int dilation, int K, int k0, int Ktot,
                     int dx_blocks, long dx_plane_stride, void* stream) {
  if (!daf || !dag || !dx_out || !wf || !wg) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || dilation <= 0 || K < 1 || da_blocks < 1 || ldw < 32 ||
      k0 < 0 || Ktot < k0 + K || dx_blocks < 1 ||
      (dx_blocks > 1 && (dx_plane_stride < (long)B * T * 32 || (dx_plane_stride & 3))))
    return WN_ERR_BAD_SHAPE;
  if (K * da_blocks > 8) return WN_ERR_UNSUPPORTED;
  const void* ptrs[] = {daf, dag, dxin, dx_out};
  for (const void* p : ptrs)
    if (p && !wn_aligned16(p)) return WN_ERR_MISALIGNED;
  if ((da_plane_stride & 3) != 0) return WN_ERR_MISALIGNED;
  dim3 grid(layer_grid(B, T, GEN_WAVES), dx_blocks), block(GEN_WG);
  const size_t lds = (((size_t)2 * K * da_blocks * 33 * 32 + 3) / 4 * 4 +
                      GEN_WAVES * 2048) * 4;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HX)                                                            \
  if (hipFuncSetAttribute((const void*)layer_bwd_blk_kernel<HX>,              \
                          hipFuncAttributeMaxDynamicSharedMemorySize,         \
                          (int)lds) != hipSuccess)                            \
    return WN_ERR_LAUNCH;                                                     \
  hipLaunchKernelGGL((layer_bwd_blk_kernel<HX>), grid, block, lds, s, daf,    \
                     dag, da_plane_stride, da_blocks, dxin, dx_out, wf, wg,   \
                     ldw, tap_stride, B, T, dilation, K, k0, Ktot,              \
                     dx_plane_stride)
  if (dxin) { LAUNCH(true); } else { LAUNCH(false); }
#undef LAUNCH
  return wn_check_launch();
}

int wn_layer_wgrad_k(const float* x, const float* daf, const float* dag,
                     const float* z, const float* dxin, float* slabs,
                     int num_slabs, int B, int T, int dilation, int K, int k0,
                     int Ktot, int CB, long plane_stride, void* stream) {
  if (!x || !daf || !dag || !slabs) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || dilation <= 0 || num_slabs <= 0 || K < 1 || k0 < 0 ||
      Ktot < k0 + K || CB < 1 || (CB > 1 && plane_stride < (long)B * T * 32))
    return WN_ERR_BAD_SHAPE;
  if (K > 8) return WN_ERR_UNSUPPORTED;
  if ((dxin != nullptr) != (z != nullptr)) return WN_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(num_slabs, CB * CB), block(256);
  if (K == 2 && k0 == 0 && Ktot == 2 && (CB == 2 || CB == 4) &&
      (long)B * T * WN_CH * 4 < (1L << 31)) {
    // 64 / 128 channels, two taps: all input blocks and two output blocks per
    // pass (every plane of a 64-channel layer read once, of a 128-channel
    // layer twice)
    dim3 grid1(num_slabs, CB / 2), block1(512);
#define LAUNCH(CBV, HD)                                                                  \
  hipLaunchKernelGGL((layer_wgrad_cbn_kernel<CBV, HD>), grid1, block1, 0, s, x, daf, dag, z, \
                     dxin, slabs, B, T, dilation, plane_stride)
    if (CB == 2) { if (dxin) LAUNCH(2, true); else LAUNCH(2, false); }
    else { if (dxin) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
    return wn_check_launch();
  }
  if (K == 2 && k0 == 0 && Ktot == 2) {
    // two taps: the one-pass kernel of the 32-channel models per block pair
    // (same slab layout; 24 instead of 32 plane reads per layer at 64 channels)
    if (dxin)
      hipLaunchKernelGGL((layer_wgrad_kernel<true>), grid, block, 0, s, x, daf, dag, z,
                         dxin, slabs, B, T, dilation, CB, plane_stride);
    else
      hipLaunchKernelGGL((layer_wgrad_kernel<false>), grid, block, 0, s, x, daf, dag, z,
                         dxin, slabs, B, T, dilation, CB, plane_stride);
    return wn_check_launch();
  }
  if (dxin)
    hipLaunchKernelGGL((layer_wgrad_gen_kernel<true>), grid, block, 0, s, x,
                       daf, dag, z, dxin, slabs, B, T, dilation, K, k0, Ktot, CB,
                       plane_stride);
  else
    hipLaunchKernelGGL((layer_wgrad_gen_kernel<false>), grid, block, 0, s, x,
                       daf, dag, z, dxin, slabs, B, T, dilation, K, k0, Ktot, CB,
                       plane_stride);
  return wn_check_launch();
}

// number of slabs (workgroups) wn_layer_bwdw writes for this shape
// out planes = addend planes + in planes * W (+ bias): the 1x1 residual conv of
// the channel-block models and its data gradient as a streaming kernel (see
// dense_planes_kernel).  CB = C / 32 planes of [rows][32] each way, 1 <= CB <= 4;
// W [C][C] row-major ([Cin][Cout]); bias [C] or NULL; addend planes or NULL.
int wn_dense_planes(const float* in, long in_plane_stride, const float* W,
                    const float* bias, const float* addend, long add_plane_stride,
                    float* out, long out_plane_stride, long rows, int C,
                    void* stream) {
  if (!in || !W || !out) return WN_ERR_NULL;
  if (rows <= 0) return WN_ERR_BAD_SHAPE;
  if (C != 32 && C != 64 && C != 96 && C != 128) return WN_ERR_UNSUPPORTED;
  const void* ptrs[] = {in, W, bias, addend, out};
  for (const void* p : ptrs)
    if (p && !wn_aligned16(p)) return WN_ERR_MISALIGNED;
  if ((in_plane_stride & 3) || (add_plane_stride & 3) || (out_plane_stride & 3))
    return WN_ERR_MISALIGNED;
  const long ntiles = (rows + 31) / 32;
  const long wgs = (ntiles + 3) / 4;
  if (wgs > 0x7fffffffL) return WN_ERR_BAD_SHAPE;
  dim3 grid((unsigned)wgs), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(CBV)                                                              \
  hipLaunchKernelGGL((dense_planes_kernel<CBV, false>), grid, block, 0, s, in, in_plane_stride, \
                     W, bias, addend, add_plane_stride, out, out_plane_stride, rows,     \
                     nullptr, nullptr, 0L, nullptr)
  if (C == 32) LAUNCH(1);
  else if (C == 64) LAUNCH(2);
  else if (C == 96) LAUNCH(3);
  else LAUNCH(4);
#undef LAUNCH
  return wn_check_launch();
}

// dz = addend + in W as wn_dense_planes, then the gate gradients instead of dz:
// daf = dz s (1 - t^2), dag = dz s t (1 - s) with the saved tanh / sigmoid planes
// th / sg (planes ts_plane_stride apart); daf / dag planes da_plane_stride apart
int wn_dense_planes_gate(const float* in, long in_plane_stride, const float* W,
                         const float* addend, long add_plane_stride, const float* th,
                         const float* sg, long ts_plane_stride, float* daf, float* dag,
                         long da_plane_stride, long rows, int C, void* stream) {
  if (!in || !W || !th || !sg || !daf || !dag) return WN_ERR_NULL;
  if (rows <= 0) return WN_ERR_BAD_SHAPE;
  if (C != 32 && C != 64 && C != 96 && C != 128) return WN_ERR_UNSUPPORTED;
  const void* ptrs[] = {in, W, addend, th, sg, daf, dag};
  for (const void* p : ptrs)
    if (p && !wn_aligned16(p)) return WN_ERR_MISALIGNED;
  if ((in_plane_stride & 3) || (add_plane_stride & 3) || (ts_plane_stride & 3) ||
      (da_plane_stride & 3))
    return WN_ERR_MISALIGNED;
  const long ntiles = (rows + 31) / 32;
  const long wgs = (ntiles + 3) / 4;
  if (wgs > 0x7fffffffL) return WN_ERR_BAD_SHAPE;
  dim3 grid((unsigned)wgs), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(CBV)                                                              \
  hipLaunchKernelGGL((dense_planes_kernel<CBV, true>), grid, block, 0, s, in, in_plane_stride, \
                     W, nullptr, addend, add_plane_stride, daf, da_plane_stride, rows,  \
                     th, sg, ts_plane_stride, dag)
  if (C == 32) LAUNCH(1);
  else if (C == 64) LAUNCH(2);
  else if (C == 96) LAUNCH(3);
  else LAUNCH(4);
#undef LAUNCH
  return wn_check_launch();
}

// number of slabs (workgroups) wn_layer_bwd2 writes for this shape (an upper
// bound over the kernel variants: slabs past the launched grid stay unused)
int wn_layer_bwd2_slabs(int B, int T) {
  return layer_grid(B, T, B2_WAVES, true);
}

int wn_layer_bwd2_wimg_floats(void) { return B2_WIMG; }

int wn_layer_bwd2_pack(const float* layer0, long layer_stride, float* wimg,
                       int L, void* stream) {
  if (!layer0 || !wimg) return WN_ERR_NULL;
  if (L <= 0) return WN_ERR_BAD_SHAPE;
  if (!wn_aligned16(wimg)) return WN_ERR_MISALIGNED;
  hipLaunchKernelGGL(bwd2_pack_kernel, dim3(L), dim3(256), 0,
                     (hipStream_t)stream, layer0, layer_stride, wimg);
  return wn_check_launch();
}

int wn_layer_bwd2(const float* x, const float* z, const float* sg,
                  const float* dZ, const float* dxin, float* dx_out,
                  const float* wblock, const float* wimg, float* slabs,
                  float* tile_colsum, int B, int T, int dilation,
                  void* stream) {
  if (B <= 0 || T <= 0 || dilation <= 0) return WN_ERR_BAD_SHAPE;
  if (!x || !z || !sg || !dZ || !dx_out || !wblock || !wimg || !slabs)
    return WN_ERR_NULL;
  const void* ptrs[] = {x, z, sg, dZ, dxin, dx_out, wblock, wimg};
  for (const void* p : ptrs)
    if (p && !wn_aligned16(p)) return WN_ERR_MISALIGNED;
  dim3 grid(layer_grid(B, T, B2_WAVES, true)), block(B2_WAVES * 64);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(KERNEL, W)                                                     \
  hipLaunchKernelGGL(KERNEL, grid, block, 0, s, x, z, sg, dZ, dxin, dx_out,   \
                     W, slabs, tile_colsum, B, T, dilation)
  if (dxin) LAUNCH((layer_bwd2d_kernel<true>), wimg);
  else LAUNCH((layer_bwd2d_kernel<false>), wimg);
#undef LAUNCH
  return wn_check_launch();
}

}  // extern "C"
