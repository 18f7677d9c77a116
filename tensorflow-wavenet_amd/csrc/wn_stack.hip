// The whole residual stack in ONE launch: a persistent, dependency-driven
// version of layer_fwd_kernel (wn_layer.hip) for
// WaveNetModel._create_dilation_layer x L (wavenet/model.py:236-330, the loop
// at model.py:417-428).
//
// A per-layer launch pays a fixed start and end (launch, weight staging, first
// loads, the slowest wave's tail: ~5 us of the forward kernel's 21 us at
// B*T = 128000) fifty times, and reloads from HBM the tile it has just written.
// Here a workgroup keeps a GROUP of consecutive 32-row tiles (one per wave, 16
// waves unless the batch is small) and walks it through all L layers:
//   * the tile's own rows x_l[t] never leave the wave's registers: the dense
//     1x1 accumulator of layer l IS the B operand of layer l+1;
//   * the dilated tap x_l[t-d] belongs to another wave (another workgroup,
//     usually another XCD): x_{l+1} is written through to memory (sc1 stores),
//     the producer publishes flag[l+1][tile] = epoch after `s_waitcnt vmcnt(0)`,
//     the consumer polls the flag(s) of the one or two tiles its rows come from
//     and reads them with sc1 loads (the per-XCD L2s are not coherent with
//     each other; sc1 accesses are performed at device scope);
//   * weights sit in a two-layer LDS ring: the wave that is LAST to finish
//     layer l refills that layer's half with layer l+2 by LDS-DMA and marks it
//     ready one layer later; the waves of a workgroup are never held at a
//     barrier inside the stack and drift up to a layer apart, so one wave's
//     flag / load latency hides under the MFMAs of the others on its SIMD.
// No grid-wide barrier anywhere: a wave waits only for the tiles it reads.
//
// Deadlock freedom does not depend on how many workgroups are resident: groups
// are handed out by an atomic ticket in increasing order, a group's rows depend
// only on rows of the same or LOWER groups (the taps are causal), and a
// workgroup finishes all layers of its group before it takes another ticket --
// so the lowest unfinished group can always run to completion.  Every poll is
// additionally bounded (2 s of s_memrealtime): on expiry the wave records
// ctl[3] = 1 and *poison = NaN (a float the host sums into the loss, which so
// turns NaN instead of being silently wrong; the backward also poisons its dx)
// and carries on without waiting, so the grid always drains.
#include "wn_common.h"

#define WN_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#define STACK_MAXL 256   // layers (LDS bookkeeping words per layer)
// Forward weight image of one layer (wn_stack_pack): the five 32 x 32
// matrices TRANSPOSED, [m][cout][SF_LD] with the contraction index (cin)
// contiguous and rows padded to 36 floats, then the dense bias.  A lane's four
// consecutive A operands (cin = 8q+4h .. +3 of its cout row) are ONE
// ds_read_b128 instead of four ds_read_b32 (row stride 36 floats: the eight
// lanes of a 128-byte LDS beat hit 32 different banks), so a tile's 80 MFMAs
// are fed by 20 LDS instructions.
// (the z / sigmoid plane stores with an nt hint: 760 -> 772 us, not kept)
#define SF_LD 36
#define SF_MT (32 * SF_LD)
#define SF_OFF_BD (5 * SF_MT)                // 5760
#define STACK_WBUF 5888                      // 23 KiB per layer (1 KiB multiple)

struct StackFwd {
  float* X;            // [L][N][32]  X[l] = input of layer l (X[0]: causal layer)
  float* Z;            // [L][N][32]
  float* SG;           // [L][N][32] sigmoid planes (SAVE == 2) or null
  const float* wimg;   // [L][STACK_WBUF] forward weight images (wn_stack_pack)
  const float* bias;   // [L][B or 1][64] filter|gate bias (+ gc), or null
  long bias_layer_stride;
  int bias_clip_stride;
  const int* dil;      // [L] dilations (device)
  unsigned* flags;     // [L][ntiles]
  unsigned* ctl;       // [0] group ticket [1] workgroups done [2] epoch [3] error
  float* poison;       // set to NaN when a bounded wait expires (or null)
  int L, B, T;
  long plane;          // N * 32 floats
  // fused skip sum (stack_fwd16_kernel<.., true>): the skip weights as
  // half-stage images (wn_stack_skip_pack), sum_l bs_l or null, h1 out [N][512]
  const float* skimg;
  const float* sk_bsum;
  float* sk_out;
#ifdef STACK_STAMPS
  unsigned long long* dbg;   // diagnostic build: [grid][16][L][12] s_memtime stamps
#endif
};

// Fused skip sum of the 16-row forward (small batches), S = 512.  Partner wave j
// of a workgroup adds the products of ALL four tiles of the group for skip
// columns 128 j .. 128 j + 127; its operands -- Ws_l[k][column] as 16x16x4 A
// fragments -- come straight from memory in the lane order it consumes them:
// image [L][4 j][8 n-tiles][2 k-halves][64 lanes][4 floats], lane (jr, g):
// Ws_l[16 half + 4 g + e][128 j + 16 nt + jr], e = 0 .. 3 (one 1-KiB load
// instruction per fragment pair), SK_AHEAD n-tiles ahead of their use.
#define SK_S 512
#define SK_NT 8                              // 16-column n-tiles of a partner wave
#define SK_WAVE (SK_NT * 2 * 256)            // floats of a (layer, wave) image: 16 KiB
#define SK_AHEAD 3
#define SK_ZB 4                              // z hand-over tiles per slot (layers a chain wave may run ahead)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// device-scope (sc1) row-contiguous tile access: same lane <-> row mapping as
// rows_load / rows_store of wn_common.h
__device__ __forceinline__ RowRegs rows_load_dev(const float* tile0, int lane,
                                                 int lo, int hi) {
  RowRegs R;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)tile0, 0, 0x7fffffff, 0x00020000);
  const int vo = ((lane >> 3) * 32 + (lane & 7) * 4) * 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = 8 * c + (lane >> 3);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= lo && r < hi)
      v = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + c * 1024, 0, 16));
    R.v[c] = v;
  }
  return R;
}

template <int AUX>   // 16 = sc1 (device scope)
__device__ __forceinline__ void rows_store_aux(float* tile0, int lane, int hi,
                                               const RowRegs& R) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)tile0, 0, 0x7fffffff, 0x00020000);
  const int vo = ((lane >> 3) * 32 + (lane & 7) * 4) * 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = 8 * c + (lane >> 3);
    if (r < hi)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, R.v[c]),
                                             rs, vo + c * 1024, 0, AUX);
  }
}
__device__ __forceinline__ void rows_store_dev(float* tile0, int lane, int hi,
                                               const RowRegs& R) {
  rows_store_aux<16>(tile0, lane, hi, R);
}

// poll until flag[idx] == epoch on every lane that has an idx >= 0
__device__ __forceinline__ void wait_flags(const unsigned* fl, int idx,
                                           unsigned epoch, unsigned* ctl,
                                           float* poison, bool& dead, int lane) {
  if (dead) return;
  unsigned spins = 0;
  unsigned long long t_start = 0;
  for (;;) {
    bool pending = false;
    if (idx >= 0)
      pending = __hip_atomic_load(fl + idx, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT) != epoch;
    if (__builtin_amdgcn_ballot_w64(pending) == 0) break;
    if ((++spins & 63u) == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t_start == 0) t_start = now;
      if (now - t_start > 200000000ull) {   // 2 s at 100 MHz
        if (lane == 0) {
          __hip_atomic_store(ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (poison) *poison = __builtin_nanf("");   // summed into the loss by the host
        }
        dead = true;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// acc^T[cout = i, time] += sum_k W[k][i] * frag[time, k] with the TRANSPOSED
// image: `wt_lane` = image + i * SF_LD + 4 * h; the four operands of
// r = 4q .. 4q+3 (cin = 8q + 4h + e) are one 16-byte LDS read.  Same MFMA order
// as mma32 (bitwise the same result).
// Chunk q + 1 is requested before the MFMAs of chunk q, order pinned (left
// alone the compiler reads two chunks, waits, issues eight MFMAs: B = 8 829 ->
// 810 us).  The matrix's first chunk arrives in `pre`; the first chunk of the
// NEXT matrix (`wt_next`) is requested before this one's last MFMAs.
__device__ __forceinline__ void mma32t_chain(f32x16& acc, const f32x16& frag,
                                             const float* wt_lane, f32x4& pre,
                                             const float* wt_next) {
  f32x4 nxt = pre;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 a4 = nxt;
    nxt = *reinterpret_cast<const f32x4*>(q < 3 ? wt_lane + 8 * (q + 1) : wt_next);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], frag[4 * q + e], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  pre = nxt;
}

// blockIdx.y == 0: forward image (transposed, + dense bias); 1: backward image
// (the matrices as they are -- there the contraction runs over cout -- dense
// [m][cin][32] with the 16-byte chunks of row cin at chunk ^ ((cin >> 1) & 7),
// see mma32s)
__global__ void stack_pack_kernel(const float* __restrict__ layer0,
                                  long layer_stride, float* __restrict__ img_f,
                                  float* __restrict__ img_b) {
  const bool bwd = blockIdx.y == 1;
  float* img = bwd ? img_b : img_f;
  if (!img) return;
  const float* blk = layer0 + (long)blockIdx.x * layer_stride;
  float* out = img + (size_t)blockIdx.x * STACK_WBUF;
  for (int i = threadIdx.x; i < STACK_WBUF; i += blockDim.x) out[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < LAYER_W_FLOATS; i += blockDim.x) {
    const int m = i >> 10, cin = (i >> 5) & 31, cout = i & 31;   // W[m][cin][cout]
    out[bwd ? m * 1024 + cin * 32 + ((((cout >> 2) ^ ((cin >> 1) & 7)) << 2) | (cout & 3))
            : m * SF_MT + cout * SF_LD + cin] = blk[i];
  }
  if (!bwd && threadIdx.x < 32)
    out[SF_OFF_BD + threadIdx.x] = blk[LAYER_OFF_BD + threadIdx.x];
}

// bounded spin on a workgroup-local LDS word; an expired wait is recorded like
// an expired flag poll (ctl[3], NaN poison) -- the wave then runs on
__device__ __forceinline__ void wait_lds(const int* p, bool& dead, unsigned* ctl,
                                         float* poison, int lane) {
  unsigned spins = 0;
  while (!dead && __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 24)) {
      dead = true;
      if (lane == 0) {
        __hip_atomic_store(ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (poison) *poison = __builtin_nanf("");
      }
    }
  }
  asm volatile("" ::: "memory");
}

// skip weights [L * 32][512] -> per-(layer, partner wave) operand images (see SK_WAVE)
__global__ void stack_skip_pack_kernel(const float* __restrict__ skip_w, float* __restrict__ img) {
  const int l = blockIdx.x >> 2, j = blockIdx.x & 3;
  float* out = img + (size_t)blockIdx.x * SK_WAVE;
  for (int i = threadIdx.x; i < SK_WAVE; i += blockDim.x) {
    const int e = i & 3, lane = (i >> 2) & 63, half = (i >> 8) & 1, nt = i >> 9;
    const int jr = lane & 15, g = lane >> 4;
    out[i] = skip_w[((size_t)l * 32 + 16 * half + 4 * g + e) * SK_S + 128 * j + 16 * nt + jr];
  }
}

template <int SAVE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stack_fwd_kernel(StackFwd a) {
  __shared__ __attribute__((aligned(1024))) float wl[2 * STACK_WBUF];
  __shared__ __attribute__((aligned(16))) float tiles[WAVES * 1024];
  __shared__ int s_group;
  // per layer: waves that no longer read its weights / its weights are in LDS
  __shared__ int s_done[STACK_MAXL], s_ready[STACK_MAXL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int T = a.T, L = a.L;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * a.B;
  const int ngroups = (ntiles + WAVES - 1) / WAVES;
  const unsigned epoch =
      __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* ta = tiles + wave * 1024;
  bool dead = false;
#ifdef STACK_STAMPS
#define SSTAMP(l, i)                                                         \
  if (lane == 0)                                                             \
    a.dbg[(((size_t)blockIdx.x * 16 + wave) * L + (l)) * 12 + (i)] =         \
        __builtin_amdgcn_s_memtime()
  // clock calibration per workgroup: {realtime, memtime} at entry and exit
  unsigned long long* cal = a.dbg + (size_t)gridDim.x * 16 * L * 12 + (size_t)blockIdx.x * 4;
  if (tid == 0) { cal[0] = __builtin_amdgcn_s_memrealtime(); cal[1] = __builtin_amdgcn_s_memtime(); }
#else
#define SSTAMP(l, i)
#endif

  // pieces [p0, p0 + step, ...) of layer l's 23 KiB weight image by LDS-DMA
  auto issue_weights = [&](int l, int p0, int step) {
    const float* wb = a.wimg + (size_t)l * STACK_WBUF;
    float* dst = wl + (l & 1) * STACK_WBUF;
    for (int p = p0; p < STACK_WBUF / 256; p += step)
      __builtin_amdgcn_global_load_lds((wn_gptr_t)(wb + p * 256 + lane * 4),
                                       (wn_lptr_t)(dst + p * 256), 16, 0, 0);
  };

  for (;;) {
    if (tid == 0) s_group = (int)atomicAdd(a.ctl, 1u);
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(s_group);
    __syncthreads();
    if (g >= ngroups) break;
    const int tile = g * WAVES + wave;
    const int nactive = min(WAVES, ntiles - g * WAVES);
    const bool any = tile < ntiles;
    const int b = any ? tile / tiles_per_clip : 0;
    const int tt = any ? tile - b * tiles_per_clip : 0;   // tile index inside the clip
    const int t0 = tt * 32;
    const int hi = any ? min(32, T - t0) : 0;
    const size_t off0 = ((size_t)b * T + t0) * WN_CH;

    for (int i = tid; i < L; i += WAVES * 64) {
      s_done[i] = 0;
      s_ready[i] = i < 2;
    }
    issue_weights(0, wave, WAVES);
    if (L > 1) issue_weights(1, wave, WAVES);
    f32x16 xc;
    {
      const RowRegs rc = rows_load_dev(a.X + off0, lane, 0, hi);
      rows_to_lds(ta, lane, rc);
      __builtin_amdgcn_wave_barrier();
      xc = frag_from_lds(ta, j, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();        // weights of layers 0 and 1 are in LDS
    int publish = 0;        // layer whose weights this wave has in flight

    for (int l = 0; any && l < L; ++l) {
      const int d = a.dil[l];
      SSTAMP(l, 0);
      wait_lds(s_ready + l, dead, a.ctl, a.poison, lane);
      SSTAMP(l, 1);
      const float* xl = a.X + (size_t)l * a.plane;
      int woff = j * SF_LD + 4 * h + (l & 1) * STACK_WBUF;
      asm volatile("" : "+v"(woff));
      const float* wlane = wl + woff;
      f32x16 af, ag;
      if (a.bias) {
        const float* bp = a.bias + (size_t)l * a.bias_layer_stride +
                          (size_t)b * a.bias_clip_stride;
        af = frag_bcast(bp, h);
        ag = frag_bcast(bp + 32, h);
      } else {
        af = frag_zero();
        ag = frag_zero();
      }
      // ---- the dilated tap: rows t0-d .. t0-d+31 of x_l, written by the
      // owners of (at most) two tiles of this clip.  Their flags are requested
      // FIRST and looked at after the current-tap products, which need only the
      // wave's own rows: 32 of the layer's 80 MFMAs leave the tile-to-tile
      // dependency chain and cover the flag's round trip.
      const int lo_row = t0 - d;
      int fidx = -1;
      unsigned fval = epoch;
      if (l > 0 && lo_row + 31 >= 0) {
        const int first = max(lo_row, 0) >> 5, last = (lo_row + 31) >> 5;
        if (lane == 0 && first != tt) fidx = b * tiles_per_clip + first;
        if (lane == 1 && last != first && last != tt) fidx = b * tiles_per_clip + last;
        if (fidx >= 0)
          fval = __hip_atomic_load(a.flags + (size_t)l * ntiles + fidx, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
      }
      // (-DSTACK_NOMMA / -DSTACK_NOACT: timing-only ablation builds, see
      // DESIGN.md 3a; their results are meaningless)
#ifndef STACK_NOMMA
      // (the first chunk of each weight matrix is requested during the matrix
      // before it: 777 -> 772 us)
      f32x4 wpre = *reinterpret_cast<const f32x4*>(wlane + 1 * SF_MT);
      mma32t_chain(af, xc, wlane + 1 * SF_MT, wpre, wlane + 3 * SF_MT);  // Wf[1]: current tap
      mma32t_chain(ag, xc, wlane + 3 * SF_MT, wpre, wlane + 0 * SF_MT);  // Wg[1]
#endif
      if (__builtin_amdgcn_ballot_w64(fval != epoch) != 0)
        wait_flags(a.flags + (size_t)l * ntiles, fidx, epoch, a.ctl, a.poison, dead, lane);
      SSTAMP(l, 2);
      f32x16 xp;
      {
        const RowRegs rp = rows_load_dev(xl + off0 - (size_t)d * WN_CH, lane,
                                         max(0, d - t0), hi);
        __builtin_amdgcn_wave_barrier();
        rows_to_lds(ta, lane, rp);
        __builtin_amdgcn_wave_barrier();
        xp = frag_from_lds(ta, j, h);
      }
      if (publish) {
        // the weight pieces issued at the end of the previous layer are older
        // than the loads just consumed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(s_ready + publish, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        publish = 0;
      }
#ifdef STACK_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      SSTAMP(l, 3);
#ifndef STACK_NOMMA
      mma32t_chain(af, xp, wlane + 0 * SF_MT, wpre, wlane + 2 * SF_MT);  // Wf[0]: past tap
      mma32t_chain(ag, xp, wlane + 2 * SF_MT, wpre, wlane + 4 * SF_MT);  // Wg[0]
#else
#pragma unroll
      for (int r = 0; r < 16; ++r) { af[r] += xp[r] * wlane[0]; ag[r] += xc[r] * wlane[SF_MT]; }
#endif
      SSTAMP(l, 4);
      f32x16 zz;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifndef STACK_NOACT
        af[r] = wn_tanh(af[r]);
        ag[r] = wn_sigmoid(ag[r]);
#endif
        zz[r] = af[r] * ag[r];
      }
      SSTAMP(l, 5);
      // x' first: it is what other waves wait for
      if (l + 1 < L) {
        const f32x16 bd = frag_bcast(wl + (l & 1) * STACK_WBUF + SF_OFF_BD, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[r] += bd[r];
#ifndef STACK_NOMMA
        mma32t_chain(xc, zz, wlane + 4 * SF_MT, wpre, wlane + 4 * SF_MT);  // Wd
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[r] += zz[r] * wlane[4 * SF_MT];
#endif
        SSTAMP(l, 6);
        __builtin_amdgcn_wave_barrier();
        frag_to_lds(ta, j, h, xc);
        __builtin_amdgcn_wave_barrier();
        rows_store_dev(a.X + (size_t)(l + 1) * a.plane + off0, lane, hi,
                       rows_from_lds(ta, lane));
        // x_{l+1} of this tile is in memory: publish it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(a.flags + (size_t)(l + 1) * ntiles + tile, epoch,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SSTAMP(l, 7);
      }
      __builtin_amdgcn_wave_barrier();
      frag_to_lds(ta, j, h, zz);
      __builtin_amdgcn_wave_barrier();
      rows_store(a.Z + (size_t)l * a.plane + off0, lane, hi, rows_from_lds(ta, lane));
      if (SAVE == 2) {
        __builtin_amdgcn_wave_barrier();
        frag_to_lds(ta, j, h, ag);
        __builtin_amdgcn_wave_barrier();
        rows_store(a.SG + (size_t)l * a.plane + off0, lane, hi, rows_from_lds(ta, lane));
      }
      // this wave no longer reads layer l's weights; the last one to say so
      // refills their buffer with layer l + 2
      {
        int old = 0;
        if (lane == 0)
          old = __hip_atomic_fetch_add(s_done + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == nactive - 1 && l + 2 < L) {
          issue_weights(l + 2, 0, 1);
          publish = l + 2;
        }
      }
      SSTAMP(l, 8);
    }
    __syncthreads();   // every wave is through the last layer: wl is free again
  }
#ifdef STACK_STAMPS
  if (tid == 0) { cal[2] = __builtin_amdgcn_s_memrealtime(); cal[3] = __builtin_amdgcn_s_memtime(); }
#endif
  // the last workgroup to leave re-arms the control block for the next launch
  if (tid == 0) {
    const unsigned done = atomicAdd(a.ctl + 1, 1u);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.ctl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 2, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------
// Backward of the whole residual stack in one launch: layer_bwd2d_kernel
// (wn_layer.hip; the hand-written gradient of model.py:236-330) wrapped in a
// layer loop l = L-1 .. 0 with the same hand-over scheme as the forward:
// DX[l] = dL/dx_l goes to memory with sc1 stores, flag[l][tile] = epoch
// publishes it, the rows t+d of DX[l+1] (the anti-causal tap) are read with sc1
// loads after their owners' flags.  A workgroup keeps a GROUP of WAVES x tpw
// consecutive tiles for all layers; groups are handed out in DECREASING order
// (a group depends on rows of the same or HIGHER groups), so the highest
// unfinished group can always finish whatever the residency.
//
// No workgroup barrier inside the stack (round 3; round 2 had four per layer):
//   * weights: a two-layer LDS ring of dense, XOR-swizzled images (20 KiB
//     each); the wave that is LAST through layer l's tiles refills that half
//     with layer l-2 and marks it ready once its DMA has landed;
//   * the weight gradients of a layer are summed over the workgroup's waves by
//     an ORDERED accumulation through one LDS slab: wave 0 writes its sums,
//     wave w adds its own after wave w-1's token, the last wave adds and
//     stores the slab to memory -- a fixed order (bitwise reproducible
//     whichever workgroup ran the group, whatever the timing) in which nobody
//     waits for a slower wave except its successors in that chain; the waves
//     settle into a stagger of one accumulation (~0.4 us) each and drift up to
//     a layer apart, so one wave's flag / load / LDS latencies hide under the
//     MFMAs of its SIMD partner instead of lining up behind a barrier;
//   * three 4-KiB LDS tiles per wave instead of four (the ring and the slab
//     need the room): the rows t+d pass through them before the rows-t DMA is
//     issued, the sigmoid rows travel in registers, and the weight-gradient
//     products run as two halves (da_f, then da_g through the same tile).
// One slab per (layer, group).
// ---------------------------------------------------------------------------
#define SB_WIMG 5120                      // dense swizzled backward image (floats)
// cache policy of a wave's own dx rows in the push formulation (store, load):
// 0 = plain, 1 = sc0 (past the vector L1), 16 = sc1 (device scope)
// (push) planes read exactly once per launch (z, dZ, sigmoid): 2 = nt hint
#ifndef SB_STREAM
#define SB_STREAM 2     // (1591 -> 1564 us)
#endif
#ifndef SB_OWN_ST
#define SB_OWN_ST 0
#endif
#ifndef SB_OWN_LD
#define SB_OWN_LD 1
#endif
// cache policy of the x[t] / x[t-d] tile loads (weight-gradient operands):
// 2 = nt measured 1621 -> 1595 us in the isolated launch A/B (tools/stack_ab.py)
// and 1493 -> 1502 us inside the training step (three interleaved bench.py
// runs per build on one box): not kept
#ifndef SB_X_AUX
#define SB_X_AUX 0
#endif

struct StackBwd {
  const float* X;      // [L][N][32]
  const float* Z;
  const float* SG;
  const float* dZ;
  float* DX;           // DX + l * dx_stride = dL/dx_l (without the tap's term, l > 0)
  long dx_stride;      // floats between the dx planes of two layers; 0 = ONE plane,
                       // rewritten in place from layer to layer (a tile's own rows
                       // have no other reader: they stay in the L2 / Infinity Cache)
  float* Q;            // [L][N][32]  q_l[s] = da_l[s] W[0]^T, the term row s sends to row s - d
  const float* wimg;   // [L][STACK_WBUF] backward weight images (wn_stack_pack)
  float* slabs;        // [L][>= groups][LAYER_BLOCK_FLOATS]
  long slab_layer_stride;
  float* tilesum;      // [L][ntiles][64] per-tile column sums of da, or null
  const int* dil;
  unsigned* flags;     // [L][ntiles]
  unsigned* ctl;       // as StackFwd
  float* poison;
  int L, B, T, tpw;    // tpw: tiles per wave and layer
  long plane;
#ifdef STACK_STAMPS
  unsigned long long* dbg;   // [grid][WAVES][L][16] + [grid][4]
#endif
};

// bounded spin until the workgroup-local LDS word reaches `want`
__device__ __forceinline__ void wait_lds_ge(const int* p, int want, bool& dead,
                                            unsigned* ctl, float* poison, int lane) {
  unsigned spins = 0;
  while (!dead && __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 24)) {
      dead = true;
      if (lane == 0) {
        __hip_atomic_store(ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (poison) *poison = __builtin_nanf("");
      }
    }
  }
  asm volatile("" ::: "memory");
}

// (timing-only ablations of the backward kernel, results meaningless:
// -DSB_NOGATE skips the gate derivatives (2143 -> 1761 us: the 2 x 16 gate
// evaluations per lane and tile sit on every tile's dependent path),
// -DSB_FAKE_ADDR loads every tile from one L2-resident megabyte (2126 -> 2011
// us: the kernel is not bound by memory))
#define sb_mfma(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// acc^T[i, time] += sum_k A[i][k] * frag[time, k] from a dense swizzled image:
// row i of a [32][32] matrix holds its eight 16-byte chunks at chunk ^ ((i >> 1)
// & 7), so the sixteen lanes of a ds_read_b128 beat (i & 15 all different)
// cover all 64 banks.  `wm` = matrix base + this lane's byte-swizzled offset of
// chunk h (q = 0): chunk 2q + h sits at float offset  off0 ^ (q << 3).
// The four 16-byte operand reads of a matrix are issued up front (one LDS
// latency per 16 MFMAs; read-wait-4 MFMAs per chunk exposed it four times).
__device__ __forceinline__ void mma32s(f32x16& acc, const f32x16& frag,
                                       const float* mat, int off0) {
  // chunk q + 1 is requested before the MFMAs of chunk q
  f32x4 nxt = *reinterpret_cast<const f32x4*>(mat + off0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 a4 = nxt;
    if (q < 3) nxt = *reinterpret_cast<const f32x4*>(mat + (off0 ^ ((q + 1) << 3)));
    // (order pinned: left alone the compiler moves the read behind the MFMAs
    // and waits for it there at some call sites; B = 8 1650 -> 1637 us,
    // B = 1 685 -> 660 us)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = sb_mfma(a4[e], frag[4 * q + e], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the same for a matrix whose first chunk is already in `pre`; the first chunk
// of the NEXT matrix is requested before this one's last MFMAs
__device__ __forceinline__ void mma32s_chain(f32x16& acc, const f32x16& frag,
                                             const float* mat, int off0, f32x4& pre,
                                             const float* next_mat) {
  f32x4 nxt = pre;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 a4 = nxt;
    nxt = *reinterpret_cast<const f32x4*>(q < 3 ? mat + (off0 ^ ((q + 1) << 3))
                                                : next_mat + off0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = sb_mfma(a4[e], frag[4 * q + e], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  pre = nxt;
}

// Transposed products from LDS tiles (weight gradients): acc0 += A0^T B,
// acc1 += A1^T B over the 32 rows of the tiles, bsum += column sums of B.
// Element [row = 2 s + h][channel = lane j] of the wave's tile T (swizzled
// layout of rows_to_lds) is  te[s & 3][T * 1024 + 64 * s]:  four per-lane
// pointers (the four values of row & 7 a lane meets) serve all tiles and
// steps with immediate offsets (the XOR swizzle written out per access costs
// about 350 address instructions per tile).
struct TileElemPtr {
  const float* p[4];
};
__device__ __forceinline__ TileElemPtr tile_elem_ptrs(const float* t0, int j, int h) {
  TileElemPtr te;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int off = 32 * h + ((((j >> 2) ^ (2 * k + h)) & 7) << 2) + (j & 3);
    asm volatile("" : "+v"(off));
    te.p[k] = t0 + off;
  }
  return te;
}

// ---- tile access through buffer resources: the plane base sits in four
// SGPRs, the tile offset in one SGPR, the lane's part of the address in ONE
// VGPR for all planes (instead of a 64-bit VGPR pointer pair per plane and
// shift: the backward kernel has no registers to spare) ----
typedef __amdgpu_buffer_rsrc_t wn_rsrc_t;
__device__ __forceinline__ wn_rsrc_t plane_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}

// rows [lo, hi) of the 32-row tile at byte offset `soff` of the plane, row-
// contiguous in registers (rows_load's lane <-> row mapping); `vrow` =
// ((lane >> 3) * 32 + (lane & 7) * 4) * 4.  AUX 16 = sc1 (device scope).
template <int AUX>
__device__ __forceinline__ RowRegs rows_ld(wn_rsrc_t rs, int soff, int vrow,
                                           int lane, int lo, int hi) {
  RowRegs R;
  if (lo <= 0 && hi >= 32) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      R.v[c] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vrow + c * 1024, soff, AUX));
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 8 * c + (lane >> 3);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r >= lo && r < hi)
        v = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vrow + c * 1024, soff, AUX));
      R.v[c] = v;
    }
  }
  return R;
}

template <int AUX>
__device__ __forceinline__ void rows_st(wn_rsrc_t rs, int soff, int vrow, int lane,
                                        int hi, const RowRegs& R) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = 8 * c + (lane >> 3);
    if (r < hi)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, R.v[c]), rs,
                                             vrow + c * 1024, soff, AUX);
  }
}

// tile_dma through a buffer resource; `vswz` = the lane's swizzled source
// offset ((rr * 32 + (((lane & 7) ^ (rr & 7)) << 2)) * 4, rr = lane >> 3)
template <int AUX>
__device__ __forceinline__ void tile_dma_rs(float* lds_tile, wn_rsrc_t rs, int soff,
                                            int vswz, int lane, int lo, int hi) {
  if (lo <= 0 && hi >= 32) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wn_lptr_t)(lds_tile + c * 256), 16,
                                               vswz, soff + c * 1024, 0, AUX);
  } else {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<f32x4*>(lds_tile + c * 256 + lane * 4) = zero;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 8 * c + (lane >> 3);
      // (the SGPR offset is added unsigned and only the VGPR offset is range
      // checked: a negative start -- the tap begins before the first clip --
      // goes into the lanes' offsets, where exactly the masked-off rows end up
      // below zero)
      const int sc = soff + c * 1024;
      if (r >= lo && r < hi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wn_lptr_t)(lds_tile + c * 256), 16,
                                                 vswz + min(sc, 0), max(sc, 0), 0, AUX);
    }
  }
}

// "Push" formulation (round 3): dx_l[t] = own_l[t] + q_l[t + d] with
//   own_l[t] = dx_{l+1}[t] + da_l[t] W[1]^T   and   q_l[s] = da_l[s] W[0]^T.
// A tile computes da for ITS OWN rows only and publishes q (what its rows
// contribute to the rows d earlier) instead of every tile re-reading four
// planes at rows t + d and re-deriving da there (the first, "pull" formulation,
// layer_bwd2d_kernel's arithmetic: 1689 -> 1642 us at B = 8 when it was
// replaced; removed in round 5): per tile and layer one gate evaluation
// instead of two, 160 instead of 176 MFMAs, 9 instead of 11 plane passes
// through memory, and the hand-over is published in the middle of a tile
// (before its weight-gradient products) instead of at its end.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stack_bwd_kernel(StackBwd a) {
  constexpr int SLAB = WAVES > 1 ? LAYER_BLOCK_FLOATS : 16;   // ordered-accumulation slab
  __shared__ __attribute__((aligned(1024))) float wl[2 * SB_WIMG];
  __shared__ __attribute__((aligned(1024))) float tiles[WAVES * 3072];
  __shared__ __attribute__((aligned(16))) float slab[SLAB];
  __shared__ int s_group;
  // per layer: waves through its tiles / its weights are in LDS
  __shared__ int s_done[STACK_MAXL], s_ready[STACK_MAXL];
  // accumulation tokens, one per (layer parity, matrix): (L - l) * 16 + the
  // number of waves that have added their part of that matrix
  __shared__ int s_tok[2][8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int vrow = ((lane >> 3) * 32 + (lane & 7) * 4) * 4;
  const int vswz = ((lane >> 3) * 32 + (((lane & 7) ^ ((lane >> 3) & 7)) << 2)) * 4;
  float* t0 = tiles + wave * 3072;
  float* t1 = t0 + 1024;
  float* t2 = t1 + 1024;
  const TileElemPtr te = tile_elem_ptrs(t0, j, h);
  const int T = a.T, L = a.L;
  const int tiles_per_clip = (T + 31) >> 5;
  const int ntiles = tiles_per_clip * a.B;
  const int GS = WAVES * a.tpw;
  const int ngroups = (ntiles + GS - 1) / GS;
  const unsigned epoch =
      __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool dead = false;
#ifdef STACK_STAMPS
#define BSTAMP(l, i)                                                         \
  if (lane == 0)                                                             \
    a.dbg[(((size_t)blockIdx.x * 8 + wave) * L + (l)) * 16 + (i)] =          \
        __builtin_amdgcn_s_memtime()
  unsigned long long* cal = a.dbg + (size_t)gridDim.x * 8 * L * 16 + (size_t)blockIdx.x * 4;
  if (tid == 0) { cal[0] = __builtin_amdgcn_s_memrealtime(); cal[1] = __builtin_amdgcn_s_memtime(); }
#else
#define BSTAMP(l, i)
#endif

  // pieces [p0, p0 + step, ...) of layer l's 20 KiB image into its ring half
  auto issue_wimg = [&](int l, int p0, int step) {
    const float* src = a.wimg + (size_t)l * STACK_WBUF;
    float* dst = wl + (l & 1) * SB_WIMG;
    for (int p = p0; p < SB_WIMG / 256; p += step)
      __builtin_amdgcn_global_load_lds((wn_gptr_t)(src + p * 256 + lane * 4),
                                       (wn_lptr_t)(dst + p * 256), 16, 0, 0);
  };

  for (;;) {
    if (tid == 0) s_group = (int)atomicAdd(a.ctl, 1u);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_group);
    __syncthreads();
    if (ticket >= ngroups) break;
    const int g = ngroups - 1 - ticket;
    const int tbase = g * GS, tend = min(ntiles, tbase + GS);
    for (int i = tid; i < L; i += WAVES * 64) {
      s_done[i] = 0;
      s_ready[i] = i >= L - 2;
    }
    if (tid < 16) s_tok[tid >> 3][tid & 7] = 0;
    issue_wimg(L - 1, wave, WAVES);
    if (L > 1) issue_wimg(L - 2, wave, WAVES);
    WN_WAIT_VM0();
    __syncthreads();        // weights of the two top layers are in LDS

    unsigned nfv_push = epoch;    // flag value(s) requested ahead for the next tile
    // lanes 0 / 1: index of the (at most two) flags the rows t + dd of tile tl
    // wait for, -1 on the other lanes / when there is nothing to wait for
    auto flag_idx2 = [&](int tl, int dd) -> int {
      if (tl >= tend) return -1;
      const int b = tl / tiles_per_clip;
      const int tt = tl - b * tiles_per_clip;
      const int tt0 = tt * 32;
      const int hif = min(min(32, T - tt0), T - dd - tt0);
      if (hif <= 0) return -1;
      const int first = (tt0 + dd) >> 5, last = (tt0 + dd + hif - 1) >> 5;
      int idx = -1;
      if (lane == 0 && first != tt) idx = b * tiles_per_clip + first;
      if (lane == 1 && last != first && last != tt) idx = b * tiles_per_clip + last;
      return idx;
    };
    for (int l = L - 1; l >= 0; --l) {
      const int d = a.dil[l];
      const bool hx = l + 1 < L;              // a gradient flows into x_{l+1}
      const int dn = hx ? a.dil[l + 1] : 0;   // the tap distance of dx_{l+1}
      const wn_rsrc_t qin = plane_rsrc(a.Q + (size_t)(hx ? l + 1 : l) * a.plane);
      const wn_rsrc_t q_out = plane_rsrc(a.Q + (size_t)l * a.plane);
      const wn_rsrc_t x = plane_rsrc(a.X + (size_t)l * a.plane);
      const wn_rsrc_t z = plane_rsrc(a.Z + (size_t)l * a.plane);
      const wn_rsrc_t sg = plane_rsrc(a.SG + (size_t)l * a.plane);
      const wn_rsrc_t dZ = plane_rsrc(a.dZ + (size_t)l * a.plane);
      const wn_rsrc_t dxin = plane_rsrc(a.DX + (size_t)(hx ? l + 1 : l) * a.dx_stride);
      const wn_rsrc_t dx_out = plane_rsrc(a.DX + (size_t)l * a.dx_stride);
      const unsigned* fl_in = a.flags + (size_t)(hx ? l + 1 : l) * ntiles;
      unsigned* fl_out = a.flags + (size_t)l * ntiles;
      float* tile_colsum = a.tilesum ? a.tilesum + (size_t)l * ntiles * 64 : nullptr;
      BSTAMP(l, 0);
      wait_lds_ge(s_ready + l, 1, dead, a.ctl, a.poison, lane);
      BSTAMP(l, 1);
      f32x16 cf0 = frag_zero(), cf1 = frag_zero(), cg0 = frag_zero(),
             cg1 = frag_zero(), cd = frag_zero();
      float sf = 0.f, sgs = 0.f, sd = 0.f;
      RowRegs a0, a1, a3;
      BSTAMP(l, 2);
      // this lane's swizzled offset of chunk h in row j of a weight matrix
      int woff = (j * 32 + ((h ^ ((j >> 1) & 7)) << 2)) + (l & 1) * SB_WIMG;
      asm volatile("" : "+v"(woff));   // opaque: no hoisting of the weight reads
      const float* const wm = wl;
      for (int tile = tbase + wave; tile < tend; tile += WAVES) {
        if (tile == tbase + wave + WAVES) { BSTAMP(l, 7); }
        const int b = tile / tiles_per_clip;
        const int tt0 = (tile - b * tiles_per_clip) * 32;
        const int hi = min(32, T - tt0);
        const int lo_p = max(0, d - tt0);       // rows whose t-d tap exists
#ifdef SB_FAKE_ADDR
        const int off0r = (b * T + tt0) * (WN_CH * 4);
        const int off0 = off0r & 0xFFFFF;
#else
        const int off0 = (b * T + tt0) * (WN_CH * 4);       // bytes
#endif
        // ---- rows t (dx_{l+1}[t] is this wave's own store of the layer
        // above); the sigmoid rows travel in registers
        if (tile == tbase + wave) { BSTAMP(l, 11); }
        tile_dma_rs<SB_STREAM>(t0, z, off0, vswz, lane, 0, hi);
        tile_dma_rs<SB_STREAM>(t2, dZ, off0, vswz, lane, 0, hi);
        a3 = rows_ld<SB_STREAM>(sg, off0, vrow, lane, 0, hi);
        if (hx) {
          // complete dx_{l+1}[t] = own rows (this wave's store of the layer
          // above) + q_{l+1}[t + dn] of the tiles d rows later, whose flags were
          // requested during the previous tile (they published q in the
          // middle of their tile of the layer above: normally long set)
          // (own rows: this wave's own store of one layer ago, same CU, same
          // L2 -- no device-scope access needed, only past the vector L1)
          a0 = rows_ld<SB_OWN_LD>(dxin, off0, vrow, lane, 0, hi);
          const int hi_q = min(hi, T - dn - tt0);
          if (hi_q > 0) {
            const int idx = flag_idx2(tile, dn);
            if (__builtin_amdgcn_ballot_w64(idx >= 0 && nfv_push != epoch) != 0)
              wait_flags(fl_in, idx, epoch, a.ctl, a.poison, dead, lane);
          }
          a1 = rows_ld<16>(qin, off0 + dn * (WN_CH * 4), vrow, lane, 0, hi_q);
        }
        if (tile == tbase + wave) { BSTAMP(l, 12); }
        WN_WAIT_VM0();
        if (hx) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a0.v[c] += a1.v[c];
          rows_to_lds(t1, lane, a0);
          __builtin_amdgcn_wave_barrier();
        }
        if (tile == tbase + wave) { BSTAMP(l, 14); }
        if (hx) {                                    // dWd += z^T dx_{l+1}
          // (operands of step s + 1 requested before the MFMA of step s,
          // pinned: see wgrad2 below)
          float az = te.p[0][0 * 1024], bd = te.p[0][1 * 1024];
#pragma unroll 1
          for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int kn = (k + 1) & 3, o = 64 * kn + 256 * ((it + (k == 3)) & 3);
              const float naz = te.p[kn][0 * 1024 + o];
              const float nbd = te.p[kn][1 * 1024 + o];
              __builtin_amdgcn_sched_barrier(0);
              cd = sb_mfma(az, bd, cd);
              sd += bd;
              __builtin_amdgcn_sched_barrier(0);
              az = naz;
              bd = nbd;
            }
          }
        }
        f32x16 dx, dz, di, zz, ss;
        zz = frag_from_lds(t0, j, h);
        if (hx) di = frag_from_lds(t1, j, h);
        dz = frag_from_lds(t2, j, h);
        __builtin_amdgcn_wave_barrier();
        rows_to_lds(t2, lane, a3);
        __builtin_amdgcn_wave_barrier();
        ss = frag_from_lds(t2, j, h);
        WN_WAIT_LGKM0();
        if (tile == tbase + wave) { BSTAMP(l, 15); }
        // the x tiles: in flight during the rows-t math
        tile_dma_rs<SB_X_AUX>(t0, x, off0, vswz, lane, 0, hi);
#ifdef SB_FAKE_ADDR
        tile_dma_rs<SB_X_AUX>(t1, x, off0 + 4096, vswz, lane, lo_p, hi);
#else
        tile_dma_rs<SB_X_AUX>(t1, x, off0 - d * (WN_CH * 4), vswz, lane, lo_p, hi);
#endif
        f32x16 dg;
        {
          if (hx) {
            dx = di;
            mma32s(dz, di, wm + 4 * 1024, woff);    // dx_{l+1}[t] * Wd^T
          } else {
            dx = frag_zero();
          }
          f32x16 df;
#ifdef SB_NOGATE
          df = dz; dg = zz;
          asm volatile("" :: "v"(ss[0]));
#else
          gate_grad(dz, zz, ss, df, dg);
#endif
          // (Wg's first chunk requested during Wf's last MFMAs: 1637 -> 1622 us)
          f32x4 wpre = *reinterpret_cast<const f32x4*>(wm + 1 * 1024 + woff);
          mma32s_chain(dx, df, wm + 1 * 1024, woff, wpre, wm + 0 * 1024);  // da_f[t] * Wf[1]^T
          // q_l[t] = da[t] W[0]^T, then own_l[t]: both to memory through t2
          // before it takes da_f (a wave's DS operations run in order)
          f32x16 qf = frag_zero();
          mma32s_chain(qf, df, wm + 0 * 1024, woff, wpre, wm + 3 * 1024);  // da_f[t] * Wf[0]^T
          mma32s_chain(dx, dg, wm + 3 * 1024, woff, wpre, wm + 2 * 1024);  // da_g[t] * Wg[1]^T
          mma32s_chain(qf, dg, wm + 2 * 1024, woff, wpre, wm + 2 * 1024);  // da_g[t] * Wg[0]^T
          if (dead) { qf[0] = __builtin_nanf(""); dx[0] = __builtin_nanf(""); }
          frag_to_lds(t2, j, h, qf);
          __builtin_amdgcn_wave_barrier();
          rows_st<16>(q_out, off0, vrow, lane, hi, rows_from_lds(t2, lane));
          __builtin_amdgcn_wave_barrier();
          frag_to_lds(t2, j, h, dx);
          __builtin_amdgcn_wave_barrier();
          rows_st<SB_OWN_ST>(dx_out, off0, vrow, lane, hi, rows_from_lds(t2, lane));
          __builtin_amdgcn_wave_barrier();
          frag_to_lds(t2, j, h, df);                 // t2 now holds da_f[t]
        }
        if (tile == tbase + wave) { BSTAMP(l, 8); }
        WN_WAIT_VM0();                               // x tiles in, q and own rows stored
        if (lane == 0)
          __hip_atomic_store(fl_out + tile, epoch, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        {
          // the flags the NEXT tile of this wave will look at: requested now
          int ntile = tile + WAVES, nl = l;
          if (ntile >= tend) { ntile = tbase + wave; nl = l - 1; }
          nfv_push = epoch;
          if (nl + 1 < L) {
            const int nidx = flag_idx2(ntile, a.dil[nl + 1]);
            if (nidx >= 0)
              nfv_push = __hip_atomic_load(a.flags + (size_t)(nl + 1) * ntiles + nidx,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (tile == tbase + wave) { BSTAMP(l, 10); }
        __builtin_amdgcn_wave_barrier();
        float tsf = 0.f, tsg = 0.f;          // this tile's column sums of da[t]
        // c1 += x[t]^T da, c0 += x[t-d]^T da, ts += column sums of da over the
        // tile's 32 rows (16 steps of two rows).  The three operands of step
        // s + 1 are requested before the MFMAs of step s and the order is
        // pinned: left alone the compiler reads right before use and waits --
        // an LDS round trip per two MFMAs with one partner wave to cover it.
        // (The last step requests step 0 again: harmless, keeps the loop one
        // body.)
        auto wgrad2 = [&](f32x16& c1, f32x16& c0, float& ts) {
          float axc = te.p[0][0 * 1024], axp = te.p[0][1 * 1024], b = te.p[0][2 * 1024];
#pragma unroll 1
          for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int kn = (k + 1) & 3, o = 64 * kn + 256 * ((it + (k == 3)) & 3);
              const float naxc = te.p[kn][0 * 1024 + o];
              const float naxp = te.p[kn][1 * 1024 + o];
              const float nb = te.p[kn][2 * 1024 + o];
              __builtin_amdgcn_sched_barrier(0);
              c1 = sb_mfma(axc, b, c1);
              c0 = sb_mfma(axp, b, c0);
              ts += b;
              __builtin_amdgcn_sched_barrier(0);
              axc = naxc;
              axp = naxp;
              b = nb;
            }
          }
        };
        wgrad2(cf1, cf0, tsf);               // dWf[1] += x[t]^T da_f, dWf[0] += x[t-d]^T da_f
        __builtin_amdgcn_wave_barrier();
        frag_to_lds(t2, j, h, dg);           // ... then da_g[t] through the same tile
        __builtin_amdgcn_wave_barrier();
        wgrad2(cg1, cg0, tsg);
        sf += tsf;
        sgs += tsg;
        if (tile_colsum) {
          const float ca = tsf + __shfl_xor(tsf, 32), cb = tsg + __shfl_xor(tsg, 32);
          if (h == 0) {
            tile_colsum[(size_t)tile * 64 + j] = ca;
            tile_colsum[(size_t)tile * 64 + 32 + j] = cb;
          }
        }
        WN_WAIT_LGKM0();                     // the three tiles are free
        if (tile == tbase + wave) { BSTAMP(l, 13); }
        __builtin_amdgcn_wave_barrier();
      }
      BSTAMP(l, 3);
      // ---- this wave no longer reads layer l's weights; the last one to say
      // so refills their ring half with layer l - 2
      bool refill = false;
      {
        int old = 0;
        if (lane == 0)
          old = __hip_atomic_fetch_add(s_done + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == WAVES - 1 && l >= 2) {
          issue_wimg(l - 2, 0, 1);
          refill = true;
        }
      }
      // ---- weight-gradient slab of this (layer, group): ordered accumulation
      // over the waves (wave 0 writes, wave w adds after wave w - 1, the last
      // wave adds and stores to memory)
      sf += __shfl_xor(sf, 32);
      sgs += __shfl_xor(sgs, 32);
      sd += __shfl_xor(sd, 32);
      float* out = a.slabs + (size_t)l * a.slab_layer_stride +
                   (size_t)g * LAYER_BLOCK_FLOATS;
      // The chain runs matrix by matrix (one token each), so wave w + 1 adds
      // matrix m while wave w is at matrix m + 1: 8 + 5 - 1 steps per layer
      // instead of 8 x 5.  It is the one serial path of the workgroup: raised
      // priority while on it.
      const int tbase_l = (L - l) * 16;
      int* tok = s_tok[l & 1];
      const int* tok_up = s_tok[(l + 1) & 1];
      const int e0 = 4 * h * 32 + j;           // + (8 (r >> 2) + (r & 3)) * 32
      __builtin_amdgcn_s_setprio(2);
      BSTAMP(l, 4);
      // copy matrix m (+ its bias row) of the finished slab to memory
      // (the slab holds a matrix in the accumulators' own layout, lane-linear
      // 16-byte pieces [m][q][lane]: a chain step is 4 + 4 LDS instructions
      // instead of 16 + 16; only the copy to memory uses the parameter layout)
      const f32x4* slab4 = reinterpret_cast<const f32x4*>(slab) + lane;
      auto slab_out = [&](float* dst, int m) {
        f32x16 p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = slab4[(m * 4 + q) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) p[4 * q + e] = v[e];
        }
        float pb = 0.f;
        if (m >= 2 && h == 0) pb = slab[LAYER_W_FLOATS + (m - 2) * 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[m * 1024 + e0 + (8 * (r >> 2) + (r & 3)) * 32] = p[r];   // (nt hint: no change)
        if (m >= 2 && h == 0) dst[LAYER_W_FLOATS + (m - 2) * 32 + j] = pb;
      };
      // position of this wave in the accumulation order (0, 4, 1, 5, ... --
      // SIMD partners next to each other -- and a static priority for either
      // half of the waves were tried: +-1 %)
      const int cpos = wave;
      auto chain = [&](f32x16& c, int m, float& bsum) {
        if (WAVES > 1) {
          if (cpos == 0) {
            // The slab's matrix m still holds layer l + 1's finished sums once
            // the last wave has added its part: wave 0 -- which would wait for
            // the slab anyway, while the last wave is the one everybody waits
            // for -- copies them to memory, then starts layer l's sums.
            if (hx) {
              wait_lds_ge(tok_up + m, tbase_l - 16 + WAVES, dead, a.ctl, a.poison, lane);
              slab_out(out + a.slab_layer_stride, m);
            }
          } else {
            wait_lds_ge(tok + m, tbase_l + cpos, dead, a.ctl, a.poison, lane);
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = slab4[(m * 4 + q) * 64];
            if (m >= 2 && h == 0) bsum += slab[LAYER_W_FLOATS + (m - 2) * 32 + j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) c[4 * q + e] += v[q][e];
          }
        }
        if (WAVES == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            out[m * 1024 + e0 + (8 * (r >> 2) + (r & 3)) * 32] = c[r];
          if (m >= 2 && h == 0) out[LAYER_W_FLOATS + (m - 2) * 32 + j] = bsum;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            reinterpret_cast<f32x4*>(slab)[(m * 4 + q) * 64 + lane] =
                f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
          if (m >= 2 && h == 0) slab[LAYER_W_FLOATS + (m - 2) * 32 + j] = bsum;
          WN_WAIT_LGKM0();                   // this wave's slab reads / writes are done
          if (lane == 0)
            __hip_atomic_store(tok + m, tbase_l + cpos + 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      float nosum = 0.f;
      chain(cf0, 0, nosum);
      chain(cf1, 1, nosum);
      chain(cg0, 2, sf);       // (the three bias sums ride with matrices 2, 3, 4)
      chain(cg1, 3, sgs);
      chain(cd, 4, sd);
      __builtin_amdgcn_s_setprio(0);
      BSTAMP(l, 5);
      if (refill) {
        WN_WAIT_VM0();                       // layer l - 2's image has landed
        if (lane == 0)
          __hip_atomic_store(s_ready + l - 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      BSTAMP(l, 6);
    }
    {
      // dx_0[t] = own_0[t] + q_0[t + d_0]: completed in place for the causal
      // layer's weight gradient (nobody reads DX[0] through a flag)
      const int d0 = a.dil[0];
      const wn_rsrc_t dx0 = plane_rsrc(a.DX), q0 = plane_rsrc(a.Q);
      for (int tile = tbase + wave; tile < tend; tile += WAVES) {
        const int b = tile / tiles_per_clip;
        const int tt0 = (tile - b * tiles_per_clip) * 32;
        const int hi = min(32, T - tt0);
        const int hi_q = min(hi, T - d0 - tt0);
        const int off0 = (b * T + tt0) * (WN_CH * 4);
        RowRegs ro = rows_ld<SB_OWN_LD>(dx0, off0, vrow, lane, 0, hi);
        if (hi_q > 0) {
          const int idx = flag_idx2(tile, d0);
          const bool ahead = tile == tbase + wave;      // (only the first one was requested ahead)
          if (__builtin_amdgcn_ballot_w64(idx >= 0 && (!ahead || nfv_push != epoch)) != 0)
            wait_flags(a.flags, idx, epoch, a.ctl, a.poison, dead, lane);
          const RowRegs rq = rows_ld<16>(q0, off0 + d0 * (WN_CH * 4), vrow, lane, 0, hi_q);
          WN_WAIT_VM0();
#pragma unroll
          for (int c = 0; c < 4; ++c) ro.v[c] += rq.v[c];
        }
        WN_WAIT_VM0();
        if (dead) ro.v[0][0] = __builtin_nanf("");
        rows_st<0>(dx0, off0, vrow, lane, hi, ro);
      }
    }
    if (WAVES > 1 && wave == 0) {          // the bottom layer's finished slab (position 0 is wave 0 in every order)
      const int e0 = 4 * h * 32 + j;
      float* dst = a.slabs + (size_t)g * LAYER_BLOCK_FLOATS;
      for (int m = 0; m < 5; ++m) {
        wait_lds_ge(s_tok[0] + m, L * 16 + WAVES, dead, a.ctl, a.poison, lane);
        f32x16 p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = reinterpret_cast<const f32x4*>(slab)[(m * 4 + q) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) p[4 * q + e] = v[e];
        }
        float pb = 0.f;
        if (m >= 2 && h == 0) pb = slab[LAYER_W_FLOATS + (m - 2) * 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[m * 1024 + e0 + (8 * (r >> 2) + (r & 3)) * 32] = p[r];
        if (m >= 2 && h == 0) dst[LAYER_W_FLOATS + (m - 2) * 32 + j] = pb;
      }
    }
    __syncthreads();        // every wave is through layer 0: LDS is free again
  }
#ifdef STACK_STAMPS
  if (tid == 0) { cal[2] = __builtin_amdgcn_s_memrealtime(); cal[3] = __builtin_amdgcn_s_memtime(); }
#endif
  if (tid == 0) {
    const unsigned done = atomicAdd(a.ctl + 1, 1u);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.ctl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 2, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------
// Small batches: the same two launches on 16-ROW tiles.
// With at most a tile or two per SIMD the launches above are bound by ONE
// wave's dependent path through a layer (B = 1, T = 16000: 500 tiles on 1024
// SIMDs; stamps: forward 5.4 us, backward 12.9 us per layer, of which the
// wave's own MFMAs 2.1 / 4.4 us and its gate math and LDS transits most of the
// rest).  A 16-row tile halves all of that and doubles the waves:
//   * time-major products on v_mfma_f32_16x16x4_f32 (32 cycles): lane (jr =
//     lane & 15, g = lane >> 4) holds, of a [16 rows][32 channels] operand, the
//     two 16-byte pieces  [row jr][16 blk + 4 g .. + 3], blk = 0, 1  (F16).  A
//     K-step (blk, e) contracts the channels 16 blk + 4 g' + e, g' = 0..3, and
//     the accumulator of output block mb holds [row jr][16 mb + 4 g + e]: the
//     output IS the next product's B operand, as in the 32-row kernels.  A
//     lane's four A operands of (mb, blk) are one ds_read_b128 of the SAME
//     weight images (wn_stack_pack);
//   * that layout is also what a lane reads from / writes to a [N][32] plane
//     with one 16-byte access per piece (64 contiguous bytes per row and
//     instruction): time-major operands go registers <-> memory directly, no
//     LDS transit; LDS tiles only where an operand is needed with the CHANNEL
//     on the lane (the weight-gradient products of the backward, which stay on
//     v_mfma_f32_32x32x2_f32 over K = 16 rows, so that accumulators, ordered
//     accumulation and slabs are the 32-row kernel's);
//   * same hand-over scheme, flags per 16-row tile.
// The products sum their 32 terms in another grouping than the 32-row kernels
// (4 + 4 + ... instead of 2 + 2 + ...): results agree to rounding, not bitwise
// (tests/test_gpu_stack.py).
// ---------------------------------------------------------------------------
struct F16 {
  f32x4 v[2];
};
#ifdef STACK_STAMPS
// diagnostic build: [grid][16][L][16] s_memtime stamps + [grid][4] clock calibration
#define S16STAMP(l, i)                                                       \
  if (lane == 0)                                                             \
    a.dbg[(((size_t)blockIdx.x * 16 + wave) * L + (l)) * 16 + (i)] =         \
        __builtin_amdgcn_s_memtime()
#define S16CAL(k)                                                                        \
  if (tid == 0) {                                                                        \
    unsigned long long* cal_ = a.dbg + (size_t)gridDim.x * 16 * L * 16 + (size_t)blockIdx.x * 4; \
    cal_[k] = __builtin_amdgcn_s_memrealtime();                                          \
    cal_[k + 1] = __builtin_amdgcn_s_memtime();                                          \
  }
#else
#define S16STAMP(l, i)
#define S16CAL(k)
#endif
#ifdef STACK_STAMPS
#define P16STAMP(l, i)                                                       \
  if (lane == 0)                                                             \
    a.dbg[(((size_t)blockIdx.x * 16 + wave_raw) * L + (l)) * 16 + (i)] =     \
        __builtin_amdgcn_s_memtime()
#else
#define P16STAMP(l, i)
#endif
#define wn_mfma16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

__device__ __forceinline__ F16 f16_zero() {
  F16 f;
  f.v[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  f.v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  return f;
}

// pieces of plane row `row_bytes / 128` (`voff` = row_bytes + 16 g: the lane's
// byte offset of piece 0; < 2^31, host check), zero when !valid.  An invalid
// lane asks for an offset past the resource's 2^31 - 1 bytes: the buffer unit
// answers 0 (drops the store) -- no branch around the access, so the
// compiler's count of outstanding loads stays exact.
#define WN_BUF_OOB ((int)0x80000000u)
template <int AUX>
__device__ __forceinline__ F16 f16_ld(wn_rsrc_t rs, int voff, bool valid) {
  const int o = valid ? voff : WN_BUF_OOB;
  F16 f;
  f.v[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, AUX));
  f.v[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + 64, 0, AUX));
  return f;
}
template <int AUX>
__device__ __forceinline__ void f16_st(wn_rsrc_t rs, int voff, bool valid, const F16& f) {
  const int o = valid ? voff : WN_BUF_OOB;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f.v[0]), rs, o, 0, AUX);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f.v[1]), rs, o + 64, 0, AUX);
}

// accA^T += WA^T frag, accB^T += WB^T frag from the FORWARD image (rows
// padded to SF_LD): `wa` / `wb` = matrix + jr * SF_LD + 4 g.  Four independent
// accumulation chains, the eight operand reads up front.
__device__ __forceinline__ void mma16t2(F16& accA, F16& accB, const F16& frag,
                                        const float* wa, const float* wb) {
  f32x4 a[2][2], b[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      a[mb][blk] = *reinterpret_cast<const f32x4*>(wa + mb * 16 * SF_LD + blk * 16);
      b[mb][blk] = *reinterpret_cast<const f32x4*>(wb + mb * 16 * SF_LD + blk * 16);
    }
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      accA.v[0] = wn_mfma16(a[0][blk][e], frag.v[blk][e], accA.v[0]);
      accA.v[1] = wn_mfma16(a[1][blk][e], frag.v[blk][e], accA.v[1]);
      accB.v[0] = wn_mfma16(b[0][blk][e], frag.v[blk][e], accB.v[0]);
      accB.v[1] = wn_mfma16(b[1][blk][e], frag.v[blk][e], accB.v[1]);
    }
}
__device__ __forceinline__ void mma16t(F16& acc, const F16& frag, const float* wa) {
  f32x4 a[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
      a[mb][blk] = *reinterpret_cast<const f32x4*>(wa + mb * 16 * SF_LD + blk * 16);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc.v[0] = wn_mfma16(a[0][blk][e], frag.v[blk][e], acc.v[0]);
      acc.v[1] = wn_mfma16(a[1][blk][e], frag.v[blk][e], acc.v[1]);
    }
}

// time-major pieces <-> a swizzled [16][32] LDS tile (chunk c of row r at
// c ^ (r & 7)): `lt` = tile + jr * 32, `sw` = jr & 7
__device__ __forceinline__ F16 f16_from_lds(const float* lt, int g, int sw) {
  F16 f;
  f.v[0] = *reinterpret_cast<const f32x4*>(lt + ((g ^ sw) << 2));
  f.v[1] = *reinterpret_cast<const f32x4*>(lt + (((4 + g) ^ sw) << 2));
  return f;
}
__device__ __forceinline__ void f16_to_lds(float* lt, int g, int sw, const F16& f) {
  *reinterpret_cast<f32x4*>(lt + ((g ^ sw) << 2)) = f.v[0];
  *reinterpret_cast<f32x4*>(lt + (((4 + g) ^ sw) << 2)) = f.v[1];
}

// (Built, measured, removed: the rows a tile hands to its tap readers as
// 8-byte {value, epoch} words that the readers poll directly -- RCCL's LL idea,
// what took the persistent generator of wn_fastgen.hip from 43.6 to 34.4 us --
// instead of "drain the stores, post a flag" / "read the flag, read the rows".
// Here it is neutral: 234 vs 229 us at B = 1 (260 before a row's words were
// laid out so that one instruction moves 64 contiguous bytes per row).  Stamps:
// the tap's wait shrinks from 0.50 to 0.43 us and the x' hand-off from 0.47 to
// 0.35, but 4 KiB of words per tile and layer on top of the planes make every
// other memory phase slower (weight ring 0.17 -> 0.47 us, z / sigmoid stores
// 0.21 -> 0.48).)
// SKIP (round 5, small batches): the launch also computes h1 = relu(sum_l z_l Ws_l
// + sum_l bs_l).  At one 16-row tile per SIMD a layer period is 4.1 us of which
// the wave's own 80 MFMAs are 1.1 us, while the skip GEMM (194 us at B = 1) waits
// for the whole stack.  Here a workgroup has WAVES more PARTNER waves (one per
// SIMD beside a chain wave): every chain wave hands its tile's z of a layer over
// through LDS (its accumulator registers, lane for lane; SK_ZB layers deep),
// partner wave j adds z_l[16 x 32] Ws_l[32 x 128 j ..] for ALL tiles of the group
// -- 256 MFMAs per layer into 128 accumulator registers, the operand fragments
// streamed from memory SK_AHEAD n-tiles ahead in the order it consumes them
// (no LDS ring: two were built first, 2 half-stages and 4 quarter-stages by
// LDS-DMA; a refill took 2 us from "slot free" to "landed" and the waves waited
// 2.4 - 4 us a layer for it).  What it buys is bounded by the SIMD the two waves
// share: under the partner's MFMA stream the chain wave's vector instructions
// issue at a fraction of their rate (wn_common.h) and the layer period grows
// from 4.1 to 7.4 us -- the launch takes 390 us against 232 + 194 for stack and
// GEMM apart; a step at B = 1 goes from 1.68 to 1.61 ms.
template <int SAVE, int WAVES, bool SKIP = false>
__global__ __launch_bounds__((SKIP ? 2 : 1) * WAVES * 64) void stack_fwd16_kernel(StackFwd a) {
  __shared__ __attribute__((aligned(1024))) float wl[2 * STACK_WBUF];
  __shared__ int s_group;
  __shared__ int s_done[STACK_MAXL], s_ready[STACK_MAXL];
  // SKIP: [WAVES][SK_ZB][512] z hand-over tiles (dynamic)
  extern __shared__ __attribute__((aligned(1024))) float sk_lds[];
  __shared__ int s_zflag[WAVES], s_zdone[WAVES];   // layers published by chain slot t / consumed by partner j
  __shared__ int s_dil[STACK_MAXL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_raw = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (waves w and w + WAVES of a workgroup share a SIMD: measured -- pairing
  // 2 s with 2 s + 1 put two partner waves on one SIMD and cost 5 % of a step)
  const bool skipw = SKIP && wave_raw >= WAVES;          // a partner wave
  const int wave = skipw ? wave_raw - WAVES : wave_raw;  // the tile slot / the partner's column quarter
  float* ztile = sk_lds + wave * (SK_ZB * 512);          // [SK_ZB][512] of this slot
  const int jr = lane & 15, g = lane >> 4;
  const int T = a.T, L = a.L;
  const int tiles_per_clip = (T + 15) >> 4;
  const int ntiles = tiles_per_clip * a.B;
  const int ngroups = (ntiles + WAVES - 1) / WAVES;
  const unsigned epoch =
      __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool dead = false;
  S16CAL(0);
  for (int i = tid; i < L; i += (SKIP ? 2 : 1) * WAVES * 64) s_dil[i] = a.dil[i];
  // (visible after the first barrier of the group loop)

  auto issue_weights = [&](int l, int p0, int step) {
    const float* wb = a.wimg + (size_t)l * STACK_WBUF;
    float* dst = wl + (l & 1) * STACK_WBUF;
    for (int p = p0; p < STACK_WBUF / 256; p += step)
      __builtin_amdgcn_global_load_lds((wn_gptr_t)(wb + p * 256 + lane * 4),
                                       (wn_lptr_t)(dst + p * 256), 16, 0, 0);
  };

  for (;;) {
    if (tid == 0) s_group = (int)atomicAdd(a.ctl, 1u);
    __syncthreads();
    const int gi = __builtin_amdgcn_readfirstlane(s_group);
    __syncthreads();
    if (gi >= ngroups) break;
    const int tile = gi * WAVES + wave;
    const int nactive = min(WAVES, ntiles - gi * WAVES);
    const bool any = tile < ntiles;
    const int b = any ? tile / tiles_per_clip : 0;
    const int tt = any ? tile - b * tiles_per_clip : 0;
    const int t0 = tt * 16;
    const int hi = any ? min(16, T - t0) : 0;
    const bool mine = jr < hi;                       // this lane's row exists
    // the lane's byte offset of piece 0 of ITS row in a plane
    const int voff = ((b * T + t0 + jr) * WN_CH + 4 * g) * 4;

    for (int i = tid; i < L; i += (SKIP ? 2 : 1) * WAVES * 64) {
      s_done[i] = 0;
      s_ready[i] = i < 2;
    }
    if (SKIP && tid < WAVES) { s_zflag[tid] = 0; s_zdone[tid] = 0; }
    if (skipw) {
      // ------------------------------------------------ partner (skip) wave
      // operand stream: fragment pair n = l * SK_NT + nt of this wave, requested
      // SK_AHEAD n-tiles ahead (the weights do not depend on the data: the
      // stream runs across layers)
      const float* wsrc = a.skimg + (size_t)wave * SK_WAVE + lane * 4;
      const long lstride = 4L * SK_WAVE;
      const int nfr = L * SK_NT;
      auto frag_ld = [&](int n, f32x4 (&w)[2]) {
        const int nn = n < nfr ? n : nfr - 1;
        const float* pw = wsrc + (size_t)(nn / SK_NT) * lstride + (nn % SK_NT) * 512;
        w[0] = *reinterpret_cast<const f32x4*>(pw);
        w[1] = *reinterpret_cast<const f32x4*>(pw + 256);
      };
      f32x4 win[SK_AHEAD + 1][2];
#pragma unroll
      for (int n0 = 0; n0 < SK_AHEAD; ++n0) frag_ld(n0, win[n0]);
      f32x4 acc[WAVES][SK_NT];
#pragma unroll
      for (int t = 0; t < WAVES; ++t)
#pragma unroll
        for (int n = 0; n < SK_NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      __syncthreads();      // (the chain waves' barrier: their weights of layers 0, 1 are in LDS)
      const bool run = nactive > 0;
      for (int l = 0; run && l < L; ++l) {
        // the z of layer l of the group's tiles (the chain waves' accumulator
        // registers, lane for lane; a slot without a tile: zeros)
        P16STAMP(l, 0);
        F16 zf[WAVES];
#pragma unroll
        for (int t = 0; t < WAVES; ++t) {
          zf[t] = f16_zero();
          if (t < nactive) {
            wait_lds_ge(s_zflag + t, l + 1, dead, a.ctl, a.poison, lane);
            const float* zt = sk_lds + t * (SK_ZB * 512) + (l & (SK_ZB - 1)) * 512 + lane * 8;
            zf[t].v[0] = *reinterpret_cast<const f32x4*>(zt);
            zf[t].v[1] = *reinterpret_cast<const f32x4*>(zt + 4);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(s_zdone + wave, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        P16STAMP(l, 1);
#pragma unroll
        for (int nt = 0; nt < SK_NT; ++nt) {
          // (static slot of the rolling window: (l * 8 + nt) % 6 would not be a
          // compile-time register index; the window is rotated by copies)
          frag_ld(l * SK_NT + nt + SK_AHEAD, win[SK_AHEAD]);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SK_AHEAD) : "memory");
#pragma unroll
          for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int t = 0; t < WAVES; ++t)
              acc[t][nt] = wn_mfma16(win[0][q >> 2][q & 3], zf[t].v[q >> 2][q & 3], acc[t][nt]);
            // (a pause of 128 cycles per eight MFMAs: the chain wave on this SIMD
            // issues its vector instructions at a fraction of their rate under a
            // running MFMA stream, and it is the chain that sets the layer
            // period.  B = 1, ms per step: no pause 1.641; 384 cycles per 32
            // MFMAs 1.626; 64 / 128 / 192 / 256 per 8: 1.629 / 1.611 / 1.649 /
            // 1.692; 128 / 256 / 384 per 16: 1.619 / 1.626 / 1.666)
            if (q & 1) {
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_s_sleep(2);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int n0 = 0; n0 < SK_AHEAD; ++n0) {
            win[n0][0] = win[n0 + 1][0];
            win[n0][1] = win[n0 + 1][1];
          }
        }
        P16STAMP(l, 3);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // h1 rows of every tile of the group, columns 128 j ..: accumulator register
      // i of n-tile nt = column 128 j + 16 nt + 4 g + i, row jr of tile t
#pragma unroll
      for (int t = 0; t < WAVES; ++t) {
        const int tl = gi * WAVES + t;
        if (tl < ntiles) {
          const int bt = tl / tiles_per_clip, t0t = (tl - bt * tiles_per_clip) * 16;
          if (jr < min(16, T - t0t)) {
            float* orow = a.sk_out + ((size_t)bt * T + t0t + jr) * SK_S + 128 * wave + 4 * g;
#pragma unroll
            for (int nt = 0; nt < SK_NT; ++nt) {
              f32x4 v = acc[t][nt];
              if (a.sk_bsum)
                v += *reinterpret_cast<const f32x4*>(a.sk_bsum + 128 * wave + 16 * nt + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
              *reinterpret_cast<f32x4*>(orow + 16 * nt) = v;
            }
          }
        }
      }
      __syncthreads();      // (the chain waves' barrier at the group's end)
      continue;
    }
    // (the chain wave's instructions go first on the SIMD it shares with a partner)
    if (SKIP) __builtin_amdgcn_s_setprio(3);
    issue_weights(0, wave, WAVES);
    if (L > 1) issue_weights(1, wave, WAVES);
    F16 xc = f16_ld<16>(plane_rsrc(a.X), voff, mine);
    // filter | gate bias of the first layer (the next layer's is requested a
    // layer ahead)
    F16 bf = f16_zero(), bg = f16_zero();
    auto bias_request = [&](int l) {
      if (!a.bias) return;
      const float* bp = a.bias + (size_t)l * a.bias_layer_stride +
                        (size_t)b * a.bias_clip_stride + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        bf.v[mb] = *reinterpret_cast<const f32x4*>(bp + 16 * mb);
        bg.v[mb] = *reinterpret_cast<const f32x4*>(bp + 32 + 16 * mb);
      }
    };
    bias_request(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();        // weights of layers 0 and 1 are in LDS
    int publish = 0;

    for (int l = 0; any && l < L; ++l) {
      // (from LDS: the load from memory waited, with vmcnt(0), for the layer's
      // plane stores as well)
      const int d = s_dil[l];
      S16STAMP(l, 0);
      wait_lds(s_ready + l, dead, a.ctl, a.poison, lane);
      S16STAMP(l, 1);
      const wn_rsrc_t xl = plane_rsrc(a.X + (size_t)l * a.plane);
      int woff = jr * SF_LD + 4 * g + (l & 1) * STACK_WBUF;
      asm volatile("" : "+v"(woff));
      const float* wlane = wl + woff;
      F16 af = bf, ag = bg;
      if (l + 1 < L) bias_request(l + 1);
      // the dilated tap: rows t0-d .. t0-d+15 of x_l, owned by at most two
      // other tiles of this clip (d < 16: and by this one -- its own stores of
      // the layer before, read back like the others'); flags requested first,
      // looked at after the current-tap products
      const int lo_row = t0 - d;
      const bool tap = mine && t0 + jr - d >= 0;          // this lane's tap row exists
      int fidx = -1;
      unsigned fval = epoch;
      if (l > 0 && lo_row + 15 >= 0) {
        const int first = max(lo_row, 0) >> 4, last = (lo_row + 15) >> 4;
        if (lane == 0 && first != tt) fidx = b * tiles_per_clip + first;
        if (lane == 1 && last != first && last != tt) fidx = b * tiles_per_clip + last;
        if (fidx >= 0)
          fval = __hip_atomic_load(a.flags + (size_t)l * ntiles + fidx, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
      }
      mma16t2(af, ag, xc, wlane + 1 * SF_MT, wlane + 3 * SF_MT);   // Wf[1], Wg[1]: current tap
      S16STAMP(l, 2);
      if (__builtin_amdgcn_ballot_w64(fval != epoch) != 0)
        wait_flags(a.flags + (size_t)l * ntiles, fidx, epoch, a.ctl, a.poison, dead, lane);
      const F16 xp = f16_ld<16>(xl, voff - d * (WN_CH * 4), tap);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      S16STAMP(l, 3);
      if (publish) {
        // (the weight pieces issued at the end of the previous layer are older
        // than the loads just waited for)
        if (lane == 0)
          __hip_atomic_store(s_ready + publish, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        publish = 0;
      }
      mma16t2(af, ag, xp, wlane + 0 * SF_MT, wlane + 2 * SF_MT);   // Wf[0], Wg[0]: past tap
      S16STAMP(l, 4);
      F16 zz;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          af.v[mb][e] = wn_tanh(af.v[mb][e]);
          ag.v[mb][e] = wn_sigmoid(ag.v[mb][e]);
          zz.v[mb][e] = af.v[mb][e] * ag.v[mb][e];
        }
      S16STAMP(l, 5);
      if (SKIP) {
        // z to the partner wave (its hand-over tile of two layers ago is free
        // once the partner has it in registers)
        if (l >= SK_ZB) {
#pragma unroll
          for (int j = 0; j < WAVES; ++j)
            wait_lds_ge(s_zdone + j, l - SK_ZB + 1, dead, a.ctl, a.poison, lane);
        }
        float* zt = ztile + (l & (SK_ZB - 1)) * 512 + lane * 8;
        *reinterpret_cast<f32x4*>(zt) = zz.v[0];
        *reinterpret_cast<f32x4*>(zt + 4) = zz.v[1];
        if (lane == 0)
          __hip_atomic_store(s_zflag + wave, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (l + 1 < L) {
        const float* bd = wl + (l & 1) * STACK_WBUF + SF_OFF_BD + 4 * g;
        xc.v[0] += *reinterpret_cast<const f32x4*>(bd);
        xc.v[1] += *reinterpret_cast<const f32x4*>(bd + 16);
        mma16t(xc, zz, wlane + 4 * SF_MT);                          // Wd
        S16STAMP(l, 6);
        // x' first: it is what other waves wait for
        f16_st<16>(plane_rsrc(a.X + (size_t)(l + 1) * a.plane), voff, mine, xc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(a.flags + (size_t)(l + 1) * ntiles + tile, epoch,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      S16STAMP(l, 7);
      f16_st<0>(plane_rsrc(a.Z + (size_t)l * a.plane), voff, mine, zz);
      if (SAVE == 2) f16_st<0>(plane_rsrc(a.SG + (size_t)l * a.plane), voff, mine, ag);
      {
        int old = 0;
        if (lane == 0)
          old = __hip_atomic_fetch_add(s_done + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == nactive - 1 && l + 2 < L) {
          // (every wave's reads of this half have returned: each waited for
          // its LDS reads before the MFMAs that used them)
          issue_weights(l + 2, 0, 1);
          publish = l + 2;
        }
      }
      S16STAMP(l, 8);
    }
    __syncthreads();
  }
  S16CAL(2);
  if (tid == 0) {
    const unsigned done = atomicAdd(a.ctl + 1, 1u);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.ctl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 2, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// LDS-DMA of a [16][32] tile into the swizzled layout of rows_to_lds (see
// tile_dma_rs; two 1-KiB pieces)
template <int AUX>
__device__ __forceinline__ void tile_dma16_rs(float* lds_tile, wn_rsrc_t rs, int soff,
                                              int vswz, int lane, int lo, int hi) {
  if (lo <= 0 && hi >= 16) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wn_lptr_t)(lds_tile + c * 256), 16,
                                               vswz, soff + c * 1024, 0, AUX);
  } else {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c)
      *reinterpret_cast<f32x4*>(lds_tile + c * 256 + lane * 4) = zero;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int r = 8 * c + (lane >> 3);
      const int sc = soff + c * 1024;     // (a negative start goes into the lanes' offsets)
      if (r >= lo && r < hi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wn_lptr_t)(lds_tile + c * 256), 16,
                                                 vswz + min(sc, 0), max(sc, 0), 0, AUX);
    }
  }
}

// accA^T[cin] += A[cin][cout] frag[cout], accB likewise with matrix B, from the
// dense swizzled BACKWARD image: `off0` = jr * 32 + ((g ^ ((jr >> 1) & 7)) << 2)
// (chunk g of row jr; chunk 4 + g sits at off0 ^ 16, rows 16 + jr 512 floats on)
__device__ __forceinline__ void mma16s2(F16& accA, F16& accB, const F16& frag,
                                        const float* ma, const float* mb_, int off0) {
  // (the operands of K-half blk + 1 requested before the MFMAs of half blk:
  // 16 + 16 registers in flight instead of 32 up front)
  f32x4 a[2][2], b[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    a[mb][0] = *reinterpret_cast<const f32x4*>(ma + mb * 512 + off0);
    b[mb][0] = *reinterpret_cast<const f32x4*>(mb_ + mb * 512 + off0);
  }
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    if (blk == 0) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        a[mb][1] = *reinterpret_cast<const f32x4*>(ma + mb * 512 + (off0 ^ 16));
        b[mb][1] = *reinterpret_cast<const f32x4*>(mb_ + mb * 512 + (off0 ^ 16));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      accA.v[0] = wn_mfma16(a[0][blk][e], frag.v[blk][e], accA.v[0]);
      accA.v[1] = wn_mfma16(a[1][blk][e], frag.v[blk][e], accA.v[1]);
      accB.v[0] = wn_mfma16(b[0][blk][e], frag.v[blk][e], accB.v[0]);
      accB.v[1] = wn_mfma16(b[1][blk][e], frag.v[blk][e], accB.v[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void mma16s(F16& acc, const F16& frag, const float* ma, int off0) {
  f32x4 a[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
      a[mb][blk] = *reinterpret_cast<const f32x4*>(ma + mb * 512 + (off0 ^ (blk << 4)));
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc.v[0] = wn_mfma16(a[0][blk][e], frag.v[blk][e], acc.v[0]);
      acc.v[1] = wn_mfma16(a[1][blk][e], frag.v[blk][e], acc.v[1]);
    }
}

// The backward on 16-row tiles ("push" formulation, one tile per wave and
// layer).  Three 2-KiB LDS tiles per wave for the operands that are needed with
// the channel on the lane (weight-gradient products), used in the 32-row
// kernel's order: z | dx_{l+1} (dWd), then x[t] | x[t-d] | da_f, then da_g
// through the third tile.  108 KiB per 8-wave workgroup: a weight-gradient GEMM
// workgroup of the side stream still fits beside it on the CU (six tiles per
// wave, everything requested at the layer's top, filled the CU's LDS and the
// GEMMs beside a B = 1 backward took 934 instead of 731 us).  Weight ring,
// ordered accumulation, slabs, tickets, flags, bounded waits: as
// stack_bwd_kernel.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stack_bwd16_kernel(StackBwd a) {
  constexpr int SLAB = WAVES > 1 ? LAYER_BLOCK_FLOATS : 16;
  __shared__ __attribute__((aligned(1024))) float wl[2 * SB_WIMG];
  __shared__ __attribute__((aligned(1024))) float tiles[WAVES * 1536];
  __shared__ __attribute__((aligned(16))) float slab[SLAB];
  __shared__ int s_group;
  __shared__ int s_done[STACK_MAXL], s_ready[STACK_MAXL];
  __shared__ int s_tok[2][8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;        // 32 x 32 products: channel, row parity
  const int jr = lane & 15, g = lane >> 4;       // 16 x 16 products: row, piece
  const int sw = jr & 7;
  const int vswz = ((lane >> 3) * 32 + (((lane & 7) ^ ((lane >> 3) & 7)) << 2)) * 4;
  float* t0 = tiles + wave * 1536;               // z[t], then x[t]
  float* t1 = t0 + 512;                          // dx_{l+1}[t] (complete), then x[t - d]
  float* t2 = t0 + 1024;                         // da_f[t], then da_g[t]
  const TileElemPtr te = tile_elem_ptrs(t0, j, h);   // [row 2 s + h][channel j] of tile n: te.p[s & 3][n * 512 + 64 s]
  const int T = a.T, L = a.L;
  const int tiles_per_clip = (T + 15) >> 4;
  const int ntiles = tiles_per_clip * a.B;
  const int ngroups = (ntiles + WAVES - 1) / WAVES;
  const unsigned epoch =
      __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool dead = false;
  S16CAL(0);

  auto issue_wimg = [&](int l, int p0, int step) {
    const float* src = a.wimg + (size_t)l * STACK_WBUF;
    float* dst = wl + (l & 1) * SB_WIMG;
    for (int p = p0; p < SB_WIMG / 256; p += step)
      __builtin_amdgcn_global_load_lds((wn_gptr_t)(src + p * 256 + lane * 4),
                                       (wn_lptr_t)(dst + p * 256), 16, 0, 0);
  };

  for (;;) {
    if (tid == 0) s_group = (int)atomicAdd(a.ctl, 1u);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_group);
    __syncthreads();
    if (ticket >= ngroups) break;
    const int gi = ngroups - 1 - ticket;
    const int tile = gi * WAVES + wave;
    const bool any = tile < ntiles;
    const int b = any ? tile / tiles_per_clip : 0;
    const int tt = any ? tile - b * tiles_per_clip : 0;
    const int tt0 = tt * 16;
    const int hi = any ? min(16, T - tt0) : 0;
    const bool mine = jr < hi;
    const int off0 = (b * T + tt0) * (WN_CH * 4);            // bytes (< 2^31: host check)
    const int voff = off0 + (jr * WN_CH + 4 * g) * 4;        // the lane's piece 0 of its row
    for (int i = tid; i < L; i += WAVES * 64) {
      s_done[i] = 0;
      s_ready[i] = i >= L - 2;
    }
    if (tid < 16) s_tok[tid >> 3][tid & 7] = 0;
    issue_wimg(L - 1, wave, WAVES);
    if (L > 1) issue_wimg(L - 2, wave, WAVES);
    WN_WAIT_VM0();
    __syncthreads();

    // lanes 0 / 1: index of the (at most two) flags the rows t + dd of this
    // tile wait for, -1 on the other lanes / when there is nothing to wait for
    auto flag_idx2 = [&](int dd) -> int {
      if (!any) return -1;
      const int hif = min(hi, T - dd - tt0);
      if (hif <= 0) return -1;
      const int first = (tt0 + dd) >> 4, last = (tt0 + dd + hif - 1) >> 4;
      int idx = -1;
      if (lane == 0 && first != tt) idx = b * tiles_per_clip + first;
      if (lane == 1 && last != first && last != tt) idx = b * tiles_per_clip + last;
      return idx;
    };
    unsigned nfv_push = epoch;    // flag value(s) requested ahead for the next layer
    for (int l = L - 1; l >= 0; --l) {
      const int d = a.dil[l];
      const bool hx = l + 1 < L;
      const int dn = hx ? a.dil[l + 1] : 0;
      const wn_rsrc_t qin = plane_rsrc(a.Q + (size_t)(hx ? l + 1 : l) * a.plane);
      const wn_rsrc_t q_out = plane_rsrc(a.Q + (size_t)l * a.plane);
      const wn_rsrc_t x = plane_rsrc(a.X + (size_t)l * a.plane);
      const wn_rsrc_t z = plane_rsrc(a.Z + (size_t)l * a.plane);
      const wn_rsrc_t sg = plane_rsrc(a.SG + (size_t)l * a.plane);
      const wn_rsrc_t dZ = plane_rsrc(a.dZ + (size_t)l * a.plane);
      const wn_rsrc_t dxin = plane_rsrc(a.DX + (size_t)(hx ? l + 1 : l) * a.dx_stride);
      const wn_rsrc_t dx_out = plane_rsrc(a.DX + (size_t)l * a.dx_stride);
      const unsigned* fl_in = a.flags + (size_t)(hx ? l + 1 : l) * ntiles;
      unsigned* fl_out = a.flags + (size_t)l * ntiles;
      float* tile_colsum = a.tilesum ? a.tilesum + (size_t)l * ntiles * 64 : nullptr;
      S16STAMP(l, 0);
      wait_lds_ge(s_ready + l, 1, dead, a.ctl, a.poison, lane);
      S16STAMP(l, 1);
      f32x16 cf0 = frag_zero(), cf1 = frag_zero(), cg0 = frag_zero(),
             cg1 = frag_zero(), cd = frag_zero();
      float sf = 0.f, sgs = 0.f, sd = 0.f;
      int woff = jr * 32 + ((g ^ ((jr >> 1) & 7)) << 2) + (l & 1) * SB_WIMG;
      asm volatile("" : "+v"(woff));
      const float* const wm = wl;
      if (any) {
        const int lo_p = max(0, d - tt0);
        // ---- rows t: z through LDS (needed both ways), the others time-major
        tile_dma16_rs<SB_STREAM>(t0, z, off0, vswz, lane, 0, hi);
        F16 dz = f16_ld<SB_STREAM>(dZ, voff, mine);
        const F16 ss = f16_ld<SB_STREAM>(sg, voff, mine);
        F16 di = f16_zero();
        if (hx) {
          // dx_{l+1}[t] = own rows (this wave's store of the layer above: same
          // CU, past the vector L1) + q_{l+1}[t + dn] of the tiles dn rows later
          di = f16_ld<SB_OWN_LD>(dxin, voff, mine);
          const int hi_q = min(hi, T - dn - tt0);
          F16 qv = f16_zero();
          if (hi_q > 0) {
            const int idx = flag_idx2(dn);
            if (__builtin_amdgcn_ballot_w64(idx >= 0 && nfv_push != epoch) != 0)
              wait_flags(fl_in, idx, epoch, a.ctl, a.poison, dead, lane);
            qv = f16_ld<16>(qin, voff + dn * (WN_CH * 4), jr < hi_q);
          }
          S16STAMP(l, 2);
          WN_WAIT_VM0();
          S16STAMP(l, 3);
          di.v[0] += qv.v[0];
          di.v[1] += qv.v[1];
          f16_to_lds(t1 + jr * 32, g, sw, di);
          __builtin_amdgcn_wave_barrier();
          // dWd += z^T dx_{l+1} over the 16 rows (8 steps of two rows; operands
          // of step s + 1 requested before the MFMA of step s)
          float ez = te.p[0][0 * 512], ei = te.p[0][1 * 512];
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const int sn = (s + 1) & 7, o = 64 * sn;
            const float nz = te.p[sn & 3][0 * 512 + o], ni = te.p[sn & 3][1 * 512 + o];
            __builtin_amdgcn_sched_barrier(0);
            cd = sb_mfma(ez, ei, cd);
            sd += ei;
            __builtin_amdgcn_sched_barrier(0);
            ez = nz;
            ei = ni;
          }
        } else {
          WN_WAIT_VM0();
        }
        const F16 zz = f16_from_lds(t0 + jr * 32, g, sw);
        WN_WAIT_LGKM0();
        S16STAMP(l, 4);
        __builtin_amdgcn_wave_barrier();
        // the x tiles: in flight during the rows-t math
        tile_dma16_rs<SB_X_AUX>(t0, x, off0, vswz, lane, 0, hi);
        tile_dma16_rs<SB_X_AUX>(t1, x, off0 - d * (WN_CH * 4), vswz, lane, lo_p, hi);
        if (hx) mma16s(dz, di, wm + 4 * 1024, woff);            // + dx_{l+1}[t] Wd^T
        F16 df, dg;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (gate_grad of wn_common.h on this layout)
            const float sgm = ss.v[mb][e], zv = zz.v[mb][e];
            const float th = zv * __builtin_amdgcn_rcpf(sgm + 1e-30f);
            df.v[mb][e] = dz.v[mb][e] * __builtin_fmaf(-zv, th, sgm);
            dg.v[mb][e] = dz.v[mb][e] * __builtin_fmaf(-zv, sgm, zv);
          }
        f16_to_lds(t2 + jr * 32, g, sw, df);
        S16STAMP(l, 5);
        // own_l[t] = dx_{l+1}[t] + da[t] W[1]^T,  q_l[t] = da[t] W[0]^T
        F16 dx = di, qf = f16_zero();
        mma16s2(dx, qf, df, wm + 1 * 1024, wm + 0 * 1024, woff);   // da_f: Wf[1], Wf[0]
        mma16s2(dx, qf, dg, wm + 3 * 1024, wm + 2 * 1024, woff);   // da_g: Wg[1], Wg[0]
        S16STAMP(l, 6);
        if (dead) { qf.v[0][0] = __builtin_nanf(""); dx.v[0][0] = __builtin_nanf(""); }
        f16_st<16>(q_out, voff, mine, qf);
        f16_st<SB_OWN_ST>(dx_out, voff, mine, dx);
        WN_WAIT_VM0();                               // x tiles in, q and own rows stored
        if (lane == 0)
          __hip_atomic_store(fl_out + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the flags this tile will look at in the layer below: requested now
        // (after layer 0: by the completion of dx_0 below)
        nfv_push = epoch;
        {
          const int nidx = flag_idx2(d);
          if (nidx >= 0)
            nfv_push = __hip_atomic_load(fl_out + nidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        S16STAMP(l, 7);
        __builtin_amdgcn_wave_barrier();
        // ---- c1 += x[t]^T da, c0 += x[t-d]^T da, ts += column sums of da
        auto wgrad2 = [&](f32x16& c1, f32x16& c0, float& ts) {
          float exc = te.p[0][0 * 512], exp_ = te.p[0][1 * 512], eb = te.p[0][2 * 512];
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const int sn = (s + 1) & 7, o = 64 * sn;
            const float nxc = te.p[sn & 3][0 * 512 + o], nxp = te.p[sn & 3][1 * 512 + o],
                        nb = te.p[sn & 3][2 * 512 + o];
            __builtin_amdgcn_sched_barrier(0);
            c1 = sb_mfma(exc, eb, c1);
            c0 = sb_mfma(exp_, eb, c0);
            ts += eb;
            __builtin_amdgcn_sched_barrier(0);
            exc = nxc;
            exp_ = nxp;
            eb = nb;
          }
        };
        float tsf = 0.f, tsg = 0.f;
        wgrad2(cf1, cf0, tsf);
        __builtin_amdgcn_wave_barrier();
        f16_to_lds(t2 + jr * 32, g, sw, dg);   // ... then da_g[t] through the same tile
        __builtin_amdgcn_wave_barrier();
        wgrad2(cg1, cg0, tsg);
        sf += tsf;
        sgs += tsg;
        if (tile_colsum) {
          const float ca = tsf + __shfl_xor(tsf, 32), cb = tsg + __shfl_xor(tsg, 32);
          if (h == 0) {
            tile_colsum[(size_t)tile * 64 + j] = ca;
            tile_colsum[(size_t)tile * 64 + 32 + j] = cb;
          }
        }
        WN_WAIT_LGKM0();                     // the tiles are free for the next layer's DMA
        S16STAMP(l, 8);
      }
      // ---- this wave no longer reads layer l's weights; the last one to say
      // so refills their ring half with layer l - 2
      bool refill = false;
      {
        int old = 0;
        if (lane == 0)
          old = __hip_atomic_fetch_add(s_done + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == WAVES - 1 && l >= 2) {
          issue_wimg(l - 2, 0, 1);
          refill = true;
        }
      }
      // ---- ordered accumulation of the weight-gradient slab over the waves
      // (as stack_bwd_kernel)
      sf += __shfl_xor(sf, 32);
      sgs += __shfl_xor(sgs, 32);
      sd += __shfl_xor(sd, 32);
      float* out = a.slabs + (size_t)l * a.slab_layer_stride +
                   (size_t)gi * LAYER_BLOCK_FLOATS;
      const int tbase_l = (L - l) * 16;
      int* tok = s_tok[l & 1];
      const int* tok_up = s_tok[(l + 1) & 1];
      const int e0 = 4 * h * 32 + j;
      __builtin_amdgcn_s_setprio(2);
      S16STAMP(l, 9);
      const f32x4* slab4 = reinterpret_cast<const f32x4*>(slab) + lane;
      auto slab_out = [&](float* dst, int m) {
        f32x16 p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = slab4[(m * 4 + q) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) p[4 * q + e] = v[e];
        }
        float pb = 0.f;
        if (m >= 2 && h == 0) pb = slab[LAYER_W_FLOATS + (m - 2) * 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[m * 1024 + e0 + (8 * (r >> 2) + (r & 3)) * 32] = p[r];
        if (m >= 2 && h == 0) dst[LAYER_W_FLOATS + (m - 2) * 32 + j] = pb;
      };
      auto chain = [&](f32x16& c, int m, float& bsum) {
        if (WAVES > 1) {
          if (wave == 0) {
            if (hx) {
              wait_lds_ge(tok_up + m, tbase_l - 16 + WAVES, dead, a.ctl, a.poison, lane);
              slab_out(out + a.slab_layer_stride, m);
            }
          } else {
            wait_lds_ge(tok + m, tbase_l + wave, dead, a.ctl, a.poison, lane);
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = slab4[(m * 4 + q) * 64];
            if (m >= 2 && h == 0) bsum += slab[LAYER_W_FLOATS + (m - 2) * 32 + j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) c[4 * q + e] += v[q][e];
          }
        }
        if (WAVES == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            out[m * 1024 + e0 + (8 * (r >> 2) + (r & 3)) * 32] = c[r];
          if (m >= 2 && h == 0) out[LAYER_W_FLOATS + (m - 2) * 32 + j] = bsum;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            reinterpret_cast<f32x4*>(slab)[(m * 4 + q) * 64 + lane] =
                f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
          if (m >= 2 && h == 0) slab[LAYER_W_FLOATS + (m - 2) * 32 + j] = bsum;
          WN_WAIT_LGKM0();
          if (lane == 0)
            __hip_atomic_store(tok + m, tbase_l + wave + 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      float nosum = 0.f;
      chain(cf0, 0, nosum);
      chain(cf1, 1, nosum);
      chain(cg0, 2, sf);
      chain(cg1, 3, sgs);
      chain(cd, 4, sd);
      __builtin_amdgcn_s_setprio(0);
      S16STAMP(l, 10);
      if (refill) {
        WN_WAIT_VM0();
        if (lane == 0)
          __hip_atomic_store(s_ready + l - 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    // dx_0[t] = own_0[t] + q_0[t + d_0], completed in place
    if (any) {
      const int d0 = a.dil[0];
      const wn_rsrc_t dx0 = plane_rsrc(a.DX), q0 = plane_rsrc(a.Q);
      F16 ro = f16_ld<SB_OWN_LD>(dx0, voff, mine);
      const int hi_q = min(hi, T - d0 - tt0);
      if (hi_q > 0) {
        const int idx = flag_idx2(d0);
        if (__builtin_amdgcn_ballot_w64(idx >= 0 && nfv_push != epoch) != 0)
          wait_flags(a.flags, idx, epoch, a.ctl, a.poison, dead, lane);
        const F16 rq = f16_ld<16>(q0, voff + d0 * (WN_CH * 4), jr < hi_q);
        WN_WAIT_VM0();
        ro.v[0] += rq.v[0];
        ro.v[1] += rq.v[1];
      }
      WN_WAIT_VM0();
      if (dead) ro.v[0][0] = __builtin_nanf("");
      f16_st<0>(dx0, voff, mine, ro);
    }
    if (WAVES > 1 && wave == 0) {          // the bottom layer's finished slab
      const int e0 = 4 * h * 32 + j;
      float* dst = a.slabs + (size_t)gi * LAYER_BLOCK_FLOATS;
      for (int m = 0; m < 5; ++m) {
        wait_lds_ge(s_tok[0] + m, L * 16 + WAVES, dead, a.ctl, a.poison, lane);
        f32x16 p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = reinterpret_cast<const f32x4*>(slab)[(m * 4 + q) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) p[4 * q + e] = v[e];
        }
        float pb = 0.f;
        if (m >= 2 && h == 0) pb = slab[LAYER_W_FLOATS + (m - 2) * 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[m * 1024 + e0 + (8 * (r >> 2) + (r & 3)) * 32] = p[r];
        if (m >= 2 && h == 0) dst[LAYER_W_FLOATS + (m - 2) * 32 + j] = pb;
      }
    }
    __syncthreads();
  }
  S16CAL(2);
  if (tid == 0) {
    const unsigned done = atomicAdd(a.ctl + 1, 1u);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.ctl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ctl + 2, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#ifdef STACK_STAMPS
static unsigned long long* g_stack_dbg = nullptr;
static unsigned long long* g_stack_dbg_b = nullptr;
extern "C" int wn_diag_stack_dbg(unsigned long long* p) { g_stack_dbg = p; return WN_OK; }
extern "C" int wn_diag_stack_dbg_b(unsigned long long* p) { g_stack_dbg_b = p; return WN_OK; }
#endif

extern "C" {

// unsigned ints of `flags` for a (B, T, L) problem
// (one word per 16-row tile: enough for either tile height)
long wn_stack_flag_count(int B, int T, int L) {
  if (B <= 0 || T <= 0 || L <= 0) return 0;
  return (long)L * B * ((T + 15) / 16);
}

// Rows of a tile in wn_stack_fwd / wn_stack_bwd for a shape: 16 while the
// batch has at most four 32-row tiles per CU (the launches are then bound by
// one wave's dependent path through a layer: halve the path), else 32.
// Measured, T = 16000, ms per training step at 32 / 16 rows: B = 1 1.780 /
// 1.683, B = 2 2.999 / 2.947, B = 3 4.14 / 4.60, B = 4 5.01 / 5.56.
// `variant` (WN_STACK_VARIANT of wavenet_hip.h) may force 16 / 32 (A/B, tests).
int wn_stack_tile_rows(int B, int T, int variant) {
  if (B <= 0 || T <= 0) return 32;
  const int r = variant & 0x3f;
  if (r == 16 || r == 32) return r;
  const long ntiles = (long)B * ((T + 31) / 32);
  return ntiles <= 4L * wn_device_cus() ? 16 : 32;
}

// waves per workgroup a variant word asks for (bits 8..11), or `dflt`
static int variant_waves(int variant, int dflt, bool allow_small) {
  const int w = (variant >> 8) & 0xf;
  if (w == 4 || w == 8 || (allow_small && (w == 1 || w == 2))) return w;
  return dflt;
}

int wn_stack_wimg_floats(void) { return STACK_WBUF; }

int wn_stack_pack(const float* layer0, long layer_stride, float* wimg_fwd,
                  float* wimg_bwd, int L, void* stream) {
  if (!layer0 || (!wimg_fwd && !wimg_bwd)) return WN_ERR_NULL;
  if (L <= 0 || layer_stride < LAYER_BLOCK_FLOATS) return WN_ERR_BAD_SHAPE;
  if ((wimg_fwd && !wn_aligned16(wimg_fwd)) || (wimg_bwd && !wn_aligned16(wimg_bwd)))
    return WN_ERR_MISALIGNED;
  hipLaunchKernelGGL(stack_pack_kernel, dim3(L, 2), dim3(256), 0,
                     (hipStream_t)stream, layer0, layer_stride, wimg_fwd, wimg_bwd);
  return wn_check_launch();
}

// floats of the skip image wn_stack_skip_pack writes for wn_stack_fwd_skip
long wn_stack_skip_img_floats(int L) { return L > 0 ? (long)4 * L * SK_WAVE : 0; }

// 1 when wn_stack_fwd_skip covers the shape: the 16-row launch with one wave per
// SIMD (small batches), 512 skip channels
int wn_stack_fwd_skip_ok(int B, int T, int S, int variant) {
  if (B <= 0 || T <= 0 || S != SK_S) return 0;
  if (wn_stack_tile_rows(B, T, variant) != 16) return 0;
  const long nt16 = (long)B * ((T + 15) / 16);
  return variant_waves(variant, (nt16 + 3) / 4 <= wn_device_cus() ? 4 : 8, false) == 4;
}

int wn_stack_skip_pack(const float* skip_w, int L, float* img, void* stream) {
  if (!skip_w || !img) return WN_ERR_NULL;
  if (L <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(stack_skip_pack_kernel, dim3(4 * L), dim3(256), 0, (hipStream_t)stream,
                     skip_w, img);
  return wn_check_launch();
}

static int stack_fwd_launch(float* X, float* Z, float* SG, const float* wimg,
                 const float* bias, long bias_layer_stride,
                 int bias_clip_stride, const int* dilations, unsigned* flags,
                 unsigned* ctl, float* poison, int L, int B, int T, int save_sg,
                 int variant, void* stream, const float* skimg, const float* sk_bsum,
                 float* sk_out) {
  if (!X || !Z || !wimg || !dilations || !flags || !ctl) return WN_ERR_NULL;
  if (save_sg && !SG) return WN_ERR_NULL;
  if (L <= 0 || B <= 0 || T <= 0) return WN_ERR_BAD_SHAPE;
  if (!wn_aligned16(X) || !wn_aligned16(Z) || (SG && !wn_aligned16(SG)) ||
      !wn_aligned16(wimg))
    return WN_ERR_MISALIGNED;
  StackFwd a;
  a.X = X; a.Z = Z; a.SG = SG; a.wimg = wimg;
  a.bias = bias; a.bias_layer_stride = bias_layer_stride;
  a.bias_clip_stride = bias_clip_stride; a.dil = dilations; a.flags = flags;
  a.ctl = ctl; a.poison = poison; a.L = L; a.B = B; a.T = T;
  a.plane = (long)B * T * WN_CH;
  a.skimg = skimg; a.sk_bsum = sk_bsum; a.sk_out = sk_out;
#ifdef STACK_STAMPS
  if (!g_stack_dbg) return WN_ERR_NULL;
  a.dbg = g_stack_dbg;
#endif
  if (L > STACK_MAXL) return WN_ERR_UNSUPPORTED;
  if (skimg) {
    // the 16-row launch with a partner wave per tile (stack_fwd16_kernel<.., true>)
    if (!sk_out || !wn_aligned16(skimg) || !wn_aligned16(sk_out) ||
        (sk_bsum && !wn_aligned16(sk_bsum)))
      return sk_out ? WN_ERR_MISALIGNED : WN_ERR_NULL;
    if (!wn_stack_fwd_skip_ok(B, T, SK_S, variant)) return WN_ERR_UNSUPPORTED;
    if ((long)B * T * WN_CH * 4 >= (1L << 31)) return WN_ERR_UNSUPPORTED;
    const long nt16 = (long)B * ((T + 15) / 16);
    long g16 = (nt16 + 3) / 4;
    if (g16 > wn_device_cus()) g16 = wn_device_cus();
    const size_t dyn = (size_t)4 * SK_ZB * 512 * sizeof(float);
    const void* kf = save_sg ? reinterpret_cast<const void*>(stack_fwd16_kernel<2, 4, true>)
                             : reinterpret_cast<const void*>(stack_fwd16_kernel<0, 4, true>);
    if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
      return WN_ERR_LAUNCH;
    if (save_sg)
      hipLaunchKernelGGL((stack_fwd16_kernel<2, 4, true>), dim3((unsigned)g16), dim3(512), dyn,
                         (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((stack_fwd16_kernel<0, 4, true>), dim3((unsigned)g16), dim3(512), dyn,
                         (hipStream_t)stream, a);
    return wn_check_launch();
  }
  if (wn_stack_tile_rows(B, T, variant) == 16) {
    // (tile offsets inside a plane are 32-bit byte offsets of a buffer resource)
    if ((long)B * T * WN_CH * 4 >= (1L << 31)) return WN_ERR_UNSUPPORTED;
    const long nt16 = (long)B * ((T + 15) / 16);
    const int cus16 = wn_device_cus();
    // one wave per SIMD while that covers the batch in one pass, else two
    const int w16 = variant_waves(variant, (nt16 + 3) / 4 <= cus16 ? 4 : 8, false);
    long g16 = (nt16 + w16 - 1) / w16;
    if (g16 > cus16) g16 = cus16;
    dim3 grid16((unsigned)g16), block16(w16 * 64);
    hipStream_t s16 = (hipStream_t)stream;
#define LAUNCH16(W)                                                              \
  do {                                                                           \
    if (save_sg)                                                                 \
      hipLaunchKernelGGL((stack_fwd16_kernel<2, W>), grid16, block16, 0, s16, a);  \
    else                                                                         \
      hipLaunchKernelGGL((stack_fwd16_kernel<0, W>), grid16, block16, 0, s16, a);  \
  } while (0)
    if (w16 == 8) LAUNCH16(8); else LAUNCH16(4);
#undef LAUNCH16
    return wn_check_launch();
  }
  // waves (= tiles) per workgroup: 16 when that fills every CU, fewer for
  // small batches so that the tiles still spread over the whole chip
  const long ntiles = (long)B * ((T + 31) / 32);
  const int cus = wn_device_cus();
  // (a layer costs a workgroup about max(7 us of dependent latency, 2.2 us of
  // matrix-pipe time per wave of a SIMD); groups beyond one per CU run in passes)
  int waves = 16;
  long best = -1;
  for (int w = 16; w >= 1; w >>= 1) {
    const long groups = (ntiles + w - 1) / w;
    const long passes = (groups + cus - 1) / cus;
    const long per_layer = w * 22 / 4 > 70 ? w * 22 / 4 : 70;   // 0.1 us
    const long cost = passes * per_layer;
    if (best < 0 || cost <= best) { best = cost; waves = w; }   // ties: fewer waves per CU
  }
  long g = (ntiles + waves - 1) / waves;
  if (g > cus) g = cus;
  dim3 grid((unsigned)g), block(waves * 64);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(W)                                                             \
  do {                                                                        \
    if (save_sg)                                                              \
      hipLaunchKernelGGL((stack_fwd_kernel<2, W>), grid, block, 0, s, a);     \
    else                                                                      \
      hipLaunchKernelGGL((stack_fwd_kernel<0, W>), grid, block, 0, s, a);     \
  } while (0)
  switch (waves) {
    case 16: LAUNCH(16); break;
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(1); break;
  }
#undef LAUNCH
  return wn_check_launch();
}

int wn_stack_fwd(float* X, float* Z, float* SG, const float* wimg,
                 const float* bias, long bias_layer_stride,
                 int bias_clip_stride, const int* dilations, unsigned* flags,
                 unsigned* ctl, float* poison, int L, int B, int T, int save_sg,
                 int variant, void* stream) {
  return stack_fwd_launch(X, Z, SG, wimg, bias, bias_layer_stride, bias_clip_stride, dilations,
                          flags, ctl, poison, L, B, T, save_sg, variant, stream, nullptr,
                          nullptr, nullptr);
}

// wn_stack_fwd + the skip sum in the same launch: h1[N][512] = relu(sum_l z_l Ws_l
// + skip_bsum) (wn_stack_fwd_skip_ok shapes; skip_img from wn_stack_skip_pack)
int wn_stack_fwd_skip(float* X, float* Z, float* SG, const float* wimg,
                      const float* bias, long bias_layer_stride,
                      int bias_clip_stride, const int* dilations, unsigned* flags,
                      unsigned* ctl, float* poison, int L, int B, int T, int save_sg,
                      int variant, const float* skip_img, const float* skip_bsum,
                      float* h1, void* stream) {
  if (!skip_img || !h1) return WN_ERR_NULL;
  return stack_fwd_launch(X, Z, SG, wimg, bias, bias_layer_stride, bias_clip_stride, dilations,
                          flags, ctl, poison, L, B, T, save_sg, variant, stream, skip_img,
                          skip_bsum, h1);
}

// (waves per workgroup, tiles per wave and layer) of wn_stack_bwd for a shape:
// big batches keep 8 waves and give every wave enough tiles that one pass of
// at most one group per CU covers the batch; small batches use fewer waves so
// that the tiles spread over the whole chip
static void stack_bwd_shape(int B, int T, int variant, int* waves_out, int* tpw_out) {
  if (wn_stack_tile_rows(B, T, variant) == 16) {     // 16-row tiles: one per wave
    *waves_out = variant_waves(variant, 8, false);
    *tpw_out = 1;
    return;
  }
  const long ntiles = (long)B * ((T + 31) / 32);
  const int cus = wn_device_cus();
  // experiments (A/B of co-residency with a side-stream GEMM, stamps): the
  // variant word may force the waves per workgroup
  if (const int w = variant_waves(variant, 0, true)) {
    *waves_out = w;
    *tpw_out = (int)((ntiles + (long)w * cus - 1) / ((long)w * cus));
    return;
  }
  long best = -1;
  int bw = 8, bt = 1;
  const int tmin = (int)((ntiles + 8L * cus - 1) / (8L * cus));
  for (int w = 8; w >= 1; w >>= 1) {
    for (int t = 1; t <= (w == 8 ? tmin + 1 : 1); ++t) {
      const long groups = (ntiles + (long)w * t - 1) / ((long)w * t);
      const long passes = (groups + cus - 1) / cus;
      const long cost = passes * (t * (w == 8 ? 144 : 100) + 70);   // 0.1 us per layer
      // (ties: MORE waves per workgroup = fewer weight-gradient slabs to write
      // and reduce; at B = 1 that is 125 instead of 250 per layer, 130 MB)
      if (best < 0 || cost < best) { best = cost; bw = w; bt = t; }
    }
  }
  *waves_out = bw;
  *tpw_out = bt;
}

// weight-gradient slabs per layer wn_stack_bwd writes (= tile groups)
int wn_stack_bwd_slabs(int B, int T, int variant) {
  if (B <= 0 || T <= 0) return 0;
  int w, t;
  stack_bwd_shape(B, T, variant, &w, &t);
  const int rows = wn_stack_tile_rows(B, T, variant);
  const long ntiles = (long)B * ((T + rows - 1) / rows);
  return (int)((ntiles + (long)w * t - 1) / ((long)w * t));
}

int wn_stack_bwd(const float* X, const float* Z, const float* SG,
                 const float* dZ, float* DX, long dx_layer_stride, float* Q,
                 const float* wimg, float* slabs,
                 long slab_layer_stride, float* tilesum, const int* dilations,
                 unsigned* flags, unsigned* ctl, float* poison, int L, int B,
                 int T, int variant, void* stream) {
  if (!X || !Z || !SG || !dZ || !DX || !Q || !wimg || !slabs || !dilations ||
      !flags || !ctl)
    return WN_ERR_NULL;
  if (L <= 0 || B <= 0 || T <= 0) return WN_ERR_BAD_SHAPE;
  if (L > STACK_MAXL) return WN_ERR_UNSUPPORTED;
  // (tile offsets inside a plane are 32-bit byte offsets of a buffer resource)
  if ((long)B * T * WN_CH * 4 >= (1L << 31)) return WN_ERR_UNSUPPORTED;
  const void* ptrs[] = {X, Z, SG, dZ, DX, wimg};
  for (const void* p : ptrs)
    if (!wn_aligned16(p)) return WN_ERR_MISALIGNED;
  if (!wn_aligned16(Q)) return WN_ERR_MISALIGNED;
  int waves, tpw;
  stack_bwd_shape(B, T, variant, &waves, &tpw);
  const int rows = wn_stack_tile_rows(B, T, variant);
  const long ntiles = (long)B * ((T + rows - 1) / rows);
  const long groups = (ntiles + (long)waves * tpw - 1) / ((long)waves * tpw);
  if (slab_layer_stride < groups * LAYER_BLOCK_FLOATS) return WN_ERR_BAD_SHAPE;
  StackBwd a;
  a.X = X; a.Z = Z; a.SG = SG; a.dZ = dZ; a.DX = DX; a.Q = Q; a.wimg = wimg;
  a.dx_stride = dx_layer_stride;
  a.slabs = slabs; a.slab_layer_stride = slab_layer_stride; a.tilesum = tilesum;
  a.dil = dilations; a.flags = flags; a.ctl = ctl; a.poison = poison;
  a.L = L; a.B = B; a.T = T;
  a.tpw = tpw; a.plane = (long)B * T * WN_CH;
#ifdef STACK_STAMPS
  if (!g_stack_dbg_b) return WN_ERR_NULL;
  a.dbg = g_stack_dbg_b;
#endif
  const int cus = wn_device_cus();
  dim3 grid((unsigned)(groups < cus ? groups : cus)), block(waves * 64);
  hipStream_t s = (hipStream_t)stream;
  // one dx plane per layer, or (nobody but the owner reads a tile's dx rows)
  // a single plane rewritten in place
  if (dx_layer_stride != (long)B * T * WN_CH && dx_layer_stride != 0)
    return WN_ERR_BAD_SHAPE;
  if (rows == 16) {
    if (waves == 8) hipLaunchKernelGGL((stack_bwd16_kernel<8>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((stack_bwd16_kernel<4>), grid, block, 0, s, a);
    return wn_check_launch();
  }
#define LAUNCH(W) hipLaunchKernelGGL((stack_bwd_kernel<W>), grid, block, 0, s, a)
  switch (waves) {
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(1); break;
  }
#undef LAUNCH
  return wn_check_launch();
}

}  // extern "C"
