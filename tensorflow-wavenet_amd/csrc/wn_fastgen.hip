// Fast generation: the incremental generator of WaveNetModel
// (_generator_conv / _generator_causal_layer / _generator_dilation_layer /
// _create_generator / predict_proba_incremental, wavenet/model.py:332-387,
// 444-516, 592-626) and the per-sample host loop of generate.py:213-241
// (one sess.run per audio sample, temperature rescale, np.random.choice) as
// ONE persistent kernel: FIFO queues live in device memory as ring buffers,
// the next-sample distribution is drawn on the device, and no host round trip
// happens between samples.
//
// Round-1 structure ("single CU"): one workgroup of 5 waves.
//   wave 0      : the serial residual chain (50 x [64x64 + 32x32] mat-vec),
//                 one conv output per lane, inputs broadcast by v_readlane;
//   waves 1..4  : skip accumulation total += z_l * Ws_l (the 3.3 MB weight
//                 stream) one layer behind the chain, then the two
//                 post-processing mat-vecs.
// One workgroup barrier per layer.  Replicas only across GPUs (the chain is
// serial); see DESIGN.md for the multi-CU pipeline planned next.
//
// Reference quirks kept: the generator ignores residual_postproc
// (model.py:511 vs 436-437); queues start as zeros (net.init_ops).
#include "wn_common.h"

#define FG_THREADS 256   // wave 0: chain; waves 1..3: skip / post (192 threads)
#define FG_SKT 192       // skip threads
#define FG_SKO 3         // skip outputs per skip thread (3 * 192 >= FG_MAXS)
#define FG_MAXS 512
#define FG_MAXQ 512

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// v_readlane_b32 on a float (the builtin is typed int: bit-cast, never convert)
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

#define FG_MAXL 64

struct FastGen {
  const float* causal;   // [2][Q][32]
  const float* layer0;   // layer blocks
  long layer_stride;
  const float* skip_w;   // [L][32][S]
  const float* skip_bsum;  // [S] sum over layers of skip biases, or null
  const float* post1_w;  // [S][S]
  const float* post1_b;  // [S] or null
  const float* post2_w;  // [S][Q]
  const float* post2_b;  // [Q] or null
  const float* bias_fg;  // [L][64] filter|gate bias (+gc) or null
  int L, S, Q;
  float* state;          // ring buffers, layer l at state + roff[l]*32
  int32_t* cursors;      // [0] steps done so far, [1] previous code (-1: none)
  int32_t* samples;      // [n_steps + 1]
  int n_given, n_steps;
  float temperature;
  uint64_t seed;
  float* proba_out;      // [ceil(n_steps/proba_every)][Q] or null
  int proba_every;
  int use_dense_bias;
  int push;              // 0: peek (do not advance the queues), like running
                         // the reference's proba op without net.push_ops
  const int32_t* dil;    // [L] device
};

// Roles inside the workgroup (a producer / consumer split, one barrier per
// layer):
//   waves 1..3 ("loaders"): stream the NEXT layers' weights -- the 20 KB chain
//     block through registers into a 2-deep LDS ring, their own skip-weight
//     columns into registers -- and accumulate the skip outputs of the layer
//     the chain finished before the last barrier;
//   wave 0 ("chain"): the serial residual chain, weights read from the LDS
//     ring (conflict-free, ~64-cycle latency instead of an L2 / Infinity-Cache
//     round trip), state value and bias prefetched one layer ahead.
// Nothing the loaders fetch depends on the data being computed, so all global
// latency sits off the critical path.
#define FG_CW LAYER_W_FLOATS   // chain weights per layer (floats)
#define FG_CW4 (FG_CW / 4)
#define FG_LDR ((FG_CW4 + FG_SKT - 1) / FG_SKT)   // float4 per loader thread

__global__ __launch_bounds__(FG_THREADS, 1) void fastgen_kernel(FastGen g) {
  __shared__ __attribute__((aligned(16))) float wring[2][FG_CW];
  __shared__ float zbuf[2][32];
  __shared__ float hbuf[FG_MAXS];     // relu(total)
  __shared__ float h2buf[FG_MAXS];    // relu(conv1)
  __shared__ float part[FG_MAXS];     // post2 partial sums
  __shared__ double pd[FG_MAXQ];
  __shared__ int s_code;
  __shared__ int pos[FG_MAXL];        // ring cursor of every layer
  __shared__ int sdil[FG_MAXL], roff[FG_MAXL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = g.S, Q = g.Q, L = g.L;
  const int st = tid - 64;            // loader-thread index 0..191 (waves 1..3)

  const int steps_done = g.cursors[0];
  int prev_code = g.cursors[1];
  if (tid == 0) s_code = g.samples[0];
  for (int l = tid; l < L; l += FG_THREADS) {
    sdil[l] = g.dil[l];
    pos[l] = steps_done % g.dil[l];
  }
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int l = 0; l < L; ++l) { roff[l] = off; off += sdil[l]; }
  }
  __syncthreads();

  for (int step = 0; step < g.n_steps; ++step) {
    const int code = s_code;
    const long tpos = (long)steps_done + step;
    // ---------------- loaders: chain-weight ring + skip columns ------------
    f32x4 cw[FG_LDR];
    auto cw_load = [&](int l) {        // global -> registers
      const f32x4* src = reinterpret_cast<const f32x4*>(
          g.layer0 + (long)l * g.layer_stride);
#pragma unroll
      for (int k = 0; k < FG_LDR; ++k) {
        const int i4 = st + FG_SKT * k;
        if (i4 < FG_CW4) cw[k] = src[i4];
      }
    };
    auto cw_store = [&](int buf) {     // registers -> LDS ring
      f32x4* dst = reinterpret_cast<f32x4*>(wring[buf]);
#pragma unroll
      for (int k = 0; k < FG_LDR; ++k) {
        const int i4 = st + FG_SKT * k;
        if (i4 < FG_CW4) dst[i4] = cw[k];
      }
    };
    float acc[FG_SKO], sw[FG_SKO][32];
    bool son[FG_SKO];
#pragma unroll
    for (int o = 0; o < FG_SKO; ++o) {
      acc[o] = 0.f;
      son[o] = wave >= 1 && st + FG_SKT * o < S;
    }
    auto skip_load = [&](int l) {
      const float* ws = g.skip_w + (long)l * 32 * S + st;
#pragma unroll
      for (int o = 0; o < FG_SKO; ++o)
        if (son[o]) {
#pragma unroll
          for (int k = 0; k < 32; ++k) sw[o][k] = ws[(long)k * S + FG_SKT * o];
        }
    };
    auto skip_fma = [&](const float* zl) {
#pragma unroll
      for (int o = 0; o < FG_SKO; ++o)
        if (son[o]) {
#pragma unroll
          for (int k = 0; k < 32; ++k) acc[o] = fmaf(zl[k], sw[o][k], acc[o]);
        }
    };
    // ---------------- chain state (wave 0) ---------------------------------
    float x = 0.f;                     // residual stream on lanes < 32
    float stv = 0.f, bias = 0.f, bdv = 0.f;  // prefetched for the coming layer
    auto chain_prefetch = [&](int l) {
      bias = g.bias_fg ? g.bias_fg[l * 64 + lane] : 0.f;
      bdv = g.use_dense_bias
                ? g.layer0[(long)l * g.layer_stride + LAYER_OFF_BD + (lane & 31)]
                : 0.f;
      stv = lane < 32 ? g.state[((long)roff[l] + pos[l]) * 32 + lane] : 0.f;
    };
    // prologue: layer 0 weights into ring[0]
    if (wave >= 1) {
      cw_load(0);
      cw_store(0);
      if (L > 1) cw_load(1);
      skip_load(0);
    } else {
      chain_prefetch(0);
      if (lane < 32) {
        float v = 0.f;
        if (prev_code >= 0 && prev_code < Q) v = g.causal[(long)prev_code * 32 + lane];
        if (code >= 0 && code < Q) v += g.causal[((long)Q + code) * 32 + lane];
        x = v;
      }
    }
    __syncthreads();
    for (int l = 0; l <= L; ++l) {
      if (wave == 0) {
        if (l < L) {
          const float* wl = wring[l & 1];
          const float cur_st = stv, cur_bias = bias, cur_bd = bdv;
          if (g.push && lane < 32)       // enqueue x_l[t] (after the dequeue)
            g.state[((long)roff[l] + pos[l]) * 32 + lane] = x;
          if (l + 1 < L) chain_prefetch(l + 1);
          const float* wcol = wl + (lane < 32 ? 0 : 2048) + (lane & 31);
          float a = cur_bias;
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            a = fmaf(readlane_f(cur_st, k), wcol[k * 32], a);      // W[0]
            a = fmaf(readlane_f(x, k), wcol[1024 + k * 32], a);    // W[1]
          }
          const float gate = __shfl(a, (lane & 31) + 32);
          const float z = wn_tanh(a) * wn_sigmoid(gate);  // valid on lanes < 32
          if (lane < 32) zbuf[l & 1][lane] = z;
          if (l + 1 < L) {
            float dsum = cur_bd;
            const float* wd = wl + 4096 + (lane & 31);
#pragma unroll
            for (int k = 0; k < 32; ++k) dsum = fmaf(readlane_f(z, k), wd[k * 32], dsum);
            if (lane < 32) x += dsum;
          }
        }
      } else {
        // ring slot (l+1)&1 was last read in iteration l-1: free to refill
        if (l + 1 < L) cw_store((l + 1) & 1);
        if (l + 2 < L) cw_load(l + 2);
        if (l >= 1) skip_fma(zbuf[(l - 1) & 1]);
        if (l >= 1 && l < L) skip_load(l);
      }
      // raw barrier: only LDS traffic is drained.  __syncthreads() would also
      // wait vmcnt(0), i.e. for the weight loads just issued for the NEXT
      // layers -- exposing a full L2 / Infinity-Cache round trip per layer.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // advance the ring cursors (after every wave is done with this step)
    if (g.push) {
      for (int l = tid; l < L; l += FG_THREADS) {
        const int p = pos[l] + 1;
        pos[l] = p == sdil[l] ? 0 : p;
      }
    }
    // ---------------- post-processing (model.py:505-514) -------------------
    if (wave >= 1) {
#pragma unroll
      for (int o = 0; o < FG_SKO; ++o)
        if (son[o]) {
          const int sc = st + FG_SKT * o;
          hbuf[sc] = fmaxf(acc[o] + (g.skip_bsum ? g.skip_bsum[sc] : 0.f), 0.f);
        }
    }
    __syncthreads();
    if (wave >= 1) {
      for (int s = st; s < S; s += FG_SKT) {
        float c0 = g.post1_b ? g.post1_b[s] : 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        const float* w = g.post1_w + s;
        int k = 0;
        for (; k + 64 <= S; k += 64) {
          float wv[64];
#pragma unroll
          for (int u = 0; u < 64; ++u) wv[u] = w[(long)(k + u) * S];
#pragma unroll
          for (int u = 0; u < 64; u += 4) {
            c0 = fmaf(hbuf[k + u], wv[u], c0);
            c1 = fmaf(hbuf[k + u + 1], wv[u + 1], c1);
            c2 = fmaf(hbuf[k + u + 2], wv[u + 2], c2);
            c3 = fmaf(hbuf[k + u + 3], wv[u + 3], c3);
          }
        }
        for (; k < S; ++k) c0 = fmaf(hbuf[k], w[(long)k * S], c0);
        h2buf[s] = fmaxf((c0 + c1) + (c2 + c3), 0.f);
      }
    }
    __syncthreads();
    // logits: thread (q, part) sums a k-range; parts = 192 / Q
    {
      int parts = FG_SKT / Q;
      if (parts < 1) parts = 1;
      if (wave >= 1) {
        for (int o = st; o < Q * parts; o += FG_SKT) {
          const int q = o % Q, p = o / Q;
          const int k0 = (int)((long)S * p / parts), k1 = (int)((long)S * (p + 1) / parts);
          float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
          const float* w = g.post2_w + q;
          int k = k0;
          for (; k + 64 <= k1; k += 64) {
            float wv[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) wv[u] = w[(long)(k + u) * Q];
#pragma unroll
            for (int u = 0; u < 64; u += 4) {
              c0 = fmaf(h2buf[k + u], wv[u], c0);
              c1 = fmaf(h2buf[k + u + 1], wv[u + 1], c1);
              c2 = fmaf(h2buf[k + u + 2], wv[u + 2], c2);
              c3 = fmaf(h2buf[k + u + 3], wv[u + 3], c3);
            }
          }
          for (; k < k1; ++k) c0 = fmaf(h2buf[k], w[(long)k * Q], c0);
          part[o] = (c0 + c1) + (c2 + c3);
        }
      }
      __syncthreads();
      for (int q = tid; q < Q; q += FG_THREADS) {
        float c = g.post2_b ? g.post2_b[q] : 0.f;
        for (int p = 0; p < parts; ++p) c += part[p * Q + q];
        pd[q] = (double)c;
      }
      __syncthreads();
    }
    // softmax in float64 (model.py:620-621), optional temperature
    // (generate.py:229-233), draw (generate.py:239-240)
    if (wave == 0) {
      double m = -1e300;
      for (int q = lane; q < Q; q += 64) m = fmax(m, pd[q]);
      for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
      double se = 0.0;
      for (int q = lane; q < Q; q += 64) {
        const double e = exp(pd[q] - m);    // evaluated once, kept in LDS
        pd[q] = e;
        se += e;
      }
      for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
      const bool want_p = g.proba_out && (step % g.proba_every == 0);
      float* po = want_p ? g.proba_out + (long)(step / g.proba_every) * Q : nullptr;
      for (int q = lane; q < Q; q += 64) {
        const float p32 = (float)(pd[q] / se);
        if (po) po[q] = p32;
        pd[q] = (double)p32;  // the float32 probabilities generate.py sees
      }
    }
    __syncthreads();
    if (step + 1 >= g.n_given) {
      if (wave == 0) {
        // temperature: exp(log(p)/tau - logsumexp) in float64, then inverse
        // CDF with a counter-based uniform (np.random.choice equivalent)
        // (sampling weights proportional to exp(log(p)/tau); at tau == 1
        // that is p itself, no transcendental needed)
        const double tau = (double)g.temperature;
        if (g.temperature != 1.0f) {
          double mx = -1e300;
          for (int q = lane; q < Q; q += 64) {
            const double lp = log(pd[q] > 0.0 ? pd[q] : 1e-300) / tau;
            pd[q] = lp;
            mx = fmax(mx, lp);
          }
          for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
          for (int q = lane; q < Q; q += 64) pd[q] = exp(pd[q] - mx);
        }
        __builtin_amdgcn_wave_barrier();
        // per-lane contiguous segment [q0, q1): segment sums -> prefix -> pick
        const int per = (Q + 63) / 64;
        const int q0 = lane * per, q1 = min(Q, q0 + per);
        double seg = 0.0;
        for (int q = q0; q < q1; ++q) seg += pd[q];
        double incl = seg;
        for (int o = 1; o < 64; o <<= 1) {
          const double v = __shfl_up(incl, o);
          if (lane >= o) incl += v;
        }
        const double total = __shfl(incl, 63);
        const uint64_t r = splitmix64(g.seed ^ splitmix64((uint64_t)tpos));
        const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0) * total;
        // (the neighbour's inclusive sum, not incl - seg: the lanes' intervals
        // then tile [0, total) exactly -- no gap a draw could fall into)
        const double up = __shfl_up(incl, 1);
        const double excl = lane == 0 ? 0.0 : up;
        int pick = -1;
        if (u >= excl && u < incl) {
          double c = excl;
          pick = q1 - 1;
          for (int q = q0; q < q1; ++q) {
            c += pd[q];
            if (u < c) { pick = q; break; }
          }
        }
        // exactly one lane holds the pick (u < total); fall back to Q-1
        int best = pick;
        for (int o = 32; o >= 1; o >>= 1) best = max(best, __shfl_xor(best, o));
        if (best < 0) best = Q - 1;
        if (lane == 0) {
          g.samples[step + 1] = best;
          s_code = best;
        }
      }
    } else if (tid == 0) {
      s_code = g.samples[step + 1];
    }
    prev_code = code;
    __syncthreads();
  }
  if (tid == 0 && g.push) {
    g.cursors[0] = steps_done + g.n_steps;
    g.cursors[1] = prev_code;
  }
}

// ===========================================================================
// Multi-CU fast generation: one generated sample = four small kernels on the
// stream (captured into a hipGraph by the host, hundreds of samples per
// replay):
//   A fg_chain_kernel   1 workgroup : the serial residual chain, CURRENT tap
//                        only.  The past-tap half of every layer's dilated
//                        conv, pre_l = x_l[t - d_l] (Wf[0] | Wg[0]) + bias_l,
//                        reads a queue entry that is at least one step old,
//                        so it is computed for ALL layers in parallel by the
//                        previous step's kernel B and the chain starts from
//                        a 12.8 KB table in LDS: no global load, half the
//                        mat-vec work and 12 KB instead of 20 KB of weights
//                        per layer on the critical path.  Four loader waves
//                        stream Wf[1] | Wg[1] | Wd through a 4-slot LDS ring
//                        three layers ahead of the chain wave.
//   B fg_skip_kernel    S/16 WGs    : total = sum_l z_l Ws_l (+bias), ReLU;
//                        + L more workgroups: pre_l of the NEXT step
//   C fg_post1_kernel   S/16 WGs    : conv1 (+bias), ReLU
//   D fg_logits_kernel  Q/16 WGs    : conv2 logits (marks a draw pending)
//   (the float64 softmax, temperature, inverse-CDF draw and cursor update of
//    step t run at the START of step t+1's kernel A, on its chain wave, while
//    the loader waves run their prologue; wn_fastgen_finish draws for the
//    last step of a run)
// Kernel boundaries are the grid-wide synchronisation (about 1.5 us each):
// no in-launch flags, nothing that can hang.  The 3.3 MB skip and 1 MB conv1
// weights are read by S/16 CUs in parallel instead of one CU's load path.
// Everything that differs between two generate() calls (first step, seed,
// temperature, number of given samples) is read from a device control block,
// so one captured graph serves every call.
// ===========================================================================
#define FGCTL_BASE 0         // cursors[0] when the call started
#define FGCTL_NGIVEN 1
#define FGCTL_PEVERY 2
#define FGCTL_TEMP 3         // float bits
#define FGCTL_SEED 4         // 4, 5: uint64 seed (lo, hi)
#define FGCTL_WORDS 8

struct FgStep {
  const float* causal;
  const float* layer0;
  long layer_stride;
  const float* skip_w;
  const float* skip_bsum;
  const float* post1_w;
  const float* post1_b;
  const float* post2_w;
  const float* post2_b;
  const float* bias_fg;
  const int32_t* dil;
  int L, S, Q;
  float* state;
  int32_t* cursors;      // [0] steps done, [1] previous code, [2] draw pending
  int32_t* samples;      // indexed by (cursors[0] - ctl[BASE])
  const int32_t* ctl;    // per-call control block (device)
  float* proba_out;
  int use_dense_bias;
  const float* cw_img;   // [L][3072] Wf[1] | Wg[1] | Wd, transposed + swizzled
  float* pre;            // [L][64] past-tap pre-activations of the next step
  float* z_all;          // [L][32]
  float* h1;             // [S]
  float* h2;             // [S]
  float* logits;         // [Q]
};

#define FGC_THREADS 320
#define FGC_SLOTS 4
#define FGC_CW 3072            // ring slot: Wf[1] | Wg[1] | Wd   (floats)

// Ring-slot image: the three 32x32 matrices TRANSPOSED ([out n][in k], k
// contiguous) with the 16-byte chunk c of row n stored at chunk c ^ (n & 7).
// The chain lane that owns output n then reads its whole weight row with
// eight ds_read_b128 per matrix (2-way conflicts at worst), and the input
// values are broadcast from LDS with ds_read_b128 instead of v_readlane.
__device__ __forceinline__ int fgc_widx(int m, int n, int k) {
  return m * 1024 + n * 32 + ((((k >> 2) ^ (n & 7)) << 2) | (k & 3));
}

// one-off: layer blocks -> ring-slot images (weights are constant while
// generating).  Source matrices 1 (Wf[1]), 3 (Wg[1]), 4 (Wd) -> slots 0, 1, 2.
__global__ void fg_pack_kernel(const float* __restrict__ layer0, long layer_stride,
                               float* __restrict__ img) {
  const int l = blockIdx.x;
  const float* blk = layer0 + (long)l * layer_stride;
  for (int i = threadIdx.x; i < FGC_CW; i += blockDim.x) {
    const int m = i >> 10, k = (i >> 5) & 31, n = i & 31;   // slot m: W[k][n]
    const int src = m == 0 ? 1 : (m == 1 ? 3 : 4);
    img[(long)l * FGC_CW + fgc_widx(m, n, k)] = blk[src * 1024 + k * 32 + n];
  }
}

// pre[l][0:32 | 32:64] = x_l[t' - d_l] (Wf[0] | Wg[0]) + bias_fg[l]  for the
// step t' = cursors[0] + ahead: the entry the queue of layer l hands out at
// that step (model.py:335-338 `state`), which is at least one step old.
__device__ __forceinline__ void fg_pre_layer(const FgStep& g, int l, int ahead,
                                             float* lds /* >= 32 + 256 floats */) {
  const int tid = threadIdx.x;                 // 256 threads
  const int tpos = g.cursors[0] + ahead;
  int roff = 0;
  for (int i = 0; i < l; ++i) roff += g.dil[i];
  const int d = g.dil[l];
  float* sv = lds;
  float* red = lds + 32;
  if (tid < 32) sv[tid] = g.state[((long)roff + tpos % d) * 32 + tid];
  __syncthreads();
  // thread -> output n = tid & 63 (filter | gate), K-slice tid >> 6 (8 inputs)
  const int n = tid & 63, ks = tid >> 6;
  const float* W = g.layer0 + (long)l * g.layer_stride +
                   (n < 32 ? 0 : 2 * 1024) + (n & 31);      // Wf[0] / Wg[0]
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) a = fmaf(sv[ks * 8 + k], W[(ks * 8 + k) * 32], a);
  red[tid] = a;
  __syncthreads();
  if (tid < 64)
    g.pre[l * 64 + tid] = (g.bias_fg ? g.bias_fg[l * 64 + tid] : 0.f) +
                          ((red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]));
}

__global__ __launch_bounds__(256) void fg_pre_kernel(FgStep g, int ahead) {
  __shared__ float lds[32 + 256];
  fg_pre_layer(g, blockIdx.x, ahead, lds);
}

__device__ __forceinline__ int fg_draw_wave(const FgStep& g, double* pd, int lane,
                                            int steps_done, const float* lg = nullptr);

#ifdef FG_STAMPS   // diagnostic build: per-layer phase stamps of the chain wave
__device__ unsigned long long fg_dbg[8 + FG_MAXL * 8];
#define FSTAMP(i) if (threadIdx.x == 0) fg_dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define FSTAMP(i)
#endif
__global__ __launch_bounds__(FGC_THREADS) void fg_chain_kernel(FgStep g) {
  FSTAMP(0);
  // ring slot: Wf[1] | Wg[1] | Wd image, then this step's pre[l][64] and the
  // dense bias bd[l][32] (fetched by the loaders with the weights, so the
  // chain needs no table of its own and the prologue no bulk load)
  __shared__ __attribute__((aligned(16))) float wring[FGC_SLOTS][FGC_CW + 128];
  __shared__ __attribute__((aligned(16))) float inv[32];   // x, broadcast
  __shared__ __attribute__((aligned(16))) float zv[32];
  __shared__ int pos[FG_MAXL], roff[FG_MAXL], sdil[FG_MAXL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = g.L, Q = g.Q;
  const int lt = tid - 64;                       // loader thread 0..255
  __shared__ double pd[FG_MAXQ];
  // the previous step's logits are still to be drawn from (cursors[2]): wave 0
  // does that first, while the loader waves run their prologue.  The values
  // it leaves in cursors[] are carried in registers here (this CU's L1 may
  // still hold the old ones).
  int steps_done = g.cursors[0];
  int prev_code = g.cursors[1];
  int code = 0;
  FSTAMP(1);
  if (wave == 0) {
    if (g.cursors[2]) {
      prev_code = g.samples[steps_done - g.ctl[FGCTL_BASE]];
      code = fg_draw_wave(g, pd, lane, steps_done);
      steps_done += 1;
    } else {
      code = g.samples[steps_done - g.ctl[FGCTL_BASE]];
    }
  }
  FSTAMP(2);
  // loaders: 256 threads x 3 float4 = one layer of the pre-packed image
  // (+ one float4 of pre[l] for loader threads 0..15, of bd[l] for 16..23)
  f32x4 s0[4], s1[4], s2[4];
  auto ld = [&](f32x4 (&r)[4], int l) {
    if (l < L) {
      const f32x4* src = reinterpret_cast<const f32x4*>(g.cw_img + (long)l * FGC_CW);
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = src[lt + 256 * k];
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      r[3] = zero;
      if (lt < 16)
        r[3] = reinterpret_cast<const f32x4*>(g.pre + (long)l * 64)[lt];
      else if (lt < 24 && g.use_dense_bias)
        r[3] = reinterpret_cast<const f32x4*>(
            g.layer0 + (long)l * g.layer_stride + LAYER_OFF_BD)[lt - 16];
    }
  };
  auto stl = [&](const f32x4 (&r)[4], int l) {   // straight 16-byte copies
    if (l < L) {
      f32x4* dst = reinterpret_cast<f32x4*>(wring[l % FGC_SLOTS]);
#pragma unroll
      for (int k = 0; k < 3; ++k) dst[lt + 256 * k] = r[k];
      if (lt < 24) dst[FGC_CW / 4 + lt] = r[3];
    }
  };
  // nothing in the prologue depends on the new sample except the causal
  // gather: every global load is issued before the first barrier
  float x = 0.f;
  if (wave >= 1) {
    ld(s0, 0);
    ld(s1, 1);
    ld(s2, 2);
  } else {
    // ring offsets = exclusive prefix sum of the dilations (wave scan, L <= 64)
    const int dl = lane < L ? g.dil[lane] : 0;
    int incl = dl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane < L) {
      sdil[lane] = dl;
      roff[lane] = incl - dl;
      pos[lane] = steps_done % dl;
    }
    if (lane < 32) {
      float v = 0.f;
      if (prev_code >= 0 && prev_code < Q) v = g.causal[(long)prev_code * 32 + lane];
      if (code >= 0 && code < Q) v += g.causal[((long)Q + code) * 32 + lane];
      x = v;
    }
  }
  if (wave >= 1) {
    stl(s0, 0);
    ld(s0, 3);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  FSTAMP(3);
  const int nn = lane & 31, gsel = lane >> 5;     // output row, 0 filter / 1 gate
  auto body = [&](int l, f32x4 (&set)[4]) {
    if (wave == 0) {
      if (l < L) {
        const float* wl = wring[l % FGC_SLOTS];
        FSTAMP(8 + l * 8 + 0);
        // the layer's weights do not depend on x: their LDS reads are issued
        // before the broadcast of x (and the dense conv's before that of z)
        // instead of behind the wave barriers, where each costs the chain an
        // LDS round trip
        const float* w1 = wl + gsel * 1024 + nn * 32;
        const float* wd = wl + 2 * 1024 + nn * 32;
        f32x4 qw[8], pw[4];
#pragma unroll
        for (int c = 0; c < 8; ++c)
          qw[c] = *reinterpret_cast<const f32x4*>(w1 + ((c ^ (nn & 7)) << 2));
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          pw[cc] = *reinterpret_cast<const f32x4*>(wd + (((gsel * 4 + cc) ^ (nn & 7)) << 2));
        float a0 = wl[FGC_CW + lane], a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float bdl = wl[FGC_CW + 64 + (lane & 31)];
        if (lane < 32) {
          g.state[((long)roff[l] + pos[l]) * 32 + lane] = x;  // enqueue x_l[t]
          inv[lane] = x;
        }
        __builtin_amdgcn_wave_barrier();
        // current tap: lane -> output nn of filter (gsel 0) or gate (gsel 1)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(inv + 4 * c);
          const f32x4 q = qw[c];
          a0 = fmaf(xv[0], q[0], a0); a1 = fmaf(xv[1], q[1], a1);
          a2 = fmaf(xv[2], q[2], a2); a3 = fmaf(xv[3], q[3], a3);
        }
        const float a = (a0 + a1) + (a2 + a3);
        FSTAMP(8 + l * 8 + 1);
        // lanes 0-31 hold the filter pre-activation, lanes 32-63 the gate's:
        // one exp + one rcp per lane (tanh(a) = 2 sigmoid(2a) - 1), then the
        // halves meet through v_permlane32_swap (no LDS round trip)
        const float s = wn_sigmoid(gsel ? a : 2.f * a);
        const float act = gsel ? s : fmaf(2.f, s, -1.f);
        const auto pr = __builtin_amdgcn_permlane32_swap(
            __float_as_uint(act), __float_as_uint(act), false, false);
        const float z = __uint_as_float(pr[0]) * __uint_as_float(pr[1]);
        FSTAMP(8 + l * 8 + 2);
        if (lane < 32) {
          g.z_all[l * 32 + lane] = z;
          zv[lane] = z;
        }
        if (l + 1 < L) {
          __builtin_amdgcn_wave_barrier();
          // dense 32 x 32: each half takes 16 of the 32 inputs
          float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int c = gsel * 4 + cc;
            const f32x4 zz = *reinterpret_cast<const f32x4*>(zv + 4 * c);
            const f32x4 p = pw[cc];
            d0 = fmaf(zz[0], p[0], d0); d1 = fmaf(zz[1], p[1], d1);
            d2 = fmaf(zz[2], p[2], d2); d3 = fmaf(zz[3], p[3], d3);
          }
          const float dh = (d0 + d1) + (d2 + d3);
          const auto pd = __builtin_amdgcn_permlane32_swap(
              __float_as_uint(dh), __float_as_uint(dh), false, false);
          if (lane < 32)
            x += bdl + (__uint_as_float(pd[0]) + __uint_as_float(pd[1]));
        }
        FSTAMP(8 + l * 8 + 3);
      }
    } else {
      stl(set, l + 1);
      ld(set, l + 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef FG_STAMPS
    if (l < L) { FSTAMP(8 + l * 8 + 4); }
#endif
  };
  for (int l = 0; l < L; l += 3) {
    body(l, s1);          // stores layer l+1 (set (l+1)%3 == 1 when l%3 == 0)
    body(l + 1, s2);
    body(l + 2, s0);
  }
}

// The three mat-vecs after the chain (skip sum, postprocess1, postprocess2) are
// latency-bound: a workgroup's time is (weight loads per thread / loads in
// flight) round trips.  16 outputs x 16 K-slices per 256-thread workgroup (twice
// the workgroups, half the loads per thread of the first 32 x 8 version) and
// every thread's loads issued in at most two batches.
#define FGM_OUTS 16
#define FGM_PARTS 16

// partial dot product of this thread's K-slice for output column `col`
// (W is [K][ld], 64-byte coalesced over the 16 output lanes); in_s is in LDS
template <int BATCH>
__device__ __forceinline__ float fg_mv_partial(const float* in_s, int K,
                                               const float* __restrict__ W,
                                               long ld, int col, int part) {
  const int per = (K + FGM_PARTS - 1) / FGM_PARTS;
  const int k0 = part * per, k1 = min(K, k0 + per);
  const float* w = W + col;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
  int k = k0;
  for (; k + BATCH <= k1; k += BATCH) {
    float wv[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) wv[u] = w[(long)(k + u) * ld];
#pragma unroll
    for (int u = 0; u + 3 < BATCH; u += 4) {
      c0 = fmaf(in_s[k + u], wv[u], c0);
      c1 = fmaf(in_s[k + u + 1], wv[u + 1], c1);
      c2 = fmaf(in_s[k + u + 2], wv[u + 2], c2);
      c3 = fmaf(in_s[k + u + 3], wv[u + 3], c3);
    }
#pragma unroll
    for (int u = BATCH & ~3; u < BATCH; ++u) c0 = fmaf(in_s[k + u], wv[u], c0);
  }
  for (; k < k1; ++k) c0 = fmaf(in_s[k], w[(long)k * ld], c0);
  return (c0 + c1) + (c2 + c3);
}

__device__ __forceinline__ float fg_mv_reduce(float (*red)[FGM_OUTS], int o) {
  float t = 0.f;
#pragma unroll
  for (int p = 0; p < FGM_PARTS; ++p) t += red[p][o];
  return t;
}

// h1[s] = relu(sum_l z_l . Ws_l[:, s] + sum_l bs_l[s])   (model.py:505-509)
// (+ g.L more workgroups: the past-tap pre-activations of the NEXT step)
__global__ __launch_bounds__(256) void fg_skip_kernel(FgStep g) {
  __shared__ float zs[FG_MAXL * 32];
  __shared__ float red[FGM_PARTS][FGM_OUTS];
  const int nskip = (g.S + FGM_OUTS - 1) / FGM_OUTS;
  if ((int)blockIdx.x >= nskip) {        // workgroup-uniform
    fg_pre_layer(g, blockIdx.x - nskip, 1, zs);
    return;
  }
  const int tid = threadIdx.x, o = tid & (FGM_OUTS - 1), part = tid / FGM_OUTS;
  const int s = blockIdx.x * FGM_OUTS + o;
  const int KK = g.L * 32;
  for (int i = tid; i < KK; i += 256) zs[i] = g.z_all[i];
  __syncthreads();
  red[part][o] = s < g.S ? fg_mv_partial<50>(zs, KK, g.skip_w, g.S, s, part) : 0.f;
  __syncthreads();
  if (part == 0 && s < g.S)
    g.h1[s] = fmaxf((g.skip_bsum ? g.skip_bsum[s] : 0.f) + fg_mv_reduce(red, o), 0.f);
}

// h2[s] = relu(sum_k h1[k] W1[k][s] + b1[s])
__global__ __launch_bounds__(256) void fg_post1_kernel(FgStep g) {
  __shared__ float hs[FG_MAXS];
  __shared__ float red[FGM_PARTS][FGM_OUTS];
  const int tid = threadIdx.x, o = tid & (FGM_OUTS - 1), part = tid / FGM_OUTS;
  const int s = blockIdx.x * FGM_OUTS + o;
  const int S = g.S;
  for (int i = tid; i < S; i += 256) hs[i] = g.h1[i];
  __syncthreads();
  red[part][o] = s < S ? fg_mv_partial<32>(hs, S, g.post1_w, S, s, part) : 0.f;
  __syncthreads();
  if (part == 0 && s < S)
    g.h2[s] = fmaxf((g.post1_b ? g.post1_b[s] : 0.f) + fg_mv_reduce(red, o), 0.f);
}

// logits[q] = sum_k h2[k] W2[k][q] + b2[q]
__global__ __launch_bounds__(256) void fg_logits_kernel(FgStep g) {
  __shared__ float hs[FG_MAXS];
  __shared__ float red[FGM_PARTS][FGM_OUTS];
  const int tid = threadIdx.x, o = tid & (FGM_OUTS - 1), part = tid / FGM_OUTS;
  const int q = blockIdx.x * FGM_OUTS + o;
  const int S = g.S, Q = g.Q;
  for (int i = tid; i < S; i += 256) hs[i] = g.h2[i];
  __syncthreads();
  red[part][o] = q < Q ? fg_mv_partial<32>(hs, S, g.post2_w, Q, q, part) : 0.f;
  __syncthreads();
  if (part == 0 && q < Q)
    g.logits[q] = (g.post2_b ? g.post2_b[q] : 0.f) + fg_mv_reduce(red, o);
  if (blockIdx.x == 0 && tid == 0) g.cursors[2] = 1;   // a draw is pending
}

// float64 softmax of the pending logits, temperature, inverse-CDF draw, cursor
// update; executed by ONE wave (pd: FG_MAXQ doubles of LDS owned by it).
// Returns the code the NEXT step consumes (the drawn sample, or the given one)
// and leaves cursors = {steps_done + 1, code, 0 (nothing pending)}.
// (Folding this into fg_logits_kernel -- the last workgroup to arrive, by an
// agent-scope ticket, does the draw -- was measured: 47.1 vs 46.8 us per
// sample; the release / acquire fences cost what the saved boundary returns.
// It runs at the START of the next step's chain kernel instead, next to that
// kernel's weight / table prologue.)
__device__ __forceinline__ int fg_draw_wave(const FgStep& g, double* pd, int lane,
                                            int steps_done, const float* lg) {
  const int Q = g.Q;
  if (!lg) lg = g.logits;      // (the persistent kernel hands in its own coherent copy)
  const int local = steps_done - g.ctl[FGCTL_BASE];
  const int code = g.samples[local];
  const int n_given = g.ctl[FGCTL_NGIVEN];
  const int proba_every = g.ctl[FGCTL_PEVERY] > 0 ? g.ctl[FGCTL_PEVERY] : 1;
  const float temperature = __int_as_float(g.ctl[FGCTL_TEMP]);
  const uint64_t seed = (uint64_t)(uint32_t)g.ctl[FGCTL_SEED] |
                        ((uint64_t)(uint32_t)g.ctl[FGCTL_SEED + 1] << 32);
  for (int q = lane; q < Q; q += 64) pd[q] = (double)lg[q];
  __builtin_amdgcn_wave_barrier();
  double m = -1e300;
  for (int q = lane; q < Q; q += 64) m = fmax(m, pd[q]);
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  // each double-precision exp / log is evaluated once and kept in LDS
  double se = 0.0;
  for (int q = lane; q < Q; q += 64) {
    const double e = exp(pd[q] - m);
    pd[q] = e;
    se += e;
  }
  for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
  const bool want_p = g.proba_out && (local % proba_every == 0);
  float* po = want_p ? g.proba_out + (long)(local / proba_every) * Q : nullptr;
  for (int q = lane; q < Q; q += 64) {
    const float p32 = (float)(pd[q] / se);
    if (po) po[q] = p32;
    pd[q] = (double)p32;
  }
  int next = 0;
  if (local + 1 >= n_given) {
    // sampling weights w_q proportional to exp(log(p_q) / tau)
    // (generate.py:229-233); at tau == 1 that is p_q itself
    const double tau = (double)temperature;
    if (temperature != 1.0f) {
      double mx = -1e300;
      for (int q = lane; q < Q; q += 64) {
        const double lp = log(pd[q] > 0.0 ? pd[q] : 1e-300) / tau;
        pd[q] = lp;
        mx = fmax(mx, lp);
      }
      for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
      for (int q = lane; q < Q; q += 64) pd[q] = exp(pd[q] - mx);
    }
    __builtin_amdgcn_wave_barrier();
    const int per = (Q + 63) / 64;
    const int q0 = lane * per, q1 = min(Q, q0 + per);
    double seg = 0.0;
    for (int q = q0; q < q1; ++q) seg += pd[q];
    double incl = seg;
    for (int o = 1; o < 64; o <<= 1) {
      const double v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    const double total = __shfl(incl, 63);
    const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)steps_done));
    const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0) * total;
    // (the neighbour's inclusive sum, not incl - seg: the lanes' intervals
    // then tile [0, total) exactly -- no gap a draw could fall into)
    const double up = __shfl_up(incl, 1);
    const double excl = lane == 0 ? 0.0 : up;
    int pick = -1;
    if (u >= excl && u < incl) {
      double c = excl;
      pick = q1 - 1;
      for (int q = q0; q < q1; ++q) {
        c += pd[q];
        if (u < c) { pick = q; break; }
      }
    }
    int best = pick;
    for (int o = 32; o >= 1; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (best < 0) best = Q - 1;
    if (lane == 0) g.samples[local + 1] = best;
    next = best;
  } else {
    next = g.samples[local + 1];           // still inside the given samples
  }
  if (lane == 0) {
    g.cursors[0] = steps_done + 1;
    g.cursors[1] = code;
    g.cursors[2] = 0;
  }
  return next;
}

// the draw of the LAST step of a run (nothing follows it)
__global__ __launch_bounds__(64) void fg_draw_kernel(FgStep g) {
  __shared__ double pd[FG_MAXQ];
  if (g.cursors[2]) fg_draw_wave(g, pd, threadIdx.x, g.cursors[0]);
}


// ===========================================================================
// Round 4: ONE persistent multi-CU launch for a whole run of samples
// (wn_fastgen_persist).  The step kernels above pay four kernel boundaries per
// sample (1.65 us each inside a replayed graph, tools/ubench/hop_vs_boundary)
// and stream the 600 KB of chain weights through one CU's LDS ring again for
// every sample, with a workgroup barrier per layer between the chain wave and
// its loaders.  Here every workgroup stays resident for the run and owns its
// weights:
//   chain segments  nseg workgroups, L / nseg consecutive layers each, their
//                   Wf[1] | Wg[1] | Wd images resident in LDS (12 KB a layer);
//                   wave 0 is the serial chain (no barrier inside a segment),
//                   waves 1..4 compute the NEXT step's past-tap
//                   pre-activations of the segment's layers meanwhile;
//   skip            S / 16 workgroups, 16 skip channels each, their [L*32][16]
//                   slice of Ws resident in LDS; they take in a segment's z as
//                   soon as its flag is up, so only the last segment's share
//                   is left after the chain;
//   post1, logits   S / 16 and Q / 16 workgroups with resident weight slices;
//   draw            one wave: float64 softmax, temperature, inverse-CDF draw.
// Hand-overs inside the launch (0.65 - 0.8 us a hop on an idle chip): the
// payload with sc1 (write-through) stores, s_waitcnt vmcnt(0), then a relaxed
// agent-scope flag / counter holding the 1-based step; consumers poll it and
// read the payload with sc1 loads.  Every wait is bounded (2 s): on expiry the
// error word is set, the waiter goes on without waiting, and the grid drains.
// All nseg + S/16 + S/16 + Q/16 + 1 workgroups must be resident together (86
// for the default stack on 256 CUs); the host refuses shapes that cannot be.
// ===========================================================================
#define FGP_THREADS 320
#define FGP_MAXSEG 8
#define FGP_SEGL 16            // most layers per chain segment (12.25 KB of LDS each)
// sync words (uint32): [0..7] segment flags, [8] h1 counter, [9] h2 counter,
// [10] logits counter, [11] draw flag, [12] error
#define FGP_SEG 0
#define FGP_H1 8
#define FGP_H2 9
#define FGP_LG 10
#define FGP_DRAW 11
#define FGP_ERR 12
#define FGP_WORDS 16

// Role of workgroup `b` of a `total`-workgroup persistent launch with `nseg`
// chain segments (roles are numbered: chain segments, skip, post1, logits,
// draw = total - 1).  Workgroups go to the eight XCDs round-robin (b % 8): the
// nseg + 1 roles of the serial chain (segments, draw) take the blocks 0, 8,
// 16, ... -- one XCD --, the mat-vec roles the other blocks in order; a grid too
// small for that keeps the identity.  A permutation of [0, total)
// (wn_fastgen_persist_role exports it for the host-side test).
__host__ __device__ inline int fgp_role_of_block(int b, int total, int nseg) {
  const int nsp = nseg + 1;
  if (total <= 8 * (nsp - 1)) return b;
  if ((b & 7) == 0 && (b >> 3) < nsp) return (b >> 3) < nseg ? (b >> 3) : total - 1;
  const int before = ((b + 7) >> 3) < nsp ? ((b + 7) >> 3) : nsp;   // chain / draw blocks below b
  return nseg + (b - before);
}

struct FgPersist {
  FgStep g;
  unsigned* sync;        // FGP_WORDS words, zero before the launch ([12]: error)
  unsigned long long* ll;   // hand-over words (wn_fastgen_persist_ll_words), zero before the launch
  int n_steps, nseg;
  unsigned long long* dbg;   // diagnostic stamps or null
};

__device__ __forceinline__ float fgp_ld(const float* p) {          // sc1 (device-scope) load
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fgp_st(float* p, float v) {        // sc1 (write-through) store
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a tail mat-vec role: OUT16 = 16 outputs of  out = act(in[K] * W[:, col0 .. col0+15] + b)
// with the weight slice resident in LDS as [K][16]
__device__ __forceinline__ float fgp_mv16(const float* in_s, const float* w_s, int K, int o,
                                          int part, int parts) {
  const int per = (K + parts - 1) / parts;
  const int k0 = part * per, k1 = min(K, k0 + per);
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    c0 = fmaf(in_s[k], w_s[k * 16 + o], c0);
    c1 = fmaf(in_s[k + 1], w_s[(k + 1) * 16 + o], c1);
    c2 = fmaf(in_s[k + 2], w_s[(k + 2) * 16 + o], c2);
    c3 = fmaf(in_s[k + 3], w_s[(k + 3) * 16 + o], c3);
  }
  for (; k < k1; ++k) c0 = fmaf(in_s[k], w_s[k * 16 + o], c0);
  return (c0 + c1) + (c2 + c3);
}

// Hand-over words ("LL" style): a 32-bit payload and the 1-based step it
// belongs to in ONE 8-byte write-through store; the consumer polls the word
// itself until the step matches.  One memory round trip per hop instead of
// three (payload drain + flag, then poll + load): the first version of this
// kernel (counters + sc1 payloads) spent 2.3 / 4.6 / 3.8 / 1.6 us in its four
// tail hand-overs and ran at 49 us per sample.  A location is rewritten once
// per step, and no producer can run a step ahead of a consumer of the same
// word (every word's next write waits, through the draw, for this step's
// reads), so a reader never sees a later step's value.
typedef unsigned long long fgp_ll_t;
__device__ __forceinline__ void fgp_put(fgp_ll_t* p, float v, unsigned step) {
  __hip_atomic_store(p, ((fgp_ll_t)step << 32) | (fgp_ll_t)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// poll until the word carries `step` (bounded: 2 s; `dead` = some wait expired)
__device__ __forceinline__ float fgp_get(const fgp_ll_t* p, unsigned step, unsigned* sync,
                                         bool& dead) {
  fgp_ll_t w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  unsigned long long t_start = 0;
  while (!dead && (unsigned)(w >> 32) != step) {
    __builtin_amdgcn_s_sleep(1);
    w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((++spins & 255u) == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t_start == 0) t_start = now;
      if (now - t_start > 200000000ull ||
          __hip_atomic_load(sync + FGP_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(sync + FGP_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
      }
    }
  }
  return __uint_as_float((unsigned)w);
}

// up to two words per thread, both polled in ONE loop (two dependent polls
// cost two memory round trips): words k0 and k0 + 256 of `src` into `dst`
// (d0, d1: where the two values go in `dst`; default: their own index)
__device__ __forceinline__ void fgp_get2(float* dst, const fgp_ll_t* src, int k0, int n,
                                         unsigned step, unsigned* sync, bool& dead,
                                         int d0 = -1, int d1 = -1) {
  const bool h0 = k0 < n, h1 = k0 + 256 < n;
  if (d0 < 0) { d0 = k0; d1 = k0 + 256; }
  if (!h0) return;
  fgp_ll_t w0 = __hip_atomic_load(src + k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  fgp_ll_t w1 = h1 ? __hip_atomic_load(src + k0 + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                   : ((fgp_ll_t)step << 32);
  unsigned spins = 0;
  unsigned long long t_start = 0;
  while (!dead && ((unsigned)(w0 >> 32) != step || (unsigned)(w1 >> 32) != step)) {
    __builtin_amdgcn_s_sleep(1);
    if ((unsigned)(w0 >> 32) != step)
      w0 = __hip_atomic_load(src + k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(w1 >> 32) != step)
      w1 = __hip_atomic_load(src + k0 + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((++spins & 255u) == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t_start == 0) t_start = now;
      if (now - t_start > 200000000ull ||
          __hip_atomic_load(sync + FGP_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(sync + FGP_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
      }
    }
  }
  dst[d0] = __uint_as_float((unsigned)w0);
  if (h1) dst[d1] = __uint_as_float((unsigned)w1);
}

// ---- wave-wide reductions by DPP (row shifts inside the four 16-lane rows,
// then lane 15 / lane 31 broadcast to the following rows): an inclusive scan in
// six VALU steps, the total in lane 63 -- no LDS round trip per step as with
// ds_bpermute shuffles.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL,
                                                    ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double x) {
  const long long o = __double_as_longlong(old), v = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)v, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
  const long long v = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)v, l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// max over the wave, the same value in every lane
__device__ __forceinline__ float wave_max_f32(float x) {
  x = fmaxf(x, dpp_f32<0x111, 0xf>(x, x));   // row_shr:1
  x = fmaxf(x, dpp_f32<0x112, 0xf>(x, x));
  x = fmaxf(x, dpp_f32<0x114, 0xf>(x, x));
  x = fmaxf(x, dpp_f32<0x118, 0xf>(x, x));
  x = fmaxf(x, dpp_f32<0x142, 0xa>(x, x));   // row_bcast:15 -> rows 1, 3
  x = fmaxf(x, dpp_f32<0x143, 0xc>(x, x));   // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ double wave_max_f64(double x) {
  x = fmax(x, dpp_f64<0x111, 0xf>(x, x));
  x = fmax(x, dpp_f64<0x112, 0xf>(x, x));
  x = fmax(x, dpp_f64<0x114, 0xf>(x, x));
  x = fmax(x, dpp_f64<0x118, 0xf>(x, x));
  x = fmax(x, dpp_f64<0x142, 0xa>(x, x));
  x = fmax(x, dpp_f64<0x143, 0xc>(x, x));
  return readlane_f64(x, 63);
}
// inclusive prefix sum over the lanes (lane 63: the wave's sum)
__device__ __forceinline__ double wave_scan_f64(double x) {
  x += dpp_f64<0x111, 0xf>(0.0, x);
  x += dpp_f64<0x112, 0xf>(0.0, x);
  x += dpp_f64<0x114, 0xf>(0.0, x);
  x += dpp_f64<0x118, 0xf>(0.0, x);
  x += dpp_f64<0x142, 0xa>(0.0, x);
  x += dpp_f64<0x143, 0xc>(0.0, x);
  return x;
}

// ---- a 32-input mat-vec without an LDS round trip (round 6).  The input
// vector sits REPLICATED in every 16-lane row of two registers (lo: inputs
// 0..15, hi: 16..31 -- one v_permlane16_swap of a register that holds input
// (lane & 31) in every lane gives both); lane j then adds input k times ITS
// weight with the input read through the DPP row broadcast of lane k of the
// lane's own row: one v_fmac_f32_dpp per product, no ds_write / ds_read /
// s_waitcnt on the dependent path (the LDS version: ~130 cycles of round trip
// per broadcast, two per layer).  Four accumulator chains, input 4c + e into
// chain e in ascending c: the order of fg_chain_kernel / fastgen_kernel, so
// the results are bitwise theirs.  (Inline asm: the compiler leaves
// update_dpp + fma as v_mov_b32_dpp + v_fmac, twice the instructions.  s_nop 1:
// a VALU write of the DPP operand needs two wait states before the read.)
#define FG_DPPF(A, W, K) "v_fmac_f32_dpp " A ", %[x], " W " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define FG_DPPM(A, W, K) "v_mul_f32_dpp " A ", %[x], " W " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define FG_MV16_TAIL \
      FG_DPPF("%[a0]", "%[w10]", 4) FG_DPPF("%[a1]", "%[w11]", 5) FG_DPPF("%[a2]", "%[w12]", 6) FG_DPPF("%[a3]", "%[w13]", 7) \
      FG_DPPF("%[a0]", "%[w20]", 8) FG_DPPF("%[a1]", "%[w21]", 9) FG_DPPF("%[a2]", "%[w22]", 10) FG_DPPF("%[a3]", "%[w23]", 11) \
      FG_DPPF("%[a0]", "%[w30]", 12) FG_DPPF("%[a1]", "%[w31]", 13) FG_DPPF("%[a2]", "%[w32]", 14) FG_DPPF("%[a3]", "%[w33]", 15)
#define FG_MV16_WOPS \
        [x] "v"(x), [w00] "v"(w0[0]), [w01] "v"(w0[1]), [w02] "v"(w0[2]), [w03] "v"(w0[3]), \
        [w10] "v"(w1[0]), [w11] "v"(w1[1]), [w12] "v"(w1[2]), [w13] "v"(w1[3]), \
        [w20] "v"(w2[0]), [w21] "v"(w2[1]), [w22] "v"(w2[2]), [w23] "v"(w2[3]), \
        [w30] "v"(w3[0]), [w31] "v"(w3[1]), [w32] "v"(w3[2]), [w33] "v"(w3[3])
// a_e += sum_c in(4c + e) * w_c[e]: sixteen FMAs.  NOP: the DPP operand was
// written by the VALU instruction just before (the row-halves swap).
template <bool NOP>
__device__ __forceinline__ void fg_mv16_dpp(float& a0, float& a1, float& a2, float& a3, float x,
                                            const f32x4& w0, const f32x4& w1, const f32x4& w2,
                                            const f32x4& w3) {
  if (NOP)
    asm volatile("s_nop 1\n\t"
        FG_DPPF("%[a0]", "%[w00]", 0) FG_DPPF("%[a1]", "%[w01]", 1) FG_DPPF("%[a2]", "%[w02]", 2) FG_DPPF("%[a3]", "%[w03]", 3)
        FG_MV16_TAIL
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3) : FG_MV16_WOPS);
  else
    asm volatile(
        FG_DPPF("%[a0]", "%[w00]", 0) FG_DPPF("%[a1]", "%[w01]", 1) FG_DPPF("%[a2]", "%[w02]", 2) FG_DPPF("%[a3]", "%[w03]", 3)
        FG_MV16_TAIL
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3) : FG_MV16_WOPS);
}
// the same with chains 1..3 (A0: chain 0 too) STARTING here: their first
// product is a v_mul (= fmaf(in, w, +0) up to the sign of a zero, which no
// later value depends on) instead of an FMA into a register zeroed by a v_mov
template <bool A0>
__device__ __forceinline__ void fg_mv16_dpp_first(float& a0, float& a1, float& a2, float& a3, float x,
                                                  const f32x4& w0, const f32x4& w1, const f32x4& w2,
                                                  const f32x4& w3) {
  if (A0)
    asm volatile("s_nop 1\n\t"
        FG_DPPM("%[a0]", "%[w00]", 0) FG_DPPM("%[a1]", "%[w01]", 1) FG_DPPM("%[a2]", "%[w02]", 2) FG_DPPM("%[a3]", "%[w03]", 3)
        FG_MV16_TAIL
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3) : FG_MV16_WOPS);
  else
    asm volatile("s_nop 1\n\t"
        FG_DPPF("%[a0]", "%[w00]", 0) FG_DPPM("%[a1]", "%[w01]", 1) FG_DPPM("%[a2]", "%[w02]", 2) FG_DPPM("%[a3]", "%[w03]", 3)
        FG_MV16_TAIL
        : [a0] "+v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3) : FG_MV16_WOPS);
}
// (lo, hi) row-replicated halves of a register holding value (lane & 31) in every lane
__device__ __forceinline__ void fg_row_halves(float v, float& lo, float& hi) {
  const auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  lo = __uint_as_float(pr[0]);
  hi = __uint_as_float(pr[1]);
}

// What the draw reads from ctl[]: constant over a launch, read once.
struct FgDrawCtl {
  int base, n_given, proba_every;
  float temperature;
  uint64_t seed;
};
__device__ __forceinline__ FgDrawCtl fg_draw_ctl(const FgStep& g) {
  FgDrawCtl c;
  c.base = g.ctl[FGCTL_BASE];
  c.n_given = g.ctl[FGCTL_NGIVEN];
  c.proba_every = g.ctl[FGCTL_PEVERY] > 0 ? g.ctl[FGCTL_PEVERY] : 1;
  c.temperature = __int_as_float(g.ctl[FGCTL_TEMP]);
  c.seed = (uint64_t)(uint32_t)g.ctl[FGCTL_SEED] | ((uint64_t)(uint32_t)g.ctl[FGCTL_SEED + 1] << 32);
  return c;
}

// The draw of fg_draw_wave by the persistent kernel's draw workgroup (its first
// 256 threads compute, Q <= 512): float64 softmax of the Q logits in LDS `lgs`,
// temperature, inverse-CDF draw with the same random number.  One exp / log /
// division per thread instead of Q / 64 per lane.  Every wave keeps its values
// in registers: sums and the prefix sum are DPP scans inside the wave plus one
// partial per wave in LDS (two workgroup barriers per step; round 4: five, and
// the reductions as ds_bpermute shuffles of doubles: 2.36 us per draw).  The
// thread whose interval holds the random number publishes the code itself.
// part: 16 doubles of LDS; nxt: 2 ints of LDS (the code drawn at step parity).
template <int PER>   // values per computing thread: 1 (Q <= 256) or 2
__device__ __forceinline__ void fg_draw_wg256(const FgStep& g, const FgDrawCtl& dc, const float* lgs,
                                              double* part, int* nxt, fgp_ll_t* x0ll,
                                              const float* ctab, int cur_code,
                                              unsigned step, bool publish, int tid,
                                              int steps_done) {
  const int Q = g.Q, lane = tid & 63, wave = tid >> 6;
  const int local = steps_done - dc.base;
  const int npl = (Q + 63) >> 6;                       // logits per lane in the max pass (<= 8)
  constexpr int per = PER;
  const int q0 = min(Q, tid * per), q1 = min(Q, q0 + per);
  // What segment 0 needs of the drawn code is the causal layer's output for the
  // NEXT step, x0[n] = W[0][cur][n] + W[1][next][n] (model.py:341-346; a code
  // outside [0, Q) contributes nothing), not the code: the wave that holds the
  // drawing thread reads the two rows from the draw workgroup's LDS copy of the
  // table and publishes the 32 values -- segment 0's two dependent global
  // loads per step (0.28 us behind the code word) are gone.  `code`: the same
  // value in every lane of the wave.
  auto publish_x0 = [&](int code) {
    if (!publish || lane >= 32) return;
    float v = 0.f;
    if (cur_code >= 0 && cur_code < Q) v = ctab[cur_code * 32 + lane];
    if (code >= 0 && code < Q) v += ctab[(Q + code) * 32 + lane];
    fgp_put(x0ll + lane, v, step);
  };
  // (the step's random number does not wait for the logits)
  const uint64_t r = splitmix64(dc.seed ^ splitmix64((uint64_t)steps_done));
  // the maximum: every wave for itself over all Q logits
  float mf = -3.0e38f;
  if (Q == 256) {                                      // (the reference's 8-bit mu-law: one read)
    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lgs + lane * 4);
    mf = fmaxf(fmaxf(l4[0], l4[1]), fmaxf(l4[2], l4[3]));
  } else {
    for (int j = 0; j < npl; ++j) {
      const int q = lane * npl + j;
      if (q < Q) mf = fmaxf(mf, lgs[q]);
    }
  }
  const double m = (double)wave_max_f32(mf);
  double e[2] = {0.0, 0.0};
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (q0 + j < q1) e[j] = exp((double)lgs[q0 + j] - m);
  // waves 0 .. 3 hold values (256 threads x per); a fifth wave adds zeros
  const double es = wave_scan_f64(PER == 2 ? e[0] + e[1] : e[0]);
  if (lane == 63 && wave < 4) part[wave] = es;
  __syncthreads();
  const double se = ((part[0] + part[1]) + part[2]) + part[3];
  const bool want_p = g.proba_out && (local % dc.proba_every == 0);
  float* po = want_p ? g.proba_out + (long)(local / dc.proba_every) * Q : nullptr;
  double pq[2] = {0.0, 0.0};
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (q0 + j < q1) {
      const float p32 = (float)(e[j] / se);
      if (po) po[q0 + j] = p32;
      pq[j] = (double)p32;
    }
  if (local + 1 < dc.n_given) {                        // still inside the given samples
    if (wave == 0) {
      const int next = g.samples[local + 1];           // (one address: a broadcast load)
      if (tid == 0) nxt[step & 1] = next;
      publish_x0(next);
    }
    return;
  }
  if (dc.temperature != 1.0f) {
    const double tau = (double)dc.temperature;
    double lp[2] = {-1e300, -1e300};
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (q0 + j < q1) lp[j] = log(pq[j] > 0.0 ? pq[j] : 1e-300) / tau;
    const double wm = wave_max_f64(PER == 2 ? fmax(lp[0], lp[1]) : lp[0]);
    if (lane == 0 && wave < 4) part[4 + wave] = wm;
    __syncthreads();
    const double mx = fmax(fmax(part[4], part[5]), fmax(part[6], part[7]));
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (q0 + j < q1) pq[j] = exp(lp[j] - mx);
  }
  // prefix sums: inside the wave by DPP, the waves' totals through LDS.  A
  // thread's exclusive bound is its neighbour's inclusive sum and a wave's
  // offset is the previous wave's offset plus that wave's total -- the same
  // additions on both sides of every boundary, so the threads' intervals tile
  // [0, total) exactly.
  double incl = wave_scan_f64(PER == 2 ? pq[0] + pq[1] : pq[0]);
  double excl = dpp_f64<0x138, 0xf>(0.0, incl);        // wave_shr:1, lane 0: 0
  if (lane == 63 && wave < 4) part[8 + wave] = incl;
  __syncthreads();
  double offs = 0.0;
  for (int w = 0; w < wave && w < 4; ++w) offs += part[8 + w];
  const double total = ((part[8] + part[9]) + part[10]) + part[11];
  incl += offs;
  excl += offs;
  const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0) * total;
  int next = -1;
  if (q1 > q0 && u >= excl && u < incl)
    next = (PER == 2 && q1 - q0 == 2 && u >= excl + pq[0]) ? q0 + 1 : q0;
  // (u == total after rounding: the last code, as fg_draw_wave)
  if (q1 == Q && q1 > q0 && u >= incl) next = Q - 1;
  if (next >= 0) {
    g.samples[local + 1] = next;
    nxt[step & 1] = next;
  }
  // the wave of the drawing thread (exactly one thread has next >= 0)
  const unsigned long long hit = __builtin_amdgcn_ballot_w64(next >= 0);
  if (hit != 0) {
    const int src = __builtin_ctzll(hit);
    publish_x0(__builtin_amdgcn_readlane(next, src));
  }
}

// A post-processing mat-vec role of the persistent launches (postprocess1 /
// logits): 16 outputs col0 .. col0 + 15 of out = act(in[S] W[:, N] + bias), the
// input from the hand-over words `src`, the outputs to `dst`, one step after
// the other.  256 threads; lds: 2 S floats (S a multiple of 64) or S + 16 S + 256.
// stamp: diagnostic builds, one slot every 16 per step, or null.
__device__ __forceinline__ void fgp_post_role(const float* W, const float* bias, int col0, int N,
                                              int S, bool lg, const fgp_ll_t* src, fgp_ll_t* dst,
                                              int n_steps, float* lds, unsigned* sync, bool& dead,
                                              int tid, unsigned long long* stamp) {
  float* in_s = lds;
  const int o = tid & 15, part = tid >> 4;
  if ((S & 63) == 0) {
    // Round 5: the thread's 32 weights stay in REGISTERS for the run, the 16
    // K-slices of an output sit on 16 adjacent lanes (slice p = lane & 15 of a
    // row), the input vector is stored slice-interleaved ([unit i of the
    // slice][slice]: the lanes' eight 16-byte reads are conflict-free), and the
    // slices' partial sums are added by fifteen DPP row shifts -- in slice
    // order with the same four chains per slice as fgp_mv16 / the step
    // kernels, so the bits are theirs.  One barrier a step (the input buffer
    // alternates), no weight or partial-sum traffic through LDS.
    const int o2 = tid >> 4, p2 = tid & 15;
    const int upp = S >> 6, per = S >> 4;          // 16-byte units / floats of a slice
    float wr[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        wr[u][e] = (u < upp && col0 + o2 < N)
                       ? W[(size_t)(p2 * per + 4 * u + e) * N + col0 + o2] : 0.f;
    const float bv = (bias && col0 + o2 < N) ? bias[col0 + o2] : 0.f;
    // word k -> float ((u % upp) * 16 + u / upp) * 4 + (k & 3), u = k >> 2
    auto spos = [&](int k) { const int u = k >> 2; return ((u % upp) * 16 + u / upp) * 4 + (k & 3); };
    const int d0 = spos(tid), d1 = spos(tid + 256);
    for (int i = 0; i < n_steps; ++i) {
      const unsigned step = (unsigned)(i + 1);
      float* buf = in_s + (i & 1) * S;
      fgp_get2(buf, src, tid, S, step, sync, dead, d0, d1);
      __syncthreads();
      const f32x4* b4 = reinterpret_cast<const f32x4*>(buf) + p2;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < upp) {
          const f32x4 v = b4[u * 16];
          c0 = fmaf(v[0], wr[u][0], c0); c1 = fmaf(v[1], wr[u][1], c1);
          c2 = fmaf(v[2], wr[u][2], c2); c3 = fmaf(v[3], wr[u][3], c3);
        }
      const float r = (c0 + c1) + (c2 + c3);
      // lane 15 of the row: ((r_0 + r_1) + ...) + r_15
      float t = r;
#pragma unroll
      for (int q = 0; q < 15; ++q) t = dpp_f32<0x111, 0xf>(0.f, t) + r;
      if (p2 == 15 && col0 + o2 < N) {
        t += bv;
        fgp_put(dst + col0 + o2, lg ? t : fmaxf(t, 0.f), step);
      }
      if (stamp && (tid & 63) == 0) stamp[(size_t)i * 16] = __builtin_amdgcn_s_memrealtime();
    }
    return;
  }
  float* w_s = lds + ((S + 3) & ~3);                // [S][16]
  float* red = w_s + (size_t)S * 16;
  for (int i = tid; i < S * 16; i += 256) {
    const int k = i >> 4, c = i & 15;
    w_s[i] = col0 + c < N ? W[(size_t)k * N + col0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = 0; i < n_steps; ++i) {
    const unsigned step = (unsigned)(i + 1);
    fgp_get2(in_s, src, tid, S, step, sync, dead);
    __syncthreads();
    red[part * 16 + o] = fgp_mv16(in_s, w_s, S, o, part, 16);
    __syncthreads();
    if (part == 0 && col0 + o < N) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 16; ++p) t += red[p * 16 + o];
      t += bias ? bias[col0 + o] : 0.f;
      fgp_put(dst + col0 + o, lg ? t : fmaxf(t, 0.f), step);
    }
    if (stamp && (tid & 63) == 0) stamp[(size_t)i * 16] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
  }
}

// ---- the serial chain of one segment of fg_persist_kernel, one step (round 6).
// A lone wave issues one instruction every ~5 cycles whatever its kind
// (tools/ubench/dpp_matvec.hip), so a layer costs its instruction COUNT: the
// chain wave's layer is written for that -- no LDS round trip and no exec-mask
// region on the dependent path, every operand of a layer in ONE per-layer LDS
// block read with immediate offsets, the queue stores after the hand-over.
//   LDS block of layer ll (FGP_BLK floats): 12 KB of weights in lane order
//   (float4 [c][lane]: c < 8 filter | gate column chunks, 8 + cc: dense column
//   n's K half lane >> 5) | float4 [lane] {past-tap pre-activation, ring row
//   (float offset, int bits), dense bias, -}.
// x: channel lane & 31 in EVERY lane.  NL > 0: the segment's layer count at
// compile time (loop unrolled, a layer's queue entry kept in registers and
// stored after the hand-over); NL == 0: any count, entries stored at once.
#define FGP_BLK (FGC_CW + 256)
struct FgpLW { f32x4 qw[8], pw[4], ms; };
__device__ __forceinline__ void fgp_lw_load(FgpLW& w, const float* wres, int ll, int lane) {
  const f32x4* b = reinterpret_cast<const f32x4*>(wres) + ll * (FGP_BLK / 4) + lane;
#pragma unroll
  for (int c = 0; c < 8; ++c) w.qw[c] = b[c * 64];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) w.pw[cc] = b[512 + cc * 64];
  w.ms = b[768];
}
template <int NL>
__device__ __forceinline__ float fgp_chain_layers(float x, const float* wres, FgpLW& wa, int nl,
                                                  int l0, int L, float* state, fgp_ll_t* zrow,
                                                  fgp_ll_t* xout, unsigned step, int lane) {
  constexpr bool UNR = NL > 0;
  const int n = UNR ? NL : nl;
  const int nn = lane & 31;
  const bool gsel = lane >= 32;
  const float kexp = gsel ? -1.4426950408889634f : -2.8853900817779268f;
  const float kmul = gsel ? 1.f : 2.f, kadd = gsel ? 0.f : -1.f;
  float xq[UNR ? NL : 1];
  int rq[UNR ? NL : 1];
  FgpLW wb;
  auto layer = [&](int ll, const FgpLW& w, FgpLW& wn) {
    float a0 = w.ms[0], a1, a2, a3;
    float xlo, xhi;
    fg_row_halves(x, xlo, xhi);
    fg_mv16_dpp_first<false>(a0, a1, a2, a3, xlo, w.qw[0], w.qw[1], w.qw[2], w.qw[3]);
    fg_mv16_dpp<false>(a0, a1, a2, a3, xhi, w.qw[4], w.qw[5], w.qw[6], w.qw[7]);
    if (UNR) {
      xq[UNR ? ll : 0] = x;                              // enqueue x_l[t]: after the hand-over
      rq[UNR ? ll : 0] = __float_as_int(w.ms[1]);
    } else {
      fgp_st(state + __float_as_int(w.ms[1]) + nn, x);
    }
    // the next layer's operands: they land under the gate and the dense part
    fgp_lw_load(wn, wres, ll + 1 < n ? ll + 1 : ll, lane);
    const float av = (a0 + a1) + (a2 + a3);
    // gate lanes: sigmoid(av); filter lanes: tanh(av) = 2 sigmoid(2 av) - 1.  The
    // lane's constants instead of selects: exp2(av * (-log2 e) * {1, 2}) is the
    // bits of wn_sigmoid(gsel ? av : 2 av) (a scaling by two is exact), and
    // fma(sg, 1, 0) = sg.
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(av * kexp));
    const float act = fmaf(sg, kmul, kadd);
    const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act),
                                                     false, false);
    const float z = __uint_as_float(pr[0]) * __uint_as_float(pr[1]);   // channel nn, every lane
    fgp_put(zrow + ll * 32, z, step);                    // (lanes n and n + 32: the same word)
    if (ll + 1 < n || l0 + n < L) {                      // (every layer but the network's last)
      // z[16 (lane >> 5) + j] to the lanes of the lane's half: rows {lo, lo, hi, hi}
      const auto p16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z),
                                                        false, false);
      const auto p32 = __builtin_amdgcn_permlane32_swap(p16[0], p16[1], false, false);
      float d0, d1, d2, d3;
      fg_mv16_dpp_first<true>(d0, d1, d2, d3, __uint_as_float(p32[0]), w.pw[0], w.pw[1], w.pw[2], w.pw[3]);
      const float dh = (d0 + d1) + (d2 + d3);
      const auto pd = __builtin_amdgcn_permlane32_swap(__float_as_uint(dh), __float_as_uint(dh),
                                                       false, false);
      x += w.ms[2] + (__uint_as_float(pd[0]) + __uint_as_float(pd[1]));
    }
  };
  constexpr int UF = UNR ? (NL + 1) / 2 : 1;
#pragma unroll UF
  for (int ll = 0; ll < n; ll += 2) {
    layer(ll, wa, wb);
    if (ll + 1 < n) layer(ll + 1, wb, wa);
  }
  // x to the next segment at once; the queue entries are for this segment's
  // own helper waves
  if (xout) fgp_put(xout + nn, x, step);
  if (UNR) {
#pragma unroll
    for (int ll = 0; ll < (UNR ? NL : 0); ++ll) fgp_st(state + rq[ll] + nn, xq[ll]);
  }
  return x;
}

__global__ __launch_bounds__(FGP_THREADS) void fg_persist_kernel(FgPersist a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const FgStep& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = g.L, S = g.S, Q = g.Q, nseg = a.nseg, n_steps = a.n_steps;
  const int nsk = (S + 15) / 16, nlg = (Q + 15) / 16;
  const int base = g.ctl[FGCTL_BASE];
  unsigned* sync = a.sync;
  // hand-over words: z [L][32] | h1 [S] | h2 [S] | logits [Q] | x [nseg][32] | x0 [32]
  fgp_ll_t* zll = a.ll;
  fgp_ll_t* h1ll = zll + L * 32;
  fgp_ll_t* h2ll = h1ll + S;
  fgp_ll_t* lgll = h2ll + S;
  fgp_ll_t* xll = lgll + Q;
  fgp_ll_t* x0ll = xll + FGP_MAXSEG * 32;            // [32] the causal layer's output for the next step
  bool dead = false;
  // Role of this workgroup.  Workgroups go to the eight XCDs round-robin
  // (blockIdx % 8) and a hand-over word between two workgroups of ONE XCD costs
  // 0.59 - 0.64 us, between two XCDs 0.76 (tools/ubench/ll_hop.hip): the serial
  // chain's hand-overs -- draw -> segment 0 -> ... -> last segment -- stay on
  // one XCD: those nseg + 1 roles take the blocks 0, 8, 16, ..., the mat-vec
  // roles (skip | post1 | logits) the rest in order.  Roles are numbered
  // chain segments, skip, post1, logits, draw.
  int role = fgp_role_of_block((int)blockIdx.x, (int)gridDim.x, nseg);
#ifdef FGP_STAMPS
#define PSTAMP(slot) if (a.dbg && (tid & 63) == 0) a.dbg[(size_t)(slot)] = __builtin_amdgcn_s_memrealtime()
#else
#define PSTAMP(slot)
#endif

  if (role < nseg) {
    // ------------------------------------------------------------ chain segment
    const int seg = role;
    const int l0 = __builtin_amdgcn_readfirstlane((int)((long)seg * L / nseg));
    const int l1 = (int)((long)(seg + 1) * L / nseg);
    const int nl = __builtin_amdgcn_readfirstlane(l1 - l0);
    float* wres = lds;                                 // [nl][FGP_BLK]: weights | {pre, row, bd, -}
    int* meta = reinterpret_cast<int*>(wres + (size_t)nl * FGP_BLK);   // [nl] ring offset (rows), [nl] dilation
    int* flags = meta + 2 * FGP_SEGL;                  // [0] pre ready for step, [1] chain done with step
    // resident weights in LANE order: chunk c of the chain lane's row at float4
    // [c][lane] (filter | gate rows: c < 8; dense rows: [8 + cc][lane], lane =
    // 32 (c >> 2) + n), so a layer's reads are one lane address plus
    // immediates and conflict-free (the ring-slot image is [matrix][n][chunk ^ (n & 7)])
    for (int i = tid; i < nl * FGC_CW / 4; i += FGP_THREADS) {
      const int ll = i / (FGC_CW / 4), q = i % (FGC_CW / 4);
      const int m = q >> 8, n = (q >> 3) & 31, c = (q & 7) ^ (n & 7);
      const int dst = m < 2 ? c * 64 + m * 32 + n : 512 + (c & 3) * 64 + (c >> 2) * 32 + n;
      reinterpret_cast<f32x4*>(wres)[ll * (FGP_BLK / 4) + dst] =
          reinterpret_cast<const f32x4*>(g.cw_img + (size_t)l0 * FGC_CW)[i];
    }
    if (tid < nl) {
      int ro = 0;
      for (int q = 0; q < l0 + tid; ++q) ro += g.dil[q];
      meta[tid] = ro;
      meta[FGP_SEGL + tid] = g.dil[l0 + tid];
    }
    __syncthreads();
    for (int i = tid; i < nl * 64; i += FGP_THREADS) {
      const int ll = i >> 6, ln = i & 63;
      f32x4 ms;
      ms[0] = g.pre[(size_t)l0 * 64 + i];
      ms[1] = __int_as_float((meta[ll] + base % meta[FGP_SEGL + ll]) * 32);
      ms[2] = g.use_dense_bias
                  ? g.layer0[(size_t)(l0 + ll) * g.layer_stride + LAYER_OFF_BD + (ln & 31)] : 0.f;
      ms[3] = 0.f;
      reinterpret_cast<f32x4*>(wres)[ll * (FGP_BLK / 4) + 768 + ln] = ms;
    }
    if (tid == 0) { flags[0] = 1; flags[1] = 0; flags[2] = 0; flags[3] = 0; }
    __syncthreads();
    const int nn = lane & 31;
    if (wave == 0) {
      // ---- the serial chain of this segment (no workgroup barrier, weights
      // resident): fgp_chain_layers
      const int prev_code = g.cursors[1];
      fgp_ll_t* zrow = zll + l0 * 32 + nn;
      fgp_ll_t* xo = seg + 1 < nseg ? xll + seg * 32 : nullptr;
      for (int i = 0; i < n_steps; ++i) {
        const unsigned step = (unsigned)(i + 1);
        // this step's past-tap pre-activations are in LDS (helper waves): the
        // first layer's operands are requested before x is waited for
        while (!dead && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < i + 1)
          __builtin_amdgcn_s_sleep(1);
        FgpLW wa;
        fgp_lw_load(wa, wres, 0, lane);
        float x = 0.f;
        if (seg == 0) {
          // (x lives in EVERY lane: channel lane & 31)
          if (i > 0) {
            // the draw workgroup's x0 of this step (fg_draw_wg256)
            x = fgp_get(x0ll + nn, (unsigned)i, sync, dead);
          } else {
            const int code = g.samples[0];
            float v = 0.f;
            if (prev_code >= 0 && prev_code < Q) v = g.causal[(long)prev_code * 32 + nn];
            if (code >= 0 && code < Q) v += g.causal[((long)Q + code) * 32 + nn];
            x = v;
          }
          PSTAMP(i * 16 + 0);
        } else {
          x = fgp_get(xll + (seg - 1) * 32 + nn, step, sync, dead);
        }
        PSTAMP(i * 16 + 1 + seg);
        if (nl == 10) x = fgp_chain_layers<10>(x, wres, wa, nl, l0, L, g.state, zrow, xo, step, lane);
        else x = fgp_chain_layers<0>(x, wres, wa, nl, l0, L, g.state, zrow, xo, step, lane);
        PSTAMP(i * 16 + 6 + seg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(flags + 1, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      // ---- helper waves: the NEXT step's past-tap pre-activations of this
      // segment's layers (model.py:335-338, the `state` half of the conv):
      // pre[l][n] = bias_fg[l][n] + sum_k x_l[t + 1 - d_l][k] * W[0][k][n],
      // one wave per layer, lane = output n (filter | gate); the queue entry
      // is read with device-scope loads after the chain wave's flag (its
      // write-through stores were acknowledged before it set the flag)
      const int hw = wave - 1;                          // 0..3
      for (int i = 0; i + 1 < n_steps; ++i) {
        while (!dead && __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < i + 1) {
          __builtin_amdgcn_s_sleep(4);
          if (__hip_atomic_load(sync + FGP_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
        }
        const int tpos = base + i + 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ll = hw; ll < nl; ll += 4) {
          const int l = l0 + ll, d = meta[FGP_SEGL + ll];
          const float xv = lane < 32 ? fgp_ld(g.state + ((long)meta[ll] + tpos % d) * 32 + lane) : 0.f;
          const float* W = g.layer0 + (long)l * g.layer_stride + (lane < 32 ? 0 : 2 * 1024) + (lane & 31);
          float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            p0 = fmaf(__shfl(xv, k), W[k * 32], p0);
            p1 = fmaf(__shfl(xv, 8 + k), W[(8 + k) * 32], p1);
            p2 = fmaf(__shfl(xv, 16 + k), W[(16 + k) * 32], p2);
            p3 = fmaf(__shfl(xv, 24 + k), W[(24 + k) * 32], p3);
          }
          acc[(ll - hw) >> 2] = (g.bias_fg ? g.bias_fg[l * 64 + lane] : 0.f) + ((p0 + p1) + (p2 + p3));
        }
        // (the chain wave is past this segment's layers of step i: the
        // {pre, row} words of the layers' blocks are free)
        for (int ll = hw; ll < nl; ll += 4) {
          float* ms = wres + (size_t)ll * FGP_BLK + 3072 + lane * 4;
          ms[0] = acc[(ll - hw) >> 2];
          ms[1] = __int_as_float((meta[ll] + tpos % meta[FGP_SEGL + ll]) * 32);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // four helper waves: the last one to finish raises the step
        if (lane == 0) {
          const int old = __hip_atomic_fetch_add(flags + 2, 1, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old == 3) {
            __hip_atomic_store(flags + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(flags, i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    return;
  }
  role -= nseg;
  if (tid >= 256) return;          // the tail roles are 256-thread workgroups
  float* in_s = lds;               // staged input vector
  const int o = tid & 15, part = tid >> 4;
  if (role < nsk) {
    // ------------------------------------------------------------------- skip
    const int KK = L * 32, col0 = role * 16;
    float* w_s = lds + ((KK + 3) & ~3);               // [KK][16]
    float* red = w_s + (size_t)KK * 16;               // [16][16]
    for (int i = tid; i < KK * 16; i += 256) {
      const int k = i >> 4, c = i & 15;
      w_s[i] = col0 + c < S ? g.skip_w[(size_t)k * S + col0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = 0; i < n_steps; ++i) {
      const unsigned step = (unsigned)(i + 1);
      float accv = 0.f;
      for (int sg = 0; sg < nseg; ++sg) {
        const int l0 = (int)((long)sg * L / nseg), l1 = (int)((long)(sg + 1) * L / nseg);
        const int n = (l1 - l0) * 32;
        fgp_get2(in_s + l0 * 32, zll + l0 * 32, tid, n, step, sync, dead);
        __syncthreads();
        accv += fgp_mv16(in_s + l0 * 32, w_s + (size_t)l0 * 32 * 16, n, o, part, 16);
      }
      red[part * 16 + o] = accv;
      __syncthreads();
      if (part == 0) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += red[p * 16 + o];
        if (col0 + o < S)
          fgp_put(h1ll + col0 + o, fmaxf((g.skip_bsum ? g.skip_bsum[col0 + o] : 0.f) + t, 0.f), step);
      }
      if (role == 0) { PSTAMP(i * 16 + 11); }
      __syncthreads();             // (red is rewritten in the next step)
    }
    return;
  }
  role -= nsk;
  if (role < nsk + nlg) {
    // ------------------------------------------------------- post1 / logits
    const bool lg = role >= nsk;
    fgp_post_role(lg ? g.post2_w : g.post1_w, lg ? g.post2_b : g.post1_b,
                  (lg ? role - nsk : role) * 16, lg ? Q : S, S, lg, lg ? h2ll : h1ll,
                  lg ? lgll : h2ll, n_steps, lds, sync, dead, tid,
                  a.dbg && role == 0 ? a.dbg + 12 : (a.dbg && role == nsk ? a.dbg + 13 : nullptr));
    return;
  }
  // ---------------------------------------------------------------------- draw
  // (256 threads: one logit per thread for Q <= 256)
  float* lgs = lds;                                             // [Q] the step's logits
  double* dpart = reinterpret_cast<double*>(lds + ((Q + 3) & ~3));  // 16 doubles: per-wave partials
  int* nxt = reinterpret_cast<int*>(dpart + 16);                 // [2] the code drawn at step parity
  float* ctab = reinterpret_cast<float*>(nxt + 8);               // [2][Q][32] the causal layer's filter
  for (int i = tid; i < 2 * Q * 32; i += 256) ctab[i] = g.causal[i];
  const FgDrawCtl dc = fg_draw_ctl(g);
  __syncthreads();
  for (int i = 0; i < n_steps; ++i) {
    const unsigned step = (unsigned)(i + 1);
    fgp_get2(lgs, lgll, tid, Q, step, sync, dead);
    PSTAMP(i * 16 + 14);
    __syncthreads();
    // the code this step consumes (drawn / given one step ago: written behind
    // the previous step's last barrier, read behind this one)
    const int cur_code = i == 0 ? g.samples[0] : nxt[i & 1];
    // (dpart[] is rewritten a step later only after this barrier, which every
    // wave reaches after its last read of the step before)
    // the drawing thread publishes the code of step i + 1 for segment 0
    if (Q <= 256) fg_draw_wg256<1>(g, dc, lgs, dpart, nxt, x0ll, ctab, cur_code, step, i + 1 < n_steps, tid, base + i);
    else fg_draw_wg256<2>(g, dc, lgs, dpart, nxt, x0ll, ctab, cur_code, step, i + 1 < n_steps, tid, base + i);
    PSTAMP(i * 16 + 15);
  }
  __syncthreads();
  // the code the last step consumed (this CU's L1 may hold an older copy of the
  // samples line: the draws were kept in LDS)
  const int cur_code = n_steps >= 2 ? nxt[(n_steps - 1) & 1] : g.samples[0];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // cursors as wn_fastgen_finish leaves them: {steps done, the last code consumed, nothing pending}
  if (tid == 0) {
    g.cursors[0] = base + n_steps;
    g.cursors[1] = cur_code;
    g.cursors[2] = 0;
  }
#undef PSTAMP
}

// ===========================================================================
// Wide fast generation: the same incremental generator for MORE than 32
// residual / dilation channels (C = 32 * blocks padded channels, weights in the
// reference's [K][C][C] layout of wavenet/blocked.py; the reference's
// generator has no width limit, model.py:444-516).  One persistent workgroup,
// correctness first: per layer three phases separated by workgroup barriers
//   1. thread (which, c): a_which[c] = bias + st . W_which[0][:, c] + x . W_which[1][:, c]
//   2. z = tanh(a_f) sigmoid(a_g); the queue entry is replaced by x (push)
//   3. x += bd + z . Wd  (every layer, as model.py:377-380); total += z . Ws_l
// then the post-processing mat-vecs, the float64 softmax and the same
// counter-based draw as fastgen_kernel (same seed -> same uniform per step).
// Queues: layer l's ring holds d_l rows of C floats at state + roff[l] * C.
// ===========================================================================
#define FGW_THREADS 256
#define FGW_MAXC 1024
#define FGW_MAXS 4096
#define FGW_MAXQ 4096
#define FGW_MAXL 1024
#define FGW_XPT (FGW_MAXC / FGW_THREADS)   // residual channels per thread
#define FGW_SPT (FGW_MAXS / FGW_THREADS)   // skip channels per thread

// dynamic LDS of the wide generator: Q doubles, then 5 C + 2 S floats, 3 L ints
static size_t fgw_lds_bytes(int C, int S, int Q, int L) {
  return (size_t)Q * 8 + ((size_t)5 * C + 2 * (size_t)S + 3 * (size_t)L) * 4;
}

// sum_k in[k] * w[k * stride]: the weight loads in batches (all of a batch in
// flight before its FMAs: 64, then 16, 4 and 1 for what is left) into four
// partial sums -- a plain loop waits for every load before the next one is
// issued, and the wide generator is nothing but such loops (64 channels,
// default stack: 3.5 ms per sample with one load per FMA, 0.77 / 0.50 / 0.37 /
// 0.31 ms with batches of 8 / 16 / 32 / 64).
#define FGW_BATCH 64
__device__ __forceinline__ float fgw_dot(const float* in, const float* __restrict__ w,
                                         long stride, int K, float init) {
  float c[4] = {init, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + FGW_BATCH <= K; k += FGW_BATCH) {
    float wv[FGW_BATCH];
#pragma unroll
    for (int u = 0; u < FGW_BATCH; ++u) wv[u] = w[(long)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < FGW_BATCH; ++u) c[u & 3] = fmaf(in[k + u], wv[u], c[u & 3]);
  }
  for (; k + 16 <= K; k += 16) {
    float wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = w[(long)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 16; ++u) c[u & 3] = fmaf(in[k + u], wv[u], c[u & 3]);
  }
  for (; k + 4 <= K; k += 4) {
    float wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = w[(long)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = fmaf(in[k + u], wv[u], c[u]);
  }
  for (; k < K; ++k) c[0] = fmaf(in[k], w[(long)k * stride], c[0]);
  return (c[0] + c[1]) + (c[2] + c[3]);
}

// two such sums at once (both batches of weight loads in flight together)
__device__ __forceinline__ void fgw_dot2(const float* in0, const float* __restrict__ w0,
                                         const float* in1, const float* __restrict__ w1,
                                         long stride, int K, float& r0, float& r1) {
  float a[4] = {r0, 0.f, 0.f, 0.f}, b[4] = {r1, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 32 <= K; k += 32) {
    float u0[32], u1[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) u0[u] = w0[(long)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 32; ++u) u1[u] = w1[(long)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 32; ++u) a[u & 3] = fmaf(in0[k + u], u0[u], a[u & 3]);
#pragma unroll
    for (int u = 0; u < 32; ++u) b[u & 3] = fmaf(in1[k + u], u1[u], b[u & 3]);
  }
  r0 = (a[0] + a[1]) + (a[2] + a[3]);
  r1 = (b[0] + b[1]) + (b[2] + b[3]);
  if (k < K) {
    r0 = fgw_dot(in0 + k, w0 + (long)k * stride, stride, K - k, r0);
    r1 = fgw_dot(in1 + k, w1 + (long)k * stride, stride, K - k, r1);
  }
}

struct FastGenWide {
  FastGen g;
  int C;        // padded channels (multiple of 32, <= FGW_MAXC)
  // cooperative launch (64 channels): FGP_WORDS sync words, then the hand-over
  // words z [L][64] | h1 [S] | h2 [S] | logits [Q]; zero before the launch
  unsigned* sync;
  fgp_ll_t* ll;
};

// Cooperative 64-channel generator (round 5).  The single workgroup above pulls
// 10.4 MB of weights per sample through ONE CU: 208 KB a layer, 128 KB of them
// the layer's skip weights, and 1.5 MB for the two post-processing mat-vecs.
// Here workgroup 0 keeps the serial part -- the layers' filter / gate / dense
// mat-vecs and the draw -- and publishes every layer's z as tagged hand-over
// words; S / 16 skip workgroups add z_l Ws_l[:, their 16 columns] as the words
// arrive (their weights from their own CUs' L2 share, a layer ahead), then
// S / 16 + Q / 16 post-processing workgroups (fgp_post_role, weights in
// registers) turn h1 into the logits, which workgroup 0 polls.  Same
// arithmetic per output as the single workgroup up to the order of the skip
// sum's additions.  Every workgroup must be resident (checked by the host).

// F64 (64 channels, at most 512 skip channels): the generic layer body pays
// three dependent weight-load latencies per layer (past-tap ring entry ->
// filter / gate weights -> dense + skip weights; 213 KB a layer through ONE
// CU, 10.6 MB a sample through one XCD's 4 MB L2: 0.30 ms per sample).  None of
// the weights depends on the data: here (512 threads) EVERY load of a layer --
// the ring entry, a thread's 32 filter / gate weights (output o = tid & 127, a
// quarter of the contraction each), its 8 dense weights (an eighth of the
// contraction) a LAYER AHEAD, the 64 weights of its skip column, which the
// layer needs last, at the layer's own top -- and the phases only synchronise
// through LDS, four barriers a layer (207 us per sample with every load at the
// layer's own top and 256 threads; without the skip weights' loads a sample
// takes 144 us: they are the 54 us still exposed).
struct FgwLayer64 {
  float ringv, wa[16], wb[16], wdv[8], bf, bg, bdv;
};
template <bool F64, bool COOP>
__global__ __launch_bounds__(F64 ? 512 : FGW_THREADS, 1) void fastgen_wide_kernel(FastGenWide a) {
  constexpr int NT = F64 ? 512 : FGW_THREADS;    // threads
  constexpr int SPT = FGW_MAXS / NT, XPT = FGW_MAXC / NT;
  const FastGen& g = a.g;
  extern __shared__ double fgw_lds[];
  const int S = g.S, Q = g.Q, L = g.L, C = a.C;
  double* pd = fgw_lds;                           // [Q]
  float* xs = reinterpret_cast<float*>(pd + Q);   // [C]
  float* sts = xs + C;                            // [C]
  float* zs = sts + C;                            // [C]
  float* apre = zs + C;                           // [2 C]
  float* hbuf = apre + 2 * C;                     // [S]
  float* h2buf = hbuf + S;                        // [S]
  int* pos = reinterpret_cast<int*>(h2buf + S);   // [L]
  int* sdil = pos + L;                            // [L]
  int* roff = sdil + L;                           // [L]
  __shared__ int s_code;
  __shared__ float f64_part[F64 ? 4 * 128 + 8 * 64 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long CC = (long)C * C;
  fgp_ll_t* zll = a.ll;                            // [L][C]
  fgp_ll_t* h1ll = COOP ? zll + (size_t)L * C : nullptr;
  fgp_ll_t* h2ll = COOP ? h1ll + S : nullptr;
  fgp_ll_t* lgll = COOP ? h2ll + S : nullptr;
  bool dead = false;
  if (COOP && blockIdx.x > 0) {
    if (tid >= 256) return;                        // the workers are 256-thread roles
    float* wlds = reinterpret_cast<float*>(fgw_lds);
    const int nsk = (S + 15) / 16;
    int role = blockIdx.x - 1;
    if (role < nsk) {
      // ---- skip worker: columns col0 .. col0 + 15; thread (o, part) adds the
      // products of z_l[k], k = part, part + 16, ... for every layer, the
      // layer's weights requested a layer ahead (C / 16 <= 64 per thread)
      const int o = tid & 15, part = tid >> 4, col = role * 16 + o;
      float* zb = wlds;                            // [2][C]
      float* red = wlds + 2 * C;                   // [16][16]
      const bool live = col < S;
      const int npt = C >> 4;                      // inputs per thread and layer (C % 32 == 0)
      const float bsum = (live && g.skip_bsum) ? g.skip_bsum[col] : 0.f;
      for (int step = 0; step < g.n_steps; ++step) {
        const unsigned tag = (unsigned)(step + 1);
        float c0 = 0.f, c1 = 0.f;
        // (two inputs per pass: npt is even)
        float wc[2], wn[2];
        auto wload = [&](int l, int i, float (&w)[2]) {
          w[0] = live ? g.skip_w[((size_t)l * C + part + 16 * i) * S + col] : 0.f;
          w[1] = live ? g.skip_w[((size_t)l * C + part + 16 * (i + 1)) * S + col] : 0.f;
        };
        wload(0, 0, wc);
        for (int l = 0; l < L; ++l) {
          float* zl = zb + (l & 1) * C;
          for (int c = tid; c < C; c += 256)
            zl[c] = fgp_get(zll + (size_t)l * C + c, tag, a.sync, dead);
          __syncthreads();
          for (int i = 0; i < npt; i += 2) {
            // the next pair's weights (of this layer, or the first of the next)
            const bool more = i + 2 < npt;
            if (more) wload(l, i + 2, wn);
            else if (l + 1 < L) wload(l + 1, 0, wn);
            c0 = fmaf(zl[part + 16 * i], wc[0], c0);
            c1 = fmaf(zl[part + 16 * (i + 1)], wc[1], c1);
            wc[0] = wn[0]; wc[1] = wn[1];
          }
        }
        red[part * 16 + o] = c0 + c1;
        __syncthreads();
        if (part == 0 && live) {
          float t = 0.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) t += red[q * 16 + o];
          fgp_put(h1ll + col, fmaxf(bsum + t, 0.f), tag);
        }
        __syncthreads();                           // (red is rewritten in the next step)
      }
      return;
    }
    role -= nsk;
    const bool lg = role >= nsk;
    fgp_post_role(lg ? g.post2_w : g.post1_w, lg ? g.post2_b : g.post1_b,
                  (lg ? role - nsk : role) * 16, lg ? Q : S, S, lg, lg ? h2ll : h1ll,
                  lg ? lgll : h2ll, g.n_steps, wlds, a.sync, dead, tid, nullptr);
    return;
  }
  const int steps_done = g.cursors[0];
  int prev_code = g.cursors[1];
  if (tid == 0) s_code = g.samples[0];
  for (int l = tid; l < L; l += NT) {
    sdil[l] = g.dil[l];
    pos[l] = steps_done % g.dil[l];
  }
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int l = 0; l < L; ++l) { roff[l] = off; off += sdil[l]; }
  }
  __syncthreads();

  for (int step = 0; step < g.n_steps; ++step) {
    const int code = s_code;
    const long tpos = (long)steps_done + step;
    float acc[SPT];
#pragma unroll
    for (int o = 0; o < SPT; ++o) acc[o] = 0.f;
    // causal layer: one-hot input = two table rows (model.py:341-346)
    for (int c = tid; c < C; c += NT) {
      float v = 0.f;
      if (prev_code >= 0 && prev_code < Q) v = g.causal[(long)prev_code * C + c];
      if (code >= 0 && code < Q) v += g.causal[((long)Q + code) * C + c];
      xs[c] = v;
    }
    __syncthreads();
    if (F64) {
      const int o1 = tid & 127, kq = tid >> 7;       // filter / gate output, quarter of K
      const int cd = tid & 63, k8 = tid >> 6;        // dense output, eighth of K
      const int c0 = tid < S ? tid : 0;              // skip column
      auto request = [&](int l) {
        FgwLayer64 w;
        const float* blk = g.layer0 + (long)l * g.layer_stride;
        w.ringv = tid < 64 ? g.state[((long)roff[l] + pos[l]) * 64 + tid] : 0.f;
        const float* w0 = blk + (long)(o1 >> 6) * 2 * 4096 + (o1 & 63) + kq * 16 * 64;
#pragma unroll
        for (int u = 0; u < 16; ++u) { w.wa[u] = w0[u * 64]; w.wb[u] = w0[4096 + u * 64]; }
        const float* wd = blk + 4 * 4096 + cd + k8 * 8 * 64;
#pragma unroll
        for (int u = 0; u < 8; ++u) w.wdv[u] = wd[u * 64];
        w.bf = tid < 64 && g.bias_fg ? g.bias_fg[(long)l * 128 + tid] : 0.f;
        w.bg = tid < 64 && g.bias_fg ? g.bias_fg[(long)l * 128 + 64 + tid] : 0.f;
        w.bdv = tid < 64 && g.use_dense_bias ? blk[5 * 4096 + 128 + tid] : 0.f;
        return w;
      };
      FgwLayer64 cur = request(0);
      if (tid < 64) sts[tid] = cur.ringv;
      __syncthreads();
      // (four workgroup barriers a layer: partial sums | gate | partial sums |
      // x' and the next layer's past tap)
      for (int l = 0; l < L; ++l) {
        // (the skip column's weights, which the layer needs last, at its own
        // top: requesting them a layer ahead -- into the registers the layer
        // before has just multiplied out of -- sends the register allocator
        // into 119 spills and the sample to 284 us)
        float ws0[COOP ? 1 : 64];
        if (!COOP) {
          const float* ws = g.skip_w + (long)l * 64 * S + c0;
#pragma unroll
          for (int u = 0; u < 64; ++u) ws0[u] = ws[(long)u * S];
        }
        FgwLayer64 nxt = cur;
        if (l + 1 < L) nxt = request(l + 1);
        float* ring = g.state + ((long)roff[l] + pos[l]) * 64;
        // ---- 1. filter / gate pre-activations, a quarter of the contraction per thread
        {
          float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
          for (int u = 0; u < 16; u += 2) {
            p0 = fmaf(sts[kq * 16 + u], cur.wa[u], p0);
            p1 = fmaf(xs[kq * 16 + u], cur.wb[u], p1);
            p2 = fmaf(sts[kq * 16 + u + 1], cur.wa[u + 1], p2);
            p3 = fmaf(xs[kq * 16 + u + 1], cur.wb[u + 1], p3);
          }
          f64_part[kq * 128 + o1] = (p0 + p1) + (p2 + p3);
        }
        __syncthreads();
        // ---- 2. gate; enqueue x_l[t] in place of the entry just read
        if (tid < 64) {
          const float af = cur.bf + ((f64_part[tid] + f64_part[128 + tid]) +
                                     (f64_part[256 + tid] + f64_part[384 + tid]));
          const float ag = cur.bg + ((f64_part[64 + tid] + f64_part[192 + tid]) +
                                     (f64_part[320 + tid] + f64_part[448 + tid]));
          const float zv = wn_tanh(af) * wn_sigmoid(ag);
          zs[tid] = zv;
          // (cooperative launch: the skip workgroups take it from here)
          if (COOP) fgp_put(zll + (size_t)l * 64 + tid, zv, (unsigned)(step + 1));
          if (g.push) ring[tid] = xs[tid];
        }
        __syncthreads();
        // ---- 3. residual 1x1 conv (an eighth of the contraction per thread)
        // and this layer's skip contribution
        {
          float d0 = 0.f, d1 = 0.f;
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            d0 = fmaf(zs[k8 * 8 + u], cur.wdv[u], d0);
            d1 = fmaf(zs[k8 * 8 + u + 1], cur.wdv[u + 1], d1);
          }
          f64_part[512 + k8 * 64 + cd] = d0 + d1;
          if (!COOP) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int u = 0; u < 64; u += 4) {
              a0 = fmaf(zs[u], ws0[u], a0);
              a1 = fmaf(zs[u + 1], ws0[u + 1], a1);
              a2 = fmaf(zs[u + 2], ws0[u + 2], a2);
              a3 = fmaf(zs[u + 3], ws0[u + 3], a3);
            }
            if (tid < S) acc[0] += (a0 + a1) + (a2 + a3);
          }
        }
        __syncthreads();
        if (tid < 64) {
          float t = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) t += f64_part[512 + q * 64 + tid];
          xs[tid] = xs[tid] + (cur.bdv + t);
          sts[tid] = nxt.ringv;          // (the next layer's past tap)
        }
        __syncthreads();
        cur = nxt;
      }
    }
    for (int l = 0; !F64 && l < L; ++l) {
      const float* blk = g.layer0 + (long)l * g.layer_stride;
      float* ring = g.state + ((long)roff[l] + pos[l]) * C;
      for (int c = tid; c < C; c += NT) sts[c] = ring[c];
      __syncthreads();
      // ---- 1. filter / gate pre-activations
      for (int o = tid; o < 2 * C; o += NT) {
        const int which = o / C, c = o - which * C;
        const float* w0 = blk + (long)which * 2 * CC + c;   // W_which[0][:, c]
        const float* w1 = w0 + CC;                          // W_which[1][:, c]
        float s0 = g.bias_fg ? g.bias_fg[(long)l * 2 * C + o] : 0.f, s1 = 0.f;
        fgw_dot2(sts, w0, xs, w1, C, C, s0, s1);
        apre[o] = s0 + s1;
      }
      __syncthreads();
      // ---- 2. gate; enqueue x_l[t] in place of the entry just read
      for (int c = tid; c < C; c += NT) {
        const float zv = wn_tanh(apre[c]) * wn_sigmoid(apre[C + c]);
        zs[c] = zv;
        if (COOP) fgp_put(zll + (size_t)l * C + c, zv, (unsigned)(step + 1));
        if (g.push) ring[c] = xs[c];
      }
      __syncthreads();
      // ---- 3. residual 1x1 conv and this layer's skip contribution
      float xn[XPT];
#pragma unroll
      for (int u = 0; u < XPT; ++u) {
        const int c = tid + u * NT;
        xn[u] = 0.f;
        if (c < C) {
          const float* wd = blk + 4 * CC + c;
          const float d0 = fgw_dot(zs, wd, C, C, g.use_dense_bias ? blk[5 * CC + 2 * C + c] : 0.f);
          xn[u] = xs[c] + d0;
        }
      }
      if (!COOP) {
        const float* ws = g.skip_w + (long)l * C * S;
#pragma unroll
        for (int oi = 0; oi < SPT; oi += 2) {
          const int sc = tid + oi * NT;
          if (sc + NT < S)             // this thread's next two columns
            fgw_dot2(zs, ws + sc, zs, ws + sc + NT, S, C, acc[oi], acc[oi + 1]);
          else if (sc < S)
            acc[oi] = fgw_dot(zs, ws + sc, S, C, acc[oi]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < XPT; ++u)
        if (tid + u * NT < C) xs[tid + u * NT] = xn[u];
      __syncthreads();
    }
    if (g.push) {
      for (int l = tid; l < L; l += NT) {
        const int p = pos[l] + 1;
        pos[l] = p == sdil[l] ? 0 : p;
      }
    }
    // ---- post-processing (model.py:505-514)
    if (COOP) {
      // (the worker workgroups: skip sum -> postprocess1 -> logits)
      for (int q = tid; q < Q; q += NT)
        pd[q] = (double)fgp_get(lgll + q, (unsigned)(step + 1), a.sync, dead);
      __syncthreads();
    } else {
      {
#pragma unroll
        for (int oi = 0; oi < SPT; ++oi) {
          const int sc = tid + oi * NT;
          if (sc < S)
            hbuf[sc] = fmaxf(acc[oi] + (g.skip_bsum ? g.skip_bsum[sc] : 0.f), 0.f);
        }
      }
      __syncthreads();
      for (int sc = tid; sc < S; sc += NT) {
        const float c0 = fgw_dot(hbuf, g.post1_w + sc, S, S, g.post1_b ? g.post1_b[sc] : 0.f);
        h2buf[sc] = fmaxf(c0, 0.f);
      }
      __syncthreads();
      for (int q = tid; q < Q; q += NT) {
        const float c0 = fgw_dot(h2buf, g.post2_w + q, Q, S, g.post2_b ? g.post2_b[q] : 0.f);
        pd[q] = (double)c0;
      }
      __syncthreads();
    }
    // ---- softmax in float64, temperature, draw: as fastgen_kernel
    if (wave == 0) {
      double m = -1e300;
      for (int q = lane; q < Q; q += 64) m = fmax(m, pd[q]);
      for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
      double se = 0.0;
      for (int q = lane; q < Q; q += 64) {
        const double e = exp(pd[q] - m);
        pd[q] = e;
        se += e;
      }
      for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
      const bool want_p = g.proba_out && (step % g.proba_every == 0);
      float* po = want_p ? g.proba_out + (long)(step / g.proba_every) * Q : nullptr;
      for (int q = lane; q < Q; q += 64) {
        const float p32 = (float)(pd[q] / se);
        if (po) po[q] = p32;
        pd[q] = (double)p32;
      }
    }
    __syncthreads();
    if (step + 1 >= g.n_given) {
      if (wave == 0) {
        const double tau = (double)g.temperature;
        if (g.temperature != 1.0f) {
          double mx = -1e300;
          for (int q = lane; q < Q; q += 64) {
            const double lp = log(pd[q] > 0.0 ? pd[q] : 1e-300) / tau;
            pd[q] = lp;
            mx = fmax(mx, lp);
          }
          for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
          for (int q = lane; q < Q; q += 64) pd[q] = exp(pd[q] - mx);
        }
        __builtin_amdgcn_wave_barrier();
        const int per = (Q + 63) / 64;
        const int q0 = lane * per, q1 = min(Q, q0 + per);
        double seg = 0.0;
        for (int q = q0; q < q1; ++q) seg += pd[q];
        double incl = seg;
        for (int o = 1; o < 64; o <<= 1) {
          const double v = __shfl_up(incl, o);
          if (lane >= o) incl += v;
        }
        const double total = __shfl(incl, 63);
        const uint64_t r = splitmix64(g.seed ^ splitmix64((uint64_t)tpos));
        const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0) * total;
        // (the neighbour's inclusive sum, not incl - seg: the lanes' intervals
        // then tile [0, total) exactly -- no gap a draw could fall into)
        const double up = __shfl_up(incl, 1);
        const double excl = lane == 0 ? 0.0 : up;
        int pick = -1;
        if (u >= excl && u < incl) {
          double c = excl;
          pick = q1 - 1;
          for (int q = q0; q < q1; ++q) {
            c += pd[q];
            if (u < c) { pick = q; break; }
          }
        }
        int best = pick;
        for (int o = 32; o >= 1; o >>= 1) best = max(best, __shfl_xor(best, o));
        if (best < 0) best = Q - 1;
        if (lane == 0) {
          g.samples[step + 1] = best;
          s_code = best;
        }
      }
    } else if (tid == 0) {
      s_code = g.samples[step + 1];
    }
    prev_code = code;
    __syncthreads();
  }
  if (tid == 0 && g.push) {
    g.cursors[0] = steps_done + g.n_steps;
    g.cursors[1] = prev_code;
  }
}


#ifdef FGP_STAMPS
static unsigned long long* g_fgp_dbg = nullptr;   // [n_steps][16] (diagnostic build only)
extern "C" int wn_diag_fgp_dbg(unsigned long long* p) { g_fgp_dbg = p; return WN_OK; }
#endif

// A launch whose workgroups wait for each other (hand-over words) must have
// them all resident at once.  hipLaunchCooperativeKernel makes that the
// RUNTIME's promise: it refuses (an error, nothing has run, the caller takes
// its single-workgroup / step-kernel path at once) where a plain launch would
// start a partial grid that spins until its 2 s bounded waits expire.
template <typename Arg>
static int fg_launch_cooperative(const void* kernel, int wgs, int threads, size_t lds, hipStream_t s,
                                 Arg& a) {
  void* args[] = {(void*)&a};
  const hipError_t e = hipLaunchCooperativeKernel(kernel, dim3(wgs), dim3(threads), args,
                                                  (unsigned)lds, s);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return e == hipErrorCooperativeLaunchTooLarge ? WN_ERR_UNSUPPORTED : WN_ERR_LAUNCH;
  }
  return wn_check_launch();
}

extern "C" {

long wn_fastgen_state_floats(const int32_t* dilations_host, int L) {
  if (!dilations_host || L <= 0) return WN_ERR_NULL;
  long n = 0;
  for (int l = 0; l < L; ++l) n += (long)dilations_host[l] * 32;
  return n;
}

int wn_fastgen_init(float* state, long state_floats, int32_t* cursors, int L,
                    void* stream) {
  if (!state || !cursors) return WN_ERR_NULL;
  if (state_floats <= 0 || L <= 0) return WN_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(state, 0, state_floats * sizeof(float), s) != hipSuccess)
    return WN_ERR_LAUNCH;
  const int32_t init[2] = {0, -1};
  // two 4-byte async sets (no host buffer lifetime issue)
  if (hipMemsetD32Async((hipDeviceptr_t)cursors, (int)init[0], 1, s) != hipSuccess)
    return WN_ERR_LAUNCH;
  if (hipMemsetD32Async((hipDeviceptr_t)(cursors + 1), (int)init[1], 1, s) != hipSuccess)
    return WN_ERR_LAUNCH;
  if (hipMemsetD32Async((hipDeviceptr_t)(cursors + 2), 0, 1, s) != hipSuccess)
    return WN_ERR_LAUNCH;
  return WN_OK;
}

int wn_fastgen_run(const float* params_causal, const float* layer0,
                   long layer_stride, const float* skip_w, const float* skip_bsum,
                   const float* post1_w, const float* post1_b,
                   const float* post2_w, const float* post2_b,
                   const float* gc_bias_fg, const int32_t* dilations_dev,
                   int L, int S, int Q, float* state, int32_t* cursors,
                   int32_t* samples_io, int n_given, int n_steps,
                   float temperature, uint64_t seed, float* proba_out,
                   int proba_every, int use_biases, int push, void* stream) {
  if (!params_causal || !layer0 || !skip_w || !post1_w || !post2_w ||
      !dilations_dev || !state || !cursors || !samples_io)
    return WN_ERR_NULL;
  if (L <= 0 || S <= 0 || Q <= 0 || n_steps <= 0 || n_given < 1)
    return WN_ERR_BAD_SHAPE;
  if (S > FG_MAXS || Q > FG_MAXQ || L > FG_MAXL) return WN_ERR_UNSUPPORTED;
  if (!(temperature > 0.f)) return WN_ERR_BAD_SHAPE;
  FastGen g;
  g.causal = params_causal; g.layer0 = layer0; g.layer_stride = layer_stride;
  g.skip_w = skip_w; g.skip_bsum = skip_bsum; g.post1_w = post1_w;
  g.post1_b = post1_b; g.post2_w = post2_w; g.post2_b = post2_b;
  g.bias_fg = gc_bias_fg; g.dil = dilations_dev; g.L = L; g.S = S; g.Q = Q;
  g.state = state; g.cursors = cursors; g.samples = samples_io;
  g.n_given = n_given; g.n_steps = n_steps; g.temperature = temperature;
  g.seed = seed; g.proba_out = proba_out;
  g.proba_every = proba_every > 0 ? proba_every : 1;
  g.use_dense_bias = use_biases;
  g.push = push;
  if (!push && n_steps != 1) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(fastgen_kernel, dim3(1), dim3(FG_THREADS), 0,
                     (hipStream_t)stream, g);
  return wn_check_launch();
}


// wn_fastgen_run for C = 32 * blocks > 32 padded channels (layer blocks
// Wf[2][C][C] Wg[2][C][C] Wd[C][C] bf[C] bg[C] bd[C], causal [2][Q][C], skip
// [L][C][S], gc_bias_fg [L][2 C], queues of C floats per entry:
// state_floats = sum(dilations) * C).
// Scratch of the cooperative 64-channel launch of wn_fastgen_run_wide (`coop`):
// FGP_WORDS sync words (after a run word [12] != 0 = a hand-over wait expired:
// the samples of that run are not valid) + the hand-over words; 0 = the shape
// has no cooperative launch (S > 512 or not a multiple of 16, Q > 512).
long wn_fastgen_wide_coop_bytes(int L, int C, int S, int Q) {
  if (C < 32 || C > FGW_MAXC || (C & 31) || S <= 0 || S > 512 || (S & 15) || Q <= 0 ||
      Q > 512 || L <= 0 || L > FGW_MAXL)
    return 0;
  return (long)FGP_WORDS * 4 + ((long)L * C + 2L * S + Q) * 8;
}

int wn_fastgen_run_wide(const float* params_causal, const float* layer0,
                        long layer_stride, const float* skip_w,
                        const float* skip_bsum, const float* post1_w,
                        const float* post1_b, const float* post2_w,
                        const float* post2_b, const float* gc_bias_fg,
                        const int32_t* dilations_dev, int L, int C, int S, int Q,
                        float* state, int32_t* cursors, int32_t* samples_io,
                        int n_given, int n_steps, float temperature,
                        uint64_t seed, float* proba_out, int proba_every,
                        int use_biases, int push, void* coop, void* stream) {
  if (!params_causal || !layer0 || !skip_w || !post1_w || !post2_w ||
      !dilations_dev || !state || !cursors || !samples_io)
    return WN_ERR_NULL;
  if (L <= 0 || S <= 0 || Q <= 0 || n_steps <= 0 || n_given < 1 || C <= 0 ||
      C % 32 != 0)
    return WN_ERR_BAD_SHAPE;
  if (S > FGW_MAXS || Q > FGW_MAXQ || L > FGW_MAXL || C > FGW_MAXC)
    return WN_ERR_UNSUPPORTED;
  const size_t lds = fgw_lds_bytes(C, S, Q, L);
  if (lds > 150 * 1024) return WN_ERR_UNSUPPORTED;
  // 64 channels, at most 512 skip channels: every load of a layer requested
  // at its top (other shapes: the generic layer body)
  const bool f64 = C == 64 && S <= 512;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(
        f64 ? reinterpret_cast<const void*>(fastgen_wide_kernel<true, false>)
            : reinterpret_cast<const void*>(fastgen_wide_kernel<false, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return WN_ERR_LAUNCH;
  }
  if (!(temperature > 0.f)) return WN_ERR_BAD_SHAPE;
  if (!push && n_steps != 1) return WN_ERR_BAD_SHAPE;
  FastGenWide a;
  FastGen& g = a.g;
  g.causal = params_causal; g.layer0 = layer0; g.layer_stride = layer_stride;
  g.skip_w = skip_w; g.skip_bsum = skip_bsum; g.post1_w = post1_w;
  g.post1_b = post1_b; g.post2_w = post2_w; g.post2_b = post2_b;
  g.bias_fg = gc_bias_fg; g.dil = dilations_dev; g.L = L; g.S = S; g.Q = Q;
  g.state = state; g.cursors = cursors; g.samples = samples_io;
  g.n_given = n_given; g.n_steps = n_steps; g.temperature = temperature;
  g.seed = seed; g.proba_out = proba_out;
  g.proba_every = proba_every > 0 ? proba_every : 1;
  g.use_dense_bias = use_biases;
  g.push = push;
  a.C = C;
  a.sync = nullptr; a.ll = nullptr;
  hipStream_t s = (hipStream_t)stream;
  // cooperative launch: a shape it covers (pushed runs and the one-step peek
  // alike, so that both give the same bits), with every
  // workgroup resident (asked of the runtime for THIS configuration); anything
  // else -- also a caller without scratch -- takes the single workgroup
  if (coop && wn_fastgen_wide_coop_bytes(L, C, S, Q) > 0) {
    if (!wn_aligned16(coop)) return WN_ERR_MISALIGNED;
    const int wgs = 1 + 2 * ((S + 15) / 16) + (Q + 15) / 16;
    // workers: at most S + 16 S + 256 floats (fgp_post_role); workgroup 0: as above
    size_t wl = ((size_t)17 * S + 512) * 4;              // post roles
    if (wl < ((size_t)2 * C + 256) * 4) wl = ((size_t)2 * C + 256) * 4;   // skip role
    const size_t cl = lds > wl ? lds : wl;
    const void* kf = f64 ? reinterpret_cast<const void*>(fastgen_wide_kernel<true, true>)
                         : reinterpret_cast<const void*>(fastgen_wide_kernel<false, true>);
    const int nthr = f64 ? 512 : FGW_THREADS;
    int per_cu = 0;
    if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cl) ==
            hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kf, nthr, cl) == hipSuccess &&
        per_cu >= 1 && (long)wgs <= (long)per_cu * wn_device_cus()) {
      a.sync = reinterpret_cast<unsigned*>(coop);
      a.ll = reinterpret_cast<fgp_ll_t*>(a.sync + FGP_WORDS);
      if (hipMemsetAsync(coop, 0, (size_t)wn_fastgen_wide_coop_bytes(L, C, S, Q), s) != hipSuccess)
        return WN_ERR_LAUNCH;
      const int rc = f64
          ? fg_launch_cooperative((const void*)fastgen_wide_kernel<true, true>, wgs, 512, cl, s, a)
          : fg_launch_cooperative((const void*)fastgen_wide_kernel<false, true>, wgs, FGW_THREADS, cl, s, a);
      if (rc == WN_OK) return rc;
      // (refused by the runtime -- CUs held elsewhere: the single workgroup below)
    }
  }
  if (f64)
    hipLaunchKernelGGL((fastgen_wide_kernel<true, false>), dim3(1), dim3(512), lds, s, a);
  else
    hipLaunchKernelGGL((fastgen_wide_kernel<false, false>), dim3(1), dim3(FGW_THREADS), lds, s, a);
  return wn_check_launch();
}


// Enqueue ONE generation step (four kernels) on `stream`.  Everything
// step- or call-dependent is read from device memory (cursors, ctl), so the
// call can be captured into a hipGraph once and replayed for every call.
// `pre` must hold the past-tap pre-activations of the step about to run
// (wn_fastgen_pre before the first step of a sequence; afterwards every step
// leaves the next step's behind).
int wn_fastgen_step(const float* params_causal, const float* layer0,
                    long layer_stride, const float* skip_w,
                    const float* skip_bsum, const float* post1_w,
                    const float* post1_b, const float* post2_w,
                    const float* post2_b, const float* gc_bias_fg,
                    const int32_t* dilations_dev, int L, int S, int Q,
                    float* state, int32_t* cursors, int32_t* samples_io,
                    const int32_t* ctl, float* proba_out, int use_biases,
                    const float* cw_img, float* pre, float* z_all, float* h1,
                    float* h2, float* logits, void* stream) {
  if (!cw_img || !logits || !ctl || !pre) return WN_ERR_NULL;
  if (!params_causal || !layer0 || !skip_w || !post1_w || !post2_w ||
      !dilations_dev || !state || !cursors || !samples_io || !z_all || !h1 ||
      !h2)
    return WN_ERR_NULL;
  if (L <= 0 || S <= 0 || Q <= 0) return WN_ERR_BAD_SHAPE;
  if (S > FG_MAXS || Q > FG_MAXQ || L > FG_MAXL) return WN_ERR_UNSUPPORTED;
  FgStep g;
  g.causal = params_causal; g.layer0 = layer0; g.layer_stride = layer_stride;
  g.skip_w = skip_w; g.skip_bsum = skip_bsum; g.post1_w = post1_w;
  g.post1_b = post1_b; g.post2_w = post2_w; g.post2_b = post2_b;
  g.bias_fg = gc_bias_fg; g.dil = dilations_dev; g.L = L; g.S = S; g.Q = Q;
  g.state = state; g.cursors = cursors; g.samples = samples_io; g.ctl = ctl;
  g.proba_out = proba_out;
  g.use_dense_bias = use_biases; g.cw_img = cw_img; g.pre = pre;
  g.z_all = z_all; g.h1 = h1; g.h2 = h2; g.logits = logits;
  hipStream_t s = (hipStream_t)stream;
  const int wgs = (S + FGM_OUTS - 1) / FGM_OUTS;
  hipLaunchKernelGGL(fg_chain_kernel, dim3(1), dim3(FGC_THREADS), 0, s, g);
  hipLaunchKernelGGL(fg_skip_kernel, dim3(wgs + L), dim3(256), 0, s, g);
  hipLaunchKernelGGL(fg_post1_kernel, dim3(wgs), dim3(256), 0, s, g);
  hipLaunchKernelGGL(fg_logits_kernel, dim3((Q + FGM_OUTS - 1) / FGM_OUTS), dim3(256), 0, s, g);
  return wn_check_launch();
}

// Draw from the logits the last wn_fastgen_step left pending (every step's
// draw otherwise happens at the start of the NEXT step's chain kernel).  Call
// once after a sequence of steps; a no-op when nothing is pending.
int wn_fastgen_finish(int Q, int32_t* cursors, int32_t* samples_io,
                      const int32_t* ctl, float* proba_out, const float* logits,
                      void* stream) {
  if (!cursors || !samples_io || !ctl || !logits) return WN_ERR_NULL;
  if (Q <= 0) return WN_ERR_BAD_SHAPE;
  if (Q > FG_MAXQ) return WN_ERR_UNSUPPORTED;
  FgStep g = {};
  g.Q = Q; g.cursors = cursors; g.samples = samples_io; g.ctl = ctl;
  g.proba_out = proba_out; g.logits = const_cast<float*>(logits);
  hipLaunchKernelGGL(fg_draw_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g);
  return wn_check_launch();
}

// ONE persistent launch for n_steps samples (see fg_persist_kernel): same
// inputs, state and results as n_steps x wn_fastgen_step + wn_fastgen_finish
// (cursors[2] must be 0 on entry: no draw pending).  `sync`: 16 uint32 (zeroed
// by this call), `xhand`: 8 x 32 floats of scratch.  After the launch
// sync[12] != 0 means a bounded wait (2 s) expired: the samples are invalid.
// Needs every workgroup resident at once: nseg + 2 * ceil(S / 16) +
// ceil(Q / 16) + 1 workgroups against the occupancy the runtime reports for the
// launch configuration x CUs, else WN_ERR_UNSUPPORTED (use wn_fastgen_step).
// 8-byte hand-over words of wn_fastgen_persist (z, h1, h2, logits, x, code)
int wn_fastgen_persist_role(int block, int total, int nseg) {
  if (block < 0 || total <= 0 || block >= total || nseg < 1 || nseg + 1 > total) return WN_ERR_BAD_SHAPE;
  return fgp_role_of_block(block, total, nseg);
}

long wn_fastgen_persist_ll_words(int L, int S, int Q) {
  if (L <= 0 || S <= 0 || Q <= 0) return 0;
  return (long)L * 32 + 2L * S + Q + FGP_MAXSEG * 32 + 32;
}

int wn_fastgen_persist_workgroups(int L, int S, int Q) {
  if (L <= 0 || S <= 0 || Q <= 0) return 0;
  int nseg = (L + 9) / 10;
  if (nseg > FGP_MAXSEG) nseg = FGP_MAXSEG;
  return nseg + 2 * ((S + 15) / 16) + (Q + 15) / 16 + 1;
}

int wn_fastgen_persist(const float* params_causal, const float* layer0,
                       long layer_stride, const float* skip_w,
                       const float* skip_bsum, const float* post1_w,
                       const float* post1_b, const float* post2_w,
                       const float* post2_b, const float* gc_bias_fg,
                       const int32_t* dilations_dev, int L, int S, int Q,
                       float* state, int32_t* cursors, int32_t* samples_io,
                       const int32_t* ctl, float* proba_out, int use_biases,
                       const float* cw_img, float* pre, float* z_all, float* h1,
                       float* h2, float* logits, unsigned* sync,
                       unsigned long long* ll, int n_steps, void* stream) {
  if (!cw_img || !logits || !ctl || !pre || !sync || !ll) return WN_ERR_NULL;
  if (!params_causal || !layer0 || !skip_w || !post1_w || !post2_w ||
      !dilations_dev || !state || !cursors || !samples_io || !z_all || !h1 ||
      !h2)
    return WN_ERR_NULL;
  if (L <= 0 || S <= 0 || Q <= 0 || n_steps <= 0) return WN_ERR_BAD_SHAPE;
  if (S > FG_MAXS || Q > FG_MAXQ || L > FG_MAXL) return WN_ERR_UNSUPPORTED;
  FgPersist a;
  FgStep& g = a.g;
  g.causal = params_causal; g.layer0 = layer0; g.layer_stride = layer_stride;
  g.skip_w = skip_w; g.skip_bsum = skip_bsum; g.post1_w = post1_w;
  g.post1_b = post1_b; g.post2_w = post2_w; g.post2_b = post2_b;
  g.bias_fg = gc_bias_fg; g.dil = dilations_dev; g.L = L; g.S = S; g.Q = Q;
  g.state = state; g.cursors = cursors; g.samples = samples_io; g.ctl = ctl;
  g.proba_out = proba_out;
  g.use_dense_bias = use_biases; g.cw_img = cw_img; g.pre = pre;
  g.z_all = z_all; g.h1 = h1; g.h2 = h2; g.logits = logits;
  int nseg = (L + 9) / 10;
  if (nseg > FGP_MAXSEG) nseg = FGP_MAXSEG;
  const int per = (L + nseg - 1) / nseg;
  if (per > FGP_SEGL) return WN_ERR_UNSUPPORTED;
  a.sync = sync; a.ll = ll; a.n_steps = n_steps; a.nseg = nseg; a.dbg = nullptr;
#ifdef FGP_STAMPS
  a.dbg = g_fgp_dbg;
#endif
  const int wgs = wn_fastgen_persist_workgroups(L, S, Q);
  // dynamic LDS: the largest role
  size_t chain = (size_t)per * FGP_BLK + 2 * FGP_SEGL + 8;
  size_t skip = (size_t)((L * 32 + 3) & ~3) + (size_t)L * 32 * 16 + 256;
  size_t post = (size_t)((S + 3) & ~3) + (size_t)S * 16 + 256;
  size_t draw = (size_t)((Q + 3) & ~3) + 48 + (size_t)2 * Q * 32;   // + the causal table
  size_t fl = chain;
  if (skip > fl) fl = skip;
  if (post > fl) fl = post;
  if (draw > fl) fl = draw;
  const size_t bytes = fl * 4;
  if (bytes > 160 * 1024 - 512) return WN_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipFuncSetAttribute((const void*)fg_persist_kernel,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return WN_ERR_LAUNCH;
  // every workgroup must be resident at once: ask the runtime how many of THIS
  // launch configuration (registers, dynamic LDS) fit on a CU instead of
  // assuming one -- a non-resident grid is refused here, before anything is
  // written, and the caller takes the step kernels
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fg_persist_kernel,
                                                   FGP_THREADS, bytes) != hipSuccess)
    return WN_ERR_LAUNCH;
  if (per_cu < 1 || (long)wgs > (long)per_cu * wn_device_cus()) return WN_ERR_UNSUPPORTED;
  if (hipMemsetAsync(sync, 0, FGP_WORDS * sizeof(unsigned), s) != hipSuccess ||
      hipMemsetAsync(ll, 0, (size_t)wn_fastgen_persist_ll_words(L, S, Q) * 8, s) != hipSuccess)
    return WN_ERR_LAUNCH;
  return fg_launch_cooperative((const void*)fg_persist_kernel, wgs, FGP_THREADS, bytes, s, a);
}

// Past-tap pre-activations pre[L][64] of the step the queues are at
// (cursors[0]); needed once before the first wn_fastgen_step of a sequence.
int wn_fastgen_pre(const float* layer0, long layer_stride,
                   const float* gc_bias_fg, const int32_t* dilations_dev,
                   int L, const float* state, const int32_t* cursors,
                   float* pre, void* stream) {
  if (!layer0 || !dilations_dev || !state || !cursors || !pre) return WN_ERR_NULL;
  if (L <= 0) return WN_ERR_BAD_SHAPE;
  if (L > FG_MAXL) return WN_ERR_UNSUPPORTED;
  FgStep g = {};
  g.layer0 = layer0; g.layer_stride = layer_stride; g.bias_fg = gc_bias_fg;
  g.dil = dilations_dev; g.L = L; g.state = const_cast<float*>(state);
  g.cursors = const_cast<int32_t*>(cursors); g.pre = pre;
  hipLaunchKernelGGL(fg_pre_kernel, dim3(L), dim3(256), 0, (hipStream_t)stream,
                     g, 0);
  return wn_check_launch();
}

// Build the chain-weight image wn_fastgen_step streams: img [L][3072]
// (Wf[1] | Wg[1] | Wd of every layer, transposed + swizzled).
int wn_fastgen_pack(const float* layer0, long layer_stride, float* img, int L,
                    void* stream) {
  if (!layer0 || !img) return WN_ERR_NULL;
  if (L <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(fg_pack_kernel, dim3(L), dim3(256), 0,
                     (hipStream_t)stream, layer0, layer_stride, img);
  return wn_check_launch();
}

#ifdef FG_STAMPS
int wn_diag_fg_stamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(fg_dbg), sizeof(fg_dbg)) == hipSuccess ? 0 : WN_ERR_LAUNCH;
}
#endif

}  // extern "C"
