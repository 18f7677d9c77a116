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

#define FG_THREADS 320
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
  const int32_t* dil;    // [L] device
  int L, S, Q;
  float* state;          // ring buffers, layer l at state_off(l)
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
};

__global__ __launch_bounds__(FG_THREADS) void fastgen_kernel(FastGen g) {
  __shared__ float zbuf[2][32];
  __shared__ float hbuf[FG_MAXS];     // relu(total) then relu(conv1)
  __shared__ float h2buf[FG_MAXS];
  __shared__ float part[FG_MAXS];     // post2 partial sums
  __shared__ double pd[FG_MAXQ];
  __shared__ int s_code;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = g.S, Q = g.Q, L = g.L;
  const int st = tid - 64;            // skip-thread index 0..255 (waves 1..4)

  int steps_done = g.cursors[0];
  int prev_code = g.cursors[1];
  if (tid == 0) s_code = g.samples[0];
  __syncthreads();

  for (int step = 0; step < g.n_steps; ++step) {
    const int code = s_code;
    const long tpos = (long)steps_done + step;
    // ---------------- chain + skip, pipelined by one layer ----------------
    float x = 0.f;  // wave 0, lanes < 32: residual stream
    if (wave == 0 && lane < 32) {
      float v = 0.f;
      if (prev_code >= 0 && prev_code < Q) v = g.causal[(long)prev_code * 32 + lane];
      if (code >= 0 && code < Q) v += g.causal[((long)Q + code) * 32 + lane];
      x = v;
    }
    float acc0 = 0.f, acc1 = 0.f;      // skip outputs st and st+256
    long qbase = 0;
    for (int l = 0; l <= L; ++l) {
      if (wave == 0) {
        if (l < L) {
          const int d = g.dil[l];
          const float* blk = g.layer0 + (long)l * g.layer_stride;
          float* ring = g.state + qbase * 32 + (tpos % d) * 32;
          qbase += d;
          float stv = 0.f;
          if (lane < 32) {
            stv = ring[lane];   // dequeue: x_l[t - d]
            if (g.push) ring[lane] = x;  // enqueue: x_l[t]
          }
          // conv: lane n -> filter ch n (n<32) / gate ch n-32
          const float* wcol = blk + (lane < 32 ? 0 : 2048) + (lane & 31);
          float a = g.bias_fg ? g.bias_fg[l * 64 + lane] : 0.f;
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            const float sk = readlane_f(stv, k);
            const float xk = readlane_f(x, k);
            a = fmaf(sk, wcol[k * 32], a);           // W[0]: past tap
            a = fmaf(xk, wcol[1024 + k * 32], a);    // W[1]: current tap
          }
          const float gate = __shfl(a, (lane & 31) + 32);
          const float z = wn_tanh(a) * wn_sigmoid(gate);  // valid on lanes < 32
          if (lane < 32) zbuf[l & 1][lane] = z;
          if (l + 1 < L) {
            float dsum = g.use_dense_bias ? blk[LAYER_OFF_BD + (lane & 31)] : 0.f;
            const float* wd = blk + 4096 + (lane & 31);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
              const float zk = readlane_f(z, k);
              dsum = fmaf(zk, wd[k * 32], dsum);
            }
            if (lane < 32) x += dsum;
          }
        }
      } else if (l >= 1) {
        // skip accumulation for layer l-1 (z written before the last barrier)
        const float* zl = zbuf[(l - 1) & 1];
        const float* ws = g.skip_w + (long)(l - 1) * 32 * S;
        if (st < S) {
#pragma unroll 8
          for (int k = 0; k < 32; ++k) acc0 = fmaf(zl[k], ws[(long)k * S + st], acc0);
        }
        if (st + 256 < S) {
#pragma unroll 8
          for (int k = 0; k < 32; ++k)
            acc1 = fmaf(zl[k], ws[(long)k * S + st + 256], acc1);
        }
      }
      __syncthreads();
    }
    // ---------------- post-processing (model.py:505-514) -------------------
    if (wave >= 1) {
      if (st < S) hbuf[st] = fmaxf(acc0 + (g.skip_bsum ? g.skip_bsum[st] : 0.f), 0.f);
      if (st + 256 < S)
        hbuf[st + 256] = fmaxf(acc1 + (g.skip_bsum ? g.skip_bsum[st + 256] : 0.f), 0.f);
    }
    __syncthreads();
    if (wave >= 1) {
      for (int s = st; s < S; s += 256) {
        float c = g.post1_b ? g.post1_b[s] : 0.f;
        const float* w = g.post1_w + s;
#pragma unroll 8
        for (int k = 0; k < S; ++k) c = fmaf(hbuf[k], w[(long)k * S], c);
        h2buf[s] = fmaxf(c, 0.f);
      }
    }
    __syncthreads();
    // logits: thread (q, part) sums a k-range; parts = 256 / Qp
    {
      int parts = 256 / Q;
      if (parts < 1) parts = 1;
      if (wave >= 1) {
        for (int o = st; o < Q * parts; o += 256) {
          const int q = o % Q, p = o / Q;
          const int k0 = (int)((long)S * p / parts), k1 = (int)((long)S * (p + 1) / parts);
          float c = 0.f;
          const float* w = g.post2_w + q;
          for (int k = k0; k < k1; ++k) c = fmaf(h2buf[k], w[(long)k * Q], c);
          part[o] = c;
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
      for (int q = lane; q < Q; q += 64) se += exp(pd[q] - m);
      for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
      const bool want_p = g.proba_out && (step % g.proba_every == 0);
      float* po = want_p ? g.proba_out + (long)(step / g.proba_every) * Q : nullptr;
      for (int q = lane; q < Q; q += 64) {
        const float p32 = (float)(exp(pd[q] - m) / se);
        if (po) po[q] = p32;
        pd[q] = (double)p32;  // the float32 probabilities generate.py sees
      }
    }
    __syncthreads();
    if (step + 1 >= g.n_given) {
      if (tid == 0) {
        // temperature: exp(log(p)/tau - logsumexp) in float64
        const double tau = (double)g.temperature;
        double mx = -1e300;
        for (int q = 0; q < Q; ++q) {
          const double lp = log(pd[q] > 0.0 ? pd[q] : 1e-300) / tau;
          pd[q] = lp;
          mx = fmax(mx, lp);
        }
        double se = 0.0;
        for (int q = 0; q < Q; ++q) se += exp(pd[q] - mx);
        const uint64_t r = splitmix64(g.seed ^ splitmix64((uint64_t)tpos));
        const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0) * se;
        double c = 0.0;
        int pick = Q - 1;
        for (int q = 0; q < Q; ++q) {
          c += exp(pd[q] - mx);
          if (u < c) { pick = q; break; }
        }
        g.samples[step + 1] = pick;
        s_code = pick;
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
  if (S > FG_MAXS || Q > FG_MAXQ) return WN_ERR_UNSUPPORTED;
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

}  // extern "C"
