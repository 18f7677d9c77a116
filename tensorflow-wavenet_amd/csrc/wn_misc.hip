// Streaming (HBM-bound) kernels around the MFMA core: mu-law companding, the
// one-hot "conv" of the causal layer as a 2-row gather, fused softmax
// cross-entropy, optimizers with TensorFlow update rules, global-conditioning
// helpers, and the thin exported ops (causal_conv, time_to_batch, ...).
#include "wn_common.h"
#include <stdlib.h>
#include <cmath>

// ---------------------------------------------------------------------------
// mu-law  (wavenet/ops.py:65-85)
//
// Bit-exactness: ops.py:65-73 is a float32 chain whose only non-IEEE-exact
// step is log().  The chain is DEFINED here with the correctly rounded
// float32 log (double log rounded once).  encode() is a monotone step
// function of x on [-1,1], so the device encodes by binary search over the
// Q-1 float32 decision thresholds of that chain (built once on the host by
// wn_mu_law_thresholds_host) -- exact by construction, independent of the
// device's logf.  |x| > 1 (outside the table) evaluates the chain itself.
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)
static inline int mu_chain_host(float x, int Q) {
  const float mu = (float)(Q - 1);
  const float ax = std::fabs(x);
  const float arg = 1.0f + mu * ax;
  const float num = (float)std::log((double)arg);
  const float den = (float)std::log((double)(1.0f + mu));
  const float mag = num / den;
  const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  float v = sgn * mag;
  v = v + 1.0f;
  v = v / 2.0f;
  v = v * mu;
  v = v + 0.5f;
  return (int)v;
}

static inline uint32_t f2key(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

__device__ __forceinline__ int mu_chain_dev(float x, int Q) {
  // same chain on the device for |x| > 1; double log keeps it within the
  // definition above (OCML double log, rounded once to float).
  const float mu = (float)(Q - 1);
  const float arg = __fadd_rn(1.0f, __fmul_rn(mu, fabsf(x)));
  const float num = (float)log((double)arg);
  const float den = (float)log((double)__fadd_rn(1.0f, mu));
  const float mag = __fdiv_rn(num, den);
  const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  float v = __fmul_rn(sgn, mag);
  v = __fadd_rn(v, 1.0f);
  v = __fdiv_rn(v, 2.0f);
  v = __fmul_rn(v, mu);
  v = __fadd_rn(v, 0.5f);
  return (int)v;
}

__global__ void mu_law_encode_kernel(const float* __restrict__ audio,
                                     int32_t* __restrict__ codes, long n,
                                     const float* __restrict__ thr, int Q) {
  extern __shared__ float sthr[];
  for (int i = threadIdx.x; i < Q - 1; i += blockDim.x) sthr[i] = thr[i];
  __syncthreads();
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (long)gridDim.x * blockDim.x) {
    const float x = audio[idx];
    int code;
    if (x >= -1.f && x <= 1.f) {
      int lo = 0, hi = Q - 1;  // count of thresholds <= x  (upper bound)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sthr[mid] <= x) lo = mid + 1; else hi = mid;
      }
      code = lo;
    } else {
      code = mu_chain_dev(x, Q);
    }
    codes[idx] = code;
  }
}

__device__ __forceinline__ float mu_decode_dev(int code, int Q) {
  const float mu = (float)(Q - 1);
  const float s = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn((float)code, mu)), 1.0f);
  const float p = (float)pow((double)__fadd_rn(1.0f, mu), (double)fabsf(s));
  const float mag = __fmul_rn(__fdiv_rn(1.0f, mu), __fsub_rn(p, 1.0f));
  const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
  return __fmul_rn(sgn, mag);
}

__global__ void mu_law_decode_kernel(const int32_t* __restrict__ codes,
                                     float* __restrict__ audio, long n,
                                     const float* __restrict__ lut, int Q) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (long)gridDim.x * blockDim.x) {
    const int c = codes[idx];
    audio[idx] = (c >= 0 && c < Q) ? lut[c] : mu_decode_dev(c, Q);
  }
}

// ---------------------------------------------------------------------------
// causal layer on one-hot input == gather   (model.py:227-234, 518-531)
//   x0[t] = sum_k Wc[k][q[t - s_k]],  s_k = (K-1-k) + (K-1)/2
//   (K = filter_width; K = 2: Wc[0][q[t-1]] + Wc[1][q[t]]; terms with t < s_k
//   vanish; an out-of-range code is an all-zero one-hot row)
// Wc: [K][Q][32] (R padded to 32).  8 threads x 16 B per row.
// ---------------------------------------------------------------------------
__global__ void causal_gather_kernel(const int32_t* __restrict__ q,
                                     const float* __restrict__ Wc,
                                     float* __restrict__ x0, long rows, int T,
                                     int Q, int K, int ldw) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx >> 3;
  const int c4 = (idx & 7) * 4;
  if (row >= rows) return;
  const int t = (int)(row % T);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const int extra = (K - 1) / 2;       // TF 'SAME' centring for K > 2
  for (int k = 0; k < K; ++k) {
    const int s = (K - 1 - k) + extra;
    if (t < s) continue;
    const int code = q[row - s];
    if (code >= 0 && code < Q)
      v += *reinterpret_cast<const f32x4*>(Wc + ((long)k * Q + code) * ldw + c4);
  }
  *reinterpret_cast<f32x4*>(x0 + row * 32 + c4) = v;
}

// ---------------------------------------------------------------------------
// causal layer on scalar input (scalar_input=True, model.py:143-153, 646-648):
//   x0[t][c] = sum_k audio[t - s_k] * W[k][0][c],  s_k = (K0-1-k) + (K0-1)/2
// (dilation 1; the extra (K0-1)/2 delay is TF's 'SAME' centring inside
// causal_conv for K0 > 2, ops.py:46-62).  W: [K0][32].
// ---------------------------------------------------------------------------
__global__ void scalar_causal_fwd_kernel(const float* __restrict__ audio,
                                         const float* __restrict__ W,
                                         float* __restrict__ x0, long rows,
                                         int T, int K0, int ldw) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx >> 3;
  const int c4 = (idx & 7) * 4;
  if (row >= rows) return;
  const int t = (int)(row % T);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const int extra = (K0 - 1) / 2;
  for (int k = 0; k < K0; ++k) {
    const int s = (K0 - 1 - k) + extra;
    if (t - s < 0) continue;
    const float a = audio[row - s];
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)k * ldw + c4);
    v += a * w;
  }
  *reinterpret_cast<f32x4*>(x0 + row * 32 + c4) = v;
}

// dW[k][c] = sum_rows audio[row - s_k] * dx0[row][c] into slabs [splits][K0*32]
__global__ __launch_bounds__(1024) void scalar_causal_wgrad_kernel(
    const float* __restrict__ audio, const float* __restrict__ dx0,
    float* __restrict__ slabs, long rows, long rows_per_split, int T, int K0) {
  const int c = threadIdx.x & 31;
  const long r0 = (long)blockIdx.x * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > rows) r1 = rows;
  // thread (k, c): taps k, k + 32, ... (initial_filter_width > 32: one pass
  // over the rows per 32 taps)
  for (int k = threadIdx.x >> 5; k < K0; k += 32) {
    const int s = (K0 - 1 - k) + (K0 - 1) / 2;
    float acc = 0.f;
    int t = (int)(r0 % T);
    for (long r = r0; r < r1; ++r) {
      if (t >= s) acc = fmaf(audio[r - s], dx0[r * 32 + c], acc);
      if (++t == T) t = 0;
    }
    slabs[(long)blockIdx.x * K0 * 32 + k * 32 + c] = acc;
  }
}

// ---------------------------------------------------------------------------
// fused softmax cross-entropy, forward + backward   (model.py:654-666)
// One wave per row.  target of row (b,t) = q[b][t+1]; the last row of every
// clip has the all-zero label row the reference pads in (model.py:659): its
// loss term is 0 but it stays in the mean's denominator, and with
// tf_quirk != 0 it back-propagates softmax/(B*T) like TF's fused kernel
// (backprop = softmax - labels).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xent_kernel(
    const float* __restrict__ logits, long ld, const int32_t* __restrict__ q,
    float* __restrict__ dlogits, float* __restrict__ loss_partials, long rows,
    int T, int Q, float inv_n, int tf_quirk) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lsum = 0.f;
  const long nwaves = (long)gridDim.x * 4;
  if (Q == 256) {
    // one 16-byte load per lane holds the whole row: a single pass, each exp
    // evaluated once, the next row's load in flight under the reductions
    long row = (long)blockIdx.x * 4 + wave;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) v = *reinterpret_cast<const f32x4*>(logits + row * ld + lane * 4);
    for (; row < rows; row += nwaves) {
      const long nrow = row + nwaves;
      f32x4 vn = {0.f, 0.f, 0.f, 0.f};
      if (nrow < rows)
        vn = *reinterpret_cast<const f32x4*>(logits + nrow * ld + lane * 4);
      const int t = (int)(row % T);
      const int label = (t + 1 < T) ? q[row + 1] : -1;
      const bool has_label = label >= 0 && label < Q;
      float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = expf(v[k] - m);
      float se = (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
      const float lse = m + logf(se);
      if (has_label) {
        // the label's logit lives in lane label >> 2
        const float mine = (label >> 2) == lane ? v[label & 3] : 0.f;
        float ll = mine;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ll += __shfl_xor(ll, o);
        if (lane == 0) lsum += lse - ll;
      }
      if (dlogits) {
        const float inv_se = 1.f / se;
        const bool back = has_label || tf_quirk;
        f32x4 g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float p = back ? e[k] * inv_se : 0.f;
          if (has_label && lane * 4 + k == label) p -= 1.f;
          g[k] = p * inv_n;
        }
        *reinterpret_cast<f32x4*>(dlogits + row * ld + lane * 4) = g;
      }
      v = vn;
    }
  } else
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += nwaves) {
    const float* lp = logits + row * ld;
    const int t = (int)(row % T);
    const int label = (t + 1 < T) ? q[row + 1] : -1;
    const bool has_label = label >= 0 && label < Q;
    float m = -INFINITY;
    for (int c = lane * 4; c < Q; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(lp + c);
      m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float se = 0.f;
    for (int c = lane * 4; c < Q; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(lp + c);
      se += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
    const float lse = m + logf(se);
    if (has_label && lane == 0) lsum += lse - lp[label];
    const float inv_se = 1.f / se;
    const bool back = has_label || tf_quirk;
    if (dlogits) {
      float* dp = dlogits + row * ld;
      for (int c = lane * 4; c < Q; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lp + c);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float p = back ? expf(v[e] - m) * inv_se : 0.f;
          if (has_label && c + e == label) p -= 1.f;
          g[e] = p * inv_n;
        }
        *reinterpret_cast<f32x4*>(dp + c) = g;
      }
    }
  }
  if (lane == 0) wsum[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0)
    loss_partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// softmax of ONE row in float64, cast to float32 (model.py:584-585, 620-621)
__global__ void softmax64_row_kernel(const float* __restrict__ logits, int Q,
                                     float* __restrict__ proba) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double m = -INFINITY;
  for (int c = tid; c < Q; c += 256) m = fmax(m, (double)logits[c]);
  red[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  double se = 0.0;
  for (int c = tid; c < Q; c += 256) se += exp((double)logits[c] - m);
  red[tid] = se;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  se = red[0];
  for (int c = tid; c < Q; c += 256)
    proba[c] = (float)(exp((double)logits[c] - m) / se);
}

// ---------------------------------------------------------------------------
// optimizers, TensorFlow-0.10 update rules (wavenet/ops.py:6-24)
//   g' = g * grad_scale + l2 * p * (l2_mask ? l2_mask[i] : 1)
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v,
                            long n, float lr_t, float b1, float b2, float eps,
                            float grad_scale, float l2,
                            const float* __restrict__ l2_mask) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float w = p[i];
    float gg = g[i] * grad_scale;
    if (l2 != 0.f) gg += l2 * w * (l2_mask ? l2_mask[i] : 1.f);
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = w - lr_t * mm / (sqrtf(vv) + eps);
  }
}

__global__ void momentum_kernel(float* __restrict__ p,
                                const float* __restrict__ g,
                                float* __restrict__ acc, long n, float lr,
                                float mom, float grad_scale, float l2,
                                const float* __restrict__ l2_mask) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float w = p[i];
    float gg = g[i] * grad_scale;
    if (l2 != 0.f) gg += l2 * w * (l2_mask ? l2_mask[i] : 1.f);
    const float a = mom * acc[i] + gg;
    acc[i] = a;
    p[i] = w - lr * a;
  }
}

__global__ void rmsprop_kernel(float* __restrict__ p,
                               const float* __restrict__ g,
                               float* __restrict__ ms, float* __restrict__ mo,
                               long n, float lr, float decay, float mom,
                               float eps, float grad_scale, float l2,
                               const float* __restrict__ l2_mask) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float w = p[i];
    float gg = g[i] * grad_scale;
    if (l2 != 0.f) gg += l2 * w * (l2_mask ? l2_mask[i] : 1.f);
    const float s = decay * ms[i] + (1.f - decay) * gg * gg;
    const float mm = mom * mo[i] + lr * gg / sqrtf(s + eps);
    ms[i] = s;
    mo[i] = mm;
    p[i] = w - mm;
  }
}

// sum of squares / 2 (tf.nn.l2_loss) partials, with optional mask
__global__ void l2_partials_kernel(const float* __restrict__ p, long n,
                                   const float* __restrict__ mask,
                                   float* __restrict__ partials) {
  __shared__ float red[256];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float w = p[i];
    s += w * w * (mask ? mask[i] : 1.f);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (unsigned k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = 0.5f * red[0];
}

// ---------------------------------------------------------------------------
// global conditioning   (model.py:272-284, 533-562)
// The GC 1x1 conv of a [B,1,G] embedding broadcast over T is a per-(clip,
// layer) bias:  bias_fg[l][b][0:32] = bf_l + emb[id_b] * Wgcf_l, [32:64] gate.
// ---------------------------------------------------------------------------
__global__ void gc_bias_kernel(const float* __restrict__ layer0, long layer_stride,
                               long off_bias, long off_gc, int G,
                               const float* __restrict__ emb, int card,
                               const int32_t* __restrict__ ids,
                               float* __restrict__ out, int B, int ch) {
  const int l = blockIdx.x, b = blockIdx.y;
  const float* blk = layer0 + (long)l * layer_stride;
  for (int c = threadIdx.x; c < 2 * ch; c += blockDim.x) {
    float v = blk[off_bias + c];  // bf (0..ch-1) then bg (ch..2ch-1)
    if (emb) {
      const int id = ids[b];
      if (id >= 0 && id < card) {
        // Wgcf [G][ch] at off_gc, Wgcg [G][ch] at off_gc + G*ch
        const float* w = blk + off_gc + (c >= ch ? (long)G * ch : 0) + (c % ch);
        const float* e = emb + (long)id * G;
        for (int g = 0; g < G; ++g) v += e[g] * w[(long)g * ch];
      }
    }
    out[((long)l * B + b) * 2 * ch + c] = v;
  }
}

__global__ __launch_bounds__(256) void colsum_clip_kernel(
    const float* __restrict__ P0, const float* __restrict__ P1, int T,
    int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[8][64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c = threadIdx.x & 31, w = threadIdx.x >> 5;
  const int t0 = chunk * rows_per_chunk;
  const int t1 = min(T, t0 + rows_per_chunk);
  const float* p0 = P0 + (long)b * T * 32;
  const float* p1 = P1 + (long)b * T * 32;
  float s0 = 0.f, s1 = 0.f;
  for (int t = t0 + w; t < t1; t += 8) {
    s0 += p0[(long)t * 32 + c];
    s1 += p1[(long)t * 32 + c];
  }
  red[w][c] = s0;
  red[w][32 + c] = s1;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][threadIdx.x];
    part[((long)b * gridDim.x + chunk) * 64 + threadIdx.x] = v;
  }
}

// dWgc_l = emb[ids]^T dsum_l ; demb[id_b] += sum_l dsum_l[b] Wgc_l^T
// dsum: [L][B][64] (f | g).  grid.x = L (+1 block for the embedding rows).
__global__ void gc_grad_kernel(const float* __restrict__ layer0, long layer_stride,
                               long off_gc, int G, const float* __restrict__ emb,
                               int card, const int32_t* __restrict__ ids,
                               const float* __restrict__ dsum, int L, int B,
                               float* __restrict__ glayer0,
                               float* __restrict__ part, int ch) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < L) {
    const int l = blockIdx.x;
    float* gw = glayer0 + (long)l * layer_stride + off_gc;
    for (int e = tid; e < 2 * G * ch; e += blockDim.x) {
      const int which = e / (G * ch), g = (e / ch) % G, c = e % ch;
      float s = 0.f;
      for (int b = 0; b < B; ++b) {
        const int id = ids[b];
        if (id < 0 || id >= card) continue;
        s += emb[(long)id * G + g] * dsum[((long)l * B + b) * 2 * ch + which * ch + c];
      }
      gw[e] = s;
    }
  } else {
    // embedding rows, stage 1 (one workgroup per layer): the layer's share
    // part[l][b][g] = dsum_l[b] . Wgc_l[g] of every clip's embedding gradient
    const int l = blockIdx.x - L;
    const float* w = layer0 + (long)l * layer_stride + off_gc;
    for (int e = tid; e < B * G; e += blockDim.x) {
      const int b = e / G, gi = e - b * G;
      const float* ds = dsum + ((long)l * B + b) * 2 * ch;
      float s = 0.f;
      for (int c = 0; c < ch; ++c)
        s += ds[c] * w[(long)gi * ch + c] + ds[ch + c] * w[(long)(G + gi) * ch + c];
      part[((long)l * B + b) * G + gi] = s;
    }
  }
}

// embedding rows, stage 2 (one workgroup): tmp[b][g] = sum_l part[l][b][g] in
// layer order, then the rows are added clip by clip in a fixed order
// (deterministic when several clips share an id).
__global__ void gc_grad_emb_kernel(const float* __restrict__ part, int L, int B,
                                   int G, int card,
                                   const int32_t* __restrict__ ids,
                                   float* __restrict__ gemb) {
  extern __shared__ float tmp[];      // [B][G]
  const int tid = threadIdx.x;
  for (int e = tid; e < card * G; e += blockDim.x) gemb[e] = 0.f;
  for (int e = tid; e < B * G; e += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += part[(long)l * B * G + e];
    tmp[e] = s;
  }
  __syncthreads();
  for (int gi = tid; gi < G; gi += blockDim.x) {
    for (int b = 0; b < B; ++b) {
      const int id = ids[b];
      if (id < 0 || id >= card) continue;
      gemb[(long)id * G + gi] += tmp[b * G + gi];
    }
  }
}

__global__ void causal_conv_kernel(const float* __restrict__ x,
                                   const float* __restrict__ w,
                                   float* __restrict__ y, int B, int T, int Cin,
                                   int Cout, int K, int d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * T * Cout;
  if (idx >= total) return;
  const int co = (int)(idx % Cout);
  const long bt = idx / Cout;
  const int t = (int)(bt % T);
  const long b = bt / T;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    // ops.py:46-62 closed form incl. TF 'SAME' centring for K > 2
    const long shift = (long)(K - 1 - k + (K - 1) / 2) * d;
    if (t - shift < 0) continue;
    const float* xr = x + ((b * T) + (t - shift)) * Cin;
    const float* wk = w + (long)k * Cin * Cout + co;
    for (int ci = 0; ci < Cin; ++ci) acc += xr[ci] * wk[(long)ci * Cout];
  }
  y[idx] = acc;
}

// time_to_batch (ops.py:27-34): out[(p*B + b), u, c] = in[b, u*d + p, c] (0 pad)
__global__ void time_to_batch_kernel(const float* __restrict__ in,
                                     float* __restrict__ out, int B, int T,
                                     int C, int d, int U) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * d * U * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  long r = idx / C;
  const int u = (int)(r % U);
  r /= U;  // r = p*B + b
  const int b = (int)(r % B), p = (int)(r / B);
  const long t = (long)u * d + p;
  out[idx] = t < T ? in[((long)b * T + t) * C + c] : 0.f;
}

// batch_to_time (ops.py:37-43): out[b, u*d + p, c] = in[(p*B + b), u, c]
__global__ void batch_to_time_kernel(const float* __restrict__ in,
                                     float* __restrict__ out, int B, int U,
                                     int C, int d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * d * U * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  long r = idx / C;
  const int tt = (int)(r % ((long)U * d));
  const int b = (int)(r / ((long)U * d));
  const int u = tt / d, p = tt % d;
  out[idx] = in[(((long)p * B + b) * U + u) * C + c];
}

// y += a * x * (mask ? mask : 1)   (gradient of the L2 term, model.py:674-680)
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x,
                            float a, const float* __restrict__ mask, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    y[i] += a * x[i] * (mask ? mask[i] : 1.f);
}

__global__ void fill_kernel(float* __restrict__ p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    p[i] = v;
}

// sum over layers of the skip biases (epilogue bias of the skip GEMM)
__global__ void sum_rows_kernel(const float* __restrict__ in, int rows, int n,
                                float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  // loads of a batch of rows are issued together (the sum itself stays in row
  // order): 50 dependent round trips were 20 us for 100 KB
  float s = 0.f;
  int r = 0;
  for (; r + 16 <= rows; r += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = in[(long)(r + u) * n + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; r < rows; ++r) s += in[(long)r * n + c];
  out[c] = s;
}

static inline int grid1d(long n, int block, int cap = 2048) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

int wn_reduce_slabs(const float* slabs, int num_slabs, long slab_stride,
                    int batch, long in_batch_stride, long offset, long n,
                    float* dst, long out_batch_stride, int replicate,
                    long rep_stride, void* stream);

int wn_version(void) { return 100; }

const char* wn_error_string(int code) {
  switch (code) {
    case WN_OK: return "ok";
    case WN_ERR_BAD_SHAPE: return "bad shape";
    case WN_ERR_UNSUPPORTED: return "unsupported configuration";
    case WN_ERR_MISALIGNED: return "pointer or leading dimension not 16-byte aligned";
    case WN_ERR_LAUNCH: return "kernel launch failed";
    case WN_ERR_NULL: return "null pointer";
    default: return "unknown error";
  }
}

int wn_mu_law_thresholds_host(int Q, float* thr_out) {
  if (!thr_out) return WN_ERR_NULL;
  if (Q < 2) return WN_ERR_BAD_SHAPE;
  const uint32_t lo0 = f2key(-1.0f), hi0 = f2key(1.0f);
  for (int k = 1; k < Q; ++k) {
    uint32_t lo = lo0, hi = hi0;
    if (mu_chain_host(key2f(lo), Q) >= k) { thr_out[k - 1] = key2f(lo); continue; }
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (mu_chain_host(key2f(mid), Q) >= k) hi = mid; else lo = mid;
    }
    thr_out[k - 1] = key2f(hi);
  }
  return WN_OK;
}

int wn_mu_law_decode_table_host(int Q, float* lut_out) {
  if (!lut_out) return WN_ERR_NULL;
  if (Q < 2) return WN_ERR_BAD_SHAPE;
  const float mu = (float)(Q - 1);
  for (int c = 0; c < Q; ++c) {
    float s = (float)c / mu;
    s = 2.0f * s;
    s = s - 1.0f;
    const float p = (float)std::pow((double)(1.0f + mu), (double)std::fabs(s));
    float mag = 1.0f / mu;
    const float pm1 = p - 1.0f;
    mag = mag * pm1;
    const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
    lut_out[c] = sgn * mag;
  }
  return WN_OK;
}

int wn_mu_law_encode(const float* audio, int32_t* codes, long n,
                     const float* thr_dev, int Q, void* stream) {
  if (!audio || !codes || !thr_dev) return WN_ERR_NULL;
  if (n <= 0 || Q < 2 || Q > 8192) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(mu_law_encode_kernel, dim3(grid1d(n, 256)), dim3(256),
                     (Q - 1) * sizeof(float), (hipStream_t)stream, audio, codes,
                     n, thr_dev, Q);
  return wn_check_launch();
}

int wn_mu_law_decode(const int32_t* codes, float* audio, long n,
                     const float* lut_dev, int Q, void* stream) {
  if (!audio || !codes || !lut_dev) return WN_ERR_NULL;
  if (n <= 0 || Q < 2) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(mu_law_decode_kernel, dim3(grid1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, codes, audio, n, lut_dev, Q);
  return wn_check_launch();
}

int wn_causal_gather(const int32_t* q, const float* Wc, float* x0, int B,
                     int T, int Q, int K, int ldw, void* stream) {
  if (!q || !Wc || !x0) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || Q <= 0 || K <= 0 || ldw < 32 || (ldw & 3))
    return WN_ERR_BAD_SHAPE;
  if (!wn_aligned16(Wc) || !wn_aligned16(x0)) return WN_ERR_MISALIGNED;
  const long rows = (long)B * T;
  const long threads = rows * 8;
  hipLaunchKernelGGL(causal_gather_kernel, dim3((unsigned)((threads + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, q, Wc, x0, rows, T, Q,
                     K, ldw);
  return wn_check_launch();
}

int wn_scalar_causal_fwd(const float* audio, const float* W, int ldw, float* x0,
                         int B, int T, int K0, void* stream) {
  if (!audio || !W || !x0) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || K0 <= 0 || ldw < 32 || (ldw & 3)) return WN_ERR_BAD_SHAPE;
  if (!wn_aligned16(W) || !wn_aligned16(x0)) return WN_ERR_MISALIGNED;
  const long rows = (long)B * T, threads = rows * 8;
  hipLaunchKernelGGL(scalar_causal_fwd_kernel,
                     dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, audio, W, x0, rows, T, K0, ldw);
  return wn_check_launch();
}

int wn_scalar_causal_wgrad(const float* audio, const float* dx0, float* slabs,
                           int splits, int B, int T, int K0, void* stream) {
  if (!audio || !dx0 || !slabs) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || K0 <= 0 || splits <= 0) return WN_ERR_BAD_SHAPE;
  const long rows = (long)B * T;
  const long rps = (rows + splits - 1) / splits;
  hipLaunchKernelGGL(scalar_causal_wgrad_kernel, dim3(splits), dim3(1024), 0,
                     (hipStream_t)stream, audio, dx0, slabs, rows, rps, T, K0);
  return wn_check_launch();
}

// ---------------------------------------------------------------------------
// Causal-layer weight gradient for one-hot input, filter width 2
// (model.py:227-234 under autodiff): dWc[tap][v][:] = sum over rows whose code
// q[t - (1 - tap)] == v of dx0[t][:].  It is a segmented sum, not a GEMM: 16 MB
// of dx0 are read once.  Every wave owns a private [2 taps][Q][32] table in
// LDS and walks a contiguous range of rows in program order, which keeps the
// result independent of scheduling; the per-wave tables go out as slabs for
// wn_reduce_slabs.  Replaces two one-hot MFMA contractions (2 x 110 us at
// B*T = 128000).  Round 3: 90 -> 34 us (one wave per SIMD: what bounds it is
// the number of instructions on the row chain, not HBM and not the LDS round
// trip: four rows per LDS round trip took 45 -> 43 us, a deeper load ring
// nothing, the per-row address arithmetic out of the vector unit 42 -> 34).
// ---------------------------------------------------------------------------
#define CWG_QMAX 256
#define CWG_NB 32     // rows per batch
__global__ __launch_bounds__(128) void causal_wgrad_kernel(
    const int32_t* __restrict__ q, const float* __restrict__ dx0,
    float* __restrict__ slabs, long rows, long rows_per_slab, int T, int Q) {
  // per wave: [2 taps][Q + 1][32]; row Q of a tap table is a dump for rows
  // without a code (the first sample of a clip for tap 0, rows past the range)
  __shared__ __attribute__((aligned(16))) float tab[2 * 2 * (CWG_QMAX + 1) * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;        // channel, tap
  const int trows = Q + 1;
  float* wtab = tab + (size_t)wave * 2 * trows * 32;
  {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < 2 * trows * 8; i += 64) reinterpret_cast<f32x4*>(wtab)[i] = z4;
  }
  __builtin_amdgcn_wave_barrier();
  const long slab = (long)blockIdx.x * 2 + wave;
  const long r0 = slab * rows_per_slab;
  long r1 = r0 + rows_per_slab;
  if (r1 > rows) r1 = rows;
  // Half-wave h accumulates tap h (tap 0 pairs dx0[t] with the PREVIOUS code),
  // lane c owns channel c: one row per step, 32 consecutive floats of one
  // table row, plain read-add-write in program order (a wave's LDS operations
  // execute in order, so rows that share a code simply chain; LDS float
  // atomics measured ~400 cycles per instruction here, hence none).  Nothing
  // on that chain but the LDS round trip: a batch's 32 rows of dx0 and its
  // codes are requested one batch ahead (one coalesced code load per lane:
  // lane (j, h) holds the table offset of row j for tap h), the 32 table
  // addresses are spread to the lanes before the chain starts, rows without a
  // code go to the dump row instead of around a branch.
  const int shift = 1 - h;
  const int lane_off = (h * trows * 32 + c) * 4;            // bytes inside wtab
  auto fetch_codes = [&](long rb, int tb) -> int {
    const long row = rb + c;                                // this lane's row
    int code = Q;
    if (row < r1) {
      int t = tb + c;                                       // tb: clip position of row rb
      if (t >= T) t %= T;
      if (t - shift >= 0) {
        const int v = q[row - shift];
        if (v >= 0 && v < Q) code = v;
      }
    }
    return code * 128;                                      // byte offset of the table row
  };
  // dx0 rows through a buffer resource based at the slab's first row: the
  // row offset is scalar, the lane's part (c * 4) loop-invariant -- no vector
  // address arithmetic per row (one wave per SIMD: the kernel is bound by the
  // number of instructions it issues).  Rows past the slab re-read its last
  // row (their code is the dump row): the scalar offset is clamped on the
  // scalar unit, the hardware's range check covers the VGPR offset only.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(dx0 + (r0 < r1 ? r0 : 0) * 32), 0,
      r0 < r1 ? (int)((r1 - r0) * 128) : 0, 0x00020000);
  const int so_last = r0 < r1 ? (int)((r1 - r0 - 1) * 128) : 0;
  auto fetch_rows = [&](long rb, float (&vv)[CWG_NB]) {
    const int so = __builtin_amdgcn_readfirstlane((int)((rb - r0) * 128));
#pragma unroll
    for (int u = 0; u < CWG_NB; ++u)
      vv[u] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rs, c * 4, min(so + u * 128, so_last), 0));
  };
  // a ring of four batches: three are in flight while one is accumulated (a
  // batch of 32 rows is worked off in ~0.7 us, a load takes ~2 us to arrive)
  float ring[4][CWG_NB];
  int codes[4] = {0, 0, 0, 0};
  long fb = r0;                                  // next batch to request
  int tb = r0 < r1 ? (int)(r0 % T) : 0;          // its clip position
  auto request = [&](int slot) {
    if (fb < r1) {
      codes[slot] = fetch_codes(fb, tb);
      fetch_rows(fb, ring[slot]);
    }
    fb += CWG_NB;
    tb += CWG_NB;
    if (tb >= T) tb %= T;
  };
  request(0);
  request(1);
  request(2);
  for (long rb = r0; rb < r1; rb += 4 * CWG_NB) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (rb + sl * CWG_NB >= r1) break;
      request((sl + 3) & 3);
      const float (&v)[CWG_NB] = ring[sl];
      // table address of row u for this lane's tap: lane (u, h)'s offset
      int addr[CWG_NB];
#pragma unroll
      for (int u = 0; u < CWG_NB; ++u)
        addr[u] = __builtin_amdgcn_ds_bpermute(((lane & 32) | u) * 4, codes[sl]) + lane_off;
      // four rows per LDS round trip: the four cells are read together, row k
      // continues from the newest earlier row of the group with the same cell
      // (else from the value read), the four results are written in row order
      // -- bit for bit the row-by-row chain
#pragma unroll
      for (int u = 0; u < CWG_NB; u += 4) {
        float* cell[4];
        float x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          cell[k] = reinterpret_cast<float*>(reinterpret_cast<char*>(wtab) + addr[u + k]);
          x[k] = *cell[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float from = x[k];
#pragma unroll
          for (int m = 0; m < k; ++m)       // ascending: the newest match wins
            from = addr[u + m] == addr[u + k] ? y[m] : from;
          y[k] = from + v[u + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *cell[k] = y[k];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // (a slab whose row range is empty still writes its zeros)
  f32x4* out = reinterpret_cast<f32x4*>(slabs + slab * (2L * Q * 32));
#pragma unroll
  for (int tap = 0; tap < 2; ++tap) {
    const f32x4* src = reinterpret_cast<const f32x4*>(wtab + tap * trows * 32);
    for (int i = lane; i < Q * 8; i += 64) out[tap * Q * 8 + i] = src[i];
  }
}

int wn_causal_wgrad_slabs(long rows) {
  long n = rows / 256;           // >= 256 rows per wave
  // (500 slabs: kernel 34 -> 23 us, their reduction 11 -> 19 us)
  if (n > 256) n = 256;
  if (n < 2) n = 2;
  return (int)(n & ~1L);         // two waves per workgroup
}

int wn_causal_wgrad(const int32_t* q, const float* dx0, float* slabs,
                    int num_slabs, int B, int T, int Q, void* stream) {
  if (!q || !dx0 || !slabs) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || Q <= 0 || num_slabs < 2 || (num_slabs & 1))
    return WN_ERR_BAD_SHAPE;
  if (Q > CWG_QMAX) return WN_ERR_UNSUPPORTED;
  const long rows = (long)B * T;
  const long rps = (rows + num_slabs - 1) / num_slabs;
  hipLaunchKernelGGL(causal_wgrad_kernel, dim3(num_slabs / 2), dim3(128), 0,
                     (hipStream_t)stream, q, dx0, slabs, rows, rps, T, Q);
  return wn_check_launch();
}

int wn_xent_partials(long rows) {
  long g = (rows + 3) / 4;
  if (g > 1024) g = 1024;
  return (int)g;
}

int wn_xent(const float* logits, long ld, const int32_t* q, float* dlogits,
            float* loss_partials, int B, int T, int Q, int tf_quirk,
            void* stream) {
  if (!logits || !q || !loss_partials) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || Q <= 0) return WN_ERR_BAD_SHAPE;
  if ((Q & 3) || (ld & 3)) return WN_ERR_UNSUPPORTED;
  if (!wn_aligned16(logits) || (dlogits && !wn_aligned16(dlogits)))
    return WN_ERR_MISALIGNED;
  const long rows = (long)B * T;
  const float inv_n = 1.0f / (float)rows;
  hipLaunchKernelGGL(xent_kernel, dim3(wn_xent_partials(rows)), dim3(256), 0,
                     (hipStream_t)stream, logits, ld, q, dlogits,
                     loss_partials, rows, T, Q, inv_n, tf_quirk);
  return wn_check_launch();
}

int wn_softmax64_row(const float* logits_row, int Q, float* proba,
                     void* stream) {
  if (!logits_row || !proba) return WN_ERR_NULL;
  if (Q <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(softmax64_row_kernel, dim3(1), dim3(256), 0,
                     (hipStream_t)stream, logits_row, Q, proba);
  return wn_check_launch();
}

int wn_adam(float* p, const float* g, float* m, float* v, long n, float lr_t,
            float beta1, float beta2, float eps, float grad_scale, float l2,
            const float* l2_mask, void* stream) {
  if (!p || !g || !m || !v) return WN_ERR_NULL;
  if (n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, p, g, m, v, n, lr_t, beta1, beta2,
                     eps, grad_scale, l2, l2_mask);
  return wn_check_launch();
}

int wn_momentum(float* p, const float* g, float* acc, long n, float lr,
                float momentum, float grad_scale, float l2,
                const float* l2_mask, void* stream) {
  if (!p || !g || !acc) return WN_ERR_NULL;
  if (n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(momentum_kernel, dim3(grid1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, p, g, acc, n, lr, momentum,
                     grad_scale, l2, l2_mask);
  return wn_check_launch();
}

int wn_rmsprop(float* p, const float* g, float* ms, float* mom, long n,
               float lr, float decay, float momentum, float eps,
               float grad_scale, float l2, const float* l2_mask,
               void* stream) {
  if (!p || !g || !ms || !mom) return WN_ERR_NULL;
  if (n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, p, g, ms, mom, n, lr, decay,
                     momentum, eps, grad_scale, l2, l2_mask);
  return wn_check_launch();
}

int wn_l2_partials_count(void) { return 256; }

int wn_l2_partials(const float* p, long n, const float* mask, float* partials,
                   void* stream) {
  if (!p || !partials) return WN_ERR_NULL;
  if (n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(l2_partials_kernel, dim3(256), dim3(256), 0,
                     (hipStream_t)stream, p, n, mask, partials);
  return wn_check_launch();
}

int wn_gc_bias(const float* layer0, long layer_stride, long off_bias,
               long off_gc, int G, const float* emb, int card,
               const int32_t* ids, float* out, int L, int B, int ch,
               void* stream) {
  if (!layer0 || !out) return WN_ERR_NULL;
  if (emb && !ids) return WN_ERR_NULL;
  if (L <= 0 || B <= 0 || ch < 32) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(gc_bias_kernel, dim3(L, B), dim3(2 * ch < 1024 ? 2 * ch : 1024), 0,
                     (hipStream_t)stream, layer0, layer_stride, off_bias,
                     off_gc, G, emb, card, ids, out, B, ch);
  return wn_check_launch();
}

int wn_colsum_clip_chunks(int T) {
  int c = (T + 511) / 512;
  if (c > 64) c = 64;
  return c < 1 ? 1 : c;
}

// out[b][0:32] = sum_t P0[b][t][:], out[b][32:64] = sum_t P1[b][t][:];
// `part` is caller scratch of B * wn_colsum_clip_chunks(T) * 64 floats.
int wn_colsum_clip(const float* plane0, const float* plane1, int B, int T,
                   float* part, float* out, void* stream) {
  if (!plane0 || !plane1 || !part || !out) return WN_ERR_NULL;
  if (B <= 0 || T <= 0) return WN_ERR_BAD_SHAPE;
  const int chunks = wn_colsum_clip_chunks(T);
  const int rpc = (T + chunks - 1) / chunks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_clip_kernel, dim3(chunks, B), dim3(256), 0, s,
                     plane0, plane1, T, rpc, part);
  // fixed-order sum over the chunks (reduce_slabs_kernel lives in wn_gemm.hip)
  return wn_reduce_slabs(part, chunks, 64, B, (long)chunks * 64, 0, 64, out, 64,
                         1, 0, stream);
}

// scratch: L * B * G floats (the per-layer shares of the embedding gradient)
int wn_gc_grad(const float* layer0, long layer_stride, long off_gc, int G,
               const float* emb, int card, const int32_t* ids,
               const float* dsum, int L, int B, float* glayer0, float* gemb,
               float* scratch, int ch, void* stream) {
  if (!layer0 || !emb || !ids || !dsum || !glayer0 || !gemb || !scratch)
    return WN_ERR_NULL;
  if (L <= 0 || B <= 0 || G <= 0 || card <= 0 || ch < 32) return WN_ERR_BAD_SHAPE;
  if ((size_t)B * G * sizeof(float) > 48 * 1024) return WN_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gc_grad_kernel, dim3(2 * L), dim3(256), 0, s, layer0,
                     layer_stride, off_gc, G, emb, card, ids, dsum, L, B,
                     glayer0, scratch, ch);
  hipLaunchKernelGGL(gc_grad_emb_kernel, dim3(1), dim3(256),
                     (size_t)B * G * sizeof(float), s, scratch, L, B, G, card,
                     ids, gemb);
  return wn_check_launch();
}

int wn_causal_conv(const float* x, const float* w, float* y, int B, int T,
                   int Cin, int Cout, int K, int dilation, void* stream) {
  if (!x || !w || !y) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || K <= 0 || dilation <= 0)
    return WN_ERR_BAD_SHAPE;
  const long total = (long)B * T * Cout;
  hipLaunchKernelGGL(causal_conv_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, x, w, y, B, T, Cin,
                     Cout, K, dilation);
  return wn_check_launch();
}

int wn_time_to_batch(const float* in, float* out, int B, int T, int C,
                     int dilation, void* stream) {
  if (!in || !out) return WN_ERR_NULL;
  if (B <= 0 || T <= 0 || C <= 0 || dilation <= 0) return WN_ERR_BAD_SHAPE;
  const int U = (T + dilation - 1) / dilation;
  const long total = (long)B * dilation * U * C;
  hipLaunchKernelGGL(time_to_batch_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, in, out, B, T, C,
                     dilation, U);
  return wn_check_launch();
}

int wn_batch_to_time(const float* in, float* out, int B_out, int U, int C,
                     int dilation, void* stream) {
  if (!in || !out) return WN_ERR_NULL;
  if (B_out <= 0 || U <= 0 || C <= 0 || dilation <= 0) return WN_ERR_BAD_SHAPE;
  const long total = (long)B_out * dilation * U * C;
  hipLaunchKernelGGL(batch_to_time_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, in, out, B_out, U, C,
                     dilation);
  return wn_check_launch();
}

int wn_axpy(float* y, const float* x, float a, const float* mask, long n,
            void* stream) {
  if (!y || !x) return WN_ERR_NULL;
  if (n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, y, x, a, mask, n);
  return wn_check_launch();
}

int wn_fill(float* p, long n, float value, void* stream) {
  if (!p) return WN_ERR_NULL;
  if (n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(fill_kernel, dim3(grid1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, p, n, value);
  return wn_check_launch();
}

int wn_sum_rows(const float* in, int rows, int n, float* out, void* stream) {
  if (!in || !out) return WN_ERR_NULL;
  if (rows <= 0 || n <= 0) return WN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sum_rows_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, in, rows, n, out);
  return wn_check_launch();
}

}  // extern "C"
