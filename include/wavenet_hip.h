/* wavenet_hip.h -- C ABI of libwavenet_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the reference (jyegerlehner/tensorflow-wavenet) is
 * its Python class API (wavenet/__init__.py:1-4); it has no FFI of its own.
 * This C ABI is the new, internal boundary under the Python host
 * (tensorflow-wavenet_amd/wavenet): each entry point names the reference
 * code it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Contract (all entry points)
 *   - plain C: raw device pointers + explicit sizes, no torch / C++ types;
 *   - every buffer is allocated and owned by the caller; the library never
 *     allocates, frees or retains a pointer past return;
 *   - asynchronous on the caller's `stream` (a hipStream_t passed as void*),
 *     no internal synchronisation, no mutable globals: re-entrant, safe for
 *     one process per GPU and for several streams;
 *   - returns WN_OK (0) or a negative WN_ERR_* code; no C++ exception
 *     crosses the boundary;
 *   - activation "planes" are [rows][32] fp32 with rows = B*T (reference
 *     layout [B,T,C], channels innermost, residual/dilation channels zero-
 *     padded to 32); weights keep the reference's [K][Cin][Cout] order.
 *   - device pointers and leading dimensions must be 16-byte aligned.
 */
#ifndef WAVENET_HIP_H_
#define WAVENET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_OK 0
#define WN_ERR_BAD_SHAPE (-1)
#define WN_ERR_UNSUPPORTED (-2)
#define WN_ERR_MISALIGNED (-3)
#define WN_ERR_LAUNCH (-4)
#define WN_ERR_NULL (-5)

int wn_version(void);
const char* wn_error_string(int code);

/* ---- mu-law companding: wavenet/ops.py:65-73 (encode), :76-85 (decode) ---
 * Host: build the Q-1 float32 decision thresholds / the Q-entry decode table
 * of the float32 chain (host memory).  Device: bit-exact encode by threshold
 * search, decode by table lookup. */
int wn_mu_law_thresholds_host(int Q, float* thr_out);
int wn_mu_law_decode_table_host(int Q, float* lut_out);
int wn_mu_law_encode(const float* audio, int32_t* codes, long n,
                     const float* thr_dev, int Q, void* stream);
int wn_mu_law_decode(const int32_t* codes, float* audio, long n,
                     const float* lut_dev, int Q, void* stream);

/* ---- causal layer on one-hot input as a gather: wavenet/model.py:227-234
 * (_create_causal_layer) + :518-531 (_one_hot).  Wc is [K][Q][ldw] (the
 * 32 channels of one block start at Wc; ldw = padded channel count, 32 for
 * <= 32 residual channels), K = filter_width taps at shifts (K-1-k) + (K-1)/2;
 * x0 is the [B*T][32] plane of that block. */
int wn_causal_gather(const int32_t* q, const float* Wc, float* x0, int B,
                     int T, int Q, int K, int ldw, void* stream);

/* ---- causal layer on scalar input (scalar_input=True): wavenet/model.py:
 * 143-153, 227-234, 646-648; W is [K0][ldw] (ldw = padded channel count; for
 * more than 32 channels one call per 32-wide block with W + 32 * block and
 * that block's plane), K0 = initial_filter_width (any).
 * The wgrad writes [splits][K0*32] slabs for wn_reduce_slabs (per block). */
int wn_scalar_causal_fwd(const float* audio, const float* W, int ldw, float* x0,
                         int B, int T, int K0, void* stream);
int wn_scalar_causal_wgrad(const float* audio, const float* dx0, float* slabs,
                           int splits, int B, int T, int K0, void* stream);
/* one-hot causal layer, filter width 2: dWc[tap][v][32] as per-wave slabs
 * [num_slabs][2][Q][32] (reduce with wn_reduce_slabs, slab stride 2*Q*32);
 * Q <= 256, num_slabs = wn_causal_wgrad_slabs(B*T) */
int wn_causal_wgrad_slabs(long rows);
int wn_causal_wgrad(const int32_t* q, const float* dx0, float* slabs,
                    int num_slabs, int B, int T, int Q, void* stream);

/* ---- fused residual block: wavenet/model.py:236-330
 * (_create_dilation_layer) incl. both causal_conv calls, ops.py:46-62.
 * wblock = Wf[2][32][32] Wg[2][32][32] Wd[32][32] bf[32] bg[32] bd[32].
 * bias_fg: [B or 1][64] per-clip (bias + global-conditioning) or NULL. */
/* save_ts: 0 = nothing kept for backward (inference); 1 = tanh and sigmoid
 * planes (th, sg; the generic-tap *_k / channel-block *_blk kernels); 2 = the
 * sigmoid plane only (sg; th may be NULL) for wn_layer_bwd2, which recovers
 * tanh = z / sigmoid. */
int wn_layer_fwd(const float* x, float* x_out, float* z, float* th, float* sg,
                 const float* wblock, const float* bias_fg,
                 int bias_clip_stride, int B, int T, int dilation,
                 int has_dense, int save_ts, void* stream);
/* floats of a per-workgroup weight-gradient slab (layout = wblock) */
int wn_layer_wgrad_slab_floats(void);

/* generic filter width K >= 2 (off-default; the K = 2 kernels above are the
 * tuned path): block = Wf[K][32][32] Wg[K][32][32] Wd[32][32] bf bg bd,
 * tap k reads x[t - (K-1-k + (K-1)/2) * dilation] (ops.py:46-62). */
int wn_layer_fwd_k(const float* x, float* x_out, float* z, float* th,
                   float* sg, const float* wblock, const float* bias_fg,
                   int bias_clip_stride, int B, int T, int dilation, int K,
                   int has_dense, int save_ts, void* stream);
int wn_layer_bwd_k(const float* daf_cur, const float* dag_cur,
                   const float* dxin, float* dx_out, const float* wblock_b,
                   const float* dZ, const float* th, const float* sg,
                   const float* wblock_a, float* daf_next, float* dag_next,
                   int B, int T, int dilation, int K, int do_b, int do_a, int blocks, long blk_stride,
                   void* stream);
/* (k0 / Ktot: the K taps k0 .. k0 + K - 1 of a filter of Ktot taps; slab
 * layout [(2K+1) * 1024 + 96]: Wf taps, Wg taps, Wd, bf | bg | bd.  CB > 1: all
 * CB x CB (input block a, output block b) pairs of a channel-block layer in one
 * launch -- x / z are the first input plane, daf / dag / dxin the first output
 * plane, planes plane_stride floats apart, slabs [pair = a * CB + b][num_slabs]) */
int wn_layer_wgrad_k(const float* x, const float* daf, const float* dag,
                     const float* z, const float* dxin, float* slabs,
                     int num_slabs, int B, int T, int dilation, int K, int k0,
                     int Ktot, int CB, long plane_stride, void* stream);

/* out planes = addend planes + in planes * W (+ bias): the 1x1 residual conv
 * x_{l+1} = x_l + z_l Wd (+ bd) of wavenet/model.py:294-300, 330 for the
 * channel-block models (C = 32, 64, 96 or 128 padded channels = C / 32 planes of
 * [rows][32]) and its data gradient dz = dZ + dx_{l+1} Wd^T, as one streaming
 * launch per layer (a wave per 32-row tile) instead of a K = N = C plane-mode
 * wn_gemm_nn.  W [C][C] row-major ([Cin][Cout]); bias [C] or NULL; addend in
 * the planes' layout or NULL.  Other widths: WN_ERR_UNSUPPORTED (use wn_gemm_nn). */
int wn_dense_planes(const float* in, long in_plane_stride, const float* W,
                    const float* bias, const float* addend, long add_plane_stride,
                    float* out, long out_plane_stride, long rows, int C,
                    void* stream);
/* The backward's dz = dZ + dx_{l+1} Wd^T (wn_dense_planes with addend = dZ) and
 * the gate gradients of autodiff of wavenet/model.py:264-282 in one launch:
 * daf = dz s (1 - t^2), dag = dz s t (1 - s) with the saved tanh / sigmoid
 * planes th / sg; dz itself is not stored.  Bitwise the planes the two launches
 * (wn_dense_planes, then wn_layer_bwd_k with do_a only) give. */
int wn_dense_planes_gate(const float* in, long in_plane_stride, const float* W,
                         const float* addend, long add_plane_stride, const float* th,
                         const float* sg, long ts_plane_stride, float* daf, float* dag,
                         long da_plane_stride, long rows, int C, void* stream);

/* more than 32 residual / dilation channels: channels are cut into 32-wide
 * blocks, each block of an activation is its own [B*T][32] plane, and one
 * launch computes ONE output block from all input blocks (weights in the
 * reference's [K][Cin][Cout] layout, row stride ldw = padded channel count).
 * fwd: z / tanh / sigmoid planes of dilation block jb; wf / wg point at column
 * 32 jb of Wf / Wg; x is input block 0, block i at x + i * in_plane_stride.
 * bwd: dx plane of residual block rb from the da planes of all dilation
 * blocks; wf / wg point at row 32 rb; tap_stride = floats between taps.
 * K * blocks <= 8 PER CALL; wider layers run in chunks of blocks: fwd with
 * in_blocks = the chunk, x / wf / wg offset to its first block, tap_rows = the
 * rows between two taps of the weight matrix (the padded channel count), the
 * partial pre-activations handed from pre_out (planes af | ag, pre_plane_stride
 * apart; no gate, z / th / sg untouched) to the next call's pre_in (which
 * replaces the bias); bwd with da_blocks = the chunk and dxin = the previous
 * chunk's dx_out.  The 1x1 convs of such a layer are wn_gemm_nn calls in
 * plane mode; its weight gradients come from wn_layer_wgrad_k per block pair.
 * K / k0 / Ktot: the call covers taps k0 .. k0 + K - 1 of a filter of Ktot
 * taps (K * blocks <= 8 per call: wide layers run in chunks of blocks, filter
 * widths above 8 in groups of taps, chained through pre_in / pre_out and
 * dxin / dx_out).  out_blocks / dx_blocks > 1: that many consecutive output
 * (residual) blocks in one launch -- planes out_plane_stride (dx_plane_stride)
 * floats apart, weight columns (rows) and bias entries 32 further per block,
 * partial pre-activation planes [block][af | ag]. */
int wn_layer_fwd_blk(const float* x, long in_plane_stride, int in_blocks,
                     float* z, float* th, float* sg, const float* wf,
                     const float* wg, int ldw, const float* bias_f,
                     const float* bias_g, int bias_clip_stride, int B, int T,
                     int dilation, int K, int save_ts, int tap_rows,
                     const float* pre_in, float* pre_out, long pre_plane_stride,
                     int k0, int Ktot, int out_blocks, long out_plane_stride,
                     void* stream);
int wn_layer_bwd_blk(const float* daf, const float* dag, long da_plane_stride,
                     int da_blocks, const float* dxin, float* dx_out,
                     const float* wf, const float* wg, int ldw, long tap_stride,
                     int B, int T, int dilation, int K, int k0, int Ktot,
                     int dx_blocks, long dx_plane_stride, void* stream);

/* default backward of one block (TF autodiff of model.py:236-330), no
 * pre-activation-gradient planes in HBM: every tile recomputes da for its
 * rows t and t+d from dZ_l, dx_{l+1}, z_l and the sigmoid plane, so the only
 * plane written is dx_l (768 B of HBM traffic per sample and layer instead of
 * 1408 B).  dxin == NULL for the last layer (its dense output is unused,
 * model.py:294-300).  slabs: [wn_layer_bwd2_slabs(B, T)][wblock layout] weight
 * gradients of layer l; tile_colsum (optional): [B * ceil(T/32)][64] per-tile
 * column sums of da_f | da_g, reduced per clip for the global-conditioning
 * gradients. */
int wn_layer_bwd2_slabs(int B, int T);
/* wimg: this layer's image out of wn_layer_bwd2_pack (the five matrices
 * transposed with row stride 33, wn_layer_bwd2_wimg_floats() floats per layer,
 * L images back to back; rebuilt once per step), which the kernel pulls into
 * LDS by LDS-DMA; wblock is the plain layer block (A/B kernel variants). */
int wn_layer_bwd2_wimg_floats(void);
int wn_layer_bwd2_pack(const float* layer0, long layer_stride, float* wimg,
                       int L, void* stream);
int wn_layer_bwd2(const float* x, const float* z, const float* sg,
                  const float* dZ, const float* dxin, float* dx_out,
                  const float* wblock, const float* wimg, float* slabs,
                  float* tile_colsum, int B, int T, int dilation,
                  void* stream);

/* ---- the whole residual stack in ONE persistent launch (csrc/wn_stack.hip):
 * replaces the per-layer loop of WaveNetModel._create_network
 * (wavenet/model.py:417-428 over _create_dilation_layer, model.py:236-330) and
 * its gradient; same arithmetic per 32-row tile as wn_layer_fwd / wn_layer_bwd2.
 * A workgroup keeps a group of consecutive tiles for all L layers; the tile's
 * own rows stay in registers (forward), the dilated tap of another tile is
 * handed over through memory: sc1 stores, one flag per (layer, tile), sc1
 * loads.  No grid barrier; groups are handed out by an atomic ticket in
 * dependency order, so the launch completes whatever the residency.
 *   X, Z, SG, dZ     : [L][B*T][32] planes, layer-major (X[0] = causal layer
 *                      output; forward writes X[1..L-1], Z, SG; backward reads
 *                      X, Z, SG, dZ)
 *   DX, dx_layer_stride : the backward writes, at DX + l * dx_layer_stride
 *                      floats, dL/dx_0 for l = 0 and for l > 0 dL/dx_l WITHOUT the
 *                      anti-causal tap's term, which is Q[l] at the rows d_l
 *                      later.  dx_layer_stride = B*T*32 keeps every layer's
 *                      plane ([L][B*T][32]); 0: ONE [B*T][32] plane rewritten in
 *                      place from layer to layer (a tile's own rows have no
 *                      other reader; they stay in the L2 / Infinity Cache
 *                      instead of travelling to HBM and back every layer)
 *   Q                : an [L][B*T][32] scratch: a tile publishes q_l[s] =
 *                      da_l[s] W[0]^T, what its rows contribute to the rows d
 *                      earlier (the "push" formulation; every tile re-deriving
 *                      da at the rows d later cost more bytes and MFMAs)
 *   wimg             : [L][wn_stack_wimg_floats()] weight images out of
 *                      wn_stack_pack (rows of 36 floats so that four MFMA
 *                      operands are one 16-byte LDS read; forward: the five
 *                      matrices transposed + the dense bias, backward: as they
 *                      are; rebuilt whenever the parameters changed), pulled
 *                      into LDS by LDS-DMA.  Forward and backward take their
 *                      own image.
 *   bias             : [L][B or 1][64] filter|gate bias (+gc) of wn_gc_bias, or NULL
 *   dilations        : [L] int32, DEVICE memory
 *   flags            : wn_stack_flag_count(B, T, L) uint32, zero before first use
 *   ctl              : 4 uint32 {0, 0, 1, 0} before first use: {group ticket,
 *                      workgroups done, epoch, error}; the kernel re-arms the
 *                      first three.  ctl[3] != 0 after a launch: a bounded flag
 *                      wait (2 s) expired -- results are invalid
 *   poison           : NULL, or one float (0 before first use) that is set to
 *                      NaN in that case: summed into the loss reduction it
 *                      makes the failure loud without a host synchronisation
 *   forward and backward use SEPARATE flags / ctl buffers.
 *   slabs            : [L][slab_layer_stride floats], slab g of layer l at
 *                      l * slab_layer_stride + g * 5216, g < wn_stack_bwd_slabs(B, T, variant)
 *   tilesum          : as wn_layer_bwd2 ([L][tiles][64] or NULL), tiles =
 *                      B * ceil(T / wn_stack_tile_rows(B, T, variant))
 * Tile height: 32 rows; 16 for small batches (wn_stack_tile_rows: at most four
 * 32-row tiles per CU -- the launches are then bound by one wave's dependent
 * path through a layer, and a 16-row tile on v_mfma_f32_16x16x4_f32 halves
 * that path).  Both launches of a shape use the same height; it sets the slab count
 * (wn_stack_bwd_slabs) and the tilesum layout.  The 16-row results agree with
 * the 32-row ones to rounding (another summation grouping), not bitwise.
 *   variant          : 0 = the library's choice for the shape (what every
 *                      production caller passes), or a word built with
 *                      WN_STACK_VARIANT for A/B runs and tests.  The same word
 *                      goes to wn_stack_tile_rows / wn_stack_bwd_slabs /
 *                      wn_stack_fwd / wn_stack_bwd of one step: behaviour is a
 *                      function of the arguments only (no process environment
 *                      is read anywhere in the library).
 * L <= 256. */
/* rows: 0 (auto), 16 or 32 rows per tile; waves: 0 (auto) or waves per
 * workgroup (16-row launches: 4 / 8; 32-row backward: 1 / 2 / 4 / 8; the 32-row
 * forward ignores it) */
#define WN_STACK_VARIANT(rows, waves) (((rows) & 0x3f) | (((waves) & 0xf) << 8))
long wn_stack_flag_count(int B, int T, int L);
int wn_stack_tile_rows(int B, int T, int variant);
int wn_stack_wimg_floats(void);
int wn_stack_pack(const float* layer0, long layer_stride, float* wimg_fwd,
                  float* wimg_bwd, int L, void* stream);
int wn_stack_fwd(float* X, float* Z, float* SG, const float* wimg,
                 const float* bias, long bias_layer_stride,
                 int bias_clip_stride, const int* dilations, unsigned* flags,
                 unsigned* ctl, float* poison, int L, int B, int T, int save_sg,
                 int variant, void* stream);
/* Small batches (wn_stack_fwd_skip_ok: the 16-row launch with one tile per
 * SIMD, 512 skip channels): wn_stack_fwd that ALSO computes the skip sum,
 *   h1[B*T][512] = relu(sum_l z_l Ws_l + skip_bsum)
 * (model.py:505-509: what wn_gemm_nn over the Z planes computes afterwards) in
 * the same launch -- a partner wave per tile takes the tile's z of every layer
 * through LDS while the chain wave goes on; the launch's matrix pipe is three
 * quarters idle otherwise.  skip_img: wn_stack_skip_img_floats(L) floats written
 * by wn_stack_skip_pack from skip_w [L * 32][512] (a function of the weights
 * only; the Python host repacks it on every forward call);
 * skip_bsum: [512] or NULL.  Same sums as the GEMM up to the order of additions. */
int wn_stack_fwd_skip_ok(int B, int T, int S, int variant);
long wn_stack_skip_img_floats(int L);
int wn_stack_skip_pack(const float* skip_w, int L, float* img, void* stream);
int wn_stack_fwd_skip(float* X, float* Z, float* SG, const float* wimg,
                      const float* bias, long bias_layer_stride,
                      int bias_clip_stride, const int* dilations, unsigned* flags,
                      unsigned* ctl, float* poison, int L, int B, int T, int save_sg,
                      int variant, const float* skip_img, const float* skip_bsum,
                      float* h1, void* stream);
int wn_stack_bwd_slabs(int B, int T, int variant);
int wn_stack_bwd(const float* X, const float* Z, const float* SG,
                 const float* dZ, float* DX, long dx_layer_stride, float* Q,
                 const float* wimg, float* slabs,
                 long slab_layer_stride, float* tilesum, const int* dilations,
                 unsigned* flags, unsigned* ctl, float* poison, int L, int B,
                 int T, int variant, void* stream);

/* ---- fp32 MFMA GEMMs: the skip sum + post-processing of
 * wavenet/model.py:303-305, 430-440 (_create_network) and their gradients.
 * C = epi(A W): + bias, relu, * (mask > 0), + addend.  A dense [M][lda] or
 * a_planes planes of [M][32]; C dense [M][ldc] or c_planes planes of [M][32]
 * (c_plane_stride floats apart); an addend passed with ld_add == 0 is in C's
 * plane layout (needs c_planes > 0): the 1x1 residual conv of the
 * channel-block models, x_{l+1} planes = x_l planes + z_l Wd. */
int wn_gemm_nn(const float* A, long lda, int a_planes, long a_plane_stride,
               const float* W, int ldw, const float* bias, const float* mask,
               long ld_mask, const float* addend, long ld_add, float* C,
               long ldc, int c_planes, long c_plane_stride, float* Cpre,
               long M, int N, int K, int relu, void* stream);

/* opt-in: same contraction with fp32 accuracy rebuilt from bf16 matrix
 * instructions (every operand split exactly into three bf16 pieces, nprod =
 * 3 / 6 / 9 piece products; 6 is as accurate as the fp32 MFMA path).
 * w_scratch: wn_gemm_split_w_bytes(K, N) bytes, caller-owned.  Not used unless
 * the host asks for it (WaveNetModel.gemm_mode). */
long wn_gemm_split_w_bytes(int K, int N);
int wn_gemm_nn_split(const float* A, long lda, int a_planes,
                     long a_plane_stride, const float* W, int ldw,
                     const float* bias, const float* mask, long ld_mask,
                     const float* addend, long ld_add, float* C, long ldc,
                     int c_planes, long c_plane_stride, float* Cpre, long M,
                     int N, int K, int relu, void* w_scratch, int nprod,
                     void* stream);
/* floats per slab of wn_gemm_tn: the Mw x Nw matrix and
 * wn_gemm_tn_tail_rows(Mw, Nw) rows of column sums behind it */
long wn_gemm_tn_slab_floats(int Mw, int Nw);
int wn_gemm_tn_tail_rows(int Mw, int Nw);
/* recommended `splits` for wn_gemm_tn (grid = one resident wave of
 * workgroups); kind: 0 dense A, 1 one-hot A, 2 wn_gemm_tn_split */
int wn_gemm_tn_splits(long rows, int Mw, int Nw, int kind);
/* opt-in split-bf16 variant of wn_gemm_tn (see wn_gemm_nn_split); dense or
 * plane A, rows % 16 == 0 */
int wn_gemm_tn_split(const float* A, long lda, int a_planes,
                     long a_plane_stride, const float* G, long ldg,
                     float* slabs, int splits, long rows, int Mw, int Nw,
                     int want_colsum, int nprod, void* stream);
/* want_colsum: 0 none; 1 the column sums of G (the bias gradient) in the first
 * tail row of every slab; 2 ("spread", shapes with wn_gemm_tn_tail_rows > 1
 * only): every tile row of a split sums its share of the rows into its own
 * tail row -- no workgroup carries all the extra adds, none falls behind and
 * re-fetches its chunks -- and wn_reduce_slabs_mt(tail_rows) adds the rows up */
int wn_gemm_tn(const float* A, long lda, int a_planes, long a_plane_stride,
               const int32_t* codes, int shift, int T, const float* G,
               long ldg, float* slabs, int splits, long rows, int Mw, int Nw,
               int want_colsum, void* stream);
int wn_reduce_slabs(const float* slabs, int num_slabs, long slab_stride,
                    int batch, long in_batch_stride, long offset, long n,
                    float* dst, long out_batch_stride, int replicate,
                    long rep_stride, void* stream);
/* the slabs of one wn_gemm_tn in ONE launch: elements [0, n_main) of every
 * slab (the matrix) -> dst_main, the n_tail elements behind them (column sums
 * = bias gradient; tail_rows partial rows of n_tail each, summed) -> dst_tail,
 * `replicate` copies rep_stride floats apart.
 * All counts / strides multiples of 4 floats, pointers 16-byte aligned. */
int wn_reduce_slabs_mt(const float* slabs, int num_slabs, long slab_stride,
                       long n_main, float* dst_main, long n_tail, float* dst_tail,
                       int replicate, long rep_stride, int tail_rows, void* stream);
/* channel-block models (more than 32 residual / dilation channels,
 * model.py:46-60 puts no limit on them): the wn_layer_wgrad_k slabs of the
 * CB x CB block pairs of one layer, slabs[pair = a * CB + b][num_slabs]
 * [(2K+1)*1024 + 96], summed in a fixed order and written into the layer's
 * gradient block (Wf [K][C][C], Wg, Wd [C][C] at 0; bf | bg | bd [C] each at
 * off_bias, from the pairs with a == 0).  The slabs hold taps tap0 .. tap0 +
 * K - 1 of a filter of Ktot taps (filter widths above 8 run in tap groups). */
int wn_reduce_pair_slabs(const float* slabs, int num_slabs, int CB, int K,
                         int has_dense, int use_bias, float* layer_grad, int C,
                         long off_bias, int tap0, int Ktot, void* stream);
int wn_transpose(const float* in, int rows, int cols, long in_ld, float* out,
                 long out_ld, void* stream);

/* ---- loss: wavenet/model.py:654-666 (shifted one-hot softmax xent, mean),
 * forward + TF-compatible backward; predict softmax in float64
 * (model.py:584-585, 620-621) */
int wn_xent_partials(long rows);
int wn_xent(const float* logits, long ld, const int32_t* q, float* dlogits,
            float* loss_partials, int B, int T, int Q, int tf_quirk,
            void* stream);
int wn_softmax64_row(const float* logits_row, int Q, float* proba,
                     void* stream);

/* ---- optimizers with TensorFlow-0.10 update rules: wavenet/ops.py:6-24 */
int wn_adam(float* p, const float* g, float* m, float* v, long n, float lr_t,
            float beta1, float beta2, float eps, float grad_scale, float l2,
            const float* l2_mask, void* stream);
int wn_momentum(float* p, const float* g, float* acc, long n, float lr,
                float momentum, float grad_scale, float l2,
                const float* l2_mask, void* stream);
int wn_rmsprop(float* p, const float* g, float* ms, float* mom, long n,
               float lr, float decay, float momentum, float eps,
               float grad_scale, float l2, const float* l2_mask,
               void* stream);
int wn_l2_partials_count(void);
int wn_l2_partials(const float* p, long n, const float* mask, float* partials,
                   void* stream);

/* ---- global conditioning: wavenet/model.py:272-284, 533-562.  ch = padded
 * dilation channel count (32, or 64 with two channel blocks): out is
 * [L][B][2 ch] = filter | gate bias (+ embedding * Wgc), dsum likewise. */
int wn_gc_bias(const float* layer0, long layer_stride, long off_bias,
               long off_gc, int G, const float* emb, int card,
               const int32_t* ids, float* out, int L, int B, int ch,
               void* stream);
int wn_colsum_clip_chunks(int T);
int wn_colsum_clip(const float* plane0, const float* plane1, int B, int T,
                   float* part, float* out, void* stream);
int wn_gc_grad(const float* layer0, long layer_stride, long off_gc, int G,
               const float* emb, int card, const int32_t* ids,
               const float* dsum, int L, int B, float* glayer0, float* gemb,
               float* scratch /* L * B * G floats */, int ch, void* stream);

/* ---- thin exported ops of wavenet/__init__.py:1-4 (arbitrary shapes):
 * causal_conv ops.py:46-62, time_to_batch :27-34, batch_to_time :37-43 */
int wn_causal_conv(const float* x, const float* w, float* y, int B, int T,
                   int Cin, int Cout, int K, int dilation, void* stream);
int wn_time_to_batch(const float* in, float* out, int B, int T, int C,
                     int dilation, void* stream);
int wn_batch_to_time(const float* in, float* out, int B_out, int U, int C,
                     int dilation, void* stream);

/* ---- diagnostics: register-only fp32 MFMA loop (clock-limited ceiling) */
int wn_diag_mfma_peak(float* out, int blocks, int iters, void* stream);

/* ---- utilities */
int wn_axpy(float* y, const float* x, float a, const float* mask, long n,
            void* stream);
int wn_fill(float* p, long n, float value, void* stream);
int wn_sum_rows(const float* in, int rows, int n, float* out, void* stream);

/* ---- fast generation: wavenet/model.py:332-387, 444-516, 592-626
 * (_create_generator / predict_proba_incremental) and the host loop of
 * generate.py:213-241, as ONE persistent kernel with the FIFO state on the
 * device and on-device sampling. */
long wn_fastgen_state_floats(const int32_t* dilations_host, int L);
int wn_fastgen_init(float* state, long state_floats, int32_t* cursors, int L,
                    void* stream);   /* cursors: int32[>= 3] */
/* samples_io holds n_steps + 1 codes; the first n_given are inputs (seed /
 * priming, generate.py:195-210), the rest are drawn on the device.  Step i
 * consumes samples_io[i]; proba_out (optional) receives the next-sample
 * distribution of every proba_every-th step.  push = 0 (n_steps must be 1)
 * evaluates a step without advancing the queues (the reference's proba op run
 * without net.push_ops, test/test_generation.py:66-68). */
int wn_fastgen_run(const float* params_causal, const float* layer0,
                   long layer_stride, const float* skip_w,
                   const float* skip_bsum, const float* post1_w,
                   const float* post1_b, const float* post2_w,
                   const float* post2_b, const float* gc_bias_fg,
                   const int32_t* dilations_dev, int L, int S, int Q,
                   float* state, int32_t* cursors, int32_t* samples_io,
                   int n_given, int n_steps, float temperature, uint64_t seed,
                   float* proba_out, int proba_every, int use_biases,
                   int push, void* stream);

/* The same generator for C = 32 * blocks > 32 padded residual / dilation
 * channels (the reference's generator has no width limit, model.py:444-516):
 * layer blocks Wf[2][C][C] Wg[2][C][C] Wd[C][C] bf[C] bg[C] bd[C], causal
 * [2][Q][C], skip [L][C][S], gc_bias_fg [L][2 C], queue entries of C floats
 * (state_floats = sum(dilations) * C).  C <= 1024, S <= 4096, Q <= 4096,
 * L <= 1024 and 8 Q + 4 (5 C + 2 S + 3 L) <= 150 KB of LDS; C = 32 is accepted
 * too (the host uses this entry for 32-channel models beyond the S / Q <= 512,
 * L <= 64 of wn_fastgen_run / wn_fastgen_step).  One persistent workgroup,
 * correctness first -- except with `coop` scratch (S <= 512 a multiple of 16,
 * Q <= 512): a COOPERATIVE launch,
 * workgroup 0 runs the layers and the draw, 2 S / 16 + Q / 16 more workgroups the
 * skip sum and the post-processing mat-vecs (hand-over words in `coop`; the
 * same samples up to the rounding of the skip sum's order).  It is launched
 * with hipLaunchCooperativeKernel (residency is the runtime's promise) behind
 * the library's own occupancy check; refused, or with coop = NULL, the single
 * workgroup runs.
 * coop: NULL, or wn_fastgen_wide_coop_bytes(L, C, S, Q) bytes (16-byte aligned,
 * zeroed by the call).  After a cooperative run ((unsigned*)coop)[12] != 0
 * means a hand-over wait expired (2 s): that run's samples are not valid. */
long wn_fastgen_wide_coop_bytes(int L, int C, int S, int Q);
int wn_fastgen_run_wide(const float* params_causal, const float* layer0,
                        long layer_stride, const float* skip_w,
                        const float* skip_bsum, const float* post1_w,
                        const float* post1_b, const float* post2_w,
                        const float* post2_b, const float* gc_bias_fg,
                        const int32_t* dilations_dev, int L, int C, int S, int Q,
                        float* state, int32_t* cursors, int32_t* samples_io,
                        int n_given, int n_steps, float temperature,
                        uint64_t seed, float* proba_out, int proba_every,
                        int use_biases, int push, void* coop, void* stream);

/* Multi-CU variant: enqueues ONE generation step as four kernels (chain on
 * one CU, current tap only, preceded by the previous step's float64 softmax +
 * draw; skip sum + the NEXT step's past-tap pre-activations, conv1 and conv2
 * on S/16 / Q/16 CUs).  Everything step- or call-dependent lives in device memory
 * (cursors, ctl), so the host captures a few hundred calls into a hipGraph
 * ONCE and replays it for every generate() call.
 * ctl: int32[8] = {base (cursors[0] when the call started), n_given,
 * proba_every, temperature (float bits), seed lo, seed hi, 0, 0}.
 * pre: float[L][64], maintained by the step kernels; wn_fastgen_pre fills it
 * for the step the queues are at (call once before a sequence of steps). */
int wn_fastgen_step(const float* params_causal, const float* layer0,
                    long layer_stride, const float* skip_w,
                    const float* skip_bsum, const float* post1_w,
                    const float* post1_b, const float* post2_w,
                    const float* post2_b, const float* gc_bias_fg,
                    const int32_t* dilations_dev, int L, int S, int Q,
                    float* state, int32_t* cursors, int32_t* samples_io,
                    const int32_t* ctl, float* proba_out, int use_biases,
                    const float* cw_img, float* pre, float* z_all, float* h1,
                    float* h2, float* logits, void* stream);
/* cursors is int32[4] for the step path: {steps done, previous code, draw
 * pending, 0}.  A step's softmax / draw / cursor update runs at the start of
 * the NEXT step's chain kernel; wn_fastgen_finish does it for the last step
 * of a sequence (no-op when nothing is pending). */
int wn_fastgen_finish(int Q, int32_t* cursors, int32_t* samples_io,
                      const int32_t* ctl, float* proba_out, const float* logits,
                      void* stream);

/* ONE persistent multi-CU launch for n_steps samples (csrc/wn_fastgen.hip,
 * fg_persist_kernel): the same inputs, state and results as n_steps x
 * wn_fastgen_step + wn_fastgen_finish, without a kernel boundary per stage and
 * with every workgroup's weights resident for the run (chain segments of ~10
 * layers in LDS, skip / post-processing slices in LDS; the stages hand over
 * through sc1 payloads + relaxed agent-scope flags).  cursors[2] must be 0 on
 * entry.  sync: 16 uint32 (zeroed by the call; sync[12] != 0 afterwards = a
 * bounded wait expired, results invalid); ll: wn_fastgen_persist_ll_words(L, S,
 * Q) 8-byte hand-over words (payload + step in one store; zeroed by the call).
 * Needs all wn_fastgen_persist_workgroups(L, S, Q) workgroups resident at once:
 * checked against hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs for the
 * launch configuration, then launched with hipLaunchCooperativeKernel, which
 * makes residency the runtime's promise -- else WN_ERR_UNSUPPORTED, returned
 * before any generator state is written: use wn_fastgen_step.  Should a
 * hand-over wait expire all the same (2 s), it surfaces as sync[12] -- the
 * caller restores its own snapshot of state / cursors / pre and takes the step
 * kernels.  The chain's workgroups (draw, segments) take the blocks 0, 8, 16,
 * ... (one XCD: a hand-over word is 0.15 us faster inside an XCD). */
int wn_fastgen_persist_workgroups(int L, int S, int Q);
/* role (0 .. total - 1: chain segments, skip, post1, logits, draw) of workgroup
 * `block` of the launch: the chain's roles on the blocks 0, 8, 16, ... (host-side
 * mirror of the kernel's mapping, for tests) */
int wn_fastgen_persist_role(int block, int total, int nseg);
long wn_fastgen_persist_ll_words(int L, int S, int Q);
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
                       unsigned long long* ll, int n_steps, void* stream);
int wn_fastgen_pre(const float* layer0, long layer_stride,
                   const float* gc_bias_fg, const int32_t* dilations_dev,
                   int L, const float* state, const int32_t* cursors,
                   float* pre, void* stream);
/* cw_img [L][3072]: Wf[1] | Wg[1] | Wd of every layer, transposed + swizzled
 * for the chain's LDS ring */
int wn_fastgen_pack(const float* layer0, long layer_stride, float* img, int L,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVENET_HIP_H_ */
