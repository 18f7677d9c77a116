"""WaveNetModel on MI355X: host-side mirror of the reference's
wavenet/model.py::WaveNetModel (same constructor signature :46-60, same
`variables` nesting/names/shapes :118-225, same public methods `loss` :628,
`predict_proba` :564, `predict_proba_incremental` :592) driving hand-written
HIP kernels through the C ABI of include/wavenet_hip.h.

MI355X-first design:
  * all parameters live in ONE flat fp32 device buffer (`params`), all
    gradients in ONE flat bucket (`grads`) -> one RCCL all-reduce, one fused
    optimizer launch; `variables[...]` are reference-shaped views into it;
  * activations are [B*T][32] planes; each residual block is one fused MFMA
    kernel; the 50 skip 1x1 convs are a single [B*T, L*32] x [L*32, S] GEMM;
  * the one-hot input tensor is never materialised (causal layer = gather,
    its weight gradient = one-hot-on-the-fly MFMA contraction);
  * backward is hand-written (the reference relies on TF autodiff).
There is no CPU / PyTorch compute fallback: without the HIP library or a GPU
every entry point raises.
"""
import math
import os

import numpy as np
import torch

from . import _lib
from .ops import mu_law_encode, mu_law_decode, mu_law_tables

CH = 32                      # channels per activation plane (one block)
# layer block (floats) for filter width K and C = 32 * blocks padded channels:
#   Wf[K][C][C] Wg[K][C][C] Wd[C][C] bf[C] bg[C] bd[C] (+ gc weights [G][C] x 2)


def layer_w(K, C=CH):
    return (2 * K + 1) * C * C


def _align(n, a=32):
    return (n + a - 1) // a * a


def _xavier_(t, gen):
    """tf.contrib.layers.xavier_initializer_conv2d (uniform), model.py:10:
    limit sqrt(6 / (fan_in + fan_out)) with the receptive field folded in."""
    shape = tuple(t.shape)
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    lim = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
    vals = (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * lim
    t.copy_(vals.to(t.device))


class _Workspace(object):
    """Caller-owned device buffers for one (B, T) shape (the library never
    allocates).  Sized for 288 GB HBM: everything stays resident.  A workspace
    for a smaller T (same B) is carved out of an existing larger one
    (`parent`) without allocating -- the windowed naive generation path calls
    predict_proba with a growing T."""

    def __init__(self, net, B, T, training, parent=None):
        dev = net.device
        L, S, Q = net.L, net.S, net.Q
        CB, CHn = net.CB, net.CHn       # channel blocks, padded channels
        LP = L * CB                     # activation planes per tensor
        N = B * T
        f32 = dict(dtype=torch.float32, device=dev)
        self.B, self.T, self.N, self.training = B, T, N, training
        # (the variant word of the stack launches is fixed per workspace)
        self.stack_variant = int(net.stack_variant)
        self.capacity = N if parent is None else parent.capacity
        lib = _lib.load()

        def alloc(name, shape, dtype=torch.float32, fill=None):
            n = int(np.prod(shape))
            if parent is not None and getattr(parent, name, None) is not None:
                t = getattr(parent, name).reshape(-1)[:n].view(shape)
            else:
                fresh.add(name)
                if fill is None:
                    t = torch.empty(shape, dtype=dtype, device=dev)
                else:
                    t = torch.full(shape, fill, dtype=dtype, device=dev)
            setattr(self, name, t)
            return t

        fresh = set()   # buffers this workspace owns (not views of the parent's)

        self.plans = {}
        alloc('q', (N,), torch.int32)
        self.gc_ids = alloc('gc_ids', (B,), torch.int32) \
            if net.card is not None else None
        self.audio = alloc('audio', (N,)) if net.scalar_input else None
        alloc('X', (LP, N, CH))
        alloc('Z', (LP, N, CH))
        alloc('h1', (N, S))
        alloc('h2', (N, S))
        alloc('logits', (N, Q))
        alloc('bias_fg', (L, B, 2 * CHn))
        # persistent residual-stack kernels (wn_stack_fwd): one "rows are in
        # memory" flag per (layer, 32-row tile) and a 4-word control block
        # {group ticket, workgroups done, epoch, error}; epochs start at 1
        alloc('stack_flags', (lib.wn_stack_flag_count(B, T, L),), torch.int32,
              fill=0)
        alloc('wimg_f', (L, lib.wn_stack_wimg_floats()))
        alloc('stack_ctl', (4,), torch.int32, fill=0)
        if 'stack_ctl' in fresh:         # (a view shares the owner's epoch)
            self.stack_ctl[2] = 1
        alloc('bsum', (S,))
        self.total = alloc('total', (N, S)) if net.residual_postproc else None
        self.nparts = lib.wn_xent_partials(N)
        # (the first 2 words: the NaN "poison" of wn_stack_fwd / wn_stack_bwd,
        # summed into the loss with the partials that follow them: an expired
        # wait turns the loss NaN; in front so that a carved-out workspace
        # shares them)
        alloc('loss_parts', (2 + self.nparts,), fill=0.0)
        alloc('loss', (1,), fill=0.0)
        alloc('proba', (Q,))
        if net.blocked:
            # partial pre-activations of a layer wider than one chunk of
            # channel blocks (wavenet/blocked.py), planes af | ag
            alloc('pre', (2 * CB, N, CH))
        if not training:
            return
        # the generic-tap / channel-block backward kernels also need the tanh
        # plane and two ping-pong pairs of pre-activation-gradient planes; the
        # default wn_stack_bwd / wn_layer_bwd2 do not
        self.legacy = net._legacy_bwd()
        self.TH = alloc('TH', (LP, N, CH)) if self.legacy else None
        alloc('SG', (LP, N, CH))
        alloc('dZ', (LP, N, CH))
        alloc('dc1', (N, S))
        alloc('dtotal', (N, S))
        self.dh2 = alloc('dh2', (N, S)) if net.residual_postproc else None
        self.c1 = alloc('c1', (N, S)) if net.residual_postproc else None
        self.da = alloc('da', (2, 2 * CB, N, CH)) if self.legacy else None
        alloc('dx', (2, CB, N, CH))
        # persistent backward (wn_stack_bwd), "push" formulation: a tile's
        # own dx rows have no reader but the wave that wrote them, so ONE
        # plane is rewritten in place from layer to layer (it stays in the L2
        # / Infinity Cache; DX[0] ends up as dL/dx_0).  The per-layer checks of
        # tests/test_gpu_stack.py (`net.stack_bwd_keep_dx`) keep dL/dx_l of
        # EVERY layer.  Plus the q planes, flags and control block (allocated
        # whenever the option could apply)
        self.stack_bwd = (net.stack_bwd and not net.blocked and not net.generic_layers
                          and L <= 256 and N * CH * 4 < 2 ** 31)
        if self.stack_bwd:
            self.keep_dx = bool(net.stack_bwd_keep_dx)
            if parent is not None and getattr(parent, 'DX', None) is not None:
                self.keep_dx = parent.keep_dx
            alloc('DX', (L if self.keep_dx else 1, N, CH))
            # q_l planes of the "push" formulation: what a tile's rows send to
            # the rows d earlier (csrc/wn_stack.hip)
            alloc('DQ', (L, N, CH))
            alloc('wimg_b', (L, lib.wn_stack_wimg_floats()))
            alloc('stack_flags_b', (lib.wn_stack_flag_count(B, T, L),),
                  torch.int32, fill=0)
            alloc('stack_ctl_b', (4,), torch.int32, fill=0)
            # a child whose parent was built without the backward stack
            # buffers owns fresh flags (all 0): its epoch must start at 1 too
            if 'stack_ctl_b' in fresh:
                self.stack_ctl_b[2] = 1
        if net.blocked:                  # channel-block path scratch
            alloc('dzb', (CB, N, CH))
            alloc('wdT', (CHn, CHn))
            alloc('blk_tmp', (max((2 * net.KW + 1) * 1024 + 96, Q * CH,
                                  net.initial_filter_width * CH),))
            alloc('cs_tmp', (B, 64))
        alloc('w2t', (Q, S))
        alloc('w1t', (S, S))
        alloc('wst', (S, L * CHn))
        ntiles = B * ((T + 31) // 32)
        self.nslab = max(1, min(512, ntiles // 4))
        if net.blocked:
            # the pair-slab scratch of the channel-block path grows with the
            # SQUARE of the block count (CB^2 x nslab x up to 17.5 K floats:
            # 36 GB at 1024 channels and 8 taps with 512 slabs): fewer, longer
            # row splits beyond 8 GB instead of an opaque allocation failure
            per = CB * CB * ((2 * min(net.KW, 8) + 1) * 1024 + 96) * 4
            self.nslab = max(1, min(self.nslab, (8 << 30) // per))
        self.nslab_2 = lib.wn_layer_bwd2_slabs(B, T)
        self.nslab_s = lib.wn_stack_bwd_slabs(B, T, self.stack_variant) \
            if self.stack_bwd else 0
        alloc('wimg', (L, lib.wn_layer_bwd2_wimg_floats()))
        if net.blocked:
            # channel-block path: one slab region per (input, output) block
            # pair of ONE layer (wavenet/blocked.py); the per-layer slabs of the
            # 32-channel kernels are not used
            alloc('pslabs', (CB * CB, self.nslab,
                             (2 * min(net.KW, 8) + 1) * 1024 + 96))
            alloc('lslabs', (1, 1, 4))
        else:
            alloc('lslabs', (L, max(self.nslab, self.nslab_2,
                                    self.nslab_s), net.LAYER_BLOCK))
        need = 0
        self.splits = {}
        for key, (mw, nw) in dict(post2=(S, Q), post1=(S, S), skip=(L * CHn, S),
                                  causal=(Q, CH)).items():
            sp = lib.wn_gemm_tn_splits(N, mw, nw, 1 if key == 'causal' else 0)
            self.splits[key] = sp
            need = max(need, sp * lib.wn_gemm_tn_slab_floats(mw, nw))
        need_tn = need
        # scalar-input causal wgrad: [splits][initial_filter_width][32] slabs
        need = max(need, max(256, self.splits['causal'])
                   * max(32, net.initial_filter_width) * CH)
        self.nslab_c = lib.wn_causal_wgrad_slabs(N)
        need = max(need, self.nslab_c * 2 * Q * CH)
        alloc('slabs', (need,))
        alloc('slabs_tn', (need_tn,))     # TN GEMMs on the side stream
        self.ev_fork = torch.cuda.Event() if dev.type == 'cuda' else None
        self.ev_join = torch.cuda.Event() if dev.type == 'cuda' else None
        self.dsum = alloc('dsum', (L, B, 2 * CHn)) if net.G else None
        self.gc_part = alloc('gc_part', (L, B, net.G)) if net.G else None
        # per-tile column sums of da: [L][tiles][64] for 32-row tiles (the
        # per-layer kernels, wn_stack_bwd on big batches) or 16-row tiles
        # (wn_stack_bwd on small ones, wn_stack_tile_rows): two views of one buffer
        self.stack_rows = lib.wn_stack_tile_rows(B, T, self.stack_variant)
        nt16 = B * ((T + 15) // 16)
        if net.G:
            buf = alloc('tilesum_buf', (L * nt16 * 64,))
            self.tilesum = buf[:L * ntiles * 64].view(L, ntiles, 64)
            self.tilesum16 = buf.view(L, nt16, 64)
        else:
            self.tilesum = self.tilesum16 = None
        self.dsum_part = alloc(
            'dsum_part', (B * lib.wn_colsum_clip_chunks(T) * 64,)) \
            if net.G else None
        alloc('l2_parts', (lib.wn_l2_partials_count(),))
        alloc('l2', (1,), fill=0.0)


class WaveNetModel(object):
    '''Implements the WaveNet network for generative audio (MI355X / HIP).

    Usage mirrors the reference (model.py:31-44):
        net = WaveNetModel(batch_size, dilations, filter_width,
                           residual_channels, dilation_channels, skip_channels)
        loss = net.loss(input_batch)          # forward + backward on the GPU
        optimizer.minimize(loss)              # fused TF-rule update
    '''

    # variant word of the stack launches a new model starts with (0 = the
    # library's choice per shape; tests / tools set it, see `stack_variant`)
    DEFAULT_STACK_VARIANT = 0

    def __init__(self,
                 batch_size,
                 dilations,
                 filter_width,
                 residual_channels,
                 dilation_channels,
                 skip_channels,
                 quantization_channels=2**8,
                 use_biases=False,
                 scalar_input=False,
                 initial_filter_width=32,
                 histograms=False,
                 global_condition_channels=None,
                 global_condition_cardinality=None,
                 residual_postproc=False,
                 device=None,
                 seed=0):
        self.batch_size = batch_size
        self.dilations = list(dilations)
        self.filter_width = filter_width
        self.residual_channels = residual_channels
        self.dilation_channels = dilation_channels
        self.quantization_channels = quantization_channels
        self.use_biases = use_biases
        self.skip_channels = skip_channels
        self.scalar_input = scalar_input
        self.initial_filter_width = initial_filter_width
        self.histograms = histograms
        self.global_condition_channels = global_condition_channels
        self.global_condition_cardinality = global_condition_cardinality
        self.residual_postproc = residual_postproc
        # TF's fused softmax-xent back-propagates softmax/(B*T) through the
        # all-zero-label last row of every clip [inferred].
        self.tf_xent_zero_label_quirk = True
        # model.py:28 passes the bias *name* as `trainable`, so the reference's
        # L2 filter "'bias' in v.name" (model.py:676) does not exclude biases.
        self.tf_bias_name_quirk = True
        # Run the three weight-gradient (TN) GEMMs of the skip / post-processing
        # convs on a second, lower-priority HIP stream next to the dZ GEMM and
        # the residual-stack backward (one fork after the dtotal GEMM, one join
        # before the slab reductions): the layer kernels keep the matrix pipes
        # about a third busy, the TN GEMMs are MFMA-bound.  None (default):
        # on for small batches only -- at most four 32-row tiles per CU, where
        # the backward stack is one dependent chain per tile that leaves most
        # of the chip idle (B = 1, T = 16000: 2.03 -> 1.88 ms per step; B = 2:
        # 2.94 -> 2.89; neutral at B = 4, 1 % SLOWER at B = 8).  True / False
        # force it.
        self.overlap_tn = None
        self.overlap_tn_split_frac = 0.6
        # channel-block models (33 - 128 channels): dz and the gate gradients
        # in one launch (False: two launches, A/B and tests; bitwise equal)
        self.wide_fuse_gate = True
        # column sums (bias gradients) of the weight-gradient GEMMs spread over
        # all tile rows of a split (False: one owner tile row, A/B)
        self.tn_spread_colsum = True
        # forward of the residual stack as ONE persistent launch
        # (wn_stack_fwd: tiles stay in registers from layer to layer, the
        # dilated taps are handed over through per-tile flags) instead of one
        # wn_layer_fwd launch per layer.  False selects the latter.
        self.stack_fwd = True
        # ... with the skip sum inside that launch where the library has it
        # (wn_stack_fwd_skip: small batches, 512 skip channels)
        self.stack_fwd_skip = True
        # the same for the backward of the stack (wn_stack_bwd instead of one
        # wn_layer_bwd2 per layer); read when a workspace is created
        self.stack_bwd = True
        # diagnostic: keep dL/dx_l of every layer instead of one plane
        # rewritten in place (read when a workspace is created)
        self.stack_bwd_keep_dx = False
        # explicit variant word of the stack launches (_lib.stack_variant:
        # tile rows, waves per workgroup, split backward), 0 = the library's
        # choice for the shape; for A/B runs and tests, read when a workspace
        # is created (setting it drops the resident workspaces).  (The library
        # reads no process environment.)
        self._stack_variant = int(self.DEFAULT_STACK_VARIANT)
        # data-parallel runs (wavenet/parallel.py): start the all-reduce of the
        # skip / post-processing gradients -- 82 % of the bucket, complete
        # before the backward stack launch -- from inside the backward pass,
        # on a communication stream beside that launch.  The training loop
        # switches it on: it promises that optimizer.minimize (which joins)
        # follows every loss().  Ignored outside torch.distributed and when L2
        # regularisation is on.
        self.dp_overlap_allreduce = False
        self._tail_work = None
        self._early_ok = False
        # generate(): four kernels per sample over many CUs, replayed from a
        # hipGraph, instead of the single-workgroup persistent kernel
        self.fastgen_multi_cu = True
        self.fastgen_graph_steps = 200
        # the multi-CU path as ONE persistent launch per generate() call
        # (wn_fastgen_persist: chain segments / skip / post-processing / draw
        # workgroups resident for the whole run, weights resident in LDS,
        # in-launch hand-overs instead of four kernel boundaries per sample);
        # False: the step kernels replayed from a hipGraph
        self.fastgen_persistent = True
        # persistent / cooperative generation launches that expired once on this
        # device (CUs held elsewhere): not tried again by any generator of this
        # model; `reset_generator_launch_failures()` clears it
        self._gen_launch_failed = {}
        # 64-channel models: wn_fastgen_run_wide's cooperative launch
        self.fastgen_wide_coop = True
        # 'fp32' (default): fp32 MFMA GEMMs.  'bf16x6' / 'bf16x9' / 'bf16x3':
        # opt-in split-bf16 products for the six NN GEMMs (wn_gemm_nn_split;
        # x6 measures the same error vs float64 as the fp32 MFMA path)
        self.gemm_mode = 'fp32'
        self._wsplit = {}
        # replay recorded (function, args) launch sequences instead of
        # re-deriving ~235 argument lists per step in Python
        self.use_launch_plans = True
        # causal-layer weight gradient as a segmented sum (K = 2, Q <= 256)
        # instead of two one-hot MFMA contractions
        self.causal_wgrad_segsum = True
        # seeds longer than this are primed from ONE batch forward pass
        # instead of one incremental step per seed sample
        self.fastgen_prime_forward_min = 64

        _lib.load()
        if device is None:
            _lib.require_gpu()
            self.device = torch.device('cuda', torch.cuda.current_device())
        else:
            # device='cpu' is allowed for parameter bookkeeping only (variable
            # names / shapes / checkpoints); every compute entry point raises.
            self.device = torch.device(device)
        self.L, self.S, self.Q = len(self.dilations), skip_channels, \
            quantization_channels
        self.R, self.D = residual_channels, dilation_channels
        self.G = global_condition_channels
        self.card = global_condition_cardinality
        self._unsupported = None
        if filter_width < 2 or filter_width > 64:
            # (above 8 the layers run in groups of 8 taps, wavenet/blocked.py;
            # tested at 11 and 19)
            self._unsupported = 'filter_width must be in [2, 64] on the HIP path'
        elif max(self.R, self.D) > 1024:
            # channel-block kernels (wavenet/blocked.py): 32-wide blocks, in
            # chunks of 8 // filter_width blocks per kernel call; tested up to
            # 320 channels, capped at 32 blocks
            self._unsupported = ('at most 1024 residual / dilation channels on '
                                 'the HIP path')
        elif self.S % 4 or self.Q % 4:
            self._unsupported = 'skip/quantization channels must be multiples of 4'
        elif self.G is not None and self.card is None:
            self._unsupported = ('dense-vector global conditioning cannot run in '
                                 'the reference either (model.py:547,553)')
        self.KW = K = int(filter_width)
        # more than 32 residual / dilation channels: 32-wide channel blocks,
        # one activation plane per block (wavenet/blocked.py)
        self.CB = max(1, (max(self.R, self.D) + CH - 1) // CH)
        self.CHn = C = CH * self.CB
        # K = 2 runs the tuned kernels; other widths (or forcing it, for
        # tests) the generic-tap kernels
        self.generic_layers = K != 2
        # channel-block kernels (wavenet/blocked.py): more than 32 channels, or
        # a filter wider than the generic-tap kernels' 8 taps
        self.blocked = self.CB > 1 or K > 8
        self.LAYER_W = layer_w(K, C)
        self.OFF_BF = self.LAYER_W
        self.OFF_BG = self.LAYER_W + C
        self.OFF_BD = self.LAYER_W + 2 * C
        self.LAYER_BLOCK = self.LAYER_W + 3 * C
        self.OFF_GC = self.LAYER_BLOCK
        self._ws = {}
        self._dil_dev = torch.tensor(self.dilations, dtype=torch.int32,
                                     device=self.device)
        self._gen = None
        self.init_ops = []
        self.push_ops = []
        self.variables = self._create_variables(seed)

    @property
    def stack_variant(self):
        return self._stack_variant

    @stack_variant.setter
    def stack_variant(self, v):
        if int(v) != self._stack_variant:
            self._stack_variant = int(v)
            self._ws = {}          # slab counts / tile sums depend on it

    # ------------------------------------------------------------------ params
    def _create_variables(self, seed):
        '''Creates all variables (model.py:118-225) as views into one flat
        buffer; same nesting, keys and [K, Cin, Cout] shapes as the reference.'''
        L, S, Q, R, D, G, card = (self.L, self.S, self.Q, self.R, self.D,
                                  self.G, self.card)
        if self._unsupported:
            self.params = torch.zeros(4, device=self.device)
            self.grads = torch.zeros(4, device=self.device)
            return {}
        C = self.CHn
        self.layer_stride = self.LAYER_BLOCK + (2 * G * C if G else 0)
        seg, off = {}, 0
        def add(name, n):
            nonlocal off
            seg[name] = (off, n)
            off = _align(off + n)
        if card is not None:
            add('emb', card * G)
        add('causal', (self.initial_filter_width if self.scalar_input
                       else self.KW * Q) * C)
        add('layers', L * self.layer_stride)
        add('skip_w', L * C * S)
        add('skip_b', L * S)
        add('post1_w', S * S)
        add('post2_w', S * Q)
        add('post1_b', S)
        add('post2_b', Q)
        self.segments = seg
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.variables = self._views(self.params)
        self.gradients = self._views(self.grads)
        self._init_variables(seed)
        return self.variables

    def _seg(self, flat, name):
        o, n = self.segments[name]
        return flat[o:o + n]

    def _views(self, flat):
        L, S, Q, R, D, G, card = (self.L, self.S, self.Q, self.R, self.D,
                                  self.G, self.card)
        var = dict()
        CH = self.CHn              # padded channel count of every weight
        if card is not None:
            var['embeddings'] = {
                'gc_embedding': self._seg(flat, 'emb').view(card, G)}
        if self.scalar_input:                               # model.py:143-153
            var['causal_layer'] = {'filter': self._seg(flat, 'causal').view(
                self.initial_filter_width, 1, CH)[:, :, :R]}
        else:
            var['causal_layer'] = {
                'filter': self._seg(flat, 'causal').view(self.KW, Q, CH)[
                    :, :, :R]}
        layers = self._seg(flat, 'layers').view(L, self.layer_stride)
        skw = self._seg(flat, 'skip_w').view(L, 1, CH, S)
        skb = self._seg(flat, 'skip_b').view(L, S)
        var['dilated_stack'] = []
        for i in range(L):
            blk = layers[i]
            cur = dict()
            K = self.KW
            OFF_GC, OFF_BF, OFF_BG, OFF_BD = (self.OFF_GC, self.OFF_BF,
                                              self.OFF_BG, self.OFF_BD)
            M = CH * CH
            cur['filter'] = blk[0:K * M].view(K, CH, CH)[:, :R, :D]
            cur['gate'] = blk[K * M:2 * K * M].view(K, CH, CH)[:, :R, :D]
            cur['dense'] = blk[2 * K * M:(2 * K + 1) * M].view(
                1, CH, CH)[:, :D, :R]
            cur['skip'] = skw[i][:, :D, :]
            if G is not None:
                gcf = blk[OFF_GC:OFF_GC + G * CH].view(1, G, CH)
                gcg = blk[OFF_GC + G * CH:OFF_GC + 2 * G * CH].view(1, G, CH)
                cur['gc_gateweights'] = gcg[:, :, :D]
                cur['gc_filtweights'] = gcf[:, :, :D]
            if self.use_biases:
                cur['filter_bias'] = blk[OFF_BF:OFF_BF + D]
                cur['gate_bias'] = blk[OFF_BG:OFF_BG + D]
                cur['dense_bias'] = blk[OFF_BD:OFF_BD + R]
                cur['skip_bias'] = skb[i]
            var['dilated_stack'].append(cur)
        post = dict()
        post['postprocess1'] = self._seg(flat, 'post1_w').view(1, S, S)
        post['postprocess2'] = self._seg(flat, 'post2_w').view(1, S, Q)
        if self.use_biases:
            post['postprocess1_bias'] = self._seg(flat, 'post1_b')
            post['postprocess2_bias'] = self._seg(flat, 'post2_b')
        var['postprocessing'] = post
        return var

    def _init_variables(self, seed):
        """Xavier / identity / zero initialisation (model.py:7-28) drawn on the
        host into a flat staging buffer of the same layout and uploaded ONCE
        (not one small copy per variable)."""
        gen = torch.Generator().manual_seed(int(seed))
        host = torch.zeros(self.params.numel(), dtype=torch.float32)
        v = self._views(host)
        with torch.no_grad():
            if self.card is not None:
                e = v['embeddings']['gc_embedding']
                if self.card == self.G:                      # model.py:16-19
                    e.copy_(torch.eye(self.card))
                else:
                    _xavier_(e, gen)
            _xavier_(v['causal_layer']['filter'], gen)
            for cur in v['dilated_stack']:
                for k in ['filter', 'gate', 'dense', 'skip', 'gc_gateweights',
                          'gc_filtweights']:
                    if k in cur:
                        _xavier_(cur[k], gen)
            _xavier_(v['postprocessing']['postprocess1'], gen)
            _xavier_(v['postprocessing']['postprocess2'], gen)
            # biases: zeros (model.py:27)
            self.params.copy_(host)

    def histogram_summaries(self, bins=30):
        """Counterpart of the `histograms=True` summaries of
        _create_dilation_layer (model.py:314-325): per layer, histograms of the
        filter / gate / dense (not for the last layer) / skip weights and,
        with biases, of the four bias vectors, under the reference's tags
        (`layer3_filter`, `layer3_biases_gate`, ...).  Returns
        {tag: (counts int64[bins], lo, hi)}; empty unless the model was built
        with histograms=True."""
        out = {}
        if not self.histograms:
            return out
        L = self.L
        tags = [('filter', '_filter'), ('gate', '_gate'), ('dense', '_dense'),
                ('skip', '_skip'), ('filter_bias', '_biases_filter'),
                ('gate_bias', '_biases_gate'), ('dense_bias', '_biases_dense'),
                ('skip_bias', '_biases_skip')]
        for i, cur in enumerate(self.variables['dilated_stack']):
            for k, suffix in tags:
                if k not in cur or (k == 'dense' and i == L - 1):
                    continue
                v = cur[k].detach().float().reshape(-1)
                lo, hi = float(v.min()), float(v.max())
                if hi <= lo:
                    hi = lo + 1e-12
                counts = torch.histc(v, bins=bins, min=lo, max=hi)
                out['layer%d%s' % (i, suffix)] = (
                    counts.to(torch.int64).cpu().numpy(), lo, hi)
        return out

    def named_variables(self, tree=None, prefix='wavenet'):
        """(reference variable name, view) pairs, creation order."""
        tree = self.variables if tree is None else tree
        out = []
        if 'embeddings' in tree:
            out.append((prefix + '/embeddings/gc_embedding',
                        tree['embeddings']['gc_embedding']))
        out.append((prefix + '/causal_layer/filter',
                    tree['causal_layer']['filter']))
        order = ['filter', 'gate', 'dense', 'skip', 'gc_gateweights',
                 'gc_filtweights', 'filter_bias', 'gate_bias', 'dense_bias',
                 'skip_bias']
        tfname = {'gc_gateweights': 'gc_gate', 'gc_filtweights': 'gc_filter',
                  'skip_bias': 'slip_bias'}       # sic, model.py:183-204
        for i, cur in enumerate(tree['dilated_stack']):
            for k in order:
                if k in cur:
                    out.append(('%s/dilated_stack/layer%d/%s'
                                % (prefix, i, tfname.get(k, k)), cur[k]))
        for k in ['postprocess1', 'postprocess2', 'postprocess1_bias',
                  'postprocess2_bias']:
            if k in tree['postprocessing']:
                out.append((prefix + '/postprocessing/' + k,
                            tree['postprocessing'][k]))
        return out

    def state_dict(self):
        return {n: v.detach().cpu().clone() for n, v in self.named_variables()}

    def load_state_dict(self, sd):
        with torch.no_grad():
            for n, v in self.named_variables():
                v.copy_(torch.as_tensor(sd[n]).to(self.device))
        self._gen = None

    def load_nested(self, tree):
        """Load a nested dict shaped like `variables` (numpy / tensors)."""
        def rec(dst, src):
            if isinstance(dst, dict):
                for k in dst:
                    rec(dst[k], src[k])
            elif isinstance(dst, list):
                for a, b in zip(dst, src):
                    rec(a, b)
            else:
                dst.copy_(torch.as_tensor(np.asarray(src),
                                          dtype=torch.float32).to(self.device))
        with torch.no_grad():
            rec(self.variables, tree)
        self._gen = None

    # ------------------------------------------------------------------ helpers
    def _overlap_tn_on(self, ws):
        if self.overlap_tn is not None:
            return bool(self.overlap_tn)
        return ws.B * ((ws.T + 31) // 32) <= 1024 and not self.blocked

    def _early_on(self):
        """The tail all-reduce starts inside this step's backward pass."""
        from . import parallel
        return bool(self.dp_overlap_allreduce and self._early_ok
                    and parallel.is_distributed())

    def _early_allreduce(self):
        from . import parallel
        _lib.call_py(lambda: parallel.begin_tail_allreduce(self))

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            # lower priority than the default stream: the residual-stack
            # kernels on the main stream are the critical path
            self._side = torch.cuda.Stream(device=self.device, priority=0)
        return self._side

    def _legacy_bwd(self):
        """The generic-tap (filter_width > 2) and channel-block (> 32 channels)
        models: their backward keeps the tanh plane and ping-pong
        pre-activation-gradient planes; the default-width two-tap model runs
        wn_stack_bwd / wn_layer_bwd2, which do not."""
        return self.generic_layers or self.blocked

    def check_device_errors(self):
        """Raise if a persistent stack launch recorded an expired dependency
        wait (its control word 3; the loss of that step is NaN by
        construction, see _Workspace.loss_parts).  Synchronises the device:
        meant for the moment a caller sees a non-finite loss."""
        for ws in self._ws.values():
            for name in ('stack_ctl', 'stack_ctl_b'):
                ctl = getattr(ws, name, None)
                if ctl is not None and int(ctl[3]) != 0:
                    raise _lib.WaveNetHipError(
                        '%s: a dependency wait inside the persistent residual-'
                        'stack launch expired (2 s); the results of that step '
                        'are invalid.  net.stack_fwd / net.stack_bwd = False '
                        'select the one-launch-per-layer kernels.' % name)

    def reset_generator_launch_failures(self):
        """Try the persistent / cooperative generation launches again."""
        self._gen_launch_failed = {}

    def reset_device_errors(self):
        """Clear the expired-wait record (control word 3 and the NaN poison
        words) of every workspace, e.g. after a caller has handled the error
        `check_device_errors` raised."""
        for ws in self._ws.values():
            for name in ('stack_ctl', 'stack_ctl_b'):
                ctl = getattr(ws, name, None)
                if ctl is not None:
                    ctl[3] = 0
            ws.loss_parts[:2] = 0.0

    def _bwd_image_with_fwd(self, ws):
        """True when the training forward runs the persistent stack launch and
        the backward will too: wn_stack_pack then writes both weight images in
        the forward's launch (the parameters do not change in between)."""
        return bool(self.stack_fwd and not self.blocked and not self.generic_layers
                    and self.L <= 256 and not self._legacy_bwd()
                    and self._stack_bwd_ok() and getattr(ws, 'stack_bwd', False)
                    and getattr(ws, 'wimg_b', None) is not None)

    def _stack_bwd_ok(self):
        """wn_stack_bwd covers what wn_layer_bwd2 covers."""
        return self.stack_bwd and not self._legacy_bwd() and self.L <= 256

    def _check_supported(self):
        if self._unsupported:
            raise NotImplementedError(self._unsupported)
        if self.device.type != 'cuda':
            raise _lib.WaveNetHipError(
                'WaveNetModel compute needs an MI355X (device=%s); there is '
                'no CPU fallback' % self.device)

    def _workspace(self, B, T, training):
        key = (B, T, bool(training))
        ws = self._ws.get(key)
        if ws is not None and training and self._legacy_bwd() and \
                not ws.legacy:
            self._ws = {}          # switched to a legacy backward: re-allocate
            ws = None
        if ws is not None:
            return ws
        # carve out of a resident larger workspace of the same batch size
        for (b, t, tr), cand in list(self._ws.items()):
            if cand.capacity == cand.N and b == B and B * T <= cand.capacity \
                    and (tr or not training):
                ws = _Workspace(self, B, T, training, parent=cand)
                break
        if ws is None:
            # grow geometrically (the naive generation path asks for T, T+1,
            # ... up to its window) and keep ONE owner per kind resident: a
            # training step and forward-only calls of another length do not
            # evict each other's buffers and launch plans
            prev = [w for (b, t, tr), w in self._ws.items()
                    if w.capacity == w.N and tr == bool(training) and b == B]
            t_alloc = T
            if prev and not training:
                t_alloc = max(T, min(2 * max(w.T for w in prev), 1 << 20))
            self._ws = {k: w for k, w in self._ws.items()
                        if self._owner_kind(w) != bool(training)}
            owner = _Workspace(self, B, t_alloc, training)
            self._ws[(B, t_alloc, bool(training))] = owner
            ws = owner if t_alloc == T else \
                _Workspace(self, B, T, training, parent=owner)
        self._ws[key] = ws
        if len(self._ws) > 64:        # views are cheap but unbounded otherwise
            owners = {k: v for k, v in self._ws.items()
                      if v.capacity == v.N}
            self._ws = owners
            self._ws[key] = ws
        return ws

    @staticmethod
    def _owner_kind(ws):
        return bool(ws.training)

    def reserve(self, batch_size, max_samples, training=False):
        """Allocate the workspace for up to `max_samples` samples per clip now
        (generate.py's window, train.py's sample_size), so that later calls
        with shorter inputs carve views out of it instead of allocating."""
        self._check_supported()
        return self._workspace(int(batch_size), int(max_samples), training)

    def _gc_ids(self, global_condition, B):
        if self.card is None or global_condition is None:
            if self.G is not None and self.card is not None:
                raise ValueError('model was built with global conditioning; '
                                 'a global_condition id batch is required')
            return None
        ids = global_condition
        if not isinstance(ids, torch.Tensor):
            ids = torch.as_tensor(np.asarray(ids))
        ids = ids.reshape(-1).to(device=self.device, dtype=torch.int32)
        if ids.numel() != B:
            raise ValueError('global_condition has %d ids, batch_size is %d'
                             % (ids.numel(), B))
        return ids.contiguous()

    def _layer_block(self, flat, l):
        o, _ = self.segments['layers']
        return flat[o + l * self.layer_stride: o + (l + 1) * self.layer_stride]

    def _bias_fg(self, ws_bias, ids, B):
        """Per-(layer, clip) filter|gate bias (+ GC 1x1 conv of the broadcast
        embedding, model.py:272-290).  Returns (tensor or None, clip stride)."""
        if not self.use_biases and ids is None:
            return None, 0
        nb = B if ids is not None else 1
        W2 = 2 * self.CHn                  # filter | gate, padded channels
        out = ws_bias.view(-1)[:self.L * nb * W2].view(self.L, nb, W2)
        emb = self._seg(self.params, 'emb') if ids is not None else None
        _lib.call('wn_gc_bias', _lib.ptr(self._layer_block(self.params, 0)),
                  self.layer_stride, self.OFF_BF, self.OFF_GC, self.G or 0,
                  _lib.ptr(emb), self.card or 0, _lib.ptr(ids), _lib.ptr(out),
                  self.L, nb, self.CHn, _lib.stream())
        return out, (W2 if ids is not None else 0)

    def _nn_seq(self, calls):
        """A sequence of row-wise dependent wn_gemm_nn calls (argument tuples
        without the stream), one launch each.  (Round 4 also ran them as ONE
        persistent launch with row-block dependencies inside: bitwise equal,
        worth at most 0.4 % of a B = 8 step and a loss at small batches;
        removed in round 5, DESIGN.md.)"""
        st = _lib.stream()
        for c in calls:
            self._nn(*(c + (st,)))

    def _nn(self, *args):
        """wn_gemm_nn (or, when `gemm_mode` asks for it, wn_gemm_nn_split),
        optionally bracketed by HIP events on the launch stream (bench.py's
        live roofline measurement)."""
        name = 'wn_gemm_nn'
        if self.gemm_mode != 'fp32':
            nprod = {'bf16x3': 3, 'bf16x6': 6, 'bf16x9': 9}[self.gemm_mode]
            M, N, K = args[-5], args[-4], args[-3]
            if K % 16 == 0:
                # one scratch buffer per WEIGHT (its address), not per shape:
                # equal-shaped GEMMs never share pieces
                key = (args[4], K, N)
                buf = self._wsplit.get(key)
                if buf is None:
                    nb = _lib.load().wn_gemm_split_w_bytes(K, N)
                    buf = torch.empty(nb // 4, dtype=torch.int32,
                                      device=self.device)
                    self._wsplit[key] = buf
                name = 'wn_gemm_nn_split'
                args = args[:-1] + (_lib.ptr(buf), nprod, args[-1])
        k = -7 if name == 'wn_gemm_nn_split' else -5
        _lib.call_timed(name, args, 2.0 * args[k] * args[k + 1] * args[k + 2],
                        getattr(self, '_gemm_events', None))

    # ------------------------------------------------------------ launch plans
    def _plan_key(self, tag, ws, ids, extra):
        return (tag, extra, ids is not None, self.generic_layers,
                self.overlap_tn,
                self.overlap_tn_split_frac, self.wide_fuse_gate, self.tn_spread_colsum, self.stack_fwd, self.stack_fwd_skip, self.stack_bwd, ws.stack_variant, self._early_on(), self.gemm_mode, self.causal_wgrad_segsum, self.tf_xent_zero_label_quirk,
                _lib.stream(), self.params.data_ptr(), self.grads.data_ptr())

    def _stage_ids(self, ws, ids):
        """GC ids into a workspace-owned buffer, so that recorded launch
        arguments never point at a caller's temporary."""
        if ids is None:
            return None
        if ids.data_ptr() != ws.gc_ids.data_ptr():
            ws.gc_ids.copy_(ids)
        return ws.gc_ids

    def _forward(self, ws, ids, save_ts):
        """Forward pass through a recorded launch plan (see _lib.record)."""
        ids = self._stage_ids(ws, ids)
        if not self.use_launch_plans:
            return self._forward_eager(ws, ids, save_ts)
        key = self._plan_key('fwd', ws, ids, save_ts)
        plan = ws.plans.get(key, 0)
        if plan == 0:                 # first use of this workspace: eager
            ws.plans[key] = None
            self._forward_eager(ws, ids, save_ts)
        elif plan is None:            # second use: record while executing
            with _lib.record() as rec:
                self._forward_eager(ws, ids, save_ts)
            ws.plans[key] = rec.plan
        else:
            _lib.replay(plan, getattr(self, '_gemm_events', None))

    def _backward(self, ws, ids):
        ids = self._stage_ids(ws, ids)
        if not self.use_launch_plans or self.blocked:
            # (the channel-block path, whose gradient-block copies are torch ops
            # a launch plan cannot replay)
            return self._backward_eager(ws, ids)
        key = self._plan_key('bwd', ws, ids, None)
        plan = ws.plans.get(key, 0)
        if plan == 0:
            ws.plans[key] = None
            self._backward_eager(ws, ids)
        elif plan is None:
            with _lib.record() as rec:
                self._backward_eager(ws, ids)
            ws.plans[key] = rec.plan
        else:
            _lib.replay(plan, getattr(self, '_gemm_events', None))

    # ------------------------------------------------------------------ forward
    def _forward_eager(self, ws, ids, save_ts):
        """_create_network (model.py:389-442) on codes ws.q -> ws.logits."""
        st = _lib.stream()
        B, T, N, L, S, Q = ws.B, ws.T, ws.N, self.L, self.S, self.Q
        P = self.params
        if self.scalar_input:
            wc = self._seg(P, 'causal')
            for cb in range(self.CB):          # one plane per channel block
                _lib.call('wn_scalar_causal_fwd', _lib.ptr(ws.audio),
                          _lib.ptr(wc[cb * CH:]), self.CHn, _lib.ptr(ws.X[cb]),
                          B, T, self.initial_filter_width, st)
        else:
            wc = self._seg(P, 'causal')
            for cb in range(self.CB):          # one plane per channel block
                _lib.call('wn_causal_gather', _lib.ptr(ws.q),
                          _lib.ptr(wc[cb * CH:]), _lib.ptr(ws.X[cb]), B, T, Q,
                          self.KW, self.CHn, st)
        bias, bstride = self._bias_fg(ws.bias_fg, ids, B)
        if self.blocked:
            from . import blocked
            blocked.forward_layers(self, ws, bias, bstride, bool(save_ts), st)
        stack = (self.stack_fwd and not self.blocked and not self.generic_layers
                 and save_ts in (0, 2) and L <= 256)
        fuse_skip = False
        if stack:
            # all L layers in one persistent launch (csrc/wn_stack.hip)
            # (its transposed weight images, one small launch per call)
            # (a training forward packs the backward stack's image in the
            # same launch: one launch per step instead of two)
            both = save_ts == 2 and self._bwd_image_with_fwd(ws)
            _lib.call('wn_stack_pack', _lib.ptr(self._layer_block(P, 0)),
                      self.layer_stride, _lib.ptr(ws.wimg_f),
                      _lib.ptr(ws.wimg_b) if both else None, L, st)
            stack_args = (_lib.ptr(ws.X), _lib.ptr(ws.Z),
                          _lib.ptr(ws.SG) if save_ts else None,
                          _lib.ptr(ws.wimg_f),
                          None if bias is None else _lib.ptr(bias),
                          0 if bias is None else bias.shape[1] * bias.shape[2],
                          bstride, _lib.ptr(self._dil_dev),
                          _lib.ptr(ws.stack_flags), _lib.ptr(ws.stack_ctl),
                          _lib.ptr(ws.loss_parts),
                          L, B, T, 1 if save_ts else 0, ws.stack_variant)
            # small batches: the skip sum h1 = relu(sum_l z_l Ws_l + sum_l bs_l)
            # inside the stack launch (wn_stack_fwd_skip: a partner wave per
            # tile; the launch's matrix pipe is three quarters idle otherwise)
            fuse_skip = bool(
                self.stack_fwd_skip and self.gemm_mode == 'fp32' and
                not self.residual_postproc and
                _lib.load().wn_stack_fwd_skip_ok(B, T, S, ws.stack_variant))
            if fuse_skip:
                if getattr(ws, 'skimg', None) is None:
                    ws.skimg = torch.empty(
                        int(_lib.load().wn_stack_skip_img_floats(L)),
                        dtype=torch.float32, device=self.device)
                bsum_f = None
                if self.use_biases:
                    _lib.call('wn_sum_rows', _lib.ptr(self._seg(P, 'skip_b')), L,
                              S, _lib.ptr(ws.bsum), st)
                    bsum_f = ws.bsum
                _lib.call('wn_stack_skip_pack', _lib.ptr(self._seg(P, 'skip_w')),
                          L, _lib.ptr(ws.skimg), st)
                _lib.call_timed('wn_stack_fwd_skip', stack_args + (
                    _lib.ptr(ws.skimg), _lib.ptr(bsum_f), _lib.ptr(ws.h1), st),
                    0.0, getattr(self, '_gemm_events', None))
            else:
                # (flops 0: timed in bench.py's instrumented pass for its HBM roofline)
                _lib.call_timed('wn_stack_fwd', stack_args + (st,), 0.0,
                                getattr(self, '_gemm_events', None))
        for l, d in enumerate(self.dilations if not self.blocked and not stack
                              else []):
            last = l == L - 1
            fargs = (_lib.ptr(ws.X[l]),
                     None if last else _lib.ptr(ws.X[l + 1]),
                     _lib.ptr(ws.Z[l]),
                     _lib.ptr(ws.TH[l]) if save_ts == 1 else None,
                     _lib.ptr(ws.SG[l]) if save_ts else None,
                     _lib.ptr(self._layer_block(P, l)),
                     None if bias is None else _lib.ptr(bias[l]), bstride,
                     B, T, int(d))
            if self.generic_layers:
                _lib.call('wn_layer_fwd_k', *fargs, self.KW,
                          0 if last else 1, 1 if save_ts else 0, st)
            else:
                _lib.call('wn_layer_fwd', *fargs, 0 if last else 1,
                          int(save_ts), st)
        bsum = None
        if self.use_biases and not fuse_skip:
            _lib.call('wn_sum_rows', _lib.ptr(self._seg(P, 'skip_b')), L, S,
                      _lib.ptr(ws.bsum), st)
            bsum = ws.bsum
        # total = sum_l z_l * Ws_l (+ sum_l bs_l); h1 = relu(total)
        LP, C = L * self.CB, self.CHn      # planes, padded channels
        b1 = self._seg(P, 'post1_b') if self.use_biases else None
        b2 = self._seg(P, 'post2_b') if self.use_biases else None
        rp = self.residual_postproc
        skip_gemm = [] if fuse_skip else [
            (_lib.ptr(ws.Z), 0, LP, N * CH,
             _lib.ptr(self._seg(P, 'skip_w')), S, _lib.ptr(bsum), None, 0,
             None, 0, _lib.ptr(ws.h1), S, 0, 0,
             _lib.ptr(ws.total) if self.residual_postproc else None,
             N, S, L * C, 1)]
        self._nn_seq(skip_gemm + [
            (_lib.ptr(ws.h1), S, 0, 0,
             _lib.ptr(self._seg(P, 'post1_w')), S, _lib.ptr(b1), None, 0,
             _lib.ptr(ws.total) if rp else None, S, _lib.ptr(ws.h2), S, 0,
             0, _lib.ptr(ws.c1) if (rp and ws.training) else None,
             N, S, S, 1),
            (_lib.ptr(ws.h2), S, 0, 0,
             _lib.ptr(self._seg(P, 'post2_w')), Q, _lib.ptr(b2), None, 0,
             None, 0, _lib.ptr(ws.logits), Q, 0, 0, None, N, Q, S, 0)])

    # ------------------------------------------------------------------ backward
    def _backward_eager(self, ws, ids):
        """Hand-written gradient of loss() (the reference uses TF autodiff of
        model.py:628-685).  Consumes ws.logits == dlogits (in place)."""
        st = _lib.stream()
        B, T, N, L, S, Q = ws.B, ws.T, ws.N, self.L, self.S, self.Q
        P, Gr = self.params, self.grads
        ub = 1 if self.use_biases else 0
        rp = self.residual_postproc
        dlog = ws.logits
        lib = _lib.load()

        deferred = []          # TN GEMMs postponed to the side stream
        ovl = self._overlap_tn_on(ws)

        def tn(*a, **kw):
            if ovl:
                deferred.append((a, kw))
            else:
                tn_now(st, ws.slabs, *a, **kw)

        def tn_now(st, slabs, A, lda, a_planes, a_pstride, codes, shift, Gm,
                   ldg, key, mw, nw, dst, dst_bias, replicate=1, rep_stride=0):
            sp = ws.splits[key]
            if ovl and 256 < ws.B * ((ws.T + 31) // 32) <= 512:
                # beside the backward stack of a very small batch: fewer, longer
                # workgroups disturb the stack's dependent chain less
                # (B = 1, T = 16000 -- 500 tiles: 1.89 -> 1.80 ms per step at
                # 0.6; 0.75 and 0.4 lose, and at B = 2 nothing changes; shapes
                # of at most one tile per CU are too short for it to matter)
                sp = max(1, int(sp * self.overlap_tn_split_frac))
            sl = lib.wn_gemm_tn_slab_floats(mw, nw)
            # the slabs' matrix and column sums go through ONE reduction launch
            # when the shapes allow; then the column sums are "spread" too:
            # every tile row of a split sums its share (wn_gemm_tn,
            # want_colsum = 2)
            mt = bool(ub and dst_bias is not None and (mw * nw) % 4 == 0 and
                      nw % 4 == 0 and sl % 4 == 0 and rep_stride % 4 == 0)
            tr = 1
            if mt and self.gemm_mode == 'fp32' and codes is None and \
                    self.tn_spread_colsum:
                tr = int(lib.wn_gemm_tn_tail_rows(mw, nw))
            if self.gemm_mode != 'fp32' and codes is None and N % 16 == 0:
                # opt-in split-bf16 products (fewer, larger splits)
                sp = min(sp, lib.wn_gemm_tn_splits(N, mw, nw, 2))
                _lib.call('wn_gemm_tn_split', A, lda, a_planes, a_pstride, Gm,
                          ldg, _lib.ptr(slabs), sp, N, mw, nw, ub,
                          int(self.gemm_mode[-1]), st)
            else:
                _lib.call_timed('wn_gemm_tn',
                                (A, lda, a_planes, a_pstride, codes, shift, T,
                                 Gm, ldg, _lib.ptr(slabs), sp, N, mw, nw,
                                 2 if tr > 1 else ub,
                                 st), 2.0 * N * mw * nw,
                                getattr(self, '_gemm_events', None))
            if mt:
                # matrix and column sums (bias gradient) in one launch
                _lib.call('wn_reduce_slabs_mt', _lib.ptr(slabs), sp, sl,
                          mw * nw, dst, nw, dst_bias, replicate, rep_stride,
                          tr, st)
                return
            _lib.call('wn_reduce_slabs', _lib.ptr(slabs), sp, sl, 1, 0, 0,
                      mw * nw, dst, 0, 1, 0, st)
            if ub and dst_bias is not None:
                _lib.call('wn_reduce_slabs', _lib.ptr(slabs), sp, sl, 1, 0,
                          mw * nw, nw, dst_bias, 0, replicate, rep_stride, st)

        # The data gradients first -- dc1 = (dlogits W2^T) * [c1 > 0],
        # dtotal = (dc1 W1^T) * [total > 0] (+ dh2 when residual_postproc),
        # dZ planes = dtotal Ws_all^T -- as ONE chained launch (a 128-row block
        # of a GEMM starts when that row block of the previous one is stored),
        # then the three weight-gradient (TN) GEMMs, whose operands all exist
        # by then: dW2 = h2^T dlogits, dW1 = h1^T dc1, dWs_all = Z^T dtotal
        # (+ column sums = the bias gradients).
        LP, C = L * self.CB, self.CHn      # planes, padded channels
        _lib.call('wn_transpose', _lib.ptr(self._seg(P, 'post2_w')), S, Q, Q,
                  _lib.ptr(ws.w2t), S, st)
        _lib.call('wn_transpose', _lib.ptr(self._seg(P, 'post1_w')), S, S, S,
                  _lib.ptr(ws.w1t), S, st)
        _lib.call('wn_transpose', _lib.ptr(self._seg(P, 'skip_w')), L * C, S,
                  S, _lib.ptr(ws.wst), L * C, st)
        nn_dc1 = (_lib.ptr(dlog), Q, 0, 0, _lib.ptr(ws.w2t), S,
                  None, _lib.ptr(ws.c1 if rp else ws.h2), S, None, 0,
                  _lib.ptr(ws.dc1), S, 0, 0, _lib.ptr(ws.dh2) if rp else None,
                  N, S, Q, 0)
        nn_dtotal = (_lib.ptr(ws.dc1), S, 0, 0, _lib.ptr(ws.w1t), S,
                     None, _lib.ptr(ws.h1), S, _lib.ptr(ws.dh2) if rp else None,
                     S, _lib.ptr(ws.dtotal), S, 0, 0, None, N, S, S, 0)
        nn_dz = (_lib.ptr(ws.dtotal), S, 0, 0, _lib.ptr(ws.wst),
                 L * C, None, None, 0, None, 0, _lib.ptr(ws.dZ), 0, LP,
                 N * CH, None, N, L * C, S, 0)
        # (small batches: the TN GEMMs run on a side stream beside the dZ GEMM
        # and the backward stack, so the dZ GEMM stays a launch of its own
        # behind the fork)
        self._nn_seq([nn_dc1, nn_dtotal] if ovl else [nn_dc1, nn_dtotal, nn_dz])
        tn(_lib.ptr(ws.h2), S, 0, 0, None, 0, _lib.ptr(dlog), Q, 'post2', S, Q,
           _lib.ptr(self._seg(Gr, 'post2_w')),
           _lib.ptr(self._seg(Gr, 'post2_b')))
        tn(_lib.ptr(ws.h1), S, 0, 0, None, 0, _lib.ptr(ws.dc1), S, 'post1', S,
           S, _lib.ptr(self._seg(Gr, 'post1_w')),
           _lib.ptr(self._seg(Gr, 'post1_b')))
        # skip convs: dbs_l = colsum(dtotal) for every l
        tn(_lib.ptr(ws.Z), 0, LP, N * CH, None, 0, _lib.ptr(ws.dtotal), S,
           'skip', L * C, S, _lib.ptr(self._seg(Gr, 'skip_w')),
           _lib.ptr(self._seg(Gr, 'skip_b')), replicate=L, rep_stride=S)
        if ovl:
            # fork: everything the three TN GEMMs read exists now
            # (forking behind the dZ GEMM instead, or another order of the
            # three, changes nothing at B = 1: 1.80 ms either way)
            main_s = torch.cuda.current_stream()
            side_s = self._side_stream()
            _lib.call_py(lambda: (ws.ev_fork.record(main_s),
                                  side_s.wait_event(ws.ev_fork)))
            for a, kw in deferred:
                tn_now(side_s.cuda_stream, ws.slabs_tn, *a, **kw)
            _lib.call_py(lambda: ws.ev_join.record(side_s))
            self._nn(*(nn_dz + (st,)))
        elif self._early_on():
            # skip / post-processing gradients are complete on this stream:
            # their all-reduce runs beside the backward stack
            self._early_allreduce()
        if self.blocked:
            # channel-block path: residual stack, causal layer and global
            # conditioning gradients (wavenet/blocked.py)
            from . import blocked
            if ovl:
                main_s = torch.cuda.current_stream()
                _lib.call_py(lambda: main_s.wait_event(ws.ev_join))
            blocked.backward_layers(self, ws, ids, st)
            return

        # residual stack, last layer first
        if not self._legacy_bwd():
            # one launch per layer; the launches are chained through dx only
            dxin, xp = None, 0
            tsum = None if ws.dsum is None else ws.tilesum
            if self._stack_bwd_ok() and getattr(ws, 'stack_bwd', False):
                if ws.stack_rows == 16 and tsum is not None:
                    tsum = ws.tilesum16
                # all L layers in one persistent launch (csrc/wn_stack.hip)
                if not self._bwd_image_with_fwd(ws):
                    _lib.call('wn_stack_pack', _lib.ptr(self._layer_block(P, 0)),
                              self.layer_stride, None, _lib.ptr(ws.wimg_b), L, st)
                _lib.call_timed('wn_stack_bwd', (
                    _lib.ptr(ws.X), _lib.ptr(ws.Z),
                    _lib.ptr(ws.SG), _lib.ptr(ws.dZ), _lib.ptr(ws.DX),
                    ws.N * CH if ws.keep_dx else 0,
                    _lib.ptr(ws.DQ), _lib.ptr(ws.wimg_b), _lib.ptr(ws.lslabs),
                    ws.lslabs.shape[1] * self.LAYER_BLOCK,
                    None if tsum is None else _lib.ptr(tsum),
                    _lib.ptr(self._dil_dev), _lib.ptr(ws.stack_flags_b),
                    _lib.ptr(ws.stack_ctl_b),
                    _lib.ptr(ws.loss_parts[1:]), L, B, T, ws.stack_variant, st), 0.0,
                    getattr(self, '_gemm_events', None))
                self._backward_tail(ws, ids, ws.DX[0], ws.nslab_s, True,
                                    tile_rows=ws.stack_rows)
                return
            # transposed weight images of all layers (the kernels DMA them
            # into LDS): one small launch per step
            _lib.call('wn_layer_bwd2_pack', _lib.ptr(self._layer_block(P, 0)),
                      self.layer_stride, _lib.ptr(ws.wimg), L, st)
            for l in range(L - 1, -1, -1):
                dxo = ws.dx[xp]
                _lib.call('wn_layer_bwd2', _lib.ptr(ws.X[l]), _lib.ptr(ws.Z[l]),
                          _lib.ptr(ws.SG[l]), _lib.ptr(ws.dZ[l]),
                          _lib.ptr(dxin), _lib.ptr(dxo),
                          _lib.ptr(self._layer_block(P, l)),
                          _lib.ptr(ws.wimg[l]), _lib.ptr(ws.lslabs[l]),
                          None if tsum is None else _lib.ptr(tsum[l]),
                          B, T, int(self.dilations[l]), st)
                dxin, xp = dxo, 1 - xp
            self._backward_tail(ws, ids, dxin, ws.nslab_2, True)
            return

        # generic filter width (wn_layer_*_k; also a K = 2 model with
        # `generic_layers` forced, tests): phase A of layer l - 1 and phase B
        # of layer l per launch, pre-activation gradients through two ping-pong
        # plane pairs, weight gradients per layer into slabs
        def da(p):
            return ws.da[p][0], ws.da[p][1]

        def layer_bwd(*a):           # (..., B, T, d, do_b, do_a, stream)
            _lib.call('wn_layer_bwd_k', *a[:14], self.KW, *a[14:16], 1, 0, a[16])

        def layer_wgrad(*a):         # (..., nslab, B, T, d, stream)
            _lib.call('wn_layer_wgrad_k', *a[:10], self.KW, 0, self.KW, 1, 0, a[10])
        cur = 0
        f, g = da(cur)
        # phase A of the last layer (no gradient flows into its x' output)
        layer_bwd(None, None, None, None, None,
                  _lib.ptr(ws.dZ[L - 1]), _lib.ptr(ws.TH[L - 1]),
                  _lib.ptr(ws.SG[L - 1]),
                  _lib.ptr(self._layer_block(P, L - 1)), _lib.ptr(f),
                  _lib.ptr(g), B, T, 1, 0, 1, st)
        dxin = None           # dL/dx' of layer l (None for the last layer)
        xp = 0
        nslab = ws.nslab
        for l in range(L - 1, -1, -1):
            d = int(self.dilations[l])
            f, g = da(cur)
            dxo = ws.dx[xp]
            layer_wgrad(_lib.ptr(ws.X[l]), _lib.ptr(f), _lib.ptr(g),
                        None if dxin is None else _lib.ptr(ws.Z[l]),
                        None if dxin is None else _lib.ptr(dxin),
                        _lib.ptr(ws.lslabs[l]), ws.nslab, B, T, d, st)
            if ws.dsum is not None:
                _lib.call('wn_colsum_clip', _lib.ptr(f), _lib.ptr(g), B, T,
                          _lib.ptr(ws.dsum_part), _lib.ptr(ws.dsum[l]), st)
            if l > 0:
                fn, gn = da(1 - cur)
                layer_bwd(_lib.ptr(f), _lib.ptr(g),
                          None if dxin is None else _lib.ptr(dxin),
                          _lib.ptr(dxo), _lib.ptr(self._layer_block(P, l)),
                          _lib.ptr(ws.dZ[l - 1]), _lib.ptr(ws.TH[l - 1]),
                          _lib.ptr(ws.SG[l - 1]),
                          _lib.ptr(self._layer_block(P, l - 1)), _lib.ptr(fn),
                          _lib.ptr(gn), B, T, d, 1, 1, st)
                cur = 1 - cur
            else:
                layer_bwd(_lib.ptr(f), _lib.ptr(g),
                          None if dxin is None else _lib.ptr(dxin),
                          _lib.ptr(dxo), _lib.ptr(self._layer_block(P, l)),
                          None, None, None, None, None, None, B, T, d, 1, 0,
                          st)
            dxin = dxo
            xp = 1 - xp
        self._backward_tail(ws, ids, dxin, nslab, False)

    def _backward_tail(self, ws, ids, dxin, nslab, fused, tile_rows=32):
        """After the residual stack: slab reductions of the layer-block
        gradients, causal-layer and global-conditioning gradients."""
        if self._overlap_tn_on(ws):
            main_s = torch.cuda.current_stream()
            _lib.call_py(lambda: main_s.wait_event(ws.ev_join))     # join
            if self._early_on():
                # (small batches: the side stream's weight-gradient GEMMs have
                # just joined; the tail's all-reduce runs beside the slab
                # reductions and the causal / conditioning gradients)
                self._early_allreduce()
        st = _lib.stream()
        B, T, N, L, S, Q = ws.B, ws.T, ws.N, self.L, self.S, self.Q
        P, Gr = self.params, self.grads
        ub = 1 if self.use_biases else 0
        lib = _lib.load()
        if fused and ws.dsum is not None:
            # per-clip sums of da_l for every layer from the per-tile sums the
            # fused kernel wrote (fixed order over a clip's tiles)
            tpc = (T + tile_rows - 1) // tile_rows
            _lib.call('wn_reduce_slabs', _lib.ptr(ws.tilesum), tpc, 64, L * B,
                      tpc * 64, 0, 64, _lib.ptr(ws.dsum), 64, 1, 0, st)
        # layer-block gradients: fixed-order sum of the per-workgroup slabs
        lo, _ = self.segments['layers']
        _lib.call('wn_reduce_slabs', _lib.ptr(ws.lslabs), nslab,
                  self.LAYER_BLOCK, L, ws.lslabs.shape[1] * self.LAYER_BLOCK,
                  0, self.LAYER_BLOCK if ub else self.LAYER_W,
                  _lib.ptr(Gr[lo:]),
                  self.layer_stride, 1, 0, st)
        # causal layer: dWc[1][v] = sum_t [q[t]==v] dx0[t]; dWc[0][v] likewise
        # with q[t-1]  (one-hot operand generated on the fly)
        gc_ = self._seg(Gr, 'causal')
        if self.scalar_input:
            K0 = self.initial_filter_width
            sp = ws.splits['causal']
            _lib.call('wn_scalar_causal_wgrad', _lib.ptr(ws.audio),
                      _lib.ptr(dxin), _lib.ptr(ws.slabs), sp, B, T, K0, st)
            _lib.call('wn_reduce_slabs', _lib.ptr(ws.slabs), sp, K0 * CH, 1,
                      0, 0, K0 * CH, _lib.ptr(gc_), 0, 1, 0, st)
        elif self.KW == 2 and Q <= 256 and self.causal_wgrad_segsum:
            # segmented sum over the codes (no one-hot contraction)
            ns = ws.nslab_c
            _lib.call('wn_causal_wgrad', _lib.ptr(ws.q), _lib.ptr(dxin),
                      _lib.ptr(ws.slabs), ns, B, T, Q, st)
            _lib.call('wn_reduce_slabs', _lib.ptr(ws.slabs), ns, 2 * Q * CH,
                      1, 0, 0, 2 * Q * CH, _lib.ptr(gc_), 0, 1, 0, st)
        else:
            K = self.KW
            for tap in range(K):
                shift = (K - 1 - tap) + (K - 1) // 2
                sp = ws.splits['causal']
                sl = lib.wn_gemm_tn_slab_floats(Q, CH)
                _lib.call('wn_gemm_tn', None, 0, 0, 0, _lib.ptr(ws.q), shift,
                          T, _lib.ptr(dxin), CH, _lib.ptr(ws.slabs), sp, N, Q,
                          CH, 0, st)
                _lib.call('wn_reduce_slabs', _lib.ptr(ws.slabs), sp, sl, 1, 0,
                          0, Q * CH, _lib.ptr(gc_[tap * Q * CH:]), 0, 1, 0, st)
        if ws.dsum is not None:
            _lib.call('wn_gc_grad', _lib.ptr(self._layer_block(P, 0)),
                      self.layer_stride, self.OFF_GC, self.G,
                      _lib.ptr(self._seg(P, 'emb')), self.card, _lib.ptr(ids),
                      _lib.ptr(ws.dsum), L, B,
                      _lib.ptr(self._layer_block(Gr, 0)),
                      _lib.ptr(self._seg(Gr, 'emb')), _lib.ptr(ws.gc_part),
                      self.CHn, st)

    # ------------------------------------------------------------------ API
    def encode(self, input_batch, B=None):
        """mu-law codes [B, T] of float audio (model.py:639-640)."""
        B = self.batch_size if B is None else B
        a = input_batch
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.asarray(a), dtype=torch.float32)
        a = a.to(device=self.device, dtype=torch.float32).reshape(B, -1)
        return mu_law_encode(a, self.Q)

    def loss(self,
             input_batch,
             global_condition_batch=None,
             l2_regularization_strength=None,
             name='wavenet',
             backward=True):
        '''Creates a WaveNet network and returns the autoencoding loss
        (model.py:628-685).  input_batch: float audio in [-1, 1], anything
        reshapeable to [batch_size, -1].  With backward=True (default) the
        gradient of the returned loss w.r.t. every variable is left in
        `self.grads` (the flat bucket `optimizer.minimize` consumes).'''
        self._check_supported()
        B = self.batch_size
        a = input_batch
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.asarray(a), dtype=torch.float32)
        a = a.to(device=self.device, dtype=torch.float32).reshape(B, -1)
        q = mu_law_encode(a, self.Q)
        return self.loss_from_codes(q, global_condition_batch,
                                    l2_regularization_strength, backward,
                                    audio=a)

    def loss_from_codes(self, q, global_condition_batch=None,
                        l2_regularization_strength=None, backward=True,
                        audio=None):
        self._check_supported()
        B = self.batch_size
        q = q.reshape(B, -1)
        T = q.shape[1]
        N = B * T
        ws = self._workspace(B, T, backward)
        if backward and self._tail_work is not None:
            # the previous backward pass started its tail all-reduce and no
            # optimizer.minimize joined it: join before this pass rewrites the
            # bucket (parallel.begin_tail_allreduce documents why this keeps
            # the ranks in step)
            import warnings
            from . import parallel
            warnings.warn('the tail all-reduce of the previous backward pass '
                          'was never joined (no optimizer.minimize followed): '
                          'joined and dropped now')
            parallel.abandon_tail_allreduce(self)
        ws.q.copy_(q.reshape(-1))
        if self.scalar_input:
            # network input is the raw float audio (model.py:645-648)
            if audio is None:
                raise ValueError('scalar_input needs the float audio')
            ws.audio.copy_(audio.reshape(-1))
        ids = self._gc_ids(global_condition_batch, B)
        st = _lib.stream()
        # 0: inference; 1: tanh + sigmoid planes (legacy backward kernels);
        # 2: sigmoid plane only (wn_layer_bwd2)
        self._forward(ws, ids, save_ts=0 if not backward else
                      (1 if self._legacy_bwd() else 2))
        _lib.call('wn_xent', _lib.ptr(ws.logits), self.Q, _lib.ptr(ws.q),
                  _lib.ptr(ws.logits) if backward else None,
                  _lib.ptr(ws.loss_parts[2:]), B, T, self.Q,
                  1 if self.tf_xent_zero_label_quirk else 0, st)
        _lib.call('wn_reduce_slabs', _lib.ptr(ws.loss_parts), ws.nparts + 2, 1, 1,
                  0, 0, 1, _lib.ptr(ws.loss), 0, 1, 0, st)
        loss = ws.loss[0] / float(N)                    # reduce_mean, :666
        if backward:
            # (L2 adds lambda * params to the WHOLE bucket after the backward
            # pass: the tail must not have been summed over ranks before that)
            self._early_ok = l2_regularization_strength is None
            try:
                self._backward(ws, ids)
            except BaseException:
                # (a launch error after the tail's all-reduce was issued: join
                # it, so that the next step does not find it dangling)
                from . import parallel
                parallel.abandon_tail_allreduce(self)
                raise
            # the backward stack launch's poison word (0, or NaN when one of
            # its dependency waits expired) is written after the loss
            # reduction above: add it here so that THIS step's loss is NaN
            # before the optimizer applies the step
            loss = loss + ws.loss_parts[1]
        if l2_regularization_strength is not None:
            lam = float(l2_regularization_strength)
            mask = None if self.tf_bias_name_quirk else self._l2_mask()
            _lib.call('wn_l2_partials', _lib.ptr(self.params),
                      self.params.numel(), _lib.ptr(mask),
                      _lib.ptr(ws.l2_parts) if backward else
                      _lib.ptr(self._l2_tmp()), st)
            parts = ws.l2_parts if backward else self._l2_tmp()
            loss = loss + lam * parts.sum()             # model.py:674-680
            if backward:
                _lib.call('wn_axpy', _lib.ptr(self.grads),
                          _lib.ptr(self.params), lam, _lib.ptr(mask),
                          self.params.numel(), st)
        loss._wn_model = self
        loss._wn_has_grads = bool(backward)
        return loss

    def _l2_tmp(self):
        if not hasattr(self, '_l2_parts'):
            self._l2_parts = torch.empty(_lib.load().wn_l2_partials_count(),
                                         dtype=torch.float32,
                                         device=self.device)
        return self._l2_parts

    def _l2_mask(self):
        """1 for weights, 0 for biases (only used when the TF bias-name quirk
        is switched off)."""
        if not hasattr(self, '_l2m'):
            m = torch.ones_like(self.params)
            tree = self._views(m)
            for n, v in self.named_variables(tree):
                if 'bias' in n.split('/')[-1]:
                    v.zero_()
            self._l2m = m
        return self._l2m

    def predict_proba(self, waveform, global_condition=None, name='wavenet'):
        '''Computes the probability distribution of the next sample based on
        all samples in the input waveform (model.py:564-590).  waveform:
        already-quantised int samples.'''
        self._check_supported()
        B = self.batch_size
        w = waveform
        if not isinstance(w, torch.Tensor):
            w = torch.as_tensor(np.asarray(w))
        w = w.to(device=self.device, dtype=torch.int32).reshape(B, -1)
        T = w.shape[1]
        ws = self._workspace(B, T, False)
        ws.q.copy_(w.reshape(-1))
        if self.scalar_input:
            # decode the codes back to floats in [-1, 1] (model.py:570-576)
            ws.audio.copy_(mu_law_decode(w, self.Q).reshape(-1))
        ids = self._gc_ids(global_condition, B)
        self._forward(ws, ids, save_ts=0)
        out = torch.empty(self.Q, dtype=torch.float32, device=self.device)
        _lib.call('wn_softmax64_row', _lib.ptr(ws.logits[B * T - 1]), self.Q,
                  _lib.ptr(out), _lib.stream())
        # 0, or NaN when a dependency wait of the forward stack launch
        # expired: wrong probabilities are never returned silently
        return out + ws.loss_parts[0]

    # ---------------------------------------------------------- fast generation
    FASTGEN_MAX_CHANNELS = 1024    # wn_fastgen_run_wide (FGW_MAXC)

    def _generator(self, global_condition):
        """Device-resident incremental-generation state (_create_generator,
        model.py:444-516): ring buffers standing in for the FIFO queues."""
        if self.CHn > self.FASTGEN_MAX_CHANNELS:
            raise NotImplementedError(
                'fast (incremental) generation supports at most %d residual / '
                'dilation channels on the HIP path; predict_proba (generate.py '
                'without --fast_generation) has no such limit'
                % self.FASTGEN_MAX_CHANNELS)
        if self._gen is None:
            lib = _lib.load()
            dil = np.asarray(self.dilations, dtype=np.int32)
            # (queue entries are rows of CHn = 32 * blocks floats)
            nfl = lib.wn_fastgen_state_floats(dil.ctypes.data, self.L) * self.CB
            g = dict(
                state=torch.zeros(nfl, dtype=torch.float32, device=self.device),
                # [0] steps done, [1] previous code, [2] draw pending
                cursors=torch.zeros(4, dtype=torch.int32, device=self.device),
                dil=torch.from_numpy(dil).to(self.device),
                bias=torch.zeros((self.L, 1, 2 * self.CHn),
                                 dtype=torch.float32, device=self.device),
                bsum=torch.zeros(self.S, dtype=torch.float32,
                                 device=self.device),
                proba=torch.empty(self.Q, dtype=torch.float32,
                                  device=self.device),
                z_all=torch.zeros(self.L * CH, dtype=torch.float32,
                                  device=self.device),
                cw_img=torch.zeros(self.L * 3072, dtype=torch.float32,
                                   device=self.device),
                pre=torch.zeros(self.L * 64, dtype=torch.float32,
                                device=self.device),
                ctl=torch.zeros(8, dtype=torch.int32, device=self.device),
                graphs={}, warm=False,
                h1=torch.zeros(self.S, dtype=torch.float32,
                               device=self.device),
                h2=torch.zeros(self.S, dtype=torch.float32,
                               device=self.device),
                logits=torch.zeros(self.Q, dtype=torch.float32,
                                   device=self.device),
                steps=0,
                io=torch.zeros(2, dtype=torch.int32, device=self.device))
            self._gen = g
            self._gen_reset()
        return self._gen

    def _gen_reset(self):
        g = self._gen
        _lib.call('wn_fastgen_init', _lib.ptr(g['state']), g['state'].numel(),
                  _lib.ptr(g['cursors']), self.L, _lib.stream())
        g['steps'] = 0

    def _gen_run(self, samples_io, n_given, n_steps, temperature, seed,
                 proba_out, proba_every, global_condition, push=True,
                 multi_cu=False):
        """Run `n_steps` generation steps.  multi_cu=False: ONE persistent
        single-workgroup kernel (also the push=False peek).  multi_cu=True:
        four kernels per step spread over many CUs, captured into a hipGraph
        and replayed (config 5 throughput path)."""
        # (wn_fastgen_step reads both from the device control block and cannot
        # reject them; the reference applies the temperature as log(p) / T,
        # generate.py:229-233)
        if not (np.isfinite(float(temperature)) and float(temperature) > 0.0):
            raise ValueError('temperature must be a finite number > 0, got %r'
                             % (temperature,))
        if int(n_given) < 1:
            raise ValueError('n_given must be >= 1, got %r' % (n_given,))
        g = self._generator(global_condition)
        ids = self._gc_ids(global_condition, 1) if self.card is not None \
            else None
        bias, _ = self._bias_fg(g['bias'], ids, 1)
        P = self.params
        bsum = None
        if self.use_biases:
            _lib.call('wn_sum_rows', _lib.ptr(self._seg(P, 'skip_b')), self.L,
                      self.S, _lib.ptr(g['bsum']), _lib.stream())
            bsum = g['bsum']
        ub = self.use_biases
        common = (_lib.ptr(self._seg(P, 'causal')),
                  _lib.ptr(self._layer_block(P, 0)), self.layer_stride,
                  _lib.ptr(self._seg(P, 'skip_w')), _lib.ptr(bsum),
                  _lib.ptr(self._seg(P, 'post1_w')),
                  _lib.ptr(self._seg(P, 'post1_b')) if ub else None,
                  _lib.ptr(self._seg(P, 'post2_w')),
                  _lib.ptr(self._seg(P, 'post2_b')) if ub else None,
                  None if bias is None else _lib.ptr(bias), _lib.ptr(g['dil']),
                  self.L, self.S, self.Q, _lib.ptr(g['state']),
                  _lib.ptr(g['cursors']), _lib.ptr(samples_io))
        sd = int(seed) & (2**64 - 1)
        if self.CB > 1 or self.S > 512 or self.Q > 512 or self.L > 64:
            # more than 32 channels, or more skip / quantization channels or
            # layers than the tuned kernels hold in LDS (FG_MAXS / FG_MAXQ /
            # FG_MAXL): the wide single-workgroup generator
            # 64 channels: the cooperative launch (skip sum and post-processing
            # on other CUs) when the library has one for the shape; it falls
            # back to the single workgroup by itself when the workgroups would
            # not all be resident
            coop = None
            if self.fastgen_wide_coop and not self._gen_launch_failed.get('coop'):
                if 'coop' not in g:
                    nb = _lib.load().wn_fastgen_wide_coop_bytes(
                        self.L, self.CHn, self.S, self.Q)
                    g['coop'] = torch.zeros(nb // 4, dtype=torch.int32,
                                            device=self.device) if nb else None
                coop = g['coop']

            def run(scratch):
                _lib.call('wn_fastgen_run_wide', *common[:11], self.L, self.CHn,
                          self.S, self.Q, *common[14:], int(n_given), int(n_steps),
                          float(temperature), sd, _lib.ptr(proba_out),
                          int(proba_every), 1 if ub else 0, 1 if push else 0,
                          _lib.ptr(scratch), _lib.stream())
            if coop is None:
                run(None)
            else:
                # What the library's residency check cannot see -- another
                # process or stream holding CUs, a CU mask -- shows as an
                # expired hand-over wait (word 12 of the scratch): the queues,
                # cursors and samples are then restored from a snapshot taken
                # here and the run is repeated by the single workgroup, which
                # always completes.
                snap = (g['state'].clone(), g['cursors'].clone(), samples_io.clone())
                run(coop)
                if int(coop[12]) != 0:     # (synchronises; the call ends on the host anyway)
                    import warnings
                    warnings.warn(
                        'wn_fastgen_run_wide: a hand-over wait inside the '
                        'cooperative generation launch expired (2 s: its '
                        'workgroups were not all resident); state restored, '
                        'continuing with the single workgroup '
                        '(net.fastgen_wide_coop = False selects it up front)')
                    g['state'].copy_(snap[0])
                    g['cursors'].copy_(snap[1])
                    samples_io.copy_(snap[2])
                    # (remembered on the MODEL: a new generator on the same busy
                    # device must not pay another expired wait)
                    self._gen_launch_failed['coop'] = True
                    run(None)
                del snap
            if push:
                g['steps'] += int(n_steps)
            return
        if not multi_cu or not push:
            _lib.call('wn_fastgen_run', *common, int(n_given), int(n_steps),
                      float(temperature), sd, _lib.ptr(proba_out),
                      int(proba_every), 1 if ub else 0, 1 if push else 0,
                      _lib.stream())
            if push:
                g['steps'] += int(n_steps)
            return
        base = g['steps']
        st = _lib.stream()
        # weights are constant while generating: pack the chain blocks once,
        # and compute the past-tap pre-activations of the first step (every
        # step then leaves the next step's behind)
        _lib.call('wn_fastgen_pack', _lib.ptr(self._layer_block(P, 0)),
                  self.layer_stride, _lib.ptr(g['cw_img']), self.L, st)
        _lib.call('wn_fastgen_pre', _lib.ptr(self._layer_block(P, 0)),
                  self.layer_stride, None if bias is None else _lib.ptr(bias),
                  _lib.ptr(g['dil']), self.L, _lib.ptr(g['state']),
                  _lib.ptr(g['cursors']), _lib.ptr(g['pre']), st)
        # per-call values live in device memory (ctl) and in two persistent
        # buffers (codes, probabilities), so a captured graph holds nothing
        # that changes between calls and is reused
        n_io = int(n_steps) + 1
        pe = max(1, int(proba_every))
        io = self._gen_buf('io_buf', n_io, torch.int32)
        io[:n_io].copy_(samples_io[:n_io])
        pb = None
        if proba_out is not None:
            rows = (int(n_steps) + pe - 1) // pe
            pb = self._gen_buf('proba_buf', rows * self.Q, torch.float32)
        ctl = np.zeros(8, np.uint32)
        ctl[0], ctl[1], ctl[2] = base, int(n_given), pe
        ctl[3] = np.float32(temperature).view(np.uint32)
        ctl[4], ctl[5] = sd & 0xffffffff, sd >> 32
        g['ctl'].copy_(torch.from_numpy(ctl.view(np.int32)))
        common = common[:-1] + (_lib.ptr(io),)
        tail = (_lib.ptr(g['ctl']), _lib.ptr(pb), 1 if ub else 0,
                _lib.ptr(g['cw_img']), _lib.ptr(g['pre']), _lib.ptr(g['z_all']),
                _lib.ptr(g['h1']), _lib.ptr(g['h2']), _lib.ptr(g['logits']))

        lib = _lib.load()
        if self.fastgen_persistent and not self._gen_launch_failed.get('persist'):
            # ONE persistent launch for the run.  Every workgroup has to be
            # resident at once; the library checks that against the launch
            # configuration's occupancy (WN_ERR_UNSUPPORTED: the step kernels
            # below).  What it cannot see -- another process or stream holding
            # CUs, a CU mask -- shows as an expired hand-over wait (sync[12]):
            # the generator's state is then restored from a snapshot taken
            # here and the run falls through to the step kernels, which always
            # complete; `steps` advances only after a successful run.
            sync = self._gen_buf('fgp_sync', 16, torch.int32)
            ll = self._gen_buf('fgp_ll', int(lib.wn_fastgen_persist_ll_words(
                self.L, self.S, self.Q)), torch.int64)
            snap = (g['state'].clone(), g['cursors'].clone(), g['pre'].clone())
            code = lib.wn_fastgen_persist(*common, *tail, _lib.ptr(sync),
                                          _lib.ptr(ll), int(n_steps), _lib.stream())
            if code == 0:
                if int(sync[12]) == 0:   # (synchronises; a generation call ends on the host anyway)
                    g['steps'] += int(n_steps)
                    samples_io[:n_io].copy_(io[:n_io])
                    if proba_out is not None:
                        proba_out.view(-1).copy_(pb[:proba_out.numel()])
                    return
                import warnings
                warnings.warn(
                    'wn_fastgen_persist: a hand-over wait inside the persistent '
                    'generation launch expired (2 s: its workgroups were not all '
                    'resident); state restored, continuing with the step kernels '
                    '(net.fastgen_persistent = False selects them up front)')
                g['state'].copy_(snap[0])
                g['cursors'].copy_(snap[1])
                g['pre'].copy_(snap[2])
                io[:n_io].copy_(samples_io[:n_io])
                self._gen_launch_failed['persist'] = True
            elif code != -2:             # WN_ERR_UNSUPPORTED: not resident / shape
                _lib.check(code, 'wn_fastgen_persist')
            del snap

        def one():
            _lib.call('wn_fastgen_step', *common, *tail, _lib.stream())

        def graph_of(nsteps):
            key = (common, tail, nsteps)
            gr = g['graphs'].get(key)
            if gr is None:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(nsteps):
                        one()
                if len(g['graphs']) > 8:
                    g['graphs'].clear()
                g['graphs'][key] = gr
            return gr
        done = 0
        if not g['warm']:
            one()                      # module load etc. outside any capture
            done, g['warm'] = 1, True
        for per in (int(self.fastgen_graph_steps),
                    max(1, int(self.fastgen_graph_steps) // 10)):
            if per > 1 and n_steps - done >= per:
                gr = graph_of(per)
                while n_steps - done >= per:
                    gr.replay()
                    done += per
        for _ in range(n_steps - done):
            one()
        # the last step's draw (every other one ran inside the next step)
        _lib.call('wn_fastgen_finish', self.Q, _lib.ptr(g['cursors']),
                  _lib.ptr(io), _lib.ptr(g['ctl']), _lib.ptr(pb),
                  _lib.ptr(g['logits']), _lib.stream())
        g['steps'] += int(n_steps)
        samples_io[:n_io].copy_(io[:n_io])
        if proba_out is not None:
            proba_out.view(-1).copy_(pb[:proba_out.numel()])

    def _gen_buf(self, name, n, dtype):
        """Persistent per-generator buffer of at least n elements (grown
        geometrically; growing drops the captured graphs, which hold its
        address)."""
        g = self._gen
        buf = g.get(name)
        if buf is None or buf.numel() < n:
            cap = max(int(n), 2 * (buf.numel() if buf is not None else 0), 4096)
            g[name] = buf = torch.zeros(cap, dtype=dtype, device=self.device)
            g['graphs'].clear()
        return buf

    def predict_proba_incremental(self, waveform, global_condition=None,
                                  name='wavenet', push=True):
        '''Computes the probability distribution of the next sample
        incrementally, based on a single sample and all previously passed
        samples (model.py:592-626).  Eager counterpart of running the
        reference's proba op together with `net.push_ops` (push=True) or
        alone (push=False); `net.reset_generator()` is `net.init_ops`.'''
        if self.filter_width > 2:
            raise NotImplementedError("Incremental generation does not "
                                      "support filter_width > 2.")
        if self.scalar_input:
            raise NotImplementedError("Scalar input is not supported by "
                                      "fast generation.")
        self._check_supported()
        g = self._generator(global_condition)
        w = waveform
        if isinstance(w, torch.Tensor):
            g['io'][0:1].copy_(w.reshape(-1)[-1:].to(torch.int32))
        else:
            g['io'][0] = int(np.asarray(w).reshape(-1)[-1])
        self._gen_run(g['io'], 1, 1, 1.0, 0, g['proba'], 1, global_condition,
                      push=push)
        return g['proba'].clone()

    def reset_generator(self):
        """net.init_ops: refill every queue with zeros (model.py:457-479)."""
        self._generator(None)
        self._gen_reset()

    def generate(self, num_samples, seed_samples=None, temperature=1.0,
                 global_condition=None, seed=0, return_proba_every=0):
        """The whole generate.py:195-241 loop on the device: prime with
        `seed_samples` (int codes; default one random-free seed 128 as in
        test_model.py:63), then draw `num_samples` samples with temperature.
        Returns int32 codes [len(seed) + num_samples] (and the probabilities
        of every `return_proba_every`-th step when requested)."""
        self._check_supported()
        if self.filter_width > 2 or self.scalar_input:
            raise NotImplementedError('fast generation needs filter_width 2 '
                                      'and one-hot input (model.py:597-603)')
        if seed_samples is None:
            seed_samples = [self.Q // 2]
        s = torch.as_tensor(np.asarray(seed_samples), dtype=torch.int32).reshape(-1)
        n_given = int(s.numel())
        n_steps = n_given - 1 + int(num_samples)
        io = torch.zeros(n_steps + 1, dtype=torch.int32, device=self.device)
        io[:n_given] = s.to(self.device)
        self.reset_generator()
        pe = int(return_proba_every)
        proba = None
        if pe > 0:
            proba = torch.empty(((n_steps + pe - 1) // pe, self.Q),
                                dtype=torch.float32, device=self.device)
        if pe == 0 and n_given - 1 >= self.fastgen_prime_forward_min and \
                int(num_samples) > 0:
            # the reference's own TODO (generate.py:199-201): fill the queues
            # from a forward pass over the seed instead of stepping through it
            self.prime_generator(io[:n_given - 1], global_condition)
            io2 = io[n_given - 1:].clone()     # [last seed | generated ...]
            self._gen_run(io2, 1, int(num_samples), temperature, seed, None,
                          1, global_condition, multi_cu=self.fastgen_multi_cu)
            io[n_given - 1:] = io2
        elif n_steps > 0:
            self._gen_run(io, n_given, n_steps, temperature, seed, proba,
                          pe if pe > 0 else 1, global_condition,
                          multi_cu=self.fastgen_multi_cu)
        out = io[:n_given + int(num_samples)]
        return (out, proba) if pe > 0 else out

    def prime_generator(self, codes, global_condition=None):
        """Set the incremental-generation queues to the state they have after
        `codes` were pushed one by one from a fresh `reset_generator()`, using
        ONE batch forward pass: layer l's queue (capacity d_l) holds the last
        d_l inputs x_l[t] of that layer (model.py:473-484), which are rows of
        the forward pass's per-layer activation planes."""
        if self.filter_width > 2 or self.scalar_input:
            raise NotImplementedError('fast generation needs filter_width 2 '
                                      'and one-hot input (model.py:597-603)')
        self._check_supported()
        g = self._generator(global_condition)
        self._gen_reset()
        w = torch.as_tensor(codes).to(device=self.device,
                                      dtype=torch.int32).reshape(-1)
        n0 = int(w.numel())
        if n0 == 0:
            return
        ws = self._workspace(1, n0, False)
        ws.q.copy_(w)
        ids = self._gc_ids(global_condition, 1)
        self._forward(ws, ids, save_ts=0)
        # a queue entry is a row of CB 32-wide blocks; block cb of layer l's
        # input is activation plane l * CB + cb
        CB = self.CB
        src, dst, roff = [], [], 0
        for l, d in enumerate(self.dilations):
            t = np.arange(max(0, n0 - d), n0, dtype=np.int64)
            for cb in range(CB):
                src.append((l * CB + cb) * n0 + t)
                dst.append((roff + t % d) * CB + cb)
            roff += d
        src = torch.from_numpy(np.concatenate(src)).to(self.device)
        dst = torch.from_numpy(np.concatenate(dst)).to(self.device)
        X = ws.X.reshape(-1, CH)[:self.L * CB * n0]
        # (+ the forward launch's poison word: 0, or NaN after an expired wait)
        g['state'].view(-1, CH).index_copy_(
            0, dst, X.index_select(0, src) + ws.loss_parts[0])
        g['cursors'][0] = n0
        g['cursors'][1:2].copy_(w[-1:])
        g['steps'] = n0

    def continue_generation(self, num_samples, last_sample, temperature=1.0,
                            global_condition=None, seed=0):
        """Draw `num_samples` more samples after `generate` (the queues stay
        on the device; `last_sample` is the last code drawn so far, which has
        not been pushed yet).  Returns the new int32 codes."""
        self._check_supported()
        n = int(num_samples)
        io = torch.zeros(n + 1, dtype=torch.int32, device=self.device)
        io[0] = int(last_sample)
        self._gen_run(io, 1, n, temperature, seed, None, 1, global_condition,
                      multi_cu=self.fastgen_multi_cu)
        return io[1:]
