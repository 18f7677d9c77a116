"""Residual stack with more than 32 residual / dilation channels
(_create_dilation_layer, wavenet/model.py:236-330, whose constructor puts no
limit on the channel counts, model.py:46-60).

Channels are cut into 32-wide blocks; block cb of layer l's activation is the
[B*T][32] plane l * CB + cb of X / Z / TH / SG / dZ, so the skip-sum, dZ and
dWs GEMMs of model.py see L * CB planes and run unchanged.  Per layer:

  forward   one wn_layer_fwd_blk per dilation-channel block (all taps and all
            input blocks contracted inside the kernel), then the 1x1 residual
            conv as a plane-mode wn_gemm_nn per residual-channel block;
  backward  dz = dZ + dx' Wd^T (plane-mode wn_gemm_nn per block), the gate
            gradients (phase A of wn_layer_bwd_k), the weight gradients from
            wn_layer_wgrad_k per (input block, output block) pair -- their
            32 x 32 results are copied into the [K][C][C] gradient matrices --
            and dx per residual block (wn_layer_bwd_blk).

Weights stay in the reference's [K][Cin][Cout] layout (C = 32 * CB padded
channels), so `net.variables` are plain views exactly as for <= 32 channels.
Layers wider than 8 / K blocks (128 channels at filter width 2) run the two
block kernels in chunks of blocks, filter widths above 8 (any channel count,
one block included) in groups of 8 taps (partial pre-activations through
`ws.pre` forward, dx chained from chunk to chunk backward).
Correctness-first: this is an off-default configuration (the default
wavenet_params.json has 32 / 32 channels and runs the fused kernels).
"""
import torch

from . import _lib

CH = 32


def _blk(net, flat, l):
    K, C = net.KW, net.CHn
    b = net._layer_block(flat, l)
    M = C * C
    return dict(wf=b[0:K * M], wg=b[K * M:2 * K * M], wd=b[2 * K * M:(2 * K + 1) * M],
                bf=b[net.OFF_BF:net.OFF_BF + C], bg=b[net.OFF_BG:net.OFF_BG + C],
                bd=b[net.OFF_BD:net.OFF_BD + C], all=b)


def _chunks(CB, K):
    """(first tap, taps, first block, blocks) pieces with taps x blocks <= 8:
    the block kernels keep that many "virtual taps" of weights in LDS.  Layers
    wider than 8 // K blocks run in chunks of blocks, filter widths above 8 in
    groups of 8 taps (one block each)."""
    per_k = min(K, 8)
    per_b = max(1, 8 // per_k)
    return [(k0, min(per_k, K - k0), i0, min(per_b, CB - i0))
            for k0 in range(0, K, per_k) for i0 in range(0, CB, per_b)]


def _tap_groups(K):
    return [(k0, min(8, K - k0)) for k0 in range(0, K, 8)]


def forward_layers(net, ws, bias, bstride, save_ts, st):
    """Layers 0..L-1 on ws.X[0:CB] (the causal layer's output planes)."""
    L, CB, K, C = net.L, net.CB, net.KW, net.CHn
    B, T, N = ws.B, ws.T, ws.N
    P = net.params
    pstride = N * CH
    for l, d in enumerate(net.dilations):
        w = _blk(net, P, l)
        x0 = ws.X[l * CB]
        chunks = _chunks(CB, K)
        bf = bg = None
        if bias is not None:
            row = bias[l].reshape(-1)
            bf, bg = row, row[C:]
        # ALL output blocks per launch (blockIdx.y); input blocks in chunks of
        # at most 8 // K and filter widths above 8 in groups of 8 taps (the
        # kernel holds a chunk's 2 x taps x blocks weight blocks in LDS);
        # partial pre-activations travel through ws.pre ([block][af | ag])
        for ci, (k0, nk, i0, nb) in enumerate(chunks):
            last = ci == len(chunks) - 1
            wo = (k0 * C + i0 * CH) * C
            _lib.call('wn_layer_fwd_blk',
                      _lib.ptr(ws.X[l * CB + i0]), pstride, nb,
                      _lib.ptr(ws.Z[l * CB]),
                      _lib.ptr(ws.TH[l * CB]) if save_ts else None,
                      _lib.ptr(ws.SG[l * CB]) if save_ts else None,
                      _lib.ptr(w['wf'][wo:]), _lib.ptr(w['wg'][wo:]),
                      C, _lib.ptr(bf), _lib.ptr(bg), bstride, B, T, int(d),
                      nk, 1 if save_ts else 0, C,
                      _lib.ptr(ws.pre) if ci > 0 else None,
                      None if last else _lib.ptr(ws.pre), pstride, k0, K,
                      CB, pstride, st)
        if l == L - 1:
            break
        # x_{l+1} = x_l + z_l Wd (+ bd)   model.py:294-300,330 -- ONE plane-mode
        # GEMM for all residual-channel blocks (planes in, planes out, the
        # addend x_l in the output's plane layout: ld_add = 0)
        if CB <= 4:
            # (up to 128 channels: a streaming kernel, a wave per 32-row tile,
            # instead of a K = N = C GEMM launch that is all prologue and
            # epilogue: 54 -> see DESIGN section 8)
            _lib.call('wn_dense_planes', _lib.ptr(ws.Z[l * CB]), pstride,
                      _lib.ptr(w['wd']),
                      _lib.ptr(w['bd']) if net.use_biases else None,
                      _lib.ptr(ws.X[l * CB]), pstride,
                      _lib.ptr(ws.X[(l + 1) * CB]), pstride, N, C, st)
            continue
        _lib.call('wn_gemm_nn', _lib.ptr(ws.Z[l * CB]), 0, CB, pstride,
                  _lib.ptr(w['wd']), C,
                  _lib.ptr(w['bd']) if net.use_biases else None,
                  None, 0, _lib.ptr(ws.X[l * CB]), 0,
                  _lib.ptr(ws.X[(l + 1) * CB]), 0, CB, pstride, None,
                  N, C, C, 0, st)


def backward_layers(net, ws, ids, st):
    """Residual stack (last layer first), causal layer and global-conditioning
    gradients; ws.dZ holds dtotal * Ws^T (model.py's dZ GEMM ran before)."""
    L, CB, K, C, Q = net.L, net.CB, net.KW, net.CHn, net.Q
    B, T, N = ws.B, ws.T, ws.N
    P, Gr = net.params, net.grads
    ub = net.use_biases
    pstride = N * CH
    lib = _lib.load()
    WF = (2 * K + 1) * 1024
    nslab = ws.nslab
    if CB in (2, 4) and K == 2:
        # the all-input-blocks weight-gradient kernel of 64- / 128-channel
        # layers holds one workgroup per CU: one round of them per pass (the
        # device's CU count: 256 on a whole MI355X, 32 on a CPX partition) --
        # 55 instead of 63 us a 64-channel layer, and half the slabs
        cus = torch.cuda.get_device_properties(net.device).multi_processor_count
        nslab = max(1, min(nslab, cus // (CB // 2)))
    # every layer-block gradient entry is rewritten below except the padding
    # and the last layer's (gradient-free) dense conv: start from zero
    lo, ln = net.segments['layers']
    Gr[lo:lo + ln].zero_()
    daf, dag = ws.da[0][:CB], ws.da[0][CB:]
    dxin, xp = None, 0
    for l in range(L - 1, -1, -1):
        d = int(net.dilations[l])
        w, g = _blk(net, P, l), _blk(net, Gr, l)
        M = C * C
        if dxin is not None:
            _lib.call('wn_transpose', _lib.ptr(w['wd']), C, C, C,
                      _lib.ptr(ws.wdT), C, st)
        gated = False
        if dxin is not None:
            # dz = dZ + dx' Wd^T for all dilation-channel blocks in one
            # plane-mode GEMM (Wd^T as [res][dil]; addend dZ in plane layout)
            if CB <= 4 and net.wide_fuse_gate:
                # ... and the gate gradients in the same launch (dz is not
                # stored: bitwise the planes of the two launches below)
                _lib.call('wn_dense_planes_gate', _lib.ptr(dxin[0]), pstride,
                          _lib.ptr(ws.wdT), _lib.ptr(ws.dZ[l * CB]), pstride,
                          _lib.ptr(ws.TH[l * CB]), _lib.ptr(ws.SG[l * CB]), pstride,
                          _lib.ptr(daf[0]), _lib.ptr(dag[0]), pstride, N, C, st)
                gated = True
            elif CB <= 4:
                _lib.call('wn_dense_planes', _lib.ptr(dxin[0]), pstride,
                          _lib.ptr(ws.wdT), None, _lib.ptr(ws.dZ[l * CB]),
                          pstride, _lib.ptr(ws.dzb[0]), pstride, N, C, st)
            else:
                _lib.call('wn_gemm_nn', _lib.ptr(dxin[0]), 0, CB, pstride,
                          _lib.ptr(ws.wdT), C, None, None, 0,
                          _lib.ptr(ws.dZ[l * CB]), 0, _lib.ptr(ws.dzb[0]), 0, CB,
                          pstride, None, N, C, C, 0, st)
        # ---- gate gradients of every dilation-channel block (one launch,
        # blockIdx.y = block; gate gradients only: the filter width just sizes
        # a weight staging area this mode does not read)
        dz0 = ws.dZ[l * CB] if dxin is None else ws.dzb[0]
        if not gated:
            _lib.call('wn_layer_bwd_k', None, None, None, None, None,
                      _lib.ptr(dz0), _lib.ptr(ws.TH[l * CB]), _lib.ptr(ws.SG[l * CB]),
                      _lib.ptr(w['all']), _lib.ptr(daf[0]), _lib.ptr(dag[0]), B, T, d,
                      min(K, 8), 0, 1, CB, pstride, st)
        for jb in range(CB):
            if ws.dsum is not None:
                _lib.call('wn_colsum_clip', _lib.ptr(daf[jb]), _lib.ptr(dag[jb]),
                          B, T, _lib.ptr(ws.dsum_part), _lib.ptr(ws.cs_tmp), st)
                dv = ws.dsum[l].view(B, 2, CB, CH)
                dv[:, 0, jb].copy_(ws.cs_tmp[:, :CH])
                dv[:, 1, jb].copy_(ws.cs_tmp[:, CH:])
        # ---- weight gradients per (input block a, output block b) pair into
        # the pair's slab region, then ONE fixed-order reduction per layer (and
        # group of 8 taps) that writes the [K][C][C] / [C][C] matrices and the
        # bias vectors directly
        flat = ws.pslabs.view(-1)
        for k0, nk in _tap_groups(K):
            dense = dxin is not None and k0 == 0
            # (all CB x CB pairs in one launch: blockIdx.y = pair)
            _lib.call('wn_layer_wgrad_k', _lib.ptr(ws.X[l * CB]),
                      _lib.ptr(daf[0]), _lib.ptr(dag[0]),
                      _lib.ptr(ws.Z[l * CB]) if dense else None,
                      _lib.ptr(dxin[0]) if dense else None,
                      _lib.ptr(flat), nslab, B, T, d, nk, k0, K, CB, pstride, st)
            _lib.call('wn_reduce_pair_slabs', _lib.ptr(flat), nslab, CB, nk,
                      1 if dense else 0, 1 if (ub and k0 == 0) else 0,
                      _lib.ptr(g['all']), C, net.OFF_BF, k0, K, st)
        # ---- dx of every residual-channel block
        dxo = ws.dx[xp]
        # ALL residual-channel blocks per launch (blockIdx.y); dilation blocks /
        # taps in chunks: a chunk's dx is the next one's dxin
        for ci, (k0, nk, j0, nb) in enumerate(_chunks(CB, K)):
            src = dxo[0] if ci > 0 else (None if dxin is None else dxin[0])
            wo = k0 * M + j0 * CH
            _lib.call('wn_layer_bwd_blk', _lib.ptr(daf[j0]),
                      _lib.ptr(dag[j0]), pstride, nb, _lib.ptr(src),
                      _lib.ptr(dxo[0]), _lib.ptr(w['wf'][wo:]),
                      _lib.ptr(w['wg'][wo:]), C, M, B, T, d, nk, k0, K, CB,
                      pstride, st)
        dxin, xp = dxo, 1 - xp
    sp = ws.splits['causal']
    if net.scalar_input:
        # ---- causal layer on scalar input (model.py:143-153): per block
        K0 = net.initial_filter_width
        gs = net._seg(Gr, 'causal').view(K0, C)
        for cb in range(CB):
            _lib.call('wn_scalar_causal_wgrad', _lib.ptr(ws.audio),
                      _lib.ptr(dxin[cb]), _lib.ptr(ws.slabs), sp, B, T, K0, st)
            _lib.call('wn_reduce_slabs', _lib.ptr(ws.slabs), sp, K0 * CH, 1,
                      0, 0, K0 * CH, _lib.ptr(ws.blk_tmp), 0, 1, 0, st)
            gs[:, cb * CH:(cb + 1) * CH].copy_(
                ws.blk_tmp[:K0 * CH].view(K0, CH))
        K = 0                                  # no one-hot taps below
    # ---- causal layer (model.py:227-234): one-hot contraction per tap / block
    gc_ = None if net.scalar_input else net._seg(Gr, 'causal').view(K, Q, C)
    sl = lib.wn_gemm_tn_slab_floats(Q, CH)
    for tap in range(K):
        shift = (K - 1 - tap) + (K - 1) // 2
        for cb in range(CB):
            _lib.call('wn_gemm_tn', None, 0, 0, 0, _lib.ptr(ws.q), shift, T,
                      _lib.ptr(dxin[cb]), CH, _lib.ptr(ws.slabs), sp, N, Q, CH,
                      0, st)
            _lib.call('wn_reduce_slabs', _lib.ptr(ws.slabs), sp, sl, 1, 0, 0,
                      Q * CH, _lib.ptr(ws.blk_tmp), 0, 1, 0, st)
            gc_[tap, :, cb * CH:(cb + 1) * CH].copy_(
                ws.blk_tmp[:Q * CH].view(Q, CH))
    if ws.dsum is not None:
        _lib.call('wn_gc_grad', _lib.ptr(net._layer_block(P, 0)),
                  net.layer_stride, net.OFF_GC, net.G,
                  _lib.ptr(net._seg(P, 'emb')), net.card, _lib.ptr(ids),
                  _lib.ptr(ws.dsum), L, B, _lib.ptr(net._layer_block(Gr, 0)),
                  _lib.ptr(net._seg(Gr, 'emb')), _lib.ptr(ws.gc_part), C, st)
