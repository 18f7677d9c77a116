"""The persistent residual-stack launch (csrc/wn_stack.hip, `net.stack_fwd`)
against the one-launch-per-layer kernels it replaces: same arithmetic in the
same order, so every activation plane, the logits, the loss and every gradient
must be BITWISE equal (model.py:236-330 x L, model.py:417-428).  These are
SELF-comparisons of two device paths.  The oracle comparisons of each path are
elsewhere: tests/test_gpu_model.py::test_loss_and_gradients_vs_oracle runs
every <= 32-channel two-tap case three times -- the library's choice (16-row
launches), `-rows32` (stack_fwd_kernel<2,16> / stack_bwd_kernel<8>, forced)
and `-perlayer` (layer_fwd_kernel / layer_bwd2d_kernel) -- and
tests/test_gpu_fullsize.py::test_full_length_vs_oracle[*-rows32] /
test_natural_32_row_batch_vs_oracle do the same at 16000 / 3 x 11000 samples."""
import json
import os

import numpy as np
import pytest
import torch

from util import ROOT, TINY, build_pair, cfg_with, synth_audio
from wavenet import WaveNetModel
from wavenet._lib import stack_variant

pytestmark = pytest.mark.gpu


def default_cfg(B, **kw):
    p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
    c = {k: p[k] for k in p if k != 'sample_rate'}
    c['batch_size'] = B
    c.update(kw)
    return c


def _pair(cfg):
    """two models with identical weights (biases N(0, 0.1)): persistent stack
    launch vs one launch per layer"""
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    assert torch.equal(a.params, b.params)
    a.stack_fwd, b.stack_fwd = True, False
    a.stack_bwd = b.stack_bwd = False
    return a, b


def _ctl(net):
    ws = list(net._ws.values())
    return [w.stack_ctl.cpu().tolist() for w in ws]


CASES = [
    ('default_B8_T16000', lambda: default_cfg(8), 8, 16000, None),
    ('default_B1_T16000', lambda: default_cfg(1), 1, 16000, None),
    ('default_B3_T5211', lambda: default_cfg(3), 3, 5211, None),       # ragged last tile
    ('default_B20_T16000', lambda: default_cfg(20), 20, 16000, None),  # more groups than CUs
    ('default_gc_B4_T7000', lambda: default_cfg(
        4, global_condition_channels=32, global_condition_cardinality=377), 4, 7000, 'gc'),
    ('tiny_B2_T100', lambda: cfg_with(TINY, batch_size=2), 2, 100, None),
    ('default_nobias_B2_T4000', lambda: default_cfg(2, use_biases=False), 2, 4000, None),
    # clips shorter than most dilations (taps entirely before the clip start),
    # shorter than one tile, one row over a tile
    ('default_B2_T40', lambda: default_cfg(2), 2, 40, None),
    ('default_B1_T31', lambda: default_cfg(1), 1, 31, None),
    ('default_B3_T33', lambda: default_cfg(3), 3, 33, None),
    ('default_B300_T64', lambda: default_cfg(300), 300, 64, None),     # many clips of two tiles
    ('one_layer_B2_T500', lambda: default_cfg(2, dilations=[4]), 2, 500, None),
    ('two_layers_B2_T500', lambda: default_cfg(2, dilations=[64, 1]), 2, 500, None),
]


@pytest.fixture(autouse=True)
def _tile_rows_32(monkeypatch):
    """The bitwise statements of this file are about the 32-row launches; small
    batches would otherwise run the 16-row ones (wn_stack_tile_rows), which sum
    in another grouping -- they have their own tests below.  (The variant word
    is an explicit argument of the C entry points; the class default is what a
    new model passes.)"""
    monkeypatch.setattr(WaveNetModel, 'DEFAULT_STACK_VARIANT', stack_variant(rows=32))


@pytest.mark.parametrize('name,mk,B,T,kind', CASES, ids=[c[0] for c in CASES])
def test_stack_forward_bitwise_equals_per_layer(hip_lib, name, mk, B, T, kind):
    cfg = mk()
    a, b = _pair(cfg)
    audio = synth_audio(B, T)
    gc = None
    if kind == 'gc':
        gc = np.array([(37 * i) % 377 for i in range(B)], np.int32)
    for rep in range(3):        # epochs 1, 2, 3 on the same workspace (and the launch plan)
        la = a.loss(audio, global_condition_batch=gc) if gc is not None else a.loss(audio)
        lb = b.loss(audio, global_condition_batch=gc) if gc is not None else b.loss(audio)
        torch.cuda.synchronize()
        wa, wb = list(a._ws.values())[0], list(b._ws.values())[0]
        for pl in ('X', 'Z', 'SG'):
            pa, pb = getattr(wa, pl), getattr(wb, pl)
            if not torch.equal(pa, pb):
                bad = [l for l in range(pa.shape[0]) if not torch.equal(pa[l], pb[l])]
                raise AssertionError('%s planes differ at layers %s (rep %d)' % (pl, bad[:8], rep))
        assert float(la) == float(lb)
        assert torch.equal(a.grads, b.grads)
        ctl = wa.stack_ctl.cpu().tolist()
        assert ctl[0] == 0 and ctl[1] == 0 and ctl[3] == 0, ctl
        assert ctl[2] == 2 + rep, ctl


@pytest.mark.parametrize('waves', ['4', '8'])
@pytest.mark.parametrize('name,mk,B,T,kind', CASES, ids=[c[0] for c in CASES])
def test_stack_forward_16_row_tiles(hip_lib, monkeypatch, name, mk, B, T, kind, waves):
    """The small-batch launch (16-row tiles on v_mfma_f32_16x16x4_f32) against
    one launch per layer: the same products summed in another grouping, so
    every plane agrees to rounding (2e-5 of the plane's largest entry after up
    to 50 layers), run to run bitwise."""
    var = stack_variant(rows=16, waves=int(waves))
    monkeypatch.setattr(WaveNetModel, 'DEFAULT_STACK_VARIANT', var)
    cfg = mk()
    a, b = _pair(cfg)
    assert hip_lib.wn_stack_tile_rows(B, T, var) == 16
    audio = synth_audio(B, T)
    gc = np.array([(37 * i) % 377 for i in range(B)], np.int32) if kind == 'gc' else None
    prev = None
    worst = {'X': 0.0, 'Z': 0.0, 'SG': 0.0}
    for rep in range(3):
        la = a.loss(audio, global_condition_batch=gc) if gc is not None else a.loss(audio)
        lb = b.loss(audio, global_condition_batch=gc) if gc is not None else b.loss(audio)
        torch.cuda.synchronize()
        wa, wb = list(a._ws.values())[0], list(b._ws.values())[0]
        for pl in ('X', 'Z', 'SG'):
            pa, pb = getattr(wa, pl), getattr(wb, pl)
            for l in range(pa.shape[0]):
                sc = float(pb[l].abs().max())
                err = float((pa[l] - pb[l]).abs().max())
                worst[pl] = max(worst[pl], err / (sc + 1e-30))
                assert err <= (2e-5 if pl == 'X' else 1e-4) * sc + 1e-30, (pl, l, err, sc, rep)
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
        scale = float(b.grads.abs().max())
        assert float((a.grads - b.grads).abs().max()) <= 2e-3 * scale
        ctl = wa.stack_ctl.cpu().tolist()
        assert ctl[0] == 0 and ctl[1] == 0 and ctl[3] == 0 and ctl[2] == 2 + rep, ctl
        cur = (wa.X.clone(), wa.Z.clone(), wa.SG.clone(), float(la))
        if prev is not None:
            assert all(torch.equal(x, y) for x, y in zip(prev[:3], cur[:3])) and prev[3] == cur[3]
        prev = cur
    print('16-row forward %s waves %s: worst plane errors %s' % (name, waves, worst))


def test_stack_inference_forward(hip_lib):
    """predict_proba / loss(backward=False): no sigmoid planes."""
    a, b = _pair(default_cfg(1))
    audio = synth_audio(1, 6000)
    a.loss(audio, backward=False)
    b.loss(audio, backward=False)
    wa, wb = list(a._ws.values())[0], list(b._ws.values())[0]
    assert torch.equal(wa.logits, wb.logits)
    q = np.random.default_rng(2).integers(0, 256, 3000).astype(np.int32)
    assert torch.equal(a.predict_proba(q), b.predict_proba(q))
    for ws in a._ws.values():
        assert ws.stack_ctl.cpu().tolist()[3] == 0
    a.check_device_errors()


def _f64_layer_grads(net, ws, l):
    """float64 weight / bias gradients of residual block l from the planes the
    stack launches leave in the workspace (hand-written gradient of
    model.py:236-330; the anti-causal tap is the only coupling in time)."""
    B, T, L = ws.B, ws.T, net.L
    d = int(net.dilations[l])
    f = lambda t: t.double().reshape(B, T, -1)
    X, Z, SG, dZ = f(ws.X[l]), f(ws.Z[l]), f(ws.SG[l]), f(ws.dZ[l])
    blk = net._layer_block(net.params, l).double()
    Wd = blk[4096:5120].reshape(32, 32)
    dz = dZ.clone()
    dxn = None
    if l + 1 < L:
        dxn = f(ws.DX[l + 1])
        # DX[l+1] holds dx_{l+1} without the term of the tap, which sits in
        # DQ[l+1] at the rows d_{l+1} later
        dd = int(net.dilations[l + 1])
        dxn = dxn.clone()
        if dd < T:
            dxn[:, :T - dd] += f(ws.DQ[l + 1])[:, dd:]
        dz = dz + dxn @ Wd.t()
    th = torch.where(SG > 1e-30, Z / SG.clamp_min(1e-30), torch.zeros_like(Z))
    da_f = dz * (SG - Z * th)
    da_g = dz * Z * (1.0 - SG)
    Xp = torch.zeros_like(X)
    if d < T:
        Xp[:, d:] = X[:, :T - d]
    mm = lambda a_, b_: torch.einsum('btc,bte->ce', a_, b_).reshape(-1)
    out = {'Wf0': mm(Xp, da_f), 'Wf1': mm(X, da_f), 'Wg0': mm(Xp, da_g),
           'Wg1': mm(X, da_g), 'bf': da_f.sum((0, 1)), 'bg': da_g.sum((0, 1))}
    if dxn is not None:
        out['Wd'] = mm(Z, dxn)
        out['bd'] = dxn.sum((0, 1))
    return out


_SECTIONS = [('Wf0', 0, 1024), ('Wf1', 1024, 2048), ('Wg0', 2048, 3072),
             ('Wg1', 3072, 4096), ('Wd', 4096, 5120), ('bf', 5120, 5152),
             ('bg', 5152, 5184), ('bd', 5184, 5216)]


def _check_layer_grads(a, b, wa, name):
    """Per variable of every residual block (Wf[0], Wf[1], Wg[0], Wg[1], Wd and
    the three biases; 32 padded channels): the persistent launch and the
    per-layer kernels sum the same per-tile products in different orders, so
    both are compared with a float64 evaluation of the same planes -- each in
    the norm of the VARIABLE (a bias or a small matrix cannot hide behind the
    bucket's largest gradient), and the persistent launch may not be further
    from float64 than 2.5 x the per-layer kernels plus 1e-6 of the variable
    (fp32 summation noise of a 128000-row sum with cancellation reaches 2e-3
    of a variable's largest entry in BOTH paths).  Everything outside the
    residual blocks comes from the same kernels on bitwise equal inputs."""
    ga, gb = a.grads, b.grads
    L = a.L
    worst = 0.0
    for l in range(L):
        ref = _f64_layer_grads(a, wa, l)
        xa = a._layer_block(ga, l)[:5216].double()
        xb = b._layer_block(gb, l)[:5216].double()
        for nm, lo, hi in _SECTIONS:
            if nm not in ref or (nm[0] == 'b' and not a.use_biases):
                continue
            r = ref[nm]
            n = float(r.norm()) + 1e-300
            ea = float((xa[lo:hi] - r).norm()) / n
            eb = float((xb[lo:hi] - r).norm()) / n
            worst = max(worst, ea)
            assert ea <= 2.5 * eb + 1e-6, (name, l, nm, ea, eb)
            assert ea <= 1e-4, (name, l, nm, ea)
    # the other variables: same kernels, bitwise equal inputs (dx_0, dZ, ...)
    scale = float(gb.abs().max())
    assert float((ga - gb).abs().max()) <= 2e-6 * max(scale, 1e-30)


_BWD_PARAMS = [pytest.param(*c, r, w, id='%s-%s-%s' % (c[0], r, w))
               for c in CASES for r, w in [(32, '8'), (16, '8'), (16, '4')]]


@pytest.mark.parametrize('name,mk,B,T,kind,rows,waves', _BWD_PARAMS)
def test_stack_backward_equals_per_layer(hip_lib, monkeypatch, name, mk, B, T, kind, rows, waves):
    """wn_stack_bwd vs one wn_layer_bwd2 per layer: dL/dx_0 and the weight
    gradients to rounding (the persistent launch sums a tile's own rows before
    the anti-causal tap, and the tiles of a slab in another order), repeated
    runs bitwise.  `a` keeps dL/dx_l of every layer (the per-layer float64
    check needs them); `c` is the product configuration -- ONE dx plane
    rewritten in place from layer to layer -- and must give the same bits.
    rows = 16: the small-batch launch (16-row tiles), forced on every shape
    here; all three models then run the 16-row FORWARD too, so the planes the
    backward paths read are bitwise the same."""
    var = stack_variant(rows=rows, waves=int(waves))
    monkeypatch.setattr(WaveNetModel, 'DEFAULT_STACK_VARIANT', var)
    assert hip_lib.wn_stack_tile_rows(B, T, var) == rows
    cfg = mk()
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    c, _ = build_pair(cfg)
    a.stack_bwd, b.stack_bwd = True, False
    a.stack_bwd_keep_dx = True
    assert c.stack_bwd and not c.stack_bwd_keep_dx
    audio = synth_audio(B, T)
    gc = np.array([(37 * i) % 377 for i in range(B)], np.int32) if kind == 'gc' else None
    prev = None
    for rep in range(3):
        la = a.loss(audio, global_condition_batch=gc) if gc is not None else a.loss(audio)
        lb = b.loss(audio, global_condition_batch=gc) if gc is not None else b.loss(audio)
        torch.cuda.synchronize()
        wa, wb = list(a._ws.values())[0], list(b._ws.values())[0]
        if not wa.stack_bwd:
            pytest.skip('configuration runs the generic backward kernels')
        assert float(la) == float(lb)
        lc = c.loss(audio, global_condition_batch=gc) if gc is not None else c.loss(audio)
        torch.cuda.synchronize()
        wc = list(c._ws.values())[0]
        assert wc.DX.shape[0] == 1 and wa.DX.shape[0] == a.L
        assert float(lc) == float(la)
        assert torch.equal(wc.DX[0], wa.DX[0])
        assert torch.equal(c.grads, a.grads)
        assert int(wc.stack_ctl_b[3]) == 0
        # (same terms, another order of summation: the persistent launch adds
        # the tile's own rows first and the anti-causal tap last)
        # (the per-layer launches ping-pong between two planes, layer L - 1
        # into the first: dL/dx_0 ends up in plane (L - 1) % 2 -- the other one
        # may never have been written)
        dxb = wb.dx[(a.L - 1) % 2][0]
        sc = float(dxb.abs().max())
        assert float((wa.DX[0] - dxb).abs().max()) <= 1e-5 * sc + 1e-30
        ga, gb = a.grads, b.grads
        _check_layer_grads(a, b, wa, name)
        if prev is not None:
            assert torch.equal(prev, ga)          # run-to-run determinism
        prev = ga.clone()
        ctl = wa.stack_ctl_b.cpu().tolist()
        assert ctl[0] == 0 and ctl[1] == 0 and ctl[3] == 0 and ctl[2] == 2 + rep, ctl


def test_child_workspace_owns_fresh_backward_control_block(hip_lib):
    """A training workspace created while `stack_bwd` was off has no backward
    stack buffers; a carved-out child created after the option is switched on
    allocates its own flags (all 0) and control block, whose epoch must start
    at 1 -- with epoch 0 every dependency wait would pass at once."""
    cfg = default_cfg(2)
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    a.stack_bwd = b.stack_bwd = False
    audio = synth_audio(2, 6000)
    a.loss(audio)
    b.loss(audio)
    a.stack_bwd = True                     # read when a workspace is created
    a.stack_bwd_keep_dx = True             # (_check_layer_grads reads every layer's dx)
    short = audio[:, :4500]
    la, lb = a.loss(short), b.loss(short)
    torch.cuda.synchronize()
    wa = [w for w in a._ws.values() if w.T == 4500][0]
    if not getattr(wa, 'stack_bwd', False):
        pytest.skip('the shorter workspace did not take the stack backward')
    assert wa.stack_ctl_b.cpu().tolist()[2] == 2       # started at 1, one launch
    assert float(la) == float(lb)
    _check_layer_grads(a, b, wa, 'child workspace')
    a.check_device_errors()


def test_bounded_wait_expires_loudly(hip_lib):
    """A launch whose dependencies can never be satisfied (the ticket counter is
    pre-advanced, so the first tile groups are never computed) must still
    drain: every poll is bounded (2 s), the control block records the error,
    the outputs are poisoned with NaN, and the block is re-armed so that the
    next launch is correct again."""
    cfg = default_cfg(1)
    a, b = _pair(cfg)
    a.stack_bwd = b.stack_bwd = False
    audio = synth_audio(1, 4000)
    a.loss(audio)                      # creates the workspace (epoch 1 -> 2)
    b.loss(audio)
    ws = list(a._ws.values())[0]
    torch.cuda.synchronize()
    ws.stack_ctl[0] = 3                # groups 0..2 will never be handed out
    bad = a.loss(audio)
    torch.cuda.synchronize()
    ctl = ws.stack_ctl.cpu().tolist()
    assert ctl[3] == 1 and ctl[0] == 0 and ctl[1] == 0, ctl
    assert np.isnan(float(bad))        # the poison word is part of the loss sum
    assert np.isnan(float(a.loss(audio)))          # ... and stays until cleared
    from wavenet._lib import WaveNetHipError
    with pytest.raises(WaveNetHipError):
        a.check_device_errors()
    a.reset_device_errors()
    good = a.loss(audio)
    ref = b.loss(audio)
    assert float(good) == float(ref)
    wb = list(b._ws.values())[0]
    assert torch.equal(ws.Z, wb.Z)
    assert ws.stack_ctl.cpu().tolist()[3] == 0
    a.check_device_errors()


@pytest.mark.parametrize('B,T,gc', [(1, 700, False), (1, 1500, False), (2, 333, True), (3, 40, False),
                                    (1, 5200, True)])
def test_forward_with_fused_skip_sum_equals_stack_plus_gemm(hip_lib, monkeypatch, B, T, gc):
    """wn_stack_fwd_skip (small batches, 512 skip channels: the skip sum by partner
    waves inside the 16-row forward launch) against wn_stack_fwd + the skip GEMM:
    the same X / Z / sigmoid planes bitwise, h1 to rounding (another order of
    the sum), loss and every gradient; forward-only too; ragged last tiles,
    groups of fewer than four tiles, global conditioning."""
    from util import DEFAULT
    monkeypatch.setattr(WaveNetModel, 'DEFAULT_STACK_VARIANT', 0)   # (the library's choice: 16-row tiles)
    kw = dict(global_condition_channels=32, global_condition_cardinality=377) if gc else {}
    cfg = cfg_with(DEFAULT, batch_size=B, **kw)
    assert hip_lib.wn_stack_fwd_skip_ok(B, T, 512, 0) == 1
    assert hip_lib.wn_stack_fwd_skip_ok(B, T, 256, 0) == 0
    assert hip_lib.wn_stack_fwd_skip_ok(8, 16000, 512, 0) == 0
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    assert a.stack_fwd_skip
    b.stack_fwd_skip = False
    audio = synth_audio(B, T)
    ids = np.array([(37 * i) % 377 for i in range(B)], np.int32) if gc else None
    la, lb = a.loss(audio, ids), b.loss(audio, ids)
    torch.cuda.synchronize()
    wa = [w for w in a._ws.values() if w.training][0]
    wb = [w for w in b._ws.values() if w.training][0]
    assert getattr(wa, 'skimg', None) is not None and getattr(wb, 'skimg', None) is None
    assert torch.equal(wa.X, wb.X) and torch.equal(wa.Z, wb.Z) and torch.equal(wa.SG, wb.SG)
    assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb)))
    scale = float(b.grads.abs().max())
    # (h1 differs by rounding, so a ReLU at a kink may flip -- more of them in a
    # longer clip; the oracle comparison at the device's kinks is test_gpu_model's)
    assert float((a.grads - b.grads).abs().max()) <= 1e-3 * scale
    assert float((a.grads - b.grads).norm()) <= 1e-5 * max(1.0, T / 250.0) * float(b.grads.norm())
    # the same twice
    g1 = a.grads.clone()
    la2 = a.loss(audio, ids)
    assert float(la2) == float(la) and torch.equal(a.grads, g1)
    # forward only
    fa, fb = a.loss(audio, ids, backward=False), b.loss(audio, ids, backward=False)
    assert abs(float(fa) - float(fb)) <= 1e-6 * max(1.0, abs(float(fb)))
    assert int(wa.stack_ctl[3]) == 0 and not bool(torch.isnan(wa.loss_parts[:2]).any())

