"""The persistent residual-stack launch (csrc/wn_stack.hip, `net.stack_fwd`)
against the one-launch-per-layer kernels it replaces: same arithmetic in the
same order, so every activation plane, the logits, the loss and every gradient
must be BITWISE equal (model.py:236-330 x L, model.py:417-428).  The
per-layer path is itself checked against the oracle in test_gpu_model.py."""
import json
import os

import numpy as np
import pytest
import torch

from util import ROOT, TINY, build_pair, cfg_with, synth_audio

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


@pytest.mark.parametrize('name,mk,B,T,kind', CASES, ids=[c[0] for c in CASES])
def test_stack_backward_equals_per_layer(hip_lib, name, mk, B, T, kind):
    """wn_stack_bwd vs one wn_layer_bwd2 per layer: dL/dx_0 bitwise (per-tile
    arithmetic is identical), weight gradients to rounding (the tiles of a
    slab are summed in another order), repeated runs bitwise."""
    cfg = mk()
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    a.stack_bwd, b.stack_bwd = True, False
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
        assert torch.equal(wa.DX[0], wb.dx[0][0]) or torch.equal(wa.DX[0], wb.dx[1][0])
        ga, gb = a.grads, b.grads
        # per variable, against that variable's own largest entry (a bias or
        # GC weight must not hide behind the bucket's largest gradient)
        ta, tb = a._views(ga), b._views(gb)
        for (n, va), (_, vb) in zip(a.named_variables(ta), b.named_variables(tb)):
            scale = float(vb.abs().max())
            err = float((va - vb).abs().max())
            assert err <= 1e-5 * scale + 1e-30, (n, err, scale)
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
    short = audio[:, :4500]
    la, lb = a.loss(short), b.loss(short)
    torch.cuda.synchronize()
    wa = [w for w in a._ws.values() if w.T == 4500][0]
    if not getattr(wa, 'stack_bwd', False):
        pytest.skip('the shorter workspace did not take the stack backward')
    assert wa.stack_ctl_b.cpu().tolist()[2] == 2       # started at 1, one launch
    assert float(la) == float(lb)
    ta, tb = a._views(a.grads), b._views(b.grads)
    for (n, va), (_, vb) in zip(a.named_variables(ta), b.named_variables(tb)):
        scale = float(vb.abs().max())
        assert float((va - vb).abs().max()) <= 1e-5 * scale + 1e-30, n
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
