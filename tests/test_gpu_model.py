"""GPU parity of the WaveNet hot path (forward, loss, hand-written backward,
optimizers, naive and incremental prediction) against the CPU oracle and the
committed golden fixtures.  fp32 tolerance: 1e-4 (BASELINE.json north_star),
gradients relative to the largest entry of each variable."""
import os

import numpy as np
import pytest
import torch

from util import (O, TINY, MID, DEFAULT, cfg_with, build_pair, flat_named,
                  tree_to_numpy, model_kwargs, synth_audio,
                  oracle_grads_at_device_kinks)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


GRAD_REL = 2e-5     # fp32 MFMA path vs float64 oracle, per variable, relative
                    # to that variable's largest gradient entry
_ERRLOG = {}


def _fp32_oracle_error(tag):
    """{variable: error of the float32 ORACLE against float64, relative to the
    variable's largest entry} for the cases tests/golden/make_golden.py
    (fp32_error_cases) holds -- the ill-conditioned deep stacks -- else {}."""
    fx = np.load(os.path.join(GOLD, 'fp32_oracle_error.npz'))
    if tag is None or tag + '/names' not in fx.files:
        return {}
    return dict(zip([str(n) for n in fx[tag + '/names']], fx[tag + '/err32']))


def check_grads(net, ref_g, tol=TOL, rel=GRAD_REL, tag=None, log_tag=None):
    """Every variable's gradient against the float64 oracle: <= GRAD_REL of
    the variable's largest entry (exact-fp32 MFMA accumulation over <= a few
    10^4 rows observes ~1e-6; a missing small term would be orders above).
    Where a float32 evaluation of the ORACLE itself is further than a quarter
    of that from float64 (the 70-layer 8-channel stack: 1.1e-5; committed in
    tests/golden/fp32_oracle_error.npz) the bound is 4 x the oracle's own
    float32 error instead: conditioning of the network, not a missing term.
    The observed ratios are logged to gpurun_out/grad_errors.json."""
    got = tree_to_numpy(net.gradients)
    e32 = _fp32_oracle_error(tag)
    bad = []
    for (n, a), (_, b) in zip(flat_named(got), flat_named(ref_g)):
        err = np.abs(a - b).max()
        scale = np.abs(b).max()
        rel_v = max(rel, 4.0 * float(e32.get(n, 0.0)))
        if tag is not None:
            _ERRLOG.setdefault(log_tag or tag, {})[n] = (float(err), float(scale))
        if not (err <= rel_v * scale + 1e-9 and err <= tol * max(1.0, scale)):
            bad.append((n, float(err), float(scale)))
    assert not bad, bad[:6]


@pytest.fixture(scope='module', autouse=True)
def _dump_errlog():
    yield
    if _ERRLOG:
        import json
        out = os.path.join(os.path.dirname(GOLD), '..', 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        worst = {t: max(((e / (s + 1e-30), n) for n, (e, s) in v.items()
                         if s > 0), default=(0, ''))
                 for t, v in _ERRLOG.items()}
        with open(os.path.join(out, 'grad_errors.json'), 'w') as f:
            json.dump({'worst_ratio': worst, 'all': _ERRLOG}, f, indent=1)


def check_planes(net, cfg, c, B, T, tol=TOL):
    """Per-layer residual-stream input x_l and gated output z_l against the
    oracle cache (network_forward(keep=True)), so that a mismatch localises to
    a layer instead of showing up only in the logits."""
    ws = [w for w in net._ws.values() if w.T == T][0]
    R, D = cfg['residual_channels'], cfg['dilation_channels']
    for l, lc in enumerate(c['layers']):
        x = ws.X[l].cpu().numpy().reshape(B, T, -1)[:, :, :R]
        z = ws.Z[l].cpu().numpy().reshape(B, T, -1)[:, :, :D]
        ex = np.abs(x - lc['x']).max()
        ez = np.abs(z - lc['z']).max()
        assert ex < tol and ez < tol, (l, float(ex), float(ez))


CASES = [
    ('tiny', cfg_with(TINY, batch_size=2), 37, False, None),
    ('tiny_T_lt_32', cfg_with(TINY, batch_size=3), 5, False, None),
    ('tiny_T1', cfg_with(TINY, batch_size=2), 1, False, None),
    ('tiny_ragged', cfg_with(TINY, batch_size=2), 131, False, None),
    ('nobias', cfg_with(TINY, batch_size=1, use_biases=False), 70, False, None),
    ('mid', cfg_with(MID, batch_size=2), 300, False, None),
    ('mid_T_lt_d', cfg_with(MID, batch_size=1), 50, False, None),
    ('gc', cfg_with(TINY, batch_size=3, global_condition_channels=4,
                    global_condition_cardinality=5), 50, True, None),
    ('gc_square', cfg_with(MID, batch_size=3, global_condition_channels=3,
                           global_condition_cardinality=3), 90, True, None),
    ('rp_l2', cfg_with(TINY, batch_size=2, residual_postproc=True), 40, False,
     0.01),
    ('r16', cfg_with(MID, batch_size=1, residual_channels=16,
                     dilation_channels=16, skip_channels=32,
                     quantization_channels=128), 200, False, None),
    ('scalar_k4', cfg_with(TINY, batch_size=2, scalar_input=True,
                           initial_filter_width=4), 70, False, None),
    ('scalar_k32', cfg_with(MID, batch_size=2, scalar_input=True,
                            initial_filter_width=32), 150, False, None),
    ('scalar_T_lt_k', cfg_with(TINY, batch_size=1, scalar_input=True,
                               initial_filter_width=32), 9, False, None),
    ('scalar_k70', cfg_with(TINY, batch_size=2, scalar_input=True,
                            initial_filter_width=70), 120, False, None),
    # more than 32 channels: two 32-wide channel blocks (wavenet/blocked.py)
    ('r64', cfg_with(MID, batch_size=2, residual_channels=64,
                     dilation_channels=64, skip_channels=32), 150, False, None),
    ('r48_d40_gc', cfg_with(MID, batch_size=2, residual_channels=48,
                            dilation_channels=40, skip_channels=32,
                            global_condition_channels=4,
                            global_condition_cardinality=5), 130, True, None),
    ('r96_d128', cfg_with(TINY, batch_size=1, residual_channels=96,
                          dilation_channels=128, skip_channels=32,
                          quantization_channels=32), 60, False, None),
    # more than 8 / K channel blocks per layer: the block kernels run in chunks
    ('r160_d136', cfg_with(TINY, batch_size=1, residual_channels=160,
                           dilation_channels=136), 70, False, None),
    # more than 256 channels (9 / 10 blocks)
    ('r288_d320', cfg_with(TINY, batch_size=1, residual_channels=288,
                           dilation_channels=320, skip_channels=32,
                           quantization_channels=32), 40, False, None),
    # the widest supported model (32 channel blocks) with biases and GC
    ('r1024_d544_gc', cfg_with(TINY, batch_size=2, residual_channels=1024,
                               dilation_channels=544, skip_channels=32,
                               quantization_channels=32, dilations=[1, 2],
                               global_condition_channels=3,
                               global_condition_cardinality=4), 24, True, None),
    # more skip / quantization channels and layers than the default stack
    ('S1024_Q512', cfg_with(MID, batch_size=1, skip_channels=1024,
                            quantization_channels=512), 120, False, None),
    ('L70', cfg_with(TINY, batch_size=1, dilations=[1, 2, 4, 8, 16] * 14), 200,
     False, None),
    # filter widths above 8: groups of 8 taps (wavenet/blocked.py)
    ('k11', cfg_with(TINY, batch_size=2, filter_width=11), 120, False, None),
    ('k19_r64_d40', cfg_with(TINY, batch_size=1, filter_width=19,
                             residual_channels=64, dilation_channels=40),
     150, False, None),
    ('r64_k5', cfg_with(TINY, batch_size=1, residual_channels=64,
                        dilation_channels=48, filter_width=5), 90, False, None),
    ('scalar_r64_d40', cfg_with(TINY, batch_size=2, scalar_input=True,
                                initial_filter_width=4, residual_channels=64,
                                dilation_channels=40), 70, False, None),
    ('r40_k3_nobias', cfg_with(TINY, batch_size=1, residual_channels=40,
                               dilation_channels=24, filter_width=3,
                               use_biases=False), 70, False, None),
    ('k3', cfg_with(TINY, batch_size=2, filter_width=3), 90, False, None),
    ('k3_mid_gc', cfg_with(MID, batch_size=2, filter_width=3,
                           global_condition_channels=4,
                           global_condition_cardinality=5), 300, True, None),
    ('k4_T_lt_shift', cfg_with(TINY, batch_size=2, filter_width=4), 11, False,
     None),
    ('default', cfg_with(DEFAULT, batch_size=1), 1500, False, None),
    ('default_gc', cfg_with(DEFAULT, batch_size=2,
                            global_condition_channels=32,
                            global_condition_cardinality=377), 700, True, None),
    # longer than the receptive field (5117) of the default stack
    ('default_T5200', cfg_with(DEFAULT, batch_size=1), 5200, False, None),
    ('default_gc_T5200', cfg_with(DEFAULT, batch_size=1,
                                  global_condition_channels=32,
                                  global_condition_cardinality=377), 5200, True,
     None),
]


def _stack_case(c):
    """the cases the persistent stack launches / per-layer kernels of the
    default-width model cover: <= 32 residual / dilation channels, two taps"""
    return max(c[1]['residual_channels'], c[1]['dilation_channels']) <= 32 and \
        c[1]['filter_width'] == 2


# kernel path of the residual stack (model.py:236-330 x L):
#   auto     -- the library's choice for the shape (wn_stack_tile_rows: every
#               case here is small enough for the 16-row launches
#               stack_fwd16_kernel / stack_bwd16_kernel; wider / more-tap
#               models run the block / generic-tap kernels)
#   rows32   -- the 32-row launches stack_fwd_kernel<2,16> / stack_bwd_kernel<8>
#               forced by the variant word: the kernels bench.py's headline
#               (8 x 16000) is measured on
#   perlayer -- one launch per layer: layer_fwd_kernel / layer_bwd2d_kernel
_PATH_CASES = [pytest.param(*c, 'auto', id=c[0]) for c in CASES] + \
    [pytest.param(*c, path, id='%s-%s' % (c[0], path)) for c in CASES
     if _stack_case(c) for path in ('rows32', 'perlayer')]


def _select_path(net, path):
    from wavenet._lib import stack_variant
    if path == 'rows32':
        net.stack_variant = stack_variant(rows=32)
    elif path == 'perlayer':
        net.stack_fwd = net.stack_bwd = False


def _assert_path(hip_lib, net, ws, cfg, B, T, path):
    """the launches the test MEANT to compare with the oracle are the ones
    that ran (the launch decisions of WaveNetModel, read back)"""
    if path == 'rows32':
        assert ws.stack_rows == 32 and ws.stack_bwd and net._stack_bwd_ok()
        assert hip_lib.wn_stack_tile_rows(B, T, ws.stack_variant) == 32
    elif path == 'perlayer':
        assert not net.stack_fwd and not net._stack_bwd_ok()
    elif _stack_case((None, cfg)):
        assert ws.stack_rows == hip_lib.wn_stack_tile_rows(B, T, 0)


@pytest.mark.parametrize('name,cfg,T,gc,l2,path', _PATH_CASES)
def test_loss_and_gradients_vs_oracle(hip_lib, name, cfg, T, gc, l2, path):
    B = cfg['batch_size']
    net, var = build_pair(cfg)
    _select_path(net, path)
    rng = np.random.default_rng(7)
    audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
    ids = rng.integers(0, cfg['global_condition_cardinality'], B) if gc \
        else None
    loss = net.loss(audio, ids, l2)
    torch.cuda.synchronize()
    # float64 oracle; at the (measure-zero) ReLU kinks it takes the device's
    # side after checking that the two differ only within forward rounding
    ref_loss, ref_g, c, flips = oracle_grads_at_device_kinks(
        net, cfg, var, audio, ids, l2)
    tag = name if path == 'auto' else '%s-%s' % (name, path)
    _ERRLOG.setdefault('relu_flips', {})[tag] = (flips, 0)
    ws = list(net._ws.values())[0]
    _assert_path(hip_lib, net, ws, cfg, B, T, path)
    # codes: bit exact
    assert np.array_equal(ws.q.cpu().numpy().reshape(B, T),
                          O.mu_law_encode(audio, cfg['quantization_channels']))
    assert abs(float(loss) - ref_loss) < TOL
    check_grads(net, ref_g, tag=name, log_tag=tag)
    if _stack_case((None, cfg)):
        check_planes(net, cfg, c, B, T)
    # forward-only path gives the same loss and identical logits to the oracle
    loss2 = net.loss(audio, ids, l2, backward=False)
    assert abs(float(loss2) - ref_loss) < TOL
    logits = ws.logits.cpu().numpy().reshape(B, T, -1)
    assert np.abs(logits - c['logits']).max() < TOL


# every case of at most 32 residual / dilation channels in the 6-product mode,
# three of them in the 9-product mode as well
_SPLIT_CASES = [pytest.param(*c, 'bf16x6', id=c[0] + '-bf16x6') for c in CASES
                if max(c[1]['residual_channels'], c[1]['dilation_channels']) <= 32] + \
               [pytest.param(*c, 'bf16x9', id=c[0] + '-bf16x9') for c in CASES
                if c[0] in ('mid', 'rp_l2', 'default')]


@pytest.mark.parametrize('name,cfg,T,gc,l2,mode', _SPLIT_CASES)
def test_split_bf16_gemm_mode_parity(hip_lib, name, cfg, T, gc, l2, mode):
    """Opt-in `gemm_mode` (NN / TN GEMM products rebuilt from three exact bf16
    pieces per operand on bf16 MFMA, fp32 accumulation): test_loss_and_
    gradients_vs_oracle's rule against the float64 oracle, unchanged, on every
    case the default-width kernels run."""
    B = cfg['batch_size']
    net, var = build_pair(cfg)
    net.gemm_mode = mode
    rng = np.random.default_rng(7)
    audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
    ids = rng.integers(0, cfg['global_condition_cardinality'], B) if gc \
        else None
    loss = net.loss(audio, ids, l2)
    ref_loss, ref_g, c, _ = oracle_grads_at_device_kinks(net, cfg, var, audio,
                                                        ids, l2)
    # (the split kernels take contractions that are multiples of 16; a case
    # without any runs the fp32 kernels, which is what the mode promises)
    if cfg['skip_channels'] % 16 == 0 and T * B >= 1:
        assert net._wsplit, 'split path not taken'
    assert abs(float(loss) - ref_loss) < TOL
    check_grads(net, ref_g)
    # (the backward pass left dlogits in ws.logits: forward only for the logits)
    loss2 = net.loss(audio, ids, l2, backward=False)
    assert abs(float(loss2) - ref_loss) < TOL
    ws = list(net._ws.values())[0]
    logits = ws.logits.cpu().numpy().reshape(B, T, -1)
    assert np.abs(logits - c['logits']).max() < TOL


def test_launch_plan_replay_tracks_new_inputs(hip_lib):
    """Steps 1 (eager), 2 (recorded) and 3+ (replayed launch plan) on the SAME
    workspace with different audio, different GC ids (fresh tensors every
    step) and a weight update in between must each match the oracle: nothing a
    plan holds may point at a caller's temporary or a stale value."""
    cfg = cfg_with(MID, batch_size=2, global_condition_channels=4,
                   global_condition_cardinality=5)
    net, var = build_pair(cfg)
    assert net.use_launch_plans
    rng = np.random.default_rng(21)
    for step in range(5):
        audio = rng.uniform(-1, 1, (2, 200)).astype(np.float32)
        ids = rng.integers(0, 5, 2)
        if step == 3:   # perturb the weights in place (same flat buffer)
            with torch.no_grad():
                net.params.mul_(1.01)
            var = tree_to_numpy(net.variables)
        ref_loss, ref_g = O.loss_and_grads(cfg, var, audio, ids,
                                           dtype=np.float64)
        loss = net.loss(torch.from_numpy(audio).cuda(),
                        torch.as_tensor(ids).cuda())
        assert abs(float(loss) - ref_loss) < TOL, step
        check_grads(net, ref_g)
    ws = list(net._ws.values())[0]
    assert any(isinstance(p, list) and len(p) > 10 for p in ws.plans.values())
    # switching a flag that changes the launch sequence takes a new plan
    net.stack_fwd = net.stack_bwd = False          # one launch per layer
    loss = net.loss(audio, ids)
    assert abs(float(loss) - ref_loss) < TOL
    check_grads(net, ref_g)
    # and the eager path agrees
    net.use_launch_plans = False
    loss = net.loss(audio, ids)
    assert abs(float(loss) - ref_loss) < TOL
    check_grads(net, ref_g)


def test_generic_tap_kernels_at_k2(hip_lib):
    """The generic-filter-width kernels forced on a K = 2 model must give the
    same loss / gradients as the oracle (and hence the tuned kernels)."""
    cfg = cfg_with(MID, batch_size=2)
    net, var = build_pair(cfg)
    net.generic_layers = True
    audio = np.random.default_rng(5).uniform(-1, 1, (2, 333)).astype(np.float32)
    ref_loss, ref_g = O.loss_and_grads(cfg, var, audio, dtype=np.float64)
    loss = net.loss(audio)
    assert abs(float(loss) - ref_loss) < TOL
    check_grads(net, ref_g)


@pytest.mark.parametrize('case', ['r64', 'r48_d40_gc', 'r96_d128'])
def test_wide_models_fused_gate_gradients_bitwise(hip_lib, case):
    """Channel-block models: dz = dZ + dx' Wd^T and the gate gradients in one
    launch (wn_dense_planes_gate, default) against the two launches it
    replaces (wn_dense_planes, then the elementwise pass that reads dz back):
    same operations in the same order, so loss and every gradient are
    bitwise equal; 64-channel models run the all-pairs weight-gradient kernel
    in both."""
    name, cfg, T, gc, l2 = [c for c in CASES if c[0] == case][0]
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    assert a.wide_fuse_gate
    b.wide_fuse_gate = False
    B = cfg['batch_size']
    audio = np.random.default_rng(7).uniform(-1, 1, (B, T)).astype(np.float32)
    ids = np.arange(B, dtype=np.int32) % cfg['global_condition_cardinality'] if gc else None
    la, lb = a.loss(audio, ids), b.loss(audio, ids)
    torch.cuda.synchronize()
    assert float(la) == float(lb)
    assert torch.equal(a.grads, b.grads)


def test_xent_quirk_switch(hip_lib):
    cfg = cfg_with(TINY, batch_size=2)
    net, var = build_pair(cfg)
    audio = np.random.default_rng(1).uniform(-1, 1, (2, 33)).astype(np.float32)
    for quirk in (True, False):
        net.tf_xent_zero_label_quirk = quirk
        ref_loss, ref_g = O.loss_and_grads(
            cfg, var, audio, dtype=np.float64, tf_xent_zero_label_quirk=quirk)
        loss = net.loss(audio)
        assert abs(float(loss) - ref_loss) < TOL
        check_grads(net, ref_g)


def test_l2_bias_name_quirk_switch(hip_lib):
    cfg = cfg_with(TINY, batch_size=2)
    net, var = build_pair(cfg)
    audio = np.random.default_rng(1).uniform(-1, 1, (2, 33)).astype(np.float32)
    for quirk in (True, False):
        net.tf_bias_name_quirk = quirk
        ref_loss, ref_g = O.loss_and_grads(cfg, var, audio, l2=0.05,
                                           dtype=np.float64,
                                           tf_bias_name_quirk=quirk)
        loss = net.loss(audio, None, 0.05)
        assert abs(float(loss) - ref_loss) < TOL
        check_grads(net, ref_g)


GOLD_CFG = {
    'tiny': cfg_with(TINY, batch_size=2),
    'tiny_nobias': cfg_with(TINY, batch_size=1, use_biases=False),
    'tiny_gc': cfg_with(TINY, batch_size=3, global_condition_channels=4,
                        global_condition_cardinality=5),
    'tiny_rp_l2': cfg_with(TINY, batch_size=2, residual_postproc=True),
    'tiny_noquirk': cfg_with(TINY, batch_size=2),
}


@pytest.mark.parametrize('name', sorted(GOLD_CFG))
def test_golden_fixtures(hip_lib, name):
    from wavenet import WaveNetModel
    z = np.load(os.path.join(GOLD, 'stack_cases.npz'))
    c = {k.split('/', 1)[1]: z[k] for k in z.files if k.startswith(name + '/')}
    cfg = GOLD_CFG[name]
    net = WaveNetModel(**model_kwargs(cfg))
    with torch.no_grad():
        for i, (n, v) in enumerate(net.named_variables()):
            v.copy_(torch.from_numpy(c['w%03d' % i]).to(v.device))
    net.tf_xent_zero_label_quirk = bool(c['quirk'])
    l2 = float(c['l2']) if 'l2' in c else None
    loss = net.loss(c['audio'], c.get('ids'), l2)
    assert abs(float(loss) - float(c['loss'])) < TOL
    ws = list(net._ws.values())[0]
    B = cfg['batch_size']
    for i, (n, g) in enumerate(net.named_variables(net.gradients)):
        ref = c['g%03d' % i]
        assert np.abs(g.cpu().numpy() - ref).max() <= \
            TOL * max(1.0, np.abs(ref).max()), n


@pytest.mark.parametrize('kind,lr,mom', [('adam', 1e-3, 0.9),
                                         ('sgd', 0.02, 0.95),
                                         ('rmsprop', 1e-3, 0.9)])
def test_optimizer_trajectory_vs_oracle(hip_lib, kind, lr, mom):
    """Three full training steps: HIP (loss + fused update) vs oracle
    (float64 loss/grads + TF-rule optimizer)."""
    from wavenet import optimizer_factory
    cfg = cfg_with(TINY, batch_size=2)
    net, var = build_pair(cfg)
    audio = np.random.default_rng(2).uniform(-1, 1, (2, 40)).astype(np.float32)
    opt = optimizer_factory[kind](learning_rate=lr, momentum=mom)
    ref_opt = O.TFOptimizer(kind, lr, mom)
    for step in range(3):
        ref_loss, ref_g = O.loss_and_grads(cfg, var, audio, dtype=np.float64)
        O.unpack_into(var, ref_opt.apply(O.pack(var), O.pack(ref_g)))
        loss = net.loss(audio)
        opt.minimize(loss)
        assert abs(float(loss) - ref_loss) < TOL, step
    got = tree_to_numpy(net.variables)
    for (n, a), (_, b) in zip(flat_named(got), flat_named(var)):
        assert np.abs(a - b).max() < 2e-5, n


def test_predict_proba_vs_oracle(hip_lib):
    # test/test_generation.py:19-32 shapes (9 layers, R=D=16, Q=128, S=32)
    cfg = dict(batch_size=1, dilations=[1, 2, 4, 8, 16, 32, 64, 128, 256],
               filter_width=2, residual_channels=16, dilation_channels=16,
               quantization_channels=128, skip_channels=32, use_biases=True)
    net, var = build_pair(cfg)
    np.random.seed(0)
    data = np.random.randint(128, size=1000)
    p = net.predict_proba(data).cpu().numpy()
    assert p.shape == (128,)
    assert np.all((p >= 0) & (p <= 1)) and abs(p.sum() - 1) < 1e-5
    ref = O.predict_proba(cfg, var, data, dtype=np.float64)
    assert np.abs(p - ref).max() < 1e-5


def test_predict_proba_scalar_input(hip_lib):
    # model.py:570-576: codes are decoded back to floats for the scalar net
    cfg = cfg_with(TINY, batch_size=1, scalar_input=True,
                   initial_filter_width=4)
    net, var = build_pair(cfg)
    data = np.random.default_rng(0).integers(0, 16, 60)
    p = net.predict_proba(data).cpu().numpy()
    ref = O.predict_proba(cfg, var, data, dtype=np.float64)
    assert np.abs(p - ref).max() < 1e-5
    with pytest.raises(NotImplementedError, match='Scalar input'):
        net.predict_proba_incremental(3)


def test_incremental_vs_oracle_and_naive(hip_lib):
    """test/test_generation.py:50-72 strengthened: compare over more than the
    receptive field (RF 32 -> 80 steps), against the golden trace, the oracle
    and the naive forward."""
    from wavenet import WaveNetModel
    z = np.load(os.path.join(GOLD, 'incremental.npz'))
    cfg = cfg_with(TINY, batch_size=1)
    net = WaveNetModel(**model_kwargs(cfg))
    with torch.no_grad():
        for i, (n, v) in enumerate(net.named_variables()):
            v.copy_(torch.from_numpy(z['inc/w%03d' % i]).to(v.device))
    wave, probs = z['inc/wave'], z['inc/probs']
    net.reset_generator()
    for i, s in enumerate(wave):
        if i == len(wave) - 1:
            # the reference's final proba op runs WITHOUT push_ops
            p = net.predict_proba_incremental(int(s), push=False).cpu().numpy()
            p_again = net.predict_proba_incremental(int(s)).cpu().numpy()
            assert np.array_equal(p, p_again)
        else:
            p = net.predict_proba_incremental(int(s)).cpu().numpy()
        assert np.abs(p - probs[i]).max() < 1e-5, i
    naive = net.predict_proba(wave).cpu().numpy()
    assert np.abs(naive - z['inc/naive_last']).max() < 1e-5
    assert np.abs(naive - p).max() < 1e-5
    # teacher-forced generate() gives the same trace on the three device paths:
    # the single-workgroup kernel, the multi-CU step kernels (hipGraph) and
    # the persistent multi-CU launch (here: one chain segment, one workgroup
    # per tail stage)
    for multi, persist in ((False, False), (True, False), (True, True)):
        net.fastgen_multi_cu = multi
        net.fastgen_persistent = persist
        out, pr = net.generate(0, seed_samples=wave, return_proba_every=1)
        assert np.array_equal(out.cpu().numpy(), wave)
        assert np.abs(pr.cpu().numpy() - probs[:len(wave) - 1]).max() < 1e-5


@pytest.mark.parametrize('multi,persist', [(False, False), (True, False), (True, True)],
                         ids=['one_wg', 'multi_cu_graph', 'multi_cu_persistent'])
def test_generate_sampling(hip_lib, multi, persist):
    cfg = cfg_with(MID, batch_size=1)
    net, var = build_pair(cfg)
    net.fastgen_multi_cu = multi
    net.fastgen_persistent = persist
    net.fastgen_graph_steps = 50          # exercise graph capture + replay
    a = net.generate(300, seed_samples=[128], seed=11).cpu().numpy()
    b = net.generate(300, seed_samples=[128], seed=11).cpu().numpy()
    c = net.generate(300, seed_samples=[128], seed=12).cpu().numpy()
    assert a.shape == (301,) and a[0] == 128
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= 0 and a.max() < 256
    if multi:
        # the persistent launch and the step kernels draw the SAME samples, also
        # with a temperature and with probabilities returned every third step
        net.fastgen_persistent = not persist
        a2 = net.generate(300, seed_samples=[128], seed=11).cpu().numpy()
        t1, p1 = net.generate(90, seed_samples=[3, 200, 17], temperature=0.7, seed=4,
                              return_proba_every=3)
        net.fastgen_persistent = persist
        t2, p2 = net.generate(90, seed_samples=[3, 200, 17], temperature=0.7, seed=4,
                              return_proba_every=3)
        assert np.array_equal(a, a2)
        assert torch.equal(t1, t2) and p1.shape == p2.shape
        assert float((p1 - p2).abs().max()) < 1e-6
    # drawn samples follow the predicted distribution: teacher-force the drawn
    # sequence through the oracle and compare empirical log-likelihood ranks
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    ll = 0.0
    for i in range(120):
        p = gen.step(int(a[i]))
        ll += np.log(p[a[i + 1]] + 1e-30)
    assert ll / 120 > np.log(1.0 / 256) - 1.0   # far better than uniform-1nat
    # low temperature -> (nearly) argmax decoding
    g = net.generate(40, seed_samples=[128], temperature=1e-3, seed=5)
    g = g.cpu().numpy()
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    for i in range(40):
        p = gen.step(int(g[i]))
        assert g[i + 1] == int(np.argmax(p)) or \
            p[g[i + 1]] > 0.999 * p.max()


@pytest.mark.parametrize('n0', [5, 40, 150])
def test_prime_generator_from_forward_pass(hip_lib, n0):
    """prime_generator (one batch forward pass; the reference's TODO at
    generate.py:199-201) leaves the queues in the state that n0 incremental
    pushes produce: same ring contents, cursor, and next-step distribution;
    also against the float64 oracle.  n0 below / above the largest dilation
    and the receptive field."""
    cfg = cfg_with(MID, batch_size=1)
    net, var = build_pair(cfg)
    rng = np.random.default_rng(n0)
    codes = rng.integers(0, cfg['quantization_channels'], n0).astype(np.int32)
    nxt = int(rng.integers(0, cfg['quantization_channels']))
    net.reset_generator()
    for c in codes:
        net.predict_proba_incremental(int(c))
    st_ref = net._gen['state'].clone()
    cur_ref = net._gen['cursors'].clone()
    p_ref = net.predict_proba_incremental(nxt, push=False).cpu().numpy()
    net.prime_generator(codes)
    assert torch.equal(net._gen['cursors'], cur_ref)
    assert (net._gen['state'] - st_ref).abs().max().item() < 1e-5
    p = net.predict_proba_incremental(nxt, push=False).cpu().numpy()
    assert np.abs(p - p_ref).max() < 1e-5
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    for c in codes:
        gen.step(int(c))
    assert np.abs(p - gen.step(nxt)).max() < 1e-5
    # generate() with a long seed takes the forward-priming path and still
    # returns seed + drawn samples, deterministically
    net.fastgen_prime_forward_min = 16
    a = net.generate(30, seed_samples=codes, seed=3).cpu().numpy()
    b = net.generate(30, seed_samples=codes, seed=3).cpu().numpy()
    assert a.shape == (n0 + 30,) and np.array_equal(a[:n0], codes)
    assert np.array_equal(a, b)


def test_prime_generator_with_global_condition(hip_lib):
    """forward-pass priming with a speaker id: same queues / distribution as
    stepping, and different from another speaker's."""
    cfg = cfg_with(MID, batch_size=1, global_condition_channels=4,
                   global_condition_cardinality=5)
    net, var = build_pair(cfg)
    codes = np.random.default_rng(4).integers(0, 256, 90).astype(np.int32)
    net.reset_generator()
    for c in codes:
        net.predict_proba_incremental(int(c), global_condition=3)
    st_ref = net._gen['state'].clone()
    p_ref = net.predict_proba_incremental(7, global_condition=3,
                                          push=False).cpu().numpy()
    net.prime_generator(codes, global_condition=3)
    assert (net._gen['state'] - st_ref).abs().max().item() < 1e-5
    p = net.predict_proba_incremental(7, global_condition=3,
                                      push=False).cpu().numpy()
    assert np.abs(p - p_ref).max() < 1e-5
    net.prime_generator(codes, global_condition=1)
    p_other = net.predict_proba_incremental(7, global_condition=1,
                                            push=False).cpu().numpy()
    assert np.abs(p_other - p_ref).max() > 1e-4


WIDE_GEN = [
    ('r64', cfg_with(MID, batch_size=1, residual_channels=64,
                     dilation_channels=64), None),
    ('r48_d40_gc', cfg_with(MID, batch_size=1, residual_channels=48,
                            dilation_channels=40, global_condition_channels=4,
                            global_condition_cardinality=5), 3),
    ('r96_d128', cfg_with(TINY, batch_size=1, residual_channels=96,
                          dilation_channels=128), None),
    ('r160_d136', cfg_with(TINY, batch_size=1, residual_channels=160,
                           dilation_channels=136), None),
    # more channels than the generator kernel has threads
    ('r288_d264', cfg_with(TINY, batch_size=1, residual_channels=288,
                           dilation_channels=264), None),
    ('r1024_d520', cfg_with(TINY, batch_size=1, residual_channels=1024,
                            dilation_channels=520, dilations=[1, 2]), None),
    # 32 channels, but more skip / quantization channels or layers than the
    # tuned generator kernels keep in LDS (512 / 512 / 64)
    ('S1024_Q1024', cfg_with(TINY, batch_size=1, skip_channels=1024,
                             quantization_channels=1024), None),
    ('L70_gc', cfg_with(TINY, batch_size=1, dilations=[1, 2, 4, 8, 16] * 14,
                        global_condition_channels=4,
                        global_condition_cardinality=5), 2),
    ('r64_S520_L66', cfg_with(TINY, batch_size=1, residual_channels=64,
                              dilation_channels=40, skip_channels=520,
                              dilations=[1, 2, 4] * 22), None),
]


@pytest.mark.parametrize('name,cfg,gc', WIDE_GEN, ids=[c[0] for c in WIDE_GEN])
def test_fast_generation_above_32_channels(hip_lib, name, cfg, gc):
    """The incremental generator with 33 - 1024 residual / dilation channels,
    or more than 512 skip / quantization channels or 64 layers
    (wn_fastgen_run_wide; the reference's generator has no size limit,
    model.py:444-516): every step of a teacher-forced trace longer than the
    receptive field against the float64 oracle, the no-push peek, forward-pass
    priming, and deterministic sampling through generate()."""
    net, var = build_pair(cfg)
    Q = cfg['quantization_channels']
    rf = sum(cfg['dilations']) + 2
    rng = np.random.default_rng(9)
    wave = rng.integers(0, Q, rf + 25).astype(np.int32)
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    gids = None if gc is None else np.array([gc])
    net.reset_generator()
    worst = 0.0
    for i, c in enumerate(wave):
        p_ref = np.asarray(gen.step(int(c), gc_ids=gids)).reshape(-1)
        if i == len(wave) - 1:
            p = net.predict_proba_incremental(int(c), global_condition=gc,
                                              push=False).cpu().numpy()
            again = net.predict_proba_incremental(
                int(c), global_condition=gc).cpu().numpy()
            assert np.array_equal(p, again)
        else:
            p = net.predict_proba_incremental(
                int(c), global_condition=gc).cpu().numpy()
        worst = max(worst, float(np.abs(p - p_ref).max()))
    assert worst < 1e-5, worst
    # the naive windowed forward agrees too
    naive = net.predict_proba(wave, global_condition=gc).cpu().numpy()
    assert np.abs(naive - p).max() < 1e-5
    # priming from ONE forward pass == stepping through the seed
    st_ref = net._gen['state'].clone()
    cur_ref = net._gen['cursors'].clone()
    net.prime_generator(wave, global_condition=gc)
    assert torch.equal(net._gen['cursors'][:2], cur_ref[:2])
    # (queue entries are activations: relative to their size in deep stacks)
    scale = max(1.0, st_ref.abs().max().item())
    assert (net._gen['state'] - st_ref).abs().max().item() < 1e-5 * scale
    # generate(): teacher-forced trace, then deterministic draws
    out, pr = net.generate(0, seed_samples=wave[:40], return_proba_every=1,
                           global_condition=gc)
    assert np.array_equal(out.cpu().numpy(), wave[:40])
    a = net.generate(60, seed_samples=[Q // 2], seed=11,
                     global_condition=gc).cpu().numpy()
    b = net.generate(60, seed_samples=[Q // 2], seed=11,
                     global_condition=gc).cpu().numpy()
    c2 = net.generate(60, seed_samples=[Q // 2], seed=12,
                      global_condition=gc).cpu().numpy()
    assert a.shape == (61,) and np.array_equal(a, b) and not np.array_equal(a, c2)
    assert a.min() >= 0 and a.max() < Q


@pytest.mark.parametrize('gc,ch', [(None, 64), (2, 64), (None, 96), (2, 160)])
def test_wide_cooperative_generator_equals_single_workgroup(hip_lib, gc, ch):
    """More than 32 channels: wn_fastgen_run_wide's cooperative launch (skip sum and
    post-processing mat-vecs on other workgroups, hand-over words) against its
    single workgroup: the same drawn samples, with a temperature, probabilities
    to rounding (the skip sum adds in another order), queues and cursors that
    continue on either path."""
    kw = dict(global_condition_channels=4, global_condition_cardinality=5) if gc is not None else {}
    cfg = cfg_with(MID, batch_size=1, residual_channels=ch, dilation_channels=ch - 8,
                   skip_channels=128, **kw)
    net, var = build_pair(cfg)
    assert hip_lib.wn_fastgen_wide_coop_bytes(len(cfg['dilations']), 64, 128, 256) > 0
    assert hip_lib.wn_fastgen_wide_coop_bytes(len(cfg['dilations']), 64, 520, 256) == 0
    res = []
    for coop in (True, False):
        net.fastgen_wide_coop = coop
        net.reset_generator()
        o, p = net.generate(200, seed_samples=[128, 7, 250], seed=5, temperature=0.9,
                            global_condition=gc, return_proba_every=3)
        assert ('coop' in net._gen and net._gen['coop'] is not None) or not coop
        res.append((o.cpu().numpy(), p.cpu().numpy(), net._gen['state'].clone(),
                    net._gen['cursors'].clone()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-6
    assert torch.equal(res[0][3][:2], res[1][3][:2])
    scale = max(1.0, res[1][2].abs().max().item())
    assert (res[0][2] - res[1][2]).abs().max().item() < 1e-5 * scale
    # a run continued on the other path
    net.fastgen_wide_coop = True
    more_a = net.continue_generation(50, int(res[1][0][-1]), 1.0, gc, 6).cpu().numpy()
    assert more_a.shape == (50,) and more_a.min() >= 0 and more_a.max() < 256


def test_wide_cooperative_launch_failure_restores_state_and_falls_back(hip_lib, monkeypatch):
    """A cooperative launch of the wide generator that reports an expired
    hand-over wait (workgroups not all resident: invisible to the launch-time
    occupancy check) after it has rewritten queues, cursors and samples: the
    host restores its snapshot, warns, repeats the run on the single workgroup
    -- the samples of a model that took the single workgroup from the start --
    and does not try the cooperative launch again on this generator."""
    from wavenet import _lib
    cfg = cfg_with(MID, batch_size=1, residual_channels=64, dilation_channels=64,
                   skip_channels=128)
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    b.fastgen_wide_coop = False
    lib = _lib.load()
    real = lib.wn_fastgen_run_wide
    coop_calls = []

    def failing(*args):
        code = real(*args)
        if args[-2]:                      # a cooperative launch: report a failed wait
            torch.cuda.synchronize()
            a._gen['coop'][12] = 1
            coop_calls.append(code)
        return code
    monkeypatch.setattr(lib, 'wn_fastgen_run_wide', failing)
    with pytest.warns(UserWarning, match='state restored'):
        out_a = a.generate(120, seed_samples=[128, 3, 77], seed=9).cpu().numpy()
    out_b = b.generate(120, seed_samples=[128, 3, 77], seed=9).cpu().numpy()
    assert coop_calls == [0] and a._gen_launch_failed['coop']
    assert np.array_equal(out_a, out_b)
    assert a._gen['steps'] == b._gen['steps']
    more_a = a.continue_generation(60, int(out_a[-1]), seed=4).cpu().numpy()
    more_b = b.continue_generation(60, int(out_b[-1]), seed=4).cpu().numpy()
    assert coop_calls == [0]              # not tried again on this generator
    assert np.array_equal(more_a, more_b)
    assert torch.equal(a._gen['state'], b._gen['state'])


def test_unsupported_configs_raise(hip_lib):
    from wavenet import WaveNetModel
    for kw in (dict(filter_width=65), dict(residual_channels=1025),
               dict(dilation_channels=1200)):
        cfg = cfg_with(TINY, batch_size=1, **kw)
        net = WaveNetModel(**model_kwargs(cfg))
        with pytest.raises(NotImplementedError):
            net.loss(np.zeros(16, np.float32))
    net = WaveNetModel(**model_kwargs(cfg_with(
        TINY, batch_size=2, global_condition_channels=4,
        global_condition_cardinality=5)))
    with pytest.raises(ValueError):
        net.loss(np.zeros((2, 16), np.float32), np.array([1, 2, 3]))
    with pytest.raises(ValueError):
        net.loss(np.zeros((2, 16), np.float32))


def test_workspace_reserve_and_owner_per_kind(hip_lib):
    """net.reserve(): growing forward-only inputs are views of one allocation
    (same plane address, no new owner); without a reservation the capacity
    grows geometrically; a training workspace and a forward-only one of
    another length do not evict each other (the recorded launch plans of the
    training workspace survive)."""
    cfg = cfg_with(MID, batch_size=1)
    net, var = build_pair(cfg)
    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, 400)
    owner = net.reserve(1, 300)
    base = owner.X.data_ptr()
    for T in (7, 120, 300):
        p = net.predict_proba(data[:T]).cpu().numpy()
        ws = net._ws[(1, T, False)]
        assert ws.X.data_ptr() == base
        assert np.abs(p - O.predict_proba(cfg, var, data[:T],
                                          dtype=np.float64)).max() < 1e-5
    owners = lambda: [w for w in net._ws.values() if w.capacity == w.N]
    assert len(owners()) == 1
    net.predict_proba(data[:301])           # past the reservation: grows x2
    assert max(w.T for w in owners()) >= 600 and len(owners()) == 1
    # training owner next to the eval owner
    audio = rng.uniform(-1, 1, (1, 150)).astype(np.float32)
    for _ in range(3):
        net.loss(audio)
    tr = net._ws[(1, 150, True)]
    assert any(isinstance(pl, list) for pl in tr.plans.values())
    net.predict_proba(data[:350])
    net.predict_proba(data[:800])           # new eval owner
    assert net._ws[(1, 150, True)] is tr    # training buffers + plans kept
    l1 = float(net.loss(audio))
    ref, _ = O.loss_and_grads(cfg, var, audio, dtype=np.float64)
    assert abs(l1 - ref) < TOL


def test_random_configurations_vs_oracle(hip_lib):
    """tools/model_fuzz.py: random constructor arguments (dilations, channel
    counts up to 96, filter widths 2 - 9, skip / quantization channels,
    biases, global conditioning, scalar input, residual post-processing, L2),
    each against the float64 oracle: loss, every gradient, and a generator
    trace where the reference's generator supports the configuration."""
    import subprocess
    import sys
    from util import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'model_fuzz.py'),
                        '24', '7'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert '0 of 24 cases failed' in r.stdout


@pytest.mark.parametrize('Q', [64, 320, 512])
def test_persistent_draw_other_quantization_channels(hip_lib, Q):
    """The persistent launch's draw workgroup away from the reference's Q = 256
    (fewer values than computing threads; two values per thread, 5 and 8 logits
    per lane in the max pass): the samples of the step kernels, with and without
    a temperature, and their probabilities."""
    cfg = cfg_with(MID, batch_size=1, quantization_channels=Q)
    net, var = build_pair(cfg)
    for temp in (1.0, 0.8):
        outs = []
        for persist in (0, 1):
            net.fastgen_persistent = persist
            o, p = net.generate(150, seed_samples=[Q // 2, 3], seed=9, temperature=temp,
                                return_proba_every=2)
            outs.append((o.cpu().numpy(), p.cpu().numpy()))
        assert np.array_equal(outs[0][0], outs[1][0]), temp
        assert np.abs(outs[0][1] - outs[1][1]).max() < 1e-6
        assert outs[1][0].min() >= 0 and outs[1][0].max() < Q
        assert len(np.unique(outs[1][0][2:])) > 8          # not stuck on one code


def test_persistent_generator_short_runs_and_state_continuity(hip_lib):
    """wn_fastgen_persist at the edges: runs of 1, 2, 3 and 17 steps (the helper
    waves' look-ahead loop is empty at 1), with global conditioning, a
    temperature and probabilities every step, draw the samples of the step
    kernels; and the queues / cursors it leaves continue on either path."""
    cfg = cfg_with(MID, batch_size=1, global_condition_channels=4,
                   global_condition_cardinality=5)
    net, var = build_pair(cfg)
    for n in (1, 2, 3, 17):
        outs = []
        for persist in (0, 1):
            net.fastgen_persistent = persist
            o, p = net.generate(n, seed_samples=[5, 9], seed=3, temperature=0.9,
                                global_condition=2, return_proba_every=1)
            outs.append((o.cpu().numpy(), p.cpu().numpy()))
        assert np.array_equal(outs[0][0], outs[1][0]), n
        assert np.abs(outs[0][1] - outs[1][1]).max() < 1e-6
    res = []
    for first, then in ((1, 0), (0, 0), (0, 1), (1, 1)):
        net.reset_generator()
        net.fastgen_persistent = first
        a = net.generate(40, seed_samples=[7], seed=1, global_condition=1).cpu().numpy()
        net.fastgen_persistent = then
        b = net.continue_generation(40, int(a[-1]), 1.0, 1, 2).cpu().numpy()
        res.append(np.concatenate([a, b]))
    for r in res[1:]:
        assert np.array_equal(res[0], r)
