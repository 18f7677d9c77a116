"""oracle/wavenet_oracle.py (closed forms + analytic backward) against the
independent op-for-op PyTorch-CPU restatement of the reference graph
(oracle/torch_graph.py, autograd) in float64, and against the committed
golden fixtures."""
import os

import numpy as np
import pytest
import torch

from util import O, TINY, cfg_with
from oracle import torch_graph as TG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CASES = [
    ('bias', cfg_with(TINY, batch_size=2), 37, False, None),
    ('nobias', cfg_with(TINY, batch_size=1, use_biases=False), 5, False, None),
    ('gc', cfg_with(TINY, batch_size=3, global_condition_channels=4,
                    global_condition_cardinality=5), 37, True, None),
    ('rp_l2', cfg_with(TINY, batch_size=2, residual_postproc=True), 20, False,
     0.01),
    ('k3', cfg_with(TINY, batch_size=2, filter_width=3), 37, False, None),
    ('scalar', cfg_with(TINY, batch_size=2, scalar_input=True,
                        initial_filter_width=4), 37, False, None),
    ('t_lt_d', cfg_with(TINY, batch_size=2), 3, False, None),
]


@pytest.mark.parametrize('name,cfg,T,gc,l2', CASES,
                         ids=[c[0] for c in CASES])
def test_loss_and_grads_match_autograd(name, cfg, T, gc, l2):
    torch.set_num_threads(2)
    B = cfg['batch_size']
    var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
    rng = np.random.default_rng(1)
    audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
    ids = rng.integers(0, cfg['global_condition_cardinality'], B) if gc \
        else None
    L, g = O.loss_and_grads(cfg, var, audio, ids, l2=l2, dtype=np.float64,
                            tf_xent_zero_label_quirk=False)
    tv = TG.to_torch(var, torch.float64, True)
    q = torch.tensor(O.mu_law_encode(
        audio, cfg['quantization_channels']).astype(np.int64))
    names = O.flatten_variables(tv)
    tl = TG.loss(cfg, tv, q, audio=torch.tensor(audio),
                 gc_ids=None if ids is None else torch.tensor(ids), l2=l2,
                 names=names)
    tl.backward()
    assert abs(float(tl.detach()) - L) < 1e-10
    for (n, ga), (_, tp) in zip(O.flatten_variables(g), names):
        tg = tp.grad.numpy() if tp.grad is not None else np.zeros_like(ga)
        assert np.abs(ga - tg).max() < 1e-10, n


def test_xent_quirk_only_touches_last_rows():
    cfg = cfg_with(TINY, batch_size=2)
    var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
    audio = np.random.default_rng(1).uniform(-1, 1, (2, 20)).astype(np.float32)
    l1, g1 = O.loss_and_grads(cfg, var, audio, dtype=np.float64,
                              tf_xent_zero_label_quirk=True)
    l2, g2 = O.loss_and_grads(cfg, var, audio, dtype=np.float64,
                              tf_xent_zero_label_quirk=False)
    assert l1 == l2
    d = np.abs(O.pack(g1) - O.pack(g2)).max()
    assert d > 0      # the zero-label rows do back-propagate under the quirk


def test_incremental_equals_naive():
    cfg = cfg_with(TINY, batch_size=1)
    var = O.create_variables(cfg, seed=2, dtype=np.float64, bias_scale=0.1)
    rng = np.random.default_rng(3)
    wave = rng.integers(0, 16, 70)
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    for i, s in enumerate(wave):
        p = gen.step(int(s))
        if i in (0, 5, 31, 32, 33, 69):
            ref = O.predict_proba(cfg, var, wave[:i + 1], dtype=np.float64)
            assert np.allclose(p, ref, atol=1e-6), i


def test_incremental_refuses_like_reference():
    # model.py:597-603
    with pytest.raises(NotImplementedError):
        O.IncrementalGenerator(cfg_with(TINY, batch_size=1, filter_width=3),
                               {}, np.float64)
    with pytest.raises(NotImplementedError):
        O.IncrementalGenerator(cfg_with(TINY, batch_size=1, scalar_input=True),
                               {}, np.float64)


def _load_case(z, name):
    ks = sorted(k for k in z.files if k.startswith(name + '/'))
    return {k.split('/', 1)[1]: z[k] for k in ks}


CASE_CFG = {
    'tiny': cfg_with(TINY, batch_size=2),
    'tiny_nobias': cfg_with(TINY, batch_size=1, use_biases=False),
    'tiny_gc': cfg_with(TINY, batch_size=3, global_condition_channels=4,
                        global_condition_cardinality=5),
    'tiny_rp_l2': cfg_with(TINY, batch_size=2, residual_postproc=True),
    'tiny_noquirk': cfg_with(TINY, batch_size=2),
}


@pytest.mark.parametrize('name', sorted(CASE_CFG))
def test_oracle_reproduces_golden(name):
    z = np.load(os.path.join(GOLD, 'stack_cases.npz'))
    c = _load_case(z, name)
    cfg = CASE_CFG[name]
    var = O.create_variables(cfg, seed=0, dtype=np.float64)
    flat = O.flatten_variables(var)
    for i, (n, a) in enumerate(flat):
        assert n == str(c['names'][i])
        a[...] = c['w%03d' % i]
    ids = c.get('ids')
    l2 = float(c['l2']) if 'l2' in c else None
    L, g = O.loss_and_grads(cfg, var, c['audio'], ids, l2=l2,
                            dtype=np.float64,
                            tf_xent_zero_label_quirk=bool(c['quirk']))
    assert abs(L - float(c['loss'])) < 1e-5
    for i, (n, ga) in enumerate(O.flatten_variables(g)):
        ref = c['g%03d' % i]
        assert np.abs(ga - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), n


def test_tf_optimizer_rules():
    """TensorFlow-0.10 update rules restated by hand."""
    w0 = np.array([1.0, -2.0, 0.5])
    g = np.array([0.1, -0.2, 0.3])
    # Momentum: acc = m*acc + g ; w -= lr*acc
    o = O.TFOptimizer('sgd', 0.1, 0.9)
    w = o.apply(w0.copy(), g)
    assert np.allclose(w, w0 - 0.1 * g)
    w = o.apply(w, g)
    assert np.allclose(w, w0 - 0.1 * g - 0.1 * (0.9 * g + g))
    # Adam, eps 1e-4 un-corrected: step 1 -> lr*sqrt(1-b2)/(1-b1) * m/(sqrt v + eps)
    o = O.TFOptimizer('adam', 1e-3)
    w = o.apply(w0.copy(), g)
    m, v = 0.1 * g, 0.001 * g * g
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(w, w0 - lr_t * m / (np.sqrt(v) + 1e-4))
    # RMSProp: ms starts at ONE
    o = O.TFOptimizer('rmsprop', 1e-3, 0.9)
    w = o.apply(w0.copy(), g)
    ms = 0.9 * 1.0 + 0.1 * g * g
    assert np.allclose(w, w0 - 1e-3 * g / np.sqrt(ms + 1e-5))
