"""Host-side logic that needs no GPU: variable names / shapes mirror the
reference (model.py:118-225), config schema, DP sharding helpers, error
behaviour of compute entry points without a device."""
import json
import os

import numpy as np
import pytest
import torch

from util import O, ROOT, DEFAULT, TINY, cfg_with, model_kwargs


def test_params_json_schema():
    # wavenet_params.json:1-17 -- the ten keys the reference reads by name
    p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
    for k in ['filter_width', 'sample_rate', 'dilations', 'residual_channels',
              'dilation_channels', 'quantization_channels', 'skip_channels',
              'use_biases', 'scalar_input', 'initial_filter_width',
              'residual_postproc']:
        assert k in p
    assert p['dilations'] == [2 ** i for i in range(10)] * 5
    assert (p['residual_channels'], p['dilation_channels'],
            p['skip_channels'], p['quantization_channels']) == (32, 32, 512, 256)


def test_variables_mirror_reference_layout(hip_lib):
    from wavenet import WaveNetModel
    cfg = cfg_with(DEFAULT, batch_size=1, global_condition_channels=32,
                   global_condition_cardinality=377)
    net = WaveNetModel(device='cpu', **model_kwargs(cfg))
    ref = O.create_variables(cfg, dtype=np.float32)
    got = net.variables
    assert set(got) == set(ref)
    assert got['embeddings']['gc_embedding'].shape == (377, 32)
    assert tuple(got['causal_layer']['filter'].shape) == (2, 256, 32)
    assert len(got['dilated_stack']) == 50
    for a, b in zip(got['dilated_stack'], ref['dilated_stack']):
        assert set(a) == set(b)
        for k in a:
            assert tuple(a[k].shape) == b[k].shape, k
    for k, v in ref['postprocessing'].items():
        assert tuple(got['postprocessing'][k].shape) == v.shape
    # parameter count of SURVEY 8a row 6 (1 630 432 with GC 32x377)
    n = sum(int(np.prod(v.shape)) for _, v in net.named_variables())
    assert n == 1630432
    names = [n for n, _ in net.named_variables()]
    assert names[0] == 'wavenet/embeddings/gc_embedding'
    assert 'wavenet/dilated_stack/layer49/gc_filter' in names
    assert 'wavenet/postprocessing/postprocess2_bias' in names


def test_small_channel_views_and_init(hip_lib):
    from wavenet import WaveNetModel
    cfg = cfg_with(TINY, batch_size=1)
    net = WaveNetModel(device='cpu', seed=3, **model_kwargs(cfg))
    v = net.variables['dilated_stack'][0]
    assert tuple(v['filter'].shape) == (2, 8, 8)
    assert tuple(v['skip'].shape) == (1, 8, 16)
    lim = np.sqrt(6.0 / (2 * 8 + 2 * 8))      # xavier, model.py:10
    assert float(v['filter'].abs().max()) <= lim
    assert float(v['filter_bias'].abs().max()) == 0.0   # zeros, model.py:27
    # padding lanes of the flat buffer stay zero
    used = torch.zeros_like(net.params)
    for _, t in net.named_variables(net._views(used)):
        t.fill_(1)
    assert float((net.params * (1 - used)).abs().max()) == 0.0
    # state_dict round trip by reference names
    sd = net.state_dict()
    net2 = WaveNetModel(device='cpu', seed=9, **model_kwargs(cfg))
    net2.load_state_dict(sd)
    assert torch.equal(net.params, net2.params)


def test_filter_width_3_shapes(hip_lib):
    from wavenet import WaveNetModel
    cfg = cfg_with(TINY, batch_size=1, filter_width=3)
    net = WaveNetModel(device='cpu', **model_kwargs(cfg))
    v = net.variables['dilated_stack'][0]
    assert tuple(v['filter'].shape) == (3, 8, 8)
    assert tuple(v['gate'].shape) == (3, 8, 8)
    assert tuple(v['dense'].shape) == (1, 8, 8)
    # causal layer keeps filter_width taps on the one-hot input (model.py:148)
    assert tuple(net.variables['causal_layer']['filter'].shape) == (3, 16, 8)


def test_scalar_input_causal_filter_shape(hip_lib):
    from wavenet import WaveNetModel
    cfg = cfg_with(TINY, batch_size=1, scalar_input=True,
                   initial_filter_width=4)
    net = WaveNetModel(device='cpu', **model_kwargs(cfg))
    assert tuple(net.variables['causal_layer']['filter'].shape) == (4, 1, 8)


def test_identity_embedding_when_square(hip_lib):
    from wavenet import WaveNetModel
    cfg = cfg_with(TINY, batch_size=3, global_condition_channels=3,
                   global_condition_cardinality=3)
    net = WaveNetModel(device='cpu', **model_kwargs(cfg))
    assert torch.equal(net.variables['embeddings']['gc_embedding'],
                       torch.eye(3))


def test_compute_without_gpu_raises(hip_lib):
    from wavenet import WaveNetModel, _lib, mu_law_encode
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    net = WaveNetModel(device='cpu', **model_kwargs(cfg_with(TINY, batch_size=1)))
    with pytest.raises(_lib.WaveNetHipError):
        net.loss(np.zeros(16, np.float32))
    with pytest.raises(_lib.WaveNetHipError):
        mu_law_encode(np.zeros(4, np.float32), 256)
    with pytest.raises(_lib.WaveNetHipError):
        WaveNetModel(**model_kwargs(cfg_with(TINY, batch_size=1)))


def test_incremental_refusals_match_reference(hip_lib):
    # model.py:597-603: raised before anything touches the device
    from wavenet import WaveNetModel
    n1 = WaveNetModel(device='cpu', **model_kwargs(
        cfg_with(TINY, batch_size=1, filter_width=3)))
    with pytest.raises(NotImplementedError, match='filter_width > 2'):
        n1.predict_proba_incremental(3)
    n2 = WaveNetModel(device='cpu', **model_kwargs(
        cfg_with(TINY, batch_size=1, scalar_input=True)))
    with pytest.raises(NotImplementedError, match='Scalar input'):
        n2.predict_proba_incremental(3)


def test_optimizer_factory_names():
    from wavenet import optimizer_factory
    assert sorted(optimizer_factory) == ['adam', 'rmsprop', 'sgd']
    a = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
    assert a.eps == 1e-4                       # ops.py:7-8
    r = optimizer_factory['rmsprop'](learning_rate=1e-3, momentum=0.5)
    assert (r.eps, r.mom, r.decay) == (1e-5, 0.5, 0.9)   # ops.py:16-19
    s = optimizer_factory['sgd'](learning_rate=0.02, momentum=0.95)
    assert (s.lr, s.mom) == (0.02, 0.95)
    with pytest.raises(ValueError):
        a.minimize(torch.zeros(()))


def test_shard_range():
    from wavenet import parallel
    assert parallel.shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        parallel.shard_range(10, 0, 4)


def test_bench_synthetic_clips():
    import bench
    a = bench.synth_audio(2, 16000, first_clip=0)
    b = bench.synth_audio(1, 16000, first_clip=0)
    assert a.dtype == np.float32 and a.shape == (2, 16000)
    assert np.array_equal(a[0], b[0]) and np.abs(a).max() <= 1.0
    assert len(np.unique(O.mu_law_encode(a, 256))) > 100
    assert bench.host_cores() >= 1
