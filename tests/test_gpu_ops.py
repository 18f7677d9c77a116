"""GPU parity of the exported ops (mu-law, causal_conv, time_to_batch,
batch_to_time) against the CPU oracle and the reference's known answers.
Bit-exact for integer work, exact/1e-6 for the float ops."""
import os

import numpy as np
import pytest
import torch

from util import O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def np_(t):
    return t.detach().cpu().numpy()


def test_mu_law_encode_precomputed(hip_lib):
    # test/test_mu_law.py:113-124 -- literal golden vector
    from wavenet import mu_law_encode
    lit = np.load(os.path.join(GOLD, 'reference_literals.npz'))
    got = np_(mu_law_encode(lit['mulaw_x'], 256))
    assert got.dtype == np.int32
    assert np.array_equal(got, lit['mulaw_codes'])


def test_mu_law_seeded_families_bit_exact(hip_lib):
    from wavenet import mu_law_encode, mu_law_decode
    fam = np.load(os.path.join(GOLD, 'mulaw_families.npz'))
    for k in ['uniform', 'const', 'ramp', 'zeros']:
        assert np.array_equal(np_(mu_law_encode(fam['enc_%s_x' % k], 256)),
                              fam['enc_%s_codes' % k]), k
    assert np.array_equal(np_(mu_law_decode(fam['dec128_codes'], 128)),
                          fam['dec128_audio'])
    assert np.array_equal(np_(mu_law_decode(np.arange(256), 256)),
                          fam['dec256_all'])


@pytest.mark.parametrize('q', [16, 123, 128, 256])
def test_mu_law_encode_dense_sweep_bit_exact(hip_lib, q):
    """2M-point sweep incl. every decision threshold and its float32
    neighbours: bit-exact against the oracle's float32 chain."""
    from wavenet import mu_law_encode
    thr = O.mu_law_thresholds(q)
    x = np.concatenate([
        np.linspace(-1, 1, 2000001).astype(np.float32), thr,
        np.nextafter(thr, np.float32(-2)), np.nextafter(thr, np.float32(2)),
        np.float32([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30])])
    assert np.array_equal(np_(mu_law_encode(x, q)), O.mu_law_encode(x, q))


def test_mu_law_out_of_range_and_shapes(hip_lib):
    from wavenet import mu_law_encode, mu_law_decode
    x = np.float32([[1.5, -1.5, 3.0], [-7.25, 1.0000001, -1.0000001]])
    got = mu_law_encode(x, 256)
    assert tuple(got.shape) == (2, 3)
    assert np.array_equal(np_(got), O.mu_law_encode(x, 256))
    codes = np.int32([-3, 0, 255, 256, 300])
    assert np.array_equal(np_(mu_law_decode(codes, 256)),
                          O.mu_law_decode(codes, 256))
    assert mu_law_encode(np.zeros((0,), np.float32), 256).numel() == 0
    # torch tensors on the device are accepted too
    t = torch.linspace(-1, 1, 1000, device='cuda')
    assert np.array_equal(np_(mu_law_encode(t, 256)),
                          O.mu_law_encode(np_(t), 256))


def test_mu_law_reference_properties(hip_lib):
    # test/test_mu_law.py:37-68, 70-111 on the GPU path
    from wavenet import mu_law_encode, mu_law_decode
    lv = np.arange(256)
    dec = mu_law_decode(lv, 256)
    assert np.array_equal(np_(mu_law_encode(dec, 256)), lv)
    d = np_(dec)
    assert abs(d.max() - 1.0) < 1e-10 and abs(d.min() + 1.0) < 1e-10
    x = np.linspace(-1, 1, 1000).astype(np.float32)
    x1 = np_(mu_law_decode(mu_law_encode(x, 256), 256))
    slope, icpt = np.polyfit(x, x1, 1)
    assert abs(slope - 1) < 1e-4 and abs(icpt) < 1e-4
    x2 = np_(mu_law_decode(mu_law_encode(x1, 256), 256))
    assert np.allclose(x1, x2)
    xs = np.linspace(-1, 1, 10000).astype(np.float32)
    assert len(np.unique(np_(mu_law_encode(xs, 123)))) == 123


def test_causal_conv_known_answers(hip_lib):
    # test/test_causal_conv.py:11-27 and :29-58, assertAllEqual
    from wavenet import causal_conv
    lit = np.load(os.path.join(GOLD, 'reference_literals.npz'))
    assert np.array_equal(np_(causal_conv(lit['cc_x'], lit['cc_f'], 4)),
                          lit['cc_y'])
    out = np_(causal_conv(lit['nts_x'], lit['nts_f'], dilation=2))
    assert out.shape == lit['nts_x'].shape
    assert np.array_equal(out, lit['nts_x'])


@pytest.mark.parametrize('K,d,T', [(2, 1, 5), (2, 4, 37), (2, 64, 20),
                                   (3, 2, 37), (4, 3, 50), (32, 1, 100)])
def test_causal_conv_random_vs_oracle(hip_lib, K, d, T):
    from wavenet import causal_conv
    rng = np.random.default_rng(K * 100 + d)
    x = rng.standard_normal((3, T, 5)).astype(np.float32)
    w = rng.standard_normal((K, 5, 7)).astype(np.float32)
    ref = O.causal_conv(x.astype(np.float64), w.astype(np.float64), d)
    assert np.abs(np_(causal_conv(x, w, d)) - ref).max() < 1e-4


@pytest.mark.parametrize('d', [1, 2, 3, 8])
def test_time_to_batch_and_back(hip_lib, d):
    from wavenet import time_to_batch, batch_to_time
    rng = np.random.default_rng(d)
    x = rng.standard_normal((2, 24, 3)).astype(np.float32)
    y = time_to_batch(x, d)
    assert np.array_equal(np_(y), O.time_to_batch(x, d))
    assert np.array_equal(np_(batch_to_time(y, d)), x)
    # ragged length: padded with zeros like ops.py:30-31
    xr = rng.standard_normal((2, 23, 3)).astype(np.float32)
    assert np.array_equal(np_(time_to_batch(xr, d)), O.time_to_batch(xr, d))
