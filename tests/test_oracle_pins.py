"""The oracle against every known answer the reference's own tests hold for
the hot path: these PIN causal_conv and mu-law."""
import os

import numpy as np
import pytest

from util import O, ROOT

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_causal_conv_known_answer():
    # test/test_causal_conv.py:11-27
    lit = np.load(os.path.join(GOLD, 'reference_literals.npz'))
    out = O.causal_conv(lit['cc_x'], lit['cc_f'], 4)
    assert np.array_equal(out, lit['cc_y'])
    assert np.array_equal(O.causal_conv_literal(lit['cc_x'], lit['cc_f'], 4),
                          lit['cc_y'])


def test_causal_conv_no_time_shift():
    # test/test_causal_conv.py:29-58
    lit = np.load(os.path.join(GOLD, 'reference_literals.npz'))
    out = O.causal_conv(lit['nts_x'], lit['nts_f'], 2)
    assert out.shape == lit['nts_x'].shape
    assert np.array_equal(out, lit['nts_x'])


def test_closed_form_equals_literal_graph():
    rng = np.random.default_rng(1)
    for K in (2, 3, 4, 32):
        for d in (1, 2, 3, 7, 64):
            for T in (5, 20, 37):
                x = rng.standard_normal((3, T, 4))
                w = rng.standard_normal((K, 4, 5))
                assert np.allclose(O.causal_conv(x, w, d),
                                   O.causal_conv_literal(x, w, d), atol=1e-12)


def test_time_to_batch_roundtrip():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 24, 3))
    for d in (1, 2, 3, 4, 8):
        y = O.time_to_batch(x, d)
        assert y.shape == (2 * d, 24 // d, 3)
        assert np.array_equal(O.batch_to_time(y, d), x)


def test_mu_law_encode_precomputed():
    # test/test_mu_law.py:113-124
    lit = np.load(os.path.join(GOLD, 'reference_literals.npz'))
    assert np.array_equal(O.mu_law_encode(lit['mulaw_x'], 256),
                          lit['mulaw_codes'])


def test_mu_law_decode_encode_all_levels():
    # test/test_mu_law.py:37-51
    x = np.arange(256)
    assert np.array_equal(O.mu_law_encode(O.mu_law_decode(x, 256), 256), x)


def test_mu_law_min_max_range():
    # test/test_mu_law.py:53-68
    d = O.mu_law_decode(np.arange(256), 256)
    assert abs(d.max() - 1.0) < 1e-10 and abs(d.min() + 1.0) < 1e-10


def test_mu_law_shift_and_invariance():
    # test/test_mu_law.py:70-104
    x = np.linspace(-1, 1, 1000).astype(np.float32)
    x1 = O.mu_law_decode(O.mu_law_encode(x, 256), 256)
    slope, icpt = np.polyfit(x, x1, 1)
    assert abs(slope - 1.0) < 1e-4 and abs(icpt) < 1e-4
    assert np.allclose(x, x1, rtol=1e-1, atol=0.05)
    x2 = O.mu_law_decode(O.mu_law_encode(x1, 256), 256)
    assert np.allclose(x1, x2)


def test_mu_law_surjective_123():
    # test/test_mu_law.py:106-111
    x = np.linspace(-1, 1, 10000).astype(np.float32)
    assert len(np.unique(O.mu_law_encode(x, 123))) == 123


def _manual_encode_f32(signal, channels):
    """test/test_mu_law.py:11-22 restated with the float32 arithmetic it had
    under the NumPy of its day (weak python scalars)."""
    mu = np.float32(channels - 1)
    s = signal.astype(np.float32)
    mag = np.log(np.float32(1) + mu * np.abs(s)) / np.log(np.float32(1) + mu)
    s = np.sign(s) * mag
    s = (s + np.float32(1)) / np.float32(2) * mu + np.float32(0.5)
    return s.astype(np.int32)


def test_mu_law_seeded_families():
    # inputs of test/test_mu_law.py:126-178; the oracle's correctly-rounded-log
    # chain must agree with the plain float32 numpy chain and the fixtures
    fam = np.load(os.path.join(GOLD, 'mulaw_families.npz'))
    for k in ['uniform', 'const', 'ramp', 'zeros']:
        x = fam['enc_%s_x' % k]
        got = O.mu_law_encode(x, 256)
        assert np.array_equal(got, fam['enc_%s_codes' % k])
        assert np.array_equal(got, _manual_encode_f32(x, 256))


def test_mu_law_decode_family():
    # test/test_mu_law.py:205-261 (128 channels)
    fam = np.load(os.path.join(GOLD, 'mulaw_families.npz'))
    assert np.array_equal(O.mu_law_decode(fam['dec128_codes'], 128),
                          fam['dec128_audio'])
    assert np.array_equal(O.mu_law_decode(np.arange(256), 256),
                          fam['dec256_all'])


def test_thresholds_reproduce_encode():
    fam = np.load(os.path.join(GOLD, 'mulaw_families.npz'))
    rng = np.random.default_rng(0)
    for q in (16, 123, 128, 256):
        thr = O.mu_law_thresholds(q)
        assert np.array_equal(thr, fam['thr_%d' % q])
        assert np.all(np.diff(thr) > 0)
        x = np.concatenate([rng.uniform(-1, 1, 50000).astype(np.float32), thr,
                            np.nextafter(thr, np.float32(-2)),
                            np.float32([-1, 1, 0, -0.0])])
        assert np.array_equal(O.mu_law_encode(x, q),
                              np.searchsorted(thr, x, side='right'))


@pytest.mark.parametrize('tag', ['config1', 'config4'])
def test_fullsize_fixture_is_the_oracle_at_full_length(tag):
    """tests/golden/config1_fullsize.npz (BASELINE.json configs[0]: default
    stack, ONE clip of 16000 samples, float64) and config4_fullsize.npz (one
    GPU's share of configs[3]: the same with global conditioning 32 x 377,
    global clip 5, speaker id 185) are what this oracle computes: loss, and
    every variable's gradient through its sum / abs-sum / max and 32 sampled
    entries.  (The GPU test of the same name pins the device to them.)"""
    import sys
    gold = os.path.join(ROOT, 'tests', 'golden')
    if gold not in sys.path:
        sys.path.insert(0, gold)
    from make_golden import fullsize_inputs, sample_index
    fx = np.load(os.path.join(gold, tag + '_fullsize.npz'))
    cfg, audio, ids = fullsize_inputs(tag)
    var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
    loss, g = O.loss_and_grads(cfg, var, audio, ids, dtype=np.float64)
    assert abs(loss - float(fx[tag + '/loss'])) < 1e-12
    flat = O.flatten_variables(g)
    assert [n for n, _ in flat] == [str(n) for n in fx[tag + '/names']]
    # 405 variables; + the embedding table and 2 x 50 conditioning convs
    assert len(flat) == (405 if ids is None else 506)
    for i, (n, a) in enumerate(flat):
        sc = float(fx[tag + '/absmax'][i])
        assert abs(np.abs(a).max() - sc) <= 1e-9 * sc + 1e-300, n
        assert abs(a.sum() - fx[tag + '/sum'][i]) <= 1e-9 * fx[tag + '/abssum'][i] + 1e-300, n
        got = a.reshape(-1)[sample_index(a.size, i)]
        assert np.abs(got - fx[tag + '/samples'][i]).max() <= 2e-7 * sc + 1e-300, n
    # the float32 oracle's own error on this case is the tests' yardstick
    assert 0 < fx[tag + '/err32'].max() < 2e-5
