"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference (TensorFlow 0.10 graph code) cannot be imported in this image
(`import wavenet` -> ModuleNotFoundError: tensorflow, an ordinary Python
error), so the vectors are (a) literal known-answer data restated from the
reference's own tests and (b) outputs of this repo's CPU oracle
(oracle/wavenet_oracle.py, float64) on small seeded inputs.  Run from the
repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import wavenet_oracle as O  # noqa: E402
from util import TINY, cfg_with  # noqa: E402


def literal():
    """Known answers held by the reference's tests (data only)."""
    out = {}
    # test/test_mu_law.py:113-124 (testEncodePrecomputed)
    out['mulaw_x'] = np.array([-1.0, 1.0, 0.6, -0.25, 0.01, 0.33, -0.9999,
                               0.42, 0.1, -0.45], np.float32)
    out['mulaw_codes'] = np.array([0, 255, 243, 32, 157, 230, 0, 235, 203,
                                   18], np.int32)
    # test/test_causal_conv.py:11-27 (testCausalConv): x = 1..20 twice,
    # filter ones[2,1,1], dilation 4  -> convolve(x,[1,0,0,0,1])[:-4]
    x1 = np.arange(1, 21, dtype=np.float32)
    out['cc_x'] = np.append(x1, x1).reshape(2, 20, 1)
    out['cc_f'] = np.ones((2, 1, 1), np.float32)
    ref = np.convolve(x1, [1, 0, 0, 0, 1])[:-4]
    out['cc_y'] = np.append(ref, ref).reshape(2, 20, 1).astype(np.float32)
    # test/test_causal_conv.py:29-58 (testNoTimeShift): filter [0,1], d=2
    out['nts_x'] = np.arange(1, 11, dtype=np.float32).reshape(1, 10, 1)
    out['nts_f'] = np.array([0.0, 1.0], np.float32).reshape(2, 1, 1)
    return out


def mulaw_families():
    """Inputs of test/test_mu_law.py:126-178, 205-261 (seeds 42/1944/40) and
    the oracle's float32-chain codes / decodes."""
    out = {}
    np.random.seed(42)
    out['enc_uniform_x'] = np.random.uniform(-1, 1, 2048).astype(np.float32)
    np.random.seed(1944)
    c = np.zeros(1024, np.float32)
    c.fill(np.random.uniform(-1, 1))
    out['enc_const_x'] = c
    out['enc_ramp_x'] = np.arange(-1.0, 1.0, 2.0 / 1024).astype(np.float32)
    out['enc_zeros_x'] = np.zeros(1024, np.float32)
    for k in ['uniform', 'const', 'ramp', 'zeros']:
        out['enc_%s_codes' % k] = O.mu_law_encode(out['enc_%s_x' % k], 256)
    np.random.seed(40)
    x = np.random.uniform(-1, 1, 512)
    y = O.mu_law_encode(x.astype(np.float32), 128)
    out['dec128_codes'] = y
    out['dec128_audio'] = O.mu_law_decode(y, 128)
    out['dec256_all'] = O.mu_law_decode(np.arange(256), 256)
    for q in (16, 123, 128, 256):
        out['thr_%d' % q] = O.mu_law_thresholds(q)
    return out


def stack_case(name, cfg, T, gc=False, l2=None, seed=0, quirk=True):
    B = cfg['batch_size']
    var = O.create_variables(cfg, seed=seed, dtype=np.float64, bias_scale=0.1)
    rng = np.random.default_rng(100 + seed)
    audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
    ids = rng.integers(0, cfg['global_condition_cardinality'], B) if gc \
        else None
    loss, g = O.loss_and_grads(cfg, var, audio, ids, l2=l2, dtype=np.float64,
                               tf_xent_zero_label_quirk=quirk)
    _, c = O.loss(cfg, var, audio, ids, l2, np.float64, keep=True)
    out = {'audio': audio, 'loss': np.float64(loss),
           'logits': c['logits'].astype(np.float32),
           'quirk': np.int32(quirk)}
    if ids is not None:
        out['ids'] = ids.astype(np.int32)
    if l2 is not None:
        out['l2'] = np.float64(l2)
    for i, (n, a) in enumerate(O.flatten_variables(var)):
        out['w%03d' % i] = a.astype(np.float32)
    for i, (n, a) in enumerate(O.flatten_variables(g)):
        out['g%03d' % i] = a.astype(np.float32)
    out['names'] = np.array([n for n, _ in O.flatten_variables(var)])
    return {name + '/' + k: v for k, v in out.items()}


def incremental_case(cfg, n, seed=0):
    var = O.create_variables(cfg, seed=seed, dtype=np.float64, bias_scale=0.1)
    rng = np.random.default_rng(7)
    Q = cfg['quantization_channels']
    wave = rng.integers(0, Q, n)
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    probs = np.stack([gen.step(int(s)) for s in wave])
    naive_last = O.predict_proba(cfg, var, wave, dtype=np.float64)
    out = {'inc/wave': wave.astype(np.int32), 'inc/probs': probs,
           'inc/naive_last': naive_last}
    for i, (nm, a) in enumerate(O.flatten_variables(var)):
        out['inc/w%03d' % i] = a.astype(np.float32)
    return out


def _relu_masks(c):
    return dict(total=c['total'] > 0, c1=c['c1'] > 0)


def fp32_error(cfg, var, audio, ids=None):
    """Per variable: max |g32 - g64| / max |g64| where g32 is THIS oracle run
    in float32 on the same inputs, taking float64's side at the ReLU kinks (so
    that only rounding is measured, not which subgradient a value that rounds
    across 0 picks).  It is what a straightforward float32 evaluation of the
    reference's graph loses on this case: the yardstick for how close a
    float32 device path can be asked to come (tests bound the device by
    max(2e-5, 4 x this))."""
    l64, g64, c = O.loss_and_grads(cfg, var, audio, ids, dtype=np.float64,
                                   return_cache=True)
    l32, g32 = O.loss_and_grads(cfg, var, audio, ids, dtype=np.float32,
                                relu_masks=_relu_masks(c))
    names, err = [], []
    for (n, a), (_, b) in zip(O.flatten_variables(g32), O.flatten_variables(g64)):
        names.append(n)
        sc = np.abs(b).max()
        err.append(np.abs(a.astype(np.float64) - b).max() / sc if sc > 0 else 0.0)
    return names, np.asarray(err), l64, g64, c, abs(float(l32) - float(l64))


def fp32_error_cases():
    """Deep narrow stacks are ill-conditioned: the float32
    oracle's own error for the cases of tests/test_gpu_model.py whose device
    error sits near the fixed 2e-5 bar.  Same weights (seed 0, biases
    N(0, 0.1)) and audio (default_rng(7)) as the test."""
    out = {}
    for name, cfg, T in (
            ('L70', cfg_with(TINY, batch_size=1, dilations=[1, 2, 4, 8, 16] * 14), 200),):
        var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
        audio = np.random.default_rng(7).uniform(-1, 1, (1, T)).astype(np.float32)
        names, err, l64, _, _, dl = fp32_error(cfg, var, audio)
        out[name + '/names'] = np.array(names)
        out[name + '/err32'] = err
        out[name + '/loss'] = np.float64(l64)
        out[name + '/loss_err32'] = np.float64(dl)
        print(name, 'float32 oracle vs float64: worst %.3e (%s)' % (
            err.max(), names[int(err.argmax())]))
    return out


N_SAMPLED = 32


def sample_index(n_elems, var_index):
    """The N_SAMPLED flat positions of a variable that the full-size fixture
    holds (seeded by the variable's index; shared with the tests)."""
    rng = np.random.default_rng(9000 + var_index)
    return rng.integers(0, n_elems, N_SAMPLED)


GC4 = dict(global_condition_channels=32, global_condition_cardinality=377)
CONFIG4_CLIP = 5          # a clip of the global batch with a non-trivial speaker id


def fullsize_inputs(tag):
    """(cfg, audio, speaker ids or None) of a full-length fixture; shared with
    the tests.  config1: BASELINE.json configs[0], clip 0.  config4: one GPU's
    share of configs[3] at one clip per GPU -- default stack + global
    conditioning 32 x 377, GLOBAL clip 5 of BASELINE.md's synthetic batch with
    its speaker id (37 * 5) mod 377 = 185."""
    from util import DEFAULT, synth_audio
    if tag == 'config1':
        return cfg_with(DEFAULT, batch_size=1), synth_audio(1, 16000), None
    assert tag == 'config4'
    audio = synth_audio(CONFIG4_CLIP + 1, 16000)[CONFIG4_CLIP:]
    ids = np.array([(37 * CONFIG4_CLIP) % 377], np.int32)
    return cfg_with(DEFAULT, batch_size=1, **GC4), audio, ids


def fullsize(tag):
    """A BASELINE.json configuration at FULL length (16000 samples, one clip):
    weights of create_variables(seed 0) with N(0, 0.1) biases, float64 oracle.
    The gradient tensors are 6 MB; committed are per variable sum, abs-sum, max
    abs and 32 sampled entries, the float32 oracle's own error (fp32_error),
    and how many post-processing pre-activations sit within 2e-5 of the ReLU
    kink (where a float32 device may legitimately take the other side)."""
    cfg, audio, ids = fullsize_inputs(tag)
    var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
    names, err, l64, g64, c, dl = fp32_error(cfg, var, audio, ids)
    flat = O.flatten_variables(g64)
    out = {'names': np.array(names), 'err32': err, 'loss': np.float64(l64),
           'loss_err32': np.float64(dl),
           'sum': np.array([a.sum() for _, a in flat]),
           'abssum': np.array([np.abs(a).sum() for _, a in flat]),
           'absmax': np.array([np.abs(a).max() for _, a in flat]),
           # (float32: 6e-8 relative, three hundred times below the 2e-5 bar)
           'samples': np.stack([a.reshape(-1)[sample_index(a.size, i)]
                                for i, (_, a) in enumerate(flat)]).astype(np.float32),
           'near_kink': np.array([int((np.abs(c[k]) < 2e-5).sum())
                                  for k in ('total', 'c1')]),
           'logits_first_last': np.stack([c['logits'][0, 0], c['logits'][0, -1]]),
           'audio_crc': np.array([audio.sum(dtype=np.float64),
                                  np.abs(audio).sum(dtype=np.float64)])}
    print('%s T=16000: loss %.9f, float32 oracle worst %.3e (%s), near-kink %s'
          % (tag, l64, err.max(), names[int(err.argmax())], out['near_kink']))
    return {tag + '/' + k: v for k, v in out.items()}


def main():
    if '--fullsize-only' not in sys.argv:
        small()
        np.savez_compressed(os.path.join(HERE, 'fp32_oracle_error.npz'),
                            **fp32_error_cases())
    for tag in ('config1', 'config4'):
        if '--only' in sys.argv and tag not in sys.argv:
            continue
        np.savez_compressed(os.path.join(HERE, tag + '_fullsize.npz'),
                            **fullsize(tag))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


def small():
    np.savez_compressed(os.path.join(HERE, 'reference_literals.npz'),
                        **literal())
    np.savez_compressed(os.path.join(HERE, 'mulaw_families.npz'),
                        **mulaw_families())
    cases = {}
    cases.update(stack_case('tiny', cfg_with(TINY, batch_size=2), 37))
    cases.update(stack_case('tiny_nobias',
                            cfg_with(TINY, batch_size=1, use_biases=False), 5))
    cases.update(stack_case('tiny_gc', cfg_with(
        TINY, batch_size=3, global_condition_channels=4,
        global_condition_cardinality=5), 50, gc=True))
    cases.update(stack_case('tiny_rp_l2', cfg_with(
        TINY, batch_size=2, residual_postproc=True), 40, l2=0.01))
    cases.update(stack_case('tiny_noquirk', cfg_with(TINY, batch_size=2), 33,
                            quirk=False, seed=3))
    np.savez_compressed(os.path.join(HERE, 'stack_cases.npz'), **cases)
    # receptive field of TINY = 2*(1+2+4+8)+2 = 32 -> 80 steps > RF
    np.savez_compressed(os.path.join(HERE, 'incremental.npz'),
                        **incremental_case(cfg_with(TINY, batch_size=1), 80))


if __name__ == '__main__':
    main()
