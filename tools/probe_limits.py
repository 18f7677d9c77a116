"""Probe configurations at and beyond the tested ranges (skip / quantization
channels, layer count): training parity against the float64 oracle and one
fast-generation trace each.  Prints one line per case; exits non-zero when a
case fails.   python tools/probe_limits.py"""
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_model as M  # noqa: E402
from util import TINY, MID, cfg_with  # noqa: E402

O = M.O
CASES = [
    ('S1024', cfg_with(MID, batch_size=1, skip_channels=1024), 120),
    ('S520_Q264', cfg_with(MID, batch_size=1, skip_channels=520,
                           quantization_channels=264), 120),
    ('Q512', cfg_with(TINY, batch_size=2, quantization_channels=512), 80),
    ('Q1024', cfg_with(TINY, batch_size=1, quantization_channels=1024), 80),
    ('L70', cfg_with(TINY, batch_size=1, dilations=[1, 2, 4, 8, 16] * 14), 200),
    ('L250', cfg_with(TINY, batch_size=1, dilations=[1, 2, 4, 8, 16] * 50), 60),
    ('L300', cfg_with(TINY, batch_size=1, dilations=[1, 2, 4] * 100), 60),
    ('S4096_Q2048', cfg_with(TINY, batch_size=1, skip_channels=4096,
                             quantization_channels=2048), 40),
    ('r64_S1024_L70', cfg_with(TINY, batch_size=1, residual_channels=64,
                               dilation_channels=40, skip_channels=1024,
                               dilations=[1, 2] * 35), 60),
]
bad = 0
for name, cfg, T in CASES:
    for what in ('train', 'fastgen'):
        if what == 'train' and len(cfg['dilations']) > 200:
            continue          # (gradient tolerance is relative to fp32 conditioning: fastgen leg only)
        try:
            net, var = M.build_pair(cfg)
            B = cfg['batch_size']
            rng = np.random.default_rng(7)
            if what == 'train':
                audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
                loss = net.loss(audio, None, None)
                ref_loss, ref_g, c, _ = M.oracle_grads_at_device_kinks(
                    net, cfg, var, audio, None, None)
                assert abs(float(loss) - ref_loss) < M.TOL, (float(loss), ref_loss)
                M.check_grads(net, ref_g, tag=name)
                msg = 'loss %.6f' % float(loss)
            else:
                Q = cfg['quantization_channels']
                wave = rng.integers(0, Q, 40).astype(np.int32)
                gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
                # the same trace through the ORACLE in float32: what float32
                # arithmetic costs on this network (hundreds of layers of a
                # narrow random stack amplify rounding), the device gets 4x that
                gen32 = O.IncrementalGenerator(cfg, O.cast_variables(var, np.float32),
                                               dtype=np.float32)
                net.reset_generator()
                worst, e32 = 0.0, 0.0
                for cc in wave:
                    p_ref = np.asarray(gen.step(int(cc))).reshape(-1)
                    p32 = np.asarray(gen32.step(int(cc))).reshape(-1)
                    p = net.predict_proba_incremental(int(cc)).cpu().numpy()
                    worst = max(worst, float(np.abs(p - p_ref).max()))
                    e32 = max(e32, float(np.abs(p32 - p_ref).max()))
                tol = max(1e-5, 4.0 * e32)
                assert worst < tol, (worst, e32)
                a = net.generate(30, seed_samples=[Q // 2], seed=3).cpu().numpy()
                b = net.generate(30, seed_samples=[Q // 2], seed=3).cpu().numpy()
                assert np.array_equal(a, b) and a.max() < Q
                out, pr = net.generate(0, seed_samples=wave, return_proba_every=1)
                pr = pr.cpu().numpy()
                gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
                for i, cc in enumerate(wave[:len(pr)]):
                    p_ref = np.asarray(gen.step(int(cc))).reshape(-1)
                    worst = max(worst, float(np.abs(pr[i] - p_ref).max()))
                assert worst < tol, (worst, e32)
                msg = 'worst %.2e (float32 oracle %.2e)' % (worst, e32)
            torch.cuda.synchronize()
            print('%-16s %-8s ok   %s' % (name, what, msg), flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            tb = traceback.format_exc().strip().splitlines()
            print('%-16s %-8s FAIL %s: %s | %s' % (
                name, what, type(e).__name__, str(e)[:200], tb[-3].strip()[:120]),
                flush=True)
sys.exit(1 if bad else 0)
