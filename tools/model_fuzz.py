"""Random model configurations (dilations, channel counts, filter widths, skip /
quantization channels, biases, global conditioning, scalar input, residual
post-processing, L2) against the float64 oracle: loss, every variable's
gradient, and -- where the reference's generator supports the configuration --
an incremental-generation trace.   python tools/model_fuzz.py [cases] [seed]
KB_BIG=1: 32 channels, filter width 2, dilations up to 512, clips of 200 - 4000
samples (the persistent stack launches' territory)."""
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_model as M  # noqa: E402

O = M.O
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    L = int(rng.integers(1, 9))
    wide = rng.random() < 0.35
    cfg = dict(
        dilations=[int(2 ** rng.integers(0, 6)) for _ in range(L)],
        filter_width=int(rng.choice([2, 2, 2, 3, 4, 5, 9])),
        residual_channels=int(rng.choice([8, 24, 32, 40, 64, 72]) if wide else rng.choice([4, 8, 16, 32])),
        dilation_channels=int(rng.choice([8, 32, 48, 64, 96]) if wide else rng.choice([4, 8, 16, 32])),
        skip_channels=int(rng.choice([4, 16, 32, 64, 260])),
        quantization_channels=int(rng.choice([8, 16, 32, 256])),
        use_biases=bool(rng.random() < 0.7),
        batch_size=int(rng.integers(1, 4)))
    if rng.random() < 0.3:
        cfg['scalar_input'] = True
        cfg['initial_filter_width'] = int(rng.choice([2, 4, 32, 33]))
    if rng.random() < 0.3:
        cfg['residual_postproc'] = True
    gc = rng.random() < 0.35
    if gc:
        cfg['global_condition_channels'] = int(rng.choice([3, 4, 16]))
        cfg['global_condition_cardinality'] = int(rng.choice([2, 5, 9]))
    l2 = float(rng.choice([0.0, 0.0, 1e-3])) or None
    T = int(rng.integers(8, 300))
    if os.environ.get('KB_BIG') == '1':
        # the default model's territory: 32 channels, width 2, dilations up to
        # 512, clips of several tiles' length (the persistent stack launches)
        cfg['dilations'] = [int(2 ** rng.integers(0, 10)) for _ in range(int(rng.integers(2, 13)))]
        cfg['filter_width'] = 2
        cfg['residual_channels'] = int(rng.choice([16, 32]))
        cfg['dilation_channels'] = int(rng.choice([16, 32]))
        cfg.pop('scalar_input', None)
        cfg.pop('initial_filter_width', None)
        T = int(rng.integers(200, 4000))
        L = len(cfg['dilations'])
    B = cfg['batch_size']
    tag = 'L%d k%d r%d d%d s%d q%d b%d B%d T%d%s%s%s%s' % (
        L, cfg['filter_width'], cfg['residual_channels'], cfg['dilation_channels'],
        cfg['skip_channels'], cfg['quantization_channels'], cfg['use_biases'], B, T,
        ' scalar%d' % cfg['initial_filter_width'] if cfg.get('scalar_input') else '',
        ' rp' if cfg.get('residual_postproc') else '', ' gc' if gc else '',
        ' l2' if l2 else '')
    try:
        net, var = M.build_pair(cfg)
        audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
        ids = rng.integers(0, cfg['global_condition_cardinality'], B) if gc else None
        loss = net.loss(audio, ids, l2)
        ref_loss, ref_g, c, _ = M.oracle_grads_at_device_kinks(net, cfg, var, audio, ids, l2)
        assert abs(float(loss) - ref_loss) < M.TOL, (float(loss), ref_loss)
        M.check_grads(net, ref_g, tag='fuzz')
        msg = 'loss %.5f' % float(loss)
        if cfg['filter_width'] == 2 and not cfg.get('scalar_input') and B == 1:
            Q = cfg['quantization_channels']
            wave = rng.integers(0, Q, 30).astype(np.int32)
            gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
            gid = None if ids is None else int(ids[0])
            net.reset_generator()
            worst = 0.0
            for cc in wave:
                p_ref = np.asarray(gen.step(int(cc), gc_ids=None if gid is None else np.array([gid]))).reshape(-1)
                p = net.predict_proba_incremental(int(cc), global_condition=gid).cpu().numpy()
                worst = max(worst, float(np.abs(p - p_ref).max()))
            assert worst < 1e-5, worst
            msg += ', generator trace %.1e' % worst
        torch.cuda.synchronize()
        print('%3d ok   %-58s %s' % (case, tag, msg), flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        tb = traceback.format_exc().strip().splitlines()
        print('%3d FAIL %-58s %s: %s | %s' % (case, tag, type(e).__name__, str(e)[:160],
                                              tb[-3].strip()[:100]), flush=True)
print('%d of %d cases failed' % (bad, n_cases))
sys.exit(1 if bad else 0)
