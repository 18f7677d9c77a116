"""Stage-by-stage parity dump (development aid; run on the GPU box)."""
import sys
import numpy as np
import torch
from util import O, TINY, MID, DEFAULT, cfg_with, build_pair, flat_named, \
    tree_to_numpy, synth_audio


def maxerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max()), float(np.abs(b).max())


def run(name, cfg, T, gc=False, l2=None):
    B = cfg['batch_size']
    print('=== %s B=%d T=%d gc=%s l2=%s' % (name, B, T, gc, l2)); sys.stdout.flush()
    net, var = build_pair(cfg)
    rng = np.random.default_rng(7)
    audio = rng.uniform(-1, 1, (B, T)).astype(np.float32)
    ids = rng.integers(0, cfg['global_condition_cardinality'], B) if gc else None
    Lo, g = O.loss_and_grads(cfg, var, audio, ids, l2=l2, dtype=np.float64)
    _, c = O.loss(cfg, var, audio, ids, l2, np.float64, keep=True)
    loss = net.loss(audio, ids, l2)
    torch.cuda.synchronize()
    ws = list(net._ws.values())[0]
    R, D = cfg['residual_channels'], cfg['dilation_channels']
    q = O.mu_law_encode(audio, cfg['quantization_channels'])
    print('codes mismatches', int((ws.q.cpu().numpy().reshape(B, T) != q).sum()))
    L = len(cfg['dilations'])
    worst = {}
    for l in range(L):
        lc = c['layers'][l]
        for nm, buf, ref, ch in (('x', ws.X, lc['x'], R), ('z', ws.Z, lc['z'], D),
                                 ('th', ws.TH, lc['tanh'], D), ('sg', ws.SG, lc['sig'], D)):
            got = buf[l].cpu().numpy().reshape(B, T, 32)
            e, m = maxerr(got[:, :, :ch], ref)
            pad = float(np.abs(got[:, :, ch:]).max()) if ch < 32 and nm in ('x', 'z') else 0.0
            worst[nm] = max(worst.get(nm, 0), e)
            if e > 1e-4 or pad > 0:
                print('  layer', l, nm, 'err', e, 'ref', m, 'pad', pad)
    print('  per-layer worst', worst)
    print('  h1', maxerr(ws.h1.cpu().numpy().reshape(B, T, -1), c['h1']))
    h2ref = c['h2']
    print('  h2', maxerr(ws.h2.cpu().numpy().reshape(B, T, -1), h2ref))
    print('  loss', float(loss), Lo, abs(float(loss) - Lo))
    gn = tree_to_numpy(net.gradients)
    bad = 0
    for (n, ga), (_, gr) in zip(flat_named(gn), flat_named(g)):
        e, m = maxerr(ga, gr)
        rel = e / (m + 1e-12)
        if rel > 2e-3 and e > 1e-7:
            bad += 1
            if bad < 25:
                print('  GRAD BAD', n, 'err', e, 'ref', m)
    print('  grads bad count', bad, 'of', len(flat_named(g)))
    sys.stdout.flush()


if __name__ == '__main__':
    torch.manual_seed(0)
    run('tiny', cfg_with(TINY, batch_size=2), 37)
    run('tiny-nobias', cfg_with(TINY, batch_size=1, use_biases=False), 5)
    run('mid', cfg_with(MID, batch_size=2), 300)
    run('tiny-gc', cfg_with(TINY, batch_size=3, global_condition_channels=4,
                            global_condition_cardinality=5), 50, gc=True)
    run('tiny-rp-l2', cfg_with(TINY, batch_size=2, residual_postproc=True), 40, l2=0.01)
    run('default', cfg_with(DEFAULT, batch_size=1), 1500)
