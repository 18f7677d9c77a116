"""Randomised cross-check of the persistent stack launches against the
per-layer kernels (same model, `stack_fwd` / `stack_bwd` on vs off): random
dilation lists, clip counts and lengths (ragged tiles, taps longer than the
clip, single-tile clips).  python tools/stack_fuzz.py [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
# KB_ROWS (default here: 32, the launches whose forward is bitwise the
# per-layer kernels'; 16: the small-batch launches, compared to rounding --
# KB_WAVES=4|8 selects their variant): an explicit variant word
# of the C entry points, set through the model class
ROWS16 = os.environ.get('KB_ROWS', '32') == '16'
import numpy as np  # noqa: E402
import torch  # noqa: E402
from util import TINY, build_pair, cfg_with, synth_audio  # noqa: E402
from wavenet import WaveNetModel  # noqa: E402
from wavenet._lib import stack_variant  # noqa: E402

WaveNetModel.DEFAULT_STACK_VARIANT = stack_variant(
    rows=16 if ROWS16 else 32, waves=int(os.environ.get('KB_WAVES', 0)))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n):
    L = int(rng.integers(1, 12))
    dil = [int(2 ** rng.integers(0, 10)) for _ in range(L)]
    B = int(rng.integers(1, 6))
    T = int(rng.choice([rng.integers(2, 40), rng.integers(40, 700), rng.integers(700, 4000)]))
    cfg = cfg_with(TINY, batch_size=B, dilations=dil, residual_channels=32, dilation_channels=32,
                   skip_channels=int(rng.choice([16, 64])), use_biases=bool(rng.integers(0, 2)))
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    b.stack_fwd = b.stack_bwd = False
    audio = synth_audio(B, T)
    la, lb = a.loss(audio), b.loss(audio)
    torch.cuda.synchronize()
    wa = list(a._ws.values())[0]
    assert int(wa.stack_ctl[3]) == 0 and (not wa.stack_bwd or int(wa.stack_ctl_b[3]) == 0)
    if ROWS16:
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb)), (case, float(la), float(lb))
        wb = list(b._ws.values())[0]
        for pl in ('X', 'Z', 'SG'):
            pa, pb = getattr(wa, pl), getattr(wb, pl)
            e = float((pa - pb).abs().max()) / max(float(pb.abs().max()), 1e-30)
            assert e <= 1e-4, (case, pl, e)
    else:
        assert float(la) == float(lb), (case, float(la), float(lb))
    sc = float(b.grads.abs().max())
    err = float((a.grads - b.grads).abs().max()) / max(sc, 1e-30)
    worst = max(worst, err)
    print('case %2d  L=%2d dil=%s B=%d T=%d  stack_bwd=%s  rel. grad diff %.2e' % (
        case, L, dil, B, T, bool(wa.stack_bwd), err), flush=True)
    # (16 rows: the forward differs in the last bits, and a ReLU of the
    # post-processing net may sit on the other side of its kink)
    assert err <= (2e-3 if ROWS16 else 5e-6), err
print('worst relative gradient difference %.2e over %d cases' % (worst, n))
