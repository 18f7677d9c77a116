"""Do the six NN GEMMs of a step run faster as `parts` independent row ranges
on `parts` streams (every op of the post-processing chain is row-wise, so a
row range can run ahead of the others and one chain's launch tails overlap the
other chain's steady state)?   python tools/nn_rows_split.py [parts ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import _lib  # noqa: E402

lib = _lib.load()
N = int(os.environ.get('KB_ROWS', 128000))
dev = torch.device('cuda')
# (name, K, Nn, a_planes, c_planes, bias, relu, cpre, mask, addend)
shapes = [('skip', 1600, 512, 50, 0, 1, 1, 0, 0, 0), ('post1', 512, 512, 0, 0, 1, 1, 1, 0, 0),
          ('post2', 512, 256, 0, 0, 1, 0, 0, 0, 0), ('dc1', 256, 512, 0, 0, 0, 0, 0, 1, 0),
          ('dtotal', 512, 512, 0, 0, 0, 0, 0, 1, 1), ('dZ', 512, 1600, 0, 50, 0, 0, 0, 0, 0)]
bufs = []
for name, K, Nn, pa, pc, ub, relu, cpre, um, ua in shapes:
    bufs.append(dict(
        A=torch.randn(N * K, device=dev), W=torch.randn(K * Nn, device=dev),
        C=torch.empty(N * Nn, device=dev),
        Cp=torch.empty(N * Nn, device=dev) if cpre else None,
        bias=torch.randn(Nn, device=dev) if ub else None,
        mask=torch.randn(N * Nn, device=dev) if um else None,
        add=torch.randn(N * Nn, device=dev) if ua else None))


def off(t, floats):
    return None if t is None else t.data_ptr() + 4 * floats


def chain(r0, rows, st):
    for (name, K, Nn, pa, pc, ub, relu, cpre, um, ua), b in zip(shapes, bufs):
        a_off = r0 * 32 if pa else r0 * K
        c_off = r0 * 32 if pc else r0 * Nn
        code = lib.wn_gemm_nn(off(b['A'], a_off), 0 if pa else K, pa, N * 32, b['W'].data_ptr(), Nn,
                              off(b['bias'], 0), off(b['mask'], r0 * Nn), Nn if um else 0,
                              off(b['add'], r0 * Nn), Nn if ua else 0, off(b['C'], c_off),
                              0 if pc else Nn, pc, N * 32, off(b['Cp'], r0 * Nn), rows, Nn, K, relu, st)
        assert code == 0, (name, code)


def run(parts, streams):
    main = torch.cuda.current_stream()
    if parts == 1:
        chain(0, N, main.cuda_stream)
        return
    ev = torch.cuda.Event()
    ev.record(main)
    per = (N // parts + 127) // 128 * 128
    for p in range(parts):
        r0 = p * per
        rows = min(per, N - r0)
        s = streams[p]
        s.wait_event(ev)
        chain(r0, rows, s.cuda_stream)
        e2 = torch.cuda.Event()
        e2.record(s)
        main.wait_event(e2)


streams = [torch.cuda.Stream() for _ in range(8)]
ref = None
for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 1, 2]:
    ts = []
    for r in range(6):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run(parts, streams)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3)
    outs = [b['C'].clone() for b in bufs]
    if ref is None:
        ref = outs
    same = all(torch.equal(a, b) for a, b in zip(outs, ref))
    print('parts %d: six NN GEMMs %8.1f us (min %8.1f)  %s' % (
        parts, float(np.median(ts)), min(ts), 'bitwise' if same else 'DIFFERS'), flush=True)
