"""wn_gemm_nn_chain against single wn_gemm_nn launches on the six NN shapes of a
training step (B x T rows, default stack), HIP events, median of KB_REPS:
every shape as a single launch, as a chain of one (the persistent kernel without
dependencies), and the forward / backward chains of three against the sum of
their single launches.   python tools/chain_ab.py   (KB_ROWS=128000)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import _lib  # noqa: E402

M = int(os.environ.get('KB_ROWS', 128000))
reps = int(os.environ.get('KB_REPS', 9))
lib = _lib.load()   # (WN_LIB_PATH selects an A/B build)
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device='cuda').manual_seed(1)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
S, Q, L = 512, 256, 50
Z, h1, h2, logits = r(L, M, 32), r(M, S), r(M, S), r(M, Q)
dc1, dtotal, dZ = r(M, S), r(M, S), r(L, M, 32)
Ws, W1, W2 = r(L * 32, S), r(S, S), r(S, Q)
W2t, W1t, Wst = r(Q, S), r(S, S), r(S, L * 32)
bs, b1, b2 = r(S), r(S), r(Q)
p = lambda t: t.data_ptr()
fwd = [(p(Z), 0, L, M * 32, p(Ws), S, p(bs), None, 0, None, 0, p(h1), S, 0, 0, None, M, S, L * 32, 1),
       (p(h1), S, 0, 0, p(W1), S, p(b1), None, 0, None, 0, p(h2), S, 0, 0, None, M, S, S, 1),
       (p(h2), S, 0, 0, p(W2), Q, p(b2), None, 0, None, 0, p(logits), Q, 0, 0, None, M, Q, S, 0)]
bwd = [(p(logits), Q, 0, 0, p(W2t), S, None, p(h2), S, None, 0, p(dc1), S, 0, 0, None, M, S, Q, 0),
       (p(dc1), S, 0, 0, p(W1t), S, None, p(h1), S, None, 0, p(dtotal), S, 0, 0, None, M, S, S, 0),
       (p(dtotal), S, 0, 0, p(Wst), L * 32, None, None, 0, None, 0, p(dZ), 0, L, M * 32, None, M, L * 32, S, 0)]
names = ['skip', 'post1', 'post2', 'dc1', 'dtotal', 'dZ']
scratch = torch.zeros(1024, dtype=torch.int32, device='cuda')
nx = lib.wn_gemm_nn_chain_probe(p(scratch), st)
ctl = torch.zeros(lib.wn_gemm_nn_chain_ctl_words(M), dtype=torch.int32, device='cuda')
print('rows %d, XCD queues %d' % (M, nx))


def timed(fn):
    ts = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def single(c):
    assert lib.wn_gemm_nn(*(c + (st,))) == 0


def chain(cs):
    assert lib.wn_gemm_nn_chain(_lib.nn_problems(cs), len(cs), nx, p(ctl), None, st) == 0


tot = {}
for name, c in zip(names, fwd + bwd):
    fl = 2.0 * c[16] * c[17] * c[18]
    a = timed(lambda: single(c))
    b = timed(lambda: chain([c]))
    tot[name] = a
    print('%-7s single %7.1f us (%5.1f TFLOP/s)   chain of one %7.1f us (%5.1f)' % (
        name, a, fl / a / 1e6, b, fl / b / 1e6))
for tag, cs, ns in (('forward', fwd, names[:3]), ('backward', bwd, names[3:])):
    a = timed(lambda: [single(c) for c in cs])
    b = timed(lambda: chain(cs))
    print('%-8s three launches %7.1f us (sum of singles %7.1f)   chain of three %7.1f us' % (
        tag, a, sum(tot[n] for n in ns), b))
    b2 = timed(lambda: chain(cs[:2]))
    print('%-8s chain of the first two %7.1f us (singles %7.1f)' % (tag, b2, sum(tot[n] for n in ns[:2])))
assert int(ctl[9]) == 0
