"""Time wn_causal_wgrad + its slab reduction (HIP events, median) and check it
against a float64 index_add:  python tools/cwg_bench.py [B] [T]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
Q = 256
lib = _lib.load()
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
q = (128 + 20 * torch.randn(B, T, generator=g)).round().clamp(0, Q - 1).to(torch.int32).to(dev)
dx = torch.randn(B * T, 32, generator=g).to(dev)
ns = lib.wn_causal_wgrad_slabs(B * T)
slabs = torch.zeros(ns * 2 * Q * 32, device=dev)
out = torch.zeros(2 * Q * 32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    _lib.call('wn_causal_wgrad', q.data_ptr(), dx.data_ptr(), slabs.data_ptr(), ns, B, T, Q, st)


def red():
    _lib.call('wn_reduce_slabs', slabs.data_ptr(), ns, 2 * Q * 32, 1, 0, 0, 2 * Q * 32,
              out.data_ptr(), 0, 1, 0, st)


run(); red()
torch.cuda.synchronize()
ref = torch.zeros(2, Q, 32, dtype=torch.float64, device=dev)
qq = q.long()
d3 = dx.double().reshape(B, T, 32)
ref[1].index_add_(0, qq.reshape(-1), d3.reshape(-1, 32))
ref[0].index_add_(0, qq[:, :-1].reshape(-1), d3[:, 1:].reshape(-1, 32))
err = float((out.double().reshape(2, Q, 32) - ref).abs().max())
print('slabs %d  max abs err vs f64 %.3e (max |ref| %.1f)' % (ns, err, float(ref.abs().max())))
for name, fn in (('causal_wgrad', run), ('reduce', red)):
    ts = []
    for _ in range(20):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print('%-14s median %.1f us  min %.1f' % (name, float(np.median(ts)), min(ts)))
