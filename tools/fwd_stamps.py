"""In-kernel phase stamps of layer_fwd_kernel (diagnostic build -DFWD_STAMPS
through WN_LIB_PATH; save mode 2, so the unused `th` pointer carries the
stamp buffer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np
import torch
from wavenet import _lib
lib = _lib.load()
dev = torch.device('cuda')
B, T, d = int(os.environ.get('KB_B', 8)), 16000, int(os.environ.get('KB_D', 4))
N = B * T
mk = lambda: torch.randn(N * 32, device=dev)
x, xo, z, sg = [mk() for _ in range(4)]
w = torch.randn(5216, device=dev) * 0.1
grid = 256
dbg = torch.zeros(grid * 2 * 16, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(5):
    dbg.zero_()
    _lib.call('wn_layer_fwd', x.data_ptr(), xo.data_ptr(), z.data_ptr(), dbg.data_ptr(), sg.data_ptr(),
              w.data_ptr(), None, 0, B, T, d, 1, 2, st)
torch.cuda.synchronize()
s = dbg.cpu().numpy().reshape(grid, 2, 16).astype(np.float64)
clk = np.median((s[:, :, 10] - s[:, :, 0]) / ((s[:, :, 15] - s[:, :, 14]) * 10.0))
names = {0: 'entry', 1: 'weight DMA + first loads issued', 2: 'loads landed', 3: 'workgroup barrier',
         4: 'x, x[t-d] in fragments', 5: '64 conv MFMAs', 6: 'tanh / sigmoid / z', 7: 'z, sigmoid stores issued',
         8: 'dense MFMAs', 9: "x' store issued", 10: 'stores drained'}
print('clock %.2f GHz' % clk)
for wv in (0, 1):
    print('--- wave %d of the workgroup' % (0 if wv == 0 else 15))
    prev = 0
    for i in sorted(names):
        v = s[:, wv, i]
        ok = v > 0
        dt = np.median((v - s[:, wv, prev])[ok]) / clk / 1e3
        at = np.median((v - s[:, wv, 0])[ok]) / clk / 1e3
        print('%-34s +%6.2f us   (at %6.2f us)' % (names[i], dt, at))
        prev = i
