"""In-kernel phase stamps of layer_bwd2d_kernel (diagnostic build with
-DB2_STAMPS, loaded through WN_LIB_PATH; the shipped kernel has no stamps).
Prints the median duration of every phase over the workgroups, for the first
and the last wave of a workgroup, and the clock the launch held."""
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
x, z, dZ, dxin, dxo = [mk() for _ in range(5)]
sg = torch.rand(N * 32, device=dev) * 0.9 + 0.05
w = torch.randn(5216, device=dev) * 0.1
wimg = torch.randn(5376, device=dev) * 0.1
nsl = lib.wn_layer_bwd2_slabs(B, T)
slabs = torch.empty(nsl * 5216, device=dev)
dbg = torch.zeros(nsl * 2 * 64, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(5):
    dbg.zero_()
    _lib.call('wn_layer_bwd2', x.data_ptr(), z.data_ptr(), sg.data_ptr(), dZ.data_ptr(), dxin.data_ptr(),
              dxo.data_ptr(), w.data_ptr(), wimg.data_ptr(), slabs.data_ptr(), dbg.data_ptr(), B, T, d, st)
torch.cuda.synchronize()
s = dbg.cpu().numpy().reshape(nsl, 2, 64).astype(np.float64)
names = {0: 'entry', 1: 'first loads issued', 2: 'weights staged + barrier'}
for k in range(2):
    b = 3 + 10 * k
    names.update({b: 't%d loop top' % k, b + 1: 't%d rows t+d in frags' % k, b + 2: 't%d rows-t DMA issued' % k,
                  b + 3: 't%d rows t+d math done' % k, b + 4: 't%d rows-t landed' % k,
                  b + 5: 't%d dWd + frags + x DMA issued' % k, b + 6: 't%d rows-t math, dx store, da->LDS' % k,
                  b + 7: 't%d x landed' % k, b + 8: 't%d dW done' % k, b + 9: 't%d end' % k})
names.update({30: 'loop done', 31: 'after barrier', 32: 'tree reduced', 33: 'slab written'})
clk = (s[:, :, 33] - s[:, :, 0]) / ((s[:, :, 41] - s[:, :, 40]) * 10.0)   # cycles per ns
print('clock %.2f GHz, kernel span (median wave) %.1f us' % (np.median(clk), np.median(s[:, :, 33] - s[:, :, 0]) / np.median(clk) / 1e3))
for wv in (0, 1):
    print('--- wave %d of the workgroup' % (0 if wv == 0 else 7))
    prev = 0
    for i in sorted(names):
        v = s[:, wv, i]
        ok = v > 0
        if not ok.any():
            continue
        dt = np.median((v - s[:, wv, prev])[ok]) / np.median(clk) / 1e3
        at = np.median((v - s[:, wv, 0])[ok]) / np.median(clk) / 1e3
        print('%-36s +%6.2f us   (at %6.2f us)' % (names[i], dt, at))
        prev = i
# spread of workgroup start / end over the grid
e0 = s[:, 0, 0]
print('workgroup entry spread: %.2f us; end spread: %.2f us' % ((e0.max() - e0.min()) / np.median(clk) / 1e3,
      (s[:, 0, 33].max() - s[:, 0, 33].min()) / np.median(clk) / 1e3))
