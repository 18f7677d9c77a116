"""Per-layer phase stamps of the fast-generation chain wave (diagnostic build
-DFG_STAMPS through WN_LIB_PATH; the shipped kernel has none)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np
import torch
from wavenet import WaveNetModel, _lib
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
net = WaveNetModel(batch_size=1, dilations=p['dilations'], filter_width=2, residual_channels=32,
                   dilation_channels=32, skip_channels=512, quantization_channels=256, use_biases=True, seed=0)
net.generate(600, seed_samples=[128], seed=1)
torch.cuda.synchronize()
lib = _lib.load()
L = 50
buf = np.zeros(8 + 64 * 8, np.uint64)
lib.wn_diag_fg_stamps.restype = ctypes.c_int
lib.wn_diag_fg_stamps.argtypes = [ctypes.c_void_p]
assert lib.wn_diag_fg_stamps(buf.ctypes.data) == 0
s = buf.astype(np.float64)
clk = 2.4   # GHz assumed (single-CU latency-bound kernel); cycles shown too
print('entry -> cursors read %d cyc; draw + code %d cyc; prologue to first barrier %d cyc (%.2f us total)' % (
    s[1] - s[0], s[2] - s[1], s[3] - s[2], (s[3] - s[0]) / clk / 1e3))
ph = np.array([[s[8 + l * 8 + i] for i in range(5)] for l in range(L)])
d = np.diff(ph, axis=1)
names = ['x broadcast + current-tap mat-vec', 'activation + permlane', 'z broadcast + dense mat-vec', 'wait + barrier']
for i, n in enumerate(names):
    print('%-36s median %5.0f cyc  (min %5.0f max %5.0f)' % (n, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
per = np.diff(ph[:, 0])
print('layer to layer: median %.0f cyc = %.3f us at %.1f GHz; chain of %d layers %.1f us' % (
    np.median(per), np.median(per) / clk / 1e3, clk, L, (ph[-1, 4] - ph[0, 0]) / clk / 1e3))
