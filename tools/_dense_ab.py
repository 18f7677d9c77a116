import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tensorflow-wavenet_amd'))
import numpy as np, torch
from wavenet import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
rows = 128000
for C in (64, 128):
    CB = C // 32
    x = torch.randn(CB, rows, 32, device='cuda'); a = torch.randn(CB, rows, 32, device='cuda')
    W = torch.randn(C, C, device='cuda'); b = torch.randn(C, device='cuda'); o = torch.empty(CB, rows, 32, device='cuda')
    def t(fn):
        ts = []
        for i in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
        return np.median(ts)
    g = t(lambda: lib.wn_gemm_nn(x.data_ptr(), 0, CB, rows * 32, W.data_ptr(), C, b.data_ptr(), None, 0, a.data_ptr(), 0, o.data_ptr(), 0, CB, rows * 32, None, rows, C, C, 0, st))
    d = t(lambda: lib.wn_dense_planes(x.data_ptr(), rows * 32, W.data_ptr(), b.data_ptr(), a.data_ptr(), rows * 32, o.data_ptr(), rows * 32, rows, C, st))
    mb = 3 * CB * rows * 128 / 1e6
    print('C=%d: wn_gemm_nn %.1f us, wn_dense_planes %.1f us (%.0f MB moved: %.2f TB/s)' % (C, g, d, mb, mb / d / 1e6 * 1e6 / 1e6))
