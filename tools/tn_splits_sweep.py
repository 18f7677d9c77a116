"""Split-count sweep of the three weight-gradient (TN) GEMMs of a step around
wn_gemm_tn_splits' recommendation (grid = tiles x splits):
    python tools/tn_splits_sweep.py        (KB_ROWS rows, default 128000)"""
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
st = torch.cuda.current_stream().cuda_stream
for name, Mw, Nw, pa in (('dWs', 1600, 512, 50), ('dW1', 512, 512, 0), ('dW2', 512, 256, 0)):
    A = torch.randn(N * Mw, device=dev)
    G = torch.randn(N * Nw, device=dev)
    rec = lib.wn_gemm_tn_splits(N, Mw, Nw, 0)
    sl = lib.wn_gemm_tn_slab_floats(Mw, Nw)
    cands = sorted({max(1, int(round(rec * f))) for f in (0.5, 0.64, 0.75, 0.8, 0.88, 0.96, 1.0, 1.04, 1.12, 1.25, 1.5, 2.0)})
    slabs = torch.zeros(max(cands) * sl, device=dev)
    dst = torch.zeros(Mw * Nw, device=dev)
    dstb = torch.zeros(Nw, device=dev)
    res = {}
    for rnd in range(4):
        for sp in cands:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                code = lib.wn_gemm_tn(A.data_ptr(), 0 if pa else Mw, pa, N * 32, None, 0, 16000,
                                      G.data_ptr(), Nw, slabs.data_ptr(), sp, N, Mw, Nw, 1, st)
                assert code == 0, code
                code = lib.wn_reduce_slabs_mt(slabs.data_ptr(), sp, sl, Mw * Nw, dst.data_ptr(), Nw,
                                              dstb.data_ptr(), 1, 0, 1, st)
                assert code == 0, code
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault(sp, []).append(e0.elapsed_time(e1) * 1e3 / 3)
    print('%s (recommended %d splits): ' % (name, rec) + '  '.join(
        '%d:%.0f' % (sp, float(np.median(res[sp]))) for sp in cands), flush=True)
