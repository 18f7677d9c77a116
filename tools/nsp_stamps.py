"""In-kernel clock of gemm_nn_split_kernel's K loop (diagnostic build
-DNSP_STAMPS through WN_LIB_PATH): s_memtime / s_memrealtime around the loop,
after two seconds of back-to-back launches on random data (MI355X_MICROARCH.md
'DVFS give-back' item 6), and the loop's cycles per 16-deep chunk against the
matrix pipe's (16 nprod bf16 MFMAs of 16 cycles per wave and 32-deep chunk, 2
waves per SIMD)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np
import torch
from wavenet import _lib
lib = _lib.load()
dev = torch.device('cuda')
N = 128000
st = torch.cuda.current_stream().cuda_stream
nprod = int(os.environ.get('KB_NPROD', 6))
for (K, Nn, name) in [(1600, 512, 'skip'), (512, 512, 'post1'), (256, 512, 'dh2'), (512, 1600, 'dZ')]:
    A = torch.randn(N, K, device=dev)
    W = torch.randn(K, Nn, device=dev) * 0.05
    C = torch.empty(N, Nn, device=dev)
    scratch = torch.empty(lib.wn_gemm_split_w_bytes(K, Nn) // 4, dtype=torch.int32, device=dev)
    nwg = ((N + 127) // 128) * ((Nn + 127) // 128)
    dbg = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20):
            _lib.call('wn_gemm_nn_split', A.data_ptr(), K, 0, 0, W.data_ptr(), Nn, None, None, 0, None, 0,
                      C.data_ptr(), Nn, 0, 0, dbg.data_ptr(), N, Nn, K, 0, scratch.data_ptr(), nprod, st)
        torch.cuda.synchronize()
    s = dbg.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    cyc = s[:, 2] - s[:, 0]
    clk = np.median(cyc / ((s[:, 3] - s[:, 1]) * 10.0))
    nk = (K + 31) // 32
    per = np.median(cyc) / nk
    pipe = 16 * nprod * 16 * 2          # MFMA cycles per chunk, 2 workgroups per CU
    print('%-6s K=%4d x%d: in-kernel clock %.2f GHz; K loop %.0f cycles per chunk per workgroup; matrix pipe '
          '%d cycles per chunk at 2 workgroups per CU = %.2f of the loop; loop %.1f us per workgroup'
          % (name, K, nprod, clk, per, pipe, pipe / per, np.median(cyc) / clk / 1e3))
    us = lambda c: np.median(c) / clk / 1e3
    span = (s[:, 7].max() - s[:, 5].min()) * 0.01
    print('        entry -> loop %.1f us, loop -> exit (epilogue) %.1f us; launch span %.1f us = %.2f rounds of '
          '%d workgroups x (%.1f us resident)' % (us(s[:, 0] - s[:, 4]), us(s[:, 6] - s[:, 2]), span,
                                                 nwg / 512.0, 512, us(s[:, 6] - s[:, 4])))
