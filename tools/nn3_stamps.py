"""In-kernel stamps of gemm_nn3_kernel (diagnostic build -DNN3_STAMPS through
WN_LIB_PATH): per workgroup, entry -> first chunk landed -> K loop done ->
epilogue done, the clock, and how the workgroups of the launch are spread in
time (how many are inside their K loop at any moment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np
import torch
from wavenet import _lib
lib = _lib.load()
dev = torch.device('cuda')
N = 128000
st = torch.cuda.current_stream().cuda_stream
for (K, Nn, pa, pc, name) in [(1600, 512, 50, 0, 'skip'), (512, 512, 0, 0, 'post1'), (256, 512, 0, 0, 'dh2'), (512, 1600, 0, 50, 'dZ')]:
    A = torch.randn(N * K, device=dev)
    W = torch.randn(K * Nn, device=dev)
    C = torch.empty(N * Nn, device=dev)
    nwg = ((N + 127) // 128) * ((Nn + 127) // 128)
    dbg = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    for it in range(4):
        dbg.zero_()
        _lib.call('wn_gemm_nn', A.data_ptr(), 0 if pa else K, pa, N * 32, W.data_ptr(), Nn, None, None, 0, None, 0,
                  C.data_ptr(), 0 if pc else Nn, pc, N * 32, dbg.data_ptr(), N, Nn, K, 1, st)
    torch.cuda.synchronize()
    s = dbg.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    clk = np.median((s[:, 4] - s[:, 0]) / ((s[:, 7] - s[:, 6]) * 10.0))
    us = lambda c: c / clk / 1e3
    span = (s[:, 7].max() - s[:, 6].min()) * 0.01          # 100 MHz realtime counter
    pro, loop, bar, epi = (us(np.median(s[:, 1] - s[:, 0])), us(np.median(s[:, 2] - s[:, 1])),
                           us(np.median(s[:, 3] - s[:, 2])), us(np.median(s[:, 4] - s[:, 3])))
    nk = K // 16
    ideal_chunk = 32 * 64 * 4 / clk / 1e3     # 32 MFMAs x 64 cycles, 4 workgroups sharing a SIMD (us)
    print('%-6s K=%4d: clock %.2f GHz  launch span %.1f us; per workgroup: first chunk %.2f us, K loop %.1f us '
          '(%.3f us per chunk, MFMA-bound %.3f), barrier %.2f us, epilogue %.2f us'
          % (name, K, clk, span, pro, loop, loop / nk, ideal_chunk, bar, epi))
    # occupancy over time: fraction of the 1024 slots inside a K loop
    # realtime stamps exist for entry / exit only: place the loop stamps by
    # each workgroup's own cycle counter relative to its entry
    r0 = s[:, 6].min()
    ent = (s[:, 6] - r0) * 0.01
    l0 = ent + us(s[:, 1] - s[:, 0])
    l1 = ent + us(s[:, 2] - s[:, 0])
    ext = ent + us(s[:, 4] - s[:, 0])
    ts = np.linspace(0, span, 400)
    inloop = [((l0 <= t) & (l1 > t)).sum() for t in ts]
    resident = [((ent <= t) & (ext > t)).sum() for t in ts]
    print('        workgroups resident (mean %.0f) / inside the K loop (mean %.0f) of 1024 slots; '
          'last 10%% of the launch: %.0f / %.0f' % (np.mean(resident), np.mean(inloop),
                                                    np.mean(resident[-40:]), np.mean(inloop[-40:])))
