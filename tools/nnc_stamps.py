"""In-kernel stamps of gemm_nn_chain_kernel (diagnostic build -DNNC_STAMPS:
tools/mk_variants.sh wn_gemm.hip stamps:-DNNC_STAMPS; WN_LIB_PATH=.../lib_stamps.so):
per tile of a persistent workgroup, 10 ns clock: [0] ticket known, [1] dependency
seen, [2] tile start, [3] first chunk landed (behind the previous tile's stores),
[4] epilogue stores issued.   KB_SHAPE=dc1|post1|dtotal|post2|skip|dZ"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import _lib  # noqa: E402

M = int(os.environ.get('KB_ROWS', 128000))
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device='cuda').manual_seed(1)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
S, Q, L = 512, 256, 50
p = lambda t: t.data_ptr()
A512, A256, Zp = r(M, S), r(M, Q), r(L, M, 32)
mask = r(M, S)
shapes = {
    'skip': lambda C: (p(Zp), 0, L, M * 32, p(r(L * 32, S)), S, None, None, 0, None, 0, p(C), S, 0, 0, None, M, S, L * 32, 1),
    'post1': lambda C: (p(A512), S, 0, 0, p(r(S, S)), S, None, None, 0, None, 0, p(C), S, 0, 0, None, M, S, S, 1),
    'post2': lambda C: (p(A512), S, 0, 0, p(r(S, Q)), Q, None, None, 0, None, 0, p(C), Q, 0, 0, None, M, Q, S, 0),
    'dc1': lambda C: (p(A256), Q, 0, 0, p(r(Q, S)), S, None, p(mask), S, None, 0, p(C), S, 0, 0, None, M, S, Q, 0),
    'dtotal': lambda C: (p(A512), S, 0, 0, p(r(S, S)), S, None, p(mask), S, None, 0, p(C), S, 0, 0, None, M, S, S, 0),
}
scratch = torch.zeros(1024, dtype=torch.int32, device='cuda')
nx = lib.wn_gemm_nn_chain_probe(p(scratch), st)
ctl = torch.zeros(lib.wn_gemm_nn_chain_ctl_words(M), dtype=torch.int32, device='cuda')
C = torch.empty(M, S, device='cuda')
for name in os.environ.get('KB_SHAPE', 'post1,dc1,dtotal,post2').split(','):
    prob = shapes[name](C)
    dbg = torch.zeros(1024 * 256, dtype=torch.int64, device='cuda')
    for it in range(3):
        dbg.zero_()
        assert lib.wn_gemm_nn_chain(_lib.nn_problems([prob]), 1, nx, p(ctl), p(dbg), st) == 0
    torch.cuda.synchronize()
    s = dbg.cpu().numpy().reshape(1024, 32, 8).astype(np.float64) / 100.0   # us
    ntile = (s[:, :, 2] > 0).sum(1)
    t0 = s[:, 0, 0][s[:, 0, 0] > 0].min()
    print('== %s: tiles per workgroup median %d (max %d); launch span %.1f us' % (
        name, np.median(ntile), ntile.max(), s[:, :, 4].max() - t0))
    for k in range(int(min(6, ntile.max()))):
        ok = ntile > k
        a = s[ok, k]
        print('  tile %d (%4d wgs): ticket->start %5.2f  start->chunk0 landed %6.2f (p90 %6.2f)  '
              'landed->epilogue issued %7.2f   gap to next ticket %5.2f' % (
                  k, ok.sum(), np.median(a[:, 2] - a[:, 0]), np.median(a[:, 3] - a[:, 2]),
                  np.percentile(a[:, 3] - a[:, 2], 90), np.median(a[:, 4] - a[:, 3]),
                  np.median(s[ok & (ntile > k + 1), k + 1, 0] - s[ok & (ntile > k + 1), k, 4])
                  if (ok & (ntile > k + 1)).any() else float('nan')))

# ---- the real chains of three (forward: skip -> post1 -> post2; backward: dc1 -> dtotal -> dZ)
if os.environ.get('KB_CHAINS', '1') == '1':
    h1, h2, logits = r(M, S), r(M, S), r(M, Q)
    dc1, dtotal, dZ = r(M, S), r(M, S), r(L, M, 32)
    Ws, W1, W2 = r(L * 32, S), r(S, S), r(S, Q)
    W2t, W1t, Wst = r(Q, S), r(S, S), r(S, L * 32)
    fwd = [(p(Zp), 0, L, M * 32, p(Ws), S, None, None, 0, None, 0, p(h1), S, 0, 0, None, M, S, L * 32, 1),
           (p(h1), S, 0, 0, p(W1), S, None, None, 0, None, 0, p(h2), S, 0, 0, None, M, S, S, 1),
           (p(h2), S, 0, 0, p(W2), Q, None, None, 0, None, 0, p(logits), Q, 0, 0, None, M, Q, S, 0)]
    bwd = [(p(logits), Q, 0, 0, p(W2t), S, None, p(h2), S, None, 0, p(dc1), S, 0, 0, None, M, S, Q, 0),
           (p(dc1), S, 0, 0, p(W1t), S, None, p(h1), S, None, 0, p(dtotal), S, 0, 0, None, M, S, S, 0),
           (p(dtotal), S, 0, 0, p(Wst), L * 32, None, None, 0, None, 0, p(dZ), 0, L, M * 32, None, M, L * 32, S, 0)]
    tiles_m = (M + 127) // 128
    for tag, chain, tns in (('forward', fwd, (4, 4, 2)), ('backward', bwd, (4, 4, 13))):
        dbg = torch.zeros(1024 * 256, dtype=torch.int64, device='cuda')
        for it in range(3):
            dbg.zero_()
            assert lib.wn_gemm_nn_chain(_lib.nn_problems(chain), 3, nx, p(ctl), p(dbg), st) == 0
        torch.cuda.synchronize()
        raw = dbg.cpu().numpy().reshape(1024, 32, 8)
        s = raw.astype(np.float64) / 100.0
        tk = raw[:, :, 6]
        valid = raw[:, :, 2] > 0
        t0 = s[:, 0, 0][valid[:, 0]].min()
        nrb = tiles_m // nx          # (exact for M = 128000)
        c1, c2 = nrb * tns[0], nrb * (tns[0] + tns[1])
        prob = np.where(tk < c1, 0, np.where(tk < c2, 1, 2))
        print('== %s chain: span %.1f us; tiles per workgroup median %d max %d (stamps cover 32)' % (
            tag, s[:, :, 4].max() - t0, np.median(valid.sum(1)), valid.sum(1).max()))
        for q in range(3):
            m = valid & (prob == q)
            print('  problem %d: %5d tiles, first start %7.1f last end %7.1f; dependency wait median %5.2f '
                  'p90 %6.2f max %7.2f us; tile (landed->stores issued) median %6.1f' % (
                      q, m.sum(), s[:, :, 2][m].min() - t0, s[:, :, 4][m].max() - t0,
                      np.median((s[:, :, 1] - s[:, :, 0])[m]), np.percentile((s[:, :, 1] - s[:, :, 0])[m], 90),
                      (s[:, :, 1] - s[:, :, 0])[m].max(), np.median((s[:, :, 4] - s[:, :, 3])[m])))
