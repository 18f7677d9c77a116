"""Direct parity of the fp32 MFMA GEMM entry points (wn_gemm_nn,
wn_gemm_tn + wn_reduce_slabs) against float64 numpy, through the C ABI:
ragged M, partial tiles in N and K, plane-mode operands, every epilogue."""
import numpy as np
import pytest
import torch

from util import PKG  # noqa: F401

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


@pytest.mark.parametrize('kind', ['nn'])
@pytest.mark.parametrize('M,N,K', [(1, 4, 4), (37, 16, 16), (300, 132, 68),
                                   (5000, 256, 512), (129, 512, 96),
                                   (256, 200, 64), (1031, 1600, 32)])
def test_gemm_dense_epilogues(hip_lib, kind, M, N, K):
    from wavenet import _lib
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    mask = rng.standard_normal((M, N)).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    dA, dW = dev(A), dev(W if kind == 'nn' else W.T)
    db, dm, dadd = dev(bias), dev(mask), dev(add)
    C = torch.empty((M, N), device='cuda')
    Cpre = torch.empty((M, N), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    ldw = N if kind == 'nn' else K
    ref0 = A.astype(np.float64) @ W.astype(np.float64) + bias
    for relu, use_mask, use_add in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 0, 1)]:
        _lib.call('wn_gemm_' + kind, dA.data_ptr(), K, 0, 0, dW.data_ptr(), ldw,
                  db.data_ptr(), dm.data_ptr() if use_mask else None, N,
                  dadd.data_ptr() if use_add else None, N, C.data_ptr(), N, 0,
                  0, Cpre.data_ptr(), M, N, K, relu, st)
        ref = ref0.copy()
        if relu:
            ref = np.maximum(ref, 0)
        if use_mask:
            ref = np.where(mask > 0, ref, 0)
        if use_add:
            ref = ref + add
        tol = 1e-4 * max(1.0, np.abs(ref0).max())
        assert np.abs(C.cpu().numpy() - ref).max() < tol
        assert np.abs(Cpre.cpu().numpy() - ref0).max() < tol


@pytest.mark.parametrize('kind', ['nn'])
def test_gemm_plane_operands(hip_lib, kind):
    """A as [P][M][32] planes (the z planes of the skip sum) and C as planes
    (dZ): K = P*32 / N = P*32."""
    from wavenet import _lib
    rng = np.random.default_rng(3)
    M, P, N = 777, 5, 64
    Ap = rng.standard_normal((P, M, 32)).astype(np.float32)
    W = rng.standard_normal((P * 32, N)).astype(np.float32)
    A = Ap.transpose(1, 0, 2).reshape(M, P * 32)
    dA, dW = dev(Ap), dev(W if kind == 'nn' else W.T)
    C = torch.empty((M, N), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('wn_gemm_' + kind, dA.data_ptr(), 0, P, M * 32, dW.data_ptr(),
              N if kind == 'nn' else P * 32, None, None, 0, None, 0,
              C.data_ptr(), N, 0, 0, None, M, N, P * 32, 0, st)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    assert np.abs(C.cpu().numpy() - ref).max() < 1e-3
    # planes out: C2[p][m][32] = (X @ W2)[:, p*32:(p+1)*32]
    X = rng.standard_normal((M, 48)).astype(np.float32)
    W2 = rng.standard_normal((48, P * 32)).astype(np.float32)
    Cp = torch.empty((P, M, 32), device='cuda')
    dX, dW2 = dev(X), dev(W2 if kind == 'nn' else W2.T)   # keep alive
    _lib.call('wn_gemm_' + kind, dX.data_ptr(), 48, 0, 0,
              dW2.data_ptr(),
              P * 32 if kind == 'nn' else 48, None, None, 0, None, 0,
              Cp.data_ptr(), 0, P, M * 32, None, M, P * 32, 48, 0, st)
    ref2 = (X.astype(np.float64) @ W2.astype(np.float64)).reshape(M, P, 32)
    assert np.abs(Cp.cpu().numpy().transpose(1, 0, 2) - ref2).max() < 1e-3


@pytest.mark.parametrize('rows,Mw,Nw,splits', [(100, 32, 32, 3), (1000, 64, 96, 7),
                                               (5000, 160, 128, 9), (3333, 512, 256, 5),
                                               (777, 96, 64, 1),
                                               # whole 16-row chunks: LDS-DMA kernel
                                               (4800, 160, 128, 9), (3328, 512, 256, 5),
                                               (2048, 512, 512, 3), (1600, 128, 128, 40),
                                               (96, 160, 256, 2)])
def test_gemm_tn_and_reduce(hip_lib, rows, Mw, Nw, splits):
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(rows)
    A = rng.standard_normal((rows, Mw)).astype(np.float32)
    G = rng.standard_normal((rows, Nw)).astype(np.float32)
    sl = lib.wn_gemm_tn_slab_floats(Mw, Nw)
    slabs = torch.zeros(splits * sl, device='cuda')
    out = torch.empty(Mw * Nw, device='cuda')
    cs = torch.empty(Nw, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    dA, dG = dev(A), dev(G)                                # keep alive
    _lib.call('wn_gemm_tn', dA.data_ptr(), Mw, 0, 0, None, 0, 1,
              dG.data_ptr(), Nw, slabs.data_ptr(), splits, rows, Mw, Nw, 1,
              st)
    _lib.call('wn_reduce_slabs', slabs.data_ptr(), splits, sl, 1, 0, 0,
              Mw * Nw, out.data_ptr(), 0, 1, 0, st)
    _lib.call('wn_reduce_slabs', slabs.data_ptr(), splits, sl, 1, 0, Mw * Nw,
              Nw, cs.data_ptr(), 0, 1, 0, st)
    ref = A.astype(np.float64).T @ G.astype(np.float64)
    assert np.abs(out.cpu().numpy().reshape(Mw, Nw) - ref).max() < \
        1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(cs.cpu().numpy() - G.astype(np.float64).sum(0)).max() < 1e-2


@pytest.mark.parametrize('nprod,tol', [(6, 1e-5), (9, 1e-5), (3, 2e-3)])
@pytest.mark.parametrize('M,N,K', [(37, 16, 16), (5000, 256, 512), (129, 512, 96),
                                   (256, 200, 64), (1031, 1600, 32), (300, 72, 48)])
def test_gemm_nn_split(hip_lib, nprod, tol, M, N, K):
    """wn_gemm_nn_split (opt-in): fp32 products rebuilt from exact bf16 pieces.
    nprod 6 / 9 must be as close to float64 as an fp32 GEMM (1e-5 of the
    result scale here), nprod 3 only ~2^-16; epilogues shared with wn_gemm_nn.  (K = 16, 48: the
    kernel's 32-deep last chunk is half empty.)"""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(M + N + K + nprod)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    mask = rng.standard_normal((M, N)).astype(np.float32)
    dA, dW, db, dm = dev(A), dev(W), dev(bias), dev(mask)
    C = torch.empty((M, N), device='cuda')
    Cpre = torch.empty((M, N), device='cuda')
    scratch = torch.empty(lib.wn_gemm_split_w_bytes(K, N) // 4,
                          dtype=torch.int32, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('wn_gemm_nn_split', dA.data_ptr(), K, 0, 0, dW.data_ptr(), N,
              db.data_ptr(), dm.data_ptr(), N, None, 0, C.data_ptr(), N, 0, 0,
              Cpre.data_ptr(), M, N, K, 1, scratch.data_ptr(), nprod, st)
    ref0 = A.astype(np.float64) @ W.astype(np.float64) + bias
    ref = np.where(mask > 0, np.maximum(ref0, 0), 0)
    scale = max(1.0, np.abs(ref0).max())
    assert np.abs(Cpre.cpu().numpy() - ref0).max() < tol * scale
    assert np.abs(C.cpu().numpy() - ref).max() < tol * scale
    # K not a multiple of 16 is refused, not silently mis-computed
    assert lib.wn_gemm_nn_split(dA.data_ptr(), K, 0, 0, dW.data_ptr(), N, None,
                                None, 0, None, 0, C.data_ptr(), N, 0, 0, None,
                                M, N, K - 4, 0, scratch.data_ptr(), nprod,
                                st) != 0


@pytest.mark.parametrize('nprod', [6, 9])
@pytest.mark.parametrize('rows,Mw,Nw,splits,planes', [(96, 32, 32, 2, 0), (4800, 160, 128, 9, 5),
                                                      (3328, 512, 256, 5, 0), (1600, 200, 132, 3, 0)])
def test_gemm_tn_split(hip_lib, nprod, rows, Mw, Nw, splits, planes):
    """wn_gemm_tn_split (opt-in) + wn_reduce_slabs vs float64, dense and
    plane-mode A, partial 128-tiles in both output dimensions, column sums."""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(rows + nprod)
    A = rng.standard_normal((rows, Mw)).astype(np.float32)
    G = rng.standard_normal((rows, Nw)).astype(np.float32)
    if planes:
        dA = dev(A.reshape(rows, planes, 32).transpose(1, 0, 2))
        a_args = (dA.data_ptr(), 0, planes, rows * 32)
    else:
        dA = dev(A)
        a_args = (dA.data_ptr(), Mw, 0, 0)
    dG = dev(G)
    sl = lib.wn_gemm_tn_slab_floats(Mw, Nw)
    slabs = torch.zeros(splits * sl, device='cuda')
    out = torch.empty(Mw * Nw, device='cuda')
    cs = torch.empty(Nw, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('wn_gemm_tn_split', *a_args, dG.data_ptr(), Nw, slabs.data_ptr(),
              splits, rows, Mw, Nw, 1, nprod, st)
    _lib.call('wn_reduce_slabs', slabs.data_ptr(), splits, sl, 1, 0, 0,
              Mw * Nw, out.data_ptr(), 0, 1, 0, st)
    _lib.call('wn_reduce_slabs', slabs.data_ptr(), splits, sl, 1, 0, Mw * Nw,
              Nw, cs.data_ptr(), 0, 1, 0, st)
    ref = A.astype(np.float64).T @ G.astype(np.float64)
    assert np.abs(out.cpu().numpy().reshape(Mw, Nw) - ref).max() < \
        1e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(cs.cpu().numpy() - G.astype(np.float64).sum(0)).max() < 1e-2
    # ragged row counts are refused (the caller falls back to wn_gemm_tn)
    assert lib.wn_gemm_tn_split(*a_args, dG.data_ptr(), Nw, slabs.data_ptr(),
                                splits, rows - 3, Mw, Nw, 1, nprod, st) != 0


def test_gemm_tn_plane_operand(hip_lib):
    """A given as [P][rows][32] planes (dWs = Z^T dtotal), both TN kernels
    (rows % 16 == 0 -> LDS-DMA, else register-staged)."""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(11)
    for rows, splits in ((1600, 6), (1000, 4)):
        P, Nw = 5, 128
        Ap = rng.standard_normal((P, rows, 32)).astype(np.float32)
        G = rng.standard_normal((rows, Nw)).astype(np.float32)
        A = Ap.transpose(1, 0, 2).reshape(rows, P * 32)
        sl = lib.wn_gemm_tn_slab_floats(P * 32, Nw)
        slabs = torch.zeros(splits * sl, device='cuda')
        out = torch.empty(P * 32 * Nw, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        dA, dG = dev(Ap), dev(G)
        _lib.call('wn_gemm_tn', dA.data_ptr(), 0, P, rows * 32, None, 0, 1,
                  dG.data_ptr(), Nw, slabs.data_ptr(), splits, rows, P * 32,
                  Nw, 1, st)
        _lib.call('wn_reduce_slabs', slabs.data_ptr(), splits, sl, 1, 0, 0,
                  P * 32 * Nw, out.data_ptr(), 0, 1, 0, st)
        ref = A.astype(np.float64).T @ G.astype(np.float64)
        assert np.abs(out.cpu().numpy().reshape(P * 32, Nw) - ref).max() < \
            1e-4 * max(1.0, np.abs(ref).max())


def test_gemm_tn_splits_policy(hip_lib):
    """wn_gemm_tn_splits: >= 1, never more than rows / 64, and a grid of at
    most one resident wave of workgroups for the LDS-staged shapes."""
    lib = hip_lib
    assert lib.wn_gemm_tn_splits(128000, 1600, 512, 0) in range(1, 2001)
    assert lib.wn_gemm_tn_splits(64, 512, 512, 0) == 1
    assert lib.wn_gemm_tn_splits(100, 256, 32, 1) == 1
    assert lib.wn_gemm_tn_splits(128000, 512, 512, 0) * 8 <= 4 * 304
    assert lib.wn_gemm_tn_splits(0, 512, 512, 0) == 1


def test_gemm_tn_one_hot(hip_lib):
    """one-hot A generated from int codes with a per-clip shift (the causal
    layer's weight gradient)."""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(9)
    B, T, Q = 3, 50, 16
    q = rng.integers(0, Q, (B, T)).astype(np.int32)
    G = rng.standard_normal((B * T, 32)).astype(np.float32)
    sl = lib.wn_gemm_tn_slab_floats(Q, 32)
    st = torch.cuda.current_stream().cuda_stream
    dq = torch.as_tensor(q).cuda()
    dG = dev(G)
    for shift in (0, 1, 3):
        slabs = torch.zeros(4 * sl, device='cuda')
        out = torch.empty(Q * 32, device='cuda')
        _lib.call('wn_gemm_tn', None, 0, 0, 0, dq.data_ptr(), shift, T,
                  dG.data_ptr(), 32, slabs.data_ptr(), 4, B * T, Q, 32, 0,
                  st)
        _lib.call('wn_reduce_slabs', slabs.data_ptr(), 4, sl, 1, 0, 0, Q * 32,
                  out.data_ptr(), 0, 1, 0, st)
        ref = np.zeros((Q, 32))
        Gr = G.reshape(B, T, 32)
        for b in range(B):
            for t in range(shift, T):
                ref[q[b, t - shift]] += Gr[b, t]
        assert np.abs(out.cpu().numpy().reshape(Q, 32) - ref).max() < 1e-4


@pytest.mark.parametrize('B,T,Q', [(3, 50, 16), (1, 1, 256), (2, 700, 256), (5, 333, 123)])
def test_causal_wgrad_segmented_sum(hip_lib, B, T, Q):
    """wn_causal_wgrad: both taps of the one-hot causal layer's weight
    gradient as a segmented sum; clip boundaries (tap 0 has no t-1 at t = 0),
    ragged slab ranges, deterministic."""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(B * T + Q)
    q = rng.integers(0, Q, (B, T)).astype(np.int32)
    G = rng.standard_normal((B * T, 32)).astype(np.float32)
    ns = lib.wn_causal_wgrad_slabs(B * T)
    assert ns >= 2 and ns % 2 == 0
    st = torch.cuda.current_stream().cuda_stream
    dq, dG = torch.as_tensor(q).cuda(), dev(G)
    outs = []
    for _ in range(2):
        slabs = torch.full((ns * 2 * Q * 32,), float('nan'), device='cuda')
        out = torch.empty(2 * Q * 32, device='cuda')
        _lib.call('wn_causal_wgrad', dq.data_ptr(), dG.data_ptr(),
                  slabs.data_ptr(), ns, B, T, Q, st)
        _lib.call('wn_reduce_slabs', slabs.data_ptr(), ns, 2 * Q * 32, 1, 0, 0,
                  2 * Q * 32, out.data_ptr(), 0, 1, 0, st)
        outs.append(out.cpu().numpy().reshape(2, Q, 32))
    assert np.array_equal(outs[0], outs[1])
    ref = np.zeros((2, Q, 32))
    Gr = G.reshape(B, T, 32).astype(np.float64)
    for b in range(B):
        for t in range(T):
            ref[1, q[b, t]] += Gr[b, t]
            if t >= 1:
                ref[0, q[b, t - 1]] += Gr[b, t]
    assert np.abs(outs[0] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('nslab,n,batch,rep', [(1026, 1, 1, 1), (300, 3, 2, 2),
                                               (255, 2, 1, 1), (2048, 4, 3, 1)])
def test_reduce_slabs_few_outputs(hip_lib, nslab, n, batch, rep):
    """wn_reduce_slabs with a handful of outputs and many slabs (the loss
    partials): the wave-parallel path (>= 256 slabs) and the plain one give the
    float64 sum to rounding, identically on every run."""
    from wavenet import _lib
    rng = np.random.default_rng(nslab + n)
    stride, bstride, off = 7, nslab * 7 + 5, 2
    src = rng.standard_normal(batch * bstride + 16).astype(np.float32)
    d = dev(src)
    out = torch.full((batch * 8 + rep * 16,), -1.0, device='cuda')
    outs = []
    for _ in range(2):
        _lib.call('wn_reduce_slabs', d.data_ptr(), nslab, stride, batch, bstride,
                  off, n, out.data_ptr(), 8, rep, 16, torch.cuda.current_stream().cuda_stream)
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])
    got = outs[0].cpu().numpy()
    for b in range(batch):
        for e in range(n):
            ref = sum(float(src[b * bstride + off + e + s * stride]) for s in range(nslab))
            for r in range(rep):
                assert abs(got[b * 8 + r * 16 + e] - ref) <= 1e-5 * max(1.0, nslab ** 0.5)


@pytest.mark.parametrize('M,P', [(300, 2), (4099, 4), (129, 8)])
def test_gemm_plane_addend(hip_lib, M, P):
    """planes in, planes out, the addend in the OUTPUT's plane layout
    (ld_add = 0): x_{l+1} planes = x_l planes + z_l Wd (+ bd) of the
    channel-block models (model.py:294-300, 330 with more than 32 channels)."""
    from wavenet import _lib
    rng = np.random.default_rng(M + P)
    C = 32 * P
    Z = rng.standard_normal((P, M, 32)).astype(np.float32)
    X = rng.standard_normal((P, M, 32)).astype(np.float32)
    W = rng.standard_normal((C, C)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    dZ, dX, dW, db = dev(Z), dev(X), dev(W), dev(bias)
    out = torch.zeros((P, M, 32), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('wn_gemm_nn', dZ.data_ptr(), 0, P, M * 32, dW.data_ptr(), C,
              db.data_ptr(), None, 0, dX.data_ptr(), 0, out.data_ptr(), 0, P,
              M * 32, None, M, C, C, 0, st)
    zd = Z.transpose(1, 0, 2).reshape(M, C).astype(np.float64)
    xd = X.transpose(1, 0, 2).reshape(M, C).astype(np.float64)
    ref = xd + zd @ W.astype(np.float64) + bias
    got = out.cpu().numpy().transpose(1, 0, 2).reshape(M, C)
    assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # a dense output cannot take a plane addend
    lib = _lib.load()
    assert lib.wn_gemm_nn(dZ.data_ptr(), 0, P, M * 32, dW.data_ptr(), C, None, None, 0,
                          dX.data_ptr(), 0, out.data_ptr(), C, 0, 0, None, M, C, C, 0,
                          st) != 0


@pytest.mark.parametrize('CB,K,dense,ub', [(2, 2, 1, 1), (3, 3, 0, 1), (2, 4, 1, 0), (5, 2, 1, 1)])
def test_reduce_pair_slabs(hip_lib, CB, K, dense, ub):
    """wn_reduce_pair_slabs: the CB x CB block pairs' weight-gradient slabs of
    one layer summed into the layer's [K][C][C] / [C][C] matrices and bias
    vectors (fixed order: two runs are bitwise equal)."""
    from wavenet import _lib
    rng = np.random.default_rng(CB * 10 + K)
    C, ns = 32 * CB, 37
    WF = (2 * K + 1) * 1024
    slabs = rng.standard_normal((CB * CB, ns, WF + 96)).astype(np.float32)
    d = dev(slabs)
    off_b = (2 * K + 1) * C * C
    g = torch.full((off_b + 3 * C,), 7.0, device='cuda')
    g2 = g.clone()
    st = torch.cuda.current_stream().cuda_stream
    for t in (g, g2):
        _lib.call('wn_reduce_pair_slabs', d.data_ptr(), ns, CB, K, dense, ub, t.data_ptr(),
                  C, off_b, 0, K, st)
    assert torch.equal(g, g2)
    tot = slabs.astype(np.float64).sum(1)                    # [pair][WF + 96]
    want = np.full(off_b + 3 * C, 7.0)
    mats = want[:off_b].reshape(2 * K + 1, C, C)
    for a in range(CB):
        for b in range(CB):
            blk = tot[a * CB + b, :WF].reshape(2 * K + 1, 32, 32)
            nm = 2 * K + 1 if dense else 2 * K
            mats[:nm, a * 32:(a + 1) * 32, b * 32:(b + 1) * 32] = blk[:nm]
            if a == 0 and ub:
                for q in range(3 if dense else 2):
                    want[off_b + q * C + b * 32:off_b + q * C + (b + 1) * 32] = \
                        tot[b, WF + q * 32:WF + (q + 1) * 32]
    assert np.abs(g.cpu().numpy() - want).max() < 1e-4
    # a group of taps of a wider filter: taps 3 .. 3 + K - 1 of K + 5
    Kt = K + 5
    off_t = (2 * Kt + 1) * C * C
    gt = torch.full((off_t + 3 * C,), 7.0, device='cuda')
    _lib.call('wn_reduce_pair_slabs', d.data_ptr(), ns, CB, K, 0, 0, gt.data_ptr(), C,
              off_t, 3, Kt, st)
    wt = np.full(off_t + 3 * C, 7.0)
    mt = wt[:off_t].reshape(2 * Kt + 1, C, C)
    for a in range(CB):
        for b in range(CB):
            blk = tot[a * CB + b, :WF].reshape(2 * K + 1, 32, 32)
            mt[3:3 + K, a * 32:(a + 1) * 32, b * 32:(b + 1) * 32] = blk[:K]
            mt[Kt + 3:Kt + 3 + K, a * 32:(a + 1) * 32, b * 32:(b + 1) * 32] = blk[K:2 * K]
    assert np.abs(gt.cpu().numpy() - wt).max() < 1e-4


@pytest.mark.parametrize('ns,n_main,n_tail,rep', [(25, 4096, 64, 1), (96, 1024, 32, 5), (3, 64, 0, 1)])
def test_reduce_slabs_matrix_and_tail(hip_lib, ns, n_main, n_tail, rep):
    """wn_reduce_slabs_mt: a weight-gradient GEMM's slabs summed in one launch,
    the matrix part and the column-sum tail (bias gradient, replicated) to
    their own destinations; fixed order (two runs bitwise equal)."""
    from wavenet import _lib
    rng = np.random.default_rng(ns + n_main)
    stride = n_main + n_tail + 8
    slabs = rng.standard_normal((ns, stride)).astype(np.float32)
    d = dev(slabs)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(2):
        m = torch.full((n_main,), 3.0, device='cuda')
        t = torch.full((rep, n_tail + 4), 3.0, device='cuda')
        _lib.call('wn_reduce_slabs_mt', d.data_ptr(), ns, stride, n_main, m.data_ptr(),
                  n_tail, t.data_ptr() if n_tail else None, rep, n_tail + 4, 1, st)
        outs.append((m, t))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    tot = slabs.astype(np.float64).sum(0)
    assert np.abs(outs[0][0].cpu().numpy() - tot[:n_main]).max() < 1e-4
    tt = outs[0][1].cpu().numpy()
    if n_tail:
        for r in range(rep):
            assert np.abs(tt[r, :n_tail] - tot[n_main:n_main + n_tail]).max() < 1e-4
    assert np.all(tt[:, n_tail:] == 3.0)


def test_gemm_plane_operands_beyond_2gb(hip_lib):
    """Plane-mode operands whose planes lie more than 2 GB apart (the z planes
    of a large batch or a wide model): the LDS-DMA kernels address a chunk /
    a tile through 64-bit scalar bases and stay correct -- and in use (a
    fallback to the register-staged kernels would hide a perf cliff)."""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(11)
    M, P, N = 1600, 5, 128
    stride = 600_000_000                     # floats between planes: 2.4 GB
    big = torch.empty(stride * (P - 1) + M * 32, device='cuda')
    Ap = rng.standard_normal((P, M, 32)).astype(np.float32)
    for p in range(P):
        big[p * stride:p * stride + M * 32] = dev(Ap[p]).reshape(-1)
    A = Ap.transpose(1, 0, 2).reshape(M, P * 32).astype(np.float64)
    W = rng.standard_normal((P * 32, N)).astype(np.float32)
    dW = dev(W)
    C = torch.empty((M, N), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('wn_gemm_nn', big.data_ptr(), 0, P, stride, dW.data_ptr(), N, None, None, 0,
              None, 0, C.data_ptr(), N, 0, 0, None, M, N, P * 32, 0, st)
    assert np.abs(C.cpu().numpy() - A @ W.astype(np.float64)).max() < 1e-3
    # TN: dW[P*32][N] = A^T G with A in the same far-apart planes
    G = rng.standard_normal((M, N)).astype(np.float32)
    dG = dev(G)
    splits = 4
    sl = lib.wn_gemm_tn_slab_floats(P * 32, N)
    slabs = torch.zeros(splits * sl, device='cuda')
    out = torch.empty(P * 32 * N, device='cuda')
    _lib.call('wn_gemm_tn', big.data_ptr(), 0, P, stride, None, 0, 1, dG.data_ptr(), N,
              slabs.data_ptr(), splits, M, P * 32, N, 0, st)
    _lib.call('wn_reduce_slabs', slabs.data_ptr(), splits, sl, 1, 0, 0, P * 32 * N,
              out.data_ptr(), 0, 1, 0, st)
    ref = A.T @ G.astype(np.float64)
    assert np.abs(out.cpu().numpy().reshape(P * 32, N) - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    # planes OUT that far apart
    X = rng.standard_normal((M, 48)).astype(np.float32)
    W2 = rng.standard_normal((48, P * 32)).astype(np.float32)
    dX, dW2 = dev(X), dev(W2)
    # (the epilogue reaches a lane's neighbour plane with a 32-bit offset: refused)
    assert lib.wn_gemm_nn(dX.data_ptr(), 48, 0, 0, dW2.data_ptr(), P * 32, None, None, 0,
                          None, 0, big.data_ptr(), 0, P, stride, None, M, P * 32, 48, 0,
                          st) == -2


def _chain_case(M, P_in, S, Q, P_out, rp, seed):
    """The two chains of a training step in miniature: forward skip sum (plane
    operand) -> post1 -> post2, backward dc1 -> dtotal -> dZ (plane output,
    N % 128 == 32 or 64: a wave half beyond N), with biases, ReLU, masks, the
    pre-activation copy and (rp) the addend a previous problem wrote."""
    rng = np.random.default_rng(seed)
    r = lambda *s: dev(rng.standard_normal(s).astype(np.float32))
    t = dict(Z=r(P_in, M, 32), Ws=r(P_in * 32, S), bs=r(S), W1=r(S, S), b1=r(S),
             W2=r(S, Q), b2=r(Q), dlog=r(M, Q), W2t=r(Q, S), W1t=r(S, S),
             Wst=r(S, P_out * 32))
    return t


def _chain_problems(t, o, M, P_in, S, Q, P_out, rp):
    p = lambda x: None if x is None else x.data_ptr()
    fwd = [
        (p(t['Z']), 0, P_in, M * 32, p(t['Ws']), S, p(t['bs']), None, 0, None, 0,
         p(o['h1']), S, 0, 0, p(o['total']) if rp else None, M, S, P_in * 32, 1),
        (p(o['h1']), S, 0, 0, p(t['W1']), S, p(t['b1']), None, 0,
         p(o['total']) if rp else None, S, p(o['h2']), S, 0, 0,
         p(o['c1']) if rp else None, M, S, S, 1),
        (p(o['h2']), S, 0, 0, p(t['W2']), Q, p(t['b2']), None, 0, None, 0,
         p(o['logits']), Q, 0, 0, None, M, Q, S, 0)]
    bwd = [
        (p(t['dlog']), Q, 0, 0, p(t['W2t']), S, None, p(o['c1'] if rp else o['h2']), S,
         None, 0, p(o['dc1']), S, 0, 0, p(o['dh2']) if rp else None, M, S, Q, 0),
        (p(o['dc1']), S, 0, 0, p(t['W1t']), S, None, p(o['h1']), S,
         p(o['dh2']) if rp else None, S, p(o['dtotal']), S, 0, 0, None, M, S, S, 0),
        (p(o['dtotal']), S, 0, 0, p(t['Wst']), P_out * 32, None, None, 0, None, 0,
         p(o['dZ']), 0, P_out, M * 32, None, M, P_out * 32, S, 0)]
    return fwd, bwd


def _chain_outputs(M, S, Q, P_out):
    e = lambda *s: torch.full(s, float('nan'), device='cuda')
    return dict(h1=e(M, S), total=e(M, S), h2=e(M, S), c1=e(M, S), logits=e(M, Q),
                dc1=e(M, S), dh2=e(M, S), dtotal=e(M, S), dZ=e(P_out, M, 32))


@pytest.mark.parametrize('M,P_in,S,Q,P_out,rp', [
    (128000 // 8 + 77, 6, 256, 128, 5, False),   # ragged M, dZ N = 160 (32 columns in the last tile)
    (3000, 4, 512, 256, 6, True),                # N = 192: a dead wave half; residual_postproc
    (300, 2, 64, 32, 2, False),
    (1, 1, 16, 16, 1, True),
])
def test_gemm_nn_step_sequence_vs_float64(hip_lib, M, P_in, S, Q, P_out, rp):
    """The six row-wise dependent NN GEMMs of a training step in miniature
    (skip sum -> post1 -> post2, dc1 -> dtotal -> dZ; model.py:430-440 and its
    gradient), each launch fed by the previous one's output, every output
    against a float64 evaluation of the same sequence."""
    from wavenet import _lib
    st = torch.cuda.current_stream().cuda_stream
    t = _chain_case(M, P_in, S, Q, P_out, rp, seed=M + S)
    o = _chain_outputs(M, S, Q, P_out)
    fr, br = _chain_problems(t, o, M, P_in, S, Q, P_out, rp)
    for c in fr + br:
        _lib.call('wn_gemm_nn', *(c + (st,)))
    torch.cuda.synchronize()
    f = lambda k: t[k].cpu().numpy().astype(np.float64)
    g = lambda k: o[k].cpu().numpy().astype(np.float64)
    Z = f('Z').transpose(1, 0, 2).reshape(M, P_in * 32)
    total = Z @ f('Ws') + f('bs')
    h1 = np.maximum(total, 0)
    c1 = h1 @ f('W1') + f('b1')
    h2 = np.maximum(c1, 0) + (total if rp else 0)       # (model.py:435-437)
    logits = h2 @ f('W2') + f('b2')
    # (masks are taken from the device's own pre-activations: a ReLU within
    # rounding of its kink may legitimately fall on either side)
    m2 = (g('c1') if rp else g('h2')) > 0
    dpre = f('dlog') @ f('W2t')
    dc1 = dpre * m2
    dtotal = (dc1 @ f('W1t')) * (g('h1') > 0) + (dpre if rp else 0)
    dZ = (dtotal @ f('Wst')).reshape(M, P_out, 32).transpose(1, 0, 2)
    want = dict(h1=h1, h2=h2, logits=logits, dc1=dc1, dtotal=dtotal, dZ=dZ)
    if rp:
        want.update(total=total, c1=c1, dh2=dpre)
    for name, w in want.items():
        err = np.abs(g(name) - w).max()
        assert err < 2e-5 * max(1.0, np.abs(w).max()), (name, err)


@pytest.mark.parametrize('C', [64, 96, 128])
@pytest.mark.parametrize('rows', [1, 77, 4000 * 3 + 5])
def test_dense_planes_vs_float64(hip_lib, C, rows):
    """wn_dense_planes: out planes = addend planes + in planes * W (+ bias), the
    1x1 residual conv of the channel-block models, against float64 and against
    the plane-mode wn_gemm_nn it replaces."""
    from wavenet import _lib
    rng = np.random.default_rng(C + rows)
    CB = C // 32
    x = rng.standard_normal((CB, rows, 32)).astype(np.float32)
    add = rng.standard_normal((CB, rows, 32)).astype(np.float32)
    W = rng.standard_normal((C, C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    dx, da, dW, db = dev(x), dev(add), dev(W), dev(b)
    st = torch.cuda.current_stream().cuda_stream
    X = x.transpose(1, 0, 2).reshape(rows, C).astype(np.float64)
    A = add.transpose(1, 0, 2).reshape(rows, C).astype(np.float64)
    for use_b, use_a in ((1, 1), (0, 1), (1, 0), (0, 0)):
        out = torch.full((CB, rows, 32), float('nan'), device='cuda')
        _lib.call('wn_dense_planes', dx.data_ptr(), rows * 32, dW.data_ptr(),
                  db.data_ptr() if use_b else None, da.data_ptr() if use_a else None,
                  rows * 32, out.data_ptr(), rows * 32, rows, C, st)
        ref = X @ W.astype(np.float64) + (b if use_b else 0) + (A if use_a else 0)
        got = out.cpu().numpy().transpose(1, 0, 2).reshape(rows, C)
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    ref_g = torch.empty((CB, rows, 32), device='cuda')
    _lib.call('wn_gemm_nn', dx.data_ptr(), 0, CB, rows * 32, dW.data_ptr(), C,
              db.data_ptr(), None, 0, da.data_ptr(), 0, ref_g.data_ptr(), 0, CB,
              rows * 32, None, rows, C, C, 0, st)
    out = torch.empty((CB, rows, 32), device='cuda')
    _lib.call('wn_dense_planes', dx.data_ptr(), rows * 32, dW.data_ptr(), db.data_ptr(),
              da.data_ptr(), rows * 32, out.data_ptr(), rows * 32, rows, C, st)
    assert float((out - ref_g).abs().max()) < 1e-4 * max(1.0, float(ref_g.abs().max()))
    assert hip_lib.wn_dense_planes(dx.data_ptr(), rows * 32, dW.data_ptr(), None, None, 0,
                                   out.data_ptr(), rows * 32, rows, 160, st) == -2
    # wn_dense_planes_gate: dz = addend + in W never stored, the gate gradients
    # da_f = dz s (1 - t^2), da_g = dz s t (1 - s) instead -- bitwise what the
    # separate elementwise pass makes of the stored dz
    t = np.tanh(rng.standard_normal((CB, rows, 32))).astype(np.float32)
    sgm = (1.0 / (1.0 + np.exp(-rng.standard_normal((CB, rows, 32))))).astype(np.float32)
    dt, ds = dev(t), dev(sgm)
    dz = torch.empty((CB, rows, 32), device='cuda')
    _lib.call('wn_dense_planes', dx.data_ptr(), rows * 32, dW.data_ptr(), None,
              da.data_ptr(), rows * 32, dz.data_ptr(), rows * 32, rows, C, st)
    daf = torch.full((CB, rows, 32), float('nan'), device='cuda')
    dag = torch.full((CB, rows, 32), float('nan'), device='cuda')
    _lib.call('wn_dense_planes_gate', dx.data_ptr(), rows * 32, dW.data_ptr(), da.data_ptr(),
              rows * 32, dt.data_ptr(), ds.data_ptr(), rows * 32, daf.data_ptr(), dag.data_ptr(),
              rows * 32, rows, C, st)
    # (to rounding here: the kernel's 1 - t^2 may be one fused operation; the
    # bitwise statement is test_gpu_model.py's, against the separate device pass)
    zs = dz * ds
    for got, want in ((daf, zs * (1.0 - dt * dt)), (dag, zs * dt * (1.0 - ds))):
        assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))
    assert hip_lib.wn_dense_planes_gate(dx.data_ptr(), rows * 32, dW.data_ptr(), None, 0,
                                        None, ds.data_ptr(), rows * 32, daf.data_ptr(),
                                        dag.data_ptr(), rows * 32, rows, C, st) == -5


@pytest.mark.parametrize('rows,Mw,Nw,splits', [(4800, 160, 128, 9), (128000 // 8, 1600, 512, 25),
                                               (3333, 512, 256, 5), (2048, 512, 512, 3)])
def test_gemm_tn_spread_column_sums(hip_lib, rows, Mw, Nw, splits):
    """want_colsum = 2: every tile row of a split sums the columns of its share
    of the chunks into its own tail row; wn_reduce_slabs_mt(tail_rows) adds
    them up.  Matrix bitwise what want_colsum = 1 gives, column sums equal to
    float64 (ragged row counts: the register-staged kernel's zero rows)."""
    from wavenet import _lib
    lib = hip_lib
    rng = np.random.default_rng(rows + Mw)
    A = rng.standard_normal((rows, Mw)).astype(np.float32)
    G = rng.standard_normal((rows, Nw)).astype(np.float32)
    tr = lib.wn_gemm_tn_tail_rows(Mw, Nw)
    assert tr == Mw // (160 if Mw % 160 == 0 else 128)
    sl = lib.wn_gemm_tn_slab_floats(Mw, Nw)
    assert sl == Mw * Nw + tr * Nw
    dA, dG = dev(A), dev(G)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for want in (1, 2):
        slabs = torch.full((splits * sl,), float('nan'), device='cuda')
        out = torch.empty(Mw * Nw, device='cuda')
        cs = torch.empty(Nw, device='cuda')
        _lib.call('wn_gemm_tn', dA.data_ptr(), Mw, 0, 0, None, 0, 1, dG.data_ptr(), Nw,
                  slabs.data_ptr(), splits, rows, Mw, Nw, want, st)
        _lib.call('wn_reduce_slabs_mt', slabs.data_ptr(), splits, sl, Mw * Nw, out.data_ptr(),
                  Nw, cs.data_ptr(), 1, Nw, tr if want == 2 else 1, st)
        outs.append((out, cs))
    assert torch.equal(outs[0][0], outs[1][0])
    ref = G.astype(np.float64).sum(0)
    for _, cs in outs:
        assert np.abs(cs.cpu().numpy() - ref).max() < 2e-3
    assert lib.wn_gemm_tn(dA.data_ptr(), 100, 0, 0, None, 0, 1, dG.data_ptr(), 36,
                          slabs.data_ptr(), 2, 64, 100, 36, 2, st) == -2    # no tile rows to spread over
