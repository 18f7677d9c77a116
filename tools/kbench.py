"""Per-kernel micro-benchmarks at the bench.py shapes (development aid).
usage: python tools/kbench.py [peak] [nn] [tn] [layer]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import torch
from wavenet import _lib
lib = _lib.load()
dev = torch.device('cuda')
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def peak():
    out = torch.empty(4096 * 256, device=dev)
    for blocks in (256, 512, 1024, 2048):
        for sign, nm in ((1, 'regs'), (-1, 'lds-fed')):
            iters = 2000
            t = timeit(lambda: _lib.call('wn_diag_mfma_peak', out.data_ptr(), blocks, sign * iters, st()), n=5)
            fl = blocks * 4 * iters * 32 * 4096
            print('mfma peak %-8s blocks=%4d (%d waves/SIMD): %.1f TFLOP/s' % (nm, blocks, max(1, blocks // 256), fl / t / 1e12))


def nn():
    N = 128000
    for (K, Nn, planes_a, planes_c, name) in [(1600, 512, 50, 0, 'skip'), (512, 512, 0, 0, 'post1'),
                                               (512, 256, 0, 0, 'post2'), (256, 512, 0, 0, 'dh2'),
                                               (512, 1600, 0, 50, 'dZ')]:
        A = torch.randn(N * K, device=dev)
        W = torch.randn(K * Nn, device=dev)
        C = torch.empty(N * Nn, device=dev)
        bias = torch.randn(Nn, device=dev)
        def f():
            _lib.call('wn_gemm_nn', A.data_ptr(), 0 if planes_a else K, planes_a, N * 32, W.data_ptr(), Nn,
                      bias.data_ptr(), None, 0, None, 0, C.data_ptr(), 0 if planes_c else Nn, planes_c, N * 32,
                      None, N, Nn, K, 1, st())
        t = timeit(f)
        print('nn %-6s K=%4d N=%4d: %7.1f us  %.1f TFLOP/s' % (name, K, Nn, t * 1e6, 2.0 * N * K * Nn / t / 1e12))


def nnacc():
    """speed and accuracy (vs float64) of the NN GEMM: fp32 MFMA vs the
    split-bf16 variants (wn_gemm_nn_split, nprod 3 / 6 / 9)"""
    N = 128000
    modes = [int(m) for m in os.environ.get('KB_MODES', '0,3,6,9').split(',')]
    for (K, Nn, name) in [(1600, 512, 'skip'), (512, 512, 'post1'), (256, 512, 'dh2'), (512, 1600, 'dZ')]:
        A = torch.randn(N, K, device=dev)
        W = torch.randn(K, Nn, device=dev) * 0.05
        C = torch.empty(N, Nn, device=dev)
        scratch = torch.empty(lib.wn_gemm_split_w_bytes(K, Nn) // 4, dtype=torch.int32, device=dev)
        ref = A[:8192].double() @ W.double()
        def f(m):
            if m == 0:
                _lib.call('wn_gemm_nn', A.data_ptr(), K, 0, 0, W.data_ptr(), Nn, None, None, 0, None, 0,
                          C.data_ptr(), Nn, 0, 0, None, N, Nn, K, 0, st())
            else:
                _lib.call('wn_gemm_nn_split', A.data_ptr(), K, 0, 0, W.data_ptr(), Nn, None, None, 0, None, 0,
                          C.data_ptr(), Nn, 0, 0, None, N, Nn, K, 0, scratch.data_ptr(), m, st())
        res = {m: [] for m in modes}
        err = {}
        for rep in range(5):
            for m in modes:
                res[m].append(timeit(lambda: f(m), n=8, warm=2))
                if rep == 0:
                    d = (C[:8192].double() - ref)
                    err[m] = (d.abs().max().item(), (d.norm() / ref.norm()).item())
        line = 'nn %-6s K=%4d N=%4d:' % (name, K, Nn)
        for m in modes:
            t = sorted(res[m])[2]
            line += '  x%d %6.1f us %5.1f TF err max %.2e rel %.2e |' % (m, t * 1e6, 2.0 * N * K * Nn / t / 1e12,
                                                                     err[m][0], err[m][1])
        print(line, flush=True)


def tnacc():
    """speed and accuracy (vs float64) of the TN GEMM: fp32 MFMA vs
    wn_gemm_tn_split"""
    N = 128000
    modes = [int(m) for m in os.environ.get('KB_MODES', '0,3,6,9').split(',')]
    for (Mw, Nw, name) in [(1600, 512, 'dWs'), (512, 512, 'dW1'), (512, 256, 'dW2')]:
        A = torch.randn(N, Mw, device=dev)
        G = torch.randn(N, Nw, device=dev) * 0.05
        sl = lib.wn_gemm_tn_slab_floats(Mw, Nw)
        sp0, sp2 = lib.wn_gemm_tn_splits(N, Mw, Nw, 0), lib.wn_gemm_tn_splits(N, Mw, Nw, 2)
        slabs = torch.empty(max(sp0, sp2) * sl, device=dev)
        out = torch.empty(Mw * Nw, device=dev)
        ref = A.double().t() @ G.double()
        def f(m):
            sp = sp0 if m == 0 else sp2
            if m == 0:
                _lib.call('wn_gemm_tn', A.data_ptr(), Mw, 0, 0, None, 0, 16000, G.data_ptr(), Nw,
                          slabs.data_ptr(), sp, N, Mw, Nw, 1, st())
            else:
                _lib.call('wn_gemm_tn_split', A.data_ptr(), Mw, 0, 0, G.data_ptr(), Nw, slabs.data_ptr(), sp, N,
                          Mw, Nw, 1, m, st())
            _lib.call('wn_reduce_slabs', slabs.data_ptr(), sp, sl, 1, 0, 0, Mw * Nw, out.data_ptr(), 0, 1, 0, st())
        res = {m: [] for m in modes}
        err = {}
        for rep in range(5):
            for m in modes:
                res[m].append(timeit(lambda: f(m), n=6, warm=2))
                if rep == 0:
                    d = out.double().view(Mw, Nw) - ref
                    err[m] = (d.abs().max().item(), (d.norm() / ref.norm()).item())
        line = 'tn %-4s %dx%d (incl. slab reduce):' % (name, Mw, Nw)
        for m in modes:
            t = sorted(res[m])[2]
            line += '  x%d %6.1f us %5.1f TF err max %.2e rel %.2e |' % (m, t * 1e6, 2.0 * N * Mw * Nw / t / 1e12,
                                                                     err[m][0], err[m][1])
        print(line, flush=True)


def vendor():
    """vendor-library fp32 GEMM (torch.mm -> rocBLAS/hipBLASLt) on the same
    shapes, as a calibration of what 'good' is on this device (not shipped)"""
    N = 128000
    torch.backends.cuda.matmul.allow_tf32 = False
    for (K, Nn, name) in [(1600, 512, 'skip'), (512, 512, 'post1'), (512, 256, 'post2'), (256, 512, 'dh2'),
                          (512, 1600, 'dZ')]:
        A = torch.randn(N, K, device=dev)
        W = torch.randn(K, Nn, device=dev)
        t = timeit(lambda: torch.mm(A, W))
        print('torch.mm %-6s K=%4d N=%4d: %7.1f us  %.1f TFLOP/s' % (name, K, Nn, t * 1e6, 2.0 * N * K * Nn / t / 1e12))
    for (Mw, Nw, name) in [(1600, 512, 'dWs'), (512, 512, 'dW1'), (512, 256, 'dW2')]:
        A = torch.randn(N, Mw, device=dev)
        G = torch.randn(N, Nw, device=dev)
        t = timeit(lambda: torch.mm(A.t(), G))
        print('torch.mm^T %-4s %dx%d: %7.1f us  %.1f TFLOP/s' % (name, Mw, Nw, t * 1e6, 2.0 * N * Mw * Nw / t / 1e12))


def tn():
    """weight-gradient GEMMs; splits as the model picks them and a few more"""
    N = 128000
    for (Mw, Nw, planes, name, sps) in [(1600, 512, 50, 'dWs', (13, 19, 25, 38, 51)), (512, 512, 0, 'dW1', (32, 48, 64, 96, 128)),
                                        (512, 256, 0, 'dW2', (64, 96, 128, 192))]:
        A = torch.randn(N * Mw, device=dev)
        G = torch.randn(N * Nw, device=dev)
        sl = lib.wn_gemm_tn_slab_floats(Mw, Nw)
        for sp in sps:
            slabs = torch.empty(sp * sl, device=dev)
            def f():
                _lib.call('wn_gemm_tn', A.data_ptr(), 0 if planes else Mw, planes, N * 32, None, 0, 16000,
                          G.data_ptr(), Nw, slabs.data_ptr(), sp, N, Mw, Nw, 1, st())
            t = sorted(timeit(f, n=6, warm=2) for _ in range(5))[2]
            print('tn %-4s %dx%d splits=%3d: %7.1f us %5.1f TF' % (name, Mw, Nw, sp, t * 1e6, 2.0 * N * Mw * Nw / t / 1e12), flush=True)


def layer():
    B, T = 8, 16000
    N = B * T
    mk = lambda: torch.randn(N * 32, device=dev)
    x, xo, z, th, sg, dz, f, g, f2, g2, dx, dxo = [mk() for _ in range(12)]
    w = torch.randn(5216, device=dev) * 0.1
    wimg = torch.randn(5376, device=dev) * 0.1
    nslab = 512
    slabs = torch.empty(nslab * 5216, device=dev)
    for d in (1, 64, 512):
        t = timeit(lambda: _lib.call('wn_layer_fwd', x.data_ptr(), xo.data_ptr(), z.data_ptr(), th.data_ptr(),
                                     sg.data_ptr(), w.data_ptr(), None, 0, B, T, d, 1, 1, st()), n=20)
        print('layer_fwd d=%3d: %6.1f us' % (d, t * 1e6), flush=True)


def bwd2():
    """wn_layer_bwd2 at several batch sizes: per-tile latency vs throughput"""
    T = 16000
    for B in (1, 2, 4, 8, 16):
        N = B * T
        mk = lambda: torch.randn(N * 32, device=dev)
        x, z, dZ, dxin, dxo, f, g, th, fn, gn = [mk() for _ in range(10)]
        sg = torch.rand(N * 32, device=dev) * 0.9 + 0.05
        w = torch.randn(5216, device=dev) * 0.1
        wimg = torch.randn(5376, device=dev) * 0.1
        for d in (4, 512):
            line = 'B=%2d d=%3d:' % (B, d)
            nsl = lib.wn_layer_bwd2_slabs(B, T)
            slabs = torch.empty(nsl * 5216, device=dev)
            t = timeit(lambda: _lib.call('wn_layer_bwd2', x.data_ptr(), z.data_ptr(), sg.data_ptr(), dZ.data_ptr(),
                                         dxin.data_ptr(), dxo.data_ptr(), w.data_ptr(), wimg.data_ptr(), slabs.data_ptr(), None,
                                         B, T, d, st()), n=20, warm=3)
            line += '  bwd2 %6.1f us' % (t * 1e6)
            print(line, flush=True)


def lk():
    """the two default layer kernels at the bench shape (B=8, T=16000):
    wn_layer_fwd (sigmoid plane only) and wn_layer_bwd2; median of 5 x 40"""
    B, T = int(os.environ.get('KB_B', 8)), 16000
    N = B * T
    mk = lambda: torch.randn(N * 32, device=dev)
    x, xo, z, dZ, dxin, dxo = [mk() for _ in range(6)]
    sg = torch.rand(N * 32, device=dev) * 0.9 + 0.05
    w = torch.randn(5216, device=dev) * 0.1
    wimg = torch.randn(5376, device=dev) * 0.1
    slabs = torch.empty(lib.wn_layer_bwd2_slabs(B, T) * 5216, device=dev)
    for d in (4, 512):
        f = lambda: _lib.call('wn_layer_fwd', x.data_ptr(), xo.data_ptr(), z.data_ptr(), None, sg.data_ptr(),
                              w.data_ptr(), None, 0, B, T, d, 1, 2, st())
        b = lambda: _lib.call('wn_layer_bwd2', x.data_ptr(), z.data_ptr(), sg.data_ptr(), dZ.data_ptr(),
                              dxin.data_ptr(), dxo.data_ptr(), w.data_ptr(), wimg.data_ptr(), slabs.data_ptr(), None, B, T, d, st())
        only = os.environ.get('KB_LK', '')          # 'fwd' / 'bwd': one kernel only
        tf = sorted(timeit(f, n=40, warm=3) for _ in range(5))[2] if only != 'bwd' else 0.0
        tb = sorted(timeit(b, n=40, warm=3) for _ in range(5))[2] if only != 'fwd' else 0.0
        print('B=%d d=%3d: fwd %6.2f us   bwd2 %6.2f us' % (B, d, tf * 1e6, tb * 1e6), flush=True)


def layerpad():
    """are the layer kernels sensitive to the relative alignment of their
    planes?  (12 planes carved from one buffer at stride N*32 + pad floats)"""
    B, T = 8, 16000
    N = B * T
    w = torch.randn(5216, device=dev) * 0.1
    wimg = torch.randn(5376, device=dev) * 0.1
    for pad in (0, 64, 1088, 8256, 65600, 524352):
        big = torch.randn(12 * (N * 32 + pad) + 64, device=dev)
        pl = [big[i * (N * 32 + pad): i * (N * 32 + pad) + N * 32] for i in range(12)]
        x, xo, z, th, sg, dz, f, g, f2, g2, dx, dxo = pl
        d = 64
        tf = timeit(lambda: _lib.call('wn_layer_fwd', x.data_ptr(), xo.data_ptr(), z.data_ptr(), th.data_ptr(),
                                      sg.data_ptr(), w.data_ptr(), None, 0, B, T, d, 1, 1, st()), n=30)
        print('pad %7d floats: fwd %5.1f us' % (pad, tf * 1e6), flush=True)


def nnsmall():
    # same GEMM at sizes whose A operand fits the 256 MB Infinity Cache
    for N in (65536, 128000, 131072, 128000 + 512):
        K, Nn = 1600, 512
        A = torch.randn(N * K, device=dev)
        W = torch.randn(K * Nn, device=dev)
        C = torch.empty(N * Nn, device=dev)
        def f():
            _lib.call('wn_gemm_nn', A.data_ptr(), 0, 50, N * 32, W.data_ptr(), Nn, None, None, 0, None, 0,
                      C.data_ptr(), Nn, 0, 0, None, N, Nn, K, 1, st())
        t = timeit(f, n=20)
        print('nn skip M=%6d (A %4d MB): %7.1f us  %.1f TFLOP/s' % (N, N * K * 4 >> 20, t * 1e6, 2.0 * N * K * Nn / t / 1e12))


if __name__ == '__main__':
    which = sys.argv[1:] or ['peak', 'nn', 'tn', 'layer']
    for w in which:
        globals()[w]()

