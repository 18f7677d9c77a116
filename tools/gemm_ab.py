"""Same-process, interleaved A/B of wn_gemm_nn of several library builds on the
six NN shapes of a training step (with the epilogue each of them uses):
    python tools/gemm_ab.py [lib_a.so lib_b.so ...]   (default: build/ab/lib_*.so)
Outputs of every variant are compared with the first one's."""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'tensorflow-wavenet_amd')
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import _lib  # noqa: E402


def open_lib(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ('wn_gemm_nn', 'wn_gemm_tn', 'wn_gemm_tn_splits', 'wn_gemm_tn_slab_floats'):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    return lib


def tn(libs, N, reps, dev, st):
    """the three weight-gradient (TN) GEMMs of a step: dWs (A = 50 planes), dW1, dW2"""
    tot = {n: 0.0 for n, _ in libs}
    for name, Mw, Nw, pa in (('dWs', 1600, 512, 50), ('dW1', 512, 512, 0), ('dW2', 512, 256, 0)):
        A = torch.randn(N * Mw, device=dev)
        G = torch.randn(N * Nw, device=dev)
        lib0 = libs[0][1]
        sp = lib0.wn_gemm_tn_splits(N, Mw, Nw, 0)
        sl = lib0.wn_gemm_tn_slab_floats(Mw, Nw)
        slabs = torch.zeros(sp * sl, device=dev)

        def run(lib):
            code = lib.wn_gemm_tn(A.data_ptr(), 0 if pa else Mw, pa, N * 32, None, 0, 16000,
                                  G.data_ptr(), Nw, slabs.data_ptr(), sp, N, Mw, Nw,
                                  int(os.environ.get('KB_CS', 0)), st)
            assert code == 0, code
        ref = None
        times = {n: [] for n, _ in libs}
        for r in range(reps + 1):
            for n, lib in libs:
                slabs.zero_()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(4):
                    run(lib)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) * 1e3 / 4)
                elif ref is None:
                    ref = slabs.clone()
                else:
                    print('%-8s %-20s %s' % (name, n, 'bitwise' if torch.equal(slabs, ref) else
                                             'DIFFERS %.3e' % float((slabs - ref).abs().max())))
        line = '%-7s M=%4d N=%4d splits %3d:' % (name, Mw, Nw, sp)
        for n, _ in libs:
            t = float(np.median(times[n]))
            tot[n] += t
            line += '  %s %7.1f us %5.1f TF' % (n.replace('lib_', '').replace('.so', ''), t,
                                                2.0 * N * Mw * Nw / t / 1e6)
        print(line, flush=True)
    print('sum TN: ' + '  '.join('%s %.1f us' % (n, t) for n, t in tot.items()))


def main():
    paths = sys.argv[1:] or [q for q in sorted(glob.glob(os.path.join(PKG, 'build', 'ab', 'lib_*.so')))
                             if 'stamps' not in q]
    libs = [(os.path.basename(q), open_lib(q)) for q in paths]
    N = int(os.environ.get('KB_ROWS', 128000))
    reps = int(os.environ.get('KB_REPS', 7))
    dev = torch.device('cuda')
    st = torch.cuda.current_stream().cuda_stream
    if os.environ.get('KB_ONLY', 'tn') == 'tn':
        tn(libs, N, reps, dev, st)
    if os.environ.get('KB_ONLY', 'nn') != 'nn':
        return
    # (name, K, Nn, a_planes, c_planes, bias, relu, cpre, mask, addend)
    shapes = [('skip', 1600, 512, 50, 0, 1, 1, 0, 0, 0), ('post1', 512, 512, 0, 0, 1, 1, 1, 0, 0),
              ('post2', 512, 256, 0, 0, 1, 0, 0, 0, 0), ('dc1', 256, 512, 0, 0, 0, 0, 0, 1, 0),
              ('dtotal', 512, 512, 0, 0, 0, 0, 0, 1, 1), ('dZ', 512, 1600, 0, 50, 0, 0, 0, 0, 0)]
    tot = {n: 0.0 for n, _ in libs}
    for name, K, Nn, pa, pc, ub, relu, cpre, um, ua in shapes:
        A = torch.randn(N * K, device=dev)
        W = torch.randn(K * Nn, device=dev)
        C = torch.empty(N * Nn, device=dev)
        Cp = torch.empty(N * Nn, device=dev) if cpre else None
        bias = torch.randn(Nn, device=dev) if ub else None
        mask = torch.randn(N * Nn, device=dev) if um else None
        add = torch.randn(N * Nn, device=dev) if ua else None
        p = lambda t: None if t is None else t.data_ptr()

        def run(lib):
            code = lib.wn_gemm_nn(p(A), 0 if pa else K, pa, N * 32, p(W), Nn, p(bias), p(mask),
                                  Nn if um else 0, p(add), Nn if ua else 0, p(C),
                                  0 if pc else Nn, pc, N * 32, p(Cp), N, Nn, K, relu, st)
            assert code == 0, code
        ref = None
        times = {n: [] for n, _ in libs}
        for r in range(reps + 1):
            for n, lib in libs:
                C.zero_()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(4):
                    run(lib)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) * 1e3 / 4)
                else:
                    out = (C.clone(), None if Cp is None else Cp.clone())
                    if ref is None:
                        ref = out
                    else:
                        same = torch.equal(out[0], ref[0]) and (Cp is None or torch.equal(out[1], ref[1]))
                        print('%-8s %-20s %s' % (name, n, 'bitwise' if same else 'DIFFERS %.3e' % float(
                            (out[0] - ref[0]).abs().max())))
        line = '%-7s K=%4d N=%4d:' % (name, K, Nn)
        for n, _ in libs:
            t = float(np.median(times[n]))
            tot[n] += t
            line += '  %s %7.1f us %5.1f TF' % (n.replace('lib_', '').replace('.so', ''), t,
                                                2.0 * N * K * Nn / t / 1e6)
        print(line, flush=True)
    print('sum:   ' + '  '.join('%s %.1f us' % (n, t) for n, t in tot.items()))


if __name__ == '__main__':
    main()
