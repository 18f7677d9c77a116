"""Same-process, interleaved A/B of the persistent residual-stack launches
(wn_stack_fwd / wn_stack_bwd) of several library builds on ONE workspace:

    python tools/stack_ab.py [lib_a.so lib_b.so ...]     (default: build/ab/lib_*.so)
    KB_B=1 python tools/stack_ab.py ...                  (clips per batch)
    KB_VARIANTS="32:0,16:8,32:0:1" python tools/stack_ab.py lib.so
                                                         (one library, several variant words
                                                          rows:waves of the launches)
    KB_DX=1 python tools/stack_ab.py lib.so              (one library: a dx plane per layer
                                                          vs one plane rewritten in place)
    KB_SAVE_SG=0                                         (forward without the sigmoid planes)

The planes the kernels read are produced once by the default library's own
training step; every variant then runs pack + launch on the same inputs
(HIP events on the launch stream, median of KB_REPS rounds).  For the
backward the weight-gradient slabs and dx_0 of every variant are compared
with the first one (max abs difference; "bitwise" when equal)."""
import ctypes
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'tensorflow-wavenet_amd')
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import _lib, WaveNetModel  # noqa: E402
from util import model_kwargs, synth_audio  # noqa: E402


def open_lib(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ('wn_stack_pack', 'wn_stack_fwd', 'wn_stack_bwd',
                 'wn_stack_bwd_slabs', 'wn_stack_wimg_floats'):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    return lib


def main():
    paths = sys.argv[1:] or [q for q in sorted(glob.glob(os.path.join(PKG, 'build', 'ab', 'lib_*.so'))) if 'stamps' not in q]
    B = int(os.environ.get('KB_B', 8))
    T = int(os.environ.get('KB_T', 16000))
    reps = int(os.environ.get('KB_REPS', 7))
    p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
    cfg = {k: p[k] for k in p if k != 'sample_rate'}
    cfg['batch_size'] = B
    net = WaveNetModel(seed=0, **model_kwargs(cfg))
    net.use_launch_plans = False
    net.stack_bwd_keep_dx = True
    audio = synth_audio(B, T)
    net.loss(audio)
    torch.cuda.synchronize()
    ws = list(net._ws.values())[0]
    L = net.L
    P = net.params
    st = _lib.stream()
    ptr = _lib.ptr
    libs = [(os.path.basename(q), open_lib(q)) for q in paths]
    variants = {}
    if os.environ.get('KB_VARIANTS'):
        # one library, several explicit variant words of the launches; the
        # first one is the reference of the diffs
        lib = libs[0][1]
        libs = []
        for spec in os.environ['KB_VARIANTS'].split(','):
            r, w = (int(v) for v in spec.split(':')[:2])
            variants[spec] = _lib.stack_variant(rows=r, waves=w)
            libs.append((spec, lib))
    if os.environ.get('KB_DX'):
        lib = libs[0][1]
        libs = [('dx_per_layer', lib), ('dx_in_place', lib)]
    nslab = max(l.wn_stack_bwd_slabs(B, T, variants.get(n, 0)) for n, l in libs)
    slabs = torch.zeros(L, nslab, net.LAYER_BLOCK, device='cuda')
    wimg = torch.zeros(L, max(l.wn_stack_wimg_floats() for _, l in libs), device='cuda')
    bias = ws.bias_fg if net.use_biases else None

    def run_bwd(lib, name=''):
        lib.wn_stack_pack(ptr(net._layer_block(P, 0)), net.layer_stride, None,
                          ptr(wimg), L, st)
        code = lib.wn_stack_bwd(ptr(ws.X), ptr(ws.Z), ptr(ws.SG), ptr(ws.dZ),
                                ptr(ws.DX), 0 if name == 'dx_in_place' or os.environ.get('KB_INPLACE') == '1' else ws.N * 32,
                                ptr(ws.DQ), ptr(wimg), ptr(slabs),
                                slabs.shape[1] * net.LAYER_BLOCK, None,
                                ptr(net._dil_dev), ptr(ws.stack_flags_b),
                                ptr(ws.stack_ctl_b), ptr(ws.loss_parts[1:]),
                                L, B, T, variants.get(name, 0), st)
        assert code == 0, code

    def run_fwd(lib, name=''):
        lib.wn_stack_pack(ptr(net._layer_block(P, 0)), net.layer_stride,
                          ptr(wimg), None, L, st)
        code = lib.wn_stack_fwd(ptr(ws.X), ptr(ws.Z), ptr(ws.SG), ptr(wimg),
                                ptr(bias) if bias is not None else None,
                                bias.stride(0) if bias is not None else 0,
                                bias.stride(1) if bias is not None else 0,
                                ptr(net._dil_dev), ptr(ws.stack_flags),
                                ptr(ws.stack_ctl), ptr(ws.loss_parts), L, B, T,
                                int(os.environ.get('KB_SAVE_SG', 1)), variants.get(name, 0), st)
        assert code == 0, code

    for what, run in (('bwd', run_bwd), ('fwd', run_fwd)):
        if os.environ.get('KB_ONLY', what) != what:
            continue
        times = {n: [] for n, _ in libs}
        ref = None
        for r in range(reps + 1):
            for n, lib in libs:
                slabs.zero_()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                run(lib, n)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) * 1e3)
                elif what == 'bwd':
                    ns = lib.wn_stack_bwd_slabs(B, T, variants.get(n, 0))
                    g = slabs[:, :ns].double().sum(1).float()
                    out = (g.clone(), ws.DX[0].clone())
                    if ref is None:
                        ref = out
                        print('%-22s reference' % n)
                    else:
                        dg = float((out[0] - ref[0]).abs().max())
                        if os.environ.get('KB_DETAIL'):
                            sec = [('Wf0', 0, 1024), ('Wf1', 1024, 2048), ('Wg0', 2048, 3072),
                                   ('Wg1', 3072, 4096), ('Wd', 4096, 5120), ('bf', 5120, 5152),
                                   ('bg', 5152, 5184), ('bd', 5184, 5216)]
                            for nm, lo, hi in sec:
                                dd = (out[0][:, lo:hi] - ref[0][:, lo:hi]).abs().amax(1)
                                rr = ref[0][:, lo:hi].abs().amax(1)
                                rel = (dd / rr.clamp_min(1e-30))
                                k = int(rel.argmax())
                                print('   %-4s worst layer %2d: diff %.3e of %.3e (rel %.2e); median rel %.2e'
                                      % (nm, k, float(dd[k]), float(rr[k]), float(rel[k]), float(rel.median())))
                        print('%-22s slab sums: max diff %.3e of %.3e; dx_0 %s' % (
                            n, dg, float(ref[0].abs().max()),
                            'bitwise' if torch.equal(out[1], ref[1]) else
                            'DIFFERS %.3e' % float((out[1] - ref[1]).abs().max())))
        for n, _ in libs:
            t = np.asarray(times[n])
            print('%s %-22s median %.1f us  min %.1f  max %.1f  (incl. pack)' % (
                what, n, np.median(t), t.min(), t.max()))
        err = [int(ws.stack_ctl[3]), int(ws.stack_ctl_b[3])]
        assert err == [0, 0], err


if __name__ == '__main__':
    main()
