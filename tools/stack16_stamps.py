"""In-kernel phase stamps of the 16-row stack launches (stack_fwd16_kernel /
stack_bwd16_kernel; diagnostic build -DSTACK_STAMPS): medians over workgroups
and layers, us per phase.  KB_B (default 1) clips of 16000 samples."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'tensorflow-wavenet_amd')
if 'WN_LIB_PATH' not in os.environ:
    out = os.path.join(PKG, 'build', 'ab', 'lib_stack_stamps.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(PKG, 'csrc', f) for f in
            ('wn_layer.hip', 'wn_stack.hip', 'wn_gemm.hip', 'wn_misc.hip', 'wn_fastgen.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                           '-DSTACK_STAMPS', '-shared', '-o', out] + os.environ.get('KB_DEFS', '').split() + srcs)
    os.environ['WN_LIB_PATH'] = out
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        sys.exit(0)
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes
import json
import numpy as np
import torch
from wavenet import _lib, WaveNetModel
from util import model_kwargs, synth_audio
lib = _lib.load()
# KB_WAVES_B: waves of the 16-row backward launch
rw = int(os.environ.get('KB_WAVES_B', 8))
VAR = _lib.stack_variant(rows=16, waves=rw)
WaveNetModel.DEFAULT_STACK_VARIANT = VAR
B, T = int(os.environ.get('KB_B', 1)), 16000
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = B
net = WaveNetModel(seed=0, **model_kwargs(cfg))
net.use_launch_plans = False
if os.environ.get('KB_OVERLAP_TN'):
    net.overlap_tn = os.environ['KB_OVERLAP_TN'] == '1'
L = net.L
# the launches' grids, as wn_stack_fwd / wn_stack_bwd choose them (the
# calibration words sit behind gridDim.x workgroups' stamps)
nt16 = B * ((T + 15) // 16)
wf = rw        # (one variant word serves both launches)
gridf = min(256, (nt16 + wf - 1) // wf)
gridb = min(256, lib.wn_stack_bwd_slabs(B, T, VAR))
dbg = torch.zeros(gridf * 16 * L * 16 + gridf * 4, dtype=torch.int64, device='cuda')
dbgb = torch.zeros(gridb * 16 * L * 16 + gridb * 4, dtype=torch.int64, device='cuda')
lib.wn_diag_stack_dbg.argtypes = [ctypes.c_void_p]
lib.wn_diag_stack_dbg(dbg.data_ptr())
lib.wn_diag_stack_dbg_b.argtypes = [ctypes.c_void_p]
lib.wn_diag_stack_dbg_b(dbgb.data_ptr())
audio = synth_audio(B, T)
for it in range(4):
    dbg.zero_()
    dbgb.zero_()
    net.loss(audio)
torch.cuda.synchronize()


def report(raw, grid, title, seq, down, waves=None, first=0):
    s = raw[:grid * 16 * L * 16].reshape(grid, 16, L, 16).astype(np.float64)
    cal = raw[grid * 16 * L * 16:].reshape(grid, 4).astype(np.float64)
    used = cal[:, 2] > 0
    cal = cal[used]
    clk = np.median((cal[:, 3] - cal[:, 1]) / ((cal[:, 2] - cal[:, 0]) * 10.0))
    print('=== %s: %d workgroups, clock %.2f GHz; workgroup entry -> exit median %.1f us, whole launch %.1f us' % (
        title, int(used.sum()), clk, np.median(cal[:, 2] - cal[:, 0]) / 100.0,
        (cal[:, 2].max() - cal[:, 0].min()) / 100.0))
    s = s[used]
    nw = int((s[:, :, L // 2, 0] > 0).any(axis=0).sum())
    for wv in (waves if waves is not None else sorted({0, nw - 1})):
        tot = 0.0
        print('--- wave %d of %d: median (p90) over workgroups and layers 1..L-2, us' % (wv, nw))
        for a_, b_, nm in seq:
            ok = (s[:, wv, 1:L - 1, b_] > 0) & (s[:, wv, 1:L - 1, a_] > 0)
            dt = ((s[:, wv, 1:L - 1, b_] - s[:, wv, 1:L - 1, a_]) / clk / 1e3)[ok]
            if dt.size:
                print('%-58s %6.2f   (%6.2f)' % (nm, np.median(dt), np.percentile(dt, 90)))
                tot += np.median(dt)
        if down:
            per = (s[:, wv, 1:L - 2, 0] - s[:, wv, 2:L - 1, 0]) / clk / 1e3
        else:
            per = (s[:, wv, 2:L - 1, 0] - s[:, wv, 1:L - 2, 0]) / clk / 1e3
        print('%-58s %6.2f   (sum of medians %.2f)' % ('layer period', np.median(per), tot))


report(dbg.cpu().numpy(), gridf, 'stack_fwd16_kernel', [
  (0, 1, 'wait for the weight ring'), (1, 2, 'bias, tap requested, 32 current-tap MFMAs'),
  (2, 3, 'tap rows in registers (flag wait + load, or word poll)'),
  (3, 4, '32 past-tap MFMAs'), (4, 5, 'tanh / sigmoid'), (5, 6, 'dense bias + 16 MFMAs'),
  (6, 7, "x' out (stored, drained, flag posted / words stored)"),
  (7, 8, 'z / sigmoid stores issued, ring bookkeeping')], False)
if os.environ.get('KB_WAVES_B') == '4':
    # the fused skip sum's partner waves (wn_stack_fwd_skip: waves 4 .. 7)
    report(dbg.cpu().numpy(), gridf, 'stack_fwd16_kernel<.., skip>, partner waves', [
        (0, 1, "the four tiles' z of the layer in registers (waits for the chain waves)"),
        (1, 3, '8 n-tiles x 4 tiles: 256 MFMAs, operand stream')], False, waves=(4, 7))
report(dbgb.cpu().numpy(), gridb, 'stack_bwd16_kernel', [
    (0, 1, 'wait for the weight ring'), (1, 2, 'loads requested (z DMA, dZ, sigmoid, own dx, flag check, q)'),
    (2, 3, 'wait for them'), (3, 4, 'dx_{l+1} to LDS, dWd (8 MFMA), z fragments'),
    (4, 5, 'x DMA issued, dz (16 MFMA), gate derivatives'),
    (5, 6, 'own / q rows (64 MFMA)'), (6, 7, 'q and own rows stored, drained, flag posted, next flags requested'),
    (7, 8, 'weight gradients (32 MFMA)'), (8, 9, 'ring bookkeeping'),
    (9, 10, 'ordered accumulation chain (incl. token waits)')], True)
