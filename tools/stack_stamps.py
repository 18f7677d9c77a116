"""In-kernel phase stamps of stack_fwd_kernel (diagnostic build -DSTACK_STAMPS,
loaded through WN_LIB_PATH): per layer, for the first and the last wave of
every workgroup."""
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
B, T = int(os.environ.get('KB_B', 8)), 16000
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = B
net = WaveNetModel(seed=0, **model_kwargs(cfg))
net.use_launch_plans = False
L = net.L
# the forward launch's grid, as wn_stack_fwd chooses it (the stamp buffer is
# indexed by workgroup: it must not be smaller than the launch)
ntiles, best, fw = B * ((T + 31) // 32), -1, 16
for w in (16, 8, 4, 2, 1):
    groups = (ntiles + w - 1) // w
    cost = ((groups + 255) // 256) * max(w * 22 // 4, 70)
    if best < 0 or cost <= best:
        best, fw = cost, w
grid = min(256, (ntiles + fw - 1) // fw)
dbg = torch.zeros(grid * 16 * L * 12 + grid * 4, dtype=torch.int64, device='cuda')
lib.wn_diag_stack_dbg.argtypes = [ctypes.c_void_p]
lib.wn_diag_stack_dbg(dbg.data_ptr())
gridb = min(256, lib.wn_stack_bwd_slabs(B, T, net.stack_variant))
dbgb = torch.zeros(gridb * 8 * L * 16 + gridb * 4, dtype=torch.int64, device='cuda')
lib.wn_diag_stack_dbg_b.argtypes = [ctypes.c_void_p]
lib.wn_diag_stack_dbg_b(dbgb.data_ptr())
audio = synth_audio(B, T)
fwd_only = os.environ.get('KB_FWDONLY', '0') == '1'     # no sigmoid planes (SAVE = 0)
for it in range(4):
    dbg.zero_()
    dbgb.zero_()
    net.loss(audio, backward=not fwd_only)
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
s = raw[:grid * 16 * L * 12].reshape(grid, 16, L, 12).astype(np.float64)
cal = raw[grid * 16 * L * 12:].reshape(grid, 4).astype(np.float64)
clk = np.median((cal[:, 3] - cal[:, 1]) / ((cal[:, 2] - cal[:, 0]) * 10.0))    # GHz
print('clock %.2f GHz; kernel (entry -> exit of a workgroup) median %.1f us, max %.1f us' % (
    clk, np.median(cal[:, 2] - cal[:, 0]) / 100.0, (cal[:, 2].max() - cal[:, 0].min()) / 100.0))
names = ['top of layer', 'weights barrier', 'flags seen', "x[t-d] in fragments", '64 conv MFMAs',
         'tanh/sigmoid, z + sigmoid stores issued', 'dense MFMAs', "x' stored, drained, flag posted",
         'weight ring bookkeeping (+ refill by the last wave)', 'z store issued', 'sigmoid store issued']
for wv in sorted({0, fw - 1}):
    print('--- wave %d of the workgroup: median over workgroups and layers 1..L-2, us per phase' % wv)
    tot = 0
    for i in range(1, 11):
        dt = (s[:, wv, 1:L - 1, i] - s[:, wv, 1:L - 1, i - 1]) / clk / 1e3
        print('%-42s %6.2f   (p90 %6.2f)' % (names[i], np.median(dt), np.percentile(dt, 90)))
        tot += np.median(dt)
    per = (s[:, wv, 2:L - 1, 0] - s[:, wv, 1:L - 2, 0]) / clk / 1e3
    print('%-42s %6.2f   (sum of medians %.2f)' % ('layer period', np.median(per), tot))
print('--- per wave: median us of flag wait / conv / dense / whole layer minus flag wait')
for wv in range(fw):
    f = (s[:, wv, 1:L - 1, 2] - s[:, wv, 1:L - 1, 1]) / clk / 1e3
    c = (s[:, wv, 1:L - 1, 4] - s[:, wv, 1:L - 1, 3]) / clk / 1e3
    dn = (s[:, wv, 1:L - 1, 6] - s[:, wv, 1:L - 1, 5]) / clk / 1e3
    per = (s[:, wv, 2:L - 1, 0] - s[:, wv, 1:L - 2, 0]) / clk / 1e3
    print('wave %2d (SIMD %d): flags %5.2f  conv %5.2f  dense %5.2f  busy %5.2f  period %5.2f' % (
        wv, wv & 3, np.median(f), np.median(c), np.median(dn), np.median(per) - np.median(f), np.median(per)))
# by dilation
dil = net.dilations
print('--- layer period by dilation (wave 0, median over workgroups)')
per = (s[:, 0, 1:, 0] - s[:, 0, :-1, 0]) / clk / 1e3
for l in range(0, L - 1):
    if l < 12 or l % 10 == 0:
        wf = (s[:, 0, l, 2] - s[:, 0, l, 1]) / clk / 1e3
        print('layer %2d d=%3d: period %6.2f us, flag wait %5.2f us' % (l, dil[l], np.median(per[:, l]), np.median(wf)))

# ---------------------------------------------------------------- backward
if not fwd_only:
    raw = dbgb.cpu().numpy()
    s = raw[:gridb * 8 * L * 16].reshape(gridb, 8, L, 16).astype(np.float64)
    cal = raw[gridb * 8 * L * 16:].reshape(gridb, 4).astype(np.float64)
    clk = np.median((cal[:, 3] - cal[:, 1]) / ((cal[:, 2] - cal[:, 0]) * 10.0))
    print('=== stack_bwd_kernel: clock %.2f GHz; workgroup entry -> exit median %.1f us, whole launch %.1f us' % (
        clk, np.median(cal[:, 2] - cal[:, 0]) / 100.0, (cal[:, 2].max() - cal[:, 0].min()) / 100.0))
    seq = [(0, 1, 'layer top: wait for the weight image'), (1, 11, 'tile start'),
           (11, 12, 'rows-t DMA + tap flags requested'), (12, 14, 'wait for the rows-t DMA'),
           (14, 15, 'dWd (16 MFMA), fragments of rows t'),
           (15, 8, 'x DMA issued, rows-t math (48 MFMA + gates)'),
           (8, 10, 'x tiles in; tap flags checked, rows t+d requested'),
           (10, 13, 'weight-gradient MFMAs (64)'),
           (13, 9, 'rows t+d: wait, LDS transit, 48 MFMA + gates, dx stored, drained, flag'),
           (9, 7, 'tile end -> second tile start'),
           (7, 3, 'second tile'), (3, 4, 'ring bookkeeping'), (4, 5, 'ordered accumulation chain (incl. token waits)'),
           (5, 6, 'refill drain + ready mark (refilling wave only)')]
    for wv in (0, 3, 7):
        print('--- wave %d: median over workgroups and layers 1..L-2 (p90), us' % wv)
        for a_, b_, nm in seq:
            dt = (s[:, wv, 1:L - 1, b_] - s[:, wv, 1:L - 1, a_]) / clk / 1e3
            dt = dt[(s[:, wv, 1:L - 1, b_] > 0) & (s[:, wv, 1:L - 1, a_] > 0)]
            if dt.size:
                print('%-46s %6.2f   (%6.2f)' % (nm, np.median(dt), np.percentile(dt, 90)))
        per = (s[:, wv, 1:L - 2, 0] - s[:, wv, 2:L - 1, 0]) / clk / 1e3     # layers run downwards
        print('%-46s %6.2f' % ('layer period', np.median(per)))

    # raw timeline of one workgroup, one layer: every wave's stamps in us from
    # the earliest one (which phases of the two waves of a SIMD coincide?)
    if os.environ.get('KB_TIMELINE'):
        g, l = gridb // 2, L // 2
        order = [0, 1, 11, 12, 14, 15, 8, 10, 13, 9, 7, 3, 4, 5, 6]
        t0 = s[g, :, l, 0].min()
        print('--- workgroup %d, layer %d: stamps (us) in program order: %s' % (g, l, order))
        for wv in range(8):
            print('wave %d: ' % wv + ' '.join('%6.2f' % ((s[g, wv, l, i] - t0) / clk / 1e3) for i in order))
