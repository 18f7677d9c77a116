"""Per-kernel roofline table from the committed rocprofv3 summaries:
profiles/<tag>_bench_kernel_stats.csv (--kernel-trace --stats) and
profiles/<tag>_pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes,
tools/pmc_summary.py).  usage: python tools/roofline_table.py r01_v6 > profiles/r01_v8_roofline_table.md

Algorithmic FLOPs per launch at the bench shape (B*T = 128000, default stack):
  gemm_nn3: 620.757 GFLOP per step over 6 launches (skip, post1, post2 and
            their data gradients); gemm_tn3<5,1>: dWs; <4,2>: dW1 and dW2;
  layer_fwd: 80 MFMA / 32-row tile = 10 240 FLOP per audio sample;
  (layer_bwdw: the round-1 fused backward, 160 MFMA / tile, removed in round 6;
   its rows in the round-1 tables stay readable through the entry below)
  layer_bwd2d: 176 MFMA / tile = 22 528 FLOP per audio sample;
  stack_fwd / stack_bwd: the same per layer, all 50 layers in one launch.
With profiles/<tag>_mfma_util.json (tools/pmc_mfma.py) the table also carries
the matrix-pipe utilisation in CYCLES and the clock the chip held.
"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01_v6'
N = 128000
FLOPS = {
    'gemm_nn3_kernel': 620.756992e9 / 6,
    'gemm_tn3_kernel<5, 1>': 2.0 * N * 1600 * 512,
    'gemm_tn3_kernel<4, 2>': (2.0 * N * 512 * 512 + 2.0 * N * 512 * 256) / 2,
    'layer_fwd_kernel<true, true>': N * 10240.0,
    'layer_bwdw_kernel<true, true>': N * 20480.0,
    'layer_fwd_kernel<true, 2>': N * 10240.0,
    'layer_bwd2d_kernel<true>': N * 22528.0,
    # the persistent whole-stack launches (50 layers; the last layer has no
    # dense 1x1: 64 instead of 80 MFMAs forward, 128 instead of 176 backward)
    'stack_fwd_kernel<2, 16>': N * (49 * 10240.0 + 8192.0),
    'stack_bwd_kernel<8>': N * (49 * 22528.0 + 16384.0),
    # "push" formulation (round 3): 160 MFMA / tile, the algorithmic minimum
    'stack_bwd_kernel<8, true>': N * (49 * 20480.0 + 16384.0),
    'stack_bwd_kernel<8, false>': N * (49 * 22528.0 + 16384.0),
}
MFMA_PEAK, HBM_SPEC, HBM_STREAM = 157.3e12, 8.0e12, 5.3e12
rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', tag + '_bench_kernel_stats.csv'))))
pmc = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc_traffic.json')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
mu_path = os.path.join(ROOT, 'profiles', tag + '_mfma_util.json')
mfma = json.load(open(mu_path)) if os.path.exists(mu_path) else {}


def key(name):
    n = name.split('(')[0].replace('void ', '').strip()
    return n


print('# Per-kernel rooflines, `bench.py` at B=8, T=16000 (%s)\n' % tag)
print('Source: `profiles/%s_bench_kernel_stats.csv` (rocprofv3 --kernel-trace --stats), '
      '`profiles/%s_pmc_traffic.json` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 per the gfx950 '
      'correction).  MFMA peak 157.3 TFLOP/s (fp32 dense), HBM 8.0 TB/s spec; a streaming copy of '
      'non-cache-resident data measures 5.3 TB/s on this part (`tools/hbm_bw.py`).\n' % (tag, tag))
if mfma:
    print('MFMA utilisation: `profiles/%s_mfma_util.json` (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... '
          'GRBM_GUI_ACTIVE with --kernel-trace; busy cycles / (1024 SIMDs x elapsed cycles); the clock is '
          'GRBM_GUI_ACTIVE / 8 / duration and reads high on dispatches shorter than ~0.3 ms).\n' % tag)
print('| kernel | launches/step | avg us | % of GPU time | TFLOP/s | frac MFMA peak | MFMA pipe busy (cycles) | clock GHz | HBM MB/launch (PMC) | TB/s | frac of 8.0 spec | frac of 5.3 streaming | bound |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
steps = None
for r in rows:
    if 'adam_kernel' in r['Name']:
        steps = int(r['Calls'])
for r in rows[:12]:
    k = key(r['Name'])
    avg = float(r['AverageNs']) * 1e-9
    share = float(r['TotalDurationNs']) / tot * 100
    fl = FLOPS.get(k)
    p = None
    for kk, v in pmc.items():
        if key(kk) == k:
            p = v
    tf = ('%.1f' % (fl / avg / 1e12)) if fl else '-'
    fr = ('%.2f' % (fl / avg / MFMA_PEAK)) if fl else '-'
    if p:
        hb = p['hbm_bytes_per_launch']
        mb, tb = '%.1f' % (hb / 1e6), '%.2f' % (hb / avg / 1e12)
        f1, f2 = '%.2f' % (hb / avg / HBM_SPEC), '%.2f' % (hb / avg / HBM_STREAM)
    else:
        mb = tb = f1 = f2 = '-'
    bound = 'MFMA' if k.startswith('gemm') else ('latency / MFMA' if k.startswith(('layer', 'stack')) else 'HBM')
    if k.startswith('stack') and p and hb / avg / HBM_STREAM >= 0.8:
        # a persistent stack launch that moves its bytes at >= 0.8 of what a
        # streaming copy reaches on this part is bound by memory, not latency
        bound = 'HBM (streaming rate) / MFMA issue'
    mu = ck = '-'
    for kk, v in mfma.items():
        if key(kk) == k and v.get('mfma_util') is not None:
            mu = '%.2f' % v['mfma_util']
            ck = '%.2f' % v['clock_ghz'] if v['avg_us'] > 300 else '(%.2f)' % v['clock_ghz']
    print('| `%s` | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s |' % (
        k, int(r['Calls']) / float(steps or 1), avg * 1e6, share, tf, fr, mu, ck, mb, tb, f1, f2, bound))
