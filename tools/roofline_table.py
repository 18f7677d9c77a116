"""Per-kernel roofline table from the committed rocprofv3 summaries:
profiles/<tag>_bench_kernel_stats.csv (--kernel-trace --stats) and
profiles/<tag>_pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes,
tools/pmc_summary.py).  usage: python tools/roofline_table.py r01_v6 > profiles/r01_v8_roofline_table.md

Algorithmic FLOPs per launch at the bench shape (B*T = 128000, default stack):
  gemm_nn3: 620.757 GFLOP per step over 6 launches (skip, post1, post2 and
            their data gradients); gemm_tn3<5,1>: dWs; <4,2>: dW1 and dW2;
  layer_fwd: 80 MFMA / 32-row tile = 10 240 FLOP per audio sample;
  layer_bwdw: 160 MFMA / tile = 20 480 FLOP per audio sample.
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
}
MFMA_PEAK, HBM_SPEC, HBM_STREAM = 157.3e12, 8.0e12, 5.3e12
rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', tag + '_bench_kernel_stats.csv'))))
pmc = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc_traffic.json')))
tot = sum(float(r['TotalDurationNs']) for r in rows)


def key(name):
    n = name.split('(')[0].replace('void ', '').strip()
    return n


print('# Per-kernel rooflines, `bench.py` at B=8, T=16000 (%s)\n' % tag)
print('Source: `profiles/%s_bench_kernel_stats.csv` (rocprofv3 --kernel-trace --stats), '
      '`profiles/%s_pmc_traffic.json` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 per the gfx950 '
      'correction).  MFMA peak 157.3 TFLOP/s (fp32 dense), HBM 8.0 TB/s spec; a streaming copy of '
      'non-cache-resident data measures 5.3 TB/s on this part (`tools/hbm_bw.py`).\n' % (tag, tag))
print('| kernel | launches/step | avg us | % of GPU time | TFLOP/s | frac MFMA peak | HBM MB/launch (PMC) | TB/s | frac of 8.0 spec | frac of 5.3 streaming | bound |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
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
    bound = 'MFMA' if k.startswith('gemm') else 'HBM'
    print('| `%s` | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s |' % (
        k, int(r['Calls']) / float(steps or 1), avg * 1e6, share, tf, fr, mb, tb, f1, f2, bound))
