"""Print (kernel, calls, average us) rows of a rocprofv3 *kernel_stats.csv,
optionally only the kernels whose name contains one of the given substrings.
    python tools/kstats.py <kernel_stats.csv> [substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    if pats and not any(p in r['Name'] for p in pats):
        continue
    print('%-62s %6s %10.1f us' % (r['Name'][:62], r['Calls'], float(r['AverageNs']) / 1e3))
