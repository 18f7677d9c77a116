"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected:
    python tools/pmc_any.py <counter_collection.csv> [kernel-substring]"""
import collections
import csv
import sys

per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    per[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in per.items():
    print(k[:60])
    for c, v in sorted(cs.items()):
        print('   %-32s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
