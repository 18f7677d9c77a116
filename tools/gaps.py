"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV:
per stream (queue) and overall, over the last `tail` kernels.
    python tools/gaps.py <kernel_trace.csv> [tail]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-tail:]
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
busy, cur_end, gaps = 0, t0, []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > cur_end:
        gaps.append((s - cur_end, r['Kernel_Name'][:50]))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
wall = t1 - t0
print('kernels %d  wall %.1f us  some kernel running %.1f us  idle %.1f us (%.1f%%) in %d gaps, median gap %.2f us' % (
    len(rows), wall / 1e3, busy / 1e3, (wall - busy) / 1e3, 100.0 * (wall - busy) / wall, len(gaps),
    sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
by = defaultdict(list)
for g, n in gaps:
    by[n].append(g)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print('  before %-50s %4d gaps, mean %.2f us, total %.1f us' % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
