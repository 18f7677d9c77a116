"""Summarise a rocprofv3 `--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE` pass into
profiles/<tag>_mfma_util.json: per kernel, average duration, the clock the
chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration; MI355X_MICROARCH.md, DVFS) and
the matrix-pipe utilisation
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles)
(busy cycles are summed over the chip's SIMDs; one v_mfma_f32_32x32x2_f32 holds
its SIMD's pipe for 64 cycles), plus the wave-state split of SQ_WAVE_CYCLES.
usage: python tools/pmc_mfma.py <counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

SIMDS = 1024
XCDS = 8


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = r['Kernel_Name'].split('(')[0]
        disp = per[k][r['Dispatch_Id']]
        if not disp:
            disp.append({'_ns': float(r['End_Timestamp']) -
                         float(r['Start_Timestamp'])})
        disp[0][r['Counter_Name']] = float(r['Counter_Value'])
    out = {}
    for k, disps in per.items():
        rows = [d[0] for d in disps.values()]
        n = len(rows)
        avg = lambda c: sum(r.get(c, 0.0) for r in rows) / n
        ns = avg('_ns')
        cyc = avg('GRBM_GUI_ACTIVE') / XCDS
        e = {'launches': n, 'avg_us': ns / 1e3,
             'clock_ghz': cyc / ns if ns else None,
             'mfma_busy_cycles': avg('SQ_VALU_MFMA_BUSY_CYCLES'),
             'mfma_util': avg('SQ_VALU_MFMA_BUSY_CYCLES') / (SIMDS * cyc)
             if cyc else None,
             'mfma_mops_f32': avg('SQ_INSTS_VALU_MFMA_MOPS_F32'),
             'wave_cycles': avg('SQ_WAVE_CYCLES'),
             'wait_any_frac': avg('SQ_WAIT_ANY') / avg('SQ_WAVE_CYCLES')
             if avg('SQ_WAVE_CYCLES') else None,
             'wait_inst_any_frac': avg('SQ_WAIT_INST_ANY') /
             avg('SQ_WAVE_CYCLES') if avg('SQ_WAVE_CYCLES') else None,
             'active_inst_any_frac': avg('SQ_ACTIVE_INST_ANY') /
             avg('SQ_WAVE_CYCLES') if avg('SQ_WAVE_CYCLES') else None,
             'sq_busy_cycles': avg('SQ_BUSY_CYCLES')}
        out[k] = e
    json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
    tot = sum(v['avg_us'] * v['launches'] for v in out.values())
    for k in sorted(out, key=lambda k: -out[k]['avg_us'] * out[k]['launches'])[:14]:
        v = out[k]
        print('%-34s %4d x %8.1f us  clock %.2f GHz  mfma_util %.2f  '
              'wait %.2f issue-stall %.2f active %.2f' % (
                  k[:34], v['launches'], v['avg_us'], v['clock_ghz'] or 0,
                  v['mfma_util'] or 0, v['wait_any_frac'] or 0,
                  v['wait_inst_any_frac'] or 0, v['active_inst_any_frac'] or 0))


if __name__ == '__main__':
    main()
