"""Summarise a rocprofv3 `--kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
GRBM_GUI_ACTIVE` pass into
profiles/<tag>_issue.json: per kernel, what its SIMDs' issue slots were spent
on.  The residual-stack launches are bound by neither HBM nor the matrix pipe
alone but by issue: on gfx950 the f32 MFMA and the vector ALU are one resource
(profiles/*_mfma_valu.txt: a vector instruction under a running MFMA issues at
about a third of its rate), so the figure of merit is
    issue_frac = (MFMA busy cycles + 4 x vector instructions) / (1024 SIMDs x
                 elapsed cycles)
(a wave64 vector instruction holds its SIMD's 16 lanes for 4 cycles, a
v_mfma_f32_32x32x2_f32 for 64).  A counter the pass could not collect is
reported as null, never guessed.
usage: python tools/pmc_issue.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash  # noqa: E402

SIMDS = 1024
XCDS = 8


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(sys.argv[1])):
        k = r['Kernel_Name'].split('(')[0]
        d = per[k][r['Dispatch_Id']]
        d['_ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        d[r['Counter_Name']] = float(r['Counter_Value'])
    out = {}
    for k, disps in per.items():
        rows = list(disps.values())
        n = len(rows)

        def avg(c):
            v = [r[c] for r in rows if c in r]
            return sum(v) / len(v) if v else None
        ns, gui = avg('_ns'), avg('GRBM_GUI_ACTIVE')
        cyc = gui / XCDS if gui else None
        slots = SIMDS * cyc if cyc else None
        valu, mfma = avg('SQ_INSTS_VALU'), avg('SQ_VALU_MFMA_BUSY_CYCLES')
        e = {'launches': n, 'avg_us': ns / 1e3, 'clock_ghz': cyc / ns if cyc else None,
             'insts_valu': valu, 'insts_lds': avg('SQ_INSTS_LDS'),
             'mfma_mops_f32': avg('SQ_INSTS_VALU_MFMA_MOPS_F32'),
             'mfma_busy_frac': mfma / slots if mfma is not None and slots else None,
             'valu_issue_frac': 4.0 * valu / slots if valu is not None and slots else None,
             'active_inst_valu_frac': (avg('SQ_ACTIVE_INST_VALU') / slots)
             if avg('SQ_ACTIVE_INST_VALU') is not None and slots else None}
        # (round 6: SQ_INST_CYCLES_VMEM and SQ_VALU_MFMA_COEXEC_CYCLES do not
        # count on gfx950 -- null / 0.0 for every kernel incl. GEMMs at 0.88
        # MFMA busy -- and are no longer collected or reported; the "f32 MFMA and
        # vector ALU are one issue resource" statement rests on the
        # microbenchmark profiles/*_mfma_valu.txt alone)
        e['issue_frac'] = (e['mfma_busy_frac'] + e['valu_issue_frac']) \
            if e['mfma_busy_frac'] is not None and e['valu_issue_frac'] is not None else None
        out[k] = e
    for k in sorted(out, key=lambda k: -out[k]['avg_us'] * out[k]['launches'])[:12]:
        v = out[k]
        f = lambda x: ' n/a' if x is None else '%.2f' % x
        print('%-34s %4d x %8.1f us  mfma %s  valu %s  issue %s' % (
            k[:34], v['launches'], v['avg_us'], f(v['mfma_busy_frac']),
            f(v['valu_issue_frac']), f(v['issue_frac'])))
    out['_meta'] = {'csrc_sha16': csrc_hash()}
    json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
