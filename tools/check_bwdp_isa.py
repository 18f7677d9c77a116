"""Static check of stack_bwdp_kernel's machine code (csrc/wn_stack.hip).

The matrix waves of that kernel issue their A-operand loads as inline asm with
a hand-counted `s_waitcnt vmcnt(31)` in front of every MFMA (the compiler's own
wait-count pass drained every outstanding load at some point of the ticket
loop, differently from build to build: DESIGN.md).  The compiler does not know
that those registers are written asynchronously, so it must not touch them
between the load and its wait.  This tool compiles the file to assembly and
checks, inside the kernel's matrix loop:
  * a register written by an inline-asm `buffer_load_dword` is read only by
    `v_mfma` instructions that directly follow an `s_waitcnt vmcnt(31)`, and
    written only by those loads (no copy, no spill, no reuse as a temporary);
  * no scratch (spill) access and no compiler-inserted vmcnt wait there.
Exit status 0 = fine.  python tools/check_bwdp_isa.py [path/to/wn_stack.hip]
(tests/test_abi.py runs it when hipcc is present.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
    ROOT, 'tensorflow-wavenet_amd', 'csrc', 'wn_stack.hip')


def regs_of(text):
    out = set()
    text = text.split(';')[0]
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r'\bv(\d+)\b', text))
    return out


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'k.s')
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'),
                               '--offload-arch=gfx950', '-O3', '-std=c++17', '-S',
                               '--cuda-device-only', '-o', asm, SRC])
        text = open(asm).read()
    m = re.search(r'^_Z17stack_bwdp_kernel8StackBwd:.*?^\.Lfunc_end\d+:', text, re.S | re.M)
    if not m:
        raise SystemExit('stack_bwdp_kernel not found in the assembly')
    lines = m.group(0).split('\n')
    # inline-asm regions
    in_asm, tagged = False, []
    for l in lines:
        if '#ASMSTART' in l:
            in_asm = True
        elif '#ASMEND' in l:
            in_asm = False
        tagged.append((l, in_asm))
    errors = []
    # ---- matrix loop: single-dword asm loads
    mat_dst = set()
    for l, ia in tagged:
        mm = re.match(r'\s*buffer_load_dword (v\d+),', l)
        if ia and mm:
            mat_dst |= regs_of(mm.group(1))
    idx = [i for i, (l, _) in enumerate(tagged) if 'v_mfma_f32_32x32x2' in l]
    if len(mat_dst) != 32 or not idx:
        errors.append('expected 32 A-operand registers / a matrix loop, found %d' % len(mat_dst))
    lo, hi = idx[0] - 8, idx[-1] + 8
    for i in range(max(lo, 0), min(hi, len(tagged))):
        l, ia = tagged[i]
        code = l.split(';')[0].strip()
        if not code or code.endswith(':'):
            continue
        if 'scratch_' in code:
            errors.append('matrix loop: spill access: ' + code)
        if 'vmcnt' in code and not ia:
            errors.append('matrix loop: compiler-inserted wait: ' + code)
        r = regs_of(code)
        if not (r & mat_dst):
            continue
        if ia and code.startswith('buffer_load_dword '):
            continue
        if ia and code.startswith('s_waitcnt'):
            continue
        if code.startswith('v_mfma_f32_32x32x2'):
            prev = [tagged[j][0] for j in range(i - 6, i) if 'vmcnt(31)' in tagged[j][0]]
            if not prev:
                errors.append('matrix loop: MFMA without its vmcnt(31): ' + code)
            continue
        errors.append('matrix loop: %s touches an asynchronously loaded register' % code)
    for e in errors:
        print('FAIL:', e)
    print('check_bwdp_isa: %s (%d A-operand registers, %d MFMAs in the ticket loop)' % (
        'FAILED' if errors else 'ok', len(mat_dst), len(idx)))
    return 1 if errors else 0


if __name__ == '__main__':
    sys.exit(main())
