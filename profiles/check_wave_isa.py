#!/usr/bin/env python3
"""Static check of the wave kernels' ISA (kernels_wave.hip).

The kernel loads its input fragments with inline-asm `global_load_dwordx4 v[hi..], ... ; XLOAD` into a register range at
the top of its VGPR budget that the compiler knows nothing about, waits for them one row later with a counted
`s_waitcnt vmcnt(N) ; XWAIT`, and only then copies them into ordinary registers (`v_mov_b32 ... ; XREAD`).  That is sound
only if no compiler-generated instruction ever touches the reserved range.  This script compiles the file to ISA and
checks, for every wave kernel:

  1. no scratch access inside the row loop and at most 32 bytes of it in total -- a spilled accumulator costs more than the
     kernel's whole margin (a value parked before the loop and reloaded behind it does not);
  2. outside the XLOAD / XREAD asm blocks no instruction names a VGPR inside [lowest XLOAD destination, budget top);
  3. no XREAD between an XLOAD and the next XWAIT in layout order (the row loop is laid out wait -> read -> ... -> load);
  4. the kernel allocates exactly its budget (the reserved range exists): vgpr_count == 256 / 168 / 128.

Usage: check_wave_isa.py [-v] [--tune] [file.s]   (without a file: compiles cosypose_amd/csrc/kernels_wave.hip with the
shipping flags, or with -DCOSY_TUNE: the experiment build must be checked too before its timings are believed)
Exit code 0 = clean.  tests/test_build_isa.py runs it.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_isa(out, tune=False):
    src = os.path.join(REPO, 'cosypose_amd', 'csrc', 'kernels_wave.hip')
    sys.path.insert(0, REPO)
    from cosypose_amd.build import HIPCC, FLAGS, FILE_FLAGS        # the shipping flags, from the one place that defines them
    cmd = [HIPCC] + [f for f in FLAGS if not f.startswith('-W')] + FILE_FLAGS.get('kernels_wave.hip', []) + ['-S', '--cuda-device-only', '-o', out, src]
    if tune:
        cmd.insert(1, '-DCOSY_TUNE')
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


VREG = re.compile(r'\bv\[(\w+)(?::(\w+))?\]|\bv(\d+)\b')


def vregs(code):
    out = []
    for m in VREG.finditer(code):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            lo = int(m.group(1), 0)
            out += list(range(lo, (int(m.group(2), 0) if m.group(2) else lo) + 1))
    return out


def check_kernel(name, lines):
    """-> (problems, lowest reserved register)"""
    problems = []
    # pass 1: the reserved range
    lo = min([min(vregs(ln.split(';')[0].split(',')[0])) for ln in lines if 'XLOAD' in ln] or [10 ** 6])
    if lo == 10 ** 6:
        return [f'{name}: no XLOAD found'], None
    # spills are tolerated only OUTSIDE the row loop (layout region first XWAIT .. last XLOAD): a value parked in scratch before
    # the loop and reloaded behind it costs nothing, one inside the loop costs more than the kernel's whole margin
    waits = [i for i, ln in enumerate(lines) if 'XWAIT' in ln]
    loads = [i for i, ln in enumerate(lines) if 'XLOAD' in ln]
    if waits and loads:
        for i, ln in enumerate(lines):
            if min(waits) <= i <= max(loads) and ln.strip().startswith('scratch_'):
                problems.append(f'{name}: line {i}: scratch access inside the row loop: {ln.strip()!r}')
    pending, in_asm, ours = False, False, False
    for i, ln in enumerate(lines):
        st = ln.strip()
        if st.startswith(';;#ASMSTART'):
            in_asm, ours = True, False
            j = i + 1
            while not lines[j].strip().startswith(';;#ASMEND'):
                ours = ours or any(t in lines[j] for t in ('XLOAD', 'XREAD', 'XWAIT', 'XRESERVE', 'XFENCE'))
                j += 1
            continue
        if st.startswith(';;#ASMEND'):
            in_asm = False
            continue
        code = ln.split(';')[0].strip()
        if not code or code.endswith(':') or code.startswith('.'):
            continue
        if in_asm and ours:
            blk_end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(';;#ASMEND'))
            blk = ''.join(lines[i:blk_end])
            if 'XLOAD' in ln:
                pending = True
            elif 'XWAIT' in ln:
                pending = False
            elif code.startswith('v_mov_b32') and pending:
                problems.append(f'{name}: line {i}: fragment read while its load is in flight')
            continue
        hit = [r for r in vregs(code) if r >= lo]
        if hit:
            problems.append(f'{name}: line {i}: {code!r} touches reserved v{hit[0]} (reserved from v{lo})')
    return problems, lo


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('-')]
    if args:
        path = args[0]
    else:
        path = os.path.join(tempfile.mkdtemp(), 'wave.s')
        compile_to_isa(path, tune='--tune' in sys.argv)
    text = open(path).read().split('\n')
    problems, n, agprs = [], 0, {}
    # kernel bodies
    cur, body = None, []
    for ln in text:
        m = re.match(r'^(_ZN4cosy18mbconv_wave_kernel\w+):', ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if ln.strip().startswith('s_endpgm') and False:
                pass
            if ln.strip().startswith('.end_amdhsa_kernel') or ln.strip().startswith('.Lfunc_end'):
                pr, na = check_kernel(cur, body)
                problems += pr
                agprs[cur] = na
                n += 1
                cur = None
    # scratch
    meta = '\n'.join(text)
    rows = []
    for blk in meta.split('  - .agpr_count:')[1:]:
        nm = re.search(r'\.name:\s+(\S+)', blk)
        if not nm or 'mbconv_wave_kernel' not in nm.group(1):
            continue
        ag = int(blk.split('\n')[0])
        mw = int(re.search(r'Lb[01]ELi(\d)EEE', nm.group(1)).group(1))
        sc = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
        vg = int(re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1))
        rows.append((nm.group(1), vg, agprs.get(nm.group(1)), sc))
        if sc > 32:
            problems.append(f'{nm.group(1)}: {sc} bytes of scratch')
        if ag != 0 or vg != {2: 256, 3: 168, 4: 128}[mw]:
            problems.append(f'{nm.group(1)}: allocates {vg} VGPRs / {ag} AGPRs, expected the whole budget of {mw} waves per SIMD and no AGPRs')
    if '-v' in sys.argv:
        for r in rows:
            print('%-90s vgprs %3d  reserved from v%s  scratch %d' % r)
    print(f'checked {n} wave kernels')
    for p in problems:
        print('PROBLEM', p)
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
