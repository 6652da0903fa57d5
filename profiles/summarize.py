#!/usr/bin/env python3
"""Summarise a profiles/collect.sh output directory: per-kernel time (rocprofv3 --kernel-trace) and
per-kernel mean PMC counters per launch.  HBM traffic follows MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
(KiB) reports half of a wide coalesced read stream -> read bytes ~= 2 * FETCH_SIZE * 1024 (upper bound for
narrow accesses); WRITE_SIZE (KiB) is taken as is (uncalibrated)."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    """Canonical kernel name from the MANGLED symbol (collect with rocprofv3 -M: its demangler garbles __bf16
    template arguments).  e.g. _ZN4cosy18pw_gemm_dma_kernelIDF16bLi4ELi2ELi3ELb1EEEvNS_7PwKArgsE ->
    pw_gemm_dma_kernel<__bf16, 4, 2, 3, true>  (same strings as cosy_effnet_b3_profile_read reports)."""
    m = re.match(r'_ZN4cosy\d+([a-z0-9_]+?)I(.*)EEv', name) or re.match(r'_ZN4cosy\d+([a-z0-9_]+?)()E', name)
    if not m:
        return re.sub(r'\(.*$', '', re.sub(r'^void ', '', name)).replace('cosy::', '')
    base, targs = m.group(1), m.group(2)
    if not targs:
        return base
    toks, i = [], 0
    while i < len(targs):
        if targs.startswith('DF16b', i): toks.append('__bf16'); i += 5
        elif targs.startswith('DF16_', i): toks.append('_Float16'); i += 5
        elif targs[i] == 'f': toks.append('float'); i += 1
        elif targs.startswith('Li', i):
            j = targs.index('E', i); toks.append(targs[i + 2:j]); i = j + 1
        elif targs.startswith('Lb', i):
            toks.append('true' if targs[i + 2] == '1' else 'false'); i += 4
        else:
            toks.append(targs[i:]); break
    return f"{base}<{', '.join(toks)}>"


def main(root):
    times = defaultdict(list)
    for f in glob.glob(f'{root}/stats/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            times[(short(r['Kernel_Name']), str(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    ctr = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f'{root}/pmc_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[(short(r['Kernel_Name']), r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
    total = sum(sum(v) for v in times.values())
    bykern = defaultdict(lambda: [0.0, 0])
    for (k, g), v in times.items():
        bykern[k][0] += sum(v); bykern[k][1] += len(v)
    # per-kernel mean HBM traffic per launch from the PMC passes (read ~ 2*FETCH_SIZE KiB on gfx950, see module docstring)
    import json
    agg = defaultdict(lambda: defaultdict(list))
    for (k, g), c in ctr.items():
        for n in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg[k][n] += c.get(n, [])
    traffic = {k: dict(read_bytes=2 * 1024 * sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE']),
                       write_bytes=1024 * sum(v['WRITE_SIZE']) / len(v['WRITE_SIZE']), launches=len(v['FETCH_SIZE']))
               for k, v in agg.items() if v['FETCH_SIZE'] and v['WRITE_SIZE'] and not k.startswith('__amd')}
    import hashlib, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(here, 'cosypose_amd', 'csrc')
    for fn in sorted(os.listdir(d)):
        h.update(fn.encode()); h.update(open(os.path.join(d, fn), 'rb').read())
    # csrc_sha ties the numbers to the kernel sources they were measured on (bench.py reports `traffic` only on a match)
    json.dump(dict(csrc_sha=h.hexdigest()[:16], dtype=os.environ.get('COSY_PROFILE_DTYPE', 'fp16'), kernels=traffic), open(f'{root}/pmc_traffic.json', 'w'), indent=1)
    # matrix-core utilisation per kernel: rocprofv3's MfmaUtil (= sum SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * #SIMD), percent)
    # and the MFMA FLOP rate from SQ_INSTS_VALU_MFMA_MOPS_{F16,BF16} * 512 over the kernel's mean duration
    mf = defaultdict(lambda: defaultdict(list))
    for (k, g), c in ctr.items():
        for n in ('MfmaUtil', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            mf[k][n] += c.get(n, [])
        # MFMA op counters of the 16-bit types (the headline runs fp16, the bf16 line beside it bf16): whichever the kernel used
        a16, b16 = c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', []), c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', [])
        mf[k]['MOPS'] += [x + y for x, y in zip(a16, b16)] if a16 and b16 else (a16 or b16)
    rows = []
    for k, v in mf.items():
        if v['MfmaUtil'] and bykern[k][1] and sum(v['MOPS']) > 0:
            t_us = bykern[k][0] / bykern[k][1]
            mops = sum(v['MOPS']) / len(v['MOPS'])
            rows.append((sum(v['MfmaUtil']) / len(v['MfmaUtil']), mops * 512 / (t_us * 1e-6) / 1e12, t_us, k))
    if rows:
        print('== matrix-core utilisation per kernel (PMC pass `mfma`): MfmaUtil %, MFMA TFLOP/s issued, avg us')
        for u, tf, t_us, k in sorted(rows, reverse=True):
            print(f'{u:6.1f}%  {tf:8.1f} TFLOP/s  {t_us:8.1f} us  {k}')
        print()
    print(f'== per-kernel time (rocprofv3 --kernel-trace), total {total / 1e3:.2f} ms')
    for k, (t, n) in sorted(bykern.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f'{100 * t / total:5.1f}%  n={n:5d}  avg {t / n:9.1f} us  {k}')
    print('\n== per (kernel, grid): avg us | counters per launch (mean)')
    names = sorted({c for v in ctr.values() for c in v})
    for key, v in sorted(times.items(), key=lambda kv: -sum(kv[1]))[:60]:
        c = ctr.get(key, {})
        m = {n: sum(c[n]) / len(c[n]) for n in c}
        extra = []
        if 'FETCH_SIZE' in m:
            extra.append(f"rd~{2 * m['FETCH_SIZE'] * 1024 / 1e6:.1f}MB")
        if 'WRITE_SIZE' in m:
            extra.append(f"wr~{m['WRITE_SIZE'] * 1024 / 1e6:.1f}MB")
        if 'TCC_HIT_sum' in m and m['TCC_HIT_sum'] + m.get('TCC_MISS_sum', 0) > 0:
            extra.append(f"L2hit={m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.2f}")
        if 'SQ_WAVE_CYCLES' in m and m['SQ_WAVE_CYCLES'] > 0:
            wc = m['SQ_WAVE_CYCLES']
            extra.append(f"wait={m.get('SQ_WAIT_ANY', 0) / wc:.2f} waitinst={m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} "
                         f"valu={m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.2f} vmem={m.get('SQ_ACTIVE_INST_VMEM', 0) / wc:.2f}")
        if 'SQ_INSTS_VALU' in m and 'GRBM_GUI_ACTIVE' in m and m['GRBM_GUI_ACTIVE'] > 0:
            # VALU wave-instructions per SIMD-cycle of the launch (1024 SIMDs): x ~3.1-8.7 issue cycles each = pipe occupancy
            extra.append(f"valu/simd/cyc={m['SQ_INSTS_VALU'] / (m['GRBM_GUI_ACTIVE'] * 1024):.3f} ldsconf={m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1):.2f}")
        if 'SQ_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m and m['GRBM_GUI_ACTIVE'] > 0:
            extra.append(f"gui={m['GRBM_GUI_ACTIVE']:.0f}")
        if 'TA_BUSY_avr' in m:
            extra.append(f"TAbusy={m['TA_BUSY_avr']:.0f}")
        if 'TCP_TOTAL_CACHE_ACCESSES_sum' in m:
            extra.append(f"L1acc={m['TCP_TOTAL_CACHE_ACCESSES_sum'] / 1e6:.2f}M L1->L2rd={m.get('TCP_TCC_READ_REQ_sum', 0) / 1e6:.2f}M "
                         f"wr={m.get('TCP_TCC_WRITE_REQ_sum', 0) / 1e6:.2f}M pend={m.get('TCP_PENDING_STALL_CYCLES_sum', 0) / 1e6:.1f}M")
        print(f'{sum(v) / len(v):9.1f} us n={len(v):3d} {key[0]} grid={key[1]} | ' + ' '.join(extra))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_run')
