#!/usr/bin/env python
"""One character per instruction, one line per barrier interval, of a kernel in hipcc's assembly output (no GPU needed):

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I csrc -S --cuda-device-only csrc/conv_halo.hip -o /tmp/conv_halo.s
    python scripts/isa_summary.py /tmp/conv_halo.s 'conv3x3_halo_kernel<1, 0, 1>'

M = v_mfma, r = ds_read, w = ds_write, G = global / buffer load (incl. LDS-DMA), S = global store, t = v_exp / v_rcp (transcendental),
v = other VALU, s = SALU, _ = s_waitcnt, | = s_barrier.  This is how the un-overlapped lumps of GroupNorm arithmetic and the fragment reads
that came out behind their MFMAs were found (profiles/r03_conv_halo_isa_interleave.txt); tests/test_isa_schedule.py asserts the
properties on every build."""
import re
import subprocess
import sys


def classify(op):
    if op.startswith('v_mfma'):
        return 'M'
    if op.startswith('ds_read'):
        return 'r'
    if op.startswith('ds_write'):
        return 'w'
    if op.startswith('global_load') or op.startswith('buffer_load'):
        return 'G'
    if op.startswith('global_store') or op.startswith('buffer_store'):
        return 'S'
    if op.startswith('v_exp') or op.startswith('v_rcp'):
        return 't'
    if op.startswith('v_'):
        return 'v'
    if op.startswith('s_barrier'):
        return '|'
    if op.startswith('s_waitcnt'):
        return '_'
    if op.startswith('s_'):
        return 's'
    return ''


def kernels(path):
    """{demangled name: [instruction text, ...]} of every kernel in an assembly file."""
    out, cur, mangled = {}, None, []
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        t = line.split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'):
            continue
        out[cur].append(t)
        if t.startswith('s_endpgm'):
            cur = None
    names = subprocess.run(['c++filt'] + list(out), capture_output=True, text=True).stdout.split('\n')
    return {n.split('(')[0].replace('void ', ''): ins for n, ins in zip(names, out.values())}


def intervals(instrs):
    """the instruction classes of a kernel, split at its barriers"""
    s = ''.join(classify(i.split()[0]) for i in instrs)
    return s.split('|')


if __name__ == '__main__':
    ks = kernels(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, ins in ks.items():
        if want and want not in name:
            continue
        print(f'== {name}: {len(ins)} instructions')
        print('|\n'.join(intervals(ins)))
