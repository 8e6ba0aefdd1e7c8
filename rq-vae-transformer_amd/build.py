#!/usr/bin/env python
"""Build librqamd.so (hipcc, gfx950 only) in-tree.  `python build.py [--force]`.

No torch involvement: the library is a plain C-ABI shared object (include/rqamd.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'librqamd.so')
SOURCES = ['api.hip', 'gemm.hip', 'quantize.hip', 'rqt_kernels.hip', 'engine_rqt.hip', 'vae_kernels.hip', 'conv_halo.hip', 'engine_vae.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-I', CSRC]


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'rqamd.h'))
    newest_h = max(os.path.getmtime(h) for h in headers)
    objs, procs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s + '.o')
        objs.append(obj)
        if force or _newer(src, obj) or newest_h > os.path.getmtime(obj):
            cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}')
    if procs or not os.path.exists(OUT):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
