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
# librqamd_f16.so: the two engines once more with IEEE fp16 as the 16-bit storage type (csrc/rq_hip.h, -DRQ_F16=1) -- what
# RQTransformer.sample(amp=True) / forward(amp=True) run on, as the reference's fp16 autocast does (transformers.py:21,206), and what the
# opt-in RQAMD_VAE=fp16 RQ-VAE engine runs on.  Same C ABI (the rqamd_rqt_* / rqamd_vae_* entry points, rqamd_abi_version,
# rqamd_last_error); the quantiser (fp32 arithmetic) is not in it.
OUT_F16 = os.path.join(HERE, 'librqamd_f16.so')
SOURCES_F16 = ['api.hip', 'gemm.hip', 'rqt_kernels.hip', 'engine_rqt.hip', 'vae_kernels.hip', 'conv_halo.hip', 'engine_vae.hip']


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    """(serialised by a file lock: several test workers / processes may ask for the library at once)"""
    import fcntl
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    with open(os.path.join(HERE, 'build', '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'rqamd.h'))
    newest_h = max(os.path.getmtime(h) for h in headers)
    objs, objs16, procs = [], [], []
    for s, f16 in [(s, False) for s in SOURCES] + [(s, True) for s in SOURCES_F16]:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s + ('.f16.o' if f16 else '.o'))
        (objs16 if f16 else objs).append(obj)
        if force or _newer(src, obj) or newest_h > os.path.getmtime(obj):
            cmd = [hipcc] + FLAGS + (['-DRQ_F16=1'] if f16 else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, f16, subprocess.Popen(cmd)))
    for s, f16, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}' + (' (-DRQ_F16=1)' if f16 else ''))
    for out, obs, mine in ((OUT, objs, any(not f for _, f, _ in procs)), (OUT_F16, objs16, any(f for _, f, _ in procs))):
        if mine or not os.path.exists(out):
            cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--no-undefined', '-o', out] + obs
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
