#!/usr/bin/env python
"""TEST INFRASTRUCTURE: compile csrc/*.hip for the HOST against the fiber emulator (rq_emu.h) into
tests/emu/librqamd_emu.so.  Used by the CPU-only tests to check kernel index math where no GPU
exists; never loaded by the product (see tests/emu/README.md)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'rq-vae-transformer_amd', 'csrc')
OUT = os.path.join(HERE, 'librqamd_emu.so')
CXX = os.environ.get('RQ_EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')


def build(force=False):
    """(serialised by a file lock: the CPU suite may run its tests in several pytest-xdist workers, each of which comes here)"""
    import fcntl
    with open(os.path.join(HERE, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(force=False):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip')) + [os.path.join(HERE, 'rq_emu.cpp')]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(HERE, 'rq_emu.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in srcs:
        obj = os.path.join(objdir, os.path.basename(s) + '.o')
        objs.append(obj)
        cmd = [CXX, '-x', 'c++', '-std=c++17', '-O1', '-g', '-fPIC', '-DRQ_EMU', '-Wno-unused-result',
               '-I', HERE, '-I', CSRC, '-c', s, '-o', obj]
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'emulator build failed on {s}')
    subprocess.check_call([CXX, '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
