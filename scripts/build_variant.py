#!/usr/bin/env python
"""Build an A/B variant of librqamd.so (diagnostics only; the product library is built by rq-vae-transformer_amd/build.py).

  python scripts/build_variant.py --out rq-vae-transformer_amd/variants/librqamd_base.so --rev HEAD
  python scripts/build_variant.py --out rq-vae-transformer_amd/variants/librqamd_noslp.so --file-flags conv_halo.hip=-fno-slp-vectorize

--rev REV builds the csrc/ of that git revision (checked out into a scratch directory), otherwise the working tree's.
The scripts under scripts/ load a variant when RQ_LIB=<path> is set; variants/ is git-ignored but travels with gpurun."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rq-vae-transformer_amd')
sys.path.insert(0, PKG)
import build as rqbuild  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--rev', default=None)
    ap.add_argument('--csrc', default=None, help='a patched copy of csrc/ (scratch experiments that should not touch the tree)')
    ap.add_argument('--flags', default='')
    ap.add_argument('--file-flags', action='append', default=[], help='file.hip=-flag[,-flag]')
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix='rqvar_')
    try:
        if a.rev:
            subprocess.check_call(f'git -C {ROOT} archive {a.rev} rq-vae-transformer_amd/csrc include | tar -x -C {tmp}', shell=True)
            csrc, inc = os.path.join(tmp, 'rq-vae-transformer_amd', 'csrc'), os.path.join(tmp, 'include')
        else:
            csrc, inc = (a.csrc or rqbuild.CSRC), os.path.join(ROOT, 'include')
        per_file = dict((kv.split('=', 1)[0], kv.split('=', 1)[1].split(',')) for kv in a.file_flags)
        objs, procs = [], []
        for s in rqbuild.SOURCES:
            obj = os.path.join(tmp, s + '.o')
            objs.append(obj)
            cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-I', csrc, '-I', inc]
            cmd += a.flags.split() + per_file.get(s, []) + ['-c', os.path.join(csrc, s), '-o', obj]
            procs.append((s, subprocess.Popen(cmd)))
        for s, p in procs:
            if p.wait() != 0:
                raise SystemExit(f'hipcc failed on {s}')
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', a.out] + objs)
        print(a.out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
