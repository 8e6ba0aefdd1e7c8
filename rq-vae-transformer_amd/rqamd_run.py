#!/usr/bin/env python
"""Run one of the reference's UNCHANGED drivers on top of this package.

    python rqamd_run.py /path/to/rq-vae-transformer/main_sampling_fid.py -a ... -v ... --top-k 1024 --top-p 0.95
    python rqamd_run.py -m measure_throughput f=32 d=4 c=16384 model=huge batch_size=100     (cwd = reference checkout)

Python puts a script's own directory (or the cwd, for -m) FIRST on sys.path, so launching the reference's scripts
directly would import the reference's `rqvae` package.  This launcher only arranges the import order -- this directory
(the MI355X-native `rqvae`) first, the reference checkout second -- and then hands over with runpy; nothing of the driver
is patched.  Sub-packages this repo does not provide (`rqvae.metrics`, `rqvae.img_datasets`, ...) still resolve to the
reference's files (see rqvae/__init__.py).  torch.distributed launchers work the same way:
    python -m torch.distributed.run --nproc-per-node 8 rqamd_run.py main_sampling_fid.py ...
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv):
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 2
    if argv[0] == '-m':
        if len(argv) < 2:
            print(__doc__)
            return 2
        ref_root = os.path.abspath(os.environ.get('RQVAE_REFERENCE_ROOT', os.getcwd()))
        sys.path[:] = [HERE, ref_root] + [p for p in sys.path if os.path.abspath(p or '.') not in (HERE, ref_root)]
        sys.argv = [argv[1]] + argv[2:]
        runpy.run_module(argv[1], run_name='__main__', alter_sys=True)
        return 0
    script = os.path.abspath(argv[0])
    ref_root = os.path.dirname(script)
    sys.path[:] = [HERE, ref_root] + [p for p in sys.path if os.path.abspath(p or '.') not in (HERE, ref_root)]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
