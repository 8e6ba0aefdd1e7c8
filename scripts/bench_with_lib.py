#!/usr/bin/env python
"""Run bench.py on another build of the kernel library (diagnostics): RQ_LIB=<path> python scripts/bench_with_lib.py <bench.py arguments>."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
from rqvae import _native  # noqa: E402
if os.environ.get('RQ_LIB'):
    _native.LIB_PATH = os.environ['RQ_LIB']
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
