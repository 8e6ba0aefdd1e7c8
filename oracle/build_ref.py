#!/usr/bin/env python
"""Recipe for oracle/_ref: the REFERENCE's own modules for this path, byte-compiled from the sources where they lie under
/root/reference into sourceless .pyc files (TEST INFRASTRUCTURE; outputs only -- no reference source enters the repo).

    python oracle/build_ref.py            (also run by __graft_entry__.build() whenever /root/reference is present)

oracle/_ref/ is git-ignored (it stays out of history) but not gpurun-ignored, so it travels to the GPU box like the built
librqamd.so does; there `bench.py`'s cpu_baseline leg runs the reference itself on the host cores through
oracle/ref_cpu_baseline.py (cpu_baseline.kind = "reference").  Nothing else may import it; /root/reference itself is never
read on the GPU box.  The reference is Python, so "compiling" it means py_compile: the .pyc files hold bytecode, are tied to
this image's CPython (3.10) and are imported as sourceless modules (module.pyc beside its package's __init__.pyc)."""
import glob
import os
import py_compile
import shutil
import sys

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
# the modules the sampling path imports (rqvae.models -> rqvae.utils.utils, rqvae.optimizer.loss); nothing of the trainers
WANT = ['rqvae/__init__.py', 'rqvae/models/**/*.py', 'rqvae/utils/__init__.py', 'rqvae/utils/utils.py',
        'rqvae/optimizer/*.py',
        # the reference's throughput DRIVER, unchanged: tests/test_gpu_reference_driver.py runs its bytecode on the MI355X on top
        # of this repo's rqvae package (rqamd_run.py -m measure_throughput with RQVAE_REFERENCE_ROOT=oracle/_ref)
        'measure_throughput/__main__.py',
        # ... and its FID-sampling DRIVER with the import closure of its module level (compute_metrics -> rqvae.metrics ->
        # rqvae.txtimg_datasets): the same test file runs main_sampling_fid's sampling loop -- sample(), one decode_code call per
        # image, all_gather_cat, the sample pickles -- on synthetic checkpoints; the metric networks themselves are never built
        'main_sampling_fid.py', 'compute_metrics.py', 'rqvae/metrics/*.py', 'rqvae/txtimg_datasets/*.py',
        'rqvae/txtimg_datasets/tokenizers/*.py']
# data files the compiled modules open next to themselves: parsed here and re-emitted (JSON is YAML), not copied
DATA = ['measure_throughput/rq_defaults.yaml']


def build(verbose=True):
    if not os.path.isdir(REF):
        if verbose:
            print(f'oracle/build_ref.py: {REF} not present -- keeping whatever oracle/_ref holds')
        return None
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    n = 0
    for pat in WANT:
        for src in sorted(glob.glob(os.path.join(REF, pat), recursive=True)):
            rel = os.path.relpath(src, REF)
            dst = os.path.join(OUT, rel[:-3] + '.pyc')
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            py_compile.compile(src, cfile=dst, dfile=rel, doraise=True, optimize=0)
            n += 1
    import json
    import yaml
    for rel in DATA:
        with open(os.path.join(REF, rel)) as f:
            parsed = yaml.safe_load(f)
        with open(os.path.join(OUT, rel), 'w') as f:
            json.dump(parsed, f)
    with open(os.path.join(OUT, 'README'), 'w') as f:
        f.write('byte-compiled modules of kakaobrain/rq-vae-transformer (see oracle/build_ref.py); build output, not source\n')
    if verbose:
        print(f'oracle/_ref: {n} modules byte-compiled from {REF} (python {sys.version_info.major}.{sys.version_info.minor})')
    return OUT


if __name__ == '__main__':
    build()
