#!/bin/bash
# round 6: decode-step attention with non-temporal K / V cache loads (variants/librqamd_attn_nt.so, -DRQ_ATTN_NT=1) against the tree, same box,
# alternating; the default bench line's AR time, attention roofline and images/s at 10752 images
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
  for lib in "" rq-vae-transformer_amd/variants/librqamd_attn_nt.so; do
    RQ_LIB=$lib python scripts/bench_with_lib.py --steps 2 --warmup 1 --sweep "" --also "" --formats 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); ra = d.get('roofline_attn') or {}
        print('${lib:-tree}: %.1f images/s, AR %.4f ms/image, attention %.0f GB/s (frac %.3f)' % (d['value'], d.get('ar_ms_per_image', float('nan')), ra.get('achieved', float('nan')), ra.get('frac', float('nan'))))
"
  done
done
