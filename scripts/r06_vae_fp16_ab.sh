#!/bin/bash
# round 6: the opt-in fp16 RQ-VAE engine (RQAMD_VAE=fp16, librqamd_f16.so) against the bf16 default, same box, alternating:
# decode half of the default bench line at 10752 images, get_codes
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
  for fmt in bf16 fp16; do
    RQAMD_VAE=$fmt python bench.py --steps 2 --warmup 1 --sweep "" --also "" --formats 0 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); e = d.get('rqvae_encode') or {}
        print('RQAMD_VAE=$fmt: %.1f images/s, decode %.4f ms/image, get_codes %.0f images/s (%.3f M codes/s)' % (d['value'], d.get('decode_ms_per_image', float('nan')), e.get('images_per_sec', float('nan')), e.get('codes_per_sec', float('nan')) / 1e6))
"
  done
done
