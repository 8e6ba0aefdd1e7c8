#!/bin/bash
# round 6: the KV workspace as uncached device memory (RQAMD_KV_UNCACHED=1: hipExtMallocWithFlags(hipDeviceMallocUncached)) against the
# default allocation read with non-temporal loads; same box, alternating; default bench line at 10752 images
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
  for v in default uncached; do
    if [ $v = uncached ]; then export RQAMD_KV_UNCACHED=1; else export RQAMD_KV_UNCACHED=0; fi
    timeout 600 python bench.py --steps 2 --warmup 1 --sweep "" --also "" --formats 0 --no-cpu-baseline 2>gpurun_out/kvu_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); ra = d.get('roofline_attn') or {}
        print('$v: %.1f images/s, AR %.4f ms/image, attention %.0f GB/s' % (d['value'], d.get('ar_ms_per_image', float('nan')), ra.get('achieved', float('nan'))))
"
    tail -2 gpurun_out/kvu_err.txt | grep -v amdgpu.ids
  done
done
