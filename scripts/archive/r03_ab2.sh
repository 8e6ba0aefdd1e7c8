#!/bin/bash
# round-3 A/B call 2: halo conv base vs new on one box, barrier timeline of the new kernel, short full-batch bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
V=rq-vae-transformer_amd/variants
for lib in base new; do
  echo "== conv_halo_bench: $lib"
  if [ $lib = new ]; then unset RQ_LIB; else export RQ_LIB=$PWD/$V/librqamd_$lib.so; fi
  timeout 300 python scripts/conv_halo_bench.py 2>&1 | tail -6
done
unset RQ_LIB
echo "== barrier timeline (new kernel)"
timeout 300 python scripts/conv_trace.py 2>&1 | tail -16 | cut -c1-600
echo "== bench (2 steps, full batch)"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --sweep "" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image','verified')}, d['roofline_decode']['frac'], d['roofline']['frac'])"
