#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for cfg in "2048 64" "2048 128" "3072 64" "4096 64"; do
  set -- $cfg
  RQAMD_VAE_CHUNK=$2 timeout 600 python bench.py --steps 2 --warmup 1 --batch $1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$1 chunk=$2', {k:round(d[k],3) for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')})"
done
