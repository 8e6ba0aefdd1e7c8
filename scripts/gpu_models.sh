#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for cfg in "medium 4096" "large 4096" "xhuge 2048"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --steps 1 --warmup 1 --batch $2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 B=$2', {k:round(d[k],3) for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')})" || echo "$1 FAILED"
done
