#!/bin/bash
# round 6: split-K slab GEMMs at mid batch -- K slices per XCD group (sched 4) against n-ranges per XCD (sched 1, RQAMD_GEMM_NO_SCHED4=1);
# same box, alternating; images/s of the whole step at the reference's own batches
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
  for b in 500 200 100; do
    for v in sched4 sched1; do
      if [ $v = sched1 ]; then export RQAMD_GEMM_NO_SCHED4=1; else unset RQAMD_GEMM_NO_SCHED4; fi
      python bench.py --batch $b --steps 4 --warmup 1 --sweep "" --also "" --formats 0 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$v batch $b: %.1f images/s, %.2f ms/step, verified %s' % (d['value'], d['ms_per_step'], d.get('verified')))
"
    done
  done
done
