#!/bin/bash
# round-3 A/B call 4: 256x256 GEMM epilogue rework, base vs new on one box; short full-batch bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
V=rq-vae-transformer_amd/variants
for lib in base new; do
  if [ $lib = new ]; then unset RQ_LIB; else export RQ_LIB=$PWD/$V/librqamd_$lib.so; fi
  echo "== gemm_p8_check: $lib"
  RQ_AUTO=1 RQ_MS=10752 timeout 300 python scripts/gemm_p8_check.py 2>&1 | grep -v "^screen" | tail -7
done
unset RQ_LIB
echo "== bench (2 steps, full batch)"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sweep "" 2>&1 | tail -1 > gpurun_out/r03w_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03w_bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image','verified')})
for k in ('roofline','roofline_decode','roofline_attn'):
    r=d.get(k) or {}
    print(k, r.get('frac'), r.get('avg_launch_us'), (r.get('in_graph') or {}).get('frac'))
PY
