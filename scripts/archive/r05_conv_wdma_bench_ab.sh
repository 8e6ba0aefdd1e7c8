mkdir -p gpurun_out
for v in base new base new; do
  if [ $v = new ]; then unset RQ_LIB; else export RQ_LIB=rq-vae-transformer_amd/variants/librqamd_$v.so; fi
  python scripts/bench_with_lib.py --steps 2 --warmup 1 --sweep "" --also "" --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('decode_ms_per_image'), (d.get('rqvae_encode') or {}).get('value'), (d.get('roofline_decode') or {}).get('frac'))"
done 2>&1 | tee gpurun_out/r05_conv_wdma_bench_ab.txt
