#!/bin/bash
# round 6: non-temporal cache policy for streams that are read once -- variants of the tree (which already reads the KV cache non-temporally):
#   attn_st  = + the KV appends        smp_nt = + the logits rows of sample_topk_kernel
#   conv_nt1 = halo conv output stores  conv_nt2 = + halo-piece and residual loads
# same box, alternating with the tree; the default bench line at 10752 images
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
V=rq-vae-transformer_amd/variants
for rep in 1 2; do
  for lib in "" ${RQ_VARIANTS:-$V/librqamd_attn_st.so $V/librqamd_smp_nt.so $V/librqamd_conv_nt1.so $V/librqamd_conv_nt2.so}; do
    RQ_LIB=$lib python scripts/bench_with_lib.py --steps 2 --warmup 1 --sweep "" --also "" --formats 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); ra = d.get('roofline_attn') or {}
        print('%-60s %.1f images/s, AR %.4f, decode %.4f ms/image, attention %.0f GB/s' % ('${lib:-tree}', d['value'], d.get('ar_ms_per_image', float('nan')), d.get('decode_ms_per_image', float('nan')), ra.get('achieved', float('nan'))))
"
  done
done
