#!/usr/bin/env python
"""Do the sampler's kernels and the decoder's kernels overlap under `bench.py --overlap`?  Reads a rocprofv3 --kernel-trace CSV and
reports, for every decoder kernel (conv / GroupNorm / vae_*), how many sampler kernels (gemm_stream / resid_ln / attn / sample) START
inside its [start, end) interval, and the total time during which kernels of both groups are in flight.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --overlap --batch 64 --steps 3 --sweep "" --no-profile --no-cpu-baseline
    python scripts/overlap_timeline.py OUT"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
rows.sort()
dec = [r for r in rows if any(k in r[2] for k in ('conv', 'gn_', 'vae_', 'splitk', 'cvt_bf16', 'rq_embed'))]
smp = [r for r in rows if any(k in r[2] for k in ('gemm_stream', 'gemm_p8', 'resid_ln', 'attn_', 'sample_', 'embed_tokens'))]
print(f'{len(rows)} kernels: {len(dec)} decoder-side, {len(smp)} sampler-side')
starts = sorted(s for s, _, _ in smp)
import bisect
inside = 0
for s, e, n in dec:
    inside += bisect.bisect_left(starts, e) - bisect.bisect_left(starts, s)
tot_dec = sum(e - s for s, e, _ in dec)
# time with both groups in flight: sweep
ev = [(s, 1, 0) for s, e, _ in dec] + [(e, -1, 0) for s, e, _ in dec] + [(s, 1, 1) for s, e, _ in smp] + [(e, -1, 1) for s, e, _ in smp]
ev.sort()
cnt = [0, 0]
both = 0
last = ev[0][0]
for t, d, g in ev:
    if cnt[0] > 0 and cnt[1] > 0:
        both += t - last
    cnt[g] += d
    last = t
print(f'decoder kernels busy {tot_dec / 1e6:.1f} ms; sampler kernels starting inside a decoder kernel: {inside}; both groups in flight for {both / 1e6:.1f} ms')
big = sorted(dec, key=lambda r: r[0] - r[1])[:3]
for s, e, n in big:
    k = bisect.bisect_left(starts, e) - bisect.bisect_left(starts, s)
    print(f'   {n[:40]:40s} {(e - s) / 1e3:8.1f} us: {k} sampler kernels started inside')
