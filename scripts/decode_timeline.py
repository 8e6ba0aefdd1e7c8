#!/usr/bin/env python
"""One RQVAE.decode_code (or, RQ_WHAT=encode, RQVAE.get_codes) at the benchmark's decode sub-batch, for an ordered kernel timeline.

    python scripts/decode_timeline.py run            # the workload (under rocprofv3 --kernel-trace --output-format csv)
    python scripts/decode_timeline.py report <dir>   # ordered list of the LAST decode's dispatches: kernel, grid, duration, gap
"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == 'run':
    sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
    import torch
    from rqvae import presets
    dev = torch.device('cuda:0')
    vae, ar, cfg = presets.build('huge', device=dev, seed=0)
    del ar
    B = int(os.environ.get('RQ_B', 128))
    if os.environ.get('RQ_WHAT', 'decode') == 'encode':      # RQ_WHAT=encode: RQVAE.get_codes (encoder + residual quantiser) instead
        x = torch.randn((B, 3, 256, 256), device=dev).clamp(-1, 1)
        for _ in range(3):
            vae.get_codes(x)
            torch.cuda.synchronize()
    else:
        codes = torch.randint(0, 16384, (B, 8, 8, 4), device=dev)
        for _ in range(3):
            vae.decode_code(codes)
            torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    # the last decode = the last third of the dispatches after the first conv-like kernel pattern repeats; simply split by count
    n = len(rows) // 3
    last = rows[-n:]
    t_prev = None
    tot = 0
    print(f'{len(rows)} dispatches, {n} per decode')
    for r in last:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:58]
        gap = 0 if t_prev is None else (s - t_prev) / 1e3
        grid = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
        wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
        print(f'{name:58s} grid {grid:>9s} wg {wg:>4s}  {(e - s) / 1e3:9.1f} us  gap {gap:6.1f}')
        tot += e - s
        t_prev = e
    print(f'sum of kernel durations {tot / 1e3:.0f} us; span {(int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3:.0f} us')
