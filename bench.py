#!/usr/bin/env python
"""bench.py -- 256x256 images/sec of the AR-sampling + decode path (BASELINE.json metric) on MI355X.

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted on --
ImageNet-256 class-conditional RQ-Transformer 1.4B (E 1536 / 24 heads / 42 body + 6 head layers /
V 16384, measure_throughput 'huge', reference measure_throughput/__main__.py:71-92) sampling 8x8x4
codes, then RQ-VAE (104 M) decode_code to 256x256 pixels and the [0,1] clamp.  Random-init weights of that
architecture (torch.manual_seed(0), module default inits), zero class condition, synthetic -- there is no network.

One "step" = one batch of B images per GPU: sample -> decode -> clamp (-> pixel all-gather when N > 1,
main_sampling_fid.py:226).  value = N * B * K / max-over-ranks time, inputs resident in HBM.  The headline step calls
``decode_code(codes)`` ONCE on the whole batch; the reference's throughput script decodes one image per call
(``torch.cat([decode_code(chunk) for chunk in codes.chunk(B)])``, :297-299) -- that exact loop is timed separately, at
the reference's own batches, as ``batch_sweep[*].driver_loop`` (it is served by the read-ahead of RQVAE.decode_code).

Launching.  `python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` re-executing this
file, one process per GPU, RCCL); under an external torchrun (RANK / WORLD_SIZE in the environment) it joins
that job.  The world size actually joined is what `n_gpus` reports.

Extra objects on the JSON line: "roofline" for the dominant kernel (the bf16 MFMA weight-streaming GEMM of the
decode step, timed live with HIP events on the engine's stream in a separate profiled pass), "roofline_attn" (KV-cache
bytes / decode-attention time, same pass), "roofline_decode" (RQ-VAE decoder conv FLOPs / decode time of the timed region),
"step_frac_of_mfma_peak" (all algorithmic FLOPs of a step / step time / 2.5 PF), "verified" (post-timed-region check of
the sampled codes), "batch_sweep" (the same measurement at the per-GPU batches SURVEY 8d names: 64 = BASELINE configs[3]
per-GPU share, 100 / 200 / 500 = the reference's Fig. 4; each with its own roofline and the reference script's
one-image-per-call loop as "driver_loop"), "per_image_decode" (that loop on its own), "per_image_recon", "roofline_rq"
(the residual quantiser), "rqvae_encode" (codes/sec) and "cpu_baseline" (the REFERENCE's own modules -- oracle/_ref -- on the
host cores, a bounded sample of a 32-image batch scaled to the metric's unit, kind "reference"; the numpy oracle port only when
oracle/_ref is absent; rank 0, N=1 only) and "baseline_8gpu_models_per_gpu_point" (round 4: the two models BASELINE.json quotes on
8 GPUs -- 3.8B at a global batch of 512, the 3.9B text-to-image shape -- at their per-GPU share of 64 images, on this one GPU)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'rq-vae-transformer_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3    # fp32-input MFMA = fp32 vector rate
A100_FIG4_IMG_S = 52.6          # reference Fig. 4, 1.4B 8x8x4, batch 500, 1x A100 (fp32)
# Per-GPU batch: as large as 288 GB allows (KV cache = 16.5 MB / image for the 1.4B model), and chosen so that the 256-row
# tiles of the decode-step GEMMs fill whole rounds of the 256 CUs: M = 10752 = 42 m-tiles gives 252 / 756 / 1008 workgroups
# for the N = 1536 / 4608 / 6144 GEMMs of the 1.4B model (98 % of 1 / 3 / 4 rounds); 16384 = 64 m-tiles for E = 1024,
# 6400 = 25 m-tiles for E = 2560.
DEFAULT_BATCH = {'huge': 10752, 'large': 10752, 'small': 10752, 'medium': 16384, 'xhuge': 6400, 'txt3900m': 2048, 'cc3m': 4096, 'tiny': 64}
# N > 1 ranks: the two models BASELINE.json quotes on 8 GPUs are quoted at a GLOBAL batch (configs[3]: 512 = 64 per GPU; configs[4]
# batch-sharded the same way), so their multi-GPU default is that per-GPU share; the 1.4B headline model keeps its N = 1 batch on
# every rank (weak scaling: per-GPU work fixed, so the driver's per-N values are comparable with the N = 1 line).
DEFAULT_BATCH_MULTI = {'xhuge': 64, 'txt3900m': 64}
WORKLOADS = {'huge': 'BASELINE configs[2]', 'medium': 'BASELINE configs[1]', 'xhuge': 'BASELINE configs[3] dims',
             'txt3900m': 'BASELINE configs[4] dims', 'cc3m': 'CC-3M 654M'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('RQ_BENCH_BATCH', 0)),
                    help='images per GPU per step (0 = the per-model default, DEFAULT_BATCH)')
    ap.add_argument('--model', default='huge')
    # BASELINE.json configs[2]: top-k=1024 / top-p=0.95 (0 / 1.0 = the reference defaults top_k=None, top_p=None)
    ap.add_argument('--top-k', type=int, default=1024)
    ap.add_argument('--top-p', type=float, default=0.95)
    ap.add_argument('--sweep', type=str, default='64,100,200,500', help='extra per-GPU batches measured after the timed region ("" = none)')
    ap.add_argument('--formats', type=int, default=1,
                    help='1 (default): after everything else, time the 1.4B model at 2048 images with the default bf16 engine, the opt-in 8-bit key / '
                         'key + value caches (RQAMD_KV=int8k / int8kv) and the fp16 build of the engine (sample(amp=True)) -> "kv_cache_formats"; 0: skip (~25 s)')
    ap.add_argument('--also', type=str, default='xhuge:64,txt3900m:64',
                    help='model:batch points measured after everything else on rank 0 at N = 1 (default: the two models BASELINE.json quotes on '
                         '8 GPUs, at their per-GPU share of 64 images; "" = none).  Adds ~40 s to a default run: after the headline model is '
                         'released, two ~3.8B-parameter models are built (random init), captured and timed; a failure is reported in the JSON '
                         'line AND on stderr')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--overlap', action='store_true', help='experiment: decode batch i while sampling batch i+1 (two streams)')
    ap.add_argument('--overlap-prio', type=int, default=0, help='with --overlap: 1 = sampling stream at high priority, 2 = decode stream at high priority')
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU/gloo rehearsal of the launcher, sharding, gather and timing logic with synthetic pixels (no kernels run; tests only)')
    args = ap.parse_args(argv)
    if args.top_k is not None and args.top_k <= 0:
        args.top_k = None
    if args.top_p is not None and args.top_p >= 1.0:
        args.top_p = None
    return args


def self_launch(args, argv):
    """No torchrun environment and --gpus N > 1: start the N ranks (one process per GPU) and relay rank 0's JSON line."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


GATHER_BUDGET_BYTES = 16 << 30


def gather_plan(world, batch, per_img_bytes):
    """(rows per all-gather call, number of calls) of gather_pixels for a local batch -- also what config.parallelism reports."""
    if world * batch * per_img_bytes <= GATHER_BUDGET_BYTES:
        return batch, 1
    step = max(1, GATHER_BUDGET_BYTES // (world * per_img_bytes))
    return step, (batch + step - 1) // step


def gather_pixels(distenv, pixels):
    """main_sampling_fid.py:226: every rank receives all ranks' pixels, rank-major (rqvae.utils.dist.all_gather_cat = one
    RCCL all_gather_into_tensor).  At the bench batch the gathered fp32 tensor is world x 8.4 GB -- next to a 180 GB KV cache
    it does not fit 288 GB for world = 8 -- so above GATHER_BUDGET_BYTES the same collective runs over slices of the local batch
    (same bytes over xGMI, same rank-major order per slice) and each gathered slice is released before the next."""
    from rqvae.utils.dist import all_gather_cat
    per_img = pixels[0].numel() * pixels.element_size()
    step, calls = gather_plan(distenv.world_size, pixels.shape[0], per_img)
    if calls == 1:
        return all_gather_cat(distenv, pixels)
    last = None
    for s0 in range(0, pixels.shape[0], step):
        last = all_gather_cat(distenv, pixels[s0:s0 + step])
    return last


def parallelism_note(world, batch, per_img_bytes=3 * 256 * 256 * 4):
    """config.parallelism, truthfully: how many all-gather calls a step makes and what survives them."""
    if world <= 1:
        return 'single GPU'
    step, calls = gather_plan(world, batch, per_img_bytes)
    if calls == 1:
        return f'replica x{world}, image batches sharded, one pixel all-gather (RCCL all_gather_into_tensor) per step'
    return (f'replica x{world}, image batches sharded; the gathered fp32 pixels ({world} x {batch} images) exceed the {GATHER_BUDGET_BYTES >> 30} GiB '
            f'gather budget, so every step runs {calls} all-gathers over slices of {step} local images (same bytes over xGMI, rank-major per '
            f'slice) and keeps only the last gathered slice')


def one_step(vae, ar, empty_sample, empty_cond, distenv, top_k, top_p):
    codes = ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=top_k, top_p=top_p)
    pixels = vae.decode_code(codes)
    pixels.mul_(0.5).add_(0.5).clamp_(0, 1)             # in place: 8.4 GB per temporary at B = 10752
    if distenv is not None and distenv.world_size > 1:
        pixels = gather_pixels(distenv, pixels)
    return codes, pixels


def cpu_baseline_reference(model, top_k, top_p, batch=32, positions=16, n_dec=8, threads=16):
    """kind "reference": the REFERENCE's own modules (oracle/_ref: byte-compiled from /root/reference by oracle/build_ref.py,
    which the build step runs; they travel with the snapshot) on this box's host cores, in a process of their own (their package
    is also called `rqvae`).  Bounded sample of the benchmarked workload (~10-30 s of CPU work): RQTransformer.sample -- the
    reference's Python loop on torch CPU kernels, fp32, KV cache on -- over the LAST `positions` of the 64 spatial positions of a
    batch of `batch` images (start_loc: the reference prefills the prefix in its first cached step), scaled to 64 positions,
    plus the throughput script's one-image-per-call decode_code + clamp (measure_throughput/__main__.py:297-300) of `n_dec`
    images.  `threads` torch threads: with all 128 hardware threads of the box the same code is ~10x slower (a full 16-image
    batch took 488 s + 47.5 s there: every small op pays a 128-way fork/join).  None when oracle/_ref is absent."""
    ref_dir = os.path.join(ROOT, 'oracle', '_ref')
    if not os.path.isdir(os.path.join(ref_dir, 'rqvae')):
        return None
    from rqvae import presets
    arch, vname = presets.RQTRANSFORMER[model]
    payload = {'rqt': arch, 'vae': {k: presets.RQVAE[vname][k] for k in ('hparams', 'ddconfig')}}
    threads = max(1, min(threads, os.cpu_count() or threads))
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py'), '--arch', json.dumps(payload), '--batch', str(batch),
           '--top-k', str(top_k or 0), '--top-p', str(top_p if top_p is not None else 1.0), '--positions', str(positions),
           '--decode', str(n_dec), '--threads', str(threads)]
    env = dict(os.environ)
    env['HIP_VISIBLE_DEVICES'] = ''                     # the reference leg is a CPU run
    env['OMP_NUM_THREADS'] = str(threads)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f'oracle/ref_cpu_baseline.py failed: {r.stderr[-400:]}')
    d = json.loads(lines[-1])
    return {'value': d['images_per_sec'], 'unit': 'images/sec', 'cores': int(d['threads']), 'kind': 'reference',
            'extrapolated': True, 'extrapolation_factor': d['of_positions'] / d['positions'],
            'sample': f"the reference's own modules (oracle/_ref), fp32 on torch CPU kernels, {d['threads']} threads: RQTransformer.sample over the "
                      f"last {d['positions']} of {d['of_positions']} spatial positions of a batch of {d['batch']} (prefix prefilled by its first "
                      f"cached step; top-k {top_k} / top-p {top_p}) = {d['ar_s']:.1f} s, scaled x{d['of_positions'] / d['positions']:.0f} -> "
                      f"{d['ar_s_per_image']:.2f} s/img (an UPPER bound on s/img, i.e. biased against the CPU: the late positions carry the "
                      f"longest prefix re-embedding, transformers.py:218-225); per-image RQVAE.decode_code + clamp of {d['decoded']} images = "
                      f"{d['decode_s_per_image']:.2f} s/img"}


def cpu_baseline(vae, ar, cfg, vcfg, n_pos=3, batch=32, n_dec=2):
    """Oracle (restatement of the reference, kind 'port') on the host cores, its heavy primitives on PyTorch's CPU kernels --
    the kernels the reference itself runs on a CPU (oracle/backend.py): `n_pos` spatial positions (n_pos body steps +
    4*n_pos head/sampler steps) of a batch-`batch` sample, scaled to the 64 positions of an image, plus one full decode_code
    of `n_dec` images.  Bounded to ~10-30 s of CPU work."""
    import oracle
    aparams = {k: v.detach().float().cpu().numpy() for k, v in ar.state_dict().items()}
    vparams = {k: v.detach().float().cpu().numpy() for k, v in vae.state_dict().items()}
    orc = oracle.RQTransformerOracle(cfg, aparams)
    ov = oracle.RQVAEOracle(vcfg['hparams'], vcfg['ddconfig'], vparams)
    H, W, D = cfg['block_size']
    part = np.zeros((batch, H, W, D), np.int64)
    cl = max(cfg.get('block_size_cond', 1), 1)
    oracle.backend.use_torch(True)
    try:
        t0 = time.time()
        xs = orc.sample(part, ov.codebooks, cond=np.zeros((batch, cl), np.int64), max_steps=n_pos * D)
        t_ar = (time.time() - t0) * (H * W / n_pos) / batch            # seconds per image
        t0 = time.time()
        ov.decode_code(xs[:n_dec])
        t_dec = (time.time() - t0) / n_dec
    finally:
        oracle.backend.use_torch(False)
    return {'value': 1.0 / (t_ar + t_dec), 'unit': 'images/sec', 'cores': int(torch.get_num_threads()), 'kind': 'port', 'extrapolated': True,
            'sample': f'oracle (numpy restatement, heavy primitives on torch-CPU kernels), fp32: {n_pos} of {H * W} spatial positions at '
                      f'batch {batch} (scaled x{H * W / n_pos:.1f}) = {t_ar:.2f} s/img AR + full 256x256 decode_code of {n_dec} images = '
                      f'{t_dec:.2f} s/img'}


def gemm_roofline(ar, vae, empty_sample, empty_cond, top_k, top_p, device, model, B, cfg):
    """Profiled pass: every decode-step GEMM launch bracketed by HIP events on the engine's stream (graphs off for this pass)."""
    eng = ar._eng()
    eng.set_profile(True)
    ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=top_k, top_p=top_p)
    torch.cuda.synchronize(device)
    pf = eng.get_profile()
    eng.set_profile(False)
    if not (pf['gemm_launches'] > 0 and pf['gemm_ms_total'] > 0):
        return None
    sec = pf['gemm_ms_total'] * 1e-3
    gbs = pf['gemm_bytes'] / sec / 1e9
    tfl = pf['gemm_flops'] / sec / 1e12
    ridge = MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    intensity = pf['gemm_flops'] / pf['gemm_bytes']
    if intensity < ridge:
        roofline = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS}
    else:
        roofline = {'bound': 'mfma', 'achieved': tfl, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tfl / MFMA_BF16_PEAK_TFLOPS}
    traffic, traffic_src, traffic_note = None, None, None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_gemm_traffic_m{B}.json')))      # the newest round's file
    tp = cands[-1] if cands else os.path.join(ROOT, 'profiles', f'r04_gemm_traffic_m{B}.json')
    if model == 'huge' and os.path.exists(tp):
        # PMC counters cannot be collected from inside the timed run; this is the committed result of `scripts/gpu.sh pmc`
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) for the same GEMM shapes at the same batch rows,
        # launch-weighted like `achieved`.  It is stamped with a hash of the GEMM kernel sources it was measured on and is
        # refused when they have changed since.
        from rqvae import _native
        with open(tp) as f:
            tj = json.load(f)
        if tj.get('kernel_sources_sha16') == _native.kernel_source_hash():
            traffic, traffic_src = tj['hbm_bytes_per_launch_weighted'], os.path.relpath(tp, ROOT)
        else:
            traffic_note = f'{os.path.relpath(tp, ROOT)} was measured on other GEMM sources ({tj.get("kernel_sources_sha16")}): refused'
    roofline.update({'traffic': traffic, 'traffic_source': traffic_src, 'traffic_measured_in_run': False, 'traffic_note': traffic_note,
                     'algorithmic_bytes_per_launch': pf['gemm_bytes'] / pf['gemm_launches'], 'kernel': 'decode-step bf16 MFMA GEMMs (gemm_* kernels)',
                     'launches_per_batch': pf['gemm_launches'], 'avg_launch_us': pf['gemm_ms_total'] * 1e3 / pf['gemm_launches'],
                     'algorithmic_GB_per_batch': pf['gemm_bytes'] / 1e9, 'algorithmic_TFLOP_per_batch': pf['gemm_flops'] / 1e12,
                     'achieved_GBps': gbs, 'achieved_TFLOPs': tfl, 'flop_per_byte': intensity,
                     'note': ('proj / fc2 launches also carry the residual update of the fp32 stream (read + write of rows x E x 4 B in their '
                              'epilogue, counted in algorithmic_bytes) when K is not split; RQAMD_NO_FUSE_RESID=1 restores the plain slab '
                              'epilogue (GEMM frac 0.42 instead of 0.40 at 10752 rows, 2.5 % fewer images/s)')
                             if not os.environ.get('RQAMD_NO_FUSE_RESID') else 'plain slab epilogues (RQAMD_NO_FUSE_RESID)'})
    # the same GEMMs inside the captured graphs: a sampling pass with and without its GEMM launches (engine profile mode 2), HIP events
    # around each pass.  At small batches the per-launch events above run eagerly and include the dispatch latency of their own markers
    # (B = 64: 10.4 us per launch against 7.7 us in the rocprof kernel trace of the graph run); this difference agrees with the trace.
    try:
        def timed_pass():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=top_k, top_p=top_p)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        timed_pass()
        t_full = min(timed_pass(), timed_pass())
        eng.set_profile(2)
        timed_pass()
        t_skip = min(timed_pass(), timed_pass())
        eng.set_profile(0)
        g_ms = t_full - t_skip
        if g_ms > 0:
            ach = (pf['gemm_bytes'] / (g_ms * 1e-3) / 1e9) if roofline['bound'] == 'hbm' else (pf['gemm_flops'] / (g_ms * 1e-3) / 1e12)
            roofline['in_graph'] = {'gemm_ms_per_batch': g_ms, 'avg_launch_us': g_ms * 1e3 / pf['gemm_launches'], 'achieved': ach,
                                    'frac': ach / roofline['peak'], 'sampling_ms_with_gemms': t_full, 'sampling_ms_without_gemms': t_skip,
                                    'what': 'time of a graph-mode sampling pass minus the same pass with the GEMM launches skipped'}
    except Exception as e:
        eng.set_profile(0)
        roofline['in_graph'] = {'error': repr(e)}
    # second roofline of the same pass: the decode-step attention reads the KV cache once per step (HBM-bound)
    attn = None
    if pf.get('attn_launches', 0) > 0 and pf['attn_ms_total'] > 0:
        kvb = kv_bytes_per_image(cfg) * B
        gb = kvb / (pf['attn_ms_total'] * 1e-3) / 1e9
        attn = {'bound': 'hbm', 'achieved': gb, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gb / HBM_PEAK_GBS,
                'kernel': 'attn_decode_kernel / attn_small_kernel (KV-cached MultiSelfAttention, attentions.py:60-104)',
                'launches_per_batch': pf['attn_launches'], 'avg_launch_us': pf['attn_ms_total'] * 1e3 / pf['attn_launches'],
                'algorithmic_GB_per_batch': kvb / 1e9, 'ms_per_batch': pf['attn_ms_total'],
                'what': 'bf16 K and V of every cached key read once per step and layer; q / output traffic not counted'}
    return roofline, attn


def kv_bytes_per_image(cfg):
    """bf16 K + V bytes the cached attention reads for one image: body step p attends p + cond_len keys in each body layer,
    head step d attends d + 1 keys in each head layer (0.56 GB per image for the 1.4B model)."""
    E, (H, W, D) = cfg['embed_dim'], cfg['block_size']
    cl = max(cfg.get('block_size_cond', 1), 1)
    body = sum(p + cl for p in range(H * W)) * cfg['body']['n_layer']
    head = H * W * sum(d + 1 for d in range(D)) * cfg['head']['n_layer']
    if os.environ.get('RQAMD_KV', 'bf16') == 'int8k':      # opt-in: body keys as 64 bytes + an fp32 scale per head (68 B per 64 components), values bf16
        return body * (E * 2 + E + (E // 64) * 4) + head * E * 2 * 2
    if os.environ.get('RQAMD_KV', 'bf16') == 'int8kv':     # opt-in: body keys AND values as bytes + scales
        return body * 2 * (E + (E // 64) * 4) + head * E * 2 * 2
    return (body + head) * E * 2 * 2


def rqt_flops_per_image(cfg):
    """2 * (weights touched per step) summed over the 64 body and 256 head steps of one image (GEMMs only)."""
    E, V, (H, W, D) = cfg['embed_dim'], cfg['vocab_size'], cfg['block_size']
    return 2.0 * (H * W * cfg['body']['n_layer'] * 12 * E * E + H * W * D * (cfg['head']['n_layer'] * 12 * E * E + E * V))


def decoder_flops_per_image(dd, embed_dim, executed=False):
    """Convolution + attention-GEMM FLOPs of Decoder.forward + post_quant_conv for one image (modules.py:171-202), from the
    ddconfig alone: 249.5 GFLOP for the released 256x256 shapes.  executed=True: what the engine's kernels multiply --
    the upsample convs whose source image has whole 8 x 32 tiles (output >= 64^2) run as four 2 x 2 convs over the source image,
    4/9 of the taps (round 5, conv_halo.hip): 225.3 GFLOP."""
    ch, mult, nrb = dd['ch'], list(dd['ch_mult']), dd['num_res_blocks']
    res = dd['resolution'] >> (len(mult) - 1)
    fl = 0.0

    def conv(cin, cout, k, hw):
        return 2.0 * hw * hw * cin * cout * k * k

    def resblock(cin, cout, hw):
        f = conv(cin, cout, 3, hw) + conv(cout, cout, 3, hw)
        return f + (conv(cin, cout, 1, hw) if cin != cout else 0.0)

    def attn(c, hw):
        t = hw * hw
        return 4 * conv(c, c, 1, hw) + 2 * 2.0 * t * t * c
    block_in = ch * mult[-1]
    fl += conv(embed_dim, dd['z_channels'], 1, res) + conv(dd['z_channels'], block_in, 3, res)
    fl += 2 * resblock(block_in, block_in, res) + attn(block_in, res)
    for lvl in reversed(range(len(mult))):
        block_out = ch * mult[lvl]
        for _ in range(nrb + 1):
            fl += resblock(block_in, block_out, res)
            block_in = block_out
            if res in dd['attn_resolutions']:
                fl += attn(block_in, res)
        if lvl != 0:
            res *= 2
            # the engine's own eligibility test for the sub-pixel form (VaeRun::conv + rq_conv_halo_subpixel_supported, engine_vae.hip /
            # conv_halo.hip): halo kernel available at that layer (output side >= 64, or 32 with the low-resolution halo rule; whole
            # 8 x 32 output tiles; Cin % 64 == 0, Cout % 128 == 0), source image in whole 8 x 32 tiles, no A/B switch set
            src = res // 2
            halo = (res >= 64 or os.environ.get('RQAMD_HALO_LOWRES', '1') != '0') and res % 32 == 0 and res % 8 == 0 \
                and block_in % 128 == 0 and res * res <= 65536
            off = any(os.environ.get(k) for k in ('RQAMD_NO_HALO', 'RQAMD_NO_HALO_UPS', 'RQAMD_NO_UPS_SUBPIXEL'))
            sub = executed and halo and not off and src % 32 == 0 and src % 8 == 0
            fl += conv(block_in, block_in, 3, res) * (4.0 / 9.0 if sub else 1.0)
    return fl + conv(block_in, dd['out_ch'], 3, res)


def encoder_flops_per_image(dd, embed_dim):
    """Convolution + attention-GEMM FLOPs of Encoder.forward + quant_conv for one image (modules.py:14-98 of the reference), from
    the ddconfig alone: 134.2 GFLOP for the released 256x256 shapes.  The residual quantiser's 2.1 GFLOP (fp32) are counted by
    rq_roofline, not here."""
    ch, mult, nrb = dd['ch'], list(dd['ch_mult']), dd['num_res_blocks']
    res = dd['resolution']

    def conv(cin, cout, k, hw):
        return 2.0 * hw * hw * cin * cout * k * k

    def resblock(cin, cout, hw):
        f = conv(cin, cout, 3, hw) + conv(cout, cout, 3, hw)
        return f + (conv(cin, cout, 1, hw) if cin != cout else 0.0)

    def attn(c, hw):
        t = hw * hw
        return 4 * conv(c, c, 1, hw) + 2 * 2.0 * t * t * c
    fl = conv(dd['in_channels'], ch, 3, res)
    block_in = ch
    for lvl in range(len(mult)):
        block_out = ch * mult[lvl]
        for _ in range(nrb):
            fl += resblock(block_in, block_out, res)
            block_in = block_out
            if res in dd['attn_resolutions']:
                fl += attn(block_in, res)
        if lvl != len(mult) - 1:
            res //= 2
            fl += conv(block_in, block_in, 3, res)                   # Downsample: 3x3 stride-2 conv (layers.py:39-57)
    fl += 2 * resblock(block_in, block_in, res) + attn(block_in, res)
    z = dd['z_channels'] * (2 if dd.get('double_z') else 1)
    return fl + conv(block_in, z, 3, res) + conv(dd['z_channels'], embed_dim, 1, res)


def driver_loop(vae, ar, B, device, top_k, top_p, steps, warmup):
    """The timed body of the reference's throughput script, verbatim (measure_throughput/__main__.py:293-301):
        codes = model_ar.sample(empty_sample, model_aux=model_aux, cond=empty_cond)
        chunks = codes.chunk(batch_size)
        pixels = torch.cat([model_aux.decode_code(chunk) for chunk in chunks], dim=0)
        _ = (0.5 * pixels + 0.5).clamp(0, 1)
    with its own event placement (start / middle / end), on this package's models."""
    es = torch.zeros((B,) + tuple(ar.block_size), device=device, dtype=torch.long)
    ec = torch.zeros((B, ar.block_size_cond), device=device, dtype=torch.long)
    st = vae._ahead
    calls0, hits0 = st.engine_calls, st.hits
    t_ar = t_dec = 0.0
    el = 0.0
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            calls0, hits0 = st.engine_calls, st.hits
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        codes = ar.sample(es, model_aux=vae, cond=ec, top_k=top_k, top_p=top_p)
        ev[1].record()
        chunks = codes.chunk(B)
        pixels = torch.cat([vae.decode_code(chunk) for chunk in chunks], dim=0)
        _ = (0.5 * pixels + 0.5).clamp(0, 1)
        ev[2].record()
        if it >= warmup:
            ev[2].synchronize()
            t_ar += ev[0].elapsed_time(ev[1])
            t_dec += ev[1].elapsed_time(ev[2])
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    return {'images_per_sec': B * steps / el, 'ar_ms_per_image': t_ar / (steps * B), 'decode_ms_per_image': t_dec / (steps * B),
            'decode_engine_calls_per_batch': (st.engine_calls - calls0) / steps, 'decode_calls_served_from_read_ahead': (st.hits - hits0) / steps,
            'what': 'sample, then torch.cat([decode_code(chunk) for chunk in codes.chunk(B)]) and the clamp: '
                    'measure_throughput/__main__.py:293-301 verbatim'}


def verify_codes(vae, ar, codes, cond, top_k, top_p, n_rows=8):
    """Correctness check of what the timed region produced (run after it): every code in range, and for n_rows rows spread
    over the batch every one of the 256 sampled codes has non-zero probability under the filtered distribution (temperature 1,
    top-k, top-p: the reference's sample_from_logits, utils.py:82-123, restated in oracle/sampler.py) of the teacher-forced
    logits of that row -- computed through the same kernels the batch used (the diagnostics row-scale hook makes the kernel
    selection see the full batch)."""
    import oracle
    from rqvae import _native
    B, V = codes.shape[0], ar.vocab_size
    in_range = bool(int(codes.min()) >= 0 and all(int(codes[..., d].max()) < V[d] for d in range(codes.shape[-1])))
    rows = sorted(set(int(r) for r in np.linspace(0, B - 1, n_rows)))
    sub = codes[rows].contiguous()
    _native.dbg_set_row_scale(max(1, (B + len(rows) - 1) // len(rows)))
    try:
        logits = ar.teacher_forced_logits(sub, vae, cond=cond[rows].contiguous())
    finally:
        _native.dbg_set_row_scale(1)
    logits = logits.cpu().numpy()
    sub = sub.cpu().numpy()
    outside, outside_relaxed = count_outside_support(logits, sub, top_k, top_p)
    n = sub.size
    return {'verified': bool(in_range and outside_relaxed == 0 and outside <= max(2, n // 200)), 'codes_in_range': in_range,
            'rows_teacher_forced': len(rows), 'codes_checked': n, 'codes_outside_filtered_support': outside,
            'codes_outside_relaxed_support': outside_relaxed,
            'what': 'post-timed-region: codes of the last timed step; filtered support via the oracle sampler on teacher-forced logits '
                    '(relaxed = top-k + 2 %, top-p + 0.01: no code may lie outside it)'}


def count_outside_support(logits, codes, top_k, top_p):
    """(strict, relaxed) numbers of codes[r, h, w, d] that have zero probability under the filtered distribution of logits[r, h, w, d, :]
    (numpy; the oracle's restatement of sample_from_logits).  See verify_codes for the two filters."""
    import oracle
    sub = codes
    rows = range(sub.shape[0])
    H, W, D = sub.shape[1:]
    # strict: the filter of the run; relaxed: top-k + 2 %, top-p + 0.01 -- the margin a code at the very edge of the support needs when
    # the teacher-forced logits come through other kernel variants than the sampled ones (text-conditioned models: the 8 checked rows
    # take the small-batch prefill / attention forms, and bf16 logits that differ in the last bits move the top-p boundary by a token:
    # 7-8 of 2048 codes on the CC-3M / 3.9B text shapes, with the round-2 library as well; 0 on the class-conditional models)
    outside = outside_relaxed = 0
    k_rel = None if top_k is None else int(np.ceil(top_k * 1.02))
    p_rel = None if top_p is None else min(1.0, top_p + 0.01)
    for h in range(H):
        for w in range(W):
            for d in range(D):
                pr = oracle.filtered_probs(logits[:, h, w, d], 1.0, top_k, top_p)
                miss = pr[np.arange(len(rows)), sub[:, h, w, d]] <= 0
                outside += int(miss.sum())
                if miss.any():
                    pr2 = oracle.filtered_probs(logits[:, h, w, d], 1.0, k_rel, p_rel)
                    outside_relaxed += int((pr2[np.arange(len(rows)), sub[:, h, w, d]] <= 0).sum())
    return outside, outside_relaxed


def timed_batch(vae, ar, B, device, top_k, top_p, steps, warmup):
    """sample -> decode -> clamp at per-GPU batch B; returns (img/s, AR ms/img, decode ms/img)."""
    es = torch.zeros((B,) + tuple(ar.block_size), device=device, dtype=torch.long)
    ec = torch.zeros((B, ar.block_size_cond), device=device, dtype=torch.long)
    for _ in range(warmup):
        one_step(vae, ar, es, ec, None, top_k, top_p)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_ar = t_dec = 0.0
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ev[0].record()
        codes = ar.sample(es, model_aux=vae, cond=ec, top_k=top_k, top_p=top_p)
        ev[1].record()
        pixels = vae.decode_code(codes)
        pixels.mul_(0.5).add_(0.5).clamp_(0, 1)
        ev[2].record()
        ev[2].synchronize()
        t_ar += ev[0].elapsed_time(ev[1])
        t_dec += ev[1].elapsed_time(ev[2])
        del pixels
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    return B * steps / el, t_ar / (steps * B), t_dec / (steps * B), es, ec


def per_image_decode(vae, ar, device, n=64):
    """The unchanged drivers decode ONE image per call (measure_throughput/__main__.py:297-299,
    main_sampling_fid.py:223): torch.cat([decode_code(codes[i:i+1]) for i in range(B)]).  `ms_per_image`: that loop as the
    drivers run it (row views of one code batch: served by the read-ahead of RQVAE.decode_code); `cold_ms_per_image`: the
    same rows passed as independent tensors, one engine call (graph replay) per image."""
    V = ar.vocab_size[0]
    codes = torch.randint(0, V, (n,) + tuple(ar.block_size), device=device)
    warm = torch.randint(0, V, (n,) + tuple(ar.block_size), device=device)
    for i in range(n):
        vae.decode_code(warm[i:i + 1])
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = vae._ahead
    calls0 = st.engine_calls
    t0 = time.perf_counter()
    e0.record()
    pixels = torch.cat([vae.decode_code(codes[i:i + 1]) for i in range(n)], dim=0)
    pixels = (0.5 * pixels + 0.5).clamp(0, 1)
    e1.record()
    e1.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / n
    ms = e0.elapsed_time(e1) / n
    calls = st.engine_calls - calls0
    singles = [codes[i:i + 1].clone() for i in range(n)]
    for i in range(4):
        vae.decode_code(singles[i])
    torch.cuda.synchronize(device)
    e0.record()
    cold = torch.cat([vae.decode_code(c) for c in singles], dim=0)
    cold = (0.5 * cold + 0.5).clamp(0, 1)
    e1.record()
    e1.synchronize()
    cold_ms = e0.elapsed_time(e1) / n
    return {'ms_per_image': ms, 'wall_ms_per_image': wall, 'images_per_sec': 1e3 / ms, 'images': n, 'engine_calls': calls,
            'cold_ms_per_image': cold_ms, 'bit_identical_to_cold': bool(torch.equal(pixels, cold)),
            'what': 'decode_code(codes[i:i+1]) one image per call + cat + clamp, as measure_throughput/__main__.py:297-299 does'}


def per_image_recon(vae, device, n=64):
    """The rFID loop (rqvae/metrics/fid.py:167-169): stage1_model(imgs[i:i+1])[0] on ONE image per call = encode + residual
    quantisation + commitment loss + decode.  `ms_per_image`: the loop as fid.py runs it (row views of one image batch: served by
    the read-ahead of RQVAE.forward); `cold_ms_per_image`: the same images as independent tensors, one engine pass per image."""
    x = torch.randn((n, 3, 256, 256), device=device).clamp(-1, 1)
    warm = torch.randn((n, 3, 256, 256), device=device).clamp(-1, 1)
    for i in range(n):
        vae(warm[i:i + 1])
    del warm
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    outs = torch.cat([vae(x[i:i + 1])[0] for i in range(n)], dim=0)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    singles = [x[i:i + 1].clone() for i in range(n)]
    for i in range(3):
        vae(singles[i])
    torch.cuda.synchronize(device)
    e0.record()
    cold = torch.cat([vae(t)[0] for t in singles], dim=0)
    e1.record()
    e1.synchronize()
    cold_ms = e0.elapsed_time(e1) / n
    return {'ms_per_image': ms, 'images_per_sec': 1e3 / ms, 'images': n, 'cold_ms_per_image': cold_ms,
            'bit_identical_to_cold': bool(torch.equal(outs, cold)),
            'what': 'stage1_model(img[i:i+1])[0]: encode -> RQ -> decode, one image per call (rqvae/metrics/fid.py:167-169)'}


def rq_roofline(vae, device, n_img=256):
    """Residual quantiser alone (RQBottleneck.quantize on encoder-shaped latents): FLOPs 2*K*D per vector-depth, bytes =
    z read + quants/codes written + codebook once."""
    q = vae.quantizer
    cbs = q.codebook_list()
    K, Dm = cbs[0].shape
    depth = len(cbs)
    z = torch.randn((n_img, 8, 8, Dm), device=device)
    q.quantize(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        q.quantize(z)
    e1.record()
    e1.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    nvec = n_img * 64
    flops = 2.0 * K * Dm * nvec * depth
    byts = nvec * Dm * 4 * (1 + depth) + nvec * depth * 8 + K * Dm * 4
    return {'bound': 'fp32-mfma', 'achieved': flops / sec / 1e12, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 'achieved_GBps': byts / sec / 1e9, 'hbm_frac': byts / sec / 1e9 / HBM_PEAK_GBS,
            'images': n_img, 'K': int(K), 'dim': int(Dm), 'depth': depth, 'us_per_launch': sec * 1e6,
            'images_per_sec': n_img / sec, 'codes_per_sec': n_img * 64 * depth / sec,
            'what': 'RQBottleneck.quantize (all depths, one launch) incl. the cumulative quant_list outputs'}


def dry_run(args, rank, world, local_rank):
    """CPU / gloo rehearsal (tests/test_dist_gloo.py): everything around the kernels -- rank environment, per-rank seeds,
    label sharding, the pixel all-gather in rank order, barrier-bracketed timing with the max over ranks, the JSON line."""
    import torch.distributed as dist
    from rqvae.utils.dist import DistEnv, all_gather_cat
    from rqvae.utils.utils import set_seed
    distenv = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo', init_method='env://', world_size=world, rank=rank)
        distenv = DistEnv(world, rank, local_rank, 1, rank == 0, 'cpu')
    set_seed(0 + rank)
    B = args.batch if args.batch > 0 else 4

    def sync():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        pass
    sync()
    t0 = time.perf_counter()
    checks = []
    for step in range(args.steps):
        pixels = torch.full((B, 3, 4, 4), float(rank)) + torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1) / 1000.0
        if distenv is not None:
            pixels = all_gather_cat(distenv, pixels)
        checks.append(bool(pixels.shape[0] == world * B and all(float(pixels[r * B, 0, 0, 0]) == float(r) for r in range(world))))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'dry-run (no kernels): launcher / sharding / gather / timing rehearsal', 'value': world * B * args.steps / elapsed,
                          'unit': 'images/sec', 'n_gpus': world, 'world_size': world, 'requested_gpus': args.gpus, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'none', 'data': 'dry-run', 'gather_rank_major_ok': all(checks),
                          'config': {'workload': 'dry-run', 'batch_per_gpu': B, 'global_batch': B * world}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args, argv)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}', file=sys.stderr)
    if args.dry_run:
        return dry_run(args, rank, world, local_rank)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback exists)'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    distenv = None
    if world > 1:
        import torch.distributed as dist
        from rqvae.utils.dist import DistEnv
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', init_method='env://', world_size=world, rank=rank)
        assert dist.get_world_size() == world
        distenv = DistEnv(world, rank, local_rank, 1, rank == 0, torch.cuda.get_device_name())
    torch.set_grad_enabled(False)

    from rqvae import presets
    from rqvae.utils.utils import set_seed
    vae, ar, cfg = presets.build(args.model, device=device, seed=0)
    vcfg = presets.RQVAE[presets.RQTRANSFORMER[args.model][1]]
    set_seed(0 + rank)                                   # main_sampling_fid.py:166-169
    B = args.batch if args.batch > 0 else DEFAULT_BATCH.get(args.model, 1024)
    if args.batch <= 0 and world > 1 and args.model in DEFAULT_BATCH_MULTI:
        B = DEFAULT_BATCH_MULTI[args.model]
    # safety net (normally a no-op): the default batch is sized for an empty 288-GB device (KV caches ~17 MB per image at the
    # 1.4B shape).  If this device has less free memory, shrink the batch to what fits -- in whole 256-row GEMM tiles, the same on
    # every rank -- instead of dying in hipMalloc; the line then reports the batch actually used (config.batch_per_gpu).
    E = cfg['embed_dim']
    t_body = cfg['block_size'][0] * cfg['block_size'][1] + max(cfg.get('block_size_cond', 1), 1) - 1
    per_row = 2 * (cfg['body']['n_layer'] * t_body + cfg['head']['n_layer'] * cfg['block_size'][2]) * E * 2 \
        + 12 * E * 4 + cfg['vocab_size'] * 4 + 16 * E * 2 + 3 * 256 * 256 * 4
    free_b, _ = torch.cuda.mem_get_info(device)
    fit = int((free_b - 42e9) // per_row)
    if world > 1:
        import torch.distributed as dist
        t_fit = torch.tensor([fit], device=device, dtype=torch.int64)
        dist.all_reduce(t_fit, op=dist.ReduceOp.MIN)
        fit = int(t_fit.item())
    batch_note = None
    if args.batch <= 0 and fit < B:
        newB = max(256, fit // 256 * 256)
        batch_note = f'default batch {B} does not fit the free device memory ({free_b / 1e9:.0f} GB); using {newB}'
        if rank == 0:
            print('bench.py: ' + batch_note, file=sys.stderr)
        B = newB
    empty_sample = torch.zeros((B,) + tuple(ar.block_size), device=device, dtype=torch.long)
    empty_cond = torch.zeros((B, ar.block_size_cond), device=device, dtype=torch.long)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_step(vae, ar, empty_sample, empty_cond, distenv, args.top_k, args.top_p)
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_ar = t_dec = 0.0
    t0 = time.perf_counter()
    if args.overlap:
        # experiment (not the default): batch i is decoded on a second stream while batch i+1 is being sampled
        s_ar = torch.cuda.Stream(device, priority=-1 if args.overlap_prio == 1 else 0)
        s_dec = torch.cuda.Stream(device, priority=-1 if args.overlap_prio == 2 else 0)
        prev = None
        for i in range(args.steps + 1):
            cur = None
            if i < args.steps:
                with torch.cuda.stream(s_ar):
                    codes = ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=args.top_k, top_p=args.top_p)
                    e = torch.cuda.Event()
                    e.record(s_ar)
                cur = (codes, e)
            if prev is not None:
                with torch.cuda.stream(s_dec):
                    s_dec.wait_event(prev[1])
                    pixels = vae.decode_code(prev[0])
                    pixels.mul_(0.5).add_(0.5).clamp_(0, 1)
                    if distenv is not None:
                        pixels = gather_pixels(distenv, pixels)
                    pixels.record_stream(s_dec)
                    del pixels
            prev = cur
        torch.cuda.synchronize(device)
    for _ in range(0 if args.overlap else args.steps):
        ev[0].record()
        codes = ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=args.top_k, top_p=args.top_p)
        ev[1].record()
        pixels = vae.decode_code(codes)
        pixels.mul_(0.5).add_(0.5).clamp_(0, 1)
        if distenv is not None:
            pixels = gather_pixels(distenv, pixels)
        ev[2].record()
        ev[2].synchronize()
        t_ar += ev[0].elapsed_time(ev[1])
        t_dec += ev[1].elapsed_time(ev[2])
        del pixels                                           # the gathered images (world x 6.4 GB) are freed before the next step
    sync()
    elapsed = time.perf_counter() - t0
    rank_times = [elapsed]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_times = [float(x.item()) for x in allt]
        elapsed = max(rank_times)

    # ---- correctness of what the timed region produced (codes of its last step), after the clock stopped
    verify = None
    if rank == 0 and not args.overlap and not args.no_profile and args.steps > 0:
        try:
            verify = verify_codes(vae, ar, codes, empty_cond, args.top_k, args.top_p)
        except Exception as e:
            verify = {'verified': False, 'error': repr(e)}

    # ---- roofline of the dominant kernel at the timed batch
    roofline = roofline_attn = None
    if rank == 0 and not args.no_profile:
        roofline, roofline_attn = gemm_roofline(ar, vae, empty_sample, empty_cond, args.top_k, args.top_p, device, args.model, B, cfg)

    # ---- what the board lets the matrix pipes do at all: MFMAs alone (no LDS, no memory), once with constant operands (the instruction
    # rate the 2.5 PFLOP/s `peak` is) and once with operands that change every instruction like real activations do -- the power-limited
    # rate.  Reported beside `roofline.frac`, which stays priced against the data-sheet peak.
    if rank == 0 and roofline is not None and roofline.get('bound') == 'mfma':
        try:
            from rqvae import _native
            const_tf = _native.dbg_mfma_rate(mode=0, secs=1.0, device=device)
            sust_tf = _native.dbg_mfma_rate(mode=1, secs=2.0, device=device)
            roofline['sustained_mfma_peak'] = {
                'value': sust_tf, 'unit': 'TFLOP/s', 'constant_operands': const_tf, 'frac_of_it': roofline['achieved'] / sust_tf,
                'in_graph_frac_of_it': (roofline.get('in_graph') or {}).get('achieved', 0.0) / sust_tf if 'achieved' in (roofline.get('in_graph') or {}) else None,
                'what': 'v_mfma_f32_32x32x16_bf16 alone from registers on every CU (rqamd_dbg_mfma_rate), ~2 s: `value` with operands that '
                        'change every instruction (~N(0,1) bf16), `constant_operands` with the same bits every clock.  The board reaches '
                        'the 2.5 PFLOP/s of `peak` only on constant data; on changing data its power limit holds the clock near 1.8 GHz '
                        '(profiles/r05_clock_under_load.txt: the GEMM itself runs at 1.69-1.87 GHz and 1.40 kW, and 1.36 x faster on all-zero activations)'}
        except Exception as e:          # noqa: BLE001
            roofline['sustained_mfma_peak'] = {'error': repr(e)}

    # ---- the same measurement at smaller per-GPU batches (BASELINE configs[3] per-GPU share, reference Fig. 4 batch)
    sweep = []
    if rank == 0 and world == 1 and args.sweep:
        for b in [int(x) for x in args.sweep.split(',') if x]:
            if b == B:
                continue
            ips, a_ms, d_ms, es, ec = timed_batch(vae, ar, b, device, args.top_k, args.top_p, steps=3 if b >= 256 else 5, warmup=1)
            entry = {'batch_per_gpu': b, 'images_per_sec': ips, 'ar_ms_per_image': a_ms, 'decode_ms_per_image': d_ms,
                     'ar_ms_per_batch': a_ms * b}
            if b == 500 and args.model == 'huge':
                entry['vs_reference_fig4'] = {'ratio': ips / A100_FIG4_IMG_S, 'reference_images_per_sec': A100_FIG4_IMG_S,
                                              'what': 'same model (1.4B, 8x8x4 codes) and batch (500) as the reference\'s Fig. 4 point; bf16 on one MI355X '
                                                      'here against fp32 on one A100 there (BASELINE.md §1): same batch, other precision and hardware'}
            if not args.no_profile:
                entry['roofline'], entry['roofline_attn'] = gemm_roofline(ar, vae, es, ec, args.top_k, args.top_p, device, args.model, b, cfg)
            del es, ec
            # the reference script's own loop (one image per decode_code call) at the same batch
            dl = driver_loop(vae, ar, b, device, args.top_k, args.top_p, steps=3 if b >= 256 else 5, warmup=1)
            dl['vs_batched'] = dl['images_per_sec'] / ips
            entry['driver_loop'] = dl
            sweep.append(entry)

    # ---- per-image decode (the drivers' call pattern), BASELINE's second metric (codes/sec) and the quantiser roofline
    pid = pir = enc = rqr = None
    if rank == 0 and not args.no_profile:
        pid = per_image_decode(vae, ar, device)
        pir = per_image_recon(vae, device)
        xb = torch.randn((256, 3, 256, 256), device=device).clamp(-1, 1)
        vae.get_codes(xb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            vae.get_codes(xb)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        enc = {'codes_per_sec': 256 * 256 / ms * 1e3, 'images_per_sec': 256 / ms * 1e3, 'batch': 256,
               'what': 'RQVAE.get_codes: 256x256 encode + depth-4 residual quantisation, 256 codes per image'}
        del xb
        rqr = rq_roofline(vae, device)
        efl = encoder_flops_per_image(vcfg['ddconfig'], vcfg['hparams']['embed_dim'])
        enc['roofline_encode'] = {'bound': 'mfma', 'achieved': efl * 256 / (ms * 1e-3) / 1e12, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': efl * 256 / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 'traffic': None,
                                  'algorithmic_GFLOP_per_image': efl / 1e9,
                                  'what': 'RQVAE.get_codes over 256 images: conv / attention-GEMM FLOPs of Encoder.forward (modules.py:73-98) over the '
                                          'device time of the whole call (the fp32 residual quantiser, ~11 % of it, included in the time, not in the '
                                          'FLOPs); kernel shares: profiles/r06_encode_kernel_stats.md'}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            try:
                cpu = cpu_baseline_reference(args.model, args.top_k, args.top_p)
            except Exception as e:                            # e.g. bytecode of another CPython: say so, use the port
                print(f'bench.py: reference CPU leg failed ({e!r}); falling back to the oracle port', file=sys.stderr)
                cpu = None
            if cpu is None:                                   # no usable oracle/_ref on this box: the oracle port, bounded + scaled
                cpu = cpu_baseline(vae, ar, cfg, vcfg)
        except Exception as e:  # the baseline is reported, never required
            cpu = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e!r}'}

    # ---- the two models BASELINE.json quotes on 8 GPUs (configs[3]: 3.8B at a global batch of 512 = 64 per GPU; configs[4]: the 3.9B
    # text-to-image shape, batch-sharded the same way) at their PER-GPU operating point, on this one GPU: what each of the 8 ranks
    # of those configurations would run.  Random-init weights of the named architecture, like the headline.  The headline model and its
    # 180-GB KV cache are released first.
    also = []
    if rank == 0 and world == 1 and args.also and args.model == 'huge' and not args.overlap:
        try:
            del empty_sample, empty_cond
            if getattr(ar, '_engine', None) is not None:
                ar._engine.close()
                ar._engine = None
            del ar
            torch.cuda.empty_cache()
            for spec in [x for x in args.also.split(',') if x]:
                name, b = spec.split(':')
                b = int(b)
                _, ar2, cfg2 = presets.build(name, device=device, seed=0)
                ips, a_ms, d_ms, es, ec = timed_batch(vae, ar2, b, device, args.top_k, args.top_p, steps=3, warmup=1)
                codes2 = ar2.sample(es, model_aux=vae, cond=ec, top_k=args.top_k, top_p=args.top_p)
                also.append({'model': name, 'workload': WORKLOADS.get(name, name), 'batch_per_gpu': b, 'images_per_sec': ips, 'ar_ms_per_batch': a_ms * b,
                             'decode_ms_per_image': d_ms, 'params_M': sum(p.numel() for p in ar2.parameters()) / 1e6,
                             'codes_in_range': bool(int(codes2.min()) >= 0 and int(codes2.max()) < cfg2['vocab_size']),
                             'what': 'sample -> decode_code -> clamp, 3 timed steps after 1 warm-up, as batch_sweep; parity of this model at full depth: '
                                     'tests/test_gpu_parity_big.py (rqt_in3800m / rqt_txt3900m fixtures)'})
                if getattr(ar2, '_engine', None) is not None:
                    ar2._engine.close()
                    ar2._engine = None
                del ar2, es, ec, codes2
                torch.cuda.empty_cache()
        except Exception as e:            # reported, never required -- and not buried: the line goes to stderr as well
            print(f'bench.py: --also {args.also!r} failed: {e!r}', file=sys.stderr)
            also.append({'error': repr(e)})

    # ---- the opt-in 8-bit key cache next to the default bf16 one ("report both ways"), same model, one moderate batch, measured live;
    # the headline batch both ways: profiles/r05_kv_int8k_ab.txt.  Engines read RQAMD_KV when they are created.
    kv_formats = None
    if rank == 0 and world == 1 and args.formats and args.model == 'huge' and not args.overlap and os.environ.get('RQAMD_KV', 'bf16') == 'bf16':
        kv_formats = []
        kv_before = os.environ.get('RQAMD_KV')
        try:
            # ('fp16': the fp16 build of the engine, what sample(amp=True) runs on -- same MFMA rate, three more mantissa bits)
            for fmt in ('bf16', 'int8k', 'int8kv', 'fp16'):
                os.environ['RQAMD_KV'] = 'bf16' if fmt == 'fp16' else fmt
                _, ar3, cfg3 = presets.build('huge', device=device, seed=0)
                if fmt == 'fp16':
                    _s = ar3.sample
                    ar3.sample = lambda *a, **k: _s(*a, **dict(k, amp=True))
                ips, a_ms, d_ms, es, ec = timed_batch(vae, ar3, 2048, device, args.top_k, args.top_p, steps=2, warmup=1)
                kv_formats.append({'kv_cache': fmt, 'batch_per_gpu': 2048, 'images_per_sec': ips, 'ar_ms_per_image': a_ms,
                                   'kv_bytes_per_image_GB': kv_bytes_per_image(cfg3) / 1e9})      # (reads RQAMD_KV, set above)
                for e3, _sig in list(ar3._engines.values()):
                    e3.close()
                ar3._engines.clear()
                del ar3, es, ec
                torch.cuda.empty_cache()
        except Exception as e:
            print(f'bench.py: storage-format comparison failed: {e!r}', file=sys.stderr)
            kv_formats.append({'error': repr(e)})
        finally:
            if kv_before is None:
                os.environ.pop('RQAMD_KV', None)
            else:
                os.environ['RQAMD_KV'] = kv_before

    # ---- the opt-in fp16 RQ-VAE engine (RQAMD_VAE=fp16: same kernels, IEEE fp16 storage; tests/test_gpu_parity.py::test_vae_fp16_engine has its
    # parity) next to the bf16 default: decode_code of 1024 code maps, measured live
    vae_formats = None
    if rank == 0 and world == 1 and args.formats and not args.overlap and os.environ.get('RQAMD_VAE', 'bf16') == 'bf16':
        vae_formats = []
        vae_before = os.environ.get('RQAMD_VAE')
        try:
            g = torch.Generator(device='cpu').manual_seed(5)
            codes_f = torch.randint(0, vcfg['hparams']['n_embed'], (1024, 8, 8, 4), generator=g).to(device)
            for fmt in ('bf16', 'fp16'):
                os.environ['RQAMD_VAE'] = fmt
                vae3, ar_tmp, _ = presets.build(args.model, device=device, seed=0)
                del ar_tmp
                vae3.decode_code(codes_f)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(2):
                    vae3.decode_code(codes_f)
                e1.record()
                torch.cuda.synchronize()
                vae_formats.append({'vae': fmt, 'batch': 1024, 'decode_ms_per_image': e0.elapsed_time(e1) / (2 * 1024)})
                if vae3._engine is not None:
                    vae3._engine.close()
                    vae3._engine = None
                del vae3
                torch.cuda.empty_cache()
        except Exception as e:
            print(f'bench.py: RQ-VAE storage-format comparison failed: {e!r}', file=sys.stderr)
            vae_formats.append({'error': repr(e)})
        finally:
            if vae_before is None:
                os.environ.pop('RQAMD_VAE', None)
            else:
                os.environ['RQAMD_VAE'] = vae_before

    if rank == 0:
        n_img = world * B * args.steps
        value = n_img / elapsed
        roofline_decode = step_frac = None
        dfl = decoder_flops_per_image(vcfg['ddconfig'], vcfg['hparams']['embed_dim'])
        if not args.overlap and t_dec > 0:
            # MFMA utilisation is priced on the FLOPs the kernels execute (the sub-pixel upsample convs multiply 4/9 of the reference's taps);
            # the reference's own count is reported next to it
            dfx = decoder_flops_per_image(vcfg['ddconfig'], vcfg['hparams']['embed_dim'], executed=True)
            tf = dfx * B * args.steps / (t_dec * 1e-3) / 1e12
            roofline_decode = {'bound': 'mfma', 'achieved': tf, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / MFMA_BF16_PEAK_TFLOPS,
                               'traffic': None,
                               'executed_GFLOP_per_image': dfx / 1e9, 'algorithmic_GFLOP_per_image': dfl / 1e9,
                               'algorithmic_TFLOPs': dfl * B * args.steps / (t_dec * 1e-3) / 1e12, 'ms_per_image': t_dec / (args.steps * B),
                               'what': 'RQ-VAE decode_code + clamp of the timed region: conv / attention-GEMM FLOPs of Decoder.forward '
                                       '(modules.py:171-202) over its device time (events around the decode half of every step)'}
        step_frac = {'achieved_TFLOPs': (dfl + rqt_flops_per_image(cfg)) * value / world / 1e12,
                     'frac': (dfl + rqt_flops_per_image(cfg)) * value / world / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                     'algorithmic_GFLOP_per_image': (dfl + rqt_flops_per_image(cfg)) / 1e9,
                     'what': 'all GEMM + conv FLOPs of one image (transformer decode steps + RQ-VAE decoder) x images/s per GPU / 2.5 PF'}
        out = {
            'metric': f'256x256 images/sec, AR sampling + decode (RQ-Transformer {args.model}, 8x8x4 codes)',
            'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            # BASELINE.json publishes no number for this metric (`published: {}`): null, as the contract says.  The one figure of the
            # reference that exists -- Fig. 4, batch 500, one A100, fp32 -- is set against batch_sweep's 500 entry (`vs_reference_fig4`)
            'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'RQ-Transformer {args.model} ({WORKLOADS.get(args.model, args.model)}) sampling 8x8x4 codes + RQ-VAE decode; '
                                   f'random-init weights, zero condition',
                       'batch_per_gpu': B, 'global_batch': B * world, 'batch_note': batch_note, 'top_k': args.top_k, 'top_p': args.top_p,
                       'kv_cache': os.environ.get('RQAMD_KV', 'bf16') + (' (opt-in: body-stack keys as bytes + one scale per token and head; default bf16)'
                                                                          if os.environ.get('RQAMD_KV', 'bf16') != 'bf16' else ''),
                       'overlap_decode_with_next_sampling': bool(args.overlap), 'world_size': world, 'requested_gpus': args.gpus,
                       'per_rank_seconds': rank_times,
                       'parallelism': parallelism_note(world, B),
                       'vs_baseline_note': 'null: BASELINE.json publishes no number for this metric; the reference\'s Fig. 4 point (batch 500) is '
                                           'compared in batch_sweep[batch_per_gpu = 500].vs_reference_fig4'},
            'ar_ms_per_image': t_ar / (args.steps * B) if not args.overlap else None,
            'decode_ms_per_image': t_dec / (args.steps * B) if not args.overlap else None,
            'verified': None if verify is None else verify['verified'], 'verify': verify,
            'roofline': roofline, 'roofline_attn': roofline_attn, 'roofline_decode': roofline_decode,
            'step_frac_of_mfma_peak': step_frac, 'batch_sweep': sweep, 'baseline_8gpu_models_per_gpu_point': also, 'kv_cache_formats': kv_formats, 'vae_formats': vae_formats, 'per_image_decode': pid, 'per_image_recon': pir, 'roofline_rq': rqr, 'cpu_baseline': cpu, 'rqvae_encode': enc,
        }
        # ---- flat scalars: the driver's record keeps the scalar members of `roofline` / `cpu_baseline` / `config` and only the
        # NAMES of other top-level keys, so everything this repo claims from the line is repeated as plain numbers -- at the top level and inside `roofline`
        flat = {}
        for e in sweep:
            flat[f"b{e['batch_per_gpu']}_images_per_sec"] = e.get('images_per_sec')
            flat[f"b{e['batch_per_gpu']}_ar_ms_per_batch"] = e.get('ar_ms_per_batch')
        for key, r in (('roofline_decode_frac', roofline_decode), ('roofline_attn_frac', roofline_attn), ('roofline_rq_frac', rqr),
                       ('roofline_encode_frac', (enc or {}).get('roofline_encode'))):
            flat[key] = (r or {}).get('frac')
        if enc:
            flat['encode_images_per_sec'] = enc.get('images_per_sec')
            flat['encode_codes_per_sec'] = enc.get('codes_per_sec')
        smp = (roofline or {}).get('sustained_mfma_peak') or {}
        flat['sustained_mfma_peak_TFLOPs'] = smp.get('value')
        flat['gemm_frac_of_sustained'] = smp.get('frac_of_it')
        flat['gemm_in_graph_frac'] = ((roofline or {}).get('in_graph') or {}).get('frac')
        flat['ar_ms_per_image'] = out['ar_ms_per_image']
        flat['decode_ms_per_image'] = out['decode_ms_per_image']
        for a in also:
            if 'model' in a:
                flat[f"{a['model']}_b{a['batch_per_gpu']}_images_per_sec"] = a.get('images_per_sec')
        out.update(flat)
        if roofline is not None:
            roofline.update({k: v for k, v in flat.items() if k not in roofline})
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
