#!/usr/bin/env python
"""bench.py -- 256x256 images/sec of the AR-sampling + decode path (BASELINE.json metric) on MI355X.

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted on --
ImageNet-256 class-conditional RQ-Transformer 1.4B (E 1536 / 24 heads / 42 body + 6 head layers /
V 16384, measure_throughput 'huge', reference measure_throughput/__main__.py:71-92) sampling 8x8x4
codes, then RQ-VAE (104 M) decode_code to 256x256 pixels and the [0,1] clamp, exactly the timed body
of the reference's throughput script (:295-301).  Random-init weights of that architecture
(torch.manual_seed(0), module default inits), zero class condition, synthetic -- there is no network.

One "step" = one batch of B images per GPU: sample -> decode -> clamp (-> pixel all-gather when N > 1,
main_sampling_fid.py:226).  value = N * B * K / max-over-ranks time, inputs resident in HBM.

Extra objects on the JSON line (prompt section 4): "roofline" for the dominant kernel (the bf16 MFMA
weight-streaming GEMM of the decode step, timed live with HIP events on the engine's stream in a
separate profiled pass) and "cpu_baseline" (the numpy oracle on the host cores, bounded sample, rank 0,
N=1 only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'rq-vae-transformer_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
A100_FIG4_IMG_S = 52.6          # reference Fig. 4, 1.4B 8x8x4, batch 500, 1x A100 (fp32)


def build_models(device, model='huge'):
    from oracle import configs as C
    from rqvae.models.rqvae import RQVAE
    from rqvae.models.rqtransformer import RQTransformer
    cfg = {'huge': C.RQT_IN_1400M, 'large': C.RQT_IN_821M, 'medium': C.RQT_FFHQ_355M, 'xhuge': C.RQT_IN_3800M,
           'tiny': C.RQT_TINY}[model]
    vcfg = C.VAE_TINY if model == 'tiny' else (C.VAE_FFHQ if model == 'medium' else C.VAE_IMAGENET)
    torch.manual_seed(0)
    with torch.device(device):
        hps, dd = vcfg
        vae = RQVAE(**hps, ddconfig=dd, checkpointing=False).eval()
        ar = RQTransformer(cfg).eval()
    return vae, ar, cfg, vcfg


def one_step(vae, ar, empty_sample, empty_cond, distenv, top_k, top_p):
    codes = ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=top_k, top_p=top_p)
    pixels = vae.decode_code(codes)
    pixels.mul_(0.5).add_(0.5).clamp_(0, 1)             # in place: 6.4 GB per temporary at B = 8192
    if distenv is not None and distenv.world_size > 1:
        from rqvae.utils.dist import all_gather_cat
        pixels = all_gather_cat(distenv, pixels)
    return codes, pixels


def cpu_baseline(vae, ar, cfg, vcfg, n_pos=3, batch=2):
    """Oracle (numpy restatement, kind 'port') on the host cores: `n_pos` spatial positions (n_pos body
    steps + 4*n_pos head/sampler steps) of a batch-`batch` sample, scaled to the 64 positions of an image,
    plus one full decode_code of one image."""
    import oracle
    import threadpoolctl
    aparams = {k: v.detach().float().cpu().numpy() for k, v in ar.state_dict().items()}
    vparams = {k: v.detach().float().cpu().numpy() for k, v in vae.state_dict().items()}
    hps, dd = vcfg
    orc = oracle.RQTransformerOracle(cfg, aparams)
    ov = oracle.RQVAEOracle(hps, dd, vparams)
    H, W, D = cfg['block_size']
    cores = max(i['num_threads'] for i in threadpoolctl.threadpool_info() if i.get('user_api') == 'blas')
    part = np.zeros((batch, H, W, D), np.int64)
    t0 = time.time()
    xs = orc.sample(part, ov.codebooks, cond=np.zeros((batch, 1), np.int64), max_steps=n_pos * D)
    t_ar = (time.time() - t0) * (H * W / n_pos) / batch            # seconds per image
    t0 = time.time()
    ov.decode_code(xs[:1])
    t_dec = time.time() - t0
    return {'value': 1.0 / (t_ar + t_dec), 'unit': 'images/sec', 'cores': int(cores), 'kind': 'port',
            'sample': f'numpy oracle, fp32: {n_pos} of {H * W} spatial positions at batch {batch} (scaled x{H * W / n_pos:.1f}) '
                      f'= {t_ar:.1f} s/img AR + one full 256x256 decode_code = {t_dec:.1f} s/img'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('RQ_BENCH_BATCH', 8192)), help='images per GPU per step')
    ap.add_argument('--model', default='huge')
    # BASELINE.json configs[2]: top-k=1024 / top-p=0.95 (0 / 1.0 = the reference defaults top_k=None, top_p=None)
    ap.add_argument('--top-k', type=int, default=1024)
    ap.add_argument('--top-p', type=float, default=0.95)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--overlap', action='store_true', help='experiment: decode batch i while sampling batch i+1 (two streams)')
    args = ap.parse_args()
    if args.top_k is not None and args.top_k <= 0:
        args.top_k = None
    if args.top_p is not None and args.top_p >= 1.0:
        args.top_p = None

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback exists)'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    distenv = None
    if world > 1:
        import torch.distributed as dist
        from rqvae.utils.dist import DistEnv
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', init_method='env://', world_size=world, rank=rank)
        distenv = DistEnv(world, rank, local_rank, 1, rank == 0, torch.cuda.get_device_name())
    torch.set_grad_enabled(False)

    from rqvae.utils.utils import set_seed
    vae, ar, cfg, vcfg = build_models(device, args.model)
    set_seed(0 + rank)                                   # main_sampling_fid.py:166-169
    B = args.batch
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
        s_ar, s_dec = torch.cuda.Stream(device), torch.cuda.Stream(device)
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
                        from rqvae.utils.dist import all_gather_cat
                        pixels = all_gather_cat(distenv, pixels)
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
            from rqvae.utils.dist import all_gather_cat
            pixels = all_gather_cat(distenv, pixels)
        ev[2].record()
        ev[2].synchronize()
        t_ar += ev[0].elapsed_time(ev[1])
        t_dec += ev[1].elapsed_time(ev[2])
        del pixels                                           # the gathered images (world x 6.4 GB) are freed before the next step
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (profiled pass: every GEMM launch bracketed by HIP events on
    # the engine's stream; graphs off for this pass only)
    roofline = None
    if rank == 0 and not args.no_profile:
        eng = ar._eng()
        eng.set_profile(True)
        ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=args.top_k, top_p=args.top_p)
        torch.cuda.synchronize(device)
        pf = eng.get_profile()
        eng.set_profile(False)
        if pf['gemm_launches'] > 0 and pf['gemm_ms_total'] > 0:
            sec = pf['gemm_ms_total'] * 1e-3
            gbs = pf['gemm_bytes'] / sec / 1e9
            tfl = pf['gemm_flops'] / sec / 1e12
            ridge = MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            intensity = pf['gemm_flops'] / pf['gemm_bytes']
            if intensity < ridge:
                roofline = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS}
            else:
                roofline = {'bound': 'mfma', 'achieved': tfl, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': tfl / MFMA_BF16_PEAK_TFLOPS}
            traffic, traffic_src = None, None
            tp = os.path.join(ROOT, 'profiles', f'r01_gemm_traffic_m{B}.json')
            if args.model == 'huge' and os.path.exists(tp):
                # PMC counters cannot be collected from inside the timed run; this is the committed result of
                # scripts/gpu_pmc2.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) for the same GEMM
                # shapes at the same batch rows, launch-weighted like `achieved`
                with open(tp) as f:
                    tj = json.load(f)
                traffic, traffic_src = tj['hbm_bytes_per_launch_weighted'], os.path.relpath(tp, ROOT)
            roofline.update({'traffic': traffic, 'traffic_source': traffic_src,
                             'algorithmic_bytes_per_launch': pf['gemm_bytes'] / pf['gemm_launches'], 'kernel': 'gemm_bf16_kernel', 'launches_per_batch': pf['gemm_launches'],
                             'avg_launch_us': pf['gemm_ms_total'] * 1e3 / pf['gemm_launches'],
                             'algorithmic_GB_per_batch': pf['gemm_bytes'] / 1e9, 'algorithmic_TFLOP_per_batch': pf['gemm_flops'] / 1e12,
                             'achieved_GBps': gbs, 'achieved_TFLOPs': tfl, 'flop_per_byte': intensity})

    # ---- BASELINE.json's second metric, codes/sec of the RQ-VAE (encode + residual quantisation), outside the timed region
    enc = None
    if rank == 0 and not args.no_profile:
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
               'what': 'RQVAE.get_codes: 256x256 encode + depth-4 residual quantisation (K=16384), 256 codes per image'}
        del xb

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(vae, ar, cfg, vcfg)
        except Exception as e:  # the baseline is reported, never required
            cpu = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e!r}'}

    if rank == 0:
        n_img = world * B * args.steps
        value = n_img / elapsed
        out = {
            'metric': '256x256 images/sec, AR sampling + decode (ImageNet RQ-Transformer 1.4B, 8x8x4 codes)',
            'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': value / (A100_FIG4_IMG_S * world) if args.model == 'huge' else None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'ImageNet-256 class-conditional RQ-Transformer {args.model} sampling 8x8x4 codes + RQ-VAE decode '
                                   f'(BASELINE configs[2]); random-init weights, zero class condition',
                       'batch_per_gpu': B, 'global_batch': B * world, 'top_k': args.top_k, 'top_p': args.top_p, 'overlap_decode_with_next_sampling': bool(args.overlap),
                       'parallelism': f'replica x{world}, image batches sharded, one pixel all-gather per step' if world > 1 else 'single GPU',
                       'vs_baseline_ref': 'reference Fig.4: 52.6 img/s, 1.4B 8x8x4, batch 500, 1x A100 fp32 (BASELINE.md §1), per GPU'},
            'ar_ms_per_image': t_ar / (args.steps * B), 'decode_ms_per_image': t_dec / (args.steps * B),
            'roofline': roofline, 'cpu_baseline': cpu, 'rqvae_encode': enc,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
