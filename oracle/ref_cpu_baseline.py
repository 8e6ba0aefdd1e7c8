#!/usr/bin/env python
"""bench.py's cpu_baseline leg, kind "reference" (TEST INFRASTRUCTURE): the REFERENCE's own modules (oracle/_ref, see
oracle/build_ref.py) on the host cores -- RQTransformer.sample (its own Python loop, torch CPU kernels, fp32, KV cache on)
followed by the throughput script's one-image-per-call decode loop and clamp (measure_throughput/__main__.py:293-301) -- on
ONE batch of `--batch` images of the named model shape with module-default random weights.  Run as a separate process: the
reference's package is also called `rqvae`, so it must not share an interpreter with the product.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', required=True, help='JSON: {"rqt": stage-2 arch dict, "vae": {"hparams": ..., "ddconfig": ...}}')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--top-k', type=int, default=0)
    ap.add_argument('--top-p', type=float, default=1.0)
    ap.add_argument('--positions', type=int, default=0, help='sample only the last N spatial positions (start_loc; the prefix is prefilled in one pass); 0 = all')
    ap.add_argument('--decode', type=int, default=0, help='decode only the first N images of the batch; 0 = all')
    ap.add_argument('--threads', type=int, default=0)
    a = ap.parse_args()
    stub = types.ModuleType('omegaconf')            # the one import the model files make that this image lacks (configs.py:18)
    stub.OmegaConf = type('OmegaConf', (), {})
    stub.MISSING = '???'
    stub.DictConfig = dict
    sys.modules['omegaconf'] = stub
    sys.path.insert(0, os.path.join(HERE, '_ref'))
    import torch
    from rqvae.models.rqvae import RQVAE              # the reference (sourceless .pyc)
    from rqvae.models.rqtransformer import RQTransformer
    assert RQVAE.__module__.startswith('rqvae.') and os.path.join('oracle', '_ref') in sys.modules['rqvae'].__file__

    class Cfg(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

        def copy(self):
            return to_cfg(json.loads(json.dumps(self)))

    def to_cfg(d):
        return Cfg({k: to_cfg(v) if isinstance(v, dict) else v for k, v in d.items()})
    arch = json.loads(a.arch)
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    vae = RQVAE(**arch['vae']['hparams'], ddconfig=arch['vae']['ddconfig'], checkpointing=False).eval()
    ar = RQTransformer(to_cfg(arch['rqt'])).eval()
    B = a.batch
    empty_sample = torch.zeros(B, *ar.block_size, dtype=torch.long)
    empty_cond = torch.zeros(B, ar.block_size_cond, dtype=torch.long)
    top_k = a.top_k if a.top_k > 0 else None
    top_p = a.top_p if a.top_p < 1.0 else None
    H, W, D = ar.block_size
    n_pos = a.positions if 0 < a.positions < H * W else H * W
    first = H * W - n_pos
    n_dec = a.decode if 0 < a.decode < B else B
    t0 = time.time()
    # start_loc: the reference skips the positions before it and prefills them (whatever partial_sample holds there) in the first
    # cached step (transformers.py:346-350) -- a bounded sample of the 64-position loop: n_pos positions + one prefill pass
    codes = ar.sample(empty_sample, model_aux=vae, cond=empty_cond, top_k=top_k, top_p=top_p, start_loc=(first // W, first % W))
    t1 = time.time()
    pixels = torch.cat([vae.decode_code(chunk) for chunk in codes[:n_dec].chunk(n_dec)], dim=0)
    _ = (0.5 * pixels + 0.5).clamp(0, 1)
    t2 = time.time()
    ar_s_img = (t1 - t0) * (H * W / n_pos) / B          # seconds per image, scaled from n_pos positions (+ the prefill) to H * W
    dec_s_img = (t2 - t1) / n_dec
    print(json.dumps({'batch': B, 'positions': n_pos, 'of_positions': H * W, 'decoded': n_dec, 'ar_s': t1 - t0, 'decode_s': t2 - t1,
                      'images_per_sec': 1.0 / (ar_s_img + dec_s_img), 'ar_s_per_image': ar_s_img, 'decode_s_per_image': dec_s_img,
                      'threads': torch.get_num_threads(), 'pixels_shape': list(pixels.shape),
                      'codes_in_range': bool(int(codes.min()) >= 0 and int(codes.max()) < arch['rqt']['vocab_size'])}))


if __name__ == '__main__':
    main()
