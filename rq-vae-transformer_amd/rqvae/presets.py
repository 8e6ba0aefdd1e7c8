"""Named model shapes of the released checkpoints / the throughput script, as plain config dicts.

Restates (does not import) the reference's YAML / builder tables: configs/imagenet256/stage1/in256-rqvae-8x8x4.yaml:10-31,
configs/ffhq/stage1/ffhq256-rqvae-8x8x4.yaml:9-31, configs/**/stage2/*.yaml:9-30, configs/cc3m/*.yaml:9-30,
measure_throughput/__main__.py:69-210 and measure_throughput/rq_defaults.yaml.  Used by bench.py and the examples; the
dicts go through rqvae.utils.config.augment_arch_defaults / create_model like a loaded config.yaml would."""
import copy


def rqvae_arch(n_embed=16384, attn_resolutions=(8,), ch=128, ch_mult=(1, 1, 2, 2, 4, 4), resolution=256, z_channels=256,
               embed_dim=256, num_res_blocks=2, depth=4):
    hw = resolution // 2 ** (len(ch_mult) - 1)
    return {'type': 'rq-vae', 'code_hier': 1,
            'hparams': dict(bottleneck_type='rq', embed_dim=embed_dim, n_embed=n_embed, latent_shape=[hw, hw, embed_dim],
                            code_shape=[hw, hw, depth], shared_codebook=True, decay=0.99, restart_unused_codes=True,
                            loss_type='mse', latent_loss_weight=0.25),
            'ddconfig': dict(double_z=False, z_channels=z_channels, resolution=resolution, in_channels=3, out_ch=3, ch=ch,
                             ch_mult=list(ch_mult), num_res_blocks=num_res_blocks, attn_resolutions=list(attn_resolutions),
                             dropout=0.0)}


def rqtransformer_arch(embed_dim, n_head, n_body, n_head_layers, vocab_size, vocab_size_cond=1000, block_size_cond=1,
                       block_size=(8, 8, 4), input_embed_dim=256):
    block = dict(embed_dim=embed_dim, n_head=n_head, mlp_bias=True, attn_bias=True, attn_pdrop=0.0, resid_pdrop=0.1, gelu='v1')
    return dict(type='rq-transformer', block_size=list(block_size), vocab_size=vocab_size, vocab_size_cond=vocab_size_cond,
                block_size_cond=block_size_cond, embed_dim=embed_dim, input_embed_dim=input_embed_dim, shared_tok_emb=True,
                shared_cls_emb=True, input_emb_vqvae=True, head_emb_vqvae=True, cumsum_depth_ctx=True, embd_pdrop=0.0,
                body=dict(n_layer=n_body, block=copy.deepcopy(block)), head=dict(n_layer=n_head_layers, block=copy.deepcopy(block)))


RQVAE = {
    'imagenet': rqvae_arch(16384, (8,)),                       # in256-rqvae-8x8x4
    'ffhq': rqvae_arch(2048, (16,)),                           # ffhq256-rqvae-8x8x4
    'tiny': rqvae_arch(500, (8,), ch=64, ch_mult=(1, 2), resolution=16, z_channels=64, embed_dim=64, num_res_blocks=1),
}

RQTRANSFORMER = {
    # name: (arch, rqvae preset)        measure_throughput names in quotes
    'medium': (rqtransformer_arch(1024, 16, 24, 4, 2048, vocab_size_cond=1), 'ffhq'),        # FFHQ 355M (BASELINE configs[1])
    'small': (rqtransformer_arch(1536, 24, 12, 4, 16384), 'imagenet'),                        # ImageNet 480M
    'large': (rqtransformer_arch(1536, 24, 24, 4, 16384), 'imagenet'),                        # ImageNet 821M
    'huge': (rqtransformer_arch(1536, 24, 42, 6, 16384), 'imagenet'),                         # ImageNet 1.4B (BASELINE configs[2])
    'xhuge': (rqtransformer_arch(2560, 40, 42, 6, 16384), 'imagenet'),                        # ImageNet 3.8B (BASELINE configs[3])
    'cc3m': (rqtransformer_arch(1280, 20, 26, 4, 16384, vocab_size_cond=16384, block_size_cond=32), 'imagenet'),   # CC-3M 654M
    # BASELINE configs[4]: the 3.9B text-to-image model = 3.8B dims + 64 BPE tokens of a 16384-word vocabulary
    'txt3900m': (rqtransformer_arch(2560, 40, 42, 6, 16384, vocab_size_cond=16384, block_size_cond=64), 'imagenet'),
    'tiny': (rqtransformer_arch(128, 2, 2, 2, 500, vocab_size_cond=10, block_size=(4, 4, 4), input_embed_dim=64), 'tiny'),
}


def build(name, device=None, seed=0):
    """(rqvae, rqtransformer, arch dict) with module-default random initialisation under torch.manual_seed(seed) -- the
    synthetic-weight setting of measure_throughput (no checkpoints are downloadable offline)."""
    import torch
    from .models import create_model
    from .utils.config import Config, augment_arch_defaults
    arch, vname = RQTRANSFORMER[name]
    torch.manual_seed(seed)
    ctx = torch.device(device) if device is not None else torch.device('cpu')
    with ctx:
        vae, _ = create_model(augment_arch_defaults(Config(copy.deepcopy(RQVAE[vname]))))
        ar, _ = create_model(augment_arch_defaults(Config(copy.deepcopy(arch))))
    return vae.eval(), ar.eval(), copy.deepcopy(arch)
