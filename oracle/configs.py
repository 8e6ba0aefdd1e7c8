"""Named shapes used by tests / bench (restating reference configs; TEST INFRASTRUCTURE).

Sources: configs/imagenet256/stage1/in256-rqvae-8x8x4.yaml:10-31,
configs/ffhq/stage1/ffhq256-rqvae-8x8x4.yaml:9-31, configs/**/stage2/*.yaml:9-30,
measure_throughput/__main__.py:69-210 and rq_defaults.yaml of the reference.
"""
import copy


def vae(n_embed=16384, attn_res=(8,), ch=128, ch_mult=(1, 1, 2, 2, 4, 4), resolution=256,
        z_channels=256, embed_dim=256, num_res_blocks=2, depth=4):
    f = 2 ** (len(ch_mult) - 1)
    hw = resolution // f
    hps = dict(bottleneck_type='rq', embed_dim=embed_dim, n_embed=n_embed,
               latent_shape=[hw, hw, embed_dim], code_shape=[hw, hw, depth], shared_codebook=True,
               decay=0.99, restart_unused_codes=True, loss_type='mse', latent_loss_weight=0.25)
    dd = dict(double_z=False, z_channels=z_channels, resolution=resolution, in_channels=3, out_ch=3,
              ch=ch, ch_mult=list(ch_mult), num_res_blocks=num_res_blocks,
              attn_resolutions=list(attn_res), dropout=0.0)
    return hps, dd


VAE_IMAGENET = vae(16384, (8,))
VAE_FFHQ = vae(2048, (16,))
# 16x16 image -> 8x8x4 codes; exercises conv s1/s2, upsample, nin_shortcut, attention, GN
VAE_TINY = vae(n_embed=500, attn_res=(8,), ch=64, ch_mult=(1, 2), resolution=16, z_channels=64,
               embed_dim=64, num_res_blocks=1)
# the same with ddconfig.resamp_with_conv = False (layers.py:20-57: bare nearest upsample / 2 x 2 average pool; no released config) -- round 6
VAE_TINY_NORESAMP = (copy.deepcopy(VAE_TINY[0]), dict(copy.deepcopy(VAE_TINY[1]), resamp_with_conv=False))


def rqt(embed_dim, n_head, n_body, n_headl, vocab, vocab_cond=1000, block_cond=1,
        block_size=(8, 8, 4), input_embed_dim=256, gelu='v1'):
    blk = dict(embed_dim=embed_dim, n_head=n_head, mlp_bias=True, attn_bias=True,
               attn_pdrop=0.0, resid_pdrop=0.1, gelu=gelu)
    return dict(type='rq-transformer', block_size=list(block_size), vocab_size=vocab,
                vocab_size_cond=vocab_cond, block_size_cond=block_cond, embed_dim=embed_dim,
                input_embed_dim=input_embed_dim, shared_tok_emb=True, shared_cls_emb=True,
                input_emb_vqvae=True, head_emb_vqvae=True, cumsum_depth_ctx=True, embd_pdrop=0.0,
                gelu=gelu,
                body=dict(n_layer=n_body, block=copy.deepcopy(blk)),
                head=dict(n_layer=n_headl, block=copy.deepcopy(blk)))


RQT_FFHQ_355M = rqt(1024, 16, 24, 4, 2048, vocab_cond=1)
RQT_IN_480M = rqt(1536, 24, 12, 4, 16384)
RQT_IN_821M = rqt(1536, 24, 24, 4, 16384)
RQT_IN_1400M = rqt(1536, 24, 42, 6, 16384)
RQT_IN_3800M = rqt(2560, 40, 42, 6, 16384)
RQT_CC3M_654M = rqt(1280, 20, 26, 4, 16384, vocab_cond=16384, block_cond=32)
# tiny: 4x4x4 codes, E=128 (2 heads x 64), pairs with a 500x64 codebook
RQT_TINY = rqt(128, 2, 2, 2, 500, vocab_cond=10, block_size=(4, 4, 4), input_embed_dim=64)
# text-conditioned variant: 4 conditioning tokens from a 20-word vocabulary (exercises the cond-prefix prefill)
RQT_TINY_TXT = rqt(128, 2, 2, 2, 500, vocab_cond=20, block_cond=4, block_size=(4, 4, 4), input_embed_dim=64)
# one real-width layer of each stack (E=1536, 24 heads, V=16384) -- layer-count-independent parity
RQT_WIDE = rqt(1536, 24, 2, 1, 16384)
# 3.8B layer shapes (E=2560, 40 heads), 2 body + 1 head layers -- width-dependent kernel variants (BASELINE configs[3])
RQT_XWIDE = rqt(2560, 40, 2, 1, 16384)
# text-to-image widths with real conditioning lengths: cc3m shape (E=1280, 20 heads, 32 text tokens -> body context 95,
# the 16-block DYN attention kernel) and the 64-token variant of the 3.9B text model (context 127, 32-block kernel),
# 3 body + 2 head layers each (configs/cc3m/stage2/*.yaml; BASELINE configs[4])
RQT_TXT32 = rqt(1280, 20, 3, 2, 16384, vocab_cond=16384, block_cond=32)
RQT_TXT64 = rqt(1280, 20, 3, 2, 16384, vocab_cond=16384, block_cond=64)
# BASELINE configs[4] at FULL depth: the 3.9B text-to-image model = the 3.8B dims (E 2560 / 40 heads / 42 + 6 layers) with 64 BPE
# tokens of a 16384-word vocabulary (README.md:60, notebooks/notebook_utils.py:32; same shape as rqvae/presets.py 'txt3900m')
RQT_TXT_3900M = rqt(2560, 40, 42, 6, 16384, vocab_cond=16384, block_cond=64)

# Variants of the stage-2 flags that no released config uses (primitives.py: TupleEmbedding / BatchLinear / LogitMask):
def _variant(base, **kw):
    c = copy.deepcopy(base)
    c.update(kw)
    return c


# learned per-depth token embeddings instead of the RQ-VAE codebook, per-depth classifiers, per-depth vocabulary sizes
RQT_TINY_TUPLE = _variant(RQT_TINY, input_emb_vqvae=False, head_emb_vqvae=False, shared_tok_emb=False, shared_cls_emb=False,
                          vocab_size=[500, 400, 300, 200])
# codebook embeddings without the depth cumsum in the head context
RQT_TINY_NOCUMSUM = _variant(RQT_TINY, cumsum_depth_ctx=False)
# codebook embeddings into the body, one shared learned embedding into the head
RQT_TINY_MIXED = _variant(RQT_TINY, head_emb_vqvae=False)

# bias-free attention / MLP layers (AttentionBlockConfig.attn_bias / mlp_bias = False, configs.py:21-40, attentions.py:48-55,117-122):
# mixed on purpose -- the body without attention biases, the head without MLP biases
RQT_TINY_NOBIAS = _variant(RQT_TINY)
RQT_TINY_NOBIAS['body']['block']['attn_bias'] = False
RQT_TINY_NOBIAS['head']['block']['mlp_bias'] = False

# the two stacks with different GELU forms (AttentionBlockConfig.gelu per stack, configs.py:21-40, attentions.py:25-36,117-122): body 'v2'
# (x * sigmoid(1.702 x)), head 'v1' (erf) -- round 6
RQT_TINY_GELUMIX = _variant(RQT_TINY)
RQT_TINY_GELUMIX['body']['block']['gelu'] = 'v2'
RQT_TINY_GELUMIX['head']['block']['gelu'] = 'v1'

# head sizes other than 64 and different head counts in the two stacks (attentions.py:44-57: any embed_dim / n_head; transformers.py:86-87:
# each stack has its own block config): body 4 heads of 32, head stack 1 head of 128 -- round 6
RQT_TINY_HEADS = _variant(RQT_TINY)
RQT_TINY_HEADS['body']['block']['n_head'] = 4
RQT_TINY_HEADS['head']['block']['n_head'] = 1
# the text-conditioned form of the same (prefix prefill through the plain attention kernel): 8 heads of 16 / 2 heads of 64
RQT_TINY_TXT_HEADS = _variant(RQT_TINY_TXT)
RQT_TINY_TXT_HEADS['body']['block']['n_head'] = 8

PARAM_COUNTS_M = {  # BASELINE.md §2 / reference README.md:38-47
    'RQT_FFHQ_355M': 355.4, 'RQT_IN_480M': 480.9, 'RQT_IN_821M': 820.9,
    'RQT_IN_1400M': 1387.5, 'RQT_IN_3800M': 3822.5, 'RQT_CC3M_654M': 654.1,
}
