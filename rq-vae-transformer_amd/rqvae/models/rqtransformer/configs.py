"""Defaults of rqvae/models/rqtransformer/configs.py:21-73 of the reference, without omegaconf
(not installable in the target image): plain dicts merged field by field."""
import copy

ATTENTION_BLOCK_DEFAULTS = dict(embed_dim=None, n_head=None, mlp_bias=True, attn_bias=True, attn_pdrop=0.0,
                                resid_pdrop=0.1, gelu='v1')
RQTRANSFORMER_DEFAULTS = dict(
    type='rq-transformer', ema=None, ar_hierarchy=None, vocab_size=None, block_size=None, vocab_size_cond=0,
    block_size_cond=0, embed_dim=None, input_embed_dim=None, use_padding_emb=False, input_emb_vqvae=False,
    head_emb_vqvae=False, scaled_head_emb_vqvae=False, cumsum_depth_ctx=False, shared_tok_emb=False, embd_pdrop=0.0,
    body=dict(n_layer=None, block=copy.deepcopy(ATTENTION_BLOCK_DEFAULTS)),
    head=dict(n_layer=None, block=copy.deepcopy(ATTENTION_BLOCK_DEFAULTS)),
    shared_cls_emb=False)


def _get(cfg, key, default=None):
    try:
        if hasattr(cfg, 'keys'):
            return cfg[key] if key in cfg.keys() else default
        return getattr(cfg, key)
    except (KeyError, AttributeError):
        return default


def _to_dict(cfg):
    if cfg is None:
        return {}
    out = {}
    for k in cfg.keys():
        v = cfg[k]
        if hasattr(v, 'keys'):
            out[k] = _to_dict(v)
        elif isinstance(v, (str, int, float, bool, type(None))):
            out[k] = v
        else:
            out[k] = list(v)
    return out


def _merge(base, over):
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def resolve(config):
    """RQTransformerConfig.create (configs.py:68-73): defaults <- config, block embed_dim = embed_dim."""
    cfg = _merge(RQTRANSFORMER_DEFAULTS, _to_dict(config))
    for stack in ('body', 'head'):
        if cfg[stack]['block'].get('embed_dim') is None:
            cfg[stack]['block']['embed_dim'] = cfg['embed_dim']
    return cfg
