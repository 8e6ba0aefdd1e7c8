"""Deterministic synthetic weights keyed by the reference's state_dict names.

TEST INFRASTRUCTURE (see oracle/__init__.py).  No pretrained checkpoints are
available offline (reference README.md:38-47 are download links), so parity is
always on seeded random weights.  Every tensor is drawn from its own
``numpy.random.default_rng([seed, crc32(name)])`` stream, so the value of a
tensor depends only on (seed, name, shape) -- the golden generator, the oracle
and the HIP path regenerate identical weights without shipping them.

The name/shape tables restate the module structure of
  rqvae/models/rqvae/rqvae.py:27-72, modules.py:10-202, layers.py:60-182,
  quantizations.py:24-41,199-214 (RQ-VAE) and
  rqvae/models/rqtransformer/transformers.py:36-107, attentions.py:39-130
(RQ-Transformer); tests/test_oracle_golden.py checks them against the key/shape
list dumped from the reference modules themselves.
"""
import zlib
from collections import OrderedDict

import numpy as np


# --------------------------------------------------------------------------- shapes
def _conv(d, name, cin, cout, k):
    d[name + '.weight'] = (cout, cin, k, k)
    d[name + '.bias'] = (cout,)


def _norm(d, name, c):
    d[name + '.weight'] = (c,)
    d[name + '.bias'] = (c,)


def _resblock(d, name, cin, cout):
    _norm(d, name + '.norm1', cin)
    _conv(d, name + '.conv1', cin, cout, 3)
    _norm(d, name + '.norm2', cout)
    _conv(d, name + '.conv2', cout, cout, 3)
    if cin != cout:
        _conv(d, name + '.nin_shortcut', cin, cout, 1)


def _attnblock(d, name, c):
    _norm(d, name + '.norm', c)
    for n in ('q', 'k', 'v', 'proj_out'):
        _conv(d, name + '.' + n, c, c, 1)


def encoder_param_shapes(dd, prefix='encoder'):
    """modules.py:10-71"""
    d = OrderedDict()
    ch, ch_mult, nrb = dd['ch'], list(dd['ch_mult']), dd['num_res_blocks']
    _conv(d, prefix + '.conv_in', dd['in_channels'], ch, 3)
    curr_res = dd['resolution']
    in_ch_mult = [1] + ch_mult
    block_in = ch
    for i_level in range(len(ch_mult)):
        block_in = ch * in_ch_mult[i_level]
        block_out = ch * ch_mult[i_level]
        for i_block in range(nrb):
            _resblock(d, f'{prefix}.down.{i_level}.block.{i_block}', block_in, block_out)
            block_in = block_out
            if curr_res in dd['attn_resolutions']:
                _attnblock(d, f'{prefix}.down.{i_level}.attn.{i_block}', block_in)
        if i_level != len(ch_mult) - 1:
            if dd.get('resamp_with_conv', True):                    # (layers.py:38-48: the conv exists only then)
                _conv(d, f'{prefix}.down.{i_level}.downsample.conv', block_in, block_in, 3)
            curr_res //= 2
    _resblock(d, prefix + '.mid.block_1', block_in, block_in)
    _attnblock(d, prefix + '.mid.attn_1', block_in)
    _resblock(d, prefix + '.mid.block_2', block_in, block_in)
    _norm(d, prefix + '.norm_out', block_in)
    zc = dd['z_channels'] * (2 if dd.get('double_z', True) else 1)
    _conv(d, prefix + '.conv_out', block_in, zc, 3)
    return d


def decoder_param_shapes(dd, prefix='decoder'):
    """modules.py:101-169 (note: ``up`` is built high-res-last but *named* by level)."""
    d = OrderedDict()
    ch, ch_mult, nrb = dd['ch'], list(dd['ch_mult']), dd['num_res_blocks']
    nres = len(ch_mult)
    block_in = ch * ch_mult[nres - 1]
    curr_res = dd['resolution'] // 2 ** (nres - 1)
    _conv(d, prefix + '.conv_in', dd['z_channels'], block_in, 3)
    _resblock(d, prefix + '.mid.block_1', block_in, block_in)
    _attnblock(d, prefix + '.mid.attn_1', block_in)
    _resblock(d, prefix + '.mid.block_2', block_in, block_in)
    for i_level in reversed(range(nres)):
        block_out = ch * ch_mult[i_level]
        for i_block in range(nrb + 1):
            _resblock(d, f'{prefix}.up.{i_level}.block.{i_block}', block_in, block_out)
            block_in = block_out
            if curr_res in dd['attn_resolutions']:
                _attnblock(d, f'{prefix}.up.{i_level}.attn.{i_block}', block_in)
        if i_level != 0:
            if dd.get('resamp_with_conv', True):
                _conv(d, f'{prefix}.up.{i_level}.upsample.conv', block_in, block_in, 3)
            curr_res *= 2
    _norm(d, prefix + '.norm_out', block_in)
    _conv(d, prefix + '.conv_out', block_in, dd['out_ch'], 3)
    return d


def rqvae_param_shapes(hps, dd):
    """state_dict key -> shape for RQVAE (rqvae.py:27-72, quantizations.py:24-41)."""
    d = OrderedDict()
    d.update(encoder_param_shapes(dd))
    d.update(decoder_param_shapes(dd))
    depth = hps['code_shape'][-1]
    n_embed, dim = hps['n_embed'], hps['embed_dim']
    for i in range(depth):
        ne = n_embed if isinstance(n_embed, int) else n_embed[i]
        d[f'quantizer.codebooks.{i}.weight'] = (ne + 1, dim)
        d[f'quantizer.codebooks.{i}.cluster_size_ema'] = (ne,)
        d[f'quantizer.codebooks.{i}.embed_ema'] = (ne, dim)
    _conv(d, 'quant_conv', dd['z_channels'], dim, 1)
    _conv(d, 'post_quant_conv', dim, dd['z_channels'], 1)
    return d


def rqt_param_shapes(cfg):
    """state_dict key -> shape for RQTransformer with the released flag set
    (shared_tok_emb/shared_cls_emb/input_emb_vqvae/head_emb_vqvae all true);
    transformers.py:36-107, attentions.py:39-130."""
    d = OrderedDict()
    E = cfg['embed_dim']
    H, W, D = cfg['block_size']
    V = cfg['vocab_size']
    vc = max(cfg.get('vocab_size_cond', 0), 1)
    bc = max(cfg.get('block_size_cond', 0), 1)
    d['pos_emb_cond'] = (1, bc, E)
    d['pos_emb_hw'] = (1, H * W, E)
    d['pos_emb_d'] = (1, D, E)
    d['cond_emb.weight'] = (vc, E)
    vs = [V] * D if isinstance(V, int) else list(V)
    in_vq, head_vq = cfg.get('input_emb_vqvae', True), cfg.get('head_emb_vqvae', True)
    if in_vq:
        d['input_mlp.weight'] = (E, cfg['input_embed_dim'])
        d['input_mlp.bias'] = (E,)
    if head_vq:
        d['head_mlp.weight'] = (E, cfg['input_embed_dim'])
        d['head_mlp.bias'] = (E,)
    if not (in_vq and head_vq):                      # transformers.py:66-70
        if cfg.get('shared_tok_emb', True):
            d['tok_emb.weight'] = (vs[0], E)
        else:
            d['tok_emb.weight'] = (sum(vs), E)
            d['tok_emb.offsets'] = (D,)              # registered buffer of TupleEmbedding (primitives.py:60-61)
    for stack, key in (('body_transformer', 'body'), ('head_transformer', 'head')):
        nl = cfg[key]['n_layer']
        attn_bias = cfg[key].get('block', {}).get('attn_bias', True)      # nn.Linear(..., bias=config.attn_bias), attentions.py:48-55
        mlp_bias = cfg[key].get('block', {}).get('mlp_bias', True)        # attentions.py:117-122
        for i in range(nl):
            p = f'{stack}.blocks.{i}'
            _norm(d, p + '.ln1', E)
            _norm(d, p + '.ln2', E)
            for n in ('key', 'query', 'value', 'proj'):
                d[f'{p}.attn.{n}.weight'] = (E, E)
                if attn_bias:
                    d[f'{p}.attn.{n}.bias'] = (E,)
            d[f'{p}.mlp.0.weight'] = (4 * E, E)
            if mlp_bias:
                d[f'{p}.mlp.0.bias'] = (4 * E,)
            d[f'{p}.mlp.2.weight'] = (E, 4 * E)
            if mlp_bias:
                d[f'{p}.mlp.2.bias'] = (E,)
    _norm(d, 'classifier.layer_norm', E)
    if cfg.get('shared_cls_emb', True):
        d['classifier.linear.weight'] = (vs[0], E)
        d['classifier.linear.bias'] = (vs[0],)
    else:                                            # BatchLinear (primitives.py:96-125): (n_vectors, in, out)
        d['classifier.linear.weight'] = (D, E, max(vs))
        d['classifier.linear.bias'] = (D, max(vs))
    if cfg.get('block_size_cond', 0) > 1:
        _norm(d, 'cond_classifier.layer_norm', E)
        d['cond_classifier.linear.weight'] = (vc, E)
        d['cond_classifier.linear.bias'] = (vc,)
    return d


# --------------------------------------------------------------------------- values
def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def make_tensor(name, shape, seed):
    """One synthetic tensor.  Scales follow torch's default inits closely enough
    that activations stay O(1) through ~50 layers."""
    rng = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    if 'codebooks' in name:
        # shared codebook => every depth aliases codebooks.0 (quantizations.py:199-205)
        base = name.split('codebooks.')[0] + 'codebooks.0.'
        r0 = _rng(seed, base + 'weight')
        if leaf == 'weight':
            w = r0.standard_normal((shape[0] - 1, shape[1]), dtype=np.float32)
            return np.concatenate([w, np.zeros((1, shape[1]), np.float32)], 0)  # padding row (quantizations.py:28)
        if leaf == 'embed_ema':
            return r0.standard_normal(shape, dtype=np.float32)
        return np.zeros(shape, np.float32)                                     # cluster_size_ema
    if name == 'tok_emb.offsets':
        raise KeyError('tok_emb.offsets is derived from the vocabulary sizes, not drawn')
    if name == 'tok_emb.weight':
        return (0.5 * rng.standard_normal(shape)).astype(np.float32)
    if name.startswith('pos_emb'):
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    if name == 'cond_emb.weight':
        return rng.standard_normal(shape, dtype=np.float32)
    is_norm = ('norm' in name) or ('.ln1.' in name) or ('.ln2.' in name)
    if leaf == 'weight' and len(shape) == 1 or (is_norm and leaf == 'weight'):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) != 3 else shape[1]      # BatchLinear weight is (n_vectors, in, out)
    b = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-b, b, size=shape).astype(np.float32)


def make_params(shapes, seed, cfg=None):
    out = OrderedDict()
    for k, s in shapes.items():
        if k == 'tok_emb.offsets':
            vs = cfg['vocab_size']
            out[k] = np.cumsum([0] + list(vs)[:-1]).astype(np.int64)
        else:
            out[k] = make_tensor(k, s, seed)
    return out
