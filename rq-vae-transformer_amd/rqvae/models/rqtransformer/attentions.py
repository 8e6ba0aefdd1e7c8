"""Parameter containers mirroring rqvae/models/rqtransformer/attentions.py:39-169 of the reference
(MultiSelfAttention, AttentionBlock, AttentionStack).  Names/shapes only -- the decode-step arithmetic
(q/k/v/proj GEMMs, KV-cache attention, MLP, LayerNorm) runs in librqamd's sampling engine."""
from torch import nn


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} is a parameter container; run RQTransformer.sample / forward')


class GELU(_Holder):
    def __init__(self, version='v1'):
        super().__init__()
        assert version == 'v1' or version == 'v2'
        self.version = version


class MultiSelfAttention(_Holder):
    def __init__(self, embed_dim, n_head, attn_bias=True, attn_pdrop=0.0, resid_pdrop=0.1, mask=True):
        super().__init__()
        assert embed_dim % n_head == 0
        self.key = nn.Linear(embed_dim, embed_dim, bias=attn_bias)
        self.query = nn.Linear(embed_dim, embed_dim, bias=attn_bias)
        self.value = nn.Linear(embed_dim, embed_dim, bias=attn_bias)
        self.attn_drop = nn.Dropout(attn_pdrop, inplace=False)
        self.resid_drop = nn.Dropout(resid_pdrop, inplace=True)
        self.proj = nn.Linear(embed_dim, embed_dim, attn_bias)
        self.n_head = n_head
        self.mask = mask


class AttentionBlock(_Holder):
    def __init__(self, block_cfg):
        super().__init__()
        E = block_cfg['embed_dim']
        # attn_bias / mlp_bias = False (configs.py:21-40): the Linear layers carry no bias parameter, as in the reference; the engine is
        # handed zero vectors for them (RQTransformer._eng), which adds nothing
        attn_bias, mlp_bias = bool(block_cfg.get('attn_bias', True)), bool(block_cfg.get('mlp_bias', True))
        self.ln1 = nn.LayerNorm(E)
        self.ln2 = nn.LayerNorm(E)
        self.attn = MultiSelfAttention(E, block_cfg['n_head'], attn_bias, block_cfg.get('attn_pdrop', 0.0),
                                       block_cfg.get('resid_pdrop', 0.1), mask=True)
        self.mlp = nn.Sequential(nn.Linear(E, 4 * E, bias=mlp_bias), GELU(block_cfg.get('gelu', 'v1')),
                                 nn.Linear(4 * E, E, bias=mlp_bias), nn.Dropout(block_cfg.get('resid_pdrop', 0.1), inplace=True))
        self._cache = None


class AttentionStack(_Holder):
    def __init__(self, stack_cfg):
        super().__init__()
        self.blocks = nn.ModuleList([AttentionBlock(stack_cfg['block']) for _ in range(stack_cfg['n_layer'])])

    def init_cache(self):
        pass   # KV caches live in the engine (fixed capacity, reset at the start of every sample())
