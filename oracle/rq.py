"""Residual-quantiser oracle (numpy).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates rqvae/models/rqvae/quantizations.py of the reference:
  compute_distances        :43-62   (||x||^2 + ||c||^2 - 2 x.c^T, fp32, expanded form)
  find_nearest_embedding   :64-69   (argmin, first minimum on ties)
  VQEmbedding.forward/embed:131-146 (gather; padding row K excluded from the search, :45)
  RQBottleneck.quantize    :237-271 (depth loop on the residual, cumulative quants)
  embed_code               :297-311 (cat over depth then .sum(-2))
  embed_code_with_depth    :313-334 (no depth reduction)
"""
import numpy as np


def compute_distances(x, codebook):
    """quantizations.py:43-62.  x (..., D) fp32, codebook (K, D) fp32 (padding row
    already dropped) -> (..., K) fp32.  Same expanded form and fp32 arithmetic as
    torch.addmm(beta=1, alpha=-2); summation order inside the GEMM is BLAS-defined
    on both sides, hence the 'unambiguous margin' notion in rq_quantize_margins."""
    x = np.asarray(x, np.float32)
    cb = np.asarray(codebook, np.float32)
    flat = x.reshape(-1, x.shape[-1])
    xn = (flat * flat).sum(1, keepdims=True, dtype=np.float32)
    cn = (cb * cb).sum(1, dtype=np.float32)[None, :]
    d = (xn + cn) + np.float32(-2.0) * (flat @ cb.T)
    return d.reshape(*x.shape[:-1], cb.shape[0]).astype(np.float32)


def rq_quantize(x, codebooks):
    """quantizations.py:237-271.  x (B,h,w,D) fp32; codebooks = list (len depth) of
    (K_i, D) fp32 arrays (same array repeated when shared_codebook).
    Returns (quant_list: depth x (B,h,w,D) cumulative fp32, codes (B,h,w,depth) int64)."""
    x = np.asarray(x, np.float32)
    residual = x.copy()
    agg = np.zeros_like(x)
    quant_list, code_list = [], []
    for cb in codebooks:
        cb = np.asarray(cb, np.float32)
        d = compute_distances(residual, cb)
        code = d.argmin(-1)                      # first minimum, like torch.argmin
        quant = cb[code]
        residual = residual - quant              # residual_feature.sub_(quant)
        agg = agg + quant                        # aggregated_quants.add_(quant)
        quant_list.append(agg.copy())
        code_list.append(code[..., None])
    return quant_list, np.concatenate(code_list, -1).astype(np.int64)


def rq_quantize_margins(x, codebooks):
    """Exact (fp64, direct ||r-c||^2) top-2 gap per vector and depth, following the
    oracle's own code path.  A vector/depth is 'unambiguous' when gap > tau
    (tau = 1e-3 at D=256, N(0,1) data: SURVEY.md §8c); bit-exact code parity is
    claimed on the prefix of depths up to the first ambiguous one."""
    x64 = np.asarray(x, np.float64)
    residual = x64.reshape(-1, x64.shape[-1]).copy()
    gaps, codes = [], []
    for cb in codebooks:
        cb64 = np.asarray(cb, np.float64)
        d = ((residual ** 2).sum(1)[:, None] + (cb64 ** 2).sum(1)[None, :]
             - 2.0 * residual @ cb64.T)
        part = np.partition(d, 1, axis=1)
        gaps.append(part[:, 1] - part[:, 0])
        code = d.argmin(1)
        codes.append(code)
        residual = residual - cb64[code]
    shp = x64.shape[:-1]
    return (np.stack(gaps, -1).reshape(*shp, -1),
            np.stack(codes, -1).reshape(*shp, -1).astype(np.int64))


def rq_embed_code(codes, codebooks):
    """quantizations.py:297-311: sum_d codebook_d[code[..., d]] (cat then sum(-2), fp32,
    summed in depth order)."""
    codes = np.asarray(codes)
    out = None
    for i, cb in enumerate(codebooks):
        e = np.asarray(cb, np.float32)[codes[..., i]]
        out = e.copy() if out is None else out + e
    return out


def rq_embed_code_with_depth(codes, codebooks):
    """quantizations.py:313-334: (..., depth, D), no reduction."""
    codes = np.asarray(codes)
    return np.stack([np.asarray(cb, np.float32)[codes[..., i]] for i, cb in enumerate(codebooks)], -2)


def rq_soft_codes(x, codebooks, temp=1.0):
    """RQBottleneck.get_soft_codes, stochastic=False (quantizations.py:371-400): per depth softmax(-distances / temp) over
    the codebook and the argmin code; x (..., D) -> soft (..., depth, K) fp32, codes (..., depth) int64."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    r = x.reshape(-1, x.shape[-1]).copy()
    softs, codes = [], []
    for cb in codebooks:
        d = compute_distances(r, cb)
        z = -d / np.float32(temp)
        z = z - z.max(-1, keepdims=True)
        e = np.exp(z)
        softs.append((e / e.sum(-1, keepdims=True)).astype(np.float32))
        k = d.argmin(-1)
        codes.append(k)
        r = r - cb[k]
    soft = np.stack(softs, 1).reshape(*lead, len(codebooks), -1)
    return soft, np.stack(codes, 1).reshape(*lead, len(codebooks)).astype(np.int64)


def vq_ema_step(weight, cluster_size_ema, embed_ema, vectors, decay=0.99, eps=1e-5, restart_vectors=None):
    """One train-mode VQEmbedding.forward (quantizations.py:131-142) on `vectors` (N, D): nearest codes with the CURRENT
    weights (:64-69), _update_buffers (:80-118: per-code counts and vector sums, EMA of both, dead-code restart from
    `restart_vectors` (n_embed, D) -- the reference draws them with torch.rand_like / torch.randperm, :70-77,109-111 -- or
    None for restart_unused_codes=False), embeds from the weights BEFORE the update (:138), _update_embedding (:120-129).
    weight (K, D) without the padding row.  Returns (embeds, codes, new weight, new cluster_size_ema, new embed_ema)."""
    w = np.asarray(weight, np.float32)
    K, D = w.shape
    v = np.asarray(vectors, np.float32).reshape(-1, D)
    codes = compute_distances(v, w).argmin(-1)
    count = np.bincount(codes, minlength=K).astype(np.float32)
    vsum = np.zeros((K, D), np.float32)
    np.add.at(vsum, codes, v)
    cs = (np.asarray(cluster_size_ema, np.float32) * np.float32(decay) + np.float32(1 - decay) * count).astype(np.float32)
    ee = (np.asarray(embed_ema, np.float32) * np.float32(decay) + np.float32(1 - decay) * vsum).astype(np.float32)
    if restart_vectors is not None:
        usage = (cs >= 1).astype(np.float32)
        ee = ee * usage[:, None] + np.asarray(restart_vectors, np.float32) * (1 - usage[:, None])
        cs = cs * usage + (1 - usage)
    embeds = w[codes]
    n = cs.sum(dtype=np.float32)
    norm = n * (cs + np.float32(eps)) / (n + np.float32(K * eps))
    return embeds, codes.astype(np.int64), (ee / norm[:, None]).astype(np.float32), cs.astype(np.float32), ee.astype(np.float32)


def rq_quantize_train(x, weight, cluster_size_ema, embed_ema, depth, decay=0.99, eps=1e-5, restart_vectors=None):
    """RQBottleneck.quantize in train mode with ONE shared EMA codebook (quantizations.py:199-205,237-271): the depth loop calls
    the same VQEmbedding `depth` times, each call updating it from that depth's residuals before the next depth searches it.
    restart_vectors: list (len depth) of (K, D) arrays or None.  Returns (quant_list, codes, weight, cluster_size_ema, embed_ema)."""
    x = np.asarray(x, np.float32)
    D = x.shape[-1]
    residual = x.reshape(-1, D).copy()
    agg = np.zeros_like(residual)
    w, cs, ee = np.asarray(weight, np.float32), cluster_size_ema, embed_ema
    quant_list, code_list = [], []
    for d in range(depth):
        rv = None if restart_vectors is None else restart_vectors[d]
        quant, code, w, cs, ee = vq_ema_step(w, cs, ee, residual, decay, eps, rv)
        residual = residual - quant
        agg = agg + quant
        quant_list.append(agg.reshape(x.shape).copy())
        code_list.append(code.reshape(x.shape[:-1])[..., None])
    return quant_list, np.concatenate(code_list, -1), w, cs, ee


def ema_restart_candidates(vectors, n_embed, prng):
    """numpy twin of VQEmbedding._tile_with_noise + the randperm pick (quantizations.py:70-77,107-111) on a shared fake random
    stream `prng` (np.random.Generator): the fixture generator and the tests patch torch.rand_like / torch.randperm to draw
    from the same stream, so that the reference, the oracle and the product on any device see the same restart vectors."""
    v = np.asarray(vectors, np.float32)
    if v.shape[0] < n_embed:
        reps = (n_embed + v.shape[0] - 1) // v.shape[0]
        std = np.ones(v.shape[1], np.float32) * np.float32(0.01 / np.sqrt(v.shape[1]))
        v = np.tile(v, (reps, 1))
        v = v + prng.random(v.shape, dtype=np.float32) * std
    return v[prng.permutation(v.shape[0])][:n_embed]
