"""ctypes binding of librqamd.so (include/rqamd.h) -- the only bridge between the Python mirror of the
reference's ``rqvae.models`` API and the gfx950 kernels.

There is no CPU path: if the library (or a GPU tensor) is missing this module raises.  PyTorch is used
for device memory and streams only; every argument crossing the ABI is a raw pointer / size."""
import contextlib
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, 'librqamd.so')
# the two engines compiled once more with IEEE fp16 as their 16-bit storage type (csrc/rq_hip.h, -DRQ_F16=1; build.py): what
# RQTransformer.sample(amp=True) / forward(amp=True) and the opt-in RQAMD_VAE=fp16 RQ-VAE run on.  Same entry points, same ABI version;
# RqtEngine(half=True) / VaeEngine(half=True) bind to it.
LIB16_PATH = os.path.join(_PKG, 'librqamd_f16.so')

_lib = None
_lib16 = None
ABI_VERSION = 7


class RqamdError(RuntimeError):
    status = None


class RqamdOutOfMemory(RqamdError):
    """RQAMD_ERR_NOMEM: an engine workspace did not fit the free device memory"""
    status = -5


class VaeConfig(C.Structure):
    _fields_ = [('ch', C.c_int), ('out_ch', C.c_int), ('in_channels', C.c_int), ('resolution', C.c_int),
                ('z_channels', C.c_int), ('num_res_blocks', C.c_int), ('n_levels', C.c_int), ('ch_mult', C.c_int * 8),
                ('n_attn_res', C.c_int), ('attn_resolutions', C.c_int * 8), ('embed_dim', C.c_int), ('double_z', C.c_int)]


class RqtConfig(C.Structure):
    _fields_ = [('embed_dim', C.c_int), ('n_head', C.c_int), ('n_layer_body', C.c_int), ('n_layer_head', C.c_int),
                ('vocab_size', C.c_int), ('input_embed_dim', C.c_int), ('vocab_size_cond', C.c_int),
                ('block_size_cond', C.c_int), ('H', C.c_int), ('W', C.c_int), ('D', C.c_int), ('gelu_v2', C.c_int),
                ('input_emb_vqvae', C.c_int), ('head_emb_vqvae', C.c_int), ('shared_tok_emb', C.c_int), ('shared_cls_emb', C.c_int),
                ('cumsum_depth_ctx', C.c_int), ('vocab_sizes', C.c_int * 8)]


_SIGS = {
    'rqamd_abi_version': (C.c_int, []),
    'rqamd_last_error': (C.c_char_p, []),
    'rqamd_rq_quantize': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int64,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'rqamd_rq_code_norms': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_rq_soft_codes': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int64,
                                      C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p]),
    'rqamd_rq_distances': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'rqamd_rq_ema_accumulate': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'rqamd_rq_ema_update': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    'rqamd_rq_ema_normalize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'rqamd_rq_embed': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int64, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_sample_logits': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'rqamd_vae_create': (C.c_int, [C.POINTER(VaeConfig), C.POINTER(C.c_void_p)]),
    'rqamd_vae_destroy': (C.c_int, [C.c_void_p]),
    'rqamd_vae_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    'rqamd_vae_set_param': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    'rqamd_vae_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_vae_encode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_rqt_create': (C.c_int, [C.POINTER(RqtConfig), C.POINTER(C.c_void_p)]),
    'rqamd_rqt_destroy': (C.c_int, [C.c_void_p]),
    'rqamd_rqt_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    'rqamd_rqt_set_param': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    'rqamd_rqt_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                   C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    'rqamd_rqt_logits': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'rqamd_rqt_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]),
    'rqamd_rqt_step_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    'rqamd_rqt_step_logits': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    'rqamd_rqt_step_set_code': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_rqt_step_end': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'rqamd_rqt_set_profile': (C.c_int, [C.c_void_p, C.c_int]),
    'rqamd_rqt_get_profile': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double)]),
    'rqamd_rqt_get_profile_attn': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'rqamd_dbg_gemm_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'rqamd_dbg_conv_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'rqamd_dbg_conv_halo_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'rqamd_dbg_set_row_scale': (C.c_int, [C.c_int]),
    'rqamd_dbg_rqt_share_params': (C.c_int, [C.c_void_p, C.c_void_p]),
    'rqamd_dbg_mfma_rate': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    'rqamd_dbg_conv_in_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_dbg_ups_subpixel_weights': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'rqamd_dbg_conv_out_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)


EXPORTS_F16 = tuple(n for n in _SIGS if n.startswith('rqamd_rqt_') or n.startswith('rqamd_vae_')) + ('rqamd_abi_version', 'rqamd_last_error', 'rqamd_dbg_set_row_scale')


def _bind(path, names=None):
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        if names is not None and name not in names:
            continue
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.rqamd_abi_version() != ABI_VERSION:
        raise RqamdError(f'{path}: ABI version {lib.rqamd_abi_version()} != {ABI_VERSION} (rebuild: python rq-vae-transformer_amd/build.py)')
    return lib


def lib():
    """The loaded librqamd.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RqamdError(f'{LIB_PATH} not found: build it with `python rq-vae-transformer_amd/build.py` '
                             '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
        _lib = _bind(LIB_PATH)
    return _lib


def lib16():
    """librqamd_f16.so (the fp16 build of the RQ-Transformer engine); raises when it has not been built."""
    global _lib16
    if _lib16 is None:
        if not os.path.exists(LIB16_PATH):
            raise RqamdError(f'{LIB16_PATH} not found: build it with `python rq-vae-transformer_amd/build.py` '
                             '(hipcc --offload-arch=gfx950 -DRQ_F16=1). There is no CPU fallback.')
        _lib16 = _bind(LIB16_PATH, EXPORTS_F16)
    return _lib16


def kernel_source_hash(files=('gemm.h', 'gemm.hip')):
    """sha256[:16] over kernel sources under csrc/: stamps measurements that were taken out of run (PMC traffic files under
    profiles/) with the kernels they were taken on, so that bench.py can refuse a stale one."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(_PKG, 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(status, L=None):
    if status != 0:
        msg = (L or lib()).rqamd_last_error().decode(errors='replace')
        if status == -1:
            raise ValueError(msg)
        if status == -2:
            raise NotImplementedError(msg)
        if status == -5:
            raise RqamdOutOfMemory(f'librqamd: out of device memory: {msg}')
        raise RqamdError(f'librqamd status {status}: {msg}')


def ptr(t, dtype=None):
    """Raw device pointer of a contiguous tensor (validated)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RqamdError('librqamd needs CUDA(HIP) tensors on an MI355X; got a CPU tensor (no CPU fallback exists)')
    if not t.is_contiguous():
        raise ValueError('non-contiguous tensor passed to librqamd')
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f'expected {dtype}, got {t.dtype}')
    return C.c_void_p(t.data_ptr())


def stream_of(t):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def on_device_of(t):
    """Context that makes t's device the current HIP device for the duration of a native call: the library allocates
    (engine workspaces, graphs) and sets kernel attributes on the CURRENT device, while the stream passed in belongs to
    the tensor's device -- model.to('cuda:1') with current device 0 must not mix the two."""
    dev = t if isinstance(t, torch.device) else t.device
    return torch.cuda.device(dev) if dev.type == 'cuda' else contextlib.nullcontext()


def _view_f32(address, shape, device):
    """A float32 tensor over `address` (device memory owned by the library; no copy, no ownership)."""
    n = 1
    for s_ in shape:
        n *= int(s_)

    class _Mem:
        __cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (int(address), False), 'version': 2}
    with on_device_of(device):
        return torch.as_tensor(_Mem(), device=device).view(*shape)


def _ptr_array(tensors, dtype=torch.float32):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t, dtype).value
    return arr


def _int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


# ---------------------------------------------------------------------------------------------- free functions
def rq_code_norms(codebook):
    """||c||^2 per code of a (K, dim) fp32 codebook (the codebook term of compute_distances)."""
    K, dim = codebook.shape
    out = torch.empty((K,), dtype=torch.float32, device=codebook.device)
    with on_device_of(codebook):
        check(lib().rqamd_rq_code_norms(ptr(codebook, torch.float32), K, dim, ptr(out), stream_of(codebook)))
    return out


def rq_quantize(x, codebooks, want_quants=True, norms=None):
    """x (n_vec, dim) fp32; codebooks: list of (K_i, dim) fp32 (padding row excluded); norms: list of rq_code_norms(cb)
    (computed here when not given -- callers that quantise repeatedly cache them per codebook version).
    -> codes (n_vec, depth) int64, quant_cum (depth, n_vec, dim) fp32 or None."""
    n_vec, dim = x.shape
    depth = len(codebooks)
    for cb in codebooks:
        if cb.dim() != 2 or cb.shape[1] != dim:
            raise ValueError(f'codebook of shape {tuple(cb.shape)} does not match vectors of dim {dim}')
    if norms is None:
        norms, seen = [], {}
        for cb in codebooks:
            key = (cb.data_ptr(), cb.shape[0])
            if key not in seen:
                seen[key] = rq_code_norms(cb)
            norms.append(seen[key])
    codes = torch.empty((n_vec, depth), dtype=torch.int64, device=x.device)
    quants = torch.empty((depth, n_vec, dim), dtype=torch.float32, device=x.device) if want_quants else None
    ws = None
    if 0 < n_vec < 96 * 64:          # small inputs: scratch for the codebook-split path (residual + per-split partial minima)
        ws = torch.empty((n_vec * dim * 4 + n_vec * 64 * 8,), dtype=torch.uint8, device=x.device)
    with on_device_of(x):
        check(lib().rqamd_rq_quantize(ptr(x, torch.float32), _ptr_array(codebooks), _ptr_array(norms),
                                      _int_array([c.shape[0] for c in codebooks]), depth, n_vec, dim, ptr(codes), ptr(quants),
                                      ptr(ws), 0 if ws is None else ws.numel(), stream_of(x)))
    return codes, quants


def rq_soft_codes(x, codebooks, norms, temp=1.0, stochastic=False, seed=0, offset=0):
    """x (n_vec, dim) fp32 -> (soft codes (n_vec, depth, K) fp32, codes (n_vec, depth) int64): RQBottleneck.get_soft_codes."""
    n_vec, dim = x.shape
    depth, K = len(codebooks), codebooks[0].shape[0]
    soft = torch.empty((n_vec, depth, K), dtype=torch.float32, device=x.device)
    codes = torch.empty((n_vec, depth), dtype=torch.int64, device=x.device)
    ws = torch.empty((max(n_vec, 1) * (dim * 4 + 512 + K * 4),), dtype=torch.uint8, device=x.device)
    with on_device_of(x):
        check(lib().rqamd_rq_soft_codes(ptr(x, torch.float32), _ptr_array(codebooks), _ptr_array(norms),
                                        _int_array([c.shape[0] for c in codebooks]), depth, n_vec, dim, float(temp), int(bool(stochastic)),
                                        int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), ptr(soft), ptr(codes), ptr(ws), ws.numel(),
                                        stream_of(x)))
    return soft, codes


def rq_distances(x, codebook, norms):
    """x (n_vec, dim) fp32, codebook (K, dim) fp32 -> (n_vec, K) fp32 squared distances ||x||^2 + ||c||^2 - 2 x.c:
    VQEmbedding.compute_distances (quantizations.py:43-62), the values the quantiser's argmin is taken over."""
    n_vec, dim = x.shape
    K = codebook.shape[0]
    out = torch.empty((n_vec, K), dtype=torch.float32, device=x.device)
    ws = torch.empty((max(n_vec, 1) * 512,), dtype=torch.uint8, device=x.device)
    with on_device_of(x):
        check(lib().rqamd_rq_distances(ptr(x, torch.float32), ptr(codebook, torch.float32), ptr(norms, torch.float32), K, n_vec, dim,
                                       ptr(out), ptr(ws), ws.numel(), stream_of(x)))
    return out


def rq_ema_accumulate(x, idx, n_embed):
    """x (n_vec, dim) fp32, idx (n_vec) int64 -> (count (n_embed,), sum (n_embed, dim)) fp32: VQEmbedding._update_buffers' one-hot sums."""
    n_vec, dim = x.shape
    count = torch.empty((n_embed,), dtype=torch.float32, device=x.device)
    vsum = torch.empty((n_embed, dim), dtype=torch.float32, device=x.device)
    with on_device_of(x):
        check(lib().rqamd_rq_ema_accumulate(ptr(x, torch.float32), ptr(idx, torch.int64), n_vec, dim, n_embed, ptr(count), ptr(vsum), stream_of(x)))
    return count, vsum


def rq_ema_update(cluster_size_ema, embed_ema, count, vsum, restart_vectors, decay):
    """in place: the EMA step of the codebook statistics + (restart_vectors given) the dead-code restart."""
    n_embed, dim = embed_ema.shape
    with on_device_of(embed_ema):
        check(lib().rqamd_rq_ema_update(ptr(cluster_size_ema, torch.float32), ptr(embed_ema, torch.float32), ptr(count, torch.float32),
                                        ptr(vsum, torch.float32), ptr(restart_vectors, torch.float32), n_embed, dim, float(decay),
                                        stream_of(embed_ema)))


def rq_ema_normalize(cluster_size_ema, embed_ema, n_total, eps, weight_out):
    """weight_out (n_embed, dim) <- embed_ema / normalised cluster size; n_total: device scalar = cluster_size_ema.sum()."""
    n_embed, dim = embed_ema.shape
    with on_device_of(embed_ema):
        check(lib().rqamd_rq_ema_normalize(ptr(cluster_size_ema, torch.float32), ptr(embed_ema, torch.float32), ptr(n_total, torch.float32),
                                           n_embed, dim, float(eps), ptr(weight_out, torch.float32), stream_of(embed_ema)))


def rq_embed(codes, codebooks, mode):
    """codes (n_vec, depth) int64 -> mode 0: (n_vec, dim); 1: (n_vec, depth, dim); 2: depth-cumsum."""
    n_vec, depth = codes.shape
    dim = codebooks[0].shape[1]
    shape = (n_vec, dim) if mode == 0 else (n_vec, depth, dim)
    out = torch.empty(shape, dtype=torch.float32, device=codes.device)
    with on_device_of(codes):
        check(lib().rqamd_rq_embed(ptr(codes, torch.int64), _ptr_array(codebooks), _int_array([c.shape[0] for c in codebooks]),
                                   depth, n_vec, dim, mode, ptr(out), stream_of(codes)))
    return out


def sample_logits(logits, temperature=1.0, top_k=None, top_p=None, seed=0, offset=0, want_probs=False, want_samples=True):
    rows, vocab = logits.shape
    samples = torch.empty((rows,), dtype=torch.int64, device=logits.device) if want_samples else None
    probs = torch.empty((rows, vocab), dtype=torch.float32, device=logits.device) if want_probs else None
    flags = torch.empty((max(rows, 1),), dtype=torch.int32, device=logits.device)
    with on_device_of(logits):
        check(lib().rqamd_sample_logits(ptr(logits, torch.float32), rows, vocab, float(temperature),
                                        0 if top_k is None else int(top_k), -1.0 if top_p is None else float(top_p),
                                        int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), ptr(samples), ptr(probs),
                                        ptr(flags), stream_of(logits)))
    return samples, probs


def dbg_gemm(a_bf16, w_bf16, bias=None, epi=3, bm=0, bn=0, splitk=0, out=None):
    """diagnostics: out[M,N] = a[M,K] @ w[N,K]^T (+bias); a, w are torch.bfloat16."""
    M, K = a_bf16.shape
    N = w_bf16.shape[0]
    if out is None:
        kind = epi % 16                      # + 16 / + 64 / + 96 / ... are launch options (include/rqamd.h)
        if kind == 4:
            out = torch.empty((splitk if splitk > 0 else 8, M, N), dtype=torch.float32, device=a_bf16.device)
        else:
            out = torch.empty((M, N), dtype=torch.float32 if kind == 3 else torch.bfloat16, device=a_bf16.device)
    check(lib().rqamd_dbg_gemm_bf16(ptr(a_bf16, torch.bfloat16), ptr(w_bf16, torch.bfloat16), M, N, K, ptr(bias), epi,
                                    ptr(out), bm, bn, splitk, stream_of(a_bf16)))
    return out


def dbg_conv(x, w, bias=None, resid=None, ksize=3, stride=1, ups=0, bm=0, bn=0, flags=0, out=None):
    """diagnostics: x (B,Hs,Ws,Cin) bf16 NHWC, w (Cout,k,k,Cin) bf16 -> (B,Ho,Wo,Cout) bf16."""
    B, Hs, Ws, Cin = x.shape
    H, W = Hs << ups, Ws << ups
    Cout = w.shape[0]
    Ho, Wo = (H // 2, W // 2) if stride == 2 else (H, W)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    check(lib().rqamd_dbg_conv_bf16(ptr(x, torch.bfloat16), ptr(w, torch.bfloat16), ptr(bias), ptr(resid), B, H, W, Cin, Cout,
                                    ksize, stride, ups, ptr(out), bm, bn, flags, stream_of(x)))
    return out


def dbg_ups_subpixel_weights(w):
    """diagnostics: (Cout,3,3,Cin) bf16 conv weights -> (4,Cout,2,2,Cin) bf16 pre-summed weights of the sub-pixel form of Upsample.conv."""
    Cout, _, _, Cin = w.shape
    wsub = torch.empty((4, Cout, 2, 2, Cin), dtype=torch.bfloat16, device=w.device)
    check(lib().rqamd_dbg_ups_subpixel_weights(ptr(w, torch.bfloat16), Cout, Cin, ptr(wsub), stream_of(w)))
    return wsub


def dbg_conv_halo(x, w, bias, gn=None, resid=None, out=None, stats=None, ups=False, persistent=None, wpx=0, subpixel=False):
    """diagnostics: halo-reuse 3x3 conv; x (B,H,W,Cin) bf16 (or (B,H/2,W/2,Cin) with ups), w (Cout,3,3,Cin) bf16, gn
    (B,Cin,2) fp32 or None; stats (B, (H/8)*(W/32), 32, 2) fp32 receives the per-tile GroupNorm partial sums.
    persistent True / False forces the persistent / per-tile form of the kernel (None: the default), wpx the persistent form's
    workgroups per XCD (0: one per CU)."""
    B, H, W, Cin = x.shape
    if ups:
        H, W = 2 * H, 2 * W
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=x.device)
    if subpixel:          # the upsample conv as four 2 x 2 convs over the source image (conv_halo.hip, UPS = 2)
        assert ups
        w = dbg_ups_subpixel_weights(w)
    check(lib().rqamd_dbg_conv_halo_bf16(ptr(x, torch.bfloat16), ptr(w, torch.bfloat16), ptr(bias, torch.float32), ptr(gn), ptr(resid),
                                         B, H, W, Cin, Cout,
                                         (1 if ups else 0) | (0 if persistent is None else 16 if persistent else 32) | ((int(wpx) & 0xff) << 8)
                                         | (64 if subpixel else 0),
                                         ptr(out), ptr(stats), stream_of(x)))
    return out


def dbg_mfma_rate(mode=1, n_per_wave=1 << 15, launches=64, secs=1.5, device='cuda'):
    """diagnostics: the dense bf16 MFMA rate (TFLOP/s) the board sustains with MFMAs alone -- mode 0 constant operands (the instruction
    rate), mode 1 operands that change every instruction (~N(0,1) bf16).  Runs for about `secs` seconds so that the clock settles under the
    power limit, and times the last batch of launches with events."""
    scratch = torch.zeros(16, dtype=torch.float32, device=device)
    flop = C.c_double(0.0)
    st = torch.cuda.current_stream(scratch.device)

    def batch():
        check(lib().rqamd_dbg_mfma_rate(int(mode), int(n_per_wave), int(launches), ptr(scratch), C.byref(flop), stream_of(scratch)))
    batch()
    torch.cuda.synchronize()
    import time
    t0, rate = time.time(), 0.0
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        batch()
        e1.record(st)
        e1.synchronize()
        rate = flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
        if time.time() - t0 >= secs:
            return rate


def dbg_set_row_scale(factor):
    """diagnostics: variant selection sees rows * factor (1 = normal)."""
    check(lib().rqamd_dbg_set_row_scale(int(factor)))


def dbg_conv_in(x, w, bias):
    """diagnostics: MFMA conv_in; x (B,3,H,W) fp32, w (3,3,3,128) fp32 = (ky,kx,ci,cout), returns (B,H,W,128) bf16."""
    B, _, H, W = x.shape
    y = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=x.device)
    check(lib().rqamd_dbg_conv_in_bf16(ptr(x, torch.float32), ptr(w, torch.float32), ptr(bias, torch.float32), B, H, W, ptr(y),
                                       stream_of(x)))
    return y


def dbg_conv_out(x, w, bias, gn=None):
    """diagnostics: MFMA conv_out; x (B,H,W,Cin) bf16, w (Cout,3,3,Cin) fp32, returns (B,Cout,H,W) fp32."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    check(lib().rqamd_dbg_conv_out_bf16(ptr(x, torch.bfloat16), ptr(w, torch.float32), ptr(bias, torch.float32), ptr(gn),
                                        B, H, W, Cin, Cout, ptr(y), stream_of(x)))
    return y


# ---------------------------------------------------------------------------------------------- engines
class _Engine:
    """Opaque native handle bound to ONE device: created, fed and run with that device current (on_device_of)."""
    _create = _destroy = _set = None

    def __init__(self, cfg_struct, device, half=False):
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self._h = C.c_void_p()
        self.half = bool(half)
        self._L = lib16() if half else lib()         # the library this handle belongs to (errors are per library, too)
        with on_device_of(self.device):
            check(getattr(self._L, self._create)(C.byref(cfg_struct), C.byref(self._h)), self._L)

    def _on_my_device(self, *tensors):
        for t in tensors:
            if t is not None and t.device != self.device:
                raise ValueError(f'tensor on {t.device}, engine on {self.device} (move the model and its inputs to one device)')

    def set_param(self, name, tensor):
        t = tensor.detach()
        self._on_my_device(t)
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        with on_device_of(self.device):
            check(getattr(self._L, self._set)(self._h, name.encode(), ptr(t, torch.float32), shape, t.dim(), stream_of(t)), self._L)
        if t.is_cuda:
            t.record_stream(torch.cuda.current_stream(t.device))

    def _run(self, fn):
        """One native call on the engine's device.  The engines allocate their workspaces with hipMalloc, outside torch's
        caching allocator: on an out-of-memory failure (RQAMD_ERR_NOMEM only -- any other error is raised as it is) the
        allocator's cached blocks are released and the call is retried once; the handle is left empty, not dangling, by a
        failed regrowth, and the library clears HIP's sticky last-error after the failed hipMalloc."""
        with on_device_of(self.device):
            try:
                return check(fn(), self._L)
            except RqamdOutOfMemory:
                if self.device.type != 'cuda':
                    raise
                torch.cuda.empty_cache()
                return check(fn(), self._L)

    def close(self):
        if self._h is not None and self._h.value and self._L is not None:
            getattr(self._L, self._destroy)(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VaeEngine(_Engine):
    _create, _destroy, _set = 'rqamd_vae_create', 'rqamd_vae_destroy', 'rqamd_vae_set_param'

    def __init__(self, ddconfig, embed_dim, device='cuda', half=False):
        c = VaeConfig()
        c.ch, c.out_ch, c.in_channels = ddconfig['ch'], ddconfig['out_ch'], ddconfig['in_channels']
        c.resolution, c.z_channels, c.num_res_blocks = ddconfig['resolution'], ddconfig['z_channels'], ddconfig['num_res_blocks']
        mult, attn = list(ddconfig['ch_mult']), list(ddconfig['attn_resolutions'])
        if len(mult) > 8 or len(attn) > 8:
            raise NotImplementedError('more than 8 resolution levels')
        c.n_levels, c.n_attn_res = len(mult), len(attn)
        for i, m in enumerate(mult):
            c.ch_mult[i] = int(m)
        for i, a in enumerate(attn):
            c.attn_resolutions[i] = int(a)
        c.embed_dim, c.double_z = int(embed_dim), int(bool(ddconfig.get('double_z', True)))
        self.cfg = c
        super().__init__(c, device, half=half)      # half: the fp16 build of the engine (opt-in RQAMD_VAE=fp16, see models/rqvae/rqvae.py)
        if not ddconfig.get('resamp_with_conv', True):      # bare nearest upsample / average pool (layers.py:20-57; no released config)
            check(self._L.rqamd_vae_set_option(self._h, b'resamp_with_conv', 0), self._L)

    def decode(self, z_q):
        """z_q (B,h,w,embed_dim) fp32 NHWC -> (B,out_ch,H,W) fp32"""
        c = self.cfg
        lr = c.resolution >> (c.n_levels - 1)
        if z_q.dim() != 4 or tuple(z_q.shape[1:]) != (lr, lr, c.embed_dim):
            # the engine is built for one resolution (its workspaces and tile schedules are sized from ddconfig.resolution)
            raise ValueError(f'decode: latent of shape {tuple(z_q.shape)}; this RQVAE decodes (B, {lr}, {lr}, {c.embed_dim})')
        self._on_my_device(z_q)
        B = z_q.shape[0]
        out = torch.empty((B, c.out_ch, c.resolution, c.resolution), dtype=torch.float32, device=z_q.device)
        self._run(lambda: self._L.rqamd_vae_decode(self._h, ptr(z_q, torch.float32), B, ptr(out), stream_of(z_q)))
        return out

    def encode(self, x):
        """x (B,in_channels,H,W) fp32 -> z_e (B,h,w,embed_dim) fp32 NHWC"""
        c = self.cfg
        if x.dim() != 4 or tuple(x.shape[1:]) != (c.in_channels, c.resolution, c.resolution):
            raise ValueError(f'encode: input of shape {tuple(x.shape)}; this RQVAE encodes (B, {c.in_channels}, {c.resolution}, {c.resolution})')
        self._on_my_device(x)
        B = x.shape[0]
        lr = c.resolution >> (c.n_levels - 1)
        out = torch.empty((B, lr, lr, c.embed_dim), dtype=torch.float32, device=x.device)
        self._run(lambda: self._L.rqamd_vae_encode(self._h, ptr(x, torch.float32), B, ptr(out), stream_of(x)))
        return out


class RqtEngine(_Engine):
    _create, _destroy, _set = 'rqamd_rqt_create', 'rqamd_rqt_destroy', 'rqamd_rqt_set_param'

    def __init__(self, *, embed_dim, n_head, n_layer_body, n_layer_head, vocab_size, input_embed_dim, vocab_size_cond,
                 block_size_cond, block_size, gelu_v2=False, device='cuda', input_emb_vqvae=True, head_emb_vqvae=True,
                 shared_tok_emb=True, shared_cls_emb=True, cumsum_depth_ctx=True, vocab_sizes=None, half=False, n_head_head=None):
        c = RqtConfig(embed_dim, n_head, n_layer_body, n_layer_head, vocab_size, input_embed_dim, vocab_size_cond,
                      block_size_cond, block_size[0], block_size[1], block_size[2], int(gelu_v2), int(bool(input_emb_vqvae)),
                      int(bool(head_emb_vqvae)), int(bool(shared_tok_emb)), int(bool(shared_cls_emb)), int(bool(cumsum_depth_ctx)))
        vs = list(vocab_sizes) if vocab_sizes is not None else [vocab_size] * block_size[2]
        for i, v in enumerate(vs[:8]):
            c.vocab_sizes[i] = int(v)
        self.cfg = c
        super().__init__(c, device, half=half)
        if n_head_head is not None and int(n_head_head) != int(n_head):       # head.block.n_head != body.block.n_head (no released config)
            check(self._L.rqamd_rqt_set_option(self._h, b'head.n_head', int(n_head_head)), self._L)

    def _check(self, codes, cond, codebooks):
        c = self.cfg
        if codes.dim() != 4 or tuple(codes.shape[1:]) != (c.H, c.W, c.D):
            raise ValueError(f'codes of shape {tuple(codes.shape)}; expected (B, {c.H}, {c.W}, {c.D})')
        if cond is not None and tuple(cond.shape) != (codes.shape[0], max(c.block_size_cond, 1)):
            raise ValueError(f'cond of shape {tuple(cond.shape)}; expected ({codes.shape[0]}, {max(c.block_size_cond, 1)})')
        if len(codebooks) < c.D:
            raise ValueError(f'{len(codebooks)} codebooks for depth {c.D}')
        if c.input_emb_vqvae or c.head_emb_vqvae:
            for d, cb in enumerate(codebooks[:c.D]):
                if cb.dim() != 2 or cb.shape[0] < c.vocab_sizes[d] or cb.shape[1] != c.input_embed_dim:
                    raise ValueError(f'codebook of shape {tuple(cb.shape)}; expected (>= {c.vocab_sizes[d]}, {c.input_embed_dim})')
        self._on_my_device(codes, cond, *codebooks[:c.D])

    def sample(self, partial, cond, codebooks, start_loc, temperature, top_k, top_p, seed, offset, use_graph):
        self._check(partial, cond, codebooks)
        B = partial.shape[0]
        out = torch.empty_like(partial)
        D = self.cfg.D
        cbs, tk, tp = _ptr_array(codebooks[:D]), _int_array(top_k[:D]), (C.c_float * D)(*[float(p) for p in top_p[:D]])
        self._run(lambda: self._L.rqamd_rqt_sample(self._h, ptr(partial, torch.int64), ptr(cond, torch.int64), B, cbs,
                                                 int(start_loc[0]), int(start_loc[1]), float(temperature), tk, tp,
                                                 int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(bool(use_graph)),
                                                 ptr(out), stream_of(partial)))
        return out

    def logits(self, codes, cond, codebooks):
        self._check(codes, cond, codebooks)
        B = codes.shape[0]
        c = self.cfg
        out = torch.empty((B, c.H, c.W, c.D, c.vocab_size), dtype=torch.float32, device=codes.device)
        cbs = _ptr_array(codebooks[:c.D])
        self._run(lambda: self._L.rqamd_rqt_logits(self._h, ptr(codes, torch.int64), ptr(cond, torch.int64), B, cbs,
                                                 ptr(out), stream_of(codes)))
        return out

    def forward(self, codes, cond, codebooks):
        """(seq_logits (B,H,W,D,V), cond_logits (B, block_size_cond-1, vocab_size_cond)) -- text-conditioned models"""
        self._check(codes, cond, codebooks)
        B = codes.shape[0]
        c = self.cfg
        out = torch.empty((B, c.H, c.W, c.D, c.vocab_size), dtype=torch.float32, device=codes.device)
        cl = torch.empty((B, c.block_size_cond - 1, max(c.vocab_size_cond, 1)), dtype=torch.float32, device=codes.device)
        cbs = _ptr_array(codebooks[:c.D])
        self._run(lambda: self._L.rqamd_rqt_forward(self._h, ptr(codes, torch.int64), ptr(cond, torch.int64), B, cbs,
                                                  ptr(out), ptr(cl), stream_of(codes)))
        return out, cl

    # ---- stepping form: the caller draws the samples (rqamd_rqt_step_*)
    def step_begin(self, partial, cond, codebooks):
        self._check(partial, cond, codebooks)
        self._step = (partial.shape[0], partial.device, [cb for cb in codebooks[:self.cfg.D]])     # keeps the codebooks alive
        cbs = _ptr_array(self._step[2])
        self._run(lambda: self._L.rqamd_rqt_step_begin(self._h, ptr(partial, torch.int64), ptr(cond, torch.int64), partial.shape[0], cbs,
                                                     stream_of(partial)))

    def step_logits(self, pos, d):
        """logits (B, V) fp32 of step (pos, d) -- a view of the engine's workspace, valid until the next engine call; d < 0:
        body stack only (returns None)."""
        B, dev, _ = self._step
        out = C.c_void_p()
        with on_device_of(dev):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream if dev.type == 'cuda' else 0)
            check(self._L.rqamd_rqt_step_logits(self._h, int(pos), int(d), C.byref(out), st), self._L)
        if d < 0:
            return None
        return _view_f32(out.value, (B, self.cfg.vocab_size), dev)

    def step_set_code(self, pos, d, codes):
        B, dev, _ = self._step
        if tuple(codes.shape) != (B,):
            raise ValueError(f'codes of shape {tuple(codes.shape)}; expected ({B},)')
        with on_device_of(dev):
            check(self._L.rqamd_rqt_step_set_code(self._h, int(pos), int(d), ptr(codes.contiguous(), torch.int64), stream_of(codes)), self._L)

    def step_end(self):
        B, dev, _ = self._step
        c = self.cfg
        out = torch.empty((B, c.H, c.W, c.D), dtype=torch.int64, device=dev)
        with on_device_of(dev):
            check(self._L.rqamd_rqt_step_end(self._h, ptr(out), stream_of(out)), self._L)
        self._step = None
        return out

    def set_profile(self, mode):
        """0 / False: off; 1 / True: HIP events around every GEMM / attention launch (eager launches); 2: skip the GEMM launches (graphs
        stay on) -- a sampling pass timed with and without them gives the GEMMs' time inside the graphs."""
        check(self._L.rqamd_rqt_set_profile(self._h, int(mode)), self._L)

    def get_profile(self):
        ms, n, by, fl = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(self._L.rqamd_rqt_get_profile(self._h, C.byref(ms), C.byref(n), C.byref(by), C.byref(fl)), self._L)
        ams, an = C.c_double(), C.c_int64()
        check(self._L.rqamd_rqt_get_profile_attn(self._h, C.byref(ams), C.byref(an)), self._L)
        return dict(gemm_ms_total=ms.value, gemm_launches=n.value, gemm_bytes=by.value, gemm_flops=fl.value,
                    attn_ms_total=ams.value, attn_launches=an.value)
