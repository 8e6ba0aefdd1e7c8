/* rqamd.h -- C ABI of librqamd.so: the MI355X (gfx950) RQ-VAE + RQ-Transformer sampling path.
 *
 * The reference (kakaobrain/rq-vae-transformer) is pure Python on PyTorch and has no FFI or
 * operator registry; its "operator API" for this path is the Python method surface
 * of rqvae.models.  Each entry point below names the reference method(s) it stands
 * behind (file:line relative to the reference root).  The host-side mirror of those classes lives in
 * rq-vae-transformer_amd/rqvae/ and binds this ABI with ctypes.
 *
 * Conventions
 *  - plain C: device pointers, sizes, an opaque stream (hipStream_t passed as void*; NULL = stream 0).
 *  - every call is asynchronous on `stream` unless stated; the library never frees or keeps caller
 *    buffers beyond the call, except the parameter tables of the engine handles (see *_set_param).
 *  - return 0 on success, negative rqamd_status otherwise; rqamd_last_error() gives the message
 *    (thread-local).  Nothing is thrown across the ABI.
 *  - tensors are dense, row-major, in the layouts the reference methods use.
 *
 * Two builds of the RQ-Transformer engine (round 6).  librqamd.so stores weights, GEMM operands and the KV cache as bfloat16
 * (BASELINE.json's compute dtype; the default everywhere).  librqamd_f16.so is the SAME source compiled with -DRQ_F16=1
 * (csrc/rq_hip.h): IEEE fp16 storage instead, fp32 accumulation / residual stream / LayerNorm / softmax / logits as before --
 * what the reference's `amp=True` (fp16 autocast, rqvae/models/rqtransformer/transformers.py:21,206; main_sampling_fid.py:216)
 * computes in.  It exports the rqamd_rqt_* and (since the end of round 6) the rqamd_vae_* entry points below with identical signatures
 * and meaning, plus rqamd_abi_version / rqamd_last_error / rqamd_dbg_set_row_scale; the quantiser (fp32 arithmetic) exists in
 * librqamd.so only.  The fp16 RQ-VAE engine is opt-in on the Python side (RQAMD_VAE=fp16): three more mantissa bits in every stored
 * activation and weight -- z_e and pixels ~8 x closer to the fp32 reference -- for ~5 % more decode time.  Values beyond 65504 become
 * inf in fp16 storage, as in torch.float16.
 */
#ifndef RQAMD_H
#define RQAMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RQAMD_ABI_VERSION 7

typedef enum {
    RQAMD_OK = 0,
    RQAMD_ERR_INVALID = -1,      /* bad argument / shape (reference raises ValueError / AssertionError) */
    RQAMD_ERR_UNSUPPORTED = -2,  /* shape outside what the gfx950 kernels implement */
    RQAMD_ERR_HIP = -3,          /* HIP runtime error */
    RQAMD_ERR_STATE = -4,        /* handle not ready (missing parameters) */
    RQAMD_ERR_NOMEM = -5         /* hipMalloc of an engine workspace ran out of device memory; the handle is left empty (not
                                    dangling) and the call may be retried after the caller released memory */
} rqamd_status;

int rqamd_abi_version(void);
const char* rqamd_last_error(void);

/* ---- residual quantiser -------------------------------------------------------------------
 * rqamd_rq_quantize <- RQBottleneck.quantize (rqvae/models/rqvae/quantizations.py:237-271), i.e.
 * VQEmbedding.compute_distances :43-62 + find_nearest_embedding :64-69 + embed :144-146 fused over
 * all depths.  x (n_vec, dim) fp32; codebooks[d] (n_embed[d], dim) fp32 WITHOUT the padding row
 * (pass weight[:-1]); the same pointer repeated = shared codebook.  codes (n_vec, depth) int64;
 * quant_cum (depth, n_vec, dim) fp32 cumulative quants (quant_list) or NULL.
 * code_norms[d] (n_embed[d]) fp32 = ||c||^2 per code from rqamd_rq_code_norms, owned and cached by the caller per codebook
 * version (no library-side scratch: calls on different streams / threads never share state).
 * Distances use the reference's expanded form ||x||^2+||c||^2-2x.c in fp32; ties -> lowest index.
 * workspace (caller-owned device scratch, n_vec * dim * 4 + n_vec * 64 * 8 bytes, or NULL): lets small inputs (< 96 tiles
 * of 64 vectors: the per-image rFID / get_codes calls) divide the codebook over the chip, one launch pair per depth; the
 * result is bit-identical to the single-launch path. */
int rqamd_rq_quantize(const float* x, const float* const* codebooks, const float* const* code_norms, const int* n_embed,
                      int depth, int64_t n_vec, int dim, int64_t* codes, float* quant_cum, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* rqamd_rq_soft_codes <- RQBottleneck.get_soft_codes (quantizations.py:371-400): per depth, soft_out[v][d][:] =
 * softmax(-distances(residual_d, codebook) / temp) (n_vec, depth, n_embed) fp32 and codes[v][d] = argmin of the distances,
 * or -- stochastic != 0 -- one multinomial draw from that soft code (Philox keyed by seed / offset); the residual is updated
 * with the chosen code.  All codebooks must have one size.  workspace: n_vec * (dim * 4 + 512 + n_embed * 4) bytes. */
int rqamd_rq_soft_codes(const float* x, const float* const* codebooks, const float* const* code_norms, const int* n_embed,
                        int depth, int64_t n_vec, int dim, float temp, int stochastic, uint64_t seed, uint64_t offset,
                        float* soft_out, int64_t* codes, void* workspace, int64_t workspace_bytes, void* stream);
/* rqamd_rq_distances <- VQEmbedding.compute_distances (quantizations.py:43-62) on its own: dist_out[v][k] =
 * ||x_v||^2 + ||c_k||^2 - 2 x_v . c_k  (n_vec, n_embed) fp32, the values the quantiser's argmin and the soft codes are formed
 * from (same kernel, same summation order), for ONE codebook.  workspace: n_vec * 64 * 8 bytes of device scratch.  (ABI v5) */
int rqamd_rq_distances(const float* x, const float* codebook, const float* code_norms, int n_embed, int64_t n_vec, int dim,
                       float* dist_out, void* workspace, int64_t workspace_bytes, void* stream);
/* rqamd_rq_code_norms <- the codebook_t.pow(2).sum(0) term of compute_distances (quantizations.py:51-52). */
int rqamd_rq_code_norms(const float* codebook, int n_embed, int dim, float* norms_out, void* stream);

/* ---- EMA codebook update (stage-1 training; VQEmbedding._update_buffers / _update_embedding, quantizations.py:80-129) ----
 * rqamd_rq_ema_accumulate <- :85-99: count_out[k] = number of vectors whose nearest code is k, sum_out[k][:] = their sum (the
 *   reference's one_hot.sum(1) and one_hot @ vectors), x (n_vec, dim) fp32, idx (n_vec) int64, added in ascending vector order
 *   (deterministic; no (n_embed x n_vec) one-hot matrix).  The caller all-reduces both over the ranks (:101-103) before
 * rqamd_rq_ema_update <- :105-118: cluster_size_ema = decay * cluster_size_ema + (1 - decay) * count, embed_ema likewise with sum;
 *   restart_vectors (n_embed, dim) fp32 or NULL: the dead-code restart with the caller's random vectors (usage = cluster_size_ema
 *   >= 1; unused codes take the random vector and cluster size 1).  In place.  decay is a double (ABI v4): torch computes the step
 *   as mul_(decay) then add_(x, alpha = 1 - decay) with alpha taken in double, and so does the kernel.
 * rqamd_rq_ema_normalize <- :120-129: weight_out[k][:] = embed_ema[k][:] / (n (cluster_size_ema[k] + eps) / (n + n_embed eps)),
 *   n = *n_total, a device scalar holding sum(cluster_size_ema) (computed by the caller). */
int rqamd_rq_ema_accumulate(const float* x, const int64_t* idx, int64_t n_vec, int dim, int n_embed, float* count_out,
                            float* sum_out, void* stream);
int rqamd_rq_ema_update(float* cluster_size_ema, float* embed_ema, const float* count, const float* sum,
                        const float* restart_vectors, int n_embed, int dim, double decay, void* stream);
int rqamd_rq_ema_normalize(const float* cluster_size_ema, const float* embed_ema, const float* n_total, int n_embed, int dim,
                           float eps, float* weight_out, void* stream);

/* rqamd_rq_embed <- RQBottleneck.embed_code :297-311 (mode 0: sum over depth),
 * embed_code_with_depth :313-334 (mode 1: (n_vec, depth, dim)), and the depth-cumsum the sampler
 * feeds to head_mlp (transformers.py:157-160; mode 2).  codes (n_vec, depth) int64; out fp32.
 * code == n_embed selects the zero padding row (quantizations.py:28).  Any other out-of-range code is an index error in
 * the reference (F.embedding's device-side assert); here the kernel traps, which surfaces as a HIP error at the next
 * synchronisation -- it cannot be reported asynchronously through the return value, and it is never silent zeros. */
int rqamd_rq_embed(const int64_t* codes, const float* const* codebooks, const int* n_embed, int depth,
                   int64_t n_vec, int dim, int mode, float* out, void* stream);

/* ---- sampler ------------------------------------------------------------------------------
 * rqamd_sample_logits <- sample_from_logits / top_k_logits / top_p_probs (rqvae/utils/utils.py:60-123):
 * fp32, /temperature, keep logits >= k-th largest (ties kept), NaN -> -inf, softmax, nucleus filter
 * (first token whose inclusive cumulative mass reaches p is kept), renormalise, one multinomial draw
 * per row by the exponential race max_i p_i/E_i (the algorithm torch.multinomial uses for one sample)
 * on Philox4x32-10 keyed by (seed, offset).  No host synchronisation.  top_k <= 0 or >= vocab: no
 * top-k; top_p < 0: no nucleus step (top_p = 1.0 still runs it, as the reference does,
 * transformers.py:323-324).  probs_out (rows, vocab) fp32 receives the filtered distribution
 * (parity hook) or NULL; samples_out (rows) int64 or NULL.  row_flags: caller-owned scratch of `rows` ints (lets rows
 * whose top-k survivors fit in registers skip the general kernel) or NULL. */
int rqamd_sample_logits(const float* logits, int rows, int vocab, float temperature, int top_k,
                        float top_p, uint64_t seed, uint64_t offset, int64_t* samples_out,
                        float* probs_out, int* row_flags, void* stream);

/* ---- RQ-VAE encoder / decoder engine -------------------------------------------------------
 * Handle = packed bf16 weights + activation workspace for Encoder/Decoder (modules.py:10-202),
 * quant_conv / post_quant_conv (rqvae.py:68-69).  Parameters are pushed by their reference
 * state_dict names ("decoder.up.3.block.1.conv2.weight", ...), fp32 device pointers in torch layout. */
typedef struct rqamd_vae rqamd_vae;
typedef struct {
    int ch, out_ch, in_channels, resolution, z_channels, num_res_blocks;
    int n_levels;             /* len(ch_mult) */
    int ch_mult[8];
    int n_attn_res;
    int attn_resolutions[8];
    int embed_dim;            /* RQVAE embed_dim (quant_conv out / post_quant_conv in) */
    int double_z;
} rqamd_vae_config;

int rqamd_vae_create(const rqamd_vae_config* cfg, rqamd_vae** out);
int rqamd_vae_destroy(rqamd_vae* h);
/* options of a handle that the config struct does not carry (ABI v7).  "resamp_with_conv" = 0 | 1 <- ddconfig.resamp_with_conv (modules.py:12,103;
 * layers.py:20-57): 1 (default, every released config) = Upsample / Downsample carry their 3 x 3 conv; 0 = bare nearest-2x
 * upsample / 2 x 2 average pool, no `*.upsample.conv.*` / `*.downsample.conv.*` parameters. */
int rqamd_vae_set_option(rqamd_vae* h, const char* name, int value);
/* copies + repacks (fp32 -> bf16, OIHW -> O,kh,kw,I) on `stream`; the source may be freed after
 * the stream reaches this point. */
int rqamd_vae_set_param(rqamd_vae* h, const char* name, const float* dev_ptr, const int64_t* shape,
                        int ndim, void* stream);
/* rqamd_vae_decode <- RQVAE.decode (rqvae.py:85-89) + Decoder.forward (modules.py:171-202):
 * z_q (batch, h, w, embed_dim) fp32 NHWC -> out (batch, out_ch, H, W) fp32 NCHW. */
int rqamd_vae_decode(rqamd_vae* h, const float* z_q, int batch, float* out, void* stream);
/* rqamd_vae_encode <- RQVAE.encode (rqvae.py:80-83) + Encoder.forward (modules.py:73-98):
 * x (batch, in_channels, H, W) fp32 NCHW -> z_e (batch, h, w, embed_dim) fp32 NHWC. */
int rqamd_vae_encode(rqamd_vae* h, const float* x, int batch, float* z_e, void* stream);

/* ---- RQ-Transformer sampling engine --------------------------------------------------------
 * Handle = packed bf16 weights, fixed-capacity KV caches, device-side step state and (optionally)
 * captured hipGraphs for RQTransformer.sample / cached_forward (transformers.py:190-369),
 * AttentionStack/AttentionBlock/MultiSelfAttention (attentions.py:39-169) and the classifier. */
typedef struct rqamd_rqt rqamd_rqt;
typedef struct {
    int embed_dim, n_head;
    int n_layer_body, n_layer_head;
    int vocab_size;           /* shared classifier / codebook size */
    int input_embed_dim;      /* codebook dim fed to input_mlp/head_mlp */
    int vocab_size_cond, block_size_cond;
    int H, W, D;              /* block_size */
    int gelu_v2;              /* attentions.py:25-36: 0 = exact erf GELU ('v1') in both stacks, 1 = x*sigmoid(1.702x) ('v2') in both;
                                 2 = body 'v1' / head 'v2', 3 = body 'v2' / head 'v1' (the stacks have their own block configs) */
    /* stage-2 flags (transformers.py:60-99; every released config sets all five): 0 selects the primitives.py variants --
     * learned token embeddings (nn.Embedding / TupleEmbedding "tok_emb") instead of the RQ-VAE codebook through
     * input_mlp / head_mlp, no depth cumsum in the head context, one classifier matrix per depth (BatchLinear) */
    int input_emb_vqvae, head_emb_vqvae, shared_tok_emb, shared_cls_emb, cumsum_depth_ctx;
    int vocab_sizes[8];       /* per-depth vocabulary (all equal to vocab_size unless shared_* are off); vocab_size = max */
} rqamd_rqt_config;

int rqamd_rqt_create(const rqamd_rqt_config* cfg, rqamd_rqt** out);
int rqamd_rqt_destroy(rqamd_rqt* h);
/* options of a handle that the config struct does not carry (ABI v7).  "head.n_head" <- head.block.n_head where it differs from
 * body.block.n_head = cfg.n_head (transformers.py:86-87: each AttentionStack has its own block config; attentions.py:44-57: any
 * embed_dim / n_head).  Head size 64 -- every released config -- takes the tuned attention kernels; any other size <= 256 a plain
 * wavefront-per-(row, head) kernel (bf16 / fp16 cache only: the RQAMD_KV 8-bit formats need 64). */
int rqamd_rqt_set_option(rqamd_rqt* h, const char* name, int value);
int rqamd_rqt_set_param(rqamd_rqt* h, const char* name, const float* dev_ptr, const int64_t* shape,
                        int ndim, void* stream);
/* rqamd_rqt_sample <- RQTransformer.sample (transformers.py:294-369) with cached=True:
 * partial (batch,H,W,D) int64 (not modified); cond (batch, block_size_cond) int64 or NULL (zeros);
 * codebooks[d] (n_embed, input_embed_dim) fp32 without padding row (model_aux.get_code_emb_with_depth);
 * top_k[D] / top_p[D] per-depth lists already normalised as transformers.py:314-330 does;
 * codes_out (batch,H,W,D) int64.  use_graph != 0 replays one captured hipGraph per (h,w). */
int rqamd_rqt_sample(rqamd_rqt* h, const int64_t* partial, const int64_t* cond, int batch,
                     const float* const* codebooks, int start_h, int start_w, float temperature,
                     const int* top_k, const float* top_p, uint64_t seed, uint64_t offset,
                     int use_graph, int64_t* codes_out, void* stream);
/* rqamd_rqt_logits <- the same cached_forward stepping (transformers.py:190-287) driven
 * teacher-forced over given codes, returning every step's logits: logits_out (batch,H,W,D,vocab) fp32.
 * This is the parity hook against RQTransformer.forward (transformers.py:113-188). */
int rqamd_rqt_logits(rqamd_rqt* h, const int64_t* codes, const int64_t* cond, int batch,
                     const float* const* codebooks, float* logits_out, void* stream);
/* rqamd_rqt_forward <- RQTransformer.forward (transformers.py:113-188) including the text-conditioned return value
 * (seq_logits, cond_logits) (:150-153,:185-186): as rqamd_rqt_logits, plus cond_logits_out (batch, block_size_cond-1,
 * vocab_size_cond) fp32 = cond_classifier over the body outputs of the conditioning prefix (NULL to skip; must be NULL
 * when block_size_cond <= 1). */
int rqamd_rqt_forward(rqamd_rqt* h, const int64_t* codes, const int64_t* cond, int batch,
                      const float* const* codebooks, float* logits_out, float* cond_logits_out, void* stream);
/* Stepping form of rqamd_rqt_sample for callers that draw the samples themselves -- e.g. torch.multinomial on the filtered
 * probabilities, which consumes the device generator exactly as the reference's sample_from_logits does
 * (rqvae/utils/utils.py:112; RQTransformer.sample transformers.py:346-364 is this loop):
 *   step_begin     copies partial / cond into the handle, runs the conditioning prefix; codebooks as for rqamd_rqt_sample
 *                  (the pointer ARRAY is copied, the codebooks must stay alive until step_end)
 *   step_logits    logits (batch, vocab_size) fp32 of step (pos, d) given the codes written so far; steps must come in order
 *                  (pos ascending, d = 0..D-1); d < 0 runs only the body stack of that position (positions before start_loc,
 *                  whose codes are given: KV cache only).  *logits_dev points into the handle's workspace and is valid until
 *                  the next call on the handle.  Columns >= vocab_sizes[d] are masked to -inf (LogitMask).
 *   step_set_code  writes codes[batch] as code (pos, d) of every image
 *   step_end       copies the codes (batch,H,W,D) out (codes_out may be NULL) and ends the sequence.
 * Any other call on the handle ends a sequence in progress. */
int rqamd_rqt_step_begin(rqamd_rqt* h, const int64_t* partial, const int64_t* cond, int batch,
                         const float* const* codebooks, void* stream);
int rqamd_rqt_step_logits(rqamd_rqt* h, int pos, int d, const float** logits_dev, void* stream);
int rqamd_rqt_step_set_code(rqamd_rqt* h, int pos, int d, const int64_t* codes, void* stream);
int rqamd_rqt_step_end(rqamd_rqt* h, int64_t* codes_out, void* stream);
/* timing hook for bench.py: average device time (ms) of the engine's weight-streaming GEMM launches
 * during the last rqamd_rqt_sample call is not observable from outside the stream, so the engine can
 * bracket every GEMM launch of one call with HIP events (profile != 0 disables graphs for that call). */
int rqamd_rqt_set_profile(rqamd_rqt* h, int profile);   /* 0 off; 1 events around every GEMM / attention launch (eager); 2 skip the
                                                          * GEMM launches (graphs on): pass time minus this = the GEMMs' time in situ */
int rqamd_rqt_get_profile(rqamd_rqt* h, double* gemm_ms_total, int64_t* gemm_launches,
                          double* gemm_bytes_total, double* gemm_flops_total);
/* the cached-attention launches (MultiSelfAttention.forward with the KV cache, attentions.py:60-104) of the same profiled call:
 * total device time and launch count (the KV bytes they read are a function of the shapes: bench.py computes them). */
int rqamd_rqt_get_profile_attn(rqamd_rqt* h, double* attn_ms_total, int64_t* attn_launches);

/* ---- diagnostics ------------------------------------------------------------------------------
 * One raw launch of the engines' bf16 MFMA GEMM: out[M,N] = A[M,K] . W[N,K]^T (+bias), both operands bf16
 * K-contiguous (W = nn.Linear.weight layout).  epi: 0 bf16 out, 1 bf16 + GELU, 3 fp32 out, 4 fp32 split-K
 * partial slabs out[splitk][M][N] (no bias); + 16: skip the epilogue (ablation); + 64 / + 96: stage the operands by
 * LDS-DMA through 2 / 3 LDS stages (tiles 128x64, 128x128, 256x128); 4 + 2048 (splitk 1): update out[M][N] in place,
 * out = (out + A.W^T) + bias -- the residual-stream epilogue of the decode step.  bm/bn <= 0: the engine's own tile choice.
 * Used by the kernel-level parity test and scripts/gemm_bench.py; not part of the reference-facing surface. */
int rqamd_dbg_gemm_bf16(const void* A, const void* W, int M, int N, int K, const float* bias, int epi,
                        void* out, int bm, int bn, int splitk, void* stream);
/* One raw implicit-GEMM convolution launch (the conv form of the same kernel): x NHWC bf16
 * [B][H>>ups][W>>ups][Cin] (H, W = virtual input size after the folded nearest-2x upsample), w bf16
 * [Cout][k][k][Cin], out NHWC bf16 (+bias, +resid if not NULL); stride 2 = Downsample (layers.py:50-54). */
int rqamd_dbg_conv_bf16(const void* x, const void* w, const float* bias, const void* resid, int B, int H, int W,
                        int Cin, int Cout, int ksize, int stride, int ups, void* out, int bm, int bn, int flags,
                        void* stream);

/* One launch of the halo-reuse 3x3 / stride-1 conv used for the high-resolution decoder/encoder layers
 * (csrc/conv_halo.hip): x NHWC bf16 [B][H][W][Cin], w bf16 [Cout][3][3][Cin], out NHWC bf16; gn = NULL or fp32
 * [B][Cin][2] (scale, shift): the input is then read as silu(x*scale + shift), i.e. GroupNorm+SiLU fused into the
 * staging (ResnetBlock norm -> swish -> conv, layers.py:100-120).  stats = NULL or fp32 [B][(H/8)*(W/32)][32][2]:
 * per-tile (sum, sum of squares) of the 32 GroupNorm groups of `out` (Cout 128/256/512), taken in the epilogue for
 * the next layer's Normalize.  ups != 0: x is [B][H/2][W/2][Cin], read through the nearest 2x upsample of
 * Upsample.forward (layers.py:31-35; gn and resid must be NULL).  Needs H % 8 == 0, W % 32 == 0, H >= 64,
 * Cin % 64 == 0, Cout % 128 == 0. */
int rqamd_dbg_conv_halo_bf16(const void* x, const void* w, const float* bias, const float* gn, const void* resid,
                             int B, int H, int W, int Cin, int Cout, int ups, void* out, float* stats, void* stream);

/* One launch of the MFMA Decoder.conv_out kernel (modules.py:165-169): x NHWC bf16 [B][H][W][Cin], w fp32
 * [Cout][3][3][Cin] (Cout <= 4), y NCHW fp32 [B][Cout][H][W]; gn as above (norm_out + swish fused into the
 * staging).  Needs H % 4 == 0, W % 32 == 0, Cin in {64, 128, 256}. */
/* diagnostics: w (Cout,3,3,Cin) bf16 -> wsub (4,Cout,2,2,Cin) bf16, the pre-summed weights of the sub-pixel form of Upsample.conv
 * (layers.py:20-35: nearest 2x + 3x3 conv = four 2x2 convs over the source image); rqamd_dbg_conv_halo_bf16 takes them with ups bit 6 set */
int rqamd_dbg_ups_subpixel_weights(const void* w, int Cout, int Cin, void* wsub, void* stream);
int rqamd_dbg_conv_out_bf16(const void* x, const float* w, const float* bias, const float* gn, int B, int H, int W,
                            int Cin, int Cout, float* y, void* stream);

/* One launch of the MFMA Encoder.conv_in kernel (modules.py:23-27): x NCHW fp32 [B][3][H][W], w fp32 [3][3][3][128]
 * = (ky, kx, ci, cout), y NHWC bf16 [B][H][W][128].  Needs H % 8 == 0, W % 32 == 0. */
int rqamd_dbg_conv_in_bf16(const float* x, const float* w, const float* bias, int B, int H, int W, void* y, void* stream);

/* Kernel variants are selected by the number of rows (batch).  factor > 1 makes the selection logic see rows * factor, so
 * that the large-batch variants run on test-sized inputs (results must not change); 1 restores normal behaviour. */
int rqamd_dbg_set_row_scale(int factor);
/* Diagnostics (scripts/lanes_probe2.py, round 6): `dst` -- a handle of the same configuration that has not run yet -- drops its own
 * parameter arena and runs on `src`'s (its own workspace, KV caches, graphs): two handles fed half a batch each on two streams are two
 * independent decode chains over one copy of the weights.  Measured slower than one chain at every batch (profiles/r06_lanes_probe.txt);
 * not used by the product path. */
int rqamd_dbg_rqt_share_params(rqamd_rqt* dst, const rqamd_rqt* src);

/* The dense bf16 MFMA rate the board sustains, measured: `launches` launches of one 512-thread workgroup per CU, every wavefront issuing
 * n_per_wave v_mfma_f32_32x32x16_bf16 from registers alone.  mode 0: constant operands (the data-sheet instruction rate); mode 1: operands
 * that change with every instruction (~N(0,1) bf16), which the board's power limit lets through at a lower clock.  scratch: >= 4 device
 * bytes; *flop_out = the FLOPs issued (the caller times the stream).  bench.py reports the mode-1 rate beside `roofline.peak`. */
int rqamd_dbg_mfma_rate(int mode, int n_per_wave, int launches, float* scratch, double* flop_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RQAMD_H */
