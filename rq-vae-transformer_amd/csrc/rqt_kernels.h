// rqt_kernels.h -- non-GEMM kernels of the RQ-Transformer decode step (gfx950).
//
// Reference call sites (rqvae/models/rqtransformer/):
//   resid_ln_kernel     <- x + attn / x + mlp residual adds and nn.LayerNorm (attentions.py:126-142,
//                          transformers.py:91); also reduces the split-K partial slabs of the
//                          producing GEMM (launch-boundary reduce) and adds its bias
//   attn_decode_kernel  <- MultiSelfAttention.forward with caching=True (attentions.py:60-104):
//                          KV append (torch.cat :75-76), q.(k^T/sqrt(hs)) :87, causal mask :88-91,
//                          softmax :92, att.v :95 -- one wavefront per (batch row, head)
//   embed_tokens_kernel <- model_aux.get_code_emb_with_depth + .sum(-2) / cumsum(-2)
//                          (transformers.py:109-111,218-225,249-257), only for the NEW position
//   cond_embed_kernel   <- cond_emb(cond) + pos_emb_cond (transformers.py:224)
//   sample_kernel       <- sample_from_logits (rqvae/utils/utils.py:82-123), no host sync
#pragma once
#include "rq_hip.h"

struct ResidLnArgs {
    const float* x_in;      // [rows][E]
    float* x_out;           // [rows][E] (may alias x_in); may be null when only y is wanted
    const float* slabs;     // [n_slabs][rows][E] split-K partials of the producing GEMM, or null
    int n_slabs;
    const float* bias;      // [E] bias of the producing GEMM, or null
    const float* addvec;    // [E] broadcast add (positional embedding), or null
    const float* gamma;     // LayerNorm weight/bias; null => no LN output
    const float* beta;
    bf16_t* y;              // [rows][E] bf16 LN output
    int rows, E;
    float eps;
};

struct AttnDecodeArgs {
    const bf16_t* qkv;      // [rows][3E]: q | k | v, head h at columns h*hd..h*hd+hd-1 of each third (hd = E / nh; 64 in every released config)
    bf16_t* kc;             // K cache [rows][nh][Tcap][hd]
    bf16_t* vc;             // V cache [rows][nh][Tcap][hd]
    float* ksc;             // null: bf16 keys.  Non-null (opt-in RQAMD_KV=int8k, body stack): `kc` holds [rows][nh][Tcap][64] BYTES,
                            // key component = (byte - 128) * ksc[row][head][position], one absmax / 127 scale per cached key
    float* vsc;             // null: bf16 values.  Non-null (opt-in RQAMD_KV=int8kv; needs ksc): `vc` holds bytes + these scales, like the keys
    bf16_t* y;              // [rows][E]
    const int* step;        // device-side step counter (or null)
    int step_off;           // t = *step + step_off = number of cached keys before this token
    int t_max;              // host-side upper bound of t for this launch (selects the register-block count), -1 = Tcap-1
    int rows, nh, E, Tcap;
};

// Multi-token prefill of the conditioning prefix (transformers.py:235-239 of the reference: the first cached body step
// runs the whole prefix through MultiSelfAttention.forward with the causal mask, attentions.py:60-104).
struct AttnPrefillArgs {
    const bf16_t* qkv;      // [n_img * P][3E], row = img * P + i (token i of image img)
    bf16_t* kc;             // K cache of the FIRST image of this chunk: [n_img][nh][Tcap][64]; positions 0..P-1 are written
    bf16_t* vc;
    float* ksc;             // as AttnDecodeArgs::ksc (of the first image of the chunk), or null
    float* vsc;             // as AttnDecodeArgs::vsc, or null
    bf16_t* y;              // [n_img * P][E]
    int n_img, P, nh, E, Tcap;
};

struct EmbedTokArgs {
    const int64_t* xs;      // [rows][HW][D] codes
    const float* cb[8];     // per-depth codebooks (K, dim), padding row excluded
    int K[8];
    const int* pos;         // device-side spatial position
    int pos_off;            // position read = *pos + pos_off
    int depth_lo;           // sum over depths [depth_lo, n_depth) (0: the depth cumsum; n_depth - 1: cumsum_depth_ctx off)
    int n_depth;
    int rows, HW, D, dim;
    bf16_t* out;            // [rows][dim]
};

struct SampleArgs {
    const float* logits;    // [rows][V]
    int rows, V;
    float temperature;
    int top_k;              // <=0 or >=V: off
    float top_p;            // <0: off
    const uint64_t* rng;    // device {seed, offset}; or null -> seed/offset below
    uint64_t seed, offset;
    const int* pos;         // device-side spatial position (or null)
    int d, D;               // depth index / depth count: draw counter = offset + (*pos * D + d)
    int64_t* out;           // samples: out[row * out_stride + (*pos * D + d)] (pos null -> out[row*out_stride])
    long out_stride;
    float* probs_out;       // [rows][V] filtered distribution, or null
    int* redo;              // [rows] workspace: rows the top-k kernel hands to the general kernel (null: general kernel only)
};

int rq_launch_resid_ln(const ResidLnArgs& a, hipStream_t s);
int rq_launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s);
int rq_launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t s);
int rq_launch_embed_tokens(const EmbedTokArgs& a, hipStream_t s);
// x[(img, i)] = cond_emb[cond[img][i]] + pos_emb_cond[i] for i < n_tok: the prefix rows of the prefill
int rq_launch_cond_embed_multi(const int64_t* cond, int cond_stride, int n_tok, const float* cond_emb, int vocab_cond,
                               const float* pos_emb_cond, float* x, int n_img, int E, hipStream_t s);
int rq_launch_cond_embed(const int64_t* cond, int cond_stride, int cond_idx, const float* cond_emb, int vocab_cond,
                         const float* pos_emb_cond, float* x, int rows, int E, hipStream_t s);
int rq_launch_sample(const SampleArgs& a, hipStream_t s);
// learned token embeddings (tok_emb: nn.Embedding, or TupleEmbedding = per-depth tables at row offsets offs[d], primitives.py:25-75):
// x[b][:] = sum_{d in [d_lo, d_hi)} table[offs[d] + code[b][pos][d]][:] + add[row][:], row = (pos_dev ? *pos_dev : 0) + add_row
struct TokEmbedArgs {
    const int64_t* xs;      // [rows][HW][D] codes
    const float* table;     // [sum V][E] fp32
    int offs[8], V[8];
    const int* pos;         // device-side spatial position (or null)
    int pos_off;            // code position = *pos + pos_off
    int d_lo, d_hi;
    const float* add;       // [*][E] positional table
    int add_by_pos;         // 1: row = *pos + add_row, 0: row = add_row
    int add_row;
    int rows, HW, D, E;
    float* out;             // [rows][E] fp32
};
int rq_launch_tok_embed(const TokEmbedArgs& a, hipStream_t s);
// logits[:, v_lo:V] = -inf (per-depth vocabularies smaller than the classifier's width)
int rq_launch_mask_logits(float* logits, int rows, int V, int v_lo, hipStream_t s);
int rq_launch_cvt_bf16(const float* src, bf16_t* dst, long n, hipStream_t s);
// dst[c][r] = bf16(src[r][c]): BatchLinear's (in, out) matrices into the GEMM's K-contiguous weight layout
int rq_launch_cvt_bf16_transpose(const float* src, bf16_t* dst, int R, int Cc, hipStream_t s);
int rq_launch_set_int(int* p, int v, hipStream_t s);
int rq_launch_add_int(int* p, int v, hipStream_t s);
