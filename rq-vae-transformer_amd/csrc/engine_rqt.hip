// engine_rqt.hip -- RQ-Transformer sampling engine (host side of the decode loop, gfx950).
//
// Stands behind RQTransformer.sample / cached_forward (rqvae/models/rqtransformer/transformers.py:190-369),
// AttentionStack.cached_forward / AttentionBlock.cached_forward / MultiSelfAttention.forward
// (attentions.py:60-104,134-142,162-165) and the classifier (transformers.py:90-99,278-285).
//
// Differences from the reference's eager loop, none of which change the math beyond rounding:
//  * only the NEW position is embedded (the reference re-embeds the whole prefix every step,
//    transformers.py:218-225,249-257); sum_d (W e_d + b) is computed as W (sum_d e_d) + D*b,
//  * q/k/v projections are one GEMM over the row-concatenated weight [query;key;value],
//  * the KV cache has fixed capacity and is appended in place (the reference torch.cat's it, :75-76),
//  * residual adds, split-K reduction and LayerNorm are one kernel (resid_ln),
//  * every step-dependent quantity (spatial position, RNG offset) lives in device memory, so the
//    per-position launch sequence (1 body step + D head steps + D sampler calls) is identical for
//    every position and can be captured once as a hipGraph and replayed H*W-1 times; the sampler has
//    no host synchronisation (the reference syncs once per step, rqvae/utils/utils.py:103).
//  * weights are bf16 (fp32 accumulate, fp32 residual stream / LayerNorm / softmax / logits).
#include <string.h>
#include <string>
#include <vector>
#include "gemm.h"
#include "rq_common.h"
#include "rqt_kernels.h"

struct RqtLayer {
    bf16_t *wqkv, *wproj, *wfc1, *wfc2;
    float *bqkv, *bproj, *bfc1, *bfc2, *ln1w, *ln1b, *ln2w, *ln2b;
    bf16_t *kc, *vc;   // KV cache (workspace, per batch capacity)
    float* ksc;        // body layers with the opt-in 8-bit key cache (RQAMD_KV=int8k / int8kv): per-key scales, `kc` then holds bytes; else null
    float* vsc;        // body layers with RQAMD_KV=int8kv: per-value-row scales, `vc` then holds bytes; else null
    int nh;            // attention heads of the layer's stack (body: cfg.n_head; head: the same unless the "head.n_head" option says otherwise)
};

struct GemmProfile {
    bool on = false;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    double bytes = 0, flops = 0;
    double ms_total = 0;
    int64_t launches = 0;
    // the decode-step attention launches of the same call, bracketed the same way (second roofline: KV-cache bytes / this time)
    std::vector<hipEvent_t> ev_attn;
    size_t used_attn = 0;
    double attn_ms_total = 0;
    int64_t attn_launches = 0;
    // profile == 2: the decode-step GEMM launches are SKIPPED (graphs stay on; everything else runs, on garbage).  The time of a
    // sampling pass minus the time of the same pass without its GEMMs is what the GEMMs cost inside the captured graphs --
    // bracketing every launch with events (profile == 1) runs eagerly and adds the dispatch latency of two markers to a ~7-us kernel.
    bool skip_gemm = false;
};

struct rqamd_rqt {
    rqamd_rqt_config cfg;
    int E, HW, D, V, Din, cond_len, Tbody;
    DevBuf arena;                 // all parameters
    std::vector<RqtLayer> body, head;
    bf16_t *w_in, *w_headin, *w_cls;
    float *b_in, *b_headin, *b_cls, *cls_lnw, *cls_lnb;
    bf16_t* w_ccls = nullptr;                      // cond_classifier (transformers.py:100-104), text-conditioned models only
    float *b_ccls = nullptr, *ccls_lnw = nullptr, *ccls_lnb = nullptr;
    int n_ccls_seen = 0;
    float *cond_emb, *pos_cond, *pos_hw, *pos_d;
    float* tok_emb = nullptr;      // learned token embeddings (variants with input_emb_vqvae / head_emb_vqvae off), fp32 [sum V][E]
    int tok_offs[8] = {}, Vd[8] = {};
    long tok_rows = 0;
    bool in_vq = true, head_vq = true, shared_cls = true, cumsum = true;
    float *body_in_bias, *head_in_bias;   // [HW][E], [D][E] derived tables
    bool tables_dirty = true;
    std::vector<std::string> seen;
    size_t n_required = 0;

    // workspace for `cap` rows
    int cap = 0;
    DevBuf ws, kv;
    float *x, *xh, *slabs, *logits;
    bf16_t *y, *qkv, *ya, *hbuf, *ain;
    int64_t *xs, *cond;
    int* st;            // [0] = spatial position
    uint64_t* rng;      // {seed, offset}
    int* smp_redo;      // [rows] sampler workspace (rows the top-k kernel hands back to the general kernel)
    int max_slabs = 8;
    int cur_gelu_v2 = 0;   // GELU form of the stack being run (cfg.gelu_v2: 0 both erf, 1 both sigmoid, 2 body erf / head sigmoid, 3 body sigmoid / head erf)
    bool kv_int8k = false;   // RQAMD_KV=int8k / int8kv when the handle was created: body-stack keys cached as 64 bytes + one fp32 scale (rqt_kernels.hip)
    bool kv_int8v = false;   // RQAMD_KV=int8kv: the body-stack values likewise (round 6)

    // graph cache
    // one captured position per 8-key bucket of the body context (the attention kernel variant is baked in)
    static constexpr int NGRAPH = 33;
    hipGraphExec_t gexec[NGRAPH] = {};
    struct Key { int B; float T; int tk[8]; float tp[8]; const float* cb[8]; void* stream; } gkey;
    bool gvalid = false;

    GemmProfile prof;

    // stepping form (rqamd_rqt_step_*): one (position, depth) step per call, the caller draws the samples
    bool step_on = false;
    int step_B = 0, step_pos = 0, step_d = 0;      // next expected step
    const float* step_cb[8] = {};
    const float* step_pend_slabs = nullptr;        // the body's last fc2 partials, consumed by depth 0 of the same position
    int step_pend_n = 0;
    const float* step_pend_bias = nullptr;
};

// -------------------------------------------------------------------------------------------------
__global__ void bias_table_kernel(const float* bias, float scale, const float* pos, float* out, int rows, int E) {
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)rows * E) return;
    int e = (int)(gid % E);
    out[gid] = scale * bias[e] + pos[gid];
}
__global__ void set_rng_kernel(uint64_t* rng, uint64_t seed, uint64_t offset) {
    if (threadIdx.x == 0) { rng[0] = seed; rng[1] = offset; }
}

static size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" int rqamd_rqt_create(const rqamd_rqt_config* c, rqamd_rqt** out) {
    if (!c || !out) return rq_fail(RQAMD_ERR_INVALID, "rqt_create: null argument");
    if (c->n_head < 1 || c->embed_dim % c->n_head || c->embed_dim / c->n_head > 256)      // 64 is what the tuned kernels take; any other size <= 256: the plain attention kernel
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "rqt_create: embed_dim=%d must be a multiple of n_head=%d with head_dim <= 256", c->embed_dim, c->n_head);
    if (c->embed_dim % 64 || c->input_embed_dim % 64 || c->embed_dim > 4096)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "rqt_create: embed_dim / input_embed_dim must be multiples of 64, embed_dim <= 4096");
    if (c->n_layer_body < 1 || c->n_layer_head < 0)      // head.n_layer = 0: the depth-1 "VQ-GAN" shapes (measure_throughput/__main__.py:166-210)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "rqt_create: needs >= 1 body layer and >= 0 head layers");
    if (c->D < 1 || c->D > 8 || c->H < 1 || c->W < 1) return rq_fail(RQAMD_ERR_INVALID, "rqt_create: bad block_size");
    if (c->gelu_v2 < 0 || c->gelu_v2 > 3) return rq_fail(RQAMD_ERR_INVALID, "rqt_create: gelu_v2 = %d (0 .. 3)", c->gelu_v2);
    rqamd_rqt* h = new rqamd_rqt();
    h->cfg = *c;
    {   // opt-in storage format of the body stack's key cache, fixed for the life of the handle (default: bf16, what BASELINE.json asks for)
        const char* kvf = getenv("RQAMD_KV");
        if (kvf && *kvf && strcmp(kvf, "bf16") != 0) {
            if (strcmp(kvf, "int8k") != 0 && strcmp(kvf, "int8kv") != 0) { delete h; return rq_fail(RQAMD_ERR_INVALID, "rqt_create: RQAMD_KV=%s (bf16, int8k or int8kv)", kvf); }
            if (c->embed_dim != c->n_head * 64) { delete h; return rq_fail(RQAMD_ERR_UNSUPPORTED, "rqt_create: RQAMD_KV=%s is written for head_dim 64 (embed_dim=%d n_head=%d)", kvf, c->embed_dim, c->n_head); }
            h->kv_int8k = true;
            h->kv_int8v = strcmp(kvf, "int8kv") == 0;
        }
    }
    h->E = c->embed_dim; h->HW = c->H * c->W; h->D = c->D; h->V = c->vocab_size; h->Din = c->input_embed_dim;
    h->cond_len = c->block_size_cond < 1 ? 1 : c->block_size_cond;
    h->Tbody = h->HW + h->cond_len - 1;
    if (h->Tbody > 256) { delete h; return rq_fail(RQAMD_ERR_UNSUPPORTED, "rqt_create: context %d > 256", h->Tbody); }
    const size_t E = h->E, V = h->V, Din = h->Din;
    const int vc = c->vocab_size_cond < 1 ? 1 : c->vocab_size_cond;
    const size_t per_layer = al(3 * E * E * 2) + al(E * E * 2) + 2 * al(4 * E * E * 2) + al(3 * E * 4) + al(4 * E * 4) + 6 * al(E * 4);
    size_t total = per_layer * (c->n_layer_body + c->n_layer_head) + 2 * al(E * Din * 2) + al(V * E * 2) + 4 * al(E * 4) + al(V * 4)
                   + al(vc * E * 4) + al(h->cond_len * E * 4) + 2 * al(h->HW * E * 4) + 2 * al(h->D * E * 4);
    if (h->cond_len > 1) total += al((size_t)vc * E * 2) + al((size_t)vc * 4) + 2 * al(E * 4);
    // stage-2 flag variants (a struct zero-filled by an old caller means "all on", i.e. the released configuration)
    const bool legacy = c->vocab_sizes[0] == 0;
    h->in_vq = legacy || c->input_emb_vqvae; h->head_vq = legacy || c->head_emb_vqvae;
    h->shared_cls = legacy || c->shared_cls_emb; h->cumsum = legacy || c->cumsum_depth_ctx;
    for (int d = 0; d < c->D; ++d) {
        h->Vd[d] = legacy ? c->vocab_size : c->vocab_sizes[d];
        if (h->Vd[d] < 1 || h->Vd[d] > c->vocab_size) { delete h; return rq_fail(RQAMD_ERR_INVALID, "rqt_create: vocab_sizes[%d] = %d not in 1..vocab_size", d, h->Vd[d]); }
    }
    if (!(h->in_vq && h->head_vq)) {
        const bool shared_tok = legacy || c->shared_tok_emb;
        // one shared table has Vd[0] rows: a depth with a larger vocabulary would read past it (the reference asserts equal
        // vocabularies for shared_tok_emb, transformers.py:55-56, and nn.Embedding raises an index error)
        for (int d = 0; shared_tok && d < c->D; ++d)
            if (h->Vd[d] > h->Vd[0]) {
                const int vd = h->Vd[d], v0 = h->Vd[0];
                delete h;
                return rq_fail(RQAMD_ERR_INVALID, "rqt_create: shared_tok_emb with vocab_sizes[%d] = %d > vocab_sizes[0] = %d", d, vd, v0);
            }
        long rows = 0;
        for (int d = 0; d < c->D; ++d) { h->tok_offs[d] = shared_tok ? 0 : (int)rows; rows += h->Vd[d]; }
        h->tok_rows = shared_tok ? h->Vd[0] : rows;
        total += al((size_t)h->tok_rows * E * 4);
    }
    if (!h->shared_cls) total += al((size_t)c->D * V * E * 2) + al((size_t)c->D * V * 4);
    { const int rc = h->arena.reserve(total); if (rc != RQAMD_OK) { delete h; return rc; } }
    char* p = (char*)h->arena.p;
    auto take = [&](size_t bytes) { char* r = p; p += al(bytes); return (void*)r; };
    auto mk = [&](std::vector<RqtLayer>& v, int n) {
        v.resize(n);
        for (auto& L : v) {
            L.wqkv = (bf16_t*)take(3 * E * E * 2); L.wproj = (bf16_t*)take(E * E * 2);
            L.wfc1 = (bf16_t*)take(4 * E * E * 2); L.wfc2 = (bf16_t*)take(4 * E * E * 2);
            L.bqkv = (float*)take(3 * E * 4); L.bfc1 = (float*)take(4 * E * 4);
            L.bproj = (float*)take(E * 4); L.bfc2 = (float*)take(E * 4);
            L.ln1w = (float*)take(E * 4); L.ln1b = (float*)take(E * 4); L.ln2w = (float*)take(E * 4); L.ln2b = (float*)take(E * 4);
            L.kc = L.vc = nullptr;
            L.ksc = L.vsc = nullptr;
            L.nh = c->n_head;
        }
    };
    mk(h->body, c->n_layer_body);
    mk(h->head, c->n_layer_head);
    h->w_in = (bf16_t*)take(E * Din * 2); h->w_headin = (bf16_t*)take(E * Din * 2);
    h->w_cls = (bf16_t*)take((h->shared_cls ? 1 : (size_t)c->D) * V * E * 2);
    h->b_in = (float*)take(E * 4); h->b_headin = (float*)take(E * 4); h->cls_lnw = (float*)take(E * 4); h->cls_lnb = (float*)take(E * 4);
    h->b_cls = (float*)take((h->shared_cls ? 1 : (size_t)c->D) * V * 4);
    if (h->tok_rows) h->tok_emb = (float*)take((size_t)h->tok_rows * E * 4);
    h->cond_emb = (float*)take(vc * E * 4); h->pos_cond = (float*)take(h->cond_len * E * 4);
    h->pos_hw = (float*)take(h->HW * E * 4); h->body_in_bias = (float*)take(h->HW * E * 4);
    h->pos_d = (float*)take(h->D * E * 4); h->head_in_bias = (float*)take(h->D * E * 4);
    if (h->cond_len > 1) {
        h->w_ccls = (bf16_t*)take((size_t)vc * E * 2); h->b_ccls = (float*)take((size_t)vc * 4);
        h->ccls_lnw = (float*)take(E * 4); h->ccls_lnb = (float*)take(E * 4);
    }
    h->n_required = 4 + 4 + 4 + 12 * 2 * 0;   // filled below
    h->n_required = 3 /*pos*/ + 1 /*cond_emb*/ + (h->in_vq ? 2 : 0) + (h->head_vq ? 2 : 0) + (h->tok_rows ? 1 : 0) + 4 /*classifier*/
                    + (size_t)16 * (c->n_layer_body + c->n_layer_head);
    *out = h;
    return RQAMD_OK;
}

extern "C" int rqamd_rqt_destroy(rqamd_rqt* h) {
    if (!h) return RQAMD_OK;
    for (auto& g : h->gexec) if (g) (void)hipGraphExecDestroy(g);
    for (auto e : h->prof.ev) (void)hipEventDestroy(e);
    for (auto e : h->prof.ev_attn) (void)hipEventDestroy(e);
    delete h;
    return RQAMD_OK;
}

// options of a handle that the config struct does not carry (include/rqamd.h)
extern "C" int rqamd_rqt_set_option(rqamd_rqt* h, const char* name, int value) {
    if (!h || !name) return rq_fail(RQAMD_ERR_INVALID, "rqt_set_option: null argument");
    if (strcmp(name, "head.n_head") == 0) {      // head.block.n_head where it differs from body.block.n_head (transformers.py:86-87: the stacks have their own block configs)
        if (value < 1 || h->E % value || h->E / value > 256)
            return rq_fail(RQAMD_ERR_UNSUPPORTED, "rqt_set_option: embed_dim=%d must be a multiple of head.n_head=%d with head_dim <= 256", h->E, value);
        for (auto& L : h->head) L.nh = value;
        h->gvalid = false;                       // captured graphs hold the old launches
        return RQAMD_OK;
    }
    return rq_fail(RQAMD_ERR_INVALID, "rqt_set_option: unknown option '%s'", name);
}

static long numel(const int64_t* shape, int ndim) {
    long n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
}

extern "C" int rqamd_rqt_set_param(rqamd_rqt* h, const char* name, const float* src, const int64_t* shape, int ndim, void* stream) {
    if (!h || !name || !src || !shape) return rq_fail(RQAMD_ERR_INVALID, "rqt_set_param: null argument");
    hipStream_t st = (hipStream_t)stream;
    const long n = numel(shape, ndim);
    const long E = h->E;
    std::string s(name);
    auto expect = [&](long want) -> int {
        if (n != want) return rq_fail(RQAMD_ERR_INVALID, "rqt_set_param(%s): %ld elements, expected %ld", name, n, want);
        return RQAMD_OK;
    };
    auto f32copy = [&](float* dst, long want) -> int {
        RQ_TRY(expect(want));
        RQ_HIP(hipMemcpyAsync(dst, src, want * 4, hipMemcpyDeviceToDevice, st));
        return RQAMD_OK;
    };
    auto bf16copy = [&](bf16_t* dst, long want) -> int {
        RQ_TRY(expect(want));
        return rq_launch_cvt_bf16(src, dst, want, st);
    };
    int rc = RQAMD_OK;
    bool known = true;
    const int vc = h->cfg.vocab_size_cond < 1 ? 1 : h->cfg.vocab_size_cond;
    if (s == "pos_emb_cond") rc = f32copy(h->pos_cond, h->cond_len * E);
    else if (s == "pos_emb_hw") rc = f32copy(h->pos_hw, h->HW * E);
    else if (s == "pos_emb_d") rc = f32copy(h->pos_d, h->D * E);
    else if (s == "cond_emb.weight") rc = f32copy(h->cond_emb, vc * E);
    else if ((s.rfind("input_mlp.", 0) == 0 && !h->in_vq) || (s.rfind("head_mlp.", 0) == 0 && !h->head_vq)) return RQAMD_OK;
    else if (s == "input_mlp.weight") rc = bf16copy(h->w_in, E * h->Din);
    else if (s == "input_mlp.bias") rc = f32copy(h->b_in, E);
    else if (s == "head_mlp.weight") rc = bf16copy(h->w_headin, E * h->Din);
    else if (s == "head_mlp.bias") rc = f32copy(h->b_headin, E);
    else if (s == "classifier.layer_norm.weight") rc = f32copy(h->cls_lnw, E);
    else if (s == "classifier.layer_norm.bias") rc = f32copy(h->cls_lnb, E);
    else if (s == "classifier.linear.weight") {
        if (h->shared_cls) rc = bf16copy(h->w_cls, (long)h->V * E);
        else {      // BatchLinear (primitives.py:96-125): (depth, in = E, out = V) -> per depth [V][E] bf16
            RQ_TRY(expect((long)h->D * E * h->V));
            if (ndim != 3) return rq_fail(RQAMD_ERR_INVALID, "rqt_set_param(%s): BatchLinear weight must be (depth, in, out)", name);
            for (int d = 0; d < h->D && rc == RQAMD_OK; ++d)
                rc = rq_launch_cvt_bf16_transpose(src + (long)d * E * h->V, h->w_cls + (long)d * h->V * E, (int)E, h->V, st);
        }
    }
    else if (s == "classifier.linear.bias") rc = f32copy(h->b_cls, (h->shared_cls ? 1 : (long)h->D) * h->V);
    else if (s == "tok_emb.weight") {
        if (!h->tok_rows) return RQAMD_OK;
        rc = f32copy(h->tok_emb, h->tok_rows * E);
    }
    else if (s == "tok_emb.offsets") return RQAMD_OK;                  // derived from the vocabulary sizes
    else if (s.rfind("cond_classifier.", 0) == 0) {                    // forward()-only head (transformers.py:150-153)
        if (h->cond_len <= 1) return RQAMD_OK;
        if (s == "cond_classifier.layer_norm.weight") rc = f32copy(h->ccls_lnw, E);
        else if (s == "cond_classifier.layer_norm.bias") rc = f32copy(h->ccls_lnb, E);
        else if (s == "cond_classifier.linear.weight") rc = bf16copy(h->w_ccls, (long)vc * E);
        else if (s == "cond_classifier.linear.bias") rc = f32copy(h->b_ccls, vc);
        else return rq_fail(RQAMD_ERR_INVALID, "rqt_set_param: unknown parameter %s", name);
        if (rc != RQAMD_OK) return rc;
        bool dupc = false;
        for (auto& k : h->seen) if (k == s) { dupc = true; break; }
        if (!dupc) { h->seen.push_back(s); h->n_ccls_seen++; }
        return RQAMD_OK;
    }
    else {
        std::vector<RqtLayer>* stack = nullptr;
        size_t off = 0;
        if (s.rfind("body_transformer.blocks.", 0) == 0) { stack = &h->body; off = strlen("body_transformer.blocks."); }
        else if (s.rfind("head_transformer.blocks.", 0) == 0) { stack = &h->head; off = strlen("head_transformer.blocks."); }
        if (!stack) known = false;
        else {
            size_t dot = s.find('.', off);
            int li = atoi(s.substr(off, dot - off).c_str());
            if (dot == std::string::npos || li < 0 || li >= (int)stack->size())
                return rq_fail(RQAMD_ERR_INVALID, "rqt_set_param: bad layer index in %s", name);
            RqtLayer& L = (*stack)[li];
            std::string leaf = s.substr(dot + 1);
            if (leaf == "ln1.weight") rc = f32copy(L.ln1w, E);
            else if (leaf == "ln1.bias") rc = f32copy(L.ln1b, E);
            else if (leaf == "ln2.weight") rc = f32copy(L.ln2w, E);
            else if (leaf == "ln2.bias") rc = f32copy(L.ln2b, E);
            else if (leaf == "attn.query.weight") rc = bf16copy(L.wqkv, E * E);
            else if (leaf == "attn.key.weight") rc = bf16copy(L.wqkv + E * E, E * E);
            else if (leaf == "attn.value.weight") rc = bf16copy(L.wqkv + 2 * E * E, E * E);
            else if (leaf == "attn.query.bias") rc = f32copy(L.bqkv, E);
            else if (leaf == "attn.key.bias") rc = f32copy(L.bqkv + E, E);
            else if (leaf == "attn.value.bias") rc = f32copy(L.bqkv + 2 * E, E);
            else if (leaf == "attn.proj.weight") rc = bf16copy(L.wproj, E * E);
            else if (leaf == "attn.proj.bias") rc = f32copy(L.bproj, E);
            else if (leaf == "mlp.0.weight") rc = bf16copy(L.wfc1, 4 * E * E);
            else if (leaf == "mlp.0.bias") rc = f32copy(L.bfc1, 4 * E);
            else if (leaf == "mlp.2.weight") rc = bf16copy(L.wfc2, 4 * E * E);
            else if (leaf == "mlp.2.bias") rc = f32copy(L.bfc2, E);
            else known = false;
        }
    }
    if (!known) return rq_fail(RQAMD_ERR_INVALID, "rqt_set_param: unknown parameter %s", name);
    if (rc != RQAMD_OK) return rc;
    bool dup = false;
    for (auto& k : h->seen) if (k == s) { dup = true; break; }
    if (!dup) h->seen.push_back(s);
    h->tables_dirty = true;
    return RQAMD_OK;
}

// -------------------------------------------------------------------------------------------------
// images per prefill chunk / workspace rows for a batch of B (text-conditioned models run the cond_len-1 prefix tokens of
// a chunk of images through the body stack at once: rows = images x (cond_len - 1))
static int prefill_chunk(const rqamd_rqt* h, int B) {
    const int P = h->cond_len - 1;
    if (P < 1) return 0;
    int pc = (B > 4096 ? B : 4096) / P;
    if (pc < 1) pc = 1;
    return pc < B ? pc : B;
}

static int ensure_batch(rqamd_rqt* h, int B) {
    if (B <= h->cap) return RQAMD_OK;
    // Failure-atomic regrowth: the old capacity is dropped BEFORE anything is freed, and the new one is recorded only after
    // every allocation succeeded -- a failed hipMalloc (the KV cache is ~135 GB at B = 8192) leaves the handle empty
    // (cap 0, no graphs), never pointing at freed memory; the caller may retry with a smaller batch.
    h->cap = 0;
    h->gvalid = false;
    for (auto& L : h->body) { L.kc = L.vc = nullptr; L.ksc = L.vsc = nullptr; }
    for (auto& L : h->head) { L.kc = L.vc = nullptr; L.ksc = L.vsc = nullptr; }
    const size_t E = h->E, V = h->V;
    const size_t brows = (size_t)B;
    const size_t prow = (size_t)prefill_chunk(h, B) * (h->cond_len - 1);
    const size_t rows = brows > prow ? brows : prow;              // activation rows (decode step or prefill chunk)
    size_t total = 2 * al(rows * E * 4) + al((size_t)h->max_slabs * rows * E * 4) + al(brows * V * 4) + 2 * al(rows * E * 2) + al(rows * 3 * E * 2)
                   + al(rows * 4 * E * 2) + al(brows * h->Din * 2) + al(brows * h->HW * h->D * 8) + al(brows * h->cond_len * 8) + al(64) + al(64) + al(brows * 4);
    RQ_TRY(h->ws.reserve(total));
    char* p = (char*)h->ws.p;
    auto take = [&](size_t bytes) { char* r = p; p += al(bytes); return (void*)r; };
    h->x = (float*)take(rows * E * 4); h->xh = (float*)take(rows * E * 4);
    h->slabs = (float*)take((size_t)h->max_slabs * rows * E * 4);
    h->logits = (float*)take(brows * V * 4);
    h->y = (bf16_t*)take(rows * E * 2); h->ya = (bf16_t*)take(rows * E * 2);
    h->qkv = (bf16_t*)take(rows * 3 * E * 2); h->hbuf = (bf16_t*)take(rows * 4 * E * 2);
    h->ain = (bf16_t*)take(brows * h->Din * 2);
    h->xs = (int64_t*)take(brows * h->HW * h->D * 8); h->cond = (int64_t*)take(brows * h->cond_len * 8);
    h->st = (int*)take(64); h->rng = (uint64_t*)take(64); h->smp_redo = (int*)take(brows * 4);
    // KV caches: body [rows][nh][Tbody][64] x2 per layer, head Tcap = D
    const size_t kvb = al(brows * E * h->Tbody * 2), kvh = al(brows * E * h->D * 2);
    // (8-bit keys: half the bytes for K plus one fp32 scale per (row, head, position))
    const size_t kkb = h->kv_int8k ? al(brows * E * h->Tbody) : kvb, ksb = h->kv_int8k ? al(brows * (E / 64) * h->Tbody * 4) : 0;
    const size_t vvb = h->kv_int8v ? kkb : kvb, vsb = h->kv_int8v ? ksb : 0;
    // The KV workspace is uncached device memory (hipDeviceMallocUncached): every line of it is written once and read once per position, GBs
    // apart -- nothing a cache could serve.  On top of the non-temporal loads of the attention kernels: 5.55 -> 5.67-5.76 TB/s on the decode
    // attention, +0.45 % on the whole step at 10752 images, level at 64 / 500 (profiles/r06_kv_uncached_ab.txt).  RQAMD_KV_UNCACHED=0: plain hipMalloc.
    static const bool kv_uncached = !(getenv("RQAMD_KV_UNCACHED") && atoi(getenv("RQAMD_KV_UNCACHED")) == 0);
    RQ_TRY(h->kv.reserve((kkb + ksb + vvb + vsb) * h->body.size() + 2 * kvh * h->head.size(), kv_uncached ? hipDeviceMallocUncached : 0u));
    char* q = (char*)h->kv.p;
    for (auto& L : h->body) {
        L.kc = (bf16_t*)q; q += kkb; L.vc = (bf16_t*)q; q += vvb;
        L.ksc = h->kv_int8k ? (float*)q : nullptr; q += ksb;
        L.vsc = h->kv_int8v ? (float*)q : nullptr; q += vsb;
    }
    for (auto& L : h->head) { L.kc = (bf16_t*)q; q += kvh; L.vc = (bf16_t*)q; q += kvh; L.ksc = L.vsc = nullptr; }
    h->cap = B;
    return RQAMD_OK;
}

static int finalize_tables(rqamd_rqt* h, hipStream_t st) {
    if (h->seen.size() - h->n_ccls_seen < h->n_required)
        return rq_fail(RQAMD_ERR_STATE, "rqt: only %zu of %zu parameters set", h->seen.size() - h->n_ccls_seen, h->n_required);
    if (!h->tables_dirty) return RQAMD_OK;
    long n = (long)h->HW * h->E;
    RQ_LAUNCH(bias_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->b_in, (float)h->D, h->pos_hw, h->body_in_bias, h->HW, h->E);
    n = (long)h->D * h->E;
    RQ_LAUNCH(bias_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->b_headin, 1.0f, h->pos_d, h->head_in_bias, h->D, h->E);
    RQ_TRY(rq_check_launch("bias_table_kernel"));
    h->tables_dirty = false;
    h->gvalid = false;
    return RQAMD_OK;
}

// one weight-streaming GEMM of the decode step; returns slab count for partial epilogues
// residual-producing GEMMs (proj, fc2): `resid` non-null asks for the branch to be added into the fp32 residual stream by
// the GEMM's own epilogue, resid = (resid + acc) + resid_bias, when the tile choice needs no K split -- then *n_slabs = 0 and
// the LayerNorm that follows reads the stream only (one fp32 read + the bf16 write instead of stream + slab in, stream + bf16
// out: 41 -> 18 us per call at 10752 rows; the slab round trip through HBM goes away too).  With a K split the partial slabs and
// their reduction in resid_ln stay as they are.  The additions happen in the same order either way: bit-identical results.
static int step_gemm(rqamd_rqt* h, const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, int epi,
                     const float* bias, const int* bias_step, int bias_stride, void* out, int ldo, int* n_slabs, hipStream_t st,
                     float* resid = nullptr, const float* resid_bias = nullptr) {
    GemmArgs a{};
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.epi = epi; a.gelu_v2 = h->cur_gelu_v2;
    a.bias = bias; a.bias_step = bias_step; a.bias_stride = bias_stride; a.out = out; a.ldo = ldo;
    int bm, bn, sk, gl = 0;
    rq_gemm_pick_tile(M, N, K, epi == EPI_F32_PARTIAL, &bm, &bn, &sk, &gl);
    if (sk > h->max_slabs) sk = h->max_slabs;
    static const bool no_fuse = getenv("RQAMD_NO_FUSE_RESID") != nullptr;      // A/B switch
    if (resid && epi == EPI_F32_PARTIAL && sk == 1 && !no_fuse && (N & 3) == 0 && !(bm == 64 && bn == 32)) {
        a.accum = 1; a.bias = resid_bias; a.out = resid; a.ldo = N;
        sk = 0;                                                              // reported slab count
    }
    a.splitk = sk > 0 ? sk : 1;
    a.glds = gl;
    if (n_slabs) *n_slabs = sk;
    GemmProfile& pf = h->prof;
    if (pf.skip_gemm) return RQAMD_OK;
    if (pf.on) {
        if (pf.used + 2 > pf.ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; RQ_HIP(hipEventCreate(&e)); pf.ev.push_back(e); }
        }
        RQ_HIP(hipEventRecord(pf.ev[pf.used], st));
    }
    RQ_TRY(rq_gemm_launch(a, bm, bn, st));
    if (pf.on) {
        RQ_HIP(hipEventRecord(pf.ev[pf.used + 1], st));
        pf.used += 2;
        pf.bytes += (double)N * K * 2 + (double)M * K * 2 + (double)M * N * (epi >= EPI_F32 ? 4.0 * (sk > 0 ? sk : 2) : 2.0);
        pf.flops += 2.0 * M * N * K;
    }
    return RQAMD_OK;
}

struct Pending { const float* slabs; int n; const float* bias; };   // un-reduced output of the previous GEMM

// one transformer block on `rows` single-token rows; x is the fp32 residual stream (updated lazily:
// `pend` carries the previous block's fc2 partials + bias into this block's first resid_ln)
struct PrefillCtx { int img0, n_img, P; };   // non-null: `rows` = n_img * P prefix tokens of images img0.. (multi-token causal attention)

static int run_block(rqamd_rqt* h, RqtLayer& L, float* x_in, float* x, Pending& pend, const float* addvec, int rows,
                     const int* step, int step_off, int t_max, int Tcap, hipStream_t st, const PrefillCtx* pf = nullptr) {
    const int E = h->E;
    ResidLnArgs r{};
    r.x_in = x_in; r.slabs = pend.slabs; r.n_slabs = pend.n; r.bias = pend.bias; r.addvec = addvec;
    r.x_out = (x_in != x || pend.slabs || pend.bias || addvec) ? x : nullptr;      // nothing to add in place: the stream is not rewritten
    r.gamma = L.ln1w; r.beta = L.ln1b; r.y = h->y; r.rows = rows; r.E = E; r.eps = 1e-5f;
    RQ_TRY(rq_launch_resid_ln(r, st));
    RQ_TRY(step_gemm(h, h->y, E, L.wqkv, rows, 3 * E, E, EPI_BF16, L.bqkv, nullptr, 0, h->qkv, 3 * E, nullptr, st));
    if (pf) {
        AttnPrefillArgs ap{};
        const long img_stride = (long)E * Tcap;
        ap.qkv = h->qkv; ap.y = h->ya;
        ap.vc = L.vsc ? (bf16_t*)((unsigned char*)L.vc + pf->img0 * img_stride) : L.vc + pf->img0 * img_stride;
        ap.vsc = L.vsc ? L.vsc + (long)pf->img0 * L.nh * Tcap : nullptr;
        // (8-bit keys: a key is 64 bytes, i.e. half the bf16 stride, and has one scale)
        ap.kc = L.ksc ? (bf16_t*)((unsigned char*)L.kc + pf->img0 * img_stride) : L.kc + pf->img0 * img_stride;
        ap.ksc = L.ksc ? L.ksc + (long)pf->img0 * L.nh * Tcap : nullptr;
        ap.n_img = pf->n_img; ap.P = pf->P; ap.nh = L.nh; ap.E = E; ap.Tcap = Tcap;
        RQ_TRY(rq_launch_attn_prefill(ap, st));
    } else {
        AttnDecodeArgs at{};
        at.qkv = h->qkv; at.kc = L.kc; at.vc = L.vc; at.ksc = L.ksc; at.vsc = L.vsc; at.y = h->ya; at.step = step; at.step_off = step_off;
        at.t_max = t_max; at.rows = rows; at.nh = L.nh; at.E = E; at.Tcap = Tcap;
        GemmProfile& pfl = h->prof;
        if (pfl.on) {
            if (pfl.used_attn + 2 > pfl.ev_attn.size())
                for (int i = 0; i < 2; ++i) { hipEvent_t e; RQ_HIP(hipEventCreate(&e)); pfl.ev_attn.push_back(e); }
            RQ_HIP(hipEventRecord(pfl.ev_attn[pfl.used_attn], st));
        }
        RQ_TRY(rq_launch_attn_decode(at, st));
        if (pfl.on) { RQ_HIP(hipEventRecord(pfl.ev_attn[pfl.used_attn + 1], st)); pfl.used_attn += 2; }
    }
    int ns = 1;
    RQ_TRY(step_gemm(h, h->ya, E, L.wproj, rows, E, E, EPI_F32_PARTIAL, nullptr, nullptr, 0, h->slabs, E, &ns, st, x, L.bproj));
    ResidLnArgs r2{};
    r2.x_in = x; r2.x_out = ns ? x : nullptr; r2.slabs = ns ? h->slabs : nullptr; r2.n_slabs = ns; r2.bias = ns ? L.bproj : nullptr;
    r2.gamma = L.ln2w; r2.beta = L.ln2b; r2.y = h->y; r2.rows = rows; r2.E = E; r2.eps = 1e-5f;
    RQ_TRY(rq_launch_resid_ln(r2, st));
    RQ_TRY(step_gemm(h, h->y, E, L.wfc1, rows, 4 * E, E, EPI_BF16_GELU, L.bfc1, nullptr, 0, h->hbuf, 4 * E, nullptr, st));
    RQ_TRY(step_gemm(h, h->hbuf, 4 * E, L.wfc2, rows, E, 4 * E, EPI_F32_PARTIAL, nullptr, nullptr, 0, h->slabs, E, &ns, st, x, L.bfc2));
    if (ns) { pend.slabs = h->slabs; pend.n = ns; pend.bias = L.bfc2; }
    else pend = Pending{nullptr, 0, nullptr};      // the stream already holds this block's output
    return RQAMD_OK;
}

struct StepCtx {
    int B;
    const float* const* codebooks;
    float temperature;
    const int* top_k;
    const float* top_p;
    bool sample;           // run the sampler (else teacher-forced)
    float* logits_out;     // teacher-forced: (B,HW,D,V)
    float* cond_logits_out; // teacher-forced, text-conditioned: (B, cond_len-1, vocab_size_cond) or null
};

// body stack for the token whose input is already in h->x; leaves the last fc2 un-reduced in `pend`
// t_max: host-side bound on the number of cached keys (selects the attention kernel's register-block count)
static int body_stack(rqamd_rqt* h, int rows, const int* step, int step_off, int t_max, Pending& pend, hipStream_t st) {
    pend = Pending{nullptr, 0, nullptr};
    h->cur_gelu_v2 = h->cfg.gelu_v2 == 1 || h->cfg.gelu_v2 == 3;
    for (auto& L : h->body) RQ_TRY(run_block(h, L, h->x, h->x, pend, nullptr, rows, step, step_off, t_max, h->Tbody, st));
    return RQAMD_OK;
}

static int embed_gemm(rqamd_rqt* h, const StepCtx& c, int pos_off, int depth_lo, int n_depth, const bf16_t* W, const float* bias_tab,
                      int bias_row_off, bool bias_by_pos, float* out, hipStream_t st) {
    EmbedTokArgs e{};
    e.xs = h->xs; e.pos = h->st; e.pos_off = pos_off; e.depth_lo = depth_lo; e.n_depth = n_depth; e.rows = c.B; e.HW = h->HW; e.D = h->D; e.dim = h->Din;
    e.out = h->ain;
    for (int d = 0; d < h->D; ++d) { e.cb[d] = c.codebooks[d]; e.K[d] = h->Vd[d]; }
    RQ_TRY(rq_launch_embed_tokens(e, st));
    return step_gemm(h, h->ain, h->Din, W, c.B, h->E, h->Din, EPI_F32, bias_tab + (long)bias_row_off * h->E,
                     bias_by_pos ? h->st : nullptr, h->E, out, h->E, nullptr, st);
}

// learned token embeddings (tok_emb) summed over depths [d_lo, d_hi) of the codes at position *st + pos_off, + a positional row
static int tok_embed(rqamd_rqt* h, const StepCtx& c, int pos_off, int d_lo, int d_hi, const float* add, bool add_by_pos, int add_row,
                     float* out, hipStream_t st) {
    TokEmbedArgs t{};
    t.xs = h->xs; t.table = h->tok_emb; t.pos = h->st; t.pos_off = pos_off; t.d_lo = d_lo; t.d_hi = d_hi; t.add = add;
    t.add_by_pos = add_by_pos ? 1 : 0; t.add_row = add_row; t.rows = c.B; t.HW = h->HW; t.D = h->D; t.E = h->E; t.out = out;
    for (int d = 0; d < h->D; ++d) { t.offs[d] = h->tok_offs[d]; t.V[d] = h->Vd[d]; }
    return rq_launch_tok_embed(t, st);
}

// body half of one spatial position (position read from h->st[0] on the device): token embedding + body stack; the last
// fc2 stays un-reduced in `pend`
static int position_body(rqamd_rqt* h, const StepCtx& c, bool first_pos, int host_pos, Pending& pend, hipStream_t st) {
    const int E = h->E, B = c.B;
    if (first_pos) {
        RQ_TRY(rq_launch_cond_embed(h->cond, h->cond_len, h->cond_len - 1, h->cond_emb, h->cfg.vocab_size_cond < 1 ? 1 : h->cfg.vocab_size_cond,
                                    h->pos_cond, h->x, B, E, st));
    } else {
        // token = sum_d input_mlp(e_d) + pos_emb_hw[pos-1]  (transformers.py:218-225), or sum_d tok_emb(code_d) + pos_emb_hw[pos-1]
        if (h->in_vq) RQ_TRY(embed_gemm(h, c, -1, 0, h->D, h->w_in, h->body_in_bias, -1, true, h->x, st));
        else RQ_TRY(tok_embed(h, c, -1, 0, h->D, h->pos_hw, true, -1, h->x, st));
    }
    // a captured graph serves every position of the same 8-key bucket: bound t by the bucket's last position
    return body_stack(h, B, h->st, h->cond_len - 1, ((host_pos + h->cond_len - 1) | 7), pend, st);
}

// head half: depth d of the position -- head stack, classifier, then the sampler / the teacher-forced copy / nothing
// (stepping form: the caller reads h->logits).  `pend`: the body's pending fc2 (used by d == 0 only).
static int position_depth(rqamd_rqt* h, const StepCtx& c, int d, const Pending& pend, int host_pos, bool mask_only, hipStream_t st) {
    const int E = h->E, B = c.B;
    Pending hp;
    const float* addvec = nullptr;
    float* x_in = h->xh;
    if (d == 0) {
        // head token 0 = spatial context + pos_emb_d[0]; the context is body x + last fc2 (+bias)
        hp = pend; addvec = h->pos_d; x_in = h->x;
    } else {
        // head token d = head_mlp(cumsum_{j<d} e_j) + pos_emb_d[d]  (transformers.py:249-267); without cumsum_depth_ctx only
        // e_{d-1}; with head_emb_vqvae off tok_emb(code_{d-1}) + pos_emb_d[d]
        if (h->head_vq) RQ_TRY(embed_gemm(h, c, 0, h->cumsum ? 0 : d - 1, d, h->w_headin, h->head_in_bias, d, false, h->xh, st));
        else RQ_TRY(tok_embed(h, c, 0, d - 1, d, h->pos_d, false, d, h->xh, st));
        hp = Pending{nullptr, 0, nullptr};
    }
    h->cur_gelu_v2 = h->cfg.gelu_v2 == 1 || h->cfg.gelu_v2 == 2;
    for (size_t li = 0; li < h->head.size(); ++li) {
        RQ_TRY(run_block(h, h->head[li], li == 0 ? x_in : h->xh, h->xh, hp, li == 0 ? addvec : nullptr, B, nullptr, d, d, h->D, st));
    }
    ResidLnArgs r{};
    r.x_in = h->xh; r.x_out = nullptr; r.slabs = hp.slabs; r.n_slabs = hp.n; r.bias = hp.bias;
    if (h->head.empty()) { r.x_in = x_in; r.addvec = addvec; }     // no head stack: the classifier sees the head token itself
    r.gamma = h->cls_lnw; r.beta = h->cls_lnb; r.y = h->y; r.rows = B; r.E = E; r.eps = 1e-5f;
    RQ_TRY(rq_launch_resid_ln(r, st));
    // shared classifier, or BatchLinear's matrix of this depth
    const long cls_off = h->shared_cls ? 0 : (long)d * h->V;
    RQ_TRY(step_gemm(h, h->y, E, h->w_cls + cls_off * E, B, h->V, E, EPI_F32, h->b_cls + cls_off, nullptr, 0, h->logits, h->V, nullptr, st));
    if (c.sample || mask_only) {
        // LogitMask (primitives.py:78-93): codes beyond this depth's vocabulary cannot be drawn.  (The reference's teacher-
        // forced logits are NOT masked -- its mask indexes the wrong axis there -- so rqamd_rqt_logits leaves them alone.)
        if (h->Vd[d] < h->V) RQ_TRY(rq_launch_mask_logits(h->logits, B, h->V, h->Vd[d], st));
    }
    if (c.sample) {
        SampleArgs s{};
        s.logits = h->logits; s.rows = B; s.V = h->V; s.temperature = c.temperature; s.top_k = c.top_k[d]; s.top_p = c.top_p[d];
        s.redo = h->smp_redo; s.rng = h->rng; s.pos = h->st; s.d = d; s.D = h->D; s.out = h->xs; s.out_stride = (long)h->HW * h->D;
        RQ_TRY(rq_launch_sample(s, st));
    } else if (c.logits_out) {
        float* dst = c.logits_out + ((long)host_pos * h->D + d) * h->V;
        RQ_HIP(hipMemcpy2DAsync(dst, (size_t)h->HW * h->D * h->V * 4, h->logits, (size_t)h->V * 4, (size_t)h->V * 4, B,
                                hipMemcpyDeviceToDevice, st));
    }
    return RQAMD_OK;
}

// everything that happens at one spatial position (position read from h->st[0] on the device)
static int position_sequence(rqamd_rqt* h, const StepCtx& c, bool first_pos, bool do_head, int host_pos, hipStream_t st) {
    Pending pend;
    RQ_TRY(position_body(h, c, first_pos, host_pos, pend, st));
    if (!do_head) return RQAMD_OK;                 // (the next position overwrites h->x)
    for (int d = 0; d < h->D; ++d) RQ_TRY(position_depth(h, c, d, pend, host_pos, false, st));
    return RQAMD_OK;
}

// inputs into the workspace, position counter to 0, conditioning prefix through the body stack
static int begin_batch(rqamd_rqt* h, const StepCtx& c, const int64_t* partial, const int64_t* cond, hipStream_t st) {
    const int B = c.B;
    h->step_on = false;                            // any new batch ends a stepping sequence (same workspace)
    RQ_TRY(ensure_batch(h, B));
    RQ_TRY(finalize_tables(h, st));
    RQ_HIP(hipMemcpyAsync(h->xs, partial, (size_t)B * h->HW * h->D * 8, hipMemcpyDeviceToDevice, st));
    if (cond) RQ_HIP(hipMemcpyAsync(h->cond, cond, (size_t)B * h->cond_len * 8, hipMemcpyDeviceToDevice, st));
    else RQ_HIP(hipMemsetAsync(h->cond, 0, (size_t)B * h->cond_len * 8, st));
    RQ_TRY(rq_launch_set_int(h->st, 0, st));
    // Conditioning prefix (text tokens): tokens 0..cond_len-2 only fill the body KV cache (transformers.py:235-239; in
    // forward() their body outputs also feed cond_classifier, :150-153).  All P = cond_len-1 tokens of a chunk of images go
    // through the body stack in ONE pass (GEMMs over images x P rows, causal attention inside the prefix), the way the
    // reference's first cached step does -- not as P sequential single-token steps.
    if (h->cond_len > 1) {
        const int P = h->cond_len - 1, pc = prefill_chunk(h, B), vc = h->cfg.vocab_size_cond < 1 ? 1 : h->cfg.vocab_size_cond;
        for (int b0 = 0; b0 < B; b0 += pc) {
            PrefillCtx pf{b0, B - b0 < pc ? B - b0 : pc, P};
            const int rows = pf.n_img * P;
            RQ_TRY(rq_launch_cond_embed_multi(h->cond + (long)b0 * h->cond_len, h->cond_len, P, h->cond_emb, vc, h->pos_cond, h->x, pf.n_img, h->E, st));
            Pending pend{nullptr, 0, nullptr};
            h->cur_gelu_v2 = h->cfg.gelu_v2 == 1 || h->cfg.gelu_v2 == 3;
            for (auto& L : h->body) RQ_TRY(run_block(h, L, h->x, h->x, pend, nullptr, rows, nullptr, 0, 0, h->Tbody, st, &pf));
            if (c.cond_logits_out) {
                if (!h->w_ccls || h->n_ccls_seen < 4) return rq_fail(RQAMD_ERR_STATE, "rqt: cond_classifier parameters not set");
                ResidLnArgs r{};
                r.x_in = h->x; r.x_out = nullptr; r.slabs = pend.slabs; r.n_slabs = pend.n; r.bias = pend.bias;
                r.gamma = h->ccls_lnw; r.beta = h->ccls_lnb; r.y = h->y; r.rows = rows; r.E = h->E; r.eps = 1e-5f;
                RQ_TRY(rq_launch_resid_ln(r, st));
                RQ_TRY(step_gemm(h, h->y, h->E, h->w_ccls, rows, vc, h->E, EPI_F32, h->b_ccls, nullptr, 0,
                                 c.cond_logits_out + (long)b0 * P * vc, vc, nullptr, st));
            }
        }
    }
    return RQAMD_OK;
}

static int run_all(rqamd_rqt* h, const StepCtx& c, const int64_t* partial, const int64_t* cond, int start_idx, bool use_graph,
                   int64_t* codes_out, hipStream_t st) {
    const int B = c.B;
    RQ_TRY(begin_batch(h, c, partial, cond, st));
    for (int pos = 0; pos < h->HW; ++pos) {
        const bool do_head = pos >= start_idx;
        const bool graphable = use_graph && c.sample && do_head && pos >= 1 && !h->prof.on;
        if (graphable) {
            if (!h->gvalid) {
                for (auto& g : h->gexec) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
                h->gvalid = true;
            }
            int bucket = (pos + h->cond_len - 1) >> 3;
            if (bucket >= rqamd_rqt::NGRAPH) bucket = rqamd_rqt::NGRAPH - 1;
            if (!h->gexec[bucket]) {
                hipGraph_t g = nullptr;
                hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) {
                    int rc = position_sequence(h, c, false, true, pos, st);
                    if (rc == RQAMD_OK) rc = rq_launch_add_int(h->st, 1, st);
                    hipError_t e2 = hipStreamEndCapture(st, &g);
                    if (rc != RQAMD_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
                    if (e2 != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e2));
                    e2 = hipGraphInstantiate(&h->gexec[bucket], g, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(g);
                    if (e2 != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
                } else {
                    // capture unavailable (e.g. the legacy default stream): eager launches, ~10x the launch count.
                    // Said once, loudly -- this is a performance cliff, not an error.
                    static bool warned = false;
                    if (!warned) { fprintf(stderr, "librqamd: hipStreamBeginCapture failed (%s); sampling runs with eager launches\n", hipGetErrorString(e)); warned = true; }
                    (void)hipGetLastError();
                    use_graph = false;
                }
            }
            if (h->gexec[bucket]) {
                RQ_HIP(hipGraphLaunch(h->gexec[bucket], st));
                continue;
            }
        }
        RQ_TRY(position_sequence(h, c, pos == 0, do_head, pos, st));
        RQ_TRY(rq_launch_add_int(h->st, 1, st));
    }
    if (codes_out) RQ_HIP(hipMemcpyAsync(codes_out, h->xs, (size_t)B * h->HW * h->D * 8, hipMemcpyDeviceToDevice, st));
    return RQAMD_OK;
}

extern "C" int rqamd_rqt_sample(rqamd_rqt* h, const int64_t* partial, const int64_t* cond, int batch,
                                const float* const* codebooks, int start_h, int start_w, float temperature,
                                const int* top_k, const float* top_p, uint64_t seed, uint64_t offset,
                                int use_graph, int64_t* codes_out, void* stream) {
    if (!h || !partial || !codebooks || !top_k || !top_p || !codes_out) return rq_fail(RQAMD_ERR_INVALID, "rqt_sample: null argument");
    if (batch < 1) return rq_fail(RQAMD_ERR_INVALID, "rqt_sample: batch < 1");
    if (!(temperature > 0.f)) return rq_fail(RQAMD_ERR_INVALID, "rqt_sample: temperature must be > 0");
    hipStream_t st = (hipStream_t)stream;
    h->step_on = false;
    RQ_TRY(ensure_batch(h, batch));
    StepCtx c{};
    c.B = batch; c.codebooks = codebooks; c.temperature = temperature; c.top_k = top_k; c.top_p = top_p; c.sample = true;
    // graph cache key: anything baked into kernel arguments
    rqamd_rqt::Key k{};
    k.B = batch; k.T = temperature; k.stream = stream;
    for (int d = 0; d < h->D; ++d) { k.tk[d] = top_k[d]; k.tp[d] = top_p[d]; k.cb[d] = codebooks[d]; }
    if (!h->gvalid || memcmp(&k, &h->gkey, sizeof(k)) != 0) { h->gvalid = false; h->gkey = k; }
    RQ_LAUNCH(set_rng_kernel, dim3(1), dim3(64), 0, st, h->rng, seed, offset);
    h->prof.used = 0; h->prof.bytes = 0; h->prof.flops = 0; h->prof.used_attn = 0;
    int start_idx = start_h * h->cfg.W + start_w;
    if (start_idx < 0) start_idx = 0;
    RQ_TRY(run_all(h, c, partial, cond, start_idx, use_graph != 0, codes_out, st));
    if (h->prof.on) {
        RQ_HIP(hipStreamSynchronize(st));
        double ms = 0;
        for (size_t i = 0; i + 1 < h->prof.used; i += 2) {
            float t = 0.f;
            RQ_HIP(hipEventElapsedTime(&t, h->prof.ev[i], h->prof.ev[i + 1]));
            ms += t;
        }
        h->prof.ms_total = ms;
        h->prof.launches = (int64_t)(h->prof.used / 2);
        ms = 0;
        for (size_t i = 0; i + 1 < h->prof.used_attn; i += 2) {
            float t = 0.f;
            RQ_HIP(hipEventElapsedTime(&t, h->prof.ev_attn[i], h->prof.ev_attn[i + 1]));
            ms += t;
        }
        h->prof.attn_ms_total = ms;
        h->prof.attn_launches = (int64_t)(h->prof.used_attn / 2);
    }
    return RQAMD_OK;
}

extern "C" int rqamd_rqt_logits(rqamd_rqt* h, const int64_t* codes, const int64_t* cond, int batch,
                                const float* const* codebooks, float* logits_out, void* stream) {
    if (!h || !codes || !codebooks || !logits_out) return rq_fail(RQAMD_ERR_INVALID, "rqt_logits: null argument");
    if (batch < 1) return rq_fail(RQAMD_ERR_INVALID, "rqt_logits: batch < 1");
    StepCtx c{};
    c.B = batch; c.codebooks = codebooks; c.temperature = 1.f; c.sample = false; c.logits_out = logits_out;
    return run_all(h, c, codes, cond, 0, false, nullptr, (hipStream_t)stream);
}

extern "C" int rqamd_rqt_forward(rqamd_rqt* h, const int64_t* codes, const int64_t* cond, int batch,
                                 const float* const* codebooks, float* logits_out, float* cond_logits_out, void* stream) {
    if (!h || !codes || !codebooks || !logits_out) return rq_fail(RQAMD_ERR_INVALID, "rqt_forward: null argument");
    if (batch < 1) return rq_fail(RQAMD_ERR_INVALID, "rqt_forward: batch < 1");
    if (cond_logits_out && h->cond_len <= 1) return rq_fail(RQAMD_ERR_INVALID, "rqt_forward: cond_logits need block_size_cond > 1");
    StepCtx c{};
    c.B = batch; c.codebooks = codebooks; c.temperature = 1.f; c.sample = false; c.logits_out = logits_out; c.cond_logits_out = cond_logits_out;
    return run_all(h, c, codes, cond, 0, false, nullptr, (hipStream_t)stream);
}

// ---- lanes (round 6): a second handle that runs on the SAME parameter arena as `src` -- its own workspace, KV caches, graphs and
// stepping state, no weights of its own.  Two handles fed half a batch each on two streams run two independent decode chains
// over one copy of the weights (L2 / MALL hits for whichever chain reaches a layer second).
extern "C" int rqamd_dbg_rqt_share_params(rqamd_rqt* dst, const rqamd_rqt* src) {
    if (!dst || !src) return rq_fail(RQAMD_ERR_INVALID, "rqt_share_params: null argument");
    if (memcmp(&dst->cfg, &src->cfg, sizeof(dst->cfg)) != 0 || (!dst->head.empty() && dst->head[0].nh != src->head[0].nh))
        return rq_fail(RQAMD_ERR_INVALID, "rqt_share_params: configurations differ");
    if (src->seen.size() - src->n_ccls_seen < src->n_required || src->tables_dirty)
        return rq_fail(RQAMD_ERR_STATE, "rqt_share_params: the source handle has not run yet (parameters incomplete or tables not derived)");
    auto cp = [](std::vector<RqtLayer>& d, const std::vector<RqtLayer>& s) {
        for (size_t i = 0; i < d.size(); ++i) {
            RqtLayer keep = d[i];
            d[i] = s[i];
            d[i].kc = keep.kc; d[i].vc = keep.vc; d[i].ksc = keep.ksc; d[i].vsc = keep.vsc;
        }
    };
    cp(dst->body, src->body); cp(dst->head, src->head);
    dst->w_in = src->w_in; dst->w_headin = src->w_headin; dst->w_cls = src->w_cls;
    dst->b_in = src->b_in; dst->b_headin = src->b_headin; dst->b_cls = src->b_cls; dst->cls_lnw = src->cls_lnw; dst->cls_lnb = src->cls_lnb;
    dst->w_ccls = src->w_ccls; dst->b_ccls = src->b_ccls; dst->ccls_lnw = src->ccls_lnw; dst->ccls_lnb = src->ccls_lnb; dst->n_ccls_seen = src->n_ccls_seen;
    dst->cond_emb = src->cond_emb; dst->pos_cond = src->pos_cond; dst->pos_hw = src->pos_hw; dst->pos_d = src->pos_d; dst->tok_emb = src->tok_emb;
    dst->body_in_bias = src->body_in_bias; dst->head_in_bias = src->head_in_bias;
    dst->seen = src->seen; dst->tables_dirty = false; dst->gvalid = false;
    dst->arena.release();
    return RQAMD_OK;
}

// ---- stepping form: the caller draws the samples (include/rqamd.h)
extern "C" int rqamd_rqt_step_begin(rqamd_rqt* h, const int64_t* partial, const int64_t* cond, int batch, const float* const* codebooks, void* stream) {
    if (!h || !partial || !codebooks) return rq_fail(RQAMD_ERR_INVALID, "rqt_step_begin: null argument");
    if (batch < 1) return rq_fail(RQAMD_ERR_INVALID, "rqt_step_begin: batch < 1");
    h->step_on = false;
    for (int d = 0; d < h->D; ++d) h->step_cb[d] = codebooks[d];
    StepCtx c{};
    c.B = batch; c.codebooks = h->step_cb; c.temperature = 1.f; c.sample = false;
    RQ_TRY(begin_batch(h, c, partial, cond, (hipStream_t)stream));
    h->step_on = true; h->step_B = batch; h->step_pos = 0; h->step_d = 0;
    return RQAMD_OK;
}

extern "C" int rqamd_rqt_step_logits(rqamd_rqt* h, int pos, int d, const float** logits_dev, void* stream) {
    if (!h || !h->step_on) return rq_fail(RQAMD_ERR_STATE, "rqt_step_logits: no rqamd_rqt_step_begin");
    const bool body_only = d < 0;
    if (pos != h->step_pos || (body_only ? h->step_d != 0 : d != h->step_d) || pos >= h->HW)
        return rq_fail(RQAMD_ERR_INVALID, "rqt_step_logits: step (%d, %d) out of order, expected (%d, %d)", pos, d, h->step_pos, h->step_d);
    if (h->cap < h->step_B) return rq_fail(RQAMD_ERR_STATE, "rqt_step_logits: the workspace was re-sized by another call");
    hipStream_t st = (hipStream_t)stream;
    StepCtx c{};
    c.B = h->step_B; c.codebooks = h->step_cb; c.temperature = 1.f; c.sample = false;
    if (body_only || d == 0) {
        RQ_TRY(rq_launch_set_int(h->st, pos, st));
        Pending pend;
        RQ_TRY(position_body(h, c, pos == 0, pos, pend, st));
        h->step_pend_slabs = pend.slabs; h->step_pend_n = pend.n; h->step_pend_bias = pend.bias;
    }
    if (body_only) { h->step_pos = pos + 1; h->step_d = 0; return RQAMD_OK; }
    if (!logits_dev) return rq_fail(RQAMD_ERR_INVALID, "rqt_step_logits: null argument");
    const Pending pend{h->step_pend_slabs, h->step_pend_n, h->step_pend_bias};
    RQ_TRY(position_depth(h, c, d, pend, pos, true, st));
    *logits_dev = h->logits;
    if (d + 1 < h->D) h->step_d = d + 1;
    else { h->step_d = 0; h->step_pos = pos + 1; }
    return RQAMD_OK;
}

__global__ void set_codes_kernel(int64_t* xs, const int64_t* codes, int rows, long stride, int slot) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) xs[(long)r * stride + slot] = codes[r];
}

extern "C" int rqamd_rqt_step_set_code(rqamd_rqt* h, int pos, int d, const int64_t* codes, void* stream) {
    if (!h || !h->step_on || !codes) return rq_fail(RQAMD_ERR_STATE, "rqt_step_set_code: no step in progress / null argument");
    if (pos < 0 || pos >= h->HW || d < 0 || d >= h->D) return rq_fail(RQAMD_ERR_INVALID, "rqt_step_set_code: step (%d, %d)", pos, d);
    RQ_LAUNCH(set_codes_kernel, dim3((unsigned)((h->step_B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->xs, codes, h->step_B,
              (long)h->HW * h->D, pos * h->D + d);
    return rq_check_launch("set_codes_kernel");
}

extern "C" int rqamd_rqt_step_end(rqamd_rqt* h, int64_t* codes_out, void* stream) {
    if (!h || !h->step_on) return rq_fail(RQAMD_ERR_STATE, "rqt_step_end: no step in progress");
    h->step_on = false;
    if (codes_out) RQ_HIP(hipMemcpyAsync(codes_out, h->xs, (size_t)h->step_B * h->HW * h->D * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return RQAMD_OK;
}

extern "C" int rqamd_rqt_set_profile(rqamd_rqt* h, int profile) {
    if (!h) return rq_fail(RQAMD_ERR_INVALID, "null handle");
    h->prof.on = profile == 1;
    if (h->prof.skip_gemm != (profile == 2)) h->gvalid = false;      // the captured graphs hold (or lack) the GEMM nodes
    h->prof.skip_gemm = profile == 2;
    return RQAMD_OK;
}
extern "C" int rqamd_rqt_get_profile_attn(rqamd_rqt* h, double* ms, int64_t* launches) {
    if (!h) return rq_fail(RQAMD_ERR_INVALID, "null handle");
    if (ms) *ms = h->prof.attn_ms_total;
    if (launches) *launches = h->prof.attn_launches;
    return RQAMD_OK;
}
extern "C" int rqamd_rqt_get_profile(rqamd_rqt* h, double* ms, int64_t* launches, double* bytes, double* flops) {
    if (!h) return rq_fail(RQAMD_ERR_INVALID, "null handle");
    if (ms) *ms = h->prof.ms_total;
    if (launches) *launches = h->prof.launches;
    if (bytes) *bytes = h->prof.bytes;
    if (flops) *flops = h->prof.flops;
    return RQAMD_OK;
}
