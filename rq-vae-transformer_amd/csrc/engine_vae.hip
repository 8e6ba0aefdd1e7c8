// engine_vae.hip -- RQ-VAE encoder / decoder engine (host side, gfx950).
//
// Stands behind RQVAE.encode / RQVAE.decode (rqvae/models/rqvae/rqvae.py:80-89), Encoder.forward /
// Decoder.forward (modules.py:73-98,171-202), ResnetBlock._forward (layers.py:100-120),
// AttnBlock.forward (:158-182), Upsample / Downsample (:31-35,50-57).
//
// Layout and precision: activations NHWC bf16 between layers, fp32 accumulation, GroupNorm statistics
// in fp32; every convolution is the implicit-GEMM MFMA kernel of gemm.h with bias / residual fused in
// the epilogue; nearest-2x upsampling and the (0,1,0,1) zero pad of the stride-2 downsample are folded
// into the conv's gather.  The drivers decode one image per call (main_sampling_fid.py:223,
// measure_throughput/__main__.py:299-300); here any batch is accepted and processed in chunks that keep
// the five activation buffers bounded.
#include <string.h>
#include <stdlib.h>
#include <map>
#include <memory>
#include <string>
#include "gemm.h"
#include "rq_common.h"
#include "rqt_kernels.h"
#include "vae_kernels.h"

struct rqamd_vae {
    rqamd_vae_config cfg;
    std::map<std::string, std::unique_ptr<DevBuf>> params;
    DevBuf slab;               // fp32 split-K partial slabs (calls of <= SPLIT_MAX_B images; reserved on the first such call)
    DevBuf ws, part, gnp;      // gnp: [chunk][C][2] GroupNorm (scale, shift) for the fused norm->swish->conv
    DevBuf stage;              // two-phase calls: the <= 16^2 activation of a super-chunk of images between the two phases
    bool no_two_phase = false; // RQAMD_VAE_TWO_PHASE=0 (A/B switch)
    bool resamp_with_conv = true;      // ddconfig.resamp_with_conv (rqamd_vae_set_option): false = bare nearest upsample / average pool (layers.py:20-57)
    bf16_t* buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap_elems = 0;
    int chunk = 0;
    int chunk_max = 128;
    bool no_halo = false, no_fuse_gn = false, no_fuse_stats = false, no_halo_ups = false, no_splitk = false, no_ups_subpixel = false;
    bool halo_lowres = true;       // 32^2 layers take the halo kernel like the >= 64^2 ones (RQAMD_HALO_LOWRES=0: implicit GEMM there)
    static constexpr int SPLIT_MAX_B = 8;      // calls of at most this many images divide the K loop of low-resolution convs over workgroups
    std::string missing;
    // Small batches (the drivers decode ONE image per call, measure_throughput/__main__.py:297-299, main_sampling_fid.py:223;
    // the rFID loop encodes and decodes one image per call, rqvae/metrics/fid.py:167-169) are launch-bound: ~200 launches
    // per image.  Their launch sequence is captured once per (direction, batch) as a hipGraph over engine-owned input /
    // output staging buffers and replayed; the caller's tensors are copied in / out on the same stream.
    static constexpr int GRAPH_MAX_B = 4;
    struct Graph { hipGraphExec_t exec = nullptr; void* stream = nullptr; unsigned gen = 0; const void* io = nullptr; };
    Graph gdec[GRAPH_MAX_B], genc[GRAPH_MAX_B];
    DevBuf gio;                // [decode in | decode out | encode in | encode out] for GRAPH_MAX_B images
    unsigned gen = 0;          // bumped whenever the workspace is reallocated (captured pointers become stale)
    bool use_graph = true, graph_warned = false;
};

static int vae_level_res(const rqamd_vae_config& c, int level) { return c.resolution >> level; }
static bool vae_has_attn(const rqamd_vae_config& c, int res) {
    for (int i = 0; i < c.n_attn_res; ++i)
        if (c.attn_resolutions[i] == res) return true;
    return false;
}

extern "C" int rqamd_vae_create(const rqamd_vae_config* c, rqamd_vae** out) {
    if (!c || !out) return rq_fail(RQAMD_ERR_INVALID, "vae_create: null argument");
    if (c->n_levels < 1 || c->n_levels > 8 || c->n_attn_res < 0 || c->n_attn_res > 8) return rq_fail(RQAMD_ERR_INVALID, "vae_create: bad config");
    if (c->ch % 64 != 0 || c->z_channels % 64 != 0 || c->embed_dim % 64 != 0)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "vae_create: ch / z_channels / embed_dim must be multiples of 64");
    if (c->in_channels > 4 || c->out_ch > 4) return rq_fail(RQAMD_ERR_UNSUPPORTED, "vae_create: in/out channels > 4");
    if (c->resolution % (1 << (c->n_levels - 1)) != 0) return rq_fail(RQAMD_ERR_INVALID, "vae_create: resolution not divisible");
    rqamd_vae* h = new rqamd_vae();
    h->cfg = *c;
    if (const char* e = getenv("RQAMD_VAE_CHUNK")) { int v = atoi(e); if (v >= 1 && v <= 1024) h->chunk_max = v; }
    h->no_halo = getenv("RQAMD_NO_HALO") != nullptr;      // A/B switches (diagnostics)
    h->no_fuse_gn = getenv("RQAMD_NO_FUSE_GN") != nullptr;
    h->no_fuse_stats = getenv("RQAMD_NO_FUSE_STATS") != nullptr;
    h->no_halo_ups = getenv("RQAMD_NO_HALO_UPS") != nullptr;
    h->no_ups_subpixel = getenv("RQAMD_NO_UPS_SUBPIXEL") != nullptr;      // A/B switch: the 9-tap folded-upsample form everywhere
    h->no_splitk = getenv("RQAMD_VAE_NO_SPLITK") != nullptr;
    if (const char* e = getenv("RQAMD_HALO_LOWRES")) h->halo_lowres = atoi(e) != 0;
    if (const char* e = getenv("RQAMD_VAE_GRAPH")) h->use_graph = atoi(e) != 0;
    if (const char* e = getenv("RQAMD_VAE_TWO_PHASE")) h->no_two_phase = atoi(e) == 0;
    *out = h;
    return RQAMD_OK;
}

// options of a handle that are not part of rqamd_vae_config (ABI v7): "resamp_with_conv" = 0 | 1 (ddconfig.resamp_with_conv, modules.py:12,103;
// default 1, what every released config uses)
extern "C" int rqamd_vae_set_option(rqamd_vae* h, const char* name, int value) {
    if (!h || !name) return rq_fail(RQAMD_ERR_INVALID, "vae_set_option: null argument");
    if (strcmp(name, "resamp_with_conv") == 0) {
        if (value != 0 && value != 1) return rq_fail(RQAMD_ERR_INVALID, "vae_set_option(resamp_with_conv): %d", value);
        if (h->resamp_with_conv != (value != 0)) h->gen++;      // captured graphs hold the other layer sequence
        h->resamp_with_conv = value != 0;
        return RQAMD_OK;
    }
    return rq_fail(RQAMD_ERR_INVALID, "vae_set_option: unknown option %s", name);
}

extern "C" int rqamd_vae_destroy(rqamd_vae* h) {
    if (!h) return RQAMD_OK;
    for (auto& g : h->gdec) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (auto& g : h->genc) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    delete h;
    return RQAMD_OK;
}

static DevBuf* vae_slot(rqamd_vae* h, const std::string& name, size_t bytes) {
    auto& u = h->params[name];
    if (!u) u.reset(new DevBuf());
    if (u->reserve(bytes) != RQAMD_OK) return nullptr;
    return u.get();
}

extern "C" int rqamd_vae_set_param(rqamd_vae* h, const char* name, const float* src, const int64_t* shape, int ndim, void* stream) {
    if (!h || !name || !src || !shape) return rq_fail(RQAMD_ERR_INVALID, "vae_set_param: null argument");
    hipStream_t st = (hipStream_t)stream;
    std::string s(name);
    if (s.rfind("quantizer.", 0) == 0) return RQAMD_OK;      // codebooks travel separately (rqamd_rq_*)
    if (ndim == 4) {
        const int O = (int)shape[0], I = (int)shape[1], kh = (int)shape[2], kw = (int)shape[3];
        const size_t n = (size_t)O * I * kh * kw;
        if (s == "encoder.conv_in.weight" || s == "decoder.conv_out.weight") {
            DevBuf* d = vae_slot(h, s, n * 4);
            if (!d) return RQAMD_ERR_HIP;
            return rq_launch_repack_conv(src, d->p, O, I, kh, kw, s[0] == 'e' ? 1 : 2, st);
        }
        // attention q/k/v 1x1 convs are fused into one [3C][C] GEMM weight
        for (int which = 0; which < 3; ++which) {
            const char* leaf = which == 0 ? ".q.weight" : which == 1 ? ".k.weight" : ".v.weight";
            const size_t ll = strlen(leaf);
            if (s.size() > ll && s.compare(s.size() - ll, ll, leaf) == 0 && kh == 1) {
                std::string fused = s.substr(0, s.size() - ll) + ".qkv.weight";
                DevBuf* d = vae_slot(h, fused, 3 * n * 2);
                if (!d) return RQAMD_ERR_HIP;
                return rq_launch_repack_conv(src, d->as<bf16_t>() + which * n, O, I, 1, 1, 0, st);
            }
        }
        DevBuf* d = vae_slot(h, s, n * 2);
        if (!d) return RQAMD_ERR_HIP;
        RQ_TRY(rq_launch_repack_conv(src, d->p, O, I, kh, kw, 0, st));
        // Upsample.conv (layers.py:20-35): also the pre-summed weights of its sub-pixel form (four 2 x 2 convs over the source image,
        // conv_halo.hip), kept next to the 3 x 3 ones -- which form runs is decided per layer shape in VaeRun::conv
        static const char* up_leaf = ".upsample.conv.weight";
        const size_t ul = strlen(up_leaf);
        if (kh == 3 && kw == 3 && s.size() > ul && s.compare(s.size() - ul, ul, up_leaf) == 0) {
            DevBuf* ds = vae_slot(h, s + ".subpixel", (size_t)16 * O * I * 2);
            if (!ds) return RQAMD_ERR_HIP;
            RQ_TRY(rq_launch_ups_subpixel_weights(d->as<bf16_t>(), ds->as<bf16_t>(), O, I, st));
        }
        return RQAMD_OK;
    }
    if (ndim == 1) {
        const size_t n = (size_t)shape[0];
        for (int which = 0; which < 3; ++which) {
            const char* leaf = which == 0 ? ".q.bias" : which == 1 ? ".k.bias" : ".v.bias";
            const size_t ll = strlen(leaf);
            if (s.size() > ll && s.compare(s.size() - ll, ll, leaf) == 0) {
                std::string fused = s.substr(0, s.size() - ll) + ".qkv.bias";
                DevBuf* d = vae_slot(h, fused, 3 * n * 4);
                if (!d) return RQAMD_ERR_HIP;
                RQ_HIP(hipMemcpyAsync(d->as<float>() + which * n, src, n * 4, hipMemcpyDeviceToDevice, st));
                return RQAMD_OK;
            }
        }
        DevBuf* d = vae_slot(h, s, n * 4);
        if (!d) return RQAMD_ERR_HIP;
        RQ_HIP(hipMemcpyAsync(d->p, src, n * 4, hipMemcpyDeviceToDevice, st));
        return RQAMD_OK;
    }
    return rq_fail(RQAMD_ERR_INVALID, "vae_set_param(%s): unexpected rank %d", name, ndim);
}

// -------------------------------------------------------------------------------------------------
struct VaeRun {
    rqamd_vae* h;
    hipStream_t st;
    int B;
    int err = RQAMD_OK;
    bf16_t *X, *Y, *T1, *T2, *T3;
    // GroupNorm partials currently held in h->part: of which tensor, how many per image (0 = none)
    const bf16_t* stats_of = nullptr;
    int stats_n = 0;

    void* P(const std::string& name) {
        auto it = h->params.find(name);
        if (it == h->params.end() || !it->second || !it->second->p) {
            if (err == RQAMD_OK) err = rq_fail(RQAMD_ERR_STATE, "vae: parameter %s was never set", name.c_str());
            return nullptr;
        }
        return it->second->p;
    }
    void swap() { bf16_t* t = X; X = Y; Y = t; }
    // Halo-reuse kernel or implicit GEMM for a 3x3 / stride-1 layer.  The choice is a function of the LAYER only, never of the
    // batch: a different kernel means a different summation order, and an image's pixels must not depend on how many other
    // images shared its call (rows of a batched decode == the same rows decoded one by one, bit for bit -- the speculative
    // batching behind RQVAE.decode_code relies on it, tests/test_gpu_parity.py::test_vae_batch_invariance_and_chunking).
    // Always at >= 64^2; at 32^2 (four 8 x 32 tiles per image) as well: at 128 images 256 -> 256 takes ~175 us fused against 215 us
    // + 45 us of GroupNorm passes as implicit GEMM (profiles/r02_decode_timeline_b128.txt); a single image pays ~0.15 ms per
    // decode for it (8-16 workgroups per launch) against the split-K implicit GEMM that round 2 used there below 64 images.
    bool halo_here(int H, int W, int Cin, int Cout) const {
        if (h->no_halo || !rq_conv_halo_supported(H, W, Cin, Cout)) return false;
        return H >= 64 || h->halo_lowres;
    }
    // K split of a low-resolution implicit-GEMM conv: a function of the layer only (see halo_here).  kt K-tiles of 64 -> the
    // largest split count <= 16 that leaves an even number (>= 4) of K-tiles per chunk; 1 = no split.
    static int k_split(int kt) {
        for (int sk = 16; sk >= 2; --sk)
            if (kt % (2 * sk) == 0 && kt / sk >= 4) return sk;
        return 1;
    }

    // src: NHWC [B][Hin>>ups][Win>>ups][Cin]  ->  dst [B][Hout][Wout][Cout]
    void conv(const std::string& name, const bf16_t* src, void* dst, int Hin, int Win, int Cin, int Cout, int ks, int stride, int ups,
              int epi, const bf16_t* resid) {
        if (err) return;
        const bf16_t* w = (const bf16_t*)P(name + ".weight");
        const float* b = (const float*)P(name + ".bias");
        if (err) return;
        int Hout = Hin, Wout = Win, pad = ks / 2;
        if (stride == 2) { Hout = Hin / 2; Wout = Win / 2; pad = 0; }      // F.pad(0,1,0,1) + conv(s2, p0), layers.py:50-54
        // high-resolution 3x3 / stride-1 layers: halo-reuse kernel (one patch staged per 64-channel chunk, 9 taps)
        if (ks == 3 && stride == 1 && (!ups || (epi == EPI_BF16 && !h->no_halo_ups)) && (epi == EPI_BF16 || epi == EPI_BF16_RESID) &&
            halo_here(Hin, Win, Cin, Cout)) {
            const bool st_ok = !h->no_fuse_stats && (Cout == 128 || Cout == 256 || Cout == 512) && stat_fits(Hin, Win);
            // the upsample conv as four 2 x 2 convs over the source image (round 5: 4 taps per output pixel instead of 9) wherever the
            // source image has whole 8 x 32 tiles -- a function of the layer, like every other kernel choice here
            int form = ups;
            if (ups && !h->no_ups_subpixel && rq_conv_halo_subpixel_supported(Hin, Win, Cin, Cout)) {
                auto it = h->params.find(name + ".weight.subpixel");
                if (it != h->params.end() && it->second && it->second->p) { w = it->second->as<bf16_t>(); form = 2; }
            }
            err = rq_launch_conv_halo(src, w, b, nullptr, epi == EPI_BF16_RESID ? resid : nullptr, (bf16_t*)dst,
                                      st_ok ? h->part.as<float>() : nullptr, B, Hin, Win, Cin, Cout, form, st);
            stats_of = st_ok ? (const bf16_t*)dst : nullptr;
            stats_n = st_ok ? rq_conv_halo_stat_tiles(Hin, Win) : 0;
            return;
        }
        if ((const void*)stats_of == (const void*)dst) stats_of = nullptr;      // dst is overwritten without new statistics
        GemmArgs a{};
        a.A = src; a.W = w; a.M = B * Hout * Wout; a.N = Cout; a.K = ks * ks * Cin; a.lda = Cin;
        a.conv = (ks == 3 || stride != 1 || ups) ? 1 : 0;
        a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.ksize = ks; a.stride = stride; a.pad = pad; a.ups = ups;
        a.epi = epi; a.bias = b; a.out = dst; a.ldo = Cout; a.resid = resid; a.ldr = Cout; a.splitk = 1;
        int bm = a.M >= 128 ? 128 : 64;
        const int bn = (Cout % 128 == 0) ? 128 : 64;
        if (bn == 128 && (long)(a.M / 256) * (Cout / 128) >= 512) bm = 256;     // 8-wave tile for the big layers
        // A low-resolution conv with a long reduction (8x8, 512 -> 512: 72 K-tiles) is summed as sk chunks of K, each from zero,
        // added in chunk order -- for EVERY batch size, so that the result does not depend on it (see halo_here):
        //  * few images (a handful of output tiles: four workgroups walking 72 K-tiles each took 47 us): the chunks go to
        //    blockIdx.z as fp32 slabs and splitk_reduce finishes (slab sum in chunk order + bias + residual, one rounding);
        //  * many images: one workgroup per tile walks all chunks and folds its accumulators into a running total at the chunk
        //    boundaries (GemmArgs::vsplit) -- the same additions in the same order, no slabs.
        const int kt = a.K / 64;
        const int sk = (a.conv && Hout * Wout <= 1024 && kt >= 16 && Cout % 4 == 0 && !h->no_splitk &&
                        (epi == EPI_BF16 || epi == EPI_BF16_RESID)) ? k_split(kt) : 1;
        if (sk > 1 && B <= rqamd_vae::SPLIT_MAX_B) {
            const size_t need = (size_t)sk * a.M * Cout * 4;
            if (need > h->slab.bytes) { err = rq_fail(RQAMD_ERR_STATE, "vae: split-K slab of %zu bytes was not reserved", need); return; }
            a.epi = EPI_F32_PARTIAL; a.bias = nullptr; a.resid = nullptr; a.out = h->slab.p; a.splitk = sk;
            err = rq_gemm_launch(a, bm, bn, st);
            if (err) return;
            err = rq_launch_splitk_reduce(h->slab.as<float>(), sk, a.M, Cout, b, epi == EPI_BF16_RESID ? resid : nullptr, dst, 0, st);
            return;
        }
        a.vsplit = sk;
        err = rq_gemm_launch(a, bm, bn, st);
    }
    // Upsample / Downsample without their conv (resamp_with_conv = False): X -> Y, then swap
    void resample(bool up, int Hout, int Wout, int C) {
        if (err) return;
        err = up ? rq_launch_upsample2(X, Y, B, Hout, Wout, C, st) : rq_launch_avgpool2(X, Y, B, Hout, Wout, C, st);
        if ((const void*)stats_of == (const void*)Y) stats_of = nullptr;
        swap();
    }
    bool stat_fits(int H, int W) const { return (size_t)B * rq_conv_halo_stat_tiles(H, W) * 32 * 2 * 4 <= h->part.bytes; }
    void norm(const std::string& name, const bf16_t* src, bf16_t* dst, int HW, int C, int silu) {
        if (err) return;
        stats_of = nullptr;                          // the statistics pass below overwrites h->part
        const float* g = (const float*)P(name + ".weight");
        const float* b = (const float*)P(name + ".bias");
        if (err) return;
        err = rq_launch_groupnorm(src, dst, h->part.as<float>(), g, b, B, HW, C, silu, st);
    }
    // GroupNorm -> swish -> 3x3 conv (layers.py:104-106,112-115).  On the high-resolution layers the normalisation is
    // applied inside the halo conv while it stages the patch (no normalised copy of the activation in HBM: only
    // the statistics pass reads the tensor); elsewhere norm() writes tmp and the implicit-GEMM conv reads it.
    void norm_conv(const std::string& nname, const std::string& cname, const bf16_t* src, bf16_t* tmp, bf16_t* dst, int H, int W,
                   int Cin, int Cout, int epi, const bf16_t* resid) {
        if (err) return;
        if (!h->no_fuse_gn && halo_here(H, W, Cin, Cout)) {
            const float* g = (const float*)P(nname + ".weight");
            const float* be = (const float*)P(nname + ".bias");
            const bf16_t* w = (const bf16_t*)P(cname + ".weight");
            const float* b = (const float*)P(cname + ".bias");
            if (err) return;
            // statistics of src: left in h->part by the conv that produced it, else one read-only pass
            err = rq_launch_gn_params(src, h->part.as<float>(), g, be, h->gnp.as<float>(), B, H * W, Cin,
                                      stats_of == src ? stats_n : 0, st);
            if (err) return;
            const bool st_ok = !h->no_fuse_stats && (Cout == 128 || Cout == 256 || Cout == 512) && stat_fits(H, W);
            err = rq_launch_conv_halo(src, w, b, h->gnp.as<float>(), epi == EPI_BF16_RESID ? resid : nullptr, dst,
                                      st_ok ? h->part.as<float>() : nullptr, B, H, W, Cin, Cout, 0, st);
            stats_of = st_ok ? dst : nullptr;
            stats_n = st_ok ? rq_conv_halo_stat_tiles(H, W) : 0;
            return;
        }
        norm(nname, src, tmp, H * W, Cin, 1);
        conv(cname, tmp, dst, H, W, Cin, Cout, 3, 1, 0, epi, resid);
    }
    // ResnetBlock._forward (layers.py:100-120): X -> Y, then swap
    void res(const std::string& name, int H, int W, int Cin, int Cout) {
        norm_conv(name + ".norm1", name + ".conv1", X, T1, T2, H, W, Cin, Cout, EPI_BF16, nullptr);
        const bf16_t* sc = X;
        if (Cin != Cout) {
            conv(name + ".nin_shortcut", X, T3, H, W, Cin, Cout, 1, 1, 0, EPI_BF16, nullptr);
            sc = T3;
        }
        norm_conv(name + ".norm2", name + ".conv2", T2, T1, Y, H, W, Cout, Cout, EPI_BF16_RESID, sc);
        swap();
    }
    // AttnBlock.forward (layers.py:158-182): X -> Y, then swap
    void attn(const std::string& name, int H, int W, int C) {
        norm(name + ".norm", X, T1, H * W, C, 0);
        conv(name + ".qkv", T1, T2, H, W, C, 3 * C, 1, 1, 0, EPI_BF16, nullptr);
        if (err) return;
        err = rq_launch_vae_attn(T2, T1, B, H * W, C, st);
        conv(name + ".proj_out", T1, Y, H, W, C, C, 1, 1, 0, EPI_BF16_RESID, X);
        swap();
    }
};

static int vae_prepare(rqamd_vae* h, int chunk) {
    const rqamd_vae_config& c = h->cfg;
    size_t per_img = 0;
    for (int l = 0; l < c.n_levels; ++l) {
        size_t hw = (size_t)vae_level_res(c, l) * vae_level_res(c, l);
        size_t cl = (size_t)c.ch * c.ch_mult[l];
        size_t cn = (l + 1 < c.n_levels) ? (size_t)c.ch * c.ch_mult[l + 1] : cl;
        size_t cm = cl > cn ? cl : cn;
        if ((size_t)c.z_channels > cm) cm = c.z_channels;
        if (vae_has_attn(c, vae_level_res(c, l))) cm *= 3;
        if (hw * cm > per_img) per_img = hw * cm;
    }
    const int lowres = vae_level_res(c, c.n_levels - 1);
    size_t mid = (size_t)lowres * lowres * c.ch * c.ch_mult[c.n_levels - 1] * 3;
    if (mid > per_img) per_img = mid;
    if (chunk <= h->chunk && h->buf[0]) return RQAMD_OK;
    // failure-atomic regrowth: forget the old capacity before anything is freed, record the new one only after every
    // allocation succeeded (a failed hipMalloc leaves the handle empty -- chunk 0, no buffers -- never dangling)
    h->chunk = 0;
    h->cap_elems = 0;
    h->gen++;                                    // captured graphs point into the old workspace
    for (int i = 0; i < 5; ++i) h->buf[i] = nullptr;
    const size_t elems = per_img * chunk;
    const size_t bytes = (elems * 2 + 255) & ~(size_t)255;
    RQ_TRY(h->ws.reserve(bytes * 5));
    for (int i = 0; i < 5; ++i) h->buf[i] = (bf16_t*)((char*)h->ws.p + bytes * i);
    {   // GroupNorm partials: gn_stats uses <= RQ_GN_MAX_CHUNK per image, the conv epilogues one per 8x32 output tile
        size_t per_img_parts = RQ_GN_MAX_CHUNK;
        const size_t tiles = (size_t)(c.resolution / 4) * ((c.resolution + 31) / 32);     // 4 x 32 pixel tiles at most
        if (tiles > per_img_parts) per_img_parts = tiles;
        RQ_TRY(h->part.reserve((size_t)chunk * per_img_parts * 32 * 2 * 4));
    }
    RQ_TRY(h->gnp.reserve((size_t)chunk * 2048 * 2 * 4));
    h->cap_elems = elems;
    h->chunk = chunk;
    return RQAMD_OK;
}

// split-K slabs for a call of B <= SPLIT_MAX_B images: <= 16 chunks x (B images x 1024 pixels) x the widest layer, fp32 --
// 32 MiB per image for the ImageNet / FFHQ shape.  Reserved on the first small call (never inside a graph capture); engines that
// only ever run large batches do not pay for it.
static int vae_prepare_slab(rqamd_vae* h, int B) {
    if (B > rqamd_vae::SPLIT_MAX_B || h->no_splitk) return RQAMD_OK;
    const rqamd_vae_config& c = h->cfg;
    size_t cmax = 0;
    for (int l = 0; l < c.n_levels; ++l) if ((size_t)c.ch * c.ch_mult[l] > cmax) cmax = (size_t)c.ch * c.ch_mult[l];
    if ((size_t)c.z_channels * (c.double_z ? 2 : 1) > cmax) cmax = (size_t)c.z_channels * (c.double_z ? 2 : 1);
    const size_t need = (size_t)16 * B * 1024 * cmax * 4;
    if (need > h->slab.bytes) { RQ_TRY(h->slab.reserve(need)); h->gen++; }      // captured graphs point into the old slab
    return RQAMD_OK;
}

// Two-phase calls (round 6).  The layers at <= 16^2 (implicit-GEMM convs over a few hundred rows per image, single-head attention,
// GroupNorm passes) run at 0.18-0.32 of the MFMA peak on the 128 images of a chunk and 1.8 x faster on 256 and more
// (scripts/conv_lowres_tiles.py); the chunk itself cannot grow, it is what bounds the 256^2 tensors.  So a call of more than one chunk
// runs these layers ONCE over a super-chunk of up to eight chunks (their tensors are 1/64 of the 256^2 ones: the same five buffers
// hold them) and the >= 32^2 layers chunk by chunk, the 16^2 activation of the super-chunk parked in `stage` in between.  The cut
// sits where the producing conv is an implicit GEMM (no GroupNorm statistics travel across it); every kernel on either side
// computes an image's values independently of the others in its launch (section 3a of DESIGN.md), so the pixels / latents are
// bit-identical to the single-phase path.  phase 0: everything; 1: the first half, ends with the copy into `stage`; 2: the second
// half, starts from `stage`.
// first level (counted from the full resolution) whose side is <= 16; two phases need a level on either side of it
static int vae_lo_level(const rqamd_vae_config& c) {
    for (int l = 0; l < c.n_levels; ++l)
        if (vae_level_res(c, l) <= 16) return l;
    return c.n_levels;
}
static bool vae_two_phase(const rqamd_vae* h) {
    const int l = vae_lo_level(h->cfg);
    return !h->no_two_phase && l >= 1 && l <= h->cfg.n_levels - 1;
}

static int decode_chunk(rqamd_vae* h, const float* z_q, int B, float* out, hipStream_t st, int phase = 0, bf16_t* stage = nullptr) {
    const rqamd_vae_config& c = h->cfg;
    VaeRun r{h, st, B};
    r.X = h->buf[0]; r.Y = h->buf[1]; r.T1 = h->buf[2]; r.T2 = h->buf[3]; r.T3 = h->buf[4];
    const int nl = c.n_levels;
    const int l_lo = vae_lo_level(c);
    int res = vae_level_res(c, nl - 1);
    int block_in = c.ch * c.ch_mult[nl - 1];
    if (phase != 2) {
        // z_q NHWC fp32 -> bf16; post_quant_conv (1x1, rqvae.py:87); Decoder.conv_in
        RQ_TRY(rq_launch_cvt_bf16(z_q, r.T1, (long)B * res * res * c.embed_dim, st));
        r.conv("post_quant_conv", r.T1, r.X, res, res, c.embed_dim, c.z_channels, 1, 1, 0, EPI_BF16, nullptr);
        r.conv("decoder.conv_in", r.X, r.Y, res, res, c.z_channels, block_in, 3, 1, 0, EPI_BF16, nullptr);
        r.swap();
        r.res("decoder.mid.block_1", res, res, block_in, block_in);
        r.attn("decoder.mid.attn_1", res, res, block_in);
        r.res("decoder.mid.block_2", res, res, block_in, block_in);
    }
    for (int l = nl - 1; l >= 0; --l) {
        const int block_out = c.ch * c.ch_mult[l];
        const std::string up = "decoder.up." + std::to_string(l);
        const bool skip = phase == 2 && l >= l_lo;             // done by phase 1
        if (skip) {
            res = vae_level_res(c, l);
            block_in = block_out;
            if (l != l_lo) continue;                           // (its upsample conv ran in phase 1 as well)
            if (r.err) return r.err;
            RQ_HIP(hipMemcpyAsync(r.X, stage, (size_t)B * res * res * block_in * 2, hipMemcpyDeviceToDevice, st));
        } else {
            for (int ib = 0; ib < c.num_res_blocks + 1; ++ib) {
                r.res(up + ".block." + std::to_string(ib), res, res, block_in, block_out);
                block_in = block_out;
                if (vae_has_attn(c, res)) r.attn(up + ".attn." + std::to_string(ib), res, res, block_in);
            }
            if (phase == 1 && l == l_lo) {
                if (r.err) return r.err;
                if (r.stats_of == r.X) return rq_fail(RQAMD_ERR_STATE, "vae: two-phase cut behind a conv that leaves GroupNorm statistics");
                RQ_HIP(hipMemcpyAsync(stage, r.X, (size_t)B * res * res * block_in * 2, hipMemcpyDeviceToDevice, st));
                return RQAMD_OK;
            }
        }
        if (l != 0) {
            res *= 2;   // nearest x2 folded into the conv gather (layers.py:31-35)
            if (h->resamp_with_conv) {
                r.conv(up + ".upsample.conv", r.X, r.Y, res, res, block_in, block_in, 3, 1, 1, EPI_BF16, nullptr);
                r.swap();
            } else r.resample(true, res, res, block_in);
        }
    }
    const float* w = (const float*)r.P("decoder.conv_out.weight");
    const float* b = (const float*)r.P("decoder.conv_out.bias");
    if (r.err) return r.err;
    if (!h->no_halo && rq_conv_out_halo_supported(res, res, block_in, c.out_ch)) {
        // norm_out -> swish -> conv_out in one pass over the activation (plus the statistics pass)
        const float* g = (const float*)r.P("decoder.norm_out.weight");
        const float* be = (const float*)r.P("decoder.norm_out.bias");
        if (r.err) return r.err;
        RQ_TRY(rq_launch_gn_params(r.X, h->part.as<float>(), g, be, h->gnp.as<float>(), B, res * res, block_in,
                                   r.stats_of == r.X ? r.stats_n : 0, st));
        return rq_launch_conv_out_halo(r.X, w, b, h->gnp.as<float>(), out, B, res, res, block_in, c.out_ch, st);
    }
    r.norm("decoder.norm_out", r.X, r.T1, res * res, block_in, 1);
    if (r.err) return r.err;
    return rq_launch_conv_out3(r.T1, w, b, out, B, res, res, block_in, c.out_ch, st);
}

static int encode_chunk(rqamd_vae* h, const float* x, int B, float* z_e, hipStream_t st, int phase = 0, bf16_t* stage = nullptr) {
    const rqamd_vae_config& c = h->cfg;
    VaeRun r{h, st, B};
    r.X = h->buf[0]; r.Y = h->buf[1]; r.T1 = h->buf[2]; r.T2 = h->buf[3]; r.T3 = h->buf[4];
    const int nl = c.n_levels;
    const int l_lo = vae_lo_level(c);
    int res = c.resolution;
    if (phase != 2) {
        const float* w = (const float*)r.P("encoder.conv_in.weight");
        const float* b = (const float*)r.P("encoder.conv_in.bias");
        if (r.err) return r.err;
        if (!h->no_halo && rq_conv_in_mfma_supported(res, res, c.in_channels, c.ch)) {
            // (round 6) its epilogue leaves the GroupNorm partials of the first ResnetBlock's input, like the halo convs do
            const bool st_ok = !h->no_fuse_stats && r.stat_fits(res, res);
            RQ_TRY(rq_launch_conv_in_mfma(x, w, b, r.X, st_ok ? h->part.as<float>() : nullptr, B, res, res, st));
            r.stats_of = st_ok ? r.X : nullptr;
            r.stats_n = st_ok ? rq_conv_halo_stat_tiles(res, res) : 0;
        }
        else RQ_TRY(rq_launch_conv_in3(x, w, b, r.X, B, res, res, c.in_channels, c.ch, st));
    }
    int block_in = c.ch;
    for (int l = 0; l < nl; ++l) {
        const int block_out = c.ch * c.ch_mult[l];
        const std::string dn = "encoder.down." + std::to_string(l);
        if (phase == 2 && l < l_lo) {                          // done by phase 1 (its last conv is the downsample out of level l_lo - 1)
            block_in = block_out;
            res = vae_level_res(c, l + 1);
            if (l == l_lo - 1) RQ_HIP(hipMemcpyAsync(r.X, stage, (size_t)B * res * res * block_in * 2, hipMemcpyDeviceToDevice, st));
            continue;
        }
        for (int ib = 0; ib < c.num_res_blocks; ++ib) {
            r.res(dn + ".block." + std::to_string(ib), res, res, block_in, block_out);
            block_in = block_out;
            if (vae_has_attn(c, res)) r.attn(dn + ".attn." + std::to_string(ib), res, res, block_in);
        }
        if (l != nl - 1) {
            if (h->resamp_with_conv) {
                r.conv(dn + ".downsample.conv", r.X, r.Y, res, res, block_in, block_in, 3, 2, 0, EPI_BF16, nullptr);
                r.swap();
            } else r.resample(false, res / 2, res / 2, block_in);
            res /= 2;
        }
        if (phase == 1 && l == l_lo - 1) {
            if (r.err) return r.err;
            if (r.stats_of == r.X) return rq_fail(RQAMD_ERR_STATE, "vae: two-phase cut behind a conv that leaves GroupNorm statistics");
            RQ_HIP(hipMemcpyAsync(stage, r.X, (size_t)B * res * res * block_in * 2, hipMemcpyDeviceToDevice, st));
            return RQAMD_OK;
        }
    }
    r.res("encoder.mid.block_1", res, res, block_in, block_in);
    r.attn("encoder.mid.attn_1", res, res, block_in);
    r.res("encoder.mid.block_2", res, res, block_in, block_in);
    r.norm("encoder.norm_out", r.X, r.T1, res * res, block_in, 1);
    const int zc = c.z_channels * (c.double_z ? 2 : 1);
    r.conv("encoder.conv_out", r.T1, r.Y, res, res, block_in, zc, 3, 1, 0, EPI_BF16, nullptr);
    // quant_conv 1x1 (rqvae.py:82) -> fp32 NHWC
    r.conv("quant_conv", r.Y, z_e, res, res, zc, c.embed_dim, 1, 1, 0, EPI_F32, nullptr);
    return r.err;
}

// images of a two-phase super-chunk: up to eight chunks, bounded by what the five activation buffers (sized for `chunk` images of the
// largest layer) hold of the <= 16^2 tensors -- the widest of them with the attention's 3 C -- and returns the bytes per image of the
// parked activation (decode: level l_lo's output; encode: the downsample into level l_lo)
static int vae_super_chunk(const rqamd_vae* h, int batch, bool dec, size_t* stage_per_img) {
    const rqamd_vae_config& c = h->cfg;
    const int l_lo = vae_lo_level(c);
    size_t per_img = 0, cmax = c.z_channels > c.embed_dim ? c.z_channels : c.embed_dim;
    for (int l = l_lo > 0 ? l_lo - 1 : 0; l < c.n_levels; ++l) if ((size_t)c.ch * c.ch_mult[l] > cmax) cmax = (size_t)c.ch * c.ch_mult[l];
    for (int l = l_lo; l < c.n_levels; ++l) {
        const size_t hw = (size_t)vae_level_res(c, l) * vae_level_res(c, l);
        if (hw * cmax * 3 > per_img) per_img = hw * cmax * 3;
    }
    const int r_lo = vae_level_res(c, l_lo);
    *stage_per_img = (size_t)r_lo * r_lo * (dec ? (size_t)c.ch * c.ch_mult[l_lo] : (size_t)c.ch * c.ch_mult[l_lo - 1]) * 2;
    size_t n = per_img ? h->cap_elems / per_img : 0;
    if (n > (size_t)8 * h->chunk) n = (size_t)8 * h->chunk;
    if (n > (size_t)batch) n = batch;
    return (int)n;
}


// returns RQAMD_OK when the call was served by a graph replay, 1 when the caller should run the eager path
static int vae_graph_run(rqamd_vae* h, bool dec, const float* in, int B, float* out, hipStream_t st) {
    const rqamd_vae_config& c = h->cfg;
    if (!h->use_graph || B < 1 || B > rqamd_vae::GRAPH_MAX_B || st == nullptr) return 1;      // the legacy stream cannot capture
    const int lowres = vae_level_res(c, c.n_levels - 1);
    const size_t lat = (size_t)lowres * lowres * c.embed_dim * 4, dpix = (size_t)c.out_ch * c.resolution * c.resolution * 4,
                 epix = (size_t)c.in_channels * c.resolution * c.resolution * 4;
    const size_t M = rqamd_vae::GRAPH_MAX_B;
    const size_t o_din = 0, o_dout = o_din + M * lat, o_ein = o_dout + M * dpix, o_eout = o_ein + M * epix, total = o_eout + M * lat;
    RQ_TRY(vae_prepare(h, B > h->chunk ? B : h->chunk));          // allocation happens outside the capture
    RQ_TRY(vae_prepare_slab(h, B));
    if (h->gio.bytes < total) { RQ_TRY(h->gio.reserve(total)); h->gen++; }
    char* io = (char*)h->gio.p;
    float* gin = (float*)(io + (dec ? o_din : o_ein));
    float* gout = (float*)(io + (dec ? o_dout : o_eout));
    rqamd_vae::Graph& g = (dec ? h->gdec : h->genc)[B - 1];
    if (!g.exec || g.stream != (void*)st || g.gen != h->gen || g.io != h->gio.p) {
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (!h->graph_warned) {
                fprintf(stderr, "librqamd: hipStreamBeginCapture failed (%s); per-image encode / decode runs with eager launches\n", hipGetErrorString(e));
                h->graph_warned = true;
            }
            h->use_graph = false;
            return 1;
        }
        const int rc = dec ? decode_chunk(h, gin, B, gout, st) : encode_chunk(h, gin, B, gout, st);
        hipGraph_t graph = nullptr;
        e = hipStreamEndCapture(st, &graph);
        if (rc != RQAMD_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { g.exec = nullptr; return rq_fail(RQAMD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
        g.stream = (void*)st; g.gen = h->gen; g.io = h->gio.p;
    }
    const size_t in_bytes = (size_t)B * (dec ? lat : epix), out_bytes = (size_t)B * (dec ? dpix : lat);
    RQ_HIP(hipMemcpyAsync(gin, in, in_bytes, hipMemcpyDeviceToDevice, st));
    RQ_HIP(hipGraphLaunch(g.exec, st));
    RQ_HIP(hipMemcpyAsync(out, gout, out_bytes, hipMemcpyDeviceToDevice, st));
    return RQAMD_OK;
}

extern "C" int rqamd_vae_decode(rqamd_vae* h, const float* z_q, int batch, float* out, void* stream) {
    if (!h || !z_q || !out) return rq_fail(RQAMD_ERR_INVALID, "vae_decode: null argument");
    if (batch < 0) return rq_fail(RQAMD_ERR_INVALID, "vae_decode: batch < 0");
    const rqamd_vae_config& c = h->cfg;
    const int lowres = vae_level_res(c, c.n_levels - 1);
    const int chunk = batch < h->chunk_max ? batch : h->chunk_max;
    if (batch == 0) return RQAMD_OK;
    {
        const int g = vae_graph_run(h, true, z_q, batch, out, (hipStream_t)stream);
        if (g != 1) return g;
    }
    RQ_TRY(vae_prepare(h, chunk));
    // chunks of <= SPLIT_MAX_B images divide K over workgroups and need the slab: the FULL chunks when RQAMD_VAE_CHUNK is that small
    // (they are the larger ones), otherwise only a tail chunk
    RQ_TRY(vae_prepare_slab(h, chunk <= rqamd_vae::SPLIT_MAX_B ? chunk : (batch % chunk ? batch % chunk : chunk)));
    if (batch > chunk && vae_two_phase(h)) {
        size_t spi = 0;
        const int ns = vae_super_chunk(h, batch, true, &spi);
        if (ns > chunk) {
            RQ_TRY(h->stage.reserve((size_t)ns * spi));
            for (int s0 = 0; s0 < batch; s0 += ns) {
                const int m = (batch - s0 < ns) ? batch - s0 : ns;
                RQ_TRY(decode_chunk(h, z_q + (size_t)s0 * lowres * lowres * c.embed_dim, m, nullptr, (hipStream_t)stream, 1, h->stage.as<bf16_t>()));
                for (int b0 = 0; b0 < m; b0 += chunk) {
                    const int n = (m - b0 < chunk) ? m - b0 : chunk;
                    RQ_TRY(decode_chunk(h, nullptr, n, out + (size_t)(s0 + b0) * c.out_ch * c.resolution * c.resolution, (hipStream_t)stream, 2,
                                        (bf16_t*)((char*)h->stage.p + (size_t)b0 * spi)));
                }
            }
            return RQAMD_OK;
        }
    }
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int n = (batch - b0 < chunk) ? batch - b0 : chunk;
        RQ_TRY(decode_chunk(h, z_q + (size_t)b0 * lowres * lowres * c.embed_dim, n,
                            out + (size_t)b0 * c.out_ch * c.resolution * c.resolution, (hipStream_t)stream));
    }
    return RQAMD_OK;
}

extern "C" int rqamd_vae_encode(rqamd_vae* h, const float* x, int batch, float* z_e, void* stream) {
    if (!h || !x || !z_e) return rq_fail(RQAMD_ERR_INVALID, "vae_encode: null argument");
    if (batch < 0) return rq_fail(RQAMD_ERR_INVALID, "vae_encode: batch < 0");
    const rqamd_vae_config& c = h->cfg;
    const int lowres = vae_level_res(c, c.n_levels - 1);
    const int chunk = batch < h->chunk_max ? batch : h->chunk_max;
    if (batch == 0) return RQAMD_OK;
    {
        const int g = vae_graph_run(h, false, x, batch, z_e, (hipStream_t)stream);
        if (g != 1) return g;
    }
    RQ_TRY(vae_prepare(h, chunk));
    RQ_TRY(vae_prepare_slab(h, chunk <= rqamd_vae::SPLIT_MAX_B ? chunk : (batch % chunk ? batch % chunk : chunk)));
    if (batch > chunk && vae_two_phase(h)) {
        size_t spi = 0;
        const int ns = vae_super_chunk(h, batch, false, &spi);
        if (ns > chunk) {
            RQ_TRY(h->stage.reserve((size_t)ns * spi));
            for (int s0 = 0; s0 < batch; s0 += ns) {
                const int m = (batch - s0 < ns) ? batch - s0 : ns;
                for (int b0 = 0; b0 < m; b0 += chunk) {
                    const int n = (m - b0 < chunk) ? m - b0 : chunk;
                    RQ_TRY(encode_chunk(h, x + (size_t)(s0 + b0) * c.in_channels * c.resolution * c.resolution, n, nullptr, (hipStream_t)stream, 1,
                                        (bf16_t*)((char*)h->stage.p + (size_t)b0 * spi)));
                }
                RQ_TRY(encode_chunk(h, nullptr, m, z_e + (size_t)s0 * lowres * lowres * c.embed_dim, (hipStream_t)stream, 2, h->stage.as<bf16_t>()));
            }
            return RQAMD_OK;
        }
    }
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int n = (batch - b0 < chunk) ? batch - b0 : chunk;
        RQ_TRY(encode_chunk(h, x + (size_t)b0 * c.in_channels * c.resolution * c.resolution, n,
                            z_e + (size_t)b0 * lowres * lowres * c.embed_dim, (hipStream_t)stream));
    }
    return RQAMD_OK;
}
