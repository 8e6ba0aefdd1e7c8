// vae_kernels.h -- launchers of the non-GEMM RQ-VAE kernels (see vae_kernels.hip)
#pragma once
#include "rq_hip.h"

#define RQ_GN_MAX_CHUNK 64

int rq_launch_groupnorm(const bf16_t* x, bf16_t* y, float* part, const float* gamma, const float* beta, int B, int HW, int C,
                        int silu, hipStream_t s);
// out[m][n] = (bf16 or fp32)(sum_z slabs[z][m][n] + bias[n] (+ resid[m][n])): finishes a split-K conv (small-batch mode)
int rq_launch_splitk_reduce(const float* slabs, int n_slabs, int M, int N, const float* bias, const bf16_t* resid, void* out, int out_f32,
                            hipStream_t s);
int rq_launch_vae_attn(const bf16_t* qkv, bf16_t* out, int B, int T, int C, hipStream_t s);
int rq_launch_conv_in3(const float* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int Cin, int Cout, hipStream_t s);
int rq_launch_conv_out3(const bf16_t* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, hipStream_t s);
int rq_launch_repack_conv(const float* src, void* dst, int O, int I, int kh, int kw, int mode, hipStream_t s);
int rq_launch_gn_stats(const bf16_t* x, float* part, int B, int HW, int C, int* nchunk_out, hipStream_t s);
// conv_halo.hip: halo-reuse 3x3 conv with optional fused GroupNorm+SiLU on the input
bool rq_conv_halo_supported(int H, int W, int Cin, int Cout);
bool rq_conv_halo_subpixel_supported(int H, int W, int Cin, int Cout);      // the upsample conv as four 2 x 2 convs over the source (ups = 2)
int rq_launch_ups_subpixel_weights(const bf16_t* w, bf16_t* wsub, int Cout, int Cin, hipStream_t s);
// stats != null: the epilogue also writes the GroupNorm partials of `out` ([B][rq_conv_halo_stat_tiles][32][2]);
// ups != 0: x is [B][H/2][W/2][Cin] and is read through a nearest 2x upsample (gn and resid must be null)
int rq_conv_halo_stat_tiles(int H, int W);
int rq_launch_conv_halo(const bf16_t* x, const bf16_t* w, const float* bias, const float* gn, const bf16_t* resid, bf16_t* out,
                        float* stats, int B, int H, int W, int Cin, int Cout, int ups, hipStream_t s);
// MFMA conv_out (Cin -> Cout <= 4, NCHW fp32 image out) with optional fused GroupNorm+SiLU of norm_out
bool rq_conv_out_halo_supported(int H, int W, int Cin, int Cout);
int rq_launch_conv_out_halo(const bf16_t* x, const float* w, const float* bias, const float* gn, float* y, int B, int H, int W,
                            int Cin, int Cout, hipStream_t s);
// MFMA Encoder.conv_in (3 -> 128 channels, NCHW fp32 image in, NHWC bf16 out)
bool rq_conv_in_mfma_supported(int H, int W, int Cin, int Cout);
// stats != null: also the GroupNorm partials of y ([B][rq_conv_halo_stat_tiles(H, W)][32][2])
int rq_launch_conv_in_mfma(const float* x, const float* w, const float* bias, bf16_t* y, float* stats, int B, int H, int W, hipStream_t s);
int rq_launch_gn_params(const bf16_t* x, float* part, const float* gamma, const float* beta, float* gn, int B, int HW, int C,
                        int nchunk_have, hipStream_t s);

// resamp_with_conv = False (layers.py:20-57): bare nearest-2x upsample / 2 x 2 average pool, NHWC bf16; Ho, Wo = OUTPUT size
int rq_launch_upsample2(const bf16_t* x, bf16_t* y, int B, int Ho, int Wo, int C, hipStream_t s);
int rq_launch_avgpool2(const bf16_t* x, bf16_t* y, int B, int Ho, int Wo, int C, hipStream_t s);
