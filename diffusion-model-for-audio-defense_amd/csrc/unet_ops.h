// Elementwise / normalisation / attention kernels of the Improved-Diffusion UNet purifier (SURVEY §8f row N1;
// reference diffusion_models/Improved_Diffusion_Unconditional/improved_diffusion/unet.py, nn.py).  All maps are fp32
// NHWC ([B][H*W][C]); the convolutions and linear layers run through gemm_f32.
#pragma once
#include "dmad_common.h"

namespace dmad {

// conv 3x3, one input channel, padding 1, bias: in [B][32][32] -> out [B][1024][Cout]   (input_blocks.0.0)
// out16: optional f16 twin of the output (the 16-bit tier's GEMMs read f16 maps; out may then be null); stats: optional GroupNorm
// statistics of the f16 twin (GemmH16Args::stats layout, 64-pixel blocks).  Returns -1 for Cout > 128.
int launch_conv1ch_3x3(const float* in, const float* w, const float* bias, float* out, int B, int Cout, hipStream_t s, h16_t* out16 = nullptr,
                       float* stats = nullptr);
// conv 3x3, 128 -> 1 channel, padding 1, bias: in [B][1024][128] (NHWC), w [9][128] (tap-major), out [B][1024]   (out.2)
void launch_conv3x3_c128_to1(const float* in, const float* w, const float* bias, float* out, int B, hipStream_t s);
void launch_conv3x3_c128_to1_h16(const h16_t* in, const float* w, const float* bias, float* out, int B, hipStream_t s);   // the same on an f16 map
// GroupNorm32(32, C) in fp32 (nn.py:15-17,92-100) over [B][HW][C], then optionally y * (1 + ss[c]) + ss[C + c]
// (scale-shift norm, unet.py:190-194; ss = one row of 2C floats shared by the batch) and optionally SiLU.
// Returns -1 for a map it is not built for (C not a multiple of 128, or more than 12288 values per group).
// x2 != nullptr: the input is the channel concatenation [x : c1 channels | x2 : C - c1 channels] (th.cat(dim=1), unet.py:473)
// read from its two parts; y is always one map of C channels.
int launch_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* ss, int silu, float* y, int B, int HW,
                          int C, hipStream_t s, const float* x2 = nullptr, int c1 = 0, h16_t* y16 = nullptr,      // y16 != nullptr: the result is written as f16 there (y unused)
                          const h16_t* x16 = nullptr, const h16_t* x2_16 = nullptr,    // x16 != nullptr: the input is read from f16 map(s) x16 (/ x2_16) instead of x (/ x2)
                          int split = 0);             // split: y (fp32 input form only) is written in the split-f16 storage format of dmad_common.h
// The 16-bit tier's GroupNorm as one streaming pass over an f16 map (or the two parts x [c1 channels] | x2 of a concatenated input)
// whose statistics the producing GEMMs left in st / st2 (GemmH16Args::stats: [B * HW / 64][channels / 4][2] floats per map):
// y = SiLU?((x - mean) * rstd * gamma + beta [* (1 + ss[c]) + ss[C + c]]) as f16 (y16) or fp32 (y32).  HW a multiple of 64, or 16 (the
// 4 x 4 maps: statistics blocks of 16 pixels, GemmH16Args::stats_px = 16).
// Returns -1 for a map it does not serve (the caller then takes launch_groupnorm_nhwc).
int launch_groupnorm16_apply(const h16_t* x, const float* st, const h16_t* x2, const float* st2, int c1, const float* gamma, const float* beta,
                             const float* ss, int silu, h16_t* y16, float* y32, int B, int HW, int C, hipStream_t s);
void launch_silu(const float* x, float* y, long n, hipStream_t s);
// nearest-neighbour x2 (F.interpolate(scale_factor=2, mode="nearest"), unet.py:72)
void launch_upsample2x_nhwc(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
void launch_upsample2x_nhwc_h16(const h16_t* in, h16_t* out, int B, int H, int W, int C, hipStream_t s);       // the same on an f16 map, C % 8 == 0
// QKVAttention (unet.py:241-258) with the reference's head-major channel split: qkv [B*T][3C] (token-major),
// head h: q = channels h*3*64 + [0,64), k = + 64, v = + 128;  out [B*T][C], channel h*64 + c.  Head width 64.
// T = 256, 64 or 16 (the 16x16, 8x8 and 4x4 maps this network attends at); returns -1 for any other T, a hipError_t > 0
// if the kernel could not be configured.
int launch_qkv_attention(const float* qkv, float* out, int B, int T, int heads, hipStream_t s, h16_t* out16 = nullptr,   // out16: write f16 there instead of out
                         int split = 0);      // split: `out` is written in the split-f16 storage format (operand of the middle tier's proj GEMM)
// The same on the 16-bit tier: qkv and out in f16, both products on the f16 matrix cores (fp32 accumulate, fp32 softmax).
int launch_qkv_attention_h16(const h16_t* qkv, h16_t* out, int B, int T, int heads, hipStream_t s);
// GaussianDiffusion.p_sample (gaussian_diffusion.py:232-257,331-387: epsilon prediction, clip_denoised, fixed variance):
//   x0 = clamp(ca * x - cb * eps, -1, 1);  out = c1 * x0 + c2 * x + sig * z      (z may be null when sig == 0)
void launch_unet_p_sample(const float* x, const float* eps, const float* z, float ca, float cb, float c1, float c2, float sig,
                          float* out, float* x0_out, long n, hipStream_t s);

// melspec_standardize + q_sample (z == nullptr: standardise only) and melspec_inv_standardize of the spec-domain purifier
// (sc09_spectrogram_dataset.py:62-81, gaussian_diffusion.py:188-206)
void launch_spec_diffuse(const float* spec, const float* z, float lo, float hi, float qa, float qb, float* xt, long n, hipStream_t s);
void launch_spec_unstandardize(const float* x, float lo, float hi, float* spec, long n, hipStream_t s);

}  // namespace dmad
