// Argument blocks and launchers of the bf16 MFMA WaveNet kernels (wn_bf16.hip, wn_layer.hip, wn_final.hip).
#pragma once
#include "dmad_common.h"

namespace dmad {

constexpr int kWnLdsBytes = 163840;   // both persistent kernels use the whole 160 KiB LDS of a CU

struct WnLayerArgs {
    const h16_t* hin;       // [B][LP][256] residual stream in  (h_n = x_n + fc_t_n(emb))
    h16_t* hout;            // [B][LP][256] residual stream out (h_{n+1})
    h16_t* gout;            // [8][B*L][32]  gate output of this layer, k-chunk-major
    const h16_t* w1p;       // [24][512][32] dilated-conv weights, packed LDS images; stage = 3 * kchunk + tap
    const h16_t* w2p;       // [8][256][32]  res-conv weights, packed LDS images
    const float* b1;         // [512] dilated-conv bias in tile-row order
    const float* epi_c;      // [256] b_res * sqrt(1/2) + fc_t_{n+1}(emb): epilogue constant (unused when last)
    int dilation, L, LP, last;
    long npos;               // B * L
    unsigned long long* dbg; // diagnostic (STAMP) builds only: [grid][8] per-phase cycle sums
};

struct WnFinalArgs {
    const h16_t* g;         // [NL][8][B*L][32]
    const h16_t* wsp;       // [NL*8][256][32] skip-conv weights, packed
    const h16_t* wf0p;      // [8][256][32]    final_conv.0 weights, packed
    const float* bskip_sum;  // [256] sum_n b_skip_n
    const float* bf0;        // [256]
    const float* wz;         // [256] final_conv.2 weight
    float* eps;              // [B][L]
    float bz, skip_scale;
    int NL, B, L;
};

// f16: operands are IEEE half instead of bfloat16 (same layouts, same kernels instantiated on the other type)
void launch_wn_init_bf16(const float* x, const float* w, const float* bias, const float* emb0, h16_t* h, int B, int L,
                         int LP, bool f16, hipStream_t s);
void launch_wn_layer_bf16_p(const WnLayerArgs& a, int B, bool f16, hipStream_t s, bool stamps = false);
void launch_wn_final_bf16_p(const WnFinalArgs& a, bool f16, hipStream_t s);
bool wn_final_p_supported(int num_res_layers);
int wn_bf16_configure();
int wn_layer_p_configure();
int wn_final_p_configure();

}  // namespace dmad
