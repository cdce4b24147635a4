// Argument blocks and launchers of the bf16 MFMA WaveNet kernels (wn_bf16.hip).
#pragma once
#include "dmad_common.h"

namespace dmad {

constexpr int kWnLdsBytesV2 = 155648; // 3-slot ring (3 x 40 KiB) + 2 dedicated GEMM2 weight buffers
constexpr int kWnLdsBytesV3 = 163840; // gate tile 64 KiB + six GEMM2 weight buffers (ring aliased underneath)
constexpr int kWnLdsBytes = 147456;   // 2x32K weight stages + 2x8K activation stages + 64K gate tile

struct WnLayerArgs {
    const bf16_t* hin;       // [B][LP][256] residual stream in  (h_n = x_n + fc_t_n(emb))
    bf16_t* hout;            // [B][LP][256] residual stream out (h_{n+1})
    bf16_t* gout;            // [B][L][256]  gate output of this layer
    const bf16_t* w1p;       // [24][512][32] dilated-conv weights, packed LDS images
    const bf16_t* w2p;       // [8][256][32]  res-conv weights, packed LDS images
    const float* b1;         // [512] dilated-conv bias in tile-row order
    const float* b2;         // [256] res-conv bias
    const float* emb_next;   // [256] fc_t_{n+1}(emb) (unused when last)
    const float* epi_c;      // [256] b_res * sqrt(1/2) + fc_t_{n+1}(emb): epilogue constant of the persistent kernel
    int dilation, L, LP, last;
    unsigned long long* dbg; // diagnostic builds only: [grid][8] phase timestamps
};

struct WnFinalArgs {
    const bf16_t* g;         // [NL][B][L][256]
    const bf16_t* wsp;       // [NL*8][256][32] skip-conv weights, packed
    const bf16_t* wf0p;      // [8][256][32]    final_conv.0 weights, packed
    const float* bskip_sum;  // [256] sum_n b_skip_n
    const float* bf0;        // [256]
    const float* wz;         // [256] final_conv.2 weight
    float* eps;              // [B][L]
    float bz, skip_scale;
    int NL, B, L;
};

void launch_wn_layer_bf16(const WnLayerArgs& a, int B, hipStream_t s, int variant = 0);   // variant > 0: timing-only ablations
void launch_wn_final_bf16(const WnFinalArgs& a, hipStream_t s);
void launch_wn_init_bf16(const float* x, const float* w, const float* bias, const float* emb0, bf16_t* h, int B, int L,
                         int LP, hipStream_t s);
int wn_bf16_configure();
void launch_wn_layer_bf16_p(const WnLayerArgs& a, int B, hipStream_t s, bool stamps = false);   // persistent production schedule (wn_layer.hip)
int wn_layer_p_configure();
void launch_wn_final_bf16_p(const WnFinalArgs& a, hipStream_t s);            // persistent production schedule (wn_final.hip)
bool wn_final_p_supported(const WnFinalArgs& a);
int wn_final_p_configure();

}  // namespace dmad
