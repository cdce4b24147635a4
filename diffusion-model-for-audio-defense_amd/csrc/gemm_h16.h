// f16 implicit-GEMM for NHWC convolutions (gemm_h16.hip): argument block and launcher.
#pragma once
#include "dmad_common.h"

namespace dmad {

struct GemmH16Args {
    const h16_t* A;       // f16 weights [taps][M][K], K contiguous, K % 64 == 0, M % 128 == 0
    const h16_t* X;       // f16 NHWC activations [B][H][W][ldx]
    float* C;             // fp32 output [N][ldc] or nullptr
    h16_t* C16;           // f16 twin of the output [N][ldc] or nullptr (for consumers that are GEMMs)
    const float* shift;   // [M] bias or nullptr
    const float* res;     // optional fp32 residual [N][ldc] added to the output
    const h16_t* res16;   // ... or the residual as an f16 map [N][ldc] (the UNet's 16-bit tier keeps its hidden state in f16 only); not both
    int M, K, taps, ldc;  // taps: 9 (3x3, zero padding 1) or 1 (1x1)
    long N;               // output pixels: B * Ho * Wo
    int H, W;             // input height / width
    int ldx;              // halves between two pixels of X
    int stride;           // 0/1 = 1, 2 = output (H-1)/2+1 x (W-1)/2+1
    const h16_t* X2;      // optional: channels [ksplit, K) of every pixel come from X2 (pixel pitch ldx2) — th.cat([h, skip], dim=1)
    int ksplit, ldx2;     //   of the UNet (unet.py:473) read in place; ksplit % 64 == 0
    float* stats;         // optional [ceil(N / 64)][ldc / 4][2]: (sum, sum of squares) of the f16-ROUNDED outputs of every (64-pixel block,
                          //   4-channel quad) — the GroupNorm statistics of the consumer, accumulated in the epilogue (groupnorm16_apply_kernel)
    int stats_px;         // pixels per statistics block: 0 / 64 (a wave's 64 pixel rows) or 16 (maps of 16 pixels per sample: 4 x 4; served by
                          //   the 384-row kernel only) — the slab is then [ceil(N / 16)][ldc / 4][2]
    int relu;             // 1: max(., 0) after bias and residual (NaN stays NaN, like torch.relu) — ResNeXt's BN-folded convs
    int groups;           // 0/1 = dense; g > 1: grouped conv (resnext.py:36-37), M and K are PER GROUP: group z reads channels
                          //   [z K, (z+1) K) of X (pixel pitch ldx), weights A + z * taps * M * K, writes channels [z M, (z+1) M) (pitch ldc)
    int up2;              // 1: X is the HALF-resolution map [B][H/2][W/2][ldx] and the conv reads its nearest-neighbour x2 upsampling
                          //   (F.interpolate(scale_factor=2) + conv of the UNet's Upsample, unet.py:72-79) without materialising it.  Served by the
                          //   slice-resident form only: ask gemm_h16_fuses_up2() first
};

// 0, or -1 for an argument block the kernel does not serve (nothing is launched; counted for gemm_h16_take_bad_shapes)
int launch_gemm_h16(const GemmH16Args& a, hipStream_t s);
// true if launch_gemm_h16 would serve this block WITH up2 = 1 (a: the block for the upsampled geometry, X = the half-resolution map)
bool gemm_h16_fuses_up2(const GemmH16Args& a);
int gemm_h16_take_bad_shapes();
int gemm_h16_configure();     // per device, from dmad_create: dynamic-LDS attribute (0 or a hipError_t)

}  // namespace dmad
