// fp32 gather-GEMM (gemm_f32.hip): argument block and launcher.
#pragma once
#include "dmad_common.h"

namespace dmad {

struct GemmF32Args {
    const float* A;       // weights [taps][M][K], K contiguous, K % 16 == 0
    const float* X;       // activation base pointer (rows of K contiguous floats, 16-B aligned)
    float* C;             // output [N][ldc]
    const float* scale;   // [M] or nullptr (eval-mode BatchNorm scale)
    const float* shift;   // [M] or nullptr (bias / folded BatchNorm shift)
    int M, K, taps, ldc, relu;
    long N;
    int mode;             // 0: row(n,tap) = (n / R) * batch_stride + (n % R) * row_stride + (tap - taps/2) * tap_stride
                          // 2: 3x3 conv over NHWC [B][H][W][Cin] with zero padding (taps == 9, K == Cin)
    long rows_per_batch, batch_stride, row_stride, tap_stride;   // floats (mode 0)
    int H, W, Cin;        // mode 2
    int splits;           // > 1: split-K over grid.z; partial sums go to `slab` [splits][N][ldc] (fixed-order reduce)
    float* slab;
};

// `slab` (device workspace of `slab_floats` floats) enables deterministic split-K for launches that would not
// fill the chip (deep VGG layers at small spatial size, batch-sized Linear layers); pass nullptr to disable.
void launch_gemm_f32(const GemmF32Args& a, hipStream_t s, float* slab = nullptr, long slab_floats = 0, long n_ref = 0);

}  // namespace dmad
