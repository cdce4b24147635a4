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
                          // 2: 3x3 (taps == 9, zero padding 1) or 1x1 (taps == 1) conv over NHWC [B][H][W][ldx]
    long rows_per_batch, batch_stride, row_stride, tap_stride;   // floats (mode 0)
    int H, W, Cin;        // mode 2: input height / width, K == Cin (channels of this group)
    int ldx;              // mode 2: floats between two pixels of X (0 = Cin; larger when X holds several groups)
    int stride;           // mode 2: spatial stride (0/1 = 1, 2 = output (H-1)/2+1 x (W-1)/2+1); taps == 1 is a 1x1 conv
    int groups;           // > 1: grouped conv, one group per grid.z; M, K and the A image are per group, X / C / scale /
                          //      shift / res advance by K resp. M per group (ldx / ldc = all groups); excludes split-K
    const float* X2;      // mode 2, optional: channels [ksplit, K) of every pixel come from X2 (pixel pitch ldx2) instead of X — the
    int ksplit, ldx2;     //   channel concatenation th.cat([h, skip], dim=1) of the UNet (unet.py:473) without materialising it;
                          //   ksplit % 16 == 0, X then holds ksplit channels per pixel (ldx = its pitch)
    const float* res;     // optional residual [N][ldc] added before the ReLU (ResNeXt bottleneck sum)
    // Fused epilogues of the exact-fp32 WaveNet layer (Residual_block.forward, WaveNet.py:86-97); M = 512, no split-K:
    //   epi 1  gate: the rows of A / shift are permuted so that every wave holds a gate channel's tanh row (accumulator
    //          tiles 0-1) and its sigmoid row (tiles 2-3): tile-local row wm*64 + i*16 + r of block bm is H row
    //          (i >= 2 ? 256 : 0) + bm*64 + wm*32 + (i&1)*16 + r.  C[n][ch] = tanh(H[ch]) * sigmoid(H[256+ch]), ldc = 256.
    //   epi 2  update (M = 256 = res_rows, the res conv):
    //          hout[row(n)][m] = (hin[row(n)][m] + v) * sqrt(1/2) + emb_next[m]
    //          (row(n) = position n inside the zero-padded residual stream; the last layer, whose residual output is never
    //          consumed, has no such launch; the skip convs run as one K = NL * 256 GEMM after the layer loop).
    //          `first` / `skip` are unused since then.
    int epi, res_rows, first, L, LP;
    const float* hin;
    float* hout;
    float* skip;
    const float* emb_next;
    // x3 != 0 (K a multiple of 32, no split-K, no groups): both operands are in the split-f16 storage format (dmad_common.h) and every
    // product is three v_mfma_f32_16x16x32_f16.  mode 0 (the WaveNet's GEMMs; M a multiple of 256): outputs that feed another GEMM
    // (epi 1: the gate, epi 2: hout) are written in that format, everything else (plain C, the skip sum) stays fp32.  mode 2 (NHWC convs:
    // 3x3 / 1x1, stride, two-part input; M a multiple of 128): plain epilogue with shift or scale / shift, fp32 residual, ReLU; fp32 out,
    // or the split format with out_split.  groups > 1 (mode 2, no two-part input): grouped conv, one group per grid.y; M, K and the A image are
    // per group (M a multiple of 128), X / C / scale / shift / res advance by K resp. M per group.
    int x3;
    int out_split;        // x3, plain epilogue: C is written in the split format (the consumer is another GEMM of the tier)
    int res_split;        // x3, plain epilogue: `res` is a map in the split format (a block output that only exists in that form)
    int diag;             // x3 only, error-attribution builds: bit 0 weights = f16(w), bit 1 the MFMA eats f16(x), bit 2 split-format outputs keep hi only
    int splits;           // > 1: split-K over grid.z; partial sums go to `slab` [splits][N][ldc] (fixed-order reduce)
    float* slab;
};

// `slab` (device workspace of `slab_floats` floats) enables deterministic split-K for launches that would not
// fill the chip (deep VGG layers at small spatial size, batch-sized Linear layers); pass nullptr to disable.
// Returns 0, or kGemmBadShape for an argument block no kernel serves (nothing is launched; the caller reports it).
constexpr int kGemmBadShape = -1;
int launch_gemm_f32(const GemmF32Args& a, hipStream_t s, float* slab = nullptr, long slab_floats = 0, long n_ref = 0);
int gemm_take_bad_shapes();   // number of launches refused on this thread since the last call (reset to 0)
int gemm_f32_configure();     // per device, from dmad_create: dynamic-LDS attribute of the 8-slot narrow-tile kernel (0 or a hipError_t)
int gemm_x3_configure();      // per device, from dmad_create: dynamic-LDS attribute of the split-f16 kernel (0 or a hipError_t)

}  // namespace dmad
