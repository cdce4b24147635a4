// DiffWave eps-network, bf16 MFMA path (gfx950): shared documentation + the init kernel.
//
// Replaces the 36x Residual_block.forward loop of the reference
// (diffusion_models/DiffWave_Unconditional/WaveNet.py:75-97,120-135) and its tail
// (WaveNet.py:135,160-162) with two hand-written persistent kernels:
//
//   wn_layer_bf16_p (wn_layer.hip)  one launch per residual layer.  Per 128-sample time tile of one clip:
//                  H = W_dil * [h(t-d); h(t); h(t+d)] + b        (implicit GEMM, M=512, K=768)
//                  g = tanh(H[:256]) * sigmoid(H[256:])          (in registers, fp32)
//                  res = W_res * g + b                           (M=256, K=256, g through LDS)
//                  h' = (h + res) * sqrt(1/2) + emb_{n+1}        (the reference's in-place alias
//                                                                 h = x + fc_t(emb))
//                  and streams g (bf16) to HBM: the 36 skip convolutions are NOT done here.
//   wn_final_bf16_p (wn_final.hip)  one launch per network evaluation: skip = sum_n W_skip_n * g_n as ONE
//                  GEMM with K = 36*256 over the stored gate outputs (fp32 accumulation in the MFMA
//                  accumulators instead of 36 fp32 read-modify-write passes over HBM), then
//                  relu(W_f0 * skip/6 + b) and the 256->1 output conv.
//
// Layouts:
//   residual stream h : bf16 [B][kPad + L + kPad rows][256] in the blocked "H16" order (dmad_common.h: 16-row blocks,
//                       chunk-major inside a block); pad rows are zero and never written
//   gate store g      : bf16 [layer][8 k-chunks][B*L positions][32 ch]  (k-chunk-major: one k-step of the
//                       skip GEMM reads 256 positions x 64 B = 16 KiB of CONTIGUOUS HBM)
//   packed weights    : bf16 LDS images [k-step][row][32 k], 64-B rows with the swz64 chunk swizzle
// MFMA: v_mfma_f32_16x16x32_bf16, D[channel][time]; A = weights (rows = out channel),
//       B = activations (cols = time); 8 waves = 4 (M) x 2 (N); fp32 accumulate.
#include "dmad_common.h"
#include "wn_bf16.h"

namespace dmad {

// h0 = relu(w_init * x + b_init) + emb_0  (WaveNet.py:147,13-19 + the F5 alias of layer 0), bf16 out
template <typename T>
__global__ void __launch_bounds__(256) wn_init_h16(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ emb0,
                                                   h16_t* __restrict__ h, int L, int LP, long total_chunks) {
    typedef typename H16<T>::v8 v8;
    // one thread per 8-channel chunk of one time position, in the order of the H16 layout: 16 consecutive lanes write the
    // 16 rows of one chunk column (256 contiguous bytes), a wave 1 KiB, a 16-row block 8 KiB (L is a multiple of 16, so
    // a block never straddles two clips)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_chunks;
         idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)((idx >> 4) & 31);
        const long pos = (idx >> 9) * 16 + (idx & 15);      // b * L + t
        const long bb = pos / L, t = pos - bb * L;
        const float xv = x[pos];
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            const float v = w[c] * xv + bias[c];
            o[j] = (T)(relu_nan(v) + emb0[c]);
        }
        *(v8*)((char*)(h + (size_t)bb * LP * kC) + h16_off((unsigned)(kPad + t), (unsigned)cg)) = o;
    }
}

void launch_wn_init_bf16(const float* x, const float* w, const float* bias, const float* emb0, h16_t* h, int B, int L,
                         int LP, bool f16, hipStream_t s) {
    const long chunks = (long)B * L * 32;
    const int grid = (int)((chunks + 255) / 256 < 8192 ? (chunks + 255) / 256 : 8192);
    if (f16) hipLaunchKernelGGL(wn_init_h16<_Float16>, dim3(grid), dim3(256), 0, s, x, w, bias, emb0, h, L, LP, chunks);
    else hipLaunchKernelGGL(wn_init_h16<__bf16>, dim3(grid), dim3(256), 0, s, x, w, bias, emb0, h, L, LP, chunks);
}

int wn_bf16_configure() {
    if (int r = wn_layer_p_configure()) return r;
    return wn_final_p_configure();
}

}  // namespace dmad
