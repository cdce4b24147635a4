// DiffWave eps-network, bf16 MFMA path (gfx950).
//
// Replaces the 36x Residual_block.forward loop of the reference
// (diffusion_models/DiffWave_Unconditional/WaveNet.py:75-97,120-135) and its tail
// (WaveNet.py:135,160-162) with two hand-written kernels:
//
//   wn_layer_bf16  one launch per residual layer.  Per 128-sample time tile of one clip it computes
//                  H = W_dil * [h(t-d); h(t); h(t+d)] + b        (implicit GEMM, M=512, K=768)
//                  g = tanh(H[:256]) * sigmoid(H[256:])          (in registers, fp32)
//                  res = W_res * g + b                           (M=256, K=256, g through LDS)
//                  h' = (h + res) * sqrt(1/2) + emb_{n+1}        (the reference's in-place alias
//                                                                 h = x + fc_t(emb))
//                  and streams g (bf16) to HBM: the 36 skip convolutions are NOT done here.
//   wn_final_bf16  one launch per network evaluation: skip = sum_n W_skip_n * g_n as ONE GEMM with
//                  K = 36*256 over the stored gate outputs (fp32 accumulation in the MFMA
//                  accumulators instead of 36 fp32 read-modify-write passes over HBM), then
//                  relu(W_f0 * skip/6 + b) and the 256->1 output conv.
//
// Layouts:
//   residual stream h : bf16 [B][kPad + L + kPad][256], pad rows are zero and never written
//   gate store g      : bf16 [layer][B][L][256]
//   packed weights    : bf16 LDS images [k-step][row][32 k], 64-B rows with the swz64 chunk swizzle
// MFMA: v_mfma_f32_16x16x32_bf16, D[channel][time]; A = weights (rows = out channel),
//       B = activations (cols = time); 8 waves = 4 (M) x 2 (N); fp32 accumulate.
#include "dmad_common.h"
#include "wn_bf16.h"

namespace dmad {

namespace {

constexpr int LDS_A = 0;              // 2 x 32 KiB weight stage buffers
constexpr int LDS_B = 65536;          // 2 x  8 KiB activation stage buffers
constexpr int LDS_G = 81920;          // 64 KiB gate tile [128 t][256 ch] bf16 (chunk ^ (t & 15))
constexpr int EPI_PITCH = 1040;       // fp32 [128 t][256 ch] epilogue tile overlay (133120 B)

// hipcc does not count an LDS-DMA (global_load_lds) as a pending LDS write at __syncthreads()
// inside a loop: drain it explicitly before the barrier that publishes the staged tile.
__device__ __forceinline__ void sync_stage() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// tanh(a) * sigmoid(b) with 2 v_exp + 1 v_rcp:  u = e^{-2a}, v = e^{-b}:  (1-u) / ((1+u)(1+v))
__device__ __forceinline__ float gate_fn(float a, float b) {
    const float u = fast_exp2(fminf(a * -2.8853900817779268f, 30.f));
    const float v = fast_exp2(fminf(b * -1.4426950408889634f, 30.f));
    const float p = 1.f + u;
    return (1.f - u) * fast_rcp(fmaf(p, v, p));
}

}  // namespace

__global__ void __launch_bounds__(512, 2) wn_layer_bf16(WnLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int tiles_per_clip = a.L / kTileT;
    const int b = blockIdx.x / tiles_per_clip;
    const int t0 = (blockIdx.x - b * tiles_per_clip) * kTileT;

    const char* hin_c = (const char*)(a.hin + ((size_t)b * a.LP + kPad + t0) * kC);
    const int brow = tid >> 2;
    const char* bsrc = hin_c + (size_t)brow * 512 + (((tid & 3) ^ swz64(brow)) * 16);
    const ptrdiff_t tap_bytes = (ptrdiff_t)a.dilation * 512;

    auto stage1 = [&](int ks, int buf) {
        const char* wsrc = (const char*)a.w1p + (size_t)ks * 32768 + tid * 16;
        char* la = smem + LDS_A + buf * 32768 + wv * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wsrc + i * 8192, la + i * 8192);
        const int tap = ks >> 3, kc = ks & 7;
        glds16(bsrc + (tap - 1) * tap_bytes + kc * 64, smem + LDS_B + buf * 8192 + wv * 1024);
    };
    auto stage2 = [&](int ks2, int buf) {
        const char* wsrc = (const char*)a.w2p + (size_t)ks2 * 16384 + tid * 16;
        char* la = smem + LDS_A + buf * 32768 + wv * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(wsrc + i * 8192, la + i * 8192);
    };

    // lane-constant part of every 64-B-row fragment address
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);

    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b1 + wm * 128 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias;
    }

    // ---------------- GEMM1: dilated conv, K = 3 taps x 256 channels, 24 k-steps of 32 -------------
    stage1(0, 0);
    for (int ks = 0; ks < 24; ++ks) {
        sync_stage();      // stage ks landed (vmcnt(0)) and buffer (ks+1)&1 is no longer being read
        if (ks + 1 < 24) stage1(ks + 1, (ks + 1) & 1);
        else if (!a.last) stage2(0, 0);
        const char* A = smem + LDS_A + (ks & 1) * 32768 + wm * 8192 + frag_off;
        const char* Bt = smem + LDS_B + (ks & 1) * 8192 + wn * 4096 + frag_off;
        bf16x8 bf[4], af[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const bf16x8*)(Bt + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) af[mt] = *(const bf16x8*)(A + mt * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[mt], bf[nt], acc[mt][nt]);
    }

    // ---------------- gate: g[ch][t] -> LDS [t][ch] bf16 ------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = (bf16_t)gate_fn(acc[mt][nt][r], acc[mt + 4][nt][r]);
            const int t = wn * 64 + nt * 16 + r16;
            const int chunk = wm * 8 + mt * 2 + (q >> 1);
            *(bf16x4*)(smem + LDS_G + t * 512 + ((chunk ^ r16) * 16) + (q & 1) * 8) = gv;
        }
    }
    sync_stage();          // gate tile complete; (stage2(0) landed)

    // ---------------- stream g to HBM (coalesced 512-B rows) --------------------------------------
    {
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            const uint4 v = *(const uint4*)(smem + LDS_G + t * 512 + ((c ^ (t & 15)) * 16));
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = v;
        }
    }
    if (a.last) return;    // the last layer's residual output is never consumed (WaveNet.py:131-135)

    // ---------------- GEMM2: res = W_res * g, M = 256, K = 256, 8 k-steps --------------------------
    f32x4 acc2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b2 + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = bias;
    }
    for (int ks2 = 0; ks2 < 8; ++ks2) {
        if (ks2 > 0) sync_stage();
        if (ks2 + 1 < 8) stage2(ks2 + 1, (ks2 + 1) & 1);
        const char* A = smem + LDS_A + (ks2 & 1) * 32768 + wm * 4096 + frag_off;
        const char* G = smem + LDS_G + (wn * 64 + r16) * 512 + (((ks2 * 4 + q) ^ r16) * 16);
        bf16x8 bf[4], af[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const bf16x8*)(G + nt * 8192);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = *(const bf16x8*)(A + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = mfma16(af[mt], bf[nt], acc2[mt][nt]);
    }

    // ---------------- epilogue: h' = (h + res) * sqrt(1/2) + emb_next, via an fp32 LDS tile ---------
    __syncthreads();       // every wave is done with the stage buffers and the gate tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = wn * 64 + nt * 16 + r16, ch = wm * 64 + mt * 16 + q * 4;
            *(f32x4*)(smem + t * EPI_PITCH + ch * 4) = acc2[mt][nt];
        }
    __syncthreads();
    {
        char* hout_c = (char*)(a.hout + ((size_t)b * a.LP + kPad + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
            const f32x4 r0 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32);
            const f32x4 r1 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32 + 16);
            const bf16x8 hv = *(const bf16x8*)(hin_c + (size_t)t * 512 + cg * 16);
            const f32x4 e0 = *(const f32x4*)(a.emb_next + cg * 8);
            const f32x4 e1 = *(const f32x4*)(a.emb_next + cg * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (bf16_t)(((float)hv[j] + r0[j]) * 0.70710678118654752440f + e0[j]);
                o[j + 4] = (bf16_t)(((float)hv[j + 4] + r1[j]) * 0.70710678118654752440f + e1[j]);
            }
            *(bf16x8*)(hout_c + (size_t)t * 512 + cg * 16) = o;
        }
    }
}

// skip = sum_n W_skip_n g_n  (K = NL*256)  ->  y = skip * sqrt(1/NL)  ->  relu(W_f0 y + b_f0)
// ->  eps = w_z . (.) + b_z      (WaveNet.py:131-135,160-162)
__global__ void __launch_bounds__(512, 2) wn_final_bf16(WnFinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int tiles_per_clip = a.L / kTileT;
    const int b = blockIdx.x / tiles_per_clip;
    const int t0 = (blockIdx.x - b * tiles_per_clip) * kTileT;
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);
    const int brow = tid >> 2;
    const size_t layer_bytes = (size_t)a.B * a.L * 512;
    const char* bsrc = (const char*)a.g + ((size_t)b * a.L + t0 + brow) * 512 + (((tid & 3) ^ swz64(brow)) * 16);

    auto stage = [&](int ks, int buf) {      // ks = layer*8 + kc
        const char* wsrc = (const char*)a.wsp + (size_t)ks * 16384 + tid * 16;
        char* la = smem + LDS_A + buf * 32768 + wv * 1024;
        glds16(wsrc, la);
        glds16(wsrc + 8192, la + 8192);
        glds16(bsrc + (size_t)(ks >> 3) * layer_bytes + (ks & 7) * 64, smem + LDS_B + buf * 8192 + wv * 1024);
    };
    auto stage3 = [&](int ks, int buf) {
        const char* wsrc = (const char*)a.wf0p + (size_t)ks * 16384 + tid * 16;
        char* la = smem + LDS_A + buf * 32768 + wv * 1024;
        glds16(wsrc, la);
        glds16(wsrc + 8192, la + 8192);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.bskip_sum + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias;
    }
    const int nks = a.NL * 8;
    stage(0, 0);
    for (int ks = 0; ks < nks; ++ks) {
        sync_stage();
        if (ks + 1 < nks) stage(ks + 1, (ks + 1) & 1);
        else stage3(0, (ks + 1) & 1);
        const char* A = smem + LDS_A + (ks & 1) * 32768 + wm * 4096 + frag_off;
        const char* Bt = smem + LDS_B + (ks & 1) * 8192 + wn * 4096 + frag_off;
        bf16x8 bf[4], af[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const bf16x8*)(Bt + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = *(const bf16x8*)(A + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[mt], bf[nt], acc[mt][nt]);
    }
    // y = skip * sqrt(1/NL) -> bf16 -> LDS [t][ch]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 yv;
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = (bf16_t)(acc[mt][nt][r] * a.skip_scale);
            const int t = wn * 64 + nt * 16 + r16;
            const int chunk = wm * 8 + mt * 2 + (q >> 1);
            *(bf16x4*)(smem + LDS_G + t * 512 + ((chunk ^ r16) * 16) + (q & 1) * 8) = yv;
        }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.bf0 + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias;
    }
    const int base = nks & 1;                     // stage3(0) went to buffer (nks & 1)
    for (int ks = 0; ks < 8; ++ks) {
        sync_stage();
        if (ks + 1 < 8) stage3(ks + 1, (base + ks + 1) & 1);
        const char* A = smem + LDS_A + ((base + ks) & 1) * 32768 + wm * 4096 + frag_off;
        const char* G = smem + LDS_G + (wn * 64 + r16) * 512 + (((ks * 4 + q) ^ r16) * 16);
        bf16x8 bf[4], af[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const bf16x8*)(G + nt * 8192);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = *(const bf16x8*)(A + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[mt], bf[nt], acc[mt][nt]);
    }
    // relu, dot with the 256 -> 1 output conv, reduce over channels (registers -> lanes -> waves)
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 wz = *(const f32x4*)(a.wz + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[nt] = fmaf(fmaxf(acc[mt][nt][r], 0.f), wz[r], part[nt]);
    }
    __syncthreads();                              // everyone is done with the LDS tiles
    float* red = (float*)smem;                    // [4 wm][128 t]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float p = part[nt];
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        if (q == 0) red[wm * 128 + wn * 64 + nt * 16 + r16] = p;
    }
    __syncthreads();
    if (tid < 128) {
        const float e = ((red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid])) + a.bz;
        a.eps[(size_t)b * a.L + t0 + tid] = e;
    }
}

// h0 = relu(w_init * x + b_init) + emb_0  (WaveNet.py:147,13-19 + the F5 alias of layer 0), bf16 out
__global__ void __launch_bounds__(256) wn_init_bf16(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ emb0,
                                                    bf16_t* __restrict__ h, int L, int LP, long total_chunks) {
    // one thread per 8-channel chunk of one time position
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_chunks;
         idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx & 31);
        const long pos = idx >> 5;                 // b * L + t
        const long bb = pos / L, t = pos - bb * L;
        const float xv = x[pos];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            const float v = w[c] * xv + bias[c];
            o[j] = (bf16_t)(fmaxf(v, 0.f) + emb0[c]);
        }
        *(bf16x8*)(h + ((bb * LP + kPad + t) * kC + cg * 8)) = o;
    }
}

void launch_wn_layer_bf16(const WnLayerArgs& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(wn_layer_bf16, dim3(B * (a.L / kTileT)), dim3(512), kWnLdsBytes, s, a);
}
void launch_wn_final_bf16(const WnFinalArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(wn_final_bf16, dim3(a.B * (a.L / kTileT)), dim3(512), kWnLdsBytes, s, a);
}
void launch_wn_init_bf16(const float* x, const float* w, const float* bias, const float* emb0, bf16_t* h, int B, int L,
                         int LP, hipStream_t s) {
    const long chunks = (long)B * L * 32;
    const int grid = (int)((chunks + 255) / 256 < 8192 ? (chunks + 255) / 256 : 8192);
    hipLaunchKernelGGL(wn_init_bf16, dim3(grid), dim3(256), 0, s, x, w, bias, emb0, h, L, LP, chunks);
}
int wn_bf16_configure() {
    hipError_t e = hipFuncSetAttribute((const void*)wn_layer_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_final_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    return (int)e;
}

}  // namespace dmad
