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
#include <type_traits>

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

template <int ABL>
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
        if (ABL != 2) {
            if (ks + 1 < 24) stage1(ks + 1, (ks + 1) & 1);
            else if (!a.last) stage2(0, 0);
        }
        const char* A = smem + LDS_A + (ks & 1) * 32768 + wm * 8192 + frag_off;
        const char* Bt = smem + LDS_B + (ks & 1) * 8192 + wn * 4096 + frag_off;
        bf16x8 bf[4], af[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const bf16x8*)(Bt + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) af[mt] = *(const bf16x8*)(A + mt * 1024);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL == 3) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt][0][0] += (float)af[mt][0] + (float)bf[mt & 3][1];
        } else {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[mt], bf[nt], acc[mt][nt]);
        }
    }

    // ---------------- gate: g[ch][t] -> LDS [t][ch] bf16 ------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = (bf16_t)(ABL == 1 ? acc[mt][nt][r] + acc[mt + 4][nt][r] : gate_fn(acc[mt][nt][r], acc[mt + 4][nt][r]));
            const int t = wn * 64 + nt * 16 + r16;
            const int chunk = wm * 8 + mt * 2 + (q >> 1);
            *(bf16x4*)(smem + LDS_G + t * 512 + ((chunk ^ r16) * 16) + (q & 1) * 8) = gv;
        }
    }
    sync_stage();          // gate tile complete; (stage2(0) landed)

    // ---------------- stream g to HBM (coalesced 512-B rows) --------------------------------------
    if (ABL != 5) {
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            const uint4 v = *(const uint4*)(smem + LDS_G + t * 512 + ((c ^ (t & 15)) * 16));
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = v;
        }
    }
    if (a.last || ABL == 4) return;    // the last layer's residual output is never consumed (WaveNet.py:131-135)

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


// ================================================================================================
// wn_layer_bf16_v2 — same math and layouts as wn_layer_bf16, deeper software pipeline:
//   * 3-slot LDS ring for the GEMM1 stages: the global_load_lds of k-step ks+3 is issued while k-step
//     ks computes (2 k-steps of latency budget, counted vmcnt, never drained inside the loop);
//   * MFMA operand fragments double-buffered in registers: the 12 ds_read_b128 of k-step ks+1 are
//     issued before the 32 MFMAs of k-step ks;
//   * the gate tile aliases ring slots 0-1 (free by then), GEMM2 streams its 8 weight stages through
//     4 buffers with 3 barriers, the residual rows h are prefetched into registers before GEMM2,
//     and no barrier waits for an outstanding store.
// LDS map: ring slot s at s*40960 (A 32 KiB + B 8 KiB), GEMM2 weight buffers {122880, 139264, 81920,
// 98304}, gate tile [0, 65536), epilogue tile [0, 133120).
// ================================================================================================
namespace {
constexpr int V2_SLOT = 40960, V2_BOFF = 32768;
constexpr int V2_A2_0 = 122880, V2_A2_1 = 139264, V2_A2_2 = 81920, V2_A2_3 = 98304;
// s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier through the builtins, so that hipcc's own wait-count
// bookkeeping knows the LDS reads have retired (an asm wait is invisible to it and it would re-wait
// lgkmcnt(0) AFTER the next k-step's fragment reads have been issued).  N <= 15.
#define DMAD_WAIT_BARRIER(N)                                        \
    do {                                                            \
        asm volatile("" ::: "memory");                              \
        __builtin_amdgcn_s_waitcnt(0x0070 | (N));                   \
        __builtin_amdgcn_s_barrier();                               \
        asm volatile("" ::: "memory");                              \
    } while (0)
#define DMAD_BARRIER_LGKM()                                         \
    do {                                                            \
        asm volatile("" ::: "memory");                              \
        __builtin_amdgcn_s_waitcnt(0xC07F);                         \
        __builtin_amdgcn_s_barrier();                               \
        asm volatile("" ::: "memory");                              \
    } while (0)
}  // namespace

template <bool STAMP, int ABL2 = 0>
__global__ void __launch_bounds__(512, 2) wn_layer_bf16_v2(WnLayerArgs a) {
    // STAMP: diagnostic build only (phase timestamps of wave 0 into a.dbg; never the timed/shipped kernel)
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            if (threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
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
    const char* w1src = (const char*)a.w1p + tid * 16;
    const char* w2src = (const char*)a.w2p + tid * 16;

    auto stage1 = [&](int ks, int slot) {
        const char* wsrc = w1src + (size_t)ks * 32768;
        char* la = smem + slot * V2_SLOT + wv * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (ABL2 == 2) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + i * 8192), (lds_ptr_t)(la + i * 8192), 16, 0, 2);
            else if constexpr (ABL2 == 3) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + i * 8192), (lds_ptr_t)(la + i * 8192), 16, 0, 1);
            else glds16(wsrc + i * 8192, la + i * 8192);
        }
        const int tap = ks >> 3, kc = ks & 7;
        if constexpr (ABL2 != 4) glds16(bsrc + (tap - 1) * tap_bytes + kc * 64, smem + slot * V2_SLOT + V2_BOFF + wv * 1024);
        else glds16(bsrc + kc * 64, smem + slot * V2_SLOT + V2_BOFF + wv * 1024);
    };
    auto stage2 = [&](int ks2, int off) {
        const char* wsrc = w2src + (size_t)ks2 * 16384;
        char* la = smem + off + wv * 1024;
        glds16(wsrc, la);
        glds16(wsrc + 8192, la + 8192);
    };
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);

    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b1 + wm * 128 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias;
    }

    bf16x8 af[2][8], bf[2][4];
    auto read_frags = [&](int slot, int set) {
        const char* A = smem + slot * V2_SLOT + wm * 8192 + frag_off;
        const char* Bt = smem + slot * V2_SLOT + V2_BOFF + wn * 4096 + frag_off;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[set][nt] = *(const bf16x8*)(Bt + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) af[set][mt] = *(const bf16x8*)(A + mt * 1024);
    };

    // ---------------- GEMM1 ----------------------------------------------------------------------
    stage1(0, 0);
    stage1(1, 1);
    stage1(2, 2);
    DMAD_WAIT_BARRIER(10);                 // stage 0 landed everywhere
    stamp(1);
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        const int cur = ks & 1;
        if (ks <= 21) { DMAD_WAIT_BARRIER(5); }        // stage ks+1 landed; stage ks+2 may still fly
        else if (ks == 22) { DMAD_WAIT_BARRIER(0); }   // stage 23 landed
        if constexpr (ABL2 != 6 && ABL2 != 7) {
            if (ks + 3 < 24) stage1(ks + 3, ks % 3);
        }
        if constexpr (ABL2 != 1 && ABL2 != 7) {
            if (ks + 1 < 24) read_frags((ks + 1) % 3, cur ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL2 != 1 && ABL2 != 5) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[ABL2 == 7 ? 0 : cur][mt], bf[ABL2 == 7 ? 0 : cur][nt], acc[mt][nt]);
        } else if constexpr (ABL2 == 5) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) asm volatile("" ::"v"(af[cur][mt]), "v"(bf[cur][mt & 3]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    stamp(2);
    if (!a.last) {                         // the first two GEMM2 weight stages land under the gate math
        stage2(0, V2_A2_0);
        stage2(1, V2_A2_1);
    }
    // ---------------- gate -> LDS [0, 64K) (ring slots 0-1 are free since the barrier of k-step 22) --
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = (bf16_t)gate_fn(acc[mt][nt][r], acc[mt + 4][nt][r]);
            const int t = wn * 64 + nt * 16 + r16;
            const int chunk = wm * 8 + mt * 2 + (q >> 1);
            *(bf16x4*)(smem + t * 512 + ((chunk ^ r16) * 16) + (q & 1) * 8) = gv;
            __builtin_amdgcn_sched_barrier(0);     // keep the 16 tiles' gate math from being interleaved (VGPR pressure)
        }
    }
    if (a.last) {
        DMAD_BARRIER_LGKM();
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = *(const uint4*)(smem + t * 512 + ((c ^ (t & 15)) * 16));
        }
        return;
    }
    stamp(3);
    DMAD_WAIT_BARRIER(0);                  // gate tile complete, GEMM2 stages 0-1 landed, ring slot 2 free
    stamp(4);

    // residual rows for the epilogue: prefetch now, consumed after GEMM2
    bf16x8 hv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
        hv[it] = *(const bf16x8*)(hin_c + (size_t)t * 512 + cg * 16);
    }
    stage2(2, V2_A2_2);
    stage2(3, V2_A2_3);

    // ---------------- GEMM2 ----------------------------------------------------------------------
    f32x4 acc2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b2 + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = bias;
    }
    auto compute2 = [&](int ks2, int off) {
        const char* A = smem + off + wm * 4096 + frag_off;
        const char* G = smem + (wn * 64 + r16) * 512 + (((ks2 * 4 + q) ^ r16) * 16);
        bf16x8 b2[4], a2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b2[nt] = *(const bf16x8*)(G + nt * 8192);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a2[mt] = *(const bf16x8*)(A + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = mfma16(a2[mt], b2[nt], acc2[mt][nt]);
    };
    compute2(0, V2_A2_0);
    compute2(1, V2_A2_1);
    DMAD_WAIT_BARRIER(0);                  // stages 2-3 (and the h rows) landed; buffers 0-1 free
    stage2(4, V2_A2_0);
    stage2(5, V2_A2_1);
    compute2(2, V2_A2_2);
    compute2(3, V2_A2_3);
    DMAD_WAIT_BARRIER(0);
    stage2(6, V2_A2_2);
    stage2(7, V2_A2_3);
    {   // stream the gate tile to HBM; issued after the last loads so that no later wait covers these stores
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = *(const uint4*)(smem + t * 512 + ((c ^ (t & 15)) * 16));
        }
    }
    compute2(4, V2_A2_0);
    compute2(5, V2_A2_1);
    DMAD_WAIT_BARRIER(8);                  // stages 6-7 landed (the 8 gate-tile stores may still fly)
    compute2(6, V2_A2_2);
    compute2(7, V2_A2_3);

    // ---------------- epilogue -------------------------------------------------------------------
    stamp(5);
    DMAD_BARRIER_LGKM();                   // every wave is done with the gate tile and the weight buffers
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = wn * 64 + nt * 16 + r16, ch = wm * 64 + mt * 16 + q * 4;
            *(f32x4*)(smem + t * EPI_PITCH + ch * 4) = acc2[mt][nt];
        }
    DMAD_BARRIER_LGKM();
    {
        char* hout_c = (char*)(a.hout + ((size_t)b * a.LP + kPad + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
            const f32x4 r0 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32);
            const f32x4 r1 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32 + 16);
            const f32x4 e0 = *(const f32x4*)(a.emb_next + cg * 8);
            const f32x4 e1 = *(const f32x4*)(a.emb_next + cg * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (bf16_t)(((float)hv[it][j] + r0[j]) * 0.70710678118654752440f + e0[j]);
                o[j + 4] = (bf16_t)(((float)hv[it][j + 4] + r1[j]) * 0.70710678118654752440f + e1[j]);
            }
            *(bf16x8*)(hout_c + (size_t)t * 512 + cg * 16) = o;
        }
    }
    stamp(6);
}

namespace {
constexpr int V3_A2 = 65536;          // six 16 KiB GEMM2 weight buffers at [64K, 160K); gate tile at [0, 64K)
}
// wn_layer_bf16_v3 = v2 + (a) the two waves of a SIMD staggered inside each k-step, (b) GEMM2 weight stages
// 0-5 prefetched under the gate math into six buffers (two barriers instead of three + no exposed load).
template <bool STAMP>
__global__ void __launch_bounds__(512, 2) wn_layer_bf16_v3(WnLayerArgs a) {
    // STAMP: diagnostic build only (phase timestamps of wave 0 into a.dbg; never the timed/shipped kernel)
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            if (threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
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
    const char* w1src = (const char*)a.w1p + tid * 16;
    const char* w2src = (const char*)a.w2p + tid * 16;

    auto stage1 = [&](int ks, int slot) {
        const char* wsrc = w1src + (size_t)ks * 32768;
        char* la = smem + slot * V2_SLOT + wv * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wsrc + i * 8192, la + i * 8192);
        const int tap = ks >> 3, kc = ks & 7;
        glds16(bsrc + (tap - 1) * tap_bytes + kc * 64, smem + slot * V2_SLOT + V2_BOFF + wv * 1024);
    };
    auto stage2 = [&](int ks2, int off) {
        const char* wsrc = w2src + (size_t)ks2 * 16384;
        char* la = smem + off + wv * 1024;
        glds16(wsrc, la);
        glds16(wsrc + 8192, la + 8192);
    };
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);

    auto body = [&](auto late_tag) {
    constexpr bool late = decltype(late_tag)::value;
    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b1 + wm * 128 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias;
    }

    bf16x8 af[2][8], bf[2][4];
    auto read_frags = [&](int slot, int set) {
        const char* A = smem + slot * V2_SLOT + wm * 8192 + frag_off;
        const char* Bt = smem + slot * V2_SLOT + V2_BOFF + wn * 4096 + frag_off;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[set][nt] = *(const bf16x8*)(Bt + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) af[set][mt] = *(const bf16x8*)(A + mt * 1024);
    };

    // ---------------- GEMM1 ----------------------------------------------------------------------
    stage1(0, 0);
    stage1(1, 1);
    stage1(2, 2);
    DMAD_WAIT_BARRIER(10);                 // stage 0 landed everywhere
    stamp(1);
    read_frags(0, 0);
    // Waves 4-7 share their SIMD with waves 0-3.  Stagger them inside every k-step: waves 0-3 issue their
    // memory block (next stage DMA + next fragments) first and then 32 MFMAs, waves 4-7 start with 16 MFMAs,
    // then the memory block, then the other 16, so the matrix pipe always has a wave that is ready.
#define DMAD_MFMA_ROWS(SET, M0, M1)                                                                   \
    _Pragma("unroll") for (int mt = (M0); mt < (M1); ++mt)                                           \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[SET][mt], bf[SET][nt], acc[mt][nt]);
    {
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const int cur = ks & 1;
            if (ks <= 21) { DMAD_WAIT_BARRIER(5); }        // stage ks+1 landed; stage ks+2 may still fly
            else if (ks == 22) { DMAD_WAIT_BARRIER(0); }   // stage 23 landed
            else { DMAD_BARRIER_LGKM(); }                  // every wave has its last fragments: the ring is free
            if constexpr (late) {
                DMAD_MFMA_ROWS(cur, 0, 4)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + 3 < 24) stage1(ks + 3, ks % 3);
            if (ks + 1 < 24) read_frags((ks + 1) % 3, cur ^ 1);
            if (ks == 23 && !a.last) {                     // all GEMM2 weight stages 0-5 land under the gate math
#pragma unroll
                for (int i = 0; i < 6; ++i) stage2(i, V3_A2 + i * 16384);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (late) {
                DMAD_MFMA_ROWS(cur, 4, 8)
            } else {
                DMAD_MFMA_ROWS(cur, 0, 8)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    stamp(2);
    // ---------------- gate -> LDS [0, 64K) (ring slots 0-1 are free since the barrier of k-step 22) --
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = (bf16_t)gate_fn(acc[mt][nt][r], acc[mt + 4][nt][r]);
            const int t = wn * 64 + nt * 16 + r16;
            const int chunk = wm * 8 + mt * 2 + (q >> 1);
            *(bf16x4*)(smem + t * 512 + ((chunk ^ r16) * 16) + (q & 1) * 8) = gv;
            __builtin_amdgcn_sched_barrier(0);     // keep the 16 tiles' gate math from being interleaved (VGPR pressure)
        }
    }
    if (a.last) {
        DMAD_BARRIER_LGKM();
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = *(const uint4*)(smem + t * 512 + ((c ^ (t & 15)) * 16));
        }
        return;
    }
    stamp(3);
    DMAD_WAIT_BARRIER(0);                  // gate tile complete, GEMM2 stages 0-5 landed
    stamp(4);

    // residual rows for the epilogue: prefetch now, consumed after GEMM2
    bf16x8 hv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
        hv[it] = *(const bf16x8*)(hin_c + (size_t)t * 512 + cg * 16);
    }
    const f32x4 e0 = *(const f32x4*)(a.emb_next + (tid & 31) * 8);
    const f32x4 e1 = *(const f32x4*)(a.emb_next + (tid & 31) * 8 + 4);

    // ---------------- GEMM2: 8 weight stages through 6 buffers, 2 barriers ---------------------------
    f32x4 acc2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b2 + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = bias;
    }
    auto compute2 = [&](int ks2, int off) {
        const char* A = smem + off + wm * 4096 + frag_off;
        const char* G = smem + (wn * 64 + r16) * 512 + (((ks2 * 4 + q) ^ r16) * 16);
        bf16x8 b2[4], a2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b2[nt] = *(const bf16x8*)(G + nt * 8192);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a2[mt] = *(const bf16x8*)(A + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = mfma16(a2[mt], b2[nt], acc2[mt][nt]);
    };
    compute2(0, V3_A2);
    compute2(1, V3_A2 + 16384);
    DMAD_BARRIER_LGKM();                   // buffers 0-1 are free
    stage2(6, V3_A2);
    stage2(7, V3_A2 + 16384);
    {   // stream the gate tile to HBM; issued after the last loads so that no later wait covers these stores
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = *(const uint4*)(smem + t * 512 + ((c ^ (t & 15)) * 16));
        }
    }
    compute2(2, V3_A2 + 2 * 16384);
    compute2(3, V3_A2 + 3 * 16384);
    compute2(4, V3_A2 + 4 * 16384);
    compute2(5, V3_A2 + 5 * 16384);
    DMAD_WAIT_BARRIER(8);                  // stages 6-7 landed (the 8 gate-tile stores may still fly)
    compute2(6, V3_A2);
    compute2(7, V3_A2 + 16384);

    // ---------------- epilogue -------------------------------------------------------------------
    stamp(5);
    DMAD_BARRIER_LGKM();                   // every wave is done with the gate tile and the weight buffers
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = wn * 64 + nt * 16 + r16, ch = wm * 64 + mt * 16 + q * 4;
            *(f32x4*)(smem + t * EPI_PITCH + ch * 4) = acc2[mt][nt];
        }
    DMAD_BARRIER_LGKM();
    {
        char* hout_c = (char*)(a.hout + ((size_t)b * a.LP + kPad + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
            const f32x4 r0 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32);
            const f32x4 r1 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32 + 16);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (bf16_t)(((float)hv[it][j] + r0[j]) * 0.70710678118654752440f + e0[j]);
                o[j + 4] = (bf16_t)(((float)hv[it][j + 4] + r1[j]) * 0.70710678118654752440f + e1[j]);
            }
            *(bf16x8*)(hout_c + (size_t)t * 512 + cg * 16) = o;
        }
    }
    stamp(6);
    };   // body
    if (wv >= 4) body(std::true_type{});
    else body(std::false_type{});
}

// wn_layer_bf16_v4 = v3's GEMM2/epilogue + v2's ring, with the DMA pieces and fragment reads of a k-step
// interleaved between its MFMAs (no wave stagger).
template <bool STAMP>
__global__ void __launch_bounds__(512, 2) wn_layer_bf16_v4(WnLayerArgs a) {
    // STAMP: diagnostic build only (phase timestamps of wave 0 into a.dbg; never the timed/shipped kernel)
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            if (threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
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
    const char* w1src = (const char*)a.w1p + tid * 16;
    const char* w2src = (const char*)a.w2p + tid * 16;

    auto stage1 = [&](int ks, int slot) {
        const char* wsrc = w1src + (size_t)ks * 32768;
        char* la = smem + slot * V2_SLOT + wv * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wsrc + i * 8192, la + i * 8192);
        const int tap = ks >> 3, kc = ks & 7;
        glds16(bsrc + (tap - 1) * tap_bytes + kc * 64, smem + slot * V2_SLOT + V2_BOFF + wv * 1024);
    };
    auto stage2 = [&](int ks2, int off) {
        const char* wsrc = w2src + (size_t)ks2 * 16384;
        char* la = smem + off + wv * 1024;
        glds16(wsrc, la);
        glds16(wsrc + 8192, la + 8192);
    };
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);

    {
    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b1 + wm * 128 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias;
    }

    bf16x8 af[2][8], bf[2][4];
    auto read_frags = [&](int slot, int set) {
        const char* A = smem + slot * V2_SLOT + wm * 8192 + frag_off;
        const char* Bt = smem + slot * V2_SLOT + V2_BOFF + wn * 4096 + frag_off;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[set][nt] = *(const bf16x8*)(Bt + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) af[set][mt] = *(const bf16x8*)(A + mt * 1024);
    };

    // ---------------- GEMM1 ----------------------------------------------------------------------
    stage1(0, 0);
    stage1(1, 1);
    stage1(2, 2);
    DMAD_WAIT_BARRIER(10);                 // stage 0 landed everywhere
    stamp(1);
    read_frags(0, 0);
    // Waves 4-7 share their SIMD with waves 0-3.  Stagger them inside every k-step: waves 0-3 issue their
    // memory block (next stage DMA + next fragments) first and then 32 MFMAs, waves 4-7 start with 16 MFMAs,
    // then the memory block, then the other 16, so the matrix pipe always has a wave that is ready.
#define DMAD_MFMA_ROWS(SET, M0, M1)                                                                   \
    _Pragma("unroll") for (int mt = (M0); mt < (M1); ++mt)                                           \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(af[SET][mt], bf[SET][nt], acc[mt][nt]);
    // One k-step = 32 MFMAs per wave.  The vector-memory path of a CU moves 64 B/clk, so the 40 KiB of the
    // next stage keep it busy for ~640 of the k-step's 1024 matrix cycles: the 5 DMA pieces and the 12
    // fragment reads are therefore spread between the MFMAs instead of being issued as one block.
    {
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks <= 21) { DMAD_WAIT_BARRIER(5); }        // stage ks+1 landed; stage ks+2 may still fly
            else if (ks == 22) { DMAD_WAIT_BARRIER(0); }   // stage 23 landed
            else { DMAD_BARRIER_LGKM(); }                  // every wave has its last fragments: the ring is free
            const int slot_w = ks % 3, slot_r = (ks + 1) % 3;
            const char* wsrc = w1src + (size_t)(ks + 3) * 32768;
            char* la = smem + slot_w * V2_SLOT + wv * 1024;
            const char* Ar = smem + slot_r * V2_SLOT + wm * 8192 + frag_off;
            const char* Br = smem + slot_r * V2_SLOT + V2_BOFF + wn * 4096 + frag_off;
#pragma unroll
            for (int p = 0; p < 5; ++p) {                  // 5 x (1 DMA piece, 4 MFMAs)
                if (ks + 3 < 24) {
                    if (p < 4) glds16(wsrc + p * 8192, la + p * 8192);
                    else glds16(bsrc + (((ks + 3) >> 3) - 1) * tap_bytes + ((ks + 3) & 7) * 64,
                                smem + slot_w * V2_SLOT + V2_BOFF + wv * 1024);
                } else if (ks == 23 && !a.last && p < 3) { // GEMM2 weight stages 0-5 land under the gate math
                    stage2(2 * p, V3_A2 + (2 * p) * 16384);
                    stage2(2 * p + 1, V3_A2 + (2 * p + 1) * 16384);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 4 * p; i < 4 * p + 4; ++i) acc[i >> 2][i & 3] = mfma16(af[cur][i >> 2], bf[cur][i & 3], acc[i >> 2][i & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < 12; ++p) {                 // 12 x (1 fragment read for k-step ks+1, 1 MFMA)
                if (ks + 1 < 24) {
                    if (p < 4) bf[nxt][p] = *(const bf16x8*)(Br + p * 1024);
                    else af[nxt][p - 4] = *(const bf16x8*)(Ar + (p - 4) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int i = 20 + p;
                acc[i >> 2][i & 3] = mfma16(af[cur][i >> 2], bf[cur][i & 3], acc[i >> 2][i & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    stamp(2);
    // ---------------- gate -> LDS [0, 64K) (ring slots 0-1 are free since the barrier of k-step 22) --
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = (bf16_t)gate_fn(acc[mt][nt][r], acc[mt + 4][nt][r]);
            const int t = wn * 64 + nt * 16 + r16;
            const int chunk = wm * 8 + mt * 2 + (q >> 1);
            *(bf16x4*)(smem + t * 512 + ((chunk ^ r16) * 16) + (q & 1) * 8) = gv;
            __builtin_amdgcn_sched_barrier(0);     // keep the 16 tiles' gate math from being interleaved (VGPR pressure)
        }
    }
    if (a.last) {
        DMAD_BARRIER_LGKM();
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = *(const uint4*)(smem + t * 512 + ((c ^ (t & 15)) * 16));
        }
        return;
    }
    stamp(3);
    DMAD_WAIT_BARRIER(0);                  // gate tile complete, GEMM2 stages 0-5 landed
    stamp(4);

    // residual rows for the epilogue: prefetch now, consumed after GEMM2
    bf16x8 hv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
        hv[it] = *(const bf16x8*)(hin_c + (size_t)t * 512 + cg * 16);
    }
    const f32x4 e0 = *(const f32x4*)(a.emb_next + (tid & 31) * 8);
    const f32x4 e1 = *(const f32x4*)(a.emb_next + (tid & 31) * 8 + 4);

    // ---------------- GEMM2: 8 weight stages through 6 buffers, 2 barriers ---------------------------
    f32x4 acc2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bias = *(const f32x4*)(a.b2 + wm * 64 + mt * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = bias;
    }
    auto compute2 = [&](int ks2, int off) {
        const char* A = smem + off + wm * 4096 + frag_off;
        const char* G = smem + (wn * 64 + r16) * 512 + (((ks2 * 4 + q) ^ r16) * 16);
        bf16x8 b2[4], a2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b2[nt] = *(const bf16x8*)(G + nt * 8192);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a2[mt] = *(const bf16x8*)(A + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = mfma16(a2[mt], b2[nt], acc2[mt][nt]);
    };
    compute2(0, V3_A2);
    compute2(1, V3_A2 + 16384);
    DMAD_BARRIER_LGKM();                   // buffers 0-1 are free
    stage2(6, V3_A2);
    stage2(7, V3_A2 + 16384);
    {   // stream the gate tile to HBM; issued after the last loads so that no later wait covers these stores
        char* gdst = (char*)(a.gout + ((size_t)b * a.L + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, c = idx & 31;
            *(uint4*)(gdst + (size_t)t * 512 + c * 16) = *(const uint4*)(smem + t * 512 + ((c ^ (t & 15)) * 16));
        }
    }
    compute2(2, V3_A2 + 2 * 16384);
    compute2(3, V3_A2 + 3 * 16384);
    compute2(4, V3_A2 + 4 * 16384);
    compute2(5, V3_A2 + 5 * 16384);
    DMAD_WAIT_BARRIER(8);                  // stages 6-7 landed (the 8 gate-tile stores may still fly)
    compute2(6, V3_A2);
    compute2(7, V3_A2 + 16384);

    // ---------------- epilogue -------------------------------------------------------------------
    stamp(5);
    DMAD_BARRIER_LGKM();                   // every wave is done with the gate tile and the weight buffers
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = wn * 64 + nt * 16 + r16, ch = wm * 64 + mt * 16 + q * 4;
            *(f32x4*)(smem + t * EPI_PITCH + ch * 4) = acc2[mt][nt];
        }
    DMAD_BARRIER_LGKM();
    {
        char* hout_c = (char*)(a.hout + ((size_t)b * a.LP + kPad + t0) * kC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tid, t = idx >> 5, cg = idx & 31;
            const f32x4 r0 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32);
            const f32x4 r1 = *(const f32x4*)(smem + t * EPI_PITCH + cg * 32 + 16);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (bf16_t)(((float)hv[it][j] + r0[j]) * 0.70710678118654752440f + e0[j]);
                o[j + 4] = (bf16_t)(((float)hv[it][j + 4] + r1[j]) * 0.70710678118654752440f + e1[j]);
            }
            *(bf16x8*)(hout_c + (size_t)t * 512 + cg * 16) = o;
        }
    }
    stamp(6);
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

void launch_wn_layer_bf16(const WnLayerArgs& a, int B, hipStream_t s, int variant) {
    if (variant == 40 || variant == 41) {
        launch_wn_layer_bf16_p(a, B, s, variant == 41);
        return;
    }
    const dim3 grid(B * (a.L / kTileT)), blk(512);
    switch (variant) {
        case 1: hipLaunchKernelGGL(wn_layer_bf16<1>, grid, blk, kWnLdsBytes, s, a); break;
        case 2: hipLaunchKernelGGL(wn_layer_bf16<2>, grid, blk, kWnLdsBytes, s, a); break;
        case 3: hipLaunchKernelGGL(wn_layer_bf16<3>, grid, blk, kWnLdsBytes, s, a); break;
        case 4: hipLaunchKernelGGL(wn_layer_bf16<4>, grid, blk, kWnLdsBytes, s, a); break;
        case 5: hipLaunchKernelGGL(wn_layer_bf16<5>, grid, blk, kWnLdsBytes, s, a); break;
        case 10: hipLaunchKernelGGL(wn_layer_bf16_v2<false>, grid, blk, kWnLdsBytesV2, s, a); break;
        case 11: hipLaunchKernelGGL(wn_layer_bf16_v2<true>, grid, blk, kWnLdsBytesV2, s, a); break;
        case 12: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 1>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 13: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 2>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 14: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 3>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 15: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 4>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 16: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 5>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 17: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 6>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 18: hipLaunchKernelGGL((wn_layer_bf16_v2<false, 7>), grid, blk, kWnLdsBytesV2, s, a); break;
        case 30: hipLaunchKernelGGL(wn_layer_bf16_v4<false>, grid, blk, kWnLdsBytesV3, s, a); break;
        case 31: hipLaunchKernelGGL(wn_layer_bf16_v4<true>, grid, blk, kWnLdsBytesV3, s, a); break;
        case 20: hipLaunchKernelGGL(wn_layer_bf16_v3<false>, grid, blk, kWnLdsBytesV3, s, a); break;
        case 21: hipLaunchKernelGGL(wn_layer_bf16_v3<true>, grid, blk, kWnLdsBytesV3, s, a); break;
        default: hipLaunchKernelGGL(wn_layer_bf16<0>, grid, blk, kWnLdsBytes, s, a);
    }
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
    if (int r = wn_layer_p_configure()) return r;
    if (int r = wn_final_p_configure()) return r;
    hipError_t e = hipSuccess;
    const void* fns[] = {(const void*)wn_layer_bf16<0>, (const void*)wn_layer_bf16<1>, (const void*)wn_layer_bf16<2>,
                         (const void*)wn_layer_bf16<3>, (const void*)wn_layer_bf16<4>, (const void*)wn_layer_bf16<5>};
    for (const void* f : fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
        if (e != hipSuccess) return (int)e;
    }
    e = hipFuncSetAttribute((const void*)wn_layer_bf16_v2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV2);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_layer_bf16_v2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV2);
    if (e != hipSuccess) return (int)e;
    {
        const void* fv[] = {(const void*)wn_layer_bf16_v2<false, 1>, (const void*)wn_layer_bf16_v2<false, 2>, (const void*)wn_layer_bf16_v2<false, 3>,
                            (const void*)wn_layer_bf16_v2<false, 4>, (const void*)wn_layer_bf16_v2<false, 5>,
                            (const void*)wn_layer_bf16_v2<false, 6>, (const void*)wn_layer_bf16_v2<false, 7>};
        for (const void* f : fv) {
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV2);
            if (e != hipSuccess) return (int)e;
        }
    }
    e = hipFuncSetAttribute((const void*)wn_layer_bf16_v3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV3);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_layer_bf16_v3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV3);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_layer_bf16_v4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV3);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_layer_bf16_v4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytesV3);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_final_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    return (int)e;
}

}  // namespace dmad
