// fp32 gather-GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains).
//
//   C[n][m] = act( scale[m] * sum_tap sum_k A[tap][m][k] * X[row(n, tap)][k] + shift[m] (+ res[n][m]) )
//
// One kernel serves every exact-fp32 GEMM-shaped op of the path:
//   * parity-mode WaveNet: dilated Conv1d as 3 row-shifted taps over the zero-padded residual
//     stream (WaveNet.py:23-34,86) and the 1x1 res/skip/final convs (WaveNet.py:66-72,160-162);
//   * mel front-end: the windowed DFT of the 32 overlapping frames of a clip (row stride = hop) and
//     the slaney filterbank product (torchaudio MelSpectrogram, certified_robustness_eval.py:85);
//   * VGG19_bn / ResNeXt29 / the Improved-Diffusion UNet: 3x3 (stride 1 or 2), 1x1 and grouped convs as implicit GEMM
//     over NHWC activations with bias or folded eval-mode BatchNorm in scale/shift, and the Linear layers.
// Tile 128(M) x 128(N) — 64(M) x 128(N) for layers / conv groups with at most 64 output channels —, k-steps of 16 floats,
// 4 waves (2x2, wave tile 64x64 = 16 accumulator tiles).  Both operands are
// row gathers of 64-byte k-chunks: every lane of an LDS-DMA instruction (global_load_lds_dwordx4) fetches the 16 bytes
// that belong at its own LDS position — rows of 64 B with the 16-byte chunks XOR-swizzled by swz64(row), so a fragment
// is ONE conflict-free ds_read_b128 (4 consecutive k of one row) feeding 4 MFMAs; lane (row, q) holds k = 4q + j for
// MFMA j in BOTH operands, so each MFMA still contracts 4 distinct k and the 4 together cover the 16.  Rows outside the
// problem (conv zero padding, M / N tails) are fetched from a zero page.  3-slot LDS ring, counted vmcnt, one barrier
// per k-step, no register staging.
#include "gemm_f32.h"
#include "gemm_x3_ablate.h"

namespace dmad {

namespace {
constexpr int BN = 128, BK = 16;          // BM = 128, or 64 for layers / conv groups with at most 64 output channels
constexpr int SLOT = 16384;                       // 128 A rows + 128 X rows of 64 B
__device__ __attribute__((aligned(64))) float g_zero_page[16];                 // 64 B of zeros: source of every out-of-problem row


#define GF_WAIT_BARRIER(N)                                                          \
    do {                                                                            \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                       \
        __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14));       \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    } while (0)

// LDS-DMA in its saddr form (as in wn_layer.hip): wave-uniform 64-bit base in an SGPR pair, 32-bit lane offset, wave-uniform LDS
// byte address in M0 (the hardware adds lane * 16).  hipcc does not count these loads: every wait on them is an explicit
// counted s_waitcnt (GF_WAIT_BARRIER).
__device__ __forceinline__ void x3_dma16(const void* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
// ... and in its vaddr form (a 64-bit address per lane): the NHWC conv gathers of the split-f16 kernel, whose out-of-image taps read a
// zero page that may lie anywhere relative to the map
__device__ __forceinline__ void x3_dma16v(const void* vaddr, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory");
}
__device__ __attribute__((aligned(128))) float g_zero_page128[32];             // 128 B of zeros: one staged row of the split-f16 kernel
}  // namespace

// s_waitcnt vmcnt(PIECES * stages) + lgkmcnt(0) + barrier for a run-time number of stages in flight (the deep ring's tail; the deep
// ring exists for 64-row tiles only: 3 pieces per stage, 2 with the narrow tile)
template <int PIECES>
__device__ __forceinline__ void gf_wait_stages(int stages) {
    static_assert(PIECES == 2 || PIECES == 3, "pieces per stage");
    if (PIECES == 3) {
        switch (stages) {
            case 0: GF_WAIT_BARRIER(0); break;
            case 1: GF_WAIT_BARRIER(3); break;
            case 2: GF_WAIT_BARRIER(6); break;
            case 3: GF_WAIT_BARRIER(9); break;
            case 4: GF_WAIT_BARRIER(12); break;
            case 5: GF_WAIT_BARRIER(15); break;
            default: GF_WAIT_BARRIER(18); break;
        }
    } else {
        switch (stages) {
            case 0: GF_WAIT_BARRIER(0); break;
            case 1: GF_WAIT_BARRIER(2); break;
            case 2: GF_WAIT_BARRIER(4); break;
            case 3: GF_WAIT_BARRIER(6); break;
            case 4: GF_WAIT_BARRIER(8); break;
            case 5: GF_WAIT_BARRIER(10); break;
            default: GF_WAIT_BARRIER(12); break;
        }
    }
}

// TWO: the input channels come from two maps (GemmF32Args::X2); an instantiation of its own because the extra test in the staging
//      loop costs the common kernel 2 %.
// NS, NT: ring slots and 16-pixel accumulator tiles per wave along N (tile width 32 NT).  3 / 4 for launches that fill the chip.
//      8 / 1 for launches of fewer workgroups than CUs (the recheck of a few samples on a large engine, whose split-K count is
//      derived from the engine's max batch): there a workgroup's K loop is serial fp32 MFMA work, 32 cycles each, and the launch
//      is bound by ONE workgroup's 64 x 128 tile — 64 x 32 tiles are four times as many workgroups; and alone on its CU a workgroup
//      sees the staging latency over its prefetch distance, hence 7 stages in flight instead of 2.  Same k order, same MFMA
//      sequence per output: the same bits.
template <int BM, bool TWO, int NS = 3, int NT = 4>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32Args a) {
    constexpr int MT = BM / 32;            // 16-row accumulator tiles per wave along M (2 M-waves)
    constexpr int BNT = 32 * NT, XP = NT == 4 ? 2 : 1;      // tile width in pixels; 64-row activation pieces per stage
    static_assert(NT == 4 || (NT == 1 && BM == 64 && !TWO), "the narrow tile is built for the plain 64-row kernel");
    static_assert(NS == 3 || (BM == 64 && NS == 8), "the deep ring is built for 64-row tiles and 8 slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // NS * SLOT bytes
    if (a.groups > 1) {                      // grouped conv: this workgroup's group = blockIdx.z
        const int g = blockIdx.z;
        a.A += (size_t)g * a.taps * a.M * a.K;
        a.X += (size_t)g * a.K;
        a.C += (size_t)g * a.M;
        if (a.scale) a.scale += (size_t)g * a.M;
        if (a.shift) a.shift += (size_t)g * a.M;
        if (a.res) a.res += (size_t)g * a.M;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1, q = lane >> 4, r16 = lane & 15;
    const long n0 = (long)blockIdx.x * BNT;
    const int m0 = blockIdx.y * BM;
    const int ksteps_per_tap = a.K / BK, nks_all = a.taps * ksteps_per_tap;
    const int S = a.splits > 1 ? a.splits : 1, z = a.groups > 1 ? 0 : blockIdx.z;
    const int ks_begin = (int)((long)nks_all * z / S), nks = (int)((long)nks_all * (z + 1) / S);

    // this thread's two staging rows per operand: R = piece * 64 + wave * 16 + lane / 4, chunk = (lane & 3) ^ swz64(R)
    const int rloc = wv * 16 + (lane >> 2), chunk4 = ((lane & 3) ^ swz64(lane >> 2)) * 4;
    const float* zero = g_zero_page;
    const float* arow[2];                     // A row base (tap 0, k 0) or nullptr
    long xbase[2];                            // mode 0: row offset in floats; mode 2: packed (b, y, x)
    int xy[2], xx[2];
    bool xok[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int m = m0 + p * 64 + rloc;
        arow[p] = (p * 64 < BM && m < a.M) ? a.A + (size_t)m * a.K + chunk4 : nullptr;
        const long n = n0 + p * 64 + rloc;
        xok[p] = n < a.N && p * 64 + rloc < BNT;
        xbase[p] = 0; xy[p] = 0; xx[p] = 0;
        if (xok[p]) {
            if (a.mode == 2) {
                const int st = a.stride > 1 ? a.stride : 1;
                const int Ho = (a.H - 1) / st + 1, Wo = (a.W - 1) / st + 1, hw = Ho * Wo;
                const long b = n / hw;
                const int pix = (int)(n - b * hw);
                xy[p] = (pix / Wo) * st; xx[p] = (pix % Wo) * st;
                xbase[p] = b * a.H * a.W;
            } else {
                const long b = n / a.rows_per_batch, r = n - b * a.rows_per_batch;
                xbase[p] = b * a.batch_stride + r * a.row_stride;
            }
        }
    }
    const long ldx = a.ldx ? a.ldx : a.Cin;
    auto stage = [&](int ks, int slot) {
        const int tap = ks / ksteps_per_tap, kc = (ks - tap * ksteps_per_tap) * BK;
        char* la = smem + slot * SLOT + wv * 1024;
#pragma unroll
        for (int p = 0; p < BM / 64; ++p) {
            const float* src = arow[p] ? arow[p] + (size_t)tap * a.M * a.K + kc : zero;
            glds16(src, la + p * 4096);
        }
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const float* src = zero;
            if (xok[p]) {
                if (a.mode == 2) {
                    const int yy = xy[p] + (a.taps == 9 ? tap / 3 - 1 : 0), xq = xx[p] + (a.taps == 9 ? tap % 3 - 1 : 0);
                    if ((unsigned)yy < (unsigned)a.H && (unsigned)xq < (unsigned)a.W) {
                        const long pix = xbase[p] + (long)yy * a.W + xq;
                        src = (TWO && kc >= a.ksplit) ? a.X2 + pix * a.ldx2 + (kc - a.ksplit) + chunk4 : a.X + pix * ldx + kc + chunk4;
                    }
                } else {
                    src = a.X + xbase[p] + (long)(tap - (a.taps >> 1)) * a.tap_stride + kc + chunk4;
                }
            }
            glds16(src, la + 8192 + p * 4096);
        }
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frag = r16 * 64 + ((q ^ swz64(r16)) * 16);
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (ks_begin + i < nks) stage(ks_begin + i, i);
    int slot = 0;
    for (int ks = ks_begin; ks < nks; ++ks) {
        // stage ks landed (the pieces of the NS - 2 stages behind it may still fly); every wave is done reading slot (ks-1) % NS
        if constexpr (NS == 3) {
            if (ks + 1 < nks) { if (BM == 128) { GF_WAIT_BARRIER(4); } else { GF_WAIT_BARRIER(3); } } else { GF_WAIT_BARRIER(0); }
        } else {
            const int behind = nks - 1 - ks;
            gf_wait_stages<BM / 64 + XP>(behind < NS - 2 ? behind : NS - 2);
        }
        if (ks + NS - 1 < nks) stage(ks + NS - 1, slot >= 1 ? slot - 1 : NS - 1);
        const char* As = smem + slot * SLOT + wm * (BM * 32) + frag;
        const char* Bs = smem + slot * SLOT + 8192 + wn * (NT * 1024) + frag;
        f32x4 af[MT], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) af[i] = *(const f32x4*)(As + i * 1024);
            if (i < NT) bf[i] = *(const f32x4*)(Bs + i * 1024);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        slot = slot == NS - 1 ? 0 : slot + 1;
    }

    if (S > 1) {   // split-K: raw partial sums to this split's slab; scale/shift/act happen in the reduce kernel
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * (BM / 2) + i * 16 + q * 4;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const long n = n0 + wn * (NT * 16) + j * 16 + r16;
                if (n >= a.N) continue;
                float* dst = a.slab + ((long)z * a.N + n) * a.ldc + m;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.M) dst[r] = acc[i][j][r];
            }
        }
        return;
    }
    if (BM == 128 && a.epi == 1) {          // gate: g = tanh(H[ch]) * sigmoid(H[256 + ch]) without the H round trip
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mrow = m0 + wm * 64 + i * 16 + q * 4;              // permuted row of the tanh half (its partner: + 32)
            const int ch = blockIdx.y * 64 + wm * 32 + i * 16 + q * 4;
            const float4 bt = *(const float4*)(a.shift + mrow), bs = *(const float4*)(a.shift + mrow + 32);
            const float bta[4] = {bt.x, bt.y, bt.z, bt.w}, bsa[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ht = acc[i][j][r] + bta[r], hs = acc[(i + 2) % MT][j][r] + bsa[r];
                    v[r] = tanhf(ht) * (1.f / (1.f + expf(-hs)));
                }
                *(float4*)(a.C + n * 256 + ch) = float4{v[0], v[1], v[2], v[3]};
            }
        }
        return;
    }
    if (BM == 128 && a.epi == 2) {          // h' = (h + res) * sqrt(1/2) + emb_next (M = 256 res-conv rows)
        long hrow[4];                       // stream row of position n, once per accumulator column (N < 2^31 positions)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned n32 = (unsigned)(n0 + wn * 64 + j * 16 + r16);
            if ((long)n32 >= a.N) n32 = (unsigned)(a.N - 1);                 // (never stored)
            const unsigned bb = n32 / (unsigned)a.L;
            hrow[j] = ((long)bb * a.LP + kPad + (n32 - bb * (unsigned)a.L)) * kC;
        }
        // all residual rows first, then the stores: hin / hout are distinct buffers, which the compiler cannot know — left to it,
        // every load waits (vmcnt(0)) behind the previous tile's store and the epilogue becomes 16 serial HBM round trips
        float4 hv[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[i][j] = *(const float4*)(a.hin + hrow[j] + m0 + wm * 64 + i * 16 + q * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 64 + i * 16 + q * 4;
            const float4 b4 = *(const float4*)(a.shift + m), e4 = *(const float4*)(a.emb_next + m);
            const float ba[4] = {b4.x, b4.y, b4.z, b4.w}, ea[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                const float4 h = hv[i][j];
                const float k = 0.70710678118654752440f;
                *(float4*)(a.hout + hrow[j] + m) = float4{__fadd_rn(__fmul_rn(__fadd_rn(h.x, acc[i][j][0] + ba[0]), k), ea[0]), __fadd_rn(__fmul_rn(__fadd_rn(h.y, acc[i][j][1] + ba[1]), k), ea[1]),
                                                          __fadd_rn(__fmul_rn(__fadd_rn(h.z, acc[i][j][2] + ba[2]), k), ea[2]), __fadd_rn(__fmul_rn(__fadd_rn(h.w, acc[i][j][3] + ba[3]), k), ea[3])};
            }
        }
        return;
    }
    // scale / shift, then per row tile the optional residual with its four loads ahead of the tile's stores (hipcc cannot know that
    // `res` and `C` do not overlap and would serialise load -> wait -> store per accumulator tile), ReLU, stores
    const bool vec = ((a.ldc & 3) == 0);
    long nrow[4];
    bool nok[4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const long n = n0 + wn * (NT * 16) + j * 16 + r16;
        nok[j] = n < a.N;
        nrow[j] = (nok[j] ? n : a.N - 1) * a.ldc;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + q * 4;
        float sc[4], sh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = (a.scale && m + r < a.M) ? a.scale[m + r] : 1.f;
            sh[r] = (a.shift && m + r < a.M) ? a.shift[m + r] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = a.scale ? acc[i][j][r] * sc[r] + sh[r] : acc[i][j][r] + sh[r];
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + q * 4;
        if (a.res) {                         // the four residual chunks of this row tile ahead of its stores
            float rr[4][4];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (vec && m + 3 < a.M) {
                    const float4 r4 = *(const float4*)(a.res + nrow[j] + m);
                    rr[j][0] = r4.x; rr[j][1] = r4.y; rr[j][2] = r4.z; rr[j][3] = r4.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rr[j][r] = m + r < a.M ? a.res[nrow[j] + m + r] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += rr[j][r];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!nok[j]) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a.relu ? relu_nan(acc[i][j][r]) : acc[i][j][r];
            float* dst = a.C + nrow[j] + m;
            if (vec && m + 3 < a.M) {
                *(float4*)dst = float4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.M) dst[r] = v[r];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Split-f16 variant (GemmF32Args::x3): the staging scheme and fragment addresses of the kernel above, but both operands are
// in the split-f16 storage format (dmad_common.h: a 16-byte chunk = 4 values as [hi0 hi1 | hi2 hi3 | lo0 lo1 | lo2 lo3]) and
// the contraction runs on v_mfma_f32_16x16x32_f16: k-steps are consumed in PAIRS — lane (row, q) reads its chunk q of both
// stages, which gives 8 hi and 8 lo halves = one K = 32 fragment of each part (the k-slot assignment is the same for both
// operands, so the contraction is unchanged) — and every product is three MFMAs:
//     main += hi_a * hi_b ;  corr += hi_a * lo_b + lo_a * hi_b ;  result = main + corr * 2^-11.
// No conversion work in the loop: producers write the format once (epilogues below, wn_init / scale kernels).
// Three MFMAs per staged byte make the loop LDS-DMA-latency bound, not matrix bound, unless several k-step pairs are in
// flight: tile 256(M) x 128(N), 8 waves (4 x 2, wave tile 64 x 64, one workgroup per CU), pair = 48 KiB (A 2 x 16 KiB,
// X 2 x 8 KiB), 3-pair ring = 144 KiB of dynamic LDS: two pairs in flight while one is contracted, one barrier per pair.
// ----------------------------------------------------------------------------------------------------------------
constexpr int X3_BM = 256, X3_PAIR = 384 * 128, X3_LDS = 3 * X3_PAIR;
// DIAG (error-attribution builds of tools/gpu_error_attribution.py, never the product launches): GemmF32Args::diag switches
// single roundings of the 16-bit path on inside this fp32-grade pipeline — bit 0: the weights' lo parts are ignored (weights
// = f16(w)), bit 1: the activations' lo parts are ignored (the MFMA eats f16(x), the stored value keeps its 22 bits), bit 2:
// outputs in the split format are written with lo = 0 (the stored gate / residual stream is f16).
// Round 3: a staged row is 128 bytes (32 values = the k-step PAIR in one row), so that eight consecutive lanes of an LDS-DMA
// instruction fetch one whole cache line (64-byte rows asked L2 for every line twice); the eight 16-byte chunks of a row are
// XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 under the hardware's lane groups, tools/lds_bank_check.py); lane
// (row, q) reads chunks q and q + 4: 8 hi and 8 lo halves = one K = 32 fragment of each part, as before.
// The X3A_* hooks are the identity in the product build: gemm_x3_ablate.h (ablation / stamp builds of tools/x3_ablation.sh only).
// Round 5: the kernel also serves NHWC convolutions (CONV: GemmF32Args::mode 2 — 3x3 with zero padding / 1x1, stride 2, two-part input;
// the UNet's convs on the split-f16 tier) and 128-row layers: WM = waves along M (4: tile 256 x 128 as before; 2: tile 128 x 256; the
// same 48 KiB pair = WM weight pieces + 6 - WM activation pieces of 8 KiB, the same wave tile 64 x 64, the same phases).  A conv's
// activation pieces take a 64-bit address per lane (x3_dma16v): input pixel of (output pixel, tap) or the zero page; a thread's two / four
// staging rows, their nine validity bits and base pointers are fixed for the whole launch.
template <bool DIAG, int WM, bool CONV>
__global__ void __launch_bounds__(512, 2) gemm_x3_kernel(GemmF32Args a) {
    constexpr int MT = 4;
    constexpr int WN = 8 / WM, XBM = WM * 64, XBN = WN * 64;     // waves along N; tile rows (M) and columns (N)
    static_assert(WM == 4 || WM == 2, "wave layouts 4 x 2 and 2 x 4");
    const bool drop_alo = DIAG && (a.diag & 1), drop_blo = DIAG && (a.diag & 2), hi_only = DIAG && (a.diag & 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (CONV) {
        if (a.groups > 1) {                  // grouped conv: this workgroup's group = blockIdx.y
            const int g = blockIdx.y;
            a.A += (size_t)g * a.taps * a.M * a.K;
            a.X += (size_t)g * a.K;
            a.C += (size_t)g * a.M;
            if (a.scale) a.scale += (size_t)g * a.M;
            if (a.shift) a.shift += (size_t)g * a.M;
            if (a.res) a.res += (size_t)g * a.M;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN, q = lane >> 4, r16 = lane & 15;
    // Workgroup -> tile: workgroups go round-robin over the 8 XCDs (id % 8), each with its own L2.  The M / XBM row blocks of one
    // column tile are consecutive on ONE XCD, so the second block's X rows (the same rows) are L2 hits instead of a second HBM read.
    const unsigned ny = (unsigned)(a.M / XBM), jx = blockIdx.x >> 3;
    const unsigned tile_x = (jx / ny) * 8u + (blockIdx.x & 7u);
    if ((long)tile_x * XBN >= a.N) return;                   // the grid is padded to 8 * ny * ceil(nx / 8)
    const long n0 = (long)tile_x * XBN;
    const int m0 = (int)(jx % ny) * XBM;
    const int pairs_per_tap = a.K / 32, npairs = a.taps * pairs_per_tap;
    // staging: six 64-row pieces of 8 KiB per pair (A rows 0 .. XBM-1, then X rows 0 .. XBN-1); this thread's row of a piece: wave * 8 +
    // lane / 8, its LDS slot lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7) of that row (4 values = one split-format chunk).
    // Source address = wave-uniform base (SGPR pair: operand + tap + k offset + piece) + one 32-bit lane offset per operand row.
    const int rloc = wv * 8 + (lane >> 3), chunk4 = ((lane & 7) ^ ((rloc >> 1) & 7)) * 4;
    const char* Ab = (const char*)(a.A + (size_t)m0 * a.K);                       // M is a multiple of XBM (launcher)
    const unsigned voffA = (unsigned)((rloc * a.K + chunk4) * 4);
    const size_t a_piece = (size_t)64 * a.K * 4, a_tap = (size_t)a.M * a.K * 4;
    auto rowoff = [&](long n) { const long xb = n / a.rows_per_batch; return xb * a.batch_stride + (n - xb * a.rows_per_batch) * a.row_stride; };
    // mode 0: rows addressed as base + lane offset
    const char* Xb = nullptr;
    unsigned voffX[WN];
    // CONV: per staging row the input pixel of tap (0, 0) in X (and in X2), and the taps that fall inside the image
    const char* xrow[WN];
    const char* xrow2[WN];
    unsigned vmask[WN];
    if constexpr (!CONV) {
        const long off0 = rowoff(n0);
        Xb = (const char*)(a.X + off0 - (long)(a.taps >> 1) * a.tap_stride);          // row n0 of tap 0
#pragma unroll
        for (int p = 0; p < WN; ++p) {       // columns past N read row N-1 (valid memory; their results are never stored)
            long n = n0 + p * 64 + rloc;
            if (n >= a.N) n = a.N - 1;
            voffX[p] = (unsigned)((rowoff(n) - off0 + chunk4) * 4);
        }
    } else {
        const int st = a.stride > 1 ? a.stride : 1;
        const int Ho = (a.H - 1) / st + 1, Wo = (a.W - 1) / st + 1, hw = Ho * Wo;
        const long ldx = a.ldx ? a.ldx : a.Cin;
#pragma unroll
        for (int p = 0; p < WN; ++p) {
            const long n = n0 + p * 64 + rloc;
            xrow[p] = xrow2[p] = (const char*)g_zero_page128;
            vmask[p] = 0u;
            if (n < a.N) {
                const long b = n / hw;
                const int pix = (int)(n - b * hw), y0 = (pix / Wo) * st, x0 = (pix % Wo) * st;
                const long ipix = (b * a.H + y0) * a.W + x0;
                xrow[p] = (const char*)(a.X + ipix * ldx + chunk4);
                if (a.X2) xrow2[p] = (const char*)(a.X2 + ipix * a.ldx2 + chunk4);
                if (a.taps == 9) {
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        if ((unsigned)(y0 + t / 3 - 1) < (unsigned)a.H && (unsigned)(x0 + t % 3 - 1) < (unsigned)a.W) vmask[p] |= 1u << t;
                } else {
                    vmask[p] = 1u;
                }
            }
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    bool x3_steady = false;            // false during the prologue (ablation builds that drop one operand's pieces still stage both there)
    (void)x3_steady;
    int st_tap = 0, st_kq = 0;               // staging cursor: the next pair to stage is (tap st_tap, k-pair st_kq)
    const char *st_a = Ab, *st_x = Xb;
    long st_xoff = 0;                        // CONV: byte offset of the cursor's (tap, k-pair) from a row's tap-(0,0) pixel
    bool st_second = false;                  // CONV, two-part input: the cursor's channels come from X2
    auto st_update = [&]() {
        st_a = Ab + (size_t)st_tap * a_tap + (size_t)st_kq * 128;
        if constexpr (!CONV) {
            st_x = Xb + ((long)st_tap * a.tap_stride + (long)st_kq * 32) * 4;
        } else {
            const int dy = a.taps == 9 ? st_tap / 3 - 1 : 0, dx = a.taps == 9 ? st_tap % 3 - 1 : 0, kk = st_kq * 32;
            st_second = a.X2 && kk >= a.ksplit;
            const long pitch = st_second ? a.ldx2 : (a.ldx ? a.ldx : a.Cin);
            st_xoff = ((long)(dy * a.W + dx) * pitch + (st_second ? kk - a.ksplit : kk)) * 4;
        }
    };
    st_update();
    auto st_advance = [&]() {
        if (++st_kq == pairs_per_tap) { st_kq = 0; ++st_tap; }
        st_update();
    };
    auto piece = [&](int k, unsigned slot_lds) {          // one of the six 8 KiB DMA pieces of the cursor's pair
        if (X3A_SKIP_PIECE(k, x3_steady)) return;
        if (k < WM) {
            x3_dma16(st_a + (size_t)k * a_piece, voffA, slot_lds + k * 8192 + wv * 1024);
        } else if constexpr (!CONV) {
            x3_dma16(st_x, voffX[k - WM], slot_lds + k * 8192 + wv * 1024);
        } else {
            const int p = k - WM;
            const char* src = ((vmask[p] >> st_tap) & 1u) ? (st_second ? xrow2[p] : xrow[p]) + st_xoff : (const char*)g_zero_page128 + chunk4 * 4;
            x3_dma16v(src, slot_lds + k * 8192 + wv * 1024);
        }
    };
    f32x4 acc[MT][4], cor[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; cor[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int sw = (r16 >> 1) & 7;
    const int f0 = r16 * 128 + ((q ^ sw) * 16), f1 = r16 * 128 + (((4 + q) ^ sw) * 16);
    const int aoff = wm * 8192, boff = WM * 8192 + wn * 8192;

    // Fragment registers: operand halves A0 / A1 (accumulator rows i = 0,1 / 2,3) and B0 / B1 (columns j = 0,1 / 2,3), each two
    // 16-row tiles x (hi, lo) = 16 registers.  A tile's two chunks c0 = [hi0-3 | lo0-3], c1 = [hi4-7 | lo4-7] are read straight
    // into (H, L) and turned into H = 8 hi, L = 8 lo by exchanging two registers (`fix`), one phase after the read.
    u32x4_t AH[2][2], AL[2][2], BH[2][2], BL[2][2];
    auto ld = [&](u32x4_t& H, u32x4_t& L, const char* tile) { X3A_LD(H, L, tile, f0, f1); };
    auto fix = [&](u32x4_t& H, u32x4_t& L) { X3A_FIX(H, L); };
    auto hv = [](const u32x4_t& v) { return __builtin_bit_cast(f16x8, v); };
#define X3_MFMA(k, ah, bh)                                                                                                        \
    do {                                                                                                                           \
        constexpr int t_ = (k) & 1, u_ = ((k) >> 1) & 1, i_ = 2 * (ah) + t_, j_ = 2 * (bh) + u_;                                   \
        if ((k) < 4) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv(AH[ah][t_]), hv(BH[bh][u_]), acc[i_][j_], 0, 0, 0);   \
        else if ((k) < 8) { if (!drop_blo) cor[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv(AH[ah][t_]), hv(BL[bh][u_]), cor[i_][j_], 0, 0, 0); } \
        else { if (!drop_alo) cor[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv(AL[ah][t_]), hv(BH[bh][u_]), cor[i_][j_], 0, 0, 0); }             \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    } while (0)
    // One phase = the 12 MFMAs of quadrant (A half ah) x (B half bh): 4 x hi*hi, 4 x hi*lo, 4 x lo*hi (the two products that share a
    // `cor` accumulator are four issues apart).  Before them the half read during the previous phase is fixed (FIXH, FIXT: which
    // operand / half); between them the four fragment reads of the half that is free (LDH: 0 = A, 1 = B; LDT: half; from `lbase`)
    // and, in the second half of an iteration, three of the six DMA pieces of the pair three ahead.
#define X3_PHASE(ah, bh, FIXB, FIXT, LDB, LDT, lbase, DMA0, dma_slot)                                                             \
    do {                                                                                                                           \
        if (FIXB) { fix(BH[FIXT][0], BL[FIXT][0]); fix(BH[FIXT][1], BL[FIXT][1]); }                                               \
        else { fix(AH[FIXT][0], AL[FIXT][0]); fix(AH[FIXT][1], AL[FIXT][1]); }                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (LDB) ld(BH[LDT][0], BL[LDT][0], (lbase) + boff + (LDT) * 4096); else ld(AH[LDT][0], AL[LDT][0], (lbase) + aoff + (LDT) * 4096); \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        X3_MFMA(0, ah, bh); X3_MFMA(1, ah, bh);                                                                                    \
        if (LDB) ld(BH[LDT][1], BL[LDT][1], (lbase) + boff + (LDT) * 4096 + 2048); else ld(AH[LDT][1], AL[LDT][1], (lbase) + aoff + (LDT) * 4096 + 2048); \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        X3_MFMA(2, ah, bh); X3_MFMA(3, ah, bh);                                                                                    \
        if ((DMA0) >= 0 && do_dma) { piece((DMA0), (dma_slot)); __builtin_amdgcn_sched_barrier(0); }                               \
        X3_MFMA(4, ah, bh); X3_MFMA(5, ah, bh); X3_MFMA(6, ah, bh);                                                                \
        if ((DMA0) >= 0 && do_dma) { piece((DMA0) + 1, (dma_slot)); __builtin_amdgcn_sched_barrier(0); }                           \
        X3_MFMA(7, ah, bh); X3_MFMA(8, ah, bh); X3_MFMA(9, ah, bh);                                                                \
        if ((DMA0) >= 0 && do_dma) { piece((DMA0) + 2, (dma_slot)); __builtin_amdgcn_sched_barrier(0); }                           \
        X3_MFMA(10, ah, bh); X3_MFMA(11, ah, bh);                                                                                  \
    } while (0)

    // prologue: pairs 0-2 in flight, pair 0 landed, A0 / B0 of pair 0 in registers
#pragma unroll
    for (int s = 0; s < 3; ++s)
        if (s < npairs) {
#pragma unroll
            for (int k = 0; k < 6; ++k) piece(k, lds0 + s * X3_PAIR);
            st_advance();
        }
    if (npairs >= 3) { GF_WAIT_BARRIER(12); } else if (npairs == 2) { GF_WAIT_BARRIER(6); } else { GF_WAIT_BARRIER(0); }
    ld(AH[0][0], AL[0][0], smem + aoff); ld(AH[0][1], AL[0][1], smem + aoff + 2048);
    ld(BH[0][0], BL[0][0], smem + boff); ld(BH[0][1], BL[0][1], smem + boff + 2048);
    fix(AH[0][0], AL[0][0]); fix(AH[0][1], AL[0][1]);
    __builtin_amdgcn_sched_barrier(0);

    // Pair p computes in four phases; its fragments are read from ring slot p % 3 during phases 3, 4 of pair p-1 and phases 1, 2 of
    // pair p.  The ONE barrier of a pair sits between phases 2 and 3: behind it pair p+1 has landed (counted vmcnt: only the six
    // pieces of pair p+2 may still fly), every wave has its last fragments of slot p % 3 in registers, and the six pieces of pair
    // p+3 go into that slot under the MFMAs of phases 3, 4.  Even and odd pairs walk the quadrants in mirrored order, so that
    // every phase replaces exactly one operand half — the one no later phase of the pair reads:
    //   even: (A0,B0) (A0,B1) | (A1,B1) (A1,B0)      odd: (A0,B1) (A0,B0) | (A1,B0) (A1,B1)
    X3A_STAMP_DECL;
    x3_steady = true;
    int slot = 0;
    for (int p = 0; p < npairs; p += 2) {
        const int s1 = slot == 2 ? 0 : slot + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        const char* L0 = smem + slot * X3_PAIR;
        const char* L1 = smem + s1 * X3_PAIR;
        const char* L2 = smem + s2 * X3_PAIR;
        bool do_dma = false;
        X3A_STAMP(0);
        // ---- even pair p (slot `slot`)
        X3_PHASE(0, 0, 1, 0, 1, 1, L0, -1, 0u);                  // fix B0;  (A0,B0);  read B1(p)
        X3_PHASE(0, 1, 1, 1, 0, 1, L0, -1, 0u);                  // fix B1;  (A0,B1);  read A1(p)
        X3A_STAMP(1);
        X3A_PAIR_BARRIER(p + 2 < npairs);
        X3A_STAMP(2);
        do_dma = X3A_DO_DMA(p + 3 < npairs);
        X3_PHASE(1, 1, 0, 1, 0, 0, L1, 0, lds0 + slot * X3_PAIR);    // fix A1;  (A1,B1);  read A0(p+1);  pieces 0-2 of pair p+3
        X3_PHASE(1, 0, 0, 0, 1, 1, L1, 3, lds0 + slot * X3_PAIR);    // fix A0;  (A1,B0);  read B1(p+1);  pieces 3-5
        if (do_dma) st_advance();
        X3A_STAMP(3);
        if (p + 1 >= npairs) break;
        // ---- odd pair p+1 (slot s1)
        do_dma = false;
        X3_PHASE(0, 1, 1, 1, 1, 0, L1, -1, 0u);                  // fix B1;  (A0,B1);  read B0(p+1)
        X3_PHASE(0, 0, 1, 0, 0, 1, L1, -1, 0u);                  // fix B0;  (A0,B0);  read A1(p+1)
        X3A_STAMP(1);
        X3A_PAIR_BARRIER(p + 3 < npairs);
        X3A_STAMP(2);
        do_dma = X3A_DO_DMA(p + 4 < npairs);
        X3_PHASE(1, 0, 0, 1, 0, 0, L2, 0, lds0 + s1 * X3_PAIR);      // fix A1;  (A1,B0);  read A0(p+2);  pieces 0-2 of pair p+4
        X3_PHASE(1, 1, 0, 0, 1, 0, L2, 3, lds0 + s1 * X3_PAIR);      // fix A0;  (A1,B1);  read B0(p+2);  pieces 3-5
        if (do_dma) st_advance();
        X3A_STAMP(3);
        slot = s2;
    }
    X3A_STAMP_PRINT(npairs, wv, lane);
#undef X3_PHASE
#undef X3_MFMA
    __builtin_amdgcn_s_waitcnt(0xC07F);          // the reads past the last pair (never used) are retired before the epilogue
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(cor[i][j][r], kSplitInv, acc[i][j][r]);

    if (a.epi == 1) {                        // gate -> split-f16 g (operand of the res / skip GEMM)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mrow = m0 + wm * 64 + i * 16 + q * 4;
            const int ch = ((m0 + wm * 64) >> 1) + i * 16 + q * 4;       // 64-row wave slab = 32 gate channels (tanh rows | sigmoid rows)
            const float4 bt = *(const float4*)(a.shift + mrow), bs = *(const float4*)(a.shift + mrow + 32);
            const float bta[4] = {bt.x, bt.y, bt.z, bt.w}, bsa[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // tanh(ht) * sigmoid(hs) = (1 - u) / ((1 + u)(1 + v)), u = e^-2ht, v = e^-hs: two v_exp + one v_rcp (~1 ulp each)
                    // instead of the libm tanhf / expf / division of the exact path, whose ~150 instructions per value would cost
                    // this tier as much as its matrix work; the difference (~3e-7 relative) is far inside the tier's bound
                    const float ht = acc[i][j][r] + bta[r], hs = acc[i + 2][j][r] + bsa[r];
                    const float u = fast_exp2(fminf(ht * -2.8853900817779268f, 30.f)), w = fast_exp2(hs * -1.4426950408889634f);
                    const float pp = 1.f + u, rr = fast_rcp(pp * w + pp);
                    v[r] = rr - u * rr;
                }
                u32x4_t gs = split4(v[0], v[1], v[2], v[3]);
                if (hi_only) { gs[2] = 0u; gs[3] = 0u; }
                *(u32x4_t*)(a.C + n * 256 + ch) = gs;
            }
        }
        return;
    }
    if (a.epi == 2) {                        // h' = (h + res) * sqrt(1/2) + emb_next in the split format (operand of the next layer)
        // row of position n inside the zero-padded residual stream, once per accumulator column (not per tile: a 64-bit division
        // per (i, j) cost the K = 256 res launches a fifth of their time); N < 2^31 positions (launcher)
        long hrow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned n32 = (unsigned)(n0 + wn * 64 + j * 16 + r16);
            if ((long)n32 >= a.N) n32 = (unsigned)(a.N - 1);                 // (never stored)
            const unsigned bb = n32 / (unsigned)a.L;
            hrow[j] = ((long)bb * a.LP + kPad + (n32 - bb * (unsigned)a.L)) * kC;
        }
        // all sixteen residual chunks first, then the stores (see the fp32 kernel: hin / hout may alias as far as the compiler knows)
        u32x4_t hv[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[i][j] = *(const u32x4_t*)(a.hin + hrow[j] + m0 + wm * 64 + i * 16 + q * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 64 + i * 16 + q * 4;
            const float4 b4 = *(const float4*)(a.shift + m), e4 = *(const float4*)(a.emb_next + m);
            const float ba[4] = {b4.x, b4.y, b4.z, b4.w}, ea[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float h[4];
                join4(hv[i][j], h);
                const float k = 0.70710678118654752440f;
                u32x4_t hs = split4(__fadd_rn(__fmul_rn(__fadd_rn(h[0], acc[i][j][0] + ba[0]), k), ea[0]), __fadd_rn(__fmul_rn(__fadd_rn(h[1], acc[i][j][1] + ba[1]), k), ea[1]),
                                    __fadd_rn(__fmul_rn(__fadd_rn(h[2], acc[i][j][2] + ba[2]), k), ea[2]), __fadd_rn(__fmul_rn(__fadd_rn(h[3], acc[i][j][3] + ba[3]), k), ea[3]));
                if (hi_only) { hs[2] = 0u; hs[3] = 0u; }
                *(u32x4_t*)(a.hout + hrow[j] + m) = hs;
            }
        }
        return;
    }
    if constexpr (!CONV) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {           // plain: fp32 out, bias, optional ReLU (final_conv.0)
            const int m = m0 + wm * 64 + i * 16 + q * 4;
            const float4 b4 = a.shift ? *(const float4*)(a.shift + m) : float4{0.f, 0.f, 0.f, 0.f};
            const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float t = acc[i][j][r] + ba[r]; v[r] = a.relu ? relu_nan(t) : t; }
                *(float4*)(a.C + n * a.ldc + m) = float4{v[0], v[1], v[2], v[3]};
            }
        }
        return;
    }
    // plain: bias (or folded BatchNorm scale / shift), optional fp32 residual, optional ReLU; fp32 out, or the split format when the
    // consumer is another GEMM of this tier (GemmF32Args::out_split).  A row tile's residual chunks are loaded ahead of its stores (the
    // compiler cannot know that `res` and `C` do not overlap).
    long nrow[4];
    bool nok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long n = n0 + wn * 64 + j * 16 + r16;
        nok[j] = n < a.N;
        nrow[j] = (nok[j] ? n : a.N - 1) * a.ldc;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * 64 + i * 16 + q * 4;
        const float4 b4 = a.shift ? *(const float4*)(a.shift + m) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 s4 = a.scale ? *(const float4*)(a.scale + m) : float4{1.f, 1.f, 1.f, 1.f};
        const float ba[4] = {b4.x, b4.y, b4.z, b4.w}, sa[4] = {s4.x, s4.y, s4.z, s4.w};
        float4 rr[4];
        if (a.res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[j] = *(const float4*)(a.res + nrow[j] + m);
            if (a.res_split) {               // the residual map is in the split format: back to fp32 values
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t4[4];
                    join4(__builtin_bit_cast(u32x4_t, rr[j]), t4);
                    rr[j] = float4{t4[0], t4[1], t4[2], t4[3]};
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!nok[j]) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a.scale ? acc[i][j][r] * sa[r] + ba[r] : acc[i][j][r] + ba[r];
            if (a.res) { v[0] += rr[j].x; v[1] += rr[j].y; v[2] += rr[j].z; v[3] += rr[j].w; }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = relu_nan(v[r]);
            }
            if (a.out_split) *(u32x4_t*)(a.C + nrow[j] + m) = split4(v[0], v[1], v[2], v[3]);
            else *(float4*)(a.C + nrow[j] + m) = float4{v[0], v[1], v[2], v[3]};
        }
    }
}

// C[n][m] = act(scale[m] * sum_z slab[z][n][m] + shift[m]), splits summed in index order (deterministic)
__global__ void gemm_f32_reduce_kernel(GemmF32Args a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N * a.M) return;
    const long n = i / a.M;
    const int m = (int)(i - n * a.M);
    float s = 0.f;
    for (int z = 0; z < a.splits; ++z) s += a.slab[((long)z * a.N + n) * a.ldc + m];
    float t = a.scale ? s * a.scale[m] + (a.shift ? a.shift[m] : 0.f) : s + (a.shift ? a.shift[m] : 0.f);
    if (a.res) t += a.res[n * a.ldc + m];
    a.C[n * a.ldc + m] = a.relu ? relu_nan(t) : t;
}

namespace { thread_local int g_bad_shapes = 0; }
// launches refused since the last call (and reset): the C ABI turns a non-zero count into DMAD_ERR_INVALID at the end of the entry point
int gemm_take_bad_shapes() { const int n = g_bad_shapes; g_bad_shapes = 0; return n; }

// once per device a process uses (dmad_create): the x3 tier's 144 KiB of dynamic LDS
int gemm_x3_configure() {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_x3_kernel<false, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_x3_kernel<true, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_x3_kernel<false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute((const void*)gemm_x3_kernel<false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
}

// once per device a process uses (dmad_create, like gemm_x3_configure: the attribute is per device, so a function-local static set on the
// first device would leave every later device without it): 128 KiB of dynamic LDS for the 8-slot narrow-tile instantiation
int gemm_f32_configure() {
    return (int)hipFuncSetAttribute((const void*)gemm_f32_kernel<64, false, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * SLOT);
}

int launch_gemm_f32(const GemmF32Args& a0, hipStream_t s, float* slab, long slab_floats, long n_ref) {
    GemmF32Args a = a0;
    // the update epilogue indexes positions in 32 bits and serves M = 256 residual rows (the skip convs are one GEMM of their own)
    if (a.epi == 2 && (a.N >= (1l << 31) || a.L < 1 || a.M != 256 || a.res_rows != a.M || !a.hin || !a.hout || !a.emb_next)) { ++g_bad_shapes; return kGemmBadShape; }
    if (a.x3) {                                   // split-f16 operands (shapes checked here, not in the kernel)
        a.splits = 1; a.slab = nullptr;
        if ((a.K % 32) || (a.ldc & 3) || a.K < 32) { ++g_bad_shapes; return kGemmBadShape; }
        if (a.mode == 0) {                        // the WaveNet's row-gather GEMMs: tile 256 x 128
            if ((a.M % X3_BM) || a.scale || a.res || a.out_split || a.X2 || a.groups > 1) { ++g_bad_shapes; return kGemmBadShape; }
            const long nx = (a.N + BN - 1) / BN;
            if (((nx + 7) / 8) * 8 * (a.M / X3_BM) > 0x7fffffffl) { ++g_bad_shapes; return kGemmBadShape; }
            const dim3 grid((unsigned)(((nx + 7) / 8) * 8 * (a.M / X3_BM)));
            if (a.diag) hipLaunchKernelGGL((gemm_x3_kernel<true, 4, false>), grid, dim3(512), X3_LDS, s, a);
            else hipLaunchKernelGGL((gemm_x3_kernel<false, 4, false>), grid, dim3(512), X3_LDS, s, a);
            return 0;
        }
        // NHWC convs (3x3 zero padding 1 / 1x1, stride 1 or 2, optional two-part input): rows of 16-byte chunks, no fused epilogue
        const long ldx = a.ldx ? a.ldx : a.Cin;
        if (a.mode != 2 || a.epi || a.diag || (a.M % 128) || (a.taps != 9 && a.taps != 1) || (ldx & 3) || a.H < 1 || a.W < 1 ||
            (a.X2 && ((a.ksplit % 32) || a.ksplit <= 0 || a.ksplit >= a.K || (a.ldx2 & 3) || a.groups > 1)) || (a.res && a.res == a.C) ||
            (a.groups > 1 && ((long)a.groups * a.K > ldx || (long)a.groups * a.M > a.ldc || a.groups > 65535))) { ++g_bad_shapes; return kGemmBadShape; }
        const bool big = a.M % 256 == 0;
        const long nx = (a.N + (big ? 127 : 255)) / (big ? 128 : 256), ny = a.M / (big ? 256 : 128);
        if (((nx + 7) / 8) * 8 * ny > 0x7fffffffl) { ++g_bad_shapes; return kGemmBadShape; }
        const dim3 grid((unsigned)(((nx + 7) / 8) * 8 * ny), (unsigned)(a.groups > 1 ? a.groups : 1));
        if (big) hipLaunchKernelGGL((gemm_x3_kernel<false, 4, true>), grid, dim3(512), X3_LDS, s, a);
        else hipLaunchKernelGGL((gemm_x3_kernel<false, 2, true>), grid, dim3(512), X3_LDS, s, a);
        return 0;
    }
    if (a.X2 && (a.mode != 2 || a.groups > 1 || a.M <= 64 || (a.ksplit % BK) || a.ksplit <= 0 || a.ksplit >= a.K)) { ++g_bad_shapes; return kGemmBadShape; }   // two-part input: plain NHWC convs only
    // 64-row tiles where a 128-row tile would be half empty — and for launches that leave most of the chip idle with 128-row tiles
    // (a handful of samples on the fp32 tiers: the recheck of a vote loop, an audit): twice the workgroups for the same work.  Every
    // output is the same K-ordered sum in both tile shapes (same bits); the split count below does not depend on the choice.
    const long gx_ = (a.N + BN - 1) / BN, gy128 = (a.M + 127) / 128;
    const bool small_grid = a.epi == 0 && !a.X2 && a.M > 64 && gx_ * gy128 * (a.groups > 1 ? a.groups : 1) < 128;
    const int BM = (a.M <= 64 || small_grid) ? 64 : 128;
    const unsigned gx = (unsigned)gx_, gy = (unsigned)((a.M + BM - 1) / BM);
    const int nks = a.taps * (a.K / BK);
    int S = 1;
    static const bool narrow_on = []() { const char* v = getenv("DMAD_F32_NARROW"); return !(v && v[0] == '0'); }();      // A/B switch
    auto launch = [&](dim3 grid) {
        if (a.X2) hipLaunchKernelGGL((gemm_f32_kernel<128, true>), grid, dim3(256), 3 * SLOT, s, a);
        else if (BM == 64 && narrow_on && (long)grid.x * grid.y * grid.z < 256 && nks >= 8)       // fewer workgroups than CUs:
            hipLaunchKernelGGL((gemm_f32_kernel<64, false, 8, 1>), dim3((unsigned)((a.N + 31) / 32), grid.y, grid.z), dim3(256), 8 * SLOT, s, a);   // 64 x 32 tiles, 8-slot ring
        else if (BM == 64) hipLaunchKernelGGL((gemm_f32_kernel<64, false>), grid, dim3(256), 3 * SLOT, s, a);
        else hipLaunchKernelGGL((gemm_f32_kernel<128, false>), grid, dim3(256), 3 * SLOT, s, a);
    };
    if (a.groups > 1) {
        a.splits = 1;
        a.slab = nullptr;
        launch(dim3(gx, gy, (unsigned)a.groups));
        return 0;
    }
    if (slab) {
        // The split count is derived from the REFERENCE row count n_ref (the engine's max batch), not from the
        // rows of this launch, so that a sample's result does not depend on the batch it was computed in.
        const long nr = n_ref > 0 ? n_ref : a.N;
        const long wgs_ref = ((nr + BN - 1) / BN) * (a.M <= 64 ? 1 : gy128);       // (in 128-row tiles, whatever this launch uses)
        if (wgs_ref < 384) {                       // fewer than half the resident workgroups (3 per CU): split K
            S = (int)(768 / wgs_ref);
            if (S > nks / 4) S = nks / 4;          // keep >= 4 k-steps per split
            while (S > 1 && (long)S * nr * a.ldc > slab_floats) --S;
            if (S < 2 || a.N > nr) S = 1;
        }
    }
    a.splits = S;
    a.slab = slab;
    launch(dim3(gx, gy, S));
    if (S > 1) {
        const long total = a.N * a.M;
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    }
    return 0;
}

}  // namespace dmad
