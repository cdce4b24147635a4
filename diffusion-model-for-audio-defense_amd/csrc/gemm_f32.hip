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

namespace dmad {

namespace {
constexpr int BN = 128, BK = 16;          // BM = 128, or 64 for layers / conv groups with at most 64 output channels
constexpr int SLOT = 16384;                       // 128 A rows + 128 X rows of 64 B
__device__ __attribute__((aligned(64))) float g_zero_page[16];                 // 64 B of zeros: source of every out-of-problem row
__device__ __attribute__((aligned(128))) float g_zero_page_x3[32];             // 128 B of zeros: the split-f16 kernel stages 128-byte rows

#define GF_WAIT_BARRIER(N)                                                          \
    do {                                                                            \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                       \
        __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14));       \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    } while (0)
}  // namespace

template <int BM, bool TWO>               // TWO: the input channels come from two maps (GemmF32Args::X2); an instantiation of its own
                                          // because the extra test in the staging loop costs the common kernel 2 %
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32Args a) {
    constexpr int MT = BM / 32;            // 16-row accumulator tiles per wave along M (2 M-waves)
    __shared__ __attribute__((aligned(16))) char smem[3 * SLOT];
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
    const long n0 = (long)blockIdx.x * BN;
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
        xok[p] = n < a.N;
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
        for (int p = 0; p < 2; ++p) {
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
    if (ks_begin < nks) stage(ks_begin, 0);
    if (ks_begin + 1 < nks) stage(ks_begin + 1, 1);
    int slot = 0;
    for (int ks = ks_begin; ks < nks; ++ks) {
        // stage ks landed (the 4 pieces of stage ks+1 may still fly); every wave is done reading slot (ks-1) % 3
        if (ks + 1 < nks) { if (BM == 128) { GF_WAIT_BARRIER(4); } else { GF_WAIT_BARRIER(3); } } else { GF_WAIT_BARRIER(0); }
        if (ks + 2 < nks) stage(ks + 2, slot >= 1 ? slot - 1 : 2);
        const char* As = smem + slot * SLOT + wm * (BM * 32) + frag;
        const char* Bs = smem + slot * SLOT + 8192 + wn * 4096 + frag;
        f32x4 af[MT], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) af[i] = *(const f32x4*)(As + i * 1024);
            bf[i] = *(const f32x4*)(Bs + i * 1024);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        slot = slot == 2 ? 0 : slot + 1;
    }

    if (S > 1) {   // split-K: raw partial sums to this split's slab; scale/shift/act happen in the reduce kernel
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * (BM / 2) + i * 16 + q * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
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
    if (BM == 128 && a.epi == 2) {          // h' = (h + res) * sqrt(1/2) + emb_next ; skip (+)= skip conv
        long hrow[4];                       // stream row of position n, once per accumulator column (N < 2^31 positions)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned n32 = (unsigned)(n0 + wn * 64 + j * 16 + r16), bb = n32 / (unsigned)a.L;
            hrow[j] = ((long)bb * a.LP + kPad + (n32 - bb * (unsigned)a.L)) * kC;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 64 + i * 16 + q * 4;
            const float4 b4 = *(const float4*)(a.shift + m);
            const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
            const bool is_res = m < a.res_rows;
            float ea[4] = {0.f, 0.f, 0.f, 0.f};
            if (is_res) { const float4 e4 = *(const float4*)(a.emb_next + m); ea[0] = e4.x; ea[1] = e4.y; ea[2] = e4.z; ea[3] = e4.w; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + ba[r];
                if (is_res) {
                    const long hoff = hrow[j] + m;
                    const float4 h = *(const float4*)(a.hin + hoff);
                    const float k = 0.70710678118654752440f;
                    *(float4*)(a.hout + hoff) = float4{__fadd_rn(__fmul_rn(__fadd_rn(h.x, v[0]), k), ea[0]), __fadd_rn(__fmul_rn(__fadd_rn(h.y, v[1]), k), ea[1]),
                                                       __fadd_rn(__fmul_rn(__fadd_rn(h.z, v[2]), k), ea[2]), __fadd_rn(__fmul_rn(__fadd_rn(h.w, v[3]), k), ea[3])};
                } else {
                    float4* ps = (float4*)(a.skip + n * 256 + (m - a.res_rows));
                    if (a.first) {
                        *ps = float4{v[0], v[1], v[2], v[3]};
                    } else {
                        const float4 o = *ps;
                        *ps = float4{__fadd_rn(o.x, v[0]), __fadd_rn(o.y, v[1]), __fadd_rn(o.z, v[2]), __fadd_rn(o.w, v[3])};
                    }
                }
            }
        }
        return;
    }
    const bool vec = ((a.ldc & 3) == 0);
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
        for (int j = 0; j < 4; ++j) {
            const long n = n0 + wn * 64 + j * 16 + r16;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = a.scale ? acc[i][j][r] * sc[r] + sh[r] : acc[i][j][r] + sh[r];
                if (a.res && m + r < a.M) t += a.res[n * a.ldc + m + r];
                v[r] = a.relu ? relu_nan(t) : t;
            }
            float* dst = a.C + n * a.ldc + m;
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
template <bool DIAG>
__global__ void __launch_bounds__(512, 2) gemm_x3_kernel(GemmF32Args a) {
    constexpr int BM = X3_BM, MT = 4;
    const bool drop_alo = DIAG && (a.diag & 1), drop_blo = DIAG && (a.diag & 2), hi_only = DIAG && (a.diag & 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1, q = lane >> 4, r16 = lane & 15;
    const long n0 = (long)blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const int pairs_per_tap = a.K / 32, npairs = a.taps * pairs_per_tap;
    // staging: six 64-row pieces of 8 KiB per pair (A rows 0-255, X rows 0-127); this thread's row of a piece: wave * 8 + lane / 8,
    // its LDS slot lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7) of that row (4 values = one split-format chunk)
    const int rloc = wv * 8 + (lane >> 3), chunk4 = ((lane & 7) ^ ((rloc >> 1) & 7)) * 4;
    const float* zero = g_zero_page_x3 + (lane & 7) * 4;
    const float* arow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) arow[p] = a.A + (size_t)(m0 + p * 64 + rloc) * a.K + chunk4;       // M is a multiple of 256 (launcher)
    const float* xrow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const long n = n0 + p * 64 + rloc;
        xrow[p] = nullptr;
        if (n < a.N) {
            const long xb = n / a.rows_per_batch;
            xrow[p] = a.X + xb * a.batch_stride + (n - xb * a.rows_per_batch) * a.row_stride + chunk4;
        }
    }
    auto stage = [&](int pr, char* base) {          // base: this pair's 48 KiB (A rows 0-255: 32 KiB, X rows 0-127: 16 KiB)
        const int tap = pr / pairs_per_tap, kc = (pr - tap * pairs_per_tap) * 32;
        char* la = base + wv * 1024;
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16(arow[p] + (size_t)tap * a.M * a.K + kc, la + p * 8192);
#pragma unroll
        for (int p = 0; p < 2; ++p) glds16(xrow[p] ? xrow[p] + (long)(tap - (a.taps >> 1)) * a.tap_stride + kc : zero, la + 32768 + p * 8192);
    };
    f32x4 acc[MT][4], cor[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; cor[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int sw = (r16 >> 1) & 7;
    const int f0 = r16 * 128 + ((q ^ sw) * 16), f1 = r16 * 128 + (((4 + q) ^ sw) * 16);
    stage(0, smem);
    if (npairs > 1) stage(1, smem + X3_PAIR);
    int slot = 0;
    for (int p = 0; p < npairs; ++p) {
        // pair p landed (the 6 pieces of pair p+1 may still fly); every wave is done reading pair p-1
        if (p + 1 < npairs) { GF_WAIT_BARRIER(6); } else { GF_WAIT_BARRIER(0); }
        if (p + 2 < npairs) stage(p + 2, smem + (slot == 0 ? 2 : slot - 1) * X3_PAIR);      // into the slot pair p-1 occupied
        const char* A0 = smem + slot * X3_PAIR + wm * 8192;
        const char* B0 = smem + slot * X3_PAIR + 32768 + wn * 8192;
        f16x8 ahi[MT], alo[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const u32x4_t c0 = *(const u32x4_t*)(A0 + i * 2048 + f0), c1 = *(const u32x4_t*)(A0 + i * 2048 + f1);
            ahi[i] = __builtin_bit_cast(f16x8, u32x4_t{c0[0], c0[1], c1[0], c1[1]});
            alo[i] = __builtin_bit_cast(f16x8, u32x4_t{c0[2], c0[3], c1[2], c1[3]});
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4_t c0 = *(const u32x4_t*)(B0 + j * 2048 + f0), c1 = *(const u32x4_t*)(B0 + j * 2048 + f1);
            const f16x8 bhi = __builtin_bit_cast(f16x8, u32x4_t{c0[0], c0[1], c1[0], c1[1]});
            const f16x8 blo = __builtin_bit_cast(f16x8, u32x4_t{c0[2], c0[3], c1[2], c1[3]});
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi, acc[i][j], 0, 0, 0);
                if (!drop_blo) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo, cor[i][j], 0, 0, 0);
                if (!drop_alo) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi, cor[i][j], 0, 0, 0);
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
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
    if (a.epi == 2) {                        // h' (split-f16, operand of the next layer) ; skip sum (fp32)
        // row of position n inside the zero-padded residual stream, once per accumulator column (not per tile: a 64-bit division
        // per (i, j) cost the K = 256 res launches a fifth of their time); N < 2^31 positions (launcher)
        long hrow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned n32 = (unsigned)(n0 + wn * 64 + j * 16 + r16), bb = n32 / (unsigned)a.L;
            hrow[j] = ((long)bb * a.LP + kPad + (n32 - bb * (unsigned)a.L)) * kC;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 64 + i * 16 + q * 4;
            const float4 b4 = *(const float4*)(a.shift + m);
            const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
            const bool is_res = m < a.res_rows;
            float ea[4] = {0.f, 0.f, 0.f, 0.f};
            if (is_res) { const float4 e4 = *(const float4*)(a.emb_next + m); ea[0] = e4.x; ea[1] = e4.y; ea[2] = e4.z; ea[3] = e4.w; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + ba[r];
                if (is_res) {
                    const long hoff = hrow[j] + m;
                    float h[4];
                    join4(*(const u32x4_t*)(a.hin + hoff), h);
                    const float k = 0.70710678118654752440f;
                    u32x4_t hs = split4(__fadd_rn(__fmul_rn(__fadd_rn(h[0], v[0]), k), ea[0]), __fadd_rn(__fmul_rn(__fadd_rn(h[1], v[1]), k), ea[1]),
                                        __fadd_rn(__fmul_rn(__fadd_rn(h[2], v[2]), k), ea[2]), __fadd_rn(__fmul_rn(__fadd_rn(h[3], v[3]), k), ea[3]));
                    if (hi_only) { hs[2] = 0u; hs[3] = 0u; }
                    *(u32x4_t*)(a.hout + hoff) = hs;
                } else {
                    float4* ps = (float4*)(a.skip + n * 256 + (m - a.res_rows));
                    if (a.first) {
                        *ps = float4{v[0], v[1], v[2], v[3]};
                    } else {
                        const float4 o = *ps;
                        *ps = float4{__fadd_rn(o.x, v[0]), __fadd_rn(o.y, v[1]), __fadd_rn(o.z, v[2]), __fadd_rn(o.w, v[3])};
                    }
                }
            }
        }
        return;
    }
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
    hipError_t e = hipFuncSetAttribute((const void*)gemm_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute((const void*)gemm_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
}

int launch_gemm_f32(const GemmF32Args& a0, hipStream_t s, float* slab, long slab_floats, long n_ref) {
    GemmF32Args a = a0;
    if (a.epi == 2 && (a.N >= (1l << 31) || a.L < 1)) { ++g_bad_shapes; return kGemmBadShape; }       // the update epilogue indexes positions in 32 bits
    if (a.x3) {                                   // split-f16 operands: WaveNet shapes only (checked here, not in the kernel)
        if (a.mode != 0 || (a.M % X3_BM) || (a.K % 32) || a.scale || a.res || a.groups > 1 || (a.ldc & 3)) { ++g_bad_shapes; return kGemmBadShape; }
        a.splits = 1; a.slab = nullptr;
        const dim3 grid((unsigned)((a.N + BN - 1) / BN), (unsigned)(a.M / X3_BM));
        if (a.diag) hipLaunchKernelGGL(gemm_x3_kernel<true>, grid, dim3(512), X3_LDS, s, a);
        else hipLaunchKernelGGL(gemm_x3_kernel<false>, grid, dim3(512), X3_LDS, s, a);
        return 0;
    }
    if (a.X2 && (a.mode != 2 || a.groups > 1 || a.M <= 64 || (a.ksplit % BK) || a.ksplit <= 0 || a.ksplit >= a.K)) { ++g_bad_shapes; return kGemmBadShape; }   // two-part input: plain NHWC convs only
    const int BM = a.M <= 64 ? 64 : 128;          // 64-row tiles where a 128-row tile would be half empty
    const unsigned gx = (unsigned)((a.N + BN - 1) / BN), gy = (unsigned)((a.M + BM - 1) / BM);
    const int nks = a.taps * (a.K / BK);
    int S = 1;
    auto launch = [&](dim3 grid) {
        if (a.X2) hipLaunchKernelGGL((gemm_f32_kernel<128, true>), grid, dim3(256), 0, s, a);
        else if (BM == 64) hipLaunchKernelGGL((gemm_f32_kernel<64, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_f32_kernel<128, false>), grid, dim3(256), 0, s, a);
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
        const long wgs_ref = ((nr + BN - 1) / BN) * gy;
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
