// f16 implicit-GEMM for NHWC convolutions on v_mfma_f32_16x16x32_f16 (fp32 accumulate): the 16-bit tier of the Improved-Diffusion
// UNet purifier (reference improved_diffusion/unet.py:278-491; every conv / 1x1 of ResBlock :107-197, AttentionBlock :200-252,
// Downsample / Upsample :49-111).
//
//   C[n][m] = sum_tap sum_k A[tap][m][k] * X[pixel(n, tap)][k] + shift[m] (+ res[n][m])
//
// Operand scheme: both operands are row gathers staged by LDS-DMA, every lane fetching the 16 bytes that belong at its own LDS
// position.  A k-step is 64 halves = ONE 128-byte row per operand row: eight consecutive lanes fetch one whole cache line (the
// fp32 gather-GEMM's 64-byte rows would ask L2 for every line twice at this kernel's 16x smaller matrix time per staged byte).
// LDS image: 128-byte rows, the eight 16-byte chunks XOR-swizzled by (row >> 1) & 7, so that the 16 rows x one chunk of an MFMA
// fragment read (ds_read_b128) cover all 64 banks exactly once.  Rows outside the problem (zero padding, N tail) come from a
// zero page.  One f16 MFMA per staged fragment pair is 16x less matrix time per staged byte than the fp32 kernel's, so the tile is
// large and the pipeline deep: 384 staged rows per k-step split as BM x BN = 256 x 128 or 128 x 256 (layers whose output
// channels are a multiple of 128 only), 8 waves of 64 x 64, 3-slot LDS ring of 48 KiB (144 KiB, one workgroup per CU, two waves
// per SIMD), two k-steps in flight while one is contracted, one barrier per k-step.
#include "gemm_h16.h"
#include <stdlib.h>
#include <stdio.h>
#include <vector>

namespace dmad {

namespace {
constexpr int HK = 64;                          // halves per k-step row (128 bytes)
constexpr int KSTEP = 384 * 128, H16_LDS = 3 * KSTEP;
__device__ __attribute__((aligned(128))) unsigned short g_zero_page_h[64];     // 128 B of zeros

#ifndef PP_PHASES
#define PP_PHASES 2               // phases per K-tile of the ping-pong kernel: 2 (32-MFMA clusters) or 4 (16)
#endif
#ifndef PP_ABLATE
#define PP_ABLATE 0               // development builds only (tools/…): 1 no staging DMA, 2 staging from cache-resident rows (5: the pixel rows only; 7: 32 real pixel rows), 3 fragments read once, 4 one MFMA per quadrant
#endif
#define GH_WAIT_BARRIER(N)                                                          \
    do {                                                                            \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                       \
        __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14));       \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    } while (0)

thread_local int g_bad = 0;
int g_h16_cus = 256;                            // workgroups of a persistent launch (one per CU, a multiple of 8: the XCD-aware tile walk)

// GroupNorm statistics in the epilogue (GemmH16Args::stats): a lane adds the four channels of its pixel of a 16 x 16 accumulator tile,
// as the f16 values the consumer will read; stats_store reduces over the 16 pixel lanes of the quad's row group (fixed order: DPP row rotations) and
// lane 0 of the group writes the (sum, sum of squares) of the wave's 64 pixels x 4 channels.
// sum over the 16 lanes of a DPP row (the 16 pixel lanes of an accumulator tile's row quad): four rotate-and-add steps, no LDS traffic
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0x128>(v);      // row_ror:8
    v = dpp_add<0x124>(v);      // row_ror:4
    v = dpp_add<0x122>(v);      // row_ror:2
    return dpp_add<0x121>(v);   // row_ror:1
}
// (v_dot2_f32_f16 on the f16 pairs: two instructions per pair for the sum and the sum of squares — the products of f16 values are
// exact in fp32 — instead of convert-back / add / fma per value: 0.3 % of a UNet evaluation, the epilogues are bound by their store issue)
__device__ __forceinline__ void stats_add(float& s1, float& s2, const f32x4& v) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 one = f16x2{(_Float16)1.f, (_Float16)1.f};
    const f16x2 a = f16x2{(_Float16)v[0], (_Float16)v[1]}, b = f16x2{(_Float16)v[2], (_Float16)v[3]};
    s1 = __builtin_amdgcn_fdot2(a, one, s1, false);
    s1 = __builtin_amdgcn_fdot2(b, one, s1, false);
    s2 = __builtin_amdgcn_fdot2(a, a, s2, false);
    s2 = __builtin_amdgcn_fdot2(b, b, s2, false);
}
// Blocks at or past nblk (the pixel tail of the last tile) are not written: the statistics array holds exactly N / stats_px blocks.
__device__ __forceinline__ void stats_store(float s1, float s2, float* stats, long blk, long nblk, int quads, int quad, int r16) {
    s1 = row_sum16(s1);
    s2 = row_sum16(s2);
    if (r16 == 0 && blk < nblk) *(float2*)(stats + ((size_t)blk * quads + quad) * 2) = float2{s1, s2};
}

// grouped conv: group z of a launch is an ordinary dense problem on shifted pointers
__device__ __forceinline__ void select_group(GemmH16Args& a, int z) {
    if (a.groups > 1) {
        a.A += (size_t)z * a.taps * a.M * a.K;
        a.X += z * a.K;
        const int mo = z * a.M;
        if (a.C) a.C += mo;
        if (a.C16) a.C16 += mo;
        if (a.shift) a.shift += mo;
        if (a.res) a.res += mo;
        if (a.res16) a.res16 += mo;
        if (a.stats) a.stats += (mo >> 2) * 2;
    }
}
}  // namespace

template <int BM, bool TWO>
__global__ void __launch_bounds__(512, 2) gemm_h16_kernel(GemmH16Args a) {
    select_group(a, blockIdx.z);
    constexpr int BN = 384 - BM, WN = BN / 64;          // waves along N (2 or 4); along M: BM / 64
    constexpr int AP = BM / 64, XP = 6 - AP;            // 64-row staging pieces (8 KiB each) holding A rows / X rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN, q = lane >> 4, r16 = lane & 15;
    const long n0 = (long)blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const int steps_per_tap = a.K / HK, nsteps = a.taps * steps_per_tap;
    // staging: piece p row p*64 + rloc (rloc = wave*8 + lane/8); LDS slot lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7),
    // and (row >> 1) & 7 = (rloc >> 1) & 7 for every piece (64 | piece base)
    const int rloc = wv * 8 + (lane >> 3), ch8 = ((lane & 7) ^ ((rloc >> 1) & 7)) * 8;
    const h16_t* zero = g_zero_page_h;
    const h16_t* arow[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) arow[p] = a.A + (size_t)(m0 + p * 64 + rloc) * a.K + ch8;      // M % BM == 0 (launcher)
    long xpix[XP];
    int xy[XP], xx[XP];
    bool xok[XP];
    const int st = a.stride > 1 ? a.stride : 1;
    const int Wo = (a.W - 1) / st + 1, Ho = (a.H - 1) / st + 1, hw = Ho * Wo;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        const long n = n0 + p * 64 + rloc;
        xok[p] = n < a.N;
        xpix[p] = 0; xy[p] = 0; xx[p] = 0;
        if (xok[p]) {
            const long b = n / hw;
            const int pix = (int)(n - b * hw);
            xy[p] = (pix / Wo) * st; xx[p] = (pix % Wo) * st;
            xpix[p] = b * a.H * a.W;
        }
    }
    auto stage = [&](int ks, char* base) {               // base: this k-step's 48 KiB (rows 0..383 of 128 B)
        const int kq = ks / a.taps, tap = ks - kq * a.taps, kc = kq * HK;      // the taps of a 64-channel slice back to back (every kernel of the family: one summation order)
        char* la = base + wv * 1024;
#pragma unroll
        for (int p = 0; p < AP; ++p) glds16(arow[p] + (size_t)tap * a.M * a.K + kc, la + p * 8192);
        const int dy = a.taps == 9 ? tap / 3 - 1 : 0, dx = a.taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const h16_t* src = zero + (lane & 7) * 8;
            const int yy = xy[p] + dy, xq = xx[p] + dx;
            if (xok[p] && (unsigned)yy < (unsigned)a.H && (unsigned)xq < (unsigned)a.W) {
                const long pix = xpix[p] + (long)yy * a.W + xq;
                src = (TWO && kc >= a.ksplit) ? a.X2 + pix * a.ldx2 + (kc - a.ksplit) + ch8 : a.X + pix * a.ldx + kc + ch8;
            }
            glds16(src, la + (AP + p) * 8192);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment of k-half h (k = 32 h .. 32 h + 31): chunk 4 h + q of row r16 of a 16-row tile, at slot chunk ^ ((r16 >> 1) & 7)
    const int sw = (r16 >> 1) & 7;
    const int frag0 = r16 * 128 + ((q ^ sw) * 16), frag1 = r16 * 128 + (((4 + q) ^ sw) * 16);
    stage(0, smem);
    if (nsteps > 1) stage(1, smem + KSTEP);
    int slot = 0;
    for (int p = 0; p < nsteps; ++p) {
        // k-step p landed (the 6 pieces of k-step p+1 may still fly); every wave is done reading k-step p-1
        if (p + 1 < nsteps) { GH_WAIT_BARRIER(6); } else { GH_WAIT_BARRIER(0); }
        if (p + 2 < nsteps) stage(p + 2, smem + (slot == 0 ? 2 : slot - 1) * KSTEP);        // into the slot k-step p-1 occupied
        const char* A0 = smem + slot * KSTEP + wm * 8192;
        const char* B0 = smem + slot * KSTEP + BM * 128 + wn * 8192;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fo = h ? frag1 : frag0;
            f16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const f16x8*)(A0 + i * 2048 + fo);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const f16x8*)(B0 + j * 2048 + fo);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    // epilogue: bias, optional residual; fp32 map and / or its f16 twin.  All loads come first: hipcc cannot know that `res` and
    // the outputs do not overlap and would otherwise wait (vmcnt(0)) for every tile's store before it loads the next residual chunk —
    // sixteen serial HBM round trips per workgroup.  (res == C, an in-place add, stays correct: a lane reads what it later writes.)
    long nn[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long n = n0 + wn * 64 + j * 16 + r16;
        ok[j] = n < a.N;
        nn[j] = (ok[j] ? n : a.N - 1) * a.ldc + m0 + wm * 64 + q * 4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 b4 = a.shift ? *(const float4*)(a.shift + m0 + wm * 64 + i * 16 + q * 4) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
    }
    if (a.res) {
        float4 rr[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[i][j] = *(const float4*)(a.res + nn[j] + i * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[i][j][0] += rr[i][j].x; acc[i][j][1] += rr[i][j].y; acc[i][j][2] += rr[i][j].z; acc[i][j][3] += rr[i][j].w; }
    }
    if (a.res16) {
        f16x4 rh[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) rh[i][j] = *(const f16x4*)(a.res16 + nn[j] + i * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)rh[i][j][r];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!ok[j]) continue;
            f32x4 v = acc[i][j];
            if (a.relu) v = f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])};
            if (a.C) *(float4*)(a.C + nn[j] + i * 16) = float4{v[0], v[1], v[2], v[3]};
            if (a.C16) *(f16x4*)(a.C16 + nn[j] + i * 16) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            stats_add(s1, s2, v);
        }
        if (a.stats && a.stats_px != 16) stats_store(s1, s2, a.stats, (n0 + wn * 64) >> 6, a.N >> 6, a.ldc >> 2, (m0 + wm * 64 + i * 16 + q * 4) >> 2, r16);
    }
    if (a.stats && a.stats_px == 16) {          // 16-pixel samples (4 x 4 maps): one statistics block per 16 x 16 accumulator tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1 = 0.f, s2 = 0.f;
                f32x4 v = acc[i][j];
                if (a.relu) v = f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])};
                if (ok[j]) stats_add(s1, s2, v);
                stats_store(s1, s2, a.stats, (n0 + wn * 64 + j * 16) >> 4, a.N >> 4, a.ldc >> 2, (m0 + wm * 64 + i * 16 + q * 4) >> 2, r16);
            }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile (layers with M % 256 == 0, one input map, enough pixels to fill the chip): wave tile 128 x 64 = 32 accumulator tiles,
// 64 MFMAs per wave and k-step for 24 fragment reads and 8 DMA pieces (the 256 x 128 kernel above: 32 MFMAs for 16 + 6).  A k-step
// (512 rows x 128 B = 64 KiB) lives in a 2-slot ring; what makes two slots enough is that fragments are prefetched into REGISTERS a
// phase ahead, as in the split-f16 kernel (gemm_f32.hip): a k-step computes in four phases of 16 MFMAs,
//     (A0,B)k0  (A1,B)k0  (A0,B)k1 | (A1,B)k1          A0 / A1: the wave's row tiles 0-3 / 4-7, k0 / k1: the two 32-wide k halves
// each phase reading the A unit (4 x ds_read_b128) of the next phase — and every second one the next B unit — into the register set
// the previous phase released (two A sets, two B sets: 64 fragment registers).  The slot of k-step s has been read completely after
// phase 3; the ONE barrier sits there: behind it k-step s+1 has landed (vmcnt(0): nothing younger is in flight), phase 4 reads its
// first units from the other slot, and the eight pieces of k-step s+2 go into the slot just freed under the MFMAs of phase 4.
// Weight pieces go out in the saddr form (inline asm), the gathered pixel rows through the builtin (64-bit lane addresses, recomputed
// once per tap); rows outside the image come from a zero page long enough to take the k offset.
// ----------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int BIG_LDS_256 = 2 * 512 * 128, BIG_LDS_128 = 2 * 640 * 128, BIG_KMAX = 2048;      // 128 KiB / 160 KiB (all of a CU's LDS)
__device__ __attribute__((aligned(128))) unsigned short g_zero_page_big[BIG_KMAX + 64];
// Optional non-temporal hint on the weight pieces (a weight K-tile is read once per workgroup and tile, and its 32 KiB pass through
// the CU's 32 KiB L1 between two taps that read nearly the same pixel rows).
#ifndef H16_NT_WEIGHTS
#define H16_NT_WEIGHTS 0          // measured: 2 % slower on the UNet's convs (the 1 x 1 / qkv convs 10 %), the pixel rows do not stay in L1 either way
#endif
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds) {
#if H16_NT_WEIGHTS
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
#endif
}
__device__ __forceinline__ void dma16v(const void* vaddr, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory");
}
}  // namespace

template <int BM>               // 256: 256 x 256 tile, waves 2 (M) x 4 (N); 128: 128 x 512 tile, waves 1 x 8 (the 128-channel layers)
__global__ void __launch_bounds__(512, 2) gemm_h16_big_kernel(GemmH16Args a) {
    select_group(a, blockIdx.y);
    constexpr int BN = BM == 256 ? 256 : 512, AP = BM / 64, XP = BN / 64, NP = AP + XP, SLOTB = (BM + BN) * 128, WN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN, q = lane >> 4, r16 = lane & 15;
    // workgroup -> tile: the M / 256 row blocks of one pixel tile run back to back on one XCD (id % 8): their pixel rows are L2 hits
    const unsigned ny = (unsigned)(a.M / BM), jx = blockIdx.x >> 3;
    const unsigned tile_x = (jx / ny) * 8u + (blockIdx.x & 7u);
    if ((long)tile_x * BN >= a.N) return;                     // the grid is padded to 8 * ny * ceil(nx / 8)
    const long n0 = (long)tile_x * BN;
    const int m0 = (int)(jx % ny) * BM;
    const int steps_per_tap = a.K / HK, nsteps = a.taps * steps_per_tap;
    const int rloc = wv * 8 + (lane >> 3), ch8 = ((lane & 7) ^ ((rloc >> 1) & 7)) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    // weights: uniform base + one lane offset; piece p adds 64 rows
    const char* Ab = (const char*)(a.A + (size_t)m0 * a.K);
    const unsigned voffA = (unsigned)((rloc * a.K + ch8) * 2);
    const size_t a_piece = (size_t)64 * a.K * 2, a_tap = (size_t)a.M * a.K * 2;
    // pixel rows: (image base pixel, y, x) per staged row of this lane, 4 pieces of 64 rows
    const int st = a.stride > 1 ? a.stride : 1;
    const int Wo = (a.W - 1) / st + 1, Ho = (a.H - 1) / st + 1, hw = Ho * Wo;
    int xpix[XP], xyx[XP];                                     // image base pixel (or -1: row past N), (y << 16) | x
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        const long n = n0 + p * 64 + rloc;
        xpix[p] = -1; xyx[p] = 0;
        if (n < a.N) {
            const int b = (int)(n / hw), pix = (int)(n - (long)b * hw);
            xpix[p] = b * a.H * a.W;
            xyx[p] = (((pix / Wo) * st) << 16) | ((pix % Wo) * st);
        }
    }
    const h16_t* xrow[XP];                                     // this tap's source row (k offset 0) per piece, or the zero page
    auto tap_rows = [&](int tap) {
        const int dy = a.taps == 9 ? tap / 3 - 1 : 0, dx = a.taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int yy = (xyx[p] >> 16) + dy, xq = (xyx[p] & 0xffff) + dx;
            const bool ok = xpix[p] >= 0 && (unsigned)yy < (unsigned)a.H && (unsigned)xq < (unsigned)a.W;
            xrow[p] = ok ? a.X + ((long)xpix[p] + yy * a.W + xq) * a.ldx + ch8 : g_zero_page_big + (lane & 7) * 8;
        }
    };
    int st_tap = 0, st_kq = 0;                                // staging cursor
    const char* st_a = Ab;
    tap_rows(0);
    auto st_advance = [&]() {
        if (++st_tap == a.taps) { st_tap = 0; ++st_kq; }
        tap_rows(st_tap);
        st_a = Ab + (size_t)st_tap * a_tap + (size_t)st_kq * 128;
    };
    auto piece = [&](int k, unsigned slot_lds) {              // k < XP: pixel rows, then the AP weight pieces
        if (k < XP) dma16v(xrow[k] + st_kq * HK, slot_lds + BM * 128 + k * 8192 + wv * 1024);
        else dma16s(st_a + (size_t)(k - XP) * a_piece, voffA, slot_lds + (k - XP) * 8192 + wv * 1024);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (r16 >> 1) & 7;
    const int fk[2] = {r16 * 128 + ((q ^ sw) * 16), r16 * 128 + (((4 + q) ^ sw) * 16)};
    const int aoff = wm * 16384, boff = BM * 128 + wn * 8192;
    f16x8 AX[4], AY[4], BP[4], BQ[4];
    auto ldA = [&](f16x8 (&U)[4], const char* slot, int half, int kh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) U[i] = *(const f16x8*)(slot + aoff + half * 8192 + i * 2048 + fk[kh]);
    };
    auto ldB = [&](f16x8 (&U)[4], const char* slot, int kh) {
#pragma unroll
        for (int j = 0; j < 4; ++j) U[j] = *(const f16x8*)(slot + boff + j * 2048 + fk[kh]);
    };
#define BIG_MFMA4(AU, BU, i0, ii)                                                                                    \
    do {                                                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                             \
            acc[(i0) + (ii)][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AU[ii], BU[j_], acc[(i0) + (ii)][j_], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)

    // prologue: k-steps 0, 1 staged; k-step 0 landed; its first units in registers
#pragma unroll
    for (int k = 0; k < NP; ++k) piece(k, lds0);
    st_advance();
    if (nsteps > 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) piece(k, lds0 + SLOTB);
        st_advance();
        if (NP == 8) { GH_WAIT_BARRIER(8); } else { GH_WAIT_BARRIER(10); }
    } else {
        GH_WAIT_BARRIER(0);
    }
    ldA(AX, smem, 0, 0);
    ldB(BP, smem, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < nsteps; ++s) {
        const char* cur = smem + (s & 1) * SLOTB;
        const char* nxt = smem + ((s & 1) ^ 1) * SLOTB;
        const unsigned cur_lds = lds0 + (s & 1) * SLOTB;
        // phase 1: (A0, B) k0; read A1 k0
        ldA(AY, cur, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        BIG_MFMA4(AX, BP, 0, 0); BIG_MFMA4(AX, BP, 0, 1); BIG_MFMA4(AX, BP, 0, 2); BIG_MFMA4(AX, BP, 0, 3);
        // phase 2: (A1, B) k0; read A0 k1, B k1
        ldA(AX, cur, 0, 1);
        ldB(BQ, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        BIG_MFMA4(AY, BP, 4, 0); BIG_MFMA4(AY, BP, 4, 1); BIG_MFMA4(AY, BP, 4, 2); BIG_MFMA4(AY, BP, 4, 3);
        // phase 3: (A0, B) k1; read A1 k1
        ldA(AY, cur, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        BIG_MFMA4(AX, BQ, 0, 0); BIG_MFMA4(AX, BQ, 0, 1); BIG_MFMA4(AX, BQ, 0, 2); BIG_MFMA4(AX, BQ, 0, 3);
        // k-step s+1 landed, every wave holds its last fragments of this slot
        GH_WAIT_BARRIER(0);
        // phase 4: (A1, B) k1; read A0 k0, B k0 of k-step s+1; stage k-step s+2 into this slot
        ldA(AX, nxt, 0, 0);
        ldB(BP, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef BIG_NO_DMA
        const bool more = false;
#else
        const bool more = s + 2 < nsteps;
#endif
        if (more) { piece(0, cur_lds); piece(1, cur_lds); if (NP == 10) piece(8, cur_lds); __builtin_amdgcn_sched_barrier(0); }
        BIG_MFMA4(AY, BQ, 4, 0);
        if (more) { piece(2, cur_lds); piece(3, cur_lds); if (NP == 10) piece(9, cur_lds); __builtin_amdgcn_sched_barrier(0); }
        BIG_MFMA4(AY, BQ, 4, 1);
        if (more) { piece(4, cur_lds); piece(5, cur_lds); __builtin_amdgcn_sched_barrier(0); }
        BIG_MFMA4(AY, BQ, 4, 2);
        if (more) { piece(6, cur_lds); piece(7, cur_lds); __builtin_amdgcn_sched_barrier(0); }
        BIG_MFMA4(AY, BQ, 4, 3);
        if (more) st_advance();
    }
#undef BIG_MFMA4
    __builtin_amdgcn_s_waitcnt(0xC07F);          // the reads past the last k-step (never used) are retired
    // epilogue: bias, optional residual (per row tile: its four loads ahead of its stores), fp32 map and / or f16 twin
    long nn[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long n = n0 + wn * 64 + j * 16 + r16;
        ok[j] = n < a.N;
        nn[j] = (ok[j] ? n : a.N - 1) * a.ldc + m0 + wm * 128 + q * 4;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 b4 = a.shift ? *(const float4*)(a.shift + m0 + wm * 128 + i * 16 + q * 4) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
        if (a.res) {
            float4 rr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[j] = *(const float4*)(a.res + nn[j] + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[i][j][0] += rr[j].x; acc[i][j][1] += rr[j].y; acc[i][j][2] += rr[j].z; acc[i][j][3] += rr[j].w; }
        }
        if (a.res16) {
            f16x4 rh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rh[j] = *(const f16x4*)(a.res16 + nn[j] + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)rh[j][r];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!ok[j]) continue;
            f32x4 v = acc[i][j];
            if (a.relu) v = f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])};
            if (a.C) *(float4*)(a.C + nn[j] + i * 16) = float4{v[0], v[1], v[2], v[3]};
            if (a.C16) *(f16x4*)(a.C16 + nn[j] + i * 16) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            stats_add(s1, s2, v);
        }
        if (a.stats) stats_store(s1, s2, a.stats, (n0 + wn * 64) >> 6, a.N >> 6, a.ldc >> 2, (m0 + wm * 128 + i * 16 + q * 4) >> 2, r16);
    }
}


// Epilogue of a persistent tile, straight from the accumulators (acc[i][j]: row tile i, pixel tile j of the wave's 128 x 64 output;
// wm / wn: the wave's position in the tile): bias, residual, ReLU, GroupNorm statistics, stores; leaves the accumulators zeroed.
// RES (16-byte form): the f16 residual as a COMPILE-TIME branch.  With run-time `if (a.res16)` blocks around the loads and around their
// uses, hipcc's wait-count pass sees a path on which a residual load is still pending at the end of the epilogue, carries it round
// the persistent loop and puts an `s_waitcnt vmcnt(0)` in front of the first fragment read that reuses the register — a full drain
// of the staging DMA in EVERY K-tile (found in the 128 x 512 ping-pong kernel's ISA).
template <int BM, bool WIDE, int BN = (BM == 256 ? 256 : 512), int MT = 8, bool RES = false>      // MT: accumulator row tiles per wave (wave rows = 16 MT)
__device__ __forceinline__ void pers_epilogue_impl(const GemmH16Args& a, f32x4 (&acc)[MT][4], long ctile, int ny, int wm, int wn, int q, int r16) {
    const long tx = ctile / ny;
    const int m0 = (int)(ctile - tx * ny) * BM;
    const long n0 = tx * BN;
    unsigned nrow[4];                                        // element offset of the lane's four pixel rows (N * ldc < 2^31: launcher)
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long n = n0 + wn * 64 + j * 16 + r16;
        ok[j] = n < a.N;
        nrow[j] = (unsigned)((ok[j] ? n : a.N - 1) * a.ldc);
    }
    const int mw = m0 + wm * (MT * 16);                            // first output channel of this wave
    if constexpr (WIDE) {
        // f16 map out (the common case): 16-BYTE stores.  A lane holds 4 channels x 1 pixel per accumulator tile (8 bytes as f16);
        // lanes q / q ^ 1 exchange halves (one v_permlane16_swap per dword) so that every lane ends up with 8 consecutive
        // channels — even q: of row tile 2p, odd q: of row tile 2p + 1 — and a row tile pair costs one 16-byte store per pixel
        // block instead of two 8-byte ones (the epilogue of a short-K tile is bound by the NUMBER of its store instructions:
        // 32 per lane were ~9 us of a 17 us K = 256 tile).  The f16 residual is fetched in the same chunks and un-swapped.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
        for (int p = 0; p < MT / 2; ++p) {                   // row tile pairs (2p, 2p + 1)
            const int t0 = 2 * p, t1 = t0 + 1;
            const unsigned coff = (unsigned)(mw + (t0 + (q & 1)) * 16 + (q >> 1) * 8);
            constexpr int JB = (BM == 256 || BN == 256) ? 4 : 2;              // residual chunks in flight (the 128 x 512 form has 16 more registers of staging state)
            u32x4 rc[JB];
            const float4 z4 = float4{0.f, 0.f, 0.f, 0.f};
            const float4 b0 = a.shift ? *(const float4*)(a.shift + mw + t0 * 16 + q * 4) : z4;
            const float4 b1 = a.shift ? *(const float4*)(a.shift + mw + t1 * 16 + q * 4) : z4;
            float s10 = 0.f, s20 = 0.f, s11 = 0.f, s21 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (RES && j % JB == 0) {
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) rc[jj] = *(const u32x4*)(a.res16 + nrow[j + jj] + coff);
                }
                f32x4 v0 = acc[t0][j], v1 = acc[t1][j];
                v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
                v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
                if (RES) {
                    const auto h0 = __builtin_amdgcn_permlane16_swap(rc[j % JB][0], rc[j % JB][2], false, false);
                    const auto h1 = __builtin_amdgcn_permlane16_swap(rc[j % JB][1], rc[j % JB][3], false, false);
                    const f16x4 hx = __builtin_bit_cast(f16x4, u32x2{h0[0], h1[0]}), hy = __builtin_bit_cast(f16x4, u32x2{h0[1], h1[1]});
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] += (float)hx[r]; v1[r] += (float)hy[r]; }
                }
                if (a.relu) {
                    v0 = f32x4{relu_nan(v0[0]), relu_nan(v0[1]), relu_nan(v0[2]), relu_nan(v0[3])};
                    v1 = f32x4{relu_nan(v1[0]), relu_nan(v1[1]), relu_nan(v1[2]), relu_nan(v1[3])};
                }
                if (ok[j]) { stats_add(s10, s20, v0); stats_add(s11, s21, v1); }
                const u32x2 ox = __builtin_bit_cast(u32x2, f16x4{(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3]});
                const u32x2 oy = __builtin_bit_cast(u32x2, f16x4{(_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]});
                const auto x0 = __builtin_amdgcn_permlane16_swap(ox[0], oy[0], false, false);
                const auto x1 = __builtin_amdgcn_permlane16_swap(ox[1], oy[1], false, false);
                if (ok[j]) __builtin_nontemporal_store(u32x4{x0[0], x1[0], x0[1], x1[1]}, (u32x4*)(a.C16 + nrow[j] + coff));     // (a map is 0.3-1 GB: streamed, −0.8 % of a UNet evaluation)
                acc[t0][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[t1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (a.stats) {
                stats_store(s10, s20, a.stats, (n0 + wn * 64) >> 6, a.N >> 6, a.ldc >> 2, (mw + t0 * 16 + q * 4) >> 2, r16);
                stats_store(s11, s21, a.stats, (n0 + wn * 64) >> 6, a.N >> 6, a.ldc >> 2, (mw + t1 * 16 + q * 4) >> 2, r16);
            }
        }
    } else {
#pragma unroll
        for (int ii = 0; ii < MT; ++ii) {                    // fp32 map out and / or fp32 residual (rare: a classifier's last block): plain form
            const float4 b4 = a.shift ? *(const float4*)(a.shift + mw + ii * 16 + q * 4) : float4{0.f, 0.f, 0.f, 0.f};
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned off = nrow[j] + mw + ii * 16 + q * 4;
                f32x4 v = acc[ii][j];
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                if (a.res) { const float4 rr = *(const float4*)(a.res + off); v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
                if (a.res16) {
                    const f16x4 rh = *(const f16x4*)(a.res16 + off);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rh[r];
                }
                if (a.relu) v = f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])};
                if (ok[j]) {
                    if (a.C) *(float4*)(a.C + off) = float4{v[0], v[1], v[2], v[3]};
                    if (a.C16) *(f16x4*)(a.C16 + off) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    stats_add(s1, s2, v);
                }
            }
            if (a.stats) stats_store(s1, s2, a.stats, (n0 + wn * 64) >> 6, a.N >> 6, a.ldc >> 2, (mw + ii * 16 + q * 4) >> 2, r16);
        }
#pragma unroll
        for (int ii = 0; ii < MT; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int BM, bool WIDE, int BN = (BM == 256 ? 256 : 512), int MT = 8>
__device__ __forceinline__ void pers_epilogue(const GemmH16Args& a, f32x4 (&acc)[MT][4], long ctile, int ny, int wm, int wn, int q, int r16) {
    if (WIDE && a.res16) pers_epilogue_impl<BM, WIDE, BN, MT, true>(a, acc, ctile, ny, wm, wn, q, r16);
    else pers_epilogue_impl<BM, WIDE, BN, MT, false>(a, acc, ctile, ny, wm, wn, q, r16);
}

// ----------------------------------------------------------------------------------------------------------------------------
// Ping-pong form of the persistent tile (round 4, second form).  Same tile (256 x 256, or 128 x 512 for the 128-channel layers),
// same 8 waves with a 128 x 64 output each, same ring that never drains across tiles — but the two waves of a SIMD no longer do the
// same thing at the same time.  Waves 0-3 and waves 4-7 (one of each per SIMD) run half a phase apart: while one group contracts
// a quadrant of its output (16 MFMAs, the matrix pipe's 256 cycles), the other reads its next fragments from LDS and issues its
// share of the staging DMA; a barrier, and the roles swap.  In the first persistent form both waves of a SIMD issued their eight
// DMA pieces in the same phase and then both waited for them: the stamps showed the matrix pipe busy half of a k-step.
//   K-tile (64 deep) = four half-tiles: A0 A1 (rows h * 64 .. + 64 of each wave row's 128) and B0 B1 (pixels h * 32 .. + 32 of each
//   wave column's 64); two K-tiles resident (2 x 64 KiB, or 2 x 80 KiB for 128 x 512).  Four phases per K-tile:
//       phase 0: read A0 (8 x ds_read_b128), B0 (4)   stage B1 of the cursor K-tile     quadrant (0, 0)
//       phase 1: read B1 (4)                          stage A1, advance the cursor      quadrant (0, 1)
//       phase 2: read A1 (8)                          stage A0 of the new cursor        quadrant (1, 1)
//       phase 3: (B0 is still in registers)           stage B0; wait                    quadrant (1, 0)
//   each phase = { reads + DMA issue | s_barrier | lgkmcnt(0), 16 MFMAs | s_barrier }.  The cursor runs 1.5 K-tiles ahead; a half-tile
//   is overwritten two phases or more after the phase that last read it (with the groups half a phase apart, one phase is not
//   enough: the later group's reads retire behind the barrier the earlier group's next phase starts at).  ONE counted wait per
//   K-tile, in phase 3 ahead of its first barrier: all of the next K-tile has landed, the two half-tiles just issued stay in flight;
//   the first read of that K-tile comes a phase later, behind a barrier both groups' waits precede.
//   At a tile's end the leading group waits one barrier for the other, both run the epilogue at once, and the trailing group waits
//   one barrier to fall half a phase behind again.
// ----------------------------------------------------------------------------------------------------------------------------
template <int BM, bool WIDE>
__global__ void __launch_bounds__(512, 2) gemm_h16_pp_kernel(GemmH16Args a, int nx) {
    constexpr int WR = BM / 128, WC = 8 / WR, BN = WC * 64;
    constexpr int AH = BM / 2, BH = BN / 2;                         // rows of an A / B half-tile
    constexpr int APW = AH / 64, BPW = BH / 64;                     // 1-KiB pieces (8 rows x 128 B) per wave and half-tile
    constexpr int AHB = AH * 128, BHB = BH * 128, BUFB = 2 * (AHB + BHB);      // a K-tile in LDS: A0 A1 B0 B1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = WR == 2 ? wv >> 2 : 0, wc = WR == 2 ? wv & 3 : wv, grp = wv >> 2, q = lane >> 4, r16 = lane & 15;
    const int ny = a.M / BM;
    const long T = (long)nx * ny;
    const int G8 = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jw = (int)(blockIdx.x >> 3);
    const long chunk = (((T + 7) / 8 + ny - 1) / ny) * ny;          // whole pixel tiles per XCD
    const long lo = xcd * chunk, hi = lo + chunk < T ? lo + chunk : T;
    long ctile = lo + jw;                                            // compute cursor (tile id)
    if (ctile >= hi) return;
    const int my_tiles = (int)((hi - ctile + G8 - 1) / G8);
    const int steps_per_tap = a.K / HK, nsteps = a.taps * steps_per_tap;
    const int l8 = lane >> 3, ch8 = ((lane & 7) ^ (((wv & 1) * 4 + (l8 >> 1)) & 7)) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned voffAp[APW][2];                                         // weight piece (p, h): its rows' byte offset + the lane's (one register each:
#pragma unroll                                                        //  the scalar form of these sums lived in spilled SGPRs, three v_readlane per piece)
    for (int p = 0; p < APW; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) voffAp[p][h] = (unsigned)(((p * 128 + h * 64 + wv * 8 + l8) * a.K + ch8) * 2);
    const size_t a_tap = (size_t)a.M * a.K * 2;
    const int st = a.stride > 1 ? a.stride : 1;
    const int Wo = (a.W - 1) / st + 1, Ho = (a.H - 1) / st + 1, hw = Ho * Wo;
    // ---- staging cursor: tile, tap, K-tile inside the tap; per-lane pixel rows of the staging tile ---------------------------
    long stile = ctile;
    int st_tap = 0, st_kq = 0, st_left = my_tiles * nsteps;         // K-tiles not completely staged yet
    unsigned st_lds = lds0 + wv * 1024;                              // the cursor K-tile's buffer, at this wave's 1-KiB piece of a 64-row group
    // per staged pixel row of the lane, [h * BPW + p]: flat input pixel of tap (0, 0) and the taps that fall inside the image (nine
    // bits; 0: row past N) — a piece's address is then one select and one multiply-add (the load parts are this kernel's critical path)
    int xflat[2 * BPW];
    unsigned xmask[2 * BPW];
    int st_toff = a.taps == 9 ? -a.W - 1 : 0;                        // the staging tap's pixel offset, dy * W + dx
    const char* Ab = nullptr;
    const char* st_a = nullptr;
    auto tile_rows = [&](long id) __attribute__((always_inline)) {       // (inlined by force: out of line, the cursor state it writes would live in memory — and in VGPRs)
        const long tx = id / ny;
        const int mb = (int)(id - tx * ny);
        Ab = (const char*)(a.A + (size_t)mb * BM * a.K);
        const unsigned n0 = (unsigned)tx * BN;                       // N < 2^31 (launcher): 32-bit pixel arithmetic
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < BPW; ++p) {
                const unsigned n = n0 + (2 * p + (wv >> 2)) * 64 + h * 32 + (wv & 3) * 8 + l8;
                xflat[h * BPW + p] = 0; xmask[h * BPW + p] = 0u;
                if (n < (unsigned)a.N) {
                    const unsigned b = n / (unsigned)hw, pix = n - b * (unsigned)hw, py = pix / (unsigned)Wo;
                    const int y = (int)(py * st), x = (int)((pix - py * Wo) * st);
                    xflat[h * BPW + p] = (int)(b * (unsigned)(a.H * a.W)) + y * a.W + x;
                    unsigned bits = 0x1ffu;
                    if (a.taps == 9) {
                        bits = 0u;
#pragma unroll
                        for (int t = 0; t < 9; ++t)
                            if ((unsigned)(y + t / 3 - 1) < (unsigned)a.H && (unsigned)(x + t % 3 - 1) < (unsigned)a.W) bits |= 1u << t;
                    }
                    xmask[h * BPW + p] = bits;
                }
            }
    };
    // K-tile order inside a tile: the nine taps of one 64-channel slice back to back, then the next slice.  The taps of a slice read
    // the SAME pixel rows shifted by a pixel or a row (32 KiB per workgroup and slice), so eight of the nine reads hit the XCD's L2;
    // tap-major order (all slices of a tap, then the next tap) re-reads the tile's whole input — 256 px x K x 2 B per workgroup,
    // 8 MB per XCD at K = 512 — nine times from beyond the 4 MB L2 (measured: 2 % of the UNet's conv time).  Every kernel of the
    // family walks the K-tiles in this order, so a sample's f16 result does not depend on which kernel its batch size selects.
    const h16_t* st_x = a.X;                                         // the cursor slice's input map, row stride and channel offset:
    int st_ld = a.ldx, st_ko = 0;                                    // the second part of a concatenated input (th.cat(dim=1)) from a.ksplit on
    const h16_t* st_xl = a.X + ch8;
    auto slice_source = [&]() {
        const int kc = st_kq * HK;
        const bool second = a.X2 != nullptr && kc >= a.ksplit;
        st_x = second ? a.X2 : a.X;
        st_ld = second ? a.ldx2 : a.ldx;
        st_ko = second ? kc - a.ksplit : kc;
        st_xl = st_x + st_ko + ch8;                                  // the lane's chunk of pixel row 0 of this slice: a piece adds one row offset
    };
    int st_col = 0;                                                  // the staging tap's column (tap % 3)
    auto st_advance = [&]() __attribute__((always_inline)) {         // after the cursor K-tile's last half-tile (A1) has been issued
        --st_left;
        st_lds = lds0 + ((st_lds - lds0) ^ BUFB);                    // (the wave's 1-KiB offset lies below BUFB's bits)
        if (++st_tap == a.taps) {                                    // next slice (or tile): the addresses from scratch
            st_tap = 0; st_col = 0;
            if (++st_kq == steps_per_tap) {
                st_kq = 0;
                stile += G8;
                if (st_left > 0) tile_rows(stile);
            }
            st_toff = a.taps == 9 ? -a.W - 1 : 0;
            st_a = Ab + (size_t)st_kq * 128;
            slice_source();
        } else {                                                     // next tap of the slice: increments only (this runs in every load part)
            st_a += a_tap;
            if (++st_col == 3) { st_col = 0; st_toff += a.W - 2; } else ++st_toff;
        }
    };
    const h16_t* zrow = g_zero_page_big + (lane & 7) * 8;
    auto stageA = [&](int h) {                                       // weights: wave-uniform base + one lane offset
#if PP_ABLATE != 1
#pragma unroll
        for (int p = 0; p < APW; ++p)
            dma16s(PP_ABLATE == 2 ? Ab : st_a, voffAp[p][h], st_lds + h * AHB + p * 8192);
#endif
    };
    auto stageB = [&](int h) {                                       // gathered pixel rows (a tap outside the image: the zero page)
#pragma unroll
        for (int p = 0; p < BPW; ++p) {
            const int k = h * BPW + p;
            const bool ok = (xmask[k] >> st_tap) & 1u;
            const h16_t* src = ok ? st_xl + (unsigned)(xflat[k] + st_toff) * (unsigned)st_ld : zrow;      // (N * ldx < 2^31: launcher)
#if PP_ABLATE == 2 || PP_ABLATE == 5
            src = ok ? zrow + 64 : zrow;
#elif PP_ABLATE == 7
            src = ok ? st_x + (long)(l8 + 8 * (wv & 3)) * st_ld + ch8 + st_ko : zrow;      // 32 real (non-zero) rows: cache-resident, same operand statistics
#endif
#if PP_ABLATE != 1
            dma16v(src, st_lds + 2 * AHB + h * BHB + p * 8192);
#else
            asm volatile("" ::"v"(src));
#endif
        }
    };
    tile_rows(stile);
    st_a = Ab;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (r16 >> 1) & 7;
    const int fk[2] = {r16 * 128 + ((q ^ sw) * 16), r16 * 128 + (((4 + q) ^ sw) * 16)};
    const int aoff = wr * 8192, boff = 2 * AHB + wc * 4096;
    f16x8 AF[4][2], B0[2][2], B1[2][2];
    bool rd = true;                                                  // (PP_ABLATE == 3: fragments read once)
    auto ldA = [&](const char* buf, int i) {
        if (PP_ABLATE == 3 && !rd) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) AF[mt][kh] = *(const f16x8*)(buf + i * AHB + aoff + mt * 2048 + fk[kh]);
    };
    auto ldB = [&](f16x8 (&U)[2][2], const char* buf, int j) {
        if (PP_ABLATE == 3 && !rd) return;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) U[nt][kh] = *(const f16x8*)(buf + j * BHB + boff + nt * 2048 + fk[kh]);
    };
#define PP_BARRIER()                                 \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
    // the quadrant (i, j) of the wave's output: 4 x 2 accumulator tiles x the K-tile's two 32-deep halves
#define PP_MFMA(BU, i, j)                                                                                                    \
    _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                                                      \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)                                                                  \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                                              \
                if (PP_ABLATE != 4 || (mt_ | nt_ | kh_) == 0)                                                                \
                    acc[(i) * 4 + mt_][(j) * 2 + nt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                              \
                        AF[mt_][kh_], BU[nt_][kh_], acc[(i) * 4 + mt_][(j) * 2 + nt_], 0, 0, 0);
#define PP_QUAD(BU, i, j)                                                                                                    \
    do {                                                                                                                     \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        PP_MFMA(BU, i, j)                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
#define PP_QUAD2(BU, i, j, BV, i2, j2)                                                                                       \
    do {                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        PP_MFMA(BU, i, j)                                                                                                    \
        PP_MFMA(BV, i2, j2)                                                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
    // prologue: K-tile 0 staged whole, A0 / B0 of K-tile 1 behind it; K-tile 0 landed
    stageA(0); stageB(0); stageB(1); stageA(1);
    st_advance();
    if (st_left > 0) {
        stageA(0); stageB(0);
        if (APW + BPW == 4) { GH_WAIT_BARRIER(4); } else { GH_WAIT_BARRIER(5); }
    } else {
        GH_WAIT_BARRIER(0);
    }
    if (grp == 1) PP_BARRIER();                                      // the second group runs half a phase behind
    const int total = my_tiles * nsteps;
    int s = 0;                                                       // K-tile inside the compute tile
    for (int g = 0; g < total; ++g) {
        const char* cur = smem + (g & 1) * BUFB;
#if PP_PHASES == 2
        // phase A: quadrants (0, 0) and (0, 1); the cursor K-tile's B1 and A1 go out (its A0 / B0 went a phase earlier), the cursor advances
        ldB(B0, cur, 0);
        ldB(B1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        ldA(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st_left > 0) { stageB(1); stageA(1); st_advance(); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                          // the reads retire AHEAD of the barrier: what they read may be restaged a phase later
        PP_BARRIER();
        PP_QUAD2(B0, 0, 0, B1, 0, 1);
        PP_BARRIER();
        // phase B: quadrants (1, 1) and (1, 0); A0 and B0 of the new cursor K-tile; the next K-tile has landed (those two are younger)
        ldA(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (st_left > 0) {
            stageA(0); stageB(0);
            __builtin_amdgcn_sched_barrier(0);
            if (APW + BPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        PP_BARRIER();
        PP_QUAD2(B1, 1, 1, B0, 1, 0);
        PP_BARRIER();
#else
        // phase 0
        ldB(B0, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        ldA(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st_left > 0) stageB(1);
        PP_BARRIER();
        PP_QUAD(B0, 0, 0);
        PP_BARRIER();
        // phase 1
        ldB(B1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (st_left > 0) { stageA(1); st_advance(); }
        PP_BARRIER();
        PP_QUAD(B1, 0, 1);
        PP_BARRIER();
        // phase 2
        ldA(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = st_left > 0;
        if (more) stageA(0);
        PP_BARRIER();
        PP_QUAD(B1, 1, 1);
        PP_BARRIER();
        // phase 3: the next K-tile has landed (younger: the A0 and B0 half-tiles of the one after it)
        if (more) {
            stageB(0);
            __builtin_amdgcn_sched_barrier(0);
            if (APW + BPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        PP_QUAD(B0, 1, 0);
        PP_BARRIER();
#endif
        if (PP_ABLATE == 3 && g >= 1) rd = false;
        if (s != nsteps - 1) { ++s; continue; }
        // ---- the tile is complete: both groups run the epilogue at the same time -------------------------------------------------
        if (grp == 0) PP_BARRIER();
        pers_epilogue<BM, WIDE>(a, acc, ctile, ny, wr, wc, q, r16);
        ctile += G8;
        s = 0;
        if (grp == 1 && g + 1 < total) PP_BARRIER();
    }
#undef PP_QUAD
#undef PP_QUAD2
#undef PP_MFMA
#undef PP_BARRIER
}

// ----------------------------------------------------------------------------------------------------------------------------
// Slice-resident 3 x 3 convolution (round 4, third form): the ping-pong kernel above, but a tile's pixel rows are staged ONCE per
// 64-channel slice and the nine taps read them from LDS shifted, instead of nine gathered copies of (nearly) the same rows.
// What it removes is half of the kernel's staging traffic — the half that was measured to cost: with the pixel pieces served from
// cache-resident rows the UNet's convs ran 11 % faster (PP_ABLATE = 7), with the weight pieces alone unchanged.
//   LDS: weight ring, 2 K-tiles x (A0 | A1) x 16 KiB  = 64 KiB     (as in the ping-pong kernel: rows h * 64 .. + 64 of each wave row)
//        pixel slice, 2 buffers x (328 rows + a zero row) x 128 B = 82.25 KiB
//   slice rows: flat pixels n0 - W - 1 .. n0 + 255 + W + 1 of the NHWC map (the tile's 256 pixels and a halo of one image row and one
//   pixel on both sides), whatever images they belong to; a row outside the tensor is staged from the zero page.  The image border
//   is applied on the READ side: a lane's fragment of pixel p and tap (dy, dx) is slice row p + W + 1 + dy * W + dx if (y + dy, x + dx)
//   lies inside p's image, else the zero row (nine validity bits per pixel, computed once per tile).  Consecutive pixels are
//   consecutive rows at ANY alignment, so the chunk swizzle is by row & 7 (conflict-free for every start row; the ring's (row >> 1) & 7
//   is conflict-free only for 4-aligned starts).
//   K-tile order: slice-major, the nine taps of a slice back to back (the order of every kernel of the family).  Per K-tile a wave
//   issues four weight pieces and, during taps 0-5 of a slice, one piece of the NEXT slice (41 pieces per slice and workgroup).
//   Phases, waits and the half-phase offset of the two wave groups are the ping-pong kernel's.
//   (A 128-row variant — 128 x 256 tiles, 64 x 64 per wave, one phase per K-tile — measured no faster than the ping-pong kernel's
//   128 x 512 tile on the 128-channel layers and is not kept: HISTORY.md, round 4.)
// ----------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int SR_XROWS = 328, SR_XB = (SR_XROWS + 1) * 128, SR_ARING = 2 * 256 * 128, SR_LDS = SR_ARING + 2 * SR_XB;     // 149 760 B
}
template <bool WIDE>
__global__ void __launch_bounds__(512, 2) gemm_h16_sr_kernel(GemmH16Args a, int nx) {
    constexpr int BM = 256, BN = 256, AHB = 128 * 128, ABUF = 2 * AHB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3, grp = wv >> 2, q = lane >> 4, r16 = lane & 15;
    const int ny = a.M / BM;
    const long T = (long)nx * ny;
    const int G8 = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jw = (int)(blockIdx.x >> 3);
    const long chunk = (((T + 7) / 8 + ny - 1) / ny) * ny;          // whole pixel tiles per XCD
    const long lo = xcd * chunk, hi = lo + chunk < T ? lo + chunk : T;
    long ctile = lo + jw;                                            // compute cursor (tile id)
    if (ctile >= hi) return;
    const int my_tiles = (int)((hi - ctile + G8 - 1) / G8);
    const int nslices = a.K / HK, nsteps = 9 * nslices;
    const int l8 = lane >> 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned voffAp[2][2];                                           // weight piece (p, h): its rows' byte offset + the lane's (one register each)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            voffAp[p][h] = (unsigned)(((p * 128 + h * 64 + wv * 8 + l8) * a.K + ((lane & 7) ^ (((wv & 1) * 4 + (l8 >> 1)) & 7)) * 8) * 2);
    const size_t a_tap = (size_t)a.M * a.K * 2;
    const int W = a.W, hw = a.H * a.W, halo = W + 1;
    const int xpieces = (BN + 2 * halo + 7) >> 3;                    // 1-KiB pieces of a slice (41 at W = 32)
    const int up_wsh = __builtin_ctz((unsigned)W), up_hwsh = __builtin_ctz((unsigned)hw);      // (up2: log2 of the upsampled map's width / pixel count)
    for (int i = tid; i < 64; i += 512) ((unsigned*)(smem + SR_ARING + (i >> 5) * SR_XB + SR_XROWS * 128))[i & 31] = 0u;      // the two zero rows
    // ---- weight cursor (K-tiles, 1.5 ahead of the compute cursor) ----------------------------------------------------------------
    long atile = ctile;
    int a_tapi = 0, a_kq = 0, a_left = my_tiles * nsteps;           // K-tiles whose weights are not completely staged yet
    unsigned a_lds = lds0 + wv * 1024;
    const char* Ab = (const char*)(a.A + (size_t)(ctile % ny) * BM * a.K);
    const char* st_a = Ab;
    auto a_advance = [&]() __attribute__((always_inline)) {
        --a_left;
        a_lds = lds0 + ((a_lds - lds0) ^ ABUF);
        if (++a_tapi == 9) {
            a_tapi = 0;
            if (++a_kq == nslices) {
                a_kq = 0;
                atile += G8;
                Ab = (const char*)(a.A + (size_t)(atile % ny) * BM * a.K);
            }
            st_a = Ab + (size_t)a_kq * 128;
        } else {
            st_a += a_tap;                                           // (the per-K-tile path: one add)
        }
    };
    auto stageA = [&](int h) {
#pragma unroll
        for (int p = 0; p < 2; ++p) dma16s(st_a, voffAp[p][h], a_lds + h * AHB + p * 8192);
    };
    // ---- slice cursor (one slice ahead of the compute cursor) ----------------------------------------------------------------
    long xtile = ctile;
    int x_kq = 0, x_left = my_tiles * nslices;                      // slices not staged yet
    unsigned x_lds = lds0 + SR_ARING + wv * 1024;
    const h16_t* x_src = a.X;
    int x_ld = a.ldx, x_ko = 0;
    long x_n0 = (ctile / ny) * BN - halo;                            // flat pixel of slice row 0
    const int xch = ((lane & 7) ^ l8) * 8;
    const h16_t* zrow = g_zero_page_big + (lane & 7) * 8;
    auto x_source = [&]() {
        const int kc = x_kq * HK;
        const bool second = a.X2 != nullptr && kc >= a.ksplit;
        x_src = second ? a.X2 : a.X;
        x_ld = second ? a.ldx2 : a.ldx;
        x_ko = second ? kc - a.ksplit : kc;
    };
    auto x_advance = [&]() {                                         // after the slice's last piece has been issued
        --x_left;
        x_lds = x_lds == lds0 + SR_ARING + wv * 1024 ? x_lds + SR_XB : x_lds - SR_XB;        // (SR_XB shares bits with the wave offset: no XOR here)
        if (++x_kq == nslices) {
            x_kq = 0;
            xtile += G8;
            x_n0 = (xtile / ny) * BN - halo;
        }
        x_source();
    };
    auto stageX = [&](int i) {                                       // piece wv + 8 i of the cursor slice (rows 8 z .. 8 z + 7)
        const int z = wv + 8 * i;
        if (z < xpieces) {
            const unsigned n = (unsigned)((int)x_n0 + z * 8 + l8);           // (N * ldx < 2^31, launcher; a row before the tensor wraps to a huge value)
            unsigned src_px = n;
            if (a.up2)                                                   // the pixel of the half-resolution map under upsampled pixel n (power-of-two maps: launcher)
                src_px = ((n >> up_hwsh) << (up_hwsh - 2)) + ((((n >> up_wsh) & (unsigned)(a.H - 1)) >> 1) << (up_wsh - 1)) + ((n & (unsigned)(W - 1)) >> 1);
            const h16_t* src = n < (unsigned)a.N ? x_src + src_px * (unsigned)x_ld + x_ko + xch : zrow;
            dma16v(src, x_lds + i * 8192);
        }
    };
    // ---- the compute tile's pixels: slice row of tap (0, 0) and the nine validity bits, per accumulator column of the lane ----------
    unsigned pm[4];                                                  // [j * 2 + nt]: row | bits << 16
    auto tile_pixels = [&](long id) {
        const long n0 = (id / ny) * BN;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = wc * 64 + c * 16 + r16;
            const long n = n0 + p;
            unsigned bits = 0;
            if (n < a.N) {
                const unsigned pix = (unsigned)n % (unsigned)hw, y = pix / (unsigned)W, x = pix - y * (unsigned)W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = (int)y + t / 3 - 1, xx = (int)x + t % 3 - 1;
                    if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)W) bits |= 1u << t;
                }
            }
            pm[c] = (unsigned)(p + halo) | (bits << 16);
        }
    };
    tile_pixels(ctile);
    x_source();
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (r16 >> 1) & 7;
    const int fk[2] = {r16 * 128 + ((q ^ sw) * 16), r16 * 128 + (((4 + q) ^ sw) * 16)};
    const int aoff = wr * 8192;
    f16x8 AF[4][2], B0[2][2], B1[2][2];
    auto ldA = [&](const char* buf, int i) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) AF[mt][kh] = *(const f16x8*)(buf + i * AHB + aoff + mt * 2048 + fk[kh]);
    };
    auto ldB = [&](f16x8 (&U)[2][2], const char* xbuf, int j, int tap, int toff) {       // toff = dy * W + dx
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const unsigned m = pm[j * 2 + nt];
            const int row = ((m >> (16 + tap)) & 1u) ? (int)(m & 0xffffu) + toff : SR_XROWS;
            const int base = row * 128, s7 = row & 7;
            U[nt][0] = *(const f16x8*)(xbuf + base + ((q ^ s7) << 4));
            U[nt][1] = *(const f16x8*)(xbuf + base + (((4 + q) ^ s7) << 4));
        }
    };
#define PP_BARRIER()                                 \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
#define PP_MFMA(BU, i, j)                                                                                                    \
    _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                                                      \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)                                                                  \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                                              \
                acc[(i) * 4 + mt_][(j) * 2 + nt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                  \
                    AF[mt_][kh_], BU[nt_][kh_], acc[(i) * 4 + mt_][(j) * 2 + nt_], 0, 0, 0);
#define PP_QUAD2(BU, i, j, BV, i2, j2)                                                                                       \
    do {                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        PP_MFMA(BU, i, j)                                                                                                    \
        PP_MFMA(BV, i2, j2)                                                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
    // prologue: slice 0 and the weights of K-tile 0 staged whole, A0 of K-tile 1 behind them; everything but that A0 landed
#pragma unroll
    for (int i = 0; i < 6; ++i) stageX(i);
    x_advance();
    stageA(0); stageA(1);
    a_advance();
    if (a_left > 0) { stageA(0); GH_WAIT_BARRIER(2); } else { GH_WAIT_BARRIER(0); }
    if (grp == 1) PP_BARRIER();                                      // the second group runs half a phase behind
    const int total = my_tiles * nsteps;
    int s = 0, tap = 0, tcol = 0, toff = -W - 1;                     // K-tile inside the compute tile, its tap (and tap % 3) and the tap's row offset
    const char* xbuf = smem + SR_ARING;
    for (int g = 0; g < total; ++g) {
        const char* cur = smem + (g & 1) * ABUF;
        // phase A: quadrants (0, 0) and (0, 1); the weight cursor's A1 goes out (its A0 went a phase earlier), the cursor advances
        ldB(B0, xbuf, 0, tap, toff);
        ldB(B1, xbuf, 1, tap, toff);
        __builtin_amdgcn_sched_barrier(0);
        ldA(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (a_left > 0) { stageA(1); a_advance(); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                          // the reads retire AHEAD of the barrier: what they read may be restaged a phase later
        PP_BARRIER();
        PP_QUAD2(B0, 0, 0, B1, 0, 1);
        PP_BARRIER();
        // phase B: quadrants (1, 1) and (1, 0); A0 of the new weight cursor, one piece of the next slice; the next K-tile's weights have landed
        ldA(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool xp = tap < 6 && x_left > 0 && wv + 8 * tap < xpieces;
        if (a_left > 0) {
            stageA(0);
            if (xp) stageX(tap);
            __builtin_amdgcn_sched_barrier(0);
            if (xp) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        PP_BARRIER();
        PP_QUAD2(B1, 1, 1, B0, 1, 0);
        PP_BARRIER();
        if (++tap == 9) {                                            // the slice is complete: the next one (staged during taps 0-5) becomes current
            tap = 0; tcol = 0;
            toff = -W - 1;
            if (x_left > 0) x_advance();
            xbuf = smem + SR_ARING + ((xbuf - smem - SR_ARING) ^ SR_XB);
        } else if (++tcol == 3) {
            tcol = 0;
            toff += W - 2;
        } else {
            ++toff;
        }
        if (s != nsteps - 1) { ++s; continue; }
        // ---- the tile is complete: both groups run the epilogue at the same time -------------------------------------------------
        if (grp == 0) PP_BARRIER();
        pers_epilogue<BM, WIDE>(a, acc, ctile, ny, wr, wc, q, r16);
        ctile += G8;
        s = 0;
        if (g + 1 < total) tile_pixels(ctile);
        if (grp == 1 && g + 1 < total) PP_BARRIER();
    }
#undef PP_QUAD2
#undef PP_MFMA
#undef PP_BARRIER
}

int gemm_h16_configure() {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_h16_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_big_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_256);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_big_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_128);
    if (e != hipSuccess) return (int)e;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
        g_h16_cus = prop.multiProcessorCount & ~7;
    e = hipFuncSetAttribute((const void*)gemm_h16_sr_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SR_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_sr_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SR_LDS);
    if (e != hipSuccess) return (int)e;

    e = hipFuncSetAttribute((const void*)gemm_h16_pp_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_256);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_pp_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_256);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_pp_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_128);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_h16_pp_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_128);
    if (e != hipSuccess) return (int)e;
    return 0;
}

int gemm_h16_take_bad_shapes() { const int n = g_bad; g_bad = 0; return n; }

namespace {
// does this block go to the slice-resident form?  (the routing of launch_gemm_h16, in one place)
bool routes_to_sr(const GemmH16Args& a) {
    static const bool pers_on = []() { const char* v = getenv("DMAD_H16_PERS"); return !(v && v[0] == '0'); }();
    static const bool big_on = []() { const char* v = getenv("DMAD_H16_BIG"); return !(v && v[0] == '0'); }();
    static const bool sr_on = []() { const char* v = getenv("DMAD_H16_SR"); return !(v && v[0] == '0'); }();
    const bool two = a.X2 != nullptr, st16 = a.stats && a.stats_px == 16;
    if (!(pers_on && big_on && sr_on) || a.groups > 1 || st16 || a.K > BIG_KMAX || a.H >= 32768 || a.W >= 32768) return false;
    if (a.N * (long)a.ldx >= (1l << 31) || (two && a.N * (long)a.ldx2 >= (1l << 31)) || a.N * (long)a.ldc >= (1l << 31)) return false;
    if (a.M % 256 || a.taps != 9 || a.stride > 1 || a.W > 32 || a.W < 1) return false;
    const long nxp = (a.N + 255) / 256, tiles = nxp * (a.M / 256);
    return tiles >= g_h16_cus && nxp < (1l << 31);
}
}  // namespace

bool gemm_h16_fuses_up2(const GemmH16Args& a) {
    const auto pow2 = [](long v) { return v > 0 && (v & (v - 1)) == 0; };
    static const bool up2_on = []() { const char* v = getenv("DMAD_H16_UP2"); return !(v && v[0] == '0'); }();      // A/B switch
    return up2_on && !a.X2 && a.H >= 2 && a.W >= 2 && pow2(a.W) && pow2((long)a.H * a.W) && routes_to_sr(a);
}

int launch_gemm_h16(const GemmH16Args& a, hipStream_t s) {
    const bool two = a.X2 != nullptr;
    const int ng = a.groups > 1 ? a.groups : 1;
    if (a.up2 && !gemm_h16_fuses_up2(a)) { ++g_bad; return -1; }        // only the slice-resident form reads through the upsampling
    if (ng > 1 && (two || (long)ng * a.K > a.ldx || (long)ng * a.M > a.ldc || ng > 65535)) { ++g_bad; return -1; }
    if ((a.taps != 9 && a.taps != 1) || (a.K % HK) || a.K < HK || (a.M % 128) || (a.ldc & 3) || a.N < 1 || (!a.C && !a.C16) ||
        (a.res && a.res16) || (a.ldx & 7) || (two && ((a.ksplit % HK) || a.ksplit <= 0 || a.ksplit >= a.K || (a.ldx2 & 7)))) {
        ++g_bad;
        return -1;
    }
    // Routing.  Dense convs whose tiles fill the chip at least once: the persistent forms (one workgroup per CU walks over tiles) —
    // slice-resident for 3 x 3 / stride 1 / 256-row blocks, ping-pong otherwise (also the two-part input).  Grouped convs with >= 256
    // tiles: the 256 x 256 / 128 x 512 tile as separate workgroups.  Everything else (small maps, 16-pixel statistics blocks): the
    // 384-row kernel.  DMAD_H16_PERS=0 / DMAD_H16_SR=0 / DMAD_H16_BIG=0 switch a form off for A/B runs.
    static const bool big_on = []() { const char* v = getenv("DMAD_H16_BIG"); return !(v && v[0] == '0'); }();
    static const bool pers_on = []() { const char* v = getenv("DMAD_H16_PERS"); return !(v && v[0] == '0'); }();
    const bool st16 = a.stats && a.stats_px == 16;           // 16-pixel statistics blocks: the 384-row kernel only
    if (a.stats && a.stats_px != 0 && a.stats_px != 16 && a.stats_px != 64) { ++g_bad; return -1; }
    const bool wide = a.C16 && !a.C && !a.res && !(a.ldc & 7);
    if (routes_to_sr(a)) {                                    // (the one predicate gemm_h16_fuses_up2 asks, too)
        const int nxp = (int)((a.N + 255) / 256);
        if (wide) hipLaunchKernelGGL((gemm_h16_sr_kernel<true>), dim3(g_h16_cus), dim3(512), SR_LDS, s, a, nxp);
        else hipLaunchKernelGGL((gemm_h16_sr_kernel<false>), dim3(g_h16_cus), dim3(512), SR_LDS, s, a, nxp);
        return 0;
    }
    if (pers_on && big_on && ng == 1 && (!two || a.N * (long)a.ldx2 < (1l << 31)) && !st16 && a.K <= BIG_KMAX && a.H < 32768 && a.W < 32768 &&
        a.N * (long)a.ldx < (1l << 31)) {
        const int bm = a.M % 256 == 0 ? 256 : (a.M == 128 ? 128 : 0);
        if (bm) {
            const long nxp = (a.N + (bm == 256 ? 255 : 511)) / (bm == 256 ? 256 : 512), tiles = nxp * (a.M / bm);
            if (tiles >= g_h16_cus && nxp < (1l << 31) && a.N * (long)a.ldc < (1l << 31)) {
                if (bm == 256 && wide) hipLaunchKernelGGL((gemm_h16_pp_kernel<256, true>), dim3(g_h16_cus), dim3(512), BIG_LDS_256, s, a, (int)nxp);
                else if (bm == 256) hipLaunchKernelGGL((gemm_h16_pp_kernel<256, false>), dim3(g_h16_cus), dim3(512), BIG_LDS_256, s, a, (int)nxp);
                else if (wide) hipLaunchKernelGGL((gemm_h16_pp_kernel<128, true>), dim3(g_h16_cus), dim3(512), BIG_LDS_128, s, a, (int)nxp);
                else hipLaunchKernelGGL((gemm_h16_pp_kernel<128, false>), dim3(g_h16_cus), dim3(512), BIG_LDS_128, s, a, (int)nxp);
                return 0;
            }
        }
    }
    if (big_on && !two && !st16 && a.K <= BIG_KMAX && a.H < 32768 && a.W < 32768 && a.N * (long)a.ldx < (1l << 31)) {
        if (a.M % 256 == 0 && ((a.N + 255) / 256) * (a.M / 256) * ng >= 256) {
            const long nx = (a.N + 255) / 256;
            hipLaunchKernelGGL(gemm_h16_big_kernel<256>, dim3((unsigned)(((nx + 7) / 8) * 8 * (a.M / 256)), ng), dim3(512), BIG_LDS_256, s, a);
            return 0;
        }
        if (a.M == 128 && (a.N + 511) / 512 * ng >= 256) {
            const long nx = (a.N + 511) / 512;
            hipLaunchKernelGGL(gemm_h16_big_kernel<128>, dim3((unsigned)(((nx + 7) / 8) * 8), ng), dim3(512), BIG_LDS_128, s, a);
            return 0;
        }
    }
    if (a.M % 256 == 0) {
        const dim3 grid((unsigned)((a.N + 127) / 128), (unsigned)(a.M / 256), ng);
        if (two) hipLaunchKernelGGL((gemm_h16_kernel<256, true>), grid, dim3(512), H16_LDS, s, a);
        else hipLaunchKernelGGL((gemm_h16_kernel<256, false>), grid, dim3(512), H16_LDS, s, a);
    } else {
        const dim3 grid((unsigned)((a.N + 255) / 256), (unsigned)(a.M / 128), ng);
        if (two) hipLaunchKernelGGL((gemm_h16_kernel<128, true>), grid, dim3(512), H16_LDS, s, a);
        else hipLaunchKernelGGL((gemm_h16_kernel<128, false>), grid, dim3(512), H16_LDS, s, a);
    }
    return 0;
}

}  // namespace dmad
