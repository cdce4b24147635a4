// wn_layer_p<T> — the residual-layer kernel of the 16-bit MFMA path (persistent form), T = __bf16 or _Float16.
//
// Math, operand orientation and HBM/LDS layouts are those documented at the top of wn_bf16.hip
// (Residual_block.forward, DiffWave_Unconditional/WaveNet.py:75-97).
//
// One workgroup (8 waves, 1 per CU) walks over time tiles; each XCD's workgroups share one contiguous tile range.
// Per tile (128 time samples of one clip, all 512 gate rows):
//   GEMM1   24 k-steps of 32 through a 3-slot LDS ring (slot = 32 KiB weights + 8 KiB activations), k-step
//           ks = 3 * kchunk + tap: the three taps of one 32-channel chunk follow each other, so the rows
//           they share (or that a neighbouring tile of the same XCD reads as its other tap) are still in L2;
//           the global_load_lds pieces of k-step ks+3 and the fragment ds_reads of k-step ks+1 are
//           spread between the 32 MFMAs of k-step ks (the CU's vector-memory path moves 64 B/clk, so a
//           40 KiB stage occupies it for ~640 of the k-step's 1024 matrix cycles);
//           counted vmcnt, one barrier per k-step, nothing drained inside the loop.
//   gate    tanh*sigmoid in fp32 registers -> bf16 tile [128 t][256 ch] in LDS (over ring slot C).
//   GEMM2   res conv, 8 weight stages through five 16 KiB buffers.  Gate channels are interleaved over the
//           waves so that GEMM2 k-steps 2mt, 2mt+1 only need the gate tiles `mt`: their MFMAs run under the
//           VALU-bound gate math of tiles mt+1 (the matrix pipe would otherwise idle through the gate).
//   epi     h' = h * sqrt(1/2) + acc2, acc2 = W_res*sqrt(1/2) g + (b_res*sqrt(1/2) + emb_{n+1}) (its start value), from the accumulators: the residual stream is kept in the
//           blocked "H16" order (dmad_common.h) so that lane pairs (v_permlane16_swap) assemble 16-byte chunks and a
//           wave-store writes 256-byte contiguous runs; no LDS round trip, no barrier.
//   While the epilogue runs, stages 0-2 of the NEXT tile are already in flight into the ring, so the next
//   tile starts without an exposed load latency.
//
// LDS map (160 KiB): ring slot A = [80K,120K), B = [120K,160K), C = [0,40K); gate tile [0,64K);
// GEMM2 weight buffers 5 x 16 KiB at [80K,160K); stage-0 activation slice 8 KiB at [68K,76K); pre-scaled dilated-conv
// bias, scaled like its weight rows (2 KiB) at [65K, 67K) — written once per kernel, never overwritten: the accumulators
// of every tile start from it, so the gate needs no affine step.
// The DMA pieces are issued from inline asm (saddr form) and waited for with explicit counted s_waitcnt;
// hipcc does not count them.  The few ordinary loads of the tile loop (bias, residual rows, epilogue
// constant) are therefore issued only where NO DMA is in flight (right after a vmcnt(0) barrier) and are
// retired by an explicit builtin wait before the next DMA-heavy phase, so that hipcc never inserts a
// (too small) vmcnt wait of its own that would drain the prefetches.
#include "dmad_common.h"
#include "wn_bf16.h"

namespace dmad {

namespace {

constexpr int SLOT_BOFF = 32768;
constexpr int GEMM2_BUF = 81920;
constexpr int CONST_OFF = 66560;
constexpr int EC_OFF = 65536;                 // 1 KiB: epilogue constant b_res*sqrt(1/2) + emb_{n+1} per output channel
constexpr int B0_OFF = 69632;                 // 8 KiB: activation part of a tile's stage 0 (issued a whole phase early)
constexpr float kGateKt = -2.8853900817779268f, kGateKs = -1.4426950408889634f;   // -2*log2(e), -log2(e)
__device__ __forceinline__ constexpr int slot_base(int i) { return i == 0 ? 81920 : (i == 1 ? 122880 : 0); }

// s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier — through the builtins so that hipcc's wait-count bookkeeping
// sees them (an asm wait is invisible to it and it would re-wait lgkmcnt(0) AFTER the next k-step's
// fragment reads have been issued).  vmcnt is 6 bits: [3:0] and [15:14].
#define WNL_WAIT_BARRIER(N)                                                         \
    do {                                                                            \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                       \
        __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14));       \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    } while (0)
#define WNL_BARRIER_LGKM()                                                          \
    do {                                                                            \
        asm volatile("" ::: "memory");                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                         \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    } while (0)

// LDS-DMA in its saddr form, written as asm because hipcc otherwise materialises a 64-bit VGPR address per
// piece (and hoists ~150 of them out of the tile loop).  sbase: wave-uniform 64-bit base (SGPR pair), voff:
// 32-bit lane offset, lds: wave-uniform LDS byte address (the hardware adds lane*16).  M0 is written in the
// same statement that uses it.  hipcc does not count these loads: every wait on them is an explicit
// s_waitcnt in WNL_WAIT_BARRIER below.
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}

// g = tanh(A) * sigmoid(B) for two values at a time with packed fp32 math, on accumulators that already hold the
// exp2 arguments: the host scales the tanh rows of the dilated-conv weights by -2*log2(e) and the sigmoid rows by
// -log2(e), and the accumulators start from the equally scaled bias, so
//   u = 2^min(at, 30) = e^{-2A},  v = 2^as = e^{-B},  g = (1-u) / ((1+u)(1+v))
// 2 v_exp + 1 v_rcp per value and no affine step.  Only the tanh side needs the clamp: v = inf gives
// (1+u)(1+v) = inf -> rcp = 0 -> g = 0, the correct limit.
// WNL_VARIANT: development builds of tools/layer_variants.sh (round 5, profiles/r05_layer_gate_variants.md); the product is variant 0.
//   1  the three FMA-class steps as single v_add_f32 / v_fma_f32 (scalar code; build the file with -fno-slp-vectorize): measured equal
//      to the packed form (3.327 vs 3.321 ms per launch of 256 clips), so the product keeps the form its error statistics were taken on
//   2  ABLATION (numerically meaningless): no transcendentals — what the gate's VALU work costs at all
//   3  GEMM2's MFMAs interleaved with the gate's VALU by sched_group_barrier instead of 8-MFMA blocks: equal
// (Never hand-write these steps as asm statements: a consumer of a v_exp / v_rcp result needs a wait state, and hipcc does not look
// inside an asm statement — such a build produced NaNs, and because NaN operands do not toggle it ran 15 % FASTER on the power-capped
// board, which looked like a win until the tests ran.)
typedef __attribute__((ext_vector_type(2))) float f32x2;
#ifndef WNL_VARIANT
#define WNL_VARIANT 0
#endif
__device__ __forceinline__ f32x2 gate2(f32x2 at, f32x2 as) {
#if WNL_VARIANT == 2
    return at * as + at;
#elif WNL_VARIANT == 1
    f32x2 g;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = __builtin_amdgcn_fmed3f(at[i], 30.f, -3.0e38f);
        const float u = fast_exp2(a), v = fast_exp2(as[i]);
        const float p = 1.f + u;
        const float d = __builtin_fmaf(p, v, p);
        const float r = fast_rcp(d);
        g[i] = __builtin_fmaf(-u, r, r);
    }
    return g;
#else
    at[0] = __builtin_amdgcn_fmed3f(at[0], 30.f, -3.0e38f);     // min(at, 30) as one v_med3 (fminf on an MFMA result costs
    at[1] = __builtin_amdgcn_fmed3f(at[1], 30.f, -3.0e38f);     // an extra canonicalising v_max)
    const f32x2 u = {fast_exp2(at[0]), fast_exp2(at[1])};
    const f32x2 v = {fast_exp2(as[0]), fast_exp2(as[1])};
    const f32x2 p = u + f32x2{1.f, 1.f};
    const f32x2 d = p * v + p;
    const f32x2 r = {fast_rcp(d[0]), fast_rcp(d[1])};
    return r - u * r;                     // (1 - u) * r as one packed FMA
#endif
}

}  // namespace

template <typename T, bool LAST, bool STAMP = false>
__global__ void __launch_bounds__(512, 2) wn_layer_p(WnLayerArgs a, int ntiles) {
    typedef typename H16<T>::v8 v8;        // 8 operands of type T (one MFMA fragment, one 16-byte chunk)
    typedef typename H16<T>::v4 v4;
    auto mfma16 = [](v8 x, v8 y, f32x4 c) { return H16<T>::mfma(x, y, c); };
    // STAMP: diagnostic build only (per-phase cycle sums of wave 0 into a.dbg[block][8]); never shipped/timed
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (k >= 0) tacc[k] += now - tprev;
            tprev = now;
        }
    };
    // slots 5 / 6 of the diagnostic build: shader cycles and 100 MHz ticks of the whole workgroup lifetime -> the clock the
    // chip held inside this kernel (MI355X_MICROARCH.md, DVFS give-back item 6)
    unsigned long long t_begin = 0, r_begin = 0;
    if constexpr (STAMP) { t_begin = __builtin_amdgcn_s_memtime(); r_begin = __builtin_amdgcn_s_memrealtime(); }
    auto flush = [&]() {
        if constexpr (STAMP) {
            tacc[5] = __builtin_amdgcn_s_memtime() - t_begin;
            tacc[6] = __builtin_amdgcn_s_memrealtime() - r_begin;
            if (threadIdx.x == 0)
                for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 8 + k] = tacc[k];
        }
    };
    stamp(-1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int tiles_per_clip = a.L / kTileT;
    // activation stage in LDS: 4 planes (16-byte k-chunks) x 128 rows x 16 B; wave w DMAs plane w>>1, rows (w&1)*64 + lane:
    // consecutive lanes read consecutive 16-byte chunks of the H16 layout (256-byte runs) and write consecutive LDS bytes,
    // and a fragment read (16 lanes = 16 consecutive rows of one plane) is bank-conflict free without a swizzle
    const unsigned brow_w = (unsigned)((wv & 1) * 64 + lane), bplane = (unsigned)(wv >> 1) * 256u;
    const int bfrag_off = q * 2048 + r16 * 16;
    const unsigned tid16 = (unsigned)tid * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;       // LDS byte address of the dynamic segment
    const char* w1b = (const char*)a.w1p;
    const char* w2b = (const char*)a.w2p;
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);

    auto tile_rows = [&](int tile, int& b, int& t0) {
        b = tile / tiles_per_clip;
        t0 = (tile - b * tiles_per_clip) * kTileT;
    };
    // a tile's activation source: clip base + first row (pad included) of the tile
    struct TileSrc { const char* clip; int row0; };
    auto tile_src = [&](int b, int t0) { return TileSrc{(const char*)(a.hin + (size_t)b * a.LP * kC), kPad + t0}; };
    // one DMA piece of GEMM1 stage ks: p < 4 -> 8 KiB of weights, p == 4 -> the 8 KiB activation slice
    auto stage1_piece = [&](const TileSrc& hc, int ks, int p) {
        const int sb = slot_base(ks % 3);
        if (p < 4) {
            dma16(w1b + ((size_t)ks * 32768 + p * 8192), tid16, lds0 + sb + wv * 1024 + p * 8192);
        } else {   // rows row0 + brow + (tap-1)*d, chunks kc*4 .. kc*4+3 of the H16 layout (16-byte gather per lane)
            const unsigned row = (unsigned)(hc.row0 + ((ks % 3) - 1) * a.dilation) + brow_w;
            dma16(hc.clip + (ks / 3) * 1024, (row >> 4) * 8192u + (row & 15u) * 16u + bplane,
                  lds0 + (ks == 0 ? B0_OFF : sb + SLOT_BOFF) + wv * 1024);
        }
    };
    auto stage1 = [&](const TileSrc& hc, int ks) {
#pragma unroll
        for (int p = 0; p < 5; ++p) stage1_piece(hc, ks, p);
    };
    auto stage2 = [&](int ks2, int buf) {
        const char* wsrc = w2b + (size_t)ks2 * 16384;
        const unsigned la = lds0 + GEMM2_BUF + buf * 16384 + wv * 1024;
        dma16(wsrc, tid16, la);
        dma16(wsrc + 8192, tid16, la + 8192);
    };
    // XCD-aware walk: workgroups are placed round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  XCD x takes
    // the contiguous tile range [x * chunk, (x+1) * chunk) and its CUs walk it side by side, so the rows a tile reads as
    // taps -d / +d are the centre rows of tiles the same XCD is processing at the same time: two of the three tap reads
    // hit that XCD's L2 instead of going to HBM.
    int tile = blockIdx.x, tile_hi = ntiles, tile_step = gridDim.x;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (ntiles + 7) >> 3, lo = (blockIdx.x & 7) * chunk;
        tile = lo + (blockIdx.x >> 3);
        tile_hi = lo + chunk < ntiles ? lo + chunk : ntiles;
        tile_step = gridDim.x >> 3;
    }
    if (tile >= tile_hi) return;
    int b, t0;
    tile_rows(tile, b, t0);
    TileSrc hin_c = tile_src(b, t0);

    // pre-scaled dilated-conv bias -> LDS (one float per thread = one per gate row), retired before any DMA
    ((float*)(smem + CONST_OFF))[tid] = a.b1[tid] * (((tid & 127) < 64) ? kGateKt : kGateKs);
    if (!LAST && tid < 256) ((float*)(smem + EC_OFF))[tid] = a.epi_c[tid];      // GEMM2's accumulators start from it
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0)
    f32x4 acc[8][4];
    stage1(hin_c, 0);
    stage1(hin_c, 1);
    stage1(hin_c, 2);
    bool first = true;

    for (;;) {
        // keep the loop-invariant DMA bases and lane offsets from being hoisted out of the tile loop (spills)
        asm volatile("" : "+s"(w1b), "+s"(w2b));
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        const int qv = (tidv & 63) >> 4, r16v = tidv & 15;     // per-tile copies of q / r16 for the post-GEMM1 address math
        // ---------------- GEMM1 ------------------------------------------------------------------
        const int next = tile + tile_step;
        const bool has_next = next < tile_hi;
        int nb = 0, nt0 = 0;
        if (has_next) tile_rows(next, nb, nt0);
        const TileSrc hin_n = tile_src(nb, nt0);
        // outstanding VMEM ops younger than stage 0: first tile / LAST: stages 1,2 (10);
        // later tiles: stages 1,2 (10) + the previous tile's 8 h' stores = 18
        if (LAST || first) { WNL_WAIT_BARRIER(10); } else { WNL_WAIT_BARRIER(18); }
        stamp(0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = *(const f32x4*)(smem + CONST_OFF + (wm * 128 + mt * 16 + q * 4) * 4);
        v8 af[2][8], bf[2][4];
        {
            const char* A = smem + slot_base(0) + wm * 8192 + frag_off;
            const char* Bt = smem + B0_OFF + wn * 1024 + bfrag_off;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[0][nt] = *(const v8*)(Bt + nt * 256);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) af[0][mt] = *(const v8*)(A + mt * 1024);
        }
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            // stage ks+1 landed; stage ks+2 (and, on a later tile's first step, 8 stores) may still fly
            if (ks <= 1) {                                 // later tiles: the 8 h' stores are younger than stages 1, 2
                if (LAST || first) { WNL_WAIT_BARRIER(5); } else { WNL_WAIT_BARRIER(13); }
            } else if (ks <= 21) { WNL_WAIT_BARRIER(5); }
            else if (ks == 22) { WNL_WAIT_BARRIER(0); }
            else { WNL_BARRIER_LGKM(); }                   // every wave holds its last fragments: ring free
            const char* Ar = smem + slot_base((ks + 1) % 3) + wm * 8192 + frag_off;
            const char* Br = smem + slot_base((ks + 1) % 3) + SLOT_BOFF + wn * 1024 + bfrag_off;
#pragma unroll
            for (int p = 0; p < 5; ++p) {                  // 5 x (1 DMA piece, 4 MFMAs)
                if (ks + 3 < 24) {
                    stage1_piece(hin_c, ks + 3, p);
                } else if (!LAST && ks == 23) {            // GEMM2 weight stages 0-4 land under the gate math
                    stage2(p, p);
                } else if (LAST && ks == 23 && p == 0 && has_next) {
                    stage1_piece(hin_n, 0, 4);             // next tile's first activation slice: hide its HBM latency
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 4 * p; i < 4 * p + 4; ++i)
                    acc[i >> 2][i & 3] = mfma16(af[cur][i >> 2], bf[cur][i & 3], acc[i >> 2][i & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < 12; ++p) {                 // 12 x (1 fragment read of k-step ks+1, 1 MFMA)
                if (ks + 1 < 24) {
                    if (p < 4) bf[nxt][p] = *(const v8*)(Br + p * 256);
                    else af[nxt][p - 4] = *(const v8*)(Ar + (p - 4) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int i = 20 + p;
                acc[i >> 2][i & 3] = mfma16(af[cur][i >> 2], bf[cur][i & 3], acc[i >> 2][i & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        stamp(1);
        // gate store, k-chunk-major [8][npos][32]: 16-B chunk c of row t -> k-chunk c>>2, 16-B sub-chunk c&3
        char* gdst = (char*)a.gout + ((size_t)b * a.L + t0) * 64;
        const size_t gkc = (size_t)a.npos * 64;
        auto store_g = [&]() {
            // thread -> (row t = 16 it + tr, chunk c): one LDS address and one global address per thread, the eight
            // iterations are immediate offsets (8 KiB in LDS, 1 KiB in the gate store) of them
            typedef __attribute__((ext_vector_type(4))) unsigned g_u32x4;
            const int tr = tidv >> 5, c = tidv & 31;
            const char* lsrc = smem + tr * 512 + ((c ^ (tr & 15)) * 16);
            char* gb = gdst + (size_t)(c >> 2) * gkc + (size_t)(tr * 64 + (c & 3) * 16);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const g_u32x4 gv4 = *(const g_u32x4*)(lsrc + it * 8192);
                // written once, read by another launch much later: non-temporal, so that the stream does not push the
                // centre rows (re-read by the epilogue) out of the XCD's L2
                __builtin_nontemporal_store(gv4, (g_u32x4*)(gb + it * 1024));
            }
        };
        // ---------------- gate: g[ch][t] -> LDS [t][ch] bf16 at [0, 64K) -------------------------------------
        // Channel ownership is interleaved over the waves: tile (wm, mt) holds gate channels mt*64 + wm*16 + [0,16),
        // so after every wave has gated its tiles `mt`, channels [64 mt, 64 mt + 64) — GEMM2 k-steps 2mt, 2mt+1 —
        // are complete and their MFMAs run under the gate math (VALU) of tiles mt+1.
        auto gate_tile = [&](int mt, int nt) {
            const f32x4 ha = acc[mt][nt], hb = acc[mt + 4][nt];
            const f32x2 g01 = gate2(f32x2{ha[0], ha[1]}, f32x2{hb[0], hb[1]});
            const f32x2 g23 = gate2(f32x2{ha[2], ha[3]}, f32x2{hb[2], hb[3]});
            const v4 gv = {(T)g01[0], (T)g01[1], (T)g23[0], (T)g23[1]};
            const int t = wn * 64 + nt * 16 + r16v;
            const int chunk = mt * 8 + wm * 2 + (qv >> 1);
            *(v4*)(smem + t * 512 + ((chunk ^ r16v) * 16) + (qv & 1) * 8) = gv;
        };

        if constexpr (LAST) {
            // the last layer's residual output is never consumed (WaveNet.py:131-135): only g leaves
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    gate_tile(mt, nt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            WNL_BARRIER_LGKM();            // gate tile complete
            store_g();
            if (!has_next) return;
            WNL_BARRIER_LGKM();            // every wave has read its part of the gate tile: slot C is free
#pragma unroll
            for (int p = 0; p < 4; ++p) stage1_piece(hin_n, 0, p);       // its activation slice is already in flight
            stage1(hin_n, 1);
            stage1(hin_n, 2);
        } else {
            {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    gate_tile(0, nt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            WNL_WAIT_BARRIER(0);           // g channels [0,64) complete, GEMM2 weight stages 0-4 landed
            stamp(2);
            // the next tile's first activation slice comes from HBM (~8k cycles): issue it three phases early into its
            // own 8 KiB LDS region so that the next tile's first wait only covers L2-hot weight pieces
            if (has_next) stage1_piece(hin_n, 0, 4);

            // ---------------- GEMM2 (res = W_res * g) pipelined against the rest of the gate ------------------
            f32x4 acc2[4][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = *(const f32x4*)(smem + EC_OFF + (wm * 64 + mt * 16 + qv * 4) * 4);
            v8 b2[4], a2[4];
            auto read2 = [&](int ks2, int buf) {
                const char* A = smem + GEMM2_BUF + buf * 16384 + wm * 4096 + r16v * 64 + ((qv ^ swz64(r16v)) * 16);
                const char* G = smem + (wn * 64 + r16v) * 512 + (((ks2 * 4 + qv) ^ r16v) * 16);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b2[nt] = *(const v8*)(G + nt * 8192);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a2[mt] = *(const v8*)(A + mt * 1024);
            };
            auto mfma2 = [&](int m0, int m1) {
#pragma unroll
                for (int mt = m0; mt < m1; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = mfma16(a2[mt], b2[nt], acc2[mt][nt]);
            };
            // one phase: gate tiles (mt, 0..3) interleaved with the 2 x 16 MFMAs of k-steps ka (buffer bufa), kb (bufb)
            auto phase = [&](int mt, int ka, int bufa, int kb, int bufb) {
#if WNL_VARIANT == 3
                // half a phase = 8 fragment reads, 2 gate tiles (~56 VALU / transcendental instructions, 2 LDS writes), 16 MFMAs: the
                // scheduler is asked for groups of (1 MFMA, 4 other vector instructions)
                auto half = [&](int k2, int buf, int nt0) {
                    read2(k2, buf);
                    gate_tile(mt, nt0);
                    mfma2(0, 2);
                    gate_tile(mt, nt0 + 1);
                    mfma2(2, 4);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);       // 4 VALU / transcendental
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                __builtin_amdgcn_sched_barrier(0);
                half(ka, bufa, 0);
                half(kb, bufb, 2);
#else
                read2(ka, bufa);
                __builtin_amdgcn_sched_barrier(0);
                gate_tile(mt, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma2(0, 2);
                __builtin_amdgcn_sched_barrier(0);
                gate_tile(mt, 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma2(2, 4);
                __builtin_amdgcn_sched_barrier(0);
                read2(kb, bufb);
                __builtin_amdgcn_sched_barrier(0);
                gate_tile(mt, 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma2(0, 2);
                __builtin_amdgcn_sched_barrier(0);
                gate_tile(mt, 3);
                __builtin_amdgcn_sched_barrier(0);
                mfma2(2, 4);
                __builtin_amdgcn_sched_barrier(0);
#endif
            };
            phase(1, 0, 0, 1, 1);
            WNL_WAIT_BARRIER(1);           // g channels [64,128) complete; buffers 0-1 free (only the early slice may fly)
            stage2(5, 0);
            stage2(6, 1);
            phase(2, 2, 2, 3, 3);
            WNL_WAIT_BARRIER(0);           // g channels [128,192) complete; stages 5-6 landed; buffers 2-3 free
            // ordinary loads, issued right behind a vmcnt(0) barrier and retired by the next one (see the header):
            // the residual rows of this lane's output channels wm*64 + mt*16 + 4q + [0,4)
            v8 hc16[4][2];             // residual h: the 16-byte chunk this lane will overwrite (tile 2p + (q&1), see epilogue)
            // H16 offset of this lane's chunk (mt, p2): one per-lane base + p2 * 16 KiB (two 16-row blocks on) + mt * 512 B
            // (two chunk columns on); row0 is a multiple of 16, so the row's block index and in-block row separate
            const unsigned h16_lane = (unsigned)((hin_c.row0 >> 4) + wn * 4 + (qv & 1)) * 8192u + (unsigned)(wm * 8 + (qv >> 1)) * 256u + (unsigned)r16v * 16u;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const char* hp = hin_c.clip + h16_lane + p2 * 16384;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) hc16[mt][p2] = *(const v8*)(hp + mt * 512);
            }
            stage2(7, 2);
            phase(3, 4, 4, 5, 0);
            WNL_WAIT_BARRIER(0);           // gate tile complete; stage 7 and the ordinary loads landed
            store_g();                     // stream the gate tile to HBM
            read2(6, 1);
            mfma2(0, 4);
            read2(7, 2);
            mfma2(0, 4);

            // ---------------- next tile's stages, then the epilogue straight from the accumulators ---------------
            // h' = (h + res) * sqrt(1/2) + c.  A lane holds 4 channels x 1 sample per accumulator tile (8 bytes); one
            // v_permlane16_swap per dword pairs the lanes q / q^1 so that every lane ends up with a full 16-byte
            // chunk (even q: tile 2p, odd q: tile 2p+1) and a wave-store writes 4 x 256 B contiguous runs of the H16
            // layout.  The residual h was fetched with the same pattern and un-swapped the same way.  No LDS
            // round trip and no barrier besides the one that frees the LDS for the next tile's stages.
            stamp(3);
            WNL_BARRIER_LGKM();            // every wave is done with the gate tile and the weight buffers
            stamp(4);
            if (has_next) {
#pragma unroll
                for (int p = 0; p < 4; ++p) stage1_piece(hin_n, 0, p);   // its activation slice is already in flight
                stage1(hin_n, 1);
                stage1(hin_n, 2);
            }
            char* hout_clip = (char*)(a.hout + (size_t)b * a.LP * kC);
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    // un-swap the residual: X = this lane's 4 channels of tile 2p, Y = of tile 2p+1
                    const u32x4 hc4 = __builtin_bit_cast(u32x4, hc16[mt][p2]);
                    const auto h0 = __builtin_amdgcn_permlane16_swap(hc4[0], hc4[2], false, false);
                    const auto h1 = __builtin_amdgcn_permlane16_swap(hc4[1], hc4[3], false, false);
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    const v4 hx = __builtin_bit_cast(v4, u32x2{h0[0], h1[0]});
                    const v4 hy = __builtin_bit_cast(v4, u32x2{h0[1], h1[1]});
                    v4 ox, oy;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ox[r] = (T)__builtin_fmaf((float)hx[r], 0.70710678118654752440f, acc2[mt][2 * p2][r]);
                        oy[r] = (T)__builtin_fmaf((float)hy[r], 0.70710678118654752440f, acc2[mt][2 * p2 + 1][r]);
                    }
                    const u32x2 oxu = __builtin_bit_cast(u32x2, ox), oyu = __builtin_bit_cast(u32x2, oy);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(oxu[0], oyu[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(oxu[1], oyu[1], false, false);
                    const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};
                    __builtin_nontemporal_store(chunk, (u32x4*)(hout_clip + h16_lane + p2 * 16384 + mt * 512));
                }
            stamp(7);
            if (!has_next) { flush(); return; }
        }
        hin_c = hin_n;
        tile = next;
        b = nb;
        t0 = nt0;
        first = false;
    }
}

static int g_num_cus = 256;

template <typename T>
static void launch_layer_t(const WnLayerArgs& a, int grid, int ntiles, hipStream_t s, bool stamps) {
    if (stamps && !a.last) {
        hipLaunchKernelGGL((wn_layer_p<T, false, true>), dim3(grid), dim3(512), kWnLdsBytes, s, a, ntiles);
        return;
    }
    if (a.last) hipLaunchKernelGGL((wn_layer_p<T, true>), dim3(grid), dim3(512), kWnLdsBytes, s, a, ntiles);
    else hipLaunchKernelGGL((wn_layer_p<T, false>), dim3(grid), dim3(512), kWnLdsBytes, s, a, ntiles);
}

void launch_wn_layer_bf16_p(const WnLayerArgs& a, int B, bool f16, hipStream_t s, bool stamps) {
    const int ntiles = B * (a.L / kTileT);
    const int grid = ntiles < g_num_cus ? ntiles : g_num_cus;
    if (f16) launch_layer_t<_Float16>(a, grid, ntiles, s, stamps);
    else launch_layer_t<__bf16>(a, grid, ntiles, s, stamps);
}

template <typename T>
static int configure_layer_t() {
    hipError_t e = hipFuncSetAttribute((const void*)wn_layer_p<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_layer_p<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)wn_layer_p<T, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    return (int)e;
}

int wn_layer_p_configure() {
    if (int r = configure_layer_t<__bf16>()) return r;
    if (int r = configure_layer_t<_Float16>()) return r;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        g_num_cus = prop.multiProcessorCount;
    return 0;
}

}  // namespace dmad
