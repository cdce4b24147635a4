// Ablation / stamp hooks of gemm_x3_kernel (gemm_f32.hip), development only.  The product library defines NONE of the X3_* macros:
// every hook below is then the identity and the kernel body reads as the product's.  tools/x3_ablation.sh builds the variants
// (libdmad_hip.so.<variant>) that switch parts of the main loop OFF — their results are numerically meaningless, only time and power
// are read (HISTORY.md, old section 5.4; profiles/r03g_x3_ablation.md):
//   X3_NO_DMA / X3_ONLY_A / X3_ONLY_X   steady-state LDS-DMA off / weight pieces only / activation pieces only
//   X3_NO_LDS                           no fragment reads            X3_NO_FIX   no hi / lo register exchange
//   X3_NO_BARRIER / X3_NO_VMWAIT        no barrier / barrier without the counted vmcnt wait
//   X3_STAMPS                           s_memtime per phase group, printed for the K = 9216 skip GEMM
#pragma once

// piece k of a pair is skipped in the steady state?
#if defined(X3_ONLY_A)
#define X3A_SKIP_PIECE(k, steady) ((k) >= 4 && (steady))
#elif defined(X3_ONLY_X)
#define X3A_SKIP_PIECE(k, steady) ((k) < 4 && (steady))
#else
#define X3A_SKIP_PIECE(k, steady) false
#endif

// fragment read of one tile: two 16-byte chunks at f0 / f1
#ifdef X3_NO_LDS
#define X3A_LD(H, L, tile, f0, f1) asm volatile("" : "+v"(H), "+v"(L))
#else
#define X3A_LD(H, L, tile, f0, f1) do { (H) = *(const u32x4_t*)((tile) + (f0)); (L) = *(const u32x4_t*)((tile) + (f1)); } while (0)
#endif

// the hi / lo exchange that turns two chunks [hi0-3 | lo0-3], [hi4-7 | lo4-7] into 8 hi / 8 lo
#ifdef X3_NO_FIX
#define X3A_FIX(H, L) do {} while (0)
#else
#define X3A_FIX(H, L) do { const unsigned x_ = (H)[2], y_ = (H)[3]; (H)[2] = (L)[0]; (H)[3] = (L)[1]; (L)[0] = x_; (L)[1] = y_; } while (0)
#endif

// the one barrier of a pair: counted wait (only the pair after next may fly) unless this was the last staged pair
#if defined(X3_NO_VMWAIT)
#define X3A_PAIR_BARRIER(more) do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); } while (0)
#elif defined(X3_NO_BARRIER)
#define X3A_PAIR_BARRIER(more) do {} while (0)
#else
#define X3A_PAIR_BARRIER(more) do { if (more) { GF_WAIT_BARRIER(6); } else { GF_WAIT_BARRIER(0); } } while (0)
#endif

#ifdef X3_NO_DMA
#define X3A_DO_DMA(cond) false
#else
#define X3A_DO_DMA(cond) (cond)
#endif

#ifdef X3_STAMPS
#define X3A_STAMP_DECL unsigned long long x3_t[4] = {0, 0, 0, 0}, x3_prev = __builtin_amdgcn_s_memtime()
#define X3A_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); x3_t[k] += now_ - x3_prev; x3_prev = now_; } while (0)
#define X3A_STAMP_PRINT(npairs, wv, lane)                                                                                                   \
    do {                                                                                                                                     \
        if ((npairs) == 288 && (blockIdx.x == 100 || blockIdx.x == 1501) && (lane) == 0 && ((wv) == 0 || (wv) == 4 || (wv) == 3))            \
            printf("x3 stamps block %d wave %d: loop-top %llu  ph1+2 %llu  barrier %llu  ph3+4 %llu cycles per pair\n", (int)blockIdx.x, (wv), \
                   x3_t[0] / 288, x3_t[1] / 288, x3_t[2] / 288, x3_t[3] / 288);                                                              \
    } while (0)
#else
#define X3A_STAMP_DECL do {} while (0)
#define X3A_STAMP(k) do {} while (0)
#define X3A_STAMP_PRINT(npairs, wv, lane) do {} while (0)
#endif
