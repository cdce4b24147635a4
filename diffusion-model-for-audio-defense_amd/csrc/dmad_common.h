// Shared device/host definitions for the dmad HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmad {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef unsigned short h16_t;    // storage type of a 16-bit operand (bf16 or f16 bits, see H16)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Geometry of the DiffWave eps-network this engine is specialised for
// (reference configs/config.json:7-17; other sizes are rejected by dmad_create).
constexpr int kC = 256;          // res_channels == skip_channels
constexpr int kPad = 2048;       // zero rows on each side of a clip in the residual stream (max dilation)
constexpr int kTileT = 128;      // time positions per workgroup tile

// Residual stream layout "H16" (bf16, per clip kPad + L + kPad rows of 256 channels): 16 consecutive rows form
// an 8 KiB block stored [32 chunks of 8 channels][16 rows][8 ch], i.e. the 16-byte chunk (row, c) lives at
// (row >> 4) * 8192 + c * 256 + (row & 15) * 16.  A 16x16 MFMA accumulator tile (16 samples x 4 channels per lane
// group) then maps to 256-byte contiguous runs, so the epilogue stores straight from registers, and the LDS-DMA
// of the next layer (per-lane source address) gathers its 16-byte chunks from the same layout.
__host__ __device__ inline unsigned h16_off(unsigned row, unsigned chunk) { return (row >> 4) * 8192u + chunk * 256u + (row & 15u) * 16u; }

// 16-byte-chunk swizzle for 64-byte LDS rows read as MFMA 16x16x32 operands with ds_read_b128
// (conflict-free for the four 16-lane groups of ds_read_b128, see DESIGN.md "LDS images").
__host__ __device__ inline int swz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }  // {0,2,3,1}[(row>>2)&3]

__device__ inline void glds16(const void* gsrc, void* lds_dst_wave_base) {
    // async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc, (lds_ptr_t)lds_dst_wave_base, 16, 0, 0);
}

// The 16-bit MFMA path is written once for both 16-bit operand formats: bf16 (8-bit significand, the fp32 exponent range)
// and f16 (11-bit significand: 8x smaller rounding error at the same v_mfma_f32_16x16x32 rate; the network's activations
// are O(1), far inside the f16 range).  H16<T> = vector types + the MFMA of operand type T.
template <typename T> struct H16;
template <> struct H16<__bf16> {
    typedef bf16x8 v8; typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct H16<_Float16> {
    typedef f16x8 v8; typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

__device__ inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ inline float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

}  // namespace dmad
