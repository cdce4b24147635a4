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

// "Split-f16" storage of fp32 GEMM operands (the middle tier of the exact-vote mode): a value x is kept as the pair
//   hi = f16(x),  lo = f16((x - hi) * 2^11)      (x ~= hi + lo * 2^-11 to 22 significant bits; the 2^11 keeps lo a normal f16),
// four values per 16-byte chunk as [hi0 hi1 | hi2 hi3 | lo0 lo1 | lo2 lo3] — the size and the chunk position of four floats,
// so buffers, row gathers and LDS images are those of the fp32 path.  A product is then three f16 MFMAs with fp32
// accumulation:  x*y ~= hi_x*hi_y + (hi_x*lo_y + lo_x*hi_y) * 2^-11  (the dropped lo*lo term is 2^-22 relative).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
constexpr float kSplitScale = 2048.f, kSplitInv = 1.f / 2048.f;
__host__ __device__ inline void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * kSplitScale);
}
__device__ inline u32x4_t split4(float x0, float x1, float x2, float x3) {
    _Float16 h[4], l[4];
    split1(x0, h[0], l[0]); split1(x1, h[1], l[1]); split1(x2, h[2], l[2]); split1(x3, h[3], l[3]);
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    return u32x4_t{__builtin_bit_cast(unsigned, f16x2_t{h[0], h[1]}), __builtin_bit_cast(unsigned, f16x2_t{h[2], h[3]}),
                   __builtin_bit_cast(unsigned, f16x2_t{l[0], l[1]}), __builtin_bit_cast(unsigned, f16x2_t{l[2], l[3]})};
}
__device__ inline void join4(u32x4_t c, float out[4]) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    // (scalars first: __builtin_bit_cast applied directly to a vector ELEMENT reads element 0 for every index with this clang)
    const unsigned c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
    const f16x2_t h01 = __builtin_bit_cast(f16x2_t, c0), h23 = __builtin_bit_cast(f16x2_t, c1);
    const f16x2_t l01 = __builtin_bit_cast(f16x2_t, c2), l23 = __builtin_bit_cast(f16x2_t, c3);
    out[0] = (float)h01[0] + (float)l01[0] * kSplitInv; out[1] = (float)h01[1] + (float)l01[1] * kSplitInv;
    out[2] = (float)h23[0] + (float)l23[0] * kSplitInv; out[3] = (float)h23[1] + (float)l23[1] * kSplitInv;
}

// torch.relu / torch.maximum / torch.clamp / max_pool keep a NaN a NaN; fmaxf / fminf return the other operand.  A NaN sample
// must come out as NaN logits (its vote is then the reference's: the first NaN index), so the path uses these forms.
__device__ inline float relu_nan(float t) { return t < 0.f ? 0.f : t; }
__device__ inline float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ inline float clamp_min_nan(float t, float lo) { return t < lo ? lo : t; }
__device__ inline float clamp_nan(float t, float lo, float hi) { return t < lo ? lo : (t > hi ? hi : t); }

__device__ inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ inline float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

}  // namespace dmad
