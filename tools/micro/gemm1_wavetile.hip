// Micro-benchmark (development only): the GEMM1 k-loop (3-slot LDS ring by LDS-DMA, one barrier per k-step, random bf16
// data) with two wave tilings of the same 512 x 128 CU tile:
//   8 waves x 128x64  (32 accumulator tiles / wave, 12 fragment reads per k-step per wave, 2 waves per SIMD) — shipped
//   4 waves x 128x128 (64 accumulator tiles / wave, 16 fragment reads per k-step per wave, 1 wave per SIMD, 512 registers)
// to price the LDS fragment traffic (96 KiB vs 64 KiB per k-step per CU) in the power-limited regime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int SLOT = 40960, SLOT_BOFF = 32768;
__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
#define WAIT_BARRIER(N)                                                             \
    do {                                                                            \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                       \
        __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14));       \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    } while (0)

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) k(const char* __restrict__ w, const char* __restrict__ h, float* __restrict__ out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = WAVES == 8 ? 4 : 8;        // 16-column accumulator tiles per wave along N (positions)
    constexpr int PIECES = 40 / WAVES;            // 1 KiB DMA pieces per wave per 40 KiB stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WAVES == 8 ? wv >> 1 : wv, wn = WAVES == 8 ? wv & 1 : 0, q = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int frag_off = r16 * 64 + ((q ^ swz64(r16)) * 16);
    const int bfrag_off = q * 2048 + r16 * 16;
    f32x4 acc[8][NT];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int tile, int ks) {
        const unsigned sb = (unsigned)((ks % 3) * SLOT);
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int piece = wv * PIECES + p;            // 0..39: 32 weight KiB then 8 activation KiB
            if (piece < 32) dma16(w + ((size_t)(ks % 24) * 32768 + piece * 1024), lane * 16u, lds0 + sb + piece * 1024);
            else dma16(h + ((size_t)(blockIdx.x * 64 + (tile & 63)) * 24 + (ks % 24)) * 8192 + (piece - 32) * 1024, lane * 16u,
                       lds0 + sb + SLOT_BOFF + (piece - 32) * 1024);
        }
    };
    stage(0, 0); stage(0, 1); stage(0, 2);
    for (int tile = 0; tile < tiles; ++tile) {
        asm volatile("" : "+s"(w), "+s"(h));
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            if (WAVES == 8) { WAIT_BARRIER(10); } else { WAIT_BARRIER(20); }      // stage ks landed; ks+1, ks+2 may fly
            const char* A = smem + (ks % 3) * SLOT + wm * 8192 + frag_off;
            const char* Bt = smem + (ks % 3) * SLOT + SLOT_BOFF + wn * 1024 + bfrag_off;
            bf16x8 af[8], bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const bf16x8*)(Bt + nt * 256);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) af[mt] = *(const bf16x8*)(A + mt * 1024);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();                                         // every wave holds its fragments: slot free
            stage(tile + (ks + 3) / 24, ks + 3);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][3];
    out[blockIdx.x * 512 + tid] = s;
}

__global__ void fill_rnd(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0x807f807fu) | 0x3d003d00u | ((h >> 3) & 0x01800180u);
    }
}

template <int WAVES>
void run(const char* name, const char* w, const char* h, float* out) {
    const int tiles = 2000;
    hipFuncSetAttribute((const void*)k<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WAVES>), dim3(256), dim3(WAVES * 64), 163840, 0, w, h, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES>), dim3(256), dim3(WAVES * 64), 163840, 0, w, h, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (tiles * 24);
    printf("%-28s %.2f ms  %.1f ns/k-step  %.0f TFLOP/s GEMM1-equivalent\n", name, ms, ns, 256.0 * 2 * 512 * 128 * 32 / ns / 1e3);
}

int main() {
    char *w, *h; float* out;
    hipMalloc(&w, 24 * 32768);
    const size_t hb = (size_t)256 * 64 * 24 * 8192;
    hipMalloc(&h, hb);
    hipLaunchKernelGGL(fill_rnd, dim3(1024), dim3(256), 0, 0, (unsigned*)w, (size_t)24 * 32768 / 4);
    hipLaunchKernelGGL(fill_rnd, dim3(4096), dim3(256), 0, 0, (unsigned*)h, hb / 4);
    hipDeviceSynchronize();
    hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<8>("8 waves x 128x64", w, h, out);
        run<4>("4 waves x 128x128", w, h, out);
    }
    return 0;
}
