// Micro-benchmark (development only): sustained rate of the two bf16 MFMA shapes on random operands (power-limited
// regime): 16x16x32 (16 cycles) vs 32x32x16 (32 cycles), 8 waves per CU, register operands only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ bf16x8 rnd8(unsigned seed) {
    bf16x8 v;
    for (int i = 0; i < 8; ++i) {
        unsigned h = (seed + i) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        v[i] = (short)((h & 0x807f) | 0x3d00 | ((h >> 3) & 0x0180));
    }
    return v;
}

template <int SHAPE>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = rnd8(tid * 131 + i * 17);
    for (int i = 0; i < 4; ++i) b[i] = rnd8(tid * 257 + i * 29 + 7);
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i * 2 + kh], b[j * 2 + kh], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    out[tid] = s;
}

template <int SHAPE>
void run(float* out) {
    const int iters = 60000;                      // 60000 x 32 (or 16) MFMAs = 1 GFLOP... per wave: 60000*32*16384 flop
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 32 * 16384.0;
    printf("mfma %dx%d: %.1f ms  %.0f TFLOP/s\n", SHAPE, SHAPE, ms, flop / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) { run<16>(out); run<32>(out); }
    return 0;
}
