// Micro-benchmark (development only): issue rate of the f16 MFMA shapes on gfx950 — the legacy K=16 form (one fp32-layout
// ds_read_b128 fragment per operand) against the K=32 form — for the split-f16 ("3-product") GEMM of the exact-vote tier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int K>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f16x8 a8[4], b8[4];
    f16x4 a4[4], b4[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            a8[i][e] = (_Float16)(0.001f * ((tid * 7 + i * 13 + e) % 97));
            b8[i][e] = (_Float16)(0.002f * ((tid * 3 + i * 5 + e) % 89));
            if (e < 4) { a4[i][e] = a8[i][e]; b4[i][e] = b8[i][e]; }
        }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (K == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[i], b8[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[i], b4[j], acc[i][j], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[tid] = s;
    if (tid == 0) *cyc = t1 - t0;
}

template <int K>
void run(float* out, unsigned long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mfma f16 16x16x%d, one wave per SIMD: %.1f cycles per MFMA, %.0f TFLOP/s chip-wide (%.1f ms)\n", K, (double)c / (iters * 16.0),
           256.0 * 4 * iters * 16 * (512.0 * K) / ms / 1e9, ms);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) { run<32>(out, cyc); run<16>(out, cyc); }
    return 0;
}
