// Micro-benchmark (development only): per-CU fill rate from an L2-resident buffer that every CU re-reads
// (the WaveNet layer's weight stream), by LDS-DMA vs by global_load_dwordx4 into registers, with and
// without MFMA work beside it.  Build: hipcc -O3 --offload-arch=gfx950 l2fill.hip -o l2fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kStage = 40960;            // bytes per K-step stage per CU
constexpr int kStages = 24;              // stages per pass over the buffer (960 KiB)
constexpr int kPieces = 5;               // 1 KiB pieces per wave per stage (8 waves)

template <int MODE, int NMFMA>
__global__ void __launch_bounds__(512) fill(const char* __restrict__ src, unsigned* __restrict__ out, int passes,
                                             long long* __restrict__ cyc) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8_t fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + lane); fb[i] = (short)(0x3f80 + i); }
    uint4 x = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < passes; ++p) {
        if (MODE == 0) {
#pragma unroll 1
            for (int s = 0; s < kStages; ++s) {
                const char* g = src + (long)s * kStage + wv * (kPieces * 1024);
                asm volatile("" : "+s"(g));
                const unsigned slot = (unsigned)((s % 3) * kStage + wv * (kPieces * 1024));
#pragma unroll
                for (int i = 0; i < kPieces; ++i) {
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                 :: "s"(slot + i * 1024), "v"((unsigned)(lane * 16 + i * 1024)), "s"(g) : "memory");
                }
#pragma unroll
                for (int i = 0; i < NMFMA; ++i)
                    acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 7], 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 r[3][kPieces];
            auto ld = [&](int s, int b) {
                const uint4* g = (const uint4*)(src + (long)s * kStage + wv * (kPieces * 1024) + lane * 16);
#pragma unroll
                for (int i = 0; i < kPieces; ++i) r[b][i] = g[i * 64];
            };
            auto use = [&](int b) {
#pragma unroll
                for (int i = 0; i < kPieces; ++i) { x.x ^= r[b][i].x; x.y ^= r[b][i].y; x.z ^= r[b][i].z; x.w ^= r[b][i].w; }
            };
            ld(0, 0); ld(1, 1);
#pragma unroll 1
            for (int s = 0; s < kStages; s += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (s + u + 2 < kStages) ld(s + u + 2, (u + 2) % 3);
#pragma unroll
                    for (int i = 0; i < NMFMA; ++i)
                        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 7], 0, 0, 0);
                    use(u);
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (MODE == 0) x.x = ((unsigned*)lds)[tid];
    out[blockIdx.x * 512 + tid] = x.x ^ x.y ^ x.z ^ x.w ^ __float_as_uint(sacc);
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NMFMA>
void run(const char* name, const char* src, unsigned* out, long long* cyc, int nwg) {
    const int passes = 40;
    hipFuncSetAttribute((const void*)fill<MODE, NMFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kStage);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<MODE, NMFMA>), dim3(nwg), dim3(512), 3 * kStage, 0, src, out, 2, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<MODE, NMFMA>), dim3(nwg), dim3(512), 3 * kStage, 0, src, out, passes, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nwg); hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= nwg;
    const double bytes = (double)passes * kStages * kStage;
    printf("%-28s wgs %3d: %.3f ms  %.1f GB/s per CU  %.2f TB/s chip | %.0f cyc/stage (%.1f B/clk) clk %.2f GHz | mfma/stage/wave %d -> min %d cyc\n",
           name, nwg, ms, bytes / ms / 1e6, bytes * nwg / ms / 1e9, mean / (passes * kStages), kStage / (mean / (passes * kStages)),
           mean / ms / 1e6, NMFMA, NMFMA * 16 * 2);
}

int main() {
    char* src; unsigned* out; long long* cyc;
    hipMalloc(&src, kStages * kStage); hipMemset(src, 1, kStages * kStage);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int nwg : {256, 32}) {
        run<0, 0>("lds-dma", src, out, cyc, nwg);
        run<1, 0>("regs", src, out, cyc, nwg);
        run<0, 32>("lds-dma + 32 mfma", src, out, cyc, nwg);
        run<1, 32>("regs + 32 mfma", src, out, cyc, nwg);
    }
    return 0;
}
