// Micro-benchmark (development only): ENERGY of the layer kernel's GEMM1 k-loop under different CU tilings, random f16 data.
//
// The layer kernel runs at the board's power cap, so the quantity to compare between tilings is joules per
// unit of work, not cycles.  Every variant runs the same loop shape — 3-slot LDS ring filled by LDS-DMA (weights from an
// L2-resident image, one activation slice in three from an HBM stream, two from L2, as the kernel's taps), one barrier per
// k-step, single-buffered fragments re-read right after their last use (so that one wave per SIMD also runs pipelined) —
// on a 512-row tile of
//   v0  8 waves 4(M) x 2(N), wave tile 128 x  64, 128 samples   (the shipped tiling: 96 KiB of fragment reads per k-step)
//   v1  4 waves 4 x 1,       wave tile 128 x 128, 128 samples   (one wave per SIMD, 512 registers: 64 KiB)
//   (a 128 x 192 wave tile — weight fill per sample - 33 % — needs 464 of the 512 registers for accumulators and single-buffered
//    fragments alone: hipcc spills 172 registers, not measurable in this form)
//   v3  8 waves 8 x 1,       wave tile  64 x 128, 128 samples   (each weight fragment feeds 8 MFMAs; 32 KiB weight + 64 KiB activation reads)
//   v4  4 waves 4 x 1,       wave tile 128 x 160, 160 samples   (16000 = 100 x 160: weight fill per sample - 20 %)
// while a host thread samples the card's hwmon power.  Output per variant: ns per k-step, GEMM1-equivalent TFLOP/s, board
// power, in-kernel clock (s_memtime / s_memrealtime over the workgroups' lifetimes) and joules above idle per 128-sample tile.
//   hipcc -O3 --offload-arch=gfx950 gemm1_energy.hip -o gemm1_energy.out -lpthread;  ./gemm1_energy.out [seconds per variant] [zero]
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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

template <int WAVES, int MW, int MT, int NT>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1)
k(const char* __restrict__ w, const char* __restrict__ hl2, const char* __restrict__ hbm, float* __restrict__ out,
  unsigned long long* __restrict__ clk, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES / MW, NTILE = NW * NT * 16;
    constexpr int ABYTES = 32768, BBYTES = NTILE * 64, SLOT = ABYTES + BBYTES, NPIECE = SLOT / 1024;
    constexpr int PPW = (NPIECE + WAVES - 1) / WAVES;                 // DMA pieces per wave and stage (the last round may be partial)
    static_assert(MW * MT * 16 == 512, "512 rows");
    static_assert(3 * SLOT <= 163840, "ring fits the LDS");
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / NW, wn = wv % NW, q = lane >> 4, r16 = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int afrag = wm * MT * 1024 + r16 * 64 + ((q ^ swz64(r16)) * 16);
    const int bfrag = ABYTES + q * (NTILE * 16) + (wn * NT * 16 + r16) * 16;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one DMA piece (1 KiB) of the stage (k-chunk kc, tap) of tile `tile`: piece < 32 weights, else the activation slice;
    // stage 3 * kc + tap lives in ring slot `tap`
    auto piece = [&](int tile, int kc, int tap, int p) {
        int pc = p * WAVES + wv;
        if (pc >= NPIECE) pc = NPIECE - 1;                            // (a partial last round repeats a piece: every wave issues PPW per stage)
        const unsigned dst = lds0 + (unsigned)(tap * SLOT) + pc * 1024;
        if (pc < 32) {
            dma16(w + ((size_t)(kc * 3 + tap) * 32768 + pc * 1024), lane * 16u, dst);
        } else {
            const char* src = tap == 1 ? hbm + ((size_t)((blockIdx.x * 64 + (tile & 63)) * 8 + kc)) * BBYTES
                                       : hl2 + ((size_t)((blockIdx.x & 31) * 16 + kc * 2 + (tap >> 1))) * BBYTES;
            dma16(src + (pc - 32) * 1024, lane * 16u, dst);
        }
    };
    for (int tap = 0; tap < 3; ++tap)
        for (int p = 0; p < PPW; ++p) piece(0, 0, tap, p);
    f16x8 af[MT], bf[NT];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const f16x8*)(smem + bfrag + nt * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(const f16x8*)(smem + afrag + mt * 1024);
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
        for (int kc = 0; kc < 8; ++kc) {
            asm volatile("" : "+s"(w), "+s"(hl2), "+s"(hbm));
            const int nkc = (kc + 1) & 7, ntile = tile + (kc == 7);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                // the next stage has landed (its fragments are read during this k-step), the one after may fly; every wave has
                // read this k-step's slot completely (during the previous k-step): it is refilled with the stage 3 k-steps on
                if (PPW == 5) { WAIT_BARRIER(5); } else if (PPW == 10) { WAIT_BARRIER(10); } else if (PPW == 11) { WAIT_BARRIER(11); } else { WAIT_BARRIER(12); }
                const char* nxt = smem + ((tap + 1) % 3) * SLOT;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        const int i = mt * NT + nt;                   // the DMA pieces spread over the first MFMAs of the k-step
                        if (i % 3 == 1 && i / 3 < PPW) piece(ntile, nkc, tap, i / 3);
                        // a B fragment is dead after the last row's MFMA: re-read it from the next slot
                        if (mt == MT - 1) bf[nt] = *(const f16x8*)(nxt + bfrag + nt * 256);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    af[mt] = *(const f16x8*)(nxt + afrag + mt * 1024);          // an A fragment is dead after its row
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t_begin;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r_begin;
    }
}

__global__ void fill_rnd(unsigned* p, size_t n, int zero) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // two f16: random sign, 10 random mantissa bits, exponent 8 .. 11 (|x| in [2^-7, 2^-3))
        p[i] = zero ? 0u : ((h & 0x83ff83ffu) | 0x20002000u | ((h >> 3) & 0x0c000c00u));
    }
}

// ---- hwmon power of the card that is HIP device 0 ------------------------------------------------------------------
static std::string read_file(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return "";
    char buf[4096];
    size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    return buf;
}
static std::string find_power_file() {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return "";
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    DIR* d = opendir("/sys/class/drm");
    if (!d) return "";
    std::string found;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) || strchr(e->d_name, '-')) continue;
        const std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
        std::string ue = read_file(dev + "/uevent");
        for (auto& c : ue) c = (char)tolower(c);
        if (ue.find(std::string("pci_slot_name=") + bus) == std::string::npos) continue;
        DIR* h = opendir((dev + "/hwmon").c_str());
        if (!h) continue;
        while (dirent* he = readdir(h)) {
            if (strncmp(he->d_name, "hwmon", 5)) continue;
            for (const char* n : {"power1_average", "power1_input"}) {
                const std::string p = dev + "/hwmon/" + he->d_name + "/" + n;
                if (!read_file(p).empty()) { found = p; break; }
            }
            if (!found.empty()) break;
        }
        closedir(h);
        if (!found.empty()) break;
    }
    closedir(d);
    return found;
}
struct Sampler {
    std::string path;
    std::atomic<bool> stop{false};
    std::vector<double> w;
    std::thread th;
    void start() {
        stop = false; w.clear();
        th = std::thread([this] {
            while (!stop) {
                const std::string s = read_file(path);
                if (!s.empty()) w.push_back(atof(s.c_str()) * 1e-6);
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
    }
    double finish(double skip_frac) {       // median of the samples after the first skip_frac of the run
        stop = true; th.join();
        if (w.empty()) return 0.0;
        std::vector<double> v(w.begin() + (size_t)(w.size() * skip_frac), w.end());
        if (v.empty()) v = w;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    }
};

static Sampler g_smp;
static double g_idle = 0.0;

template <int WAVES, int MW, int MT, int NT>
void run(const char* name, const char* w, const char* hl2, const char* hbm, float* out, unsigned long long* clk, double seconds) {
    constexpr int NTILE = (WAVES / MW) * NT * 16;
    auto kern = k<WAVES, MW, MT, NT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), 163840, 0, w, hl2, hbm, out, clk, 50);
    hipDeviceSynchronize();
    // size a launch to ~0.25 s
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), 163840, 0, w, hl2, hbm, out, clk, 400);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int tiles = std::max(400, (int)(400 * 250.0 / ms));
    const int launches = std::max(2, (int)(seconds / 0.25));
    if (!g_smp.path.empty()) g_smp.start();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), 163840, 0, w, hl2, hbm, out, clk, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    const double pw = g_smp.path.empty() ? 0.0 : g_smp.finish(0.3);
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(512);
    hipMemcpy(c.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < 256; ++b) if (c[2 * b + 1]) ghz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double ns = ms * 1e6 / ((double)launches * tiles * 24);                       // per k-step
    const double tf = 256.0 * 2 * 512 * NTILE * 32 / ns / 1e3;
    const double ns128 = ns * 128.0 / NTILE;                                            // per k-step of 128 samples
    const double uj_tile = (pw - g_idle) * ns128 * 24 * 1e-3 / 256.0;                   // micro-joules above idle per CU tile of 512 x 128, K = 768
    printf("%-34s %7.1f ns/k-step (%3d samples)  %6.0f TFLOP/s  %6.0f W  clock %.3f GHz  cycles/k-step/128 %6.0f  %.2f uJ above idle per 512x128 tile\n", name, ns, NTILE,
           tf, pw, ghz.empty() ? 0.0 : ghz[ghz.size() / 2], ns128 * (ghz.empty() ? 0.0 : ghz[ghz.size() / 2]), uj_tile);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int zero = argc > 2 && !strcmp(argv[2], "zero");
    g_smp.path = find_power_file();
    printf("power file: %s   data: %s\n", g_smp.path.empty() ? "(none readable)" : g_smp.path.c_str(), zero ? "zeros" : "random f16");
    char *w, *hl2, *hbm; float* out; unsigned long long* clk;
    const size_t wb = 24 * 32768, l2b = (size_t)32 * 16 * 12288, hb = (size_t)256 * 64 * 8 * 12288;
    hipMalloc(&w, wb); hipMalloc(&hl2, l2b); hipMalloc(&hbm, hb); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 512 * 8);
    hipLaunchKernelGGL(fill_rnd, dim3(1024), dim3(256), 0, 0, (unsigned*)w, wb / 4, zero);
    hipLaunchKernelGGL(fill_rnd, dim3(1024), dim3(256), 0, 0, (unsigned*)hl2, l2b / 4, zero);
    hipLaunchKernelGGL(fill_rnd, dim3(4096), dim3(256), 0, 0, (unsigned*)hbm, hb / 4, zero);
    hipDeviceSynchronize();
    if (!g_smp.path.empty()) {
        g_smp.start();
        std::this_thread::sleep_for(std::chrono::milliseconds(2000));
        g_idle = g_smp.finish(0.2);
        printf("idle %.0f W\n", g_idle);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 4, 8, 4>("v0 8w 4x2 wave 128x64  (shipped)", w, hl2, hbm, out, clk, seconds);
        run<4, 4, 8, 8>("v1 4w 4x1 wave 128x128", w, hl2, hbm, out, clk, seconds);
        run<8, 8, 4, 8>("v3 8w 8x1 wave 64x128", w, hl2, hbm, out, clk, seconds);
        run<4, 4, 8, 10>("v4 4w 4x1 wave 128x160", w, hl2, hbm, out, clk, seconds);
    }
    return 0;
}
