// Standalone check + timing of the f16 conv GEMM (csrc/gemm_h16.hip) against a host reference.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../diffusion-model-for-audio-defense_amd/csrc gemm_h16_check.hip -o gemm_h16_check.out
// Cases: 3x3 stride 1 / stride 2, 1x1, the two-part (concatenated) input, M = 128 / 256 / 384 tiles, N tails; then the
// UNet's big layers (B = 512) timed: TFLOP/s of the f16 matrix work.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../diffusion-model-for-audio-defense_amd/csrc/gemm_h16.hip"

using namespace dmad;

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static float rnd() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

static int check(int B, int H, int cin, int cout, int taps, int stride, int c1 /* 0: one input map */, bool with_res) {
    const int st = stride > 1 ? stride : 1, Ho = (H - 1) / st + 1;
    const long Nin = (long)B * H * H, N = (long)B * Ho * Ho;
    std::vector<uint16_t> A((size_t)taps * cout * cin), X1((size_t)Nin * (c1 ? c1 : cin)), X2(c1 ? (size_t)Nin * (cin - c1) : 1);
    std::vector<float> bias(cout), res(with_res ? (size_t)N * cout : 1);
    for (auto& v : A) v = f2h(rnd() * 0.1f);
    for (auto& v : X1) v = f2h(rnd());
    for (auto& v : X2) v = f2h(rnd());
    for (auto& v : bias) v = rnd();
    for (auto& v : res) v = rnd();
    h16_t *dA, *dX1, *dX2, *dC16; float *dB, *dR, *dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dX1, X1.size() * 2); hipMalloc(&dX2, X2.size() * 2); hipMalloc(&dC16, (size_t)N * cout * 2);
    hipMalloc(&dB, cout * 4); hipMalloc(&dR, res.size() * 4); hipMalloc(&dC, (size_t)N * cout * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dX1, X1.data(), X1.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dX2, X2.data(), X2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, bias.data(), cout * 4, hipMemcpyHostToDevice);
    hipMemcpy(dR, res.data(), res.size() * 4, hipMemcpyHostToDevice);
    GemmH16Args g{};
    g.A = dA; g.X = dX1; g.C = dC; g.C16 = dC16; g.shift = dB; g.res = with_res ? dR : nullptr; g.M = cout; g.K = cin; g.taps = taps; g.ldc = cout;
    g.N = N; g.H = H; g.W = H; g.ldx = c1 ? c1 : cin; g.stride = stride;
    if (c1) { g.X2 = dX2; g.ksplit = c1; g.ldx2 = cin - c1; }
    if (launch_gemm_h16(g, 0)) { printf("refused\n"); return 1; }
    std::vector<float> C((size_t)N * cout);
    std::vector<uint16_t> C16((size_t)N * cout);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(C16.data(), dC16, C16.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0, worst16 = 0;
    for (long n = 0; n < N; n += (N > 4096 ? 37 : 1)) {
        const long b = n / (Ho * Ho); const int pix = (int)(n % (Ho * Ho)), y = pix / Ho * st, x = pix % Ho * st;
        for (int m = 0; m < cout; m += (cout > 128 ? 5 : 1)) {
            double acc = bias[m];
            for (int t = 0; t < taps; ++t) {
                const int yy = y + (taps == 9 ? t / 3 - 1 : 0), xx = x + (taps == 9 ? t % 3 - 1 : 0);
                if (yy < 0 || yy >= H || xx < 0 || xx >= H) continue;
                const long p = b * H * H + (long)yy * H + xx;
                for (int k = 0; k < cin; ++k) {
                    const float xv = (c1 && k >= c1) ? h2f(X2[p * (cin - c1) + k - c1]) : h2f(X1[p * (c1 ? c1 : cin) + k]);
                    acc += (double)h2f(A[((size_t)t * cout + m) * cin + k]) * xv;
                }
            }
            if (with_res) acc += res[n * cout + m];
            worst = fmax(worst, fabs(acc - C[n * cout + m]));
            worst16 = fmax(worst16, fabs(acc - h2f(C16[n * cout + m])));
        }
    }
    const bool ok = worst < 2e-3 && worst16 < 2e-2;
    printf("%s B=%d H=%d cin=%d cout=%d taps=%d stride=%d c1=%d res=%d: max err fp32 out %.2e, f16 twin %.2e\n", ok ? "ok  " : "FAIL", B, H, cin, cout, taps,
           stride, c1, (int)with_res, worst, worst16);
    hipFree(dA); hipFree(dX1); hipFree(dX2); hipFree(dC16); hipFree(dB); hipFree(dR); hipFree(dC);
    return ok ? 0 : 1;
}

static void timeit(int B, int H, int cin, int cout, int taps) {
    const long N = (long)B * H * H;
    h16_t *dA, *dX, *dC16; float* dC;
    hipMalloc(&dA, (size_t)taps * cout * cin * 2); hipMalloc(&dX, (size_t)N * cin * 2); hipMalloc(&dC16, (size_t)N * cout * 2); hipMalloc(&dC, (size_t)N * cout * 4);
    std::vector<uint16_t> A((size_t)taps * cout * cin), X((size_t)N * cin);
    for (auto& v : A) v = f2h(rnd() * 0.05f);
    for (auto& v : X) v = f2h(rnd());
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice);
    GemmH16Args g{};
    g.A = dA; g.X = dX; g.C = dC; g.C16 = dC16; g.M = cout; g.K = cin; g.taps = taps; g.ldc = cout; g.N = N; g.H = H; g.W = H; g.ldx = cin; g.stride = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm_h16(g, 0);
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch_gemm_h16(g, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("time B=%d %dx%d cin=%d cout=%d taps=%d: %.3f ms = %.0f TFLOP/s\n", B, H, H, cin, cout, taps, ms, 2.0 * N * cout * cin * taps / (ms * 1e-3) / 1e12);
    hipFree(dA); hipFree(dX); hipFree(dC16); hipFree(dC);
}

int main() {
    if (gemm_h16_configure()) { printf("configure failed\n"); return 2; }
    int bad = 0;
    if (getenv("ONLY_TIME")) { timeit(512, 16, 256, 256, 9); timeit(512, 32, 128, 128, 9); timeit(512, 32, 128, 128, 9); timeit(512, 32, 384, 128, 9); return 0; }
    bad += check(2, 8, 64, 128, 9, 1, 0, false);
    bad += check(3, 8, 128, 256, 9, 1, 0, true);
    bad += check(2, 16, 128, 128, 9, 2, 0, false);
    bad += check(5, 4, 256, 256, 9, 1, 0, true);
    bad += check(2, 8, 256, 384, 1, 1, 0, false);
    bad += check(2, 8, 384, 128, 1, 1, 256, false);
    bad += check(3, 16, 256, 768, 1, 1, 0, false);
    bad += check(2, 8, 512, 256, 1, 1, 256, true);
    bad += check(1, 32, 128, 128, 9, 1, 0, true);
    // the 256 x 256 tile (>= 256 workgroups): 3x3 with residual, N tail, stride 2, 1x1 with three row blocks
    bad += check(256, 16, 256, 256, 9, 1, 0, true);
    bad += check(257, 16, 256, 256, 9, 1, 0, true);
    bad += check(256, 32, 128, 256, 9, 2, 0, false);
    bad += check(128, 16, 256, 768, 1, 1, 0, false);
    // the 128 x 512 tile (M = 128, >= 256 workgroups): 3x3 with residual, N tail, stride 2, K = 384
    bad += check(128, 32, 128, 128, 9, 1, 0, true);
    bad += check(129, 32, 128, 128, 9, 1, 0, false);
    bad += check(512, 32, 128, 128, 9, 2, 0, false);
    bad += check(128, 32, 384, 128, 9, 1, 0, true);
    printf(bad ? "FAILED %d case(s)\n" : "all cases ok\n", bad);
    timeit(512, 32, 128, 128, 9);
    timeit(512, 32, 384, 128, 9);
    timeit(512, 16, 256, 256, 9);
    timeit(512, 8, 256, 256, 9);
    timeit(512, 16, 256, 768, 1);
    return bad ? 1 : 0;
}
