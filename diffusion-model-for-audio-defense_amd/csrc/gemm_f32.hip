// fp32 gather-GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains).
//
//   C[n][m] = act( scale[m] * sum_tap sum_k A[tap][m][k] * X[row(n, tap)][k] + shift[m] )
//
// One kernel serves every exact-fp32 GEMM-shaped op of the path:
//   * parity-mode WaveNet: dilated Conv1d as 3 row-shifted taps over the zero-padded residual
//     stream (WaveNet.py:23-34,86) and the 1x1 res/skip/final convs (WaveNet.py:66-72,160-162);
//   * mel front-end: the windowed DFT of the 32 overlapping frames of a clip (row stride = hop) and
//     the slaney filterbank product (torchaudio MelSpectrogram, certified_robustness_eval.py:85);
//   * VGG19_bn: 3x3 convs as implicit GEMM over NHWC activations with eval-mode BatchNorm folded
//     into scale/shift, and the three Linear layers (models/vgg.py:48-52,69-81).
// Tile 128(M) x 128(N) x 16(K), 4 waves (2x2), register-staged double buffering, LDS k-major with a
// 16-float pad so both the operand reads (ds_read_b32) are bank-conflict free.
#include "gemm_f32.h"

namespace dmad {

namespace {
constexpr int BM = 128, BN = 128, BK = 16, PITCH = BM + 16;

__device__ __forceinline__ const float* row_ptr(const GemmF32Args& a, long n, int tap, int kc) {
    // returns nullptr for a zero row
    if (n >= a.N) return nullptr;
    if (a.mode == 2) {
        const int st = a.stride > 1 ? a.stride : 1;
        const int Ho = (a.H - 1) / st + 1, Wo = (a.W - 1) / st + 1, hw = Ho * Wo;
        const long b = n / hw;
        const int p = (int)(n - b * hw), y = p / Wo, x = p - y * Wo;
        const int yy = y * st + (a.taps == 9 ? tap / 3 - 1 : 0), xx = x * st + (a.taps == 9 ? tap % 3 - 1 : 0);
        if ((unsigned)yy >= (unsigned)a.H || (unsigned)xx >= (unsigned)a.W) return nullptr;
        return a.X + ((b * a.H + yy) * a.W + xx) * (long)(a.ldx ? a.ldx : a.Cin) + kc;
    }
    const long b = n / a.rows_per_batch, r = n - b * a.rows_per_batch;
    return a.X + b * a.batch_stride + r * a.row_stride + (long)(tap - (a.taps >> 1)) * a.tap_stride + kc;
}
}  // namespace

__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32Args a) {
    __shared__ float As[2][BK][PITCH];
    __shared__ float Bs[2][BK][PITCH];
    if (a.groups > 1) {                      // grouped conv: this workgroup's group = blockIdx.z
        const int g = blockIdx.z;
        a.A += (size_t)g * a.taps * a.M * a.K;
        a.X += (size_t)g * a.K;
        a.C += (size_t)g * a.M;
        if (a.scale) a.scale += (size_t)g * a.M;
        if (a.shift) a.shift += (size_t)g * a.M;
        if (a.res) a.res += (size_t)g * a.M;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, q = lane >> 4, r16 = lane & 15;
    const long n0 = (long)blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const int ksteps_per_tap = a.K / BK, nks_all = a.taps * ksteps_per_tap;
    const int S = a.splits > 1 ? a.splits : 1, z = a.groups > 1 ? 0 : blockIdx.z;
    const int ks_begin = (int)((long)nks_all * z / S), nks = (int)((long)nks_all * (z + 1) / S);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[2], rb[2];
    auto gload = [&](int ks) {
        const int tap = ks / ksteps_per_tap, kc = (ks - tap * ksteps_per_tap) * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = i * 256 + tid, row = id >> 2, kg = id & 3;
            const int m = m0 + row;
            ra[i] = (m < a.M) ? *(const float4*)(a.A + ((long)tap * a.M + m) * a.K + kc + kg * 4) : float4{0, 0, 0, 0};
            const float* p = row_ptr(a, n0 + row, tap, kc + kg * 4);
            rb[i] = p ? *(const float4*)p : float4{0, 0, 0, 0};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = i * 256 + tid, row = id >> 2, kg = id & 3;
            As[buf][kg * 4 + 0][row] = ra[i].x; As[buf][kg * 4 + 1][row] = ra[i].y;
            As[buf][kg * 4 + 2][row] = ra[i].z; As[buf][kg * 4 + 3][row] = ra[i].w;
            Bs[buf][kg * 4 + 0][row] = rb[i].x; Bs[buf][kg * 4 + 1][row] = rb[i].y;
            Bs[buf][kg * 4 + 2][row] = rb[i].z; Bs[buf][kg * 4 + 3][row] = rb[i].w;
        }
    };

    gload(ks_begin);
    lstore(ks_begin & 1);
    __syncthreads();
    for (int ks = ks_begin; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) gload(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = As[buf][kk * 4 + q][wm * 64 + i * 16 + r16];
                bf[i] = Bs[buf][kk * 4 + q][wn * 64 + i * 16 + r16];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nks) lstore(buf ^ 1);
        __syncthreads();
    }

    if (S > 1) {   // split-K: raw partial sums to this split's slab; scale/shift/act happen in the reduce kernel
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + q * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long n = n0 + wn * 64 + j * 16 + r16;
                if (n >= a.N) continue;
                float* dst = a.slab + ((long)z * a.N + n) * a.ldc + m;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.M) dst[r] = acc[i][j][r];
            }
        }
        return;
    }
    const bool vec = ((a.ldc & 3) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + q * 4;
        float sc[4], sh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = (a.scale && m + r < a.M) ? a.scale[m + r] : 1.f;
            sh[r] = (a.shift && m + r < a.M) ? a.shift[m + r] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long n = n0 + wn * 64 + j * 16 + r16;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = a.scale ? acc[i][j][r] * sc[r] + sh[r] : acc[i][j][r] + sh[r];
                if (a.res && m + r < a.M) t += a.res[n * a.ldc + m + r];
                v[r] = a.relu ? fmaxf(t, 0.f) : t;
            }
            float* dst = a.C + n * a.ldc + m;
            if (vec && m + 3 < a.M) {
                *(float4*)dst = float4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.M) dst[r] = v[r];
            }
        }
    }
}

// C[n][m] = act(scale[m] * sum_z slab[z][n][m] + shift[m]), splits summed in index order (deterministic)
__global__ void gemm_f32_reduce_kernel(GemmF32Args a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N * a.M) return;
    const long n = i / a.M;
    const int m = (int)(i - n * a.M);
    float s = 0.f;
    for (int z = 0; z < a.splits; ++z) s += a.slab[((long)z * a.N + n) * a.ldc + m];
    float t = a.scale ? s * a.scale[m] + (a.shift ? a.shift[m] : 0.f) : s + (a.shift ? a.shift[m] : 0.f);
    if (a.res) t += a.res[n * a.ldc + m];
    a.C[n * a.ldc + m] = a.relu ? fmaxf(t, 0.f) : t;
}

void launch_gemm_f32(const GemmF32Args& a0, hipStream_t s, float* slab, long slab_floats, long n_ref) {
    GemmF32Args a = a0;
    const unsigned gx = (unsigned)((a.N + BN - 1) / BN), gy = (unsigned)((a.M + BM - 1) / BM);
    const int nks = a.taps * (a.K / BK);
    int S = 1;
    if (a.groups > 1) {
        a.splits = 1;
        a.slab = nullptr;
        hipLaunchKernelGGL(gemm_f32_kernel, dim3(gx, gy, (unsigned)a.groups), dim3(256), 0, s, a);
        return;
    }
    if (slab) {
        // The split count is derived from the REFERENCE row count n_ref (the engine's max batch), not from the
        // rows of this launch, so that a sample's result does not depend on the batch it was computed in.
        const long nr = n_ref > 0 ? n_ref : a.N;
        const long wgs_ref = ((nr + BN - 1) / BN) * gy;
        if (wgs_ref < 128) {                       // fewer than half a wave of workgroups: split K
            S = (int)(256 / wgs_ref);
            if (S > nks / 4) S = nks / 4;          // keep >= 4 k-steps per split
            while (S > 1 && (long)S * nr * a.ldc > slab_floats) --S;
            if (S < 2 || a.N > nr) S = 1;
        }
    }
    a.splits = S;
    a.slab = slab;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(gx, gy, S), dim3(256), 0, s, a);
    if (S > 1) {
        const long total = a.N * a.M;
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    }
}

}  // namespace dmad
