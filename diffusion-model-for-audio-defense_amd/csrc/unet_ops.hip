// See unet_ops.h.
#include "unet_ops.h"

namespace dmad {

namespace {
inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1) / b); }

__global__ void conv1ch_3x3_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ out, int Cout, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one (b, y, x, co) each
    if (i >= total) return;
    const int co = (int)(i % Cout);
    const long p = i / Cout, b = p >> 10;
    const int y = (int)((p >> 5) & 31), x = (int)(p & 31);
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if ((unsigned)yy < 32u && (unsigned)xx < 32u) s = fmaf(w[co * 9 + ky * 3 + kx], in[(b << 10) + yy * 32 + xx], s);
        }
    out[i] = s + bias[co];
}

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wv] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per (group, sample): the group's C/32 channels x HW pixels (at most 12288 floats in this network) are read
// once into LDS as float4 (C/32 is a multiple of 4), then mean, biased variance of the deviations, and the affine /
// scale-shift / SiLU pass run from LDS
constexpr int GN_MAX = 12288;
__global__ void __launch_bounds__(256) groupnorm_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ ss, int silu,
                                                             float* __restrict__ y, int HW, int C) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float buf[GN_MAX];
    const int g = blockIdx.x, cg = C >> 5, c4 = cg >> 2, n4 = c4 * HW, n = cg * HW;
    const long base = (long)blockIdx.y * HW * C + g * cg;
    float s = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = *(const float4*)(x + base + (long)(i / c4) * C + (i % c4) * 4);
        ((float4*)buf)[i] = v;
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = ((const float4*)buf)[i];
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)n + 1e-5f);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int c = g * cg + (i % c4) * 4;
        const float4 v = ((const float4*)buf)[i];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = (o[r] - mean) * rstd * gamma[c + r] + beta[c + r];
            if (ss) t = t * (1.f + ss[c + r]) + ss[C + c + r];
            if (silu) t = t / (1.f + expf(-t));
            o[r] = t;
        }
        *(float4*)(y + base + (long)(i / c4) * C + (i % c4) * 4) = float4{o[0], o[1], o[2], o[3]};
    }
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v / (1.f + expf(-v)); }
}

__global__ void upsample2x_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C4, long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of output
    if (i >= total4) return;
    const int c = (int)(i % C4);
    long p = i / C4;
    const int xo = (int)(p % (2 * W)); p /= 2 * W;
    const int yo = (int)(p % (2 * H));
    const long b = p / (2 * H);
    ((float4*)out)[i] = ((const float4*)in)[((b * H + (yo >> 1)) * W + (xo >> 1)) * C4 + c];
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int C4, long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long r = i / C4;
    const int c = (int)(i - r * C4);
    *(float4*)(dst + r * ld_dst + c * 4) = *(const float4*)(src + r * ld_src + c * 4);
}

// one thread per query; keys / values of one (sample, head) stream through LDS in chunks of 64 with an online softmax
constexpr int HD = 64, KCH = 64;
__global__ void __launch_bounds__(256) qkv_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int heads) {
    __shared__ float Ks[KCH][HD];
    __shared__ float Vs[KCH][HD];
    const int h = blockIdx.x, b = blockIdx.y, t = blockIdx.z * blockDim.x + threadIdx.x;
    const int C3 = 3 * HD * heads;
    const float* base = qkv + (long)b * T * C3 + h * 3 * HD;
    float q[HD], acc[HD];
    const bool live = t < T;
#pragma unroll
    for (int c = 0; c < HD; ++c) { q[c] = live ? base[(long)t * C3 + c] * 0.125f : 0.f; acc[c] = 0.f; }   // (q*s).(k*s), s^2 = 1/8
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += KCH) {
        __syncthreads();
        for (int i = threadIdx.x; i < KCH * HD; i += blockDim.x) {
            const int r = i / HD, c = i % HD;
            const bool ok = k0 + r < T;
            Ks[r][c] = ok ? base[(long)(k0 + r) * C3 + HD + c] : 0.f;
            Vs[r][c] = ok ? base[(long)(k0 + r) * C3 + 2 * HD + c] : 0.f;
        }
        __syncthreads();
        const int kn = (T - k0) < KCH ? (T - k0) : KCH;
        for (int r = 0; r < kn; ++r) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) d = fmaf(q[c], Ks[r][c], d);
            const float mn = fmaxf(m, d);
            const float corr = expf(m - mn), p = expf(d - mn);
            l = l * corr + p;
#pragma unroll
            for (int c = 0; c < HD; ++c) acc[c] = fmaf(p, Vs[r][c], acc[c] * corr);
            m = mn;
        }
    }
    if (live) {
        const float inv = 1.f / l;
        float* o = out + ((long)b * T + t) * (HD * heads) + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = acc[c] * inv;
    }
}

__global__ void unet_p_sample_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ z, float ca,
                                     float cb, float c1, float c2, float sig, float* __restrict__ out, float* __restrict__ x0_out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i];
    float x0 = __fsub_rn(__fmul_rn(ca, xv), __fmul_rn(cb, eps[i]));      // reference op order, no FMA contraction
    x0 = clamp_nan(x0, -1.f, 1.f);
    float r = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, xv));
    if (z) r = __fadd_rn(r, __fmul_rn(sig, z[i]));
    out[i] = r;
    if (x0_out) x0_out[i] = x0;
}
// melspec_standardize (sc09_spectrogram_dataset.py:65-72) then q_sample (gaussian_diffusion.py:188-206), reference op order:
//   x0 = 2 * (spec - lo) / (hi - lo) - 1 ;  x_t = qa * x0 + qb * z
__global__ void spec_diffuse_kernel(const float* __restrict__ spec, const float* __restrict__ z, float lo, float range, float qa, float qb,
                                    float* __restrict__ xt, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, __fsub_rn(spec[i], lo)), range), 1.f);
    xt[i] = z ? __fadd_rn(__fmul_rn(qa, x0), __fmul_rn(qb, z[i])) : x0;
}
// melspec_inv_standardize (l.74-81): spec = (x + 1) * (hi - lo) / 2 + lo
__global__ void spec_unstandardize_kernel(const float* __restrict__ x, float lo, float range, float* __restrict__ spec, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    spec[i] = __fadd_rn(__fdiv_rn(__fmul_rn(__fadd_rn(x[i], 1.f), range), 2.f), lo);
}
}  // namespace

void launch_spec_diffuse(const float* spec, const float* z, float lo, float hi, float qa, float qb, float* xt, long n, hipStream_t s) {
    hipLaunchKernelGGL(spec_diffuse_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, spec, z, lo, hi - lo, qa, qb, xt, n);
}
void launch_spec_unstandardize(const float* x, float lo, float hi, float* spec, long n, hipStream_t s) {
    hipLaunchKernelGGL(spec_unstandardize_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, lo, hi - lo, spec, n);
}

void launch_conv1ch_3x3(const float* in, const float* w, const float* bias, float* out, int B, int Cout, hipStream_t s) {
    const long total = (long)B * 1024 * Cout;
    hipLaunchKernelGGL(conv1ch_3x3_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, in, w, bias, out, Cout, total);
}
void launch_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* ss, int silu, float* y, int B, int HW,
                           int C, hipStream_t s) {
    // C is a multiple of 128 and (C / 32) * HW <= GN_MAX for every map of this network (checked by the caller's shapes)
    hipLaunchKernelGGL(groupnorm_nhwc_kernel, dim3(32, (unsigned)B), dim3(256), 0, s, x, gamma, beta, ss, silu, y, HW, C);
}
void launch_silu(const float* x, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, y, n);
}
void launch_upsample2x_nhwc(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
    const long total4 = (long)B * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3(nblk(total4, 256)), dim3(256), 0, s, in, out, H, W, C / 4, total4);
}
void launch_copy_channels(const float* src, int ld_src, float* dst, int ld_dst, int C, long rows, hipStream_t s) {
    const long total4 = rows * (C / 4);
    hipLaunchKernelGGL(copy_channels_kernel, dim3(nblk(total4, 256)), dim3(256), 0, s, src, ld_src, dst, ld_dst, C / 4, total4);
}
void launch_qkv_attention(const float* qkv, float* out, int B, int T, int heads, hipStream_t s) {
    const int threads = T >= 256 ? 256 : (T >= 64 ? 64 * ((T + 63) / 64) : 64);
    hipLaunchKernelGGL(qkv_attention_kernel, dim3((unsigned)heads, (unsigned)B, (unsigned)((T + threads - 1) / threads)), dim3(threads), 0, s,
                       qkv, out, T, heads);
}
void launch_unet_p_sample(const float* x, const float* eps, const float* z, float ca, float cb, float c1, float c2, float sig,
                          float* out, float* x0_out, long n, hipStream_t s) {
    hipLaunchKernelGGL(unet_p_sample_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, eps, z, ca, cb, c1, c2, sig, out, x0_out, n);
}

}  // namespace dmad
