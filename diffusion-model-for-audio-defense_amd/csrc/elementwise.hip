// Small HBM-bound kernels of the path: noise + scale (Philox), step-embedding MLP, scheduler
// updates, fp32-mode gate/update, mel post-processing, VGG helpers, arg-max votes.
#include "elementwise.h"

namespace dmad {

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011).  counter = (block, sample_lo,
// sample_hi, stream), key = (seed_lo, seed_hi): sample i's noise is a pure function of
// (seed, i, stream) -> Monte Carlo votes do not depend on batch size or on the number of ranks.
// ----------------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ inline void philox_normal4(uint64_t seed, uint64_t sample, uint32_t stream, uint32_t block, float z[4]) {
    uint32_t c[4] = {block, (uint32_t)sample, (uint32_t)(sample >> 32), stream};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // Box-Muller on (0,1) uniforms with 24 random bits each
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u1 = ((float)(c[1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u3 = ((float)(c[3] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

__global__ void philox_raw_kernel(uint64_t seed, uint64_t sample, uint32_t stream, uint32_t nblocks, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    uint32_t c[4] = {i, (uint32_t)sample, (uint32_t)(sample >> 32), stream};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[4 * i + 0] = c[0]; out[4 * i + 1] = c[1]; out[4 * i + 2] = c[2]; out[4 * i + 3] = c[3];
}

// z[b][l] ~ N(0,1), sample index = sample0 + b, or idx[b] when an index list is given (the recheck passes)
__global__ void philox_normal_kernel(uint64_t seed, uint64_t sample0, uint32_t stream, float* z, int B, int L, const long long* __restrict__ idx) {
    const int per = L / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * per) return;
    const int b = (int)(i / per), blk = (int)(i - (long)b * per);
    float v[4];
    philox_normal4(seed, idx ? (uint64_t)idx[b] : sample0 + b, stream, blk, v);
    *(float4*)(z + (long)b * L + blk * 4) = float4{v[0], v[1], v[2], v[3]};
}

// certified_robust.py:46-54:  x_in = x.repeat(B) + delta ; x_in = alpha_bar_star**0.5 * x_in
// delta = host noise [B][L] (parity mode) or sigma * Philox normal (fast mode)
__global__ void mc_noise_scale_kernel(const float* __restrict__ clip, const float* __restrict__ delta, float sigma,
                                      float scale, uint64_t seed, uint64_t sample0, float* __restrict__ xt, int B, int L) {
    const int per = L / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * per) return;
    const int b = (int)(i / per), blk = (int)(i - (long)b * per);
    const float4 x = *(const float4*)(clip + blk * 4);
    float d[4];
    if (delta) {
        const float4 dv = *(const float4*)(delta + (long)b * L + blk * 4);
        d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
    } else {
        philox_normal4(seed, sample0 + b, 0u, blk, d);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __fmul_rn(sigma, d[j]);
    }
    float4 o;
    o.x = __fmul_rn(scale, __fadd_rn(x.x, d[0])); o.y = __fmul_rn(scale, __fadd_rn(x.y, d[1]));
    o.z = __fmul_rn(scale, __fadd_rn(x.z, d[2])); o.w = __fmul_rn(scale, __fadd_rn(x.w, d[3]));
    *(float4*)(xt + (long)b * L + blk * 4) = o;
}

// The same sample, addressed through an index list (exact-vote recheck): row b is global sample idx[b]; its noise is
// the Philox draw keyed (seed, idx[b]) or row idx[b] - sample0 of the caller's delta — bit-identical to what the sample
// received in its first (bf16) evaluation.
__global__ void mc_noise_scale_idx_kernel(const float* __restrict__ clip, const float* __restrict__ delta, float sigma,
                                          float scale, uint64_t seed, uint64_t sample0, const long long* __restrict__ idx,
                                          float* __restrict__ xt, int B, int L) {
    const int per = L / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * per) return;
    const int b = (int)(i / per), blk = (int)(i - (long)b * per);
    const uint64_t sample = (uint64_t)idx[b];
    const float4 x = *(const float4*)(clip + blk * 4);
    float d[4];
    if (delta) {
        const float4 dv = *(const float4*)(delta + (long)(sample - sample0) * L + blk * 4);
        d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
    } else {
        philox_normal4(seed, sample, 0u, blk, d);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __fmul_rn(sigma, d[j]);
    }
    float4 o;
    o.x = __fmul_rn(scale, __fadd_rn(x.x, d[0])); o.y = __fmul_rn(scale, __fadd_rn(x.y, d[1]));
    o.z = __fmul_rn(scale, __fadd_rn(x.z, d[2])); o.w = __fmul_rn(scale, __fadd_rn(x.w, d[3]));
    *(float4*)(xt + (long)b * L + blk * 4) = o;
}

// dst[idx[b] - base][0:W] = src[b][0:W]  (rechecked rows of logits_out / x0_out); W % 4 == 0 or W < 4 handled scalar
__global__ void scatter_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, long long base,
                                    float* __restrict__ dst, int B, int W) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * W) return;
    const int b = (int)(i / W), c = (int)(i - (long)b * W);
    dst[(idx[b] - base) * W + c] = src[i];
}

// out[i] = x[(row0 + i) % B]: rows [row0, row0 + nrows) of x.repeat(R, 1, 1)  (EOT's x_batch.repeat, _EOT.py:36)
__global__ void repeat_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int B, long row0, int L, long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int per = L / 4;
    const long row = i / per;
    const int blk = (int)(i - row * per);
    ((float4*)out)[i] = ((const float4*)x)[((row0 + row) % B) * per + blk];
}

// ----------------------------------------------------------------------------------------------
// diffusion-step embedding (util.py:68-93) -> fc_t1, fc_t2 with swish (WaveNet.py:124-126) ->
// the 36 per-layer fc_t (WaveNet.py:82-83).  t is the same for every row of the batch in every
// inference caller, so the result is a [NL][256] bias table.  One block per layer.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) embed_table_kernel(float t, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ wt, const float* __restrict__ bt,
                                                          float* __restrict__ table, float* __restrict__ emb2_out,
                                                          const float* __restrict__ b_res, float* __restrict__ epi_c) {
    __shared__ float e0[128], e1[512], e2[512];
    const int tid = threadIdx.x, n = blockIdx.x;
    if (tid < 64) {
        const float f = expf((float)tid * -0.14619587892025687f);     // -ln(10000)/63 (util.py:88-89)
        const float v = t * f;
        e0[tid] = sinf(v);
        e0[64 + tid] = cosf(v);
    }
    __syncthreads();
    {
        float s = 0.f;
        for (int k = 0; k < 128; ++k) s = fmaf(w1[tid * 128 + k], e0[k], s);
        s += b1[tid];
        e1[tid] = s / (1.f + expf(-s));
    }
    __syncthreads();
    {
        float s = 0.f;
        for (int k = 0; k < 512; ++k) s = fmaf(w2[tid * 512 + k], e1[k], s);
        s += b2[tid];
        e2[tid] = s / (1.f + expf(-s));
    }
    __syncthreads();
    if (n == 0 && emb2_out) emb2_out[tid] = e2[tid];
    if (tid < 256) {
        const float* w = wt + ((long)n * 256 + tid) * 512;
        float s = 0.f;
        for (int k = 0; k < 512; ++k) s = fmaf(w[k], e2[k], s);
        const float v = s + bt[n * 256 + tid];
        table[n * 256 + tid] = v;
        // epilogue constant of layer n-1 (bf16 path): b_res_{n-1} * sqrt(1/2) + fc_t_n(emb)
        if (epi_c && n > 0) epi_c[(n - 1) * 256 + tid] = fmaf(b_res[(n - 1) * 256 + tid], 0.70710678118654752440f, v);
    }
}

// ----------------------------------------------------------------------------------------------
// scheduler updates (diffwave_ddpm.py); op order as in the reference, no FMA contraction
// ----------------------------------------------------------------------------------------------
__global__ void lincomb_kernel(int op, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                               float c0, float c1, float c2, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
        case 0:  // one-shot x0 (l.199-203): c0 * x_t - c1 * eps
            r = __fsub_rn(__fmul_rn(c0, x[i]), __fmul_rn(c1, y[i])); break;
        case 1:  // diffusion (l.66-67): c0 * x0 + c1 * z
            r = __fadd_rn(__fmul_rn(c0, x[i]), __fmul_rn(c1, z[i])); break;
        case 2:  // reverse step (l.159-160,100): mu = (x - c0 * eps) / c1 ; x = mu + c2 * z
            r = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(c0, y[i])), c1);
            if (z) r = __fadd_rn(r, __fmul_rn(c2, z[i]));
            break;
        default: r = 0.f;
    }
    out[i] = r;
}

// ----------------------------------------------------------------------------------------------
// fp32 (parity) WaveNet helpers
// ----------------------------------------------------------------------------------------------
template <bool SPLIT>      // SPLIT: write the split-f16 storage format (dmad_common.h) for the x3 GEMM tier
__global__ void wn_init_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ emb0, float* __restrict__ h, int L, int LP, long total, int hi_only) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 (4 channels) each
    if (idx >= total) return;
    const int c4 = (int)(idx & 63);
    const long pos = idx >> 6, bb = pos / L, t = pos - bb * L;
    const float xv = x[pos];
    float4 o;
    float* po = &o.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c4 * 4 + j;
        po[j] = __fadd_rn(relu_nan(__fadd_rn(__fmul_rn(w[c], xv), bias[c])), emb0[c]);
    }
    if (SPLIT) {
        u32x4_t v = split4(o.x, o.y, o.z, o.w);
        if (hi_only) { v[2] = 0u; v[3] = 0u; }      // error-attribution runs: the stored stream is f16 (GemmF32Args::diag bit 2)
        *(u32x4_t*)(h + ((bb * LP + kPad + t) * kC + c4 * 4)) = v;
    } else *(float4*)(h + ((bb * LP + kPad + t) * kC + c4 * 4)) = o;
}

template <bool SPLIT>
__global__ void scale_kernel(const float* __restrict__ x, float c, float* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = ((const float4*)x)[i];
    v.x = __fmul_rn(v.x, c); v.y = __fmul_rn(v.y, c); v.z = __fmul_rn(v.z, c); v.w = __fmul_rn(v.w, c);
    if (SPLIT) ((u32x4_t*)y)[i] = split4(v.x, v.y, v.z, v.w);
    else ((float4*)y)[i] = v;
}

// eps[n] = w . f[n][:256] + b  (final_conv.2, WaveNet.py:160-162): one wave per position
__global__ void __launch_bounds__(256) dot256_kernel(const float* __restrict__ f, const float* __restrict__ w, float bias,
                                                     float* __restrict__ out, long N) {
    const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const float4 v = *(const float4*)(f + n * 256 + lane * 4);
    const float4 ww = *(const float4*)(w + lane * 4);
    float s = v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[n] = s + bias;
}

// ----------------------------------------------------------------------------------------------
// mel front-end post-processing (torchaudio MelSpectrogram power=2 + AmplitudeToDB 'power')
// ----------------------------------------------------------------------------------------------
// xp[b][0:1024] = 0, xp[b][1024:1024+L] = x[b], xp[b][1024+L:] = 0       (center=True, constant pad)
__global__ void mel_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int L, int LPm, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long b = i / LPm;
    const int p = (int)(i - b * LPm) - 1024;
    xp[i] = (p >= 0 && p < L) ? x[b * L + p] : 0.f;
}

// P[n][f] = re^2 + im^2, f < 1025 ; zero for the K padding up to ldp.  D is [n][ldd]: re at f, im at 1025 + f
__global__ void mel_power_kernel(const float* __restrict__ D, float* __restrict__ P, int ldd, int ldp, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long n = i / ldp;
    const int f = (int)(i - n * ldp);
    float v = 0.f;
    if (f < 1025) {
        const float re = D[n * ldd + f], im = D[n * ldd + 1025 + f];
        v = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
    }
    P[i] = v;
}

// spec[b][mel][frame] = 10 * log10(max(M[b*32 + frame][mel], 1e-10))
__global__ void mel_db_kernel(const float* __restrict__ M, float* __restrict__ spec, long total, int to_db) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long b = i >> 10;
    const int mel = (int)((i >> 5) & 31), fr = (int)(i & 31);
    if (!to_db) { spec[i] = M[(b * 32 + fr) * 32 + mel]; return; }     // MelSpectrogram alone: power, [b][mel][frame]
    const float v = clamp_min_nan(M[(b * 32 + fr) * 32 + mel], 1e-10f);
    // fp32 log10 rounded from a double evaluation (10*log10(1e-10f) must be exactly -100 like on the CPU)
    spec[i] = __fmul_rn(10.f, (float)log10((double)v));
}

// ----------------------------------------------------------------------------------------------
// VGG helpers
// ----------------------------------------------------------------------------------------------
// first conv (Cin = 1) + folded BN + relu: in [B][32][32] -> out NHWC [B][32][32][64]
__global__ void vgg_conv1_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                                 const float* __restrict__ shift, float* __restrict__ out, long total, h16_t* __restrict__ out16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one (b, y, x, co) each
    if (i >= total) return;
    const int co = (int)(i & 63);
    const long p = i >> 6, b = p >> 10;
    const int y = (int)((p >> 5) & 31), x = (int)(p & 31);
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if ((unsigned)yy < 32u && (unsigned)xx < 32u) s = fmaf(w[co * 9 + ky * 3 + kx], in[(b << 10) + yy * 32 + xx], s);
        }
    const float v = relu_nan(s * scale[co] + shift[co]);
    if (out) out[i] = v;
    if (out16) { const _Float16 hv = (_Float16)v; out16[i] = __builtin_bit_cast(unsigned short, hv); }     // f16 twin (the classifier's 16-bit tier)
}

// 2x2 max pool, NHWC
__global__ void maxpool2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of output
    if (i >= total4) return;
    const int c4n = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int c4 = (int)(i % c4n);
    long p = i / c4n;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const long b = p / Ho;
    const float* s = in + (((b * H + 2 * yo) * W + 2 * xo) * (long)C) + c4 * 4;
    const float4 a = *(const float4*)s, bq = *(const float4*)(s + C), c = *(const float4*)(s + (long)W * C),
                 d = *(const float4*)(s + (long)W * C + C);
    float4 o;
    o.x = max_nan(max_nan(a.x, bq.x), max_nan(c.x, d.x)); o.y = max_nan(max_nan(a.y, bq.y), max_nan(c.y, d.y));
    o.z = max_nan(max_nan(a.z, bq.z), max_nan(c.z, d.z)); o.w = max_nan(max_nan(a.w, bq.w), max_nan(c.w, d.w));
    ((float4*)out)[i] = o;
}

// global average pool over the HW pixels of an NHWC map (F.avg_pool2d(x, 8, 1) on the 8x8 map of ResNeXt29,
// models/resnext.py:139): pixels summed in index order, then one division
__global__ void avgpool_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one output element [b][c]
    if (i >= total) return;
    const long b = i / C;
    const int c = (int)(i - b * C);
    const float* s = in + b * HW * (long)C + c;
    float acc = 0.f;
    for (int p = 0; p < HW; ++p) acc += s[(long)p * C];
    out[i] = acc / (float)HW;
}

// ----------------------------------------------------------------------------------------------
// votes: arg-max (first maximum wins, like torch.max) + per-class count (certified_robust.py:59-65)
// ----------------------------------------------------------------------------------------------
__global__ void vote_kernel(const float* __restrict__ logits, int B, int C, unsigned long long* __restrict__ counts,
                            int* __restrict__ pred_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int best = 0;
    float bv = logits[(long)b * C];
    for (int c = 1; c < C; ++c) {
        const float v = logits[(long)b * C + c];
        if (v > bv || (v != v && bv == bv)) { bv = v; best = c; }
    }
    if (pred_out) pred_out[b] = best;
    if (counts) atomicAdd(&counts[best], 1ull);
}

// Exact-vote mode: a sample votes from these (bf16-path) logits only if its top-2 margin is >= tau; otherwise its global
// index sample_base + b is queued for the exact-fp32 re-evaluation (any NaN fails the comparison and is queued too).
__global__ void vote_margin_kernel(const float* __restrict__ logits, int B, int C, unsigned long long* __restrict__ counts,
                                   float tau, long long sample_base, const long long* __restrict__ idx, long long* __restrict__ list,
                                   unsigned long long* __restrict__ list_n, long long list_cap, int* __restrict__ pred_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int best = 0;
    float bv = logits[(long)b * C], second = -INFINITY;
    bool nan = bv != bv;
    for (int c = 1; c < C; ++c) {
        const float v = logits[(long)b * C + c];
        nan |= v != v;
        if (v > bv) { second = bv; bv = v; best = c; }
        else if (v > second) second = v;
    }
    if (pred_out) pred_out[b] = best;
    if (!nan && (C == 1 || bv - second >= tau)) {
        atomicAdd(&counts[best], 1ull);
    } else {
        // the counter keeps counting past the capacity (the host reads it and reports the overflow); nothing is written there
        const unsigned long long slot = atomicAdd(list_n, 1ull);
        if (slot < (unsigned long long)list_cap) list[slot] = idx ? idx[b] : sample_base + b;     // row b is global sample idx[b] (a recheck pass) or sample_base + b
    }
}

// ---------------------------------------------------------------------------------------------- launchers
static inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_philox_raw(uint64_t seed, uint64_t sample, uint32_t stream, uint32_t nblocks, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(philox_raw_kernel, dim3(nblk(nblocks, 256)), dim3(256), 0, s, seed, sample, stream, nblocks, out);
}
void launch_philox_normal(uint64_t seed, uint64_t sample0, uint32_t stream, float* z, int B, int L, hipStream_t s, const long long* idx) {
    hipLaunchKernelGGL(philox_normal_kernel, dim3(nblk((long)B * (L / 4), 256)), dim3(256), 0, s, seed, sample0, stream, z, B, L, idx);
}
void launch_mc_noise_scale(const float* clip, const float* delta, float sigma, float scale, uint64_t seed, uint64_t sample0,
                           float* xt, int B, int L, hipStream_t s) {
    hipLaunchKernelGGL(mc_noise_scale_kernel, dim3(nblk((long)B * (L / 4), 256)), dim3(256), 0, s, clip, delta, sigma, scale,
                       seed, sample0, xt, B, L);
}
void launch_mc_noise_scale_idx(const float* clip, const float* delta, float sigma, float scale, uint64_t seed, uint64_t sample0,
                               const long long* idx, float* xt, int B, int L, hipStream_t s) {
    hipLaunchKernelGGL(mc_noise_scale_idx_kernel, dim3(nblk((long)B * (L / 4), 256)), dim3(256), 0, s, clip, delta, sigma, scale,
                       seed, sample0, idx, xt, B, L);
}
void launch_scatter_rows(const float* src, const long long* idx, long long base, float* dst, int B, int W, hipStream_t s) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(nblk((long)B * W, 256)), dim3(256), 0, s, src, idx, base, dst, B, W);
}
void launch_repeat_rows(const float* x, float* out, int B, long row0, int nrows, int L, hipStream_t s) {
    const long total4 = (long)nrows * (L / 4);
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(nblk(total4, 256)), dim3(256), 0, s, x, out, B, row0, L, total4);
}
void launch_vote_margin(const float* logits, int B, int C, unsigned long long* counts, float tau, long long sample_base,
                        const long long* idx, long long* list, unsigned long long* list_n, long long list_cap, int* pred_out, hipStream_t s) {
    hipLaunchKernelGGL(vote_margin_kernel, dim3(nblk(B, 64)), dim3(64), 0, s, logits, B, C, counts, tau, sample_base, idx, list, list_n,
                       list_cap, pred_out);
}
void launch_embed_table(float t, const float* w1, const float* b1, const float* w2, const float* b2, const float* wt,
                        const float* bt, float* table, float* emb2_out, const float* b_res, float* epi_c, int NL, hipStream_t s) {
    hipLaunchKernelGGL(embed_table_kernel, dim3(NL), dim3(512), 0, s, t, w1, b1, w2, b2, wt, bt, table, emb2_out, b_res, epi_c);
}
void launch_lincomb(int op, const float* x, const float* y, const float* z, float c0, float c1, float c2, float* out, long n,
                    hipStream_t s) {
    hipLaunchKernelGGL(lincomb_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, op, x, y, z, c0, c1, c2, out, n);
}
void launch_wn_init_f32(const float* x, const float* w, const float* bias, const float* emb0, float* h, int B, int L, int LP,
                        hipStream_t s, bool split, bool hi_only) {
    const long total = (long)B * L * 64;
    if (split) hipLaunchKernelGGL(wn_init_f32_kernel<true>, dim3(nblk(total, 256)), dim3(256), 0, s, x, w, bias, emb0, h, L, LP, total, hi_only ? 1 : 0);
    else hipLaunchKernelGGL(wn_init_f32_kernel<false>, dim3(nblk(total, 256)), dim3(256), 0, s, x, w, bias, emb0, h, L, LP, total, 0);
}
void launch_scale(const float* x, float c, float* y, long n, hipStream_t s, bool split) {
    if (split) hipLaunchKernelGGL(scale_kernel<true>, dim3(nblk(n / 4, 256)), dim3(256), 0, s, x, c, y, n / 4);
    else hipLaunchKernelGGL(scale_kernel<false>, dim3(nblk(n / 4, 256)), dim3(256), 0, s, x, c, y, n / 4);
}
void launch_dot256(const float* f, const float* w, float bias, float* out, long N, hipStream_t s) {
    hipLaunchKernelGGL(dot256_kernel, dim3(nblk(N, 4)), dim3(256), 0, s, f, w, bias, out, N);
}
void launch_mel_pad(const float* x, float* xp, int B, int L, int LPm, hipStream_t s) {
    const long total = (long)B * LPm;
    hipLaunchKernelGGL(mel_pad_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, x, xp, L, LPm, total);
}
void launch_mel_power(const float* D, float* P, int ldd, int ldp, long rows, hipStream_t s) {
    const long total = rows * ldp;
    hipLaunchKernelGGL(mel_power_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, D, P, ldd, ldp, total);
}
void launch_mel_db(const float* M, float* spec, int B, int to_db, hipStream_t s) {
    const long total = (long)B * 1024;
    hipLaunchKernelGGL(mel_db_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, M, spec, total, to_db);
}
// AmplitudeToDB(stype='power') on its own: y = 10 * log10(max(x, 1e-10))
__global__ void power_to_db_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __fmul_rn(10.f, (float)log10((double)clamp_min_nan(x[i], 1e-10f)));
}
void launch_power_to_db(const float* x, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(power_to_db_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, y, n);
}
void launch_vgg_conv1(const float* in, const float* w, const float* scale, const float* shift, float* out, int B, hipStream_t s, h16_t* out16) {
    const long total = (long)B * 1024 * 64;
    hipLaunchKernelGGL(vgg_conv1_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, in, w, scale, shift, out, total, out16);
}
void launch_maxpool2_nhwc(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
    const long total4 = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_nhwc_kernel, dim3(nblk(total4, 256)), dim3(256), 0, s, in, out, H, W, C, total4);
}
void launch_avgpool_nhwc(const float* in, float* out, int B, int HW, int C, hipStream_t s) {
    const long total = (long)B * C;
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, in, out, HW, C, total);
}
void launch_vote(const float* logits, int B, int C, unsigned long long* counts, int* pred_out, hipStream_t s) {
    hipLaunchKernelGGL(vote_kernel, dim3(nblk(B, 64)), dim3(64), 0, s, logits, B, C, counts, pred_out);
}

// out[0:128] = v: a small host-computed vector travels as a kernel argument (copied at launch: no host buffer to keep alive,
// no stream synchronisation)
struct Vec128 { float v[128]; };
__global__ void store_vec128_kernel(Vec128 a, float* __restrict__ out) { out[threadIdx.x] = a.v[threadIdx.x]; }
void launch_store_vec128(const float* host128, float* out, hipStream_t s) {
    Vec128 a;
    for (int i = 0; i < 128; ++i) a.v[i] = host128[i];
    hipLaunchKernelGGL(store_vec128_kernel, dim3(1), dim3(128), 0, s, a, out);
}

void philox4x32_10_host(uint32_t c[4], uint32_t k0, uint32_t k1) { philox4x32_10(c, k0, k1); }

}  // namespace dmad
