// See unet_ops.h.
#include "unet_ops.h"
#include <stdlib.h>

namespace dmad {

namespace {
inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1) / b); }

// one workgroup per pair of image rows (64 pixels: one GroupNorm statistics block), one thread per output channel: its 9 weights live
// in registers, the input values a row needs are wave-uniform (scalar loads), every store instruction writes whole channel rows.
// out (fp32 map) and out16 (f16 twin) are both optional; stats != nullptr: the (sum, sum of squares) of the f16-rounded outputs per
// 64-pixel block and channel quad, in GemmH16Args::stats layout, so that the consumer GroupNorm is the one-pass kernel.
__global__ void __launch_bounds__(128) conv1ch_3x3_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int Cout, h16_t* __restrict__ out16, float* __restrict__ stats) {
    const int co = threadIdx.x, yp = blockIdx.x & 15;
    const long b = blockIdx.x >> 4;
    if (co >= Cout) return;
    float k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = w[co * 9 + t];
    const float bv = bias[co];
    const float* img = in + (b << 10);
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < 2; ++r) {
        const int y = yp * 2 + r;
        const long row = ((b << 10) + y * 32) * Cout + co;
        for (int x = 0; x < 32; ++x) {
            float s = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    if ((unsigned)yy < 32u && (unsigned)xx < 32u) s = fmaf(k[ky * 3 + kx], img[yy * 32 + xx], s);
                }
            const _Float16 hv = (_Float16)(s + bv);
            if (out) out[row + (long)x * Cout] = s + bv;
            if (out16) out16[row + (long)x * Cout] = __builtin_bit_cast(h16_t, hv);
            const float t = (float)hv;
            s1 += t;
            s2 = fmaf(t, t, s2);
        }
    }
    if (stats) {                                   // Cout % 4 == 0 (launcher): the four channels of a quad are four adjacent lanes
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
        if ((co & 3) == 0) *(float2*)(stats + (((b << 4) + yp) * (Cout >> 2) + (co >> 2)) * 2) = float2{s1, s2};
    }
}

// conv 3x3, 128 channels -> ONE output channel, padding 1, on 32x32 maps (the network's last layer, unet.py:421): a
// 1152-term dot product per pixel, far too thin for a matrix tile (the 64-row GEMM tile spends 63/64 of its MFMAs on padding).
// One lane per pixel; the weights are wave-uniform (scalar loads), each lane streams the 512-byte rows of its 9 neighbours.
__global__ void __launch_bounds__(256) conv3x3_c128_to1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out, long total) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int y = (int)((p >> 5) & 31), x = (int)(p & 31);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if ((unsigned)yy >= 32u || (unsigned)xx >= 32u) continue;
        const float4* row = (const float4*)(in + (((p >> 10) << 10) + yy * 32 + xx) * 128);
        const float4* wr = (const float4*)(w + tap * 128);
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float4 v = row[c], k = wr[c];
            acc[0] = fmaf(v.x, k.x, acc[0]); acc[1] = fmaf(v.y, k.y, acc[1]); acc[2] = fmaf(v.z, k.z, acc[2]); acc[3] = fmaf(v.w, k.w, acc[3]);
        }
    }
    out[p] = (acc[0] + acc[1]) + (acc[2] + acc[3]) + bias[0];
}

// The same conv on an f16 map (the 16-bit tier's last layer), on the matrix cores: per pixel the nine taps' dot products with its OWN
// 128 channels — a 16 x 16 x 32 MFMA chain with the taps as rows (9 of 16) and 16 pixels as columns, the weights split into f16
// hi + lo parts so that they keep fp32 precision — go to LDS, and an output pixel is the sum of nine neighbours' partials.  Every
// input byte is read once (the direct form above re-reads each row nine times through L2: 675 us per 2048 spectrograms; this 1).
// One workgroup per image: 4 waves x 16 pixels per step, 16 steps, then 4 outputs per thread.
__global__ void __launch_bounds__(256) conv3x3_c128_to1_h16_kernel(const h16_t* __restrict__ in, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ out) {
    __shared__ float P[1024 * 9];                          // [pixel][tap]: 9-float rows (odd stride: conflict-free column reads)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, r16 = lane & 15;
    const long b = blockIdx.x;
    f16x8 ah[4], al[4];                                    // A fragments: row = tap r16 (rows 9-15 zero), k = 32 kk + 8 q ..
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float wf = r16 < 9 ? w[r16 * 128 + kk * 32 + q * 8 + r] : 0.f;
            const _Float16 h = (_Float16)wf;
            ah[kk][r] = h;
            al[kk][r] = (_Float16)(wf - (float)h);
        }
    const h16_t* img = in + b * 1024 * 128;
    for (int it = 0; it < 16; ++it) {
        const int px = it * 64 + wv * 16 + r16;
        f16x8 x[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x[kk] = *(const f16x8*)(img + px * 128 + kk * 32 + q * 8);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kk], x[kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk], x[kk], acc, 0, 0, 0);
        }
        if (q < 2) {                                       // lane (q, r16): taps 4 q .. 4 q + 3 of pixel r16
#pragma unroll
            for (int r = 0; r < 4; ++r) P[px * 9 + q * 4 + r] = acc[r];
        } else if (q == 2) {
            P[px * 9 + 8] = acc[0];
        }
    }
    __syncthreads();
    const float bv = bias[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + i * 256, y = p >> 5, xq = p & 31;
        float sum = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = xq + tap % 3 - 1;
            if ((unsigned)yy < 32u && (unsigned)xx < 32u) sum += P[(yy * 32 + xx) * 9 + tap];
        }
        out[b * 1024 + p] = sum + bv;
    }
}

// GroupNorm32(32, C) over an NHWC map.  One workgroup per (sample, G neighbouring groups), G chosen so that the G groups'
// slice of a pixel is whole 128-byte lines (C/32 = 4, 8, 12 or 16 channels per group -> G = 8, 4, 8 or 4, 2): every load and
// store of a wave is then full cache lines (a single group's slice is only 16..64 bytes of its line).  The slab — HW pixels x
// R = G * C/128 float4s — stays in REGISTERS: thread t owns float4s t, t + NT, ... (PER of them) and, because NT is a
// multiple of R, all of them belong to the same group and the same 4 channels.  All loads are in flight at once; then the
// mean, the biased variance of the deviations (two-pass, like torch) and the affine / scale-shift / SiLU pass run without
// touching memory again.  Group sums: per-thread partials to LDS, wave w adds the partials of group w in a fixed order
// (deterministic: no atomics), everybody reads the G results.
// IN16 (the UNet's 16-bit tier): the input map(s) are read from their f16 twins (x / x2 then point at f16 data; half the bytes of a
// kernel that is bound by its memory traffic); statistics and arithmetic stay fp32.
template <int PER, bool IN16>
__global__ void __launch_bounds__(1024) groupnorm_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ ss, int silu,
                                                              float* __restrict__ y, int HW, int C, int G,
                                                              const float* __restrict__ x2, int c1, h16_t* __restrict__ y16) {
    __shared__ float part[1024];
    __shared__ float gsum[8];
    const bool split_out = (silu & 2) != 0;                    // flags: bit 0 SiLU, bit 1 the fp32 output in the split-f16 storage format
    silu &= 1;
    const int NT = blockDim.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, nwaves = NT >> 6;
    const int c4 = C >> 7, R = G * c4, upb = 32 / G;          // upb: workgroups per sample
    const int b = blockIdx.x / upb, g0 = (blockIdx.x % upb) * G;
    const int j = t % R, grp = j / c4, c = (g0 + grp) * (C >> 5) + (j % c4) * 4;
    const long base = (long)b * HW * C + c;                                  // + pixel * C   (c = g0 * C/32 + 4 j)
    // input: one map of C channels, or the concatenation [x (c1 channels) | x2 (C - c1 channels)] read in place
    const bool second = x2 && c >= c1;
    const int pitch = x2 ? (second ? C - c1 : c1) : C;
    const long soff = second ? (long)b * HW * pitch + (c - c1) : (long)b * HW * pitch + c;        // in elements of the input type
    const float* src = (second ? x2 : x) + soff;
    const h16_t* src16 = (const h16_t*)(second ? x2 : x) + soff;
    const int p0 = t / R, pstep = NT / R;
    const float n = (float)((C >> 5) * HW);
    auto group_sum = [&](float v) -> float {
        __syncthreads();
        part[t] = v;
        __syncthreads();
        const int members = c4 * pstep;
        for (int w = wv; w < G; w += nwaves) {
            float a = 0.f;
            for (int i = lane; i < members; i += 64) a += part[(i / c4) * R + w * c4 + (i % c4)];
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) gsum[w] = a;
        }
        __syncthreads();
        return gsum[grp];
    };
    float4 v[PER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pix = p0 + k * pstep;
        if (pix >= HW) v[k] = float4{0.f, 0.f, 0.f, 0.f};
        else if (IN16) { const f16x4 hv = *(const f16x4*)(src16 + (long)pix * pitch); v[k] = float4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}; }
        else v[k] = *(const float4*)(src + (long)pix * pitch);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float mean = group_sum(s) / n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (p0 + k * pstep < HW) {
            const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const float rstd = 1.0f / sqrtf(group_sum(q) / n + 1e-5f);
    const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
    float4 s1 = float4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (ss) { s1 = *(const float4*)(ss + c); s2 = *(const float4*)(ss + C + c); }
    const float gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bea[4] = {be.x, be.y, be.z, be.w}, s1a[4] = {s1.x, s1.y, s1.z, s1.w}, s2a[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pix = p0 + k * pstep;
        if (pix >= HW) continue;
        float o[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = (o[r] - mean) * rstd * gaa[r] + bea[r];
            if (ss) u = u * (1.f + s1a[r]) + s2a[r];
            // SiLU.  On the 16-bit tier (f16 output: 2^-11 relative) one v_exp and one v_rcp (1 ulp each) instead of libm's expf and
            // an IEEE division — ~40 of the kernel's ~47 vector instructions per value, which made it compute- not bandwidth-bound
            if (silu) u = IN16 ? u * fast_rcp(1.f + fast_exp2(u * -1.4426950408889634f)) : u / (1.f + expf(-u));
            o[r] = u;
        }
        if (y16) *(f16x4*)(y16 + base + (long)pix * C) = f16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
        else if (split_out) *(u32x4_t*)(y + base + (long)pix * C) = split4(o[0], o[1], o[2], o[3]);      // operand of a split-f16 GEMM (same 16 bytes)
        else *(float4*)(y + base + (long)pix * C) = float4{o[0], o[1], o[2], o[3]};
    }
}

#ifndef GN_UNROLL
#define GN_UNROLL 4
#endif
// GroupNorm + SiLU (+ scale-shift) of the UNet's 16-bit tier in streaming form: f16 map(s) in, f16 (or fp32) map out, the same
// (sample, G groups) slab per workgroup as above — but nothing is held in registers: pass 1 streams the slab and accumulates the
// shifted sums S1 = sum(x - K), S2 = sum((x - K)^2) per group (K = the group's first value: the cancellation in S2 - S1^2 / n is then
// bounded by the group's spread, not by its mean), pass 2 re-reads it (64 KiB per workgroup: L2 hits) and writes the result.  A
// workgroup is 192 / 256 threads with 76 registers, so 6 of them share a CU and their load / reduce / store phases overlap, where the
// register-resident form runs ONE 1024-thread workgroup per CU through its phases in turn (2.2 TB/s on the 32 x 32 maps); and it takes
// TWICE the groups per workgroup, so that a pixel's slice is whole 128-byte lines of the f16 map too (with the fp32 form's G a pixel
// contributes 64 bytes: that alone made this kernel 3 % slower than the register-resident one; with it, 2.4 % faster on C5).  The
// fp32 tier keeps the register-resident two-pass form (torch's arithmetic); here the output is f16 and the statistics differ from
// it by ~1e-6 relative.  Fixed summation order: a sample's result does not depend on the batch.
__global__ void __launch_bounds__(256) groupnorm16_stream_kernel(const h16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ ss, int silu, float* __restrict__ y, int HW, int C, int G,
                                                                 const h16_t* __restrict__ x2, int c1, h16_t* __restrict__ y16) {
    __shared__ float part[2][256];
    __shared__ float gs[2][16];
    const int NT = blockDim.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, nwaves = NT >> 6;
    const int c4 = C >> 7, R = G * c4, upb = 32 / G;
    const int b = blockIdx.x / upb, g0 = (blockIdx.x % upb) * G;
    const int j = t % R, grp = j / c4, cpg = C >> 5, c = (g0 + grp) * cpg + (j % c4) * 4;
    const bool second = x2 && c >= c1;
    const int pitch = x2 ? (second ? C - c1 : c1) : C;
    const h16_t* src = (second ? x2 : x) + (second ? (long)b * HW * pitch + (c - c1) : (long)b * HW * pitch + c);
    // the group's shift: its first channel at pixel 0 (every thread of the group reads the same value)
    const int cg = (g0 + grp) * cpg;
    const bool gsecond = x2 && cg >= c1;
    const int gpitch = x2 ? (gsecond ? C - c1 : c1) : C;
    const _Float16 kh = *(const _Float16*)((gsecond ? x2 : x) + (gsecond ? (long)b * HW * gpitch + (cg - c1) : (long)b * HW * gpitch + cg));
    const float K = (float)kh;
    const int p0 = t / R, pstep = NT / R;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll GN_UNROLL
    for (int pix = p0; pix < HW; pix += pstep) {
        const f16x4 hv = *(const f16x4*)(src + (long)pix * pitch);
        const float d0 = (float)hv[0] - K, d1 = (float)hv[1] - K, d2 = (float)hv[2] - K, d3 = (float)hv[3] - K;
        s1 += (d0 + d1) + (d2 + d3);
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    part[0][t] = s1; part[1][t] = s2;
    __syncthreads();
    const int members = c4 * pstep;
    for (int w = wv; w < 2 * G; w += nwaves) {                    // wave w: sum `w & 1` (S1 / S2) of group w >> 1, fixed order
        const int gi = w >> 1, which = w & 1;
        float a = 0.f;
        for (int i = lane; i < members; i += 64) a += part[which][(i / c4) * R + gi * c4 + (i % c4)];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) gs[which][gi] = a;
    }
    __syncthreads();
    const float n = (float)(cpg * HW);
    const float S1 = gs[0][grp], S2 = gs[1][grp];
    const float mean = K + S1 / n;
    const float var = fmaxf((S2 - S1 * (S1 / n)) / n, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
    float4 q1 = float4{0.f, 0.f, 0.f, 0.f}, q2 = q1;
    if (ss) { q1 = *(const float4*)(ss + c); q2 = *(const float4*)(ss + C + c); }
    // (x - mean) * rstd * gamma + beta, then * (1 + scale) + shift: one FMA per value with the constants folded
    float mul[4] = {rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w}, add[4] = {be.x, be.y, be.z, be.w};
    const float s1a[4] = {q1.x, q1.y, q1.z, q1.w}, s2a[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        add[r] -= mean * mul[r];
        if (ss) { mul[r] *= 1.f + s1a[r]; add[r] = add[r] * (1.f + s1a[r]) + s2a[r]; }
    }
    const long obase = (long)b * HW * C + c;
#pragma unroll GN_UNROLL
    for (int pix = p0; pix < HW; pix += pstep) {
        const f16x4 hv = *(const f16x4*)(src + (long)pix * pitch);
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = (float)hv[r] * mul[r] + add[r];
            if (silu) u = u * fast_rcp(1.f + fast_exp2(u * -1.4426950408889634f));
            o[r] = u;
        }
        if (y16) *(f16x4*)(y16 + obase + (long)pix * C) = f16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
        else *(float4*)(y + obase + (long)pix * C) = float4{o[0], o[1], o[2], o[3]};
    }
}

// GroupNorm + SiLU (+ scale-shift) of the 16-bit tier as ONE streaming pass (round 4): the statistics come from the producing GEMMs'
// epilogues (GemmH16Args::stats: (sum, sum of squares) per 64-pixel block and 4-channel quad of the f16 map), so this kernel only
// sums a sample's partials (fixed order: a sample's result does not depend on the batch), folds mean / rstd / gamma / beta /
// scale-shift into one multiplier and one addend per channel, and applies them: 16-byte loads and stores of whole pixel rows (every
// 128-byte line touched once), where the two-pass forms above re-read their slab and move 8 bytes per lane.
// x / x2: the map, or the two parts [c1 | C - c1 channels] of a concatenated input, each with the slab of the GEMM that wrote it;
// workgroup = (sample, one of `wps` pixel ranges), 256 threads (192 for C = 384: a multiple of the C / 8 octets per pixel).
__global__ void __launch_bounds__(256) groupnorm16_apply_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ x2, int c1,
                                                                const float* __restrict__ st1, const float* __restrict__ st2,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ ss, int silu, h16_t* __restrict__ y16,
                                                                float* __restrict__ y32, int HW, int C, int wps, int bshift) {
    __shared__ float mulc[512], addc[512], gsum[2][32];
    const int NT = blockDim.x, t = threadIdx.x;
    const int b = blockIdx.x / wps, part = blockIdx.x - b * wps;
    const int cpg = C >> 5, qpg = cpg >> 2, nblk = HW >> bshift;     // statistics blocks of 64 pixels (bshift 6), or of 16 on the 4 x 4 maps
    const int ca = x2 ? c1 : C, cb = C - ca;                         // channels of the first / second part
    {   // group sums: 8 threads per group, entry e = (pixel block, quad of the group), fixed xor tree over the 8
        const int k = t & 7, entries = nblk * qpg;
        for (int g = t >> 3; g < 32; g += NT >> 3) {                 // (a wave's 8 groups are all below 32 or all above: uniform trip count)
            float s1 = 0.f, s2 = 0.f;
            for (int e = k; e < entries; e += 8) {
                const int blk = e / qpg, quad = g * qpg + (e - blk * qpg);
                const float2 v = (4 * quad < ca) ? *(const float2*)(st1 + ((size_t)(b * nblk + blk) * (ca >> 2) + quad) * 2)
                                                 : *(const float2*)(st2 + ((size_t)(b * nblk + blk) * (cb >> 2) + (quad - (ca >> 2))) * 2);
                s1 += v.x; s2 += v.y;
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (k == 0) { gsum[0][g] = s1; gsum[1][g] = s2; }
        }
    }
    __syncthreads();
    const float n = (float)(cpg * HW);
    for (int c = t; c < C; c += NT) {
        const int g = c / cpg;
        const float mean = gsum[0][g] / n;
        const float var = fmaxf(gsum[1][g] / n - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        float m = rstd * gamma[c], a = beta[c] - mean * m;
        if (ss) { const float sc = 1.f + ss[c]; m *= sc; a = a * sc + ss[C + c]; }
        mulc[c] = m; addc[c] = a;
    }
    __syncthreads();
    const int opp = C >> 3, o = t % opp, ppp = NT / opp;             // octets per pixel, this thread's octet, pixels per pass
    const int c0 = o * 8;
    const bool second = x2 && c0 >= ca;
    const h16_t* src = second ? x2 + (size_t)b * HW * cb + (c0 - ca) : x + (size_t)b * HW * ca + c0;
    const int pitch = second ? cb : ca;
    float mu[8], ad[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { mu[r] = mulc[c0 + r]; ad[r] = addc[c0 + r]; }
    const int per = HW / wps, p_lo = part * per, p_hi = p_lo + per;
    const size_t obase = (size_t)b * HW * C + c0;
#pragma unroll 4
    for (int pix = p_lo + t / opp; pix < p_hi; pix += ppp) {
        const f16x8 hv = *(const f16x8*)(src + (size_t)pix * pitch);
        float u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float w = __builtin_fmaf((float)hv[r], mu[r], ad[r]);
            if (silu) w = w * fast_rcp(1.f + fast_exp2(w * -1.4426950408889634f));
            u[r] = w;
        }
        if (y16) {
            *(f16x8*)(y16 + obase + (size_t)pix * C) = f16x8{(_Float16)u[0], (_Float16)u[1], (_Float16)u[2], (_Float16)u[3],
                                                             (_Float16)u[4], (_Float16)u[5], (_Float16)u[6], (_Float16)u[7]};
        } else {
            *(float4*)(y32 + obase + (size_t)pix * C) = float4{u[0], u[1], u[2], u[3]};
            *(float4*)(y32 + obase + (size_t)pix * C + 4) = float4{u[4], u[5], u[6], u[7]};
        }
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

__global__ void upsample2x_nhwc_h16_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, int H, int W, int C8, long total8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // 8 halves (16 bytes) of output
    if (i >= total8) return;
    const int c = (int)(i % C8);
    long p = i / C8;
    const int xo = (int)(p % (2 * W)); p /= 2 * W;
    const int yo = (int)(p % (2 * H));
    const long b = p / (2 * H);
    ((u32x4_t*)out)[i] = ((const u32x4_t*)in)[((b * H + (yo >> 1)) * W + (xo >> 1)) * C8 + c];
}

// QKVAttention on the fp32 matrix cores, transposed so that no fragment ever changes layout:
//   S^T = K Q^T   (A = key rows, B = query rows, contraction over the 64 head channels)
//   O^T = V^T P^T (A = V read column-wise from LDS, B = the softmax weights exactly where the first product left them)
// A 16x16 tile of S^T leaves lane (c, g) (c = lane & 15, g = lane >> 4) with keys 4g + r (r = 0..3) of query c: query = lane
// column, so the softmax over keys is a reduction over this lane's registers and the 4 lanes sharing c (two xor-shuffles),
// and register r of that tile is at once the B operand of the second product for the k-slot assignment
// "MFMA j contracts keys {4g + j}".  One workgroup per (head, sample): K and V of the head in LDS (rows of 68 floats: the
// ds_read_b128 of a key row and the ds_read_b32 of a V column are both conflict free), each wave owns query tiles of 16,
// holds the whole S^T column block (T / 16 tiles) in registers and does the exact two-pass softmax torch does.
constexpr int HD = 64, AROW = 68;
template <int KT>                                       // T = 16 * KT keys / queries
__global__ void __launch_bounds__(KT >= 8 ? 512 : 64 * KT) qkv_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int heads, h16_t* __restrict__ out16, int split) {
    extern __shared__ __attribute__((aligned(16))) float att_lds[];
    constexpr int T = 16 * KT, NW = KT >= 8 ? 8 : KT;
    float* Ks = att_lds;
    float* Vs = att_lds + T * AROW;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C3 = 3 * HD * heads;
    const float* base = qkv + (long)b * T * C3 + h * 3 * HD;
    for (int i = tid; i < T * 16; i += NW * 64) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        *(float4*)(Ks + r * AROW + c4) = *(const float4*)(base + (long)r * C3 + HD + c4);
        *(float4*)(Vs + r * AROW + c4) = *(const float4*)(base + (long)r * C3 + 2 * HD + c4);
    }
    __syncthreads();
    for (int qt = wv; qt < KT; qt += NW) {
        f32x4 qf[4];                                    // query row c of the tile, channels 16 cc + 4 g + j, times scale^2 = 1/8
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float4 v = *(const float4*)(base + (long)(qt * 16 + c) * C3 + cc * 16 + g * 4);
            qf[cc] = f32x4{v.x * 0.125f, v.y * 0.125f, v.z * 0.125f, v.w * 0.125f};
        }
        f32x4 sacc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const f32x4 kf = *(const f32x4*)(Ks + (kt * 16 + c) * AROW + cc * 16 + g * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qf[cc][j], sacc[kt], 0, 0, 0);
                if (kt & 1) __builtin_amdgcn_sched_barrier(0);           // keep the fragment reads next to their MFMAs (register budget)
            }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) m = fmaxf(fmaxf(fmaxf(sacc[kt][0], sacc[kt][1]), fmaxf(sacc[kt][2], sacc[kt][3])), m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = expf(sacc[kt][r] - m); sacc[kt][r] = p; l += p; }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vrow = Vs + (kt * 16 + g * 4 + j) * AROW + c;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[dt * 16], sacc[kt][j], oacc[dt], 0, 0, 0);
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
        const float inv = 1.f / l;
        const long ooff = ((long)b * T + qt * 16 + c) * (HD * heads) + h * HD + g * 4;       // O^T tile dt: channels dt * 16 + 4 g + r of query c
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (out16) *(f16x4*)(out16 + ooff + dt * 16) = f16x4{(_Float16)(oacc[dt][0] * inv), (_Float16)(oacc[dt][1] * inv), (_Float16)(oacc[dt][2] * inv), (_Float16)(oacc[dt][3] * inv)};
            else if (split) *(u32x4_t*)(out + ooff + dt * 16) = split4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);      // operand of a split-f16 GEMM
            else *(float4*)(out + ooff + dt * 16) = float4{oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv};
        }
    }
}

// The same attention on the 16-bit tier: f16 q / k / v (the qkv conv's f16 output), v_mfma_f32_16x16x32_f16 for both products, fp32
// scores / softmax / output accumulation, f16 output (proj_out's operand).  Same transposed scheme — S^T = K Q^T leaves lane (c, g)
// with keys 4g + r of query c per 16-key tile — with the k-slot assignment of the second product chosen so that its B operand is
// register-local again: one MFMA contracts the keys of TWO tiles, k-slot (q, j) = key 4q + j of tile 2p (j < 4) or of tile 2p + 1
// (j >= 4), i.e. B = {P[2p][0..3], P[2p+1][0..3]} of the lane itself.  LDS: K as [key][64 ch] f16 (128-byte rows, chunks
// XOR-swizzled by (key >> 1) & 7: conflict-free ds_read_b128 fragments); V as [key / 4][64 ch][4 keys] f16 with 128 bytes of padding
// per key group (640-byte rows), so that the A fragment of the second product — 4 consecutive keys of one channel, twice — is two
// conflict-free ds_read_b64.  16x less matrix time than the fp32 form; what remains is the softmax's exponentials and the LDS fill.
constexpr int VG = 640;
template <int KT>
__global__ void __launch_bounds__(KT >= 8 ? 512 : 64 * KT) qkv_attention_h16_kernel(const h16_t* __restrict__ qkv, h16_t* __restrict__ out, int heads) {
    extern __shared__ __attribute__((aligned(16))) char att_h[];
    constexpr int T = 16 * KT, NW = KT >= 8 ? 8 : KT, KP = (KT + 1) / 2;
    char* Ks = att_h;                                   // T rows of 128 B
    char* Vg = att_h + T * 128;                         // T / 4 key groups of VG bytes
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C3 = 3 * HD * heads;
    const h16_t* base = qkv + (long)b * T * C3 + h * 3 * HD;
    for (int i = tid; i < T * 8; i += NW * 64) {
        const int r = i >> 3, ch = i & 7;               // key r, channels 8 ch .. 8 ch + 7
        const u32x4_t kv = *(const u32x4_t*)(base + (long)r * C3 + HD + ch * 8);
        *(u32x4_t*)(Ks + r * 128 + ((ch ^ ((r >> 1) & 7)) * 16)) = kv;
        const u32x4_t vv = *(const u32x4_t*)(base + (long)r * C3 + 2 * HD + ch * 8);
        unsigned short* vd = (unsigned short*)(Vg + (r >> 2) * VG + (ch * 8) * 8 + (r & 3) * 2);      // channel d: + d * 8 bytes
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned w = vv[e];
            vd[(2 * e) * 4] = (unsigned short)(w & 0xffffu);
            vd[(2 * e + 1) * 4] = (unsigned short)(w >> 16);
        }
    }
    __syncthreads();
    const int sw = (c >> 1) & 7;
#pragma unroll 1
    for (int qt = wv; qt < KT; qt += NW) {
        // query row c of the tile: channels 8g .. 8g+7 and 32 + 8g .. , times scale^2 = 1/8 (exact in f16)
        f16x8 qf[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f16x8 v = *(const f16x8*)(base + (long)(qt * 16 + c) * C3 + hh * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[hh][e] = v[e] * (_Float16)0.125f;
        }
        f32x4 sacc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f16x8 kf = *(const f16x8*)(Ks + (kt * 16 + c) * 128 + (((4 * hh + g) ^ sw) * 16));
                sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[hh], sacc[kt], 0, 0, 0);
            }
            // hipcc otherwise hoists all 2 KT fragment reads above the first MFMA: 128 registers at KT = 16, 92 spilled
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) m = fmaxf(fmaxf(fmaxf(sacc[kt][0], sacc[kt][1]), fmaxf(sacc[kt][2], sacc[kt][3])), m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        // exp(s - m) as one FMA + one v_exp (1 ulp; the weights go to the second product as f16 anyway) instead of libm's expf: the
        // 4 KT exponentials per lane and query tile were most of this kernel's vector instructions
        const float m2 = m * 1.4426950408889634f;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = fast_exp2(__builtin_fmaf(sacc[kt][r], 1.4426950408889634f, -m2)); sacc[kt][r] = p; l += p; }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const bool two = 2 * kp + 1 < KT;
            f16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (_Float16)sacc[2 * kp][r];
                pf[4 + r] = two ? (_Float16)sacc[two ? 2 * kp + 1 : 0][r] : (_Float16)0.f;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + c;
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const u32x2_t v0 = *(const u32x2_t*)(Vg + (8 * kp + g) * VG + d * 8);
                u32x2_t v1 = u32x2_t{0u, 0u};
                if (two) v1 = *(const u32x2_t*)(Vg + (8 * kp + 4 + g) * VG + d * 8);
                const f16x8 vf = __builtin_bit_cast(f16x8, u32x4_t{v0[0], v0[1], v1[0], v1[1]});
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc[dt], 0, 0, 0);
            }
            if (kp & 1) __builtin_amdgcn_sched_barrier(0);
        }
        const float inv = 1.f / l;
        const long ooff = ((long)b * T + qt * 16 + c) * (HD * heads) + h * HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *(f16x4*)(out + ooff + dt * 16) = f16x4{(_Float16)(oacc[dt][0] * inv), (_Float16)(oacc[dt][1] * inv), (_Float16)(oacc[dt][2] * inv), (_Float16)(oacc[dt][3] * inv)};
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

int launch_conv1ch_3x3(const float* in, const float* w, const float* bias, float* out, int B, int Cout, hipStream_t s, h16_t* out16, float* stats) {
    if (Cout > 128 || (stats && (Cout & 3)) || (!out && !out16)) return -1;      // one thread per output channel (this network: 128)
    hipLaunchKernelGGL(conv1ch_3x3_kernel, dim3((unsigned)B * 16u), dim3(128), 0, s, in, w, bias, out, Cout, out16, stats);
    return 0;
}
void launch_conv3x3_c128_to1(const float* in, const float* w, const float* bias, float* out, int B, hipStream_t s) {
    const long total = (long)B * 1024;
    hipLaunchKernelGGL(conv3x3_c128_to1_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, in, w, bias, out, total);
}
void launch_conv3x3_c128_to1_h16(const h16_t* in, const float* w, const float* bias, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(conv3x3_c128_to1_h16_kernel, dim3((unsigned)B), dim3(256), 0, s, in, w, bias, out);
}
int launch_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* ss, int silu, float* y, int B, int HW,
                          int C, hipStream_t s, const float* x2, int c1, h16_t* y16, const h16_t* x16, const h16_t* x2_16, int split) {
    const bool in16 = x16 != nullptr;
    if (split && (in16 || y16)) return -1;          // the split-format output belongs to the fp32-input kernel
    silu = (silu ? 1 : 0) | (split ? 2 : 0);        // the kernel's flag word
    if (in16) { x = (const float*)x16; x2 = (const float*)x2_16; }      // the kernel reinterprets them (IN16)
    const int c4 = C >> 7;
    if (C % 128 || c4 < 1 || c4 > 4 || B < 1 || HW < 1) return -1;
    if (x2 && (c1 < 4 || c1 >= C || (c1 & 3))) return -1;       // 32 groups of 4, 8, 12 or 16 channels
    // groups per workgroup: whole 128-byte lines per pixel; the largest map (32x32, 12 channels per group) takes 4 groups
    // (192 bytes) to keep the slab at 16 float4s per thread
    const int G = c4 == 1 ? 8 : c4 == 2 ? 4 : c4 == 3 ? (HW > 256 ? 4 : 8) : 2;
    const int R = G * c4;
    const long n4 = (long)HW * R;
    int NT = R * (1024 / R);                        // largest multiple of R (8, 12 or 24) and of 64 not above 1024: 1024 or 768
    if (R == 12 || R == 24) NT = 768;
    int per = (int)((n4 + NT - 1) / NT);
    const int want = HW >= 1024 ? 8 : HW >= 256 ? 4 : 2;          // float4s per thread: small maps are latency bound and want more threads
    while (per < want && NT > 64 && (NT / 2) % R == 0 && (NT / 2) % 64 == 0) { NT /= 2; per = (int)((n4 + NT - 1) / NT); }
    const dim3 grid((unsigned)(B * (32 / G)));
    static const bool stream_on = []() { const char* v = getenv("DMAD_GN_STREAM"); return !(v && v[0] == '0'); }();
    if (in16 && stream_on) {
        const int G2 = 2 * G;                                   // f16 input: twice the groups for the same whole 128-byte lines per pixel
        const int R2 = G2 * c4, nt = (R2 % 3 == 0) ? 192 : 256; // a multiple of R2 (16, 24 or 48) and of 64 (128 / 512 threads measured slower)
        hipLaunchKernelGGL(groupnorm16_stream_kernel, dim3((unsigned)(B * (32 / G2))), dim3(nt), 0, s, x16, gamma, beta, ss, silu, y, HW, C, G2, x2_16, c1, y16);
        return 0;
    }
#define GN_LAUNCH(PER)                                                                                                                  \
    do {                                                                                                                               \
        if (in16) hipLaunchKernelGGL((groupnorm_nhwc_kernel<PER, true>), grid, dim3(NT), 0, s, x, gamma, beta, ss, silu, y, HW, C, G, x2, c1, y16);  \
        else hipLaunchKernelGGL((groupnorm_nhwc_kernel<PER, false>), grid, dim3(NT), 0, s, x, gamma, beta, ss, silu, y, HW, C, G, x2, c1, y16);     \
    } while (0)
    if (per <= 1) GN_LAUNCH(1);
    else if (per <= 2) GN_LAUNCH(2);
    else if (per <= 4) GN_LAUNCH(4);
    else if (per <= 8) GN_LAUNCH(8);
    else if (per <= 16) GN_LAUNCH(16);
    else return -1;                                 // larger than any map of this network (32x32 x 384 channels)
#undef GN_LAUNCH
    return 0;
}
int launch_groupnorm16_apply(const h16_t* x, const float* st, const h16_t* x2, const float* st2, int c1, const float* gamma, const float* beta,
                             const float* ss, int silu, h16_t* y16, float* y32, int B, int HW, int C, hipStream_t s) {
    if (C % 128 || C < 128 || C > 512 || B < 1 || HW < 16 || (HW & 15) || (HW > 16 && (HW & 63)) || !x || !st || (!y16 && !y32)) return -1;
    if (x2 && (!st2 || c1 < 8 || c1 >= C || (c1 & 7))) return -1;
    const int wps = HW >= 1024 ? HW / 256 : 1;                      // 256 pixels per workgroup on the large maps, a whole sample otherwise
    const int nt = C == 384 ? 192 : 256;
    hipLaunchKernelGGL(groupnorm16_apply_kernel, dim3((unsigned)(B * wps)), dim3(nt), 0, s, x, x2, c1, st, st2, gamma, beta, ss, silu, y16, y32, HW, C, wps,
                       HW >= 64 ? 6 : 4);
    return 0;
}
void launch_silu(const float* x, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, y, n);
}
void launch_upsample2x_nhwc_h16(const h16_t* in, h16_t* out, int B, int H, int W, int C, hipStream_t s) {
    const long total8 = (long)B * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_nhwc_h16_kernel, dim3(nblk(total8, 256)), dim3(256), 0, s, in, out, H, W, C / 8, total8);
}
void launch_upsample2x_nhwc(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
    const long total4 = (long)B * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3(nblk(total4, 256)), dim3(256), 0, s, in, out, H, W, C / 4, total4);
}
int launch_qkv_attention(const float* qkv, float* out, int B, int T, int heads, hipStream_t s, h16_t* out16, int split) {
    const size_t lds = (size_t)2 * T * AROW * sizeof(float);
    if (T == 256) {
        static const hipError_t once = hipFuncSetAttribute((const void*)qkv_attention_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return (int)once;
        hipLaunchKernelGGL(qkv_attention_kernel<16>, dim3((unsigned)heads, (unsigned)B), dim3(512), lds, s, qkv, out, heads, out16, split);
    } else if (T == 64) {
        hipLaunchKernelGGL(qkv_attention_kernel<4>, dim3((unsigned)heads, (unsigned)B), dim3(256), lds, s, qkv, out, heads, out16, split);
    } else if (T == 16) {
        hipLaunchKernelGGL(qkv_attention_kernel<1>, dim3((unsigned)heads, (unsigned)B), dim3(64), lds, s, qkv, out, heads, out16, split);
    } else {
        return -1;                  // this network attends at 16x16, 8x8 (script_util.py attention_resolutions "16,8") and 4x4 (middle block)
    }
    return 0;
}
int launch_qkv_attention_h16(const h16_t* qkv, h16_t* out, int B, int T, int heads, hipStream_t s) {
    const size_t lds = (size_t)T * 128 + (size_t)(T / 4) * VG;
    if (T == 256) {
        static const hipError_t once = hipFuncSetAttribute((const void*)qkv_attention_h16_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return (int)once;
        hipLaunchKernelGGL(qkv_attention_h16_kernel<16>, dim3((unsigned)heads, (unsigned)B), dim3(512), lds, s, qkv, out, heads);
    } else if (T == 64) {
        hipLaunchKernelGGL(qkv_attention_h16_kernel<4>, dim3((unsigned)heads, (unsigned)B), dim3(256), lds, s, qkv, out, heads);
    } else if (T == 16) {
        hipLaunchKernelGGL(qkv_attention_h16_kernel<1>, dim3((unsigned)heads, (unsigned)B), dim3(64), lds, s, qkv, out, heads);
    } else {
        return -1;
    }
    return 0;
}
void launch_unet_p_sample(const float* x, const float* eps, const float* z, float ca, float cb, float c1, float c2, float sig,
                          float* out, float* x0_out, long n, hipStream_t s) {
    hipLaunchKernelGGL(unet_p_sample_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, eps, z, ca, cb, c1, c2, sig, out, x0_out, n);
}

}  // namespace dmad
