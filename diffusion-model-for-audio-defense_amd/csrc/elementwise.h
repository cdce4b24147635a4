// Launchers of the small kernels in elementwise.hip.
#pragma once
#include "dmad_common.h"

namespace dmad {

void launch_philox_raw(uint64_t seed, uint64_t sample, uint32_t stream, uint32_t nblocks, uint32_t* out, hipStream_t s);
void launch_philox_normal(uint64_t seed, uint64_t sample0, uint32_t stream, float* z, int B, int L, hipStream_t s,
                          const long long* idx = nullptr);      // idx != nullptr: row b is sample idx[b] instead of sample0 + b
void launch_mc_noise_scale(const float* clip, const float* delta, float sigma, float scale, uint64_t seed, uint64_t sample0,
                           float* xt, int B, int L, hipStream_t s);
void launch_mc_noise_scale_idx(const float* clip, const float* delta, float sigma, float scale, uint64_t seed, uint64_t sample0,
                               const long long* idx, float* xt, int B, int L, hipStream_t s);
void launch_scatter_rows(const float* src, const long long* idx, long long base, float* dst, int B, int W, hipStream_t s);
void launch_repeat_rows(const float* x, float* out, int B, long row0, int nrows, int L, hipStream_t s);
void launch_vote_margin(const float* logits, int B, int C, unsigned long long* counts, float tau, long long sample_base,
                        const long long* idx, long long* list, unsigned long long* list_n, long long list_cap, int* pred_out, hipStream_t s);
void launch_embed_table(float t, const float* w1, const float* b1, const float* w2, const float* b2, const float* wt,
                        const float* bt, float* table, float* emb2_out, const float* b_res, float* epi_c, int NL, hipStream_t s);
void launch_lincomb(int op, const float* x, const float* y, const float* z, float c0, float c1, float c2, float* out, long n,
                    hipStream_t s);
void launch_wn_init_f32(const float* x, const float* w, const float* bias, const float* emb0, float* h, int B, int L, int LP,
                        hipStream_t s, bool split = false, bool hi_only = false);
void launch_scale(const float* x, float c, float* y, long n, hipStream_t s, bool split = false);
void launch_dot256(const float* f, const float* w, float bias, float* out, long N, hipStream_t s);
void launch_mel_pad(const float* x, float* xp, int B, int L, int LPm, hipStream_t s);
void launch_mel_power(const float* D, float* P, int ldd, int ldp, long rows, hipStream_t s);
void launch_mel_db(const float* M, float* spec, int B, int to_db, hipStream_t s);
void launch_power_to_db(const float* x, float* y, long n, hipStream_t s);
void launch_vgg_conv1(const float* in, const float* w, const float* scale, const float* shift, float* out, int B, hipStream_t s,
                      h16_t* out16 = nullptr);     // out / out16: fp32 map and / or its f16 twin (either may be null)
void launch_maxpool2_nhwc(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
void launch_avgpool_nhwc(const float* in, float* out, int B, int HW, int C, hipStream_t s);
void launch_vote(const float* logits, int B, int C, unsigned long long* counts, int* pred_out, hipStream_t s);
void launch_store_vec128(const float* host128, float* out, hipStream_t s);
void philox4x32_10_host(uint32_t c[4], uint32_t k0, uint32_t k1);

}  // namespace dmad
