/* dmad.h — C ABI of libdmad_hip.so, the MI355X (gfx950) engine behind the reference's Python
 * surfaces for the certified-smoothing hot path.
 *
 * The reference (cychomatica/Diffusion-Model-for-Audio-Defense) is 100 % Python on PyTorch and has
 * no FFI; the "plugin API" of this path is a set of Python call sites.  Each entry point below
 * names the reference call it replaces (paths relative to the reference repo root).  Device pointers
 * in, device pointers out, an explicit hipStream_t, no hidden allocation on the data path after
 * dmad_create() / dmad_finalize_weights() (the two diagnostic hooks at the end create HIP events), no torch types.  Every function returns 0 on success or a negative dmad_status; dmad_last_error()
 * gives the message (thread-local).  One engine per process per GPU; calls on one engine must come
 * from one thread at a time, and work given to one engine is ordered by the stream it is given on: an
 * engine owns ONE set of work buffers and per-step tables (activations, step embeddings, the recheck
 * queue), so two streams must not have calls on the same engine in flight together — switch streams
 * only after synchronising the previous one, or create one engine per stream.
 */
#ifndef DMAD_H
#define DMAD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmad_engine dmad_engine;
typedef void* dmad_stream;      /* hipStream_t (torch.cuda.current_stream().cuda_stream) */

enum dmad_status {
    DMAD_OK = 0,
    DMAD_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    DMAD_ERR_STATE = -2,        /* weights not finalised, batch larger than max_batch, ... */
    DMAD_ERR_HIP = -3           /* a HIP runtime call failed */
};

enum dmad_precision {
    DMAD_BF16 = 0,              /* WaveNet on the 16-bit MFMA path alone (operands: dmad_half_type, fp32 accumulate); mel +
                                 * classifier fp32.  (The name is historical: the operand format is half_type's.) */
    DMAD_FP32 = 1,              /* everything on the exact-fp32 matrix path (parity mode) */
    DMAD_EXACT = 2              /* both WaveNet paths resident: the 16-bit path for throughput + re-evaluation, on the
                                 * split-f16 and exact-fp32 tiers, of every Monte Carlo sample whose 16-bit top-2 logit margin is
                                 * below the recheck bound, so that the vote counts of dmad_smooth_votes are the fp32 path's
                                 * up to the measured bound documented at dmad_set_mode (an empirical guarantee) */
};

/* Operand format of the 16-bit MFMA WaveNet path (DMAD_BF16 / DMAD_EXACT engines), fp32 accumulation either way. */
enum dmad_half_type {
    DMAD_HALF_BF16 = 0,         /* bfloat16: 8-bit significand (the format BASELINE.json's configuration names) */
    DMAD_HALF_F16 = 1           /* IEEE half: 11-bit significand, 8x smaller rounding error at the same MFMA rate; the
                                 * network's activations are O(1), far inside the half range */
};

/* Run-time mode of a DMAD_EXACT engine (the other two precisions have exactly one mode). */
enum dmad_mode {
    DMAD_MODE_FAST = 0,         /* 16-bit WaveNet, no recheck (what a DMAD_BF16 engine does) */
    DMAD_MODE_EXACT_VOTES = 1,  /* 16-bit WaveNet + margin-triggered recheck (split-f16 tier, then exact fp32) inside
                                 * dmad_smooth_votes; the waveform-returning entry points run the tier dmad_set_waveform_tier
                                 * selects (default: split-f16, fp32-grade) */
    DMAD_MODE_FP32 = 2          /* every WaveNet evaluation on the exact-fp32 path (what a DMAD_FP32 engine does) */
};

/* configs/config.json (wavenet_config + diffusion_config) as read by
 * diffusion_models/diffwave_ddpm.py:395-411 create_diffwave_model(). */
typedef struct dmad_config {
    int32_t struct_size;        /* sizeof(dmad_config) of the header the caller was built with: dmad_create refuses any
                                 * other value, so a caller of an older revision fails with a message instead of having
                                 * fields read past the end of its struct */
    int32_t res_channels;       /* 256 (the only supported value)       */
    int32_t skip_channels;      /* 256                                   */
    int32_t num_res_layers;     /* <= 64; 36 in the reference            */
    int32_t dilation_cycle;     /* <= 12 (max dilation 2048)             */
    int32_t embed_dim_in;       /* 128 */
    int32_t embed_dim_mid;      /* 512 */
    int32_t embed_dim_out;      /* 512 */
    int32_t clip_len;           /* 16000; must be a multiple of 128      */
    int32_t max_batch;          /* clips resident per pass               */
    int32_t num_classes;        /* 10  */
    int32_t precision;          /* enum dmad_precision                   */
    int32_t with_classifier;    /* 1: VGG19_bn + mel front-end buffers   */
    int32_t recheck_batch;      /* DMAD_EXACT: clips per pass of the recheck tiers (0 = 32; clamped to max_batch; the
                                 * Python engine and bench.py pass 64) */
    int32_t half_type;          /* enum dmad_half_type                   */
    int32_t with_wavenet;       /* 1: WaveNet workspace (residual streams, gate store: 330 MB per clip of max_batch on the
                                 * 16-bit path).  0: an engine for the spec-domain purifier / classifier only (BASELINE C5): no
                                 * WaveNet weights can be loaded, none of that memory is held */
} dmad_config;

int dmad_create(const dmad_config* cfg, dmad_engine** out);
void dmad_destroy(dmad_engine* e);
const char* dmad_last_error(void);
const char* dmad_version(void);
/* Non-fatal findings of the last dmad_finalize_weights() on this thread ("" if none): e.g. folded WaveNet weights that leave the
 * f16 normal range on an f16 engine (subnormals below 6.1e-5 lose significant bits, values above 65504 become inf). */
const char* dmad_last_warning(void);

/* Weights arrive as FOLDED fp32 host arrays (weight-norm and eval BatchNorm already folded by the
 * Python loader, exactly as the reference's modules compute them on every forward:
 * DiffWave_Unconditional/WaveNet.py:27-28,66-72; models/vgg.py:69-81).  Names:
 *   init.w[256] init.b[256]  fc_t1.w[512,128] fc_t1.b  fc_t2.w[512,512] fc_t2.b
 *   fc_t.{n}.w[256,512] fc_t.{n}.b   dil.{n}.w[512,256,3] dil.{n}.b   res.{n}.w[256,256] res.{n}.b
 *   skip.{n}.w[256,256] skip.{n}.b   f0.w[256,256] f0.b   f2.w[256] f2.b[1]
 *   vgg.conv{i}.w[cout,cin,3,3] vgg.conv{i}.scale[cout] vgg.conv{i}.shift[cout]  (i = 0..15)
 *   vgg.fc{j}.w[out,in] vgg.fc{j}.b   (j = 0..2)
 * or, instead of the vgg.* set, ResNeXt29 8x64d (models/resnext.py:23-142; bottleneck i = 3 * stage + k, i = 0..8):
 *   rx.conv1.w[64,1,3,3]  rx.b{i}.reduce.w[D,cin]  rx.b{i}.conv.w[D,D/8,3,3]  rx.b{i}.expand.w[cout,D]
 *   rx.b{i}.short.w[cout,cin] (where cin != cout), each with .scale[M] .shift[M];  rx.fc.w[classes,1024] rx.fc.b
 * dmad_finalize_weights() packs whichever complete set (WaveNet, classifier) has been loaded and is not
 * packed yet into the MFMA/LDS layouts and uploads it; it may be called once per set. */
int dmad_load_weight(dmad_engine* e, const char* name, const float* host, const int64_t* shape, int32_t ndim);
int dmad_finalize_weights(dmad_engine* e);

/* eps = WaveNet((x_t, t * ones))  — DiffWave.model(...) at diffwave_ddpm.py:157-158,169-170,177-178
 * (WaveNet_Speech_Commands.forward, WaveNet.py:164-172).  x_t, eps: device fp32 [B][clip_len]. */
int dmad_wavenet_eps(dmad_engine* e, const float* x_t, int32_t t, int32_t B, float* eps, dmad_stream s);

/* x0_hat = c_a * x_t - c_b * eps(x_t, t)  — DiffWave.one_shot_denoise, diffwave_ddpm.py:174-182,195-205.
 * c_a = sqrt(1/Alpha_bar)[t], c_b = sqrt(1/Alpha_bar - 1)[t] are computed by the caller in fp32
 * exactly as the reference does (tables on the CPU, then indexed). */
int dmad_one_shot(dmad_engine* e, const float* x_t, int32_t t, float c_a, float c_b, int32_t B, float* x0, dmad_stream s);

/* One reverse step  x <- (x - c_eps * eps(x, t)) / c_div (+ c_sig * z)  — DiffWave.compute_coefficients
 * + the loop body of DiffWave._reverse, diffwave_ddpm.py:95-102,143-164.  z: device fp32 [B][L] noise
 * supplied by the caller, or NULL to draw Philox N(0,1) keyed (seed, sample0 + b, stream = 1 + t);
 * pass c_sig = 0 for the t == 0 step. */
int dmad_ddpm_step(dmad_engine* e, float* x, int32_t t, float c_eps, float c_div, float c_sig, const float* z,
                   uint64_t seed, uint64_t sample0, int32_t B, dmad_stream s);

/* x_t = c_a * x0 + c_b * z  — DiffWave._diffusion, diffwave_ddpm.py:49-73 (z as above, stream 0xD1FF). */
int dmad_diffuse(dmad_engine* e, const float* x0, float c_a, float c_b, const float* z, uint64_t seed,
                 uint64_t sample0, int32_t B, float* x_t, dmad_stream s);

/* spec = AmplitudeToDB('power')(MelSpectrogram(n_fft=2048, hop=512, n_mels=32, slaney)(x))
 * — the Wave2Spect transform built at certified_robustness_eval.py:85-87.  x: [B][clip_len],
 * spec: [B][32 mel][32 frames] fp32. */
int dmad_mel_db(dmad_engine* e, const float* x, int32_t B, float* spec, dmad_stream s);

/* The two stages of that transform on their own (for callers that keep torchaudio's two-module Compose):
 * mel = MelSpectrogram(...)(x): power mel spectrogram [B][32][32]; y = AmplitudeToDB('power')(x) elementwise. */
int dmad_mel_power(dmad_engine* e, const float* x, int32_t B, float* mel, dmad_stream s);
int dmad_power_to_db(dmad_engine* e, const float* x, int64_t n, float* y, dmad_stream s);

/* DiffWave.forward = _diffusion + _reverse in one call (diffwave_ddpm.py:36-104) with on-device Philox noise keyed
 * (seed, sample0 + row; stream 0xD1FF for the diffusion draw, 1 + t for reverse step t):
 *   x <- c_a * x0 + c_b * z;  for t = t_star-1 .. 0:  x <- (x - c_eps[t] * eps(x, t)) / c_div[t] (+ c_sig[t] * z_t, t > 0).
 * c_a = sqrt(Alpha_bar[t*-1]), c_b = sqrt(1 - Alpha_bar[t*-1]); c_eps / c_div / c_sig: HOST fp32 arrays of t_star
 * entries ((1 - Alpha[t]) / sqrt(1 - Alpha_bar[t]), sqrt(Alpha[t]), Sigma[t]), computed by the caller from the fp32
 * tables as the reference does.  x0, out: device fp32 [B][clip_len] (may alias). */
int dmad_ddpm_purify(dmad_engine* e, const float* x0, int32_t t_star, float c_a, float c_b, const float* c_eps, const float* c_div,
                     const float* c_sig, uint64_t seed, uint64_t sample0, int32_t B, float* out, dmad_stream s);

/* Improved-Diffusion UNet purifier on 1x32x32 mel spectrograms (the reference's configuration C5:
 * diffusion_models/improved_diffusion_ddpm.py:64-93 -> improved_diffusion/script_util.py:11-34,100-131).  Weights: the
 * reference's UNetModel state dict, names prefixed "un." (un.time_embed.0.weight, un.input_blocks.5.0.in_layers.2.weight,
 * ...), loaded with dmad_load_weight + dmad_finalize_weights like the other sets.
 * eps = UNetModel.forward(x_t, t * ones)  — improved_diffusion/unet.py:453-477.  x_t, eps: device fp32 [B][32][32]. */
int dmad_unet_eps(dmad_engine* e, const float* x_t, int32_t t, int32_t B, float* eps, dmad_stream s);

/* One GaussianDiffusion.p_sample step in place on x (gaussian_diffusion.py:232-257,331-387; epsilon prediction, fixed
 * variance, clip_denoised):  x0 = clamp(c_a * x - c_b * eps(x, t), -1, 1);  x <- c_1 * x0 + c_2 * x + c_sig * z.
 * c_a = sqrt_recip_alphas_cumprod[t], c_b = sqrt_recipm1_alphas_cumprod[t], c_1 / c_2 = posterior_mean_coef1/2[t],
 * c_sig = exp(0.5 * log_variance[t]) (0 at t == 0), all computed by the caller from the float64 tables.  z: device
 * fp32 [B][32][32] (the reference's randn_like draws) or NULL for on-device Philox noise keyed (seed, sample0 + row).
 * x0_out: optional [B][32][32] (pred_xstart). */
int dmad_unet_p_sample(dmad_engine* e, float* x, int32_t t, float c_a, float c_b, float c_1, float c_2, float c_sig, const float* z,
                       uint64_t seed, uint64_t sample0, int32_t B, float* x0_out, dmad_stream s);

/* logits = classifier(spec)  — VGG.forward, audio_models/ConvNets_SpeechCommands/models/vgg.py:48-52, or
 * CifarResNeXt.forward, models/resnext.py:133-142, whichever weight set was loaded.
 * spec: [B][32][32] fp32, logits: [B][num_classes] fp32. */
int dmad_classify(dmad_engine* e, const float* spec, int32_t B, float* logits, dmad_stream s);

/* The classifier's tiers.  Engines of precision DMAD_BF16 / DMAD_EXACT that hold ResNeXt29 — the default classifier of the
 * reference's certification script (certified_robustness_eval.py:57; models/resnext.py:23-142) — also hold a 16-BIT TIER of it:
 * every conv (1x1 reduce / expand / shortcut, the grouped 3x3) on f16 operands with fp32 accumulation, the eval-mode BatchNorm
 * scale folded into the f16 weights, shift / shortcut add / ReLU in fp32, maps kept as f16 between the convs; average pool and
 * the linear head stay fp32.  It is the classifier of DMAD_MODE_FAST (the vote loops' pass and the mode-default path of
 * dmad_eval_samples there).  Measured on the calibrated synthetic stand-in, its leader-difference error is 0.08-0.16 against
 * 0.016-0.030 for the f16 WaveNet in front of the fp32 classifier (profiles/r05b_resnext29_error_attribution.json) — a recheck bound
 * covering it would send a quarter of the samples to the recheck tiers — so since round 5 the first pass of the exact-vote mode runs
 * the classifier's SPLIT-F16 TIER instead (DMAD_EXACT engines): every conv on split-f16 operands (three f16 MFMAs per product, ~22
 * significant bits; BatchNorm scale / shift, shortcut add and ReLU in the fp32 epilogue), fp32-grade at a third of the fp32 tier's
 * time.  The split-f16 WaveNet recheck tier is paired with the same classifier tier (together 2.2e-4 from the all-fp32 logits, under
 * tau2 = 1e-3); the exact-fp32 recheck tier, dmad_query_logits and dmad_classify use the fp32 matrix cores, so a sample that reaches the
 * last tier carries the fp32 path's logits bit for bit.  dmad_classify_tier evaluates an explicit tier (0: fp32, 1: 16-bit, 2: split-f16; VGG19_bn has one tier and is
 * served on fp32 either way) — test / measurement hook. */
int dmad_classify_tier(dmad_engine* e, const float* spec, int32_t B, int32_t tier, float* logits, dmad_stream s);

/* The Monte Carlo loop of RobustCertificate.smooth_predict (+ forward, compute_t_star's result),
 * robustness_eval/certified_robust.py:17-31,33-67:  for samples i in [sample0, sample0 + n):
 *   x_in = sqrt(alpha_bar_star) * (clip + delta_i);  x0 = one_shot(x_in, t);  logits = classifier(mel_db(x0));
 *   counts[argmax logits] += 1.
 * clip: device fp32 [clip_len].  delta: device fp32 [n][clip_len] (the reference's CPU torch.normal
 * draws, parity mode) or NULL for on-device Philox noise keyed (seed, sample index).  counts: device
 * int64[num_classes], ACCUMULATED into (zero it first).  logits_out: optional [n][num_classes].
 * batch <= max_batch.  With with_classifier == 0, x0_out (optional, [n][clip_len]) receives the
 * purified clips and the caller classifies them. */
int dmad_smooth_votes(dmad_engine* e, const float* clip, float sigma, float sqrt_alpha_bar_star, int32_t t,
                      float c_a, float c_b, int64_t n, int32_t batch, uint64_t seed, uint64_t sample0,
                      const float* delta, int64_t* counts, float* logits_out, float* x0_out, dmad_stream s);

/* DMAD_EXACT engines.  dmad_set_mode selects the dmad_mode (default DMAD_MODE_EXACT_VOTES).  dmad_set_recheck_margin sets
 * the bound tau: a sample whose 16-bit logits have (largest - second largest) < tau (or any NaN) does not vote from the 16-bit
 * logits; its global sample index is queued and the sample is re-evaluated from the SAME noise (Philox key (seed, index),
 * or its row of `delta`) on the higher tiers, and that result votes (and replaces its row of logits_out / x0_out).
 * With tau >= the largest error the 16-bit path makes on a logit DIFFERENCE AGAINST THE EXACT LEADER (E = max_j |e_j - e_i|,
 * e = 16-bit minus exact logits, i = the exact arg-max: a 16-bit leader j != i with margin >= tau would need e_j - e_i >= tau)
 * the counts equal the fp32 path's exactly (robustness_eval/certified_robust.py:59-65 is an arg-max: it only depends on the
 * order of the logits).  Defaults: 0.034 for f16 operands (measured E = 0.0244 over 36 864 samples), 0.30 for bf16 (0.207).
 *
 * The queued samples pass through two tiers.  Tier 2 is the fp32 pipeline on SPLIT-f16 operands: every fp32 value is kept
 * as hi = f16(x), lo = f16((x - hi) * 2^11) and every product is three f16 MFMAs (hi*hi + (hi*lo + lo*hi) * 2^-11, fp32
 * accumulate) — about 22 significant bits at several times the fp32 matrix rate.  It settles every queued sample whose
 * margin exceeds ITS error bound tau2 (dmad_set_recheck_margin2; default 1e-3, tau2 < 0 switches the tier off); the rest
 * (margins inside tau2) are evaluated on the exact-fp32 path, tier 3.  dmad_recheck_stats: samples voted, samples that
 * left the 16-bit pass, samples that reached the fp32 path.  dmad_wavenet_eps_path evaluates the eps-network on an
 * explicit path (0: the mode's default, 1: exact fp32, 2: split-f16) — test / measurement hook for the tiers. */
int dmad_set_mode(dmad_engine* e, int32_t mode);
/* Which WaveNet tier the WAVEFORM-returning entry points of a DMAD_EXACT engine run in DMAD_MODE_EXACT_VOTES — dmad_wavenet_eps,
 * dmad_one_shot, dmad_ddpm_step, dmad_ddpm_purify and the purifier inside dmad_query_logits (DiffWave.forward / one_shot_denoise /
 * compute_eps_t and AcousticSystem's query path, diffusion_models/diffwave_ddpm.py:36-47,166-182): 2 = the split-f16 tier (the DEFAULT:
 * eps within 8e-5 of the reference's, the fp32 tolerance class, at ~3.6 x the cost of the 16-bit path), 1 = exact fp32, 0 = the 16-bit
 * path (4e-3 with f16 operands).  Only the vote loop has a margin-triggered recheck, so these surfaces get their accuracy from the tier
 * itself.  DMAD_MODE_FAST / DMAD_MODE_FP32 keep their meaning (16-bit / fp32 everywhere); DMAD_BF16 / DMAD_FP32 engines have one path.
 * Tolerance delivered per surface: the vote counts of dmad_smooth_votes — exact (see below); its x0_out / logits_out rows — the tier the
 * row voted on; the entry points above — this setting. */
int dmad_set_waveform_tier(dmad_engine* e, int32_t tier);
int dmad_set_recheck_margin(dmad_engine* e, float tau);
int dmad_set_recheck_margin2(dmad_engine* e, float tau2);
int dmad_recheck_stats(dmad_engine* e, int64_t* samples, int64_t* rechecked, int64_t* rechecked_fp32, int32_t reset);
int dmad_wavenet_eps_path(dmad_engine* e, const float* x_t, int32_t t, int32_t B, int32_t path, float* eps, dmad_stream s);

/* The forward of RobustCertificate.smooth_predict's loop body (certified_robust.py:46-56) for an explicit LIST of Monte Carlo
 * samples on an explicit WaveNet path — the audit of the exact-vote mode (RobustCertificate.certify(audit=k): k samples that
 * voted on the 16-bit tier are re-evaluated on a higher one) and the measurement tools' hook:  row i of logits_out [n][num_classes]
 * / x0_out [n][clip_len] (either optional) is sample idx[i] (device int64, GLOBAL sample indices; noise = Philox key
 * (seed, idx[i]) or row idx[i] - sample0 of `delta`), evaluated on path 0 (the mode's default), 1 (exact fp32) or 2 (split-f16).
 * Nothing votes, no queue is touched. */
int dmad_eval_samples(dmad_engine* e, const float* clip, float sigma, float sqrt_alpha_bar_star, int32_t t, float c_a, float c_b,
                      uint64_t seed, uint64_t sample0, const float* delta, const int64_t* idx, int64_t n, int32_t path, float* logits_out,
                      float* x0_out, dmad_stream s);

/* Measurement hook (tools/gpu_error_attribution.py): switch single roundings of the 16-bit path on INSIDE the fp32-grade split-f16
 * tier, to attribute the 16-bit tier's logit error to its sources.  masks[0..3] act on the tier's dilated-conv, res-conv, skip and
 * final_conv.0 GEMM launches: bit 0 = weights rounded to f16, bit 1 = the MFMA eats f16(activation) while the stored value keeps
 * its 22 bits, bit 2 = the launch's split-format output (gate / residual stream) is stored as f16; masks[4] != 0 = the init
 * conv's output is stored as f16.  All zero (the default) = the product tier. */
int dmad_debug_rounding(dmad_engine* e, const int32_t masks[5]);

/* BASELINE configuration C5: the Monte Carlo vote loop with the SPEC-domain purifier (Improved-Diffusion UNet on 1x32x32 mel
 * spectrograms) in place of the waveform purifier.  The reference has no working composite for it
 * (diffusion_models/improved_diffusion_ddpm.py:53-59 discards its reverse chain), so the loop is DEFINED here as what its parts
 * are for — randomized smoothing in the input domain (certified_robust.py:46-48, no denoiser => no sqrt(alpha_bar*) scale), then
 * AcousticSystem's defense_type = 'spec' order (acoustic_system.py:40-49):  for samples i in [sample0, sample0 + n):
 *   x_i = clip + sigma * delta_i                    (Philox keyed (seed, i), stream 0)
 *   s   = mel_dB(x_i) ; s0 = 2 (s - lo) / (hi - lo) - 1            (melspec_standardize, sc09_spectrogram_dataset.py:62-72)
 *   s_t = q_a * s0 + q_b * z                        (q_sample at t = t_star, gaussian_diffusion.py:188-206; Philox stream 0x5BEC)
 *   for t = t_star .. 0:  s_t <- p_sample(s_t, t)   (dmad_unet_p_sample with c_a/c_b/c_1/c_2/c_sig[t]: HOST arrays of t_star + 1 entries)
 *   logits = classifier((s_0 + 1)(hi - lo) / 2 + lo) ; counts[argmax] += 1.
 * q_a / q_b = sqrt_alphas_cumprod[t_star] / sqrt_one_minus_alphas_cumprod[t_star].  counts: device int64[num_classes],
 * accumulated; logits_out [n][num_classes] and spec_out [n][32][32] (the purified dB spectrograms) optional. */
int dmad_spec_smooth_votes(dmad_engine* e, const float* clip, float sigma, int32_t t_star, float q_a, float q_b, const float* c_a,
                           const float* c_b, const float* c_1, const float* c_2, const float* c_sig, float mel_lo, float mel_hi, int64_t n,
                           int32_t batch, uint64_t seed, uint64_t sample0, int64_t* counts, float* logits_out, float* spec_out, dmad_stream s);

/* The UNet's tiers.  Engines of precision DMAD_BF16 / DMAD_EXACT hold, beside the exact-fp32 UNet, a 16-BIT TIER of it: every
 * conv / 1x1 (unet.py:107-252) on f16 operands with fp32 accumulation (v_mfma_f32_16x16x32_f16); the hidden state exists as f16 maps
 * only, GroupNorm statistics are sums over the f16-rounded outputs taken in the producing GEMM's epilogue (evaluated as E[x^2] - mean^2
 * in fp32), softmax, bias and residual sums in fp32 (measured: 4e-3 of max|eps| per evaluation).  DMAD_EXACT engines also hold a
 * SPLIT-F16 MIDDLE TIER: the fp32 pipeline (fp32 maps, GroupNorm, softmax, residual sums) with every conv / 1x1 on split-f16 operands
 * (three f16 MFMAs per product, ~22 significant bits: fp32-grade at several times the fp32 matrix rate).
 * dmad_unet_eps / dmad_unet_p_sample / dmad_spec_query_logits — map- and logit-returning surfaces without a recheck — follow
 * dmad_set_waveform_tier like the waveform-returning ones: on an exact-vote engine the split-f16 tier by default (within 1e-4 of the
 * reference fixtures at 2.2 x the fp32 rate), the exact-fp32 UNet with tier 1 (fp32) or in DMAD_MODE_FP32, the 16-bit tier with tier 0
 * or in DMAD_MODE_FAST (DMAD_FP32 engines have only the fp32 one).  In DMAD_MODE_EXACT_VOTES dmad_spec_smooth_votes runs every sample's chain on the 16-bit tier, queues the samples whose
 * top-2 logit margin is below tau_spec (dmad_set_spec_recheck_margin; default 0.13 = 1.5 x the largest leader-difference error (0.084;
 * Gaussian scale 0.020) of the 16-bit chain measured on 6 144 samples of the calibrated synthetic stand-in) and
 * re-runs their WHOLE chain from the same Philox keys on the split-f16 tier; a sample whose margin is still below tau_spec2
 * (dmad_set_spec_recheck_margin2, default 5e-4 = 2.2 x its measured error; < 0: no middle tier) goes on to the exact-fp32 UNet.  An EMPIRICAL guarantee like the
 * waveform loop's (dmad_set_mode), and like it a property of the WEIGHTS: calibrate (Engine.calibrate_spec_recheck) before certifying
 * with other checkpoints.  dmad_spec_recheck_stats: samples voted by dmad_spec_smooth_votes, samples whose chain left the 16-bit tier;
 * dmad_spec_recheck_stats2: + those that reached the exact-fp32 UNet. */
/* dmad_unet_eps on an explicit tier (0: exact fp32, 1: 16-bit, 2: split-f16) — test / measurement hook (DMAD_ERR_STATE for a tier the
 * engine's precision does not hold). */
int dmad_unet_eps_tier(dmad_engine* e, const float* x_t, int32_t t, int32_t B, int32_t tier, float* eps, dmad_stream s);
int dmad_set_spec_recheck_margin(dmad_engine* e, float tau);
int dmad_spec_recheck_stats(dmad_engine* e, int64_t* samples, int64_t* rechecked, int32_t reset);
int dmad_set_spec_recheck_margin2(dmad_engine* e, float tau2);
int dmad_spec_recheck_stats2(dmad_engine* e, int64_t* samples, int64_t* rechecked, int64_t* rechecked_fp32, int32_t reset);
/* The chain of dmad_spec_smooth_votes for an explicit LIST of Monte Carlo samples on an explicit UNet tier (0: exact fp32, 1: 16-bit,
 * 2: split-f16; the classifier is the fp32 one on tiers 0 and 2): row i of logits_out [n][num_classes] / spec_out [n][32][32] (either optional) is sample
 * idx[i] (device int64, GLOBAL indices: every draw of the row is Philox-keyed by it).  Nothing votes, no queue is touched — the hook
 * behind the calibration of tau_spec for the resident weights (Engine.calibrate_spec_recheck) and RobustCertificate.certify(audit=k)
 * on the spec-domain loop. */
int dmad_spec_eval_samples(dmad_engine* e, const float* clip, float sigma, int32_t t_star, float q_a, float q_b, const float* c_a,
                           const float* c_b, const float* c_1, const float* c_2, const float* c_sig, float mel_lo, float mel_hi, uint64_t seed,
                           const int64_t* idx, int64_t n, int32_t tier, float* logits_out, float* spec_out, dmad_stream s);

/* Batched query of the whole system for the gradient-free attack drivers: EOT.forward evaluates
 * model(x_batch.repeat(EOT_batch_size, 1, 1)) EOT_num_batches times (robustness_eval/_EOT.py:30-64; callers
 * black_box_attack.py:186-220, _NES.py:15-55) where model = AcousticSystem(classifier, transform, defender)
 * (acoustic_system.py:27-51).  One call does all of it:  row i = r * B + b  (r < repeats, b < B) is clip x[b];
 *   sampler 0: no wave defender;  1: DiffWave.forward (diffusion + t_star reverse steps, diffwave_ddpm.py:36-104;
 *   coefficients as for dmad_ddpm_purify);  2: one_shot_denoise at t = t_star - 1 (c_a, c_b as for dmad_one_shot);
 *   then mel dB -> classifier.  Noise of row i is Philox keyed (seed, sample0 + i): a row's logits do not depend on how
 *   the rows are batched.  x: device fp32 [B][clip_len]; logits: [repeats * B][num_classes]; decisions: optional
 *   int32 [repeats * B] arg-max (first maximum wins). */
int dmad_query_logits(dmad_engine* e, const float* x, int32_t B, int32_t repeats, int32_t sampler, int32_t t_star, float c_a, float c_b,
                      const float* c_eps, const float* c_div, const float* c_sig, uint64_t seed, uint64_t sample0, float* logits,
                      int32_t* decisions, dmad_stream s);

/* The same batched query for AcousticSystem(defense_type = 'spec') (acoustic_system.py:40-49: transform, THEN the defender on the
 * spectrogram): row i = r * B + b is clip x[b] through  mel dB -> standardise -> q_sample(t_star) -> t_star + 1 p_sample steps ->
 * un-standardise -> classifier  — the chain of dmad_spec_smooth_votes without the smoothing noise, every draw of row i Philox-keyed
 * (seed, sample0 + i) (q_sample: stream 0x5BEC, p_sample at t: stream 0x0E70 + t), so a row's logits do not depend on how the rows are
 * batched.  The UNet runs the tier of the map-returning surfaces (dmad_set_waveform_tier: split-f16 by default on an exact-vote engine),
 * the classifier the fp32 one: a query hands logits back and has no recheck.  Coefficients as for dmad_spec_smooth_votes (HOST arrays of t_star + 1
 * entries).  logits: [repeats * B][num_classes]; decisions: optional int32 [repeats * B]. */
int dmad_spec_query_logits(dmad_engine* e, const float* x, int32_t B, int32_t repeats, int32_t t_star, float q_a, float q_b, const float* c_a,
                           const float* c_b, const float* c_1, const float* c_2, const float* c_sig, float mel_lo, float mel_hi, uint64_t seed,
                           uint64_t sample0, float* logits, int32_t* decisions, dmad_stream s);

/* counts[argmax_c logits[b][c]] += 1 (first maximum wins) — certified_robust.py:59-65. */
int dmad_vote(dmad_engine* e, const float* logits, int32_t B, int64_t* counts, dmad_stream s);

/* Test hooks: raw Philox4x32-10 words / N(0,1) draws of the generator used above. */
int dmad_philox_raw(dmad_engine* e, uint64_t seed, uint64_t sample, uint32_t stream, uint32_t nblocks, uint32_t* out, dmad_stream s);
int dmad_philox_normal(dmad_engine* e, uint64_t seed, uint64_t sample0, uint32_t stream, int32_t B, float* z, dmad_stream s);

/* Timing hook for bench.py: runs `iters` launches of residual layer `layer` (bf16 path) on the resident
 * buffers between two HIP events recorded on `s` and returns the average milliseconds per launch. */
int dmad_time_layer(dmad_engine* e, int32_t layer, int32_t B, int32_t iters, float* ms_per_launch, dmad_stream s);

/* Live timing of the dominant kernel (the fused residual layer, wn_layer_bf16) for bench.py's roofline:
 * after dmad_profile_layers(e, max_launches > 0) every non-final layer launch of the bf16 path is
 * bracketed by a HIP event pair recorded on the launch stream (until max_launches pairs are used);
 * dmad_profile_read() waits for the last pair, returns the summed elapsed milliseconds and the number of
 * launches, and switches the bracketing off. */
int dmad_profile_layers(dmad_engine* e, int32_t max_launches);
int dmad_profile_read(dmad_engine* e, float* total_ms, int32_t* launches);
/* The same for the launches of the tail kernel (wn_final: skip GEMM over the gate store + final convs) bracketed since
 * dmad_profile_layers(); call it BEFORE dmad_profile_read (which switches the bracketing off). */
int dmad_profile_read_final(dmad_engine* e, float* total_ms, int32_t* launches);

/* Test hook of the f16 conv-GEMM family (csrc/gemm_h16.hip: the kernels behind the UNet's and ResNeXt29's 16-bit tiers), standalone —
 * no engine state is read.  NHWC convolution  out[n][g*M + m] = relu?( sum_tap sum_k w[g][tap][m][k] x[pixel(n, tap)][g*K + k] + bias[g*M + m]
 * + res[n][g*M + m] )  with f16 operands and fp32 accumulation.  x: device f16 [B][H][H][ldx]; x2 / ksplit: optional second input
 * map holding channels [ksplit, K) (the UNet's th.cat read in place; x then has ksplit channels per pixel); w: device f16
 * [groups][taps][M][K] (taps 9 = 3x3 with zero padding 1, or 1); bias: fp32 [groups * M] or NULL; res16: optional f16 residual
 * [N][groups * M]; out32 / out16: fp32 map and / or f16 twin [N][groups * M], N = B * Ho * Ho, Ho = (H - 1) / stride + 1.
 * M and K are per group.  Which of the family's kernels serves a shape is the launcher's choice (the product's). */
int dmad_conv_h16(const uint16_t* x, const uint16_t* x2, int32_t ksplit, const uint16_t* w, const float* bias, const uint16_t* res16,
                  int32_t B, int32_t H, int32_t M, int32_t K, int32_t taps, int32_t stride, int32_t groups, int32_t relu,
                  float* out32, uint16_t* out16, dmad_stream s);
/* The UNet's Upsample (F.interpolate(scale_factor=2, mode="nearest") + 3x3 conv, unet.py:72-79) as ONE launch of the family's
 * slice-resident form: x_half is the HALF-resolution f16 map [B][H/2][H/2][K], the conv output is [B][H][H][M] (H even; stats: optional
 * GroupNorm statistics slab as below).  DMAD_ERR_STATE when the shape is not served by that form (fewer tiles than CUs, M % 256, maps wider
 * than 32 pixels ...: the product then materialises the x2 map and calls the plain conv).  Test hook like dmad_conv_h16. */
int dmad_conv_h16_up2(const uint16_t* x_half, const uint16_t* w, const float* bias, const uint16_t* res16, int32_t B, int32_t H, int32_t M, int32_t K,
                      float* out32, uint16_t* out16, float* stats, dmad_stream s);
/* The same with the consumer's GroupNorm statistics accumulated in the epilogue: stats [N / blk][groups * M / 4][2] fp32 = (sum, sum of
 * squares) of the f16-rounded outputs per block of blk = 64 pixels (16 when Ho * Ho == 16) and 4-channel quad — and the one-pass
 * GroupNorm32 + SiLU (+ scale-shift) that consumes them (nn.py:15-17, unet.py:186-199 on the 16-bit tier):
 * y = SiLU?(GroupNorm_32groups(cat(x, x2)) * gamma + beta [* (1 + ss[c]) + ss[C + c]]), x / x2 f16 NHWC maps [B][HW][c1 | C - c1] with
 * their slabs st / st2 (x2, st2 NULL: one map), y16 f16 or y32 fp32 [B][HW][C].  Test hooks like dmad_conv_h16. */
int dmad_conv_h16_stats(const uint16_t* x, const uint16_t* w, const float* bias, const uint16_t* res16, int32_t B, int32_t H, int32_t M, int32_t K,
                        int32_t taps, int32_t stride, uint16_t* out16, float* stats, dmad_stream s);
int dmad_groupnorm16_apply(const uint16_t* x, const float* st, const uint16_t* x2, const float* st2, int32_t c1, const float* gamma,
                           const float* beta, const float* ss, int32_t silu, int32_t B, int32_t HW, int32_t C, uint16_t* y16, float* y32,
                           dmad_stream s);

/* Test hooks of the split-f16 conv GEMM (csrc/gemm_f32.hip, gemm_x3_kernel in its NHWC form: the kernel behind the UNet's middle tier),
 * standalone.  dmad_split_f16: y = the split-f16 storage form of the n fp32 values x (n % 4 == 0; hi = f16(v), lo = f16((v - hi) 2^11), four
 * values per 16-byte chunk: the bytes of four floats; x == y allowed).  dmad_conv_x3: NHWC convolution of split-format operands — x
 * [B][H][H][groups * K] (or, dense only, x | x2 with ksplit channels in x), w [groups][taps][M][K] (taps 9 = 3x3 zero padding 1, or 1),
 * fp32 bias [groups * M], optional residual [N][groups * M] (fp32, or a split-format map with res_split), stride 1 / 2, optional ReLU —
 * every product as three f16 MFMAs; out [N][groups * M] fp32, or in the split format (out_split).  M and K are per group;
 * M % 128 == 0, K % 32 == 0 (ksplit % 32 == 0). */
int dmad_split_f16(const float* x, int64_t n, float* y, dmad_stream s);
int dmad_conv_x3(const float* x, const float* x2, int32_t ksplit, const float* w, const float* bias, const float* res, int32_t B, int32_t H,
                 int32_t M, int32_t K, int32_t taps, int32_t stride, int32_t groups, int32_t relu, int32_t out_split, int32_t res_split, float* out,
                 dmad_stream s);

/* Bytes of device memory held by the engine. */
int64_t dmad_device_bytes(const dmad_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* DMAD_H */
