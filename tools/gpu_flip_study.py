#!/usr/bin/env python3
"""Development / evidence: how often does the bf16 engine vote differently from the exact-fp32 engine?

For every (clip, sigma) the SAME Philox keys (seed, sample index) go through both engines' fused Monte Carlo loop
(dmad_smooth_votes) and the per-sample logits are compared: flip count, logit error, the top-2 margin histogram, and —
for a list of candidate recheck bounds tau — the fraction of samples whose bf16 margin is below tau (they would be
re-evaluated in fp32) and the flips that would survive (bf16 margin >= tau yet a different arg-max).
Writes gpurun_out/flip_study.json (+ .npz with the raw logits).

    N=4096 CLIPS=0,1,2 SIGMAS=0.25,0.5,1.0 python tools/gpu_flip_study.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth  # noqa: E402
from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams  # noqa: E402

N = int(os.environ.get('N', 4096))
CLIPS = [int(c) for c in os.environ.get('CLIPS', '0,1,2').split(',')]
SIGMAS = [float(s) for s in os.environ.get('SIGMAS', '0.25,0.5,1.0').split(',')]
CLASSIFIER = os.environ.get('CLASSIFIER', 'vgg19_bn')
FIRSTPASS = os.environ.get('FIRSTPASS', '0') == '1'      # '16-bit' leg = the exact-vote mode's first pass instead of the fast mode (differs for ResNeXt29: split-f16 vs f16 classifier)
TAUS = (0.01, 0.02, 0.03, 0.034, 0.04, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.5)
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)

hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
ab = hp['Alpha_bar']
wsd = synth.wavenet_state_dict(1234)
HALF = os.environ.get('HALF', 'f16')                      # operand format of the 16-bit path under study
eng = E.Engine(max_batch=256, precision=E.EXACT, recheck_batch=64, half_type=E.HALF_F16 if HALF == 'f16' else E.HALF_BF16)
eng.load_wavenet(wsd)
if CLASSIFIER == 'resnext29':
    eng.load_resnext29(synth.resnext29_state_dict(2929))
else:
    eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
engs = {'bf16': (eng, E.MODE_FAST), 'fp32': (eng, E.MODE_FP32)}      # 'bf16' = the 16-bit path (either operand format)
if os.environ.get('X3', '1') == '1':
    engs['x3'] = (eng, 'x3')             # every sample through the split-f16 tier: tau1 = inf (nothing votes from the 16-bit pass), tau2 = 0
TAU1, TAU2 = eng.recheck_margin, eng.recheck_margin2

report, raw = [], {}
for ci in CLIPS:
    clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
    for sigma in SIGMAS:
        abar_star = 1 / (1 + sigma ** 2)
        t = int(torch.abs(ab - abar_star).min(0, keepdim=True)[1].item())
        c_a, c_b = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])
        sc = float(torch.tensor(abar_star ** 0.5, dtype=torch.float32))
        lg, cnt, secs = {}, {}, {}
        for name, (e, mode) in engs.items():
            if mode == 'x3':
                e.set_mode(E.MODE_EXACT_VOTES); e.set_recheck_margin(1e30); e.set_recheck_margin2(0.0)
            else:
                e.set_mode(mode); e.set_recheck_margin(TAU1); e.set_recheck_margin2(TAU2)
            torch.cuda.synchronize()
            t0 = time.time()
            if name == 'bf16' and FIRSTPASS:      # the exact-vote loop's FIRST PASS (16-bit WaveNet + the classifier tier it runs there), nothing rechecked
                e.set_mode(E.MODE_EXACT_VOTES)
                l = torch.cat([e.eval_samples(clip, sigma, sc, t, c_a, c_b, torch.arange(i, min(N, i + 8192), device='cuda'), path=0, seed=1000 + ci)
                               for i in range(0, N, 8192)])
                c = torch.bincount(l.argmax(1), minlength=l.shape[1])
            else:
                c, l, _ = e.smooth_votes(clip, sigma, sc, t, c_a, c_b, N, seed=1000 + ci, sample0=0, want_logits=True)
            torch.cuda.synchronize()
            secs[name] = time.time() - t0
            lg[name], cnt[name] = l.cpu().numpy().astype(np.float64), c.cpu().tolist()
        b, f = lg['bf16'], lg['fp32']
        srt_b, srt_f = np.sort(b, 1), np.sort(f, 1)
        mb_, mf_ = srt_b[:, -1] - srt_b[:, -2], srt_f[:, -1] - srt_f[:, -2]
        flips = b.argmax(1) != f.argmax(1)
        err = np.abs(b - f)
        # what decides a flip is the error of logit DIFFERENCES, not of the logits: d_ij error for the two fp32 leaders
        top2 = np.argsort(f, 1)[:, -2:]
        rows = np.arange(N)
        dd = (b[rows, top2[:, 1]] - b[rows, top2[:, 0]]) - (f[rows, top2[:, 1]] - f[rows, top2[:, 0]])
        pair_err = np.abs((b[:, :, None] - b[:, None, :]) - (f[:, :, None] - f[:, None, :])).max((1, 2))
        e_ = b - f                       # what the recheck bound has to cover: the error of a difference AGAINST THE fp32 LEADER
        lead_err = np.abs(e_ - e_[rows, f.argmax(1)][:, None]).max(1)
        rec = {'half': HALF, 'classifier': CLASSIFIER, 'first_pass_of_exact_mode': FIRSTPASS, 'clip': ci, 'sigma': sigma, 't_star': t + 1, 'n': N, 'counts_bf16': cnt['bf16'], 'counts_fp32': cnt['fp32'],
               'flips': int(flips.sum()), 'logit_err_max': float(err.max()), 'logit_err_rms': float(np.sqrt((err ** 2).mean())),
               'top2_diff_err_max': float(np.abs(dd).max()), 'top2_diff_err_rms': float(np.sqrt((dd ** 2).mean())),
               'pair_diff_err_max': float(pair_err.max()), 'leader_diff_err_max': float(lead_err.max()),
               'logit_std_fp32': float(f.std()), 'margin_fp32_median': float(np.median(mf_)),
               'margin_bf16_hist_edges': [0, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 1e9],
               'margin_bf16_hist': np.histogram(mb_, bins=[0, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 1e9])[0].tolist(),
               'flip_margins_bf16': sorted(float(v) for v in mb_[flips]), 'flip_margins_fp32': sorted(float(v) for v in mf_[flips]),
               'tau': {str(tau): {'recheck_frac': float((mb_ < tau).mean()), 'surviving_flips': int((flips & (mb_ >= tau)).sum())} for tau in TAUS},
               'clips_per_s': {k: N / v for k, v in secs.items()}}
        if 'x3' in lg:
            xx = lg['x3']
            pe = np.abs((xx[:, :, None] - xx[:, None, :]) - (f[:, :, None] - f[:, None, :])).max((1, 2))
            ex = xx - f
            le = np.abs(ex - ex[rows, f.argmax(1)][:, None]).max(1)
            srt_x = np.sort(xx, 1)
            mx_ = srt_x[:, -1] - srt_x[:, -2]
            xflips = xx.argmax(1) != f.argmax(1)
            rec['x3'] = {'flips': int(xflips.sum()), 'logit_err_max': float(np.abs(xx - f).max()), 'pair_diff_err_max': float(pe.max()), 'leader_diff_err_max': float(le.max()),
                         'pair_diff_err_rms': float(np.sqrt((pe ** 2).mean())), 'flip_margins_x3': sorted(float(v) for v in mx_[xflips]),
                         'frac_below': {str(t): float((mx_ < t).mean()) for t in (1e-4, 3e-4, 1e-3, 2e-3, 5e-3)}, 'clips_per_s': N / secs['x3']}
            print('   x3 tier vs fp32:', json.dumps(rec['x3']), flush=True)
        report.append(rec)
        raw['bf16_c%d_s%g' % (ci, sigma)] = b.astype(np.float32)
        raw['fp32_c%d_s%g' % (ci, sigma)] = f.astype(np.float32)
        print(json.dumps({k: rec[k] for k in ('clip', 'sigma', 'flips', 'logit_err_max', 'top2_diff_err_max', 'pair_diff_err_max',
                                              'margin_fp32_median', 'tau', 'clips_per_s')}), flush=True)
        with open(os.path.join(OUT, 'flip_study_%s.json' % HALF), 'w') as fh:
            json.dump(report, fh, indent=1)
np.savez_compressed(os.path.join(OUT, 'flip_study_%s.npz' % HALF), **raw)
tot = sum(r['n'] for r in report)
print('TOTAL samples %d, flips %d, worst pair-difference error %.4f' % (tot, sum(r['flips'] for r in report),
                                                                        max(r['pair_diff_err_max'] for r in report)))
