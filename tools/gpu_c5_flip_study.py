#!/usr/bin/env python3
"""Evidence for the spec-domain loop's recheck bound (BASELINE C5): the UNet's 16-bit tier against the exact-fp32 UNet on the
SAME Philox keys — per-sample logits of the whole chain (mel -> standardise -> q_sample(t*) -> t* + 1 p_sample steps ->
classifier), flips, the leader-difference error statistic the bound has to cover, the margin distribution, and the time of
each tier.  Writes gpurun_out/c5_flip_study.json (+ .npz with the raw logits, readable by tools/fit_recheck_tail.py).

    N=2048 CLIPS=0,1,2 SIGMAS=0.5 T=25 python tools/gpu_c5_flip_study.py
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
from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion  # noqa: E402

N = int(os.environ.get('N', 2048))
CLIPS = [int(c) for c in os.environ.get('CLIPS', '0,1,2').split(',')]
SIGMAS = [float(s) for s in os.environ.get('SIGMAS', '0.5').split(',')]
T = int(os.environ.get('T', 25))
BATCH = int(os.environ.get('BATCH', 2048))         # the engine batch of bench.py's C5 leg
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)

eng = E.Engine(max_batch=BATCH, precision=E.EXACT, recheck_batch=0, with_wavenet=False)
eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321, calibrated='c5'))
pur = create_improved_diffusion(None, reverse_timestep=T, state_dict=synth.unet_state_dict(31), engine=eng)
coef = tuple(pur.purify_coefficients())
report, raw = [], {}
for ci in CLIPS:
    clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
    for sigma in SIGMAS:
        args = (clip, sigma) + coef + (-100.0, 38.22)
        lg, secs, cnt = {}, {}, {}
        for name, mode in (('h16', E.MODE_FAST), ('fp32', E.MODE_FP32)):
            eng.set_mode(mode)
            eng.spec_smooth_votes(*args, min(BATCH, N), seed=1)          # warm-up (per-step tables)
            torch.cuda.synchronize()
            t0 = time.time()
            c, l, _ = eng.spec_smooth_votes(*args, N, seed=7000 + ci, want_logits=True)
            torch.cuda.synchronize()
            secs[name] = time.time() - t0
            lg[name], cnt[name] = l.cpu().numpy().astype(np.float64), c.cpu().tolist()
        # the split-f16 middle tier on the same keys (every sample through dmad_spec_eval_samples, tier 2)
        torch.cuda.synchronize(); t0 = time.time()
        lx = eng.spec_eval_samples(clip, sigma, *coef, -100.0, 38.22, torch.arange(N, device='cuda'), tier=2, seed=7000 + ci)
        torch.cuda.synchronize(); secs['x3'] = time.time() - t0
        lg['x3'] = lx.cpu().numpy().astype(np.float64)
        b, f = lg['h16'], lg['fp32']
        rows = np.arange(N)
        e_ = b - f
        le = np.abs(e_ - e_[rows, f.argmax(1)][:, None]).max(1)
        sb, sf = np.sort(b, 1), np.sort(f, 1)
        mb, mf = sb[:, -1] - sb[:, -2], sf[:, -1] - sf[:, -2]
        flips = b.argmax(1) != f.argmax(1)
        rec = {'clip': ci, 'sigma': sigma, 't_star': T, 'n': N, 'counts_h16': cnt['h16'], 'counts_fp32': cnt['fp32'], 'flips': int(flips.sum()),
               'logit_err_max': float(np.abs(e_).max()), 'leader_diff_err': {'max': float(le.max()), 'rms': float(np.sqrt((le ** 2).mean())),
                                                                             'p99': float(np.quantile(le, 0.99)), 'p999': float(np.quantile(le, 0.999))},
               'margin_fp32_median': float(np.median(mf)), 'flip_margins_h16': sorted(float(v) for v in mb[flips]),
               'margin_h16_frac_below': {str(x): float((mb < x).mean()) for x in (0.005, 0.01, 0.02, 0.03, 0.05, 0.1, 0.2, 0.5)},
               'samples_per_s': {k: N / v for k, v in secs.items()}}
        ex = lg['x3'] - f
        lex = np.abs(ex - ex[rows, f.argmax(1)][:, None]).max(1)
        sx = np.sort(lg['x3'], 1)
        rec['x3'] = {'flips': int((lg['x3'].argmax(1) != f.argmax(1)).sum()), 'logit_err_max': float(np.abs(ex).max()),
                     'leader_diff_err': {'max': float(lex.max()), 'rms': float(np.sqrt((lex ** 2).mean()))},
                     'margin_frac_below': {str(x): float(((sx[:, -1] - sx[:, -2]) < x).mean()) for x in (1e-4, 3e-4, 1e-3, 3e-3)}}
        e1 = b - lg['x3']                  # what calibrate_spec_recheck measures: the 16-bit tier against the split-f16 tier
        le1 = np.abs(e1 - e1[rows, lg['x3'].argmax(1)][:, None]).max(1)
        rec['h16_vs_x3_leader_diff_err_max'] = float(le1.max())
        report.append(rec)
        raw['bf16_c%d_s%g' % (ci, sigma)] = b.astype(np.float32)         # key names of tools/fit_recheck_tail.py ('bf16' = the 16-bit tier)
        raw['fp32_c%d_s%g' % (ci, sigma)] = f.astype(np.float32)
        print(json.dumps(rec), flush=True)
        with open(os.path.join(OUT, 'c5_flip_study.json'), 'w') as fh:
            json.dump(report, fh, indent=1)
np.savez_compressed(os.path.join(OUT, 'c5_flip_study.npz'), **raw)
# the exact-vote mode on the same keys as the last cell: counts must equal the fp32 tier's
eng.set_mode(E.MODE_EXACT_VOTES)
eng.spec_recheck_stats(reset=True)
torch.cuda.synchronize(); t0 = time.time()
c, _, _ = eng.spec_smooth_votes(*args, N, seed=7000 + CLIPS[-1])
torch.cuda.synchronize(); dt = time.time() - t0
voted, rechecked, re32 = eng.spec_recheck_stats(detail=True)
print(json.dumps({'exact_vote_mode': {'tau_spec': eng.spec_recheck_margin, 'tau_spec2': eng.spec_recheck_margin2, 'rechecked_fp32': re32, 'counts': c.cpu().tolist(), 'equals_fp32': c.cpu().tolist() == report[-1]['counts_fp32'],
                                      'samples_per_s': N / dt, 'rechecked': rechecked, 'voted': voted}}), flush=True)
eng.close()
