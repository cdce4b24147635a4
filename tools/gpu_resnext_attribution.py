#!/usr/bin/env python3
"""Evidence: where does tier 1's logit error come from when the classifier is ResNeXt29?  Per (clip, sigma), on the same Philox keys:
  (a) the classifier's 16-bit tier alone: fp32-path purified waveforms -> mel -> dmad_classify_tier(1) against tier 0;
  (b) the 16-bit WaveNet alone: 16-bit-path purified waveforms -> fp32 classifier against the all-fp32 logits;
  (c) both (what tier 1 of the vote loop runs).
Statistic: the leader-difference error max_j |e_j - e_leader| (what a recheck bound has to cover), max and rms over the samples.
Writes gpurun_out/resnext_attribution.json.

    N=1024 CLIPS=0,1,2 SIGMAS=0.25,0.5,1.0 python tools/gpu_resnext_attribution.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth  # noqa: E402
from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams  # noqa: E402

N = int(os.environ.get('N', 1024))
CLIPS = [int(c) for c in os.environ.get('CLIPS', '0,1,2').split(',')]
SIGMAS = [float(s) for s in os.environ.get('SIGMAS', '0.25,0.5,1.0').split(',')]
CLASSIFIER = os.environ.get('CLASSIFIER', 'resnext29')
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
ab = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)['Alpha_bar']
eng = E.Engine(max_batch=256, precision=E.EXACT, recheck_batch=64)
eng.load_wavenet(synth.wavenet_state_dict(1234))
if CLASSIFIER == 'resnext29':
    eng.load_resnext29(synth.resnext29_state_dict(2929))
else:
    eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))


def lead_err(a, ref):
    a, ref = a.double().cpu().numpy(), ref.double().cpu().numpy()
    e = a - ref
    le = np.abs(e - e[np.arange(len(e)), ref.argmax(1)][:, None]).max(1)
    return {'max': float(le.max()), 'rms': float(np.sqrt((le ** 2).mean())), 'flips': int((a.argmax(1) != ref.argmax(1)).sum())}


def classify(x0, tier):
    out = []
    for i in range(0, x0.shape[0], 256):
        spec = eng.mel_db(x0[i:i + 256])
        out.append(eng.classify_tier(spec, tier) if CLASSIFIER == 'resnext29' else eng.classify(spec))
    return torch.cat(out)


report = []
for ci in CLIPS:
    clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
    for sigma in SIGMAS:
        t = int(torch.abs(ab - 1 / (1 + sigma ** 2)).min(0, keepdim=True)[1].item())
        c_a, c_b = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])
        sc = float(torch.tensor((1 / (1 + sigma ** 2)) ** 0.5, dtype=torch.float32))
        x0 = {}
        for name, mode in (('fp32', E.MODE_FP32), ('h16', E.MODE_FAST)):
            eng.set_mode(mode)
            _, _, x0[name] = eng.smooth_votes(clip, sigma, sc, t, c_a, c_b, N, seed=1000 + ci, sample0=0, want_x0=True)
        ref = classify(x0['fp32'], 0)
        rec = {'clip': ci, 'sigma': sigma, 'n': N, 'logit_std_fp32': float(ref.std()),
               'classifier_16bit_alone': lead_err(classify(x0['fp32'], 1), ref) if CLASSIFIER == 'resnext29' else None,
               'wavenet_16bit_alone': lead_err(classify(x0['h16'], 0), ref),
               'both': lead_err(classify(x0['h16'], 1), ref) if CLASSIFIER == 'resnext29' else None,
               'x0_relerr_max': float((x0['h16'] - x0['fp32']).abs().max() / x0['fp32'].abs().max())}
        report.append(rec)
        print(json.dumps(rec), flush=True)
        with open(os.path.join(OUT, 'resnext_attribution.json'), 'w') as fh:
            json.dump(report, fh, indent=1)
eng.close()
