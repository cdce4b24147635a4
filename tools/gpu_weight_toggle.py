#!/usr/bin/env python3
"""Low-toggle weight images (round-4 experiment): what do f16 weight images rounded to 10 - k mantissa bits
(DMAD_WEIGHT_MASK_BITS = k: the k low bits zero, fewer toggling bits on the weight path) buy in joules / time per launch of the
layer kernel, and what do they cost in logit error (hence in recheck fraction)?

Per k: an exact-vote engine is built with the masked images; (a) layer 5 is launched back to back for SECONDS under the hwmon
power sampler: ms, W, J above idle per launch; (b) the leader-difference error of the 16-bit tier against the split-f16 tier
(unmasked fp32-grade weights) on N Philox samples of three clips at sigma = 0.5 — the statistic the recheck bound is set from;
(c) the share of those samples whose 16-bit margin is below 1.4 x that error (what would leave tier 1).
Writes gpurun_out/weight_toggle.json.     B=128 SECONDS=4 N=1024 KS=0,1,2,3 python tools/gpu_weight_toggle.py
Since round 5 the mask path exists only in a library built with -DDMAD_DEV_WEIGHT_MASK (make CXXFLAGS+=-DDMAD_DEV_WEIGHT_MASK, pointed at with
DMAD_LIB): the product library ignores the variables, so that a leaked environment cannot void the exact-vote bounds.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd'), os.path.join(ROOT, 'tools')]
from dmad_hip import engine as E, synth  # noqa: E402
from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams  # noqa: E402
import power_sampler as PS  # noqa: E402

B = int(os.environ.get('B', 128))
SECONDS = float(os.environ.get('SECONDS', 4))
N = int(os.environ.get('N', 1024))
KS = [int(k) for k in os.environ.get('KS', '0,1,2,3,-1,0').split(',')]       # -1 = bf16 operands; 0 twice: drift of the box
WHICH = os.environ.get('WHICH', '3')
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)

src = PS.sysfs_sources()
_, idle_w, _ = PS.measure(lambda: time.sleep(2.0), src, settle=0.5)
hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
ab = hp['Alpha_bar']
sigma = 0.5
abar = 1 / (1 + sigma ** 2)
t = int(torch.abs(ab - abar).min(0, keepdim=True)[1].item())
c_a, c_b = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])
sc = float(torch.tensor(abar ** 0.5, dtype=torch.float32))
wsd, csd = synth.wavenet_state_dict(1234), synth.vgg19_bn_state_dict(4321)
clips = [torch.from_numpy(synth.synthetic_clip(i)).cuda() for i in range(3)]
report = {'B': B, 'seconds': SECONDS, 'n_per_clip': N, 'idle_w': idle_w, 'which_images': WHICH, 'sysfs': src, 'rows': []}
x = torch.randn(B, 16000, device='cuda') * 0.3
for k in KS:
    os.environ['DMAD_WEIGHT_MASK_BITS'] = str(k)
    os.environ['DMAD_WEIGHT_MASK_WHICH'] = WHICH
    if k < 0:                                 # reference row: the same kernel on bf16 operands (timing / power only)
        os.environ['DMAD_WEIGHT_MASK_BITS'] = '0'
        eng = E.Engine(max_batch=B, precision=E.BF16, half_type=E.HALF_BF16, with_classifier=False)
        eng.load_wavenet(wsd)
    else:
        eng = E.Engine(max_batch=B, precision=E.EXACT, half_type=E.HALF_F16, recheck_batch=min(64, B))
        eng.load_wavenet(wsd)
        eng.load_vgg19_bn(csd)
        eng.set_mode(E.MODE_FAST)
    eng.wavenet_eps(x, t)                     # fills the residual stream the timed layer reads
    torch.cuda.synchronize()
    ms = eng.time_layer(5, B, 20)
    iters = max(20, int(SECONDS * 1e3 / ms))
    ms, pw, fq = PS.measure(lambda: eng.time_layer(5, B, iters), src)
    row = {'mask_bits': k, 'mantissa_bits': 10 - k, 'ms_per_launch': ms, 'power_w': pw, 'sclk_mhz': fq,
           'tflops': 2.0 * 16000 * (512 * 768 + 256 * 256) * B / (ms * 1e-3) / 1e12,
           'joules_above_idle_per_launch': (pw - idle_w) * ms * 1e-3 if pw and idle_w else None}
    if k < 0:
        row['operands'] = 'bf16'
        print(json.dumps(row), flush=True)
        report['rows'].append(row)
        eng.close()
        continue
    errs, margins = [], []
    for ci, clip in enumerate(clips):
        idx = torch.arange(N, dtype=torch.int64, device='cuda')
        a = (clip, sigma, sc, t, c_a, c_b)
        fast = eng.eval_samples(*a, idx, path=0, seed=77 + ci).double()
        mid = eng.eval_samples(*a, idx, path=2, seed=77 + ci).double()
        e = fast - mid
        le = (e - e.gather(1, mid.argmax(1, keepdim=True))).abs().max(1).values
        errs.append(le)
        top2 = fast.topk(2, dim=1).values
        margins.append(top2[:, 0] - top2[:, 1])
    le, mg = torch.cat(errs), torch.cat(margins)
    emax = float(le.max())
    row.update({'lead_err_max': emax, 'lead_err_rms': float(le.pow(2).mean().sqrt()), 'lead_err_q99': float(torch.quantile(le, 0.99)),
                'bound_1p4x': 1.4 * emax, 'frac_margin_below_bound': float((mg < 1.4 * emax).double().mean()),
                'frac_margin_below_0p034': float((mg < 0.034).double().mean())})
    print(json.dumps(row), flush=True)
    report['rows'].append(row)
    eng.close()
    del eng
    torch.cuda.empty_cache()
with open(os.path.join(OUT, 'weight_toggle.json'), 'w') as f:
    json.dump(report, f, indent=1)
