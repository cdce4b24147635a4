#!/usr/bin/env python3
"""Evidence: WHICH rounding of the 16-bit tier produces its logit error?

The exact-vote mode rechecks every Monte Carlo sample whose 16-bit top-2 margin is below tau1 = headroom x E, E = the largest
error the 16-bit tier makes on a logit difference against the exact leader.  The recheck fraction (3.2 % of the sigma = 0.5
samples at tau1 = 0.034) is what the mode costs, so the question is which of the tier's roundings E comes from:
    weights (f16 images of W_dil, W_res, W_skip, W_f0) | the MFMA operand f16(h) of the dilated conv |
    the STORED residual stream (f16: its rounding is carried through every later layer) | the stored gate (f16 operand of the
    res conv and of the K = 9216 skip GEMM) | the f16 operand of final_conv.0.
The split-f16 tier is an fp32-grade pipeline (error 2e-4) with the same dataflow, and dmad_debug_rounding switches each of those
roundings on inside it, one at a time, all together (= an emulation of the 16-bit tier, checked against the real one) and all
but one (= what a fix of that one source would leave).  Same Philox keys everywhere; reference = the unmodified split-f16 tier.

    N=2048 CLIPS=0,1,2 SIGMAS=0.5 python tools/gpu_error_attribution.py      -> gpurun_out/error_attribution.{json,md}
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

N = int(os.environ.get('N', 2048))
CLIPS = [int(c) for c in os.environ.get('CLIPS', '0,1,2').split(',')]
SIGMAS = [float(s) for s in os.environ.get('SIGMAS', '0.5').split(',')]
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)

W, X, S = 1, 2, 4            # GemmF32Args::diag bits: weights f16 | the MFMA eats f16(activation) | the output is stored as f16
VARIANTS = [
    # name, masks (dil, res, skip, f0, init), what it is
    ('all (emulated 16-bit tier)', dict(dil=W | X | S, res=W | X | S, skip=W | X, f0=W | X, init=1), 'every rounding of the 16-bit tier'),
    ('weights, all four GEMMs', dict(dil=W, res=W, skip=W, f0=W), 'f16 weight images only'),
    ('  W_dil', dict(dil=W), ''),
    ('  W_res', dict(res=W), ''),
    ('  W_skip', dict(skip=W), ''),
    ('  W_f0', dict(f0=W), ''),
    ('h as MFMA operand', dict(dil=X), 'the dilated conv eats f16(h); the stored stream keeps 22 bits'),
    ('h stored as f16 (carry + operand)', dict(res=S, init=1), 'the residual stream itself is f16: what the 16-bit tier does'),
    ('gate stored as f16', dict(dil=S), 'operand of the res conv and of the skip GEMM'),
    ('y as f0 operand', dict(f0=X), 'final_conv.0 eats f16(skip sum / 6)'),
    # what fixing ONE source would leave
    ('all but: stream carry exact', dict(dil=W | X | S, res=W | X, skip=W | X, f0=W | X), 'h kept hi + lo for the epilogue, MFMA still eats f16(h)'),
    ('all but: weights exact', dict(dil=X | S, res=X | S, skip=X, f0=X, init=1), ''),
    ('all but: W_dil exact', dict(dil=X | S, res=W | X | S, skip=W | X, f0=W | X, init=1), ''),
    ('all but: gate exact', dict(dil=W | X, res=W | S, skip=W, f0=W | X, init=1), ''),
    ('all but: skip path exact (W_skip, gate->skip, f0)', dict(dil=W | X | S, res=W | X | S, init=1), 'the final kernel on fp32-grade operands'),
]
if os.environ.get('VARIANTS'):
    keep = [int(v) for v in os.environ['VARIANTS'].split(',')]
    VARIANTS = [VARIANTS[i] for i in keep]

hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
ab = hp['Alpha_bar']
eng = E.Engine(max_batch=256, precision=E.EXACT, recheck_batch=64)
eng.load_wavenet(synth.wavenet_state_dict(1234))
eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
TAU1 = eng.recheck_margin


def lead_err(a, ref):
    """per-sample error of a logit difference against the reference's leader: max_j |e_j - e_i|, i = argmax ref."""
    e = a - ref
    rows = np.arange(len(ref))
    return np.abs(e - e[rows, ref.argmax(1)][:, None]).max(1)


def stats(le):
    return {'max': float(le.max()), 'rms': float(np.sqrt((le ** 2).mean())), 'p99': float(np.quantile(le, 0.99)), 'p999': float(np.quantile(le, 0.999))}


cells, t_start = [], time.time()
for ci in CLIPS:
    clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
    for sigma in SIGMAS:
        abar_star = 1 / (1 + sigma ** 2)
        t = int(torch.abs(ab - abar_star).min(0, keepdim=True)[1].item())
        args = (clip, sigma, float(torch.tensor(abar_star ** 0.5, dtype=torch.float32)), t, float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t]))
        idx = torch.arange(N, device='cuda')
        seed = 4000 + ci
        eng.debug_rounding()
        ref = eng.eval_samples(*args, idx, path=2, seed=seed).cpu().numpy().astype(np.float64)
        eng.set_mode(E.MODE_FAST)
        fast = eng.eval_samples(*args, idx, path=0, seed=seed).cpu().numpy().astype(np.float64)
        eng.set_mode(E.MODE_EXACT_VOTES)
        srt = np.sort(fast, 1)
        margin = srt[:, -1] - srt[:, -2]
        cell = {'clip': ci, 'sigma': sigma, 'n': N, 'tier1': stats(lead_err(fast, ref)), 'tier1_flips': int((fast.argmax(1) != ref.argmax(1)).sum()),
                'tier1_margin_frac_below': {str(x): float((margin < x).mean()) for x in (0.005, 0.01, 0.0126, 0.015, 0.02, 0.025, 0.03, 0.034, 0.04)},
                'variants': {}}
        for name, masks, _ in VARIANTS:
            eng.debug_rounding(**masks)
            got = eng.eval_samples(*args, idx, path=2, seed=seed).cpu().numpy().astype(np.float64)
            eng.debug_rounding()
            cell['variants'][name] = dict(stats(lead_err(got, ref)), flips=int((got.argmax(1) != ref.argmax(1)).sum()))
        cells.append(cell)
        print(json.dumps({'clip': ci, 'sigma': sigma, 'tier1': cell['tier1'], 'elapsed_s': round(time.time() - t_start, 1)}), flush=True)
        for name, _, _ in VARIANTS:
            print('   %-52s %s' % (name, json.dumps(cell['variants'][name])), flush=True)
        with open(os.path.join(OUT, 'error_attribution.json'), 'w') as fh:
            json.dump({'tau1': TAU1, 'cells': cells}, fh, indent=1)

# summary over all cells: max of the maxima, rms over everything; the recheck fraction a bound of 1.4 x max would cost
lines = ['| variant | max | rms | p99.9 | share of the emulated tier\'s variance | recheck frac at tau = 1.4 x max |', '|---|---|---|---|---|---|']


def agg(get):
    mx = max(get(c)['max'] for c in cells)
    rms = float(np.sqrt(np.mean([get(c)['rms'] ** 2 for c in cells])))
    p = max(get(c)['p999'] for c in cells)
    return mx, rms, p


def frac_below(tau):
    """fraction of the 16-bit tier's margins below tau, interpolated over the recorded grid (all cells pooled)."""
    grid = sorted(float(k) for k in cells[0]['tier1_margin_frac_below'])
    vals = [np.mean([c['tier1_margin_frac_below'][str(g)] for c in cells]) for g in grid]
    return float(np.interp(tau, [0.0] + grid, [0.0] + vals))


t1 = agg(lambda c: c['tier1'])
all_rms = agg(lambda c: c['variants'][VARIANTS[0][0]])[1] if VARIANTS else float('nan')
lines.append('| the real 16-bit tier | %.4f | %.4f | %.4f | — | %.2f %% |' % (t1[0], t1[1], t1[2], 100 * frac_below(1.4 * t1[0])))
for name, _, what in VARIANTS:
    mx, rms, p = agg(lambda c: c['variants'][name])
    lines.append('| %s%s | %.4f | %.4f | %.4f | %.0f %% | %.2f %% |' % (name.strip() if not name.startswith('  ') else '&nbsp;&nbsp;' + name.strip(),
                                                                      (' — ' + what) if what else '', mx, rms, p, 100 * (rms / all_rms) ** 2, 100 * frac_below(1.4 * mx)))
md = ('Leader-difference logit error (max_j |e_j - e_i| against the unmodified split-f16 tier) over %d samples (%d per clip, clips %s, sigma %s), '
      'synthetic WaveNet seed 1234 + VGG19_bn seed 4321; committed tau1 = %.3g.\n\n' % (N * len(cells), N, CLIPS, SIGMAS, TAU1)) + '\n'.join(lines) + '\n'
with open(os.path.join(OUT, 'error_attribution.md'), 'w') as fh:
    fh.write(md)
print(md)
eng.close()
