#!/usr/bin/env python3
"""Evidence: the exact-vote mode's per-class counts against the exact-fp32 path's on the same Philox keys, at an N the fp32 path
can still run (N = 16 384 per sigma by default).  Writes gpurun_out/exact_vs_fp32_clipC.json.
REF=profiles/earlier_record.json: take the fp32 (and 16-bit) counts of the same (clip, sigma, n) from an earlier record of this tool
instead of running those paths again (the fp32 path is the slow one: 546 s at N = 100 000) — for re-validating a new build or bound."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
N = int(os.environ.get('N', 16384))
SIGMAS = [float(s) for s in os.environ.get('SIGMAS', '0.5,0.25').split(',')]
HALF = os.environ.get('HALF', 'f16')           # operand format of tier 1 (bf16: bound 0.30 instead of 0.034, far more rechecks)
eng = E.Engine(max_batch=256, precision=E.EXACT, recheck_batch=64, half_type=E.HALF_F16 if HALF == 'f16' else E.HALF_BF16)
eng.load_wavenet(synth.wavenet_state_dict(1234))
CLASSIFIER = os.environ.get('CLASSIFIER', 'vgg19_bn')     # resnext29: the reference script's default classifier (calibrated stand-in)
if CLASSIFIER == 'resnext29':
    eng.load_resnext29(synth.resnext29_state_dict(2929))
else:
    eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
ab = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)['Alpha_bar']
CLIP = int(os.environ.get('CLIP', 0))
clip = torch.from_numpy(synth.synthetic_clip(CLIP)).cuda()
REF = json.load(open(os.environ['REF'])) if os.environ.get('REF') else None
out = []
for sigma in SIGMAS:
    t = int(torch.abs(ab - 1 / (1 + sigma ** 2)).min(0, keepdim=True)[1].item())
    args = (clip, sigma, float(torch.tensor((1 / (1 + sigma ** 2)) ** 0.5, dtype=torch.float32)), t, float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t]), N)
    rec = {'clip': CLIP, 'half': HALF, 'classifier': CLASSIFIER, 'sigma': sigma, 't_star': t + 1, 'n': N, 'margins': [eng.recheck_margin, eng.recheck_margin2]}
    ref = next((r for r in (REF or []) if r.get('clip', 0) == CLIP and r['sigma'] == sigma and r['n'] == N), None)
    if REF is not None and ref is None:
        raise SystemExit('no record for clip %d sigma %g n %d in %s' % (CLIP, sigma, N, os.environ['REF']))
    if ref is not None:
        rec['fp32'], rec['fast'] = ref['fp32'], ref['fast']          # (the stored 16-bit counts are those of the REF run's operand format)
        rec['fp32_and_fast_counts_from'] = os.environ['REF']
    for name, mode in (('exact', E.MODE_EXACT_VOTES),) + ((('fp32', E.MODE_FP32), ('fast', E.MODE_FAST)) if ref is None else ()):
        eng.set_mode(mode); eng.recheck_stats(reset=True)
        torch.cuda.synchronize(); t0 = time.time()
        c, done = None, 0
        while done < N:                     # chunks: a progress line every ~45 s of fp32 work keeps the run visibly alive
            k = min(8192, N - done)
            c, _, _ = eng.smooth_votes(*args[:-1], k, seed=4242, sample0=done, counts=c)
            done += k
            torch.cuda.synchronize()
            print('  %s: %d / %d samples, %.0f s' % (name, done, N, time.time() - t0), flush=True)
        rec[name] = {'counts': c.cpu().tolist(), 'seconds': time.time() - t0, 'stats': eng.recheck_stats(detail=True)}
    rec['exact_equals_fp32'] = rec['exact']['counts'] == rec['fp32']['counts']
    rec['fast_differs_by'] = sum(abs(a - b) for a, b in zip(rec['fast']['counts'], rec['fp32']['counts'])) // 2
    print(json.dumps(rec), flush=True)
    out.append(rec)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'exact_vs_fp32_%sclip%d.json' % ('' if CLASSIFIER == 'vgg19_bn' else CLASSIFIER + '_', CLIP)), 'w'), indent=1)
assert all(r['exact_equals_fp32'] for r in out)
