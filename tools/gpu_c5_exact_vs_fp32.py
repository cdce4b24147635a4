#!/usr/bin/env python3
"""Evidence: the spec-domain loop's exact-vote counts (16-bit -> split-f16 -> fp32 tiers) against the exact-fp32 UNet's on the same
Philox keys, per clip.  Writes gpurun_out/c5_exact_vs_fp32.json.       N=8192 CLIPS=0,1 T=25 python tools/gpu_c5_exact_vs_fp32.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion
N = int(os.environ.get('N', 8192))
CLIPS = [int(c) for c in os.environ.get('CLIPS', '0,1').split(',')]
T = int(os.environ.get('T', 25))
SIGMA = float(os.environ.get('SIGMA', 0.5))
eng = E.Engine(max_batch=2048, precision=E.EXACT, recheck_batch=0, with_wavenet=False)
eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321, calibrated='c5'))
pur = create_improved_diffusion(None, reverse_timestep=T, state_dict=synth.unet_state_dict(31), engine=eng)
coef = tuple(pur.purify_coefficients())
out = []
for ci in CLIPS:
    clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
    args = (clip, SIGMA) + coef + (-100.0, 38.22)
    rec = {'clip': ci, 'sigma': SIGMA, 't_star': T, 'n': N, 'margins': [eng.spec_recheck_margin, eng.spec_recheck_margin2]}
    for name, mode in (('exact', E.MODE_EXACT_VOTES), ('fp32', E.MODE_FP32), ('fast', E.MODE_FAST)):
        eng.set_mode(mode); eng.spec_recheck_stats(reset=True)
        eng.spec_smooth_votes(*args, 64, seed=1)
        eng.spec_recheck_stats(reset=True)
        torch.cuda.synchronize(); t0 = time.time()
        c, done = None, 0
        while done < N:
            k = min(2048, N - done)
            c, _, _ = eng.spec_smooth_votes(*args, k, seed=9090 + ci, sample0=done, counts=c)
            done += k
            torch.cuda.synchronize()
            print('  clip %d %s: %d / %d samples, %.0f s' % (ci, name, done, N, time.time() - t0), flush=True)
        rec[name] = {'counts': c.cpu().tolist(), 'seconds': time.time() - t0, 'stats': eng.spec_recheck_stats(detail=True)}
    rec['exact_equals_fp32'] = rec['exact']['counts'] == rec['fp32']['counts']
    rec['fast_differs_by'] = sum(abs(a - b) for a, b in zip(rec['fast']['counts'], rec['fp32']['counts'])) // 2
    print(json.dumps(rec), flush=True)
    out.append(rec)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'c5_exact_vs_fp32.json'), 'w'), indent=1)
assert all(r['exact_equals_fp32'] for r in out)
