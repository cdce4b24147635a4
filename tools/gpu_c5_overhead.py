#!/usr/bin/env python3
"""Development: where does the exact-vote mode of the spec-domain loop (C5) spend its time beyond the 16-bit tier?  Runs the same
N samples in fast / exact / exact-again / fp32-of-a-few and prints seconds and recheck counts."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion
N, B = int(os.environ.get('N', 10000)), int(os.environ.get('B', 2048))
eng = E.Engine(max_batch=B, precision=E.EXACT, recheck_batch=0, with_wavenet=False)
eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321, calibrated='c5'))
pur = create_improved_diffusion(None, reverse_timestep=25, state_dict=synth.unet_state_dict(31), engine=eng)
args = (torch.from_numpy(synth.synthetic_clip(0)).cuda(), 0.5) + tuple(pur.purify_coefficients()) + (-100.0, 38.22)
eng.spec_smooth_votes(*args, B, seed=1)


def run(mode, n, seed=2024):
    eng.set_mode(mode)
    eng.spec_recheck_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, _, _ = eng.spec_smooth_votes(*args, n, seed=seed)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, c.cpu().tolist(), eng.spec_recheck_stats()


for name, mode, n in (('fast', E.MODE_FAST, N), ('exact', E.MODE_EXACT_VOTES, N), ('exact again', E.MODE_EXACT_VOTES, N), ('fast again', E.MODE_FAST, N),
                      ('fp32 x1', E.MODE_FP32, 1), ('fp32 x1 again', E.MODE_FP32, 1), ('fp32 x8', E.MODE_FP32, 8)):
    dt, c, st = run(mode, n)
    print('%-12s n=%5d  %.3f s  %.1f samples/s  recheck %s  votes %s' % (name, n, dt, n / dt, st, c), flush=True)
