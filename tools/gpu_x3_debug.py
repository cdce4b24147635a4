#!/usr/bin/env python3
"""Development: the split-f16 tier (path 2) and the 16-bit path (path 0) against the exact-fp32 path (path 1), by WaveNet depth."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
for nl in (1, 2, 5, 36):
    cfg = dict(synth.WAVENET_CONFIG); cfg.update(num_res_layers=nl, dilation_cycle=min(12, max(1, nl)))
    eng = E.Engine(wavenet_config=cfg, max_batch=4, precision=E.EXACT, recheck_batch=4, with_classifier=False)
    eng.load_wavenet(synth.wavenet_state_dict(77, cfg))
    x = torch.randn(3, 16000, generator=torch.Generator().manual_seed(8)).cuda() * 0.4
    a = eng.wavenet_eps_path(x, 20, 1).cpu().numpy().astype(np.float64)
    b = eng.wavenet_eps_path(x, 20, 2).cpu().numpy().astype(np.float64)
    c = eng.wavenet_eps_path(x, 20, 0).cpu().numpy().astype(np.float64)
    rms = lambda v: np.sqrt((v ** 2).mean())
    print('layers %2d: x3-fp32 relmax %.3e rms %.3e | f16-fp32 relmax %.3e rms %.3e' % (
        nl, np.abs(b - a).max() / np.abs(a).max(), rms(b - a) / rms(a), np.abs(c - a).max() / np.abs(a).max(), rms(c - a) / rms(a)), flush=True)
    eng.close()
