#!/usr/bin/env python3
"""Development: one exact-fp32 UNet evaluation of B spectrograms (default 1: the spec loop's recheck of a single low-margin sample)
for a rocprofv3 kernel trace, and its wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
import torch
from dmad_hip import engine as E, synth
B = int(os.environ.get('B', 1))
eng = E.Engine(max_batch=int(os.environ.get("MAXB", max(B, 8))), precision=E.EXACT, with_classifier=False, with_wavenet=False)
eng.set_mode(E.MODE_FP32)
eng.load_unet(synth.unet_state_dict(5252))
x = torch.randn(B, 32, 32, device='cuda') * 0.5
for t in (40, 39):
    eng.unet_eps(x, t)
torch.cuda.synchronize()
t0 = time.time()
for i in range(10):
    eng.unet_eps(x, 40 - (i & 1))
torch.cuda.synchronize()
print('fp32 UNet, B = %d: %.3f ms per evaluation' % (B, (time.time() - t0) / 10 * 1e3), flush=True)
