#!/usr/bin/env python3
"""Development: latency of one eps-network evaluation at small batches (the attack drivers' query path) — wall time per
call vs the sum of kernel durations shows how launch-bound the chain of ~40 launches is."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
eng = E.Engine(max_batch=8, precision=E.BF16, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
for B in (1, 2, 4, 8):
    x = torch.randn(B, 16000, device='cuda') * 0.3
    for _ in range(3):
        eng.wavenet_eps(x, 65)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    for _ in range(n):
        eng.wavenet_eps(x, 65)
    torch.cuda.synchronize()
    print('B=%d: %.3f ms per eps evaluation (%.1f clips/s)' % (B, (time.time() - t0) / n * 1e3, B * n / (time.time() - t0)), flush=True)
