#!/usr/bin/env python3
"""Development: time one WaveNet evaluation on each path of an exact-vote engine (0 = 16-bit, 1 = exact fp32, 2 = split-f16);
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import _lib
if os.environ.get('DMAD_LIB'):                   # A/B of two builds on one box
    _lib.LIB_PATH = os.environ['DMAD_LIB']
from dmad_hip import engine as E, synth
B = int(os.environ.get('B', 16))
PATHS = [int(p) for p in os.environ.get('PATHS', '1,2').split(',')]
eng = E.Engine(max_batch=B, precision=E.EXACT, recheck_batch=B, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
for path in PATHS:
    eng.wavenet_eps_path(x, 65, path); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        eng.wavenet_eps_path(x, 65, path)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(os.environ.get('DMAD_LIB', 'in-tree'), 'path %d B=%d: %.1f ms -> %.1f clips/s, %.1f TFLOP/s fp32-equivalent (606.1 GFLOP/clip)' % (path, B, dt * 1e3, B / dt, B * 606.1e9 / dt / 1e12), flush=True)
