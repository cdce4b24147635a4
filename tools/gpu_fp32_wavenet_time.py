#!/usr/bin/env python3
"""Development: time the exact-fp32 (parity mode) WaveNet, i.e. gemm_f32 on large clean shapes (M 512/256, N = B * 16000)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
B = int(os.environ.get('B', 8))
eng = E.Engine(max_batch=B, precision=E.FP32, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
eng.wavenet_eps(x, 65); torch.cuda.synchronize()
t0 = time.time()
eng.wavenet_eps(x, 65); torch.cuda.synchronize()
dt = time.time() - t0
print('fp32 wavenet B=%d: %.1f ms -> %.1f TFLOP/s (606.1 GFLOP/clip)' % (B, dt * 1e3, B * 606.1e9 / dt / 1e12))
