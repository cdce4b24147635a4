#!/usr/bin/env python3
"""Development: run a few launches of one residual-layer kernel variant (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import _lib
if os.environ.get('DMAD_LIB'):
    _lib.LIB_PATH = os.environ['DMAD_LIB']
from dmad_hip import engine as E, synth
B = int(os.environ.get('B', 128))
eng = E.Engine(max_batch=B, precision=E.BF16, half_type=E.HALF_F16 if os.environ.get('HALF', 'f16') == 'f16' else E.HALF_BF16, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
eng.wavenet_eps(x, 65); torch.cuda.synchronize()
print('ms', eng.time_layer(5, B, int(os.environ.get('ITERS', 5))))
