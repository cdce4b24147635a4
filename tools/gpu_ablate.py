#!/usr/bin/env python3
"""Development: time the residual-layer kernel for several layers (dilations) in one process, interleaved rounds."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
B = int(os.environ.get('B', 128))
eng = E.Engine(max_batch=B, precision=E.BF16)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
eng.wavenet_eps(x, 65); torch.cuda.synchronize()
names = sys.argv[1:] or ['0', '5', '11']          # layer indices (dilation 2^(n%12))
res = {n: [] for n in names}
for rnd in range(4):
    for n in names:
        res[n].append(eng.time_layer(int(n), B, 10))
flop = B * 16000 * 2 * (512 * 768 + 256 * 256)
for n in names:
    ms = sorted(res[n])[len(res[n]) // 2]
    print('layer %s: median %.3f ms (min %.3f)  -> %.0f TF-equivalent' % (n, ms, min(res[n]), flop / ms / 1e9), flush=True)
