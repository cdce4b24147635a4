#!/usr/bin/env python3
"""Development: time the wn_layer_bf16 ablation variants (DMAD_LAYER_VARIANT) in one process, interleaved rounds."""
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
names = sys.argv[1:] or ['0', '1', '2', '3', '4', '5']
res = {n: [] for n in names}
for rnd in range(4):
    for n in names:
        os.environ['DMAD_LAYER_VARIANT'] = n
        res[n].append(eng.time_layer(5, B, 10))
flop = B * 16000 * 2 * (512 * 768 + 256 * 256)
for n in names:
    ms = sorted(res[n])[len(res[n]) // 2]
    print('variant %s: median %.3f ms (min %.3f)  -> %.0f TF-equivalent' % (n, ms, min(res[n]), flop / ms / 1e9), flush=True)
