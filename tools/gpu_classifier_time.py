#!/usr/bin/env python3
"""Development: time the classifier stage (mel spectrogram [B,1,32,32] -> logits) for VGG19_bn and ResNeXt29."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
B = int(os.environ.get('B', 128))
spec = torch.randn(B, 1, 32, 32, device='cuda') * 15 - 25
for name, flop in (('vgg19_bn', 0.83e9), ('resnext29', 10.8e9)):
    eng = E.Engine(max_batch=B, precision=E.BF16)
    if name == 'vgg19_bn':
        eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
    else:
        eng.load_resnext29(synth.resnext29_state_dict(2929))
    eng.classify(spec); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        eng.classify(spec)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print('%s B=%d: %.2f ms  (%.1f TFLOP/s fp32 at %.2f GFLOP/sample)' % (name, B, dt * 1e3, B * flop / dt / 1e12, flop / 1e9), flush=True)
    del eng

# Improved-Diffusion UNet eps-network (one NFE), 16.76 GFLOP/sample
eng = E.Engine(max_batch=B, precision=E.BF16, with_classifier=False)
eng.load_unet(synth.unet_state_dict(5252))
x = torch.randn(B, 32, 32, device='cuda') * 0.5
eng.unet_eps(x, 40); torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    eng.unet_eps(x, 40)
torch.cuda.synchronize()
dt = (time.time() - t0) / 3
print('unet eps B=%d: %.2f ms  (%.1f TFLOP/s fp32 at 16.76 GFLOP/sample, %.0f spectrograms/s per NFE)' % (B, dt * 1e3, B * 16.76e9 / dt / 1e12, B / dt), flush=True)
