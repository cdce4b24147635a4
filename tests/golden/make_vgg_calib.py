#!/usr/bin/env python3
"""BUILD-CONTAINER tool: BatchNorm running statistics for the synthetic VGG19_bn (synth seed 4321).

A randomly initialised 19-layer net with arbitrary BN statistics is degenerate (one class wins every
Monte Carlo sample).  This script runs ONE calibration pass over 32 noisy synthetic clips and stores
the per-layer batch mean/variance as the BN running statistics, like a trained checkpoint would hold.
The result is DATA (11 008 floats) committed as dmad_hip/data/vgg19_bn_calib_seed4321.npz so that
the build container and the GPU box load bit-identical parameters.
"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')
sys.path[:0] = [ROOT, PKG]
from dmad_hip import synth
from oracle import dmad_oracle as orc

torch.set_num_threads(8)
sd = synth.vgg19_bn_state_dict(4321, calibrated=False)
clips = torch.from_numpy(np.stack([synth.synthetic_clip(i) for i in range(4)]))
g = torch.Generator().manual_seed(5)
c = clips[torch.arange(32) % 4]
x = orc.mel_db((1 / 1.25) ** 0.5 * (c + 0.5 * torch.randn(c.shape, generator=g)))
out, idx = {}, 0
for v in synth.VGG19_CFG:
    if v == 'M':
        x = F.max_pool2d(x, 2, 2); idx += 1; continue
    x = F.conv2d(x, torch.from_numpy(sd['features.%d.weight' % idx]), torch.from_numpy(sd['features.%d.bias' % idx]), padding=1)
    b = idx + 1
    m, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
    out['features.%d.running_mean' % b] = m.numpy().astype(np.float32)
    out['features.%d.running_var' % b] = var.numpy().astype(np.float32)
    x = F.relu(F.batch_norm(x, m, var, torch.from_numpy(sd['features.%d.weight' % b]), torch.from_numpy(sd['features.%d.bias' % b]), False, 0., 1e-5))
    idx += 3
np.savez_compressed(os.path.join(PKG, 'dmad_hip', 'data', 'vgg19_bn_calib_seed4321.npz'), **out)
print('wrote', len(out), 'arrays')
