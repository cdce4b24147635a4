#!/usr/bin/env python3
"""Measured error of the UNet tiers against the reference fixture (tests/golden/unet.npz): relmax of eps at t = 3 / 40 on the exact-fp32
tier and the 16-bit tier, absolute error of p_sample's pred_xstart / sample — the numbers the tolerances in
tests/test_gpu_parity.py::test_unet_purifier_vs_reference_fixture are set from.  Writes gpurun_out/unet_tolerance.json."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth  # noqa: E402
from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion  # noqa: E402


def relmax(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / np.abs(ref).max())


z = np.load(os.path.join(ROOT, 'tests', 'golden', 'unet.npz'))
eng = E.Engine(max_batch=4, precision=E.EXACT, with_classifier=False, with_wavenet=False)
pur = create_improved_diffusion(None, reverse_timestep=3, state_dict=synth.unet_state_dict(int(z['seed'])), engine=eng)
model, gd = pur.model, pur.diffusion
out = {}
for mode, name in ((E.MODE_FP32, 'fp32'), (E.MODE_FAST, 'f16')):
    eng.set_mode(mode)
    for t in (3, 40):
        tt = torch.full((2,), t, dtype=torch.long).cuda()
        eps = model(torch.from_numpy(z['x_t%d' % t]).cuda(), tt)
        out['%s_eps_t%d_relmax' % (name, t)] = relmax(eps.cpu().numpy(), z['eps_t%d' % t])
        out['%s_eps_t%d_absmax' % (name, t)] = float(np.abs(eps.cpu().numpy() - z['eps_t%d' % t]).max())
        out['eps_t%d_ref_absmax' % t] = float(np.abs(z['eps_t%d' % t]).max())
    for t in (3, 0):
        r = gd.p_sample(model, torch.from_numpy(z['x_t3']).cuda(), torch.full((2,), t), noise=torch.from_numpy(z['p_noise_t%d' % t]).cuda())
        out['%s_p_xstart_t%d_abs' % (name, t)] = float((r['pred_xstart'].cpu() - torch.from_numpy(z['p_xstart_t%d' % t])).abs().max())
        out['%s_p_sample_t%d_abs' % (name, t)] = float((r['sample'].cpu() - torch.from_numpy(z['p_sample_t%d' % t])).abs().max())
        out['p_xstart_t%d_ref_absmax' % t] = float(np.abs(z['p_xstart_t%d' % t]).max())
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'unet_tolerance.json'), 'w') as f:
    json.dump(out, f, indent=1)
eng.close()
