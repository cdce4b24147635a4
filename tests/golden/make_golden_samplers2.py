#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY generator of tests/golden/samplers2.npz: the sampler variants of SURVEY §8 row N4
(`DiffWave.fast_reverse`, `_predict_x1_from_eps`, `_predict_x0_from_x1`, `ReffWave.forward`) run by the
imported reference (/root/reference, through the harness shims of make_golden.py) on the build's seeded
weights and inputs, with the CPU noise draws the reference consumed captured next to the outputs.

Usage:  python tests/golden/make_golden_samplers2.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shims + path setup only)


def main():
    torch.set_num_threads(8)
    mg.install_shims()
    from dmad_hip import synth
    # the reference's `diffusion_models` is a namespace package: take this repository's package directory (which
    # holds a regular package of the same name) off the path before importing it
    pkg = os.path.join(mg.ROOT, 'diffusion-model-for-audio-defense_amd')
    sys.path[:] = [p for p in sys.path if os.path.abspath(p) != pkg]
    for name in [m for m in sys.modules if m.split('.')[0] == 'diffusion_models']:
        del sys.modules[name]
    from util import calc_diffusion_hyperparams
    from WaveNet import WaveNet_Speech_Commands
    from diffusion_models.diffwave_ddpm import DiffWave, ReffWave

    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    net = WaveNet_Speech_Commands(**synth.WAVENET_CONFIG)
    net.load_state_dict(mg.to_torch_sd(synth.wavenet_state_dict(1234)))
    net.eval()
    x0 = torch.from_numpy(synth.synthetic_clip(2))[None]          # [1,1,16000]
    out = {'x0': x0.numpy()}
    t0 = time.time()

    den = DiffWave(model=net, diffusion_hyperparams=hp, reverse_timestep=9)
    den.eval()
    torch.manual_seed(501)
    with torch.no_grad():
        x_t = den._diffusion(x0.clone())
        fr = den.fast_reverse(x_t.clone())
    torch.manual_seed(501)
    zs = [torch.normal(0, 1, size=x0.shape).numpy() for _ in range(1 + 3)]   # 1 diffusion + K=3 strided steps
    out.update(fast_t9_x_t=x_t.numpy(), fast_t9=fr.numpy(), fast_t9_noise=np.stack(zs))
    with torch.no_grad():
        eps = den.compute_eps_t(x_t.clone(), 8)
        x1 = den._predict_x1_from_eps(x_t.clone(), 8, eps)
        x0p = den._predict_x0_from_x1(x1.clone())
        x0e = den._predict_x0_from_eps(x_t.clone(), 8, eps)
    out.update(eps_t9=eps.numpy(), x1_t9=x1.numpy(), x0_from_x1_t9=x0p.numpy(), x0_from_eps_t9=x0e.numpy())

    rw = ReffWave(model=net, diffusion_hyperparams=hp, reverse_timestep=4, num_re=3)
    rw.eval()
    torch.manual_seed(502)
    with torch.no_grad():
        pur = rw(x0.clone())
    torch.manual_seed(502)
    zs = [torch.normal(0, 1, size=x0.shape).numpy() for _ in range(3)]
    out.update(reff_t4_n3=pur.numpy(), reff_t4_n3_noise=np.stack(zs))
    print('generated in %.1fs' % (time.time() - t0))
    np.savez_compressed(os.path.join(HERE, 'samplers2.npz'), **out)


if __name__ == '__main__':
    main()
