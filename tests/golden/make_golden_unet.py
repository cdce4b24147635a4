#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY generator of tests/golden/unet.npz: the Improved-Diffusion UNet purifier components of
SURVEY §8 row N1 run by the imported reference (improved_diffusion.script_util.create_model_and_diffusion with the
arguments of improved_diffusion_ddpm.py:64-93) on the build's seeded weights: UNet forward at two timesteps (with
intermediate taps), GaussianDiffusion.q_sample and p_sample with captured noise, the mel standardisation pair.

Usage:  python tests/golden/make_golden_unet.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shims + path setup only)


def main():
    torch.set_num_threads(8)
    mg.install_shims()
    from dmad_hip import synth
    sys.path.insert(0, os.path.join(mg.REF, 'diffusion_models', 'Improved_Diffusion_Unconditional'))
    from improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    import types
    tv = types.ModuleType('torchvision.transforms')          # the data-set module star-imports it at import time
    tv.Compose = lambda ts: ts
    sys.modules['torchvision.transforms'] = tv
    sys.modules['torchvision'].transforms = tv
    sys.modules['torchvision'].__path__ = []
    from improved_diffusion.sc09_spectrogram_dataset import melspec_standardize, melspec_inv_standardize
    args = model_and_diffusion_defaults()
    args.update(image_size=32, num_channels=128, num_res_blocks=3, learn_sigma=False, diffusion_steps=1000, noise_schedule='linear')
    model, diffusion = create_model_and_diffusion(**args)
    sd = synth.unet_state_dict(5252)
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(mg.to_torch_sd(sd))
    model.eval()

    with np.load(os.path.join(HERE, 'classifiers.npz')) as z:
        spec = torch.from_numpy(z['spec_in'])[:2]                       # [2,1,32,32] mel-dB
    x0 = melspec_standardize(spec)
    out = {'spec': spec.numpy(), 'x0': x0.numpy(), 'inv_std': melspec_inv_standardize(x0).numpy(), 'seed': np.array(5252)}
    g = torch.Generator().manual_seed(31)
    noise = torch.randn(x0.shape, generator=g)
    for t in (3, 40):
        tt = torch.full((2,), t, dtype=torch.long)
        x_t = diffusion.q_sample(x0, tt, noise=noise)
        taps = {}
        hooks = [model.input_blocks[7].register_forward_hook(lambda m, i, o: taps.__setitem__('in7', o.detach()[:, ::32, ::3, ::3].numpy().copy())),
                 model.middle_block.register_forward_hook(lambda m, i, o: taps.__setitem__('mid', o.detach()[:, ::32].numpy().copy())),
                 model.output_blocks[11].register_forward_hook(lambda m, i, o: taps.__setitem__('out11', o.detach()[:, ::32, ::5, ::5].numpy().copy()))]
        with torch.no_grad():
            eps = model(x_t, tt)
        for h in hooks:
            h.remove()
        out['x_t%d' % t] = x_t.numpy(); out['eps_t%d' % t] = eps.numpy()
        for k, v in taps.items():
            out['%s_t%d' % (k, t)] = v
    # p_sample: captured noise = the randn_like draw inside
    for t in (3, 0):
        tt = torch.full((2,), t, dtype=torch.long)
        x_t = torch.from_numpy(out['x_t3'])
        torch.manual_seed(900 + t)
        with torch.no_grad():
            r = diffusion.p_sample(model, x_t, tt)
        torch.manual_seed(900 + t)
        z = torch.randn_like(x_t)
        out['p_sample_t%d' % t] = r['sample'].numpy(); out['p_xstart_t%d' % t] = r['pred_xstart'].numpy()
        out['p_noise_t%d' % t] = z.numpy()
    print({k: v.shape for k, v in out.items()})
    np.savez_compressed(os.path.join(HERE, 'unet.npz'), **out)


if __name__ == '__main__':
    main()
