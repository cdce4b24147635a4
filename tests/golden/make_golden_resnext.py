#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY generator of tests/golden/resnext29.npz: logits (and two intermediate taps) of the imported
reference's `models.resnext.CifarResNeXt(nlabels=10, in_channels=1)` — the default classifier of
certified_robustness_eval.py:57 — loaded with the build's seeded synthetic weights (dmad_hip/synth.py), on the mel
spectrograms of tests/golden/classifiers.npz.

Usage:  python tests/golden/make_golden_resnext.py
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
    from models.resnext import CifarResNeXt          # /root/reference/audio_models/ConvNets_SpeechCommands/models
    assert '/root/reference/' in sys.modules['models.resnext'].__file__
    net = CifarResNeXt(nlabels=10, in_channels=1)
    net.load_state_dict(mg.to_torch_sd(synth.resnext29_state_dict(2929)))
    net.eval()
    with np.load(os.path.join(HERE, 'classifiers.npz')) as z:
        spec = torch.from_numpy(z['spec_in'])
    taps = {}        # the reference calls stage.forward() directly, so the hooks sit on the last bottleneck of a stage
    hooks = [net.stage_1[-1].register_forward_hook(lambda m, i, o: taps.__setitem__('stage_1', o.detach()[:, ::16, ::5, ::7].numpy().copy())),
             net.stage_3[-1].register_forward_hook(lambda m, i, o: taps.__setitem__('stage_3', o.detach()[:, ::64].numpy().copy()))]
    with torch.no_grad():
        logits = net(spec)
    for h in hooks:
        h.remove()
    print(logits)
    np.savez_compressed(os.path.join(HERE, 'resnext29.npz'), spec_in=spec.numpy(), logits=logits.numpy(),
                        stage_1=taps['stage_1'], stage_3=taps['stage_3'], seed=np.array(2929))


if __name__ == '__main__':
    main()
