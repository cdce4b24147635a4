#!/usr/bin/env python3
"""BUILD-CONTAINER tool: calibration data that keep the synthetic classifier stand-ins from being degenerate.

The real ConvNets_SpeechCommands checkpoints are not available offline, so the Monte Carlo loops run on seeded
synthetic weights (dmad_hip/synth.py).  A randomly initialised deep net with arbitrary BatchNorm statistics votes ONE class on
every sample, with top-2 margins of tens of logit units: such a stand-in never exercises the recheck tiers of the exact-vote
mode and makes "exact votes" evidence vacuous (round-4 review).  tests/golden/make_vgg_calib.py fixed that for the VGG19_bn of
the waveform loop; this script does the same for

  resnext   the synthetic ResNeXt29 8x64d (seed 2929; the reference script's default classifier,
            certified_robustness_eval.py:57): ONE BatchNorm calibration pass over mel spectrograms of PURIFIED noisy
            synthetic clips (the oracle's one-shot DiffWave purifier on the seeded WaveNet, sigma in {0.25, 0.5, 1.0}), then the
            linear head is made insensitive to the noise level (the sigma-mean feature directions are projected out of its
            rows), centred on the mean feature and scaled to the within-cell logit spread of the VGG headline (a trained head is
            balanced over its data; a random one on pooled ReLU features is dominated by a constant offset: BN calibration
            alone still votes ONE class on every sample, with margins of 0.6-1.0).
  vgg_c5    a second BatchNorm statistics set for the synthetic VGG19_bn (seed 4321) measured on the spectrograms the
            SPEC-domain chain hands it (BASELINE C5: mel-dB -> standardise -> q_sample(t*) -> t* + 1 p_sample steps of the
            seeded Improved-Diffusion UNet -> un-standardise), whose distribution differs from the waveform loop's.

The results are DATA (BN statistics + the adjusted head), committed under dmad_hip/data/ so that the build container, the
fixtures' generators and the GPU box load bit-identical parameters.  Everything runs through the CPU oracle.

    python tests/golden/make_classifier_calib.py resnext
    python tests/golden/make_classifier_calib.py vgg_c5
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')
sys.path[:0] = [ROOT, PKG]
from dmad_hip import synth  # noqa: E402
from oracle import dmad_oracle as orc  # noqa: E402

DATA = os.path.join(PKG, 'dmad_hip', 'data')
TARGET_LOGIT_STD = 1.5            # pooled logit spread of the VGG headline at sigma = 0.5 (profiles/r03e_flip_study_f16.json)


def purified_specs(n_per_cell, sigmas=(0.25, 0.5, 1.0), clips=(0, 1, 2, 3), seed=11):
    """mel-dB spectrograms of one-shot purified noisy clips, the way RobustCertificate.smooth_predict produces them
    (certified_robust.py:46-54, diffwave_ddpm.py:174-182): [n, 1, 32, 32], plus the (clip, sigma) of every row."""
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    w = orc.folded_weights(synth.wavenet_state_dict(1234))
    g = torch.Generator().manual_seed(seed)
    rows, tags = [], []
    for ci in clips:
        clip = torch.from_numpy(synth.synthetic_clip(ci))[None]            # [1, 1, L]
        for sigma in sigmas:
            abar = 1.0 / (1.0 + sigma ** 2)
            t_star = orc.compute_t_star(hp['Alpha_bar'], sigma)
            dw = orc.DiffWaveOracle(w, hp, reverse_timestep=t_star)
            x = clip.repeat(n_per_cell, 1, 1)
            x = (abar ** 0.5) * (x + sigma * torch.randn(x.shape, generator=g))
            t0 = time.time()
            with torch.no_grad():
                x0 = dw.one_shot_denoise(x)
            rows.append(orc.mel_db(x0))
            tags += [(ci, sigma)] * n_per_cell
            print('  purified clip %d sigma %.2f (t* = %d): %d samples in %.0f s' % (ci, sigma, t_star, n_per_cell, time.time() - t0), flush=True)
    return torch.cat(rows), tags


def resnext_calibrate(sd, x):
    """One pass of models/resnext.py's forward (resnext.py:56-65,133-142) in which every BatchNorm takes the statistics of
    this batch (and keeps them as its running statistics).  Returns (new statistics, pooled features [n, 1024])."""
    T = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    out = {}

    def bn(h, p):
        m, v = h.mean((0, 2, 3)), h.var((0, 2, 3), unbiased=False)
        out[p + '.running_mean'], out[p + '.running_var'] = m.numpy().astype(np.float32), v.numpy().astype(np.float32)
        return F.batch_norm(h, m, v, T[p + '.weight'], T[p + '.bias'], False, 0.0, 1e-5)

    h = F.relu(bn(F.conv2d(x, T['conv_1_3x3.weight'], None, 1, 1), 'bn_1'))
    for st in (1, 2, 3):
        for k in range(3):
            p = 'stage_%d.stage_%d_bottleneck_%d.' % (st, st, k)
            stride = 2 if (k == 0 and st > 1) else 1
            b = F.relu(bn(F.conv2d(h, T[p + 'conv_reduce.weight']), p + 'bn_reduce'))
            b = F.relu(bn(F.conv2d(b, T[p + 'conv_conv.weight'], None, stride, 1, 1, 8), p + 'bn'))
            b = bn(F.conv2d(b, T[p + 'conv_expand.weight']), p + 'bn_expand')
            r = h
            if p + 'shortcut.shortcut_conv.weight' in T:
                r = bn(F.conv2d(h, T[p + 'shortcut.shortcut_conv.weight'], None, stride), p + 'shortcut.shortcut_bn')
            h = F.relu(r + b)
    return out, F.avg_pool2d(h, 8, 1).view(-1, 1024)


def report(name, logits, tags):
    lg = logits.numpy().astype(np.float64)
    srt = np.sort(lg, 1)
    print('%s: pooled logit std %.3f, median top-2 margin %.3f' % (name, lg.std(), np.median(srt[:, -1] - srt[:, -2])))
    for cell in sorted(set(tags)):
        rows = [i for i, t in enumerate(tags) if t == cell]
        votes = np.bincount(lg[rows].argmax(1), minlength=10)
        m = srt[rows, -1] - srt[rows, -2]
        print('   clip %d sigma %.2f: votes %s  margin median %.3f  below 0.05: %d/%d' % (cell[0], cell[1], votes.tolist(), np.median(m), (m < 0.05).sum(), len(rows)))


def main_resnext():
    sd = synth.resnext29_state_dict(2929, calibrated=False)
    spec, tags = purified_specs(int(os.environ.get('N_PER_CELL', 4)))
    with torch.no_grad():
        stats, feat = resnext_calibrate(sd, spec)
        W, b = torch.from_numpy(sd['classifier.weight']), torch.from_numpy(sd['classifier.bias'])
        report('BN-calibrated, random head', F.linear(feat, W, b), tags)
        # The pooled features move with the noise level (sigma changes the spectrogram's energy), and a random head turns that
        # shift into a class: every sample of a sigma would vote the same label.  A trained head is insensitive to the noise level
        # it was trained under, so the sigma-mean directions are projected out of the head's rows; what is left of a logit is
        # driven by the sample's own noise draw (and its clip).
        sig = sorted(set(t[1] for t in tags))
        mus = torch.stack([feat[[i for i, t in enumerate(tags) if t[1] == s_]].mean(0) for s_ in sig])
        ref = sig.index(0.5) if 0.5 in sig else 0
        D = torch.stack([mus[i] - mus[ref] for i in range(len(sig)) if i != ref], 1).double()       # [1024, nsigma - 1]
        Q, _ = torch.linalg.qr(D)
        Wp = (W.double() - (W.double() @ Q) @ Q.T).float()
        mu = mus[ref]
        z = F.linear(feat - mu, Wp)
        cell = {c: [i for i, t in enumerate(tags) if t == c] for c in set(tags)}
        within = torch.cat([z[r] - z[r].mean(0) for r in cell.values()])
        s = TARGET_LOGIT_STD / float(within.std())
        W2 = (Wp * s).float()
        b2 = (b - F.linear(mu[None], W2)[0]).float()
        report('noise-level directions projected out, centred + scaled head (x %.2f)' % s, F.linear(feat, W2, b2), tags)
    stats['classifier.weight'] = W2.numpy().astype(np.float32)
    stats['classifier.bias'] = b2.numpy().astype(np.float32)
    path = os.path.join(DATA, 'resnext29_calib_seed2929.npz')
    np.savez_compressed(path, **stats)
    print('wrote', len(stats), 'arrays ->', path, os.path.getsize(path), 'bytes')
    # the committed parameters, on held-out draws
    sd2 = synth.resnext29_state_dict(2929)
    spec2, tags2 = purified_specs(int(os.environ.get('N_CHECK', 8)), sigmas=(0.5,), clips=(0, 1), seed=12)
    with torch.no_grad():
        report('held-out (calibrated state dict)', orc.resnext29_forward(sd2, spec2), tags2)


def c5_specs(n_per_clip, clips=(0, 1, 2, 3), sigma=0.5, t_star=25, seed=13):
    """The spectrograms the spec-domain chain hands the classifier (include/dmad.h, dmad_spec_smooth_votes): mel-dB of the
    noisy clip, standardised, diffused to t*, t* + 1 p_sample steps of the seeded UNet (seed 31), un-standardised."""
    layout = synth.unet_layout()
    usd = synth.unet_state_dict(31)
    gd = orc.GaussianDiffusionOracle(1000)
    g = torch.Generator().manual_seed(seed)
    rows, tags = [], []
    model = lambda x, t: orc.unet_forward(usd, x, torch.full((x.shape[0],), t, dtype=torch.long), layout)
    for ci in clips:
        clip = torch.from_numpy(synth.synthetic_clip(ci))[None]
        x = clip.repeat(n_per_clip, 1, 1)
        x = x + sigma * torch.randn(x.shape, generator=g)                 # no wave denoiser: no sqrt(alpha_bar*) scale
        t0 = time.time()
        with torch.no_grad():
            s = orc.melspec_standardize(orc.mel_db(x))
            s = gd.q_sample(s, t_star, torch.randn(s.shape, generator=g))
            for t in range(t_star, -1, -1):
                s, _ = gd.p_sample(model, s, t, torch.randn(s.shape, generator=g))
            rows.append(orc.melspec_inv_standardize(s))
        tags += [(ci, sigma)] * n_per_clip
        print('  spec chain clip %d: %d samples in %.0f s' % (ci, n_per_clip, time.time() - t0), flush=True)
    return torch.cat(rows), tags


def vgg_calibrate(sd, x):
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
    return out


def main_vgg_c5():
    sd = synth.vgg19_bn_state_dict(4321, calibrated=False)
    spec, tags = c5_specs(int(os.environ.get('N_PER_CLIP', 8)))
    print('purified spectrograms: mean %.2f dB, std %.2f dB' % (float(spec.mean()), float(spec.std())))
    with torch.no_grad():
        stats = vgg_calibrate(sd, spec)
        sd.update(stats)
        report('VGG19_bn on the spec chain, calibration set', orc.vgg19_bn_forward(sd, spec), tags)
    path = os.path.join(DATA, 'vgg19_bn_calib_c5_seed4321.npz')
    np.savez_compressed(path, **stats)
    print('wrote', len(stats), 'arrays ->', path, os.path.getsize(path), 'bytes')
    spec2, tags2 = c5_specs(int(os.environ.get('N_CHECK', 8)), clips=(0, 1), seed=14)
    with torch.no_grad():
        report('held-out (calibrated state dict)', orc.vgg19_bn_forward(synth.vgg19_bn_state_dict(4321, calibrated='c5'), spec2), tags2)


if __name__ == '__main__':
    torch.set_num_threads(8)
    which = sys.argv[1] if len(sys.argv) > 1 else ''
    if which == 'resnext':
        main_resnext()
    elif which == 'vgg_c5':
        main_vgg_c5()
    else:
        sys.exit(__doc__)
