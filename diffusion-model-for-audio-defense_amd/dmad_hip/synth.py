"""Seeded synthetic weights and inputs.

The pretrained DiffWave / ConvNets_SpeechCommands checkpoints are not available offline
(SURVEY F9), so every parity case runs on weights produced HERE from a seed.  The same
generator feeds (a) the reference modules inside tests/golden/make_golden.py, (b) the CPU
oracle and (c) the HIP engine, so all three see bit-identical fp32 parameters.

Key names and shapes follow the reference checkpoints:
  WaveNet  : diffusion_models/DiffWave_Unconditional/WaveNet.py:23-34,53-72,138-162
  VGG19_bn : audio_models/ConvNets_SpeechCommands/models/vgg.py:31-52,69-89,190-201
Only numpy's Generator(PCG64) is used, so the stream does not depend on torch.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

WAVENET_CONFIG = dict(in_channels=1, res_channels=256, skip_channels=256, out_channels=1,
                      num_res_layers=36, dilation_cycle=12,
                      diffusion_step_embed_dim_in=128,
                      diffusion_step_embed_dim_mid=512,
                      diffusion_step_embed_dim_out=512)
DIFFUSION_CONFIG = dict(T=200, beta_0=0.0001, beta_T=0.02)

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
             512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _wn_pair(rng, prefix, out_c, in_c, k, sd):
    """weight-normed conv: weight_v ~ N(0, 2/fan_in), weight_g = |v| * U(0.8, 1.2)."""
    fan_in = in_c * k
    v = rng.standard_normal((out_c, in_c, k)) * np.sqrt(2.0 / fan_in)
    norm = np.sqrt((v.astype(np.float32) ** 2).sum(axis=(1, 2), keepdims=True))
    g = norm * rng.uniform(0.8, 1.2, size=(out_c, 1, 1))
    sd[prefix + '.bias'] = _f32(rng.standard_normal(out_c) * 0.02)
    sd[prefix + '.weight_g'] = _f32(g)
    sd[prefix + '.weight_v'] = _f32(v)


def _linear(rng, prefix, out_f, in_f, sd):
    bound = 1.0 / np.sqrt(in_f)
    sd[prefix + '.weight'] = _f32(rng.uniform(-bound, bound, size=(out_f, in_f)))
    sd[prefix + '.bias'] = _f32(rng.uniform(-bound, bound, size=(out_f,)))


def wavenet_state_dict(seed: int = 1234, cfg: dict | None = None) -> "OrderedDict[str, np.ndarray]":
    """fp32 numpy state dict with the reference's 408 keys (for the default config)."""
    cfg = dict(WAVENET_CONFIG if cfg is None else cfg)
    C, S = cfg['res_channels'], cfg['skip_channels']
    E_in, E_mid, E_out = (cfg['diffusion_step_embed_dim_in'], cfg['diffusion_step_embed_dim_mid'],
                          cfg['diffusion_step_embed_dim_out'])
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    _wn_pair(rng, 'init_conv.0.conv', C, cfg['in_channels'], 1, sd)
    _linear(rng, 'residual_layer.fc_t1', E_mid, E_in, sd)
    _linear(rng, 'residual_layer.fc_t2', E_out, E_mid, sd)
    for n in range(cfg['num_res_layers']):
        p = 'residual_layer.residual_blocks.%d' % n
        _linear(rng, p + '.fc_t', C, E_out, sd)
        _wn_pair(rng, p + '.dilated_conv_layer.conv', 2 * C, C, 3, sd)
        _wn_pair(rng, p + '.res_conv', C, C, 1, sd)
        _wn_pair(rng, p + '.skip_conv', S, C, 1, sd)
    _wn_pair(rng, 'final_conv.0.conv', S, S, 1, sd)
    # the reference zero-initialises this conv (WaveNet.py:39-44): re-initialise, SURVEY F6
    sd['final_conv.2.conv.weight'] = _f32(rng.standard_normal((cfg['out_channels'], S, 1)) * 0.05)
    sd['final_conv.2.conv.bias'] = _f32(rng.standard_normal(cfg['out_channels']) * 0.02)
    return sd


def vgg19_bn_state_dict(seed: int = 4321, num_classes: int = 10, in_channels: int = 1, calibrated: bool = True):
    """fp32 numpy state dict for models/vgg.py vgg19_bn(num_classes=10, in_channels=1).

    With `calibrated` (and seed 4321) the BatchNorm running statistics come from the committed data
    file dmad_hip/data/vgg19_bn_calib_seed4321.npz (one calibration pass over noisy synthetic clips,
    tests/golden/make_vgg_calib.py) so that the synthetic classifier is not degenerate and the Monte
    Carlo votes spread over several classes, like a trained checkpoint's would.
    """
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    idx, cin, first = 0, in_channels, True
    for v in VGG19_CFG:
        if v == 'M':
            idx += 1
            continue
        n = 9 * v
        w = rng.standard_normal((v, cin, 3, 3)) * np.sqrt(2.0 / n)
        sd['features.%d.weight' % idx] = _f32(w)
        sd['features.%d.bias' % idx] = _f32(rng.standard_normal(v) * 0.05)
        b = idx + 1
        sd['features.%d.weight' % b] = _f32(rng.uniform(0.6, 1.4, size=v))
        sd['features.%d.bias' % b] = _f32(rng.standard_normal(v) * 0.2)
        if first:
            # conv(1->64) of a dB image: response ~ sum(w)*mean_dB, spread ~ |w|*std_dB
            wsum = w.reshape(v, -1).sum(1)
            wl2 = np.sqrt((w.reshape(v, -1) ** 2).sum(1))
            sd['features.%d.running_mean' % b] = _f32(wsum * (-20.0))
            sd['features.%d.running_var' % b] = _f32((wl2 * 12.0) ** 2 + 1.0)
            first = False
        else:
            sd['features.%d.running_mean' % b] = _f32(rng.standard_normal(v) * 0.3)
            sd['features.%d.running_var' % b] = _f32(rng.uniform(0.5, 1.5, size=v))
        sd['features.%d.num_batches_tracked' % b] = np.asarray(1000, dtype=np.int64)
        idx += 3
        cin = v
    for i, (o, k) in zip((0, 3, 6), ((4096, 512), (4096, 4096), (num_classes, 4096))):
        sd['classifier.%d.weight' % i] = _f32(rng.standard_normal((o, k)) * np.sqrt(2.0 / k))
        sd['classifier.%d.bias' % i] = _f32(rng.standard_normal(o) * 0.1)
    if calibrated and seed == 4321:
        # calibrated == 'c5': the statistics measured on the spectrograms the SPEC-domain chain (BASELINE C5) hands the classifier
        _load_calibration(sd, 'vgg19_bn_calib_c5_seed4321.npz' if calibrated == 'c5' else 'vgg19_bn_calib_seed4321.npz')
    return sd


def _load_calibration(sd, name):
    """Overwrites entries of `sd` with the committed calibration data dmad_hip/data/<name> (BatchNorm running statistics, and for
    ResNeXt29 the centred / scaled head): tests/golden/make_vgg_calib.py, tests/golden/make_classifier_calib.py."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', name)
    with np.load(path) as z:
        for k in z.files:
            assert sd[k].shape == z[k].shape, k
            sd[k] = _f32(z[k])


def synthetic_clip(seed: int = 0, length: int = 16000) -> np.ndarray:
    """A 1 s 'utterance': decaying harmonic bursts + weak noise, in [-1, 1] (fp32 [1, L])."""
    rng = np.random.default_rng(seed)
    t = np.arange(length) / 16000.0
    x = np.zeros(length)
    for _ in range(4):
        f0 = rng.uniform(110, 420)
        t0 = rng.uniform(0.05, 0.7)
        dur = rng.uniform(0.08, 0.25)
        env = np.exp(-0.5 * ((t - t0) / dur) ** 2)
        for h in range(1, 6):
            x += env * rng.uniform(0.2, 1.0) / h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28))
    x += 0.01 * rng.standard_normal(length)
    x = 0.5 * x / np.abs(x).max()
    return _f32(x[None, :])


def resnext29_state_dict(seed: int = 2929, num_classes: int = 10, in_channels: int = 1, calibrated: bool = True):
    """fp32 numpy state dict for models/resnext.py CifarResNeXt(nlabels=10, cardinality=8, depth=29, base_width=64,
    widen_factor=4, in_channels=1): kaiming (fan_out) conv weights, BatchNorm affine/statistics drawn so that the
    29-layer stack stays O(1) on dB-scaled mel images (first BN sized for inputs around -20 dB +- 12, like the VGG's).

    With `calibrated` (and seed 2929) the BatchNorm running statistics and the linear head come from the committed data file
    dmad_hip/data/resnext29_calib_seed2929.npz (one calibration pass over mel spectrograms of purified noisy synthetic clips,
    head centred on the calibration set and scaled to the VGG headline's logit spread: tests/golden/make_classifier_calib.py), so
    that the Monte Carlo votes spread over several classes with margins of order one, like a trained checkpoint's — the random
    statistics alone vote one class on every sample, which exercises no recheck tier."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def conv(name, cout, cin_g, k):
        w = rng.standard_normal((cout, cin_g, k, k)) * np.sqrt(2.0 / (cout * k * k))      # fan_out mode
        sd[name + '.weight'] = _f32(w)
        return w

    def bn(name, c, gamma=(0.6, 1.4), mean=None, var=None):
        sd[name + '.weight'] = _f32(rng.uniform(*gamma, size=c))
        sd[name + '.bias'] = _f32(rng.standard_normal(c) * 0.1)
        sd[name + '.running_mean'] = _f32(rng.standard_normal(c) * 0.2 if mean is None else mean)
        sd[name + '.running_var'] = _f32(rng.uniform(0.5, 1.5, size=c) if var is None else var)
        sd[name + '.num_batches_tracked'] = np.asarray(1000, dtype=np.int64)

    w = conv('conv_1_3x3', 64, in_channels, 3)
    wsum = w.reshape(64, -1).sum(1)
    wl2 = np.sqrt((w.reshape(64, -1) ** 2).sum(1))
    bn('bn_1', 64, mean=wsum * (-20.0), var=(wl2 * 12.0) ** 2 + 1.0)
    stages = [64, 256, 512, 1024]
    for st in (1, 2, 3):
        for k in range(3):
            cin = stages[st - 1] if k == 0 else stages[st]
            cout = stages[st]
            D = 8 * (64 * cout // 256)
            p = 'stage_%d.stage_%d_bottleneck_%d.' % (st, st, k)
            # a conv's output variance under this init is ~ 2 * fan_in / fan_out of its input's: the BN statistics follow it
            conv(p + 'conv_reduce', D, cin, 1)
            bn(p + 'bn_reduce', D, var=rng.uniform(0.5, 1.5, size=D) * (2.0 * cin / D))
            conv(p + 'conv_conv', D, D // 8, 3)
            bn(p + 'bn', D, var=rng.uniform(0.5, 1.5, size=D) * (2.0 / 8))
            conv(p + 'conv_expand', cout, D, 1)
            bn(p + 'bn_expand', cout, gamma=(0.3, 0.7), var=rng.uniform(0.5, 1.5, size=cout) * (2.0 * D / cout))
            if cin != cout:
                conv(p + 'shortcut.shortcut_conv', cout, cin, 1)
                bn(p + 'shortcut.shortcut_bn', cout, gamma=(0.5, 0.9), var=rng.uniform(0.5, 1.5, size=cout) * (2.0 * cin / cout))
    sd['classifier.weight'] = _f32(rng.standard_normal((num_classes, 1024)) * np.sqrt(2.0 / 1024))
    sd['classifier.bias'] = _f32(rng.standard_normal(num_classes) * 0.1)
    if calibrated and seed == 2929 and num_classes == 10 and in_channels == 1:
        _load_calibration(sd, 'resnext29_calib_seed2929.npz')
    return sd


UNET_CONFIG = dict(in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3, attention_resolutions=(2, 4),
                   channel_mult=(1, 2, 2, 2), num_heads=4, use_scale_shift_norm=True)


def unet_layout(cfg=None):
    """Block list of improved_diffusion.unet.UNetModel (unet.py:278-421) for `cfg`: a list of
    (state-dict prefix, kind, cin, cout) with kind in conv_in / res / attn / down / up / out, in forward order of the
    three containers.  Shared by the synthetic weights, the oracle restatement and the HIP engine's loader."""
    c = dict(UNET_CONFIG)
    c.update(cfg or {})
    mc, nrb, mult, att = c['model_channels'], c['num_res_blocks'], c['channel_mult'], c['attention_resolutions']
    inp, mid, outp = [[('input_blocks.0.0', 'conv_in', c['in_channels'], mc)]], [], []
    chans, ch, ds = [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            i = len(inp)
            blk = [('input_blocks.%d.0' % i, 'res', ch, m * mc)]
            ch = m * mc
            if ds in att:
                blk.append(('input_blocks.%d.1' % i, 'attn', ch, ch))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([('input_blocks.%d.0' % len(inp), 'down', ch, ch)])
            chans.append(ch)
            ds *= 2
    mid = [('middle_block.0', 'res', ch, ch), ('middle_block.1', 'attn', ch, ch), ('middle_block.2', 'res', ch, ch)]
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            j = len(outp)
            blk = [('output_blocks.%d.0' % j, 'res', ch + chans.pop(), mc * m)]
            ch = mc * m
            if ds in att:
                blk.append(('output_blocks.%d.1' % j, 'attn', ch, ch))
            if level and i == nrb:
                blk.append(('output_blocks.%d.%d' % (j, len(blk)), 'up', ch, ch))
                ds //= 2
            outp.append(blk)
    return c, inp, mid, outp


def unet_state_dict(seed: int = 5252, cfg=None):
    """fp32 numpy state dict for improved_diffusion.unet.UNetModel with UNET_CONFIG (52.5 M parameters): the reference's
    parameter names; conv / linear weights N(0, g / fan_in) with gains that keep the 35-block stack O(1); the
    zero-initialised modules of the reference (out_layers.3, proj_out, out.2 — non-zero after training) get small
    random weights so that every path contributes."""
    c, inp, mid, outp = unet_layout(cfg)
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    mc = c['model_channels']
    ted = 4 * mc

    def dense(name, shape, gain):
        fan_in = int(np.prod(shape[1:]))
        sd[name + '.weight'] = _f32(rng.standard_normal(shape) * np.sqrt(gain / fan_in))
        sd[name + '.bias'] = _f32(rng.standard_normal(shape[0]) * 0.02)

    def gn(name, ch):
        sd[name + '.weight'] = _f32(rng.uniform(0.7, 1.3, size=ch))
        sd[name + '.bias'] = _f32(rng.standard_normal(ch) * 0.1)

    dense('time_embed.0', (ted, mc), 1.0)
    dense('time_embed.2', (ted, ted), 1.0)
    for blk in inp + [mid] + outp:
        for prefix, kind, cin, cout in blk:
            if kind == 'conv_in':
                dense(prefix, (cout, cin, 3, 3), 1.0)
            elif kind == 'res':
                gn(prefix + '.in_layers.0', cin)
                dense(prefix + '.in_layers.2', (cout, cin, 3, 3), 2.0)
                dense(prefix + '.emb_layers.1', (2 * cout, ted), 0.5)
                gn(prefix + '.out_layers.0', cout)
                dense(prefix + '.out_layers.3', (cout, cout, 3, 3), 0.3)
                if cin != cout:
                    dense(prefix + '.skip_connection', (cout, cin, 1, 1), 1.0)
            elif kind == 'attn':
                gn(prefix + '.norm', cin)
                dense(prefix + '.qkv', (3 * cin, cin, 1), 1.0)
                dense(prefix + '.proj_out', (cin, cin, 1), 0.3)
            elif kind == 'down':
                dense(prefix + '.op', (cout, cin, 3, 3), 1.0)
            elif kind == 'up':
                dense(prefix + '.conv', (cout, cin, 3, 3), 1.0)
    gn('out.0', mc)
    dense('out.2', (c['out_channels'], mc, 3, 3), 1.0)
    return sd
