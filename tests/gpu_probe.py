#!/usr/bin/env python3
"""Development probe for the GPU box: runs each stage of the HIP path against the golden fixtures /
the oracle and prints error statistics (keeps going after a failing stage).  Test infrastructure (it imports the oracle, which
only tests/, smoke() and bench.py's cpu_baseline may do): `python tests/gpu_probe.py [stage ...]`; not collected by pytest."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
G = os.path.join(ROOT, 'tests', 'golden')

from dmad_hip import engine as E, synth   # noqa: E402
from oracle import dmad_oracle as orc      # noqa: E402

want = set(sys.argv[1:])


def section(name):
    def deco(fn):
        if want and name not in want:
            return fn
        print('==== %s' % name, flush=True)
        t0 = time.time()
        try:
            fn()
        except Exception:
            traceback.print_exc()
        torch.cuda.synchronize()
        print('     (%.1fs)' % (time.time() - t0), flush=True)
        return fn
    return deco


def stats(name, got, ref):
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    print('  %-28s max|err| %.3e  rel-to-max %.3e  rms-rel %.3e  (ref max %.3e, nan %d)' % (
        name, err.max(), err.max() / (np.abs(ref).max() + 1e-30),
        np.sqrt((err ** 2).mean()) / (np.sqrt((ref ** 2).mean()) + 1e-30), np.abs(ref).max(), int(np.isnan(got).sum())), flush=True)


print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
sd = synth.wavenet_state_dict(1234)
vsd = synth.vgg19_bn_state_dict(4321)
engines = {}


@section('create')
def _():
    for name, prec in (('fp32', E.FP32), ('bf16', E.BF16)):
        t0 = time.time()
        eng = E.Engine(max_batch=4, precision=prec)
        eng.load_wavenet(sd)
        eng.load_vgg19_bn(vsd)
        engines[name] = eng
        print('  engine %s: %.1f MB device, %.1fs' % (name, eng.device_bytes() / 1e6, time.time() - t0), flush=True)


@section('philox')
def _():
    eng = engines['fp32']
    raw = eng.philox_raw(0x123456789ABCDEF, 77, 3, 1000).cpu().numpy().view(np.uint32).reshape(-1, 4)
    ctr, key = orc.philox_counters(0x123456789ABCDEF, 77, 3, 1000)
    print('  raw words bit-exact:', bool((raw == orc.philox4x32_10(ctr, key)).all()))
    z = eng.philox_normal(5, 10, 0, 2).cpu().numpy()
    stats('normal vs f64 box-muller', z[1], orc.philox_normal(5, 11, 0, 16000))
    print('  mean %.4f std %.4f' % (z.mean(), z.std()))


@section('mel')
def _():
    eng = engines['fp32']
    x = np.stack([synth.synthetic_clip(i) for i in range(3)] + [np.zeros((1, 16000), np.float32)])
    got = eng.mel_db(torch.from_numpy(x).cuda()).cpu().numpy()
    ref64 = orc.mel_db_f64(x)
    ref32 = orc.mel_db(torch.from_numpy(x)).numpy()
    stats('mel_db vs f64', got[:3], ref64[:3]); stats('mel_db vs torch', got[:3], ref32[:3])
    print('  silence all -100:', bool((got[3] == -100).all()))


@section('vgg')
def _():
    z = np.load(os.path.join(G, 'classifiers.npz'))
    for name in ('fp32', 'bf16'):
        got = engines[name].classify(torch.from_numpy(z['spec_in']).cuda()).cpu().numpy()
        stats('vgg logits (%s engine)' % name, got, z['vgg_logits'])
    print('  argmax', got.argmax(1), z['vgg_logits'].argmax(1))


@section('wavenet_fp32')
def _():
    z = np.load(os.path.join(G, 'wavenet_full.npz'))
    x_t = torch.from_numpy(z['x_t']).cuda()
    t0 = time.time()
    got = engines['fp32'].wavenet_eps(x_t, int(z['t']))
    torch.cuda.synchronize()
    print('  fp32 eps B=2: %.2fs' % (time.time() - t0))
    stats('eps fp32', got.cpu().numpy(), z['eps'][:, 0])


@section('wavenet_bf16')
def _():
    z = np.load(os.path.join(G, 'wavenet_full.npz'))
    x_t = torch.from_numpy(z['x_t']).cuda()
    got = engines['bf16'].wavenet_eps(x_t, int(z['t']))
    torch.cuda.synchronize()
    stats('eps bf16', got.cpu().numpy(), z['eps'][:, 0])
    a = got.cpu().numpy()
    print('  corr', np.corrcoef(a.reshape(-1), z['eps'][:, 0].reshape(-1))[0, 1])


@section('samplers')
def _():
    z = np.load(os.path.join(G, 'samplers.npz'))
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    for name in ('fp32', 'bf16'):
        eng = engines[name]
        t = 65
        got = eng.one_shot(torch.from_numpy(z['x_t']).cuda(), t, float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t]))
        stats('one_shot t66 (%s)' % name, got.cpu().numpy(), z['one_shot_t66'][:, 0])


@section('votes')
def _():
    z = np.load(os.path.join(G, 'smooth_predict.npz'))
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    t = 65
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    torch.manual_seed(int(z['vgg_seed']))
    delta = torch.cat([torch.normal(0, 0.5, size=(b, 1, 16000)) for b in (16, 16, 8)]).cuda()
    for name in ('fp32', 'bf16'):
        eng = engines[name]
        counts, logits, _ = eng.smooth_votes(clip, 0.5, float(torch.tensor((1 / 1.25) ** 0.5)), t, float((1 / ab).sqrt()[t]),
                                             float((1 / ab - 1).sqrt()[t]), 40, batch=4, delta=delta, want_logits=True)
        print('  %s counts %s  ref %s' % (name, counts.cpu().tolist(), z['vgg_counts'].tolist()))
        stats('logits (%s)' % name, logits.cpu().numpy(), z['vgg_logits'])
        ref = z['vgg_logits']; srt = np.sort(ref, 1)
        print('   ref min margin %.4f; flips %d' % ((srt[:, -1] - srt[:, -2]).min(), int((logits.cpu().numpy().argmax(1) != ref.argmax(1)).sum())))


@section('perf')
def _():
    eng = E.Engine(max_batch=64, precision=E.BF16)
    eng.load_wavenet(sd); eng.load_vgg19_bn(vsd)
    x = torch.randn(64, 16000, device='cuda') * 0.3
    for B in (16, 64):
        eng.wavenet_eps(x[:B], 65); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            eng.wavenet_eps(x[:B], 65)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        print('  bf16 wavenet B=%d: %.1f ms -> %.1f clips/s, %.1f TFLOP/s' % (B, dt * 1e3, B / dt, B * 606.1e9 / dt / 1e12))
    for layer in (0, 5, 11):
        ms = eng.time_layer(layer, 64, 10)
        print('  layer %d (d=%d) B=64: %.3f ms/launch -> %.1f TFLOP/s (layer flops incl. res conv)' % (
            layer, 1 << layer, ms, 64 * 16000 * 2 * (512 * 768 + 256 * 256) / ms / 1e9))
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG); ab = hp['Alpha_bar']; t = 65
    args = (clip, 0.5, float(torch.tensor((1 / 1.25) ** 0.5)), t, float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t]))
    eng.smooth_votes(*args, 64); torch.cuda.synchronize()
    t0 = time.time()
    c, _, _ = eng.smooth_votes(*args, 512)
    c = c.cpu()
    dt = time.time() - t0
    print('  smooth_votes N=512: %.2fs -> %.1f clips/s, counts %s' % (dt, 512 / dt, c.tolist()))
