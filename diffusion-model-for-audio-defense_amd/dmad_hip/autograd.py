"""The differentiation branch of the host mirrors ("Autograd").

The HIP engine is inference-only.  The reference's white-box attack drivers differentiate THROUGH the system
(`AcousticSystem.forward` with `x.requires_grad`, adaptive_attack_eval.py:176,262; kws_adaptive_attack_eval.py:111 builds the
DDPM `DiffWave` as defender), so the survey's boundary asks the mirrors to take a torch restatement on exactly that branch:
`torch.is_grad_enabled() and x.requires_grad`.  This module holds those restatements — plain differentiable torch ops on the
caller's CUDA tensors, fed with the same folded weights the engine packs:

    wavenet_eps   WaveNet_Speech_Commands.forward   DiffWave_Unconditional/WaveNet.py:75-97,120-135,164-172; util.py:68-93
    mel_db        MelSpectrogram + AmplitudeToDB    certified_robustness_eval.py:85-87 (torchaudio 0.11 semantics)

(the classifiers' own nn layers are their restatement: models/vgg.py:48-52, models/resnext.py:47-62,133-142).
Scope: gradients only.  Nothing here runs when no gradient is requested — inference calls go to libdmad_hip.so and fail loudly
without it — tensors must live on the GPU like everywhere else in this package, nothing here is timed by bench.py, and nothing
here imports `oracle/` (test infrastructure).  tests/test_gpu_parity.py::test_autograd_branch checks the forward values of this
branch against the HIP fp32 path and its gradients against finite differences taken WITH the HIP fp32 path.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from ._lib import DmadError


def needs_grad(x) -> bool:
    return isinstance(x, torch.Tensor) and torch.is_grad_enabled() and x.requires_grad


def _require_cuda(x):
    if not x.is_cuda:
        raise DmadError('input must live on the GPU (this package has no CPU path, the differentiation branch included)')


class FoldedWaveNet:
    """Folded fp32 WaveNet weights (dmad_hip.engine.fold_wavenet_state_dict) as device tensors, created on first use."""

    def __init__(self, folded: dict, num_res_layers: int, dilation_cycle: int):
        self.host, self.NL, self.cycle = folded, int(num_res_layers), int(dilation_cycle)
        self._dev = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in self.host.items()}
            for n in range(self.NL):                        # conv1d weight shapes: [out, in, k]
                w['res.%d.w' % n] = w['res.%d.w' % n].reshape(256, 256, 1)
                w['skip.%d.w' % n] = w['skip.%d.w' % n].reshape(256, 256, 1)
            w['init.w'] = w['init.w'].reshape(256, 1, 1)
            w['f0.w'] = w['f0.w'].reshape(256, 256, 1)
            w['f2.w'] = w['f2.w'].reshape(1, 256, 1)
            self._dev[key] = w
        return self._dev[key]


def wavenet_eps(fw: FoldedWaveNet, audio: torch.Tensor, t: int) -> torch.Tensor:
    """eps = WaveNet((audio [B,1,L], t * ones)), differentiable in `audio`."""
    _require_cuda(audio)
    w = fw.on(audio.device)
    B = audio.shape[0]
    x = torch.relu(F.conv1d(audio, w['init.w'], w['init.b']))
    # calc_diffusion_step_embedding (util.py:84-91): cat(sin, cos)(t * exp(-j ln(1e4) / 63)), j < 64
    half = w['fc_t1.w'].shape[1] // 2
    freq = torch.exp(torch.arange(half, device=audio.device) * -(math.log(10000.0) / (half - 1))).float()
    arg = float(t) * freq
    emb = torch.cat([torch.sin(arg), torch.cos(arg)]).unsqueeze(0).expand(B, -1)
    emb = F.silu(F.linear(emb, w['fc_t1.w'], w['fc_t1.b']))
    emb = F.silu(F.linear(emb, w['fc_t2.w'], w['fc_t2.b']))
    skip = 0
    for n in range(fw.NL):
        d = 2 ** (n % fw.cycle)
        h = x + F.linear(emb, w['fc_t.%d.w' % n], w['fc_t.%d.b' % n]).view(B, -1, 1)       # the reference's in-place alias
        H = F.conv1d(h, w['dil.%d.w' % n], w['dil.%d.b' % n], dilation=d, padding=d)
        g = torch.tanh(H[:, :256]) * torch.sigmoid(H[:, 256:])
        x = (h + F.conv1d(g, w['res.%d.w' % n], w['res.%d.b' % n])) * math.sqrt(0.5)
        skip = skip + F.conv1d(g, w['skip.%d.w' % n], w['skip.%d.b' % n])
    y = torch.relu(F.conv1d(skip * math.sqrt(1.0 / fw.NL), w['f0.w'], w['f0.b']))
    return F.conv1d(y, w['f2.w'], w['f2.b'])


_MEL_CACHE = {}


def _mel_constants(device):
    key = str(device)
    if key not in _MEL_CACHE:
        def hz2mel(f):
            return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3))

        def mel2hz(m):
            return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)
        pts = mel2hz(np.linspace(hz2mel(np.float64(0.0)), hz2mel(np.float64(8000.0)), 34))
        freqs = np.linspace(0.0, 8000.0, 1025)
        down = (freqs[None, :] - pts[:-2, None]) / (pts[1:-1, None] - pts[:-2, None])
        up = (pts[2:, None] - freqs[None, :]) / (pts[2:, None] - pts[1:-1, None])
        fb = np.maximum(0.0, np.minimum(down, up)) * (2.0 / (pts[2:] - pts[:-2]))[:, None]           # slaney norm, [32][1025]
        _MEL_CACHE[key] = (torch.hann_window(2048, periodic=True, device=device), torch.from_numpy(fb).float().to(device))
    return _MEL_CACHE[key]


def mel_db(x: torch.Tensor) -> torch.Tensor:
    """[B,1,16000] -> [B,1,32,32] dB mel spectrogram, differentiable in x."""
    _require_cuda(x)
    win, fb = _mel_constants(x.device)
    spec = torch.stft(x[:, 0], n_fft=2048, hop_length=512, win_length=2048, window=win, center=True, pad_mode='constant',
                      normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                                        # [B, 1025, 32]
    mel = torch.matmul(fb, power)                                                  # [B, 32, 32]
    return (10.0 * torch.log10(torch.clamp(mel, min=1e-10))).unsqueeze(1)
