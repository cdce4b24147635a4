"""Host mirror of the reference's diffusion_models/diffwave_ddpm.py surface, backed by the MI355X
engine (libdmad_hip.so).  Same names, argument meaning and error behaviour as the reference:

  create_diffwave_model(model_path, config_path, reverse_timestep=25) -> DiffWave   (ref l.395-411)
  DiffWave.model((audio [B,1,L], steps [B,1])) -> eps [B,1,L]                       (ref l.158,170,178)
  DiffWave.forward / _diffusion / _reverse / compute_coefficients / compute_eps_t /
  one_shot_denoise / two_shot_denoise / fast_reverse / _predict_x0_from_eps / _predict_x1_from_eps /
  _predict_x0_from_x1 / _extract_into_tensor, .diffusion_hyperparams, .reverse_timestep     (ref l.16-249)
  ReffWave(model, diffusion_hyperparams, reverse_timestep, num_re).forward / diffusion /
  one_shot_denoise                                                                       (ref l.251-349)

The eps-network runs in hand-written HIP kernels without autograd.  There is no CPU path; inputs must be
CUDA tensors (the reference itself hard-codes .cuda()).  Callers that DIFFERENTIATE through the
purifier (the white-box attack drivers: `x.requires_grad` with gradients enabled) get the torch
restatement of dmad_hip/autograd.py on that branch — `DiffWave.forward`, `one_shot_denoise`, `compute_eps_t`
and `model((x, t))`; every other call is the HIP engine's.

Noise: the reference draws every Gaussian on the CPU default generator and copies it over
(ref l.66,100).  `noise_source='torch_cpu'` reproduces exactly that stream (parity);
`noise_source='device'` (default) draws counter-based Philox noise on the GPU."""
import json
from typing import Union

import numpy as np
import torch

from dmad_hip import autograd as _ag
from dmad_hip import engine as _eng
from .DiffWave_Unconditional.util import calc_diffusion_hyperparams


class WaveNetHIP(torch.nn.Module):
    """Stands in for WaveNet_Speech_Commands (DiffWave_Unconditional/WaveNet.py:138-172):
    callable on the tuple (audio [B,1,L], diffusion_steps [B,1]); all rows must carry the same step,
    which is what every inference caller of the reference passes (t * ones)."""

    def __init__(self, engine: "_eng.Engine", state_dict=None):
        super().__init__()
        self.engine = engine
        # the differentiation branch (dmad_hip/autograd.py) evaluates the same folded weights with torch ops
        self._folded = None
        if state_dict is not None:
            cyc = dict(engine.wavenet_geometry)['dilation_cycle']
            self._folded = _ag.FoldedWaveNet(_eng.fold_wavenet_state_dict(state_dict, engine.num_res_layers), engine.num_res_layers, cyc)

    def forward(self, input_data):
        audio, diffusion_steps = input_data
        steps = torch.as_tensor(diffusion_steps).detach().reshape(-1).float().cpu()
        t = float(steps[0])
        if not bool((steps == t).all()) or t != int(t):
            raise NotImplementedError('per-row / fractional diffusion steps are not supported by the HIP engine')
        if _ag.needs_grad(audio):
            if self._folded is None:
                raise NotImplementedError('the HIP eps-network has no autograd; build the model with create_diffwave_model(...) or '
                                          'WaveNetHIP(engine, state_dict=...) to get the torch restatement on the gradient branch')
            return _ag.wavenet_eps(self._folded, audio, int(t))
        return self.engine.wavenet_eps(audio, int(t)).unsqueeze(1)


class DiffWave(torch.nn.Module):

    def __init__(self, model, diffusion_hyperparams: dict, reverse_timestep: int = 200, grad_enable=True,
                 noise_source: str = 'device', seed: int = 0):
        super().__init__()
        self.model = model
        self.diffusion_hyperparams = diffusion_hyperparams
        self.reverse_timestep = reverse_timestep
        self.freeze = False
        self.grad_enable = grad_enable
        assert noise_source in ('device', 'torch_cpu')
        self.noise_source = noise_source
        self.seed = seed
        self._draws = 0           # sample counter for device noise

    # -- plumbing --------------------------------------------------------------------------------
    @property
    def engine(self) -> "_eng.Engine":
        return self.model.engine

    def _tables(self):
        hp = self.diffusion_hyperparams
        T, Alpha, Alpha_bar, Sigma = hp["T"], hp["Alpha"], hp["Alpha_bar"], hp["Sigma"]
        assert len(Alpha) == T
        assert len(Alpha_bar) == T
        assert len(Sigma) == T
        return T, Alpha, Alpha_bar, Sigma

    @staticmethod
    def _to_tensor(x):
        return torch.from_numpy(x) if isinstance(x, np.ndarray) else x

    def _noise(self, shape, device):
        """N(0,1) like the reference's torch.normal(0, 1, size).cuda(); None = let the engine draw Philox."""
        if self.noise_source == 'torch_cpu':
            return torch.normal(0, 1, size=tuple(shape)).to(device)
        return None

    def purify_coefficients(self):
        """(t*, c_a, c_b, c_eps[t*], c_div[t*], c_sig[t*]) of forward() = _diffusion + _reverse, from the fp32 tables exactly
        as ref l.66-67,159-160 compute them: the argument block of dmad_ddpm_purify / dmad_query_logits."""
        _, Alpha, Alpha_bar, Sigma = self._tables()
        ts = self.reverse_timestep
        return (ts, float(torch.sqrt(Alpha_bar[ts - 1])), float(torch.sqrt(1 - Alpha_bar[ts - 1])),
                [float((1 - Alpha[t]) / torch.sqrt(1 - Alpha_bar[t])) for t in range(ts)],
                [float(torch.sqrt(Alpha[t])) for t in range(ts)], [float(Sigma[t]) for t in range(ts)])

    # -- reference API ---------------------------------------------------------------------------
    def _forward_autograd(self, x_0):
        """DiffWave.forward = _diffusion + _reverse (ref l.36-104,143-164) as differentiable torch ops: the branch the white-box
        attack drivers take.  Noise: the reference's CPU stream (noise_source='torch_cpu') or the engine's Philox draws keyed as
        in the inference path (row i of this call = sample _draws + i; stream 0xD1FF for the diffusion draw, 1 + t per step)."""
        _, Alpha, Alpha_bar, Sigma = self._tables()
        assert x_0.ndim == 3
        B, dev, base = x_0.shape[0], x_0.device, self._draws

        def draw(stream):
            z = self._noise(x_0.shape, dev)
            return z if z is not None else self.engine.philox_normal(self.seed, base, stream, B).unsqueeze(1)
        ts = self.reverse_timestep
        x = torch.sqrt(Alpha_bar[ts - 1]).to(dev) * x_0 + torch.sqrt(1 - Alpha_bar[ts - 1]).to(dev) * draw(0xD1FF)
        for t in range(ts - 1, -1, -1):
            eps = self.model((x, t * torch.ones((B, 1))))
            c = ((1 - Alpha[t]) / torch.sqrt(1 - Alpha_bar[t])).to(dev)
            x = (x - c * eps) / torch.sqrt(Alpha[t]).to(dev)
            if t > 0:
                x = x + Sigma[t].to(dev) * draw(1 + t)
        self._draws += B
        return x

    def forward(self, waveforms: Union[torch.Tensor, np.ndarray]):
        waveforms = self._to_tensor(waveforms)
        if _ag.needs_grad(waveforms):
            return self._forward_autograd(waveforms)
        if self.noise_source == 'device':     # the whole chain in one library call (dmad_ddpm_purify)
            assert waveforms.ndim == 3
            ts, c_a, c_b, c_eps, c_div, c_sig = self.purify_coefficients()
            out = self.engine.ddpm_purify(waveforms, ts, c_a, c_b, c_eps, c_div, c_sig, seed=self.seed, sample0=self._draws)
            self._draws += waveforms.shape[0]
            return out.unsqueeze(1)
        base = self._draws                    # device noise: row i of this call is sample base + i in BOTH phases
        output = self._diffusion(waveforms)   # (different Philox streams), so a clip's purification does not depend on
        self._draws = base                    # the batch it is in
        output = self._reverse(output)
        return output

    @torch.no_grad()
    def _diffusion(self, x_0) -> torch.Tensor:
        x_0 = self._to_tensor(x_0)
        _, _, Alpha_bar, _ = self._tables()
        assert x_0.ndim == 3
        t = self.reverse_timestep - 1
        c_a = float(torch.sqrt(Alpha_bar[t]))
        c_b = float(torch.sqrt(1 - Alpha_bar[t]))
        z = self._noise(x_0.shape, x_0.device)
        out = self.engine.diffuse(x_0, c_a, c_b, z, seed=self.seed, sample0=self._draws)
        self._draws += x_0.shape[0]
        return out.unsqueeze(1)

    @torch.no_grad()
    def _reverse(self, x_t) -> torch.Tensor:
        x_t = self._to_tensor(x_t)
        _, Alpha, Alpha_bar, Sigma = self._tables()
        assert x_t.ndim == 3
        x = x_t.detach()[:, 0].contiguous().float().clone()
        base = self._draws
        for t in range(self.reverse_timestep - 1, -1, -1):
            c_eps = float((1 - Alpha[t]) / torch.sqrt(1 - Alpha_bar[t]))
            c_div = float(torch.sqrt(Alpha[t]))
            c_sig = float(Sigma[t]) if t > 0 else 0.0
            z = self._noise(x_t.shape, x_t.device) if t > 0 else None
            self.engine.ddpm_step(x, t, c_eps, c_div, c_sig, z, seed=self.seed, sample0=base)
        self._draws += x.shape[0]
        return x.unsqueeze(1)

    @torch.no_grad()
    def compute_coefficients(self, x_t, t: int):
        x_t = self._to_tensor(x_t)
        _, Alpha, Alpha_bar, Sigma = self._tables()
        diffusion_steps = t * torch.ones((x_t.shape[0], 1))
        epsilon_theta = self.model((x_t, diffusion_steps))
        c = ((1 - Alpha[t]) / torch.sqrt(1 - Alpha_bar[t])).to(x_t.device)
        mu_theta = (x_t - c * epsilon_theta) / torch.sqrt(Alpha[t]).to(x_t.device)
        return epsilon_theta, mu_theta, Sigma[t]

    def compute_eps_t(self, x_t, t):
        x_t = self._to_tensor(x_t)
        if _ag.needs_grad(x_t):
            return self.model((x_t, t * torch.ones((x_t.shape[0], 1))))
        with torch.no_grad():
            return self.model((x_t, t * torch.ones((x_t.shape[0], 1))))

    def one_shot_denoise(self, x_t):
        x_t = self._to_tensor(x_t)
        t = self.reverse_timestep - 1
        Alpha_bar = self.diffusion_hyperparams["Alpha_bar"]
        c_a = float((1 / Alpha_bar).sqrt()[t])
        c_b = float((1 / Alpha_bar - 1).sqrt()[t])
        if _ag.needs_grad(x_t):               # ref l.174-182,195-205 through the differentiable eps-network
            return c_a * x_t - c_b * self.model((x_t, t * torch.ones((x_t.shape[0], 1))))
        with torch.no_grad():
            return self.engine.one_shot(x_t, t, c_a, c_b).unsqueeze(1)

    @torch.no_grad()
    def two_shot_denoise(self, x_t):
        x_t = self._to_tensor(x_t)
        t = self.reverse_timestep - 1
        hp = self.diffusion_hyperparams
        Alpha, Alpha_bar, Beta = hp["Alpha"], hp["Alpha_bar"], hp["Beta"]
        eps = self.model((x_t, t * torch.ones((x_t.shape[0], 1))))
        x_1 = self._predict_x1_from_eps(x_t, t, eps)
        return self._predict_x0_from_x1(x_1)

    def _predict_x0_from_eps(self, x_t, t, eps):
        """ref l.195-205"""
        assert x_t.shape == eps.shape
        Alpha_bar = self.diffusion_hyperparams["Alpha_bar"]
        sqrt_recip = (1 / Alpha_bar).sqrt()
        sqrt_recipm1 = (1 / Alpha_bar - 1).sqrt()
        return self._extract_into_tensor(sqrt_recip, t, x_t.shape) * x_t - self._extract_into_tensor(sqrt_recipm1, t, x_t.shape) * eps

    def _predict_x1_from_eps(self, x_t, t, eps):
        """ref l.207-218"""
        hp = self.diffusion_hyperparams
        Alpha, Alpha_bar, Beta = hp["Alpha"], hp["Alpha_bar"], hp["Beta"]
        mu = (Alpha_bar[t] / Alpha[0]).sqrt().to(x_t.device)
        sigma = (1 - Alpha_bar[t] - (Alpha_bar[t] / Alpha[0]) * Beta[0] ** 2).sqrt().to(x_t.device)
        return (x_t - sigma * eps) / mu

    def _predict_x0_from_x1(self, x_1):
        """ref l.220-226"""
        return self.compute_coefficients(x_1, 0)[1]

    @staticmethod
    def _extract_into_tensor(arr_or_func, timesteps, broadcast_shape, device=None):
        """ref l.228-249: a table (tensor / ndarray) or a callable, indexed by `timesteps`, broadcast to the shape."""
        device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        if callable(arr_or_func):
            res = arr_or_func(timesteps).float()
        elif isinstance(arr_or_func, torch.Tensor):
            res = arr_or_func.to(device)[timesteps].float()
        elif isinstance(arr_or_func, np.ndarray):
            res = torch.from_numpy(arr_or_func).to(device)[timesteps].float()
        else:
            raise TypeError('Unsupported data type {} in arr_or_func'.format(type(arr_or_func)))
        while len(res.shape) < len(broadcast_shape):
            res = res[..., None]
        return res.expand(broadcast_shape)

    @torch.no_grad()
    def fast_reverse(self, x_t):
        """K = 3 strided sampler (ref l.106-141); note the reference uses Beta_tilde (not its sqrt) as sigma."""
        x_t = self._to_tensor(x_t)
        Alpha_bar = self.diffusion_hyperparams["Alpha_bar"]
        K = 3
        S = torch.round(torch.linspace(1, self.reverse_timestep, K)).int() - 1
        beta_new, beta_tilde_new = torch.zeros(K), torch.zeros(K)
        for i in range(K):
            if i > 0:
                beta_new[i] = 1 - Alpha_bar[S[i]] / Alpha_bar[S[i - 1]]
                beta_tilde_new[i] = (1 - Alpha_bar[S[i - 1]]) / (1 - Alpha_bar[S[i]]) * beta_new[i]
            else:
                beta_new[i] = 1 - Alpha_bar[S[i]]
        alpha_new = 1 - beta_new
        alpha_bar_new = torch.cumprod(alpha_new, dim=0)
        x = x_t
        for t in range(K - 1, -1, -1):
            eps = self.model((x, int(S[t]) * torch.ones((x.shape[0], 1))))
            c = ((1 - alpha_new[t]) / torch.sqrt(1 - alpha_bar_new[t])).to(x.device)
            mu = (x - c * eps) / torch.sqrt(alpha_new[t]).to(x.device)
            z = self._noise(x.shape, x.device)
            if z is None:
                z = self.engine.philox_normal(self.seed, self._draws, 0xFA57 + t, x.shape[0]).unsqueeze(1)
            x = mu + beta_tilde_new[t].to(x.device) * z
        self._draws += x.shape[0]
        return x


class ReffWave(torch.nn.Module):
    """ref l.251-349: `num_re` rounds of (diffuse to t*, one-shot denoise); same engine calls as DiffWave."""

    def __init__(self, model, diffusion_hyperparams: dict, reverse_timestep: int = 200, num_re: int = 5,
                 noise_source: str = 'device', seed: int = 0):
        super().__init__()
        self.model = model
        self.diffusion_hyperparams = diffusion_hyperparams
        self.reverse_timestep = reverse_timestep
        self.freeze = False
        self.num_re = num_re
        assert noise_source in ('device', 'torch_cpu')
        self.noise_source = noise_source
        self.seed = seed
        self._draws = 0

    @torch.no_grad()
    def forward(self, waveforms: Union[torch.Tensor, np.ndarray]):
        output = DiffWave._to_tensor(waveforms)
        for _ in range(self.num_re):
            output = self.diffusion(output)
            output = self.one_shot_denoise(output)
        return output

    @torch.no_grad()
    def diffusion(self, x_0) -> torch.Tensor:
        x_0 = DiffWave._to_tensor(x_0)
        hp = self.diffusion_hyperparams
        T, Alpha, Alpha_bar, Sigma = hp["T"], hp["Alpha"], hp["Alpha_bar"], hp["Sigma"]
        assert len(Alpha) == T
        assert len(Alpha_bar) == T
        assert len(Sigma) == T
        assert x_0.ndim == 3
        t = self.reverse_timestep - 1
        z = torch.normal(0, 1, size=tuple(x_0.shape)).to(x_0.device) if self.noise_source == 'torch_cpu' else None
        out = self.model.engine.diffuse(x_0, float(torch.sqrt(Alpha_bar[t])), float(torch.sqrt(1 - Alpha_bar[t])), z,
                                        seed=self.seed, sample0=self._draws)
        self._draws += x_0.shape[0]
        return out.unsqueeze(1)

    @torch.no_grad()
    def one_shot_denoise(self, x_t):
        x_t = DiffWave._to_tensor(x_t)
        t = self.reverse_timestep - 1
        Alpha_bar = self.diffusion_hyperparams["Alpha_bar"]
        return self.model.engine.one_shot(x_t, t, float((1 / Alpha_bar).sqrt()[t]), float((1 / Alpha_bar - 1).sqrt()[t])).unsqueeze(1)

    _predict_x0_from_eps = DiffWave._predict_x0_from_eps
    _extract_into_tensor = DiffWave._extract_into_tensor


def create_diffwave_model(model_path, config_path, reverse_timestep=25, state_dict=None, noise_source='device',
                          precision=None, max_batch=None, engine=None):
    """Reference signature (ref l.395-411) plus optional keyword-only extras.  Reads the JSON keys
    `wavenet_config` and `diffusion_config`, loads checkpoint['model_state_dict'] (weight_g/weight_v
    layout), folds and uploads the weights.  `state_dict` may be passed instead of
    a checkpoint path (synthetic weights)."""
    with open(config_path) as f:
        cfg = json.loads(f.read())
    wavenet_config = cfg["wavenet_config"]
    diffusion_hyperparams = calc_diffusion_hyperparams(**cfg["diffusion_config"])
    if state_dict is None:
        checkpoint = torch.load(model_path, map_location='cpu')
        state_dict = checkpoint['model_state_dict']
    eng = engine or _eng.get_engine(wavenet_config, precision=precision, max_batch=max_batch)
    if engine is None and eng.has_wavenet and eng.wavenet_owner != _eng.state_fingerprint(state_dict):
        eng = _eng.get_engine(wavenet_config, precision=precision, max_batch=max_batch, fresh=True)   # a second, different DiffWave
    eng.bind('wavenet', state_dict, eng.load_wavenet)
    return DiffWave(model=WaveNetHIP(eng, state_dict=state_dict), diffusion_hyperparams=diffusion_hyperparams,
                    reverse_timestep=reverse_timestep, noise_source=noise_source)
