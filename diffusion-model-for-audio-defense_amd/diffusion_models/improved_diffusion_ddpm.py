"""Host mirror of the reference's diffusion_models/improved_diffusion_ddpm.py: the Improved-Diffusion purifier on
standardised mel spectrograms.  The reference's `_reverse` discards the result of p_sample_loop and starts it from pure
noise (the wrapper is broken as committed); this mirror implements what the wrapper is for — diffuse x_0 to
t*, run the reverse chain from x_{t*} down to x_0, map back to the mel-dB scale — and says so here."""
from typing import Union

import numpy as np
import torch

from .Improved_Diffusion_Unconditional.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
from .Improved_Diffusion_Unconditional.improved_diffusion.sc09_spectrogram_dataset import melspec_standardize, melspec_inv_standardize  # noqa: F401


class ImprovedDiffusion(torch.nn.Module):

    def __init__(self, model=None, diffusion=None, reverse_timestep: int = 0, seed: int = 0):
        super().__init__()
        self.model = model
        self.diffusion = diffusion
        self.reverse_timestep = reverse_timestep
        self.seed = seed

    def forward(self, waveforms: Union[torch.Tensor, np.ndarray]):
        if isinstance(waveforms, np.ndarray):
            waveforms = torch.from_numpy(waveforms)
        output = self._diffusion(waveforms)
        output = self._reverse(output)
        return melspec_inv_standardize(output)

    def _diffusion(self, x_0):
        if isinstance(x_0, np.ndarray):
            x_0 = torch.from_numpy(x_0)
        t = torch.full((x_0.shape[0],), self.reverse_timestep, dtype=torch.long, device=x_0.device)
        return self.diffusion.q_sample(x_0, t=t)

    def purify_coefficients(self):
        """The argument block of dmad_spec_smooth_votes: (t*, q_a, q_b, c_a[], c_b[], c_1[], c_2[], c_sig[]) for q_sample at t* and
        p_sample at t = 0..t*, each the fp32 table entry GaussianDiffusion.q_sample / p_sample use."""
        gd, ts = self.diffusion, self.reverse_timestep
        f = gd._f32
        sig = [0.0 if t == 0 else float(torch.exp(torch.tensor(0.5 * f(gd.model_log_variance, t)))) for t in range(ts + 1)]
        return (ts, f(gd.sqrt_alphas_cumprod, ts), f(gd.sqrt_one_minus_alphas_cumprod, ts),
                [f(gd.sqrt_recip_alphas_cumprod, t) for t in range(ts + 1)], [f(gd.sqrt_recipm1_alphas_cumprod, t) for t in range(ts + 1)],
                [f(gd.posterior_mean_coef1, t) for t in range(ts + 1)], [f(gd.posterior_mean_coef2, t) for t in range(ts + 1)], sig)

    @torch.no_grad()
    def _reverse(self, x_t):
        if isinstance(x_t, np.ndarray):
            x_t = torch.from_numpy(x_t)
        return self.diffusion.p_sample_loop(model=self.model, shape=x_t.shape, noise=x_t, start_timestep=self.reverse_timestep + 1,
                                            seed=self.seed)


class SpecDefense(torch.nn.Module):
    """Waveform -> purified mel-dB spectrogram: the `transform` + spec `defender` of an AcousticSystem(defense_type='spec')
    (acoustic_system.py:40-49) as one callable, so that RobustCertificate(classifier, transform=SpecDefense(mel, purifier))
    is the certified-smoothing loop of BASELINE configuration C5.  With all three stages on one engine that loop is ONE
    C-ABI call (dmad_spec_smooth_votes); called directly it is mel -> standardise -> ImprovedDiffusion.forward."""

    def __init__(self, mel, purifier: ImprovedDiffusion):
        super().__init__()
        self.mel, self.purifier = mel, purifier

    @torch.no_grad()
    def forward(self, x):
        return self.purifier(melspec_standardize(self.mel(x)))


class SpecPurifier(torch.nn.Module):
    """The spec-domain `defender` of AcousticSystem(classifier, transform, defender, defense_type='spec') (acoustic_system.py:40-49):
    mel-dB spectrograms [B,1,32,32] in, purified mel-dB spectrograms out — standardise, q_sample(t*), the t* + 1 p_sample steps, map
    back.  Every draw of row b is Philox-keyed (seed, draws so far + b) with the streams of dmad_spec_smooth_votes (include/dmad.h), so
    `AcousticSystem.query` can run the same rows as ONE engine call (dmad_spec_query_logits) and a row's result does not depend on how
    the rows were batched."""
    noise_source = 'device'

    def __init__(self, purifier: ImprovedDiffusion, seed: int = 0):
        super().__init__()
        self.purifier, self.seed, self._draws = purifier, seed, 0

    @property
    def engine(self):
        return getattr(self.purifier.model, 'engine', None)

    @torch.no_grad()
    def forward(self, spec_db):
        pur, eng = self.purifier, self.engine
        if eng is None:
            raise RuntimeError('SpecPurifier needs a UNet bound to a dmad engine (create_improved_diffusion(..., engine=...))')
        B, ts, s0 = spec_db.shape[0], pur.reverse_timestep, self._draws
        x0 = melspec_standardize(spec_db.float())
        zq = eng.philox_normal(self.seed, s0, 0x5BEC, B)[:, :1024].reshape(B, 1, 32, 32)
        x = pur.diffusion.q_sample(x0, torch.full((B,), ts, dtype=torch.long, device=x0.device), noise=zq)
        for t in range(ts, -1, -1):
            x = pur.diffusion.p_sample(pur.model, x, torch.full((B,), t), seed=self.seed, sample0=s0)['sample']
        self._draws += B
        return melspec_inv_standardize(x)


def create_improved_diffusion(model_path, reverse_timestep=25, state_dict=None, engine=None):
    """reference l.64-93: image_size 32, 128 channels, 3 ResBlocks, fixed sigma, 1000 linear steps."""
    args = model_and_diffusion_defaults()
    args.update(image_size=32, num_channels=128, num_res_blocks=3, learn_sigma=False, diffusion_steps=1000, noise_schedule='linear')
    model, diffusion = create_model_and_diffusion(**args)
    if state_dict is None:
        state_dict = torch.load(model_path, map_location='cpu')
    model.load_state_dict({k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()})
    model.eval()
    if engine is not None:
        model.bind_engine(engine)
    return ImprovedDiffusion(model=model, diffusion=diffusion, reverse_timestep=reverse_timestep)
