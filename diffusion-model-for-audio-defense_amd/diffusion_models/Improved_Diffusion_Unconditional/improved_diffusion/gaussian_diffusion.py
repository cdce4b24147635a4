"""GaussianDiffusion of the reference (improved_diffusion/gaussian_diffusion.py:100-470) for the configuration its wrapper
builds: epsilon prediction, fixed-large variance, clip_denoised.  Tables in float64 exactly as the reference computes
them; q_sample / p_sample / p_sample_loop run on the HIP engine when the model is the HIP UNet (one fused elementwise
kernel per step after the eps-network), with the reference's CPU-free `randn_like` noise replaced by on-device Philox
noise unless `noise` tensors are passed."""
import numpy as np
import torch


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name != 'linear':
        raise NotImplementedError(f'unknown beta schedule: {schedule_name}')
    return np.linspace(0.0001, 0.02, num_diffusion_timesteps, dtype=np.float64)      # the reference's unscaled ends (l.30-35)


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    res = torch.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)


class GaussianDiffusion:

    def __init__(self, *, betas, rescale_timesteps=False):
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        self.rescale_timesteps = rescale_timesteps
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.model_log_variance = np.log(np.append(self.posterior_variance[1], betas[1:]))     # FIXED_LARGE

    @staticmethod
    def _step(t):
        steps = torch.as_tensor(t).reshape(-1)
        t0 = int(steps[0])
        assert bool((steps == t0).all()), 'the HIP path takes one timestep per batch'
        return t0

    def _f32(self, arr, t):
        return float(torch.from_numpy(arr)[t].float())           # what _extract_into_tensor yields: the fp32 table entry

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        t = torch.as_tensor(t, device=x_start.device).long().reshape(-1)
        if t.numel() == 1:
            t = t.expand(x_start.shape[0])
        return (_extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, noise=None, seed=0, sample0=0):
        """One reverse step (l.331-387) on the engine: returns {'sample', 'pred_xstart'}.  `noise`: the step's N(0, 1)
        draw ([B,1,32,32]); None = device Philox noise keyed (seed, sample0 + row, t)."""
        assert clip_denoised and denoised_fn is None and not model_kwargs, 'the HIP step implements the wrapper configuration'
        t0 = self._step(t)
        eng = model.engine if hasattr(model, 'engine') else model.bind_engine().engine
        xs = x.detach()[:, 0].contiguous().float().clone()
        sig = 0.0 if t0 == 0 else float(torch.exp(torch.tensor(0.5 * self._f32(self.model_log_variance, t0))))
        x0 = eng.unet_p_sample(xs, t0, self._f32(self.sqrt_recip_alphas_cumprod, t0), self._f32(self.sqrt_recipm1_alphas_cumprod, t0),
                               self._f32(self.posterior_mean_coef1, t0), self._f32(self.posterior_mean_coef2, t0), sig,
                               z=noise, seed=seed, sample0=sample0, want_x0=True)
        return {'sample': xs.unsqueeze(1), 'pred_xstart': x0.unsqueeze(1)}

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None, device=None,
                      progress=False, start_timestep=None, seed=0):
        """Reverse chain from `noise` (x_T, or x_{t*} when `start_timestep` = t* is given) down to x_0 (l.389-470)."""
        img = noise if noise is not None else torch.randn(*shape, device=device or 'cuda')
        for i in range((start_timestep or self.num_timesteps) - 1, -1, -1):
            img = self.p_sample(model, img, torch.full((shape[0],), i, dtype=torch.long), clip_denoised=clip_denoised, seed=seed)['sample']
        return img
