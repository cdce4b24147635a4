"""create_model_and_diffusion / model_and_diffusion_defaults of the reference (improved_diffusion/script_util.py:11-131,
160-213) for the keys its wrapper sets; unsupported combinations raise instead of silently building something else."""
from . import gaussian_diffusion as gd
from .unet import UNetModel


def model_and_diffusion_defaults():
    return dict(image_size=32, num_channels=128, num_res_blocks=3, num_heads=4, num_heads_upsample=-1, attention_resolutions="16,8",
                dropout=0.3, learn_sigma=False, sigma_small=False, class_cond=False, diffusion_steps=200, noise_schedule="linear",
                timestep_respacing="", use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=True,
                use_checkpoint=False, use_scale_shift_norm=True)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, sigma_small, num_channels, num_res_blocks, num_heads,
                               num_heads_upsample, attention_resolutions, dropout, diffusion_steps, noise_schedule,
                               timestep_respacing, use_kl, predict_xstart, rescale_timesteps, rescale_learned_sigmas,
                               use_checkpoint, use_scale_shift_norm):
    if image_size != 32 or class_cond or learn_sigma or sigma_small or timestep_respacing or use_kl or predict_xstart or rescale_timesteps:
        raise NotImplementedError('only the configuration of improved_diffusion_ddpm.create_improved_diffusion is built for MI355X')
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    model = UNetModel(in_channels=1, model_channels=num_channels, out_channels=1, num_res_blocks=num_res_blocks,
                      attention_resolutions=attention_ds, dropout=dropout, channel_mult=(1, 2, 2, 2), num_classes=None,
                      use_checkpoint=use_checkpoint, num_heads=num_heads, num_heads_upsample=num_heads_upsample,
                      use_scale_shift_norm=use_scale_shift_norm)
    diffusion = gd.GaussianDiffusion(betas=gd.get_named_beta_schedule(noise_schedule, diffusion_steps), rescale_timesteps=rescale_timesteps)
    return model, diffusion
