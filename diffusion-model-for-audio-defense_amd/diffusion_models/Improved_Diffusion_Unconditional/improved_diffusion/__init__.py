"""Inference-side mirror of the reference's improved_diffusion package (Improved-Diffusion UNet purifier on mel
spectrograms): `unet.UNetModel`, `gaussian_diffusion.GaussianDiffusion`, `script_util.
create_model_and_diffusion`, `sc09_spectrogram_dataset.melspec_standardize / melspec_inv_standardize`."""
