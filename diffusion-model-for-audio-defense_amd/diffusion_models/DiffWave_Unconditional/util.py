"""Diffusion schedule helper kept under the reference's module path
(diffusion_models/DiffWave_Unconditional/util.py:96-123) because create_diffwave_model() exposes its
result as DiffWave.diffusion_hyperparams and RobustCertificate.compute_t_star() reads it."""
import torch


def calc_diffusion_hyperparams(T, beta_0, beta_T):
    """fp32 CPU tables {T, Beta, Alpha, Alpha_bar, Sigma}.  Alpha_bar and Beta_tilde are running
    products evaluated one step at a time in fp32 — the rounding order decides t*(sigma), so this is
    not replaced by cumprod."""
    beta = torch.linspace(beta_0, beta_T, T)
    alpha = 1 - beta
    alpha_bar = alpha.clone()
    beta_tilde = beta.clone()
    for t in range(1, T):
        alpha_bar[t] = alpha_bar[t] * alpha_bar[t - 1]
        beta_tilde[t] = beta_tilde[t] * ((1 - alpha_bar[t - 1]) / (1 - alpha_bar[t]))
    return {"T": T, "Beta": beta, "Alpha": alpha, "Alpha_bar": alpha_bar, "Sigma": torch.sqrt(beta_tilde)}
