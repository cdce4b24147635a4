import sys, os, numpy as np, torch
ROOT='/root/repo'
sys.path[:0]=[ROOT, ROOT+'/diffusion-model-for-audio-defense_amd']
from dmad_hip import engine as E, synth
from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
ab = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)['Alpha_bar']
eng = E.Engine(max_batch=64, precision=E.EXACT, recheck_batch=32)
eng.load_wavenet(synth.wavenet_state_dict(1234)); eng.load_resnext29(synth.resnext29_state_dict(2929))
clip = torch.from_numpy(synth.synthetic_clip(0)).cuda(); sigma=0.5
t = int(torch.abs(ab - 1 / (1 + sigma ** 2)).min(0, keepdim=True)[1].item())
c_a, c_b = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t]); sc=float(torch.tensor((1/(1+sigma**2))**0.5))
print('modes', E.MODE_FAST, E.MODE_EXACT_VOTES, E.MODE_FP32)
out={}
for mode in (E.MODE_FAST, E.MODE_FP32, E.MODE_EXACT_VOTES):
    eng.set_mode(mode)
    c,l,_ = eng.smooth_votes(clip, sigma, sc, t, c_a, c_b, 512, seed=700, sample0=9000, want_logits=True)
    out[mode]=l.cpu().numpy().astype(np.float64); print(mode, c.tolist(), out[mode][0,:4], 'nonfinite rows', int((~np.isfinite(out[mode])).any(1).sum()))
print('fast-fp32 max', np.abs(out[E.MODE_FAST]-out[E.MODE_FP32]).max(), 'exact-fp32', np.abs(out[E.MODE_EXACT_VOTES]-out[E.MODE_FP32]).max())

bad = np.where((~np.isfinite(out[E.MODE_FAST])).any(1))[0]
print('bad rows', bad[:10])
for tier in (0,1,2):
    pass
