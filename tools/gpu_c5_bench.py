#!/usr/bin/env python3
"""BASELINE configuration C5 on one GPU: certified smoothing with the Improved-Diffusion UNet purifier on mel spectrograms
(dmad_spec_smooth_votes), N = 10 000 Monte Carlo samples of one clip.  Prints / writes one JSON record.

    N=10000 TSTAR=25 B=512 MODE=exact python tools/gpu_c5_bench.py
MODE: exact (default: UNet chain on the 16-bit tier + fp32 re-run of the low-margin samples), fast (16-bit tier alone), fp32 (the
exact-fp32 UNet: engine batch 128 -> 208 samples/s, 512 -> 227, 2048 -> 234).
"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import _lib
if os.environ.get('DMAD_LIB'):                   # A/B of two builds on one box
    _lib.LIB_PATH = os.environ['DMAD_LIB']
from dmad_hip import engine as E, synth
from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion
N, TSTAR, B = int(os.environ.get('N', 10000)), int(os.environ.get('TSTAR', 25)), int(os.environ.get('B', 512))
MODE = os.environ.get('MODE', 'exact')
eng = E.Engine(max_batch=B, precision=E.EXACT, recheck_batch=0, with_wavenet=False)
eng.set_mode({'exact': E.MODE_EXACT_VOTES, 'fast': E.MODE_FAST, 'fp32': E.MODE_FP32}[MODE])
eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321, calibrated='c5'))
pur = create_improved_diffusion(None, reverse_timestep=TSTAR, state_dict=synth.unet_state_dict(31), engine=eng)
ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig = pur.purify_coefficients()
clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
args = (clip, 0.5, ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, -100.0, 38.22)
eng.spec_smooth_votes(*args, 2 * B, seed=1)
eng.spec_recheck_stats(reset=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
counts, _, _ = eng.spec_smooth_votes(*args, N, seed=2024)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nfe = TSTAR + 1
rec = {"config": "C5: certified smoothing N=%d sigma=0.5, spec-domain purifier (Improved-Diffusion UNet, t*=%d: %d network evaluations per sample) + VGG19_bn, 1x MI355X" % (N, TSTAR, nfe),
       "mode": MODE, "samples_per_s": N / dt, "seconds": dt, "engine_batch": B, "votes": counts.cpu().tolist(),
       "rechecked_on_fp32": eng.spec_recheck_stats()[1], "tau_spec": eng.spec_recheck_margin,
       "unet_tflops": N * nfe * 16.76e9 / dt / 1e12, "frac_of_fp32_matrix_peak": N * nfe * 16.76e9 / dt / 1e12 / 157.3}
print(json.dumps(rec))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, 'gpurun_out', 'c5_bench_%s.json' % MODE), 'w'))
