#!/usr/bin/env python3
"""Board power / sclk (hwmon) while UNet evaluations of B spectrograms run back to back on tier TIER (0 fp32, 1 16-bit, 2 split-f16):
is the conv-GEMM family at the board's power cap like the WaveNet kernels?      TIER=1 B=2048 SECONDS=4 python tools/gpu_unet_power.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd'), os.path.join(ROOT, 'tools')]
from dmad_hip import engine as E, synth
import power_sampler as ps
TIER, B, SECONDS = int(os.environ.get('TIER', 1)), int(os.environ.get('B', 2048)), float(os.environ.get('SECONDS', 4))
eng = E.Engine(max_batch=B, precision=E.EXACT if TIER == 2 else (E.FP32 if TIER == 0 else E.BF16), with_classifier=False, with_wavenet=False)
eng.load_unet(synth.unet_state_dict(5252))
x = torch.randn(B, 32, 32, device='cuda') * 0.5
eng.unet_eps(x, 40, tier=TIER); torch.cuda.synchronize()
t0 = time.time(); eng.unet_eps(x, 39, tier=TIER); torch.cuda.synchronize(); per = time.time() - t0
reps = max(4, int(SECONDS / per))


def load():
    t0 = time.time()
    for i in range(reps):
        eng.unet_eps(x, 40 - (i & 1), tier=TIER)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps


src = ps.sysfs_sources()
_, idle_w, idle_clk = ps.measure(lambda: time.sleep(1.5), src, settle=0.3)
ms, watts, sclk = ps.measure(load, src)
print(json.dumps({'tier': TIER, 'B': B, 'ms_per_evaluation': ms * 1e3, 'tflops_nominal': B * 16.76e9 / ms / 1e12, 'board_w_median': watts, 'sclk_mhz_median': sclk,
                  'idle_w': idle_w, 'power_cap_w': ps.read_num(src['cap'], 1e-6) if src.get('cap') else None}), flush=True)
eng.close()
