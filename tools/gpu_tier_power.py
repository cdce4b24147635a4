#!/usr/bin/env python3
"""Board power / sclk (hwmon) while WaveNet evaluations of B clips run back to back on path PATH of an exact-vote engine (0 = 16-bit,
1 = exact fp32, 2 = split-f16): which of the recheck tiers are at the board's power cap?      PATH_ID=2 B=32 SECONDS=4 python tools/gpu_tier_power.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd'), os.path.join(ROOT, 'tools')]
from dmad_hip import engine as E, synth
import power_sampler as ps
PATH, B, SECONDS = int(os.environ.get('PATH_ID', os.environ.get('TIER', 2))), int(os.environ.get('B', 32)), float(os.environ.get('SECONDS', 4))
eng = E.Engine(max_batch=B, precision=E.EXACT, recheck_batch=B, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
eng.wavenet_eps_path(x, 65, PATH); torch.cuda.synchronize()
t0 = time.time(); eng.wavenet_eps_path(x, 65, PATH); torch.cuda.synchronize(); per = time.time() - t0
reps = max(4, int(SECONDS / per))


def load():
    t0 = time.time()
    for _ in range(reps):
        eng.wavenet_eps_path(x, 65, PATH)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps


src = ps.sysfs_sources()
_, idle_w, idle_clk = ps.measure(lambda: time.sleep(1.5), src, settle=0.3)
ms, watts, sclk = ps.measure(load, src)
print(json.dumps({'path': PATH, 'B': B, 'ms_per_evaluation': ms * 1e3, 'clips_per_s': B / ms, 'board_w_median': watts, 'sclk_mhz_median': sclk, 'idle_w': idle_w,
                  'power_cap_w': ps.read_num(src['cap'], 1e-6) if src.get('cap') else None}), flush=True)
eng.close()
