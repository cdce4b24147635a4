#!/usr/bin/env python3
"""Development / evidence: the tail kernel wn_final_p alone — ms per launch (HIP events of dmad_profile_layers around every launch),
GB/s of the gate store it streams, PFLOP/s of its skip GEMM, and the board power / sclk while whole evaluations run back to back
(the layer launches in between are the real mix the kernel lives in).  DMAD_LIB selects a build (tools/final_variants.sh).
    B=512 SECONDS=4 python tools/gpu_final_time.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd'), os.path.join(ROOT, 'tools')]
from dmad_hip import _lib  # noqa: E402
if os.environ.get('DMAD_LIB'):
    _lib.LIB_PATH = os.environ['DMAD_LIB']
from dmad_hip import engine as E, synth  # noqa: E402
import power_sampler as ps  # noqa: E402

B = int(os.environ.get('B', 512))
SECONDS = float(os.environ.get('SECONDS', 4))
eng = E.Engine(max_batch=B, precision=E.BF16, half_type=E.HALF_F16, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
eng.wavenet_eps(x, 65); torch.cuda.synchronize()
t0 = time.time(); eng.wavenet_eps(x, 65); torch.cuda.synchronize(); per_eval = time.time() - t0
reps = max(3, int(SECONDS / per_eval))
eng.profile_layers(reps * 36)


def load():
    for _ in range(reps):
        eng.wavenet_eps(x, 65)
    torch.cuda.synchronize()


_, watts, sclk = ps.measure(load, ps.sysfs_sources())
fms, fn = eng.profile_read_final()
lms, ln = eng.profile_read()
gate_bytes = 36 * 256 * 2.0 * B * 16000
flops = 2.0 * B * 16000 * 256 * (36 * 256 + 256)
rec = {'lib': os.path.basename(_lib.LIB_PATH), 'B': B, 'final_ms_per_launch': fms / max(fn, 1), 'final_launches': fn,
       'gate_store_GBps': gate_bytes / (fms / max(fn, 1) * 1e-3) / 1e9, 'skip_gemm_PFLOPs': flops / (fms / max(fn, 1) * 1e-3) / 1e15,
       'layer_ms_per_launch': lms / max(ln, 1), 'board_w_median': watts, 'sclk_mhz_median': sclk}
print(json.dumps(rec), flush=True)
eng.close()
