#!/usr/bin/env python3
"""Evidence: the exact-vote mode's per-class counts against the exact-fp32 path's on the same Philox keys, at an N the fp32 path
can still run (N = 16 384 per sigma by default).  Writes gpurun_out/exact_vs_fp32.json."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import engine as E, synth
from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
N = int(os.environ.get('N', 16384))
SIGMAS = [float(s) for s in os.environ.get('SIGMAS', '0.5,0.25').split(',')]
eng = E.Engine(max_batch=256, precision=E.EXACT, recheck_batch=64)
eng.load_wavenet(synth.wavenet_state_dict(1234))
eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
ab = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)['Alpha_bar']
CLIP = int(os.environ.get('CLIP', 0))
clip = torch.from_numpy(synth.synthetic_clip(CLIP)).cuda()
out = []
for sigma in SIGMAS:
    t = int(torch.abs(ab - 1 / (1 + sigma ** 2)).min(0, keepdim=True)[1].item())
    args = (clip, sigma, float(torch.tensor((1 / (1 + sigma ** 2)) ** 0.5, dtype=torch.float32)), t, float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t]), N)
    rec = {'clip': CLIP, 'sigma': sigma, 't_star': t + 1, 'n': N}
    for name, mode in (('exact', E.MODE_EXACT_VOTES), ('fp32', E.MODE_FP32), ('fast', E.MODE_FAST)):
        eng.set_mode(mode); eng.recheck_stats(reset=True)
        torch.cuda.synchronize(); t0 = time.time()
        c, done = None, 0
        while done < N:                     # chunks: a progress line every ~45 s of fp32 work keeps the run visibly alive
            k = min(8192, N - done)
            c, _, _ = eng.smooth_votes(*args[:-1], k, seed=4242, sample0=done, counts=c)
            done += k
            torch.cuda.synchronize()
            print('  %s: %d / %d samples, %.0f s' % (name, done, N, time.time() - t0), flush=True)
        rec[name] = {'counts': c.cpu().tolist(), 'seconds': time.time() - t0, 'stats': eng.recheck_stats(detail=True)}
    rec['exact_equals_fp32'] = rec['exact']['counts'] == rec['fp32']['counts']
    rec['fast_differs_by'] = sum(abs(a - b) for a, b in zip(rec['fast']['counts'], rec['fp32']['counts'])) // 2
    print(json.dumps(rec), flush=True)
    out.append(rec)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'exact_vs_fp32_clip%d.json' % CLIP), 'w'), indent=1)
assert all(r['exact_equals_fp32'] for r in out)
