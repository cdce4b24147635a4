#!/usr/bin/env python3
"""bench.py — certified-smoothing throughput of the HIP path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (BASELINE.json `metric`): purified+classified 1 s clips/s of the Monte Carlo loop of
RobustCertificate.smooth_predict at sigma = 0.5 (t* = 66): Philox noise -> sqrt(alpha_bar*) scale ->
DiffWave one-shot purification (36-layer WaveNet, bf16 MFMA) -> mel dB -> VGG19_bn -> arg-max votes,
N = 100 000 samples per certified clip.  One STEP = `--samples-per-step` (default 512) Monte Carlo
samples per GPU through dmad_smooth_votes + the vote all-reduce (RCCL int64[10]) — weak scaling: every
rank works on its own shard of the sample index range; steps = 196 at 512 samples/step is one full
N = 100 000 certification on one GPU.  Inputs (clip, weights) are resident in HBM before the timed region.

The JSON line also carries
  roofline     — the dominant kernel (wn_layer_bf16): algorithmic FLOPs per launch / average launch
                 duration measured live with HIP event pairs on the launch stream over the timed steps,
                 against the dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s);
  cpu_baseline — the CPU oracle (a restatement of the reference's arithmetic, kind "port") timed on the
                 host cores of the same box on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd'), ROOT]

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0
# HBM-side bytes per clip per wn_layer_bf16_p launch from the rocprofv3 PMC passes committed under profiles/
# (FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, see profiles/r01h_kernel_stats.md); None if the file is absent
try:
    with open(os.path.join(ROOT, 'profiles', 'r01h_layer_traffic.json')) as _f:
        LAYER_TRAFFIC_PER_CLIP = float(json.load(_f)['layer_traffic_bytes_per_clip'])
except Exception:
    LAYER_TRAFFIC_PER_CLIP = None
L = 16000
LAYER_FLOP_PER_CLIP = 2.0 * L * (512 * 768 + 256 * 256)       # dilated conv + res conv of one layer


def cpu_baseline(n_samples: int):
    """oracle/dmad_oracle.py timed on the host cores: the checker's leg, never the measured product."""
    from dmad_hip import synth
    from oracle import dmad_oracle as orc
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = max(1, min(threads, 16))        # the GPU box gives one GPU's share of host cores (16)
    torch.set_num_threads(threads)
    sd, vsd = synth.wavenet_state_dict(1234), synth.vgg19_bn_state_dict(4321)
    w = orc.folded_weights(sd)
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    den = orc.DiffWaveOracle(w, hp)
    co = orc.CertifyOracle(lambda s: orc.vgg19_bn_forward(vsd, s), orc.mel_db, den)
    clip = torch.from_numpy(synth.synthetic_clip(0))
    torch.manual_seed(0)
    t0 = time.time()
    co.smooth_predict(clip, num_sampling=1, sigma=0.5, batch_size=1)          # warm-up, also sizes the sample
    one = time.time() - t0
    n_samples = max(1, min(n_samples, int(25.0 / max(one, 1e-3))))            # keep the CPU leg to ~10-30 s
    t0 = time.time()
    co.smooth_predict(clip, num_sampling=n_samples, sigma=0.5, batch_size=min(4, n_samples))
    dt = time.time() - t0
    return {"value": n_samples / dt, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "%d Monte Carlo samples (batch %d) of the same workload through oracle/dmad_oracle.py "
                      "(torch CPU fp32), %.1f s" % (n_samples, min(4, n_samples), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--samples-per-step', type=int, default=512)
    ap.add_argument('--max-batch', type=int, default=512,
                    help='clips per engine launch; 512 = one launch chain per step (gate store 151 GB of the 288 GB HBM)')
    ap.add_argument('--sigma', type=float, default=0.5)
    ap.add_argument('--cpu-samples', type=int, default=6)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--classifier', choices=['vgg19_bn', 'resnext29'], default='vgg19_bn',
                    help='vgg19_bn = the configuration BASELINE.json names; resnext29 = the reference script\'s default classifier')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # DMAD_BENCH_BACKEND=gloo is a rehearsal switch only (several ranks sharing one GPU of a 1-GPU box); the measured
    # configuration is one rank per GPU over RCCL
    backend = os.environ.get('DMAD_BENCH_BACKEND', 'nccl')
    if backend == 'gloo':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'gloo':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from dmad_hip import engine as E, synth
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    eng = None
    for mb in sorted({args.max_batch, min(args.max_batch, 256), min(args.max_batch, 128)}, reverse=True):
        try:                                   # 512 clips per launch chain needs ~165 GB of HBM: step down if it is not free
            eng = E.Engine(max_batch=mb, precision=E.BF16)
            args.max_batch = mb
            break
        except E.DmadError as exc:
            print('bench: engine batch %d not available (%s)' % (mb, exc), file=sys.stderr, flush=True)
    if eng is None:
        raise SystemExit('bench.py: could not create the engine')
    eng.load_wavenet(synth.wavenet_state_dict(1234))
    if args.classifier == 'resnext29':
        eng.load_resnext29(synth.resnext29_state_dict(2929))
    else:
        eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    sigma = args.sigma
    abar_star = 1 / (1 + sigma ** 2)
    t = int(torch.abs(ab - abar_star).min(0, keepdim=True)[1].item())      # t* - 1
    c_a, c_b = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])
    sc = float(torch.tensor(abar_star ** 0.5, dtype=torch.float32))
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    S = args.samples_per_step
    total = torch.zeros(10, dtype=torch.int64, device='cuda')

    def step(i):
        # global sample index range of this step: [i*S*world, (i+1)*S*world), rank r takes its slice
        base = (i * world + rank) * S
        counts, _, _ = eng.smooth_votes(clip, sigma, sc, t, c_a, c_b, S, seed=2024, sample0=base)
        if dist is not None:
            if backend == 'gloo':
                c = counts.cpu()
                dist.all_reduce(c)
                counts = c.cuda()
            else:
                dist.all_reduce(counts)
        total.add_(counts)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    chunks_per_step = (S + args.max_batch - 1) // args.max_batch
    fence()
    eng.profile_layers(args.steps * chunks_per_step * 35)
    total.zero_()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    layer_ms, launches = eng.profile_read()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if backend == 'gloo' else 'cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    votes = total.cpu().tolist()
    assert sum(votes) == args.steps * S * world, 'vote conservation violated: %s' % votes

    if rank == 0:
        clips = args.steps * S * world
        # launches of a step's last (possibly smaller) chunk carry fewer clips: weight by clips
        clips_per_launch = (args.steps * S) / (launches / 35.0) if launches else 0.0
        avg_ms = layer_ms / launches if launches else float('nan')
        achieved = LAYER_FLOP_PER_CLIP * clips_per_launch / (avg_ms * 1e-3) / 1e12 if launches else float('nan')
        out = {
            "metric": "purified+classified 1s clips/sec at N=100k sigma=0.5; 1/2/4/8 GPUs",
            "value": clips / dt, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "certified smoothing N=100000 sigma=%.2f (t*=%d): DiffWave one-shot purify (36x256 WaveNet) "
                                   "+ mel-dB + %s + votes; step = %d Monte Carlo samples per GPU, %d steps = one "
                                   "N=100000 clip" % (sigma, t + 1, 'VGG19_bn' if args.classifier == 'vgg19_bn' else 'ResNeXt29', S, -(-100000 // S)),
                       "samples_per_step_per_gpu": S, "engine_batch": args.max_batch, "sigma": sigma, "t_star": t + 1,
                       "noise": "device Philox4x32-10", "classifier": "VGG19_bn (synthetic seed 4321)" if args.classifier == 'vgg19_bn' else "ResNeXt29 8x64d (synthetic seed 2929)",
                       "parallelism": "mc-samples sharded x%d, one int64[10] all-reduce per step" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": (LAYER_TRAFFIC_PER_CLIP * clips_per_launch if LAYER_TRAFFIC_PER_CLIP else None),
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/r01h_kernel_stats.md "
                                           "(2 x FETCH_SIZE + WRITE_SIZE, bytes per launch)",
                         "kernel": "wn_layer_bf16", "avg_launch_ms": avg_ms, "launches_timed": launches,
                         "flop_per_launch": LAYER_FLOP_PER_CLIP * clips_per_launch},
            "end_to_end_tflops": clips / dt / world * 606.94e9 / 1e12,
            "votes": votes,
        }
        if world == 1 and not args.no_cpu_baseline and args.classifier == 'vgg19_bn':
            out["cpu_baseline"] = cpu_baseline(args.cpu_samples)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
