#!/usr/bin/env python3
"""bench.py — certified-smoothing throughput of the HIP path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

One process per GPU over RCCL.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(WORLD_SIZE set) this process is one rank.  Started plainly with --gpus N > 1 (WORLD_SIZE unset) the parent — before any
HIP call — starts the N ranks itself as a CHILD `python -m torch.distributed.run` process, forwards rank 0's single
JSON line and exits with the child's code; fewer than N visible GPUs is an error, never a silent single-rank run.

Workload (BASELINE.json `metric`): purified+classified 1 s clips/s of the Monte Carlo loop of
RobustCertificate.smooth_predict at sigma = 0.5 (t* = 66): Philox noise -> sqrt(alpha_bar*) scale -> DiffWave one-shot
purification (36-layer WaveNet) -> mel dB -> VGG19_bn -> arg-max votes, N = 100 000 samples per certified clip.
One STEP = `--samples-per-step` (default 512) Monte Carlo samples per GPU through dmad_smooth_votes + the vote
all-reduce (RCCL int64[10]) — weak scaling: every rank works on its own shard of the sample index range; 196 steps of
512 samples are one full N = 100 000 certification on one GPU.  Clip and weights are resident in HBM before the timed region.

`value` is measured in the engine's EXACT-VOTE mode (the drop-in default): f16-operand MFMA WaveNet, and every sample
whose top-2 logit margin is below the recheck bound is re-evaluated on the split-f16 / exact-fp32 tiers from the same
Philox key.  The bound is a measured one, so the equality of the counts with the fp32 path's is an
empirical guarantee — and the line itself checks it: `exact_equals_fp32` re-runs the first timed steps' sample range in
the exact-vote mode and on the exact-fp32 path and prints both vote vectors.  The line also carries
  fast_mode / fp32_mode — the same step with the recheck off (16-bit path alone) and on the exact-fp32 path alone;
  sigma_grid     — the other sigmas of BASELINE C4's grid (0.25 and 1.0; README.md:12-14, scripts/certified_robust_eval.sh:3-6):
                   a few timed steps each in the exact-vote mode and on the 16-bit tier alone, their recheck fractions, and
                   exact == fp32 on the first timed step's keys of that sigma;
  resnext29_mode — the same step with the reference script's DEFAULT classifier (ResNeXt29 8x64d, certified_robustness_eval.py:57):
                   tier 1 runs the classifier's 16-bit tier (gemm_h16), the recheck tiers the fp32 one; clips/s, recheck fractions,
                   exact == fp32 on its first timed step's keys;
  c2_ddpm_mode   — BASELINE C2: DiffWave DDPM purification t* = 5 of a batch of 256 clips + mel-dB + VGG19_bn on a bf16
                   engine (dmad_query_logits, sampler 1): clips/s, network evaluations/s, layer-kernel roofline fraction;
  c3_certify_n1000 — BASELINE C3: RobustCertificate.certify(n_0=100, n=1000) through the host mirror, clips/s;
  roofline       — the dominant kernel (wn_layer_p): algorithmic FLOPs per launch / average launch duration measured
                   live with HIP event pairs on the launch stream over the timed steps, against the dense 16-bit MFMA peak
                   (MI355X_MICROARCH.md: ~2.5 PFLOP/s);
  roofline_final — the tail kernel (wn_final_p, HBM-bound: reads the 295 MB/clip gate store once) against 8 TB/s;
  cpu_baseline   — the CPU oracle (a restatement of the reference's arithmetic, kind "port") timed on the host cores of
                   the same box on a bounded sample (rank 0, N = 1 only);
  certify_full   — RobustCertificate.certify(x, n_0=100, n = 100000, sigma) through the host mirror (the surface the
                   reference's driver calls, scripts/certified_robust_eval.sh:3-6), timed end to end on every GPU count: the
                   N = 100 000 samples of ONE clip sharded over the ranks + one int64[10] all-reduce per pass — the strong-scaling
                   quantity BASELINE.json's metric names (--full-n to shorten it).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd'), ROOT]

PEAK_MFMA16_TFLOPS = 2500.0
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
L = 16000
LAYER_FLOP_PER_CLIP = 2.0 * L * (512 * 768 + 256 * 256)       # dilated conv + res conv of one layer
FINAL_BYTES_PER_CLIP = 36 * 256 * 2.0 * L + 4.0 * L            # gate store read once + eps written
CLIP_FLOP = 606.94e9
UNET_FLOP_PER_SPEC = 16.76e9                                    # one Improved-Diffusion UNet evaluation of a 32x32 spectrogram (convs, linears, attention)


def layer_traffic_per_clip():
    """HBM-side bytes per clip per wn_layer launch from the newest rocprofv3 PMC passes committed under profiles/
    (2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE); (None, None) if no such file exists — or if the newest one was taken on
    ANOTHER version of the WaveNet kernels (the file carries one hash over csrc/wn_layer.hip, wn_final.hip, wn_bf16.h and dmad_common.h): a
    stale figure is dropped, not shown."""
    import glob
    import hashlib
    try:
        h = hashlib.sha256()
        for name in ('wn_layer.hip', 'wn_final.hip', 'wn_bf16.h', 'dmad_common.h'):      # tools/make_profile_summary.py KERNEL_SOURCES
            with open(os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd', 'csrc', name), 'rb') as f:
                h.update(f.read())
        now = h.hexdigest()[:16]
    except Exception:
        now = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_layer_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec.get('layer_kernel_sha16') not in (None, now):
                return None, None
            return float(rec['layer_traffic_bytes_per_clip']), os.path.basename(path)
        except Exception:
            continue
    return None, None


def cpu_baseline(n_samples: int):
    """oracle/dmad_oracle.py timed on the host cores: the checker's leg, never the measured product."""
    import torch
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


def spawn_ranks(n: int) -> int:
    """Parent of a plain `python bench.py --gpus N`: start the N ranks as a child torch.distributed.run process."""
    import torch
    have = torch.cuda.device_count()          # counts devices without initialising HIP in this process
    if have < n:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) are visible; refusing to run fewer ranks than asked\n' % (n, have))
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(proc.stdout)
    return proc.returncode if proc.returncode else (0 if lines else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--samples-per-step', type=int, default=512)
    ap.add_argument('--max-batch', type=int, default=512,
                    help='clips per engine launch; 512 = one launch chain per step (gate store 151 GB of the 288 GB HBM)')
    ap.add_argument('--sigma', type=float, default=0.5)
    ap.add_argument('--mode', choices=['exact', 'fast', 'fp32'], default='exact',
                    help='what `value` measures: exact = 16-bit path + fp32 recheck of the close votes (the drop-in default), '
                         'fast = 16-bit path alone, fp32 = the exact-fp32 path alone')
    ap.add_argument('--half', choices=['f16', 'bf16'], default='f16', help='operand format of the 16-bit MFMA path')
    ap.add_argument('--recheck-margin', type=float, default=None, help='override the engine default recheck bound')
    ap.add_argument('--recheck-batch', type=int, default=64)
    ap.add_argument('--side-steps', type=int, default=None, help='steps of the two side measurements (fast / exact, fp32); 0 = skip')
    ap.add_argument('--cpu-samples', type=int, default=6)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--full', action='store_true', help='kept for older command lines: the default run already times certify(n_0=100, n=100000)')
    ap.add_argument('--full-n', type=int, default=100000, help='n of the certify() run through the host mirror (about 65 s / n_gpus at 100000)')
    ap.add_argument('--check-steps', type=int, default=2, help='steps of the timed sample range re-run in the exact-vote mode and on the '
                                                               'exact-fp32 path to print exact_equals_fp32 (0 = skip; fp32 costs ~2.8 s per step)')
    ap.add_argument('--c2-iters', type=int, default=4, help='timed iterations of the BASELINE C2 leg (DDPM t*=5, B=256, bf16); 0 = skip')
    ap.add_argument('--c2-batch', type=int, default=256, help='clips per batch of the C2 leg (BASELINE C2 names 256)')
    ap.add_argument('--c3-clips', type=int, default=4, help='clips certified with n=1000 for the BASELINE C3 leg; 0 = skip')
    ap.add_argument('--no-certify', action='store_true', help='skip the certify() run through the host mirror')
    ap.add_argument('--c5-n', type=int, default=10000, help='Monte Carlo samples PER RANK of the BASELINE C5 leg (spec-domain vote loop, '
                                                            'Improved-Diffusion UNet purifier, t* = 25; BASELINE C5 names N = 10 000: '
                                                            'about 13 s in the exact-vote mode); 0 = skip')
    ap.add_argument('--sigma-grid', type=str, default='0.25,1.0',
                    help='the other sigmas of BASELINE C4\'s grid (README.md:12-14): each gets --grid-steps timed steps in the exact-vote mode '
                         '(+ the 16-bit tier alone, + exact == fp32 on its own keys); empty = skip')
    ap.add_argument('--grid-steps', type=int, default=4)
    ap.add_argument('--resnext-steps', type=int, default=6, help='timed steps of the ResNeXt29 leg (the reference script\'s default classifier on an '
                                                                 'exact-vote engine of its own, ~30 s incl. the engine build); 0 = skip')
    ap.add_argument('--c5-batch', type=int, default=2048, help='engine batch of the C5 leg (spectrograms per UNet evaluation; ~45 GB of HBM at 2048)')
    ap.add_argument('--classifier', choices=['vgg19_bn', 'resnext29'], default='vgg19_bn',
                    help='vgg19_bn = the configuration BASELINE.json names; resnext29 = the reference script\'s default classifier')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))      # nothing in this process has touched the GPU

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)'
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # DMAD_BENCH_BACKEND=gloo is a rehearsal switch only (several ranks sharing one GPU of a 1-GPU box); the measured
    # configuration is one rank per GPU over RCCL
    backend = os.environ.get('DMAD_BENCH_BACKEND', 'nccl')
    if backend == 'gloo':
        local_rank %= torch.cuda.device_count()
    elif torch.cuda.device_count() < world:
        raise SystemExit('bench.py: %d ranks but %d visible GPU(s)' % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'gloo':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from dmad_hip import engine as E, synth
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    half = E.HALF_F16 if args.half == 'f16' else E.HALF_BF16
    eng = None
    for mb in sorted({args.max_batch, min(args.max_batch, 256), min(args.max_batch, 128)}, reverse=True):
        try:                                   # 512 clips per launch chain needs ~170 GB of HBM: step down if it is not free
            eng = E.Engine(max_batch=mb, precision=E.EXACT, half_type=half, recheck_batch=min(args.recheck_batch, mb),
                           recheck_margin=args.recheck_margin)
            args.max_batch = mb
            break
        except E.DmadError as exc:
            print('bench: engine batch %d not available (%s)' % (mb, exc), file=sys.stderr, flush=True)
    if eng is None:
        raise SystemExit('bench.py: could not create the engine')
    wsd = synth.wavenet_state_dict(1234)
    eng.load_wavenet(wsd)
    csd = synth.resnext29_state_dict(2929) if args.classifier == 'resnext29' else synth.vgg19_bn_state_dict(4321)
    (eng.load_resnext29 if args.classifier == 'resnext29' else eng.load_vgg19_bn)(csd)
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    sigma = args.sigma

    def sigma_cfg(sig):
        """(sigma, sqrt(alpha_bar*), t* - 1, c_a, c_b) of certified_robust.py:51-54,102-110 + diffwave_ddpm.py:199-203."""
        abar = 1 / (1 + sig ** 2)
        tt = int(torch.abs(ab - abar).min(0, keepdim=True)[1].item())
        return (sig, float(torch.tensor(abar ** 0.5, dtype=torch.float32)), tt, float((1 / ab).sqrt()[tt]), float((1 / ab - 1).sqrt()[tt]))
    cfg0 = sigma_cfg(sigma)
    _, sc, t, c_a, c_b = cfg0
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    S = args.samples_per_step
    total = torch.zeros(10, dtype=torch.int64, device='cuda')
    MODES = {'exact': E.MODE_EXACT_VOTES, 'fast': E.MODE_FAST, 'fp32': E.MODE_FP32}

    cur = {'eng': eng}                         # the engine the step / timed helpers drive (the ResNeXt29 leg swaps it)

    def step(i, cfg=cfg0):
        # global sample index range of this step: [i*S*world, (i+1)*S*world), rank r takes its slice
        base = (i * world + rank) * S
        counts, _, _ = cur['eng'].smooth_votes(clip, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], S, seed=2024, sample0=base)
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

    def timed(mode, steps, warmup, first_step, profile=False, cfg=cfg0):
        """`steps` timed steps in `mode`; -> (seconds (max over ranks), votes, recheck fraction, layer/final timings)."""
        eng = cur['eng']
        eng.set_mode(MODES[mode])
        for i in range(warmup):
            step(first_step + i, cfg)
        fence()
        chunks_per_step = (S + args.max_batch - 1) // args.max_batch
        if profile:
            eng.profile_layers(steps * chunks_per_step * 35)
        eng.recheck_stats(reset=True)
        total.zero_()
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(first_step + warmup + i, cfg)
        fence()
        dt = time.perf_counter() - t0
        prof = None
        if profile:
            fin = eng.profile_read_final()
            prof = (eng.profile_read(), fin)
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if backend == 'gloo' else 'cuda')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        votes = total.cpu().tolist()
        assert sum(votes) == steps * S * world, 'vote conservation violated: %s' % votes
        voted, rechecked, rechecked32 = eng.recheck_stats(detail=True)
        return dt, votes, ((rechecked / voted, rechecked32 / voted) if voted else (0.0, 0.0)), prof

    recheck_margin, recheck_margin2 = eng.recheck_margin, eng.recheck_margin2
    # ---- the measured region: exactly --steps steps after --warmup untimed ones, in --mode --------------------------
    dt, votes, recheck_frac, prof = timed(args.mode, args.steps, args.warmup, 0, profile=(args.mode != 'fp32'))
    clips = args.steps * S * world
    # ---- side measurements on the same engine (outside the timed region) -----------------------------------------
    side = {}
    side_steps = args.side_steps if args.side_steps is not None else (4 if world == 1 else 0)
    first = args.warmup + args.steps
    if side_steps > 0:
        for mode, k in (('fast', side_steps), ('exact', side_steps), ('fp32', max(1, side_steps // 4))):
            if mode == args.mode:
                continue
            sdt, svotes, sfrac, _ = timed(mode, k, 1, first)
            first += k + 1
            side[mode] = {"clips_per_s": k * S * world / sdt, "steps": k, "votes": svotes}
            if mode == 'exact':
                side[mode]["recheck_frac"], side[mode]["recheck_frac_fp32"] = sfrac
            if mode == 'fp32':
                side[mode]["tflops"] = k * S / sdt * CLIP_FLOP / 1e12
                side[mode]["frac_of_fp32_matrix_peak"] = side[mode]["tflops"] / PEAK_FP32_TFLOPS

    # ---- exact == fp32 on the SAME keys: the first --check-steps timed steps' sample range, once per mode -------------------
    check = None
    if args.check_steps > 0 and args.mode == 'exact':
        k = min(args.check_steps, args.steps)

        def votes_of(mode):
            eng.set_mode(MODES[mode])
            total.zero_()
            for i in range(k):
                step(args.warmup + i)            # the global sample range of timed steps 0..k-1
            fence()
            return total.cpu().tolist()
        v_exact, v_fp32 = votes_of('exact'), votes_of('fp32')
        check = {"exact_equals_fp32": v_exact == v_fp32, "votes_exact": v_exact, "votes_fp32": v_fp32, "samples": k * S * world,
                 "sample_range": [args.warmup * S * world, (args.warmup + k) * S * world],
                 "note": "the same Philox keys as the first %d timed steps, evaluated in the exact-vote mode and on the exact-fp32 path" % k}

    # ---- the other sigmas of BASELINE C4's grid, a few steps each (same engine, same clip, their own sample range) -------------
    grid = []
    for sg in [float(v) for v in args.sigma_grid.split(',') if v.strip()] if args.grid_steps > 0 else []:
        if abs(sg - sigma) < 1e-9:
            continue
        gcfg = sigma_cfg(sg)
        gdt, gvotes, gfrac, _ = timed('exact', args.grid_steps, 1, first, cfg=gcfg)
        fdt, _, _, _ = timed('fast', args.grid_steps, 1, first, cfg=gcfg)

        def gvotes_of(mode):
            eng.set_mode(MODES[mode])
            total.zero_()
            step(first + 1, gcfg)               # the first timed step's keys
            fence()
            return total.cpu().tolist()
        gx, gp = gvotes_of('exact'), gvotes_of('fp32')
        first += args.grid_steps + 1
        grid.append({"sigma": sg, "t_star": gcfg[2] + 1, "steps": args.grid_steps, "clips_per_s": args.grid_steps * S * world / gdt,
                     "fast_mode_clips_per_s": args.grid_steps * S * world / fdt, "exact_over_fast": fdt / gdt,
                     "recheck_frac": gfrac[0], "recheck_frac_fp32": gfrac[1], "votes": gvotes,
                     "exact_equals_fp32": gx == gp, "votes_exact_first_step": gx, "votes_fp32_first_step": gp, "check_samples": S * world})

    full = None
    if not args.no_certify and args.classifier == 'vgg19_bn':
        # the surface the reference's driver calls: RobustCertificate.certify through the host mirror, n_0 pass included
        from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
        from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
        from dmad_hip.transforms import MelSpectrogramDB
        from robustness_eval.certified_robust import RobustCertificate
        eng.set_mode(MODES[args.mode])
        den = DiffWave(WaveNetHIP(eng), hp)
        clf = vgg19_bn(num_classes=10, in_channels=1).eval()
        clf.load_state_dict({k: torch.from_numpy(v) for k, v in csd.items()})
        clf.bind_engine(eng)
        rc = RobustCertificate(classifier=clf, transform=MelSpectrogramDB(eng), denoiser=den, seed=2024)
        assert rc._fused()
        x = clip.reshape(1, 1, L)
        rc.certify(x, torch.tensor([0], device='cuda'), sigma=sigma, n_0=100, n=min(1024, args.full_n), batch_size=args.max_batch)   # warm-up
        eng.recheck_stats(reset=True)
        fence()
        t0 = time.perf_counter()
        y_pred, radius = rc.certify(x, torch.tensor([0], device='cuda'), sigma=sigma, n_0=100, n=args.full_n, batch_size=args.max_batch)
        fence()
        fdt = time.perf_counter() - t0
        voted, rechecked = eng.recheck_stats()
        if dist is not None:
            tmax = torch.tensor([fdt], dtype=torch.float64, device='cpu' if backend == 'gloo' else 'cuda')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            fdt = float(tmax.item())
        full = {"seconds": fdt, "clips_per_s": (100 + args.full_n) * 1.0 / fdt, "n_0": 100, "n": args.full_n, "n_gpus": world,
                "scaling": "strong: one clip's %d samples sharded over %d rank(s)" % (args.full_n, world),
                "y_pred": int(y_pred[0]), "radius": float(radius[0]), "recheck_frac": rechecked / max(voted, 1),
                "vs_steady_state": ((100 + args.full_n) / fdt) / (clips / dt)}

        # BASELINE C3: certify(n_0 = 100, n = 1000) per clip through the same mirror (scripts' N = 1 000 setting), a few clips
        c3 = None
        if args.c3_clips > 0:
            xs = [torch.from_numpy(synth.synthetic_clip(10 + i)).cuda().reshape(1, 1, L) for i in range(args.c3_clips)]
            rc.certify(xs[0], torch.tensor([0], device='cuda'), sigma=sigma, n_0=100, n=1000, batch_size=args.max_batch)
            fence()
            t0 = time.perf_counter()
            outs = [rc.certify(xc, torch.tensor([0], device='cuda'), sigma=sigma, n_0=100, n=1000, batch_size=args.max_batch) for xc in xs]
            fence()
            c3dt = time.perf_counter() - t0
            c3 = {"workload": "BASELINE C3: certify(n_0=100, n=1000, sigma=%.2f) per clip through RobustCertificate, %d clips, %d rank(s)"
                              % (sigma, args.c3_clips, world),
                  "clips_per_s": args.c3_clips * 1100 / c3dt, "seconds_per_certified_clip": c3dt / args.c3_clips,
                  "y_pred": [int(o[0][0]) for o in outs], "radius": [float(o[1][0]) for o in outs]}

    # BASELINE C2: DiffWave DDPM purification t* = 5 of 256 clips + mel-dB + VGG19_bn on a bf16 engine of its own (the main engine
    # is released first: 512-clip exact-vote engine 172 GB + 256-clip bf16 engine 85 GB would not leave room for C5's)
    c2 = None
    if args.c2_iters > 0 and args.classifier == 'vgg19_bn':
        eng.close()
        c2_b, c2_t = args.c2_batch, 5
        eng2 = E.Engine(max_batch=c2_b, precision=E.BF16, half_type=E.HALF_BF16)
        eng2.load_wavenet(wsd)
        eng2.load_vgg19_bn(csd)
        xb = torch.stack([torch.from_numpy(synth.synthetic_clip(100 + i)) for i in range(c2_b)]).cuda()
        from diffusion_models.diffwave_ddpm import DiffWave as _DW, WaveNetHIP as _WN
        ts2, ca2, cb2, ce2, cd2, cs2 = _DW(_WN(eng2), hp, reverse_timestep=c2_t).purify_coefficients()      # the fp32 tables, as DiffWave.forward uses them
        q = dict(sampler=1, t_star=ts2, c_a=ca2, c_b=cb2, c_eps=ce2, c_div=cd2, c_sig=cs2)
        eng2.query_logits(xb, 1, seed=7, **q)                                  # warm-up: step embeddings of t = 0..4
        fence()
        eng2.profile_layers(args.c2_iters * c2_t * 35)
        t0 = time.perf_counter()
        for it in range(args.c2_iters):
            lg2, dec2 = eng2.query_logits(xb, 1, seed=7, sample0=it * c2_b, **q)
        fence()
        c2dt = time.perf_counter() - t0
        eng2.profile_read_final()
        lms, ln = eng2.profile_read()
        c2_ach = LAYER_FLOP_PER_CLIP * c2_b / (lms / ln * 1e-3) / 1e12 if ln else float('nan')
        c2 = {"workload": "BASELINE C2: DiffWave DDPM purify t*=%d (diffusion + %d reverse steps) of %d clips + mel-dB + VGG19_bn, bf16 MFMA engine, "
                          "one dmad_query_logits call per batch" % (c2_t, c2_t, c2_b),
              "clips_per_s": args.c2_iters * c2_b / c2dt, "network_evals_per_s": args.c2_iters * c2_b * c2_t / c2dt,
              "ms_per_batch": c2dt / args.c2_iters * 1e3, "dtype": "bf16", "batch": c2_b, "t_star": c2_t, "iters": args.c2_iters,
              "end_to_end_tflops": args.c2_iters * c2_b * (c2_t * 606.10e9 + 1.10e9) / c2dt / 1e12,
              "roofline": {"bound": "mfma", "kernel": "wn_layer_p<__bf16>", "achieved": c2_ach, "peak": PEAK_MFMA16_TFLOPS, "unit": "TFLOP/s",
                           "frac": c2_ach / PEAK_MFMA16_TFLOPS, "avg_launch_ms": lms / ln if ln else None, "launches_timed": ln},
              "decisions_histogram": torch.bincount(dec2.long().cpu(), minlength=10).tolist()}
        eng2.close()

    c5 = None
    if args.c5_n > 0 and args.classifier == 'vgg19_bn':
        # BASELINE configuration C5 beside the headline: the same vote loop with the spec-domain purifier (dmad_spec_smooth_votes:
        # mel-dB -> standardise -> q_sample(t*) -> 26 UNet evaluations -> classifier) on an engine of its own without WaveNet workspace,
        # in the exact-vote mode (UNet chain on the 16-bit tier, low-margin samples re-run on the exact-fp32 UNet)
        from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion
        c5_b, c5_t = args.c5_batch, 25          # engine batch 2048: the 8x8 / 4x4 maps of the UNet fill the chip (512 -> 2048: + 11 %, same votes)
        eng5 = E.Engine(max_batch=c5_b, precision=E.EXACT, recheck_batch=0, with_wavenet=False)
        eng5.load_vgg19_bn(synth.vgg19_bn_state_dict(4321, calibrated='c5'))      # the statistics of the spec chain's output distribution
        pur = create_improved_diffusion(None, reverse_timestep=c5_t, state_dict=synth.unet_state_dict(31), engine=eng5)
        c5_args = (clip, sigma) + tuple(pur.purify_coefficients()) + (-100.0, 38.22)
        eng5.spec_smooth_votes(*c5_args, c5_b, seed=1)                       # warm-up: one batch fills the per-step tables of all 26 steps

        def c5_run(mode, n_local):
            eng5.set_mode(mode)
            eng5.spec_recheck_stats(reset=True)
            fence()
            t0 = time.perf_counter()
            # every rank takes n_local samples of the global index range (weak scaling), one int64[10] all-reduce at the end
            cnt, _, _ = eng5.spec_smooth_votes(*c5_args, n_local, seed=2024, sample0=rank * n_local)
            if dist is not None:
                if backend == 'gloo':
                    cc = cnt.cpu(); dist.all_reduce(cc); cnt = cc
                else:
                    dist.all_reduce(cnt)
            fence()
            dt5 = time.perf_counter() - t0
            if dist is not None:
                tmax = torch.tensor([dt5], dtype=torch.float64, device='cpu' if backend == 'gloo' else 'cuda')
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt5 = float(tmax.item())
            return dt5, cnt.cpu().tolist(), eng5.spec_recheck_stats(detail=True)
        n32 = max(1, min(args.c5_n, 1024))                                   # the exact-fp32 leg: the same first keys, fewer of them
        c5_dt, c5_counts, (c5_voted, c5_re, c5_re32) = c5_run(E.MODE_EXACT_VOTES, args.c5_n)
        f_dt, f_counts, _ = c5_run(E.MODE_FAST, n32)
        x_dt, x_counts, _ = c5_run(E.MODE_EXACT_VOTES, n32)
        p_dt, p_counts, _ = c5_run(E.MODE_FP32, n32)
        evals = (c5_t + 1) * UNET_FLOP_PER_SPEC
        c5 = {"workload": "BASELINE C5: certified smoothing sigma=%.2f, spec-domain purifier (Improved-Diffusion UNet, t*=%d: %d network "
                          "evaluations per sample) + VGG19_bn" % (sigma, c5_t, c5_t + 1),
              "mode": "exact-vote: UNet chain on the 16-bit tier (f16 operands and maps, fp32 accumulate / GroupNorm statistics / softmax), samples "
                      "with top-2 margin < %.3g re-run on the split-f16 tier (fp32 pipeline, three f16 MFMAs per product), those still below "
                      "%.3g on the exact-fp32 UNet" % (eng5.spec_recheck_margin, eng5.spec_recheck_margin2),
              "classifier": "VGG19_bn (synthetic seed 4321, BatchNorm statistics calibrated on the spec chain's output: several classes vote)",
              "samples_per_s": args.c5_n * world / c5_dt, "n": args.c5_n * world, "n_gpus": world, "seconds": c5_dt, "engine_batch": c5_b,
              "dtype": "f16", "votes": c5_counts, "recheck_frac": c5_re / max(c5_voted, 1), "recheck_frac_fp32": c5_re32 / max(c5_voted, 1),
              "voted_classes": sum(1 for v in c5_counts if v > 0),
              "unet_tflops_per_gpu": args.c5_n * evals / c5_dt / 1e12, "frac_of_mfma16_peak": args.c5_n * evals / c5_dt / 1e12 / PEAK_MFMA16_TFLOPS,
              "fast_mode": {"samples_per_s": n32 * world / f_dt, "n": n32 * world, "votes": f_counts},
              "fp32_mode": {"samples_per_s": n32 * world / p_dt, "n": n32 * world, "votes": p_counts, "unet_tflops_per_gpu": n32 * evals / p_dt / 1e12,
                            "frac_of_fp32_matrix_peak": n32 * evals / p_dt / 1e12 / PEAK_FP32_TFLOPS},
              "exact_over_fast": (args.c5_n / c5_dt) / (n32 / f_dt),
              "exact_equals_fp32": x_counts == p_counts, "votes_exact_same_keys": x_counts, "check_samples": n32 * world}
        eng5.close()

    # The reference script's DEFAULT classifier (certified_robustness_eval.py:57: ResNeXt29 8x64d) in place of VGG19_bn: the same step on
    # an exact-vote engine of its own (tier 1 = f16 WaveNet + the classifier's 16-bit tier, recheck tiers = split-f16 / fp32 WaveNet + fp32
    # classifier), a few timed steps, the 16-bit tiers alone, and exact == fp32 on the first timed step's keys
    rxm = None
    if args.resnext_steps > 0 and args.classifier == 'vgg19_bn':
        eng.close()
        engr = E.Engine(max_batch=args.max_batch, precision=E.EXACT, half_type=half, recheck_batch=min(args.recheck_batch, args.max_batch),
                        recheck_margin=args.recheck_margin)
        engr.load_wavenet(wsd)
        engr.load_resnext29(synth.resnext29_state_dict(2929))
        cur['eng'] = engr
        rdt, rvotes, rfrac, _ = timed('exact', args.resnext_steps, 2, first)
        fdt, _, _, _ = timed('fast', args.resnext_steps, 1, first)

        def rvotes_of(mode):
            engr.set_mode(MODES[mode])
            total.zero_()
            step(first + 2)                     # the first timed step's keys
            fence()
            return total.cpu().tolist()
        rx_, rp_ = rvotes_of('exact'), rvotes_of('fp32')
        rxm = {"workload": "the headline's step with the reference script's default classifier: ResNeXt29 8x64d (synthetic seed 2929), sigma=%.2f" % sigma,
               "clips_per_s": args.resnext_steps * S * world / rdt, "steps": args.resnext_steps, "fast_mode_clips_per_s": args.resnext_steps * S * world / fdt,
               "vs_vgg_headline": (args.resnext_steps * S * world / rdt) / (clips / dt), "recheck_frac": rfrac[0], "recheck_frac_fp32": rfrac[1],
               "recheck_margin": engr.recheck_margin, "votes": rvotes, "voted_classes": sum(1 for v in rvotes if v > 0),
               "exact_equals_fp32": rx_ == rp_, "votes_exact_first_step": rx_,
               "votes_fp32_first_step": rp_, "check_samples": S * world,
               "classifier_tiers": "exact-vote mode: the classifier's split-f16 tier (three f16 MFMAs per product, fp32-grade) in the first pass and "
                                   "the split-f16 recheck tier, fp32 matrix cores in the last tier (the f16 classifier's error does not fit under the "
                                   "bound: profiles/r05b_resnext29_error_attribution.json); fast mode: gemm_h16 (f16 operands, fp32 accumulate)"}
        engr.close()
        cur['eng'] = eng

    if rank == 0:
        out = {
            "metric": "purified+classified 1s clips/sec at N=100k sigma=0.5; 1/2/4/8 GPUs",
            "value": clips / dt, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"exact": args.half, "fast": args.half, "fp32": "f32"}[args.mode], "data": "synthetic",
            "config": {"workload": "certified smoothing N=100000 sigma=%.2f (t*=%d): DiffWave one-shot purify (36x256 WaveNet) "
                                   "+ mel-dB + %s + votes; step = %d Monte Carlo samples per GPU, %d steps = one "
                                   "N=100000 clip" % (sigma, t + 1, 'VGG19_bn' if args.classifier == 'vgg19_bn' else 'ResNeXt29', S, -(-100000 // S)),
                       "samples_per_step_per_gpu": S, "engine_batch": args.max_batch, "sigma": sigma, "t_star": t + 1,
                       "mode": {"exact": "exact-vote: %s MFMA WaveNet + split-f16 / exact-fp32 re-evaluation of samples with top-2 logit margin < %.3g "
                                         "(empirical bound: no vote differed from the fp32 path's on 236 864 samples (36 864 + 2 x 100 000); this "
                                         "line's own check: exact_equals_fp32)" % (args.half, recheck_margin),
                                "fast": "%s MFMA WaveNet alone (no recheck)" % args.half, "fp32": "exact-fp32 WaveNet alone"}[args.mode],
                       "noise": "device Philox4x32-10", "classifier": "VGG19_bn (synthetic seed 4321)" if args.classifier == 'vgg19_bn' else "ResNeXt29 8x64d (synthetic seed 2929)",
                       "parallelism": "mc-samples sharded x%d, one int64[10] all-reduce per step" % world},
            "end_to_end_tflops": clips / dt / world * CLIP_FLOP / 1e12,
            "votes": votes,
        }
        if args.mode == 'exact':
            out["recheck"] = {"margin": recheck_margin, "frac": recheck_frac[0], "margin_split_f16_tier": recheck_margin2,
                              "frac_fp32": recheck_frac[1], "batch": min(args.recheck_batch, args.max_batch),
                              "tiers": "16-bit MFMA -> (margin < %.3g) split-f16 three-MFMA fp32 pipeline -> (margin < %.3g) exact fp32"
                                       % (recheck_margin, recheck_margin2)}
            if check is not None:
                out["exact_equals_fp32"] = check["exact_equals_fp32"]
                out["exact_vs_fp32_check"] = check
        if prof is not None:
            (layer_ms, launches), (final_ms, flaunches) = prof
            # launches of a step's last (possibly smaller) chunk and of the recheck passes carry fewer clips; the recheck
            # passes run on the fp32 path and are not bracketed, so every bracketed launch belongs to a 16-bit chunk
            clips_per_launch = (args.steps * S) / (launches / 35.0) if launches else 0.0
            avg_ms = layer_ms / launches if launches else float('nan')
            achieved = LAYER_FLOP_PER_CLIP * clips_per_launch / (avg_ms * 1e-3) / 1e12 if launches else float('nan')
            traffic, traffic_file = layer_traffic_per_clip()
            out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_MFMA16_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / PEAK_MFMA16_TFLOPS,
                               "traffic": (traffic * clips_per_launch if traffic else None),
                               "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/%s "
                                                  "(2 x FETCH_SIZE + WRITE_SIZE, bytes per launch)" % traffic_file) if traffic else None,
                               "kernel": "wn_layer_p<%s>" % ('_Float16' if args.half == 'f16' else '__bf16'), "avg_launch_ms": avg_ms,
                               "launches_timed": launches, "flop_per_launch": LAYER_FLOP_PER_CLIP * clips_per_launch}
            if flaunches:
                fclips = (args.steps * S) / flaunches
                favg = final_ms / flaunches
                fach = FINAL_BYTES_PER_CLIP * fclips / (favg * 1e-3) / 1e9
                out["roofline_final"] = {"bound": "hbm", "achieved": fach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": fach / PEAK_HBM_GBS,
                                         "traffic": None, "kernel": "wn_final_p", "avg_launch_ms": favg, "launches_timed": flaunches,
                                         "bytes_per_launch": FINAL_BYTES_PER_CLIP * fclips,
                                         "matrix_tflops_same_launch": 2.0 * L * 256 * (36 * 256 + 256) * fclips / (favg * 1e-3) / 1e12}
        for mode, rec in side.items():
            out[mode + "_mode"] = rec
        if grid:
            out["sigma_grid"] = grid
        if full is not None:
            out["certify_full"] = full
            if c3 is not None:
                out["c3_certify_n1000"] = c3
        if c2 is not None:
            out["c2_ddpm_mode"] = c2
        if c5 is not None:
            out["c5_spec_mode"] = c5
        if rxm is not None:
            out["resnext29_mode"] = rxm
        if world == 1 and not args.no_cpu_baseline and args.classifier == 'vgg19_bn':
            out["cpu_baseline"] = cpu_baseline(args.cpu_samples)
        print(json.dumps(out), flush=True)
    eng.close()                                # (a no-op when the C2 leg already released it)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
