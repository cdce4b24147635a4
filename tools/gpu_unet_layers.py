#!/usr/bin/env python3
"""Development: one UNet eps evaluation of B spectrograms (after a warm-up at the same step) for a rocprofv3 kernel
trace; with --analyse DIR prints per-launch TFLOP/s of the gemm_f32 launches of the LAST forward in launch order."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
B = int(os.environ.get('B', 128))


def launches():
    from dmad_hip import synth
    _, inp, mid, outp = synth.unet_layout()
    out, H = [], 32

    def ops(blk):
        nonlocal H
        for p, kind, cin, cout in blk:
            n = B * H * H
            if kind == 'res':
                out.append(('%s conv1 %d->%d @%d' % (p, cin, cout, H), 2.0 * n * cin * cout * 9))
                if cin != cout:
                    out.append(('%s skip %d->%d @%d' % (p, cin, cout, H), 2.0 * n * cin * cout))
                out.append(('%s conv2 %d->%d @%d' % (p, cout, cout, H), 2.0 * n * cout * cout * 9))
            elif kind == 'attn':
                out.append(('%s qkv %d @%d' % (p, cin, H), 2.0 * n * cin * 3 * cin))
                out.append(('%s proj %d @%d' % (p, cin, H), 2.0 * n * cin * cin))
            elif kind == 'down':
                H //= 2
                out.append(('%s down %d @%d' % (p, cin, H), 2.0 * B * H * H * cin * cout * 9))
            elif kind == 'up':
                H *= 2
                out.append(('%s up %d @%d' % (p, cin, H), 2.0 * B * H * H * cin * cout * 9))
    for b in inp:
        ops(b)
    ops(mid)
    for b in outp:
        ops(b)
    return out              # the 128 -> 1 output conv is a direct kernel (conv3x3_c128_to1), not a gemm_f32 launch


if len(sys.argv) > 2 and sys.argv[1] == '--analyse':
    path = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    pat = os.environ.get('KERNEL', 'gemm_f32_kernel')          # KERNEL=gemm_h16: the 16-bit tier's launches (same launch order)
    rows = [r for r in csv.DictReader(open(path)) if pat in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    L = launches()
    rows = rows[-len(L):]
    tot, worst = 0.0, []
    for (name, fl), r in zip(L, rows):
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += us
        worst.append((us, fl / us / 1e6, name))
    if os.environ.get('ALL'):                                  # every launch in launch order, with the kernel that served it
        for (name, fl), r in zip(L, rows):
            us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            kn = r['Kernel_Name'].split('(')[0].replace('void dmad::', '')
            print('%-52s %9.1f us  %6.1f TFLOP/s  %s grid %s' % (name, us, fl / us / 1e6, kn, r.get('Grid_Size', '?')))
    for us, tf, name in sorted(worst, reverse=True)[:25]:
        print('%-52s %9.1f us  %6.1f TFLOP/s' % (name, us, tf))
    print('total gemm %.1f us over %d launches; flops %.2f T -> %.1f TFLOP/s' % (tot, len(L), sum(f for _, f in L) / 1e12, sum(f for _, f in L) / tot / 1e6))
else:
    import time
    import torch
    from dmad_hip import _lib
    if os.environ.get('DMAD_LIB'):               # A/B of two builds on one box: DMAD_LIB=/path/to/other/libdmad_hip.so
        _lib.LIB_PATH = os.environ['DMAD_LIB']
    from dmad_hip import engine as E, synth
    TIER = int(os.environ.get('TIER', 1))          # 1: the 16-bit tier (default), 0: exact fp32, 2: split-f16 (KERNEL=gemm_x3 for --analyse)
    eng = E.Engine(max_batch=B, precision=E.EXACT if TIER == 2 else E.BF16, with_classifier=False, with_wavenet=False)     # no WaveNet workspace
    eng.load_unet(synth.unet_state_dict(5252))
    x = torch.randn(B, 32, 32, device='cuda') * 0.5
    _eps = eng.unet_eps
    eng.unet_eps = lambda xx, t: _eps(xx, t, tier=TIER)
    eng.unet_eps(x, 40); torch.cuda.synchronize()
    eng.unet_eps(x, 40); torch.cuda.synchronize()
    if len(sys.argv) > 1 and sys.argv[1] == '--time':          # wall time of REPS evaluations (alternating steps, as in the sampler)
        reps = int(os.environ.get('REPS', 20))
        t0 = time.time()
        for i in range(reps):
            eng.unet_eps(x, 40 - (i & 1))
        torch.cuda.synchronize()
        ms = (time.time() - t0) / reps * 1e3
        print('%s: %.3f ms per evaluation of %d spectrograms -> %.1f TFLOP/s fp32' % (os.environ.get('DMAD_LIB', 'in-tree build'), ms, B, sum(f for _, f in launches()) / ms / 1e9), flush=True)
