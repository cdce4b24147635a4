#!/usr/bin/env python3
"""Development: one VGG19_bn classify of B spectrograms (after a warm-up) for a rocprofv3 kernel trace; with --analyse DIR
prints per-launch TFLOP/s of the gemm_f32 launches of the LAST forward (15 convs + 3 linears) in launch order."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(os.environ.get('B', 512))
CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def launches():
    out, H, cin, first = [], 32, 1, True
    for v in CFG:
        if v == 'M':
            H //= 2
            continue
        if not first:
            out.append(('conv %d->%d @%d' % (cin, v, H), 2.0 * B * H * H * cin * v * 9))
        first, cin = False, v
    for o, k in ((4096, 512), (4096, 4096), (10, 4096)):
        out.append(('fc %d->%d' % (k, o), 2.0 * B * o * k))
    return out


if len(sys.argv) > 2 and sys.argv[1] == '--analyse':
    path = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if 'gemm_f32_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    L = launches()
    rows = rows[-len(L):]
    tot = 0.0
    for (name, fl), r in zip(L, rows):
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += us
        print('%-24s %9.1f us  %6.1f TFLOP/s' % (name, us, fl / us / 1e6))
    print('total gemm %.1f us; %.1f TFLOP/s' % (tot, sum(f for _, f in L) / tot / 1e6))
else:
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
    from dmad_hip import engine as E, synth
    eng = E.Engine(max_batch=B, precision=E.BF16, clip_len=16000)
    eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321))
    spec = torch.randn(B, 1, 32, 32, device='cuda') * 15 - 25
    eng.classify(spec); torch.cuda.synchronize()
    eng.classify(spec); torch.cuda.synchronize()
