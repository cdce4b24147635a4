#!/usr/bin/env python3
"""Development: one ResNeXt29 classify of B spectrograms (after a warm-up) for a rocprofv3 kernel trace; with
--analyse DIR prints per-launch TFLOP/s of the gemm_f32 launches of the LAST forward in launch order."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(os.environ.get('B', 128))


def launches():
    """(name, flops) of every gemm launch of CifarResNeXt.forward in order (conv1 is not a gemm)."""
    out, H = [], 32
    stages = [64, 256, 512, 1024]
    for st in range(3):
        for k in range(3):
            cin = stages[st] if k == 0 else stages[st + 1]
            cout = stages[st + 1]
            D = 8 * (64 * cout // 256)
            stride = 2 if (k == 0 and st > 0) else 1
            Ho = H // stride
            out.append(('s%d.b%d reduce %dx%d->%d' % (st + 1, k, H, cin, D), 2.0 * B * H * H * cin * D))
            out.append(('s%d.b%d conv3x3 g8 %d->%dx%d' % (st + 1, k, H, Ho, D), 2.0 * B * Ho * Ho * D * (D // 8) * 9))
            if cin != cout:
                out.append(('s%d.b%d short %d->%d' % (st + 1, k, cin, cout), 2.0 * B * Ho * Ho * cin * cout))
            out.append(('s%d.b%d expand %d->%d' % (st + 1, k, D, cout), 2.0 * B * Ho * Ho * D * cout))
            H = Ho
    out.append(('fc', 2.0 * B * 1024 * 10))
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
        print('%-34s %9.1f us  %6.1f TFLOP/s  grid %s' % (name, us, fl / us / 1e6, r.get('Grid_Size', '?')))
    print('total gemm %.1f us' % tot)
else:
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
    from dmad_hip import engine as E, synth
    eng = E.Engine(max_batch=B, precision=E.BF16)
    eng.load_resnext29(synth.resnext29_state_dict(2929))
    spec = torch.randn(B, 1, 32, 32, device='cuda') * 15 - 25
    eng.classify(spec); torch.cuda.synchronize()
    eng.classify(spec); torch.cuda.synchronize()
