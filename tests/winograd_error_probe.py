#!/usr/bin/env python3
"""CPU probe for the Winograd decision: how much logit error would a Winograd F(2,3) form of the dilated
conv add on top of the 16-bit tier's operand roundings?  Test infrastructure (it imports the oracle, which only tests/, smoke()
and bench.py's cpu_baseline may do); not collected by pytest:   python tests/winograd_error_probe.py [samples]

The dilated 3-tap conv (WaveNet.py:86, dilation d) restricted to one residue class t = r (mod d) is an ordinary 3-tap conv over
s_j = h[r + j d].  F(2,3) computes the output pair (y_2m, y_2m+1) from the inputs s_2m-1 .. s_2m+2 with FOUR channel-contracting
products instead of six:
    U = [d0 - d2, d1 + d2, d2 - d1, d1 - d3],   G = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2],
    M_i = G_i U_i,   y_2m = M_0 + M_1 + M_2,   y_2m+1 = M_1 - M_2 - M_3.
On the 16-bit tier BOTH transformed operands would be rounded to f16 (the MFMA eats f16): the transformed inputs carry the
rounding of a sum / difference of two stream values, the transformed weights that of a three-term combination.  Variants,
all with fp32 accumulation and everything outside the dilated conv exact, against the exact-fp32 oracle on the same noise:
    direct : f16(W_dil) * f16(h)                 — what the 16-bit tier's GEMM1 does today
    wino   : f16(G) * f16(U), U from f16(h)      — Winograd on the stored f16 stream (the realistic form)
    wino_x : f16(G) * f16(U), U from exact h     — transform error alone
Prints the leader-difference logit error statistic (max_j |e_j - e_i|) of each variant.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import synth                      # noqa: E402
from oracle import dmad_oracle as orc           # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
SIGMA = 0.5


def r16(x):
    return x.half().float()


def wino_conv(h, w, b, d, round_in, round_u=True):
    """Dilated 3-tap conv as F(2,3) over the residue classes mod d; h [B,C,L], w [M,C,3]."""
    B, C, L = h.shape
    if round_in:
        h = r16(h)
    J = -(-L // d)
    J2 = J + (J & 1)                                        # an even number of positions per residue class
    hp = F.pad(h, (0, J2 * d - L)).view(B, C, J2, d)        # [B, C, j, r], t = j d + r
    s = F.pad(hp, (0, 0, 1, 1))                             # one zero position before and after each class (the conv's zero padding)
    d0, d1, d2, d3 = s[:, :, 0:J2:2], s[:, :, 1:J2 + 1:2], s[:, :, 2:J2 + 2:2], s[:, :, 3:J2 + 3:2]      # [B, C, J2/2, d]
    U = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    G = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
    if round_u:
        U = [r16(u) for u in U]
        G = [r16(g) for g in G]
    M = [torch.einsum('mc,bcjr->bmjr', G[i], U[i]) for i in range(4)]
    y0, y1 = M[0] + M[1] + M[2], M[1] - M[2] - M[3]
    y = torch.stack([y0, y1], dim=3).reshape(B, w.shape[0], J2, d).reshape(B, w.shape[0], J2 * d)[:, :, :L]
    return y + b.view(1, -1, 1)


def forward(w, audio, steps, conv):
    """oracle.wavenet_forward with the dilated conv replaced by `conv(h, W, b, d)` (WaveNet.py:75-97,120-135,164-172)."""
    x = F.conv1d(audio, w['init.w'], w['init.b'])
    x = torch.maximum(x, torch.zeros_like(x))
    emb = orc.step_embedding(steps, w['fc_t1.w'].shape[1])
    emb = orc.swish(F.linear(emb, w['fc_t1.w'], w['fc_t1.b']))
    emb = orc.swish(F.linear(emb, w['fc_t2.w'], w['fc_t2.b']))
    skip = 0
    Bn, C, _ = x.shape
    for n in range(36):
        d = 2 ** (n % 12)
        h = x + F.linear(emb, w['fc_t.%d.w' % n], w['fc_t.%d.b' % n]).view(Bn, C, 1)
        H = conv(h, w['dil.%d.w' % n], w['dil.%d.b' % n], d)
        out = torch.tanh(H[:, :C]) * torch.sigmoid(H[:, C:])
        x = (h + F.conv1d(out, w['res.%d.w' % n], w['res.%d.b' % n])) * np.sqrt(0.5)
        skip = skip + F.conv1d(out, w['skip.%d.w' % n], w['skip.%d.b' % n])
    y = F.relu(F.conv1d(skip * np.sqrt(1.0 / 36), w['f0.w'], w['f0.b']))
    return F.conv1d(y, w['f2.w'], w['f2.b'])


def main():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    sd, vsd = synth.wavenet_state_dict(1234), synth.vgg19_bn_state_dict(4321)
    w = orc.folded_weights(sd)
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    t = orc.compute_t_star(ab, SIGMA) - 1
    c_a, c_b = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])
    clip = torch.from_numpy(synth.synthetic_clip(0))
    z = torch.from_numpy(np.stack([orc.philox_normal(2024, i, 0, 16000) for i in range(N)])).float()
    x_t = (float(torch.tensor((1 / (1 + SIGMA ** 2)) ** 0.5)) * (clip.view(1, 1, -1) + SIGMA * z.unsqueeze(1))).float()
    steps = torch.full((N, 1), float(t))
    variants = {
        'exact (oracle fp32)': lambda h, W, b, d: F.conv1d(h, W, b, dilation=d, padding=d),
        'winograd, no rounding (algebra check)': lambda h, W, b, d: wino_conv(h, W, b, d, False, False),
        'direct: f16(W) * f16(h)': lambda h, W, b, d: F.conv1d(r16(h), r16(W), b, dilation=d, padding=d),
        'wino: f16(G) * f16(U), U from f16(h)': lambda h, W, b, d: wino_conv(h, W, b, d, True),
        'wino_x: f16(G) * f16(U), U from exact h': lambda h, W, b, d: wino_conv(h, W, b, d, False),
    }
    logits = {}
    with torch.no_grad():
        for name, conv in variants.items():
            t0 = time.time()
            eps = forward(w, x_t, steps, conv)
            x0 = c_a * x_t - c_b * eps
            logits[name] = orc.vgg19_bn_forward(vsd, orc.mel_db(x0)).double()
            print('%-44s %.0f s' % (name, time.time() - t0), flush=True)
    ref = logits['exact (oracle fp32)']
    lead = ref.argmax(1, keepdim=True)
    print('\nleader-difference logit error over %d samples (sigma = %.2f, clip 0), dilated conv alone perturbed:' % (N, SIGMA))
    out = {}
    for name, lg in logits.items():
        e = lg - ref
        le = (e - e.gather(1, lead)).abs().max(1).values
        out[name] = (float(le.max()), float((le ** 2).mean().sqrt()))
        print('  %-44s max %.3e   rms %.3e' % (name, *out[name]))
    d, wv = out['direct: f16(W) * f16(h)'][1], out['wino: f16(G) * f16(U), U from f16(h)'][1]
    print('\nWinograd / direct (rms): %.2f x;  in variance: %.2f x of GEMM1\'s share of the 16-bit tier\'s error' % (wv / d, (wv / d) ** 2))


if __name__ == '__main__':
    main()
