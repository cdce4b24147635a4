#!/usr/bin/env python3
"""CPU: turn a flip study's raw logits into a STATED miss probability for the recheck bound tau1.

A sample is mis-voted by the exact-vote mode only if it votes on the 16-bit tier (margin >= tau1) with a leader other than the
exact one, which needs E_s = max_j |e_j - e_i| >= tau1 for that sample (e = 16-bit minus exact logits, i = exact arg-max; see
dmad_hip/engine.py) AND an exact top-2 margin m_s < E_s - tau1 (the error has to carry the wrong class from behind the exact
leader to tau1 in front of it).  So
    P(E_s >= tau1)  >=  P(miss per sample)  ~=  Integral_{x > tau1} p_E(x) F_m(x - tau1) dx        (`stated`, `with_margin`),
F_m = the empirical CDF of the exact path's top-2 margin near zero (error and margin taken as independent: their measured
correlation is printed).  The study gives E_s for 36 864 samples; the tail beyond the largest observation is extrapolated two
ways and the more pessimistic one is reported:
  * Gaussian-like body: E_s is the maximum of 9 differences of roughly normal errors -> fit sigma to the UPPER quantiles
    (q90..q99.9: a body fit would underestimate a heavier tail), P = 9 * 2 * Q(tau / sigma);
  * peaks over threshold: the excesses of the top 1 % over their threshold, fitted as a generalised Pareto (shape, scale by
    probability-weighted moments); P = 0.01 * GPD tail(tau - u).
    python tools/fit_recheck_tail.py gpurun_out/flip_study_f16.npz [tau ...]      -> prints a markdown table + JSON
"""
import json
import math
import sys

import numpy as np


def lead_err(b, f):
    e = b.astype(np.float64) - f.astype(np.float64)
    rows = np.arange(len(f))
    return np.abs(e - e[rows, f.argmax(1)][:, None]).max(1)


def qnorm_tail(x):          # Q(x) = P(N(0,1) > x)
    return 0.5 * math.erfc(x / math.sqrt(2.0))


def margins(f):
    srt = np.sort(f.astype(np.float64), 1)
    return srt[:, -1] - srt[:, -2]


def fit(le, taus, mg=None):
    le = np.sort(le)
    n = len(le)
    out = {'n': n, 'max': float(le[-1]), 'rms': float(np.sqrt((le ** 2).mean())), 'quantiles': {q: float(np.quantile(le, q)) for q in (0.5, 0.9, 0.99, 0.999, 0.9999)}}
    # (a) Gaussian scale from the upper quantiles of the max of 9 |normal| differences: P(E > x) ~= 18 Q(x / s)
    ss = []
    for q in (0.9, 0.99, 0.999):
        x = np.quantile(le, q)
        lo, hi = 1e-6, 10.0          # solve 18 Q(z) = 1 - q for z
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if 18 * qnorm_tail(mid) > 1 - q:
                lo = mid
            else:
                hi = mid
        ss.append(x / lo)
    s_gauss = max(ss)
    out['gauss_sigma'] = float(s_gauss)
    # (b) generalised Pareto over the top 1 %
    k = max(50, n // 100)
    u = le[-k - 1]
    exc = le[-k:] - u
    m0 = exc.mean()
    # probability-weighted moments (Hosking & Wallis): xi = 2 - m0 / (m0 - 2 m1), beta = 2 m0 m1 / (m0 - 2 m1)
    ranks = (np.arange(1, k + 1) - 0.35) / k
    m1 = (exc * (1 - ranks)).mean()
    xi = 2 - m0 / (m0 - 2 * m1)
    beta = 2 * m0 * m1 / (m0 - 2 * m1)
    out['gpd'] = {'threshold': float(u), 'k': int(k), 'shape_xi': float(xi), 'scale_beta': float(beta),
                  'endpoint': float(u - beta / xi) if xi < 0 else None}
    out['miss_probability'] = {}
    for tau in taus:
        pg = min(1.0, 18 * qnorm_tail(tau / s_gauss))
        y = tau - u
        if xi < 0 and y >= -beta / xi:
            pp = 0.0
        elif abs(xi) < 1e-9:
            pp = (k / n) * math.exp(-y / beta)
        else:
            pp = (k / n) * max(0.0, 1 + xi * y / beta) ** (-1 / xi)
        rec = {'gaussian': pg, 'gpd': pp, 'stated': max(pg, pp), 'empirical_frac_above': float((le >= tau).mean())}
        if mg is not None:
            # density of the exact margin near zero (per unit logit), from the samples below 0.05; F_m(y) ~= rho * y for small y
            rho = float((mg < 0.05).mean() / 0.05)
            # GPD tail beyond tau: mean excess = (beta + xi (tau - u)) / (1 - xi); Gaussian: mean excess ~= s^2 / tau
            me_gpd = max(0.0, (beta + xi * (tau - u)) / (1 - xi)) if xi < 1 else float('inf')
            me_g = s_gauss ** 2 / tau
            rec['margin_density_per_unit'] = rho
            rec['with_margin'] = max(pg * rho * me_g, pp * rho * me_gpd)
        out['miss_probability'][str(tau)] = rec
    return out


def main():
    path = sys.argv[1]
    taus = [float(t) for t in sys.argv[2:]] or [0.0244, 0.03, 0.034, 0.04, 0.045, 0.05]
    z = np.load(path)
    cells = sorted(k[5:] for k in z.files if k.startswith('bf16_'))
    per, groups = {}, {}
    for c in cells:
        le, mg = lead_err(z['bf16_' + c], z['fp32_' + c]), margins(z['fp32_' + c])
        per[c] = {'max': float(le.max()), 'rms': float(np.sqrt((le ** 2).mean())), 'n': int(len(le)),
                  'corr_err_margin': float(np.corrcoef(le, np.minimum(mg, 1.0))[0, 1])}
        for g in ('all', 'sigma=' + c.split('_s')[1]):
            groups.setdefault(g, ([], []))
            groups[g][0].append(le); groups[g][1].append(mg)
    res = {'file': path, 'cells': per, 'groups': {g: fit(np.concatenate(a), taus, np.concatenate(b)) for g, (a, b) in groups.items()}}
    print(json.dumps(res, indent=1))
    for g, r in res['groups'].items():
        print('\n%s: n = %d, max %.4f, rms %.4f, q99.9 %.4f; Gaussian sigma %.5f; GPD over the top %d: shape %.3f, scale %.5f' %
              (g, r['n'], r['max'], r['rms'], r['quantiles'][0.999], r['gauss_sigma'], r['gpd']['k'], r['gpd']['shape_xi'], r['gpd']['scale_beta']))
        print('| tau1 | P(E_s >= tau1): Gaussian tail | generalised-Pareto tail | stated P(E_s >= tau1) (max) | P(miss) with the margin condition |')
        print('|---|---|---|---|---|')
        for tau in taus:
            p = r['miss_probability'][str(tau)]
            print('| %.4g | %.2e | %.2e | %.2e | %.2e |' % (tau, p['gaussian'], p['gpd'], p['stated'], p['with_margin']))


if __name__ == '__main__':
    main()
