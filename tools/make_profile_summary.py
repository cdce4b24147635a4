#!/usr/bin/env python3
"""Turn the rocprofv3 output of a bench.py run into the summaries kept under profiles/.

    python tools/make_profile_summary.py NAME --stats DIR [--fetch DIR --write DIR --mfma DIR] [--bench-line FILE]

DIR are `rocprofv3 -d` output directories (CSV output format):
  --stats  : `--kernel-trace --stats` run           -> profiles/NAME_kernel_stats.{md,csv}
  --fetch  : `--pmc FETCH_SIZE` pass                 \\
  --write  : `--pmc WRITE_SIZE` pass                  > profiles/NAME_layer_traffic.json (+ a PMC section in the .md)
  --mfma   : `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass
FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request; MI355X_MICROARCH.md, HBM section).
"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-name patterns, demangled (kernel stats) or mangled (counter collection): a name matches if ALL substrings of ONE
# alternative occur in it (either operand type)
LAYER = (('wn_layer_p<', 'false, false>'), ('wn_layer_pI', 'Lb0ELb0E'))
FINAL = (('wn_final_p<',), ('wn_final_pI',))
KERNEL_SOURCES = ('wn_layer.hip', 'wn_final.hip', 'wn_bf16.h', 'dmad_common.h')      # what the committed counters belong to (bench.py hashes the same files)


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    return hits[0] if hits else None


def counters(d):
    """{(kernel substring key, counter): [values per dispatch]}"""
    out = defaultdict(list)
    path = one(os.path.join(d, '**', '*counter_collection.csv'))
    if not path:
        return out
    for r in csv.DictReader(open(path)):
        out[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
    return out


def mean_for(cnt, kernel_sub, counter):
    vals = [v for (k, c), vs in cnt.items() if any(all(sub in k for sub in alt) for alt in kernel_sub) and c == counter for v in vs]
    return sum(vals) / len(vals) if vals else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('name')
    ap.add_argument('--stats', required=True)
    ap.add_argument('--fetch'); ap.add_argument('--write'); ap.add_argument('--mfma')
    ap.add_argument('--bench-line')
    ap.add_argument('--clips-per-launch', type=int, default=None, help='clips one layer launch carries (default: engine_batch of the bench line, else 512)')
    ap.add_argument('--command', default='python3 bench.py --no-cpu-baseline --side-steps 0 --c5-n 0 --c2-iters 0 --check-steps 0 --grid-steps 0 --resnext-steps 0 --no-certify --steps 5 --warmup 1')
    a = ap.parse_args()
    prof = os.path.join(ROOT, 'profiles')
    os.makedirs(prof, exist_ok=True)
    stats = one(os.path.join(a.stats, '**', '*kernel_stats.csv'))
    shutil.copy(stats, os.path.join(prof, a.name + '_kernel_stats.csv'))
    rows = list(csv.DictReader(open(stats)))
    md = ['# %s — rocprofv3 of `%s` on 1x MI355X' % (a.name, a.command), '',
          'Command: `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- %s`' % a.command, '']
    if a.bench_line and os.path.exists(a.bench_line):
        line = [l for l in open(a.bench_line) if l.strip().startswith('{')][-1]
        j = json.loads(line)
        shutil.copy(a.bench_line, os.path.join(prof, a.name + '_profiled_bench_line.json'))
        md += ['bench line of the profiled run: %.1f %s, roofline.achieved %.0f %s (layer kernel by HIP events).' % (
            j['value'], j['unit'], j['roofline']['achieved'], j['roofline']['unit']), '']
    if a.clips_per_launch is None:
        a.clips_per_launch = int(j['config']['engine_batch']) if (a.bench_line and os.path.exists(a.bench_line)) else 512
    layer_us = next((float(r['AverageNs']) / 1e3 for r in rows if any(all(p in r['Name'] for p in alt) for alt in LAYER)), None)
    md += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in rows[:18]:
        md.append('| %s | %s | %.3f | %.1f | %s |' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                     float(r['AverageNs']) / 1e3, r['Percentage']))
    if a.fetch and a.write:
        f, w = counters(a.fetch), counters(a.write)
        fl, wl = mean_for(f, LAYER, 'FETCH_SIZE'), mean_for(w, LAYER, 'WRITE_SIZE')
        ff, wf = mean_for(f, FINAL, 'FETCH_SIZE'), mean_for(w, FINAL, 'WRITE_SIZE')
        traffic = (2 * fl + wl) * 1024 / a.clips_per_launch
        # the kernel source these counters belong to: bench.py drops `roofline.traffic` when the layer kernel has changed since
        # (since round 5 the hash covers the tail kernel and the shared headers too: KERNEL_SOURCES, the same list bench.py hashes)
        import hashlib
        h = hashlib.sha256()
        for name in KERNEL_SOURCES:
            h.update(open(os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd', 'csrc', name), 'rb').read())
        json.dump({'layer_traffic_bytes_per_clip': traffic, 'fetch_kb': fl, 'write_kb': wl,
                   'final_traffic_bytes_per_clip': (2 * ff + wf) * 1024 / a.clips_per_launch if ff and wf else None,
                   'kernel_sources': list(KERNEL_SOURCES), 'layer_kernel_sha16': h.hexdigest()[:16]},
                  open(os.path.join(prof, a.name + '_layer_traffic.json'), 'w'))
        md += ['', '## PMC (separate --pmc passes of the same program)', '',
               'Per launch of `%s` (%d clips), mean over the launches: FETCH_SIZE %.4g KB, WRITE_SIZE %.4g KB.' % ('wn_layer_p<T, false, false>', a.clips_per_launch, fl, wl),
               'gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> x2.  Traffic = 2 x %.3f GB + %.3f GB = %.3f GB per launch = %.1f MB per clip per launch'
               % (fl * 1024 / 1e9, wl * 1024 / 1e9, (2 * fl + wl) * 1024 / 1e9, traffic / 1e6),
               '(algorithmic: read h 8.19 MB + write h\' 8.19 MB + write g 8.19 MB = 24.6 MB per clip).']
        if ff:
            md.append('Per launch of `wn_final_p<T>`: FETCH_SIZE %.4g KB (x2 = %.2f GB), WRITE_SIZE %.4g KB.' % (ff, 2 * ff * 1024 / 1e9, wf or 0))
    if a.mfma:
        m = counters(a.mfma)
        busy, gui = mean_for(m, LAYER, 'SQ_VALU_MFMA_BUSY_CYCLES'), mean_for(m, LAYER, 'GRBM_GUI_ACTIVE')
        if busy and gui:
            md += ['', 'MFMA (layer kernel, per launch): SQ_VALU_MFMA_BUSY_CYCLES %.4g, GRBM_GUI_ACTIVE %.4g (sum over 8 XCDs).' % (busy, gui),
                   'MfmaUtil = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs) = %.1f %%' % (100 * busy / (gui / 8 * 1024))]
            if layer_us:
                md += ['Effective clock = (GUI_ACTIVE / 8) cycles per launch / %.1f us (average launch of the --stats pass) = %.2f GHz'
                       % (layer_us, gui / 8 / layer_us / 1e3)]
    open(os.path.join(prof, a.name + '_kernel_stats.md'), 'w').write('\n'.join(md) + '\n')
    print('\n'.join(md))


if __name__ == '__main__':
    main()
