#!/usr/bin/env python3
"""Development: split a rocprofv3 --kernel-trace database of tools/gpu_tier_time.py (PATHS=2) into the split-f16 tier's launch
kinds (in launch order: (dilated conv, res conv) x 35, dilated conv, skip GEMM, final_conv.0) — median µs of the last evaluation."""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = db.execute('select s.kernel_name, d.start, d.end, d.grid_size_y from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()
    x3 = [(e - s) / 1e3 for n, s, e, gy in rows if 'gemm_x3' in n][-73:]      # one evaluation: (dilated, res) x 35, dilated, skip, f0
    dil = sorted(x3[0:71:2])[18]
    res = sorted(x3[1:70:2])[17]
    print('%-40s dilated %6.1f  res %6.1f  skip %7.1f  f0 %6.1f   -> per evaluation %.2f ms' % (
        path[-40:], dil, res, x3[71], x3[72], sum(x3) / 1e3))
