#!/usr/bin/env python3
"""Development: summarise a rocprofv3 --pmc counter CSV per layer-kernel dispatch (mean by dilation position)."""
import csv, glob, sys
from collections import defaultdict
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(path)))
    per = defaultdict(list)
    for r in rows:
        if 'wn_layer' in r['Kernel_Name'] or 'wn_final' in r['Kernel_Name']:
            per[(r['Kernel_Name'][:48], r['Counter_Name'])].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for (k, c), v in per.items():
        v.sort()
        vals = [x for _, x in v]
        print(k, c, 'n=%d' % len(vals), 'first 12:', ' '.join('%.0f' % x for x in vals[:12]), ' last 6:', ' '.join('%.0f' % x for x in vals[-6:]))
