#!/usr/bin/env python3
"""Average the counters of the last dispatch of every kernel matching a substring over rocprofv3 counter_collection CSVs."""
import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else 'conv'
vals = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
    if not rows: continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id']) == last:
            vals[r['Counter_Name']] = vals.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k in sorted(vals): print(f'{k:45s} {vals[k]:18.1f}')
