"""Per-kernel mean of every counter in a rocprofv3 counter_collection.csv:  python tools/pmc_summary.py <csv> [substr]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
with open(sys.argv[1], newline='') as f:
    for r in csv.DictReader(f):
        name = r['Kernel_Name']
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        short = name.replace('(anonymous namespace)::', '').replace('void ', '', 1).split('(')[0]
        grid = r.get('Grid_Size', '')
        acc[(short, grid, r['Counter_Name'])].append(float(r['Counter_Value']))
for (short, grid, counter), vals in sorted(acc.items()):
    print('%-34s grid %-8s %-28s n=%-3d mean %.1f' % (short, grid, counter, len(vals), sum(vals) / len(vals)))
