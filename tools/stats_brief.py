"""Print rocprofv3 kernel_stats.csv rows compactly: calls, avg us, total ms, short name."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2:] 
for r in rows[:int(1e9)]:
    name = r['Name']
    if pat and not any(p in name for p in pat):
        continue
    short = name.replace('void (anonymous namespace)::', '')[:70]
    print('%6s %10.1f us %9.2f ms  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, short))
