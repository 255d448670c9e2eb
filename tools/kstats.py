"""Prints a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py FILE [calls_per_pass]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    nm = r['Name'].split('(')[0].replace('void ds::', '')
    print(f"{nm:44s} calls {r['Calls']:>5s}  {int(r['TotalDurationNs']) / n / 1e6:8.3f} ms/pass  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
print(f'sum {tot / n / 1e6:.3f} ms/pass')
