#!/usr/bin/env python3
"""Print the head of a rocprofv3 *kernel_stats.csv (short kernel names): python tools/kstats.py DIR_OR_FILE [rows]"""
import csv, os, re, sys
path = sys.argv[1]
if os.path.isdir(path):
    found = [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs if f.endswith('kernel_stats.csv')]
    path = sorted(found)[0]
rows = list(csv.DictReader(open(path)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = re.sub(r'\(.*', '', r['Name']).replace('void ds::', '')[:58]
    print(f"{n:58s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.1f} ms  {float(r['Percentage']):5.2f}%")
