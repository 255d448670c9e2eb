#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh: per kernel, counter total
per launch, calibrated against the copy kernel of known traffic (MI355X_MICROARCH.md: on gfx950 the
raw FETCH_SIZE of a coalesced stream is not the byte count; calibrate in your own access pattern)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
CALIB_BYTES = 8 * (1 << 28)
out = {'calib_bytes_each_way': CALIB_BYTES, 'walkers_per_launch': int(os.environ.get('PMC_WALKERS', 4096)), 'system': os.environ.get('PMC_SYSTEM', 'bcc_li'),
       'dtype': os.environ.get('PMC_DTYPE', 'f64'), 'kernels': {}}
dur = defaultdict(list)          # kernel durations inside the profile passes (ms): bench.py only trusts the counters when they agree with its own
raw = {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(root, counter, '**', '*counter_collection.csv'), recursive=True)
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            tot[name] += float(row['Counter_Value'])
            cnt[name] += 1
            if row.get('Start_Timestamp') and row.get('End_Timestamp'):
                dur[name].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6)
    raw[counter] = {k: (tot[k], cnt[k]) for k in tot}
    calib = [v for k, v in raw[counter].items() if 'k_calib_copy' in k]
    out[counter + '_units_per_byte'] = (calib[0][0] / calib[0][1]) / CALIB_BYTES if calib else None
for name in sorted(set(raw['FETCH_SIZE']) | set(raw['WRITE_SIZE'])):
    if not name.startswith('ds::'):
        continue
    e = {}
    for counter, key in (('FETCH_SIZE', 'read'), ('WRITE_SIZE', 'write')):
        if name in raw[counter] and out[counter + '_units_per_byte']:
            t, c = raw[counter][name]
            e[key + '_bytes_per_launch'] = t / c / out[counter + '_units_per_byte']
            e['launches'] = c
    if dur.get(name):
        e['avg_launch_ms'] = sorted(dur[name])[len(dur[name]) // 2]          # median over the profile run's launches
    out['kernels'][name] = e
print(json.dumps(out, indent=1))
