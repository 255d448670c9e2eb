#!/bin/bash
# Shader clock during each kernel of the chain: GRBM_GUI_ACTIVE (cycles at the shader clock) / kernel duration.
# One rocprofv3 pass, --kernel-trace only.  usage (GPU box, repo root): tools/pmc_clock.sh TAG
TAG=${1:-r02}
R=$PWD
mkdir -p $R/gpurun_out/clk_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$TAG -o pmc -- python $R/tools/pmc_driver.py > $R/gpurun_out/clk_$TAG/driver.out 2> $R/gpurun_out/clk_$TAG/driver.err < /dev/null
cd $R
python - <<'PY' $TAG
import csv, glob, json, os, sys
from collections import defaultdict
root = 'gpurun_out/clk_' + sys.argv[1]
cyc, dur, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    rows = list(csv.DictReader(open(f)))
    if rows:
        print('columns:', list(rows[0].keys()), file=sys.stderr)
    for row in rows:
        if row.get('Counter_Name') != 'GRBM_GUI_ACTIVE':
            continue
        name = row['Kernel_Name'].split('(')[0].replace('void ', '')
        cyc[name] += float(row['Counter_Value']); cnt[name] += 1
        if 'Start_Timestamp' in row:
            dur[name] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
out = {k: {'launches': cnt[k], 'gui_active_cycles_per_launch': cyc[k] / cnt[k], 'ns_per_launch': dur[k] / cnt[k] if dur[k] else None,
           'clock_ghz': cyc[k] / dur[k] if dur[k] else None} for k in cyc if k.startswith('ds::')}
print(json.dumps(out, indent=1))
PY
find gpurun_out/clk_$TAG -name "*kernel_trace.csv" -delete
