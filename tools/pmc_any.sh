#!/bin/bash
# Generic PMC pass (GPU box, repo root): one rocprofv3 --kernel-trace --pmc run per counter GROUP (quoted, space separated),
# per-kernel averages printed for the kernels matching $PMC_KERNELS (default: the layer kernels).
# usage: tools/pmc_any.sh TAG "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS" ...
TAG=$1; shift
R=$PWD
DRIVER=${PMC_DRIVER:-tools/pmc_driver.py}
cd /tmp && export TMPDIR=/tmp
i=0
for G in "$@"; do
  i=$((i+1))
  D=$R/gpurun_out/pmc_$TAG/g$i
  mkdir -p $D
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o pmc -- python $R/$DRIVER > $D/driver.out 2> $D/driver.err < /dev/null
done
cd $R
python - "$TAG" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root = 'gpurun_out/pmc_' + sys.argv[1]
pat = os.environ.get('PMC_KERNELS', 'k_jet_gemm<double, 4, 5, 2>,k_jet_gemm<double, 4, 5, 1>').split(',')
tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if not any(p in name for p in pat):
            continue
        tot[name][row['Counter_Name']] += float(row['Counter_Value']); cnt[name][row['Counter_Name']] += 1
out = {k: {c: tot[k][c] / cnt[k][c] for c in sorted(tot[k])} | {'launches': max(cnt[k].values())} for k in tot}
print(json.dumps(out, indent=1))
PY
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
