#!/bin/bash
# Executed-MFMA counters of the chain ("executed" FLOPs next to the algorithmic count), one rocprofv3 pass per
# counter group, --kernel-trace only.  usage (GPU box, repo root): tools/pmc_mfma.sh TAG
TAG=${1:-r01}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  mkdir -p $R/gpurun_out/mfma_$TAG/$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/mfma_$TAG/$C -o pmc -- python $R/tools/pmc_driver.py > $R/gpurun_out/mfma_$TAG/$C/driver.out 2> $R/gpurun_out/mfma_$TAG/$C/driver.err < /dev/null
done
cd $R
python - <<'PY' $TAG
import csv, glob, json, os, sys
from collections import defaultdict
root = 'gpurun_out/mfma_' + sys.argv[1]
out = {}
for c in ('SQ_INSTS_VALU_MFMA_MOPS_F64', 'SQ_INSTS_VALU_MFMA_MOPS_I8', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES'):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, c, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != c:
                continue
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            tot[name] += float(row['Counter_Value']); cnt[name] += 1
    for k in tot:
        if k.startswith('ds::k_jet_gemm') or k.startswith('ds::i8::') or k.startswith('ds::k_layer') or k.startswith('ds::k_shared') or k.startswith('ds::k_det_trace') or k.startswith('ds::k_two'):
            out.setdefault(k, {})[c + '_per_launch'] = tot[k] / cnt[k]
            out[k]['launches'] = cnt[k]
print(json.dumps(out, indent=1))
PY
find gpurun_out/mfma_$TAG -name "*kernel_trace.csv" -delete
