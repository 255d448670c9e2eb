#!/bin/bash
# SQ counters of ONE kernel (name substring) under an arbitrary python driver, one rocprofv3 pass per counter (--kernel-trace only).
# usage (GPU box, repo root): tools/pmc_kernel.sh TAG KERNEL_SUBSTRING driver.py [args] -- prints counter -> average per launch
TAG=$1; KN=$2; shift 2
R=$PWD
cd /tmp && export TMPDIR=/tmp
CNT="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
for C in $CNT; do
  mkdir -p $R/gpurun_out/pmck_$TAG/$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmck_$TAG/$C -o pmc -- python $R/"$@" > $R/gpurun_out/pmck_$TAG/$C/driver.out 2> $R/gpurun_out/pmck_$TAG/$C/driver.err < /dev/null
done
cd $R
python - "$TAG" "$KN" $CNT <<'PY'
import csv, glob, os, sys
tag, kn, cnts = sys.argv[1], sys.argv[2], sys.argv[3:]
for c in cnts:
    tot = n = 0
    for f in glob.glob(os.path.join('gpurun_out/pmck_' + tag, c, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == c and kn in row['Kernel_Name']:
                tot += float(row['Counter_Value']); n += 1
    print(f'{c:32s} {tot / n if n else float("nan"):16.1f}  ({n} launches)')
PY
find gpurun_out/pmck_$TAG -name "*kernel_trace.csv" -delete
