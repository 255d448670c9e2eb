#!/bin/bash
# HBM traffic counters, one rocprofv3 pass per counter (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2 -- cannot share a pass).
# Counter passes use --kernel-trace only (no other trace domains).  usage: tools/pmc_traffic.sh TAG
TAG=${1:-r01}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/pmc_$TAG/$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/$C -o pmc -- python $R/tools/pmc_driver.py > $R/gpurun_out/pmc_$TAG/$C/driver.out 2> $R/gpurun_out/pmc_$TAG/$C/driver.err < /dev/null
done
cd $R
python tools/pmc_summarize.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.json
cat gpurun_out/pmc_$TAG/summary.json
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
