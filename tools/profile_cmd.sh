#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script: tools/profile_cmd.sh TAG script.py [args]
TAG=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o $TAG -- python $R/"$@" > $R/gpurun_out/$TAG/out.txt 2> $R/gpurun_out/$TAG/err.txt < /dev/null
cd $R
find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1 | xargs -r head -16 | cut -c1-150
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
