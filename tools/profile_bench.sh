#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench.py command; copies the CSV summary to gpurun_out/.
# usage (on the GPU box, from the repo root): tools/profile_bench.sh TAG
TAG=${1:-r01}
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err < /dev/null
cd $R
find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1 | xargs -r head -14
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
du -sh gpurun_out/$TAG
