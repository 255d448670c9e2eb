#!/bin/bash
# One box, one call: the PMC traffic pass first, its summary installed as profiles/${TAG}_pmc_traffic.json, then bench.py (which
# reads that file and checks its kernel durations against the in-run ones) and the same command under rocprofv3 --kernel-trace
# --stats.  The shader clock differs by a few per cent between boxes (2.20 .. 2.32 GHz seen), which alone can trip bench.py's
# 5 % agreement rule when the profile comes from another box.   usage: tools/bench_with_traffic.sh TAG   (GPU box, repo root)
TAG=${1:-r03}
F=gpurun_out/bwt_$TAG
mkdir -p $F
bash tools/pmc_traffic.sh ${TAG}c > $F/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_${TAG}c/summary.json profiles/${TAG}_pmc_traffic.json
cp gpurun_out/pmc_${TAG}c/summary.json $F/${TAG}_pmc_traffic.json
python bench.py > $F/bench.json 2> $F/bench.err
bash tools/profile_bench.sh ${TAG}_bench > $F/prof.txt 2>&1
tail -c 400 $F/bench.json
