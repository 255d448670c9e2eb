#!/bin/bash
# Round-end measurement batch (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC traffic,
# gradient / value-chain profiles.  Results land in gpurun_out/; copy what should be judged into profiles/.
set -x
mkdir -p gpurun_out/final gpurun_out/r01b
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.json
bash tools/profile_bench.sh r01b > gpurun_out/final/prof.txt 2>&1
bash tools/pmc_traffic.sh r01b > gpurun_out/final/pmc.txt 2>&1
bash tools/profile_cmd.sh r01_vjp tools/grad_bench.py bcc_li 4096 vjp > /dev/null 2>&1
bash tools/profile_cmd.sh r01_value tools/value_driver.py > /dev/null 2>&1
python tools/grad_bench.py bcc_li 4096 > gpurun_out/final/grad_bench.txt 2>&1
for s in h2 lih; do python bench.py --system $s --no-cpu-baseline > gpurun_out/final/bench_$s.json 2>> gpurun_out/final/bench.err; done
python bench.py --system graphene --batch 512 --steps 2 --no-cpu-baseline > gpurun_out/final/bench_graphene.json 2>> gpurun_out/final/bench.err
python bench.py --system diamond --dtype f32 --batch 1024 --steps 2 --no-cpu-baseline > gpurun_out/final/bench_diamond.json 2>> gpurun_out/final/bench.err
ls -la gpurun_out/final
